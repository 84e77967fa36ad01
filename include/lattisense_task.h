/*
 * lattisense_task.h — TASK LAYER of the MI355X executor: the drop-in for the GPU entry points of
 * mega_ag_runners/wrapper.h:67-85, with the data ABI of abi/c_types.h:26-60 and mega_ag_runners/c_argument.h:26-46.
 *
 * A caller (lattisense::FheTaskGpu cxx_sdk_v2/cxx_fhe_task_gpu.cpp:33-103, plug-in/SEAL/acc/gpu_runner.cpp:12-52,
 * plug-in/lattigo/acc/gpu_runner.go:88-128) does:
 *     h = create_fhe_gpu_task(project_dir);                       // loads <dir>/mega_ag.json
 *     bind_gpu_task_abi_bridge_executors(h, &export_fn, &import_fn);   // caller-side Handle <-> C-struct converters
 *     run_fhe_gpu_task(h, in_args, n_in, out_args, n_out, progress_cb, user, gpu_device);
 *     release_fhe_gpu_task(h);
 * The two executor pointers are `ExecutorFunc*` (std::function, see lattisense_task.hpp); they are copied during bind.
 *
 * Differences from the reference, all deliberate:
 *  - run_fhe_gpu_task never lets a C++ exception cross the C frame: it returns non-zero and lsa_last_error() holds the
 *    message (the reference throws through extern "C"; a cgo caller cannot catch that).
 *  - the device context (NTT tables, conversion constants) is built once per task and device, not on every run
 *    (reference: gpu_wrapper.cu:155).
 *  - identical operators of one topological level are executed as ONE batched launch sequence.
 */
#ifndef LATTISENSE_TASK_H
#define LATTISENSE_TASK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data ABI: same layout as abi/c_types.h:26-60 (LP64: CComponent 16 B, CPolynomial 16 B, CCiphertext 16 B) ---- */
typedef struct {
    int n;          /* ring degree */
    uint64_t* data; /* n residues of one RNS limb */
} CComponent;

typedef struct {
    int n_component; /* RNS limbs */
    CComponent* components;
} CPolynomial;

typedef struct {
    int level;
    CPolynomial poly;
} CPlaintext;

typedef struct {
    int level;
    int degree; /* polys = degree + 1 */
    CPolynomial* polys;
} CCiphertext;

typedef CCiphertext CPublicKey; /* one gadget digit: degree-1 pair over Q[0..level] u P */

typedef struct {
    int n_public_key; /* beta digits */
    CPublicKey* public_keys;
} CKeySwitchKey;

typedef CKeySwitchKey CRelinKey;

typedef struct {
    int n_key_switch_key;
    uint64_t* galois_elements;
    CKeySwitchKey* key_switch_keys;
} CGaloisKey;

/* ---- argument ABI: same layout and enumerator values as mega_ag_runners/c_argument.h:26-46 ---- */
typedef enum { TYPE_PLAINTEXT, TYPE_CIPHERTEXT, TYPE_RELIN_KEY, TYPE_GALOIS_KEY, TYPE_SWITCH_KEY, TYPE_CUSTOM } DataType;
typedef enum { ALGO_BFV, ALGO_CKKS } Algo;

typedef struct {
    const char* id;
    DataType type;
    void* data; /* void*[size]: opaque caller handles, interpreted only by the caller's export/import executors */
    int level;
    int size;
} CArgument;

/* ---- task entry points: wrapper.h:31-42 (callback), :67-85 (GPU task functions) ---- */
typedef struct fhe_task_handle_st* fhe_task_handle;
typedef void (*progress_callback_t)(int completed, int total, void* user_data);

fhe_task_handle create_fhe_gpu_task(const char* project_path); /* NULL on failure (see lsa_last_error) */
void release_fhe_gpu_task(fhe_task_handle handle);
void bind_gpu_task_abi_bridge_executors(fhe_task_handle handle, void* abi_export_executor, void* abi_import_executor);
void bind_gpu_task_custom_executors(fhe_task_handle handle, const char** custom_types, void** executors,
                                    uint64_t n_executors);
int run_fhe_gpu_task(fhe_task_handle handle, CArgument* input_args, uint64_t n_in_args, CArgument* output_args,
                     uint64_t n_out_args, progress_callback_t progress_cb, void* user_data, int gpu_device);

/* ---- C-struct allocation helpers: the role of abi/c_structs.c:23-98 (alloc_* / free_*), used by the STORE executor
 * for the structs it hands to the caller's import executor. Prefixed so both libraries can live in one process. */
void lsa_alloc_component(CComponent* c, int n);
void lsa_alloc_polynomial(CPolynomial* p, int n_component, int n);
void lsa_alloc_ciphertext(CCiphertext* ct, int degree, int level, int n);
void lsa_free_polynomial(CPolynomial* p);
void lsa_free_ciphertext(CCiphertext* ct);

/* ---- native front-end: the analogue of plug-in/SEAL/acc (argument.h, abi_bridge_executors.h, gpu_runner.cpp) for
 * callers whose objects are plain host limb buffers.  CArgument.data[i] points at one of these handles. ---- */
typedef struct {
    int level;
    int degree;
    int n;
    uint64_t* data; /* [degree+1][level+1][n] */
} lsa_host_ciphertext;

typedef struct {
    int level;
    int n;
    uint64_t* data; /* [level+1][n] */
} lsa_host_plaintext;

typedef struct {
    int level; /* key level: level+1 Q-limbs + np special limbs per polynomial */
    int n_special;
    int n;
    uint64_t* data; /* compact [beta][2][level+1+n_special][n], NTT domain, non-Montgomery */
} lsa_host_kskey;

typedef struct {
    int n_keys;
    uint64_t* galois_elements;
    lsa_host_kskey* keys;
} lsa_host_galois_key;

/* binds the native front-end's export/import executors to a task (calls bind_gpu_task_abi_bridge_executors) */
int lsa_frontend_bind(fhe_task_handle handle);

/* frees every pooled device / pinned buffer of the task (all devices); the next run allocates afresh.  Pools are keyed by
 * (device, lane): one task handle may be run on any gpu_device, one run at a time (reference: README.md:195-202) */
int lsa_task_trim_pools(fhe_task_handle handle);

/* Multi-device execution behind run_fhe_gpu_task (the reference's only multi-GPU mode is one run per device,
 * README.md:195-202, which exports and uploads every key once per device and run).  With a device list set, a run whose graph
 * splits into independent subgraphs deals them out to the listed devices -- one pair of execution lanes per entry; an index may
 * repeat (two logical shards on one device) -- exports and uploads every evaluation key ONCE on the first device and copies it
 * device-to-device to the others.  Results are imported as always.  n_devices = 0 clears the list.  With a list set (or
 * gpu_device = -1: every visible device) the gpu_device argument of run_fhe_gpu_task is ignored. */
int lsa_task_set_devices(fhe_task_handle handle, const int* device_ids, int n_devices);
int lsa_task_last_run_shards(fhe_task_handle handle, int* n_shards, int* n_chunks, int* key_peer_copies);

/* Zero-copy ingestion for callers of the native front-end: limb buffers inside a range registered here (pinned in
 * place) are DMA'd from / into directly -- an input ciphertext or plaintext whose limbs are one contiguous block is copied to
 * the device from where it lies, a result is copied straight into the pre-allocated output ciphertext -- instead of going
 * through the pinned staging slabs.  The caller owns the lifetime: keep the range allocated until lsa_host_unregister.
 * Everything else (foreign executors' per-limb structs, unregistered buffers) takes the staged path. */
int lsa_host_register(void* ptr, size_t bytes);
int lsa_host_unregister(void* ptr);
/* pinned memory allocated for the caller (the fast path: full PCIe rate; pin-in-place registration measured slower than staging) */
int lsa_host_alloc(size_t bytes, void** out);
int lsa_host_free(void* ptr);
int lsa_task_last_run_direct(fhe_task_handle handle, int* loads, int* stores);

/* Evaluation keys stay resident on the device(s) across run() calls: a run whose export executor yields the same caller
 * handle and the same fingerprint (shape + three sampled words of every limb of the exported C struct) for a key datum reuses
 * the converted device copy instead of staging, uploading and converting it again (the reference re-exports and re-uploads every
 * key on every run, cxx_sdk_v2/cxx_argument.h:178-260).  A regenerated key or another key object is detected and replaces the
 * copy; a caller that rewrites a key in place must call lsa_task_drop_keys (frees the device copies; the next run uploads).
 * LSA_NO_KEY_CACHE=1 in the environment restores upload-per-run. */
int lsa_task_drop_keys(fhe_task_handle handle);
int lsa_task_last_run_keys(fhe_task_handle handle, int* uploaded, int* reused);

/* introspection used by tests and INTEGRATION.md examples */
int lsa_task_counts(fhe_task_handle handle, int* n_data, int* n_compute, int* n_inputs, int* n_outputs);
/* number of batched launch groups vs. compute nodes in the last run (how much graph-level batching happened) */
int lsa_task_last_run_stats(fhe_task_handle handle, int* gpu_nodes, int* gpu_batches, double* run_ms);

#ifdef __cplusplus
}
#endif
#endif
