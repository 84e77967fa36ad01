/*
 * lattisense_amd.h — C-ABI of the MI355X executor for LattiSense's RNS polynomial-arithmetic hot path.
 *
 * Two layers, both plain C (pointers + sizes, no C++/torch types):
 *
 *  (1) OPERATOR LAYER (lsa_*): what the reference obtains from the absent HEonGPU library through
 *      heongpu::HEContext / HEArithmeticOperator (call sites mega_ag_runners/gpu/gpu_wrapper.cu:53-138 and
 *      mega_ag_runners/gpu/mega_ag_executors_gpu.cu:71-426).  Ciphertexts are device-resident u64 limbs laid out
 *      [poly][limb][N] exactly like the reference's device buffers (gpu_abi_bridge_executors.h:76-80, :185-189);
 *      every call takes a `batch` of independent ciphertexts (batch_stride u64 elements apart) and a HIP stream.
 *
 *  (2) TASK LAYER (create/bind/run/release_fhe_gpu_task): declared in lattisense_task.h, the drop-in for
 *      mega_ag_runners/wrapper.h:67-85.
 *
 * All functions return 0 on success and a non-zero code on failure; lsa_last_error() returns the message of the
 * last failure on the calling thread (the reference throws std::runtime_error through extern "C",
 * gpu_abi_bridge_executors.h:51-55 — unusable from cgo/ctypes callers, SURVEY §8b "Errors").
 * There is no CPU fallback: without a HIP device every compute entry point fails with LSA_ERR_NO_DEVICE.
 */
#ifndef LATTISENSE_AMD_H
#define LATTISENSE_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsa_context_st* lsa_context;
typedef struct lsa_key_st* lsa_key; /* one key-switch key (relin key, or the key of one Galois element) */

enum { LSA_OK = 0, LSA_ERR_ARG = 1, LSA_ERR_NO_DEVICE = 2, LSA_ERR_HIP = 3, LSA_ERR_INTERNAL = 4 };
enum { LSA_ALGO_BFV = 0, LSA_ALGO_CKKS = 1 }; /* same values as Algo in mega_ag_runners/c_argument.h:35-38 */

const char* lsa_last_error(void);
const char* lsa_version(void);
/* every compile-time switch (LSA_* macro) the library was built with; "" for the product build (csrc/build_flags.h) */
const char* lsa_build_flags(void);

/* ---- context: replaces init_gpu_context (gpu_wrapper.cu:53-138).  q = Q chain (max_level+1 primes),
 * p = special primes, t = BFV plaintext modulus (0 for CKKS).  Tables are built once and cached. */
int lsa_context_create(int algo, int n, const uint64_t* q, int nq, const uint64_t* p, int np, uint64_t t,
                       int device, lsa_context* out);
int lsa_context_destroy(lsa_context ctx);
/* all moduli in context order: Q chain, P, then the BFV auxiliary basis the context generated */
int lsa_context_moduli(lsa_context ctx, uint64_t* out, int capacity, int* count);

/* ---- device memory / streams (thin wrappers so a caller needs no HIP bindings; torch pointers work too) */
int lsa_malloc(lsa_context ctx, void** dptr, size_t bytes);
int lsa_free(lsa_context ctx, void* dptr);
int lsa_memcpy_h2d(lsa_context ctx, void* dst, const void* src, size_t bytes, void* stream);
int lsa_memcpy_d2h(lsa_context ctx, void* dst, const void* src, size_t bytes, void* stream);
int lsa_memcpy_d2d(lsa_context ctx, void* dst, const void* src, size_t bytes, void* stream);
int lsa_stream_create(lsa_context ctx, void** stream);
int lsa_stream_destroy(lsa_context ctx, void* stream);
int lsa_stream_synchronize(lsa_context ctx, void* stream);
/* HIP-event timing on `stream` (bench.py roofline leg): returns elapsed ms between two recorded events */
int lsa_event_create(lsa_context ctx, void** ev);
int lsa_event_record(lsa_context ctx, void* ev, void* stream);
int lsa_event_elapsed_ms(lsa_context ctx, void* ev_start, void* ev_stop, float* ms);
int lsa_event_destroy(lsa_context ctx, void* ev);

/* ---- evaluation keys.  `compact` is the ABI order of plug-in/lattigo/acc/c_struct_import_export.go:41-135:
 * [beta][2][key_level+1+np][N], beta = ceil((key_level+1)/np), NTT domain, non-Montgomery (GPU_MFORM_BITS = 0,
 * cxx_sdk_v2/cxx_fhe_task_gpu.cpp:30).  Replaces export_relin_key / export_galois_key / export_switching_key
 * (gpu_abi_bridge_executors.h:86-176). */
int lsa_key_upload(lsa_context ctx, const uint64_t* compact_host, int key_level, void* stream, lsa_key* out);
/* adopt a device buffer already holding the compact key (e.g. after an RCCL broadcast); converted in place */
int lsa_key_adopt_device(lsa_context ctx, uint64_t* compact_dev, int key_level, void* stream, lsa_key* out);
int lsa_key_destroy(lsa_context ctx, lsa_key key);
size_t lsa_key_bytes(lsa_context ctx, int key_level);

/* ---- K1/K2: batched negacyclic NTT / INTT, in place.  data = [batch][rows][N]; row r uses modulus index
 * mod_of[r % period] in context order (0xFF = leave the row untouched). */
int lsa_ntt(lsa_context ctx, uint64_t* data, int batch, long long batch_stride, int rows, const int* mod_of,
            int period, int inverse, void* stream);

/* ---- K3: limb-wise add / sub / negate over `polys` polynomials of level+1 limbs (ct+ct, ct-ct, -ct).
 * op: 0 add, 1 sub, 2 neg (b ignored).  Replaces HEArithmeticOperator::add/sub/negate (executors_gpu.cu:79-173). */
int lsa_poly_addsub(lsa_context ctx, int op, int level, int polys, const uint64_t* a, const uint64_t* b,
                    uint64_t* out, int batch, long long stride_a, long long stride_b, long long stride_out,
                    void* stream);

/* ---- CKKS (NTT-domain ciphertexts) ---------------------------------------------------------------------- */
/* multiply (executors_gpu.cu:185,223): a,b = [2][L][N] -> d3 = [3][L][N] */
int lsa_ckks_mult(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, uint64_t* d3, int batch,
                  long long stride_a, long long stride_b, long long stride_d, void* stream);
/* relinearize (executors_gpu.cu:236): d3 -> out [2][L][N] */
int lsa_ckks_relin(lsa_context ctx, int level, const uint64_t* d3, lsa_key rlk, uint64_t* out, int batch,
                   long long stride_d, long long stride_out, void* stream);
/* rescale (executors_gpu.cu:246): in [polys][L][N] -> out [polys][L-1][N] */
int lsa_ckks_rescale(lsa_context ctx, int level, int polys, const uint64_t* in, uint64_t* out, int batch,
                     long long stride_in, long long stride_out, void* stream);
/* rotate_rows / conjugate (executors_gpu.cu:275,289): Galois element g, key of that element */
int lsa_ckks_rotate(lsa_context ctx, int level, const uint64_t* in, uint64_t galois_element, lsa_key glk,
                    uint64_t* out, int batch, long long stride_in, long long stride_out, void* stream);
/* mod_drop (executors_gpu.cu:257): keep the first L-1 limbs of each polynomial */
int lsa_drop_level(lsa_context ctx, int level, int polys, const uint64_t* in, uint64_t* out, int batch,
                   long long stride_in, long long stride_out, void* stream);
/* fused HMult + relinearize + rescale: the BASELINE.json headline operator. out = [2][L-1][N] */
int lsa_ckks_mult_relin_rescale(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, lsa_key rlk,
                                uint64_t* out, int batch, long long stride_a, long long stride_b,
                                long long stride_out, void* stream);

/* ---- BFV (coefficient-domain ciphertexts) ---------------------------------------------------------------- */
int lsa_bfv_mult(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, uint64_t* d3, int batch,
                 long long stride_a, long long stride_b, long long stride_d, void* stream);
int lsa_bfv_relin(lsa_context ctx, int level, const uint64_t* d3, lsa_key rlk, uint64_t* out, int batch,
                  long long stride_d, long long stride_out, void* stream);
int lsa_bfv_rotate(lsa_context ctx, int level, const uint64_t* in, uint64_t galois_element, lsa_key glk,
                   uint64_t* out, int batch, long long stride_in, long long stride_out, void* stream);
int lsa_bfv_rescale(lsa_context ctx, int level, int polys, const uint64_t* in, uint64_t* out, int batch,
                    long long stride_in, long long stride_out, void* stream);
int lsa_bfv_mult_relin(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, lsa_key rlk,
                       uint64_t* out, int batch, long long stride_a, long long stride_b, long long stride_out,
                       void* stream);

/* ---- tuning / introspection */
/* ciphertexts processed per kernel wave inside the fused operators (0 = automatic) */
int lsa_set_tile_batch(lsa_context ctx, int tile_batch);
/* NTT butterfly engine for limbs with q < 2^47: 1 (default) = exact FP64-FMA butterflies, 0 = integer Montgomery for
 * every limb.  Both produce identical residues; the switch exists for A/B measurement and parity tests. */
int lsa_set_fp64_ntt(lsa_context ctx, int enable);
/* 1: alternate tiles of a batched operator run on the caller's stream and on an internal auxiliary stream (fork/join
 * with events inside the call), so two independent tiles overlap (+5 % measured); 0 (default): everything on the
 * caller's stream, which keeps per-kernel timings attributable. */
int lsa_set_dual_stream(lsa_context ctx, int enable);
/* 1 (default): the ModDown and rescale element-wise tails run inside the NTT kernel's load/store phases; 0: separate
 * kernels (A/B measurement; identical results). */
int lsa_set_fuse_tails(lsa_context ctx, int enable);
/* Two-pass NTTs (N > 2^12) run both passes over a chunk of at most `mib` MiB of limbs before moving on, so that the
 * second pass is served by the 256 MiB Infinity Cache (0 = one launch per pass over the whole batch). */
int lsa_set_ntt_chunk_mib(lsa_context ctx, int mib);
/* Rotations of the same ciphertexts by several Galois elements with ONE decomposition of the input ("hoisting"):
 * outs[i] = rotate(in, galois_elements[i]), each bit-identical to lsa_ckks_rotate's result. */
int lsa_ckks_rotate_many(lsa_context ctx, int level, const uint64_t* in, int n_rot, const uint64_t* galois_elements,
                         const lsa_key* glk, uint64_t* const* outs, int batch, long long sin, long long sout, void* stream);

/* ---- CKKS bootstrapping: the `bootstrap` node of a task graph (reference: mega_ag_executors_gpu.cu:410-426 calls HEonGPU's
 * regular_bootstrapping_v2; configuration gpu_wrapper.cu:86-117 / custom_task.py:383-468).  A plan holds the encoded
 * CoeffsToSlots / SlotsToCoeffs diagonals (depths cts_depth / stc_depth, full-slot encoding), the EvalMod constants
 * (cosine of range k with `double_angle` doublings, message ratio) and the list of Galois elements a run needs -- the
 * rotation set of the reference's planner (frontend/bootstrap_params.py:104-263) plus the conjugation.  in_scale: scale
 * of the level-0 input; out_scale: scale the refreshed ciphertext must have (0: whatever falls out).  The context's chain
 * must have cts_depth + 5 + double_angle + stc_depth levels above the output level.  log_slots: 0 or log2(N)-1 for dense
 * packing, smaller for sparsely packed ciphertexts (SubSum, one EvalMod on the packed real|imaginary halves, repacking
 * SlotsToCoeffs).  The matrices of a sparse plan (lsa_bootstrap_info: sparse = 1) are ordered
 * [n_cts leading CoeffsToSlots ..., P1, P2, SlotsToCoeffs ...], those of a dense plan [n_cts CoeffsToSlots, SlotsToCoeffs ...].
 * lsa_ckks_bootstrap: in [batch][2][1][N] -> out [batch][2][out_level+1][N]; swk_dts / swk_std (both or neither) are the
 * sparse-secret encapsulation keys at level 0 / top level (custom_task.py:1989-1996). */
typedef struct lsa_bootstrap_st* lsa_bootstrap;
int lsa_bootstrap_create(lsa_context ctx, int cts_depth, int stc_depth, int k, int double_angle, double message_ratio,
                         double in_scale, double out_scale, int log_slots, void* stream, lsa_bootstrap* out);
/* the same with the EvalMod polynomial degrees the reference forwards (gpu_wrapper.cu:100-103): sine_deg 1..63 (the cosine
 * interpolant takes ceil(log2(sine_deg+1)) levels), arcsine_deg 0 (none) or odd <= 15 (ceil(log2(arcsine_deg+1)) more levels) */
int lsa_bootstrap_create_ex(lsa_context ctx, int cts_depth, int stc_depth, int k, int double_angle, double message_ratio,
                            double in_scale, double out_scale, int log_slots, int sine_deg, int arcsine_deg, void* stream,
                            lsa_bootstrap* out);
int lsa_bootstrap_evalmod_constants(lsa_bootstrap b, int* n_cheb, double* cheb, int* n_asin, double* asin_coef);
void lsa_bootstrap_destroy(lsa_bootstrap b);
int lsa_bootstrap_info(lsa_bootstrap b, int* out_level, double* out_scale, int* n_galois, int* n_matrices, int* n_cts,
                       int* sparse);
int lsa_bootstrap_galois_elements(lsa_bootstrap b, uint64_t* out, int capacity);
/* the plan's floating-point constants, exported so that a checker can replay the program with the same integers */
int lsa_bootstrap_chebyshev(lsa_bootstrap b, double* out32);
int lsa_bootstrap_matrix_info(lsa_bootstrap b, int index, int* level, int* n1, int* n_diagonals, int* diagonals, int capacity);
/* lsa_bootstrap_plaintext writes the (level + 1) * N words at q_0..q_level.  A baby-step / giant-step matrix of a double-hoisting
 * plan (the default; LSA_BT_DOUBLE_HOIST=0 at plan creation turns it off) also carries the residues at the k special primes, because
 * its inner sums are formed over Q u P before ONE division by P per giant step (Lattigo v4 ckks/linear_transform.go,
 * MultiplyByDiagMatrixBSGS): lsa_bootstrap_plaintext_rows reports level + 1 or level + 1 + k, lsa_bootstrap_plaintext_ext writes
 * all rows * N words (capacity checked). */
int lsa_bootstrap_plaintext(lsa_bootstrap b, int matrix, int diag_pos, uint64_t* host_out);
int lsa_bootstrap_plaintext_rows(lsa_bootstrap b, int matrix, int* rows);
int lsa_bootstrap_plaintext_ext(lsa_bootstrap b, int matrix, int diag_pos, uint64_t* host_out, long long capacity_words);
int lsa_ckks_bootstrap(lsa_context ctx, lsa_bootstrap b, const uint64_t* in, uint64_t* out, int batch, long long sin, long long sout,
                       lsa_key rlk, int n_glk, const uint64_t* glk_elements, const lsa_key* glk, lsa_key swk_dts, lsa_key swk_std,
                       void* stream);

/* diagnostic builds only (-DLSA_NTT_DIAG_STAMPS): device buffer of 8192*8 u64 receiving per-workgroup phase time stamps
 * of every following NTT launch; NULL turns it off.  Ignored by the normal build. */
int lsa_debug_set_ntt_stamps(lsa_context ctx, void* device_buffer);
/* Sampled HIP-event timing of the library's own kernel launches, recorded on the stream they are launched on (every
 * `stride`-th launch of each kind gets an event pair).  kind: 0 NTT pass, 1 base conversion, 2 key-switch MAC,
 * 3 tensor, 4 other element-wise.  total_bytes = ALGORITHMIC bytes of the sampled launches. */
int lsa_profile_begin(lsa_context ctx, int stride);
int lsa_profile_end(lsa_context ctx);
int lsa_profile_read(lsa_context ctx, int kind, double* total_ms, double* total_bytes, long long* sampled,
                     long long* launched);
int lsa_profile_read_primary(lsa_context ctx, int kind, double* total_bytes_primary);
/* micro-benchmark kernels used by bench.py / DESIGN.md to report the integer-multiply and copy ceilings */
int lsa_probe_copy(lsa_context ctx, uint64_t* dst, const uint64_t* src, size_t n_u64, void* stream);
int lsa_probe_mulhi(lsa_context ctx, uint64_t* buf, size_t n_u64, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif
