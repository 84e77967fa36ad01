// task_runtime.hip — the GPU task runner: drop-in for mega_ag_runners/gpu/gpu_wrapper.cu (FheGpuTask, _run_mega_ag_impl,
// the extern "C" entry points :481-530), mega_ag_executors_gpu.cu (bind_gpu_executor and the per-op executors) and
// gpu_abi_bridge_executors.h (LOAD_TO_BACKEND / STORE_FROM_BACKEND), re-designed for MI355X:
//
//   * LEVEL-BATCHED scheduling instead of a 1 ms-polling dispatcher + 2 streams: nodes of one topological level that
//     perform the same operator on the same shapes (the frontend emits n_op identical disjoint subgraphs, e.g.
//     examples/benchmark_gpu/benchmark_gpu.py:29-33) are executed as ONE batched launch sequence of the operator layer.
//     A 1024-op task becomes a handful of large launches that fill 256 CUs, not 7k tiny ones.
//   * device data of a batch lives in one slab with a fixed stride, so operands of the next level are usually already
//     contiguous; otherwise they are gathered with device-to-device copies.
//   * H2D/D2H go through one pinned staging slab per level and ONE hipMemcpyAsync per group (the reference issues one
//     pageable copy per limb, gpu_abi_bridge_executors.h:60-191).
//   * the device context (tables) is created once per (task, device) and reused across run() calls.
//   * CPU-side nodes (caller's export/import executors, custom nodes) of a level run on a small thread pool.
//   * every failure is turned into a non-zero return code + lsa_last_error(); nothing throws across extern "C".
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <optional>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_set>

#include "buf_pool.h"
#include "lsa_internal.h"
#include "shard_plan.h"
#include "task_graph.h"

namespace lsa {
void ckks_relin(Context&, int, const u64*, const Key&, u64*, int, long long, long long, hipStream_t);
void ckks_rescale(Context&, int, int, const u64*, u64*, int, long long, long long, hipStream_t);
void ckks_rotate(Context&, int, const u64*, u64, const Key&, u64*, int, long long, long long, hipStream_t);
void ckks_rotate_many(Context&, int, const u64*, int, const u64*, const Key* const*, u64* const*, int, long long, long long,
                      hipStream_t);
void ckks_mult_relin_rescale(Context&, int, const u64*, const u64*, const Key&, u64*, int, long long, long long, long long,
                             hipStream_t);
void bfv_mult(Context&, int, const u64*, const u64*, u64*, int, long long, long long, long long, hipStream_t);
void bfv_relin(Context&, int, const u64*, const Key&, u64*, int, long long, long long, hipStream_t);
void bfv_rotate(Context&, int, const u64*, u64, const Key&, u64*, int, long long, long long, hipStream_t);
void bfv_rescale(Context&, int, int, const u64*, u64*, int, long long, long long, hipStream_t);
}  // namespace lsa

using namespace lsa;

// ------------------------------------------------------------------------------------------------ C-struct helpers
extern "C" {
void lsa_alloc_component(CComponent* c, int n) {
    c->n = n;
    c->data = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
}
void lsa_alloc_polynomial(CPolynomial* p, int n_component, int n) {
    p->n_component = n_component;
    p->components = (CComponent*)malloc(sizeof(CComponent) * (size_t)n_component);
    for (int i = 0; i < n_component; i++) lsa_alloc_component(&p->components[i], n);
}
void lsa_alloc_ciphertext(CCiphertext* ct, int degree, int level, int n) {
    ct->level = level;
    ct->degree = degree;
    ct->polys = (CPolynomial*)malloc(sizeof(CPolynomial) * (size_t)(degree + 1));
    for (int i = 0; i <= degree; i++) lsa_alloc_polynomial(&ct->polys[i], level + 1, n);
}
void lsa_free_polynomial(CPolynomial* p) {
    if (!p || !p->components) return;
    for (int i = 0; i < p->n_component; i++) free(p->components[i].data);
    free(p->components);
    p->components = nullptr;
}
void lsa_free_ciphertext(CCiphertext* ct) {
    if (!ct || !ct->polys) return;
    for (int i = 0; i <= ct->degree; i++) lsa_free_polynomial(&ct->polys[i]);
    free(ct->polys);
    ct->polys = nullptr;
}
}

namespace {

// ------------------------------------------------------------------------------------------------ caller-pinned host memory
// Zero-copy ingestion: a caller that keeps its limb buffers in memory it has registered with lsa_host_register
// (pinned in place, hipHostRegister) gets its ciphertexts DMA'd straight from / into those buffers -- no gather into a staging
// slab on the way in, no copy out of one on the way back.  The caller owns the lifetime: the range must stay allocated until
// lsa_host_unregister.  Unregistered buffers take the pinned-staging path as before.
struct HostRegistry {
    std::mutex mu;
    std::map<uintptr_t, size_t> ranges;   // base -> bytes
    bool covers(const void* p, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        if (ranges.empty()) return false;
        const uintptr_t a = (uintptr_t)p;
        auto it = ranges.upper_bound(a);
        if (it == ranges.begin()) return false;
        --it;
        return a >= it->first && a + bytes <= it->first + it->second;
    }
};
HostRegistry& host_registry() {
    static HostRegistry r;
    return r;
}

// ------------------------------------------------------------------------------------------------ device data
// Buffers (device or pinned host) recycled across levels and run() calls: hipMalloc/hipFree and pinned allocation cost
// milliseconds and synchronise the device, so a task keeps what it allocated (buf_pool.h: best fit, a cap on what stays
// pooled).  Every device pool belongs to ONE (device, lane) = one in-order stream, so handing a released device buffer to a
// later kernel is ordered after its earlier readers, and a run on another device never sees this device's allocations.
void* hip_buf_alloc(size_t bytes, int device, bool pinned) {
    LSA_HIP(hipSetDevice(device));
    void* p = nullptr;
    if (pinned) LSA_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    else LSA_HIP(hipMalloc(&p, bytes));
    return p;
}
void hip_buf_release(void* p, int device, bool pinned) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    if (pinned) (void)hipHostFree(p);
    else (void)hipFree(p);
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
}
size_t pool_cap_bytes(const char* env, double default_gib) {
    const char* v = getenv(env);
    return (size_t)((v ? atof(v) : default_gib) * 1073741824.0);
}

struct Slab {
    u64* ptr = nullptr;
    size_t words = 0;       // requested
    size_t cap_words = 0;   // what the pool handed out (>= words)
    BufPool* pool;
    Slab(BufPool& p, size_t w) : words(w), pool(&p) { ptr = p.take(w, &cap_words); }
    ~Slab() { pool->give(cap_words, ptr); }
    Slab(const Slab&) = delete;
    Slab& operator=(const Slab&) = delete;
};

// Limb copy into the pinned staging slab with non-temporal stores: the destination is written once and read by the DMA
// engine, so the read-for-ownership traffic of an ordinary memcpy (a third of the gather's memory traffic) is wasted.
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) static void stream_copy_avx2(u64* dst, const u64* src, size_t words) {
    typedef long long v4 __attribute__((vector_size(32)));
    size_t i = 0;
    for (; i + 4 <= words; i += 4) {
        v4 v;
        __builtin_memcpy(&v, src + i, 32);
        __builtin_nontemporal_store(v, reinterpret_cast<v4*>(dst + i));
    }
    for (; i < words; i++) dst[i] = src[i];
}
static void stream_copy(u64* dst, const u64* src, size_t words) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && (reinterpret_cast<uintptr_t>(dst) & 31) == 0) stream_copy_avx2(dst, src, words);
    else memcpy(dst, src, words * sizeof(u64));
}
#else
static void stream_copy(u64* dst, const u64* src, size_t words) { memcpy(dst, src, words * sizeof(u64)); }
#endif

// Parallel loop on a few PERSISTENT host threads (memcpy-bound staging work; the staging loop calls this once per 32 MiB of
// input).  One loop at a time (callers on different shard threads queue on `run_mu_`).  LSA_STAGE_THREADS overrides the count.
// Measured (profiles/r03/t2_staging_and_lane_handback.log): the copies themselves bound the CKKS x64 graph -- 208 MiB per chunk
// staged at ~55 GB/s read + 55 GB/s written on the box's 16-core share while the DMA engine reads the previous 32 MiB --, not the
// thread start-up (this pool against a spawn per call: no change) and not the lane turnaround (handing a lane back before its
// chunk's import: no change either).
class StagePool {
  public:
    static StagePool& get() {
        static StagePool p;
        return p;
    }
    template <typename F> void run(size_t n, F&& fn) {
        if (n == 0) return;
        if (workers_.empty() || n == 1) {
            for (size_t i = 0; i < n; i++) fn(i);
            return;
        }
        std::lock_guard<std::mutex> one(run_mu_);
        std::function<void(size_t)> f = std::ref(fn);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f;
            n_ = n;
            next_.store(0);
            active_ = (int)workers_.size();
            gen_++;
        }
        cv_.notify_all();
        work(f, n);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return active_ == 0; });   // every worker has seen this generation and left work()
        fn_ = nullptr;
    }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }

  private:
    StagePool() {
        const int hw = (int)std::thread::hardware_concurrency();
        int nthreads = std::max(1, std::min(16, hw > 0 ? hw : 1) - 2);
        if (const char* e = std::getenv("LSA_STAGE_THREADS")) nthreads = std::max(1, std::atoi(e));
        nthreads = std::min(nthreads, 32);
        for (int t = 1; t < nthreads; t++) workers_.emplace_back([this] { loop(); });
    }
    void work(const std::function<void(size_t)>& f, size_t n) {
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= n) return;
            f(i);
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(size_t)>* f;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                f = fn_;
                n = n_;
            }
            work(*f, n);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--active_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    int active_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};
template <typename F> void parallel_for(size_t n, F&& fn) { StagePool::get().run(n, fn); }

struct DevDatum {  // a ciphertext or plaintext living in (a slice of) a slab: [polys][level+1][N]
    std::shared_ptr<Slab> slab;
    u64* ptr = nullptr;
    int polys = 0, level = 0;
    bool is_plain = false;
};
struct DevKey {
    std::shared_ptr<Slab> slab;
    Key key;
};
using DatumP = std::shared_ptr<DevDatum>;
using KeyP = std::shared_ptr<DevKey>;

const char* op_name(OperationType op) {
    switch (op) {
        case OperationType::ADD: return "add";
        case OperationType::SUB: return "sub";
        case OperationType::NEGATE: return "neg";
        case OperationType::MULTIPLY: return "mult";
        case OperationType::RELINEARIZE: return "relin";
        case OperationType::RESCALE: return "rescale";
        case OperationType::DROP_LEVEL: return "drop_level";
        case OperationType::ROTATE_COL: return "rotate_col";
        case OperationType::ROTATE_ROW: return "rotate_row";
        case OperationType::MAC_WO_PARTIAL_SUM: return "cmp_sum";
        case OperationType::MAC_W_PARTIAL_SUM: return "cmpac_sum";
        case OperationType::BOOTSTRAP: return "bootstrap";
        case OperationType::FUSED_MULT_RELIN_RESCALE: return "mult+relin+rescale";
        default: return "?";
    }
}

// which (device, lane) the calling host thread is enqueueing on: every shard of a run has its own thread (run()), a handle runs
// one run() at a time, so the pools / temporaries a helper touches follow from the thread it is called on
struct ExecTls {
    int device = 0, lane = 0;
};
thread_local ExecTls tls_exec;

bool is_plain_node(const DatumNode* d) { return d->datum_type == TYPE_PLAINTEXT; }
bool is_ringt_node(const DatumNode* d) { return d->fhe_prop && d->fhe_prop->p && d->fhe_prop->p->is_ringt; }

}  // namespace

// The operator surface of mega_ag_runners/mega_ag_executors.h:53-54: validates that this backend implements the node
// (the reference throws at bind time for unsupported combinations, mega_ag_executors_gpu.cu:212,481,498).  Backend nodes
// are dispatched in batches by FheGpuTask::run_gpu_bucket; this records nothing but the verdict.
void bind_gpu_executor(ComputeNode& node, Algo algorithm) {
    if (!node.fhe_prop) throw std::runtime_error("FHE property not found for compute node");
    const OperationType op = node.op();
    auto unsupported = [&](const std::string& why) {
        throw std::runtime_error(std::string("Unsupported operation type for GPU ") + (algorithm == ALGO_BFV ? "BFV" : "CKKS") +
                                 ": " + op_name(op) + " (" + why + ")");
    };
    for (auto* in : node.input_nodes)
        if (!in->fhe_prop) throw std::runtime_error("FHE property not found for input node " + std::to_string(in->index));
    switch (op) {
        case OperationType::ADD:
        case OperationType::SUB: break;   // ct+-ct, ct+-pt, ct+-ring-t pt
        case OperationType::MULTIPLY:
            if (node.input_nodes.size() == 2 && is_plain_node(node.input_nodes[1]) && !is_ringt_node(node.input_nodes[1]) &&
                algorithm == ALGO_BFV)
                throw std::runtime_error("Multiply with plaintext only supported for CKKS scheme");  // executors_gpu.cu:212
            break;
        case OperationType::NEGATE:
        case OperationType::RELINEARIZE:
        case OperationType::RESCALE:
        case OperationType::ROTATE_ROW:
        case OperationType::FUSED_MULT_RELIN_RESCALE:
            break;
        case OperationType::ROTATE_COL:
            if (!node.fhe_prop->p) throw std::runtime_error("Rotation step not found in FHE property");
            break;
        case OperationType::DROP_LEVEL:
            if (algorithm == ALGO_BFV) throw std::runtime_error("DROP_LEVEL only supported for CKKS scheme");
            break;
        case OperationType::MAC_WO_PARTIAL_SUM:
        case OperationType::MAC_W_PARTIAL_SUM: {
            if (!node.fhe_prop->p) throw std::runtime_error("Sum count not found in FHE property");
            const int n = node.fhe_prop->p->sum_cnt;
            const size_t pt0 = (size_t)n + (op == OperationType::MAC_W_PARTIAL_SUM ? 1 : 0);
            if (node.input_nodes.size() != pt0 + (size_t)n) unsupported("compressed plaintext blocks");
            if (algorithm == ALGO_BFV && !is_ringt_node(node.input_nodes[pt0]))
                throw std::runtime_error("Multiply with plaintext only supported for CKKS scheme");  // executors_gpu.cu:349,405
            break;
        }
        case OperationType::BOOTSTRAP:   // inputs [ct, rlk, glk..., swk_dts, swk_std] (frontend/custom_task.py:1952-2002)
            if (algorithm != ALGO_CKKS) throw std::runtime_error("BOOTSTRAP only supported for CKKS scheme");  // executors_gpu.cu:424
            if (node.input_nodes.size() < 5) unsupported("bootstrap node without its keys");
            break;
        default: unsupported("unknown");
    }
}

struct fhe_task_handle_st {
    // Two execution lanes (stream + context + device-buffer pool each).  A run whose graph splits into independent
    // subgraphs is pipelined over them: while lane A's chunk computes and copies its results back, lane B's chunk is
    // staged and copied in (PCIe is full duplex; the reference's runner overlaps nothing across its 2 streams' copies).
    // A released device buffer only returns to ITS lane's pool, so reuse stays ordered by that lane's in-order stream.
    // Pools are keyed by (device, lane): one task handle may be run on any device, one run() at a time (the reference's
    // multi-GPU mode, README.md:195-202 / gpu_wrapper.cu:148-149); run() calls on one handle are serialised by run_mu.
    LanePools pools{BufAllocator{hip_buf_alloc, hip_buf_release}, pool_cap_bytes("LSA_POOL_MAX_DEV_GIB", 48.0),
                    pool_cap_bytes("LSA_POOL_MAX_PIN_GIB", 16.0)};   // declared first: destroyed last
    std::mutex run_mu;
    std::shared_ptr<Slab> dslab(size_t words) { return std::make_shared<Slab>(pools.device_pool(tls_exec.device, tls_exec.lane), words); }
    std::shared_ptr<Slab> pslab(size_t words) { return std::make_shared<Slab>(pools.pinned_pool(tls_exec.device), words); }
    std::vector<int> devices_;   // lsa_task_set_devices: the shards of a run (a device may repeat); empty = the run's gpu_device alone
    // Evaluation keys stay on the device across run() calls (SURVEY f3 "persistent state"; the reference re-exports and
    // re-uploads them every run, cxx_sdk_v2/cxx_argument.h:178-260): per (device, key datum) the converted key is kept together
    // with the caller's handle and a fingerprint of the exported C struct (shape + three words of every limb).  A run whose
    // export yields the same handle and fingerprint skips staging, upload and conversion; anything else (another key object, a
    // regenerated key) replaces the entry.  CONTRACT: a caller that rewrites a key IN PLACE so that the sampled words stay the
    // same must call lsa_task_drop_keys.  lsa_task_drop_keys / release free the device copies.
    struct CachedKey {
        const void* handle = nullptr;
        uint64_t fingerprint = 0;
        KeyP key;
    };
    std::map<std::pair<int, NodeIndex>, CachedKey> key_cache;   // touched by the thread that runs the shared levels / the fan-out only
    bool keep_keys = getenv("LSA_NO_KEY_CACHE") == nullptr;
    int last_key_uploads = 0, last_key_hits = 0;
    static uint64_t ksk_fingerprint(const CKeySwitchKey* k, int n) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)k->n_public_key;
        auto mix = [&](uint64_t v) { h = (h ^ v) * 0x100000001B3ull + (h >> 29); };
        for (int d = 0; d < k->n_public_key; d++) {
            const CPublicKey& pk = k->public_keys[d];
            mix((uint64_t)pk.level * 131 + (uint64_t)pk.degree);
            for (int p = 0; p <= pk.degree; p++)
                for (int j = 0; j < pk.polys[p].n_component; j++) {
                    const uint64_t* w = pk.polys[p].components[j].data;
                    mix(w[0]);
                    mix(w[n / 2]);
                    mix(w[n - 1]);
                }
        }
        return h;
    }
    TaskGraph g;
    std::vector<std::vector<ComputeNode*>> levels;
    std::vector<std::vector<ComputeNode*>> shared_levels;                  // key export/load: before every chunk
    std::vector<std::vector<std::vector<ComputeNode*>>> chunk_levels;      // [chunk][level] -> nodes; empty: not pipelined
    std::map<int, std::unique_ptr<Context>> contexts;                      // key = LanePools::key(device, lane)
    std::map<int, hipStream_t> streams;
    std::map<int, std::vector<std::shared_ptr<Slab>>> pending_free_;   // temporaries still referenced by enqueued work, per (device, lane)
    // (entries are created by run() before any shard thread starts: concurrent callers only look their own up)
    std::vector<std::shared_ptr<Slab>>& pending_free() { return pending_free_.at(LanePools::key(tls_exec.device, tls_exec.lane)); }
    std::atomic<int> last_gpu_nodes{0}, last_gpu_batches{0};
    int last_shards = 1, last_chunks = 0, last_key_peer_copies = 0;
    double last_ms = 0;
    std::mutex bootstrap_mu;
    struct BtDeleter {
        void operator()(Bootstrap* b) const { bootstrap_destroy(b); }
    };
    std::map<Context*, std::unique_ptr<Bootstrap, BtDeleter>> bootstrap_plans;   // built on first use, per lane context

    // bootstrapping plan from the task's `parameter` block (reference: gpu_wrapper.cu:86-117)
    Bootstrap& bootstrap_plan(Context& c, hipStream_t s) {
        std::lock_guard<std::mutex> lk(bootstrap_mu);
        auto it = bootstrap_plans.find(&c);
        if (it != bootstrap_plans.end()) return *it->second;
        const mjson::Value& P = g.parameter;
        LSA_REQUIRE(P.contains("btp_output_level"), "bootstrap node in a task without bootstrapping parameters");
        // (the sine TYPE is not among the fields the reference forwards to its GPU library, gpu_wrapper.cu:94-103; Cos1 is what the
        // frontend's parameter sets say and what is implemented)
        LSA_REQUIRE(!P.contains("btp_eval_mod_sine_type") || P["btp_eval_mod_sine_type"].as_string() == "Cos1",
                    "bootstrap: only the Cos1 sine type is implemented");
        const int sine_deg = (int)P["btp_eval_mod_sine_deg"].as_int(), arcsine_deg = (int)P["btp_eval_mod_arcsine_deg"].as_int();
        int log_slots = 0;
        if (P.contains("slots")) {
            const long long slots = P["slots"].as_int();
            LSA_REQUIRE(slots >= 2 && slots <= c.n / 2 && (slots & (slots - 1)) == 0, "bootstrap: slot count must be a power of two <= N/2");
            while ((1LL << log_slots) < slots) log_slots++;
        }
        LSA_REQUIRE(sine_deg >= 1 && sine_deg <= 63, "bootstrap: sine degree outside 1..63");
        LSA_REQUIRE(arcsine_deg >= 0 && arcsine_deg <= 15 && (arcsine_deg == 0 || (arcsine_deg & 1)), "bootstrap: arcsine degree must be 0 or odd and at most 15");
        const int cts_depth = (int)P["btp_cts_depth"].as_int(), stc_depth = (int)P["btp_stc_depth"].as_int();
        LSA_REQUIRE(P["btp_cts_start_level"].as_int() == c.nq - 1 && P["btp_eval_mod_start_level"].as_int() == c.nq - 1 - cts_depth,
                    "bootstrap: level plan differs from the one implemented");
        const double scale = P["scale"].as_double();
        Bootstrap* b = bootstrap_create(c, cts_depth, stc_depth, (int)P["btp_eval_mod_k"].as_int(),
                                        (int)P["btp_eval_mod_double_angle"].as_int(), P["btp_eval_mod_message_ratio"].as_double(),
                                        scale, scale, log_slots, s, sine_deg, arcsine_deg);
        LSA_REQUIRE(bootstrap_out_level(*b) == P["btp_output_level"].as_int() &&
                        P["btp_stc_start_level"].as_int() == bootstrap_out_level(*b) + stc_depth,
                    "bootstrap: level plan differs from the one implemented");
        bootstrap_plans[&c].reset(b);
        return *b;
    }

    explicit fhe_task_handle_st(const std::string& project_path) {
        g = TaskGraph::load_for_gpu(project_path + "/mega_ag.json");
        for (auto& kv : g.computes) {
            ComputeNode& c = kv.second;
            if (c.custom_prop || c.on_cpu) continue;
            const OperationType op = c.op();
            if (op == OperationType::LOAD_TO_BACKEND || op == OperationType::STORE_FROM_BACKEND) continue;
            bind_gpu_executor(c, g.algo);
        }
        levels.assign(g.max_top_level + 1, {});
        for (auto& kv : g.computes) levels[kv.second.sched_meta.top_level].push_back(&kv.second);
        for (auto& lv : levels)
            std::sort(lv.begin(), lv.end(), [](const ComputeNode* a, const ComputeNode* b) { return a->index < b->index; });
        if (!getenv("LSA_NO_PIPELINE")) plan_pipeline(1);
    }

    // Independent subgraphs = connected components of the compute nodes over the non-key data (evaluation keys are shared
    // read-only inputs).  Pipelining needs the simple shape every benchmark graph has: per chunk, CPU nodes only before
    // the loads and after the stores, and all stores in one level.
    int planned_shards_ = 0;
    void plan_pipeline(int n_shards) {
        planned_shards_ = n_shards;
        shared_levels.clear();
        chunk_levels.clear();
        auto is_key = [](const DatumNode* d) {
            return d->datum_type == TYPE_RELIN_KEY || d->datum_type == TYPE_GALOIS_KEY || d->datum_type == TYPE_SWITCH_KEY;
        };
        std::unordered_map<const ComputeNode*, const ComputeNode*> parent;
        std::function<const ComputeNode*(const ComputeNode*)> find = [&](const ComputeNode* x) {
            while (parent[x] != x) x = parent[x] = parent[parent[x]];
            return x;
        };
        for (auto& kv : g.computes) parent[&kv.second] = &kv.second;
        std::unordered_set<const ComputeNode*> key_only;
        for (auto& kv : g.computes) {
            bool all_key = true;
            for (auto* d : kv.second.input_nodes) all_key = all_key && is_key(d);
            for (auto* d : kv.second.output_nodes) all_key = all_key && is_key(d);
            if (all_key) key_only.insert(&kv.second);
        }
        for (auto& kv : g.data) {
            const DatumNode& d = kv.second;
            if (is_key(&d)) continue;
            const ComputeNode* first = nullptr;
            auto join = [&](const ComputeNode* c) {
                if (!first) first = c;
                else parent[find(c)] = find(first);
            };
            for (auto* c : d.predecessors) join(c);
            for (auto* c : d.successors) join(c);
        }
        std::map<NodeIndex, const ComputeNode*> comps;   // smallest node index -> representative
        std::unordered_map<const ComputeNode*, NodeIndex> lowest;
        for (auto& kv : g.computes) {
            if (key_only.count(&kv.second)) continue;
            const ComputeNode* r = find(&kv.second);
            auto it = lowest.find(r);
            if (it == lowest.end() || kv.first < it->second) lowest[r] = kv.first;
        }
        for (auto& kv : lowest) comps[kv.second] = kv.first;
        if (comps.size() < 4) return;
        // worth it only when the copies dominate: small graphs keep the whole-level batches (fewer, larger launches)
        double in_bytes = 0;
        const double n_ring = (double)g.parameter["n"].as_int();
        for (NodeIndex idx : g.inputs) {
            const DatumNode& d = g.data.at(idx);
            if (is_key(&d) || !d.fhe_prop) continue;
            const bool ringt = d.fhe_prop->p && d.fhe_prop->p->is_ringt;
            in_bytes += 8.0 * n_ring * (d.datum_type == TYPE_CIPHERTEXT ? d.fhe_prop->degree + 1 : 1) * (ringt ? 1 : d.fhe_prop->level + 1);
        }
        const char* min_mib = getenv("LSA_PIPELINE_MIN_MIB");   // tests force the pipelined path on small graphs with 0
        if (in_bytes < (min_mib ? atof(min_mib) : 256.0) * 1048576.0) return;
        const int nchunks = plan_chunk_count(comps.size(), n_shards);
        std::unordered_map<const ComputeNode*, int> chunk_of;
        int ci = 0;
        for (auto& kv : comps) chunk_of[kv.second] = (int)((long long)ci++ * nchunks / (long long)comps.size());
        std::vector<std::vector<std::vector<ComputeNode*>>> cl(nchunks, std::vector<std::vector<ComputeNode*>>(levels.size()));
        std::vector<std::vector<ComputeNode*>> sh(levels.size());
        for (size_t l = 0; l < levels.size(); l++)
            for (ComputeNode* n : levels[l]) {
                if (key_only.count(n)) sh[l].push_back(n);
                else cl[chunk_of.at(find(n))][l].push_back(n);
            }
        for (auto& chunk : cl) {   // shape check
            int first_load = -1, store_level = -1;
            for (size_t l = 0; l < chunk.size(); l++)
                for (ComputeNode* n : chunk[l]) {
                    if (n->op() == OperationType::LOAD_TO_BACKEND && first_load < 0) first_load = (int)l;
                    if (n->op() == OperationType::STORE_FROM_BACKEND) {
                        if (store_level >= 0 && store_level != (int)l) return;
                        store_level = (int)l;
                    }
                }
            if (first_load < 0 || store_level < 0) return;
            for (size_t l = 0; l < chunk.size(); l++)
                for (ComputeNode* n : chunk[l]) {
                    if (n->on_cpu && (int)l >= first_load && (int)l <= store_level) return;
                    // from the store level on: nothing but the stores themselves and CPU-side nodes
                    if ((int)l >= store_level && !n->on_cpu && n->op() != OperationType::STORE_FROM_BACKEND) return;
                }
        }
        shared_levels = std::move(sh);
        chunk_levels = std::move(cl);
    }
    ~fhe_task_handle_st() {
        for (auto& kv : streams) {
            (void)hipSetDevice(LanePools::device_of(kv.first));
            (void)hipStreamDestroy(kv.second);
        }
    }

    Context& context(int device, int lane = 0) {
        const int key = LanePools::key(device, lane);
        auto it = contexts.find(key);
        if (it != contexts.end()) {
            it->second->use_device();
            return *it->second;
        }
        const mjson::Value& P = g.parameter;
        const int n = (int)P["n"].as_int();
        const int max_level = (int)P["max_level"].as_int();
        std::vector<u64> q = P["q"].as_u64_vector(), p = P["p"].as_u64_vector();
        // frontend/parameter.json lists 30 primes for CKKS n=65536 but says max_level 33: use what exists
        if ((int)q.size() > max_level + 1) q.resize(max_level + 1);
        u64 t = 0;
        if (g.algo == ALGO_BFV) t = P["t"].as_u64();
        auto c = std::make_unique<Context>(g.algo == ALGO_BFV ? LSA_ALGO_BFV : LSA_ALGO_CKKS, n, q.data(), (int)q.size(),
                                           p.data(), (int)p.size(), t, device);
        hipStream_t s;
        LSA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        streams[key] = s;
        Context& ref = *c;
        contexts[key] = std::move(c);
        return ref;
    }

    // ---------------------------------------------------------------- LOAD_TO_BACKEND (batched H2D)
    // returns the pinned staging slab: it must outlive the enqueued copies (the caller synchronises or keeps it)
    std::shared_ptr<Slab> run_loads(Context& c, hipStream_t s, const std::vector<ComputeNode*>& nodes,
                                    std::unordered_map<NodeIndex, std::any>& avail) {
        const long long N = c.n;
        // 1. ciphertexts / plaintexts grouped into one slab per (kind, polys, level) in node order
        struct Item {
            ComputeNode* node;
            std::shared_ptr<CCiphertext> ct;
            std::shared_ptr<CPlaintext> pt;
            int polys, level;
            size_t off;
        };
        std::map<std::tuple<int, int, int>, std::vector<Item>> groups;
        std::vector<ComputeNode*> key_nodes;
        for (ComputeNode* node : nodes) {
            const DatumNode* in = node->input_nodes[0];
            const std::any& cs = avail.at(in->index);
            // Every operand is sized from the graph's fhe_prop downstream (gather / run_gpu_bucket): a C struct that disagrees
            // with the task's declaration would make those kernels read past the loaded slab, so it is refused here.
            auto bad = [&](const std::string& what) {
                // name the caller's datum: the C struct is the output of the export node inserted in front of this load
                const DatumNode* orig = in;
                if (!in->predecessors.empty() && in->predecessors[0]->op() == OperationType::EXPORT_TO_ABI && !in->predecessors[0]->input_nodes.empty())
                    orig = in->predecessors[0]->input_nodes[0];
                throw Error(LSA_ERR_ARG, "input '" + orig->id + "' (datum " + std::to_string(orig->index) + "): " + what);
            };
            if (in->datum_type == TYPE_CIPHERTEXT) {
                auto ct = std::any_cast<std::shared_ptr<CCiphertext>>(cs);
                if (!ct || !ct->polys) bad("null ciphertext C struct");
                if (in->fhe_prop && (ct->level != in->fhe_prop->level || ct->degree != in->fhe_prop->degree))
                    bad("ciphertext C struct has level/degree " + std::to_string(ct->level) + "/" + std::to_string(ct->degree) +
                        ", the task declares " + std::to_string(in->fhe_prop->level) + "/" + std::to_string(in->fhe_prop->degree));
                if (ct->level < 0 || ct->level >= c.nq || ct->degree < 0) bad("ciphertext level/degree out of range");
                for (int p = 0; p <= ct->degree; p++) {
                    if (!ct->polys[p].components || ct->polys[p].n_component != ct->level + 1) bad("ciphertext C struct: limb count != level+1");
                    for (int j = 0; j <= ct->level; j++)
                        if (ct->polys[p].components[j].n != c.n || !ct->polys[p].components[j].data) bad("ciphertext C struct has a wrong ring degree");
                }
                groups[{0, ct->degree + 1, ct->level}].push_back({node, ct, nullptr, ct->degree + 1, ct->level, 0});
            } else if (in->datum_type == TYPE_PLAINTEXT) {
                auto pt = std::any_cast<std::shared_ptr<CPlaintext>>(cs);
                if (!pt || !pt->poly.components) bad("null plaintext C struct");
                const bool ringt = is_ringt_node(in);
                const int want = ringt ? 1 : (in->fhe_prop ? in->fhe_prop->level + 1 : pt->poly.n_component);
                if (pt->poly.n_component != want)
                    bad("plaintext C struct has " + std::to_string(pt->poly.n_component) + " limbs, the task declares " + std::to_string(want));
                if (want < 1 || want > c.nq) bad("plaintext level out of range");
                for (int j = 0; j < want; j++)
                    if (pt->poly.components[j].n != c.n || !pt->poly.components[j].data) bad("plaintext C struct has a wrong ring degree");
                groups[{1, 1, pt->poly.n_component - 1}].push_back({node, nullptr, pt, 1, pt->poly.n_component - 1, 0});
            } else {
                key_nodes.push_back(node);
            }
        }
        // a group whose every item is one contiguous block of caller-registered (pinned) memory is copied from where it lies
        auto item_base = [&](const Item& it) -> const u64* {
            const u64* base = (it.ct ? it.ct->polys[0] : it.pt->poly).components[0].data;
            const u64* want = base;
            for (int p = 0; p < it.polys; p++) {
                const CPolynomial& poly = it.ct ? it.ct->polys[p] : it.pt->poly;
                for (int j = 0; j <= it.level; j++, want += N)
                    if (poly.components[j].data != want) return nullptr;
            }
            return host_registry().covers(base, (size_t)it.polys * (it.level + 1) * N * sizeof(u64)) ? base : nullptr;
        };
        std::map<std::tuple<int, int, int>, std::vector<const u64*>> direct;   // group -> per-item host base (all or nothing)
        for (auto& kv : groups) {
            std::vector<const u64*> bases;
            for (auto& it : kv.second) {
                const u64* b = item_base(it);
                if (!b) break;
                bases.push_back(b);
            }
            if (bases.size() == kv.second.size()) direct[kv.first] = std::move(bases);
        }
        size_t total = 0;
        for (auto& kv : groups) {
            if (direct.count(kv.first)) continue;
            for (auto& it : kv.second) {
                it.off = total;
                total += (size_t)it.polys * (it.level + 1) * N;
            }
        }
        // keys: compact order [beta][2][comp][N]
        struct KeyItem {
            ComputeNode* node;
            const CKeySwitchKey* ksk;
            std::any keep;
            int level, beta, comp;
            size_t off;
            const void* handle;
            uint64_t fingerprint;
        };
        std::vector<KeyItem> keys;
        for (ComputeNode* node : key_nodes) {
            const DatumNode* in = node->input_nodes[0];
            const std::any& cs = avail.at(in->index);
            KeyItem k{};
            k.node = node;
            k.keep = cs;
            if (in->datum_type == TYPE_RELIN_KEY) {
                k.ksk = std::any_cast<std::shared_ptr<CRelinKey>>(cs).get();
            } else if (in->datum_type == TYPE_SWITCH_KEY) {
                k.ksk = std::any_cast<std::shared_ptr<CKeySwitchKey>>(cs).get();
            } else {
                auto glk = std::any_cast<std::shared_ptr<CGaloisKey>>(cs);
                const uint32_t want = in->fhe_prop->p ? in->fhe_prop->p->galois_element : 0;
                k.ksk = nullptr;
                for (int i = 0; i < glk->n_key_switch_key; i++)
                    if (glk->galois_elements[i] == want) k.ksk = &glk->key_switch_keys[i];
                LSA_REQUIRE(k.ksk != nullptr, "Galois key for element " + std::to_string(want) + " not found in the C struct");
            }
            LSA_REQUIRE(k.ksk && k.ksk->n_public_key >= 1, "empty key-switch key");
            k.level = k.ksk->public_keys[0].level;
            k.beta = k.ksk->n_public_key;
            k.comp = k.ksk->public_keys[0].polys[0].n_component;
            LSA_REQUIRE(k.comp == k.level + 1 + c.np, "key-switch key: limbs per polynomial != level+1+#special primes");
            LSA_REQUIRE(k.beta == (k.level + 1 + c.np - 1) / c.np, "key-switch key: digit count != ceil((level+1)/k)");
            LSA_REQUIRE(k.level >= 0 && k.level < c.nq, "key-switch key: level out of range");
            for (int d = 0; d < k.beta; d++) {
                const CPublicKey& pk = k.ksk->public_keys[d];
                LSA_REQUIRE(pk.polys && pk.degree == 1 && pk.level == k.level, "key-switch key: digits differ in level or degree");
                for (int h = 0; h < 2; h++) {
                    LSA_REQUIRE(pk.polys[h].components && pk.polys[h].n_component == k.comp, "key-switch key: limb count differs between digits");
                    for (int j = 0; j < k.comp; j++)
                        LSA_REQUIRE(pk.polys[h].components[j].n == c.n && pk.polys[h].components[j].data, "key-switch key has a wrong ring degree");
                }
            }
            // resident already?  (same caller handle behind the export node, same fingerprint of what it exported)
            const DatumNode* orig = (!in->predecessors.empty() && !in->predecessors[0]->input_nodes.empty()) ? in->predecessors[0]->input_nodes[0] : in;
            const std::any* hv = avail.count(orig->index) ? &avail.at(orig->index) : nullptr;
            const std::shared_ptr<void>* hp = hv ? std::any_cast<std::shared_ptr<void>>(hv) : nullptr;
            k.handle = hp ? hp->get() : nullptr;
            k.fingerprint = ksk_fingerprint(k.ksk, c.n);
            if (keep_keys) {
                auto hit = key_cache.find({c.device, node->output_nodes[0]->index});
                if (hit != key_cache.end() && hit->second.handle == k.handle && hit->second.fingerprint == k.fingerprint &&
                    hit->second.key->key.level == k.level) {
                    avail[node->output_nodes[0]->index] = hit->second.key;
                    last_key_hits++;
                    continue;
                }
            }
            k.off = total;
            total += (size_t)k.beta * 2 * k.comp * N;
            keys.push_back(std::move(k));
        }
        // 1b. direct groups: device slab + one copy per item straight from the caller's pinned buffer
        for (auto& kv : direct) {
            auto& items = groups.at(kv.first);
            const size_t per = (size_t)items[0].polys * (items[0].level + 1) * N;
            auto slab = dslab(per * items.size());
            for (size_t i = 0; i < items.size(); i++) {
                LSA_HIP(hipMemcpyAsync(slab->ptr + per * i, kv.second[i], per * sizeof(u64), hipMemcpyHostToDevice, s));
                auto d = std::make_shared<DevDatum>();
                d->slab = slab;
                d->ptr = slab->ptr + per * i;
                d->polys = items[i].polys;
                d->level = items[i].level;
                d->is_plain = std::get<0>(kv.first) == 1;
                avail[items[i].node->output_nodes[0]->index] = d;
            }
            last_direct_loads += (int)items.size();
        }
        if (total == 0) return nullptr;
        // 2. gather limbs into a pinned staging slab, one H2D copy per group
        auto hstage = pslab(total);
        u64* host = hstage->ptr;
        struct Job {
            u64* dst;
            const u64* src;
        };
        std::vector<Job> jobs;   // one limb each, gathered by a few threads (single-threaded this was ~55 % of LOAD)
        for (auto& kv : groups) {
            if (direct.count(kv.first)) continue;
            for (auto& it : kv.second) {
                u64* dst = host + it.off;
                for (int p = 0; p < it.polys; p++) {
                    const CPolynomial& poly = it.ct ? it.ct->polys[p] : it.pt->poly;
                    for (int j = 0; j <= it.level; j++) {
                        jobs.push_back({dst, poly.components[j].data});
                        dst += N;
                    }
                }
            }
        }
        for (auto& k : keys) {
            u64* dst = host + k.off;
            for (int d = 0; d < k.beta; d++)
                for (int h = 0; h < 2; h++)
                    for (int j = 0; j < k.comp; j++) {
                        jobs.push_back({dst, k.ksk->public_keys[d].polys[h].components[j].data});
                        dst += N;
                    }
        }
        // device destinations first (one slab per group / key), as segments of the staging order
        struct Seg {
            size_t off, words;
            u64* dev;
        };
        std::vector<Seg> segs;
        for (auto& kv : groups) {
            if (direct.count(kv.first)) continue;
            auto& items = kv.second;
            const size_t per = (size_t)items[0].polys * (items[0].level + 1) * N;
            auto slab = dslab(per * items.size());
            segs.push_back({items[0].off, per * items.size(), slab->ptr});
            for (size_t i = 0; i < items.size(); i++) {
                auto d = std::make_shared<DevDatum>();
                d->slab = slab;
                d->ptr = slab->ptr + per * i;
                d->polys = items[i].polys;
                d->level = items[i].level;
                d->is_plain = std::get<0>(kv.first) == 1;
                avail[items[i].node->output_nodes[0]->index] = d;
            }
        }
        std::vector<std::shared_ptr<DevKey>> dkeys;
        for (auto& k : keys) {
            const size_t words = (size_t)k.beta * 2 * k.comp * N;
            auto dk = std::make_shared<DevKey>();
            dk->slab = dslab(ks_fused_enabled(c) ? 2 * words : words);   // (+ the key as doubles behind it, launch_key_prepare)
            segs.push_back({k.off, words, dk->slab->ptr});
            dkeys.push_back(dk);
        }
        // gather and copy in chunks: while the DMA engine moves chunk k, the host threads gather chunk k+1 (the staging
        // slab holds the whole level, so no chunk waits for a buffer)
        const size_t chunk_jobs = std::max<size_t>(1, (32u << 20) / (sizeof(u64) * (size_t)N));
        for (size_t j0 = 0; j0 < jobs.size(); j0 += chunk_jobs) {
            const size_t j1 = std::min(jobs.size(), j0 + chunk_jobs);
            parallel_for(j1 - j0, [&](size_t i) { stream_copy(jobs[j0 + i].dst, jobs[j0 + i].src, (size_t)N); });
            const size_t h0 = j0 * (size_t)N, h1 = j1 * (size_t)N;   // jobs are in staging order, one limb each
            for (const Seg& sg : segs) {
                const size_t a = std::max(h0, sg.off), b = std::min(h1, sg.off + sg.words);
                if (a < b)
                    LSA_HIP(hipMemcpyAsync(sg.dev + (a - sg.off), host + a, (b - a) * sizeof(u64), hipMemcpyHostToDevice, s));
            }
        }
        for (size_t i = 0; i < keys.size(); i++) {
            auto& k = keys[i];
            auto& dk = dkeys[i];
            dk->key.data = dk->slab->ptr;
            dk->key.level = k.level;
            dk->key.owned = false;
            double* fp = ks_fused_enabled(c) ? reinterpret_cast<double*>(dk->slab->ptr + (size_t)k.beta * 2 * k.comp * N) : nullptr;
            launch_key_prepare(c, dk->key.data, fp, k.level, s);
            dk->key.fp = fp;
            avail[k.node->output_nodes[0]->index] = dk;
            last_key_uploads++;
            if (keep_keys) key_cache[{c.device, k.node->output_nodes[0]->index}] = CachedKey{k.handle, k.fingerprint, dk};
        }
        return hstage;
    }

    // ---------------------------------------------------------------- STORE_FROM_BACKEND (batched D2H)
    struct StoreJob {
        std::vector<ComputeNode*> nodes;
        std::vector<std::pair<DatumP, size_t>> items;   // offset into the pinned slab (staged results)
        std::vector<u64*> direct;                       // non-null: the result was copied straight into the caller's buffer
        std::shared_ptr<Slab> hslab;
    };
    bool native_frontend = false;                                  // lsa_frontend_bind: output handles are lsa_host_ciphertext
    const std::unordered_map<NodeIndex, void*>* run_out_handles = nullptr;   // the current run's output handles (read-only)
    std::atomic<int> last_direct_loads{0}, last_direct_stores{0};
    void run_stores(Context& c, hipStream_t s, const std::vector<ComputeNode*>& nodes,
                    std::unordered_map<NodeIndex, std::any>& avail) {
        StoreJob j = stores_enqueue(c, s, nodes, avail);
        LSA_HIP(hipStreamSynchronize(s));
        stores_finish(c, j, avail);
    }
    // where a store node's result can be written directly: the native front-end's pre-allocated output ciphertext behind the
    // import node that follows, if the caller registered (pinned) its buffer and it has the result's shape
    u64* direct_store_target(const ComputeNode* store, const DevDatum& d, int n) {
        if (!native_frontend || !run_out_handles) return nullptr;
        const DatumNode* cs = store->output_nodes[0];
        if (cs->successors.size() != 1 || cs->successors[0]->op() != OperationType::IMPORT_FROM_ABI) return nullptr;
        auto it = run_out_handles->find(cs->successors[0]->output_nodes[0]->index);
        if (it == run_out_handles->end() || !it->second) return nullptr;
        const auto* h = (const lsa_host_ciphertext*)it->second;
        if (!h->data || h->n != n || h->level != d.level || h->degree != d.polys - 1) return nullptr;
        const size_t bytes = (size_t)d.polys * (d.level + 1) * n * sizeof(u64);
        return host_registry().covers(h->data, bytes) ? h->data : nullptr;
    }
    StoreJob stores_enqueue(Context& c, hipStream_t s, const std::vector<ComputeNode*>& nodes,
                            std::unordered_map<NodeIndex, std::any>& avail) {
        const long long N = c.n;
        size_t total = 0;
        std::vector<std::pair<DatumP, size_t>> items;
        std::vector<u64*> direct;
        for (ComputeNode* node : nodes) {
            const DatumNode* in = node->input_nodes[0];
            LSA_REQUIRE(in->datum_type == TYPE_CIPHERTEXT, "Unsupported data type for D2H transfer");
            auto d = std::any_cast<DatumP>(avail.at(in->index));
            u64* tgt = direct_store_target(node, *d, c.n);
            direct.push_back(tgt);
            items.push_back({d, total});
            if (!tgt) total += (size_t)d->polys * (d->level + 1) * N;
        }
        // staged results land in ONE pooled pinned slab; the C structs handed to the caller's import executor only index it
        // (no malloc per limb, no second host copy).  The slab returns to the pool when the last struct is released.
        std::shared_ptr<Slab> hslab = total ? pslab(total) : nullptr;
        u64* host = hslab ? hslab->ptr : nullptr;
        // merge runs that are contiguous on the device into single copies
        for (size_t i = 0; i < items.size();) {
            if (direct[i]) {
                const size_t words = (size_t)items[i].first->polys * (items[i].first->level + 1) * N;
                LSA_HIP(hipMemcpyAsync(direct[i], items[i].first->ptr, words * sizeof(u64), hipMemcpyDeviceToHost, s));
                last_direct_stores++;
                i++;
                continue;
            }
            size_t j = i, words = 0;
            while (j < items.size() && !direct[j] && items[j].first->ptr == items[i].first->ptr + words) {
                words += (size_t)items[j].first->polys * (items[j].first->level + 1) * N;
                j++;
            }
            LSA_HIP(hipMemcpyAsync(host + items[i].second, items[i].first->ptr, words * sizeof(u64), hipMemcpyDeviceToHost, s));
            i = j;
        }
        return StoreJob{nodes, std::move(items), std::move(direct), hslab};
    }
    // after the stream has been synchronised: wrap the pinned result slab into C structs for the import executor
    void stores_finish(Context& c, StoreJob& job, std::unordered_map<NodeIndex, std::any>& avail) {
        const long long N = c.n;
        const auto& nodes = job.nodes;
        const auto& items = job.items;
        auto hslab = job.hslab;
        u64* host = hslab ? hslab->ptr : nullptr;
        for (size_t i = 0; i < nodes.size(); i++) {
            const DatumP& d = items[i].first;
            auto* ct = (CCiphertext*)malloc(sizeof(CCiphertext));
            ct->level = d->level;
            ct->degree = d->polys - 1;
            ct->polys = (CPolynomial*)malloc(sizeof(CPolynomial) * (size_t)d->polys);
            u64* src = job.direct[i] ? job.direct[i] : host + items[i].second;   // (direct: the struct indexes the caller's own buffer)
            for (int p = 0; p < d->polys; p++) {
                ct->polys[p].n_component = d->level + 1;
                ct->polys[p].components = (CComponent*)malloc(sizeof(CComponent) * (size_t)(d->level + 1));
                for (int j = 0; j <= d->level; j++) {
                    ct->polys[p].components[j].n = c.n;
                    ct->polys[p].components[j].data = src;
                    src += N;
                }
            }
            std::shared_ptr<CCiphertext> sp(ct, [hslab](CCiphertext* q) {
                for (int p = 0; p <= q->degree; p++) free(q->polys[p].components);
                free(q->polys);
                free(q);
            });
            avail[nodes[i]->output_nodes[0]->index] = sp;
        }
    }

    // ---------------------------------------------------------------- batched operator dispatch
    struct Operand {
        const u64* ptr;
        long long stride;
        std::shared_ptr<Slab> keep;
    };
    // operand `pos` of every node of the bucket as (base, stride); gathers with D2D copies if not already strided
    Operand gather(Context& c, hipStream_t s, const std::vector<ComputeNode*>& nodes, int pos,
                   std::unordered_map<NodeIndex, std::any>& avail, size_t words) {
        std::vector<DatumP> d;
        for (auto* n : nodes) d.push_back(std::any_cast<DatumP>(avail.at(n->input_nodes[pos]->index)));
        Operand o{d[0]->ptr, (long long)words, nullptr};
        if (d.size() == 1) return o;
        const long long st = d[1]->ptr - d[0]->ptr;
        bool strided = st >= (long long)words || st == 0;
        for (size_t i = 1; i < d.size() && strided; i++) strided = (d[i]->ptr - d[0]->ptr) == st * (long long)i;
        if (strided && st != 0) {
            o.stride = st;
            return o;
        }
        o.keep = dslab(words * d.size());
        for (size_t i = 0; i < d.size(); i++)
            LSA_HIP(hipMemcpyAsync(o.keep->ptr + words * i, d[i]->ptr, words * sizeof(u64), hipMemcpyDeviceToDevice, s));
        o.ptr = o.keep->ptr;
        o.stride = (long long)words;
        pending_free().push_back(o.keep);
        return o;
    }

    static std::string signature(const ComputeNode* n) {
        std::string sg = std::to_string((int)n->op());
        for (auto* in : n->input_nodes) {
            sg += "|" + std::to_string((int)in->datum_type) + ":" + std::to_string(in->fhe_prop->level) + ":" +
                  std::to_string(in->fhe_prop->degree);
            // pt, pt_mul and pt_ringt all arrive as TYPE_PLAINTEXT: the plaintext's treatment (lift, transform) is decided per
            // bucket from its first node, so the flavour is part of the signature
            if (in->datum_type == TYPE_PLAINTEXT) {
                sg += (in->fhe_prop->p && in->fhe_prop->p->is_ringt) ? "r" : (in->fhe_prop->is_ntt ? "n" : "c");
                if (in->fhe_prop->is_mform) sg += "m";
            }
            // all nodes of a bucket must use the SAME key datum
            if (in->datum_type != TYPE_CIPHERTEXT && in->datum_type != TYPE_PLAINTEXT) sg += "#" + std::to_string(in->index);
        }
        sg += ">" + std::to_string(n->output_nodes[0]->fhe_prop->level) + ":" + std::to_string(n->input_nodes.size());
        return sg;
    }

    void run_gpu_bucket(Context& c, hipStream_t s, const std::vector<ComputeNode*>& nodes,
                        std::unordered_map<NodeIndex, std::any>& avail) {
        const long long N = c.n;
        const ComputeNode* n0 = nodes[0];
        const OperationType op = n0->op();
        const int m = (int)nodes.size();
        const DatumNode* in0 = n0->input_nodes[0];
        const int lvl = in0->fhe_prop->level, L = lvl + 1;
        const int polys_in = in0->fhe_prop->degree + 1;
        const int out_lvl = n0->output_nodes[0]->fhe_prop->level;
        const bool bfv = g.algo == ALGO_BFV;
        const size_t w_in = (size_t)polys_in * L * N;
        int out_polys = polys_in;
        if (op == OperationType::MULTIPLY && !(n0->input_nodes.size() == 2 && is_plain_node(n0->input_nodes[1]))) out_polys = 3;
        if (op == OperationType::RELINEARIZE || op == OperationType::FUSED_MULT_RELIN_RESCALE) out_polys = 2;
        const size_t w_out = (size_t)out_polys * (out_lvl + 1) * N;
        auto out_slab = dslab(w_out * m);
        u64* out = out_slab->ptr;
        const long long so = (long long)w_out;
        Operand a = gather(c, s, nodes, 0, avail, w_in);
        RowMap rmL;
        rmL.period = L;
        for (int i = 0; i < L; i++) rmL.mod_of[i] = (unsigned char)i;

        auto key_of = [&](int pos) -> const Key& { return std::any_cast<KeyP>(avail.at(n0->input_nodes[pos]->index))->key; };
        // plaintext operand `pos` of every node as [m][L][N] limbs in the domain the operator needs.
        //   full plaintext: used as loaded (CKKS: NTT domain; BFV: coefficient domain, already scaled)
        //   ring-t plaintext (one limb): lifted per `ringt_mode` (kernels.hip k_lift_ringt), then NTT'd if `to_ntt`
        auto plain_operand = [&](int pos, int ringt_mode, bool to_ntt) -> Operand {
            if (!is_ringt_node(n0->input_nodes[pos])) return gather(c, s, nodes, pos, avail, (size_t)L * N);
            Operand raw = gather(c, s, nodes, pos, avail, (size_t)N);
            Operand o{nullptr, (long long)L * N, dslab((size_t)m * L * N)};
            pending_free().push_back(o.keep);
            launch_lift_ringt(c, ringt_mode, lvl, raw.ptr, raw.stride, o.keep->ptr, o.stride, m, s);
            if (to_ntt) launch_ntt(c, o.keep->ptr, o.keep->ptr, m, o.stride, L, rmL, false, s);
            o.ptr = o.keep->ptr;
            return o;
        };
        auto temp = [&](size_t words) {
            auto sl = dslab(words);
            pending_free().push_back(sl);
            return sl->ptr;
        };

        switch (op) {
            case OperationType::ADD:
            case OperationType::SUB: {
                const EwOp ew = op == OperationType::ADD ? EW_ADD : EW_SUB;
                if (n0->input_nodes.size() == 1) {
                    launch_elementwise(c, ew, a.ptr, a.ptr, out, m, a.stride, a.stride, so, polys_in * L, rmL, s);
                } else if (is_plain_node(n0->input_nodes[1])) {
                    // CKKS: plaintext limbs in the NTT domain (ring-t: centred lift + NTT); BFV: coefficient domain
                    // (ring-t: scaled up by Q/t with rounding)
                    Operand b = plain_operand(1, bfv ? 2 : 0, !bfv);
                    std::vector<int> rows(polys_in * L);
                    for (size_t i = 0; i < rows.size(); i++) rows[i] = (int)i;
                    launch_copy_rows(c, a.ptr, a.stride, out, so, polys_in * L, rows.data(), m, s);
                    launch_elementwise(c, ew, a.ptr, b.ptr, out, m, a.stride, b.stride, so, L, rmL, s);  // c0 +/- pt
                } else {
                    Operand b = gather(c, s, nodes, 1, avail, w_in);
                    launch_elementwise(c, ew, a.ptr, b.ptr, out, m, a.stride, b.stride, so, polys_in * L, rmL, s);
                }
                break;
            }
            case OperationType::NEGATE:
                launch_elementwise(c, EW_NEG, a.ptr, nullptr, out, m, a.stride, 0, so, polys_in * L, rmL, s);
                break;
            case OperationType::MULTIPLY: {
                if (n0->input_nodes.size() == 2 && is_plain_node(n0->input_nodes[1])) {
                    if (!bfv) {  // CKKS ct * pt, both NTT domain (ring-t: centred lift + NTT first)
                        Operand b = plain_operand(1, 0, true);
                        for (int p = 0; p < polys_in; p++)
                            launch_elementwise(c, EW_MUL, a.ptr + (size_t)p * L * N, b.ptr, out + (size_t)p * L * N, m,
                                               a.stride, b.stride, so, L, rmL, s);
                    } else {     // BFV ct * ring-t pt: NTT(ct) . NTT(pt as residues), back to coefficients
                        Operand b = plain_operand(1, 1, true);
                        std::vector<int> rows(polys_in * L);
                        for (size_t i = 0; i < rows.size(); i++) rows[i] = (int)i;
                        launch_copy_rows(c, a.ptr, a.stride, out, so, polys_in * L, rows.data(), m, s);
                        launch_ntt(c, out, out, m, so, polys_in * L, rmL, false, s);
                        for (int p = 0; p < polys_in; p++)
                            launch_elementwise(c, EW_MUL, out + (size_t)p * L * N, b.ptr, out + (size_t)p * L * N, m, so,
                                               b.stride, so, L, rmL, s);
                        launch_ntt(c, out, out, m, so, polys_in * L, rmL, true, s);
                    }
                    break;
                }
                Operand b = n0->input_nodes.size() == 1 ? a : gather(c, s, nodes, 1, avail, w_in);
                LSA_REQUIRE(polys_in == 2, "ciphertext multiply expects degree-1 operands");
                if (bfv) bfv_mult(c, lvl, a.ptr, b.ptr, out, m, a.stride, b.stride, so, s);
                else launch_tensor(c, a.ptr, b.ptr, out, m, a.stride, b.stride, so, L, rmL, s);
                break;
            }
            case OperationType::FUSED_MULT_RELIN_RESCALE: {   // inputs [a, (b,) rlk]
                LSA_REQUIRE(!bfv && polys_in == 2 && out_lvl == lvl - 1, "fused mult+relin+rescale: unexpected shape");
                const int kpos = (int)n0->input_nodes.size() - 1;
                Operand b = kpos == 1 ? a : gather(c, s, nodes, 1, avail, w_in);
                ckks_mult_relin_rescale(c, lvl, a.ptr, b.ptr, key_of(kpos), out, m, a.stride, b.stride, so, s);
                break;
            }
            case OperationType::BOOTSTRAP: {
                LSA_REQUIRE(!bfv && polys_in == 2 && lvl == 0, "bootstrap expects a degree-1 CKKS ciphertext at level 0");
                Bootstrap& plan = bootstrap_plan(c, s);
                LSA_REQUIRE(out_lvl == bootstrap_out_level(plan), "bootstrap: output datum is not at the bootstrap output level");
                std::map<u64, const Key*> glk;
                std::vector<const Key*> swk;
                for (size_t i = 2; i < n0->input_nodes.size(); i++) {
                    const DatumNode* kd = n0->input_nodes[i];
                    if (kd->datum_type == TYPE_GALOIS_KEY) {
                        LSA_REQUIRE(kd->fhe_prop && kd->fhe_prop->p, "Galois element missing on the key datum");
                        glk[kd->fhe_prop->p->galois_element] = &key_of((int)i);
                    } else if (kd->datum_type == TYPE_SWITCH_KEY) {
                        swk.push_back(&key_of((int)i));
                    }
                }
                LSA_REQUIRE(swk.empty() || swk.size() == 2, "bootstrap: swk_dts and swk_std come as a pair");
                bootstrap_run(plan, a.ptr, a.stride, out, so, m, key_of(1), glk, swk.empty() ? nullptr : swk[0],
                              swk.empty() ? nullptr : swk[1], s);
                break;
            }
            case OperationType::RELINEARIZE:
                LSA_REQUIRE(polys_in == 3, "relinearize expects a degree-2 ciphertext");
                if (bfv) bfv_relin(c, lvl, a.ptr, key_of(1), out, m, a.stride, so, s);
                else ckks_relin(c, lvl, a.ptr, key_of(1), out, m, a.stride, so, s);
                break;
            case OperationType::RESCALE:
                LSA_REQUIRE(out_lvl == lvl - 1, "rescale must drop exactly one level");
                if (bfv) bfv_rescale(c, lvl, polys_in, a.ptr, out, m, a.stride, so, s);
                else ckks_rescale(c, lvl, polys_in, a.ptr, out, m, a.stride, so, s);
                break;
            case OperationType::DROP_LEVEL: {
                LSA_REQUIRE(out_lvl < lvl && out_lvl >= 0, "drop_level must lower the level");
                std::vector<int> rows;
                for (int p = 0; p < polys_in; p++)
                    for (int i = 0; i <= out_lvl; i++) rows.push_back(p * L + i);
                launch_copy_rows(c, a.ptr, a.stride, out, so, (int)rows.size(), rows.data(), m, s);
                break;
            }
            case OperationType::ROTATE_COL:
            case OperationType::ROTATE_ROW: {
                LSA_REQUIRE(polys_in == 2, "rotation expects a degree-1 ciphertext");
                const DatumNode* kd = n0->input_nodes[1];
                u64 gel = op == OperationType::ROTATE_ROW ? 2 * (u64)c.n - 1 : (kd->fhe_prop->p ? kd->fhe_prop->p->galois_element : 0);
                LSA_REQUIRE(gel != 0, "Galois element missing on the key datum");
                if (bfv) bfv_rotate(c, lvl, a.ptr, gel, key_of(1), out, m, a.stride, so, s);
                else ckks_rotate(c, lvl, a.ptr, gel, key_of(1), out, m, a.stride, so, s);
                break;
            }
            case OperationType::MAC_WO_PARTIAL_SUM:
            case OperationType::MAC_W_PARTIAL_SUM: {
                // inputs: ct_0..ct_{n-1}, (ct_partial,) pt_0..pt_{n-1}   (frontend/custom_task.py:1753-1836)
                // out = sum_i ct_i * pt_i (+ ct_partial); mega_ag_executors_gpu.cu:294-408 does multiply_plain + add_inplace
                const int n = n0->fhe_prop->p->sum_cnt;
                const bool with_partial = op == OperationType::MAC_W_PARTIAL_SUM;
                const int pt0 = n + (with_partial ? 1 : 0);
                LSA_REQUIRE((int)n0->input_nodes.size() == pt0 + n, "MAC node: unexpected number of inputs");
                const int rows = polys_in * L;
                if (!bfv) {   // CKKS: every operand is in the NTT domain already -> groups of <= 16 terms per launch
                    Operand part{nullptr, 0, nullptr};
                    if (with_partial) part = gather(c, s, nodes, n, avail, w_in);
                    for (int i0 = 0; i0 < n; i0 += LSA_MAC_MAX_TERMS) {
                        const int cnt = std::min(LSA_MAC_MAX_TERMS, n - i0);
                        const u64* cp[LSA_MAC_MAX_TERMS];
                        const u64* pp[LSA_MAC_MAX_TERMS];
                        long long cs_[LSA_MAC_MAX_TERMS], ps_[LSA_MAC_MAX_TERMS];
                        for (int i = 0; i < cnt; i++) {
                            Operand ci = i0 + i == 0 ? a : gather(c, s, nodes, i0 + i, avail, w_in);
                            Operand pi = plain_operand(pt0 + i0 + i, 0, true);
                            cp[i] = ci.ptr;
                            cs_[i] = ci.stride;
                            pp[i] = pi.ptr;
                            ps_[i] = pi.stride;
                        }
                        // the first group adds the node's partial sum, later groups continue from `out`
                        const u64* acc = i0 == 0 ? part.ptr : out;
                        launch_mac_plain(c, cnt, cp, cs_, pp, ps_, acc, i0 == 0 ? part.stride : so, out, so, m, polys_in, L, rmL, s);
                    }
                    break;
                }
                u64* tmp = temp((size_t)m * w_in);
                std::vector<int> all(rows);
                for (int i = 0; i < rows; i++) all[i] = i;
                for (int i = 0; i < n; i++) {
                    Operand ci = i == 0 ? a : gather(c, s, nodes, i, avail, w_in);
                    Operand pi = plain_operand(pt0 + i, bfv ? 1 : 0, true);
                    const u64* cptr = ci.ptr;
                    long long cstride = ci.stride;
                    if (bfv) {  // accumulate in the NTT domain, one inverse transform at the end (linear => identical residues)
                        launch_copy_rows(c, ci.ptr, ci.stride, tmp, (long long)w_in, rows, all.data(), m, s);
                        launch_ntt(c, tmp, tmp, m, (long long)w_in, rows, rmL, false, s);
                        cptr = tmp;
                        cstride = (long long)w_in;
                    }
                    for (int p = 0; p < polys_in; p++)
                        launch_muladd(c, EW_MUL, cptr + (size_t)p * L * N, pi.ptr, i == 0 ? nullptr : out + (size_t)p * L * N, so,
                                      out + (size_t)p * L * N, m, cstride, pi.stride, so, L, rmL, s);
                }
                if (bfv) launch_ntt(c, out, out, m, so, rows, rmL, true, s);
                if (with_partial) {
                    Operand part = gather(c, s, nodes, n, avail, w_in);
                    launch_elementwise(c, EW_ADD, out, part.ptr, out, m, so, part.stride, so, rows, rmL, s);
                }
                break;
            }
            default: throw Error(LSA_ERR_ARG, std::string("operation not implemented on this backend: ") + op_name(op));
        }
        for (int i = 0; i < m; i++) {
            auto d = std::make_shared<DevDatum>();
            d->slab = out_slab;
            d->ptr = out + w_out * i;
            d->polys = out_polys;
            d->level = out_lvl;
            avail[nodes[i]->output_nodes[0]->index] = d;
        }
        pending_free().push_back(out_slab);  // (cheap: shared) keeps frees off the critical path until the level ends
    }

    // Buckets of one level.  CKKS rotations of the SAME ciphertexts by different Galois elements (the frontend's rotate_cols
    // emits them for convolutions: examples/benchmark_convolution) are hoisted: one decomposition of the inputs, then only
    // the key MAC + ModDown + permutation per element (ckks_rotate_many; same residues as separate rotations).
    void run_buckets(Context& c, hipStream_t s, std::map<std::string, std::vector<ComputeNode*>>& buckets,
                     const std::vector<std::string>& order, std::unordered_map<NodeIndex, std::any>& avail) {
        std::map<std::vector<NodeIndex>, std::vector<const std::string*>> rot_groups;
        if (g.algo == ALGO_CKKS)
            for (auto& sg : order) {
                auto& nodes = buckets[sg];
                const OperationType op = nodes[0]->op();
                if (op != OperationType::ROTATE_COL && op != OperationType::ROTATE_ROW) continue;
                std::vector<NodeIndex> ins;
                for (auto* n : nodes) ins.push_back(n->input_nodes[0]->index);
                rot_groups[ins].push_back(&sg);
            }
        std::unordered_set<const std::string*> done;
        for (auto& sg : order) {
            if (done.count(&sg)) continue;
            auto& nodes = buckets[sg];
            const OperationType op = nodes[0]->op();
            std::vector<const std::string*>* group = nullptr;
            if (g.algo == ALGO_CKKS && (op == OperationType::ROTATE_COL || op == OperationType::ROTATE_ROW)) {
                std::vector<NodeIndex> ins;
                for (auto* n : nodes) ins.push_back(n->input_nodes[0]->index);
                auto& gr = rot_groups[ins];
                if (gr.size() >= 2) group = &gr;
            }
            if (!group) {
                run_gpu_bucket(c, s, nodes, avail);
                last_gpu_nodes += (int)nodes.size();
                last_gpu_batches++;
                continue;
            }
            // hoisted group: same inputs, one Galois element per member bucket
            const long long N = c.n;
            const ComputeNode* n0 = nodes[0];
            const int lvl = n0->input_nodes[0]->fhe_prop->level, L = lvl + 1, m = (int)nodes.size();
            LSA_REQUIRE(n0->input_nodes[0]->fhe_prop->degree == 1, "rotation expects a degree-1 ciphertext");
            const size_t w = (size_t)2 * L * N;
            Operand a = gather(c, s, nodes, 0, avail, w);
            std::vector<u64> els;
            std::vector<const Key*> keys;
            std::vector<u64*> outs;
            std::vector<std::shared_ptr<Slab>> slabs;
            for (const std::string* member : *group) {
                auto& mn = buckets[*member];
                const ComputeNode* r0 = mn[0];
                const DatumNode* kd = r0->input_nodes[1];
                const u64 gel = r0->op() == OperationType::ROTATE_ROW ? 2 * (u64)c.n - 1
                                                                      : (kd->fhe_prop->p ? kd->fhe_prop->p->galois_element : 0);
                LSA_REQUIRE(gel != 0, "Galois element missing on the key datum");
                els.push_back(gel);
                keys.push_back(&std::any_cast<KeyP>(avail.at(kd->index))->key);
                slabs.push_back(dslab(w * m));
                outs.push_back(slabs.back()->ptr);
            }
            ckks_rotate_many(c, lvl, a.ptr, (int)els.size(), els.data(), keys.data(), outs.data(), m, a.stride, (long long)w, s);
            for (size_t gi = 0; gi < group->size(); gi++) {
                auto& mn = buckets[*(*group)[gi]];
                for (int i = 0; i < m; i++) {
                    auto d = std::make_shared<DevDatum>();
                    d->slab = slabs[gi];
                    d->ptr = outs[gi] + w * i;
                    d->polys = 2;
                    d->level = lvl;
                    avail[mn[i]->output_nodes[0]->index] = d;
                }
                pending_free().push_back(slabs[gi]);
                last_gpu_nodes += m;
                done.insert((*group)[gi]);
            }
            last_gpu_batches++;
        }
    }

    // ---------------------------------------------------------------- CPU-side nodes (export / import / custom)
    void run_cpu_nodes(const std::vector<ComputeNode*>& nodes, std::unordered_map<NodeIndex, std::any>& avail,
                       const std::unordered_map<NodeIndex, void*>& out_handles) {
        if (nodes.empty()) return;
        std::vector<std::any> outputs(nodes.size());
        std::vector<std::string> errors(nodes.size());
        const int hw = (int)std::thread::hardware_concurrency();
        const int nthreads = std::max(1, std::min({(int)nodes.size(), std::min(16, hw > 0 ? hw : 1) - 2, 14}));
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= nodes.size()) return;
                const ComputeNode* node = nodes[i];
                try {
                    if (!node->executor) throw std::runtime_error("no executor bound for CPU node '" + node->id + "'");
                    std::unordered_map<NodeIndex, std::any> ins;
                    for (auto* in : node->input_nodes) ins[in->index] = avail.at(in->index);
                    ExecutionContext ec;
                    if (node->op() == OperationType::IMPORT_FROM_ABI) {
                        auto it = out_handles.find(node->output_nodes[0]->index);
                        if (it != out_handles.end()) ec.other_args.push_back(it->second);
                    }
                    node->executor(ec, ins, outputs[i], *node);
                } catch (const std::exception& e) {
                    errors[i] = e.what()[0] ? e.what() : "executor failed";
                } catch (...) {
                    errors[i] = "executor failed with a non-standard exception";
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; t++) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
        for (size_t i = 0; i < nodes.size(); i++) {
            if (!errors[i].empty()) throw Error(LSA_ERR_INTERNAL, "node '" + nodes[i]->id + "': " + errors[i]);
            avail[nodes[i]->output_nodes[0]->index] = outputs[i];
        }
    }

    // ---------------------------------------------------------------- run
    // the devices a run is spread over: lsa_task_set_devices' list, else the caller's gpu_device (-1: every visible device)
    std::vector<int> run_devices(int device) {
        if (device >= 0 && devices_.empty()) return {device};
        if (!devices_.empty()) return devices_;
        int n = 0;
        LSA_HIP(hipGetDeviceCount(&n));
        LSA_REQUIRE(n >= 1, "no HIP device");
        std::vector<int> all(n);
        for (int i = 0; i < n; i++) all[i] = i;
        return all;
    }

    void run(CArgument* in_args, uint64_t n_in, CArgument* out_args, uint64_t n_out, progress_callback_t cb, void* user,
             int device) {
        std::lock_guard<std::mutex> run_lock(run_mu);   // one run at a time per handle (the graph state is shared)
        const auto t_start = std::chrono::steady_clock::now();
        const std::vector<int> devs = run_devices(device);
        if (!getenv("LSA_NO_PIPELINE") && planned_shards_ != (int)devs.size()) plan_pipeline((int)devs.size());   // chunk count follows the shard count
        // shards: one device + two lanes each; chunks of independent subgraphs are dealt out to them (shard_plan.h)
        const ShardPlan plan = plan_shards(chunk_levels.empty() ? std::vector<int>{devs[0]} : devs, (int)chunk_levels.size());
        const int up_dev = plan.upload_device();
        // contexts, streams and the per-lane temporaries lists exist before any worker thread looks them up
        for (const ShardPlan::Shard& sh : plan.shards)
            for (int l = 0; l < 2; l++) {
                pending_free_[LanePools::key(sh.device, sh.lane0 + l)];
                if (!chunk_levels.empty() || l == 0) context(sh.device, sh.lane0 + l);
            }
        tls_exec.device = up_dev;
        tls_exec.lane = 0;
        Context& c = context(up_dev, 0);
        hipStream_t s = streams.at(LanePools::key(up_dev, 0));
        // inputs: flatten every CArgument's handle array, consume in mega_ag.inputs order; all Galois-key data nodes share
        // the first Galois handle (cpu_task_utils.h:235-319)
        std::vector<void*> handles;
        for (uint64_t i = 0; i < n_in; i++) {
            void** arr = (void**)in_args[i].data;
            for (int j = 0; j < in_args[i].size; j++) handles.push_back(arr[j]);
        }
        std::unordered_map<NodeIndex, std::any> avail;
        std::shared_ptr<void> glk_shared;
        size_t hi = 0;
        for (NodeIndex idx : g.inputs) {
            const DatumNode& d = g.data.at(idx);
            if (d.datum_type == TYPE_GALOIS_KEY) {
                if (!glk_shared) {
                    LSA_REQUIRE(hi < handles.size(), "not enough input handles for the task's inputs");
                    glk_shared = std::shared_ptr<void>(handles[hi++], [](void*) {});
                }
                avail[idx] = glk_shared;
            } else {
                LSA_REQUIRE(hi < handles.size(), "not enough input handles for the task's inputs");
                avail[idx] = std::shared_ptr<void>(handles[hi++], [](void*) {});
            }
        }
        std::unordered_map<NodeIndex, void*> out_handles;
        size_t oi = 0;
        for (uint64_t i = 0; i < n_out; i++) {
            void** arr = (void**)out_args[i].data;
            for (int j = 0; j < out_args[i].size; j++) {
                LSA_REQUIRE(oi < g.outputs.size(), "more output handles than task outputs");
                out_handles[g.outputs[oi++]] = arr[j];
            }
        }
        LSA_REQUIRE(oi == g.outputs.size(), "fewer output handles than task outputs");
        run_out_handles = &out_handles;
        struct ClearOut {
            fhe_task_handle_st* h;
            ~ClearOut() { h->run_out_handles = nullptr; }
        } clear_out{this};
        last_direct_loads = 0;
        last_direct_stores = 0;

        // remaining-consumer counts: device data is dropped as soon as its last consumer has been enqueued
        std::unordered_map<NodeIndex, int> refs;
        for (auto& kv : g.data) refs[kv.first] = (int)kv.second.successors.size();
        const int total = (int)g.computes.size();
        int completed = 0;
        auto last_cb = std::chrono::steady_clock::now() - std::chrono::seconds(1);
        last_gpu_nodes = 0;
        last_gpu_batches = 0;
        last_shards = 1;
        last_chunks = (int)chunk_levels.size();
        last_key_peer_copies = 0;
        last_key_uploads = last_key_hits = 0;

        const bool trace = getenv("LSA_TASK_TRACE") != nullptr;
        auto tick = [&]() { return std::chrono::steady_clock::now(); };
        auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
            return std::chrono::duration<double, std::milli>(tick() - t0).count();
        };
        std::mutex progress_mu;   // the callback fires from whichever thread finished something (wrapper.h:39-40)
        auto progress = [&](size_t nodes_done) {
            std::lock_guard<std::mutex> lk(progress_mu);
            completed += (int)nodes_done;
            const auto now = std::chrono::steady_clock::now();
            if (cb && (completed == total || now - last_cb >= std::chrono::milliseconds(100))) {
                cb(completed, total, user);
                last_cb = now;
            }
        };
        using Avail = std::unordered_map<NodeIndex, std::any>;
        using Refs = std::unordered_map<NodeIndex, int>;
        auto release_inputs = [](const std::vector<ComputeNode*>& level, Avail& av, Refs& rf) {
            for (ComputeNode* n : level)
                for (auto* in : n->input_nodes)
                    if (--rf[in->index] <= 0 && !in->is_input && !in->is_output) av.erase(in->index);
        };
        struct Split {
            std::vector<ComputeNode*> cpu, loads, stores;
            std::map<std::string, std::vector<ComputeNode*>> buckets;
            std::vector<std::string> bucket_order;
        };
        auto split = [&](const std::vector<ComputeNode*>& level) {
            Split sp;
            for (ComputeNode* n : level) {
                if (n->on_cpu) sp.cpu.push_back(n);
                else if (n->op() == OperationType::LOAD_TO_BACKEND) sp.loads.push_back(n);
                else if (n->op() == OperationType::STORE_FROM_BACKEND) sp.stores.push_back(n);
                else {
                    const std::string sg = signature(n);
                    if (!sp.buckets.count(sg)) sp.bucket_order.push_back(sg);
                    sp.buckets[sg].push_back(n);
                }
            }
            return sp;
        };
        // one level, everything in order on one lane, host-synchronous at the copies (graphs that are not pipelined)
        auto run_level_sync = [&](const std::vector<ComputeNode*>& level) {
            if (level.empty()) return;
            Split sp = split(level);
            auto t0 = tick();
            if (!sp.loads.empty()) {
                auto keep = run_loads(c, s, sp.loads, avail);
                LSA_HIP(hipStreamSynchronize(s));
            }
            const double t_load = ms_since(t0);
            t0 = tick();
            run_buckets(c, s, sp.buckets, sp.bucket_order, avail);
            if (trace && !sp.bucket_order.empty()) LSA_HIP(hipStreamSynchronize(s));
            const double t_gpu = ms_since(t0);
            t0 = tick();
            if (!sp.stores.empty()) run_stores(c, s, sp.stores, avail);
            const double t_store = ms_since(t0);
            t0 = tick();
            run_cpu_nodes(sp.cpu, avail, out_handles);
            const double t_cpu = ms_since(t0);
            if (trace)
                fprintf(stderr, "[lsa task] level: %zu nodes  load %.2f ms  gpu %.2f ms  store %.2f ms  cpu %.2f ms\n",
                        level.size(), t_load, t_gpu, t_store, t_cpu);
            release_inputs(level, avail, refs);
            if (!pending_free().empty()) {  // slabs whose last reference is dropped here are freed after their readers ran
                LSA_HIP(hipStreamSynchronize(s));
                pending_free().clear();
            }
            progress(level.size());
        };

        if (chunk_levels.empty()) {
            for (auto& level : levels) run_level_sync(level);
        } else {
            // shared evaluation keys first: exported, uploaded and converted ONCE, on the first device of the list
            for (auto& level : shared_levels) run_level_sync(level);
            LSA_HIP(hipStreamSynchronize(s));
            last_shards = (int)plan.shards.size();
            // ... then copied device-to-device to every other distinct device (shards of one device share its copy)
            std::map<int, Avail> dev_avail;
            dev_avail[up_dev] = avail;
            if (plan.key_devices.size() > 1) {
                std::vector<NodeIndex> key_idx;
                std::vector<KeyP> key_src;
                std::vector<void*> src;
                std::vector<size_t> bytes;
                for (auto& kv : avail)
                    if (auto* kp = std::any_cast<KeyP>(&kv.second)) {
                        key_idx.push_back(kv.first);
                        key_src.push_back(*kp);
                        src.push_back((*kp)->key.data);
                        bytes.push_back((*kp)->slab->words * sizeof(u64));
                    }
                struct PeerOps {
                    fhe_task_handle_st* self;
                    const ShardPlan* plan;
                    std::map<void*, std::shared_ptr<Slab>> slabs;
                    int copies = 0;
                    int lane0(int d) const {
                        for (auto& sh : plan->shards)
                            if (sh.device == d) return sh.lane0;
                        return 0;
                    }
                    void* alloc(int d, size_t nbytes) {
                        auto sl = std::make_shared<Slab>(self->pools.device_pool(d, lane0(d)), nbytes / sizeof(u64));
                        slabs[sl->ptr] = sl;
                        return sl->ptr;
                    }
                    void peer_copy(void* dst, int dd, const void* sp, int sd, size_t nbytes) {
                        LSA_HIP(hipSetDevice(dd));
                        int can = 0;
                        if (hipDeviceCanAccessPeer(&can, dd, sd) == hipSuccess && can) {
                            const hipError_t e = hipDeviceEnablePeerAccess(sd, 0);   // direct xGMI copies; already enabled is fine
                            if (e != hipSuccess) (void)hipGetLastError();
                        }
                        LSA_HIP(hipMemcpyPeerAsync(dst, dd, sp, sd, nbytes, self->streams.at(LanePools::key(dd, lane0(dd)))));
                        copies++;
                    }
                } ops{this, &plan, {}, 0};
                // a peer copy made by an earlier run is still valid while the upload device's entry it was made from is
                // (same handle and fingerprint): those keys are taken from the cache, the others are copied now
                std::vector<char> cached(key_idx.size() * plan.key_devices.size(), 0);
                if (keep_keys)
                    for (size_t i = 1; i < plan.key_devices.size(); i++)
                        for (size_t k = 0; k < key_idx.size(); k++) {
                            auto up = key_cache.find({up_dev, key_idx[k]});
                            auto pe = key_cache.find({plan.key_devices[i], key_idx[k]});
                            cached[i * key_idx.size() + k] = up != key_cache.end() && pe != key_cache.end() && up->second.key == key_src[k] &&
                                                             pe->second.handle == up->second.handle && pe->second.fingerprint == up->second.fingerprint;
                        }
                for (size_t i = 1; i < plan.key_devices.size(); i++) {
                    const int d = plan.key_devices[i];
                    std::vector<void*> src_d;
                    std::vector<size_t> bytes_d;
                    std::vector<size_t> which;
                    for (size_t k = 0; k < key_idx.size(); k++)
                        if (!cached[i * key_idx.size() + k]) {
                            src_d.push_back(src[k]);
                            bytes_d.push_back(bytes[k]);
                            which.push_back(k);
                        }
                    ShardPlan one = plan;   // fan-out of the missing keys to this device only
                    one.key_devices = {up_dev, d};
                    auto table = fan_out_keys(one, src_d, bytes_d, ops);
                    LSA_HIP(hipSetDevice(d));
                    LSA_HIP(hipStreamSynchronize(streams.at(LanePools::key(d, ops.lane0(d)))));
                    Avail av = avail;
                    for (size_t k = 0; k < key_idx.size(); k++)
                        if (cached[i * key_idx.size() + k]) av[key_idx[k]] = key_cache.at({d, key_idx[k]}).key;
                    for (size_t j = 0; j < which.size(); j++) {
                        const size_t k = which[j];
                        auto dk = std::make_shared<DevKey>();
                        dk->slab = ops.slabs.at(table.at(d)[j]);
                        dk->key = key_src[k]->key;
                        dk->key.data = (u64*)table.at(d)[j];
                        dk->key.owned = false;
                        if (key_src[k]->key.fp)   // the double copy travelled in the same slab
                            dk->key.fp = reinterpret_cast<const double*>(dk->key.data + (reinterpret_cast<const u64*>(key_src[k]->key.fp) - key_src[k]->key.data));
                        av[key_idx[k]] = dk;
                        if (keep_keys) {
                            auto up = key_cache.find({up_dev, key_idx[k]});
                            if (up != key_cache.end() && up->second.key == key_src[k])
                                key_cache[{d, key_idx[k]}] = CachedKey{up->second.handle, up->second.fingerprint, dk};
                        }
                    }
                    dev_avail[d] = std::move(av);
                }
                last_key_peer_copies = ops.copies;
                LSA_HIP(hipSetDevice(up_dev));
            }

            struct InFlight {
                int chunk = -1, lane = 0;
                size_t resume_level = 0;
                StoreJob job;
                std::vector<std::shared_ptr<Slab>> keep;
            };
            // One shard: its chunks in order, alternating its two lanes (stream + context + device-buffer pool each).  While
            // one lane's chunk computes and copies its results out, the other lane's chunk is staged and copied in.
            auto run_shard = [&](int si) {
                const ShardPlan::Shard sh = plan.shards[(size_t)si];
                tls_exec.device = sh.device;
                tls_exec.lane = sh.lane0;
                Context* lane_ctx[2] = {&context(sh.device, sh.lane0), &context(sh.device, sh.lane0 + 1)};
                hipStream_t lane_stream[2] = {streams.at(LanePools::key(sh.device, sh.lane0)), streams.at(LanePools::key(sh.device, sh.lane0 + 1))};
                Avail my_avail = dev_avail.at(sh.device);   // the shard's own view: its chunks' data + the device's keys
                Refs my_refs = refs;
                // finish(): wait for the chunk's stream, wrap the results into C structs, run the import executors.  It runs on
                // its own thread while the shard's thread stages and enqueues the next chunk, and touches no shared container:
                // the structs live in a local map (only the import nodes read them), the per-lane temporaries are released by
                // this thread while the shard's thread is, by construction, busy with the OTHER lane.
                auto finish = [&](InFlight* f) {
                    const int li = f->lane;
                    Context& lc = *lane_ctx[li];
                    hipStream_t ls = lane_stream[li];
                    lc.use_device();
                    auto t0 = tick();
                    LSA_HIP(hipStreamSynchronize(ls));
                    const double t_wait = ms_since(t0);
                    pending_free_.at(LanePools::key(sh.device, sh.lane0 + li)).clear();
                    f->keep.clear();
                    auto& cl = chunk_levels[f->chunk];
                    t0 = tick();
                    Avail local;
                    stores_finish(lc, f->job, local);
                    f->job = StoreJob{};
                    progress(cl[f->resume_level].size());
                    for (size_t l = f->resume_level + 1; l < cl.size(); l++) {
                        if (cl[l].empty()) continue;
                        Split sp = split(cl[l]);
                        LSA_REQUIRE(sp.loads.empty() && sp.stores.empty() && sp.bucket_order.empty(), "pipeline plan violated");
                        run_cpu_nodes(sp.cpu, local, out_handles);
                        progress(cl[l].size());
                    }
                    if (trace) fprintf(stderr, "[lsa task] chunk %d device %d lane %d: waited %.2f ms, import %.2f ms\n", f->chunk, sh.device, sh.lane0 + li, t_wait, ms_since(t0));
                    f->chunk = -1;
                };
                InFlight fly[2];
                std::future<void> done[2];
                auto join = [&](int li) {
                    if (done[li].valid()) done[li].get();   // rethrows what the finisher threw
                };
                try {
                    int mine = 0;
                    for (size_t ch = 0; ch < chunk_levels.size(); ch++) {
                        if (plan.chunk_shard[ch] != si) continue;
                        const int li = mine++ & 1;
                        join(li);                   // the lane's previous chunk (two chunks in flight at most per shard)
                        tls_exec.lane = sh.lane0 + li;
                        Context& lc = *lane_ctx[li];
                        hipStream_t ls = lane_stream[li];
                        lc.use_device();
                        InFlight& f = fly[li];
                        f.chunk = (int)ch;
                        f.lane = li;
                        auto& cl = chunk_levels[ch];
                        auto t0 = tick();
                        for (size_t l = 0; l < cl.size(); l++) {
                            if (cl[l].empty()) continue;
                            Split sp = split(cl[l]);
                            if (!sp.stores.empty()) {     // copy-out enqueued; the rest of the chunk happens in finish()
                                f.job = stores_enqueue(lc, ls, sp.stores, my_avail);   // holds the device data alive until the copy ran
                                f.resume_level = l;
                                release_inputs(cl[l], my_avail, my_refs);
                                for (size_t l2 = l + 1; l2 < cl.size(); l2++) release_inputs(cl[l2], my_avail, my_refs);
                                break;
                            }
                            if (!sp.cpu.empty()) run_cpu_nodes(sp.cpu, my_avail, out_handles);   // export executors
                            if (!sp.loads.empty()) f.keep.push_back(run_loads(lc, ls, sp.loads, my_avail));
                            run_buckets(lc, ls, sp.buckets, sp.bucket_order, my_avail);
                            release_inputs(cl[l], my_avail, my_refs);
                            progress(cl[l].size());
                        }
                        if (trace) fprintf(stderr, "[lsa task] chunk %zu device %d lane %d: enqueued in %.2f ms\n", ch, sh.device, sh.lane0 + li, ms_since(t0));
                        done[li] = std::async(std::launch::async, finish, &f);
                    }
                    join(0);
                    join(1);
                } catch (...) {
                    for (int li = 0; li < 2; li++)     // never leave a finisher running on our stack frame
                        if (done[li].valid()) {
                            try {
                                done[li].get();
                            } catch (...) {
                            }
                        }
                    throw;
                }
            };
            if (plan.shards.size() == 1) {
                run_shard(0);
            } else {
                // one host thread per shard; the first failure is reported once every shard has stopped
                std::vector<std::exception_ptr> errs(plan.shards.size());
                std::vector<std::thread> workers;
                for (size_t si = 0; si < plan.shards.size(); si++)
                    workers.emplace_back([&, si]() {
                        try {
                            run_shard((int)si);
                        } catch (...) {
                            errs[si] = std::current_exception();
                        }
                    });
                for (auto& w : workers) w.join();
                for (auto& e : errs)
                    if (e) {
                        tls_exec.device = up_dev;
                        tls_exec.lane = 0;
                        std::rethrow_exception(e);
                    }
            }
            tls_exec.device = up_dev;
            tls_exec.lane = 0;
            c.use_device();
        }
        LSA_HIP(hipStreamSynchronize(s));
        last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    }
};

// ------------------------------------------------------------------------------------------------ native front-end
namespace {

template <typename T> std::shared_ptr<T> owned_struct(T* p, void (*fin)(T*)) {
    return std::shared_ptr<T>(p, [fin](T* q) {
        fin(q);
        free(q);
    });
}

void view_polynomial(CPolynomial* poly, uint64_t* base, int limbs, int n) {  // components point INTO the caller's buffer
    poly->n_component = limbs;
    poly->components = (CComponent*)malloc(sizeof(CComponent) * (size_t)limbs);
    for (int j = 0; j < limbs; j++) {
        poly->components[j].n = n;
        poly->components[j].data = base + (size_t)j * n;
    }
}

void fill_ksk_view(CKeySwitchKey* dst, const lsa_host_kskey* k) {
    const int comp = k->level + 1 + k->n_special;
    const int beta = (k->level + 1 + k->n_special - 1) / k->n_special;
    dst->n_public_key = beta;
    dst->public_keys = (CPublicKey*)malloc(sizeof(CPublicKey) * (size_t)beta);
    for (int d = 0; d < beta; d++) {
        CPublicKey& pk = dst->public_keys[d];
        pk.level = k->level;
        pk.degree = 1;
        pk.polys = (CPolynomial*)malloc(sizeof(CPolynomial) * 2);
        for (int h = 0; h < 2; h++) view_polynomial(&pk.polys[h], k->data + ((size_t)(d * 2 + h) * comp) * k->n, comp, k->n);
    }
}
void free_ksk_view(CKeySwitchKey* k) {
    for (int d = 0; d < k->n_public_key; d++) {
        for (int h = 0; h < 2; h++) free(k->public_keys[d].polys[h].components);
        free(k->public_keys[d].polys);
    }
    free(k->public_keys);
}

// handle -> C struct, zero-copy: the structs only index the caller's limb buffers (no per-limb malloc+copy)
ExecutorFunc frontend_export() {
    return [](ExecutionContext&, const std::unordered_map<NodeIndex, std::any>& inputs, std::any& output, const ComputeNode& self) {
        const DatumNode* in = self.input_nodes[0];
        void* h = std::any_cast<std::shared_ptr<void>>(inputs.at(in->index)).get();
        if (!h) throw std::runtime_error("null input handle for '" + in->id + "'");
        switch (in->datum_type) {
            case TYPE_CIPHERTEXT: {
                auto* src = (lsa_host_ciphertext*)h;
                if (src->level != in->fhe_prop->level || src->degree != in->fhe_prop->degree)
                    throw std::runtime_error("ciphertext '" + in->id + "' has level/degree " + std::to_string(src->level) + "/" +
                                             std::to_string(src->degree) + ", task expects " + std::to_string(in->fhe_prop->level) +
                                             "/" + std::to_string(in->fhe_prop->degree));
                auto* ct = (CCiphertext*)malloc(sizeof(CCiphertext));
                ct->level = src->level;
                ct->degree = src->degree;
                ct->polys = (CPolynomial*)malloc(sizeof(CPolynomial) * (size_t)(src->degree + 1));
                for (int p = 0; p <= src->degree; p++)
                    view_polynomial(&ct->polys[p], src->data + (size_t)p * (src->level + 1) * src->n, src->level + 1, src->n);
                output = owned_struct<CCiphertext>(ct, [](CCiphertext* c) {
                    for (int p = 0; p <= c->degree; p++) free(c->polys[p].components);
                    free(c->polys);
                });
                break;
            }
            case TYPE_PLAINTEXT: {
                auto* src = (lsa_host_plaintext*)h;
                auto* pt = (CPlaintext*)malloc(sizeof(CPlaintext));
                pt->level = src->level;
                view_polynomial(&pt->poly, src->data, src->level + 1, src->n);
                output = owned_struct<CPlaintext>(pt, [](CPlaintext* p) { free(p->poly.components); });
                break;
            }
            case TYPE_RELIN_KEY:
            case TYPE_SWITCH_KEY: {
                auto* src = (lsa_host_kskey*)h;
                auto* k = (CKeySwitchKey*)malloc(sizeof(CKeySwitchKey));
                fill_ksk_view(k, src);
                if (in->datum_type == TYPE_RELIN_KEY) output = std::shared_ptr<CRelinKey>(k, [](CRelinKey* q) { free_ksk_view(q); free(q); });
                else output = std::shared_ptr<CKeySwitchKey>(k, [](CKeySwitchKey* q) { free_ksk_view(q); free(q); });
                break;
            }
            case TYPE_GALOIS_KEY: {
                auto* src = (lsa_host_galois_key*)h;
                const uint64_t want = in->fhe_prop->p ? in->fhe_prop->p->galois_element : 0;
                const lsa_host_kskey* found = nullptr;
                for (int i = 0; i < src->n_keys; i++)
                    if (src->galois_elements[i] == want) found = &src->keys[i];
                if (!found) throw std::runtime_error("The rotation key glk_" + std::to_string(want) + " is not prepared");
                auto* gk = (CGaloisKey*)malloc(sizeof(CGaloisKey));
                gk->n_key_switch_key = 1;
                gk->galois_elements = (uint64_t*)malloc(sizeof(uint64_t));
                gk->galois_elements[0] = want;
                gk->key_switch_keys = (CKeySwitchKey*)malloc(sizeof(CKeySwitchKey));
                fill_ksk_view(&gk->key_switch_keys[0], found);
                output = std::shared_ptr<CGaloisKey>(gk, [](CGaloisKey* q) {
                    free_ksk_view(&q->key_switch_keys[0]);
                    free(q->key_switch_keys);
                    free(q->galois_elements);
                    free(q);
                });
                break;
            }
            case TYPE_CUSTOM:
                // custom input data (e.g. a message a custom "encode" node turns into a plaintext) never reaches the device: the
                // opaque caller handle is handed through to the custom executors that consume it
                // (cxx_abi_bridge_executors.h:212-220 does the same with CustomData)
                output = inputs.at(in->index);
                break;
            default:   // an unknown or garbled type must not reach downstream any_casts (the reference throws here too, ibid.)
                throw std::runtime_error("Unsupported data type " + std::to_string((int)in->datum_type) + " for input '" + in->id + "' (datum " +
                                         std::to_string(in->index) + ")");
        }
    };
}

// an intermediate ciphertext handle owned by the run (a device result that a custom CPU node consumes, or a custom node's
// input for the next device stage): header + limbs in one allocation, released with the last reference
struct OwnedHostCiphertext {
    lsa_host_ciphertext h;
    std::vector<uint64_t> limbs;
};
std::shared_ptr<void> new_intermediate_ciphertext(int degree, int level, int n) {
    auto o = std::make_shared<OwnedHostCiphertext>();
    o->limbs.resize((size_t)(degree + 1) * (level + 1) * n);
    o->h.level = level;
    o->h.degree = degree;
    o->h.n = n;
    o->h.data = o->limbs.data();
    return std::shared_ptr<void>(o, &o->h);   // aliasing: callers see the lsa_host_ciphertext, the block stays alive
}

// C struct -> pre-allocated output handle (other_args[0], as in gpu_wrapper.cu:354-365); without one (a device result that
// feeds a custom CPU node) -> a fresh intermediate handle (cxx_abi_bridge_executors.h:428-431).  A custom node's own
// output that is a task output arrives as a handle already and is copied into the caller's.
ExecutorFunc frontend_import() {
    return [](ExecutionContext& ctx, const std::unordered_map<NodeIndex, std::any>& inputs, std::any& output, const ComputeNode& self) {
        const DatumNode* in = self.input_nodes[0];
        const std::any& src_any = inputs.at(in->index);
        lsa_host_ciphertext* dst = nullptr;
        std::shared_ptr<void> owned;
        if (!ctx.other_args.empty()) {
            dst = (lsa_host_ciphertext*)std::any_cast<void*>(ctx.other_args[0]);
            if (!dst || !dst->data) throw std::runtime_error("import: null output handle");
        }
        if (auto* hp = std::any_cast<std::shared_ptr<void>>(&src_any)) {   // produced by a custom node: already a handle
            auto* src = (lsa_host_ciphertext*)hp->get();
            if (!src || !src->data) throw std::runtime_error("import: custom node '" + in->id + "' produced no ciphertext handle");
            if (!dst) {
                output = *hp;
                return;
            }
            if (dst->level != src->level || dst->degree != src->degree || dst->n != src->n)
                throw std::runtime_error("output ciphertext '" + self.output_nodes[0]->id + "' was allocated at level/degree " +
                                         std::to_string(dst->level) + "/" + std::to_string(dst->degree) + ", result has " +
                                         std::to_string(src->level) + "/" + std::to_string(src->degree));
            memcpy(dst->data, src->data, sizeof(uint64_t) * (size_t)(src->degree + 1) * (src->level + 1) * src->n);
            output = std::shared_ptr<void>(dst, [](void*) {});
            return;
        }
        auto ct = std::any_cast<std::shared_ptr<CCiphertext>>(src_any);
        const int n = ct->polys[0].components[0].n;
        if (!dst) {
            owned = new_intermediate_ciphertext(ct->degree, ct->level, n);
            dst = (lsa_host_ciphertext*)owned.get();
        }
        if (dst->level != ct->level || dst->degree != ct->degree)
            throw std::runtime_error("output ciphertext '" + self.output_nodes[0]->id + "' was allocated at level/degree " +
                                     std::to_string(dst->level) + "/" + std::to_string(dst->degree) + ", result has " +
                                     std::to_string(ct->level) + "/" + std::to_string(ct->degree));
        for (int p = 0; p <= ct->degree; p++)
            for (int j = 0; j <= ct->level; j++) {
                uint64_t* to = dst->data + ((size_t)p * (ct->level + 1) + j) * n;
                if (to != ct->polys[p].components[j].data)   // (equal: the backend wrote the result straight into this handle's pinned buffer)
                    memcpy(to, ct->polys[p].components[j].data, sizeof(uint64_t) * (size_t)n);
            }
        output = owned ? owned : std::shared_ptr<void>(dst, [](void*) {});
    };
}

template <typename F> int task_guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code ? e.code : LSA_ERR_INTERNAL;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return LSA_ERR_INTERNAL;
    } catch (...) {
        set_last_error("unknown error");
        return LSA_ERR_INTERNAL;
    }
}

}  // namespace

extern "C" {

fhe_task_handle create_fhe_gpu_task(const char* project_path) {
    fhe_task_handle h = nullptr;
    task_guard([&] {
        LSA_REQUIRE(project_path != nullptr, "null project path");
        h = new fhe_task_handle_st(project_path);
    });
    return h;
}

void release_fhe_gpu_task(fhe_task_handle handle) {
    task_guard([&] { delete handle; });
}

void bind_gpu_task_abi_bridge_executors(fhe_task_handle handle, void* abi_export_executor, void* abi_import_executor) {
    task_guard([&] {
        LSA_REQUIRE(handle && abi_export_executor && abi_import_executor, "null argument");
        // copied by value, the caller may free its std::function objects afterwards (gpu_wrapper.cu:492-497)
        handle->g.bind_bridge_executors(*reinterpret_cast<ExecutorFunc*>(abi_export_executor),
                                        *reinterpret_cast<ExecutorFunc*>(abi_import_executor));
        handle->native_frontend = false;   // output handles are the caller's own objects: never interpreted here
    });
}

void bind_gpu_task_custom_executors(fhe_task_handle handle, const char** custom_types, void** executors, uint64_t n_executors) {
    task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        std::unordered_map<std::string, ExecutorFunc> m;
        for (uint64_t i = 0; i < n_executors; i++) m[custom_types[i]] = *reinterpret_cast<ExecutorFunc*>(executors[i]);
        handle->g.bind_custom_executors(m);
    });
}

int run_fhe_gpu_task(fhe_task_handle handle, CArgument* input_args, uint64_t n_in_args, CArgument* output_args,
                     uint64_t n_out_args, progress_callback_t progress_cb, void* user_data, int gpu_device) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        handle->run(input_args, n_in_args, output_args, n_out_args, progress_cb, user_data, gpu_device);
    });
}

int lsa_frontend_bind(fhe_task_handle handle) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        handle->g.bind_bridge_executors(frontend_export(), frontend_import());
        handle->native_frontend = true;
    });
}

int lsa_task_set_devices(fhe_task_handle handle, const int* device_ids, int n_devices) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr && n_devices >= 0 && (n_devices == 0 || device_ids != nullptr), "bad device list");
        std::lock_guard<std::mutex> lk(handle->run_mu);
        std::vector<int> ids(device_ids, device_ids + n_devices);
        try {
            if (!ids.empty()) (void)plan_shards(ids, 0);   // validates the list
        } catch (const std::invalid_argument& e) {
            throw Error(LSA_ERR_ARG, e.what());
        }
        handle->devices_ = ids;
    });
}

int lsa_host_register(void* ptr, size_t bytes) {
    return task_guard([&] {
        LSA_REQUIRE(ptr != nullptr && bytes > 0, "null range");
        LSA_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
        std::lock_guard<std::mutex> lk(host_registry().mu);
        host_registry().ranges[(uintptr_t)ptr] = bytes;
    });
}

int lsa_host_unregister(void* ptr) {
    return task_guard([&] {
        {
            std::lock_guard<std::mutex> lk(host_registry().mu);
            LSA_REQUIRE(host_registry().ranges.erase((uintptr_t)ptr) == 1, "range was not registered");
        }
        LSA_HIP(hipHostUnregister(ptr));
    });
}

// pinned memory allocated FOR the caller (hipHostMalloc: the DMA engines reach it at full PCIe rate; memory pinned in place
// with lsa_host_register measured slower than the staged path on MI355X hosts, profiles/r03/t2_zero_copy_ab.log)
int lsa_host_alloc(size_t bytes, void** out) {
    return task_guard([&] {
        LSA_REQUIRE(out != nullptr && bytes > 0, "null argument");
        void* p = nullptr;
        LSA_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        std::lock_guard<std::mutex> lk(host_registry().mu);
        host_registry().ranges[(uintptr_t)p] = bytes;
        *out = p;
    });
}

int lsa_host_free(void* ptr) {
    return task_guard([&] {
        {
            std::lock_guard<std::mutex> lk(host_registry().mu);
            LSA_REQUIRE(host_registry().ranges.erase((uintptr_t)ptr) == 1, "not an lsa_host_alloc block");
        }
        LSA_HIP(hipHostFree(ptr));
    });
}

int lsa_task_last_run_direct(fhe_task_handle handle, int* loads, int* stores) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        if (loads) *loads = handle->last_direct_loads;
        if (stores) *stores = handle->last_direct_stores;
    });
}

int lsa_task_drop_keys(fhe_task_handle handle) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        std::lock_guard<std::mutex> lk(handle->run_mu);
        handle->key_cache.clear();   // the device copies return to their pools
    });
}

int lsa_task_last_run_keys(fhe_task_handle handle, int* uploaded, int* reused) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        if (uploaded) *uploaded = handle->last_key_uploads;
        if (reused) *reused = handle->last_key_hits;
    });
}

int lsa_task_last_run_shards(fhe_task_handle handle, int* n_shards, int* n_chunks, int* key_peer_copies) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        if (n_shards) *n_shards = handle->last_shards;
        if (n_chunks) *n_chunks = handle->last_chunks;
        if (key_peer_copies) *key_peer_copies = handle->last_key_peer_copies;
    });
}

int lsa_task_trim_pools(fhe_task_handle handle) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        std::lock_guard<std::mutex> lk(handle->run_mu);
        for (auto& kv : handle->pending_free_) kv.second.clear();
        handle->pools.trim_all();
    });
}

int lsa_task_counts(fhe_task_handle handle, int* n_data, int* n_compute, int* n_inputs, int* n_outputs) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        if (n_data) *n_data = (int)handle->g.data.size();
        if (n_compute) *n_compute = (int)handle->g.computes.size();
        if (n_inputs) *n_inputs = (int)handle->g.inputs.size();
        if (n_outputs) *n_outputs = (int)handle->g.outputs.size();
    });
}

int lsa_task_last_run_stats(fhe_task_handle handle, int* gpu_nodes, int* gpu_batches, double* run_ms) {
    return task_guard([&] {
        LSA_REQUIRE(handle != nullptr, "null task");
        if (gpu_nodes) *gpu_nodes = handle->last_gpu_nodes;
        if (gpu_batches) *gpu_batches = handle->last_gpu_batches;
        if (run_ms) *run_ms = handle->last_ms;
    });
}

}  // extern "C"
