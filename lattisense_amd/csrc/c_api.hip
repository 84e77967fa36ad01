// c_api.hip — extern "C" operator layer (include/lattisense_amd.h).  Every entry point converts C++ exceptions into
// error codes: nothing throws across the C boundary (the reference does, SURVEY §8b "Errors").
#include "build_flags.h"
#include "lsa_internal.h"

namespace lsa {
const char* kernels_build_flags();   // kernels.hip / context.hip: what THOSE translation units were compiled with
const char* context_build_flags();
void ckks_mult(Context&, int, const u64*, const u64*, u64*, int, long long, long long, long long, hipStream_t);
void ckks_relin(Context&, int, const u64*, const Key&, u64*, int, long long, long long, hipStream_t);
void ckks_rescale(Context&, int, int, const u64*, u64*, int, long long, long long, hipStream_t);
void ckks_rotate(Context&, int, const u64*, u64, const Key&, u64*, int, long long, long long, hipStream_t);
void ckks_rotate_many(Context&, int, const u64*, int, const u64*, const Key* const*, u64* const*, int, long long, long long,
                      hipStream_t);
void ckks_mult_relin_rescale(Context&, int, const u64*, const u64*, const Key&, u64*, int, long long, long long,
                             long long, hipStream_t);
void drop_level(Context&, int, int, const u64*, u64*, int, long long, long long, hipStream_t);
void poly_addsub(Context&, int, int, int, const u64*, const u64*, u64*, int, long long, long long, long long,
                 hipStream_t);
void bfv_mult(Context&, int, const u64*, const u64*, u64*, int, long long, long long, long long, hipStream_t);
void bfv_relin(Context&, int, const u64*, const Key&, u64*, int, long long, long long, hipStream_t);
void bfv_rotate(Context&, int, const u64*, u64, const Key&, u64*, int, long long, long long, hipStream_t);
void bfv_rescale(Context&, int, int, const u64*, u64*, int, long long, long long, hipStream_t);
}  // namespace lsa

using namespace lsa;

struct lsa_context_st {
    Context ctx;
    lsa_context_st(int algo, int n, const u64* q, int nq, const u64* p, int np, u64 t, int dev)
        : ctx(algo, n, q, nq, p, np, t, dev) {}
};
struct lsa_key_st {
    Key key;
    double* fp_owned = nullptr;   // the double copy of an adopted key (the caller's buffer has no room for it)
};

template <typename F>
static int guard(F&& f) {
    try {
        f();
        return LSA_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return LSA_ERR_INTERNAL;
    } catch (...) {
        set_last_error("unknown error");
        return LSA_ERR_INTERNAL;
    }
}

static Context& C(lsa_context h) {
    LSA_REQUIRE(h != nullptr, "null context");
    h->ctx.use_device();
    return h->ctx;
}
static const Key& K(lsa_key k) {
    LSA_REQUIRE(k != nullptr && k->key.data != nullptr, "null key");
    return k->key;
}
static hipStream_t S(void* s) { return (hipStream_t)s; }

extern "C" {

const char* lsa_last_error(void) { return last_error().c_str(); }
const char* lsa_version(void) { return "lattisense_amd 0.1 (gfx950)"; }
// every LSA_* switch the library was compiled with (build_flags.h); "" for the product build.  The three translation units
// that carry switches must agree; if they do not, each view is reported.
const char* lsa_build_flags(void) {
    static const std::string text = [] {
        auto strip = [](const char* t) { return std::string(t[0] == ' ' ? t + 1 : t); };
        const std::string a = strip(LSA_BUILD_FLAGS_TEXT), k = strip(kernels_build_flags()), c = strip(context_build_flags());
        if (a == k && a == c) return a;
        return "MIXED c_api=[" + a + "] kernels=[" + k + "] context=[" + c + "]";
    }();
    return text.c_str();
}

int lsa_context_create(int algo, int n, const uint64_t* q, int nq, const uint64_t* p, int np, uint64_t t, int device,
                       lsa_context* out) {
    return guard([&] {
        LSA_REQUIRE(out != nullptr && q != nullptr && (np == 0 || p != nullptr), "null argument");
        *out = new lsa_context_st(algo, n, q, nq, p, np, t, device);
    });
}
int lsa_context_destroy(lsa_context ctx) {
    return guard([&] { delete ctx; });
}
int lsa_context_moduli(lsa_context ctx, uint64_t* out, int capacity, int* count) {
    return guard([&] {
        Context& c = C(ctx);
        if (count) *count = c.nmod;
        for (int i = 0; i < c.nmod && i < capacity; i++) out[i] = c.T.mod[i];
    });
}

int lsa_malloc(lsa_context ctx, void** dptr, size_t bytes) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipMalloc(dptr, bytes));
    });
}
int lsa_free(lsa_context ctx, void* dptr) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipFree(dptr));
    });
}
int lsa_memcpy_h2d(lsa_context ctx, void* dst, const void* src, size_t bytes, void* stream) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream)));
    });
}
int lsa_memcpy_d2h(lsa_context ctx, void* dst, const void* src, size_t bytes, void* stream) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream)));
        LSA_HIP(hipStreamSynchronize(S(stream)));
    });
}
int lsa_memcpy_d2d(lsa_context ctx, void* dst, const void* src, size_t bytes, void* stream) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(stream)));
    });
}
int lsa_stream_create(lsa_context ctx, void** stream) {
    return guard([&] {
        C(ctx);
        hipStream_t s;
        LSA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *stream = (void*)s;
    });
}
int lsa_stream_destroy(lsa_context ctx, void* stream) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipStreamDestroy(S(stream)));
    });
}
int lsa_stream_synchronize(lsa_context ctx, void* stream) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipStreamSynchronize(S(stream)));
    });
}
int lsa_event_create(lsa_context ctx, void** ev) {
    return guard([&] {
        C(ctx);
        hipEvent_t e;
        LSA_HIP(hipEventCreate(&e));
        *ev = (void*)e;
    });
}
int lsa_event_record(lsa_context ctx, void* ev, void* stream) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipEventRecord((hipEvent_t)ev, S(stream)));
    });
}
int lsa_event_elapsed_ms(lsa_context ctx, void* ev_start, void* ev_stop, float* ms) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
        LSA_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    });
}
int lsa_event_destroy(lsa_context ctx, void* ev) {
    return guard([&] {
        C(ctx);
        LSA_HIP(hipEventDestroy((hipEvent_t)ev));
    });
}

// ---- keys
size_t lsa_key_bytes(lsa_context ctx, int key_level) {
    if (!ctx) return 0;
    const Context& c = ctx->ctx;
    if (c.np < 1 || key_level < 0 || key_level >= c.nq) return 0;
    const size_t beta = (key_level + 1 + c.np - 1) / c.np;
    return beta * 2 * (size_t)(key_level + 1 + c.np) * c.n * sizeof(u64);
}

struct lsa_key_fp_owner;   // (see lsa_key_st::fp_owned)

int lsa_key_upload(lsa_context ctx, const uint64_t* compact_host, int key_level, void* stream, lsa_key* out) {
    return guard([&] {
        Context& c = C(ctx);
        LSA_REQUIRE(out != nullptr && compact_host != nullptr, "null argument");
        const size_t bytes = lsa_key_bytes(ctx, key_level);
        LSA_REQUIRE(bytes > 0, "bad key level (or context has no special primes)");
        auto k = std::make_unique<lsa_key_st>();
        k->key.level = key_level;
        k->key.owned = true;
        const bool with_fp = ks_fused_enabled(c);   // the double copy sits behind the key in the same allocation
        LSA_HIP(hipMalloc((void**)&k->key.data, with_fp ? 2 * bytes : bytes));
        LSA_HIP(hipMemcpyAsync(k->key.data, compact_host, bytes, hipMemcpyHostToDevice, S(stream)));
        double* fp = with_fp ? reinterpret_cast<double*>(k->key.data + bytes / sizeof(u64)) : nullptr;
        launch_key_prepare(c, k->key.data, fp, key_level, S(stream));
        k->key.fp = fp;
        LSA_HIP(hipStreamSynchronize(S(stream)));  // host buffer may be released by the caller on return
        *out = k.release();
    });
}
int lsa_key_adopt_device(lsa_context ctx, uint64_t* compact_dev, int key_level, void* stream, lsa_key* out) {
    return guard([&] {
        Context& c = C(ctx);
        LSA_REQUIRE(out != nullptr && compact_dev != nullptr, "null argument");
        const size_t bytes = lsa_key_bytes(ctx, key_level);
        LSA_REQUIRE(bytes > 0, "bad key level (or context has no special primes)");
        auto k = std::make_unique<lsa_key_st>();
        k->key.level = key_level;
        k->key.owned = false;
        k->key.data = compact_dev;
        if (ks_fused_enabled(c)) {   // the caller's buffer has no room: the double copy is this handle's own allocation
            LSA_HIP(hipMalloc((void**)&k->fp_owned, bytes));
            k->key.fp = k->fp_owned;
        }
        launch_key_prepare(c, k->key.data, k->fp_owned, key_level, S(stream));
        *out = k.release();
    });
}
int lsa_key_destroy(lsa_context ctx, lsa_key key) {
    return guard([&] {
        C(ctx);
        if (key) {
            if (key->key.owned && key->key.data) LSA_HIP(hipFree(key->key.data));
            if (key->fp_owned) LSA_HIP(hipFree(key->fp_owned));
            delete key;
        }
    });
}

// ---- polynomial ops
int lsa_ntt(lsa_context ctx, uint64_t* data, int batch, long long batch_stride, int rows, const int* mod_of, int period,
            int inverse, void* stream) {
    return guard([&] {
        Context& c = C(ctx);
        LSA_REQUIRE(data && mod_of && period >= 1 && period <= LSA_MAX_PERIOD, "bad arguments");
        RowMap rm;
        rm.period = period;
        for (int i = 0; i < period; i++) {
            LSA_REQUIRE(mod_of[i] == LSA_ROW_SKIP || (mod_of[i] >= 0 && mod_of[i] < c.nmod), "modulus index out of range");
            rm.mod_of[i] = (unsigned char)mod_of[i];
        }
        launch_ntt(c, data, data, batch, batch_stride, rows, rm, inverse != 0, S(stream));
    });
}
int lsa_poly_addsub(lsa_context ctx, int op, int level, int polys, const uint64_t* a, const uint64_t* b, uint64_t* out,
                    int batch, long long sa, long long sb, long long so, void* stream) {
    return guard([&] { poly_addsub(C(ctx), op, level, polys, a, b, out, batch, sa, sb, so, S(stream)); });
}

// ---- CKKS
int lsa_ckks_mult(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, uint64_t* d3, int batch,
                  long long sa, long long sb, long long sd, void* stream) {
    return guard([&] { ckks_mult(C(ctx), level, a, b, d3, batch, sa, sb, sd, S(stream)); });
}
int lsa_ckks_relin(lsa_context ctx, int level, const uint64_t* d3, lsa_key rlk, uint64_t* out, int batch, long long sd,
                   long long so, void* stream) {
    return guard([&] { ckks_relin(C(ctx), level, d3, K(rlk), out, batch, sd, so, S(stream)); });
}
int lsa_ckks_rescale(lsa_context ctx, int level, int polys, const uint64_t* in, uint64_t* out, int batch, long long si,
                     long long so, void* stream) {
    return guard([&] { ckks_rescale(C(ctx), level, polys, in, out, batch, si, so, S(stream)); });
}
int lsa_ckks_rotate(lsa_context ctx, int level, const uint64_t* in, uint64_t g, lsa_key glk, uint64_t* out, int batch,
                    long long si, long long so, void* stream) {
    return guard([&] { ckks_rotate(C(ctx), level, in, g, K(glk), out, batch, si, so, S(stream)); });
}
int lsa_drop_level(lsa_context ctx, int level, int polys, const uint64_t* in, uint64_t* out, int batch, long long si,
                   long long so, void* stream) {
    return guard([&] { drop_level(C(ctx), level, polys, in, out, batch, si, so, S(stream)); });
}
int lsa_ckks_mult_relin_rescale(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, lsa_key rlk,
                                uint64_t* out, int batch, long long sa, long long sb, long long so, void* stream) {
    return guard([&] { ckks_mult_relin_rescale(C(ctx), level, a, b, K(rlk), out, batch, sa, sb, so, S(stream)); });
}

// ---- BFV
int lsa_bfv_mult(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, uint64_t* d3, int batch, long long sa,
                 long long sb, long long sd, void* stream) {
    return guard([&] { bfv_mult(C(ctx), level, a, b, d3, batch, sa, sb, sd, S(stream)); });
}
int lsa_bfv_relin(lsa_context ctx, int level, const uint64_t* d3, lsa_key rlk, uint64_t* out, int batch, long long sd,
                  long long so, void* stream) {
    return guard([&] { bfv_relin(C(ctx), level, d3, K(rlk), out, batch, sd, so, S(stream)); });
}
int lsa_bfv_rotate(lsa_context ctx, int level, const uint64_t* in, uint64_t g, lsa_key glk, uint64_t* out, int batch,
                   long long si, long long so, void* stream) {
    return guard([&] { bfv_rotate(C(ctx), level, in, g, K(glk), out, batch, si, so, S(stream)); });
}
int lsa_bfv_rescale(lsa_context ctx, int level, int polys, const uint64_t* in, uint64_t* out, int batch, long long si,
                    long long so, void* stream) {
    return guard([&] { bfv_rescale(C(ctx), level, polys, in, out, batch, si, so, S(stream)); });
}
int lsa_bfv_mult_relin(lsa_context ctx, int level, const uint64_t* a, const uint64_t* b, lsa_key rlk, uint64_t* out,
                       int batch, long long sa, long long sb, long long so, void* stream) {
    return guard([&] {
        Context& c = C(ctx);
        const long long sd = 3LL * (level + 1) * c.n;
        // d3 lives in the second arena so that the two pipelines' own arena use cannot overlap it
        u64* d3 = c.workspace2((size_t)sd * batch, S(stream));
        bfv_mult(c, level, a, b, d3, batch, sa, sb, sd, S(stream));
        bfv_relin(c, level, d3, K(rlk), out, batch, sd, so, S(stream));
    });
}

int lsa_profile_begin(lsa_context ctx, int stride) {
    return guard([&] {
        Context& c = C(ctx);
        LSA_REQUIRE(stride >= 1, "stride must be >= 1");
        for (auto& sm : c.prof_samples) {
            c.prof_pool.push_back(sm.e0);
            c.prof_pool.push_back(sm.e1);
        }
        c.prof_samples.clear();
        for (auto& n : c.prof_launched) n = 0;
        c.prof_stride = stride;
        c.prof_on = true;
    });
}
int lsa_profile_end(lsa_context ctx) {
    return guard([&] { C(ctx).prof_on = false; });
}
int lsa_profile_read(lsa_context ctx, int kind, double* total_ms, double* total_bytes, long long* sampled,
                     long long* launched) {
    return guard([&] {
        Context& c = C(ctx);
        LSA_REQUIRE(kind >= 0 && kind < LSA_PROF_KINDS, "unknown kernel kind");
        double ms = 0, by = 0;
        long long n = 0;
        for (auto& sm : c.prof_samples) {
            if (sm.kid != kind) continue;
            LSA_HIP(hipEventSynchronize(sm.e1));
            float t = 0;
            LSA_HIP(hipEventElapsedTime(&t, sm.e0, sm.e1));
            ms += t;
            by += sm.bytes;
            n++;
        }
        if (total_ms) *total_ms = ms;
        if (total_bytes) *total_bytes = by;
        if (sampled) *sampled = n;
        if (launched) *launched = c.prof_launched[kind];
    });
}

// the same samples with only the kind's OWN algorithmic bytes (a fused launch -- the transform pass that also performs the key
// MAC -- counts both functions' bytes in lsa_profile_read and the transform's alone here)
int lsa_profile_read_primary(lsa_context ctx, int kind, double* total_bytes_primary) {
    return guard([&] {
        Context& c = C(ctx);
        LSA_REQUIRE(kind >= 0 && kind < LSA_PROF_KINDS && total_bytes_primary, "bad argument");
        double by = 0;
        for (auto& sm : c.prof_samples)
            if (sm.kid == kind) by += sm.bytes_primary;
        *total_bytes_primary = by;
    });
}

int lsa_set_fuse_tails(lsa_context ctx, int enable) {
    return guard([&] { C(ctx).fuse_tails = enable ? 1 : 0; });
}
int lsa_set_dual_stream(lsa_context ctx, int enable) {
    return guard([&] { C(ctx).dual_stream = enable ? 1 : 0; });
}
int lsa_debug_set_ntt_stamps(lsa_context ctx, void* device_buffer) {
    return guard([&] { C(ctx).ntt_diag = static_cast<unsigned long long*>(device_buffer); });
}

int lsa_ckks_rotate_many(lsa_context ctx, int level, const uint64_t* in, int n_rot, const uint64_t* galois_elements,
                         const lsa_key* glk, uint64_t* const* outs, int batch, long long sin, long long sout, void* stream) {
    return guard([&] {
        LSA_REQUIRE(in != nullptr && n_rot >= 0 && (n_rot == 0 || (galois_elements && glk && outs)), "null argument");
        std::vector<const Key*> keys(n_rot);
        for (int i = 0; i < n_rot; i++) {
            LSA_REQUIRE(glk[i] != nullptr && outs[i] != nullptr, "null key or output");
            keys[i] = &K(glk[i]);
        }
        ckks_rotate_many(C(ctx), level, in, n_rot, galois_elements, keys.data(), outs, batch, sin, sout, S(stream));
    });
}

// ---- CKKS bootstrapping
struct lsa_bootstrap_st {
    Bootstrap* b;
    Context* c;
};
int lsa_bootstrap_create_ex(lsa_context ctx, int cts_depth, int stc_depth, int k, int double_angle, double message_ratio,
                            double in_scale, double out_scale, int log_slots, int sine_deg, int arcsine_deg, void* stream,
                            lsa_bootstrap* out) {
    return guard([&] {
        LSA_REQUIRE(out != nullptr, "null argument");
        LSA_REQUIRE(k >= 1 && double_angle >= 0 && double_angle <= 8 && message_ratio > 0 && in_scale > 0, "bad bootstrap parameters");
        auto h = std::make_unique<lsa_bootstrap_st>();
        h->c = &C(ctx);
        LSA_REQUIRE(log_slots >= 0 && (log_slots == 0 || (2 << log_slots) <= C(ctx).n), "bad slot count");
        h->b = bootstrap_create(C(ctx), cts_depth, stc_depth, k, double_angle, message_ratio, in_scale, out_scale, log_slots, S(stream),
                                sine_deg, arcsine_deg);
        *out = h.release();
    });
}
int lsa_bootstrap_create(lsa_context ctx, int cts_depth, int stc_depth, int k, int double_angle, double message_ratio,
                         double in_scale, double out_scale, int log_slots, void* stream, lsa_bootstrap* out) {
    // the reference's default EvalMod: sine degree 30 (32 Chebyshev coefficients), no arcsine (frontend/custom_task.py:443-453)
    return lsa_bootstrap_create_ex(ctx, cts_depth, stc_depth, k, double_angle, message_ratio, in_scale, out_scale, log_slots, 30, 0, stream, out);
}
// polynomial constants of the plan's EvalMod: n_cheb Chebyshev coefficients, n_asin monomial coefficients of the arcsine
// correction (0: none); either output pointer may be null to query the counts
int lsa_bootstrap_evalmod_constants(lsa_bootstrap b, int* n_cheb, double* cheb, int* n_asin, double* asin_coef) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr, "null bootstrap handle");
        const auto& c = bootstrap_chebyshev(*b->b);
        const auto& a = bootstrap_arcsine(*b->b);
        if (n_cheb) *n_cheb = (int)c.size();
        if (n_asin) *n_asin = (int)a.size();
        if (cheb) std::copy(c.begin(), c.end(), cheb);
        if (asin_coef) std::copy(a.begin(), a.end(), asin_coef);
    });
}
void lsa_bootstrap_destroy(lsa_bootstrap b) {
    if (!b) return;
    bootstrap_destroy(b->b);
    delete b;
}
int lsa_bootstrap_info(lsa_bootstrap b, int* out_level, double* out_scale, int* n_galois, int* n_matrices, int* n_cts,
                       int* sparse) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr, "null bootstrap handle");
        if (out_level) *out_level = bootstrap_out_level(*b->b);
        if (out_scale) *out_scale = bootstrap_out_scale(*b->b);
        if (n_galois) *n_galois = (int)bootstrap_galois(*b->b).size();
        if (n_matrices) *n_matrices = bootstrap_matrices(*b->b);
        if (n_cts) *n_cts = bootstrap_cts_matrices(*b->b);
        if (sparse) *sparse = bootstrap_is_sparse(*b->b) ? 1 : 0;
    });
}
int lsa_bootstrap_galois_elements(lsa_bootstrap b, uint64_t* out, int capacity) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr && out != nullptr, "null argument");
        const auto& g = bootstrap_galois(*b->b);
        LSA_REQUIRE((int)g.size() <= capacity, "buffer too small");
        for (size_t i = 0; i < g.size(); i++) out[i] = g[i];
    });
}
int lsa_bootstrap_chebyshev(lsa_bootstrap b, double* out32) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr && out32 != nullptr, "null argument");
        const auto& cf = bootstrap_chebyshev(*b->b);
        LSA_REQUIRE(cf.size() <= 32, "more than 32 Chebyshev coefficients: use lsa_bootstrap_evalmod_constants");
        for (size_t i = 0; i < 32; i++) out32[i] = i < cf.size() ? cf[i] : 0.0;
    });
}
int lsa_bootstrap_matrix_info(lsa_bootstrap b, int index, int* level, int* n1, int* n_diagonals, int* diagonals, int capacity) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr, "null bootstrap handle");
        int lv, n1v;
        const std::vector<int>* ks;
        const std::vector<u64*>* pl;
        bootstrap_matrix(*b->b, index, &lv, &n1v, &ks, &pl);
        if (level) *level = lv;
        if (n1) *n1 = n1v;
        if (n_diagonals) *n_diagonals = (int)ks->size();
        if (diagonals) {
            LSA_REQUIRE((int)ks->size() <= capacity, "buffer too small");
            for (size_t i = 0; i < ks->size(); i++) diagonals[i] = (*ks)[i];
        }
    });
}
int lsa_bootstrap_plaintext(lsa_bootstrap b, int matrix, int diag_pos, uint64_t* host_out) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr && host_out != nullptr, "null argument");
        int lv, n1v;
        const std::vector<int>* ks;
        const std::vector<u64*>* pl;
        bootstrap_matrix(*b->b, matrix, &lv, &n1v, &ks, &pl);
        LSA_REQUIRE(diag_pos >= 0 && diag_pos < (int)pl->size(), "diagonal position out of range");
        b->c->use_device();
        // the rows at q_0..q_level only, whatever the plan keeps behind them: what this entry point has always written
        LSA_HIP(hipMemcpy(host_out, (*pl)[diag_pos], (size_t)(lv + 1) * b->c->n * sizeof(u64), hipMemcpyDeviceToHost));
    });
}
int lsa_bootstrap_plaintext_ext(lsa_bootstrap b, int matrix, int diag_pos, uint64_t* host_out, long long capacity_words) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr && host_out != nullptr, "null argument");
        int lv, n1v, rows = 0;
        const std::vector<int>* ks;
        const std::vector<u64*>* pl;
        bootstrap_matrix(*b->b, matrix, &lv, &n1v, &ks, &pl, &rows);
        LSA_REQUIRE(diag_pos >= 0 && diag_pos < (int)pl->size(), "diagonal position out of range");
        LSA_REQUIRE(capacity_words >= (long long)rows * b->c->n, "buffer too small for the plaintext's rows (lsa_bootstrap_plaintext_rows)");
        b->c->use_device();
        LSA_HIP(hipMemcpy(host_out, (*pl)[diag_pos], (size_t)rows * b->c->n * sizeof(u64), hipMemcpyDeviceToHost));
    });
}
int lsa_bootstrap_plaintext_rows(lsa_bootstrap b, int matrix, int* rows) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr && rows != nullptr, "null argument");
        int lv, n1v;
        const std::vector<int>* ks;
        const std::vector<u64*>* pl;
        bootstrap_matrix(*b->b, matrix, &lv, &n1v, &ks, &pl, rows);
    });
}
int lsa_ckks_bootstrap(lsa_context ctx, lsa_bootstrap b, const uint64_t* in, uint64_t* out, int batch, long long sin, long long sout,
                       lsa_key rlk, int n_glk, const uint64_t* glk_elements, const lsa_key* glk, lsa_key swk_dts, lsa_key swk_std,
                       void* stream) {
    return guard([&] {
        LSA_REQUIRE(b != nullptr && in != nullptr && out != nullptr && rlk != nullptr, "null argument");
        LSA_REQUIRE(b->c == &C(ctx), "bootstrap plan belongs to another context");
        LSA_REQUIRE((swk_dts == nullptr) == (swk_std == nullptr), "swk_dts and swk_std come as a pair");
        std::map<u64, const Key*> g;
        for (int i = 0; i < n_glk; i++) g[glk_elements[i]] = &K(glk[i]);
        bootstrap_run(*b->b, in, sin, out, sout, batch, K(rlk), g, swk_dts ? &K(swk_dts) : nullptr,
                      swk_std ? &K(swk_std) : nullptr, S(stream));
    });
}

int lsa_set_ntt_chunk_mib(lsa_context ctx, int mib) {
    return guard([&] {
        LSA_REQUIRE(mib >= 0, "chunk size must be >= 0");
        C(ctx).ntt_chunk_mib = mib;
    });
}
int lsa_set_fp64_ntt(lsa_context ctx, int enable) {
    return guard([&] { C(ctx).fp64_ntt = enable ? 1 : 0; });
}
int lsa_set_tile_batch(lsa_context ctx, int tile_batch) {
    return guard([&] {
        LSA_REQUIRE(tile_batch >= 0, "tile_batch must be >= 0");
        C(ctx).tile_batch = tile_batch;
    });
}
int lsa_probe_copy(lsa_context ctx, uint64_t* dst, const uint64_t* src, size_t n_u64, void* stream) {
    return guard([&] {
        C(ctx);
        launch_probe_copy(dst, src, n_u64, S(stream));
    });
}
int lsa_probe_mulhi(lsa_context ctx, uint64_t* buf, size_t n_u64, int iters, void* stream) {
    return guard([&] {
        C(ctx);
        launch_probe_mulhi(buf, n_u64, iters, S(stream));
    });
}

}  // extern "C"
