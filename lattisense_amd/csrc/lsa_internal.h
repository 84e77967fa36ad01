// lsa_internal.h — context, device tables, workspace and launcher declarations (host side, C++).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/lattisense_amd.h"
#include "ntt_plan.h"
#include "tables.h"

namespace lsa {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& m);

#define LSA_HIP(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw lsa::Error(_e == hipErrorNoDevice || _e == hipErrorInvalidDevice ? LSA_ERR_NO_DEVICE    \
                                                                                   : LSA_ERR_HIP,        \
                             std::string(#expr) + ": " + hipGetErrorString(_e));                          \
    } while (0)

#define LSA_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) throw lsa::Error(LSA_ERR_ARG, (msg));       \
    } while (0)

// ---------------------------------------------------------------- exact RNS base conversion plan (device constants)
#define LSA_BC_MAX_SRC 16
#define LSA_BC_MAX_DST 64

struct BaseConvConsts {
    int ns, nd, centered;
    int src_mod[LSA_BC_MAX_SRC];
    int dst_mod[LSA_BC_MAX_DST];
    u64 shat_inv_m[LSA_BC_MAX_SRC];                 // (S/q_i)^-1 mod q_i, Montgomery form
    u64 half_src[LSA_BC_MAX_SRC];                   // floor(S/2) mod q_i
    double qf[LSA_BC_MAX_SRC];                      // (double) q_i
    double rf[LSA_BC_MAX_SRC];                      // RN(1 / qf): reciprocal for the 3-operation exact division
    u64 shat_m[LSA_BC_MAX_DST][LSA_BC_MAX_SRC];     // (S/q_i) mod p_j, Montgomery form
    u64 vs[LSA_BC_MAX_DST][LSA_BC_MAX_SRC + 1];     // v*S mod p_j, v = 0..ns
    u64 half_dst[LSA_BC_MAX_DST];                   // floor(S/2) mod p_j
    // every source and target modulus below 2^58: shat_m split into 29-bit halves for the carry-free accumulate of
    // k_baseconv<.., SPLIT> (shat_m[j][i] = hi * 2^29 + lo)
    int split29;
    u32 shat_lo[LSA_BC_MAX_DST][LSA_BC_MAX_SRC];
    u32 shat_hi[LSA_BC_MAX_DST][LSA_BC_MAX_SRC];
    u32 shat_sum[LSA_BC_MAX_DST][LSA_BC_MAX_SRC];      // lo + hi: the middle column as ONE product (y0 + y1)(w0 + w1) - y0 w0 - y1 w1
    // the output corrections -v*S - [centred] floor(S/2), Montgomery form, as ONE addend v * corr_a + corr_b (< 17 * 2^58) of the
    // 128-bit sum ahead of its single REDC: corr_a = (-S) * 2^64 mod p_j, corr_b = (-floor(S/2)) * 2^64 mod p_j or 0
    u64 corr_a[LSA_BC_MAX_DST], corr_b[LSA_BC_MAX_DST];
};

struct BaseConvPlan {
    BaseConvConsts* dev = nullptr;
    int ns = 0, nd = 0;
    bool split29 = false;
};

enum ProfKind { PROF_NTT = 0, PROF_BASECONV = 1, PROF_KSMAC = 2, PROF_TENSOR = 3, PROF_ELEMWISE = 4, LSA_PROF_KINDS = 5 };

struct Key {
    u64* data = nullptr;   // device, compact order [beta][2][klvl+1+np][N], NTT domain, MONTGOMERY form
    int level = 0;
    bool owned = true;
    // the same key as plain integer-valued doubles, same layout, valid for the limbs of the FP64 engine (q < 2^47) only: the
    // operand of the key MAC that is fused into the extension transform's second pass (k_ntt_r16_ksmac); null = not built
    // (that path is then not taken).  Lives and dies with `data` (same allocation, or owned by the key's holder).
    const double* fp = nullptr;
};

struct Context {
    int algo, n, logn, nq, np, nmul, nmod, device;
    u64 t;
    HostTables T;
    NttPlan plan;        // 4096-point tiles: one pass up to N = 2^12, two passes above
    NttPlan plan_wide;   // N = 2^13 / 2^14 only: the whole limb in one 512 / 1024-thread workgroup, one pass
    int wide_mode = 2;   // LSA_NTT_WIDE: 0 never, 1 always, unset = per launch (launch_ntt)
    ModDev* d_mods = nullptr;
    u64* d_psi = nullptr;
    u64* d_psiinv = nullptr;
    u64* d_scale = nullptr;
    double* d_psi_d = nullptr;      // FP64-engine copies of the twiddle tables
    double* d_psiinv_d = nullptr;
    u64* d_psi_w = nullptr;         // the same four tables in the order of plan_wide (N = 2^13 / 2^14 only)
    u64* d_psiinv_w = nullptr;
    double* d_psi_d_w = nullptr;
    double* d_psiinv_d_w = nullptr;
    double* d_scale_d = nullptr;
    int fp64_ntt = 1;               // use the FP64 butterfly engine for limbs with q < 2^47
    int tile_batch = 0;
    int fp_raw = 1;                 // FP64-engine limbs cross between the two NTT passes as doubles (LSA_NTT_FP_RAW=0: canonical u64)
    int fuse_tails = 1;             // ModDown / rescale element-wise tails fused into the NTT load/store phases
    int dual_stream = 0;            // 1: overlap alternate tiles of an operator on an auxiliary stream (+5% throughput,
                                    // but per-kernel timings then include the co-running kernel; off for clean accounting)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void fork_aux(hipStream_t s);   // aux stream waits for everything enqueued on s so far
    void join_aux(hipStream_t s);   // s waits for everything enqueued on the aux stream
    unsigned long long* ntt_diag = nullptr;   // device buffer for LSA_NTT_DIAG_STAMPS builds (8 stamps per workgroup)
    int ntt_chunk_mib = 0;          // >0: two-pass NTTs run pass A+B per chunk of this many MiB (Infinity-Cache reuse)

    // sampled HIP-event timing of kernel launches (bench.py roofline leg); off unless lsa_profile_begin was called
    struct ProfSample {
        hipEvent_t e0, e1;
        int kid;
        double bytes;
        double bytes_primary;   // the kind's own function only (a fused launch also does another kernel's work: `bytes` counts both)
    };
    bool prof_on = false;
    int prof_stride = 1;
    long long prof_launched[LSA_PROF_KINDS] = {0};
    std::vector<ProfSample> prof_samples;
    std::vector<hipEvent_t> prof_pool;

    std::mutex mu;
    std::map<std::string, BaseConvPlan> bconv;       // device copies, keyed by "c|src..|dst.."
    std::map<u64, u32*> perm_ntt;                   // galois element -> device gather table (NTT domain)
    std::map<u64, u32*> perm_coeff;                 // galois element -> device scatter table with sign bit
    std::map<std::string, u64*> consts;             // misc per-level device constant vectors

    // workspace arena: grows on demand, reused across calls (single in-flight operator per context)
    u64* ws = nullptr;
    size_t ws_words = 0;

    Context(int algo_, int n_, const u64* q, int nq_, const u64* p, int np_, u64 t_, int device_);
    ~Context();
    void use_device() const { LSA_HIP(hipSetDevice(device)); }
    u64* workspace(size_t words, hipStream_t s);
    u64* ws2 = nullptr;   // second arena: operator-to-operator intermediates of composed entry points
    size_t ws2_words = 0;
    u64* workspace2(size_t words, hipStream_t s);
    int p_mod(int i) const { return nq + i; }
    int aux_mod(int i) const { return nq + np + i; }
    // pinv_scaled: the outputs of every target but the last are multiplied by P^-1 mod p_j (merged ModDown + rescale)
    // fold: element-wise steps on either side of a conversion folded into its constants (both linear, so the residues are the
    // ones the separate steps give): src_pre[i] multiplies source limb i BEFORE the conversion (the centring offset and
    // (S/q_i)^-1 absorb it), dst_scale[j] multiplies target j's output; `tag` names the variant in the plan cache
    struct BaseConvFold {
        std::string tag;
        std::vector<u64> src_pre, dst_scale;   // plain residues; empty = none
    };
    const BaseConvPlan* baseconv(const std::vector<int>& src, const std::vector<int>& dst, bool centered,
                                 bool pinv_scaled = false, const BaseConvFold* fold = nullptr);
    const u64* pinv_vec(int level);
    const u64* pmodq_vec(int level);   // [level+1] P mod q_j, Montgomery form (extended ciphertexts: c0 * P)
    const u64* qlinv_vec(int level);
    const u32* ntt_perm(u64 g);
    const u32* coeff_perm(u64 g);
    // per-modulus constant vector on device, Montgomery form, built by `gen(mod_index)`
    const u64* const_vec(const std::string& name, const std::vector<int>& mods, const std::vector<u64>& plain_vals);
    const u64* raw_vec(const std::string& name, const std::vector<u64>& vals);   // cached device copy, no conversion
};

// RAII sample of one kernel launch: records an event pair around every prof_stride-th launch of a kind
struct ProfScope {
    Context& c;
    hipStream_t s;
    bool active = false;
    Context::ProfSample smp{};
    ProfScope(Context& c_, int kid, double bytes, hipStream_t s_, double bytes_primary = -1.0) : c(c_), s(s_) {
        if (!c.prof_on) return;
        const long long idx = c.prof_launched[kid]++;
        // every stride-th launch on average, picked by a hash of the launch index: a fixed stride would lock onto the
        // period of the operator's launch sequence and always time the same launch types
        if (c.prof_stride > 1 && ((unsigned long long)(idx + 1) * 0x9E3779B97F4A7C15ull >> 33) % (unsigned)c.prof_stride != 0) return;
        auto take = [&]() {
            hipEvent_t e;
            if (!c.prof_pool.empty()) {
                e = c.prof_pool.back();
                c.prof_pool.pop_back();
            } else {
                LSA_HIP(hipEventCreate(&e));
            }
            return e;
        };
        smp.e0 = take();
        smp.e1 = take();
        smp.kid = kid;
        smp.bytes = bytes;
        smp.bytes_primary = bytes_primary < 0 ? bytes : bytes_primary;
        LSA_HIP(hipEventRecord(smp.e0, s));
        active = true;
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(smp.e1, s);
        c.prof_samples.push_back(smp);
    }
};

// ---------------------------------------------------------------- launchers (kernels.hip)
struct RowMap {   // rows of a batch item -> modulus index (0xFF = skip)
    int period;
    unsigned char mod_of[LSA_MAX_PERIOD];
    int row0 = 0, row_step = 1;   // the launch's i-th row is row row0 + i*row_step of the batch item (mod_of[row % period])
};

void launch_ntt(Context& c, const u64* src, u64* dst, int batch, long long batch_stride, int rows, const RowMap& rm,
                bool inverse, hipStream_t s);
// element-wise work fused into a forward transform's first-pass load / last-pass store (see NttPassArgs::fz_*)
struct NttFusion {
    int epi = 0, pro = 0, limbs = 1, base_polys = 0, ql_mod = 0;
    const u64* a = nullptr;
    long long a_stride = 0;
    int a_rpp = 0;
    const u64* base = nullptr;
    long long base_stride = 0;
    int base_rpp = 0;
    const u64* k = nullptr;
    u64* out = nullptr;
    long long out_stride = 0;
    int out_rpp = 0;
    const u64* last = nullptr;
    long long last_stride = 0;
    int last_rpp = 1;            // rows between the polynomials' last limbs in `last`
    const u64* k2 = nullptr;     // epi == 2: out = (a*k - v + base) * k2
    const u32* scatter = nullptr;   // epilogue only: out[scatter[x]] = value(x) within each row (the automorphism of a rotation)
};
// passes: bit 0 = the first executed pass, bit 1 = the second (two-pass plans; a caller that fuses the second pass into
// another kernel asks for 1 only)
void launch_ntt(Context& c, const u64* src, u64* dst, int batch, long long src_stride, long long dst_stride, int rows,
                const RowMap& rm, bool inverse, hipStream_t s, const NttFusion* fz = nullptr, int passes = 3);

// limb-wise binary/unary ops on [batch][rows][N]; row r uses modulus rm.mod_of[r % period]
enum EwOp { EW_ADD = 0, EW_SUB = 1, EW_NEG = 2, EW_MUL = 3 };
void launch_elementwise(Context& c, EwOp op, const u64* a, const u64* b, u64* out, int batch, long long sa,
                        long long sb, long long so, int rows, const RowMap& rm, hipStream_t s);
void launch_muladd(Context& c, EwOp op, const u64* a, const u64* b, const u64* acc, long long sacc, u64* out, int batch,
                   long long sa, long long sb, long long so, int rows, const RowMap& rm, hipStream_t s);
// out[p][j] = (partial[p][j]) + sum_{i<terms} ct_i[p][j] * pt_i[j], one launch; each operand is (base, batch stride)
#define LSA_MAC_MAX_TERMS 16
#define LSA_MACM_MAX 8   // baby and giant steps per k_mac_plain_multi launch
void launch_mac_plain(Context& c, int terms, const u64* const* ct, const long long* sct, const u64* const* pt,
                      const long long* spt, const u64* partial, long long spartial, u64* out, long long so, int batch,
                      int polys, int limbs, const RowMap& rm, hipStream_t s);
void launch_mac_plain_multi(Context& c, int nb, const u64* const* ct, const long long* sct, int ng, const u64* const* pt,
                            u64* const* out, long long so, int batch, int polys, int limbs, const RowMap& rm, hipStream_t s);
// ring-t plaintext limb -> [level+1][N] residues: mode 0 centred lift from q_0 (CKKS), 1 direct (BFV multiply),
// 2 scale-up by Q/t (BFV add/sub)
void launch_lift_ringt(Context& c, int mode, int level, const u64* pt, long long spt, u64* out, long long sout, int batch,
                       hipStream_t s);
// CKKS/BFV tensor: a,b [2][T][N] -> d [3][T][N]; limb i uses modulus rm.mod_of[i]
// a_rpp / b_rpp: rows per polynomial of the operands (0 = limbs; more when an operand is kept at a higher level: its first
// `limbs` rows of each polynomial are the operand at this level -- no copy needed to "drop" it)
void launch_tensor(Context& c, const u64* a, const u64* b, u64* d, int batch, long long sa, long long sb, long long sd,
                   int limbs, const RowMap& rm, hipStream_t s, int a_rpp = 0, int b_rpp = 0);
// exact base conversion: src limbs at rows src_row[i] of the source item, dst limbs at rows dst_row[j] of the dest item
struct BaseConvRows {
    int src_row[LSA_BC_MAX_SRC];
    int dst_row[LSA_BC_MAX_DST];
};
// sub (optional): converted is src - sub, limb for limb (rows.src_row indexes both; its own batch stride and row offsets
// sub_row) -- the subtraction of a preceding element-wise step done on the conversion's source load
void launch_baseconv(Context& c, const BaseConvPlan* k, const BaseConvRows& rows, const u64* src, u64* dst, int batch,
                     long long ssrc, long long sdst, hipStream_t s, const u64* sub = nullptr, long long ssub = 0,
                     const int* sub_row = nullptr);
// key-switch inner product: acc[h][tl] = sum_d ext(d,tl) * key[d][h][tl];  ext(d,tl) = cx[tl] when tl is in digit d
// engine: -1 every target limb; 0 / 1 only the target limbs of the integer / FP64 butterfly engine
// scatter (engine -1 only): the result leaves as the ROTATED EXTENDED ciphertext acc[h][tl][scatter[x]] = sum(x) + (h == 0, tl < L:
// P * base[tl][x]) -- gadget product, c0 * P and the automorphism of a baby-step rotation in one pass
void launch_ks_mac(Context& c, int level, const u64* cx, long long scx, const u64* ext, long long sext,
                   const Key& key, u64* acc, long long sacc, int batch, hipStream_t s, int engine = -1,
                   const u32* scatter = nullptr, const u64* base = nullptr, long long sbase = 0);
// out[h][i] = base[h][i] + (acc[h][i] - conv[h][i]) * Pinv_i       (base may be null)
void launch_moddown_final(Context& c, int level, const u64* acc, long long sacc, int acc_rows_per_poly, const u64* conv,
                          long long sconv, const u64* base, long long sbase, int base_rows_per_poly, int base_polys,
                          u64* out, long long sout, int batch, hipStream_t s);
// out[p][i] = base[p][i] + (a[p][i] - b[p][i]) * kvec[i]; row of operand X = p*X_rpp + i; b/base optional
void launch_sub_mul_general(Context& c, int polys, int limbs, const unsigned char* limb_mod, const u64* kvec,
                            const u64* a, long long sa, int a_rpp, const u64* b, long long sb, int b_rpp,
                            const u64* base, long long sbase, int base_rpp, int base_polys, u64* out, long long so,
                            int out_rpp, int batch, hipStream_t s);
// rescale helpers (divide-and-round by the last modulus of `level`)
void launch_rescale_prep(Context& c, int level, int polys, const u64* last, long long slast, u64* tmp, long long stmp,
                         int batch, hipStream_t s);
void launch_rescale_final(Context& c, int level, int polys, const u64* in, long long sin, const u64* tmp, long long stmp,
                          u64* out, long long sout, int batch, hipStream_t s);
// gather rows: out[r][i] = in[r][perm[i]] (NTT domain automorphism) ; scatter with sign for coefficient domain
void launch_permute_ntt(Context& c, const u32* perm, const u64* in, long long sin, u64* out, long long sout, int rows,
                        int batch, hipStream_t s);
void launch_permute_coeff(Context& c, const u32* perm, const u64* in, long long sin, u64* out, long long sout, int rows,
                          const RowMap& rm, int batch, hipStream_t s);
// extended (Q_level u P) ciphertext: out (+)= perm(acc + P * base) -- a rotation without its division by P (k_permute_ext)
void launch_permute_ext(Context& c, int level, const u32* perm, const u64* acc, long long sacc, const u64* base, long long sbase,
                        int base_polys, u64* out, long long sout, bool accumulate, int batch, hipStream_t s);
// strided row copy: out[b][r] = in[b][src_row[r]]
void launch_copy_rows(Context& c, const u64* in, long long sin, u64* out, long long sout, int rows, const int* src_row,
                      int batch, hipStream_t s);
void launch_to_mont(Context& c, u64* data, int rows, const RowMap& rm, hipStream_t s);
// a freshly loaded key (plain residues, compact order) -> Montgomery form in place (+ the double copy when fp != nullptr)
void launch_key_prepare(Context& c, u64* data, double* fp, int key_level, hipStream_t s);
bool ks_fused_enabled(const Context& c);   // the fused second-pass + key-MAC kernel applies to this context (and is not switched off)
// second pass of the extension transform + gadget inner product in one launch (see k_ntt_r16_ksmac); false = shape not covered
// engines: bit 0 integer-engine target limbs, bit 1 FP64-engine target limbs
bool launch_ntt_ksmac(Context& c, int level, const u64* cx, long long scx, u64* ext, long long sext, const Key& key, u64* acc,
                      long long sacc, int batch, hipStream_t s, int engines = 3);
int ks_fused_engines(const Context& c);   // which engines' target limbs take the fused kernel (LSA_KS_FUSED_ENGINES, default FP64 only)
// out = (a - b) * k_i  with per-row constant (Montgomery form) ; out = a * k_i
void launch_sub_mul_const(Context& c, const u64* a, long long sa, const u64* b, long long sb, const u64* kvec, u64* out,
                          long long so, int rows, const RowMap& rm, int batch, hipStream_t s);
void launch_mul_const(Context& c, const u64* a, long long sa, const u64* kvec, u64* out, long long so, int rows,
                      const RowMap& rm, int batch, hipStream_t s);
// out[row] = a[row] * mvec[row] + kvec[row] (kvec: plain residues, mvec: Montgomery-form factors or null)
void launch_add_const(Context& c, const u64* a, long long sa, const u64* kvec, u64* out, long long so, int rows,
                      const RowMap& rm, int batch, hipStream_t s, const u64* mvec = nullptr);
void launch_probe_copy(u64* dst, const u64* src, size_t n, hipStream_t s);
void launch_probe_mulhi(u64* buf, size_t n, int iters, hipStream_t s);

// ---------------------------------------------------------------- CKKS bootstrapping (bootstrap.hip)
struct Bootstrap;
Bootstrap* bootstrap_create(Context& c, int cts_depth, int stc_depth, int K, int double_angle, double message_ratio,
                            double in_scale, double out_scale, int log_slots, hipStream_t s, int sine_deg = 30, int arcsine_deg = 0);
const std::vector<double>& bootstrap_arcsine(const Bootstrap& bt);
bool bootstrap_is_sparse(const Bootstrap& bt);
void bootstrap_destroy(Bootstrap* b);
int bootstrap_out_level(const Bootstrap& bt);
double bootstrap_out_scale(const Bootstrap& bt);
const std::vector<u64>& bootstrap_galois(const Bootstrap& bt);
const std::vector<double>& bootstrap_chebyshev(const Bootstrap& bt);
int bootstrap_matrices(const Bootstrap& bt);
int bootstrap_cts_matrices(const Bootstrap& bt);
void bootstrap_matrix(const Bootstrap& bt, int i, int* level, int* n1, const std::vector<int>** ks, const std::vector<u64*>** plains,
                      int* rows = nullptr);
void bootstrap_run(Bootstrap& bt, const u64* in, long long sin, u64* out, long long sout, int batch, const Key& rlk,
                   const std::map<u64, const Key*>& glk, const Key* swk_dts, const Key* swk_std, hipStream_t s);

// ---------------------------------------------------------------- operator pipelines (ops.hip)
const std::string& last_error();

}  // namespace lsa
