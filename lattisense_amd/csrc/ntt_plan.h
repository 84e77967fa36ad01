// ntt_plan.h — pass planning shared by the launcher and the CPU replay harness.
#pragma once
#include "ntt_core.h"

struct NttPassShape { int s_lo, mu, lambda, tau; };
struct NttPlan { int npass; NttPassShape pass[2]; };

// logn <= tau_max: one pass with the whole limb in LDS.  Otherwise two passes:
// A = stages [0, floor(logn/2)) on strided columns, B = the remaining stages on contiguous chunks.
inline NttPlan make_ntt_plan(int logn, int tau_max = 12, int mu_a_override = 0) {
    NttPlan p;
    if (logn <= tau_max) {
        p.npass = 1;
        p.pass[0] = {0, logn, 0, logn};
    } else {
        // pass A keeps >= 16 columns (128-byte segments) per tile: mu_a <= tau - 4
        int mu_a = logn / 2;
        if (mu_a > tau_max - 4) mu_a = tau_max - 4;
        // fewer first-pass stages = longer contiguous column segments (2^(tau - mu_a) elements) per tile row
        if (mu_a_override > 0 && mu_a_override <= tau_max - 4 && logn - mu_a_override <= tau_max) mu_a = mu_a_override;
        int mu_b = logn - mu_a;
        p.npass = 2;
        p.pass[0] = {0, mu_a, tau_max - mu_a, tau_max};
        p.pass[1] = {mu_a, mu_b, 0, tau_max};
    }
    return p;
}

// fill the shape/direction fields of `a` for pass index `k` of a forward (inverse=0) or inverse transform;
// passes of an inverse transform run in the order npass-1 .. 0.
inline void ntt_fill_pass(NttPassArgs& a, const NttPlan& plan, int logn, int k, int inverse) {
    const NttPassShape& s = plan.pass[k];
    a.logn = logn;
    a.s_lo = s.s_lo;
    a.mu = s.mu;
    a.lambda = s.lambda;
    a.tau = s.tau;
    a.inverse = inverse;
    a.apply_scale = inverse && s.s_lo == 0;
    a.final_reduce = inverse ? (k == 0) : (k == plan.npass - 1);
}

// j of every stage under `plan`: its position inside the radix group (sub-pass) that executes it -- what the twiddle
// tables' per-stage order is keyed on (ntt_core.h, "twiddle table layout")
inline void ntt_plan_stage_j(const NttPlan& plan, int logn, unsigned char* jl /*[logn]*/) {
    for (int k = 0; k < plan.npass; k++) {
        int rho[4];
        const int np = ntt_split(plan.pass[k].mu, rho);
        int s = plan.pass[k].s_lo;
        for (int i = 0; i < np; i++)
            for (int j = 0; j < rho[i]; j++) jl[s++] = (unsigned char)j;
    }
    (void)logn;
}

// natural-order table of one limb (entry x = the twiddle of merged index x, `words` u64/double words per entry) -> the
// layout the kernel reads under `plan`; entry 0 is unused in both
template <class T>
inline void ntt_permute_twiddles(const NttPlan& plan, int logn, const T* nat, T* out, int words, bool fp) {
    unsigned char jl[32] = {0};
    ntt_plan_stage_j(plan, logn, jl);
    for (int w = 0; w < words; w++) out[w] = nat[w];
    for (int s = 0; s < logn; s++) {
        const int j = jl[s];
        for (long long i = 0; i < (1LL << s); i++) {
            const long long G = i >> j;
            const int k = (int)(i & ((1 << j) - 1));
            const long long pos = (1LL << s) + (fp ? ntt_tw_pos_fp(s, j, G, k) : ntt_tw_pos_int(s, j, G, k));
            for (int w = 0; w < words; w++) out[pos * words + w] = nat[((1LL << s) + i) * words + w];
        }
    }
}
