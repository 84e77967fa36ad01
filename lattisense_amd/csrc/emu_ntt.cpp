// emu_ntt.cpp — CPU replay of the NTT kernel's phase functions, one simulated thread at a time
// (g++ -DLSA_EMULATE).  Debugging aid for the kernel's indexing; used only by tests/test_emulate_ntt.py.
#define LSA_EMULATE 1
#include <cstddef>
#include <vector>
#include "ntt_plan.h"
#include "ntt_r16.h"
#include <array>
#include "tables.h"

template <int NT>
static void emu_block(const NttPassArgs& a, const NttBlockCtx& bc, u64* lds) {
    for (int t = 0; t < NT; t++) ntt_phase_load<true, NT>(a, bc, t, lds);
    int rho[4];
    const int np = ntt_split(a.mu, rho);
    if (!a.inverse) {
        int sig = 0;
        for (int i = 0; i < np; i++) {
            for (int t = 0; t < NT; t++) ntt_phase_sub_dyn<NT>(a, bc, t, lds, sig, rho[i]);
            sig += rho[i];
        }
    } else {
        int sig = a.mu;
        for (int i = np - 1; i >= 0; i--) {
            sig -= rho[i];
            for (int t = 0; t < NT; t++) ntt_phase_sub_dyn<NT>(a, bc, t, lds, sig, rho[i]);
        }
    }
    for (int t = 0; t < NT; t++) ntt_phase_store<true, NT>(a, bc, t, lds);
}

// the radix-16-squared pass (ntt_r16.h): each phase for every thread in turn, a thread's registers kept between its phases
template <int PASS, int MU>
static void emu_block_r16(const NttPassArgs& a, const NttBlockCtx& bc, u64* lds) {
    std::vector<std::array<u64, 16>> regs(LSA_R16_THREADS);
    for (int phase = 0; phase < 3; phase++)
        for (int t = 0; t < LSA_R16_THREADS; t++) {
            u64(&v)[16] = *reinterpret_cast<u64(*)[16]>(regs[t].data());
            r16_phase<PASS, 3, MU>(a, bc, t, lds, phase, v);
        }
}

// the nine-stage second pass (three radix-8 groups per point, four phases)
static void emu_block_r8x3(const NttPassArgs& a, const NttBlockCtx& bc, u64* lds) {
    std::vector<std::array<u64, 16>> regs(LSA_R16_THREADS);
    for (int phase = 0; phase < 4; phase++)
        for (int t = 0; t < LSA_R16_THREADS; t++) {
            u64(&v)[16] = *reinterpret_cast<u64(*)[16]>(regs[t].data());
            r8x3_phase<3>(a, bc, t, lds, phase, v);
        }
}

extern "C" int lsa_emu_ntt(int n, const u64* moduli, int nmod, u64* data, int batch, long long batch_stride, int rows,
                           const unsigned char* mod_of, int period, int inverse, int tau_max, int allow_fp64) {
    const int row_inner = (allow_fp64 >> 1) & 1;   // bit 1: the (tile, row, batch) workgroup order
    const bool r16 = (allow_fp64 >> 2) & 1;        // bit 2: 8-stage passes through the radix-16-squared kernel
    allow_fp64 &= 1;
    lsa::HostTables T;
    T.build(n, std::vector<u64>(moduli, moduli + nmod));
    NttPlan plan = make_ntt_plan(T.logn, tau_max);
    // tables in the plan's order, as Context uploads them
    std::vector<u64> tw_i(T.psi.size());
    std::vector<double> tw_d(T.psi_d.size());
    for (int m = 0; m < nmod; m++) {
        const size_t o = (size_t)m * n;
        ntt_permute_twiddles(plan, T.logn, (inverse ? T.psiinv : T.psi).data() + 2 * o, tw_i.data() + 2 * o, 2, false);
        ntt_permute_twiddles(plan, T.logn, (inverse ? T.psiinv_d : T.psi_d).data() + o, tw_d.data() + o, 1, true);
    }
    NttPassArgs a{};
    a.src = data;
    a.dst = data;
    a.src_stride = batch_stride;
    a.dst_stride = batch_stride;
    a.batch = batch;
    a.rows = rows;
    a.mods = T.mods.data();
    a.tw = tw_i.data();
    a.scale = T.scale.data();
    a.twd = tw_d.data();
    a.scaled = T.scale_d.data();
    a.allow_fp64 = allow_fp64;
    a.period = period;
    a.row0 = 0;
    a.row_step = 1;
    a.row_inner = row_inner;
    for (int i = 0; i < period; i++) a.mod_of[i] = mod_of[i];
    // as launch_ntt: the grid covers the active rows only
    int launch_rows = 0;
    for (int r = 0; r < rows; r++)
        if (mod_of[r % period] != LSA_ROW_SKIP) a.row_tbl[launch_rows++] = (unsigned short)r;
    a.compact = 1;
    a.rows = launch_rows;
    for (int step = 0; step < plan.npass; step++) {
        int k = inverse ? plan.npass - 1 - step : step;
        ntt_fill_pass(a, plan, T.logn, k, inverse);
        a.fp_raw_out = plan.npass == 2 && step == 0;   // as launch_ntt sets them
        a.fp_raw_in = plan.npass == 2 && step == 1;
        std::vector<u64> lds(lds_words(a.tau));
        long long nblocks = (long long)batch * launch_rows * (1 << (a.logn - a.tau));
        for (long long bid = 0; bid < nblocks; bid++) {
            NttBlockCtx bc = ntt_decode_block(a, bid);
            if (bc.mod == LSA_ROW_SKIP) continue;
            if (r16 && ntt_r16_shape_ok(a, plan.npass)) {
                if (a.mu == 9) emu_block_r8x3(a, bc, lds.data());
                else if (a.lambda && a.mu == 8) emu_block_r16<0, 8>(a, bc, lds.data());
                else if (a.lambda) emu_block_r16<0, 7>(a, bc, lds.data());
                else if (a.mu == 8) emu_block_r16<1, 8>(a, bc, lds.data());
                else emu_block_r16<1, 7>(a, bc, lds.data());
            } else if (a.tau <= 12) emu_block<LSA_NTT_THREADS>(a, bc, lds.data());
            else if (a.tau == 13) emu_block<512>(a, bc, lds.data());
            else emu_block<1024>(a, bc, lds.data());
        }
    }
    return 0;
}
