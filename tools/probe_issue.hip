// probe_issue.hip — issue interval (cycles per wave64 instruction and SIMD) of the integer instructions the Shoup butterfly is
// made of, measured as streams of 8 independent chains per wave at 1, 2, 4 and 8 waves per SIMD, plus one dependent chain
// (latency).  Clock taken from hipDeviceProp (clockRate); cycles = time * clock / (instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { MAD64, MULLO, MULHI, ADD64, ADDCO, ADD32, FMA64, CMP64, NKIND };
static const char* kind_name[NKIND] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64", "v_add_co+v_addc (pair)", "v_add_u32",
                                        "v_fma_f64", "v_cmp_u64+2 cndmask"};

template <int KIND, int CH> __global__ __launch_bounds__(256) void k_issue(u64* out, int iters, u64 seed) {
    u64 a[CH];
    u32 x = (u32)seed + threadIdx.x, y = (u32)(seed >> 32) | 1;
#pragma unroll
    for (int c = 0; c < CH; c++) a[c] = seed * (c + 1) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                u64 cy;
                if (KIND == MAD64) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a[c]), "=s"(cy) : "v"(x), "v"(y));
                else if (KIND == MULLO) { u32 t = (u32)a[c]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(y)); a[c] = t; }
                else if (KIND == MULHI) { u32 t = (u32)a[c]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t) : "v"(y)); a[c] = t; }
                else if (KIND == ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[c]) : "v"(seed));
                else if (KIND == ADDCO) a[c] = a[c] + (seed ^ a[(c + 1) % CH]);
                else if (KIND == ADD32) { u32 t = (u32)a[c]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(y)); a[c] = t; }
                else if (KIND == FMA64) { double d = __longlong_as_double(a[c]); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "v"(__longlong_as_double(seed))); a[c] = __double_as_longlong(d); }
                else if (KIND == CMP64) a[c] = a[c] >= seed ? a[c] - seed : a[c];
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s ^= a[c];
    if (s == 0x1234567) out[threadIdx.x] = s;
}

template <int KIND, int CH> static void run(int waves_per_simd, double clock_hz, int ncu) {
    // one workgroup of 256 threads = 1 wave per SIMD; waves_per_simd workgroups per CU
    const int blocks = ncu * waves_per_simd, iters = 4000;
    u64* out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_issue<KIND, CH><<<blocks, 256>>>(out, 10, 0x9E3779B97F4A7C15ull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_issue<KIND, CH><<<blocks, 256>>>(out, iters, 0x9E3779B97F4A7C15ull);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)iters * 8 * CH * waves_per_simd;
    printf("%-26s chains %d waves/SIMD %d: %.2f cycles per wave instruction\n", kind_name[KIND], CH, waves_per_simd, ms * 1e-3 * clock_hz / instr_per_simd);
    CK(hipFree(out));
}

template <int KIND> static void sweep(double clk, int ncu) {
    run<KIND, 1>(1, clk, ncu);
    run<KIND, 8>(1, clk, ncu);
    run<KIND, 8>(2, clk, ncu);
    run<KIND, 8>(4, clk, ncu);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const double clk = p.clockRate * 1e3;
    printf("%s, %d CUs, clock %.0f MHz (nominal; a lower sustained clock shows as proportionally more cycles)\n", p.name, p.multiProcessorCount, clk / 1e6);
    sweep<MAD64>(clk, p.multiProcessorCount);
    sweep<MULLO>(clk, p.multiProcessorCount);
    sweep<MULHI>(clk, p.multiProcessorCount);
    sweep<ADD64>(clk, p.multiProcessorCount);
    sweep<ADDCO>(clk, p.multiProcessorCount);
    sweep<ADD32>(clk, p.multiProcessorCount);
    sweep<FMA64>(clk, p.multiProcessorCount);
    sweep<CMP64>(clk, p.multiProcessorCount);
    return 0;
}
