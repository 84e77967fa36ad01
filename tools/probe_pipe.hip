// probe_pipe.hip — does a workgroup that prefetches its NEXT tile into registers while it computes on the current one
// (compute phase free of vector-memory loads) overlap HBM traffic with VALU work?  Synthetic stand-in for an NTT pass:
// 4096-point tiles through LDS, R rounds of LDS-only radix-16-shaped work (16 points per thread, W dependent 64-bit
// multiply-adds per point and round), then the store.  Variants: one tile per workgroup (today's kernel shape) against
// K tiles per workgroup with the register prefetch; workgroups per CU set through the LDS allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int pad(int l) { return l + (l >> 4); }

template <int W> __device__ __forceinline__ void compute_round(u64* lds, int tid, int round, u64 c1, u64 c2) {
    // 16 points per thread, stride 256 (round even) or contiguous (round odd): the two LDS access shapes of the NTT sub-passes
    u64 v[16];
    const int base = (round & 1) ? tid * 16 : tid;
    const int st = (round & 1) ? 1 : 256;
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = lds[pad(base + e * st)];
#pragma unroll
    for (int w = 0; w < W; w++)
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = v[e] * c1 + (v[(e + 1) & 15] ^ c2);
#pragma unroll
    for (int e = 0; e < 16; e++) lds[pad(base + e * st)] = v[e];
}

// MODE 0: memory only (no compute); 1: compute only (no global traffic); 2: both
template <int K, int R, int W, int MODE> __global__ __launch_bounds__(256) void k_pipe(u64* dst, const u64* src, long long ntiles, u64 c1, u64 c2) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * K;
    ulonglong2 st[8];
    if (MODE != 1 && t0 < ntiles) {
#pragma unroll
        for (int p = 0; p < 8; p++) st[p] = *reinterpret_cast<const ulonglong2*>(src + (t0 << 12) + 2 * (tid + p * 256));
    }
    for (int k = 0; k < K; k++) {
        const long long t = t0 + k;
        if (t >= ntiles) break;
        if (MODE != 1) {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int l = 2 * (tid + p * 256);
                lds[pad(l)] = st[p].x;
                lds[pad(l + 1)] = st[p].y;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int l = 2 * (tid + p * 256);
                lds[pad(l)] = l * c1;
                lds[pad(l + 1)] = l + c2;
            }
        }
        __syncthreads();
        if (MODE != 1 && k + 1 < K && t + 1 < ntiles) {   // next tile's loads: in flight during the rounds below
#pragma unroll
            for (int p = 0; p < 8; p++) st[p] = *reinterpret_cast<const ulonglong2*>(src + ((t + 1) << 12) + 2 * (tid + p * 256));
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                compute_round<W>(lds, tid, r, c1, c2);
                __syncthreads();
            }
        }
        if (MODE != 1) {
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int l = 2 * (tid + p * 256);
                ulonglong2 w;
                w.x = lds[pad(l)];
                w.y = lds[pad(l + 1)];
                *reinterpret_cast<ulonglong2*>(dst + (t << 12) + l) = w;
            }
        } else if (lds[pad(tid)] == 0x123456789ull) dst[tid] = 1;
        __syncthreads();
    }
}

static u64 *A, *B;
static long long NT;
template <int K, int R, int W, int MODE> static float run(int wg_per_cu) {
    const size_t ldsb = (size_t)(160 * 1024 / wg_per_cu) - 512;
    if (ldsb > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pipe<K, R, W, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned grid = (unsigned)((NT + K - 1) / K);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) k_pipe<K, R, W, MODE><<<grid, 256, ldsb, 0>>>(B, A, NT, 3, 5);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = 6;
    for (int i = 0; i < reps; i++) k_pipe<K, R, W, MODE><<<grid, 256, ldsb, 0>>>(B, A, NT, 3, 5);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int R, int W> static void sweep(const char* label) {
    const double bytes = (double)NT * 4096 * 16;
    for (int wg = 3; wg <= 4; wg++) {
        const float m = run<1, R, W, 0>(wg), c = run<1, R, W, 1>(wg), b1 = run<1, R, W, 2>(wg), b4 = run<4, R, W, 2>(wg), b16 = run<16, R, W, 2>(wg);
        printf("%-22s wg/CU %d: mem-only %.3f ms (%.0f GB/s)  compute-only %.3f ms  | both, 1 tile/wg %.3f (%.0f GB/s)  4 tiles/wg pipelined %.3f (%.0f)  16 tiles/wg %.3f (%.0f)\n",
               label, wg, m, bytes / m / 1e6, c, b1, bytes / b1 / 1e6, b4, bytes / b4 / 1e6, b16, bytes / b16 / 1e6);
        fflush(stdout);
    }
}

int main() {
    const size_t n = (size_t)1 << 28;   // 2 GiB per buffer
    CK(hipMalloc(&A, n * 8));
    CK(hipMalloc(&B, n * 8));
    CK(hipMemset(A, 1, n * 8));
    NT = (long long)(n >> 12);
    sweep<2, 2>("light (2 rounds x 2)");
    sweep<2, 6>("medium (2 rounds x 6)");
    sweep<2, 12>("heavy (2 rounds x 12)");
    sweep<2, 20>("very heavy (2 x 20)");
    return 0;
}
