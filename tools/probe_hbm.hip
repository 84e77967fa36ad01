// probe_hbm.hip — HBM ceilings for the access shapes of the NTT passes (stand-alone; build: tools/build_probes.sh).
// Prints GB/s (read bytes + written bytes) for: streaming copies with U loads in flight per thread, one-trip and
// persistent grids, temporal / non-temporal, out-of-place / in-place; the NTT pass shapes (4096-point tile through LDS,
// contiguous = second pass, 128-B column segments = first pass) at 2..5 workgroups per CU; read-only and write-only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ ulonglong2 ldv(const ulonglong2* p) {
    if (NT) {
        ulonglong2 v;
        v.x = __builtin_nontemporal_load(&p->x);
        v.y = __builtin_nontemporal_load(&p->y);
        return v;
    }
    return *p;
}
template <bool NT> __device__ __forceinline__ void stv(ulonglong2* p, ulonglong2 v) {
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
    } else *p = v;
}

template <int U, bool NT> __global__ __launch_bounds__(256) void k_copy(ulonglong2* dst, const ulonglong2* src, size_t n2) {
    for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n2; base += (size_t)gridDim.x * 256 * U) {
        ulonglong2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ldv<NT>(src + base + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) stv<NT>(dst + base + u * 256, v[u]);
    }
}
template <int U> __global__ __launch_bounds__(256) void k_read(u64* sink, const ulonglong2* src, size_t n2) {
    u64 acc = 0;
    for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n2; base += (size_t)gridDim.x * 256 * U) {
        ulonglong2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = src[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u].x + v[u].y;
    }
    if (acc == 0x1234567812345678ull) sink[0] = acc;
}
template <int U> __global__ __launch_bounds__(256) void k_write(ulonglong2* dst, size_t n2) {
    ulonglong2 v;
    v.x = threadIdx.x;
    v.y = blockIdx.x;
    for (size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x; base < n2; base += (size_t)gridDim.x * 256 * U) {
#pragma unroll
        for (int u = 0; u < U; u++) dst[base + u * 256] = v;
    }
}

// the NTT pass shape: one workgroup = one 4096-point tile of one limb of N = 2^16 points, staged through LDS.
// COL: first-pass tile (256 rows 2 KiB apart x 16 consecutive points = 128-B segments); else 32 KiB contiguous.
template <bool COL> __global__ __launch_bounds__(256, 4) void k_tile(u64* dst, const u64* src, size_t limbs) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const size_t limb = blockIdx.x >> 4;
    const int tile = blockIdx.x & 15;
    const u64* g = src + (limb << 16);
    u64* o = dst + (limb << 16);
    ulonglong2 v[8];
    int xs[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int l = 2 * (tid + p * 256);
        xs[p] = COL ? ((l >> 4) << 8) + (tile << 4) + (l & 15) : (tile << 12) + l;
        v[p] = *reinterpret_cast<const ulonglong2*>(g + xs[p]);
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int l = 2 * (tid + p * 256);
        lds[l + (l >> 4)] = v[p].x;
        lds[l + 1 + ((l + 1) >> 4)] = v[p].y;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int l = 2 * (tid + p * 256);
        ulonglong2 w;
        w.x = lds[l + (l >> 4)] + 1;
        w.y = lds[l + 1 + ((l + 1) >> 4)] + 1;
        *reinterpret_cast<ulonglong2*>(o + xs[p]) = w;
    }
}

// the launch order of the real passes: batch item fastest.  Workgroup b -> batch item b % batch, tile (b / batch) of that item;
// items are `stride` words apart (the operator workspaces use strides that are multiples of 2^16 words = 512 KiB)
template <bool COL> __global__ __launch_bounds__(256, 4) void k_tile_batch(u64* dst, const u64* src, int batch, long long stride) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x % batch;
    const unsigned rt = blockIdx.x / batch;
    const size_t limb = rt >> 4;
    const int tile = rt & 15;
    const u64* g = src + (long long)b * stride + (limb << 16);
    u64* o = dst + (long long)b * stride + (limb << 16);
    ulonglong2 v[8];
    int xs[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int l = 2 * (tid + p * 256);
        xs[p] = COL ? ((l >> 4) << 8) + (tile << 4) + (l & 15) : (tile << 12) + l;
        v[p] = *reinterpret_cast<const ulonglong2*>(g + xs[p]);
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int l = 2 * (tid + p * 256);
        lds[l + (l >> 4)] = v[p].x;
        lds[l + 1 + ((l + 1) >> 4)] = v[p].y;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int l = 2 * (tid + p * 256);
        ulonglong2 w;
        w.x = lds[l + (l >> 4)] + 1;
        w.y = lds[l + 1 + ((l + 1) >> 4)] + 1;
        *reinterpret_cast<ulonglong2*>(o + xs[p]) = w;
    }
}

static float timed(void (*fn)(void*), void* ctx, int reps = 8) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    fn(ctx);
    fn(ctx);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) fn(ctx);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

struct Ctx {
    u64 *a, *b;
    size_t n;   // u64 elements
    unsigned grid;
};
#define RUN(label, bytes, ...)                                                           \
    do {                                                                                 \
        auto f = [](void* p) { Ctx& c = *(Ctx*)p; (void)c; __VA_ARGS__; };                      \
        float ms = timed(f, &cx);                                                        \
        printf("%-58s %8.0f GB/s  (%.3f ms)\n", label, (double)(bytes) / ms / 1e6, ms);  \
        fflush(stdout);                                                                  \
    } while (0)

int main() {
    Ctx cx;
    cx.n = (size_t)1 << 28;   // 2 GiB per buffer
    CK(hipMalloc(&cx.a, cx.n * 8));
    CK(hipMalloc(&cx.b, cx.n * 8));
    CK(hipMemset(cx.a, 1, cx.n * 8));
    CK(hipMemset(cx.b, 2, cx.n * 8));
    const size_t n2 = cx.n / 2;
    const double B = (double)cx.n * 8;
#define ONE(U) (unsigned)((c.n / 2) / (256 * (U)))
    RUN("copy U=1 one-trip", 2 * B, k_copy<1, false><<<dim3(ONE(1)), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=2 one-trip", 2 * B, k_copy<2, false><<<dim3(ONE(2)), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=4 one-trip", 2 * B, k_copy<4, false><<<dim3(ONE(4)), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=8 one-trip", 2 * B, k_copy<8, false><<<dim3(ONE(8)), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=1 persistent 2048 wg", 2 * B, k_copy<1, false><<<dim3(2048), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=4 persistent 2048 wg", 2 * B, k_copy<4, false><<<dim3(2048), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=8 persistent 1024 wg", 2 * B, k_copy<8, false><<<dim3(1024), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=4 one-trip non-temporal", 2 * B, k_copy<4, true><<<dim3(ONE(4)), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=8 one-trip non-temporal", 2 * B, k_copy<8, true><<<dim3(ONE(8)), dim3(256), 0, 0>>>((ulonglong2*)c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=4 one-trip IN PLACE", 2 * B, k_copy<4, false><<<dim3(ONE(4)), dim3(256), 0, 0>>>((ulonglong2*)c.a, (const ulonglong2*)c.a, c.n / 2));
    RUN("copy U=8 one-trip IN PLACE", 2 * B, k_copy<8, false><<<dim3(ONE(8)), dim3(256), 0, 0>>>((ulonglong2*)c.a, (const ulonglong2*)c.a, c.n / 2));
    RUN("read-only U=4 one-trip", B, k_read<4><<<dim3(ONE(4)), dim3(256), 0, 0>>>(c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("read-only U=8 one-trip", B, k_read<8><<<dim3(ONE(8)), dim3(256), 0, 0>>>(c.b, (const ulonglong2*)c.a, c.n / 2));
    RUN("write-only U=4 one-trip", B, k_write<4><<<dim3(ONE(4)), dim3(256), 0, 0>>>((ulonglong2*)c.b, c.n / 2));
    const size_t limbs = cx.n >> 16;
    (void)limbs;
    const size_t ldsb = (4096 + 256 + 16) * 8;
    (void)ldsb;
    // workgroups per CU are set through the LDS allocation: 160 KiB / W
#define TILE(COL, W, DST) k_tile<COL><<<dim3((unsigned)((c.n >> 16) * 16)), dim3(256), (size_t)(160 * 1024 / (W)) - 512, 0>>>(DST, (const u64*)c.a, c.n >> 16)
    RUN("tile contiguous, 1 wg/CU-limit, out of place", 2 * B, TILE(false, 1, c.b));
    RUN("tile contiguous, 2 wg/CU-limit, out of place", 2 * B, TILE(false, 2, c.b));
    RUN("tile contiguous, 3 wg/CU-limit, out of place", 2 * B, TILE(false, 3, c.b));
    RUN("tile contiguous, 4 wg/CU-limit, out of place", 2 * B, TILE(false, 4, c.b));
    RUN("tile contiguous, 4 wg/CU-limit, IN PLACE", 2 * B, TILE(false, 4, c.a));
    RUN("tile columns 128B, 3 wg/CU-limit, out of place", 2 * B, TILE(true, 3, c.b));
    RUN("tile columns 128B, 4 wg/CU-limit, out of place", 2 * B, TILE(true, 4, c.b));
    RUN("tile columns 128B, 4 wg/CU-limit, IN PLACE", 2 * B, TILE(true, 4, c.a));
    // batch-fastest order, 64 items of `rows` limbs each, item stride = rows limbs (+ pad words)
    {
        const int batch = 64;
        for (int rows : {26, 32, 68}) {
            for (long long pad : {0LL, 512LL, 2080LL, 8192LL + 272}) {
                const long long stride = ((long long)rows << 16) + pad;
                if ((size_t)stride * batch > cx.n) continue;
                const unsigned grid = (unsigned)(batch * rows * 16);
                const double bytes = 2.0 * batch * rows * 65536 * 8;
                for (int col = 0; col < 2; col++) {
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                    auto go = [&]() {
                        if (col) k_tile_batch<true><<<grid, 256, 40 * 1024 - 512, 0>>>(cx.a, cx.a, batch, stride);
                        else k_tile_batch<false><<<grid, 256, 40 * 1024 - 512, 0>>>(cx.a, cx.a, batch, stride);
                    };
                    go();
                    go();
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, 0));
                    for (int i = 0; i < 8; i++) go();
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("batch-fastest in place, %s, %d limbs/item, pad %5lld words: %8.0f GB/s\n", col ? "columns 128B" : "contiguous  ", rows, pad, bytes / (ms / 8) / 1e6);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
