// CPU-only unit test of the task runtime's buffer pools (lattisense_amd/csrc/buf_pool.h) with a counting fake allocator:
// a buffer released on (device, lane) is never handed to another device or lane, frees go to the owning device, near-fit
// reuse, and the cap on pooled bytes.  Reference behaviour to match: one task object run on any device,
// /root/reference/README.md:195-202, mega_ag_runners/gpu/gpu_wrapper.cu:148-149,215,332.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <stdexcept>

#include "../../lattisense_amd/csrc/buf_pool.h"

namespace {
struct Rec {
    int device;
    bool pinned;
    size_t bytes;
};
std::map<void*, Rec> live;
int n_alloc = 0, n_free = 0, wrong_device_free = 0;
size_t budget = (size_t)-1, in_use = 0;   // case 7: a device that runs out of memory

void* fake_alloc(size_t bytes, int device, bool pinned) {
    if (in_use + bytes > budget) throw std::runtime_error("out of memory");
    in_use += bytes;
    void* p = malloc(16);
    live[p] = Rec{device, pinned, bytes};
    n_alloc++;
    return p;
}
void fake_release(void* p, int device, bool pinned) {
    auto it = live.find(p);
    assert(it != live.end());
    if (it->second.device != device || it->second.pinned != pinned) wrong_device_free++;
    in_use -= it->second.bytes;
    live.erase(it);
    free(p);
    n_free++;
}
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
}  // namespace

int main() {
    using namespace lsa;
    {
        LanePools pools(BufAllocator{fake_alloc, fake_release}, /*dev cap*/ 1 << 20, /*pin cap*/ 1 << 20);
        // 1. same (device, lane): a released buffer comes back
        size_t cap = 0;
        uint64_t* a = pools.device_pool(0, 0).take(1000, &cap);
        CHECK(cap == 1000 && live.at(a).device == 0 && !live.at(a).pinned);
        pools.device_pool(0, 0).give(cap, a);
        uint64_t* a2 = pools.device_pool(0, 0).take(1000, &cap);
        CHECK(a2 == a && n_alloc == 1);
        pools.device_pool(0, 0).give(cap, a2);
        // 2. another device, another lane: never the same buffer, and allocated on its own device
        uint64_t* b = pools.device_pool(1, 0).take(1000, &cap);
        CHECK(b != a && live.at(b).device == 1);
        uint64_t* c = pools.device_pool(0, 1).take(1000, &cap);
        CHECK(c != a && c != b && live.at(c).device == 0);
        CHECK(pools.device_pool(0, 0).free_count() == 1);   // a is still pooled where it was released
        pools.device_pool(1, 0).give(cap, b);
        pools.device_pool(0, 1).give(cap, c);
        // back on device 0 lane 0 after the run on device 1: its own buffer again
        uint64_t* a3 = pools.device_pool(0, 0).take(1000, &cap);
        CHECK(a3 == a);
        pools.device_pool(0, 0).give(cap, a3);
        // 3. pinned pools are per device and separate from device memory
        uint64_t* h = pools.pinned_pool(1).take(64, &cap);
        CHECK(live.at(h).pinned && live.at(h).device == 1);
        pools.pinned_pool(1).give(cap, h);
        uint64_t* h0 = pools.pinned_pool(0).take(64, &cap);
        CHECK(h0 != h && live.at(h0).device == 0);
        pools.pinned_pool(0).give(cap, h0);
        // 4. near fit (<= 1.25x) is reused with its real capacity; a much smaller request allocates
        BufPool& p = pools.device_pool(2, 0);
        uint64_t* big = p.take(1000, &cap);
        p.give(cap, big);
        uint64_t* near = p.take(900, &cap);
        CHECK(near == big && cap == 1000);
        p.give(cap, near);
        const int before = n_alloc;
        uint64_t* small = p.take(100, &cap);
        CHECK(small != big && cap == 100 && n_alloc == before + 1);
        p.give(cap, small);
        // 5. cap on pooled bytes: least recently returned buffers are freed (on their own device)
        BufPool& q = pools.device_pool(3, 1);
        std::set<uint64_t*> got;
        for (int i = 0; i < 6; i++) {
            uint64_t* x = q.take((size_t)(40000 + 4000 * i), &cap);   // 320..480 KB each, distinct sizes
            got.insert(x);
            q.give(cap, x);
        }
        CHECK(q.free_bytes() <= (1u << 20));
        CHECK(q.free_count() < 6 && n_free > 0);
        // 6. explicit flush, then the pool still works (what the GPU test forces between two runs)
        pools.trim_all();
        CHECK(pools.device_pool(0, 0).free_count() == 0 && q.free_count() == 0);
        uint64_t* z = pools.device_pool(0, 0).take(1000, &cap);
        CHECK(live.at(z).device == 0);
        pools.device_pool(0, 0).give(cap, z);
        // 7. out of memory with reclaimable buffers pooled on the SAME device (other lane): they are given back and the
        //    allocation is retried once; a request that cannot fit even then still fails
        pools.trim_all();
        budget = in_use + 8 * 1000;
        uint64_t* l0 = pools.device_pool(5, 0).take(600, &cap);
        pools.device_pool(5, 0).give(cap, l0);                       // 4800 bytes pooled on lane 0
        uint64_t* other = pools.device_pool(6, 0).take(100, &cap);   // another device: must survive the pressure trim
        pools.device_pool(6, 0).give(cap, other);
        uint64_t* l1 = pools.device_pool(5, 1).take(900, &cap);      // 7200 bytes: only fits once lane 0's buffer is freed
        CHECK(live.at(l1).device == 5 && pools.device_pool(5, 0).free_count() == 0);
        CHECK(pools.device_pool(6, 0).free_count() == 1);
        bool threw = false;
        try {
            size_t c2;
            pools.device_pool(5, 0).take(2000, &c2);
        } catch (const std::exception&) {
            threw = true;
        }
        CHECK(threw);
        pools.device_pool(5, 1).give(cap, l1);
        budget = (size_t)-1;
    }
    // destruction frees everything that was pooled, each on its owning device
    CHECK(live.empty());
    CHECK(wrong_device_free == 0);
    CHECK(n_alloc == n_free);
    printf("OK buf_pool: %d allocations, %d frees, no cross-device hand-outs\n", n_alloc, n_free);
    return 0;
}
