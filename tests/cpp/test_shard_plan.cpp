// CPU-only unit test of the multi-device plan behind run_fhe_gpu_task (lattisense_amd/csrc/shard_plan.h) with a recording fake
// device layer: chunk -> shard assignment, lanes of shards that share a device, and the key fan-out (every key uploaded once,
// copied device-to-device exactly once per OTHER distinct device, never to the upload device, never twice to one device).
// Reference behaviour this replaces: one run per device, each exporting and uploading every key
// (/root/reference/README.md:195-202, mega_ag_runners/gpu/gpu_wrapper.cu:148-149).
#include <cstdio>
#include <cstdlib>
#include <set>
#include <tuple>

#include "../../lattisense_amd/csrc/shard_plan.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

namespace {
struct FakeOps {
    struct Copy {
        void* dst;
        int dst_dev;
        const void* src;
        int src_dev;
        size_t bytes;
    };
    std::vector<Copy> copies;
    std::vector<std::pair<int, size_t>> allocs;
    char arena[4096];
    size_t used = 0;
    void* alloc(int device, size_t bytes) {
        allocs.push_back({device, bytes});
        return &arena[used++];   // distinct addresses are all the plan needs
    }
    void peer_copy(void* dst, int dd, const void* src, int sd, size_t bytes) { copies.push_back({dst, dd, src, sd, bytes}); }
};
}  // namespace

int main() {
    using namespace lsa;
    // 1. one device: one shard on lanes 0/1, every chunk on it, no key copies
    {
        ShardPlan p = plan_shards({3}, 8);
        CHECK(p.shards.size() == 1 && p.shards[0].device == 3 && p.shards[0].lane0 == 0);
        for (int c : p.chunk_shard) CHECK(c == 0);
        CHECK(p.key_devices.size() == 1 && p.upload_device() == 3);
        FakeOps ops;
        int k0, k1;
        auto t = fan_out_keys(p, {&k0, &k1}, {100, 200}, ops);
        CHECK(ops.copies.empty() && ops.allocs.empty() && t.size() == 1 && t.at(3)[1] == &k1);
    }
    // 2. the same device twice: two shards with their own lane pairs, ONE key copy shared (no peer copy at all)
    {
        ShardPlan p = plan_shards({0, 0}, 8);
        CHECK(p.shards.size() == 2 && p.shards[0].lane0 == 0 && p.shards[1].lane0 == 2 && p.shards[1].device == 0);
        CHECK(p.key_devices.size() == 1);
        int on0 = 0, on1 = 0;
        for (int c : p.chunk_shard) (c == 0 ? on0 : on1)++;
        CHECK(on0 == 4 && on1 == 4);
        CHECK(p.chunk_shard[0] == 0 && p.chunk_shard[1] == 1);   // dealt round-robin: both shards start at once
        FakeOps ops;
        int k0;
        fan_out_keys(p, {&k0}, {64}, ops);
        CHECK(ops.copies.empty());
    }
    // 3. eight devices, 16 chunks: two chunks per shard; each of 3 keys goes to the 7 other devices exactly once, from device 0
    {
        std::vector<int> devs = {0, 1, 2, 3, 4, 5, 6, 7};
        ShardPlan p = plan_shards(devs, plan_chunk_count(1024, 8));
        CHECK(p.chunk_shard.size() == 16);
        std::vector<int> per(8, 0);
        for (int c : p.chunk_shard) per[(size_t)c]++;
        for (int n : per) CHECK(n == 2);
        FakeOps ops;
        int k[3];
        auto t = fan_out_keys(p, {&k[0], &k[1], &k[2]}, {10, 20, 30}, ops);
        CHECK(ops.copies.size() == 21 && ops.allocs.size() == 21);
        std::set<std::tuple<int, const void*>> seen;
        for (auto& c : ops.copies) {
            CHECK(c.src_dev == 0 && c.dst_dev != 0);
            CHECK(seen.insert({c.dst_dev, c.src}).second);       // a key reaches a device once
            CHECK(c.bytes == (c.src == &k[0] ? 10u : c.src == &k[1] ? 20u : 30u));
        }
        for (int d = 1; d < 8; d++) CHECK(t.at(d).size() == 3 && t.at(d)[0] != &k[0]);
        CHECK(t.at(0)[2] == &k[2]);
    }
    // 4. mixed list {2, 5, 2, 5, 5}: lanes count up per device, keys fan out to ONE other device
    {
        ShardPlan p = plan_shards({2, 5, 2, 5, 5}, 10);
        CHECK(p.shards[2].device == 2 && p.shards[2].lane0 == 2);
        CHECK(p.shards[3].lane0 == 2 && p.shards[4].lane0 == 4);
        CHECK(p.key_devices.size() == 2 && p.upload_device() == 2 && p.key_devices[1] == 5);
        FakeOps ops;
        int k0;
        fan_out_keys(p, {&k0}, {8}, ops);
        CHECK(ops.copies.size() == 1 && ops.copies[0].dst_dev == 5 && ops.copies[0].src_dev == 2);
    }
    // 5. chunk counts: never fewer than the single-device plan's 8, two per shard, at least two components per chunk
    CHECK(plan_chunk_count(1024, 1) == 8 && plan_chunk_count(1024, 8) == 16 && plan_chunk_count(10, 8) == 5 && plan_chunk_count(3, 1) == 1);
    // 6. bad lists are refused
    bool threw = false;
    try {
        plan_shards({}, 4);
    } catch (const std::invalid_argument&) {
        threw = true;
    }
    CHECK(threw);
    threw = false;
    try {
        plan_shards({0, -1}, 4);
    } catch (const std::invalid_argument&) {
        threw = true;
    }
    CHECK(threw);
    printf("OK shard_plan\n");
    return 0;
}
