// shard_plan.h — spreading one task run over several devices ("one run with a device-sharding scheduler behind
// run_fhe_gpu_task").  Pure planning + the key fan-out driver, no HIP calls here: the runtime injects the device operations,
// tests/cpp/test_shard_plan.cpp a recording fake (CPU-only test).
//
// The reference's only multi-device mode is one run() per device on sub-batches (README.md:195-202,
// mega_ag_runners/gpu/gpu_wrapper.cu:148-149), which exports and uploads every evaluation key once per device and run.  Here
// the independent subgraphs the runtime already finds (task_runtime.hip, plan_pipeline: connected components over the
// non-key data, grouped into chunks) are dealt out to SHARDS, a shard being one device plus its own pair of execution lanes;
// every key is exported and uploaded ONCE, on the first device of the list, converted there, and copied device-to-device
// to each other distinct device.  A device may appear more than once in the list (two logical shards on one device: the CPU
// test's and the one-GPU box's way to run the scheduler): shards of one device share that device's key copy.
#pragma once
#include <cstddef>
#include <map>
#include <stdexcept>
#include <vector>

namespace lsa {

struct ShardPlan {
    struct Shard {
        int device;   // HIP device index
        int lane0;    // first of the shard's two lanes on that device (lanes of one device are numbered across its shards)
    };
    std::vector<Shard> shards;
    std::vector<int> chunk_shard;   // chunk index -> shard index
    std::vector<int> key_devices;   // distinct devices in order of first appearance; [0] is where keys are uploaded
    int upload_device() const { return key_devices.front(); }
};

// chunks are dealt round-robin: every shard has work from the start, and a shard's consecutive chunks alternate its lanes
inline ShardPlan plan_shards(const std::vector<int>& devices, int n_chunks) {
    if (devices.empty()) throw std::invalid_argument("shard plan: empty device list");
    ShardPlan p;
    std::map<int, int> lanes_used;
    for (int d : devices) {
        if (d < 0) throw std::invalid_argument("shard plan: negative device index");
        int& used = lanes_used[d];
        if (used + 2 > 64) throw std::invalid_argument("shard plan: too many shards on one device");
        p.shards.push_back({d, used});
        used += 2;
        bool seen = false;
        for (int k : p.key_devices) seen = seen || k == d;
        if (!seen) p.key_devices.push_back(d);
    }
    const int S = (int)p.shards.size();
    p.chunk_shard.resize((size_t)(n_chunks > 0 ? n_chunks : 0));
    for (int c = 0; c < n_chunks; c++) p.chunk_shard[(size_t)c] = c % S;
    return p;
}

// how many chunks to cut `components` independent subgraphs into for S shards: at least two per shard (so that each shard's
// two lanes overlap copies with compute), at least two components per chunk, never fewer than the single-device plan's 8
inline int plan_chunk_count(size_t components, int n_shards) {
    const size_t want = (size_t)(2 * n_shards > 8 ? 2 * n_shards : 8);
    const size_t most = components / 2;
    return (int)(want < most ? want : most);
}

// Key fan-out: `src[k]` is key k on the upload device (already converted to the device form); returns [device][k] -> pointer,
// the upload device's entries being the sources themselves.  ops.alloc(device, bytes) -> pointer on that device;
// ops.peer_copy(dst, dst_device, src, src_device, bytes) enqueues the copy.
template <class Ops>
std::map<int, std::vector<void*>> fan_out_keys(const ShardPlan& plan, const std::vector<void*>& src, const std::vector<size_t>& bytes, Ops& ops) {
    if (src.size() != bytes.size()) throw std::invalid_argument("key fan-out: sizes and sources differ in count");
    std::map<int, std::vector<void*>> out;
    out[plan.upload_device()] = src;
    for (size_t i = 1; i < plan.key_devices.size(); i++) {
        const int d = plan.key_devices[i];
        std::vector<void*>& dst = out[d];
        dst.resize(src.size());
        for (size_t k = 0; k < src.size(); k++) {
            dst[k] = ops.alloc(d, bytes[k]);
            ops.peer_copy(dst[k], d, src[k], plan.upload_device(), bytes[k]);
        }
    }
    return out;
}

}  // namespace lsa
