// buf_pool.h — recycled device / pinned-host buffers of the task runtime, keyed by (device, lane).
//
// The reference runs one task object on any device, one call at a time (`task.run(..., gpu_device=d)`,
// README.md:195-202; every worker re-issues cudaSetDevice, mega_ag_runners/gpu/gpu_wrapper.cu:215,332).  A buffer is
// therefore only ever handed back to the pool of the device that allocated it, and within a device to the lane (in-order
// stream) that used it last, so reuse stays ordered by that stream.  The allocator is injected: the runtime passes
// hipMalloc / hipHostMalloc wrappers, tests/cpp/test_buf_pool.cpp a counting fake (CPU-only test of the keying).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace lsa {

struct BufAllocator {
    void* (*alloc)(size_t bytes, int device, bool pinned);   // throws on failure
    void (*release)(void* p, int device, bool pinned);       // must free on `device` whatever the current device is
};

class BufPool {
  public:
    BufPool(int device, bool pinned, BufAllocator ops, size_t max_free_bytes)
        : device_(device), pinned_(pinned), ops_(ops), max_free_(max_free_bytes) {}
    BufPool(const BufPool&) = delete;
    BufPool& operator=(const BufPool&) = delete;
    ~BufPool() { trim(0); }

    int device() const { return device_; }
    bool pinned() const { return pinned_; }

    // a buffer of at least `words` 64-bit words; *cap_words = its real capacity (give it back with that).
    // Best fit among the free buffers no larger than 1.25x the request (a long-lived task that sees many shapes reuses
    // near-fits instead of growing by one buffer per shape); otherwise a fresh allocation.
    uint64_t* take(size_t words, size_t* cap_words) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = free_.lower_bound(words);
            if (it != free_.end() && it->first <= words + words / 4) {
                uint64_t* p = it->second.ptr;
                *cap_words = it->first;
                free_bytes_ -= it->first * sizeof(uint64_t);
                free_.erase(it);
                return p;
            }
        }
        *cap_words = words;
        try {
            return static_cast<uint64_t*>(ops_.alloc(words * sizeof(uint64_t), device_, pinned_));
        } catch (...) {
            // out of memory while this handle may be sitting on gigabytes of reclaimable pooled buffers (every lane of the
            // device keeps its own, up to the cap): give them back and try once more before failing
            if (on_pressure) on_pressure();
            else trim(0);
            return static_cast<uint64_t*>(ops_.alloc(words * sizeof(uint64_t), device_, pinned_));
        }
    }
    std::function<void()> on_pressure;   // set by the owner: frees what the sibling pools of this device hold
    void give(size_t cap_words, uint64_t* p) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            free_.emplace(cap_words, Entry{p, ++seq_});
            free_bytes_ += cap_words * sizeof(uint64_t);
        }
        trim(max_free_);
    }
    // frees least-recently-returned buffers until at most `keep_bytes` stay pooled
    void trim(size_t keep_bytes) {
        for (;;) {
            uint64_t* victim = nullptr;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (free_bytes_ <= keep_bytes || free_.empty()) return;
                auto oldest = free_.begin();
                for (auto it = free_.begin(); it != free_.end(); ++it)
                    if (it->second.seq < oldest->second.seq) oldest = it;
                victim = oldest->second.ptr;
                free_bytes_ -= oldest->first * sizeof(uint64_t);
                free_.erase(oldest);
            }
            ops_.release(victim, device_, pinned_);
        }
    }
    size_t free_bytes() const {
        std::lock_guard<std::mutex> lk(mu_);
        return free_bytes_;
    }
    size_t free_count() const {
        std::lock_guard<std::mutex> lk(mu_);
        return free_.size();
    }

  private:
    struct Entry {
        uint64_t* ptr;
        uint64_t seq;
    };
    const int device_;
    const bool pinned_;
    const BufAllocator ops_;
    const size_t max_free_;
    mutable std::mutex mu_;
    std::multimap<size_t, Entry> free_;
    size_t free_bytes_ = 0;
    uint64_t seq_ = 0;
};

// one device pool per (device, lane) and one pinned pool per device, created on first use
class LanePools {
  public:
    LanePools(BufAllocator ops, size_t max_free_dev_bytes, size_t max_free_pin_bytes)
        : ops_(ops), max_dev_(max_free_dev_bytes), max_pin_(max_free_pin_bytes) {}
    BufPool& device_pool(int device, int lane) { return get(dev_, key(device, lane), device, false, max_dev_); }
    BufPool& pinned_pool(int device) { return get(pin_, device, device, true, max_pin_); }
    static int key(int device, int lane) { return 64 * device + lane; }   // lanes of one device are numbered across its shards (shard_plan.h)
    static int device_of(int key) { return key / 64; }
    void trim_all() {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto& kv : dev_) kv.second->trim(0);
        for (auto& kv : pin_) kv.second->trim(0);
    }

  private:
    BufPool& get(std::map<int, std::unique_ptr<BufPool>>& m, int k, int device, bool pinned, size_t cap) {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = m.find(k);
        if (it == m.end()) {
            it = m.emplace(k, std::make_unique<BufPool>(device, pinned, ops_, cap)).first;
            it->second->on_pressure = [this, device, pinned]() { trim_device(device, pinned); };
        }
        return *it->second;
    }
    // every pooled buffer of one kind on one device (all lanes): called by a pool whose allocation failed
    void trim_device(int device, bool pinned) {
        std::vector<BufPool*> victims;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (auto& kv : (pinned ? pin_ : dev_))
                if (kv.second->device() == device) victims.push_back(kv.second.get());
        }
        for (BufPool* p : victims) p->trim(0);
    }
    BufAllocator ops_;
    size_t max_dev_, max_pin_;
    std::mutex mu_;
    std::map<int, std::unique_ptr<BufPool>> dev_, pin_;
};

}  // namespace lsa
