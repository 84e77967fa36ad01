// lattisense_task.hpp — C++ companion of lattisense_task.h: the SDK-level class a C++ application uses.
//
// Mirrors lattisense::FheTaskGpu of the reference SDK (cxx_sdk_v2/cxx_fhe_task.h:132-148, cxx_fhe_task_gpu.cpp:30-117):
//   FheTaskGpu task(project_path);                         // throws std::runtime_error if the task cannot be loaded
//   uint64_t ns = task.run(args, progress_cb, gpu_device); // elapsed nanoseconds; throws on failure
// Differences, because this front-end has no host crypto library behind it: arguments are plain host limb buffers
// (HostCiphertext / HostPlaintext / HostKeySwitchKey / HostGaloisKey) instead of Lattigo `Handle`s, and evaluation keys
// are passed explicitly as trailing arguments with the ids the SDK uses (rlk_ntt, glk_ntt; cxx_argument.h:185,205).
#pragma once
#include <chrono>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lattisense_amd.h"
#include "lattisense_task.h"

namespace lattisense {

// [degree+1][level+1][n] residues (BFV: coefficient domain, CKKS: NTT domain)
struct HostCiphertext {
    std::vector<uint64_t> data;
    lsa_host_ciphertext h{};
    HostCiphertext(int degree, int level, int n) : data((size_t)(degree + 1) * (level + 1) * n) { bind(degree, level, n); }
    HostCiphertext(const HostCiphertext& o) : data(o.data) { bind(o.h.degree, o.h.level, o.h.n); }
    HostCiphertext(HostCiphertext&& o) noexcept : data(std::move(o.data)) { bind(o.h.degree, o.h.level, o.h.n); }
    uint64_t* limb(int poly, int j) { return data.data() + ((size_t)poly * (h.level + 1) + j) * h.n; }

private:
    void bind(int degree, int level, int n) {
        h.degree = degree;
        h.level = level;
        h.n = n;
        h.data = data.data();
    }
};

struct HostPlaintext {  // [level+1][n]; a ring-t plaintext has level 0
    std::vector<uint64_t> data;
    lsa_host_plaintext h{};
    HostPlaintext(int level, int n) : data((size_t)(level + 1) * n) {
        h.level = level;
        h.n = n;
        h.data = data.data();
    }
    HostPlaintext(const HostPlaintext&) = delete;
};

struct HostKeySwitchKey {  // compact [beta][2][level+1+n_special][n], NTT domain, non-Montgomery
    std::vector<uint64_t> data;
    lsa_host_kskey h{};
    HostKeySwitchKey(int level, int n_special, int n)
        : data((size_t)((level + 1 + n_special - 1) / n_special) * 2 * (level + 1 + n_special) * n) {
        h.level = level;
        h.n_special = n_special;
        h.n = n;
        h.data = data.data();
    }
    HostKeySwitchKey(const HostKeySwitchKey&) = delete;
};

struct HostGaloisKey {
    std::vector<uint64_t> elements;
    std::vector<lsa_host_kskey> keys;
    lsa_host_galois_key h{};
    void add(uint64_t galois_element, HostKeySwitchKey& k) {
        elements.push_back(galois_element);
        keys.push_back(k.h);
        h.n_keys = (int)keys.size();
        h.galois_elements = elements.data();
        h.keys = keys.data();
    }
};

// one task argument: id + the handles of its objects (CxxVectorArgument, cxx_argument.h:108-133)
struct TaskArgument {
    std::string id;
    DataType type;
    std::vector<void*> handles;
    int level = 0;
    TaskArgument(std::string id_, std::vector<HostCiphertext>& v) : id(std::move(id_)), type(TYPE_CIPHERTEXT) {
        for (auto& c : v) handles.push_back(&c.h);
        if (!v.empty()) level = v[0].h.level;
    }
    TaskArgument(std::string id_, std::vector<HostPlaintext*>& v) : id(std::move(id_)), type(TYPE_PLAINTEXT) {
        for (auto* p : v) handles.push_back(&p->h);
        if (!v.empty()) level = v[0]->h.level;
    }
    TaskArgument(std::string id_, HostKeySwitchKey& k) : id(std::move(id_)), type(TYPE_RELIN_KEY), level(k.h.level) {
        handles.push_back(&k.h);
    }
    TaskArgument(std::string id_, HostGaloisKey& k) : id(std::move(id_)), type(TYPE_GALOIS_KEY) { handles.push_back(&k.h); }
};

using ProgressCallback = std::function<void(int completed, int total)>;

class FheTaskGpu {
public:
    explicit FheTaskGpu(const std::string& project_path) {
        handle_ = create_fhe_gpu_task(project_path.c_str());
        if (!handle_) throw std::runtime_error(lsa_last_error());
        if (lsa_frontend_bind(handle_) != 0) {
            std::string msg = lsa_last_error();
            release_fhe_gpu_task(handle_);
            throw std::runtime_error(msg);
        }
    }
    ~FheTaskGpu() {
        if (handle_) release_fhe_gpu_task(handle_);
    }
    FheTaskGpu(const FheTaskGpu&) = delete;
    FheTaskGpu& operator=(const FheTaskGpu&) = delete;

    // inputs in task order with the evaluation keys last; outputs are pre-allocated at the right level/degree.
    // Returns the elapsed time in nanoseconds (cxx_fhe_task_gpu.cpp:110-116).
    uint64_t run(std::vector<TaskArgument>& inputs, std::vector<TaskArgument>& outputs, ProgressCallback cb = nullptr,
                 int gpu_device = 0) {
        std::vector<CArgument> in, out;
        for (auto& a : inputs) in.push_back(CArgument{a.id.c_str(), a.type, a.handles.data(), a.level, (int)a.handles.size()});
        for (auto& a : outputs) out.push_back(CArgument{a.id.c_str(), a.type, a.handles.data(), a.level, (int)a.handles.size()});
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = run_fhe_gpu_task(handle_, in.data(), in.size(), out.data(), out.size(),
                                        cb ? &FheTaskGpu::trampoline : nullptr, cb ? &cb : nullptr, gpu_device);
        if (rc != 0) throw std::runtime_error(lsa_last_error());
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }

private:
    static void trampoline(int completed, int total, void* user) { (*static_cast<ProgressCallback*>(user))(completed, total); }
    fhe_task_handle handle_ = nullptr;
};

}  // namespace lattisense
