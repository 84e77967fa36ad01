// Test shim: lets a Python test supply the CUSTOM executors of a GPU task (bind_gpu_task_custom_executors,
// mega_ag_runners/wrapper.h:75-78) the way a plug-in does -- executors that capture their own state and ignore
// ExecutionContext (plug-in/SEAL/acc/abi_bridge_executors.h:73-74).  Each ExecutorFunc built here gathers the node's input
// handles (the native front-end's handles are std::shared_ptr<void> -> lsa_host_* structs or the caller's own custom data),
// calls the test's C callback and wraps the handle it returns.  The callbacks run concurrently from the runtime's CPU-pool
// threads, as in the reference (gpu_wrapper.cu:175-177).
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../lattisense_amd/csrc/task_graph.h"

extern "C" {
// returns 0 and sets *out_handle, or non-zero (the run then fails with the node's id in lsa_last_error)
typedef int (*lsa_test_custom_cb)(const char* type, const char* node_id, int attr_level, void** in_handles, int n_in,
                                  void** out_handle, void* user);

int lsa_test_bind_custom(fhe_task_handle task, const char** types, int n_types, lsa_test_custom_cb cb, void* user) {
    std::vector<ExecutorFunc> execs;
    std::vector<void*> ptrs;
    execs.reserve((size_t)n_types);
    for (int i = 0; i < n_types; i++) {
        const std::string type = types[i];
        execs.emplace_back([cb, user, type](ExecutionContext&, const std::unordered_map<NodeIndex, std::any>& inputs, std::any& output,
                                            const ComputeNode& self) {
            if (!self.custom_prop || self.custom_prop->type != type) throw std::runtime_error("custom executor bound to the wrong node");
            std::vector<void*> in;
            for (const DatumNode* d : self.input_nodes) {
                const auto* h = std::any_cast<std::shared_ptr<void>>(&inputs.at(d->index));
                if (!h) throw std::runtime_error("custom node input '" + d->id + "' is not a handle");
                in.push_back(h->get());
            }
            int level = -1;
#if !defined(LSA_WITH_NLOHMANN)
            if (self.custom_prop->attributes.contains("level")) level = (int)self.custom_prop->attributes["level"].as_int();
#endif
            void* out = nullptr;
            if (cb(type.c_str(), self.id.c_str(), level, in.data(), (int)in.size(), &out, user) != 0 || !out)
                throw std::runtime_error("custom executor '" + type + "' failed");
            output = std::shared_ptr<void>(out, [](void*) {});   // the test owns what it returned
        });
    }
    for (auto& e : execs) ptrs.push_back(&e);
    bind_gpu_task_custom_executors(task, types, ptrs.data(), (uint64_t)n_types);   // copies the std::function objects
    return 0;
}
}
