// task_graph.h — in-memory task graph ("MegaAG") of one compiled FHE task, and the executor interface.
//
// Interface restated from the reference so that callers' executors keep their shape:
//   ExecutorFunc / ExecutionContext        mega_ag_runners/mega_ag.h:40-66
//   OperationType (graph + bridge ops)     mega_ag_runners/mega_ag.h:68-91, type strings mega_ag.cpp:27-47
//   DatumNode / ComputeNode properties     mega_ag_runners/mega_ag.h:99-177
// The loader (task_graph.cpp) follows MegaAG::load / from_json / insert_backend_abi_bridge_nodes /
// compute_properties (mega_ag.cpp:125-657) in behaviour; the implementation is this project's own.
// NOTE: std::function / std::any / these structs cross the `void*` executor boundary, so an
// in-tree build of the reference must compile this runtime against the same headers and libstdc++ ABI.
#pragma once
#include <any>
#include <cstdint>
#include <functional>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lattisense_task.h"
#include "mini_json.h"

// The node structs cross the executor boundary (custom executors receive `const ComputeNode& self` and may read
// custom_prop->attributes).  Every member up to `attributes` has the reference's type, order and offset
// (tests/test_abi_layout.py static_asserts it against the reference header where that is present, and against
// tests/golden/abi_offsets.json everywhere); `attributes` itself is nlohmann::json in the reference.  A build inside the
// reference tree defines LSA_WITH_NLOHMANN (and adds lib/ to the include path): the member then IS nlohmann::json, filled
// from the parsed task file, and the structs are layout-identical including sizeof.  The stand-alone build has no
// nlohmann and keeps this project's own JSON value there; executors that read attributes need the in-tree build.
#if defined(LSA_WITH_NLOHMANN)
#include "nlohmann/json.hpp"
using LsaAttrJson = nlohmann::json;
inline LsaAttrJson lsa_attr_from(const mjson::Value& v) {
    switch (v.kind) {
        case mjson::Value::Null: return nullptr;
        case mjson::Value::Bool: return v.b;
        case mjson::Value::Int: return v.is_unsigned ? LsaAttrJson((uint64_t)v.i) : LsaAttrJson(v.i);
        case mjson::Value::Float: return v.f;
        case mjson::Value::String: return v.s;
        case mjson::Value::Array: {
            LsaAttrJson a = LsaAttrJson::array();
            for (auto& e : v.arr) a.push_back(lsa_attr_from(e));
            return a;
        }
        default: {
            LsaAttrJson o = LsaAttrJson::object();
            for (auto& kv : v.obj) o[kv.first] = lsa_attr_from(kv.second);
            return o;
        }
    }
}
#else
using LsaAttrJson = mjson::Value;
inline const LsaAttrJson& lsa_attr_from(const mjson::Value& v) { return v; }
#endif

using NodeIndex = uint64_t;
struct ComputeNode;

struct ExecutionContext {
    std::any context;                  // backend-specific arithmetic context (empty for CPU-side bridge/custom nodes here)
    std::vector<std::any> other_args;  // e.g. the pre-allocated output handle (void*) for IMPORT_FROM_ABI
    template <typename T> T* get_arithmetic_context() {
        auto* p = std::any_cast<T*>(&context);
        return p ? *p : nullptr;
    }
    template <typename T> T* get_other_arg(size_t index = 0) {
        if (index >= other_args.size() || !other_args[index].has_value()) return nullptr;
        return std::any_cast<T*>(other_args[index]);
    }
};

using ExecutorFunc = std::function<void(ExecutionContext& ctx, const std::unordered_map<NodeIndex, std::any>& inputs,
                                        std::any& output, const ComputeNode& self)>;

enum class OperationType {
    UNKNOWN,
    ADD,
    SUB,
    NEGATE,
    MULTIPLY,
    RELINEARIZE,
    RESCALE,
    DROP_LEVEL,
    ROTATE_COL,
    ROTATE_ROW,
    MAC_WO_PARTIAL_SUM,
    MAC_W_PARTIAL_SUM,
    BOOTSTRAP,
    FPGA_KERNEL,
    EXPORT_TO_ABI,       // caller handle  -> C struct      (caller's executor, CPU)
    IMPORT_FROM_ABI,     // C struct       -> caller handle (caller's executor, CPU)
    LOAD_TO_BACKEND,     // C struct       -> device datum  (this library)
    STORE_FROM_BACKEND,  // device datum   -> C struct      (this library)
    // ---- everything above has the reference's enumerator values (mega_ag.h:68-91); values from here on are private to this
    // backend, never appear in a task file and never on a node handed to a caller's executor as `self` (only backend nodes
    // carry them; a custom executor that walks the graph must treat values >= BACKEND_PRIVATE_BEGIN as opaque)
    BACKEND_PRIVATE_BEGIN = 1000,
    FUSED_MULT_RELIN_RESCALE = 1000,  // mult -> relin -> rescale chain merged at load time, inputs [a, (b,) rlk]
};

struct DatumNode {
    NodeIndex index = 0;
    std::string id;
    std::vector<ComputeNode*> predecessors;
    std::vector<ComputeNode*> successors;
    bool is_input = false;
    bool is_output = false;
    DataType datum_type = TYPE_CUSTOM;
    struct FheProperty {
        int32_t level = 0;
        int32_t degree = 0;
        bool is_ntt = false;
        bool is_mform = false;
        struct ExtraProperty {
            bool is_ringt = false;
            bool is_compressed = false;
            uint32_t galois_element = 0;
        };
        std::optional<ExtraProperty> p;
        int32_t sp_level = 0;
    };
    std::optional<FheProperty> fhe_prop;
    struct CustomProperty {
        std::string type;
        LsaAttrJson attributes;
    };
    std::optional<CustomProperty> custom_prop;
};

struct ComputeNode {
    NodeIndex index = 0;
    std::string id;
    std::vector<DatumNode*> input_nodes;
    std::vector<DatumNode*> output_nodes;
    ExecutorFunc executor;   // CPU-side nodes (export / import / custom); backend nodes are dispatched by op type
    bool on_cpu = false;
    int priority = 0;
    struct ScheduleMeta {
        int top_level = 0;
        int bottom_level = 0;
    };
    ScheduleMeta sched_meta;
    struct FheProperty {
        OperationType op_type = OperationType::UNKNOWN;
        struct ExtraProperty {
            int32_t rotation_step = 0;
            int32_t sum_cnt = 0;
        };
        std::optional<ExtraProperty> p;
    };
    std::optional<FheProperty> fhe_prop;
    struct CustomProperty {
        std::string type;
        LsaAttrJson attributes;
    };
    std::optional<CustomProperty> custom_prop;

    OperationType op() const { return fhe_prop ? fhe_prop->op_type : OperationType::UNKNOWN; }
};

struct TaskGraph {
    std::unordered_map<NodeIndex, DatumNode> data;      // node addresses are stable (unordered_map never moves values)
    std::unordered_map<NodeIndex, ComputeNode> computes;
    std::vector<NodeIndex> inputs, outputs;
    mjson::Value parameter;
    Algo algo = ALGO_BFV;
    int max_top_level = 0;

    // parse <path>, insert the ABI bridge nodes for a device backend, compute levels and priorities
    static TaskGraph load_for_gpu(const std::string& json_path);

    void bind_bridge_executors(const ExecutorFunc& abi_export, const ExecutorFunc& abi_import);
    void bind_custom_executors(const std::unordered_map<std::string, ExecutorFunc>& custom);

private:
    void parse(const std::string& json_path);
    void validate_structure() const;
    void fuse_accumulations();
    void fuse_mult_relin_rescale();
    void insert_bridges();
    void link_bridge(OperationType op, const std::string& id, DatumNode* in, DatumNode* out);
    void assign_processors();
    void compute_levels();
    NodeIndex next_data = 0, next_compute = 0;
    DatumNode& clone_datum(const DatumNode& src, const std::string& id);
};
