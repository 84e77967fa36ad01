// task_graph.cpp — loader for mega_ag.json: parsing, ABI-bridge insertion for a device backend, level computation.
// Behaviour follows mega_ag_runners/mega_ag.cpp:125-657 (see task_graph.h); written for this runtime's level-batched
// scheduler (top_level drives execution order, bottom_level is kept as the reference's priority).
#include "task_graph.h"

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <queue>
#include <stdexcept>

namespace {

DataType datum_type_of(const std::string& s) {
    if (s == "ct" || s == "ct3") return TYPE_CIPHERTEXT;
    if (s == "pt" || s == "pt_mul" || s == "pt_ringt") return TYPE_PLAINTEXT;
    if (s == "rlk") return TYPE_RELIN_KEY;
    if (s == "glk") return TYPE_GALOIS_KEY;
    if (s == "swk") return TYPE_SWITCH_KEY;
    throw std::runtime_error("unknown datum type '" + s + "' in mega_ag.json");
}

OperationType op_type_of(const std::string& s) {
    static const std::pair<const char*, OperationType> table[] = {
        {"add", OperationType::ADD},
        {"sub", OperationType::SUB},
        {"neg", OperationType::NEGATE},
        {"mult", OperationType::MULTIPLY},
        {"relin", OperationType::RELINEARIZE},
        {"rescale", OperationType::RESCALE},
        {"drop_level", OperationType::DROP_LEVEL},
        {"rotate_row", OperationType::ROTATE_ROW},
        {"rotate_col", OperationType::ROTATE_COL},
        {"cmp_sum", OperationType::MAC_WO_PARTIAL_SUM},
        {"cmpac_sum", OperationType::MAC_W_PARTIAL_SUM},
        {"bootstrap", OperationType::BOOTSTRAP},
        {"fpga_kernel", OperationType::FPGA_KERNEL},
    };
    for (auto& e : table)
        if (s == e.first) return e.second;
    throw std::runtime_error("unknown operation type '" + s + "' in mega_ag.json");
}

bool is_custom_json(const mjson::Value& v) { return v.contains("is_custom") && v["is_custom"].as_bool(); }

const mjson::Value& member_of_kind(const mjson::Value& root, const char* key, mjson::Value::Kind kind) {
    const mjson::Value& v = root[key];
    if (v.kind != kind)
        throw std::runtime_error(std::string("mega_ag.json: '") + key + "' is not " + (kind == mjson::Value::Array ? "an array" : "an object"));
    return v;
}

// a node index: a non-negative integer (a float or a negative number is a damaged file, not an index to truncate)
NodeIndex index_of(const mjson::Value& v) {
    if (v.kind != mjson::Value::Int || (!v.is_unsigned && v.i < 0)) throw std::runtime_error("mega_ag.json: node index is not a non-negative integer");
    return (NodeIndex)v.as_u64();
}

NodeIndex index_of(const std::string& key) {
    if (key.empty() || key.size() > 19 || key.find_first_not_of("0123456789") != std::string::npos)
        throw std::runtime_error("mega_ag.json: node key '" + key + "' is not a non-negative integer");
    return (NodeIndex)std::stoull(key);
}

}  // namespace

TaskGraph TaskGraph::load_for_gpu(const std::string& json_path) {
    TaskGraph g;
    g.parse(json_path);
    g.validate_structure();
    if (!getenv("LSA_NO_GRAPH_FUSION")) {
        g.fuse_accumulations();
        g.fuse_mult_relin_rescale();
    }
    g.insert_bridges();
    g.assign_processors();
    g.compute_levels();
    return g;
}

void TaskGraph::parse(const std::string& json_path) {
    mjson::Value root;
    try {
        root = mjson::parse_file(json_path);
    } catch (const std::exception& e) {
        throw std::runtime_error(std::string("Cannot open MegaAG file ") + json_path + " (" + e.what() + ")");
    }
    const std::string& algo_s = root["algorithm"].as_string();
    if (algo_s == "BFV") algo = ALGO_BFV;
    else if (algo_s == "CKKS") algo = ALGO_CKKS;
    else throw std::runtime_error("Unknown algorithm: " + algo_s);
    parameter = root["parameter"];

    for (auto& kv : member_of_kind(root, "data", mjson::Value::Object).obj) {
        const mjson::Value& v = kv.second;
        DatumNode n;
        n.index = index_of(kv.first);
        n.id = v["id"].as_string();
        const std::string& type_s = v["type"].as_string();
        if (is_custom_json(v)) {
            DatumNode::CustomProperty cp;
            cp.type = type_s;
            if (v.contains("attributes")) cp.attributes = lsa_attr_from(v["attributes"]);
            n.custom_prop = cp;
        } else {
            DatumNode::FheProperty fp;
            const int64_t lvl = v["level"].as_int(), deg = v["degree"].as_int();
            if (lvl != (int32_t)lvl || deg != (int32_t)deg) throw std::runtime_error("datum " + n.id + ": level / degree out of range");
            fp.level = (int32_t)lvl;
            fp.degree = (int32_t)deg;
            fp.is_ntt = v["is_ntt"].as_bool();
            fp.is_mform = v["is_mform"].as_bool();
            fp.sp_level = v.contains("sp_level") ? (int32_t)v["sp_level"].as_int() : -1;
            n.datum_type = datum_type_of(type_s);
            if (n.datum_type == TYPE_GALOIS_KEY) {
                DatumNode::FheProperty::ExtraProperty ep;
                ep.galois_element = (uint32_t)v["galois_element"].as_u64();
                fp.p = ep;
            } else if (type_s == "pt_ringt") {
                DatumNode::FheProperty::ExtraProperty ep;
                ep.is_ringt = true;
                fp.p = ep;
            }
            n.fhe_prop = fp;
        }
        next_data = std::max(next_data, n.index + 1);
        data.emplace(n.index, std::move(n));
    }

    for (auto& kv : member_of_kind(root, "compute", mjson::Value::Object).obj) {
        const mjson::Value& v = kv.second;
        ComputeNode c;
        c.index = index_of(kv.first);
        c.id = v["id"].as_string();
        const std::string& type_s = v["type"].as_string();
        if (is_custom_json(v)) {
            ComputeNode::CustomProperty cp;
            cp.type = type_s;
            if (v.contains("attributes")) cp.attributes = lsa_attr_from(v["attributes"]);
            c.custom_prop = cp;
        } else {
            ComputeNode::FheProperty fp;
            fp.op_type = op_type_of(type_s);
            if (fp.op_type == OperationType::ROTATE_COL) {
                ComputeNode::FheProperty::ExtraProperty ep;
                ep.rotation_step = (int32_t)v["step"].as_int();
                fp.p = ep;
            } else if (fp.op_type == OperationType::MAC_WO_PARTIAL_SUM || fp.op_type == OperationType::MAC_W_PARTIAL_SUM) {
                ComputeNode::FheProperty::ExtraProperty ep;
                ep.sum_cnt = (int32_t)v["sum_cnt"].as_int();
                fp.p = ep;
            }
            c.fhe_prop = fp;
        }
        auto datum = [&](const mjson::Value& e) {
            auto it = data.find(index_of(e));
            if (it == data.end()) throw std::runtime_error("compute node " + c.id + " names datum " + std::to_string(index_of(e)) + ", which the graph does not have");
            return &it->second;
        };
        for (auto& e : member_of_kind(v, "inputs", mjson::Value::Array).arr) c.input_nodes.push_back(datum(e));
        for (auto& e : member_of_kind(v, "outputs", mjson::Value::Array).arr) c.output_nodes.push_back(datum(e));
        next_compute = std::max(next_compute, c.index + 1);
        computes.emplace(c.index, std::move(c));
    }
    for (auto& kv : computes) {
        for (auto* d : kv.second.input_nodes) d->successors.push_back(&kv.second);
        for (auto* d : kv.second.output_nodes) d->predecessors.push_back(&kv.second);
    }
    auto terminal = [&](const mjson::Value& e, const char* what) -> DatumNode& {
        auto it = data.find(index_of(e));
        if (it == data.end()) throw std::runtime_error(std::string("task ") + what + " " + std::to_string(index_of(e)) + " is not a datum of the graph");
        return it->second;
    };
    for (auto& e : member_of_kind(root, "inputs", mjson::Value::Array).arr) {
        inputs.push_back(index_of(e));
        terminal(e, "input").is_input = true;
    }
    for (auto& e : member_of_kind(root, "outputs", mjson::Value::Array).arr) {
        outputs.push_back(index_of(e));
        terminal(e, "output").is_output = true;
    }
}

// What every later stage indexes without looking (input_nodes[0], output_nodes[0], predecessors[0]): each compute node has
// exactly one output (one std::any per executor call, mega_ag.h:63-66) and the operands its operator takes; each datum has
// one producer, or none when it is a task input.  The reference relies on its frontend to emit nothing else
// (frontend/custom_task.py); a task file is an input of this library, so it is checked here.
void TaskGraph::validate_structure() const {
    auto bad = [](const ComputeNode& c, const std::string& why) {
        throw std::runtime_error("compute node " + c.id + " (" + std::to_string(c.index) + "): " + why);
    };
    for (auto& kv : computes) {
        const ComputeNode& c = kv.second;
        if (c.output_nodes.size() != 1) bad(c, "has " + std::to_string(c.output_nodes.size()) + " outputs, expected 1");
        if (c.custom_prop) continue;   // a caller's executor defines its own operands
        const size_t n = c.input_nodes.size();
        size_t lo = 1, hi = 1;
        switch (c.op()) {
            case OperationType::ADD:
            case OperationType::SUB:
            case OperationType::MULTIPLY: hi = 2; break;
            case OperationType::RELINEARIZE:
            case OperationType::ROTATE_COL:
            case OperationType::ROTATE_ROW: lo = hi = 2; break;
            case OperationType::MAC_WO_PARTIAL_SUM:
            case OperationType::MAC_W_PARTIAL_SUM: lo = 2, hi = SIZE_MAX; break;
            case OperationType::BOOTSTRAP: lo = 5, hi = SIZE_MAX; break;
            default: break;   // neg, rescale, drop_level: one operand
        }
        if (n < lo || n > hi) bad(c, "has " + std::to_string(n) + " inputs");
        for (const DatumNode* d : c.input_nodes)
            if (!d->fhe_prop) bad(c, "operand " + d->id + " is custom data");
        if (c.input_nodes[0]->datum_type != TYPE_CIPHERTEXT) bad(c, "first operand " + c.input_nodes[0]->id + " is not a ciphertext");
        if (!c.output_nodes[0]->fhe_prop || c.output_nodes[0]->datum_type != TYPE_CIPHERTEXT) bad(c, "result " + c.output_nodes[0]->id + " is not a ciphertext");
        if (lo == 2 && hi == 2) {
            const DataType k = c.input_nodes[1]->datum_type;
            if (k != TYPE_RELIN_KEY && k != TYPE_GALOIS_KEY && k != TYPE_SWITCH_KEY) bad(c, "second operand " + c.input_nodes[1]->id + " is not a key");
        }
        // Relations between an operator's operands and its result.  The runtime sizes the result slab from the result's
        // declared level / degree and launches over the first operand's rows (task_runtime.hip, run_gpu_bucket): a file
        // that declares, say, an `add` whose result is lower than its operands, or a second operand of lower level or
        // degree, would make a kernel write or read past a slab.  What the frontend emits always satisfies these
        // (frontend/custom_task.py:971-1371: results are created from the operands' level and degree).
        const DatumNode::FheProperty& a = *c.input_nodes[0]->fhe_prop;
        const DatumNode::FheProperty& r = *c.output_nodes[0]->fhe_prop;
        auto is_ct = [](const DatumNode* d) { return d->datum_type == TYPE_CIPHERTEXT; };
        auto is_pt = [](const DatumNode* d) { return d->datum_type == TYPE_PLAINTEXT; };
        auto ringt = [](const DatumNode* d) { return d->fhe_prop->p && d->fhe_prop->p->is_ringt; };
        auto same_ct = [&](const DatumNode* d, const char* what) {
            if (d->fhe_prop->level != a.level || d->fhe_prop->degree != a.degree)
                bad(c, std::string(what) + " " + d->id + " has level/degree " + std::to_string(d->fhe_prop->level) + "/" + std::to_string(d->fhe_prop->degree) +
                           ", the first operand " + std::to_string(a.level) + "/" + std::to_string(a.degree));
        };
        auto plain_ok = [&](const DatumNode* d) {   // a full plaintext lives at the ciphertext's level; a ring-t one is a single limb
            if (!ringt(d) && d->fhe_prop->level != a.level)
                bad(c, "plaintext " + d->id + " is at level " + std::to_string(d->fhe_prop->level) + ", the ciphertext at " + std::to_string(a.level));
        };
        auto key_ok = [&](const DatumNode* d) {
            if (d->fhe_prop->level < a.level) bad(c, "key " + d->id + " was exported at level " + std::to_string(d->fhe_prop->level) + ", below the ciphertext's " + std::to_string(a.level));
        };
        auto result = [&](int level, int degree) {
            if (r.level != level || r.degree != degree)
                bad(c, "result " + c.output_nodes[0]->id + " is declared at level/degree " + std::to_string(r.level) + "/" + std::to_string(r.degree) +
                           ", the operation produces " + std::to_string(level) + "/" + std::to_string(degree));
        };
        switch (c.op()) {
            case OperationType::ADD:
            case OperationType::SUB:
                if (n == 2 && is_ct(c.input_nodes[1])) same_ct(c.input_nodes[1], "second operand");
                else if (n == 2 && is_pt(c.input_nodes[1])) plain_ok(c.input_nodes[1]);
                else if (n == 2) bad(c, "second operand " + c.input_nodes[1]->id + " is neither a ciphertext nor a plaintext");
                result(a.level, a.degree);
                break;
            case OperationType::NEGATE: result(a.level, a.degree); break;
            case OperationType::MULTIPLY:
                if (n == 2 && is_pt(c.input_nodes[1])) {
                    plain_ok(c.input_nodes[1]);
                    result(a.level, a.degree);
                } else {
                    if (n == 2 && !is_ct(c.input_nodes[1])) bad(c, "second operand " + c.input_nodes[1]->id + " is neither a ciphertext nor a plaintext");
                    if (n == 2) same_ct(c.input_nodes[1], "second operand");
                    if (a.degree != 1) bad(c, "ciphertext multiply expects degree-1 operands");
                    result(a.level, 2);
                }
                break;
            case OperationType::RELINEARIZE:
                if (a.degree != 2) bad(c, "relinearize expects a degree-2 ciphertext");
                key_ok(c.input_nodes[1]);
                result(a.level, 1);
                break;
            case OperationType::RESCALE:
                if (a.level < 1) bad(c, "rescale needs level >= 1");
                result(a.level - 1, a.degree);
                break;
            case OperationType::DROP_LEVEL:
                if (r.level >= a.level) bad(c, "drop_level must lower the level");
                result(r.level, a.degree);
                break;
            case OperationType::ROTATE_COL:
            case OperationType::ROTATE_ROW:
                if (a.degree != 1) bad(c, "rotation expects a degree-1 ciphertext");
                key_ok(c.input_nodes[1]);
                result(a.level, 1);
                break;
            case OperationType::MAC_WO_PARTIAL_SUM:
            case OperationType::MAC_W_PARTIAL_SUM:
                for (size_t i = 1; i < n; i++) {
                    const DatumNode* d = c.input_nodes[i];
                    if (is_ct(d)) same_ct(d, "operand");
                    else if (is_pt(d)) plain_ok(d);
                    else bad(c, "operand " + d->id + " is neither a ciphertext nor a plaintext");
                }
                result(a.level, a.degree);
                break;
            case OperationType::BOOTSTRAP:
                if (a.degree != 1 || r.degree != 1) bad(c, "bootstrap expects and produces degree-1 ciphertexts");
                break;
            default: break;
        }
    }
    // levels index the chain the context is built on: the first max_level+1 primes of `q` (frontend/parameter.json lists 30
    // primes for CKKS n=65536 under max_level 33, SURVEY App. A: what exists bounds it the other way)
    int64_t n_q = parameter.contains("q") ? (int64_t)parameter["q"].size() : INT32_MAX;
    if (parameter.contains("max_level") && parameter["max_level"].kind == mjson::Value::Int) n_q = std::min<int64_t>(n_q, parameter["max_level"].as_int() + 1);
    for (auto& kv : data) {
        const DatumNode& d = kv.second;
        if (d.fhe_prop && (d.fhe_prop->level < 0 || d.fhe_prop->level >= n_q || d.fhe_prop->degree < 0 || d.fhe_prop->degree > 2))
            throw std::runtime_error("datum " + d.id + ": level " + std::to_string(d.fhe_prop->level) + " / degree " +
                                     std::to_string(d.fhe_prop->degree) + " outside the parameter set");
        if (d.predecessors.size() > 1) throw std::runtime_error("datum " + d.id + " is produced by " + std::to_string(d.predecessors.size()) + " compute nodes");
        if (d.is_input && !d.predecessors.empty()) throw std::runtime_error("task input " + d.id + " is also produced by compute node " + d.predecessors[0]->id);
        if (!d.is_input && d.predecessors.empty() && !d.successors.empty()) throw std::runtime_error("datum " + d.id + " is read but neither a task input nor produced");
        if (d.is_output && !d.is_input && d.predecessors.empty()) throw std::runtime_error("task output " + d.id + " is never produced");
    }
}

// Peephole: a tree of `add` nodes whose leaves are single-use ciphertext x plaintext products (the shape the frontend
// emits for convolutions and matrix-vector products, e.g. examples/benchmark_convolution: mult(ct,pt) + add chains) is
// rewritten into the graph's own multiply-accumulate nodes (cmp_sum / cmpac_sum, <= 16 terms each, chained through the
// partial-sum input).  Modular addition is associative and commutative, so every residue of the result is unchanged; a
// 72-term accumulation becomes 5 launches instead of 72 products + 71 dependent additions.
// Peephole: CKKS mult(ct,ct) -> relin -> rescale whose intermediates have no other reader becomes one node that runs the
// fused operator (merged ModDown+rescale tail, ops.hip): same residues, 104 instead of 128 limb transforms.
void TaskGraph::fuse_mult_relin_rescale() {
    if (algo != ALGO_CKKS) return;
    auto private_to = [](const DatumNode* d) {
        return d->predecessors.size() == 1 && d->successors.size() == 1 && !d->is_output && !d->is_input;
    };
    std::vector<NodeIndex> tails;
    for (auto& kv : computes)
        if (kv.second.op() == OperationType::RESCALE) tails.push_back(kv.first);
    std::sort(tails.begin(), tails.end());
    for (NodeIndex ti : tails) {
        ComputeNode* rs = &computes.at(ti);
        DatumNode* d1 = rs->input_nodes[0];
        if (!private_to(d1) || d1->predecessors[0]->op() != OperationType::RELINEARIZE) continue;
        ComputeNode* rl = d1->predecessors[0];
        DatumNode* d0 = rl->input_nodes[0];
        if (!private_to(d0) || d0->predecessors[0]->op() != OperationType::MULTIPLY) continue;
        ComputeNode* mu = d0->predecessors[0];
        bool ct_ct = !mu->input_nodes.empty() && mu->input_nodes.size() <= 2;
        for (DatumNode* in : mu->input_nodes)
            ct_ct = ct_ct && in->datum_type == TYPE_CIPHERTEXT && in->fhe_prop && in->fhe_prop->degree == 1;
        if (!ct_ct || rl->input_nodes.size() != 2 || d0->fhe_prop->level < 1) continue;
        ComputeNode c;
        c.index = next_compute++;
        c.id = rs->id + "_fused";
        ComputeNode::FheProperty fp;
        fp.op_type = OperationType::FUSED_MULT_RELIN_RESCALE;
        c.fhe_prop = fp;
        c.input_nodes = mu->input_nodes;
        c.input_nodes.push_back(rl->input_nodes[1]);   // the relinearisation key
        c.output_nodes = rs->output_nodes;
        for (ComputeNode* old : {mu, rl, rs}) {
            for (DatumNode* in : old->input_nodes) {
                auto& v = in->successors;
                v.erase(std::remove(v.begin(), v.end(), old), v.end());
            }
            for (DatumNode* o : old->output_nodes) {
                auto& v = o->predecessors;
                v.erase(std::remove(v.begin(), v.end(), old), v.end());
            }
        }
        const NodeIndex i0 = d0->index, i1 = d1->index, m_i = mu->index, l_i = rl->index, r_i = rs->index;
        data.erase(i0);
        data.erase(i1);
        computes.erase(m_i);
        computes.erase(l_i);
        computes.erase(r_i);
        auto it = computes.emplace(c.index, std::move(c)).first;
        for (DatumNode* in : it->second.input_nodes) in->successors.push_back(&it->second);
        for (DatumNode* o : it->second.output_nodes) o->predecessors.push_back(&it->second);
    }
}

void TaskGraph::fuse_accumulations() {
    auto is_ct = [](const DatumNode* d) { return d->datum_type == TYPE_CIPHERTEXT && d->fhe_prop && d->fhe_prop->degree == 1; };
    auto is_pt = [](const DatumNode* d) { return d->datum_type == TYPE_PLAINTEXT && d->fhe_prop; };
    auto sole_producer = [](const DatumNode* d) -> ComputeNode* {
        return (d->predecessors.size() == 1 && d->successors.size() == 1 && !d->is_output && !d->is_input) ? d->predecessors[0]
                                                                                                              : nullptr;
    };
    auto is_product = [&](const ComputeNode* c) {
        return c && c->op() == OperationType::MULTIPLY && c->input_nodes.size() == 2 && is_ct(c->input_nodes[0]) &&
               is_pt(c->input_nodes[1]) && c->output_nodes.size() == 1;
    };
    auto is_add = [&](const ComputeNode* c) {
        return c && c->op() == OperationType::ADD && c->input_nodes.size() == 2 && is_ct(c->input_nodes[0]) &&
               is_ct(c->input_nodes[1]) && c->output_nodes.size() == 1;
    };
    std::vector<NodeIndex> roots;
    for (auto& kv : computes) {
        ComputeNode* c = &kv.second;
        if (!is_add(c)) continue;
        DatumNode* out = c->output_nodes[0];
        ComputeNode* user = (out->successors.size() == 1 && !out->is_output) ? out->successors[0] : nullptr;
        if (!is_add(user)) roots.push_back(kv.first);   // the top of a tree
    }
    std::sort(roots.begin(), roots.end());
    for (NodeIndex ri : roots) {
        ComputeNode* root = &computes.at(ri);
        const int level = root->output_nodes[0]->fhe_prop->level;
        std::vector<ComputeNode*> adds{root}, prods;
        std::vector<DatumNode*> others;   // operands that are neither an inner add nor a product
        for (size_t i = 0; i < adds.size(); i++)
            for (DatumNode* in : adds[i]->input_nodes) {
                ComputeNode* p = sole_producer(in);
                if (in->fhe_prop->level != level) p = nullptr;
                if (is_add(p)) adds.push_back(p);
                else if (is_product(p) && p->input_nodes[0]->fhe_prop->level == level &&
                         p->input_nodes[1]->fhe_prop->level == level)
                    prods.push_back(p);
                else others.push_back(in);
            }
        if (prods.size() < 2 || others.size() > 1) continue;   // nothing to gain / more than one partial sum
        std::sort(prods.begin(), prods.end(), [](ComputeNode* a, ComputeNode* b) { return a->index < b->index; });
        DatumNode* final_out = root->output_nodes[0];
        DatumNode* partial = others.empty() ? nullptr : others[0];
        // detach everything that disappears
        auto unlink = [&](ComputeNode* c) {
            for (DatumNode* in : c->input_nodes) {
                auto& v = in->successors;
                v.erase(std::remove(v.begin(), v.end(), c), v.end());
            }
            for (DatumNode* o : c->output_nodes) {
                auto& v = o->predecessors;
                v.erase(std::remove(v.begin(), v.end(), c), v.end());
            }
        };
        std::vector<DatumNode*> cts, pts;
        for (ComputeNode* p : prods) {
            cts.push_back(p->input_nodes[0]);
            pts.push_back(p->input_nodes[1]);
        }
        std::vector<NodeIndex> dead_compute, dead_data;
        for (ComputeNode* p : prods) {
            dead_data.push_back(p->output_nodes[0]->index);
            dead_compute.push_back(p->index);
            unlink(p);
        }
        for (ComputeNode* a : adds) {
            if (a != root) dead_data.push_back(a->output_nodes[0]->index);
            dead_compute.push_back(a->index);
            unlink(a);
        }
        for (NodeIndex d : dead_data) data.erase(d);
        for (NodeIndex c : dead_compute) computes.erase(c);
        // chain of multiply-accumulate nodes
        const size_t kmax = 16;
        for (size_t i0 = 0; i0 < cts.size(); i0 += kmax) {
            const size_t cnt = std::min(kmax, cts.size() - i0);
            const bool last = i0 + cnt == cts.size();
            DatumNode* out = last ? final_out : &clone_datum(*final_out, final_out->id + "_mac" + std::to_string(i0));
            ComputeNode c;
            c.index = next_compute++;
            c.id = final_out->id + "_fused_mac" + std::to_string(i0);
            ComputeNode::FheProperty fp;
            fp.op_type = partial ? OperationType::MAC_W_PARTIAL_SUM : OperationType::MAC_WO_PARTIAL_SUM;
            ComputeNode::FheProperty::ExtraProperty ep;
            ep.sum_cnt = (int32_t)cnt;
            fp.p = ep;
            c.fhe_prop = fp;
            for (size_t i = 0; i < cnt; i++) c.input_nodes.push_back(cts[i0 + i]);
            if (partial) c.input_nodes.push_back(partial);
            for (size_t i = 0; i < cnt; i++) c.input_nodes.push_back(pts[i0 + i]);
            c.output_nodes.push_back(out);
            auto it = computes.emplace(c.index, std::move(c)).first;
            for (DatumNode* in : it->second.input_nodes) in->successors.push_back(&it->second);
            out->predecessors.push_back(&it->second);
            partial = out;
        }
    }
}

DatumNode& TaskGraph::clone_datum(const DatumNode& src, const std::string& id) {
    DatumNode n = src;
    n.index = next_data++;
    n.id = id;
    n.is_input = n.is_output = false;
    n.predecessors.clear();
    n.successors.clear();
    auto it = data.emplace(n.index, std::move(n)).first;
    return it->second;
}

void TaskGraph::link_bridge(OperationType op, const std::string& id, DatumNode* in, DatumNode* out) {
    ComputeNode c;
    c.index = next_compute++;
    c.id = id;
    ComputeNode::FheProperty fp;
    fp.op_type = op;
    c.fhe_prop = fp;
    c.input_nodes.push_back(in);
    c.output_nodes.push_back(out);
    auto it = computes.emplace(c.index, std::move(c)).first;
    in->successors.push_back(&it->second);
    out->predecessors.push_back(&it->second);
}

// For every original datum decide where its bytes natively live (caller handle vs. device) and splice in
// export->load (host to device) or store->import (device to host) chains where a consumer lives on the other side.
void TaskGraph::insert_bridges() {
    std::vector<NodeIndex> originals;
    for (auto& kv : data) originals.push_back(kv.first);
    std::sort(originals.begin(), originals.end());

    auto retarget_inputs = [](std::vector<ComputeNode*>& consumers, DatumNode& from, DatumNode& to) {
        for (ComputeNode* c : consumers) {
            for (auto& in : c->input_nodes)
                if (in == &from) in = &to;
            to.successors.push_back(c);
        }
    };

    for (NodeIndex idx : originals) {
        DatumNode& d = data.at(idx);
        const bool host_native = d.is_input || (!d.predecessors.empty() && d.predecessors[0]->custom_prop.has_value());
        std::vector<ComputeNode*> dev_consumers, host_consumers;
        for (ComputeNode* c : d.successors) (c->custom_prop ? host_consumers : dev_consumers).push_back(c);
        const std::string tag = std::to_string(idx);

        if (d.is_input && d.custom_prop) {  // custom input data: only made concrete on the host
            DatumNode& conc = clone_datum(d, d.id + "_concrete");
            std::vector<ComputeNode*> all = d.successors;
            d.successors.clear();
            retarget_inputs(all, d, conc);
            link_bridge(OperationType::EXPORT_TO_ABI, "export_to_abi_" + tag, &d, &conc);
            continue;
        }
        if (host_native && !dev_consumers.empty()) {  // handle -> C struct -> device
            DatumNode& cs = clone_datum(d, d.id + "_c_struct_h2d");
            DatumNode& dev = clone_datum(d, d.id + "_gpu");
            d.successors = host_consumers;
            retarget_inputs(dev_consumers, d, dev);
            link_bridge(OperationType::EXPORT_TO_ABI, "export_to_abi_" + tag, &d, &cs);
            link_bridge(OperationType::LOAD_TO_BACKEND, "load_to_gpu_" + tag, &cs, &dev);
        }
        if (!host_native && (!host_consumers.empty() || d.is_output)) {  // device -> C struct -> handle
            DatumNode& dev = clone_datum(d, d.id + "_gpu");
            DatumNode& cs = clone_datum(d, d.id + "_c_struct_d2h");
            // device producers now write the device datum
            for (ComputeNode* p : d.predecessors) {
                for (auto& out : p->output_nodes)
                    if (out == &d) out = &dev;
                dev.predecessors.push_back(p);
            }
            d.predecessors.clear();
            d.successors.clear();
            retarget_inputs(dev_consumers, d, dev);
            link_bridge(OperationType::STORE_FROM_BACKEND, "store_from_gpu_" + tag, &dev, &cs);
            if (d.is_output) {
                link_bridge(OperationType::IMPORT_FROM_ABI, "import_from_abi_" + tag, &cs, &d);
                retarget_inputs(host_consumers, d, d);  // custom consumers of an output read the imported handle
            } else {
                DatumNode& h = clone_datum(d, d.id + "_handle");
                link_bridge(OperationType::IMPORT_FROM_ABI, "import_from_abi_" + tag, &cs, &h);
                retarget_inputs(host_consumers, d, h);
            }
        }
        if (host_native && d.is_output && !d.is_input) {  // produced by a custom node and returned to the caller
            DatumNode& conc = clone_datum(d, d.id + "_concrete");
            for (ComputeNode* p : d.predecessors) {
                for (auto& out : p->output_nodes)
                    if (out == &d) out = &conc;
                conc.predecessors.push_back(p);
            }
            d.predecessors.clear();
            link_bridge(OperationType::IMPORT_FROM_ABI, "import_from_abi_" + tag, &conc, &d);
        }
    }
}

void TaskGraph::assign_processors() {
    for (auto& kv : computes) {
        ComputeNode& c = kv.second;
        if (c.custom_prop) c.on_cpu = true;
        else c.on_cpu = c.op() == OperationType::EXPORT_TO_ABI || c.op() == OperationType::IMPORT_FROM_ABI;
    }
}

// top_level = longest path from a source, bottom_level = longest path to a sink (Kahn, O(V+E)); priority = bottom_level
void TaskGraph::compute_levels() {
    std::unordered_map<NodeIndex, int> indeg, outdeg;
    for (auto& kv : computes) {
        ComputeNode& c = kv.second;
        c.sched_meta = {};
        int in = 0, out = 0;
        for (auto* d : c.input_nodes) in += (int)d->predecessors.size();
        for (auto* d : c.output_nodes) out += (int)d->successors.size();
        indeg[c.index] = in;
        outdeg[c.index] = out;
    }
    std::queue<ComputeNode*> q;
    for (auto& kv : computes)
        if (indeg[kv.first] == 0) q.push(&kv.second);
    size_t seen = 0;
    while (!q.empty()) {
        ComputeNode* u = q.front();
        q.pop();
        seen++;
        for (auto* d : u->output_nodes)
            for (ComputeNode* v : d->successors) {
                v->sched_meta.top_level = std::max(v->sched_meta.top_level, u->sched_meta.top_level + 1);
                if (--indeg[v->index] == 0) q.push(v);
            }
    }
    if (seen != computes.size()) throw std::runtime_error("task graph has a cycle");
    for (auto& kv : computes)
        if (outdeg[kv.first] == 0) q.push(&kv.second);
    while (!q.empty()) {
        ComputeNode* v = q.front();
        q.pop();
        for (auto* d : v->input_nodes)
            for (ComputeNode* u : d->predecessors) {
                u->sched_meta.bottom_level = std::max(u->sched_meta.bottom_level, v->sched_meta.bottom_level + 1);
                if (--outdeg[u->index] == 0) q.push(u);
            }
    }
    max_top_level = 0;
    for (auto& kv : computes) {
        kv.second.priority = kv.second.sched_meta.bottom_level;
        max_top_level = std::max(max_top_level, kv.second.sched_meta.top_level);
    }
}

void TaskGraph::bind_bridge_executors(const ExecutorFunc& abi_export, const ExecutorFunc& abi_import) {
    for (auto& kv : computes) {
        if (kv.second.op() == OperationType::EXPORT_TO_ABI) kv.second.executor = abi_export;
        else if (kv.second.op() == OperationType::IMPORT_FROM_ABI) kv.second.executor = abi_import;
    }
}

void TaskGraph::bind_custom_executors(const std::unordered_map<std::string, ExecutorFunc>& custom) {
    for (auto& kv : computes) {
        if (!kv.second.custom_prop) continue;
        auto it = custom.find(kv.second.custom_prop->type);
        if (it != custom.end()) kv.second.executor = it->second;
    }
}
