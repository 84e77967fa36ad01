// Test shim: a PLUG-IN-SHAPED caller of the task entry points, built as its own shared object -- the way
// plug-in/SEAL/acc and plug-in/lattigo/acc reach a GPU task (plug-in/SEAL/acc/gpu_runner.cpp:12-52,
// plug-in/lattigo/acc/gpu_runner.go:88-128):
//   * the plug-in keeps its own objects (here: PlgCiphertext / PlgPlaintext / PlgKeys, limb vectors in the plug-in's own
//     heap -- the roles of seal::Ciphertext / seal::RelinKeys / seal::GaloisKeys);
//   * its EXPORT executor turns one into a C struct the way abi/c_structs.c:23-98 allocates them -- malloc per struct, per
//     polynomial and PER LIMB, contents copied -- owned by a std::shared_ptr<CCiphertext|CPlaintext|CRelinKey|CGaloisKey>
//     whose deleter frees every level (plug-in/SEAL/acc/abi_bridge_executors.h:70-140); a Galois export carries exactly ONE
//     element (set_galois_key_steps(c_glk, &galois_element, 1), ibid. :127);
//   * its IMPORT executor any_casts the backend's std::shared_ptr<CCiphertext> -- created in ANOTHER shared object, so the
//     cast only works if the type identity of std::shared_ptr<CCiphertext> agrees across the two libraries --
//     and copies it into the pre-allocated destination it gets through ctx.other_args[0] (ibid. :150-179);
//   * per-run state lives in file-static globals set by the runner around each run (ibid. :49-59), the executors ignore
//     ExecutionContext.context, and the std::function objects are destroyed right after bind (gpu_runner.go:93-94).
// The key digit shape is the caller's: `n_special` special primes per digit (hybrid, Lattigo) or one (SEAL: level+1
// single-prime digits, plug-in/SEAL/acc/c_struct_import_export.h:93-142) -- the C structs look the same.
#include <any>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../lattisense_amd/csrc/task_graph.h"   // ExecutorFunc, ExecutionContext, ComputeNode, DatumNode + the C structs

namespace {

// ---- the plug-in's own objects
struct PlgCiphertext {
    int level, degree, n;
    std::vector<uint64_t> limbs;   // [degree+1][level+1][n]
    int lie_level = -1, lie_degree = -1, drop_limbs = 0;   // negative tests: export a struct that disagrees with the task
};
struct PlgPlaintext {
    int limbs_n, n;
    std::vector<uint64_t> limbs;   // [limbs_n][n]
};
struct PlgKsKey {
    int level, n_special, n;
    std::vector<uint64_t> limbs;   // [beta][2][level+1+n_special][n]
    int beta() const { return (level + 1 + n_special - 1) / n_special; }
    int comp() const { return level + 1 + n_special; }
};
struct PlgGaloisKeys {
    std::map<uint64_t, PlgKsKey> keys;
};

// ---- per-run state, as the SEAL plug-in keeps it (abi_bridge_executors.h:49-59): set before a run, cleared after
struct PlgContext {
    int n;
};
static PlgContext* g_plg_context = nullptr;
static std::atomic<long> g_exports{0}, g_imports{0}, g_struct_frees{0}, g_limb_mallocs{0};

// ---- C-struct allocation exactly as abi/c_structs.c:23-98 does it: one malloc per limb
void plg_alloc_component(CComponent* c, int n) {
    c->n = n;
    c->data = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    g_limb_mallocs++;
}
void plg_alloc_polynomial(CPolynomial* p, int n_component, int n) {
    p->n_component = n_component;
    p->components = (CComponent*)malloc(sizeof(CComponent) * (size_t)n_component);
    for (int i = 0; i < n_component; i++) plg_alloc_component(&p->components[i], n);
}
void plg_free_polynomial(CPolynomial* p) {
    for (int i = 0; i < p->n_component; i++) free(p->components[i].data);
    free(p->components);
}
void plg_free_ciphertext(CCiphertext* ct) {
    for (int i = 0; i <= ct->degree; i++) plg_free_polynomial(&ct->polys[i]);
    free(ct->polys);
    g_struct_frees++;
}
void plg_free_ksk(CKeySwitchKey* k) {
    for (int d = 0; d < k->n_public_key; d++) plg_free_ciphertext(&k->public_keys[d]);
    free(k->public_keys);
}
void plg_export_ksk(CKeySwitchKey* dst, const PlgKsKey& k) {
    const int beta = k.beta(), comp = k.comp();
    dst->n_public_key = beta;
    dst->public_keys = (CPublicKey*)malloc(sizeof(CPublicKey) * (size_t)beta);
    for (int d = 0; d < beta; d++) {
        CPublicKey& pk = dst->public_keys[d];
        pk.level = k.level;
        pk.degree = 1;
        pk.polys = (CPolynomial*)malloc(sizeof(CPolynomial) * 2);
        for (int h = 0; h < 2; h++) {
            plg_alloc_polynomial(&pk.polys[h], comp, k.n);
            for (int j = 0; j < comp; j++)
                memcpy(pk.polys[h].components[j].data, k.limbs.data() + ((size_t)(d * 2 + h) * comp + j) * k.n, sizeof(uint64_t) * (size_t)k.n);
        }
    }
}

ExecutorFunc make_export() {
    return [](ExecutionContext& /*ctx: ignored, as the plug-ins do*/, const std::unordered_map<NodeIndex, std::any>& inputs, std::any& output,
              const ComputeNode& self) {
        if (!g_plg_context) throw std::runtime_error("plug-in context not set");
        const DatumNode* in = self.input_nodes[0];
        if (!in->fhe_prop.has_value()) throw std::runtime_error("Input node missing FHE properties for EXPORT_TO_ABI");
        auto handle = std::any_cast<std::shared_ptr<void>>(inputs.at(in->index));
        g_exports++;
        switch (in->datum_type) {
            case TYPE_CIPHERTEXT: {
                auto* src = static_cast<PlgCiphertext*>(handle.get());
                auto* ct = (CCiphertext*)malloc(sizeof(CCiphertext));
                ct->level = src->lie_level >= 0 ? src->lie_level : src->level;
                ct->degree = src->lie_degree >= 0 ? src->lie_degree : src->degree;
                const int polys = ct->degree + 1, L = ct->level + 1 - src->drop_limbs;
                ct->polys = (CPolynomial*)malloc(sizeof(CPolynomial) * (size_t)polys);
                for (int p = 0; p < polys; p++) {
                    plg_alloc_polynomial(&ct->polys[p], L, src->n);
                    for (int j = 0; j < L; j++) {
                        const int sp = p <= src->degree ? p : src->degree, sj = j <= src->level ? j : src->level;
                        memcpy(ct->polys[p].components[j].data, src->limbs.data() + ((size_t)sp * (src->level + 1) + sj) * src->n,
                               sizeof(uint64_t) * (size_t)src->n);
                    }
                }
                output = std::shared_ptr<CCiphertext>(ct, [](CCiphertext* p) {
                    plg_free_ciphertext(p);
                    free(p);
                });
                break;
            }
            case TYPE_PLAINTEXT: {
                auto* src = static_cast<PlgPlaintext*>(handle.get());
                auto* pt = (CPlaintext*)malloc(sizeof(CPlaintext));
                pt->level = src->limbs_n - 1;
                plg_alloc_polynomial(&pt->poly, src->limbs_n, src->n);
                for (int j = 0; j < src->limbs_n; j++)
                    memcpy(pt->poly.components[j].data, src->limbs.data() + (size_t)j * src->n, sizeof(uint64_t) * (size_t)src->n);
                output = std::shared_ptr<CPlaintext>(pt, [](CPlaintext* p) {
                    plg_free_polynomial(&p->poly);
                    free(p);
                });
                break;
            }
            case TYPE_RELIN_KEY: {
                auto* src = static_cast<PlgKsKey*>(handle.get());
                auto* k = (CRelinKey*)malloc(sizeof(CRelinKey));
                plg_export_ksk(k, *src);
                output = std::shared_ptr<CRelinKey>(k, [](CRelinKey* p) {
                    plg_free_ksk(p);
                    free(p);
                });
                break;
            }
            case TYPE_GALOIS_KEY: {
                auto* src = static_cast<PlgGaloisKeys*>(handle.get());
                const uint64_t el = in->fhe_prop->p.has_value() ? in->fhe_prop->p->galois_element : 0;
                auto it = src->keys.find(el);
                if (it == src->keys.end()) throw std::runtime_error("Galois key for element " + std::to_string(el) + " not generated");
                auto* g = (CGaloisKey*)malloc(sizeof(CGaloisKey));
                g->n_key_switch_key = 1;   // ONE element per export
                g->galois_elements = (uint64_t*)malloc(sizeof(uint64_t));
                g->galois_elements[0] = el;
                g->key_switch_keys = (CKeySwitchKey*)malloc(sizeof(CKeySwitchKey));
                plg_export_ksk(&g->key_switch_keys[0], it->second);
                output = std::shared_ptr<CGaloisKey>(g, [](CGaloisKey* p) {
                    plg_free_ksk(&p->key_switch_keys[0]);
                    free(p->key_switch_keys);
                    free(p->galois_elements);
                    free(p);
                });
                break;
            }
            default: throw std::runtime_error("Unsupported data type in plug-in EXPORT_TO_ABI");
        }
    };
}

ExecutorFunc make_import() {
    return [](ExecutionContext& ctx, const std::unordered_map<NodeIndex, std::any>& inputs, std::any& output, const ComputeNode& self) {
        if (!g_plg_context) throw std::runtime_error("plug-in context not set");
        const DatumNode* in = self.input_nodes[0];
        if (ctx.other_args.empty()) throw std::runtime_error("plug-in IMPORT_FROM_ABI requires pre-allocated dest via other_args");
        if (in->datum_type != TYPE_CIPHERTEXT) throw std::runtime_error("Unsupported data type in plug-in IMPORT_FROM_ABI");
        // created by the backend library: the cast succeeds only if std::shared_ptr<CCiphertext> is ONE type across the objects
        auto c_ct = std::any_cast<std::shared_ptr<CCiphertext>>(inputs.at(in->index));
        void* dest_raw = ctx.get_other_arg<void>(0);
        auto* dst = static_cast<PlgCiphertext*>(dest_raw);
        if (!dst) throw std::runtime_error("null destination");
        if (dst->level != c_ct->level || dst->degree != c_ct->degree)
            throw std::runtime_error("destination allocated at another level / degree");
        for (int p = 0; p <= c_ct->degree; p++)
            for (int j = 0; j <= c_ct->level; j++) {
                if (c_ct->polys[p].components[j].n != dst->n) throw std::runtime_error("ring degree mismatch");
                memcpy(dst->limbs.data() + ((size_t)p * (dst->level + 1) + j) * dst->n, c_ct->polys[p].components[j].data,
                       sizeof(uint64_t) * (size_t)dst->n);
            }
        g_imports++;
        output = std::shared_ptr<void>(dest_raw, [](void*) {});
    };
}

}  // namespace

extern "C" {

void* plg_ct_new(int level, int degree, int n, const uint64_t* data) {
    auto* c = new PlgCiphertext{level, degree, n, {}};
    c->limbs.assign((size_t)(degree + 1) * (level + 1) * n, 0);
    if (data) memcpy(c->limbs.data(), data, c->limbs.size() * sizeof(uint64_t));
    return c;
}
void plg_ct_read(void* h, uint64_t* out) {
    auto* c = static_cast<PlgCiphertext*>(h);
    memcpy(out, c->limbs.data(), c->limbs.size() * sizeof(uint64_t));
}
// negative tests: make the export disagree with what the task declares
void plg_ct_lie(void* h, int lie_level, int lie_degree, int drop_limbs) {
    auto* c = static_cast<PlgCiphertext*>(h);
    c->lie_level = lie_level;
    c->lie_degree = lie_degree;
    c->drop_limbs = drop_limbs;
}
void plg_ct_free(void* h) { delete static_cast<PlgCiphertext*>(h); }

void* plg_pt_new(int limbs_n, int n, const uint64_t* data) {
    auto* p = new PlgPlaintext{limbs_n, n, {}};
    p->limbs.assign(data, data + (size_t)limbs_n * n);
    return p;
}
void plg_pt_free(void* h) { delete static_cast<PlgPlaintext*>(h); }

void* plg_ksk_new(int level, int n_special, int n, const uint64_t* data) {
    auto* k = new PlgKsKey{level, n_special, n, {}};
    k->limbs.assign(data, data + (size_t)k->beta() * 2 * k->comp() * n);
    return k;
}
void plg_ksk_free(void* h) { delete static_cast<PlgKsKey*>(h); }

void* plg_glk_new() { return new PlgGaloisKeys(); }
void plg_glk_add(void* h, uint64_t element, int level, int n_special, int n, const uint64_t* data) {
    PlgKsKey k{level, n_special, n, {}};
    k.limbs.assign(data, data + (size_t)k.beta() * 2 * k.comp() * n);
    static_cast<PlgGaloisKeys*>(h)->keys[element] = std::move(k);
}
void plg_glk_free(void* h) { delete static_cast<PlgGaloisKeys*>(h); }

// what the plug-in's FheTaskGpu constructor does: build the two executors, hand their ADDRESSES to the task, destroy them
void plg_bind(fhe_task_handle task) {
    auto* ex = new ExecutorFunc(make_export());
    auto* im = new ExecutorFunc(make_import());
    bind_gpu_task_abi_bridge_executors(task, ex, im);   // copies the std::function objects (gpu_wrapper.cu:492-497)
    delete ex;
    delete im;
}
// what its run() does around run_fhe_gpu_task: set the file-static context, run, clear
int plg_run(fhe_task_handle task, int n, CArgument* in, uint64_t n_in, CArgument* out, uint64_t n_out, int device) {
    PlgContext ctx{n};
    g_plg_context = &ctx;
    const int rc = run_fhe_gpu_task(task, in, n_in, out, n_out, nullptr, nullptr, device);
    g_plg_context = nullptr;
    return rc;
}
void plg_counters(long* exports, long* imports, long* struct_frees, long* limb_mallocs) {
    *exports = g_exports;
    *imports = g_imports;
    *struct_frees = g_struct_frees;
    *limb_mallocs = g_limb_mallocs;
}
}
