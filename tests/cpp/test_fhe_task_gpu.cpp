// C++ test of the SDK-level class lattisense::FheTaskGpu (include/lattisense_task.hpp), written like the reference's own
// Catch2 cases (unittests/test_gpu_ckks.cpp "CKKS cmc": build inputs, FheTaskGpu(path), run, compare) but self-contained:
// the expected ct x ct tensor is computed here with 128-bit host arithmetic from its definition.
//   usage: test_fhe_task_gpu <tasks_dir>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "lattisense_task.hpp"

using namespace lattisense;
typedef unsigned __int128 u128;

static int failures = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); \
            failures++;                                                  \
        }                                                                \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string tasks = argv[1];
    // CKKS N=4096, first 4 primes of the fixture's chain (tests/golden/tasks/ckks_n4096_cmc/mega_ag.json "parameter")
    const uint64_t q[4] = {35184372121601ull, 17179967489ull, 17179672577ull, 17180262401ull};
    const int n = 4096, lvl = 3, n_op = 4;

    // error behaviour: a missing task directory throws at construction (mega_ag.cpp:135-137)
    bool threw = false;
    try {
        FheTaskGpu bad("/nonexistent/task");
    } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("Cannot open MegaAG file") != std::string::npos;
    }
    CHECK(threw);

    std::mt19937_64 rng(42);
    std::vector<HostCiphertext> xs, ys, zs;
    for (int i = 0; i < n_op; i++) {
        xs.emplace_back(1, lvl, n);
        ys.emplace_back(1, lvl, n);
        zs.emplace_back(2, lvl, n);   // ct3 is a legal task output (unittests/test_gpu_bfv.cpp:288-312)
        for (int p = 0; p < 2; p++)
            for (int j = 0; j <= lvl; j++)
                for (int k = 0; k < n; k++) {
                    xs[i].limb(p, j)[k] = rng() % q[j];
                    ys[i].limb(p, j)[k] = rng() % q[j];
                }
    }
    FheTaskGpu task(tasks + "/ckks_n4096_cmc");
    std::vector<TaskArgument> in = {TaskArgument("in_x_list", xs), TaskArgument("in_y_list", ys)};
    std::vector<TaskArgument> out = {TaskArgument("out_z_list", zs)};
    int last_done = -1, last_total = -2;
    const uint64_t ns = task.run(in, out, [&](int d, int t) { last_done = d; last_total = t; });
    CHECK(ns > 0);
    CHECK(last_done == last_total && last_total > 0);   // the final callback is guaranteed (wrapper.h:39-42)
    for (int i = 0; i < n_op; i++)
        for (int j = 0; j <= lvl; j++)
            for (int k = 0; k < n; k++) {
                const u128 a0 = xs[i].limb(0, j)[k], a1 = xs[i].limb(1, j)[k], b0 = ys[i].limb(0, j)[k], b1 = ys[i].limb(1, j)[k];
                const uint64_t d0 = (uint64_t)(a0 * b0 % q[j]);
                const uint64_t d1 = (uint64_t)((a0 * b1 % q[j] + a1 * b0 % q[j]) % q[j]);
                const uint64_t d2 = (uint64_t)(a1 * b1 % q[j]);
                if (zs[i].limb(0, j)[k] != d0 || zs[i].limb(1, j)[k] != d1 || zs[i].limb(2, j)[k] != d2) {
                    failures++;
                    if (failures < 5) std::fprintf(stderr, "mismatch op %d limb %d coeff %d\n", i, j, k);
                }
            }
    // outputs allocated at the wrong degree are refused by the import executor; run() throws, nothing crashes
    std::vector<HostCiphertext> wrong;
    for (int i = 0; i < n_op; i++) wrong.emplace_back(1, lvl, n);
    std::vector<TaskArgument> out_bad = {TaskArgument("out_z_list", wrong)};
    threw = false;
    try {
        task.run(in, out_bad);
    } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("was allocated at level/degree") != std::string::npos;
    }
    CHECK(threw);
    if (failures == 0) std::printf("test_fhe_task_gpu: OK (%d ops, %.3f ms)\n", n_op, ns / 1e6);
    return failures == 0 ? 0 : 1;
}
