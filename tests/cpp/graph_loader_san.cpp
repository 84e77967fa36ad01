// Sanitizer driver for the task-graph loader (compiled by tests/test_graph_loader_san.py with -fsanitize=address,undefined
// together with lattisense_amd/csrc/task_graph.cpp).  Reads one mega_ag.json path per line from the file named in argv[1];
// prints "ok <nodes> <levels>" or "err <message>" per path.  A malformed graph must end in an exception, never in a
// sanitizer report or a crash.  The loader's behaviour follows mega_ag_runners/mega_ag.cpp:125-657.
#include <cstdio>
#include <exception>
#include <fstream>
#include <iostream>
#include <string>

#include "../../lattisense_amd/csrc/task_graph.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream list(argv[1]);
    std::string path;
    long ok = 0, err = 0;
    while (std::getline(list, path)) {
        if (path.empty()) continue;
        try {
            TaskGraph g = TaskGraph::load_for_gpu(path);
            // walk what the scheduler walks: every edge must point at a node the graph owns
            size_t edges = 0;
            for (auto& [idx, c] : g.computes) {
                for (auto* d : c.input_nodes) edges += d->successors.size();
                for (auto* d : c.output_nodes) edges += d->predecessors.size();
                if (c.sched_meta.top_level < 0 || c.sched_meta.top_level > g.max_top_level) throw std::runtime_error("level out of range");
            }
            std::printf("ok %zu %d %zu\n", g.computes.size(), g.max_top_level, edges);
            ok++;
        } catch (const std::exception& e) {
            std::printf("err %s\n", e.what());
            err++;
        }
    }
    std::printf("DONE ok=%ld err=%ld\n", ok, err);
    return 0;
}
