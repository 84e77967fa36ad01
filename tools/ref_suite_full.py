"""Extended parity run (GPU box): EVERY task directory of the reference's GPU test suite (tests/golden/ref_gpu_suite.tar.gz,
1252 graphs: every shape, every level, every parameter set) through run_fhe_gpu_task against the CPU oracle, ALL outputs
compared bit for bit.  The pytest suite runs a subset of this (tests/test_gpu_ref_suite.py); custom-node and bootstrap graphs
have their own tests and are skipped here.  usage: python tools/ref_suite_full.py [tag-substring]"""
import collections, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import ref_suite as rs

want = sys.argv[1] if len(sys.argv) > 1 else ""
root = rs.unpack(tempfile.mkdtemp(prefix="ref_suite_"))
t0 = time.time()
ran, skipped, nodes = collections.Counter(), collections.Counter(), 0
for k, (ptag, name, lv, path) in enumerate(rs.tasks(root)):
    if want not in ptag:
        continue
    g = rs.load(path)
    if rs.is_custom(g) or rs.has_type(g, "bootstrap"):
        skipped[ptag] += 1
        continue
    n, _ = rs.run_and_compare(path, seed=k)
    nodes += n
    ran[ptag] += 1
    if sum(ran.values()) % 100 == 0:
        print("... %d graphs, %.0f s" % (sum(ran.values()), time.time() - t0), flush=True)
for tag in sorted(set(ran) | set(skipped)):
    print("%-44s %4d graphs bit-exact on every output, %3d skipped (custom / bootstrap nodes: own tests)" % (tag, ran[tag], skipped[tag]))
print("ref_suite_full: %d graphs, %d compute nodes, all outputs bit-exact (%.0f s)" % (sum(ran.values()), nodes, time.time() - t0))
