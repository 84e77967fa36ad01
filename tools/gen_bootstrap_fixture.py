"""Golden data for the bootstrapping linear transforms, produced with the reference's own rotation planner
(frontend/bootstrap_params.py:104-263).  Runs ONLY in the build container (imports /root/reference/frontend); the output
is plain JSON: per (log_n, cts_depth, stc_depth) the diagonal index set of every merged CoeffsToSlots / SlotsToCoeffs
matrix and the rotation list a caller generates Galois keys for (full-slot encoding, log_slots = log_n - 1)."""
import json
import os
import sys

sys.path.insert(0, "/root/reference")
from frontend.bootstrap_params import EncodingMatrixParams, LinearTransformType  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for log_n in (10, 11, 13, 16):
    for cts_depth, stc_depth in ((4, 3), (3, 3), (2, 2)):
        entry = {}
        for name, lt, depth in (("cts", LinearTransformType.CoeffsToSlots, cts_depth), ("stc", LinearTransformType.SlotsToCoeffs, stc_depth)):
            p = EncodingMatrixParams(linear_transform_type=lt, repack_imag_2_real=True, level_start=24, bit_reversed=False,
                                     bsgs_ratio=2.0, scaling_factor=[[1.0]] * depth, log_n=log_n, log_slots=log_n - 1)
            idx = p.compute_bootstrapping_dft_index_map()
            entry[name] = {"diagonals": [sorted(int(k) for k in idx[i]) for i in range(depth)],
                           "rotations": sorted(int(r) for r in p.rotations())}
        out["logn%d_cts%d_stc%d" % (log_n, cts_depth, stc_depth)] = entry
# sparse packing: the whole rotation list of a bootstrap (SubSum + both transforms), frontend/custom_task.py:469-486
for log_n, log_slots in ((11, 8), (11, 9), (13, 9), (13, 11), (16, 11)):
    rots = [1 << i for i in range(log_slots, log_n - 1)]
    try:
        for lt, depth in ((LinearTransformType.CoeffsToSlots, 4), (LinearTransformType.SlotsToCoeffs, 3)):
            p = EncodingMatrixParams(linear_transform_type=lt, repack_imag_2_real=True, level_start=24, bit_reversed=False,
                                     bsgs_ratio=2.0, scaling_factor=[[1.0]] * depth, log_n=log_n, log_slots=log_slots)
            rots += p.rotations()
    except ZeroDivisionError:      # the planner divides by the giant-step count, which is 0 for very small matrices
        continue
    out["sparse_logn%d_slots%d_cts4_stc3" % (log_n, log_slots)] = {"rotations": sorted(set(int(r) for r in rots))}
path = os.path.join(ROOT, "tests", "golden", "bootstrap", "planner_rotations.json")
json.dump(out, open(path, "w"))
print(path, os.path.getsize(path))
