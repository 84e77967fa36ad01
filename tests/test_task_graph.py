"""CPU-side checks of the task layer: graph loading, ABI-bridge insertion (node counts follow
mega_ag_runners/mega_ag.cpp:388-544), bind-time rejection of unsupported operators, loud failure without a GPU."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TASKS = os.path.join(ROOT, "tests", "golden", "tasks")


@pytest.fixture(scope="module", autouse=True)
def built():
    from lattisense_amd import build
    build.build_native()


def expected_counts(name):
    g = json.load(open(os.path.join(TASKS, name, "mega_ag.json")))
    nd, nc = len(g["data"]), len(g["compute"])
    n_in, n_out = len(g["inputs"]), len(g["outputs"])
    # every input handle: export + load (2 data, 2 computes); every output: store + import (2 data, 2 computes)
    return {"data": nd + 2 * n_in + 2 * n_out, "compute": nc + 2 * n_in + 2 * n_out, "inputs": n_in, "outputs": n_out}


@pytest.mark.parametrize("name", sorted(d for d in os.listdir(TASKS) if "unsupported" not in d))
def test_graph_loads_with_bridges(name, monkeypatch):
    from lattisense_amd.task import FheTaskGpu
    monkeypatch.setenv("LSA_NO_GRAPH_FUSION", "1")   # the graph exactly as compiled
    t = FheTaskGpu(os.path.join(TASKS, name))
    assert t.counts() == expected_counts(name)
    t.close()


def test_accumulation_fusion_rewrites_product_sums():
    """mult(ct,pt) + add trees become the graph's own multiply-accumulate nodes (<= 16 terms each): the conv2d fixture's
    18 products and 17 accumulating adds collapse into 2 nodes, one intermediate datum instead of 34."""
    from lattisense_amd.task import FheTaskGpu
    name = "ckks_n4096_conv2d_1in_1out_32x32_3x3"
    want = expected_counts(name)
    t = FheTaskGpu(os.path.join(TASKS, name))
    got = t.counts()
    assert got["compute"] == want["compute"] - 18 - 17 + 2
    assert got["data"] == want["data"] - 18 - 16 + 1
    assert got["inputs"] == want["inputs"] and got["outputs"] == want["outputs"]
    t.close()
    # graphs without such trees are untouched
    t = FheTaskGpu(os.path.join(TASKS, "ckks_n4096_cmp_cap"))
    assert t.counts() == expected_counts("ckks_n4096_cmp_cap")
    t.close()


def test_missing_task_file_raises():
    from lattisense_amd.task import FheTaskGpu
    with pytest.raises(RuntimeError, match="Cannot open MegaAG file"):
        FheTaskGpu("/nonexistent/task")


def test_unsupported_operator_is_rejected_at_bind_time():
    from lattisense_amd.task import FheTaskGpu
    with pytest.raises(RuntimeError, match="Multiply with plaintext only supported for CKKS scheme"):
        FheTaskGpu(os.path.join(TASKS, "bfv_n4096_cmp_unsupported"))


def test_run_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lattisense_amd._native import LsaError
    from lattisense_amd.task import Argument, Ciphertext, FheTaskGpu
    t = FheTaskGpu(os.path.join(TASKS, "ckks_n4096_cmc"))
    xs = [Ciphertext.empty(1, 3, 4096) for _ in range(4)]
    ys = [Ciphertext.empty(1, 3, 4096) for _ in range(4)]
    zs = [Ciphertext.empty(2, 3, 4096) for _ in range(4)]
    with pytest.raises(LsaError) as e:
        t.run([Argument("in_x_list", xs), Argument("in_y_list", ys)], [Argument("out_z_list", zs)])
    assert e.value.code == 2   # LSA_ERR_NO_DEVICE: there is no CPU fallback
