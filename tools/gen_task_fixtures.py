"""Generates the task-graph fixtures under tests/golden/tasks/ with the reference's Python frontend.

Runs ONLY in the build container (imports /root/reference/frontend, pure Python + networkx); the output is plain
JSON data (mega_ag.json + task_signature.json per task), committed, and is all that travels to the GPU box.
Parameter sets are small custom ones (prime chains from the reference's own defaults, truncated) so the oracle can
check whole task runs in seconds; graph shapes are the reference's test/benchmark shapes:
  unittests/test_gpu_ckks.py:287-317 (cmc_relin, cmc_relin_rescale), :396-445 (rotate_col / rotate_row),
  unittests/test_gpu_bfv.py (cmc_relin, rotate), examples/benchmark_gpu/benchmark_gpu.py:25-75 (n_op disjoint subgraphs).
"""
import os
import shutil
import sys

REF = "/root/reference"
sys.path.insert(0, REF)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from frontend.custom_task import *  # noqa: E402,F401,F403
from lattisense_amd import params as P  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "tasks")
N_OP = 4


def emit(name, inputs, outputs):
    d = os.path.join(OUT, name)
    if os.path.exists(d):
        shutil.rmtree(d)
    os.makedirs(d)
    process_custom_task(input_args=inputs, offline_input_args=[], output_args=outputs, output_instruction_path=d,
                        fpga_acc=False)
    # keep only the two files the runtime reads
    for f in os.listdir(d):
        if f not in ("mega_ag.json", "task_signature.json"):
            p = os.path.join(d, f)
            shutil.rmtree(p) if os.path.isdir(p) else os.remove(p)


def ckks_param(n, nq):
    D = P.CKKS_DEFAULT[16384]
    return CkksParam.create_custom_param(n=n, q=D["q"][:nq], p=D["p"], scale=float(2 ** 34))


def bfv_param(n, nq):
    D = P.BFV_DEFAULT[16384]
    return BfvParam.create_custom_param(n=n, q=D["q"][:nq], p=D["p"], t=D["t"])


def main():
    # ---- CKKS N=4096, 5 primes
    set_fhe_param(ckks_param(4096, 5))
    lv = 4
    xs = [CkksCiphertextNode(f"x_{i}", level=lv) for i in range(N_OP)]
    ys = [CkksCiphertextNode(f"y_{i}", level=lv) for i in range(N_OP)]
    zs = [rescale(mult_relin(xs[i], ys[i]), f"z_{i}") for i in range(N_OP)]
    emit("ckks_n4096_cmc_relin_rescale", [Argument("in_x_list", xs), Argument("in_y_list", ys)],
         [Argument("out_z_list", zs)])

    set_fhe_param(ckks_param(4096, 5))
    xs = [CkksCiphertextNode(f"x_{i}", level=3) for i in range(N_OP)]
    ys = [CkksCiphertextNode(f"y_{i}", level=3) for i in range(N_OP)]
    zs = [mult(xs[i], ys[i], f"z_{i}") for i in range(N_OP)]
    emit("ckks_n4096_cmc", [Argument("in_x_list", xs), Argument("in_y_list", ys)], [Argument("out_z_list", zs)])

    set_fhe_param(ckks_param(4096, 5))
    xs = [CkksCiphertextNode(f"x_{i}", level=3) for i in range(N_OP)]
    steps = [1, 2, 5]
    ys = [advanced_rotate_cols(xs[i], steps, [f"y_{i}_{s}" for s in steps]) for i in range(N_OP)]
    emit("ckks_n4096_advanced_rotate_col", [Argument("arg_x", xs)], [Argument("arg_y", ys)])

    set_fhe_param(ckks_param(4096, 5))
    xs = [CkksCiphertextNode(f"x_{i}", level=2) for i in range(N_OP)]
    ys = [rotate_rows(xs[i], f"y_{i}") for i in range(N_OP)]
    emit("ckks_n4096_rotate_row", [Argument("in_x_list", xs)], [Argument("out_y_list", ys)])

    set_fhe_param(ckks_param(4096, 5))
    xs = [CkksCiphertextNode(f"x_{i}", level=3) for i in range(N_OP)]
    ys = [CkksCiphertextNode(f"y_{i}", level=3) for i in range(N_OP)]
    ws = [sub(add(xs[i], ys[i]), neg(xs[i]), f"w_{i}") for i in range(N_OP)]      # (x+y) - (-x)
    ds = [drop_level(ws[i], 1, f"d_{i}") for i in range(N_OP)]
    emit("ckks_n4096_add_sub_neg_drop", [Argument("in_x_list", xs), Argument("in_y_list", ys)],
         [Argument("out_d_list", ds)])

    # ring-t plaintext operands (one coefficient-domain limb): ct + pt, ct - pt, ct * pt
    for name, f in (("cap_ringt", add), ("csp_ringt", sub), ("cmp_ringt", mult)):
        set_fhe_param(ckks_param(4096, 5))
        xs = [CkksCiphertextNode(f"x_{i}", level=2) for i in range(N_OP)]
        ys = [CkksPlaintextRingtNode(f"y_{i}") for i in range(N_OP)]
        zs = [f(xs[i], ys[i], f"z_{i}") for i in range(N_OP)]
        emit("ckks_n4096_" + name, [Argument("in_x_list", xs), Argument("in_y_list", ys)], [Argument("out_z_list", zs)])

    # ct-pt multiply-accumulate nodes (cmp_sum / cmpac_sum): m = 5 -> one 4-way MAC + one 1-way MAC with partial sum
    for name, mk in (("cmpac", lambda i: CkksPlaintextNode(f"p_{i}", level=2)), ("cmpac_ringt", lambda i: CkksPlaintextRingtNode(f"p_{i}"))):
        set_fhe_param(ckks_param(4096, 5))
        cs = [CkksCiphertextNode(f"c_{i}", level=2) for i in range(5)]
        ps = [mk(i) for i in range(5)]
        z = ct_pt_mult_accumulate(cs, ps)
        emit("ckks_n4096_" + name, [Argument("in_c_list", cs), Argument("in_p_list", ps)], [Argument("out_z_list", [z])])

    # a graph this backend must REJECT at bind time, exactly as the reference does (mega_ag_executors_gpu.cu:212):
    # BFV ciphertext x full plaintext
    set_fhe_param(bfv_param(4096, 4))
    xs = [BfvCiphertextNode(f"x_{i}", level=2) for i in range(N_OP)]
    ys = [BfvPlaintextNode(f"y_{i}", level=2) for i in range(N_OP)]
    zs = [mult(xs[i], ys[i], f"z_{i}") for i in range(N_OP)]
    emit("bfv_n4096_cmp_unsupported", [Argument("xs", xs), Argument("ys", ys)], [Argument("zs", zs)])

    for name, f in (("cap_ringt", add), ("cmp_ringt", mult)):
        set_fhe_param(bfv_param(4096, 4))
        xs = [BfvCiphertextNode(f"x_{i}", level=2) for i in range(N_OP)]
        ys = [BfvPlaintextRingtNode(f"y_{i}") for i in range(N_OP)]
        zs = [f(xs[i], ys[i], f"z_{i}") for i in range(N_OP)]
        emit("bfv_n4096_" + name, [Argument("xs", xs), Argument("ys", ys)], [Argument("zs", zs)])

    set_fhe_param(bfv_param(4096, 4))
    cs = [BfvCiphertextNode(f"c_{i}", level=2) for i in range(3)]
    ps = [BfvPlaintextRingtNode(f"p_{i}") for i in range(3)]
    z = ct_pt_mult_accumulate(cs, ps)
    emit("bfv_n4096_cmpac_ringt", [Argument("cs", cs), Argument("ps", ps)], [Argument("zs", [z])])

    # CKKS ct+pt, ct*pt with full (NTT-domain) plaintexts
    set_fhe_param(ckks_param(4096, 5))
    xs = [CkksCiphertextNode(f"x_{i}", level=2) for i in range(N_OP)]
    ys = [CkksPlaintextNode(f"y_{i}", level=2) for i in range(N_OP)]
    zs = [add(mult(xs[i], ys[i]), ys[i], f"z_{i}") for i in range(N_OP)]
    emit("ckks_n4096_cmp_cap", [Argument("in_x_list", xs), Argument("in_y_list", ys)], [Argument("out_z_list", zs)])

    # ---- BFV N=4096, 4 primes
    set_fhe_param(bfv_param(4096, 4))
    lv = 3
    xs = [BfvCiphertextNode(f"x_{i}", level=lv) for i in range(N_OP)]
    ys = [BfvCiphertextNode(f"y_{i}", level=lv) for i in range(N_OP)]
    zs = [mult_relin(xs[i], ys[i], f"z_{i}") for i in range(N_OP)]
    emit("bfv_n4096_cmc_relin", [Argument("xs", xs), Argument("ys", ys)], [Argument("zs", zs)])

    set_fhe_param(bfv_param(4096, 4))
    xs = [BfvCiphertextNode(f"x_{i}", level=2) for i in range(N_OP)]
    ys = [rotate_cols(xs[i], 3, f"y_{i}") for i in range(N_OP)]     # NAF decomposition: 3 = 4 - 1 -> two rotate_col nodes
    emit("bfv_n4096_rotate_col3", [Argument("xs", xs)], [Argument("ys", ys)])

    set_fhe_param(bfv_param(4096, 4))
    xs = [BfvCiphertextNode(f"x_{i}", level=2) for i in range(N_OP)]
    ys = [rotate_rows(xs[i], f"y_{i}") for i in range(N_OP)]
    emit("bfv_n4096_rotate_row", [Argument("xs", xs)], [Argument("ys", ys)])

    # CKKS bootstrapping: the reference's toy bootstrap parameter set (custom_task.py:284-380: 25+5 primes, CtS depth 4,
    # Cos1 K=16 deg 30 with 3 double angles, StC depth 3, output level 9) on a ring small enough for the CPU oracle
    bp = CkksBtpParam.create_toy_param()
    bp.n = 2048
    bp.slots = 1024
    set_fhe_param(bp)
    xs = [CkksCiphertextNode(f"x_{i}", level=0) for i in range(2)]
    ys = [bootstrap(xs[i], f"y_{i}") for i in range(2)]
    emit("ckks_n2048_bootstrap", [Argument("in_x_list", xs)], [Argument("out_y_list", ys)])

    # sparsely packed bootstrap (unittests/test_gpu_ckks.py:618-646: the toy sparse set): 2^9 of the 2^10 slots
    bp = CkksBtpParam.create_toy_param()
    bp.n = 2048
    bp.slots = 512
    set_fhe_param(bp)
    xs = [CkksCiphertextNode(f"x_{i}", level=0) for i in range(2)]
    ys = [bootstrap(xs[i], f"y_{i}") for i in range(2)]
    emit("ckks_n2048_slots512_bootstrap", [Argument("in_x_list", xs)], [Argument("out_y_list", ys)])

    # unittests/test_gpu_ckks.py:596-616: multiply at level 3, rescale, drop to level 0, bootstrap
    bp = CkksBtpParam.create_toy_param()
    bp.n = 2048
    bp.slots = 1024
    set_fhe_param(bp)
    xs = [CkksCiphertextNode(f"x_{i}", level=3) for i in range(2)]
    ys = [CkksCiphertextNode(f"y_{i}", level=3) for i in range(2)]
    zs = [bootstrap(drop_level(rescale(mult_relin(xs[i], ys[i])), 2), f"result_{i}") for i in range(2)]
    emit("ckks_n2048_cmc_relin_rescale_bootstrap", [Argument("in_x_list", xs), Argument("in_y_list", ys)],
         [Argument("out_z_list", zs)])

    # application-shaped graph: packed conv2d, 1 -> 1 channels of 32x32, 3x3 kernel (two channel slots per ciphertext at N=4096)
    conv_fixture("ckks_n4096_conv2d_1in_1out_32x32_3x3", ckks_param(4096, 5), 4096, 1, 1, (32, 32), (3, 3), 2)


def conv2d_graph(n, n_in, n_out, shape, kernel, level):
    """The packed 2-D convolution layer of the reference's application benchmark, restated through the frontend API
    (graph shape of examples/benchmark_convolution/benchmark_convolution.py:44-163, stride 1, skip 1): every packed
    input ciphertext is rotated to each channel slot, each of those to every kernel offset (rows, then columns), each
    rotated copy is multiplied by its weight plaintext and accumulated, then one rescale and a bias plaintext."""
    h, w = shape
    kh, kw = kernel
    per_ct = (n // 2) // (h * w)                      # channels packed in one ciphertext
    pin, pout = -(-n_in // per_ct), -(-n_out // per_ct)
    xs = [CkksCiphertextNode(f"input_0_{i}", level=level) for i in range(pin)]
    wts = [[[CkksPlaintextNode(f"convw_{o}_{c}_{k}", level) for k in range(kh * kw)] for c in range(pin * per_ct)]
           for o in range(pout)]
    bias = [CkksPlaintextNode(f"convb_{o}", level - 1) for o in range(pout)]

    def both_sides(x, reach, unit):                   # [x rotated by -reach*unit, ..., x, ..., +reach*unit]
        if reach == 0:
            return [x]
        steps = [-i * unit for i in range(1, reach + 1)] + [i * unit for i in range(1, reach + 1)]
        r = rotate_cols(x, steps)
        return list(reversed(r[:reach])) + [x] + r[reach:]

    chans = []
    for x in xs:                                      # channel slots of each packed input
        chans.append(x)
        if per_ct > 1:
            chans += rotate_cols(x, [i * h * w for i in range(1, per_ct)])
    taps = [[t for r in both_sides(c, kw // 2, w) for t in both_sides(r, kh // 2, 1)] for c in chans]
    outs = []
    for o in range(pout):
        acc = None
        for c in range(pin * per_ct):
            for k in range(kh * kw):
                prod = mult(taps[c][k], wts[o][c][k])
                acc = prod if acc is None else add(acc, prod)
        outs.append(add(rescale(acc), bias[o]))
    return xs, wts, bias, outs


def conv_fixture(name, param, n, n_in, n_out, shape, kernel, level):
    set_fhe_param(param)
    xs, wts, bias, outs = conv2d_graph(n, n_in, n_out, shape, kernel, level)
    emit(name, [Argument("input_0", xs), Argument("convw", wts), Argument("convb", bias)], [Argument("output", outs)])


def bench_fixtures():
    """Benchmark-shaped graphs (examples/benchmark_gpu/benchmark_gpu.py:25-58: n_op disjoint mult_relin subgraphs),
    CKKS at the BASELINE N=2^16 / level-12 shape with the rescale the headline metric includes."""
    global OUT
    OUT = os.path.join(ROOT, "tests", "golden", "tasks_bench")
    D = P.CKKS_DEFAULT[65536]
    set_fhe_param(CkksParam.create_custom_param(n=65536, q=D["q"][:13], p=D["p"], scale=float(2 ** 45)))
    n_op, lv = 64, 12
    xs = [CkksCiphertextNode(f"x_{i}", level=lv) for i in range(n_op)]
    ys = [CkksCiphertextNode(f"y_{i}", level=lv) for i in range(n_op)]
    zs = [rescale(mult_relin(xs[i], ys[i]), f"z_{i}") for i in range(n_op)]
    emit("ckks_n65536_l12_cmc_relin_rescale_x64", [Argument("xs", xs), Argument("ys", ys)], [Argument("zs", zs)])
    B = P.BFV_DEFAULT[16384]
    set_fhe_param(BfvParam.create_custom_param(n=16384, q=B["q"], p=B["p"], t=B["t"]))
    n_op, lv = 256, 3
    xs = [BfvCiphertextNode(f"x_{i}", level=lv) for i in range(n_op)]
    ys = [BfvCiphertextNode(f"y_{i}", level=lv) for i in range(n_op)]
    zs = [mult_relin(xs[i], ys[i], f"z_{i}") for i in range(n_op)]
    emit("bfv_n16384_l3_cmc_relin_x256", [Argument("xs", xs), Argument("ys", ys)], [Argument("zs", zs)])
    # examples/benchmark_convolution config (4, 4, (32,32), (3,3)) on the default CKKS N=16384 chain, init level 2
    conv_fixture("ckks_n16384_conv2d_4in_4out_32x32_3x3", CkksParam.create_default_param(16384), 16384, 4, 4, (32, 32),
                 (3, 3), 2)


if __name__ == "__main__":
    main()
    bench_fixtures()
    for d in sorted(os.listdir(OUT)):
        print(d, os.listdir(os.path.join(OUT, d)))
