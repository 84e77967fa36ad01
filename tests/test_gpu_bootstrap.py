"""CKKS bootstrapping on the device (lattisense_amd/csrc/bootstrap.hip, the `bootstrap` node of a task graph) against its
oracle (oracle/ckks_bootstrap.py): the device plan's floating-point constants (encoded diagonals, Chebyshev coefficients) are
fed to the oracle program, after which every step is integer arithmetic on both sides -- the refreshed ciphertext must be
identical bit for bit -- and the reference's own assertion (mean precision >= 10 bits, unittests/test_gpu_ckks.cpp:763-781)
is applied to the device result."""
import numpy as np
import pytest

from tests.gpu_util import need_gpu

pytestmark = pytest.mark.gpu


def _setup(log_n, hamming, seed):
    from lattisense_amd import params
    from lattisense_amd.device import ALGO_CKKS, DeviceContext
    from oracle.client import Client
    from oracle.pyoracle import Oracle
    B = params.CKKS_BOOTSTRAP_65536       # the reference's bootstrap chain: 25 Q + 5 P (custom_task.py:387-420)
    N = 1 << log_n
    o = Oracle(N, B["q"], B["p"], 0)
    c = Client(o, seed=seed, hamming=hamming)
    ctx = DeviceContext(ALGO_CKKS, N, B["q"], B["p"])
    return B, N, o, c, ctx


def _oracle_plains(plan):
    return plan.oracle_plains()


@pytest.mark.parametrize("log_n,encapsulate,double_hoist,scatter", [(10, False, True, True), (11, True, True, True), (10, False, False, True),
                                                                     (10, False, True, False), (10, False, False, False)])
def test_bootstrap_bit_exact_against_the_oracle_program(log_n, encapsulate, double_hoist, scatter, monkeypatch):
    """double_hoist (the default): the baby-step / giant-step matrices keep their sums over Q u P and divide by P once per giant
    step and once at the end; False (LSA_BT_DOUBLE_HOIST=0 when the plan is made): one division per rotation.  The oracle
    program follows the plan's choice; either way device == oracle bit for bit."""
    need_gpu()
    if not double_hoist:
        monkeypatch.setenv("LSA_BT_DOUBLE_HOIST", "0")
    if not scatter:   # rotations as MAC / ModDown + a permutation kernel (k_permute_ext / k_permute) instead of scattered stores
        monkeypatch.setenv("LSA_ROT_SCATTER", "0")
    from lattisense_amd.device import BootstrapPlan
    from oracle.ckks_bootstrap import Bootstrapper, Ct, Evaluator
    from oracle.client import Client, mean_precision_bits
    B, N, o, c, ctx = _setup(log_n, None if encapsulate else 32, 11 + log_n)
    top = len(B["q"]) - 1
    D = float(2 ** 40)
    plan = BootstrapPlan(ctx, in_scale=D, out_scale=D)
    assert plan.out_level == 9 and plan.out_scale == D          # btp_output_level, parameter scale
    assert plan.double_hoist == double_hoist
    ev = Evaluator(o, c, top)
    rlk = ctx.upload_key(ev.rlk, top)
    keys = {e: c.gen_galois_key(e, top) for e in plan.galois_elements}
    ev.glk = dict(keys)          # both sides rotate with the same keys (freshly generated ones differ in their noise)
    glk = {e: ctx.upload_key(k, top) for e, k in keys.items()}
    dts = std = kd = ks = None
    if encapsulate:
        sparse = Client(o, seed=99, hamming=32)
        dts = c.gen_switching_key(c.s_ntt, sparse.s_ntt, 0)
        std = c.gen_switching_key(sparse.s_ntt, c.s_ntt, top)
        kd, ks = ctx.upload_key(dts, 0), ctx.upload_key(std, top)
    rng = np.random.default_rng(log_n)
    batch = 2
    zs = [rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2) for _ in range(batch)]
    cts = np.stack([c.ckks_encrypt(z, 0, D) for z in zs])          # [batch][2][1][N]
    out = plan.run(ctx.upload(cts), batch, rlk, glk, kd, ks)
    got = ctx.download(out, (batch, 2, plan.out_level + 1, N))
    # the oracle program with the device plan's constants
    bt = Bootstrapper(ev, out_scale=D, plains=_oracle_plains(plan), coeffs=plan.chebyshev(), double_hoist=plan.double_hoist)
    for b in range(batch):
        want = bt.bootstrap(Ct(cts[b], 0, D), top, dts, std)
        assert want.level == plan.out_level and want.scale == D
        assert np.array_equal(got[b], want.data)
        re, im = mean_precision_bits(zs[b], c.ckks_decrypt(got[b], D))
        assert re >= 10 and im >= 10
    # the oracle program needed no key beyond the ones the device plan listed (the reference planner's rotations + conjugation)
    assert sorted(ev.glk) == sorted(plan.galois_elements)
    plan.close()


@pytest.mark.parametrize("fixture", ["ckks_n2048_bootstrap", "ckks_n2048_slots512_bootstrap"])
def test_bootstrap_node_through_the_task_boundary(fixture):
    """Tasks compiled by the reference's frontend from its toy bootstrap parameter set (ring reduced to N=2048 for the CPU
    oracle), densely and sparsely packed: one `bootstrap` node per ciphertext, keys rlk / column rotations / conjugation /
    swk_dts / swk_std as the frontend lists them.  Through run_fhe_gpu_task: level 0 -> 9, >= 10 bits, and identical to the
    oracle program fed with the constants of an operator-level plan built from the same parameters."""
    need_gpu()
    import json
    import os
    from lattisense_amd.device import ALGO_CKKS, BootstrapPlan, DeviceContext
    from lattisense_amd.task import Argument, Ciphertext, FheTaskGpu, GaloisKey, KeySwitchKey
    from oracle.ckks_bootstrap import Bootstrapper, Ct, Evaluator
    from oracle.client import Client, mean_precision_bits
    from oracle.pyoracle import Oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from oracle.ckks_bootstrap import SparseBootstrapper
    path = os.path.join(root, "tests", "golden", "tasks", fixture)
    P = json.load(open(os.path.join(path, "mega_ag.json")))["parameter"]
    sig = json.load(open(os.path.join(path, "task_signature.json")))
    n, q, p, D = P["n"], P["q"], P["p"], float(P["scale"])
    top, k = len(q) - 1, len(p)
    o = Oracle(n, q, p, 0)
    c, sparse = Client(o, seed=41), Client(o, seed=42, hamming=32)      # dense main secret, sparse ephemeral secret
    ev = Evaluator(o, c, top)
    keys = {int(e): c.gen_galois_key(int(e), top) for e in sig["key"]["glk"]}
    ev.glk = dict(keys)
    dts = c.gen_switching_key(c.s_ntt, sparse.s_ntt, 0)
    std = c.gen_switching_key(sparse.s_ntt, c.s_ntt, top)
    rng = np.random.default_rng(43)
    ns = P["slots"]
    sparse = ns < n // 2
    zs = [rng.uniform(-1, 1, ns) + 1j * rng.uniform(-1, 1, ns) for _ in range(2)]
    cts = [c.ckks_encrypt(np.tile(z, (n // 2) // ns), 0, D) for z in zs]
    t = FheTaskGpu(path)
    ys = [Ciphertext.empty(1, P["btp_output_level"], n) for _ in range(2)]
    t.run([Argument("in_x_list", [Ciphertext(x) for x in cts]), Argument("rlk_ntt", [KeySwitchKey(ev.rlk, top, k)]),
           Argument("glk_ntt", [GaloisKey({e: KeySwitchKey(kk, top, k) for e, kk in keys.items()})]),
           Argument("swk_dts", [KeySwitchKey(dts, 0, k)]), Argument("swk_std", [KeySwitchKey(std, top, k)])],
          [Argument("out_y_list", ys)])
    assert t.last_run_stats()["gpu_batches"] == 1                       # both bootstraps in one batched program
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    plan = BootstrapPlan(ctx, P["btp_cts_depth"], P["btp_stc_depth"], P["btp_eval_mod_k"], P["btp_eval_mod_double_angle"],
                         P["btp_eval_mod_message_ratio"], D, D, log_slots=ns.bit_length() - 1)
    plains = plan.oracle_plains()
    cfg = (P["btp_cts_depth"], P["btp_stc_depth"], P["btp_eval_mod_k"], P["btp_eval_mod_double_angle"], P["btp_eval_mod_message_ratio"])
    if sparse:
        bt = SparseBootstrapper(ev, ns.bit_length() - 1, *cfg, out_scale=D, plains=plains, coeffs=plan.chebyshev(), double_hoist=plan.double_hoist)
    else:
        bt = Bootstrapper(ev, *cfg, out_scale=D, plains=plains, coeffs=plan.chebyshev(), double_hoist=plan.double_hoist)
    for i in range(2):
        re, im = mean_precision_bits(zs[i], c.ckks_decrypt(ys[i].data, D)[:ns])
        assert re >= 10 and im >= 10
        assert np.array_equal(ys[i].data, bt.bootstrap(Ct(cts[i], 0, D), top, dts, std).data)
    plan.close()
    t.close()


def test_sparse_slot_bootstrap_bit_exact_against_the_oracle_program():
    """Sparsely packed ciphertexts (2^9 of 2^10 slots at N = 2^11; the reference's sparse bootstrap parameter set is
    log_slots = 11 at N = 2^16, unittests/fixture.hpp:152-162)."""
    need_gpu()
    from lattisense_amd.device import BootstrapPlan
    from oracle.ckks_bootstrap import Ct, Evaluator, SparseBootstrapper
    from oracle.client import mean_precision_bits
    log_n, log_slots = 11, 9
    B, N, o, c, ctx = _setup(log_n, 32, 77)
    ns = 1 << log_slots
    top = len(B["q"]) - 1
    D = float(2 ** 40)
    plan = BootstrapPlan(ctx, in_scale=D, out_scale=D, log_slots=log_slots)
    assert plan.sparse and plan.out_level == 9 and plan.out_scale == D
    ev = Evaluator(o, c, top)
    keys = {e: c.gen_galois_key(e, top) for e in plan.galois_elements}
    ev.glk = dict(keys)
    rlk = ctx.upload_key(ev.rlk, top)
    glk = {e: ctx.upload_key(k, top) for e, k in keys.items()}
    rng = np.random.default_rng(78)
    z = rng.uniform(-1, 1, ns) + 1j * rng.uniform(-1, 1, ns)
    ct = c.ckks_encrypt(np.tile(z, (N // 2) // ns), 0, D)
    out = plan.run(ctx.upload(ct[None]), 1, rlk, glk)
    got = ctx.download(out, (1, 2, plan.out_level + 1, N))[0]
    want = SparseBootstrapper(ev, log_slots, out_scale=D, plains=plan.oracle_plains(), coeffs=plan.chebyshev(), double_hoist=plan.double_hoist).bootstrap(
        Ct(ct, 0, D), top)
    assert np.array_equal(got, want.data)
    re, im = mean_precision_bits(z, c.ckks_decrypt(got, D)[:ns])
    assert re >= 10 and im >= 10
    assert sorted(ev.glk) == sorted(plan.galois_elements)
    plan.close()


def test_multiply_then_bootstrap_task():
    """unittests/test_gpu_ckks.py:596-616 / test_gpu_ckks.cpp:783-808: mult+relin at level 3, rescale, drop to level 0,
    bootstrap -- one task.  The relinearisation key is the level-24 one (used at level 3); the bootstrap is
    plaintext-preserving, so the output keeps the product's scale (scale^2 / q_3), as the reference test declares it."""
    need_gpu()
    import json
    import os
    from lattisense_amd.device import ALGO_CKKS, BootstrapPlan, DeviceContext
    from lattisense_amd.task import Argument, Ciphertext, FheTaskGpu, GaloisKey, KeySwitchKey
    from oracle.ckks_bootstrap import Bootstrapper, Ct, Evaluator
    from oracle.client import Client, mean_precision_bits
    from oracle.pyoracle import Oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "tasks", "ckks_n2048_cmc_relin_rescale_bootstrap")
    P = json.load(open(os.path.join(path, "mega_ag.json")))["parameter"]
    sig = json.load(open(os.path.join(path, "task_signature.json")))
    n, q, p, D = P["n"], P["q"], P["p"], float(P["scale"])
    top, k, lvl = len(q) - 1, len(p), 3
    o = Oracle(n, q, p, 0)
    c, sparse = Client(o, seed=51), Client(o, seed=52, hamming=32)
    ev = Evaluator(o, c, top)
    keys = {int(e): c.gen_galois_key(int(e), top) for e in sig["key"]["glk"]}
    ev.glk = dict(keys)
    dts = c.gen_switching_key(c.s_ntt, sparse.s_ntt, 0)
    std = c.gen_switching_key(sparse.s_ntt, c.s_ntt, top)
    rng = np.random.default_rng(53)
    xm = [rng.uniform(-1, 1, n // 2) + 1j * rng.uniform(-1, 1, n // 2) for _ in range(2)]
    ym = [rng.uniform(-1, 1, n // 2) + 1j * rng.uniform(-1, 1, n // 2) for _ in range(2)]
    xs = [c.ckks_encrypt(m, lvl, D) for m in xm]
    ys = [c.ckks_encrypt(m, lvl, D) for m in ym]
    t = FheTaskGpu(path)
    zs = [Ciphertext.empty(1, P["btp_output_level"], n) for _ in range(2)]
    t.run([Argument("in_x_list", [Ciphertext(x) for x in xs]), Argument("in_y_list", [Ciphertext(y) for y in ys]),
           Argument("rlk_ntt", [KeySwitchKey(ev.rlk, top, k)]),
           Argument("glk_ntt", [GaloisKey({e: KeySwitchKey(kk, top, k) for e, kk in keys.items()})]),
           Argument("swk_dts", [KeySwitchKey(dts, 0, k)]), Argument("swk_std", [KeySwitchKey(std, top, k)])],
          [Argument("out_z_list", zs)])
    ctx = DeviceContext(ALGO_CKKS, n, q, p)
    plan = BootstrapPlan(ctx, P["btp_cts_depth"], P["btp_stc_depth"], P["btp_eval_mod_k"], P["btp_eval_mod_double_angle"],
                         P["btp_eval_mod_message_ratio"], D, D)
    bt = Bootstrapper(ev, P["btp_cts_depth"], P["btp_stc_depth"], P["btp_eval_mod_k"], P["btp_eval_mod_double_angle"],
                      P["btp_eval_mod_message_ratio"], out_scale=D, plains=plan.oracle_plains(), coeffs=plan.chebyshev(), double_hoist=plan.double_hoist)
    prod_scale = D * D / q[lvl]
    for i in range(2):
        z = o.ckks_mult_relin_rescale(lvl, xs[i], ys[i], ev.rlk, top)[:, :1]          # level 2, dropped to level 0
        want = bt.bootstrap(Ct(z, 0, D), top, dts, std)
        assert np.array_equal(zs[i].data, want.data)
        re, im = mean_precision_bits(xm[i] * ym[i], c.ckks_decrypt(zs[i].data, prod_scale))
        assert re >= 10 and im >= 10
    plan.close()
    t.close()


@pytest.mark.parametrize("log_n", [11, 13])
def test_device_constants_against_independently_computed_ones(log_n):
    """The only floating-point work of the device bootstrap is its CONSTANTS (encoded DFT diagonals, Chebyshev coefficients,
    lsa_bootstrap_plaintext / lsa_bootstrap_chebyshev).  The bit-exactness tests above feed them to the oracle program; here
    they are checked against the oracle module's OWN computation (numpy FFT-layer algebra + chebinterpolate,
    oracle/ckks_bootstrap.py expected_constants -- nothing shared with csrc/bootstrap.hip).  Tolerance: every encoded
    coefficient within 2^-30 of the plaintext scale (the scale is the level's prime, 2^39..2^60, and both sides round
    double-precision products of magnitude <= scale, so the honest bound is a few ulp = scale * 2^-50; 2^-30 leaves room for the
    different evaluation order of the merged layers), identical across the limbs of a plaintext; Chebyshev coefficients
    within 1e-9 absolute (they enter at scale 2^60: an error of 1e-9 there is 30 bits below the 10-bit precision the
    reference asserts)."""
    need_gpu()
    from lattisense_amd.device import BootstrapPlan
    from oracle.ckks_bootstrap import Bootstrapper, Evaluator
    B, N, o, c, ctx = _setup(log_n, 32, 5)
    top = len(B["q"]) - 1
    D = float(2 ** 40)
    plan = BootstrapPlan(ctx, in_scale=D, out_scale=D)
    ev = Evaluator.__new__(Evaluator)          # no keys needed: only q() and encode() are used
    ev.o, ev.c, ev.n = o, c, N
    want, coeffs, out_level = Bootstrapper(ev, out_scale=D, double_hoist=plan.double_hoist).expected_constants(D, top)
    assert out_level == plan.out_level
    dev = plan.oracle_plains()
    levels = plan.oracle_levels()
    assert sorted(dev) == sorted(want)
    assert plan.double_hoist                                        # the default: special-prime rows on the BSGS matrices
    worst = 0.0
    for key in sorted(want):
        assert sorted(dev[key]) == sorted(want[key]), key          # the same diagonal index set
        for k in want[key]:
            a, b = dev[key][k], want[key][k]
            assert a.shape == b.shape, (key, k)
            lvl = levels[key]
            assert a.shape[0] in (lvl + 1, lvl + 1 + len(o.p))
            deltas = []
            for row in (0, lvl, a.shape[0] - 1):                    # first and last prime of the level, last row (a special prime
                j = ev._mi(lvl, row)                                # when the matrix is double-hoisted)
                q = o.mod[j]
                d = (o.intt(j, a[row]).astype(object) - o.intt(j, b[row]).astype(object)) % q
                d = np.array([int(x) - q if int(x) > q // 2 else int(x) for x in d], dtype=np.float64)
                deltas.append(d)
            assert np.array_equal(deltas[0], deltas[1]) and np.array_equal(deltas[0], deltas[2]), (key, k)   # one integer polynomial, not per-limb noise
            rel = np.max(np.abs(deltas[0])) / float(o.mod[lvl])
            worst = max(worst, rel)
            assert rel < 2.0 ** -30, (key, k, rel)
    dc = np.max(np.abs(plan.chebyshev()[: len(coeffs)] - coeffs))
    assert dc < 1e-9, dc
    print("max relative constant deviation 2^%.1f, chebyshev %.2e" % (np.log2(max(worst, 1e-300)), dc))
    plan.close()


def test_bootstrap_reference_parameter_set_full_size():
    """The reference's bootstrap test at its real size (unittests/test_gpu_ckks.cpp:763-781, parameter set
    unittests/fixture.hpp:120-162 = frontend default N16QP1546H192H32): N = 2^16, 25 Q + 5 P primes, dense packing, main secret
    of Hamming weight 192, ephemeral secret of weight 32 (swk_dts / swk_std), REAL keys from the test client -- relinearisation
    key, the 47 planner rotations + conjugation, both switching keys.  One level-0 ciphertext in, level 9 out, and the
    reference's own assertion: decrypted mean precision >= 10 bits.  (The oracle program is not replayed at this size: a
    CPU bootstrap takes minutes; its bit-exactness is pinned at N = 2^10..2^11 above.)"""
    need_gpu()
    from lattisense_amd.device import BootstrapPlan
    from oracle.client import Client, mean_precision_bits
    B, N, o, c, ctx = _setup(16, 192, 2026)
    top = len(B["q"]) - 1
    D = float(2 ** 40)
    plan = BootstrapPlan(ctx, in_scale=D, out_scale=D)
    assert plan.out_level == 9 and not plan.sparse and len(plan.galois_elements) == 48
    sparse = Client(o, seed=2027, hamming=32)
    rlk = ctx.upload_key(c.gen_relin_key(top), top)
    glk = {}
    for e in plan.galois_elements:             # one key at a time: 157 MB each on the host, resident on the device afterwards
        glk[e] = ctx.upload_key(c.gen_galois_key(e, top), top)
    kd = ctx.upload_key(c.gen_switching_key(c.s_ntt, sparse.s_ntt, 0), 0)
    ks = ctx.upload_key(c.gen_switching_key(sparse.s_ntt, c.s_ntt, top), top)
    rng = np.random.default_rng(16)
    z = rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2)
    ct = c.ckks_encrypt(z, 0, D)[None]
    out = plan.run(ctx.upload(ct), 1, rlk, glk, kd, ks)
    got = ctx.download(out, (1, 2, plan.out_level + 1, N))[0]
    re, im = mean_precision_bits(z, c.ckks_decrypt(got, D))
    print("N=2^16 bootstrap: level 0 -> %d, mean precision %.1f / %.1f bits" % (plan.out_level, re, im))
    assert re >= 10 and im >= 10
    plan.close()


@pytest.mark.parametrize("sine_deg,arcsine_deg", [(63, 0), (30, 7), (63, 7)])
def test_wider_evalmod_configurations(sine_deg, arcsine_deg):
    """EvalMod configurations the reference forwards besides its default (gpu_wrapper.cu:100-103: btp_eval_mod_sine_deg,
    btp_eval_mod_arcsine_deg): a degree-63 cosine interpolant (6 levels) and the arcsine correction (degree 7: 3 more levels), on a
    chain with enough 60-bit EvalMod primes.  Device == oracle program bit for bit (fed the device plan's constants), the
    device's arcsine coefficients == the oracle's own Taylor coefficients, and the reference's precision bar (>= 10 bits)."""
    need_gpu()
    from lattisense_amd import params
    from lattisense_amd.device import ALGO_CKKS, BootstrapPlan, DeviceContext
    from oracle.ckks_bootstrap import Bootstrapper, Ct, Evaluator, arcsine_coeffs
    from oracle.client import Client, mean_precision_bits
    from oracle.pyoracle import Oracle
    N = 1 << 10
    B = params.CKKS_BOOTSTRAP_65536
    depth = (6 if sine_deg > 31 else 5) + 3 + (3 if arcsine_deg else 0)
    # the reference chain's shape (custom_task.py:387-420) with `depth` EvalMod primes instead of 8: q0, output levels, StC, sine, CtS
    sine = params.ntt_primes_below(60, 1 << 16, depth + 4, avoid=B["q"])[4:]
    q = B["q"][:13] + sine + B["q"][21:]
    o = Oracle(N, q, B["p"], 0)
    c = Client(o, seed=77 + sine_deg + arcsine_deg, hamming=32)
    ctx = DeviceContext(ALGO_CKKS, N, q, B["p"])
    top = len(q) - 1
    D = float(2 ** 40)
    plan = BootstrapPlan(ctx, in_scale=D, out_scale=D, sine_deg=sine_deg, arcsine_deg=arcsine_deg)
    assert plan.out_level == 9
    cheb, asin = plan.evalmod_constants()
    assert len(cheb) == (64 if sine_deg > 31 else 32)
    if arcsine_deg:
        assert np.allclose(asin, arcsine_coeffs(arcsine_deg), rtol=1e-15, atol=0)
        assert abs(asin[1] - 1.0) < 1e-15 and abs(asin[3] - 1.0 / 6) < 1e-15 and abs(asin[5] - 3.0 / 40) < 1e-15
    else:
        assert asin is None
    ev = Evaluator(o, c, top)
    rlk = ctx.upload_key(ev.rlk, top)
    keys = {e: c.gen_galois_key(e, top) for e in plan.galois_elements}
    ev.glk = dict(keys)
    glk = {e: ctx.upload_key(k, top) for e, k in keys.items()}
    rng = np.random.default_rng(sine_deg)
    z = rng.uniform(-1, 1, N // 2) + 1j * rng.uniform(-1, 1, N // 2)
    cts = np.stack([c.ckks_encrypt(z, 0, D)])
    out = plan.run(ctx.upload(cts), 1, rlk, glk)
    got = ctx.download(out, (1, 2, plan.out_level + 1, N))
    bt = Bootstrapper(ev, out_scale=D, plains=plan.oracle_plains(), coeffs=cheb, sine_deg=sine_deg, arcsine_deg=arcsine_deg, asin=asin, double_hoist=plan.double_hoist)
    want = bt.bootstrap(Ct(cts[0], 0, D), top)
    assert want.level == plan.out_level and want.scale == D
    assert np.array_equal(got[0], want.data)
    re, im = mean_precision_bits(z, c.ckks_decrypt(got[0], D))
    assert re >= 10 and im >= 10, (re, im)
    plan.close()


@pytest.mark.parametrize("log_n,cts_depth,stc_depth,log_slots", [(10, 2, 2, 0), (10, 3, 3, 0), (11, 2, 3, 7)])
def test_other_matrix_depths_bit_exact(log_n, cts_depth, stc_depth, log_slots):
    """CoeffsToSlots / SlotsToCoeffs depths other than the reference default (btp_cts_depth / btp_stc_depth are plain parameters,
    gpu_wrapper.cu:94-99): depth 2 gives 32- and 63-diagonal matrices -- more than 8 baby steps, so the inner sums take the
    one-launch-per-giant-step path of the double-hoisted transform instead of the single multi-MAC launch -- also with sparse
    packing.  Device == oracle program bit for bit, precision above the reference's bar."""
    need_gpu()
    from lattisense_amd import params
    from lattisense_amd.device import ALGO_CKKS, BootstrapPlan, DeviceContext
    from oracle.ckks_bootstrap import Bootstrapper, Ct, Evaluator, SparseBootstrapper
    from oracle.client import Client, mean_precision_bits
    from oracle.pyoracle import Oracle
    B = params.CKKS_BOOTSTRAP_65536
    N = 1 << log_n
    # the reference chain's shape (custom_task.py:387-420: q0, 9 output levels, StC primes, 8 EvalMod primes, CtS primes) with as
    # many StC / CtS primes as the depths ask for
    q = B["q"][:10] + B["q"][10:10 + stc_depth] + B["q"][13:21] + B["q"][21:21 + cts_depth]
    o = Oracle(N, q, B["p"], 0)
    c = Client(o, seed=log_n + cts_depth, hamming=32)
    ctx = DeviceContext(ALGO_CKKS, N, q, B["p"])
    top = len(q) - 1
    D = float(2 ** 40)
    plan = BootstrapPlan(ctx, cts_depth, stc_depth, 16, 3, 256.0, D, D, log_slots=log_slots)
    assert plan.double_hoist
    ev = Evaluator(o, c, top)
    keys = {e: c.gen_galois_key(e, top) for e in plan.galois_elements}
    ev.glk = dict(keys)
    rlk = ctx.upload_key(ev.rlk, top)
    glk = {e: ctx.upload_key(k, top) for e, k in keys.items()}
    ns = (1 << log_slots) if log_slots else N // 2
    rng = np.random.default_rng(log_n)
    z = rng.uniform(-1, 1, ns) + 1j * rng.uniform(-1, 1, ns)
    ct = c.ckks_encrypt(np.tile(z, (N // 2) // ns), 0, D)
    got = ctx.download(plan.run(ctx.upload(ct[None]), 1, rlk, glk), (1, 2, plan.out_level + 1, N))[0]
    cfg = dict(out_scale=D, plains=plan.oracle_plains(), coeffs=plan.chebyshev(), double_hoist=True)
    if plan.sparse:
        bt = SparseBootstrapper(ev, log_slots, cts_depth, stc_depth, 16, 3, 256.0, **cfg)
    else:
        bt = Bootstrapper(ev, cts_depth, stc_depth, 16, 3, 256.0, **cfg)
    want = bt.bootstrap(Ct(ct, 0, D), top)
    assert want.level == plan.out_level
    assert np.array_equal(got, want.data)
    re, im = mean_precision_bits(z, c.ckks_decrypt(got, D)[:ns])
    assert re >= 10 and im >= 10
    assert sorted(ev.glk) == sorted(plan.galois_elements)
    plan.close()
    ctx.close()
