"""Generic CKKS task-graph interpreters for the parity tests (test infrastructure, like oracle/).

`eval_cipher` walks a compiled task graph (mega_ag.json, reference: mega_ag_runners/mega_ag.cpp:125-657) node by node with
the CPU oracle, giving the bit-exact expected ciphertext of every output; `eval_plain` walks the same graph on slot
vectors (rotate_col = cyclic shift of the slots, mult = element-wise product ...), giving the message the output must
decrypt to.  Op semantics follow the reference's executors (mega_ag_runners/gpu/mega_ag_executors_gpu.cu:71-426).
"""
import numpy as np


def _order(g):
    done = set(int(i) for i in g["inputs"])
    pending = {int(k): v for k, v in g["compute"].items()}
    while pending:
        ready = [k for k, v in pending.items() if all(i in done for i in v["inputs"])]
        assert ready, "graph has a cycle or a dangling input"
        for k in sorted(ready):
            yield pending.pop(k)
            done.update(g["compute"][str(k)]["outputs"])


def _limbwise(o, op, a, b, lvl):
    return np.stack([np.stack([o.vec(op, j, a[pl, j], b[pl, j]) for j in range(lvl + 1)]) for pl in range(a.shape[0])])


def eval_cipher(g, o, values, glk):
    """values: {data index: ndarray} for every graph input of type ct / pt ([L][N] NTT-domain residues);
    glk: {galois element: compact key array}.  Returns the dict extended with every computed datum."""
    data = g["data"]
    vals = dict(values)
    for node in _order(g):
        ins = node["inputs"]
        kinds = [data[str(i)]["type"] for i in ins]
        out = node["outputs"][0]
        t = node["type"]
        lvl = data[str(ins[0])]["level"]
        a = vals[ins[0]]
        if t in ("add", "sub", "mult") and len(ins) >= 2 and kinds[1] == "pt":
            p = vals[ins[1]]
            if t == "mult":      # every polynomial times the plaintext
                r = np.stack([np.stack([o.vec("mul", j, a[pl, j], p[j]) for j in range(lvl + 1)]) for pl in range(a.shape[0])])
            else:                # plaintext added to / subtracted from c0 only
                r = a.copy()
                r[0] = np.stack([o.vec(t, j, a[0, j], p[j]) for j in range(lvl + 1)])
        elif t in ("add", "sub"):
            r = _limbwise(o, t, a, vals[ins[1]] if len(ins) > 1 and kinds[1] in ("ct", "ct3") else a, lvl)
        elif t == "neg":
            r = np.stack([np.stack([o.vec("neg", j, a[pl, j]) for j in range(lvl + 1)]) for pl in range(a.shape[0])])
        elif t == "mult":
            r = o.ckks_mult(lvl, a, vals[ins[1]] if len(ins) > 1 and kinds[1] == "ct" else a)
        elif t == "rescale":
            r = o.ckks_rescale(lvl, a)
        elif t == "drop_level":
            r = a[:, : data[str(out)]["level"] + 1].copy()
        elif t in ("rotate_col", "rotate_row"):
            e = data[str(ins[1])]["galois_element"]
            r = o.ckks_rotate(lvl, a, e, glk[e], data[str(ins[1])]["level"])
        else:
            raise NotImplementedError(t)
        vals[out] = r
    return vals


def eval_plain(g, values, q):
    """values: {data index: (slot vector, scale)}; q: the modulus chain (rescale divides the scale by q[level])."""
    data = g["data"]
    vals = dict(values)
    for node in _order(g):
        ins = node["inputs"]
        out = node["outputs"][0]
        t = node["type"]
        (a, sa) = vals[ins[0]]
        second = vals[ins[1]] if len(ins) > 1 and data[str(ins[1])]["type"] in ("ct", "pt") else None
        if t == "add":
            r = (a + (second[0] if second else a), sa)
        elif t == "sub":
            r = (a - (second[0] if second else a), sa)
        elif t == "neg":
            r = (-a, sa)
        elif t == "mult":
            b, sb = second if second else (a, sa)
            r = (a * b, sa * sb)
        elif t == "rescale":
            r = (a, sa / q[data[str(ins[0])]["level"]])
        elif t == "drop_level":
            r = (a, sa)
        elif t == "rotate_col":
            r = (np.roll(a, -node["step"]), sa)
        elif t == "rotate_row":
            r = (np.conj(a), sa)
        else:
            raise NotImplementedError(t)
        vals[out] = r
    return vals
