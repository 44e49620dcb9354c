"""Which set should an active-set solve start from?  Rounds needed (numpy prototype, inputs-first + local-maximum rule) for the QPs of consecutive
closed-loop steps (gpurun_out/cl_qps.npz) from different starting sets."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities, qp_box
from as_polish import boxes, Pd, N, n, nx, nu, nz
from as_inputs_first import polish3
d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))


def shift(act):
    a = np.zeros_like(act)
    a[:n - nz] = act[nz:]
    a[:nx] = 0
    return a


def sets_of(step, b):
    A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
    E, e = build_equalities(A, Bm, c, -x0a)
    ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
    lo1, hi1 = boxes(ub1); lo2, hi2 = boxes(ub2)
    z1, _, lu1, ll1, ok1, _ = qp_box(Pd, q, E, e, lo1, hi1)
    z2, _, lu2, ll2, ok2, _ = qp_box(Pd, q, E, e, lo2, hi2)
    a1 = np.where(lu1 > hi1 - z1, 1, np.where(ll1 > z1 - lo1, -1, 0)); a1[:nx] = 0
    a2 = np.where(lu2 > hi2 - z2, 1, np.where(ll2 > z2 - lo2, -1, 0)); a2[:nx] = 0
    return dict(E=E, e=e, q=q, b1=(lo1, hi1), b2=(lo2, hi2), a1=a1, a2=a2)


tot = {}
for b in range(0, 64, 7):
    prev = None
    for step in range(1, 6):
        if not d[f"success_{step}"][b] or (prev is not None and not d[f"success_{step-1}"][b]):
            prev = None
        cur = sets_of(step, b)
        if prev is not None:
            uni = lambda x, y: np.where(x != 0, x, y)
            cands1 = {"A1(t-1)": prev["a1"], "shift A1(t-1)": shift(prev["a1"]), "A1 u shiftA1": uni(prev["a1"], shift(prev["a1"]))}
            cands2 = {"A1(t)": cur["a1"], "shift A2(t-1)": shift(prev["a2"]), "A1(t) u shiftA2(t-1)": uni(cur["a1"], shift(prev["a2"])), "A1(t) u A2(t-1)": uni(cur["a1"], prev["a2"]),
                      "A1(t) u (A2-A1)(t-1)": uni(cur["a1"], np.where(prev["a1"] == 0, prev["a2"], 0))}
            line = f"inst {b} step {step}:"
            for nm, a0 in cands1.items():
                a, z, r, ok, h = polish3(cur["E"], cur["e"], cur["q"], *cur["b1"], a0, "inputs_first")
                tot.setdefault("QP1 " + nm, []).append(r if ok else 99)
                line += f" QP1<{nm}>={r if ok else 'F'}"
            for nm, a0 in cands2.items():
                a, z, r, ok, h = polish3(cur["E"], cur["e"], cur["q"], *cur["b2"], a0, "inputs_first")
                tot.setdefault("QP2 " + nm, []).append(r if ok else 99)
                line += f" QP2<{nm}>={r if ok else 'F'}"
            print(line, flush=True)
        prev = cur
for k, v in tot.items():
    v = np.array(v)
    print(f"{k:22s} mean rounds {v[v < 99].mean():.2f}  fails {np.sum(v == 99)}/{len(v)}")
