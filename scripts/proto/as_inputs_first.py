import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities, qp_box
from as_polish import boxes, Pd, N, n, nx, nu, nz, as_solve
from robust_nonlinear_mpc_amd import make_batch


def polish3(E, e, q, lo, hi, act0, rule, max_rounds=12, tol=1e-9):
    act = act0.copy()
    hist = []
    for r in range(max_rounds):
        z, nu_, gr, cond = as_solve(E, e, q, lo, hi, act)
        qs = max(1.0, np.abs(q).max()); t = tol * qs
        rel = ((act > 0) & (gr > t)) | ((act < 0) & (-gr > t))
        vu = np.where((act == 0) & (z > hi + t), z - hi, 0.0); vl = np.where((act == 0) & (z < lo - t), lo - z, 0.0)
        vu[:nx] = 0; vl[:nx] = 0
        v = np.maximum(vu, vl)
        hist.append((int(rel.sum()), int((v > 0).sum())))
        if rel.sum() + (v > 0).sum() == 0:
            return act, z, r, True, hist
        V = np.zeros((N + 1, nz)); V.flat[:n] = v
        keep = np.zeros_like(V, dtype=bool)
        any_input = (V[:, nx:] > 0).any()
        for i in range(nz):
            col = V[:, i]
            if i >= nx:
                keep[:, i] = col > 0
            elif not (rule == "inputs_first" and any_input):
                for k in range(N + 1):
                    if col[k] > 0 and col[k] >= (col[k - 1] if k > 0 else 0) and col[k] >= (col[k + 1] if k < N else 0):
                        keep[k, i] = True
        add = keep.flat[:n] & (v > 0)
        act[rel] = 0
        act[add & (vu > 0)] = 1; act[add & (vl > 0)] = -1
    return act, z, max_rounds, False, hist


which = sys.argv[1] if len(sys.argv) > 1 else "syn"
if which == "syn":
    bt = make_batch("rocket", os.path.join(ROOT, "tests", "golden", "sweep_rocket_N20_s0.npz"), 64, seed=1234)
    cases = [(None, b) for b in range(64)]
else:
    d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))
    cases = [(step, b) for step in (1, 3, 5) for b in range(0, 256, 32)]
tot = {"localmax": [0, 0, 0], "inputs_first": [0, 0, 0]}
for step, b in cases:
    if which == "syn":
        A, Bm, c, g, gN, q, x0a = (bt[k][b] for k in ("A", "B", "c", "g", "gN", "q", "x0_arg")); ub2 = None
    else:
        A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
    E, e = build_equalities(A, Bm, c, -x0a)
    ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
    lo1, hi1 = boxes(ub1)
    line = f"{which} {step} {b}:"
    probs = [("QP1", lo1, hi1, np.zeros(n, dtype=int))]
    if ub2 is not None:
        lo2, hi2 = boxes(ub2)
        probs.append(("QP2cold", lo2, hi2, np.zeros(n, dtype=int)))
    for nm, lo, hi, a0 in probs:
        for rule in ("localmax", "inputs_first"):
            a, z, r, okp, hist = polish3(E, e, q, lo, hi, a0, rule)
            tot[rule][0] += 1; tot[rule][1] += int(okp); tot[rule][2] += r if okp else 0
            line += f" {nm} {rule}: {r if okp else 'FAIL'} {hist[:3]} |"
    print(line, flush=True)
print({k: (v[0], v[1], v[2] / max(1, v[1])) for k, v in tot.items()})
