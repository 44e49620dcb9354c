import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities, qp_box
from as_polish import boxes, Pd, N, n, nx, nu, nz, as_solve


def stage_comp(e):
    return e // nz, e % nz


def polish2(E, e, q, lo, hi, act0, rule, max_rounds=12, tol=1e-9, verbose=False):
    act = act0.copy()
    hist = []
    for r in range(max_rounds):
        z, nu_, gr, cond = as_solve(E, e, q, lo, hi, act)
        qs = max(1.0, np.abs(q).max()); t = tol * qs
        rel = ((act > 0) & (gr > t)) | ((act < 0) & (-gr > t))
        vu = np.where((act == 0) & (z > hi + t), z - hi, 0.0); vl = np.where((act == 0) & (z < lo - t), lo - z, 0.0)
        vu[:nx] = 0; vl[:nx] = 0
        v = np.maximum(vu, vl)
        add = v > 0
        hist.append((int(rel.sum()), int(add.sum())))
        if rel.sum() + add.sum() == 0:
            return act, z, r, True, hist
        if rule == "localmax":
            # state components: only local maxima of the violation along the horizon; inputs: all
            V = np.zeros((N + 1, nz)); V.flat[:n] = v
            keep = np.zeros_like(V, dtype=bool)
            for i in range(nz):
                col = V[:, i]
                if i >= nx:
                    keep[:, i] = col > 0
                else:
                    for k in range(N + 1):
                        if col[k] > 0 and col[k] >= (col[k - 1] if k > 0 else 0) and col[k] >= (col[k + 1] if k < N else 0):
                            keep[k, i] = True
            add = keep.flat[:n] & add
        elif rule == "localmax_all":     # local maxima for every component
            V = np.zeros((N + 1, nz)); V.flat[:n] = v
            keep = np.zeros_like(V, dtype=bool)
            for i in range(nz):
                col = V[:, i]
                for k in range(N + 1):
                    if col[k] > 0 and col[k] >= (col[k - 1] if k > 0 else 0) and col[k] >= (col[k + 1] if k < N else 0):
                        keep[k, i] = True
            add = keep.flat[:n] & add
        act[rel] = 0
        act[add & (vu > 0)] = 1; act[add & (vl > 0)] = -1
    return act, z, max_rounds, False, hist


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cl"
    if which == "cl":
        d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))
        cases = [(step, b) for step in (1, 2, 3, 4) for b in range(0, 256, 40)]
    else:
        from robust_nonlinear_mpc_amd import make_batch
        bt = make_batch("rocket", os.path.join(ROOT, "tests", "golden", "sweep_rocket_N20_s0.npz"), 24, seed=1234)
        cases = [(None, b) for b in range(24)]
    for step, b in cases:
        if which == "cl":
            A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
        else:
            A, Bm, c, g, gN, q, x0a = (bt[k][b] for k in ("A", "B", "c", "g", "gN", "q", "x0_arg"))
            ub2 = None
        E, e = build_equalities(A, Bm, c, -x0a)
        ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
        lo1, hi1 = boxes(ub1)
        z1, nu1, lu1, ll1, ok, its = qp_box(Pd, q, E, e, lo1, hi1)
        act1 = np.where(lu1 > hi1 - z1, 1, np.where(ll1 > z1 - lo1, -1, 0)); act1[:nx] = 0
        line = f"{which} step {step} inst {b}: |A1|={np.sum(act1!=0)}"
        # (a) QP1 from the empty set (cold active-set start from the equality-constrained optimum)
        for rule in ("all", "localmax", "localmax_all"):
            a, z, r, okp, hist = polish2(E, e, q, lo1, hi1, np.zeros(n, dtype=int), rule)
            line += f" | QP1 cold {rule}: {r if okp else 'FAIL'} {hist[:4]}"
        print(line, flush=True)
        if ub2 is not None:
            lo2, hi2 = boxes(ub2)
            line = "        QP2 from A1:"
            for rule in ("all", "localmax", "localmax_all"):
                a, z, r, okp, hist = polish2(E, e, q, lo2, hi2, act1, rule)
                line += f" {rule}: {r if okp else 'FAIL'} {hist[:5]} |"
            print(line, flush=True)
