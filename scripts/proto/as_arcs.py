"""Hard regime (closed loop from the script's x0, ~34 active bounds): rounds of the warm active-set iteration of QP #2 (from QP #1's set) under
variants of the rule for STATE bounds: touch points only (local maxima of the violation, the kernel's rule) against local maxima plus their
neighbours along the arc whose violation is at least theta x the maximum."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities, qp_box
from as_polish import boxes, Pd, N, n, nx, nu, nz, as_solve
d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))


def run(E, e, q, lo, hi, act0, theta, inputs_first=True, max_rounds=30, ctol=1e-6):
    act = act0.copy(); hist = []
    qs = max(1.0, np.abs(q).max()); t = ctol * qs
    seen = set()
    for r in range(max_rounds):
        z, nu_, gr, cond = as_solve(E, e, q, lo, hi, act)
        rel = ((act > 0) & (gr > t)) | ((act < 0) & (-gr > t))
        vu = np.where((act == 0) & (z > hi + t), z - hi, 0.0); vl = np.where((act == 0) & (z < lo - t), lo - z, 0.0)
        vu[:nx] = 0; vl[:nx] = 0
        v = np.maximum(vu, vl)
        nv = int((v > 0).sum())
        if rel.sum() + nv == 0:
            return r, True, hist
        if nv > 64 or not np.isfinite(z).all():
            return r, False, hist + ["blow-up"]
        V = np.zeros((N + 1, nz)); V.flat[:n] = v
        keep = np.zeros_like(V, dtype=bool)
        any_input = (V[:, nx:] > 0).any()
        for i in range(nz):
            col = V[:, i]
            if i >= nx:
                keep[:, i] = col > 0
            elif not (inputs_first and any_input):
                for k in range(N + 1):
                    if col[k] > 0 and col[k] >= (col[k - 1] if k > 0 else 0) and col[k] >= (col[k + 1] if k < N else 0):
                        keep[k, i] = True
                        if theta < 1.0:     # neighbours along the arc, as long as their violation stays above theta x this maximum
                            for dk in (-1, 1):
                                kk = k + dk
                                while 0 <= kk <= N and col[kk] >= theta * col[k] and col[kk] > 0:
                                    keep[kk, i] = True; kk += dk
        add = keep.flat[:n] & (v > 0)
        hist.append((int(rel.sum()), nv, int(add.sum())))
        act[rel] = 0
        act[add & (vu > 0)] = 1; act[add & (vl > 0)] = -1
        key = act.tobytes()
        if key in seen:
            return r, False, hist + ["cycle"]
        seen.add(key)
    return max_rounds, False, hist


steps = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3]
variants = [("touch (kernel)", 1.0, True), ("theta 0.9", 0.9, True), ("theta 0.7", 0.7, True), ("theta 0.5", 0.5, True), ("theta 0.3", 0.3, True), ("theta 0.5, no inputs-first", 0.5, False)]
tot = {v[0]: [] for v in variants}
for step in steps:
    for b in range(0, 64, 5):
        if f"success_{step}" in d and d[f"qp_stats_{step}"][b, 1, 6] != 0:
            continue
        A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
        E, e = build_equalities(A, Bm, c, -x0a)
        ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
        lo1, hi1 = boxes(ub1); lo2, hi2 = boxes(ub2)
        z1, _, lu1, ll1, ok1, _ = qp_box(Pd, q, E, e, lo1, hi1)
        a1 = np.where(lu1 > hi1 - z1, 1, np.where(ll1 > z1 - lo1, -1, 0)); a1[:nx] = 0
        line = f"step {step} inst {b} (gpu: rounds {d[f'qp_stats_{step}'][b, 1, 5]}, path {d[f'qp_stats_{step}'][b, 1, 7]}):"
        for name, theta, inf in variants:
            r, ok, hist = run(E, e, q, lo2, hi2, a1, theta, inf)
            tot[name].append(r if ok else 99)
            line += f" | {name}: {r if ok else 'F' + str(r)}"
        print(line, flush=True)
for k, v in tot.items():
    v = np.array(v)
    print(f"{k:28s} mean rounds (successes) {v[v < 99].mean() if (v < 99).any() else float('nan'):.2f}  fails {np.sum(v == 99)}/{len(v)}")
