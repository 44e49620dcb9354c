"""Hard regime (script x0, ~35 active bounds): how does the active-set iteration (inputs-first + local-maximum rule) fare from the empty set, and why
does it fail?  Prints per round (released, violated, added)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities, qp_box
from as_polish import boxes, Pd, N, n, nx, nu, nz, as_solve
d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))
names = ['x','y','z','vx','vy','vz','qw','qx','qy','qz','wx','wy','wz','thr','tq','sa1','sa2','u_thr','u_tq','u_sa1','u_sa2']


def run(E, e, q, lo, hi, act0, variant, max_rounds=40, tol=1e-9, verbose=False):
    act = act0.copy(); hist = []
    qs = max(1.0, np.abs(q).max()); t = 1e-6 * qs
    for r in range(max_rounds):
        z, nu_, gr, cond = as_solve(E, e, q, lo, hi, act)
        rel = ((act > 0) & (gr > t)) | ((act < 0) & (-gr > t))
        vu = np.where((act == 0) & (z > hi + t), z - hi, 0.0); vl = np.where((act == 0) & (z < lo - t), lo - z, 0.0)
        vu[:nx] = 0; vl[:nx] = 0
        v = np.maximum(vu, vl)
        nv = int((v > 0).sum())
        if rel.sum() + nv == 0:
            return r, True, hist
        V = np.zeros((N + 1, nz)); V.flat[:n] = v
        keep = np.zeros_like(V, dtype=bool)
        any_input = (V[:, nx:] > 0).any()
        for i in range(nz):
            col = V[:, i]
            if i >= nx:
                keep[:, i] = col > 0
            elif not (any_input and variant != "no_inputs_first"):
                for k in range(N + 1):
                    if col[k] > 0 and col[k] >= (col[k - 1] if k > 0 else 0) and col[k] >= (col[k + 1] if k < N else 0):
                        keep[k, i] = True
        add = keep.flat[:n] & (v > 0)
        hist.append((int(rel.sum()), nv, int(add.sum()), f"{np.abs(z).max():.1e}"))
        if variant == "release_one" and rel.sum() > 1:    # release only the most wrong multiplier
            w = np.where(rel, np.abs(gr), 0); rel = np.zeros_like(rel); rel[int(np.argmax(w))] = True
        act[rel] = 0
        act[add & (vu > 0)] = 1; act[add & (vl > 0)] = -1
    return max_rounds, False, hist


step = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for b in range(0, 96, 16):
    A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
    E, e = build_equalities(A, Bm, c, -x0a)
    ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
    lo1, hi1 = boxes(ub1)
    z1, nu1, lu1, ll1, ok, its = qp_box(Pd, q, E, e, lo1, hi1)
    a1 = np.where(lu1 > hi1 - z1, 1, np.where(ll1 > z1 - lo1, -1, 0)); a1[:nx] = 0
    comp = {}
    for i in np.nonzero(a1)[0]:
        comp.setdefault(names[i % nz] + ("+" if a1[i] > 0 else "-"), []).append(int(i // nz))
    print(f"inst {b} step {step}: QP1 |A|={np.sum(a1 != 0)} ipm its {its}: {comp}")
    for variant in ("default", "no_inputs_first", "release_one"):
        r, okp, hist = run(E, e, q, lo1, hi1, np.zeros(n, dtype=int), variant)
        print(f"   {variant:16s}: {'ok' if okp else 'FAIL'} after {r} rounds; (released, violated, added, |z|max) {hist[:14]}")
