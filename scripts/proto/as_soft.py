"""Prototype: active-set iteration where 'active' means a stiff quadratic penalty (Pi = 1/(pd + rho)) instead of elimination (Pi = 0), so that sets
which pin both ends of a dynamics row (LICQ failure) do not make the solve singular."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities, qp_box
from as_polish import boxes, Pd, N, n, nx, nu, nz
from robust_nonlinear_mpc_amd import make_batch


def solve_soft(E, e, q, lo, hi, act, rho):
    """min 1/2 z'Pz + q'z + rho/2 sum_act (z - bound)^2  s.t. E z = e   via the normal equations Y = E Pi E'"""
    bound = np.where(act > 0, hi, np.where(act < 0, lo, 0.0))
    pd = Pd + np.where(act != 0, rho, 0.0)
    qq = q - np.where(act != 0, rho * bound, 0.0)
    pi = 1.0 / pd
    pi[:nx] = 0.0                                     # x0 pinned through its equality rows anyway
    # KKT: pd z + qq + E'nu = 0, E z = e  ->  (E Pi E') nu = -E Pi qq - e ... keep x0 rows: use full pi for generality
    pi = 1.0 / pd
    Y = (E * pi) @ E.T
    nu = np.linalg.solve(Y, -(E * pi) @ qq - e)
    z = -pi * (qq + E.T @ nu)
    lam = np.where(act != 0, rho * (z - bound), 0.0)   # multiplier estimate of the penalised bounds (sign: >0 pushes down for upper)
    return z, nu, lam, np.linalg.cond(Y)


def iterate(E, e, q, lo, hi, rho, max_rounds=12, tol=1e-9):
    act = np.zeros(n, dtype=int)
    hist = []
    qs = max(1.0, np.abs(q).max()); t = tol * qs
    for r in range(max_rounds):
        z, nu, lam, cond = solve_soft(E, e, q, lo, hi, act, rho)
        rel = ((act > 0) & (lam < -t)) | ((act < 0) & (lam > t))
        vu = np.where((act == 0) & (z > hi + t), z - hi, 0.0); vl = np.where((act == 0) & (z < lo - t), lo - z, 0.0)
        vu[:nx] = 0; vl[:nx] = 0
        v = np.maximum(vu, vl)
        hist.append((int(rel.sum()), int((v > 0).sum()), f"{cond:.0e}"))
        if rel.sum() + (v > 0).sum() == 0:
            return act, z, r, True, hist
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
        add = keep.flat[:n] & (v > 0)
        act[rel] = 0
        act[add & (vu > 0)] = 1; act[add & (vl > 0)] = -1
    return act, z, max_rounds, False, hist


bt = make_batch("rocket", os.path.join(ROOT, "tests", "golden", "sweep_rocket_N20_s0.npz"), 24, seed=1234)
for b in (18, 21, 23, 1, 12):
    A, Bm, c, g, gN, q, x0a = (bt[k][b] for k in ("A", "B", "c", "g", "gN", "q", "x0_arg"))
    E, e = build_equalities(A, Bm, c, -x0a)
    ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
    lo1, hi1 = boxes(ub1)
    z1, nu1, lu1, ll1, ok, its = qp_box(Pd, q, E, e, lo1, hi1)
    act1 = np.where(lu1 > hi1 - z1, 1, np.where(ll1 > z1 - lo1, -1, 0)); act1[:nx] = 0
    names = ['x','y','z','vx','vy','vz','qw','qx','qy','qz','wx','wy','wz','thr','tq','sa1','sa2','u_thr','u_tq','u_sa1','u_sa2']
    print(f"inst {b}: optimal active set:", [(int(i // nz), names[i % nz], int(act1[i])) for i in np.nonzero(act1)[0]])
    for rho in (1e6, 1e8, 1e10):
        a, z, r, okp, hist = iterate(E, e, q, lo1, hi1, rho)
        print(f"   rho {rho:.0e}: rounds {r} ok {okp} err vs optimum {np.abs(z - z1).max():.1e} set==opt {np.array_equal(a, act1)} hist {hist[:6]}")
