"""CPU prototype (numpy, dense) of the GPU's active-set polish on QPs dumped from a closed loop (scripts/dump_cl_qps.py): why does the warm
polish of QP #2 fall back, and which correction rule fixes it?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_ipm import build_equalities, qp_box
from robust_nonlinear_mpc_amd import get_model

m = get_model("rocket")
nx, nu, N = 17, 4, 20
nz, ni, nif = nx + nu, 42, 34
SR = nx + ni
n = nz * N + nx
Pd = 2.0 * np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])


def boxes(ub):
    hi, lo = np.full(n, 1e20), np.full(n, -1e20)
    for k in range(N):
        hi[k * nz:(k + 1) * nz] = ub[k * SR + nx:k * SR + nx + nz]
        lo[k * nz:(k + 1) * nz] = -ub[k * SR + nx + nz:k * SR + nx + 2 * nz]
    hi[N * nz:], lo[N * nz:] = ub[N * SR:N * SR + nx], -ub[N * SR + nx:N * SR + 2 * nx]
    hi[:nx], lo[:nx] = 1e20, -1e20
    return lo, hi


def as_solve(E, e, q, lo, hi, act):
    """equality-constrained solve with the variables in act fixed at their bounds; returns z, gr = P z + q + E' nu"""
    z = np.zeros(n)
    fixed = act != 0
    fixed[:nx] = False
    z[act > 0] = hi[act > 0]; z[act < 0] = lo[act < 0]
    fr = ~fixed
    Ef = E[:, fr]
    K = np.block([[np.diag(Pd[fr]), Ef.T], [Ef, np.zeros((E.shape[0], E.shape[0]))]])
    rhs = np.concatenate([-q[fr], e - E[:, fixed] @ z[fixed]])
    try:
        sol = np.linalg.solve(K, rhs)
    except np.linalg.LinAlgError:
        sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
    z[fr] = sol[:fr.sum()]
    nu = sol[fr.sum():]
    gr = Pd * z + q + E.T @ nu
    return z, nu, gr, np.linalg.cond(K)


def polish(E, e, q, lo, hi, act0, rule="all", max_rounds=20, tol=1e-9, verbose=False):
    act = act0.copy()
    seen = {}
    for r in range(max_rounds):
        z, nu, gr, cond = as_solve(E, e, q, lo, hi, act)
        qs = max(1.0, np.abs(q).max())
        t = tol * qs
        rel_u = (act > 0) & (gr > t)         # lambda_u = -gr < 0
        rel_l = (act < 0) & (-gr > t)
        add_u = (act == 0) & (z > hi + t)
        add_l = (act == 0) & (z < lo - t)
        add_u[:nx] = False; add_l[:nx] = False
        nch = rel_u.sum() + rel_l.sum() + add_u.sum() + add_l.sum()
        if verbose:
            print(f"   round {r}: |A|={np.sum(act!=0)} release {rel_u.sum()+rel_l.sum()} add {add_u.sum()+add_l.sum()} cond {cond:.1e} maxviol {max((z-hi).max(), (lo-z).max()):.2e}")
        if nch == 0:
            return act, z, r, True
        key = act.tobytes()
        cyc = key in seen
        seen[key] = r
        if rule == "all" and not cyc or rule == "all_nocyc":
            act[rel_u | rel_l] = 0; act[add_u] = 1; act[add_l] = -1
        else:   # single exchange: most violated bound in, else most negative multiplier out
            viol = np.where(add_u, z - hi, 0) + np.where(add_l, lo - z, 0)
            if viol.max() > 0:
                j = int(np.argmax(viol)); act[j] = 1 if add_u[j] else -1
            else:
                w = np.where(rel_u, gr, 0) + np.where(rel_l, -gr, 0)
                j = int(np.argmax(w)); act[j] = 0
    return act, z, max_rounds, False


if __name__ == "__main__":
    d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for b in range(nb):
        A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
        E, e = build_equalities(A, Bm, c, -x0a)
        # QP 1: un-tightened bounds (+eps)
        ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
        lo1, hi1 = boxes(ub1)
        z1, nu1, lu1, ll1, ok, its = qp_box(Pd, q, E, e, lo1, hi1)
        act1 = np.where(lu1 > hi1 - z1, 1, np.where(ll1 > z1 - lo1, -1, 0)); act1[:nx] = 0
        lo2, hi2 = boxes(ub2)
        z2, nu2, lu2, ll2, ok2, its2 = qp_box(Pd, q, E, e, lo2, hi2)
        act2 = np.where(lu2 > hi2 - z2, 1, np.where(ll2 > z2 - lo2, -1, 0)); act2[:nx] = 0
        print(f"inst {b}: QP1 ipm its {its} |A1|={np.sum(act1!=0)}; QP2 its {its2} |A2|={np.sum(act2!=0)}; A1->A2 changes {np.sum(act1!=act2)}; gpu primal err {np.abs(z2-d[f'primal_vec_{step}'][b]).max():.1e}")
        for rule in ("all_nocyc", "all", "single"):
            a, z, r, okp = polish(E, e, q, lo2, hi2, act1, rule=rule, verbose=(b < 2))
            print(f"  rule {rule}: rounds {r} ok {okp} final == A2 {np.array_equal(a, act2)} err {np.abs(z-z2).max():.1e}")
