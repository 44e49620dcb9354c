"""A second, independent CPU solver for the path's QP (test infrastructure): dense Mehrotra predictor-corrector interior point in numpy on
    min 1/2 z' diag(Pd) z + q' z   s.t.  E z = e,  lo <= z <= hi.
Nothing in common with the oracle's OSQP-class ADMM restatement (oracle/sls_oracle.c) nor with the GPU's block-tridiagonal solver beyond the
problem data, so agreement between them is evidence that all of them reach the QP's unique optimum (QP parity is otherwise "unpinned": no
OSQP output exists in the reference)."""
import numpy as np
import scipy.linalg as sla


def build_equalities(A, B, c, x0val):
    """Rows x_0 = x0val, then A_k x_k + B_k u_k - x_{k+1} = -c_k, in the stage-ordered variable layout of qp_jit.py:77-192."""
    N, nx, _ = A.shape
    nu = B.shape[2]
    nz = nx + nu
    n = nz * N + nx
    E = np.zeros((nx * (N + 1), n))
    e = np.zeros(nx * (N + 1))
    E[:nx, :nx] = np.eye(nx)
    e[:nx] = x0val
    for k in range(N):
        r = nx * (k + 1)
        E[r:r + nx, k * nz:k * nz + nx] = A[k]
        E[r:r + nx, k * nz + nx:(k + 1) * nz] = B[k]
        E[r:r + nx, (k + 1) * nz:(k + 1) * nz + nx] = -np.eye(nx)
        e[r:r + nx] = -c[k]
    return E, e


def qp_box(Pd, q, E, e, lo, hi, tol=1e-12, max_it=100):
    n = len(q)
    fu, fl = hi < 1e19, lo > -1e19
    K = np.block([[np.diag(Pd), E.T], [E, np.zeros((E.shape[0], E.shape[0]))]])
    sol = np.linalg.solve(K + 1e-13 * np.eye(K.shape[0]), np.concatenate([-q, e]))
    z, nu = sol[:n], sol[n:]
    su = np.where(fu, np.maximum(hi - z, 1.0), 1.0)
    sl = np.where(fl, np.maximum(z - lo, 1.0), 1.0)
    lam0 = max(1.0, 0.1 * np.abs(q).max())
    lu, ll = np.where(fu, lam0, 0.0), np.where(fl, lam0, 0.0)
    m = max(1, fu.sum() + fl.sum())
    scale = max(1.0, np.abs(q).max())

    def steplen(ds, s, mask):
        idx = mask & (ds < 0)
        return min(1.0, (-s[idx] / ds[idx]).min()) if idx.any() else 1.0

    for it in range(max_it):
        rd = Pd * z + q + E.T @ nu + lu - ll
        rp = E @ z - e
        ru, rl = np.where(fu, z + su - hi, 0.0), np.where(fl, lo - z + sl, 0.0)
        mu = ((su * lu)[fu].sum() + (sl * ll)[fl].sum()) / m
        res = max(np.abs(rd).max(), np.abs(rp).max(), np.abs(ru).max(), np.abs(rl).max())
        if res < tol * scale and mu < tol * scale:
            return z, nu, lu, ll, True, it
        if not np.isfinite(res):
            break
        Wu, Wl = np.where(fu, lu / su, 0.0), np.where(fl, ll / sl, 0.0)
        K = np.block([[np.diag(Pd + Wu + Wl), E.T], [E, np.zeros((E.shape[0], E.shape[0]))]])
        lup = sla.lu_factor(K)

        def newton(cu, cl):
            tu = np.where(fu, (cu - su * lu) / su + Wu * ru, 0.0)
            tl = np.where(fl, (cl - sl * ll) / sl + Wl * rl, 0.0)
            s = sla.lu_solve(lup, np.concatenate([-(rd + tu - tl), -rp]))
            dz = s[:n]
            return dz, s[n:], np.where(fu, -ru - dz, 0.0), np.where(fl, -rl + dz, 0.0), np.where(fu, tu + Wu * dz, 0.0), np.where(fl, tl - Wl * dz, 0.0)

        dz, dnu, dsu, dsl, dlu, dll = newton(np.zeros(n), np.zeros(n))
        a = min(steplen(dsu, su, fu), steplen(dsl, sl, fl), steplen(dlu, lu, fu), steplen(dll, ll, fl))
        muaff = (((su + a * dsu) * (lu + a * dlu))[fu].sum() + ((sl + a * dsl) * (ll + a * dll))[fl].sum()) / m
        sig = (muaff / mu) ** 3 if mu > 0 else 0.0
        dz, dnu, dsu, dsl, dlu, dll = newton(sig * mu - dsu * dlu, sig * mu - dsl * dll)
        a = min(1.0, 0.99 * min(steplen(dsu, su, fu), steplen(dsl, sl, fl), steplen(dlu, lu, fu), steplen(dll, ll, fl)))
        z, nu, su, sl, lu, ll = z + a * dz, nu + a * dnu, su + a * dsu, sl + a * dsl, lu + a * dlu, ll + a * dll
    return z, nu, lu, ll, False, max_it


def polish(Pd, q, E, e, lo, hi, z, lu, ll, tol=1e-9, max_rounds=12):
    """Active-set refinement of an interior-point answer (what OSQP's polish does for the reference, qp_jit.py:546): the bounds the iterate
    identifies as active are fixed, the remaining equality-constrained QP is solved exactly (dense KKT system + one refinement step); multipliers
    of the wrong sign leave the set and violated bounds enter it (primal-dual active-set rounds: the iterate is next to the solution, a few
    suffice) until the result is feasible with multipliers of the right sign.  Primal and multipliers then carry rounding error only -- the
    interior point's own answer is off by up to 1e-4 in the primal near weakly active bounds although its residuals are at 1e-12 |q|inf (measured
    against the GPU's certified solutions on closed-loop rocket QPs).  Returns (z, nu, lu, ll, polished)."""
    n = len(q)
    fu, fl = hi < 1e19, lo > -1e19
    scale = max(1.0, np.abs(q).max())
    act = np.where(fu & (lu > hi - z), 1, np.where(fl & (ll > z - lo), -1, 0))
    ne = E.shape[0]
    for _ in range(max_rounds):
        aU, aL = act > 0, act < 0
        fixed = aU | aL
        fr = ~fixed
        zf = np.where(aU, hi, np.where(aL, lo, 0.0))
        Ef = E[:, fr]
        K = np.block([[np.diag(Pd[fr]), Ef.T], [Ef, np.zeros((ne, ne))]])
        rhs = np.concatenate([-q[fr], e - E[:, fixed] @ zf[fixed]])
        try:
            lup = sla.lu_factor(K)
            sol = sla.lu_solve(lup, rhs)
            sol += sla.lu_solve(lup, rhs - K @ sol)
        except Exception:
            break
        if not np.isfinite(sol).all():
            break
        zp = zf.copy(); zp[fr] = sol[:fr.sum()]
        nup = sol[fr.sum():]
        gr = Pd * zp + q + E.T @ nup                   # = -(lu - ll) on fixed elements, 0 on free ones
        lup_, llp = np.where(aU, -gr, 0.0), np.where(aL, gr, 0.0)
        rel = (aU & (lup_ < -tol * scale)) | (aL & (llp < -tol * scale))
        addu, addl = fr & fu & (zp - hi > tol), fr & fl & (lo - zp > tol)
        if not (rel.any() or addu.any() or addl.any()):
            ok = np.abs(gr[fr]).max(initial=0.0) < tol * scale and np.abs(E @ zp - e).max() < 1e-9 * max(1.0, np.abs(e).max())
            return (zp, nup, np.maximum(lup_, 0.0), np.maximum(llp, 0.0), True) if ok else (z, None, lu, ll, False)
        act = act.copy()
        act[rel] = 0; act[addu] = 1; act[addl] = -1
    return z, None, lu, ll, False
