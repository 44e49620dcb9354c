"""CPU prototype: how many Mehrotra iterations does QP #2 (tightened bounds) need when started from an iterate of QP #1's central path?"""
import sys, os
import numpy as np
import scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities
from as_polish import boxes, Pd, m, nx, nu, N, nz, SR, n


def ipm(Pd, q, E, e, lo, hi, start=None, tol=1e-6, max_it=60, hist=None, init_s=1.0):
    fu, fl = hi < 1e19, lo > -1e19
    mm = max(1, fu.sum() + fl.sum())
    scale = max(1.0, np.abs(q).max())
    if start is None:
        K = np.block([[np.diag(Pd), E.T], [E, np.zeros((E.shape[0], E.shape[0]))]])
        sol = np.linalg.solve(K + 1e-13 * np.eye(K.shape[0]), np.concatenate([-q, e]))
        z, nu = sol[:n], sol[n:]
        su = np.where(fu, np.maximum(hi - z, init_s), 1.0); sl = np.where(fl, np.maximum(z - lo, init_s), 1.0)
        lam0 = max(1.0, 0.1 * np.abs(q).max())
        lu, ll = np.where(fu, lam0, 0.0), np.where(fl, lam0, 0.0)
    else:
        z, nu, su, sl, lu, ll = (v.copy() for v in start)

    def steplen(ds, s, mask):
        idx = mask & (ds < 0)
        return min(1.0, (-s[idx] / ds[idx]).min()) if idx.any() else 1.0
    for it in range(max_it):
        rd = Pd * z + q + E.T @ nu + lu - ll
        rp = E @ z - e
        ru, rl = np.where(fu, z + su - hi, 0.0), np.where(fl, lo - z + sl, 0.0)
        mu = ((su * lu)[fu].sum() + (sl * ll)[fl].sum()) / mm
        res = max(np.abs(rd).max(), np.abs(rp).max(), np.abs(ru).max(), np.abs(rl).max())
        if hist is not None:
            hist.append((z.copy(), nu.copy(), su.copy(), sl.copy(), lu.copy(), ll.copy(), mu, res))
        if res < tol * scale and mu < tol * scale:
            return z, it, True
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
        muaff = (((su + a * dsu) * (lu + a * dlu))[fu].sum() + ((sl + a * dsl) * (ll + a * dll))[fl].sum()) / mm
        sig = (muaff / mu) ** 3 if mu > 0 else 0.0
        dz, dnu, dsu, dsl, dlu, dll = newton(sig * mu - dsu * dlu, sig * mu - dsl * dll)
        a = min(1.0, 0.99 * min(steplen(dsu, su, fu), steplen(dsl, sl, fl), steplen(dlu, lu, fu), steplen(dll, ll, fl)))
        z, nu, su, sl, lu, ll = z + a * dz, nu + a * dnu, su + a * dsu, sl + a * dsl, lu + a * dlu, ll + a * dll
    return z, max_it, False


if __name__ == "__main__":
    d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for b in range(nb):
        A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
        E, e = build_equalities(A, Bm, c, -x0a)
        ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
        lo1, hi1 = boxes(ub1); lo2, hi2 = boxes(ub2)
        h1 = []
        z1, it1, ok1 = ipm(Pd, q, E, e, lo1, hi1, hist=h1)
        z2, it2, ok2 = ipm(Pd, q, E, e, lo2, hi2)
        fu, fl = hi2 < 1e19, lo2 > -1e19
        res = []
        for j in range(1, len(h1)):
            z, nu, su, sl, lu, ll, mu, r = h1[j]
            for mode in ("keep", "center"):
                if mode == "keep":       # keep multipliers, push slacks inside
                    smin = max(np.sqrt(mu), 1e-3) if mu > 0 else 1e-3
                    su2 = np.where(fu, np.maximum(hi2 - z, np.minimum(su, smin)), 1.0); sl2 = np.where(fl, np.maximum(z - lo2, np.minimum(sl, smin)), 1.0)
                    lu2, ll2 = lu, ll
                else:                    # re-centre: s = max(hi - z, smin), lambda = mu / s
                    smin = np.sqrt(max(mu, 1e-8))
                    su2 = np.where(fu, np.maximum(hi2 - z, smin), 1.0); sl2 = np.where(fl, np.maximum(z - lo2, smin), 1.0)
                    lu2, ll2 = np.where(fu, mu / su2, 0.0), np.where(fl, mu / sl2, 0.0)
                zz, itw, okw = ipm(Pd, q, E, e, lo2, hi2, start=(z, nu, su2, sl2, lu2, ll2))
                res.append((j, mode, f"{mu:.1e}", itw, okw, f"{np.abs(zz - z2).max():.0e}"))
        print(f"inst {b}: QP1 its {it1}, QP2 cold its {it2}; warm from iterate j (mode, mu_j, its, ok, err):")
        for r in res:
            print("    ", r)
