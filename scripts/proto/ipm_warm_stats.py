import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "proto"))
from ref_ipm import build_equalities
from as_polish import boxes, Pd, N, n
from ipm_warm import ipm
d = dict(np.load(os.path.join(ROOT, "gpurun_out", "cl_qps.npz")))
rows = []
for step in (0, 1, 2, 3):
    for b in range(0, 128, 16):
        A, Bm, c, g, gN, q, x0a, ub2 = (d[f"{k}_{step}"][b] for k in ("A", "Bm", "c", "g", "gN", "q", "x0_arg", "ubg"))
        E, e = build_equalities(A, Bm, c, -x0a)
        ub1 = np.concatenate([np.concatenate([-c[k] + 1e-10, g[k] + 1e-10]) for k in range(N)] + [gN + 1e-10])
        lo1, hi1 = boxes(ub1); lo2, hi2 = boxes(ub2)
        h1 = []
        z1, it1, ok1 = ipm(Pd, q, E, e, lo1, hi1, hist=h1)
        z2, it2, ok2 = ipm(Pd, q, E, e, lo2, hi2)
        fu, fl = hi2 < 1e19, lo2 > -1e19
        qs = max(1.0, np.abs(q).max())
        out = {}
        for j in range(1, len(h1)):
            z, nu, su, sl, lu, ll, mu, r = h1[j]
            smin = max(np.sqrt(mu), 1e-3)
            su2 = np.where(fu, np.maximum(hi2 - z, np.minimum(su, smin)), 1.0); sl2 = np.where(fl, np.maximum(z - lo2, np.minimum(sl, smin)), 1.0)
            zz, itw, okw = ipm(Pd, q, E, e, lo2, hi2, start=(z, nu, su2, sl2, lu, ll))
            out[j] = (itw if okw else 99, mu / qs)
        rows.append((step, b, it1, it2, qs, out))
        print(step, b, "qp1", it1, "qp2 cold", it2, "qscale %.1f" % qs, {j: (v[0], "%.0e" % v[1]) for j, v in out.items()}, flush=True)
