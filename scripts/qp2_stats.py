import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from robust_nonlinear_mpc_amd import BatchedFastSLS, make_batch
from robust_nonlinear_mpc_amd.fast_sls import DeviceBatch
B = 4096
batch = make_batch("rocket", os.path.join(ROOT, "tests", "golden", "sweep_rocket_N20_s0.npz"), B, seed=1234)
m, N = batch["model"], batch["N"]
f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=B)
f.set_rti_steps(1); f.opts.warm_start = 0
f.opts.warm_rounds = int(os.environ.get("QP_WARM_ROUNDS", "4"))
dev = DeviceBatch(f, batch)
dev.step(); dev.step()
kk = f.get("kkt", (8,)); it = f.get("qp_iters", (), np.int32)
ticks = kk[:, 7].astype(int); ft = kk[:, 6].astype(int)
print("QP#2 ticks histogram:", dict(zip(*np.unique(ticks, return_counts=True))))
print("QP#2 factor sweeps histogram:", dict(zip(*np.unique(ft, return_counts=True))))
print("ipm iters (fallback only):", dict(zip(*np.unique(it[it > 0], return_counts=True))))
pv = f.get("primal_vec", (f.n,)); bo = f.get("backoff", (N, m.ni))
print("max backoff", bo.max(), "mean nact (|dual|>1e-9 ineq)", (np.abs(f.get("dual_vec", (f.mb,))) > 1e-9).sum(axis=1).mean() - N * m.nx)
