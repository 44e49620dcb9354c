"""Diagnostic: build the library with -DNE_STAMP (s_memtime stamps around the segments of the factor forward sweep),
run the interior point's first factorisation, print where a stage's cycles go.  The stamps land in the kkt debug array,
never in an output; this build is never shipped or timed."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "robust-nonlinear-mpc_amd", "csrc")
so = "/tmp/libslsqp_stamp_bwd.so"
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DNE_STAMP", "-DNE_STAMP_BWD", "-Wno-unused-result",
                       "-Wno-unused-value", "-Wno-pass-failed", "-o", so, os.path.join(csrc, "slsqp_api.hip")])
import torch, numpy as np  # noqa
from robust_nonlinear_mpc_amd import _lib
_lib.SO_PATH = so
from robust_nonlinear_mpc_amd import BatchedFastSLS, make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = make_batch("rocket", os.path.join(ROOT, "tests", "golden", "sweep_rocket_N20_s0.npz"), B, seed=1)
m, N = batch["model"], batch["N"]
f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=B)
f.opts.qp_max_iter = 0   # init solve only, then max-iter exit: the only k_ne_fwd launch that does work is the factor sweep of P_INIT
f.opts.warm_start = 0
f.update_dynamics_list(batch["A"], batch["B"], batch["E"], batch["g"], batch["gN"], batch["c"])
f.update_linear_cost(batch["q"])
x0 = batch["x0_arg"]
ub = f.get("ubg", (f.mb,)); lb = f.get("lbg", (f.mb,))
l = np.concatenate([lb, -x0 - 1e-10], axis=1); u = np.concatenate([ub, -x0 + 1e-10], axis=1)
f.qp_update_data_vec(batch["q"], l, u)
x, y, st, it, t = f.qp_solve()
k = f.get("kkt", (8,)).view(np.int64)
names = ["stage->LDS", "prefetch issue", "Dinv matvec", "A,B matvecs", "bwd total", "phase_update", "-", "-"]
for b in range(min(B, 1)):
    tot = k[b][:4].sum()
    print("B", B, "instance", b, "total cycles", tot, "per stage", tot // N, "status", st[b])
    for i in range(6):
        print("  %-14s %9d  %5.1f%%  per stage %7d" % (names[i], k[b][i], 100 * k[b][i] / tot, k[b][i] // N))
if B > 1:
    tot = k[:, :7].sum(axis=1)
    print("mean total over batch", tot.mean(), "share per segment", (k[:, :7].sum(axis=0) / tot.sum()).round(3))
