"""Dump the QPs a rocket closed loop actually solves (for CPU prototyping of active-set strategies): for consecutive closed-loop steps the problem
data of the step's last linearisation, the tightened bounds of QP #2, its solution and the per-QP statistics.  Writes gpurun_out/cl_qps.npz."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model
m = get_model("rocket")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
only_unsolved = len(sys.argv) > 4 and sys.argv[4] == "unsolved"      # keep only the instances one of whose QPs ended unsolved (status 1 / 3) in that step, plus two solved ones
N = 20
x0 = m.x_ref + scale * (m.extra["x0"] - m.x_ref)
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
cl = ClosedLoopMPC(m, N, B)
cl.reset(np.tile(x0, (B, 1)), solve_nominal=True, continuation=2 if scale > 0.6 else 1)
f = cl.f
nz = m.nx + m.nu
n, mb = nz * N + m.nx, N * (m.nx + m.ni) + m.ni_f
out = {}
for i in range(steps):
    cl.step(W[i], fetch=False)
    qs_now = f.get("qp_stats", (2, 8), np.int32)
    sel = np.arange(B)
    if only_unsolved:
        bad = np.flatnonzero(np.isin(qs_now[:, 0, 6], (1, 3)) | np.isin(qs_now[:, 1, 6], (1, 3)))
        sel = np.concatenate([bad[:12], np.flatnonzero((qs_now[:, 0, 6] == 0) & (qs_now[:, 1, 6] == 0))[:2]])
    out[f"sel_{i}"] = sel
    for k, shp in (("A", (N, m.nx, m.nx)), ("Bm", (N, m.nx, m.nu)), ("c", (N, m.nx)), ("g", (N, m.ni)), ("gN", (m.ni_f,)), ("q", (n,)), ("x0_arg", (m.nx,)),
                   ("ubg", (mb,)), ("primal_vec", (n,)), ("dual_vec", (mb,)), ("backoff", (N, m.ni)), ("backoff_f", (m.ni_f,))):
        out[f"{k}_{i}"] = f.get(k, shp)[sel]
    out[f"qp_stats_{i}"] = qs_now[sel]
    out[f"success_{i}"] = f.get("scp_success", (), np.int32)[sel]
    qs = qs_now
    pf = lambda k: "/".join(f"{np.mean(qs[:, k, 7] == v):.2f}" for v in range(4))
    print(f"step {i}: success {out[f'success_{i}'].mean():.3f} | QP1 its>0 {np.mean(qs[:,0,0]>0):.3f} ticks {qs[:,0,1].mean():.1f} (max {qs[:,0,1].max()}) rounds {qs[:,0,5].mean():.1f} nact {qs[:,0,3].mean():.1f} warm {qs[:,0,4].mean():.2f} path {pf(0)}"
          f" | QP2 its>0 {np.mean(qs[:,1,0]>0):.3f} ticks {qs[:,1,1].mean():.1f} (max {qs[:,1,1].max()}) rounds {qs[:,1,5].mean():.1f} nact {qs[:,1,3].mean():.1f} path {pf(1)} | status(-1..4) {np.bincount(qs[:,1,6] + 1, minlength=6)}", flush=True)
cl.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
name = "cl_qps_unsolved.npz" if only_unsolved else "cl_qps.npz"
np.savez_compressed(os.path.join(ROOT, "gpurun_out", name), **out)
print("wrote", name)
