"""Persistent closed-loop launch (k_cl_loop) on the bench's default workload: launch time, mean time a wave spends on an MPC step, how busy the
instance queue keeps the waves -- for a list of wave counts (SLSQP_LOOP_WAVES).  usage: loop_probe.py [B] [steps] [waves,waves,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
waves = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
m = get_model("rocket")
N = 20
x0 = np.tile(m.extra["x0"], (B, 1))
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
cl = ClosedLoopMPC(m, N, B)
cl.f.opts.time_kernels = 1
for w in waves:
    if w > 0:
        os.environ["SLSQP_LOOP_WAVES"] = str(w)
    else:
        os.environ.pop("SLSQP_LOOP_WAVES", None)
    t0 = time.perf_counter()
    out = cl.run_decoupled(x0, steps, W, solve_nominal=True, continuation=2)
    wall = time.perf_counter() - t0
    ls = out["loop_stats"]
    qs = out["qp_stats"]
    ticks = qs[:, :, :, 0].sum() if qs.ndim == 4 else 0
    ct = cl.f.get("chain_times", (4,), np.uint64).astype(np.float64) * 1e-5      # last step of every instance: QP #1, sweep part, QP #2, whole chain (ms)
    ran = ct[:, 3] > 0
    print(f"   chain of the last step (instances that ran it: {ran.mean():.2f}): mean qp1 {ct[ran, 0].mean():.3f} sweep {ct[ran, 1].mean():.3f} qp2 {ct[ran, 2].mean():.3f} total {ct[ran, 3].mean():.3f} ms; p50 {np.median(ct[ran, 3]):.3f} max {ct[ran, 3].max():.3f}")
    import ctypes as C
    st = (C.c_double * 16)()
    cl.f.lib.slsqp_cl_run_stats(cl.f.h, st, 16)
    if sum(st[4:11]) > 0:
        names = ["shift+reset", "linearise", "x0/solve_begin", "scp_update", "infeas", "log", "plant"]
        print("   parts per MPC step (ms): " + ", ".join(f"{nm} {st[4 + i] / max(1, ls['mpc_steps']):.3f}" for i, nm in enumerate(names)))
        print(f"   waves that ran {int(st[15])}; first wave start -> last wave exit {st[14]:.2f} ms; mean wave life time {st[13] / max(1.0, st[15]):.2f} ms; per wave: in steps {ls['busy_ms'] / max(1.0, st[15]):.2f}, "
              f"pop + acquire {st[11] / max(1.0, st[15]):.2f}, release + push {st[12] / max(1.0, st[15]):.2f} ms")
    print(f"waves {ls['waves']:5d}: launch {ls['launch_ms']:8.2f} ms = {ls['launch_ms'] / steps:6.2f} ms/step | per MPC step in a wave {ls['busy_ms'] / max(1, ls['mpc_steps']):6.3f} ms | "
          f"waves busy {ls['busy_ms'] / (ls['waves'] * ls['launch_ms']):5.3f} | success {out['success'].mean():.4f} (wall incl. nominal initialiser {wall:.2f} s)", flush=True)
cl.close()
