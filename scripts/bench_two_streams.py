"""Experiment: split the batch over K handles (K HIP streams) driven by K host threads, so the latency-bound tails of one
half overlap the bulk of the other."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from robust_nonlinear_mpc_amd import BatchedFastSLS, make_batch
from robust_nonlinear_mpc_amd.fast_sls import DeviceBatch
B = 4096
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 3
fx = os.path.join(ROOT, "tests", "golden", "sweep_rocket_N20_s0.npz")
devs = []
for k in range(K):
    batch = make_batch("rocket", fx, B // K, seed=100 + k)
    m, N = batch["model"], batch["N"]
    f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=B // K)
    f.set_rti_steps(1); f.opts.warm_start = 0
    devs.append(DeviceBatch(f, batch))
def run(d, n):
    for _ in range(n): d.step()
for d in devs: d.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(d, steps)) for d in devs]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("K=%d handles: %.2f ms per MPC step of %d instances -> %.0f QP solves/s" % (K, 1e3 * dt / steps, B, 2 * B * steps / dt))
