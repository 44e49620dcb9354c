"""Whole closed-loop MPC steps per second (BASELINE config 5 shape on one GPU): linearise -> fast-SLS (2 QPs + sweep) -> nominal update ->
warm-start shift -> plant + noise, records kept on the device; 4096 rocket runs (disturbance seeds), 3 free-running slices."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import get_model, run_monte_carlo
m = get_model("rocket")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
x0 = m.x_ref + 0.3 * (m.extra["x0"] - m.x_ref)
for slices in (1, 3):
    run_monte_carlo(m, 20, np.arange(64), 2, x0, slices=1, solve_nominal=True)      # warm the library
    t0 = time.perf_counter()
    r = run_monte_carlo(m, 20, np.arange(S), steps, x0, slices=slices, solve_nominal=True)
    dt = time.perf_counter() - t0
    gpu = r["t_qp"].sum() + r["t_riccati"].sum()
    print(f"slices {slices}: {S} runs x {steps} closed-loop MPC steps (N=20) in {dt:.2f} s wall incl. nominal initialiser, set-up and read-back "
          f"-> {S*steps/dt/1e3:.1f} k MPC steps/s; solved {r['success'].mean():.3f}; QP+sweep GPU time per step {gpu/steps:.1f} ms")
