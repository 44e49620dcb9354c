#!/usr/bin/env python3
"""Closed-loop robust MPC of the rockETH model for a batch of disturbance seeds on one MI355X -- the batched counterpart of the
reference's expe/main_rocket_robust_closed_loop.py `generate()` (same weights, E, x0, rti settings, 30 steps, seed-s noise streams).

    python examples/rocket_closed_loop.py --seeds 64 --steps 30 [--N 15] [--out results/]

The reference starts from an IPOPT nominal trajectory (SCP_SLS.solve_nominal_trajectory); here `--init sqp` (default) solves the same
nominal NLP on the GPU (slsqp_nominal_solve, trust-region SCP from a hover roll-out) and `--init rollout` uses the bare roll-out.
`--x0-scale s` starts from x_ref + s (x0_script - x_ref); the script's own x0 (s = 1, the default) needs the initial-state continuation
(`--continuation 2`, default).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_nonlinear_mpc_amd import get_model, run_monte_carlo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--N", type=int, default=15)          # the script's default horizon (main_rocket...:63)
    ap.add_argument("--x0-scale", type=float, default=1.0)
    ap.add_argument("--init", default="sqp", choices=["sqp", "rollout"])
    ap.add_argument("--continuation", type=int, default=2, help="stages of the initial-state continuation of the nominal NLP (far-away x0)")
    ap.add_argument("--slices", type=int, default=3, help="independent slices (own stream + host thread) the seeds are cut into")
    ap.add_argument("--round-budget-ms", type=float, default=None, help="run the loop through slsqp_cl_run: instances advance independently, rounds of this length")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    m = get_model("rocket")
    x0 = m.x_ref + a.x0_scale * (m.extra["x0"] - m.x_ref)
    t0 = time.perf_counter()
    r = run_monte_carlo(m, a.N, np.arange(a.seeds), a.steps, x0, solve_nominal=(a.init == "sqp"), slices=a.slices, continuation=a.continuation,
                        budget_ms=a.round_budget_ms)
    dt = time.perf_counter() - t0
    ok = r["success"]
    if "nlp_status" in r:
        print(f"nominal NLP: status counts {np.bincount(r['nlp_status'], minlength=3).tolist()} (0 KKT point, 1 max QPs, 2 failed); "
              f"accepted steps mean {r['nlp_iterations'].mean():.1f}")
    print(f"{a.seeds} seeds x {a.steps} MPC steps (N={a.N}) in {dt:.2f} s; solved steps: {ok.mean():.3f}; "
          f"final |pos| mean {np.linalg.norm(r['state_trajectory'][:, :3, -1], axis=1).mean():.3f} "
          f"(start {np.linalg.norm(x0[:3]):.3f}); QP {r['t_qp'].sum():.1f} ms, Riccati sweeps {r['t_riccati'].sum():.1f} ms")
    if a.out:
        os.makedirs(a.out, exist_ok=True)
        from robust_nonlinear_mpc_amd import ClosedLoopMPC
        cl = ClosedLoopMPC(m, a.N, 1)
        cl.save_npz(os.path.join(a.out, "rockETH_robust_closed_loop_seed0.npz"), r, 0)
        cl.close()


if __name__ == "__main__":
    main()
