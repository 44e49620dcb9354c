#!/usr/bin/env python3
"""Batched counterpart of the reference's three closed-loop scripts (expe/main_pendulum_robust_closed_loop.py,
main_quadrotor_robust_closed_loop.py, main_rocket_robust_closed_loop.py) on one MI355X: same weights, E, regularisers, rti /
fast_sls_rti_steps, step counts and (rocket) seed-s disturbance streams; B independent runs at once.

    python examples/closed_loop.py --model pendulum  [--runs 256]          # x0 = [0.5, 0.5, 0, 0], 60 steps, no noise (main_pendulum...:27-60,96)
    python examples/closed_loop.py --model quadrotor [--runs 256]          # random x0 around hover (the script's x0 is unseeded), 30 steps
    python examples/closed_loop.py --model rocket    [--runs 256] [--x0-scale 0.3]

The first nominal comes from the GPU initialiser (slsqp_nominal_solve) in place of IPOPT."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="pendulum", choices=["pendulum", "quadrotor", "rocket"])
    ap.add_argument("--runs", type=int, default=256)
    ap.add_argument("--N", type=int, default=None)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--x0-scale", type=float, default=0.3, help="rocket: x0 = x_ref + s (x0_script - x_ref); quadrotor: spread around hover")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    m = get_model(a.model)
    N = a.N or 15                                                            # the scripts' default horizon (main_*_robust_closed_loop.py: N = 15)
    steps = a.steps or m.extra.get("sim_steps", 30)
    B = a.runs
    rng = np.random.default_rng(0)
    if a.model == "pendulum":
        x0 = np.tile(m.extra["x0"], (B, 1)) * (1.0 + 0.2 * rng.uniform(-1, 1, (B, 1)))
        x0[0] = m.extra["x0"]                                                # run 0 is the script's own
        W = None
    elif a.model == "quadrotor":
        D = np.array([2.0] * 3 + [1.0] * 3 + [0.0, 0.1, 0.1, 0.1] + [0.5] * 3)
        x0 = m.x_ref + a.x0_scale * D * rng.uniform(-1, 1, (B, m.nx))
        x0[:, 6:10] /= np.linalg.norm(x0[:, 6:10], axis=1, keepdims=True)
        W = None
    else:
        x0 = np.tile(m.x_ref + a.x0_scale * (m.extra["x0"] - m.x_ref), (B, 1))
        W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)   # seed 0 = the script's stream
    cl = ClosedLoopMPC(m, N, B)
    t0 = time.perf_counter()
    out = cl.run_on_device(x0, steps, W, solve_nominal=True, continuation=2 if a.model == "rocket" else 1)
    dt = time.perf_counter() - t0
    dist0 = np.linalg.norm(out["state_trajectory"][:, :, 0] - m.x_ref, axis=1).mean()
    dist1 = np.linalg.norm(out["state_trajectory"][:, :, -1] - m.x_ref, axis=1).mean()
    print(f"{a.model}: {B} runs x {steps} MPC steps (N={N}, rti={cl.rti}) in {dt:.2f} s; nominal NLP solved for {np.mean(cl.nlp_status == 0):.3f}; "
          f"MPC steps solved {out['success'].mean():.3f}; mean |x - x_ref| {dist0:.3f} -> {dist1:.3f}; QP {out['t_qp'].sum():.0f} ms, sweeps {out['t_riccati'].sum():.0f} ms")
    if a.out:
        os.makedirs(a.out, exist_ok=True)
        cl.save_npz(os.path.join(a.out, f"{a.model}_robust_closed_loop_run0.npz"), out, 0)
    cl.close()


if __name__ == "__main__":
    main()
