"""Closed-loop Monte-Carlo over disturbance seeds, sharded across ranks (BASELINE.json config 5).

Seed s reproduces the disturbance stream of the reference script for `np.random.seed(s)`
(expe/main_rocket_robust_closed_loop.py:30,180: w_t = 2*rand(nx) - 1 per closed-loop step; seed 0 is the script's own run).
Rank r owns the contiguous seed slice shard_range(S, r, world); the only collective is one all-gather of the trajectories.
"""
import numpy as np

from .closed_loop import ClosedLoopMPC
from .sharding import gather_rows, shard_range


def disturbance_stream(seed, steps, nx):
    rs = np.random.RandomState(int(seed))
    return np.stack([2.0 * rs.rand(nx) - 1.0 for _ in range(steps)])


def run_monte_carlo(model, N, seeds, steps, x0, rank=0, world=1, device=0, noise=True, gather=True, solve_nominal=False):
    seeds = np.asarray(seeds)
    S = len(seeds)
    lo, hi = shard_range(S, rank, world)
    mine = seeds[lo:hi]
    B = len(mine)
    W = np.stack([disturbance_stream(s, steps, model.nx) for s in mine], axis=1) if noise else None   # (steps, B, nx)
    cl = ClosedLoopMPC(model, N, B, device=device)
    out = cl.run(np.tile(np.asarray(x0, dtype=float), (B, 1)), steps, W, solve_nominal=solve_nominal)
    nlp = None if cl.nlp_status is None else dict(nlp_status=cl.nlp_status, nlp_iterations=cl.nlp_iterations)
    cl.close()
    res = dict(seeds=mine, **out)
    if nlp:
        res.update(nlp)
    if gather and world > 1:
        import torch
        dev = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
        for k in ("state_trajectory", "input_trajectory"):
            t = torch.from_numpy(np.ascontiguousarray(out[k].reshape(B, -1))).to(dev)
            full = gather_rows(t, S, world).cpu().numpy()
            res[k + "_all"] = full.reshape((S,) + out[k].shape[1:])
    return res
