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


def _run_slice(model, N, seeds, steps, x0, device, noise, solve_nominal, continuation=1, budget_ms=None):
    B = len(seeds)
    W = np.stack([disturbance_stream(s, steps, model.nx) for s in seeds], axis=1) if noise else None   # (steps, B, nx)
    cl = ClosedLoopMPC(model, N, B, device=device)
    X0 = np.tile(np.asarray(x0, dtype=float), (B, 1))
    if budget_ms != 0 and cl.rti == 1 and model.fast_sls_rti_steps == 1:      # instances advance independently (slsqp_cl_run): same bits
        out = cl.run_decoupled(X0, steps, W, solve_nominal=solve_nominal, continuation=continuation, budget_ms=8.0 if budget_ms is None else budget_ms)
    else:
        out = cl.run_on_device(X0, steps, W, solve_nominal=solve_nominal, continuation=continuation)
    if cl.nlp_status is not None:
        out.update(nlp_status=cl.nlp_status, nlp_iterations=cl.nlp_iterations)
    cl.close()
    return out


def run_monte_carlo(model, N, seeds, steps, x0, rank=0, world=1, device=0, noise=True, gather=True, solve_nominal=False, slices=1, continuation=1,
                    budget_ms=None):
    """The rocket script's setting (rti = 1, one fast-SLS step) runs every slice's loop through slsqp_cl_run -- by default ONE persistent launch per slice in
    which no instance waits for another (budget_ms only matters for the round-based variant, ClosedLoopMPC.f.opts.cl_persistent = 0); budget_ms = 0 runs
    one slsqp_cl_step per step for the whole slice instead.  The results are the same bit for bit either way.
    slices > 1: the rank's seeds are cut into that many independent slices, each with its own handle (HIP stream) and host thread
    (as in fast_sls.SlicedDeviceBatch): results are bit-identical, the slices' solver tails overlap each other's bulk launches."""
    import threading
    seeds = np.asarray(seeds)
    S = len(seeds)
    lo, hi = shard_range(S, rank, world)
    mine = seeds[lo:hi]
    B = len(mine)
    K = max(1, min(int(slices), B))
    cuts = [(B * k // K, B * (k + 1) // K) for k in range(K)]
    parts, err = [None] * K, []

    def work(k):
        try:
            parts[k] = _run_slice(model, N, mine[cuts[k][0]:cuts[k][1]], steps, x0, device, noise, solve_nominal, continuation, budget_ms)
        except Exception as e:
            err.append(e)

    if K == 1:
        work(0)
    else:
        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    if err:
        raise err[0]
    out = {}
    for key, v in parts[0].items():
        if key in ("rounds", "loop_stats"):
            out[key] = [p[key] for p in parts]
        elif isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == cuts[0][1] - cuts[0][0] and key not in ("t_jac", "t_qp", "t_riccati"):
            out[key] = np.concatenate([p[key] for p in parts], axis=0)
        elif key in ("t_qp", "t_riccati", "t_jac"):
            out[key] = np.max(np.stack([p[key] for p in parts]), axis=0)      # slices run concurrently
        else:
            out[key] = v
    res = dict(seeds=mine, **out)
    if gather and world > 1:
        import torch
        dev = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
        for k in ("state_trajectory", "input_trajectory"):
            t = torch.from_numpy(np.ascontiguousarray(out[k].reshape(B, -1))).to(dev)
            full = gather_rows(t, S, world).cpu().numpy()
            res[k + "_all"] = full.reshape((S,) + out[k].shape[1:])
    return res
