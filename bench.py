#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fast-SLS QP path (BASELINE.json: "QP solves/sec (whole node) + ms/MPC-step,
rockETH N=20 batch=4096").

A "step" is one MPC step of the hot path over one batch of synthetic instances resident in HBM:
update_dynamics_list + update_linear_cost + fast_SLS.solve in the reference's closed-loop setting for the rocket
(rti=1, fast_sls_rti_steps=1: 2 QP solves + 1 SLS sweep per instance, expe/main_rocket_robust_closed_loop.py:80-85).
value = QP solves / s over all ranks (weak scaling: every rank owns its own 4096 instances; the only collective is one
RCCL all-gather of the resulting first inputs / nominal trajectories at the end).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
FIXTURE = {"rocket": "sweep_rocket_N20_s0.npz", "quadrotor": "sweep_quadrotor_N20_s0.npz", "pendulum": "sweep_pendulum_N10_s0.npz"}
QP_BYTES = {"rocket": 93656, "quadrotor": 64504, "pendulum": 6112}     # algorithmic bytes per QP solve


def cpu_baseline(batch, n_inst, budget_s=12.0):
    """Reference-class CPU path (oracle: OSQP-class ADMM + polish with upstream default settings, numba-kernel restatement for
    the sweep) on the first instances of the same workload: first one thread, then one instance per thread on all host cores
    this process may use.  The oracle's C kernels are called through ctypes, which releases the GIL."""
    import concurrent.futures as cf
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    m, N = batch["model"], batch["N"]
    d = O.dims_of(m.nx, m.nu, m.nw, N, m.ni, m.ni_f)

    def one(b):
        f = O.OracleFastSLS(d, m.G, m.Gf, m.g, m.gf, batch["E"], m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, O.default_settings())
        f.set_rti_steps(1)
        f.update_dynamics_list(batch["A"][b], batch["B"][b], batch["E"], list(batch["g"][b]) + [batch["gN"][b]], batch["c"][b])
        f.update_linear_cost(batch["q"][b])
        f.solve(batch["x0_arg"][b])
        return 2

    def run(threads, budget):
        t0 = time.perf_counter()
        done = 0
        if threads == 1:
            for b in range(n_inst):
                one(b)
                done += 1
                if time.perf_counter() - t0 > budget:
                    break
        else:
            with cf.ThreadPoolExecutor(threads) as ex:
                chunk = 4 * threads
                for lo in range(0, n_inst, chunk):
                    done += len(list(ex.map(one, range(lo, min(n_inst, lo + chunk)))))
                    if time.perf_counter() - t0 > budget:
                        break
        dt = time.perf_counter() - t0
        return 2 * done / dt, done, dt

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    v1, n1, t1 = run(1, budget_s)
    vc, nc, tc = (v1, n1, t1) if cores == 1 else run(cores, budget_s)
    return {"value": vc, "unit": "QP solves/s", "cores": cores, "kind": "port", "single_thread_value": v1,
            "sample": f"rocket-class instances x 1 RTI MPC step (2 QP + 1 sweep) each, OSQP-class restatement with upstream default settings + "
                      f"polish: {nc} instances on {cores} threads in {tc:.1f} s; {n1} instances on 1 thread in {t1:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--model", default="rocket")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--slices", type=int, default=3, help="independent slices of the rank's batch, each with its own handle / HIP stream / host thread")
    ap.add_argument("--qp-eps", type=float, default=None, help="interior-point tolerance before the polish (default: the library's 1e-6)")
    ap.add_argument("--precision", type=int, default=0, help="0: fp64 (headline); 1: mixed fp32 factorisation / fp64 residuals (secondary figure)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from robust_nonlinear_mpc_amd import BatchedFastSLS, make_batch
    from robust_nonlinear_mpc_amd.fast_sls import SlicedDeviceBatch
    fixture = os.path.join(ROOT, "tests", "golden", FIXTURE[args.model])
    B = args.batch
    batch = make_batch(args.model, fixture, B, seed=1234 + rank)
    m, N = batch["model"], batch["N"]

    def make_solver(nb):
        f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=nb, device=local_rank)
        f.set_rti_steps(m.fast_sls_rti_steps if args.model == "rocket" else 1)
        f.opts.warm_rounds = int(os.environ.get('QP_WARM_ROUNDS', '4'))
        f.opts.precision = args.precision
        if args.qp_eps is not None:
            f.opts.qp_eps = args.qp_eps
        f.opts.warm_start = 0   # QP#1 of every step is solved cold (fresh Monte-Carlo instances); QP#2 warm-starts from QP#1
        return f

    # the rank's 4096 instances as `--slices` independent slices (own handle / HIP stream / host thread each): instances are
    # independent, so one slice's few-instance solver tails overlap the other slices' bulk launches (fast_sls.SlicedDeviceBatch)
    dev = SlicedDeviceBatch(make_solver, batch, args.slices)
    f0 = dev.solvers[0]
    n_var, n_con = f0.n, f0.mb + m.nx

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def collect():
        u0 = dev.fetch_device("primal_vec", (n_var,))[:, m.nx:m.nx + m.nu].contiguous()
        if world > 1:
            src = u0 if args.backend == "nccl" else u0.cpu()
            gathered = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(gathered, src)          # RCCL over xGMI: collect the first inputs of every instance
            return gathered
        return u0

    dev.run(args.warmup)
    collect()   # warm the gather path too (first-use kernel loads are not part of the step)
    barrier()
    dev.kernel_timing()   # reset the per-kernel accumulators
    t0 = time.perf_counter()
    acc = dev.run(args.steps)
    collect()
    barrier()
    dt = time.perf_counter() - t0
    fwd_total_ms, fwd_launches, inst_sweeps, mx_retries = dev.kernel_timing()
    fact_sweeps = dev.fwd_factor_sweeps
    if world > 1:
        tmax = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st = dev.get("status", (), np.int32)
    its = dev.get("qp_iters", (), np.int32)
    qps_per_step = 2 * B            # RTI: QP#1 + QP#2 per instance
    n_sl = len(dev.slices)
    value = qps_per_step * world * args.steps / dt
    out = {
        "metric": ("QP solves/sec (whole node), rockETH N=20 batch=4096 RTI MPC step" if (args.model, B) == ("rocket", 4096)
                   else f"QP solves/sec (whole node), {args.model} N={N} batch={B} RTI MPC step"), "value": value, "unit": "QP solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == 0 else "f32 factorisation + f64 residuals (mixed)", "data": "synthetic",
        "config": {"workload": f"{args.model} N={N} batch={B}/GPU, fast-SLS RTI step (rti_steps=1: 2 QP solves + 1 SLS sweep per instance)",
                   "qp_n": n_var, "qp_m": n_con, "slices_per_gpu": n_sl, "solved_frac": float(np.mean((st == 0) | (st == 4))),
                   "polished_frac": float(np.mean(st == 0)), "ipm_iters_mean_last_qp": float(its.mean()), "ipm_iters_max_last_qp": int(its.max()),
                   "qp2_cold_fallback_frac": float(np.mean(its > 0)), "tightened_frac": float(np.mean(dev.get("backoff_x", (N + 1, m.nx)).max(axis=(1, 2)) > 0))},
    }
    if rank == 0:
        kk = dev.get("kkt", (8,))
        # dominant kernel: k_ne_fwd (block-tridiagonal forward sweep; re-factorises in half of its launches).  Algorithmic bytes of one
        # instance sweep: A_k,B_k of all stages in, rhs slices (Pi, v) in, u out; a launch moves that for every instance it works on
        # (device counter of instance sweeps / launches); time = HIP events around every launch on the launching stream.
        nz = m.nx + m.nu
        per_inst = 8 * (N * m.nx * nz + 2 * n_var + N * m.nx)
        fwd_ms = fwd_total_ms / max(1, fwd_launches)
        alg_bytes_launch = per_inst * float(inst_sweeps) / max(1, fwd_launches)
        achieved = alg_bytes_launch / (fwd_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("k_ne_fwd_bytes_per_launch")
        solves = 2 * args.steps * n_sl                          # QP solve calls in the timed region (each over one slice)
        qp_ms = sum(a["qp"] for a in acc) / solves
        sw_ms = sum(a["sweep"] for a in acc) / (args.steps * n_sl)
        out["roofline"] = {"bound": "hbm", "kernel": "k_ne_fwd", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                           "traffic": traffic, "avg_launch_ms": fwd_ms, "launches": fwd_launches, "algorithmic_bytes_per_launch": alg_bytes_launch,
                           "qp_solve": {"avg_ms": qp_ms, "instances": B / n_sl, "algorithmic_bytes": QP_BYTES[args.model] * B / n_sl,
                                        "achieved_GBps": QP_BYTES[args.model] * B / n_sl / (qp_ms * 1e-3) / 1e9,
                                        "note": "one solve call = one slice; slices run concurrently, so these times overlap"},
                           "last_qp_block_solves_per_instance": {"factorising": float(kk[:, 6].mean()), "all": float(kk[:, 7].mean())},
                           # second ceiling: fp64 work of the timed region / wall time against the vector = matrix fp64 peak.  Per instance
                           # sweep of N stages: 28.6 kflop per factorising stage, 2 kflop per forward / backward substitution stage;
                           # SLS sweep 9.0 Mflop per rocket instance (scaled with nx^3 for the other plants)
                           "fp64": {"achieved_TFLOPs": (fact_sweeps * N * 28.6e3 * (m.nx / 17.0) ** 3 + (2 * inst_sweeps - fact_sweeps) * N * 2.0e3 * (m.nx / 17.0) ** 2
                                                        + args.steps * B * 9.0e6 * (m.nx / 17.0) ** 3 * (N / 20.0) ** 2) / dt / 1e12,
                                    "peak_TFLOPs": 78.6, "note": "vector fp64 peak = matrix fp64 peak on MI355X (BASELINE.md, AMD public figure)"},
                           "sweep_avg_launch_ms": sw_ms, "slice_gpu_ms_per_step": [round(a["total"] / args.steps, 2) for a in acc]}
        if n_sl > 1:
          try:
              # the same kernel with the whole batch in ONE slice (no concurrent launches), two extra steps outside the timed region: with
              # several slices the HIP-event duration of a launch includes the time it shares the GPU with the other slices' launches
              pmc1 = os.path.join(ROOT, "profiles", "r01", "pmc_traffic_single_slice.json")
              one = SlicedDeviceBatch(make_solver, batch, 1)
              one.run(1)
              one.kernel_timing()
              one.run(2)
              ms1, n1, sw1, _ = one.kernel_timing()
              one.close()
              a1 = per_inst * float(sw1) / max(1, n1) / (ms1 / max(1, n1) * 1e-3) / 1e9
              out["roofline"]["single_slice"] = {"achieved": a1, "frac": a1 / 8000.0, "avg_launch_ms": ms1 / max(1, n1), "launches": n1,
                                                 "algorithmic_bytes_per_launch": per_inst * float(sw1) / max(1, n1),
                                                 "traffic": json.load(open(pmc1)).get("k_ne_fwd_bytes_per_launch") if os.path.exists(pmc1) else None,
                                                 "note": "whole batch as one slice, 2 steps after the timed region (traffic: profiles/r01/pmc_traffic_single_slice.json)"}
          except Exception as e:      # never lose the headline line to an auxiliary measurement
            out["roofline"]["single_slice"] = {"error": repr(e)}
        if getattr(m, "model_id", None) is not None:
          try:
              # the step in front of the path (SCP_SLS.update_jacobian -> slsqp_linearize: RK4 + forward-mode AD Jacobians, c, g, q, bounds), timed on
              # its own after the timed region: the synthetic instances above come with their A, B, so it is not part of `value`
              import ctypes as C
              from robust_nonlinear_mpc_amd import _lib as L
              g0 = dict(np.load(fixture))
              rng = np.random.default_rng(7)
              lf = make_solver(B)
              Xn = torch.from_numpy(np.tile(g0["X"], (B, 1, 1)) + 1e-3 * rng.standard_normal((B, N + 1, m.nx))).cuda()
              Un = torch.from_numpy(np.tile(g0["U"], (B, 1, 1)) + 1e-3 * rng.standard_normal((B, N, m.nu))).cuda()
              ptr = lambda t: C.c_void_p(t.data_ptr())
              for _ in range(2):
                  L.check(lf.lib.slsqp_linearize(lf.h, ptr(Xn), ptr(Un), L.DEVICE))
              torch.cuda.synchronize()
              tl = time.perf_counter()
              for _ in range(5):
                  L.check(lf.lib.slsqp_linearize(lf.h, ptr(Xn), ptr(Un), L.DEVICE))
              torch.cuda.synchronize()
              lin_ms = 1e3 * (time.perf_counter() - tl) / 5
              lf.close()
              out["config"]["linearise_ms_per_batch"] = lin_ms
              out["config"]["ms_per_step_with_linearisation_upper_bound"] = out["ms_per_step"] + lin_ms
          except Exception as e:
            out["config"]["linearise_ms_per_batch"] = None
            out["config"]["linearise_error"] = repr(e)
        if not args.no_cpu and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(batch, min(B, 4096), budget_s=12.0)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
