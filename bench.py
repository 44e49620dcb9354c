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
    from robust_nonlinear_mpc_amd.fast_sls import DeviceBatch
    fixture = os.path.join(ROOT, "tests", "golden", FIXTURE[args.model])
    B = args.batch
    batch = make_batch(args.model, fixture, B, seed=1234 + rank)
    m, N = batch["model"], batch["N"]
    f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=B, device=local_rank)
    f.set_rti_steps(m.fast_sls_rti_steps if args.model == "rocket" else 1)
    f.opts.warm_rounds = int(os.environ.get('QP_WARM_ROUNDS', '4'))
    f.opts.precision = args.precision
    f.opts.warm_start = 0   # QP#1 of every step is solved cold (fresh Monte-Carlo instances); QP#2 warm-starts from QP#1
    dev = DeviceBatch(f, batch)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def collect():
        u0 = dev.fetch_device("primal_vec", (f.n,))[:, m.nx:m.nx + m.nu].contiguous()
        if world > 1:
            src = u0 if args.backend == "nccl" else u0.cpu()
            gathered = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(gathered, src)          # RCCL over xGMI: collect the first inputs of every instance
            return gathered
        return u0

    for _ in range(args.warmup):
        dev.step()
    collect()   # warm the gather path too (first-use kernel loads are not part of the step)
    barrier()
    t_qp = t_sw = t_tot = 0.0
    f.kernel_timing()   # reset the per-kernel accumulators
    host_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        dev.step()
        host_ms.append(1e3 * (time.perf_counter() - ts))
        tm = f.timing_ms()
        t_qp += tm["qp"]
        t_sw += tm["sweep"]
        t_tot += tm["total"]
    collect()
    barrier()
    dt = time.perf_counter() - t0
    fwd_total_ms, fwd_launches = f.kernel_timing()
    if world > 1:
        tmax = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st = f.get("status", (), np.int32)
    its = f.get("qp_iters", (), np.int32)
    qps_per_step = 2 * B            # RTI: QP#1 + QP#2 per instance
    launches = 2 * args.steps
    value = qps_per_step * world * args.steps / dt
    out = {
        "metric": "QP solves/sec (whole node), rockETH N=20 batch=4096 RTI MPC step", "value": value, "unit": "QP solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == 0 else "f32 factorisation + f64 residuals (mixed)", "data": "synthetic",
        "config": {"workload": f"{args.model} N={N} batch={B}/GPU, fast-SLS RTI step (rti_steps=1: 2 QP solves + 1 SLS sweep per instance)",
                   "qp_n": f.n, "qp_m": f.mb + m.nx, "solved_frac": float(np.mean((st == 0) | (st == 4))),
                   "polished_frac": float(np.mean(st == 0)), "ipm_iters_mean_last_qp": float(its.mean()), "ipm_iters_max_last_qp": int(its.max()),
                   "qp2_cold_fallback_frac": float(np.mean(its > 0)), "tightened_frac": float(np.mean(f.get("backoff_x", (N + 1, m.nx)).max(axis=(1, 2)) > 0))},
    }
    if rank == 0:
        kk = f.get("kkt", (8,))
        # dominant kernel: k_ne_fwd (block-tridiagonal forward sweep; re-factorises in half of its launches).  Algorithmic bytes
        # of one launch = for every instance that does work: A_k,B_k of all stages in, rhs slices (Pi, v) in, u out.
        nz = m.nx + m.nu
        per_inst = 8 * (N * m.nx * nz + 2 * f.n + N * m.nx)
        fwd_ms = fwd_total_ms / max(1, fwd_launches)
        inst_launches_last_qp = float(kk[:, 7].sum())           # instance-launches of the last QP solve (device counters)
        # all QP solves of the timed region: scale the last solve's instance-launch count by the measured launch counts
        alg_bytes_launch = per_inst * inst_launches_last_qp / max(1.0, kk[:, 7].max())
        achieved = alg_bytes_launch / (fwd_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("k_ne_fwd_bytes_per_launch")
        qp_ms = t_qp / launches
        # fp64 work of one QP solve: 28.6 kflop per factorised stage, 2 kflop per solve-only stage sweep
        flop_last = float((kk[:, 6] * N * 28.6e3 + (kk[:, 7] - kk[:, 6]) * N * 2.0e3 + kk[:, 7] * N * 2.0e3).sum())
        out["roofline"] = {"bound": "hbm", "kernel": "k_ne_fwd", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                           "traffic": traffic, "avg_launch_ms": fwd_ms, "launches": fwd_launches, "algorithmic_bytes_per_launch": alg_bytes_launch,
                           "qp_solve": {"avg_ms": qp_ms, "algorithmic_bytes": QP_BYTES[args.model] * B,
                                        "achieved_GBps": QP_BYTES[args.model] * B / (qp_ms * 1e-3) / 1e9},
                           "fp64": {"achieved_TFLOPs_last_qp": flop_last / (qp_ms * 1e-3) / 1e12 if qp_ms > 0 else None, "peak_TFLOPs": 78.6,
                                    "note": "vector fp64 peak from BASELINE.md (AMD public figure); the guide lists no fp64 peak"},
                           "sweep_avg_launch_ms": t_sw / args.steps, "solve_call_gpu_ms": t_tot / args.steps,
                           "host_ms_per_step": [round(x, 2) for x in host_ms]}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(batch, min(B, 4096), budget_s=12.0)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    f.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
