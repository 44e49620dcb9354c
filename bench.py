#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fast-SLS QP path (BASELINE.json: "QP solves/sec (whole node) + ms/MPC-step,
rockETH N=20 batch=4096").

A "step" is one closed-loop MPC step of the whole batch, entirely on the device (slsqp_cl_step):
    warm-start shift + solver reset -> linearise (RK4 + forward-mode AD) -> fast-SLS RTI solve (QP #1, eta, SLS sweep, tightening,
    QP #2; rti = 1, fast_sls_rti_steps = 1 as in expe/main_rocket_robust_closed_loop.py:80-85) -> nominal += delta -> plant + noise
for 4096 rocket runs per GPU that differ in their disturbance seed (BASELINE config 5's shape: seed s reproduces the stream of
np.random.seed(s), w_t = 2 rand(17) - 1), started from THE SCRIPT'S OWN INITIAL STATE (expe/main_rocket_robust_closed_loop.py:110-126), nominal
from the GPU initialiser (untimed set-up, the role IPOPT has in the script).  The timed steps are closed-loop steps 0 .. K-1 exactly as the
script runs them (default K = 30, :128): `--warmup W` runs W untimed closed-loop steps of a DISJOINT seed batch of the same size first, so
the warm-up never shifts which steps are timed.
value = QP solves / s over all ranks (weak scaling: every rank owns its own seeds; the only collective is one RCCL all-gather
of the measured states and applied inputs at the end of the timed region).

Secondary figures of the default run (labelled, after the timed region): the same loop from the state scaled to 0.3 of its distance from
hover (round 2's headline regime: few active bounds), and the whole batch as one slice.  `--workload synthetic` times round 1's step.
`--config 5` = BASELINE config 5's per-GPU shape: 1024 seeds per rank (8192 on 8 ranks), script x0, 30 steps.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
FIXTURE = {"rocket": "sweep_rocket_N20_s0.npz", "quadrotor": "sweep_quadrotor_N20_s0.npz", "pendulum": "sweep_pendulum_N10_s0.npz"}
QP_BYTES = {"rocket": 93656, "quadrotor": 64504, "pendulum": 6112}     # algorithmic bytes per QP solve
SWEEP_BYTES = {"rocket": 344000, "quadrotor": 190000, "pendulum": 12000}  # ... per SLS sweep: A, B, eta, eta_f in; beta, beta_f, back-offs out (rocket 197 + 147 KB)
SWEEP_MFLOP = 9.0                                                      # per rocket N=20 instance, scaled with nx^3 N^2 otherwise
X0_SCALE = {"rocket": 1.0, "quadrotor": 1.0, "pendulum": 1.0}      # 1.0 = the script's own initial state
X0_SCALE_SECONDARY = 0.3                                               # round 2's headline regime, kept as a labelled secondary figure
WARM_SEED0 = 1 << 20                                                   # warm-up batches use seeds from here on (disjoint from every rank's own)
PROFILE_DIR = "r03"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="timed closed-loop steps 0 .. steps-1 (the rocket script runs 30)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed closed-loop steps of a disjoint seed batch before the timed region")
    ap.add_argument("--batch", type=int, default=None, help="instances (seeds) per GPU; default 4096")
    ap.add_argument("--config", type=int, default=None, choices=[5], help="5: BASELINE config 5's per-GPU shape (--batch 1024 --x0-scale 1.0 --steps 30)")
    ap.add_argument("--model", default="rocket")
    ap.add_argument("--workload", default="closed_loop", choices=["closed_loop", "synthetic"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (synthetic step, single-slice roofline block)")
    ap.add_argument("--slices", type=int, default=None, help="independent slices of the rank's batch, each with its own handle / HIP stream / host thread "
                    "(default: 1 for the persistent launch, 3 otherwise)")
    ap.add_argument("--qp-eps", type=float, default=None, help="interior-point tolerance before the polish (default: the library's 1e-6)")
    ap.add_argument("--precision", type=int, default=0, help="0: fp64 (headline); 1: mixed fp32 factorisation / fp64 residuals (secondary figure)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--x0-scale", type=float, default=None, help="closed loop: initial state = hover + s (script x0 - hover); default 1.0 = the script's own state "
                    "(the nominal initialiser then runs its two-stage continuation)")
    ap.add_argument("--decoupled", type=int, default=2, help="2 (default): the timed closed loop is ONE persistent launch of slsqp_cl_run (k_cl_loop: wavefronts take "
                    "instances from a device-side FIFO and run one whole MPC step each time -- shift, linearisation, RTI chain, nominal update, plant); 1: slsqp_cl_run in "
                    "rounds (a chain of QP solves still running --round-budget-ms after its launch started suspends itself and resumes in the next round); both for the "
                    "rocket script's setting only; 0: one slsqp_cl_step per step for the whole slice")
    ap.add_argument("--round-budget-ms", type=float, default=8.0)
    ap.add_argument("--round-cut-frac", type=float, default=0.0, help="a round's unfinished chains also suspend once this fraction of its participants are done (0: off)")
    ap.add_argument("--secondary-synthetic", action="store_true", help="also time round 1's synthetic step as a secondary figure")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the all-core leg of cpu_baseline")
    args = ap.parse_args()
    if args.config == 5:
        args.batch = 1024 if args.batch is None else args.batch
        args.x0_scale = 1.0 if args.x0_scale is None else args.x0_scale
    if args.batch is None:
        args.batch = 4096
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU) as fresh children -- before this process
    makes any GPU call -- and pass their single JSON line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


# --------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle; C threads)
# --------------------------------------------------------------------------------------------------------------------
def cpu_baseline(m, N, data, sample, budget_s=12.0):
    """The oracle's RTI fast-SLS step (OSQP-class ADMM restatement with upstream default settings + polish, numba-kernel restatement for the
    sweep) on QPs the GPU solved, SAME data, driven by C threads (oracle/sls_oracle.c so_rti_step_batch: no interpreter lock): one thread,
    then one instance per thread on all host cores this process may use.  `sample` says which QPs `data` holds."""
    from oracle import oracle as O
    d = O.dims_of(m.nx, m.nu, m.nw, N, m.ni, m.ni_f)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    import numpy as np
    E = np.stack([m.E] * (N + 1))
    n_inst = data["A"].shape[0]

    def run(threads, budget, sel):
        args = (d, data["A"][sel], data["Bm"][sel], data["g"][sel], data["gN"][sel], data["c"][sel], data["q"][sel], data["x0_arg"][sel], m.G, m.Gf, m.gf, E,
                m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, O.default_settings())
        t0 = time.perf_counter()
        _, ok, done = O.rti_step_batch(*args, nthreads=threads, budget_s=budget)
        dt = time.perf_counter() - t0
        return 2 * done / dt, done, dt, float(ok[:done].mean()) if done else 0.0

    # one thread: a strided sub-sample, so that a budget that ends early has still seen every sampled step
    stride = max(1, n_inst // 200)
    v1, n1, t1, ok1 = run(1, 0.5 * budget_s, np.arange(0, n_inst, stride))
    vc, nc, tc, okc = (v1, n1, t1, ok1) if cores == 1 else run(cores, budget_s, np.arange(n_inst))
    return {"value": vc, "unit": "QP solves/s", "cores": cores, "kind": "port", "single_thread_value": v1, "solved_frac": okc, "single_thread_solved_frac": ok1,
            "sample": f"{sample}; 1 RTI MPC step (2 QP + 1 sweep) each, OSQP-class restatement with upstream default settings + polish, C threads: "
                      f"{nc} of {n_inst} instances on {cores} threads in {tc:.1f} s; {n1} instances (every {stride}th) on 1 thread in {t1:.1f} s"}


# --------------------------------------------------------------------------------------------------------------------
# closed-loop workload: K free-running slices of the rank's seeds
# --------------------------------------------------------------------------------------------------------------------
class ClosedLoopSlices:
    def __init__(self, m, N, seeds, n_slices, steps_total, device, tune):
        import numpy as np
        import torch
        from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream
        self.m, self.N, self.torch = m, N, torch
        B = len(seeds)
        K = max(1, min(int(n_slices), B))
        self.bounds = [(B * k // K, B * (k + 1) // K) for k in range(K)]
        self.cl, self.W = [], []
        dev = torch.device("cuda", device)
        for lo, hi in self.bounds:
            cl = ClosedLoopMPC(m, N, hi - lo, device=device)
            tune(cl.f)
            self.cl.append(cl)
            W = np.stack([disturbance_stream(s, steps_total, m.nx) for s in seeds[lo:hi]], axis=1)      # (steps, b, nx)
            self.W.append(torch.from_numpy(np.ascontiguousarray(W)).to(dev))
        torch.cuda.synchronize()
        self.step_no = 0
        self.stats = [[] for _ in self.cl]       # per slice: list over steps of qp_stats (b,2,8)
        self.step_ms = [[] for _ in self.cl]     # per slice: GPU ms of every step run so far (HIP events of slsqp_cl_step)

    def setup(self, x0, continuation=1):
        import numpy as np
        self._threads(lambda k: self.cl[k].reset(np.tile(x0, (self.cl[k].B, 1)), solve_nominal=True, continuation=continuation))
        return np.concatenate([cl.nlp_status for cl in self.cl])

    def _threads(self, fn):
        err = []

        def work(k):
            try:
                fn(k)
            except Exception as e:      # surface worker failures in the caller
                err.append(e)
        if len(self.cl) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(k,)) for k in range(len(self.cl))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if err:
            raise err[0]

    def run_decoupled(self, steps, budget_ms, cut_frac=0.0):
        """The whole closed loop of every slice through slsqp_cl_run (instances advance independently, see include/slsqp.h): one call per slice.  Only
        from closed-loop step 0.  Per-step statistics come from the device-side copy of qp_stats."""
        import ctypes as C
        import numpy as np
        from robust_nonlinear_mpc_amd import _lib as L
        assert self.step_no == 0
        acc = [dict(jac=0.0, qp=0.0, sweep=0.0, total=0.0) for _ in self.cl]
        self.rounds = [0] * len(self.cl)
        self.loop_stats = [None] * len(self.cl)

        def work(k):
            cl, W = self.cl[k], self.W[k]
            f = cl.f
            rounds = C.c_int(0)
            L.check(f.lib.slsqp_cl_run(f.h, steps, C.c_void_p(W.data_ptr()), L.DEVICE, C.byref(f.opts), float(budget_ms), float(cut_frac), C.byref(rounds)))
            self.rounds[k] = rounds.value
            if f.opts.cl_persistent:
                st = (C.c_double * L.CL_RUN_STATS_LEN)()
                L.check(f.lib.slsqp_cl_run_stats(f.h, st, L.CL_RUN_STATS_LEN))
                self.loop_stats[k] = {"waves": int(st[0]), "busy_ms": float(st[1]), "mpc_steps": int(st[2]), "launch_ms": float(st[3])}
            t = f.timing_ms()
            for key in acc[k]:
                acc[k][key] += t[key]
            self.step_ms[k] = [t["total"] / steps] * steps
        self._threads(work)
        self.step_no += steps
        return acc

    def fetch_run_stats(self, steps):
        import numpy as np
        for k, cl in enumerate(self.cl):
            qs = cl.f.get("log_qp_stats", (steps, 2, 8), np.int32)
            self.stats[k] = [np.ascontiguousarray(qs[:, s]) for s in range(steps)]

    def run(self, steps, collect_stats=True):
        """`steps` closed-loop MPC steps of every slice, each slice on its own thread / stream without waiting for the others.
        Returns per-slice sums of the GPU times (ms) of the linearisations, QP solves and sweeps."""
        import ctypes as C
        import numpy as np
        from robust_nonlinear_mpc_amd import _lib as L
        acc = [dict(jac=0.0, qp=0.0, sweep=0.0, total=0.0) for _ in self.cl]
        first = self.step_no

        def work(k):
            cl, W = self.cl[k], self.W[k]
            f = cl.f
            for i in range(first, first + steps):
                L.check(f.lib.slsqp_cl_step(f.h, cl.rti, C.c_void_p(W[i].data_ptr()), L.DEVICE, C.byref(f.opts)))
                t = f.timing_ms()
                for key in acc[k]:
                    acc[k][key] += t[key]
                self.step_ms[k].append(t["total"])
                if collect_stats:
                    self.stats[k].append(f.get("qp_stats", (2, 8), np.int32))
        self._threads(work)
        self.step_no += steps
        return acc

    def fetch_device(self, name, shape):
        import ctypes as C
        from robust_nonlinear_mpc_amd import _lib as L
        outs = []
        for cl in self.cl:
            f = cl.f
            out = self.torch.empty((f.B,) + tuple(shape), dtype=self.torch.float64, device=self.W[0].device)
            L.check(f.lib.slsqp_get(f.h, name.encode(), C.c_void_p(out.data_ptr()), L.DEVICE))
            outs.append(out)
        return self.torch.cat(outs, dim=0)

    def get(self, name, shape, dtype=None):
        import numpy as np
        return np.concatenate([cl.f.get(name, shape, dtype or np.float64) for cl in self.cl], axis=0)

    def kernel_timing(self):
        tot = [0.0, 0, 0, 0]
        self.fwd_factor_sweeps = self.factor_stages = self.qp_solves = 0
        for cl in self.cl:
            ms, n = cl.f.kernel_timing()
            tot[0] += ms; tot[1] += n; tot[2] += cl.f.fwd_instance_sweeps - 0.5 * cl.f.bwd_sweeps_skipped;      # a forward-only residual check counts as half a block solve
            tot[3] += cl.f.mx_retries
            self.fwd_factor_sweeps += cl.f.fwd_factor_sweeps; self.factor_stages += cl.f.factor_stages; self.qp_solves += cl.f.qp_solves
        return tuple(tot)

    def close(self):
        for cl in self.cl:
            cl.close()


def qp_statistics(stats):
    """stats: list over steps of (B,2,8) int arrays -> the per-QP figures the bench line reports (slot 0 = QP #1, slot 1 = QP #2)."""
    import numpy as np
    out = {}
    S = np.stack(stats)                                   # (steps, B, 2, 8)
    for slot, name in ((0, "qp1"), (1, "qp2")):
        q = S[:, :, slot, :]
        st_all = q[..., 6]
        # status -1: the instance was not part of this solve (its step had failed before); 2: the measured state contradicts its own stage-0 box,
        # flagged before any work.  The figures below are over the solves that ran.
        ran = (st_all != -1) & (st_all != 2)
        head = {"not_run_frac": float((st_all == -1).mean()), "infeasible_x0_frac": float((st_all == 2).mean()), "ran_frac": float(ran.mean())}
        nact_all = q[..., 3]
        q = q[ran]
        its, blk, fac, nact, warm, rounds, st, fb = (q[..., i] for i in range(8))
        solved = (st == 0) | (st == 4)
        cold = its > 0
        out[name] = {
            **head,
            "solved_frac": float(solved.mean()), "certified_frac": float((st == 0).mean()),
            # status 5: primal infeasible by the interior point's Farkas certificate (the reference's OSQP reports these as primal infeasible too);
            # 1 / 3: ended without an answer (iteration cap, stagnation, numerical)
            "infeasible_certified_frac": float((st == 5).mean()), "unsolved_frac": float(((st == 1) | (st == 3)).mean()),
            "started_warm_frac": float((warm > 0).mean()), "started_from_previous_last_qp_set_frac": float((warm == 2).mean()), "interior_point_frac": float(cold.mean()),
            # how the solve got to its answer: first active-set attempt (warm set, or the empty set on a cold solve) / interior point from scratch /
            # interior point restarted from the first QP's iterate / active set from the empty set after a failed warm attempt
            "path_frac": {"first_attempt": float((fb == 0).mean()), "ipm_cold": float((fb == 1).mean()), "ipm_restart": float((fb == 2).mean()),
                          "empty_set_after_warm": float((fb == 3).mean())},
            "cold_fallback_frac": float(((fb == 1) | (fb == 2)).mean()), "active_set_rounds_mean": float(rounds.mean()),
            "ipm_iters_mean": float(its[cold].mean()) if cold.any() else 0.0, "ipm_iters_p99": float(np.percentile(its[cold], 99)) if cold.any() else 0.0,
            "ipm_iters_max": int(its.max()), "block_solves_mean": float(blk.mean()), "block_solves_p99": float(np.percentile(blk, 99)), "block_solves_max": int(blk.max()),
            "factorisations_mean": float(fac.mean()),
            "active_inequalities": {"mean": float(nact.mean()), "p50": float(np.percentile(nact, 50)), "p99": float(np.percentile(nact, 99)), "max": int(nact.max()),
                                    "histogram_per_step": {"edges": [0, 1, 6, 11, 21, 41, 10 ** 9],
                                                           "counts": [np.histogram(nact_all[s][ran[s]], bins=[0, 1, 6, 11, 21, 41, 10 ** 9])[0].tolist() for s in range(S.shape[0])]}},
        }
    return out


def bench_command(args, n_slices, x0_scale):
    """What a PMC pass must have been taken on to describe this run (profiles/<round>/pmc_traffic*.json hold the same dict under "command")."""
    return {"model": args.model, "batch": args.batch, "steps": args.steps, "warmup": args.warmup, "slices": n_slices, "x0_scale": x0_scale,
            "precision": args.precision, "workload": args.workload, "decoupled": int(args.decoupled), "round_budget_ms": args.round_budget_ms, "round_cut_frac": args.round_cut_frac}


def read_traffic(fname, key, command):
    """HBM bytes per launch from the committed PMC passes -- only from a pass that was taken on exactly this command (any profiles/<round>/pmc_traffic*.json
    whose "command" dictionary equals `command`); otherwise null.  `fname` is looked at first."""
    import glob
    d0 = os.path.join(ROOT, "profiles", PROFILE_DIR)
    files = [os.path.join(d0, fname)] + sorted(f for f in glob.glob(os.path.join(d0, "pmc_traffic*.json")) if os.path.basename(f) != fname)
    seen = []
    for p in files:
        if not os.path.exists(p):
            continue
        d = json.load(open(p))
        if d.get("command") != command:
            seen.append(os.path.basename(p))
            continue
        k = key + "_timed_region" if key + "_timed_region" in d else key      # per launch of the timed region's launches only (the file also averages the untimed set-up's)
        return d.get(k), (f"profiles/{PROFILE_DIR}/{os.path.basename(p)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, scripts/pmc_traffic.py; "
                          f"build {d.get('build', '?')})")
    return None, f"no PMC pass of this command in profiles/{PROFILE_DIR} (looked at: {seen})"


def reduce_over_ranks(dt, qp_done, world, backend):
    """The timed region's wall time is the slowest rank's, the work the sum over ranks (torch.distributed all_reduce; no-op for one rank)."""
    if world <= 1:
        return dt, qp_done
    import torch
    import torch.distributed as dist
    dev = "cuda" if backend == "nccl" else "cpu"
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = torch.tensor([qp_done], device=dev, dtype=torch.float64)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return float(tmax.item()), float(tsum.item())


def gather_results(t, world, backend):
    """The only collective of the path: one all-gather of the measured states and applied inputs (RCCL over xGMI with backend nccl)."""
    if world <= 1:
        return t
    import torch
    import torch.distributed as dist
    src = t if backend == "nccl" else t.cpu()
    out = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(out, src)
    return out


QP_DATA_KEYS = ("A", "Bm", "c", "g", "gN", "q", "x0_arg")


def sample_qp_data(m, N, seeds, x0, cont, steps, every, local_rank, tune, ClosedLoopSlicesCls):
    """The CPU baseline's sample: the closed loops of `seeds` run again as their own small batch (bit-identical to their run inside the big one: instances
    are independent) and the QP data of every `every`-th step is fetched for the instances whose two QPs the GPU solved in that step."""
    import numpy as np
    n_var = (m.nx + m.nu) * N + m.nx
    shapes = {"A": (N, m.nx, m.nx), "Bm": (N, m.nx, m.nu), "c": (N, m.nx), "g": (N, m.ni), "gN": (m.ni_f,), "q": (n_var,), "x0_arg": (m.nx,)}
    small = ClosedLoopSlicesCls(m, N, seeds, 1, steps, local_rank, tune)
    small.setup(x0, cont)
    parts, taken = {k: [] for k in QP_DATA_KEYS}, []
    for i in range(steps):
        small.run(1, collect_stats=False)
        if i % every:
            continue
        f0 = small.cl[0].f
        qs = f0.get("qp_stats", (2, 8), np.int32)
        sel = np.flatnonzero(np.isin(qs[:, 0, 6], (0, 4)) & np.isin(qs[:, 1, 6], (0, 4)))
        taken.append((i, int(sel.size)))
        for k in QP_DATA_KEYS:
            parts[k].append(f0.get(k, shapes[k])[sel])
    small.close()
    data = {k: np.concatenate(v) for k, v in parts.items()}
    # interleave the steps, so that a CPU budget that ends early has seen all of them
    order = np.argsort(np.concatenate([np.arange(n) for _, n in taken]), kind="stable")
    return {k: v[order] for k, v in data.items()}, taken


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or run `python bench.py --gpus N` "
                         f"without a launcher\n")
        sys.exit(2)

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from robust_nonlinear_mpc_amd import BatchedFastSLS, get_model, make_batch
    from robust_nonlinear_mpc_amd.fast_sls import SlicedDeviceBatch
    fixture = os.path.join(ROOT, "tests", "golden", FIXTURE[args.model])
    B = args.batch
    m = get_model(args.model)
    N = int(np.load(fixture)["N"])
    nz = m.nx + m.nu
    n_var = nz * N + m.nx

    def tune(f, synthetic=False):
        if os.environ.get('QP_WARM_ROUNDS'):
            f.opts.warm_rounds = int(os.environ['QP_WARM_ROUNDS'])
        for k in ('as_first', 'as_rounds', 'as_max_viol', 'ipm_restart', 'as_warm_max_set', 'as_warm_last', 'fuse_rti'):
            if os.environ.get('QP_' + k.upper()):
                setattr(f.opts, k, int(os.environ['QP_' + k.upper()]))
        f.opts.precision = args.precision
        f.opts.time_kernels = 1          # HIP events around every launch of the dominant kernel, on its own stream (roofline leg)
        f.opts.cl_persistent = 1 if args.decoupled == 2 else 0
        if args.qp_eps is not None:
            f.opts.qp_eps = args.qp_eps
        if synthetic:
            f.opts.warm_start = 0        # fresh instances every step: QP#1 solved cold; QP#2 warm-starts from QP#1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather(t):
        return gather_results(t, world, args.backend)

    def make_synth(n_slices, seed):
        batch = make_batch(args.model, fixture, B, seed=seed)

        def make_solver(nb):
            f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=nb, device=local_rank)
            f.set_rti_steps(m.fast_sls_rti_steps if args.model == "rocket" else 1)
            tune(f, synthetic=True)
            return f
        return SlicedDeviceBatch(make_solver, batch, n_slices), batch

    # Work figures of the dominant kernel k_qp_solve (one launch = the whole QP solve of every instance of a slice: all its block-tridiagonal
    # solves and the logic between them), from device counters:
    #   fp64 flops   28.6 kflop per factorised stage (T = M1 Dinv, D_k build, Gauss-Jordan inverse; rocket, scaled with nx^3) + 2 kflop per
    #                substitution stage of a forward and of a backward sweep (scaled with nx^2)                        [DESIGN.md section 4]
    #   bytes        SURVEY.md 8(d): 93 656 B compulsory per rocket QP solve (A, B, q, l, u in; x, y out); beside it the bytes the kernel's sweeps
    #                must move at its LDS capacity, per instance forward sweep: A_k, B_k in, rhs slices in, u out (round 1's accounting)
    per_sweep_bytes = 8 * (N * m.nx * nz + 2 * n_var + N * m.nx)
    kf, ks = 28.6e3 * (m.nx / 17.0) ** 3, 2.0e3 * (m.nx / 17.0) ** 2

    sweep_flop = SWEEP_MFLOP * 1e6 * (m.nx / 17.0) ** 3 * (N / 20.0) ** 2

    def roof_block(ms_total, launches, inst_sweeps, factor_stages, qp_solves, sls_sweeps=0):
        """sls_sweeps > 0: the launches are fused RTI chains (k_rti_chain) that also ran that many SLS sweeps (9.0 Mflop per rocket instance)."""
        L = max(1, launches)
        ms = ms_total / L
        flops = (factor_stages * kf + 2.0 * inst_sweeps * N * ks + sls_sweeps * sweep_flop) / L
        tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        alg = (QP_BYTES[args.model] * float(qp_solves) + SWEEP_BYTES[args.model] * float(sls_sweeps)) / L
        work = per_sweep_bytes * float(inst_sweeps) / L
        return {"achieved": tf, "frac": tf / 78.6, "avg_launch_ms": ms, "launches": launches, "flops_per_launch": flops,
                "qp_solves_per_launch": float(qp_solves) / L, "block_solves_per_qp": float(inst_sweeps) / max(1, qp_solves),
                "factorised_stages_per_qp": float(factor_stages) / max(1, qp_solves),
                "hbm": {"algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "frac_of_8TBps": alg / (ms * 1e-3) / 8e12 if ms > 0 else 0.0,
                        "sweep_bytes_per_launch": work, "sweep_GBps": work / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}}

    def x0_of(scale):
        return m.x_ref + scale * (m.extra["x0"] - m.x_ref) if "x0" in m.extra else m.x_ref + 0.02 * (m.x_ub - m.x_lb)

    out = {}
    x0_scale = X0_SCALE[args.model] if args.x0_scale is None else args.x0_scale
    cont = 2 if x0_scale > 0.6 else 1
    if args.workload == "closed_loop":
        seeds = rank * B + np.arange(B)
        x0 = x0_of(x0_scale)
        can_decouple = bool(args.decoupled) and args.precision == 0 and m.rti == 1 and m.fast_sls_rti_steps == 1 and int(os.environ.get("QP_FUSE_RTI", "1")) != 0
        if args.slices is None:      # the persistent launch takes the rank's whole batch; every other loop runs three free-running slices
            args.slices = 1 if (can_decouple and args.decoupled == 2) else 3
        if args.warmup > 0:
            # warm the code paths (kernel code objects, allocator, clocks, the gather) on a DISJOINT seed batch of the same size: the timed region
            # below is then closed-loop steps 0 .. steps-1 of the rank's own seeds whatever --warmup says
            wdev = ClosedLoopSlices(m, N, WARM_SEED0 + rank * B + np.arange(B), args.slices, args.warmup, local_rank, tune)
            wdev.setup(x0, cont)
            if can_decouple:
                wdev.run_decoupled(args.warmup, args.round_budget_ms, args.round_cut_frac)      # the same launches as the timed region
            else:
                wdev.run(args.warmup, collect_stats=False)
            gather(torch.cat([wdev.fetch_device("x_meas", (m.nx,)), wdev.fetch_device("u0", (m.nu,))], dim=1))     # warm the gather path
            wdev.close()
        dev = ClosedLoopSlices(m, N, seeds, args.slices, args.steps, local_rank, tune)
        nlp = dev.setup(x0, cont)
        barrier()
        dev.kernel_timing()
        decoupled = can_decouple
        persistent = decoupled and args.decoupled == 2
        t0 = time.perf_counter()
        acc = dev.run_decoupled(args.steps, args.round_budget_ms, args.round_cut_frac) if decoupled else dev.run(args.steps)
        gather(torch.cat([dev.fetch_device("x_meas", (m.nx,)), dev.fetch_device("u0", (m.nu,))], dim=1))
        barrier()
        dt = time.perf_counter() - t0
        if decoupled:
            dev.fetch_run_stats(args.steps)
        x0_txt = "the script's own initial state (expe/main_rocket_robust_closed_loop.py:110-126)" if (x0_scale == 1.0 and args.model == "rocket") else \
            f"hover + {x0_scale} (script x0 - hover)"
        workload = (f"{args.model} N={N} batch={B}/GPU closed-loop Monte-Carlo from script x0" if (x0_scale == 1.0 and "x0" in m.extra) else
                    f"{args.model} N={N} batch={B}/GPU closed-loop Monte-Carlo")
        workload += (f": per instance and step shift + reset, linearise, fast-SLS RTI ({m.rti * (m.fast_sls_rti_steps + 1)} QP solves + "
                     f"{m.rti * m.fast_sls_rti_steps} SLS sweep(s)), nominal update, plant + seeded noise; x0 = {x0_txt}, nominal from the GPU "
                     f"initialiser (untimed); timed steps = closed-loop steps 0..{args.steps - 1}; warm-up = {args.warmup} step(s) of a disjoint seed batch")
        if persistent:
            workload += ("; the whole loop is ONE persistent launch (slsqp_cl_run, k_cl_loop): wavefronts take instances from a device-side FIFO and run one whole MPC step "
                         "each time, so every instance advances independently")
        elif decoupled:
            workload += (f"; the steps run through slsqp_cl_run: every instance advances independently, chains still running {args.round_budget_ms} ms after their launch "
                         f"started resume in the next round")
        per_step = [np.concatenate([dev.stats[k][s] for k in range(len(dev.cl))]) for s in range(args.steps)]
        qstat = qp_statistics(per_step)
        succ = dev.get("scp_success", (), np.int32)
        step_ms = np.mean(np.array(dev.step_ms), axis=0)          # per closed-loop step: GPU ms of slsqp_cl_step, mean over the rank's slices (they run concurrently)
        extra_cfg = {"nominal_initialiser_converged_frac": float((nlp == 0).mean()), "mpc_step_success_frac_last_step": float(succ.mean()), "qp": qstat,
                     "linearise_ms_per_step": float(np.mean([a["jac"] for a in acc])) / args.steps,
                     "rounds_per_slice": getattr(dev, "rounds", None), "round_budget_ms": args.round_budget_ms if (decoupled and not persistent) else None,
                     "persistent_launch": ([dict(ls, wave_busy_frac=ls["busy_ms"] / max(1e-9, ls["waves"] * ls["launch_ms"]), ms_per_mpc_step_in_a_wave=ls["busy_ms"] / max(1, ls["mpc_steps"]))
                                            for ls in dev.loop_stats] if persistent else None),
                     "per_step": {"slice_gpu_ms": [round(float(v), 3) for v in step_ms],
                                  "qp_solves_run": [int(((st[:, :, 6] != -1) & (st[:, :, 6] != 2)).sum()) for st in per_step],
                                  "note": "slice_gpu_ms: HIP-event time of one slice's slsqp_cl_step, mean over the slices of rank 0; slices overlap, so the sum "
                                          "over steps exceeds the wall time"}}
    else:
        decoupled = persistent = False
        args.slices = 3 if args.slices is None else args.slices
        dev, batch = make_synth(args.slices, 1234 + rank)
        dev.run(max(1, args.warmup))
        gather(dev.fetch_device("primal_vec", (n_var,))[:, m.nx:m.nx + m.nu].contiguous())
        barrier()
        dev.kernel_timing()
        t0 = time.perf_counter()
        acc = dev.run(args.steps)
        gather(dev.fetch_device("primal_vec", (n_var,))[:, m.nx:m.nx + m.nu].contiguous())
        barrier()
        dt = time.perf_counter() - t0
        workload = (f"{args.model} N={N} batch={B}/GPU synthetic instances (seeded perturbations of one nominal; 0-8 active inequalities), fast-SLS RTI step "
                    f"(update_dynamics + update_linear_cost + solve: 2 QP solves + 1 SLS sweep per instance), no linearisation")
        extra_cfg = {}
    k_ms, k_launches, inst_sweeps, mx_retries = dev.kernel_timing()
    fact_stages, qp_solves = dev.factor_stages, dev.qp_solves
    qp_done = float(qp_solves)      # QP solves of the timed region that actually ran (device counter), this rank
    dt, qp_done = reduce_over_ranks(dt, qp_done, world, args.backend)
    n_sl = len(dev.bounds)
    # QP solves per instance and step: closed loop = rti SCP iterations x (fast_sls_rti_steps + 1) QPs (rocket script: 1 x 2; pendulum / quadrotor: 3 x 3)
    qp_per_inst = (m.rti * (m.fast_sls_rti_steps + 1)) if args.workload == "closed_loop" else ((m.fast_sls_rti_steps if args.model == "rocket" else 1) + 1)
    # value counts the QP solves that RAN: an MPC step whose measured state lies outside its own stage-0 box (noise pushed it over a state bound the
    # nominal touches) is flagged infeasible before any work, as the reference's QP would be, and its two QPs are not counted
    qp_nominal = qp_per_inst * B * world * args.steps
    value = qp_done / dt
    headline = (args.model, B, args.workload, x0_scale) == ("rocket", 4096, "closed_loop", 1.0)
    out = {
        "metric": ("QP solves/sec (whole node), rockETH N=20 batch=4096 closed-loop RTI MPC step from script x0" if headline
                   else f"QP solves/sec (whole node), {args.model} N={N} batch={B} {args.workload} RTI MPC step"), "value": value, "unit": "QP solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == 0 else "f32 factorisation + f64 residuals (mixed)",
        "data": "synthetic",
        "config": dict({"workload": workload, "qp_n": n_var, "qp_m": N * (m.nx + m.ni) + m.ni_f + m.nx, "slices_per_gpu": n_sl,
                        "qp_solves_counted": qp_done, "qp_solves_nominal": qp_nominal, "qp_solves_executed_frac": qp_done / qp_nominal}, **extra_cfg),
    }
    if rank == 0:
        st = dev.get("status", (), np.int32)
        out["config"]["last_qp_solved_frac"] = float(np.mean((st == 0) | (st == 4)))
        out["config"]["last_qp_certified_frac"] = float(np.mean(st == 0))
        # dominant kernel: k_qp_solve; time = HIP events around every launch on the launching stream (opts.time_kernels), work = device counters.
        # The kernel is bound by vector-ALU instruction issue: its roof is the fp64 peak (matrix = vector = 78.6 TFLOP/s).
        # rocket script setting (rti = 1, one fast-SLS step) in fp64: the whole RTI solve of an instance is ONE launch (k_rti_chain: QP -> eta -> Riccati /
        # propagation -> tightened bounds -> QP), the timed launches are those and their work includes the SLS sweeps of the instances whose first QP solved
        fused = (args.workload == "closed_loop" and m.fast_sls_rti_steps == 1 and args.precision == 0 and int(os.environ.get("QP_FUSE_RTI", "1")) != 0
                 and (decoupled or (B // max(1, n_sl)) * (N + 1) >= 3072)
                 and int(os.environ.get("SLSQP_FUSE_RTI", "1")) != 0)
        sls_sweeps = int(sum(int(np.isin(st_[:, 0, 6], (0, 4)).sum()) for st_ in per_step)) if fused else 0
        dom_kernel = "k_cl_loop" if persistent else ("k_rti_chain" if fused else "k_qp_solve")
        rb = roof_block(k_ms, k_launches, inst_sweeps, fact_stages, qp_solves, sls_sweeps)
        # HBM traffic from PMC counters: attached only when the committed passes were taken on exactly this command
        command = bench_command(args, n_sl, x0_scale)
        traffic, tsrc = read_traffic("pmc_traffic.json", dom_kernel + "_bytes_per_launch", command)
        calls = (sum(dev.rounds) if getattr(dev, "rounds", None) else args.steps * n_sl)
        step_tf = (fact_stages * kf + 2.0 * inst_sweeps * N * ks + args.steps * B * sweep_flop) / dt / 1e12
        if persistent:
            rb["note_work"] = ("one launch = the whole closed loop; the flops counted are those of the QP solves and SLS sweeps only (the linearisation, nominal update and "
                               "plant steps the kernel also runs are not counted)")
        out["roofline"] = dict({"bound": "mfma", "kernel": dom_kernel, "peak": 78.6, "unit": "TFLOP/s", "traffic": traffic, "traffic_source": tsrc, "command": command}, **rb)
        out["roofline"].update({
            "note": "fp64: matrix peak = vector peak on MI355X; the kernel issues its block products on the matrix core and is bound by vector-ALU issue",
            "qp_solve": (None if persistent else
                         {"avg_ms": sum(a["qp"] for a in acc) / (qp_per_inst * calls), "instances": B / n_sl,
                          "note": ("fused chain: (launch duration - the sweep part as instance 0 spent it in the kernel) / 2" if fused else "one solve call = one slice") +
                                  "; slices run concurrently, so these times overlap"}),
            # the whole step: fp64 work of the timed region (QP kernel + 9.0 Mflop per rocket instance for the SLS sweep) over wall time
            "whole_step_fp64": {"achieved_TFLOPs": step_tf, "peak_TFLOPs": 78.6, "frac": step_tf / 78.6},
            "sweep_avg_launch_ms": (None if (args.workload == "closed_loop" and decoupled) else sum(a["sweep"] for a in acc) / calls), "slice_gpu_ms_per_step": [round(a["total"] / args.steps, 2) for a in acc]})
        cpu_data, cpu_sample = None, ""
        if args.workload == "synthetic" and not args.no_cpu:
            ncpu = min(B, 1024)
            cpu_data = {"A": batch["A"][:ncpu], "Bm": batch["B"][:ncpu], "c": batch["c"][:ncpu], "g": batch["g"][:ncpu], "gN": batch["gN"][:ncpu], "q": batch["q"][:ncpu],
                        "x0_arg": batch["x0_arg"][:ncpu]}
            cpu_sample = f"the first {ncpu} synthetic instances of rank 0"
        dev.close()
        if not args.no_secondary and world == 1:
            try:
                # the same kernels with the whole batch in ONE slice (no concurrent launches) after the timed region: with several slices the HIP-event
                # duration of a launch includes the time it shares the GPU with the other slices' launches
                if n_sl > 1 and args.workload == "closed_loop":
                    one = ClosedLoopSlices(m, N, seeds, 1, args.steps, local_rank, tune)
                    one.setup(x0, cont)
                    torch.cuda.synchronize()
                    one.kernel_timing()
                    t1 = time.perf_counter()
                    a1 = one.run(args.steps, collect_stats=False)
                    torch.cuda.synchronize()
                    dt1 = time.perf_counter() - t1
                    ms1, n1, sw1, _ = one.kernel_timing()
                    fs1, qs1 = one.factor_stages, one.qp_solves
                    one.close()
                    tr1, ts1 = read_traffic("pmc_traffic_single_slice.json", dom_kernel + "_bytes_per_launch", bench_command(args, 1, x0_scale))
                    out["roofline"]["single_slice"] = dict(roof_block(ms1, n1, sw1, fs1, qs1, sls_sweeps), traffic=tr1, traffic_source=ts1, ms_per_step=1e3 * dt1 / args.steps,
                                                           gpu_ms_per_step={k: a1[0][k] / args.steps for k in a1[0]},
                                                           note="same closed-loop steps with the whole batch as one slice, after the timed region")
            except Exception as e:      # never lose the headline line to an auxiliary measurement
                out["roofline"]["single_slice"] = {"error": repr(e)}
            try:
                if args.workload == "closed_loop" and headline:
                    # round 2's headline regime as a labelled secondary figure: the same loop from the state scaled to 0.3 of its distance from hover,
                    # closed-loop steps 1..5 timed after step 0 (what profiles/r02/bench_line.json timed)
                    sec_slices = 3
                    sec = ClosedLoopSlices(m, N, seeds, sec_slices, 6, local_rank, tune)
                    sec.setup(x0_of(X0_SCALE_SECONDARY), 1)
                    sec.run(1, collect_stats=False)
                    torch.cuda.synchronize()
                    sec.kernel_timing()
                    t1 = time.perf_counter()
                    sec.run(5, collect_stats=False)
                    torch.cuda.synchronize()
                    dts = (time.perf_counter() - t1) / 5
                    sec.kernel_timing()
                    qs_sec = sec.qp_solves
                    sec.close()
                    out["secondary"] = {"workload": f"round-2 headline regime: the same closed loop from hover + {X0_SCALE_SECONDARY} (script x0 - hover), closed-loop steps 1..5 "
                                                    "(few active bounds, no interior point)", "ms_per_step": 1e3 * dts, "qp_solves_per_s": qs_sec / (5 * dts), "steps": 5,
                                        "slices": sec_slices, "loop": "one slsqp_cl_step per step and slice"}
            except Exception as e:
                out["secondary"] = {"error": repr(e)}
            if args.secondary_synthetic:
                try:
                    syn, _ = make_synth(args.slices, 1234)
                    syn.run(1)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    syn.run(3)
                    torch.cuda.synchronize()
                    dts = (time.perf_counter() - t1) / 3
                    sst = syn.get("status", (), np.int32)
                    syn.close()
                    out["secondary_synthetic"] = {"workload": "round-1 headline: synthetic instances (0-8 active inequalities of 874), update_dynamics + update_linear_cost + solve, "
                                                              "no linearisation, QP#1 cold every step", "ms_per_step": 1e3 * dts, "qp_solves_per_s": 2 * B / dts,
                                                  "certified_frac": float(np.mean(sst == 0)), "steps": 3, "slices": args.slices}
                except Exception as e:
                    out["secondary_synthetic"] = {"error": repr(e)}
        if args.workload == "closed_loop" and not args.no_cpu:
            try:
                # QPs sampled ACROSS the timed steps: the first 128 seeds of rank 0, every third closed-loop step (~1000 MPC steps: 20-30 s of CPU work)
                ns, every = min(128, B), 3
                cpu_data, taken = sample_qp_data(m, N, seeds[:ns], x0, cont, args.steps, every, local_rank, tune, ClosedLoopSlices)
                cpu_sample = (f"the MPC steps of seeds {int(seeds[0])}..{int(seeds[ns - 1])} at closed-loop steps {[t for t, _ in taken]} of the timed run whose two QPs the GPU "
                              f"solved ({[c for _, c in taken]} instances per step; the steps interleaved)")
            except Exception as e:
                cpu_data, cpu_sample = None, repr(e)
        if cpu_data is not None and cpu_data["A"].shape[0] > 0:
            try:
                out["cpu_baseline"] = cpu_baseline(m, N, cpu_data, cpu_sample, args.cpu_budget)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        else:
            out["cpu_baseline"] = {"error": cpu_sample} if cpu_sample else None
        print(json.dumps(out))
    else:
        dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
