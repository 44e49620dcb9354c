"""CPU-side tests (no GPU): C-ABI exports, host layout logic, oracle QP certificates, multi-rank shard/gather over gloo."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT
from problems import make_instance, oracle_dims, qp1_bounds, run_oracle_fastsls


def test_cabi_library_exports_every_declared_symbol():
    from robust_nonlinear_mpc_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.SO_PATH)
    header = open(os.path.join(ROOT, "include", "slsqp.h")).read()
    declared = set(re.findall(r"\b(slsqp_[A-Za-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    lib.slsqp_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.slsqp_version()


def test_no_gpu_fails_loudly():
    """Without a HIP device slsqp_create must fail with a message, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from robust_nonlinear_mpc_amd import BatchedFastSLS, get_model
    m = get_model("pendulum")
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        BatchedFastSLS(10, m.Q, m.R, m, m.Qf, batch=2)


@pytest.mark.parametrize("name,N,n,m,nnz", [("pendulum", 10, 54, 152, 352), ("quadrotor", 20, 353, 979, 5399), ("rocket", 20, 437, 1231, 8371)])
def test_structural_known_answers(name, N, n, m, nnz):
    """n, m, nnz(A) of the reference's QP (derivable from qp_jit.py:77-192 without running it)."""
    from robust_nonlinear_mpc_amd import get_model, _lib
    md = get_model(name)
    assert (md.n_var(N), md.m_con(N)) == (n, m)
    lib = ctypes.CDLL(_lib.SO_PATH)
    d = _lib.Dims(md.nx, md.nu, md.nw, N, md.ni, md.ni_f)
    a, b, c, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.slsqp_qp_nnz(ctypes.byref(d), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(e))
    assert (a.value, b.value, e.value) == (n, m, nnz)


def test_models_match_reference_constants(golden_dir):
    from robust_nonlinear_mpc_amd import get_model
    for name in ("pendulum", "quadrotor", "rocket"):
        g = dict(np.load(os.path.join(golden_dir, f"dyn_{name}.npz")))
        m = get_model(name)
        assert list(g["dims"]) == [m.nx, m.nu, m.nw, m.ni, m.ni_f]
        assert np.array_equal(g["G"], m.G) and np.array_equal(g["Gf"], m.Gf)
        assert np.allclose(g["g"], m.g) and np.allclose(g["gf"], m.gf)
        assert np.allclose(g["E_script"], m.E, rtol=1e-14)
        assert np.allclose(g["x_ref"], m.x_ref) and np.allclose(g["u_ref"], m.u_ref)


@pytest.mark.parametrize("model", ["pendulum", "quadrotor", "rocket"])
def test_oracle_qp_kkt_certificate(model):
    """The OSQP-class restatement driven to eps 1e-9 + polish satisfies the KKT conditions of the reference's QP."""
    from oracle import oracle as O
    inst = make_instance(model, 0, 1.0)
    m, d = inst.m, oracle_dims(inst)
    l, u = qp1_bounds(inst)
    x, y, info = O.qp_solve(d, inst.A, inst.B, m.G, m.Gf, m.Q, m.R, m.Qf, inst.q, l, u, O.tight_settings())
    assert info.status == 1
    k = O.qp_kkt(d, inst.A, inst.B, m.G, m.Gf, m.Q, m.R, m.Qf, inst.q, l, u, x, y)
    assert k["stationarity"] < 1e-8 and k["primal"] < 1e-8 and k["dual_sign"] < 1e-8 and k["complementarity"] < 1e-7
    # pinned x0 (qp_jit.py:376-379) and dynamics rows hold
    assert np.allclose(x[: m.nx], -inst.x0_arg, atol=1e-9)


@pytest.mark.parametrize("model,amp", [("pendulum", 1.0), ("quadrotor", 1.0), ("rocket", 0.5)])
def test_oracle_qp_agrees_with_an_independent_interior_point(model, amp):
    """QP parity is "unpinned" (no OSQP output in the reference), so the oracle's OSQP-class ADMM restatement is cross-checked on the CPU
    against a second algorithm that shares nothing with it: a dense Mehrotra interior point in numpy (tests/ref_ipm.py)."""
    from oracle import oracle as O
    from problems import make_instance, oracle_dims, qp1_bounds
    from ref_ipm import build_equalities, qp_box
    for seed in range(2):
        inst = make_instance(model, seed, amp)
        m, N = inst.m, inst.N
        nx, nz = m.nx, m.nz
        n = nz * N + nx
        l, u = qp1_bounds(inst)
        xo, yo, info = O.qp_solve(oracle_dims(inst), inst.A, inst.B, m.G, m.Gf, m.Q, m.R, m.Qf, inst.q, l, u, O.tight_settings())
        assert info.status in (1, 2)
        SR = nx + m.ni
        hi, lo = np.full(n, 1e20), np.full(n, -1e20)
        for k in range(N):
            hi[k * nz:(k + 1) * nz] = u[k * SR + nx:k * SR + nx + nz]
            lo[k * nz:(k + 1) * nz] = -u[k * SR + nx + nz:k * SR + nx + 2 * nz]
        hi[N * nz:], lo[N * nz:] = u[N * SR:N * SR + nx], -u[N * SR + nx:N * SR + 2 * nx]
        hi[:nx], lo[:nx] = 1e20, -1e20                                    # x_0 is pinned by its equality rows
        c = np.stack([-0.5 * (u[k * SR:k * SR + nx] + l[k * SR:k * SR + nx]) for k in range(N)])
        E, e = build_equalities(inst.A, inst.B, c, 0.5 * (l[-nx:] + u[-nx:]))
        Pd = 2.0 * np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
        z, nu, lu, ll, ok, its = qp_box(Pd, inst.q, E, e, lo, hi)
        assert ok, its
        # like OSQP, the restatement only promises eps-accuracy when its polish step fails; with a successful polish it is exact
        tol = 1e-6 if (info.status == 1 and info.polish_status == 1) else 1e-4
        assert np.max(np.abs(z - xo)) < tol * max(1.0, np.max(np.abs(xo)))


def test_oracle_qp_unconstrained_equals_lq_riccati():
    """With slack bounds the QP optimum is the LQ solution obtained with OCP.riccati_step (solver/ocp.py:103-109)."""
    from oracle import oracle as O
    inst = make_instance("pendulum", 1, 0.05)
    m, d, N = inst.m, oracle_dims(inst), inst.N
    big = [1e6 * np.ones(m.ni)] * N + [1e6 * np.ones(m.ni_f)]
    qp = O.OracleQP(d, m.G, m.Gf, m.g, m.gf, m.Q, m.R, m.Qf, O.tight_settings())
    qp.update_dynamics(inst.A, inst.B, big)
    qp.offset_constraints(np.zeros((m.nx, N)))
    sol = qp.solve(inst.x0_arg)
    assert sol["success"]
    # Riccati recursion (cost x'Qx + u'Ru, terminal Qf), u_k = K_k x_k
    S, Ks = m.Qf, []
    for k in range(N - 1, -1, -1):
        A, B = inst.A[k], inst.B[k]
        x_, y_ = B.T @ S, A.T @ S
        K = -np.linalg.solve(m.R + x_ @ B, x_ @ A)
        S = m.Q + y_ @ A + y_ @ B @ K
        Ks.append(K)
    Ks = Ks[::-1]
    x = -inst.x0_arg
    for k in range(N):
        u = Ks[k] @ x
        assert np.allclose(sol["primal_u"][:, k], u, atol=1e-7)
        x = inst.A[k] @ x + inst.B[k] @ u
    assert np.allclose(sol["primal_x"][:, N], x, atol=1e-7)


def test_oracle_fastsls_quirk_q5_skips_tightening():
    inst = make_instance("pendulum", 0, 0.5)
    r1 = run_oracle_fastsls(inst, 1)
    r2 = run_oracle_fastsls(inst, 1, prev_primal=r1["primal_vec"] + 1e-4)
    assert r1["iteration_number"] == 1 and r2["iteration_number"] == 0
    assert np.allclose(r2["backoff"], inst.N * 1e-5) and np.allclose(r2["backoff_x"], 0.0)


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from robust_nonlinear_mpc_amd.sharding import shard_range, gather_rows
    B = 11
    lo, hi = shard_range(B, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float64)[:, None] * torch.ones(1, 3, dtype=torch.float64)
    full = gather_rows(local, B, world)
    q.put((rank, lo, hi, full.numpy().copy()))
    dist.destroy_process_group()


def test_shard_and_gather_world_size_2_gloo():
    """N>1 path: contiguous instance shards per rank, one all_gather of the results."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(60) for p in ps]
    ranges = sorted((lo, hi) for _, lo, hi, _ in res)
    assert ranges == [(0, 6), (6, 11)]
    for _, _, _, full in res:
        assert full.shape == (11, 3) and np.array_equal(full[:, 0], np.arange(11.0))


def test_disturbance_stream_matches_reference_seed0(golden_dir):
    """np.random.seed(0); w = 2*rand(17)-1 per step (expe/main_rocket_robust_closed_loop.py:30,180)."""
    from robust_nonlinear_mpc_amd import disturbance_stream
    W = np.load(os.path.join(golden_dir, "rocket_noise_seed0.npz"))["W"]
    assert np.array_equal(disturbance_stream(0, 30, 17), W)


def _fake_run_slice(model, N, seeds, steps, x0, device, noise, solve_nominal, continuation=1, budget_ms=None):
    """Stand-in for the GPU closed loop of one slice: trajectories that encode (seed, step) so the gather can be checked exactly."""
    seeds = np.asarray(seeds)
    S = len(seeds)
    st = seeds[:, None, None] * 1000.0 + np.arange(model.nx)[None, :, None] * 10.0 + np.arange(steps)[None, None, :]
    ut = -(seeds[:, None, None] * 1000.0 + np.arange(model.nu)[None, :, None] * 10.0 + np.arange(steps - 1)[None, None, :])
    return dict(state_trajectory=st, input_trajectory=ut, success=np.ones((S, steps), dtype=bool),
                t_qp=np.full((steps, 1), float(S)), t_riccati=np.zeros((steps, 1)), t_jac=np.zeros((steps, 1)))


def _mc_gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from robust_nonlinear_mpc_amd import get_model, monte_carlo
    monte_carlo._run_slice = _fake_run_slice                     # no GPU here: only the sharding / slicing / gather logic runs
    m = get_model("pendulum")
    seeds = np.arange(100, 111)                                   # 11 seeds over 2 ranks: uneven shards (6 + 5), 2 slices per rank
    r = monte_carlo.run_monte_carlo(m, 10, seeds, 4, m.extra["x0"], rank=rank, world=world, slices=2)
    q.put((rank, r["seeds"].copy(), r["state_trajectory"].copy(), r["state_trajectory_all"].copy(), r["input_trajectory_all"].copy(), r["t_qp"].copy()))
    dist.destroy_process_group()


def test_run_monte_carlo_gather_branch_world_size_2_gloo():
    """run_monte_carlo(rank, world): contiguous seed shards, slices inside a rank, one all_gather of the trajectories to every rank
    (BASELINE config 5's multi-GPU leg) -- driven on two gloo ranks with the GPU slice replaced by a stand-in."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_mc_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    seeds = np.arange(100, 111)
    want = _fake_run_slice(__import__("robust_nonlinear_mpc_amd").get_model("pendulum"), 10, seeds, 4, None, 0, True, False)
    assert list(res[0][1]) == list(seeds[:6]) and list(res[1][1]) == list(seeds[6:])
    for rank, mine, st, st_all, u_all, t_qp in res:
        lo = 0 if rank == 0 else 6
        assert np.array_equal(st, want["state_trajectory"][lo:lo + len(mine)])           # slices concatenated in seed order
        assert np.array_equal(st_all, want["state_trajectory"])                            # every rank holds every seed's trajectory
        assert np.array_equal(u_all, want["input_trajectory"])
        assert t_qp.max() == 3.0                                                           # slices run concurrently: max over the 3 + 3 (or 2 + 3) split


class _FakeSolver:
    """What bench.py touches on ClosedLoopMPC.f, without a GPU."""
    def __init__(self, B, nx, nu):
        self.B, self.nx, self.nu = B, nx, nu
        self.fwd_instance_sweeps = self.bwd_sweeps_skipped = self.mx_retries = self.fwd_factor_sweeps = self.factor_stages = self.qp_solves = 0

    def kernel_timing(self):
        self.fwd_instance_sweeps, self.factor_stages, self.fwd_factor_sweeps = self.pending * 6, self.pending * 40, self.pending * 2
        self.qp_solves, self.pending = self.pending, 0
        return 1.0, 2


class _FakeSlices:
    """Stand-in for bench.ClosedLoopSlices: every step "solves" both QPs of every instance; x_meas encodes the seed so the gather can be checked."""
    instances = []

    def __init__(self, m, N, seeds, n_slices, steps_total, device, tune):
        import types
        self.m, self.seeds = m, np.asarray(seeds)
        B = len(seeds)
        K = max(1, min(int(n_slices), B))
        self.bounds = [(B * k // K, B * (k + 1) // K) for k in range(K)]
        self.cl = [types.SimpleNamespace(f=_FakeSolver(hi - lo, m.nx, m.nu), B=hi - lo) for lo, hi in self.bounds]
        for c in self.cl:
            c.f.pending = 0
        self.stats, self.step_ms = [[] for _ in self.cl], [[] for _ in self.cl]
        _FakeSlices.instances.append(self)

    def setup(self, x0, continuation=1):
        return np.zeros(len(self.seeds), dtype=np.int32)

    def run(self, steps, collect_stats=True):
        for k, c in enumerate(self.cl):
            for _ in range(steps):
                c.f.pending += 2 * c.B
                self.step_ms[k].append(1.0)
                if collect_stats:
                    st = np.zeros((c.B, 2, 8), dtype=np.int32); st[..., 1] = 3
                    self.stats[k].append(st)
        return [dict(jac=0.1 * steps, qp=0.5 * steps, sweep=0.2 * steps, total=1.0 * steps) for _ in self.cl]

    def run_decoupled(self, steps, budget_ms, cut_frac=0.0):
        self.rounds = [1] * len(self.cl)      # the persistent launch: no rounds
        self.loop_stats = [dict(waves=c.B, busy_ms=0.8 * steps * c.B, mpc_steps=steps * c.B, launch_ms=1.0 * steps) for c in self.cl]
        return self.run(steps)

    def fetch_run_stats(self, steps):
        pass

    def fetch_device(self, name, shape):
        import torch
        return torch.from_numpy(np.tile(self.seeds[:, None].astype(float), (1,) + tuple(shape)))

    def get(self, name, shape, dtype=None):
        return np.ones((len(self.seeds),) + tuple(shape), dtype=dtype or np.float64) if name == "scp_success" else np.zeros((len(self.seeds),) + tuple(shape), dtype=dtype or np.float64)

    def kernel_timing(self):
        tot = [0.0, 0, 0, 0]
        self.fwd_factor_sweeps = self.factor_stages = self.qp_solves = 0
        for c in self.cl:
            ms, n = c.f.kernel_timing()
            tot[0] += ms; tot[1] += n; tot[2] += c.f.fwd_instance_sweeps
            self.factor_stages += c.f.factor_stages; self.qp_solves += c.f.qp_solves
        return tuple(tot)

    def close(self):
        pass


def _bench_gloo_worker(rank, world, port, q):
    import io, contextlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import bench
    torch.cuda.set_device = lambda d: None                       # no GPU here: bench.py's own rank / reduce / gather / JSON logic runs, the slice is a stand-in
    torch.cuda.synchronize = lambda *a: None
    bench.ClosedLoopSlices = _FakeSlices
    gathered = []
    real_gather = bench.gather_results
    bench.gather_results = lambda t, w, b: gathered.append(real_gather(t, w, b)) or gathered[-1]
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "8", "--backend", "gloo", "--no-cpu", "--no-secondary", "--model", "rocket"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    q.put((rank, buf.getvalue(), [[t.numpy().copy() for t in g] for g in gathered]))


def test_bench_multi_rank_path_world_size_2_gloo():
    """bench.py's own N > 1 path end to end on two gloo ranks (the GPU slice replaced by a stand-in): rank r owns seeds [r B, (r+1) B), the warm-up
    batch is disjoint from both, value = QP solves of ALL ranks over the slowest rank's time, the one all-gather carries every rank's rows, and only
    rank 0 prints the JSON line (n_gpus 2, weak scaling)."""
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    ps = [ctx.Process(target=_bench_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=240) for _ in ps], key=lambda t: t[0])
    [p.join(60) for p in ps]
    assert res[1][1].strip() == ""                                             # rank 1 prints nothing
    line = json.loads(res[0][1].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["qp_solves_counted"] == 2 * 8 * 3 * 2                 # 2 QPs x 8 instances x 3 steps x 2 ranks (the warm-up batch is not counted)
    assert line["config"]["qp_solves_nominal"] == 96 and abs(line["value"] * line["ms_per_step"] * 3e-3 - 96) < 1e-6
    assert "closed-loop steps 0..2" in line["config"]["workload"] and "disjoint seed batch" in line["config"]["workload"]
    assert line["roofline"]["traffic"] is None and line["cpu_baseline"] is None
    assert line["roofline"]["kernel"] == "k_cl_loop" and line["config"]["slices_per_gpu"] == 1 and "ONE persistent launch" in line["config"]["workload"]
    assert abs(line["config"]["persistent_launch"][0]["wave_busy_frac"] - 0.8) < 1e-12
    for rank, _, gathered in res:
        warm, timed = gathered[0], gathered[-1]
        assert len(timed) == 2 and np.array_equal(timed[0][:, 0], np.arange(8.0)) and np.array_equal(timed[1][:, 0], 8.0 + np.arange(8.0))
        assert min(w[:, 0].min() for w in warm) >= (1 << 20)                   # warm-up seeds: disjoint from every rank's own


def test_bench_gpus_flag_launches_ranks_or_refuses(monkeypatch):
    """`bench.py --gpus N` without a launcher starts N rank processes itself (torch.distributed.run, 127.0.0.1) before touching the GPU;
    under a launcher with a different WORLD_SIZE it refuses instead of printing a line for the wrong N."""
    import subprocess
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd: calls.append(cmd) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--backend", "gloo"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus", "2", "--steps", "1", "--backend", "gloo"]
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2 and len(calls) == 1


def test_oracle_c_threaded_rti_batch_equals_the_python_driven_oracle():
    """bench.py's cpu_baseline runs the oracle's RTI step from C threads (so_rti_step_batch); it must be the same computation as
    OracleFastSLS.solve(rti_steps=1) driven from Python, instance by instance (bit-identical), for any thread count."""
    from oracle import oracle as O
    from problems import stack
    insts = [make_instance("pendulum", s, 0.5) for s in range(5)]
    m, d, N = insts[0].m, oracle_dims(insts[0]), insts[0].N
    g = np.stack([np.stack(i.g_list[:N]) for i in insts])
    gN = np.stack([i.g_list[N] for i in insts])
    ref = [run_oracle_fastsls(i, rti_steps=1, settings=O.default_settings()) for i in insts]
    for nt in (1, 3):
        P, ok, done = O.rti_step_batch(d, stack(insts, "A"), stack(insts, "B"), g, gN, stack(insts, "c"), stack(insts, "q"), stack(insts, "x0_arg"), m.G, m.Gf, m.gf,
                                       insts[0].E, m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, O.default_settings(), nthreads=nt)
        assert done == 5 and ok.all()
        for b in range(5):
            assert np.array_equal(P[b], ref[b]["primal_vec"])


@pytest.mark.parametrize("model", ["pendulum", "rocket"])
def test_closed_loop_oracle_back_ends_agree(model):
    """The closed-loop oracle can solve its QPs with the ADMM restatement (default) or with the dense interior point of tests/ref_ipm.py
    (problems.ipm_backend, used where the ADMM does not converge): on a fast-SLS RTI step both give the same primal, duals and back-offs."""
    from oracle import oracle as O
    from problems import ipm_backend
    inst = make_instance(model, 1, 0.5)
    m = inst.m
    r1 = run_oracle_fastsls(inst, 1)
    f = O.OracleFastSLS(oracle_dims(inst), m.G, m.Gf, m.g, m.gf, inst.E, m.Q, m.R, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, O.tight_settings())
    f.qp.backend = ipm_backend
    f.set_rti_steps(1)
    f.update_dynamics_list(inst.A, inst.B, inst.E, inst.g_list, inst.c)
    f.update_linear_cost(inst.q)
    r2 = f.solve(inst.x0_arg)
    assert r1["success"] and r2["success"]
    assert np.max(np.abs(r1["primal_vec"] - r2["primal_vec"])) < 1e-7 * max(1.0, np.abs(r1["primal_vec"]).max())
    assert np.max(np.abs(r1["dual_vec"] - r2["dual_vec"])) < 1e-6 * max(1.0, np.abs(r1["dual_vec"]).max())
    assert np.allclose(r1["backoff"], r2["backoff"], rtol=1e-7, atol=1e-10)


def test_bench_qp_statistics_are_over_the_solves_that_ran():
    """bench.py's per-QP figures: solves flagged at x0 (status 2) and solves an instance took no part in (status -1) are reported as
    fractions and left out of the iteration statistics."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    st = np.zeros((2, 4, 2, 8), dtype=np.int32)            # (steps, B, slot, 8)
    st[..., 1] = 5; st[..., 3] = 3; st[..., 5] = 2
    st[0, 1, 0] = [0, 0, 0, 0, 0, 0, 2, 0]; st[0, 1, 1] = [0, 0, 0, 0, 0, 0, -1, 0]      # instance 1, step 0: flagged, second QP not run
    st[1, 2, 1] = [0, 0, 0, 0, 0, 0, 2, 0]                                                # instance 2, step 1: tightened QP flagged
    out = bench.qp_statistics([st[0], st[1]])
    assert out["qp1"]["infeasible_x0_frac"] == 1 / 8 and out["qp1"]["not_run_frac"] == 0 and out["qp1"]["ran_frac"] == 7 / 8
    assert out["qp2"]["infeasible_x0_frac"] == 1 / 8 and out["qp2"]["not_run_frac"] == 1 / 8 and out["qp2"]["ran_frac"] == 6 / 8
    assert out["qp1"]["block_solves_mean"] == 5 and out["qp2"]["block_solves_mean"] == 5 and out["qp2"]["certified_frac"] == 1.0
    assert out["qp2"]["active_inequalities"]["histogram_per_step"]["counts"][0][1] == 3        # three solves of step 0 ran, 3 active rows each


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """slsqp_opts / slsqp_dims as the Python mirror declares them against the C header, field by field (gcc + offsetof): a field added on one side
    only would make slsqp_default_opts write past the Python object."""
    import ctypes as C
    import subprocess
    from robust_nonlinear_mpc_amd import _lib as L
    fields = {"slsqp_opts": [n for n, _ in L.Opts._fields_], "slsqp_dims": [n for n, _ in L.Dims._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "slsqp.h"', 'int main(void) {']
    for st, names in fields.items():
        src.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        src += [f'  printf("{st}.{n} %zu\\n", offsetof({st}, {n}));' for n in names]
    src += ['  printf("SLSQP_TIMING_LEN %d\\n", SLSQP_TIMING_LEN);', '  printf("SLSQP_KERNEL_TIMING_LEN %d\\n", SLSQP_KERNEL_TIMING_LEN);', '  printf("SLSQP_CL_RUN_STATS_LEN %d\\n", SLSQP_CL_RUN_STATS_LEN);']
    src += ['  return 0;', '}']
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(cfile)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(got["slsqp_opts"]) == C.sizeof(L.Opts) and int(got["slsqp_dims"]) == C.sizeof(L.Dims)
    # buffer lengths of the two timing queries (the library refuses shorter buffers; the mirror sizes its own from these)
    assert int(got["SLSQP_TIMING_LEN"]) == L.TIMING_LEN and int(got["SLSQP_KERNEL_TIMING_LEN"]) == L.KERNEL_TIMING_LEN and int(got["SLSQP_CL_RUN_STATS_LEN"]) == L.CL_RUN_STATS_LEN
    for cls, st in ((L.Opts, "slsqp_opts"), (L.Dims, "slsqp_dims")):
        for n in fields[st]:
            assert int(got[f"{st}.{n}"]) == getattr(cls, n).offset, (st, n)
    # and the header has no field the mirror lacks: count the members of the struct bodies
    hdr = open(os.path.join(ROOT, "include", "slsqp.h")).read()
    for st in fields:
        body = hdr[:hdr.index("} " + st + ";")]
        body = body[body.rindex("typedef struct"):]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        decl = [d for d in body.split(";") if re.search(r"\b(int|double)\b", d)]
        n_members = sum(len(d.split(",")) for d in decl)
        assert n_members == len(fields[st]), (st, n_members, len(fields[st]))


@pytest.mark.parametrize("B,steps,workers,keep", [(97, 50, 13, 1), (97, 50, 13, 0), (8, 40, 32, 1), (300, 20, 3, 1), (1, 30, 5, 1)])
def test_instance_queue_protocol_model_under_thread_sanitizer(tmp_path, B, steps, workers, keep):
    """The protocol of k_cl_loop's device-side instance FIFO (clq_pop / clq_push in csrc/slsqp_api.hip: tail / head tickets, a count of published items,
    workers that exit when the count is <= 0, an instance behind the mean keeps its worker) as a host model on std::atomic, one thread per wavefront,
    under ThreadSanitizer: every instance runs exactly `steps` steps in order, every hand-over carries the previous step's writes (release on push, acquire
    on pop), nothing is stranded when workers exit, no worker hangs -- with more workers than instances, fewer, and a single instance."""
    import subprocess
    exe = tmp_path / "queue_model"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", str(exe), os.path.join(ROOT, "tests", "queue_model.cpp")])
    r = subprocess.run([str(exe), str(B), str(steps), str(workers), str(keep)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok") and "ThreadSanitizer" not in r.stderr, (r.stdout, r.stderr[-2000:])
