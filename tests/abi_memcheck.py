"""Run by test_python_mirror_under_debug_allocators in a child process with MALLOC_CHECK_=3 and PYTHONMALLOC=malloc_debug: every entry point of the
Python mirror that hands a host buffer to the C ABI is exercised once, so that a buffer smaller than what the library writes aborts the child (a
four-element buffer for slsqp_last_timing's five values once corrupted the heap of the whole test process, intermittently and far from its cause)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robust_nonlinear_mpc_amd import BatchedFastSLS, ClosedLoopMPC, disturbance_stream, fast_SLS, get_model, make_batch
import robust_nonlinear_mpc_amd.osqp_generated as cg
GOLDEN = os.path.join(ROOT, "tests", "golden")

# 1. batched mirror: solve, every result array, timings, QP-level and sweep-level boundaries
batch = make_batch("pendulum", os.path.join(GOLDEN, "sweep_pendulum_N10_s0.npz"), 4, seed=1)
m, N = batch["model"], batch["N"]
f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=4)
f.update_dynamics_list(batch["A"], batch["B"], batch["E"], batch["g"], batch["gN"], batch["c"])
f.update_linear_cost(batch["q"])
f.opts.time_kernels = 1
out = f.solve(batch["x0_arg"])
f.timing_ms(); f.kernel_timing()
# the timing queries take the buffer's length and refuse a short one (nothing written) instead of overrunning it
import ctypes as C
short = (C.c_double * 8)(*([-7.0] * 8))
assert f.lib.slsqp_last_timing(f.h, short, 4) != 0 and b"SLSQP_TIMING_LEN" in f.lib.slsqp_last_error() and list(short) == [-7.0] * 8
assert f.lib.slsqp_kernel_timing(f.h, short, 7) != 0 and b"SLSQP_KERNEL_TIMING_LEN" in f.lib.slsqp_last_error() and list(short) == [-7.0] * 8
assert f.lib.slsqp_last_timing(f.h, short, 8) == 0 and short[5] == -7.0      # a longer buffer is fine: five values written
for name, shape, dt in (("eta", (N, N, m.ni), np.float64), ("eta_f", (N + 1, m.ni_f), np.float64), ("K", (N, N + 1, m.nu, m.nx), np.float64), ("kkt", (8,), np.float64),
                        ("qp_stats", (2, 8), np.int32), ("status", (), np.int32), ("pin_dual", (m.nx,), np.float64), ("ubg", (f.mb,), np.float64)):
    f.get(name, shape, dt)
ub, lb = f.get("ubg", (f.mb,)), f.get("lbg", (f.mb,))
x0 = batch["x0_arg"]
f.qp_update_data_vec(batch["q"], np.concatenate([lb, -x0 - 1e-10], axis=1), np.concatenate([ub, -x0 + 1e-10], axis=1))
x, y, st, it, _ = f.qp_solve()
f.sweep(f.get("eta", (N, N, m.ni)), f.get("eta_f", (N + 1, m.ni_f)))
l1 = np.concatenate([lb[0], -x0[0] - 1e-10]); u1 = np.concatenate([ub[0], -x0[0] + 1e-10]); q1 = batch["q"][0].copy()
f.close()

# 2. the osqp_generated stand-in (module-level singleton): matrices in the reference's frozen CSC order, vectors, solve (which reads the timings)
import scipy.sparse as sp
nx, nu, nz, ni, nif = m.nx, m.nu, m.nx + m.nu, m.ni, m.ni_f
nvar = nz * N + nx
A0, B0 = batch["A"][0], batch["B"][0]
rows = []
for k in range(N):
    r = np.zeros((nx, nvar)); r[:, k * nz:k * nz + nx] = A0[k]; r[:, k * nz + nx:(k + 1) * nz] = B0[k]; r[:, (k + 1) * nz:(k + 1) * nz + nx] = -np.eye(nx)
    gk = np.zeros((ni, nvar)); gk[:, k * nz:(k + 1) * nz] = m.G
    rows += [r, gk]
gfm = np.zeros((nif, nvar)); gfm[:, N * nz:] = m.Gf
pin = np.zeros((nx, nvar)); pin[:, :nx] = np.eye(nx)
Amat = np.vstack(rows + [gfm, pin])
mask = (Amat != 0)
for k in range(N):
    mask[k * (nx + ni):k * (nx + ni) + nx, k * nz:(k + 1) * nz] = True
A_csc = sp.csc_matrix((Amat[mask.nonzero()], mask.nonzero()), shape=Amat.shape); A_csc.sort_indices()
Hd = np.concatenate([np.concatenate([np.diag(m.Q), np.diag(m.R)])] * N + [np.diag(m.Qf)])
cg.reset()
assert cg.update_data_mat(P_x=sp.triu(sp.diags(2.0 * Hd), format="csc").data, A_x=A_csc.data) == 0
assert cg.update_data_vec(q=q1, l=np.maximum(l1, -1e20), u=np.minimum(u1, 1e20)) == 0
xg, yg, code, iters, rt = cg.solve()
assert code == 0 and xg.shape == (nvar,) and yg.shape == (Amat.shape[0],)
cg.reset()

# 3. single-instance view with the reference's list arguments
s1 = fast_SLS(N, m.Q, m.R, m, m.Qf)
s1.update_dynamics_list([a for a in batch["A"][0]], [b for b in batch["B"][0]], [e for e in batch["E"]], [g for g in batch["g"][0]] + [batch["gN"][0]],
                        [c for c in batch["c"][0]])
s1.update_linear_cost(batch["q"][0])
s1.solve(batch["x0_arg"][0])
s1.close()

# 4. closed loop: nominal initialiser, steps with fetch, device-side log, npz
steps = 3
cl = ClosedLoopMPC(m, 10, 4)
cl.reset(np.tile(m.extra["x0"], (4, 1)), solve_nominal=True)
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(4)], axis=1)
for i in range(steps):
    cl.step(W[i])
res = cl.run(np.tile(m.extra["x0"], (4, 1)), steps, W)
res_dev = cl.run_on_device(np.tile(m.extra["x0"], (4, 1)), steps, W)
# the whole loop as one persistent launch (slsqp_cl_run: rti = 1 loops only) and its statistics; a short statistics buffer is refused untouched
cl.rti = 1
cl.f.set_rti_steps(1)
res_run = cl.run_decoupled(np.tile(m.extra["x0"], (4, 1)), steps, W)
assert res_run["rounds"] == 1 and res_run["loop_stats"]["mpc_steps"] == 4 * steps
short = (C.c_double * 8)(*([-7.0] * 8))
assert cl.f.lib.slsqp_cl_run_stats(cl.f.h, short, 3) != 0 and b"SLSQP_CL_RUN_STATS_LEN" in cl.f.lib.slsqp_last_error() and list(short) == [-7.0] * 8
cl.save_npz(os.path.join(os.environ.get("TMPDIR", "/tmp"), "abi_memcheck.npz"), res, 0)
cl.close()
print("abi_memcheck ok")
