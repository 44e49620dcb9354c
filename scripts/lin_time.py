"""Time of the batched linearisation (slsqp_linearize: RK4 + forward-mode AD Jacobians, c, g, q, bounds) for a full rocket batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from robust_nonlinear_mpc_amd import BatchedFastSLS, get_model, _lib as L
for model, N, B in (("rocket", 20, 4096), ("quadrotor", 20, 2048), ("pendulum", 10, 1024)):
    m = get_model(model)
    f = BatchedFastSLS(N, m.Q, m.R, m, m.Qf, m.Q_reg, m.R_reg, m.Q_reg_f, batch=B)
    rng = np.random.default_rng(0)
    X = np.tile(m.x_ref, (B, N + 1, 1)) + 0.02 * rng.standard_normal((B, N + 1, m.nx))
    U = np.tile(m.u_ref, (B, N, 1)) + 0.02 * rng.standard_normal((B, N, m.nu))
    Xd, Ud = torch.from_numpy(X).cuda(), torch.from_numpy(U).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(2):
        L.check(f.lib.slsqp_linearize(f.h, p(Xd), p(Ud), L.DEVICE))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        L.check(f.lib.slsqp_linearize(f.h, p(Xd), p(Ud), L.DEVICE))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{model} N={N} B={B}: linearise {1e3*dt:.3f} ms per batch ({B*N/dt/1e6:.1f} M stage linearisations/s)")
    f.close()
