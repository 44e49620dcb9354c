"""Stand-in for the reference's generated OSQP extension module `osqp_generated` (built by QP.generate_c_code, solver/qp_jit.py:812-836,
imported by solver/fast_SLS_jit.py:16-35 and driven by QP._cg_push_updates :671-698 and QP.solve :404-485), on top of libslsqp_hip.so.

Same three module-level functions and the same module-level (singleton) state as the generated extension:

    update_data_mat(P_x=..., A_x=...) -> 0      CSC data of triu(2P) and of the (m x n) constraint matrix, frozen pattern of qp_jit.py:77-192
    update_data_vec(q, l, u)          -> 0
    solve() -> (x, y, status_code, iter, run_time)     status_code 0 = solved (qp_jit.py:460-462)

The problem size is inferred from the array lengths of the first push (the generated module has it compiled in).  To use it from the
reference:  `import robust_nonlinear_mpc_amd.osqp_generated as m; sys.modules["osqp_generated"] = m`  before importing solver.fast_SLS_jit,
or copy / symlink this file as `build/osqp_fast/osqp_generated.py`.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_DIMS = ((4, 1), (13, 4), (17, 4))       # (nx, nu) the HIP kernels are instantiated for
_state = dict(h=None, n=0, m=0, device=0, opts=None)


def _infer(n, m):
    for nx, nu in _DIMS:
        nz = nx + nu
        if (n - nx) % nz == 0:
            N = (n - nx) // nz
            if N >= 1 and m == N * (nx + 2 * nz) + 2 * nx + nx:
                return nx, nu, N
    raise RuntimeError(f"osqp_generated (HIP): no supported (nx, nu, N) has n={n}, m={m}")


def configure(n, m, device=0):
    """Create the solver for a QP with n variables and m constraint rows (called implicitly by the first update_data_vec)."""
    lib = L.load()
    if _state["h"] and (_state["n"], _state["m"]) == (n, m):
        return
    reset()
    nx, nu, N = _infer(n, m)
    d = L.Dims(nx, nu, nx, N, 2 * (nx + nu), 2 * nx)
    h = lib.slsqp_create(C.byref(d), 1, device)
    if not h:
        raise RuntimeError("slsqp_create: " + lib.slsqp_last_error().decode())
    o = L.Opts()
    lib.slsqp_default_opts(C.byref(o))
    o.warm_start = 0                       # the reference sets warm_starting=False (qp_jit.py:546)
    _state.update(h=h, n=n, m=m, device=device, opts=o, pending_mat=None)


def reset():
    if _state["h"]:
        L.load().slsqp_destroy(_state["h"])
    _state.update(h=None, n=0, m=0)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def update_data_mat(P_x=None, A_x=None):
    P_x = None if P_x is None else np.ascontiguousarray(P_x, dtype=np.float64)
    A_x = None if A_x is None else np.ascontiguousarray(A_x, dtype=np.float64)
    if not _state["h"]:
        _state["pending_mat"] = (P_x, A_x)          # sizes are only known once q, l, u arrive
        return 0
    return int(L.load().slsqp_qp_update_data_mat(_state["h"], _p(P_x), _p(A_x), L.HOST))


def update_data_vec(q=None, l=None, u=None):
    q, l, u = (None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in (q, l, u))
    if not _state["h"]:
        if q is None or l is None:
            raise RuntimeError("osqp_generated (HIP): the first update_data_vec must carry q, l and u")
        pend = _state.get("pending_mat")
        configure(q.size, l.size)
        if pend is not None:
            rc = update_data_mat(*pend)
            if rc:
                return rc
    return int(L.load().slsqp_qp_update_data_vec(_state["h"], _p(q), _p(l), _p(u), L.HOST))


def solve():
    lib = L.load()
    n, m = _state["n"], _state["m"]
    x, y = np.empty(n), np.empty(m)
    st, it = C.c_int(), C.c_int()
    rc = lib.slsqp_qp_solve(_state["h"], _p(x), _p(y), C.byref(st), C.byref(it), L.HOST, C.byref(_state["opts"]))
    if rc:
        raise RuntimeError("slsqp_qp_solve: " + lib.slsqp_last_error().decode())
    ms = (C.c_double * L.TIMING_LEN)()      # total, qp, sweep, other, jac; the library refuses a shorter buffer
    L.check(lib.slsqp_last_timing(_state["h"], ms, L.TIMING_LEN))
    return x, y, (0 if st.value in (0, 4) else st.value), it.value, ms[1] * 1e-3
