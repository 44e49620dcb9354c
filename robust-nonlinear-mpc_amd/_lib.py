"""ctypes loader of csrc/libslsqp_hip.so (the C-ABI of include/slsqp.h).  No fallback: if the HIP library is
missing or no GPU is visible the product path fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SLSQP_SO") or os.path.join(_HERE, "csrc", "libslsqp_hip.so")   # SLSQP_SO: experiment builds only
HOST, DEVICE = 0, 1
TIMING_LEN, KERNEL_TIMING_LEN, CL_RUN_STATS_LEN = 5, 8, 4      # SLSQP_TIMING_LEN / SLSQP_KERNEL_TIMING_LEN / SLSQP_CL_RUN_STATS_LEN of include/slsqp.h


class Dims(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("nx", "nu", "nw", "N", "ni", "ni_f")]


class Opts(C.Structure):
    _fields_ = [("rti_steps", C.c_int), ("max_sls_iter", C.c_int), ("qp_max_iter", C.c_int), ("qp_eps", C.c_double),
                ("conv_tol", C.c_double), ("eps_backoff", C.c_double), ("want_K", C.c_int), ("warm_start", C.c_int), ("warm_rounds", C.c_int),
                ("max_scp_iter", C.c_int), ("scp_eps", C.c_double), ("precision", C.c_int), ("as_first", C.c_int), ("as_rounds", C.c_int), ("as_max_viol", C.c_int), ("ipm_restart", C.c_int), ("time_kernels", C.c_int), ("as_warm_max_set", C.c_int), ("as_warm_last", C.c_int), ("fuse_rti", C.c_int), ("cl_persistent", C.c_int)]


EXPORTS = [
    "slsqp_default_opts", "slsqp_last_error", "slsqp_version", "slsqp_create", "slsqp_destroy", "slsqp_set_costs",
    "slsqp_set_constraints", "slsqp_update_dynamics", "slsqp_update_linear_cost", "slsqp_solve", "slsqp_get", "slsqp_reset",
    "slsqp_sync", "slsqp_qp_nnz", "slsqp_qp_update_data_mat", "slsqp_qp_update_data_vec", "slsqp_qp_solve", "slsqp_sweep",
    "slsqp_last_timing", "slsqp_kernel_timing", "slsqp_stream", "slsqp_set_model", "slsqp_set_E", "slsqp_linearize", "slsqp_cl_init", "slsqp_cl_step", "slsqp_nominal_solve", "slsqp_set", "slsqp_cl_log", "slsqp_selftest", "slsqp_result_bytes", "slsqp_cl_run", "slsqp_cl_run_stats",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                           "There is no CPU fallback.")
    try:
        # Load torch's HIP runtime first when torch is present: two different libamdhip64 copies in one process
        # (torch's bundled one and /opt/rocm's) do not mix, and the first one loaded wins.
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(SO_PATH)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p
    lib.slsqp_last_error.restype = C.c_char_p
    lib.slsqp_version.restype = C.c_char_p
    lib.slsqp_create.restype = vp
    lib.slsqp_create.argtypes = [C.POINTER(Dims), C.c_int, C.c_int]
    lib.slsqp_destroy.argtypes = [vp]
    lib.slsqp_destroy.restype = None
    lib.slsqp_default_opts.argtypes = [C.POINTER(Opts)]
    lib.slsqp_default_opts.restype = None
    lib.slsqp_set_costs.argtypes = [vp] + [dp] * 6
    lib.slsqp_set_constraints.argtypes = [vp] + [dp] * 3
    lib.slsqp_update_dynamics.argtypes = [vp] + [dp] * 6 + [C.c_int]
    lib.slsqp_update_linear_cost.argtypes = [vp, dp, C.c_int]
    lib.slsqp_solve.argtypes = [vp, dp, C.c_int, C.POINTER(Opts)]
    lib.slsqp_get.argtypes = [vp, C.c_char_p, vp, C.c_int]
    lib.slsqp_result_bytes.argtypes = [vp, C.c_char_p]
    lib.slsqp_result_bytes.restype = C.c_longlong
    lib.slsqp_set.argtypes = [vp, C.c_char_p, vp, C.c_int]
    lib.slsqp_reset.argtypes = [vp]
    lib.slsqp_sync.argtypes = [vp]
    lib.slsqp_qp_nnz.argtypes = [C.POINTER(Dims)] + [C.POINTER(C.c_int)] * 4
    lib.slsqp_qp_update_data_mat.argtypes = [vp, dp, dp, C.c_int]
    lib.slsqp_qp_update_data_vec.argtypes = [vp, dp, dp, dp, C.c_int]
    lib.slsqp_qp_solve.argtypes = [vp, dp, dp, ip, ip, C.c_int, C.POINTER(Opts)]
    lib.slsqp_sweep.argtypes = [vp] + [dp] * 7 + [C.c_int]
    lib.slsqp_last_timing.argtypes = [vp, dp, C.c_int]
    lib.slsqp_kernel_timing.argtypes = [vp, dp, C.c_int]
    lib.slsqp_set_model.argtypes = [vp, C.c_int, dp]
    lib.slsqp_set_E.argtypes = [vp, dp, C.c_int]
    lib.slsqp_linearize.argtypes = [vp, dp, dp, C.c_int]
    lib.slsqp_cl_init.argtypes = [vp, dp, dp, dp, dp, C.c_int]
    lib.slsqp_cl_step.argtypes = [vp, C.c_int, dp, C.c_int, C.POINTER(Opts)]
    lib.slsqp_nominal_solve.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.POINTER(Opts)]
    lib.slsqp_cl_log.argtypes = [vp, C.c_int]
    lib.slsqp_cl_run.argtypes = [vp, C.c_int, dp, C.c_int, C.POINTER(Opts), C.c_double, C.c_double, C.POINTER(C.c_int)]
    lib.slsqp_cl_run_stats.argtypes = [vp, dp, C.c_int]
    lib.slsqp_selftest.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int]
    lib.slsqp_stream.argtypes = [vp]
    lib.slsqp_stream.restype = vp
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("slsqp: " + load().slsqp_last_error().decode())
