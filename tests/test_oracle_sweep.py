"""Pins the CPU oracle's SLS sweep against golden vectors produced by the reference's own kernels
(_backward_solve_numba / _propagate / _backoff_from_phi, solver/fast_SLS_jit.py:65-188, run as plain
NumPy by tests/golden/gen_golden.py).  Tolerance: 1e-11 relative (same formulas, different summation order)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O

CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "sweep_*.npz")))


def relerr(a, b):
    return np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))


@pytest.mark.parametrize("case", CASES)
def test_sweep_matches_reference_kernels(case):
    g = dict(np.load(os.path.join(GOLDEN, case)))
    d = O.dims_of(int(g["nx"]), int(g["nu"]), int(g["nw"]), int(g["N"]), int(g["ni"]), int(g["ni_f"]))
    S, K = O.backward(d, g["A"], g["B"], g["G"], g["Gf"], g["eta"], g["eta_f"], g["Q_reg"], g["R_reg"], g["Q_reg_f"])
    assert relerr(K, g["K"]) < 1e-10
    Px, Pu = O.propagate(d, g["A"], g["B"], g["E"], K)
    beta, beta_f, bo, bof = O.backoff(d, Px, Pu, g["G"], g["Gf"], 1e-10)
    assert relerr(beta, g["beta"]) < 1e-9
    assert relerr(beta_f, g["beta_f"]) < 1e-9
    assert relerr(bo, g["backoff"]) < 1e-10
    assert relerr(bof, g["backoff_f"]) < 1e-10
    if "S" in g:
        assert relerr(S, g["S"]) < 1e-11
        assert relerr(Px, g["Phi_x"]) < 1e-11
        assert relerr(Pu, g["Phi_u"]) < 1e-11
    else:
        nx, nw = d.nx, d.nw
        w = np.cos(np.arange(nx * nx)).reshape(nx, nx)
        assert np.allclose(np.linalg.norm(S, axis=(2, 3)), g["S_fro"], rtol=1e-10, atol=0)
        assert np.allclose(np.einsum("kjab,ab->kj", S, w), g["S_chk"], rtol=1e-9, atol=1e-9 * np.abs(g["S_chk"]).max())
        assert np.allclose(np.linalg.norm(Px, axis=(2, 3)), g["Phix_fro"], rtol=1e-10, atol=1e-300)
        assert np.allclose(np.linalg.norm(Pu, axis=(2, 3)), g["Phiu_fro"], rtol=1e-9, atol=1e-300)


def test_riccati_step_known_answer():
    """OCP.riccati_step (solver/ocp.py:103-109) iterated 6 times == column 0 of the sweep with eta = 0."""
    g = dict(np.load(os.path.join(GOLDEN, "riccati_lq.npz")))
    nx, nu = g["B"].shape
    N = g["K"].shape[0]
    d = O.dims_of(nx, nu, nx, N, 2, 2)
    A = np.stack([g["A"]] * N)
    B = np.stack([g["B"]] * N)
    G = np.zeros((2, nx + nu))
    Gf = np.zeros((2, nx))
    S, K = O.backward(d, A, B, G, Gf, np.zeros((N, N, 2)), np.zeros((N + 1, 2)), g["Cx"], g["Cu"], g["S0"])
    for t in range(N):  # golden step t is stage N-1-t
        assert np.allclose(K[N - 1 - t, 0], g["K"][t], rtol=1e-10, atol=1e-12)
        # the reference's njit kernel symmetrises S, ocp.riccati_step does not: compare the symmetric part
        assert np.allclose(S[N - 1 - t, 0], 0.5 * (g["S"][t] + g["S"][t].T), rtol=1e-8, atol=1e-10)
