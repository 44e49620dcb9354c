"""Pins csrc/dynamics.hpp (host instantiation) against values produced by the reference's own ODE / RK4 source
(tests/golden/dyn_*.npz): ode and ddyn to 1e-12, forward-mode AD Jacobians against central differences of the reference's
ddyn to 1e-6 (the differences themselves are only ~1e-8 accurate)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def dynlib():
    so = os.path.join(ROOT, "tests", "_build", "libdyn_host.so")
    src = os.path.join(ROOT, "tests", "dyn_host.cpp")
    hdr = os.path.join(ROOT, "robust-nonlinear-mpc_amd", "csrc", "dynamics.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("name,mid", [("pendulum", 0), ("quadrotor", 1), ("rocket", 2)])
def test_ode_ddyn_and_jacobians(dynlib, name, mid):
    g = dict(np.load(os.path.join(GOLDEN, f"dyn_{name}.npz")))
    nx, nu = int(g["dims"][0]), int(g["dims"][1])
    for i in range(g["X"].shape[0]):
        x, u = np.ascontiguousarray(g["X"][i]), np.ascontiguousarray(g["U"][i])
        o = np.zeros(nx)
        dynlib.dyn_ode(mid, _p(x), _p(u), _p(o))
        assert np.allclose(o, g["ode"][i], rtol=1e-12, atol=1e-12)
        dynlib.dyn_ddyn(mid, _p(x), _p(u), _p(o))
        assert np.allclose(o, g["ddyn"][i], rtol=1e-12, atol=1e-13)
        A, B, f = np.zeros((nx, nx)), np.zeros((nx, nu)), np.zeros(nx)
        dynlib.dyn_jac(mid, _p(x), _p(u), _p(A), _p(B), _p(f))
        assert np.allclose(f, g["ddyn"][i], rtol=1e-12, atol=1e-13)
        assert np.allclose(A, g["A_fd"][i], rtol=1e-6, atol=1e-7)
        assert np.allclose(B, g["B_fd"][i], rtol=1e-6, atol=1e-7)
