"""CPU restatement of the three plants' ODEs, their RK4 map and its Jacobians -- TEST INFRASTRUCTURE ONLY.

Independent of the product's csrc/dynamics.hpp (different language, different differentiation method), so that the closed-loop
oracle in tests/problems.py no longer shares the plant with the code under test.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file.

What it follows (citations relative to antoineleeman/robust-nonlinear-mpc):
  pendulum_ode    dyn/pendulum.py:26-44        (cart-pole)
  quadrotor_ode   dyn/quadrotor.py:108-174     (rigid body, quaternion [w,x,y,z], X-configuration rotor moments)
  rocket_ode      dyn/rocket.py:165-242        (rockETH: gimballed thrust, first-order actuator lags)
  gimbal_angle    dyn/rocket.py:244-254        (four-bar linkage servo -> gimbal angle)
  ddyn            dyn/model.py:15-34           (classical RK4, h = 0.05 regardless of m.dt: SURVEY quirk q8)
Pinned by tests/test_dynamics_cpu.py against tests/golden/dyn_*.npz, which tests/golden/gen_golden.py produced by running the
reference's own ode/ddyn source (near neutral, along a roll-out from the rocket script's x0, at saturated servo angles).

Jacobians: the reference takes CasADi's exact derivatives (solver/SCP_SLS_jit.py:190-228), which cannot be run here.  The ODEs
are analytic (sin, cos, atan, sqrt, rational), so this oracle differentiates by the complex-step method, exact to round-off
(no subtractive cancellation): d f / d x_i = Im f(x + i h e_i) / h with h = 1e-30.
"""
import numpy as np

H_RK4 = 0.05

_ROCKET = dict(mass=1.16, grav=9.81, Jxx=0.00210, Jyy=0.1, Jzz=0.1, cog=0.42, tau_thrust=0.06, tau_servo=0.10,
               ga=5.0, gb=35.2, gc=33.0, gd=28.0, ge=35.2, hover=11.3796)
_QUAD = dict(m=1.0, g=9.81, l=0.15, Jx=0.02, Jy=0.02, Jz=0.04, kM=0.01)


def _rot_and_qdot(q, w):
    """Body->world rotation of the unit-norm-agnostic quaternion formula both rigid bodies use, and q_dot = 0.5 Omega(w) q."""
    qw, qx, qy, qz = q
    R = np.array([[1 - 2 * qy * qy - 2 * qz * qz, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw],
                  [2 * qx * qy + 2 * qz * qw, 1 - 2 * qx * qx - 2 * qz * qz, 2 * qy * qz - 2 * qx * qw],
                  [2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx * qx - 2 * qy * qy]])
    wx, wy, wz = w
    qd = 0.5 * np.array([-wx * qx - wy * qy - wz * qz,
                         wx * qw + wz * qy - wy * qz,
                         wy * qw - wz * qx + wx * qz,
                         wz * qw + wy * qx - wx * qy])
    return R, qd


def pendulum_ode(x, u):
    m1, m2, l, g = 1.0, 0.1, 0.5, 9.81
    th, thd = x[2], x[3]
    s, c = np.sin(th), np.cos(th)
    den = m1 + m2 * (1 - c * c)
    xdd = (u[0] + m2 * l * thd * thd * s - m2 * g * s * c) / den
    thdd = (-u[0] * c - m2 * l * thd * thd * s * c + (m1 + m2) * g * s) / (l * den)
    return np.array([x[1], xdd, thd, thdd])


def quadrotor_ode(x, u):
    p = _QUAD
    R, qd = _rot_and_qdot(x[6:10], x[10:13])
    Fz = u[0] + u[1] + u[2] + u[3]
    a = R[:, 2] * (Fz / p["m"])
    a = np.array([a[0], a[1], a[2] - p["g"]])
    tau = np.array([p["l"] * (u[1] - u[3]), p["l"] * (u[2] - u[0]), p["kM"] * (u[0] - u[1] + u[2] - u[3])])
    J = np.array([p["Jx"], p["Jy"], p["Jz"]])
    w = x[10:13]
    wd = (tau - np.cross(w, J * w)) / J
    return np.concatenate([x[3:6], a, qd, wd])


def gimbal_angle(servo, tilt):
    p = _ROCKET
    iv1 = p["gd"] + p["ga"] * np.cos(servo)
    iv2 = p["ge"] - p["ga"] * np.sin(servo)
    uu = p["gb"] ** 2 - p["gc"] ** 2 - iv1 * iv1 - iv2 * iv2
    vv = 2 * p["gc"] * np.cos(tilt) * iv2
    ww = -2 * p["gc"] * iv1
    return 2 * np.arctan((vv - np.sqrt(ww * ww + vv * vv - uu * uu)) / (uu + ww))


def rocket_ode(x, u):
    p = _ROCKET
    R, qd = _rot_and_qdot(x[6:10], x[10:13])
    thrust = x[13] + p["hover"]
    g1 = gimbal_angle(x[15], 0.0)
    g2 = gimbal_angle(x[16], g1)
    Fb = np.array([-thrust * np.sin(g1) * np.cos(g2), thrust * np.sin(g2), thrust * np.cos(g1) * np.cos(g2)])
    acc = (R @ Fb) / p["mass"]
    acc = np.array([acc[0], acc[1], acc[2] - p["grav"]])
    J = np.array([p["Jxx"], p["Jyy"], p["Jzz"]])
    w = x[10:13]
    arm = np.array([0.0, 0.0, -p["cog"]])
    wd = (np.cross(arm, Fb) - np.cross(w, J * w)) / J
    lag = np.array([(u[0] + p["hover"] - thrust) / p["tau_thrust"], (u[1] - x[14]) / p["tau_thrust"],
                    (u[2] - x[15]) / p["tau_servo"], (u[3] - x[16]) / p["tau_servo"]])
    return np.concatenate([x[3:6], acc, qd, wd, lag])


ODES = {0: pendulum_ode, 1: quadrotor_ode, 2: rocket_ode, "pendulum": pendulum_ode, "quadrotor": quadrotor_ode, "rocket": rocket_ode}


def ode(model, x, u):
    return ODES[model](np.asarray(x), np.asarray(u))


def ddyn(model, x, u, h=H_RK4):
    f = ODES[model]
    x, u = np.asarray(x), np.asarray(u)
    k1 = f(x, u)
    k2 = f(x + 0.5 * h * k1, u)
    k3 = f(x + 0.5 * h * k2, u)
    k4 = f(x + h * k3, u)
    return x + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)


def jac(model, x, u, h=H_RK4):
    """A = d ddyn / d x, B = d ddyn / d u (complex step), f = ddyn(x, u)."""
    x, u = np.asarray(x, dtype=float), np.asarray(u, dtype=float)
    nx, nu = x.size, u.size
    A, B = np.zeros((nx, nx)), np.zeros((nx, nu))
    eps = 1e-30
    for i in range(nx):
        xc = x.astype(complex)
        xc[i] += 1j * eps
        A[:, i] = ddyn(model, xc, u.astype(complex), h).imag / eps
    for i in range(nu):
        uc = u.astype(complex)
        uc[i] += 1j * eps
        B[:, i] = ddyn(model, x.astype(complex), uc, h).imag / eps
    return A, B, ddyn(model, x, u, h)
