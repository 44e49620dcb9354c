"""Problem data of the three plants the reference ships (dims, box constraints, disturbance matrix E,
weights), restated as plain numbers.  Citations are relative to /root/reference.

Only what the fast-SLS QP path consumes is here ("problem-data source").
The ODEs live in csrc/dynamics.hpp (templated C++, host + device).
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class ModelData:
    name: str
    nx: int
    nu: int
    nw: int
    x_ub: np.ndarray
    x_lb: np.ndarray
    u_ub: np.ndarray
    u_lb: np.ndarray
    E: np.ndarray            # (nx, nw) disturbance matrix as the closed-loop script sets it
    Q: np.ndarray
    R: np.ndarray
    Qf: np.ndarray
    Q_reg: np.ndarray
    R_reg: np.ndarray
    Q_reg_f: np.ndarray
    x_ref: np.ndarray        # neutral state the cost is centred on (deviation coordinates: zero)
    u_ref: np.ndarray
    rti: int                 # SCP iterations per MPC step in the script
    fast_sls_rti_steps: int
    model_id: int            # id understood by csrc/dynamics.hpp
    extra: dict = field(default_factory=dict)

    @property
    def nz(self):
        return self.nx + self.nu

    @property
    def ni(self):
        return 2 * self.nz

    @property
    def ni_f(self):
        return 2 * self.nx

    # G = [I; -I], g = [ub; -lb]  (dyn/pendulum.py:12-21, dyn/quadrotor.py:93-99, dyn/rocket.py:138-147)
    @property
    def G(self):
        return np.vstack([np.eye(self.nz), -np.eye(self.nz)])

    @property
    def Gf(self):
        return np.vstack([np.eye(self.nx), -np.eye(self.nx)])

    @property
    def g(self):
        return np.concatenate([self.x_ub, self.u_ub, -self.x_lb, -self.u_lb])

    @property
    def gf(self):
        return np.concatenate([self.x_ub, -self.x_lb])

    def n_var(self, N):
        return self.nz * N + self.nx

    def m_con(self, N):
        return N * (self.nx + self.ni) + self.ni_f + self.nx


def pendulum():
    """dyn/pendulum.py:8-24 with the overrides of expe/main_pendulum_robust_closed_loop.py:24-48."""
    nx, nu = 4, 1
    return ModelData(
        name="pendulum", nx=nx, nu=nu, nw=4,
        x_ub=10.0 * np.ones(nx), x_lb=-10.0 * np.ones(nx), u_ub=5.0 * np.ones(nu), u_lb=-5.0 * np.ones(nu),
        E=0.003 * np.eye(nx), Q=np.eye(nx), R=np.eye(nu), Qf=10.0 * np.eye(nx),
        Q_reg=1e3 * np.eye(nx), R_reg=1e3 * np.eye(nu), Q_reg_f=1e4 * np.eye(nx),
        x_ref=np.zeros(nx), u_ref=np.zeros(nu), rti=3, fast_sls_rti_steps=2, model_id=0,
        extra=dict(x0=np.array([0.5, 0.5, 0.0, 0.0]), sim_steps=60),
    )


def quadrotor():
    """dyn/quadrotor.py:31-106 with the overrides of expe/main_quadrotor_robust_closed_loop.py:35-69."""
    nx, nu = 13, 4
    x_ub = np.array([20.0] * 3 + [10.0] * 3 + [1.5] * 4 + [20.0] * 3)
    st = np.deg2rad(2.0)
    qv = 0.5 * st
    qw = 0.1 * qv
    Q = np.diag([10.0] * 3 + [1.0] * 3 + [1.0] * 4 + [2.0] * 3)
    f_hover = 1.0 * 9.81 / 4.0
    return ModelData(
        name="quadrotor", nx=nx, nu=nu, nw=nx,
        x_ub=x_ub, x_lb=-x_ub, u_ub=20.0 * np.ones(nu), u_lb=np.zeros(nu),
        E=0.05 * 5.0 * np.diag([0.10] * 3 + [0.15] * 3 + [qw, qv, qv, qv] + [0.2] * 3),
        Q=Q, R=np.eye(nu), Qf=10.0 * Q,
        Q_reg=1e4 * np.eye(nx), R_reg=1e4 * np.eye(nu), Q_reg_f=1e4 * np.eye(nx),
        x_ref=np.array([0, 0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0]), u_ref=f_hover * np.ones(nu),
        rti=3, fast_sls_rti_steps=2, model_id=1, extra=dict(sim_steps=30),
    )


def rocket():
    """dyn/rocket.py:22-163 with the overrides of expe/main_rocket_robust_closed_loop.py:32-126."""
    nx, nu = 17, 4
    x_ub = np.array([10.0] * 3 + [1.0] * 3 + [1.5] * 4 + [2.0] * 3 + [50.0, 2.0, 1.0, 1.0])
    u_ub = np.array([50.0, 2.0, 1.0, 1.0])
    st = np.deg2rad(2.0)
    qv = 0.5 * st
    qw = 0.1 * qv
    Q = np.diag([10.0] * 3 + [1.0] * 8 + [5.0, 5.0] + [1.0] * 4)
    x0 = np.array([1.75729, 4.15951, 4.72757, -0.18913, -0.38367, -0.08697, -0.79487, 0.00768, -0.21110,
                   -0.56883, -0.12752, -0.58026, -0.76542, 0.20555, 0.54610, -0.40116, -0.35401])
    return ModelData(
        name="rocket", nx=nx, nu=nu, nw=nx,
        x_ub=x_ub, x_lb=-x_ub, u_ub=u_ub, u_lb=-u_ub,
        E=0.05 * np.diag([0.20] * 6 + [qv, qv, qv, qw] + [0.2] * 3 + [0.8, 0.2, 0.04, 0.04]),
        Q=Q, R=np.eye(nu), Qf=10.0 * Q,
        Q_reg=1e4 * np.eye(nx), R_reg=1e4 * np.eye(nu), Q_reg_f=1e4 * np.eye(nx),
        x_ref=np.array([0, 0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]), u_ref=np.zeros(nu),
        rti=1, fast_sls_rti_steps=1, model_id=2, extra=dict(x0=x0, sim_steps=30),
    )


_MODELS = {"pendulum": pendulum, "quadrotor": quadrotor, "rocket": rocket, "rockETH": rocket}


def get_model(name):
    return _MODELS[name]()
