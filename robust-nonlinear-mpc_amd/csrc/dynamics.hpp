// dynamics.hpp -- the three plants of the reference as templated C++ (host + device), RK4 discretisation and
// forward-mode AD, used by the batched linearisation kernel (the step that feeds the fast-SLS path).
//
// Restated from (citations relative to antoineleeman/robust-nonlinear-mpc):
//   Pendulum.ode   dyn/pendulum.py:26-44      (cart-pole, m1=1, m2=0.1, l=0.5, g=9.81)
//   Quadrotor.ode  dyn/quadrotor.py:108-174   (m=1, g=9.81, l=0.15, J=diag(0.02,0.02,0.04), kM=0.01)
//   Rocket.ode     dyn/rocket.py:165-254      (parameters :24-38; gimbal linkage `compute_gimbal_angle` :246-254)
//   Model.ddyn     dyn/model.py:15-34         (RK4, h = 0.05 always: SURVEY quirk q8)
// Pinned by tests/golden/dyn_*.npz (values of ode/ddyn produced by the reference's own source; Jacobians by central
// differences of the reference's ddyn).
#pragma once
#include <math.h>
#ifndef __HIPCC__
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif
#define DYN_HD __host__ __device__ inline

namespace dyn {

// ---- scalar types: double, or value + one directional derivative -------------------------------------------------
struct Dual {
    double v, d;
    DYN_HD Dual() : v(0.0), d(0.0) {}
    DYN_HD Dual(double a) : v(a), d(0.0) {}
    DYN_HD Dual(double a, double b) : v(a), d(b) {}
};
DYN_HD Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
DYN_HD Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
DYN_HD Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
DYN_HD Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
DYN_HD Dual operator/(Dual a, Dual b) { const double q = a.v / b.v; return Dual(q, (a.d - q * b.d) / b.v); }
DYN_HD Dual msin(Dual a) { return Dual(sin(a.v), cos(a.v) * a.d); }
DYN_HD Dual mcos(Dual a) { return Dual(cos(a.v), -sin(a.v) * a.d); }
DYN_HD Dual msqrt(Dual a) { const double s = sqrt(a.v); return Dual(s, 0.5 * a.d / s); }
DYN_HD Dual matan(Dual a) { return Dual(atan(a.v), a.d / (1.0 + a.v * a.v)); }
// sine and cosine of one argument together: one range reduction instead of two (double), or instead of four (Dual: msin and mcos
// each need both)
DYN_HD void msincos(double a, double &s, double &c) {
#ifdef __HIP_DEVICE_COMPILE__
    sincos(a, &s, &c);
#else
    s = sin(a); c = cos(a);
#endif
}
DYN_HD void msincos(Dual a, Dual &s, Dual &c) {
    double sv, cv;
    msincos(a.v, sv, cv);
    s = Dual(sv, cv * a.d); c = Dual(cv, -sv * a.d);
}
DYN_HD double msin(double a) { return sin(a); }
DYN_HD double mcos(double a) { return cos(a); }
DYN_HD double msqrt(double a) { return sqrt(a); }
DYN_HD double matan(double a) { return atan(a); }
DYN_HD double val(double a) { return a; }
DYN_HD double val(Dual a) { return a.v; }

// ---- how an ODE evaluation gets its transcendental functions ------------------------------------------------------
// MathPlain computes them.  The linearisation evaluates the ODE at the same four RK4 points once in plain doubles (ddyn_stages) and then once
// per direction in dual numbers (ddyn_tangent: 21 directions for the rocket): the sines, cosines, arctangents and square roots of the primal
// are the same every time, and they are most of the instructions (and of the registers) of an evaluation.  MathRecord (doubles) computes them and
// keeps them on a tape, in call order; MathReplay (dual numbers) reads the primal values back from the tape and only forms the derivative
// parts.  The tape entries per evaluation: Dims<MODEL>::NT.
struct MathPlain {
    template <typename T> DYN_HD void sincos(T a, T &s, T &c) { msincos(a, s, c); }
    template <typename T> DYN_HD T sqrt(T a) { return msqrt(a); }
    template <typename T> DYN_HD T atan(T a) { return matan(a); }
};
struct MathRecord {
    double *tape; int n;
    DYN_HD explicit MathRecord(double *t) : tape(t), n(0) {}
    DYN_HD void sincos(double a, double &s, double &c) { msincos(a, s, c); tape[n++] = s; tape[n++] = c; }
    DYN_HD double sqrt(double a) { const double r = msqrt(a); tape[n++] = r; return r; }
    DYN_HD double atan(double a) { const double r = matan(a); tape[n++] = r; return r; }
};
struct MathReplay {
    const double *tape; int n;
    DYN_HD explicit MathReplay(const double *t) : tape(t), n(0) {}
    DYN_HD void sincos(Dual a, Dual &s, Dual &c) { const double sv = tape[n++], cv = tape[n++]; s = Dual(sv, cv * a.d); c = Dual(cv, -sv * a.d); }
    DYN_HD Dual sqrt(Dual a) { const double r = tape[n++]; return Dual(r, 0.5 * a.d / r); }
    DYN_HD Dual atan(Dual a) { const double r = tape[n++]; return Dual(r, a.d / (1.0 + a.v * a.v)); }
};

constexpr int MODEL_PENDULUM = 0, MODEL_QUADROTOR = 1, MODEL_ROCKET = 2;
constexpr double RK4_H = 0.05;

// ---- ODEs ---------------------------------------------------------------------------------------------------------
template <typename T, typename M>
DYN_HD void ode_pendulum(const T *X, const T *U, T *dX, M &mf) {
    const T xd = X[1], th = X[2], thd = X[3], u = U[0];
    const double m1 = 1.0, m2 = 0.1, l = 0.5, g = 9.81;
    T s, c;
    mf.sincos(th, s, c);
    const T den = T(m1) + T(m2) * (T(1.0) - c * c);
    const T xdd = (u + T(m2 * l) * thd * thd * s - T(m2 * g) * s * c) / den;
    const T thdd = (-u * c - T(m2 * l) * thd * thd * s * c + T((m1 + m2) * g) * s) / (T(l) * den);
    dX[0] = xd; dX[1] = xdd; dX[2] = thd; dX[3] = thdd;
}

// rotation matrix body->world of q = [qw,qx,qy,qz] applied to a body vector (quadrotor.py:124-138 / rocket.py:196-210)
template <typename T>
DYN_HD void rot_apply(T qw, T qx, T qy, T qz, T bx, T by, T bz, T &wx, T &wy, T &wz) {
    const T two(2.0), one(1.0);
    const T r00 = one - two * qy * qy - two * qz * qz, r01 = two * qx * qy - two * qz * qw, r02 = two * qx * qz + two * qy * qw;
    const T r10 = two * qx * qy + two * qz * qw, r11 = one - two * qx * qx - two * qz * qz, r12 = two * qy * qz - two * qx * qw;
    const T r20 = two * qx * qz - two * qy * qw, r21 = two * qy * qz + two * qx * qw, r22 = one - two * qx * qx - two * qy * qy;
    wx = r00 * bx + r01 * by + r02 * bz;
    wy = r10 * bx + r11 * by + r12 * bz;
    wz = r20 * bx + r21 * by + r22 * bz;
}

template <typename T, typename M>
DYN_HD void ode_quadrotor(const T *X, const T *U, T *dX, M &) {
    const double m = 1.0, g = 9.81, l = 0.15, Jx = 0.02, Jy = 0.02, Jz = 0.04, kM = 0.01;
    const T qw = X[6], qx = X[7], qy = X[8], qz = X[9], wx = X[10], wy = X[11], wz = X[12];
    const T f1 = U[0], f2 = U[1], f3 = U[2], f4 = U[3];
    const T Fz = f1 + f2 + f3 + f4;
    T ax, ay, az;
    rot_apply(qw, qx, qy, qz, T(0.0), T(0.0), Fz, ax, ay, az);
    dX[0] = X[3]; dX[1] = X[4]; dX[2] = X[5];
    dX[3] = ax * T(1.0 / m); dX[4] = ay * T(1.0 / m); dX[5] = az * T(1.0 / m) - T(g);
    // qdot = 0.5 * Omega(omega) q
    dX[6] = T(0.5) * (-wx * qx - wy * qy - wz * qz);
    dX[7] = T(0.5) * (wx * qw + wz * qy - wy * qz);
    dX[8] = T(0.5) * (wy * qw - wz * qx + wx * qz);
    dX[9] = T(0.5) * (wz * qw + wy * qx - wx * qy);
    const T tx = T(l) * (f2 - f4), ty = T(l) * (f3 - f1), tz = T(kM) * (f1 - f2 + f3 - f4);
    // J wdot = tau - w x (J w)
    const T Jwx = T(Jx) * wx, Jwy = T(Jy) * wy, Jwz = T(Jz) * wz;
    dX[10] = (tx - (wy * Jwz - wz * Jwy)) * T(1.0 / Jx);
    dX[11] = (ty - (wz * Jwx - wx * Jwz)) * T(1.0 / Jy);
    dX[12] = (tz - (wx * Jwy - wy * Jwx)) * T(1.0 / Jz);
}

template <typename T, typename M>
DYN_HD T gimbal_angle(T servo, T cos_tilt, M &mf) {   // rocket.py:246-254; takes cos(tilt_axis_angle): the caller has it already
    const double a = 5.0, b = 35.2, c = 33.0, d = 28.0, e = 35.2;
    T ss, cs;
    mf.sincos(servo, ss, cs);
    const T iv1 = T(d) + T(a) * cs;
    const T iv2 = T(e) - T(a) * ss;
    const T u = T(b * b - c * c) - iv1 * iv1 - iv2 * iv2;
    const T v = T(2.0 * c) * cos_tilt * iv2;
    const T w = T(-2.0 * c) * iv1;
    const T iv3 = w * w + v * v - u * u;
    return T(2.0) * mf.atan((v - mf.sqrt(iv3)) / (u + w));
}

template <typename T, typename M>
DYN_HD void ode_rocket(const T *X, const T *U, T *dX, M &mf) {
    const double mass = 1.16, grav = 9.81, Jxx = 0.00210, Jyy = 0.1, Jzz = 0.1, off = 0.42, tau_t = 0.06, tau_s = 0.10, hover = 11.3796;
    const T qw = X[6], qx = X[7], qy = X[8], qz = X[9], wx = X[10], wy = X[11], wz = X[12];
    const T thrust = X[13] + T(hover), torque_x = X[14], sa1 = X[15], sa2 = X[16];
    const T thrust_in = U[0] + T(hover), torque_in = U[1], sa1_in = U[2], sa2_in = U[3];
    T s1, c1, s2, c2;
    const T g1 = gimbal_angle(sa1, T(1.0), mf);
    mf.sincos(g1, s1, c1);
    const T g2 = gimbal_angle(sa2, c1, mf);
    mf.sincos(g2, s2, c2);
    const T Bx = -thrust * s1 * c2, By = thrust * s2, Bz = thrust * c1 * c2;
    T ax, ay, az;
    rot_apply(qw, qx, qy, qz, Bx, By, Bz, ax, ay, az);
    dX[0] = X[3]; dX[1] = X[4]; dX[2] = X[5];
    dX[3] = ax * T(1.0 / mass); dX[4] = ay * T(1.0 / mass); dX[5] = az * T(1.0 / mass) - T(grav);
    dX[6] = T(0.5) * (-wx * qx - wy * qy - wz * qz);
    dX[7] = T(0.5) * (wx * qw + wz * qy - wy * qz);
    dX[8] = T(0.5) * (wy * qw - wz * qx + wx * qz);
    dX[9] = T(0.5) * (wz * qw + wy * qx - wx * qy);
    // torque = cog_offset x B_thrust with cog_offset = (0,0,-off);  J wdot = torque - w x (J w)   (torque_x state is not fed back: rocket.py:228-231)
    const T tqx = T(off) * By, tqy = -T(off) * Bx, tqz = T(0.0);
    const T Jwx = T(Jxx) * wx, Jwy = T(Jyy) * wy, Jwz = T(Jzz) * wz;
    dX[10] = (tqx - (wy * Jwz - wz * Jwy)) * T(1.0 / Jxx);
    dX[11] = (tqy - (wz * Jwx - wx * Jwz)) * T(1.0 / Jyy);
    dX[12] = (tqz - (wx * Jwy - wy * Jwx)) * T(1.0 / Jzz);
    dX[13] = (thrust_in - thrust) * T(1.0 / tau_t);
    dX[14] = (torque_in - torque_x) * T(1.0 / tau_t);
    dX[15] = (sa1_in - sa1) * T(1.0 / tau_s);
    dX[16] = (sa2_in - sa2) * T(1.0 / tau_s);
}

template <int MODEL> struct Dims;
// NT: transcendental values one ODE evaluation puts on the tape (pendulum: sin, cos; rocket: 2 x (sin, cos, sqrt, atan) of the gimbal linkage + 2 x (sin, cos))
template <> struct Dims<MODEL_PENDULUM> { static constexpr int NX = 4, NU = 1, NT = 2; };
template <> struct Dims<MODEL_QUADROTOR> { static constexpr int NX = 13, NU = 4, NT = 0; };
template <> struct Dims<MODEL_ROCKET> { static constexpr int NX = 17, NU = 4, NT = 12; };
constexpr int NT_MAX = 12;

template <int MODEL, typename T, typename M>
DYN_HD void ode(const T *X, const T *U, T *dX, M &mf) {
    if (MODEL == MODEL_PENDULUM) ode_pendulum<T, M>(X, U, dX, mf);
    else if (MODEL == MODEL_QUADROTOR) ode_quadrotor<T, M>(X, U, dX, mf);
    else ode_rocket<T, M>(X, U, dX, mf);
}
template <int MODEL, typename T>
DYN_HD void ode(const T *X, const T *U, T *dX) { MathPlain mf; ode<MODEL, T, MathPlain>(X, U, dX, mf); }

// x+ = RK4(x, u), h = 0.05   (dyn/model.py:27-32); k's are accumulated on the fly to keep the register footprint small
template <int MODEL, typename T>
DYN_HD void ddyn(const T *X, const T *U, T *Xp) {
    constexpr int NX = Dims<MODEL>::NX;
    const T h(RK4_H);
    T k[NX], t[NX], acc[NX];
    ode<MODEL, T>(X, U, k);
    for (int i = 0; i < NX; i++) { acc[i] = k[i]; t[i] = X[i] + T(0.5) * h * k[i]; }
    ode<MODEL, T>(t, U, k);
    for (int i = 0; i < NX; i++) { acc[i] = acc[i] + T(2.0) * k[i]; t[i] = X[i] + T(0.5) * h * k[i]; }
    ode<MODEL, T>(t, U, k);
    for (int i = 0; i < NX; i++) { acc[i] = acc[i] + T(2.0) * k[i]; t[i] = X[i] + h * k[i]; }
    ode<MODEL, T>(t, U, k);
    for (int i = 0; i < NX; i++) Xp[i] = X[i] + T(1.0 / 6.0) * (acc[i] + k[i]) * h;
}

// One column of [A B] = d ddyn / d (x,u)_dir by forward-mode AD, in two steps that a GPU thread can hold in registers (pushing dual numbers
// through the whole RK4 step carries X, k, t, acc as duals: 288 registers for the rocket, 189 of them spilled in round 1's kernel):
// (1) ddyn_stages: the plain RK4 step, keeping the three intermediate stage points; (2) ddyn_tangent: the tangent of the step along one
// direction, the stage points' values read back instead of carried.
// tape (4 * Dims<MODEL>::NT doubles; may be NULL when NT == 0): the transcendental values of the four evaluations, for ddyn_tangent
template <int MODEL>
DYN_HD void ddyn_stages(const double *x, const double *u, double *stage /* 3*NX: x + h/2 k1, x + h/2 k2, x + h k3 */, double *xp, double *tape) {
    constexpr int NX = Dims<MODEL>::NX, NT = Dims<MODEL>::NT;
    const double h = RK4_H;
    double k[NX], t[NX], acc[NX];
    { MathRecord mf(tape); ode<MODEL, double, MathRecord>(x, u, k, mf); }
    for (int i = 0; i < NX; i++) { acc[i] = k[i]; t[i] = x[i] + 0.5 * h * k[i]; stage[i] = t[i]; }
    { MathRecord mf(tape + NT); ode<MODEL, double, MathRecord>(t, u, k, mf); }
    for (int i = 0; i < NX; i++) { acc[i] = acc[i] + 2.0 * k[i]; t[i] = x[i] + 0.5 * h * k[i]; stage[NX + i] = t[i]; }
    { MathRecord mf(tape + 2 * NT); ode<MODEL, double, MathRecord>(t, u, k, mf); }
    for (int i = 0; i < NX; i++) { acc[i] = acc[i] + 2.0 * k[i]; t[i] = x[i] + h * k[i]; stage[2 * NX + i] = t[i]; }
    { MathRecord mf(tape + 3 * NT); ode<MODEL, double, MathRecord>(t, u, k, mf); }
    for (int i = 0; i < NX; i++) xp[i] = x[i] + (1.0 / 6.0) * (acc[i] + k[i]) * h;
}
template <int MODEL>
DYN_HD void ddyn_tangent(const double *x, const double *u, const double *stage, const double *tape, int dir, double *col) {
    constexpr int NX = Dims<MODEL>::NX, NU = Dims<MODEL>::NU, NT = Dims<MODEL>::NT;
    const double h = RK4_H;
    Dual T[NX], U[NU], K[NX];
    double acc[NX];
    for (int i = 0; i < NU; i++) U[i] = Dual(u[i], NX + i == dir ? 1.0 : 0.0);
    for (int i = 0; i < NX; i++) { T[i] = Dual(x[i], i == dir ? 1.0 : 0.0); acc[i] = 0.0; }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        { MathReplay mf(tape + s * NT); ode<MODEL, Dual, MathReplay>(T, U, K, mf); }
        const double w = (s == 0 || s == 3) ? 1.0 : 2.0, cn = (s == 2) ? h : 0.5 * h;
        for (int i = 0; i < NX; i++) {
            acc[i] += w * K[i].d;
            if (s < 3) T[i] = Dual(stage[s * NX + i], (i == dir ? 1.0 : 0.0) + cn * K[i].d);
        }
    }
    for (int i = 0; i < NX; i++) col[i] = (i == dir ? 1.0 : 0.0) + (h / 6.0) * acc[i];
}

}  // namespace dyn
