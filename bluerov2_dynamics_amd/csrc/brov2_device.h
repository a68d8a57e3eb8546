// brov2_device.h -- device-side Fossen 6-DOF right-hand side for gfx950 (fp64 VALU).
//
// One wavefront lane owns one vehicle; everything here is straight-line fp64 code on
// registers.  The algebra is the reference's (fossen/BlueROV2.py:357-400 and the two wrench
// variants) with the sparse 6x6 products written out:
//   M is diagonal                       -> Minv is 6 multiplies           (BlueROV2.py:101-126)
//   C(nu) nu = [P x w ; P x v + L x w]-like cross terms with the author's sign fix kept
//              (C[3,4] = +Iz r, C[4,3] = -Iz r, BlueROV2.py:293,297) -> 12 products
//   D(nu_r) nu_r = (dl + dq |nu_r|) nu_r                                  (BlueROV2.py:327-338)
// and sin/cos of (phi, theta, psi) evaluated once per RHS (the reference evaluates them in
// rotation_matrix, euler_kinematics_matrix and _restoring separately).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace brov {

constexpr int MODEL_THRUSTER_EULER = 0;
constexpr int MODEL_WRENCH_EULER = 1;
constexpr int MODEL_WRENCH_QUAT = 2;
// learned double-integrator baseline of the comparison scripts (training/train_tank_brov2_full_comparison.py:531-573,
// ..._rk4.py:461-525, ..._wrench_comp.py:293-341, ..._wrench_quat.py:324-372): state-independent accelerations U K
constexpr int MODEL_DI_THRUSTER_EULER = 3;
constexpr int MODEL_DI_WRENCH_EULER = 4;
constexpr int MODEL_DI_WRENCH_QUAT = 5;
constexpr bool model_is_quat(int m) { return m == MODEL_WRENCH_QUAT || m == MODEL_DI_WRENCH_QUAT; }
constexpr bool model_is_di(int m) { return m >= MODEL_DI_THRUSTER_EULER; }
constexpr int INTEG_EULER = 0;
constexpr int INTEG_RK4 = 1;
constexpr int LAYOUT_BTU = 0;
constexpr int LAYOUT_TUB = 1;
constexpr int LAYOUT_TPB = 2;   // time-major, channel PAIRS interleaved: [T][ceil(n/2)][B][2] -> 16-byte accesses per lane

template <int MODEL> struct Dims {
    static constexpr int NX = model_is_quat(MODEL) ? 13 : 12;
    static constexpr int NU = (MODEL == MODEL_THRUSTER_EULER || MODEL == MODEL_DI_THRUSTER_EULER) ? 8 : 6;
    static constexpr int NV = NX - 6;  // offset of nu inside x
};

// Everything a kernel needs, precomputed on the host in fp64 (capi.hip: derive_params).
// Passed by value as a kernel argument (uniform -> SGPR / scalar loads).
struct DevParams {
    double md[6];        // diag(M) = (m - Xu_dot, ..., Iz - Nr_dot)
    double minv[6];      // 1 / md
    double dl[6];        // linear damping  (-Xu ...)  >= 0
    double dq[6];        // quadratic damping (-Xu_abs ...) >= 0
    double WmB;          // W - B
    double xbB, ybB, zbB;
    double cur[3];
    int has_current;
    int pad0;
    double alloc[6][8];  // tau = alloc F
    double poly[5];      // F_cmd = V (c0 + V^2 (c1 + V^2 (c2 + V^2 (c3 + V^2 c4))))
    // thruster lag, discretised at this call's dt.  For s = 1..4 consecutive samples with the
    // same input f:  y_s = lag_c[s-1] . x + lag_d[s-1] f ;  x_after_s = lag_A[s-1] x + lag_b[s-1] f
    // (lag_A[s-1] = Ad^s, lag_b[s-1] = (I + Ad + .. + Ad^(s-1)) Bd, lag_c = Cc Ad^s, lag_d = Cc lag_b).
    double lag_A[4][9];
    double lag_b[4][3];
    double lag_c[4][3];
    double lag_d[4];
};

struct SinCos { double s, c; };
__device__ __forceinline__ SinCos sincos_f64(double a) {
    SinCos r;
    sincos(a, &r.s, &r.c);
    return r;
}

// F_cmd polynomial (fossen/BlueROV2.py:251-257), Horner in V^2.
__device__ __forceinline__ double thrust_poly(const DevParams& p, double V) {
    double V2 = V * V;
    double h = fma(V2, p.poly[4], p.poly[3]);
    h = fma(V2, h, p.poly[2]);
    h = fma(V2, h, p.poly[1]);
    h = fma(V2, h, p.poly[0]);
    return V * h;
}

// tau = alloc . F  (fossen/BlueROV2.py:265-278)
__device__ __forceinline__ void allocate(const DevParams& p, const double F[8], double tau[6]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double a = p.alloc[k][0] * F[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) a = fma(p.alloc[k][i], F[i], a);
        tau[k] = a;
    }
}

// nu_dot = Minv (tau - C nu - D nu_r - g), R = R_{b->n} (row-major 9), g = restoring (6).
__device__ __forceinline__ void nu_dot(const DevParams& p, const double R[9], const double nu[6],
                                       const double tau[6], const double g[6], double out[6]) {
    double nr[6] = {nu[0], nu[1], nu[2], nu[3], nu[4], nu[5]};
    if (p.has_current) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < 3; ++i) nr[i] -= fma(R[6 + i], p.cur[2], fma(R[3 + i], p.cur[1], R[i] * p.cur[0]));
    }
    const double u = nu[0], v = nu[1], w = nu[2], pp = nu[3], q = nu[4], r = nu[5];
    const double Pu = p.md[0] * u, Pv = p.md[1] * v, Pw = p.md[2] * w;
    const double Lp = p.md[3] * pp, Lq = p.md[4] * q, Lr = p.md[5] * r;
    double cn[6];
    cn[0] = fma(Pw, q, -(Pv * r));
    cn[1] = fma(Pu, r, -(Pw * pp));
    cn[2] = fma(Pv, pp, -(Pu * q));
    cn[3] = fma(Pw, v, -(Pv * w)) + fma(Lr, q, -(Lq * r));
    cn[4] = fma(Pu, w, -(Pw * u)) + fma(Lp, r, -(Lr * pp));
    cn[5] = fma(Pv, u, -(Pu * v)) + fma(Lq, pp, -(Lp * q));
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double d = fma(p.dq[i], fabs(nr[i]), p.dl[i]) * nr[i];
        out[i] = p.minv[i] * (tau[i] - cn[i] - d - g[i]);
    }
}

__device__ __forceinline__ void restoring(const DevParams& p, double sth, double cth_sphi, double cth_cphi, double g[6]) {
    g[0] = p.WmB * sth;
    g[1] = -p.WmB * cth_sphi;
    g[2] = -p.WmB * cth_cphi;
    g[3] = fma(p.ybB, cth_cphi, -(p.zbB * cth_sphi));
    g[4] = -fma(p.zbB, sth, p.xbB * cth_cphi);
    g[5] = fma(p.xbB, cth_sphi, p.ybB * sth);
}

// xdot = f(x, tau) for the Euler-angle state (thruster and wrench models share it).
__device__ __forceinline__ void rhs_euler_angles(const DevParams& p, const double x[12], const double tau[6], double xd[12]) {
    const SinCos a = sincos_f64(x[3]), b = sincos_f64(x[4]), c = sincos_f64(x[5]);
    const double sphi = a.s, cphi = a.c, sth = b.s, cth = b.c, spsi = c.s, cpsi = c.c;
    double R[9];
    R[0] = cpsi * cth; R[1] = fma(cpsi * sth, sphi, -(spsi * cphi)); R[2] = fma(cpsi * cphi, sth, spsi * sphi);
    R[3] = spsi * cth; R[4] = fma(sphi * sth, spsi, cpsi * cphi);    R[5] = fma(sth * spsi, cphi, -(cpsi * sphi));
    R[6] = -sth;       R[7] = cth * sphi;                            R[8] = cth * cphi;
    double g[6];
    restoring(p, sth, R[7], R[8], g);
    const double* nu = x + 6;
    nu_dot(p, R, nu, tau, g, xd + 6);
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[i] = fma(R[3 * i + 2], nu[2], fma(R[3 * i + 1], nu[1], R[3 * i] * nu[0]));
    // Euler-angle kinematics with the reference's cos(theta) clamp (fossen/BlueROV2.py:52-61)
    double cc = cth;
    if (fabs(cc) < 1e-7) cc = 1e-7 * ((cc > 0.0) - (cc < 0.0));
    const double ic = 1.0 / cc;
    const double tth = sth * ic;
    const double pq = nu[3], qq = nu[4], rq = nu[5];
    xd[3] = fma(cphi * tth, rq, fma(sphi * tth, qq, pq));
    xd[4] = fma(cphi, qq, -(sphi * rq));
    xd[5] = fma(cphi * ic, rq, (sphi * ic) * qq);
}

// quat_normalize (fossen/BlueROV2_wrench.py:27-36)
__device__ __forceinline__ void quat_normalize(double q[4]) {
    const double n = sqrt(fma(q[3], q[3], fma(q[2], q[2], fma(q[1], q[1], q[0] * q[0]))));
    if (n < 1e-12) { q[0] = 1.0; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0; return; }
    const double inv = 1.0 / n;
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}

// xdot = f(x, tau) for the quaternion state (fossen/BlueROV2_wrench.py:322-367)
__device__ __forceinline__ void rhs_quat(const DevParams& p, const double x[13], const double tau[6], double xd[13]) {
    double q[4] = {x[3], x[4], x[5], x[6]};
    quat_normalize(q);
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    double R[9];
    R[0] = 1.0 - 2.0 * fma(qy, qy, qz * qz); R[1] = 2.0 * fma(qx, qy, -(qz * qw));    R[2] = 2.0 * fma(qx, qz, qy * qw);
    R[3] = 2.0 * fma(qx, qy, qz * qw);       R[4] = 1.0 - 2.0 * fma(qx, qx, qz * qz); R[5] = 2.0 * fma(qy, qz, -(qx * qw));
    R[6] = 2.0 * fma(qx, qz, -(qy * qw));    R[7] = 2.0 * fma(qy, qz, qx * qw);       R[8] = 1.0 - 2.0 * fma(qx, qx, qy * qy);
    double g[6];
    restoring(p, -R[6], R[7], R[8], g);
    const double* nu = x + 7;
    nu_dot(p, R, nu, tau, g, xd + 7);
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[i] = fma(R[3 * i + 2], nu[2], fma(R[3 * i + 1], nu[1], R[3 * i] * nu[0]));
    const double wx = nu[3], wy = nu[4], wz = nu[5];
    xd[3] = 0.5 * (-(qx * wx) - qy * wy - qz * wz);
    xd[4] = 0.5 * (qw * wx + qy * wz - qz * wy);
    xd[5] = 0.5 * (qw * wy - qx * wz + qz * wx);
    xd[6] = 0.5 * (qw * wz + qx * wy - qy * wx);
}

template <int MODEL>
__device__ __forceinline__ void rhs_state(const DevParams& p, const double* x, const double tau[6], double* xd) {
    if constexpr (MODEL == MODEL_WRENCH_QUAT) rhs_quat(p, x, tau, xd);
    else rhs_euler_angles(p, x, tau, xd);
}

// Thruster lag bank of one vehicle: 8 filters x 3 states.
// forces_after(s): thrust seen by the s-th dynamics() call since the last commit (s = 1..4);
// advance(s): commit s samples.  Input fcmd is held over those samples (same u in all RK4 stages).
struct LagBank {
    double x[8][3];
    __device__ __forceinline__ void forces_after(const DevParams& p, int s, const double fcmd[8], double F[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            F[i] = fma(p.lag_c[s - 1][2], x[i][2], fma(p.lag_c[s - 1][1], x[i][1], fma(p.lag_c[s - 1][0], x[i][0], p.lag_d[s - 1] * fcmd[i])));
    }
    __device__ __forceinline__ void advance(const DevParams& p, int s, const double fcmd[8]) {
        const double* A = p.lag_A[s - 1];
        const double* b = p.lag_b[s - 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double a0 = x[i][0], a1 = x[i][1], a2 = x[i][2];
            x[i][0] = fma(A[2], a2, fma(A[1], a1, fma(A[0], a0, b[0] * fcmd[i])));
            x[i][1] = fma(A[5], a2, fma(A[4], a1, fma(A[3], a0, b[1] * fcmd[i])));
            x[i][2] = fma(A[8], a2, fma(A[7], a1, fma(A[6], a0, b[2] * fcmd[i])));
        }
    }
};

}  // namespace brov
