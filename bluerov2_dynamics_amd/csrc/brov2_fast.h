// brov2_fast.h -- the time-loop form of the Fossen model used by the rollout / window kernels.
//
// Same equations as brov2_device.h (which is the literal per-call form used by brov_rhs), re-arranged
// so that one RK4 step costs ~760 fp64 instructions instead of ~3k:
//
//  * Minv is folded into everything: the kernels integrate nu_dot = a_thr - c(nu) - d(nu_r) - g(eta)
//    with a_thr = Minv tau, c = Minv C(nu) nu written as 12 products with 12 pre-multiplied constants
//    (the author's Coriolis sign fix, fossen/BlueROV2.py:293,297, is inside E30..E51), d = (da+db|nu_r|)nu_r,
//    g = Minv g(eta).
//  * The thruster lag is carried in "acceleration space": the eight filters share (Ad, Bd, Cc), so
//    Z = (Minv T) X (6x3) obeys the same recurrence as the 8x3 per-thruster state X and
//    Minv tau_s = Z c_s + (Minv T f_cmd) d_s for the s-th dynamics() call of a step (quirk Q1).
//    That is 6 filters instead of 8 and one allocation product per step instead of four.  The
//    per-thruster state is advanced alongside only when the caller wants it back (lag_io != NULL).
//  * sin/cos: 3-term Cody-Waite reduction + the fdlibm kernel polynomials (~1 ulp for |x| < 3e9),
//    instead of OCML's double-double reduction (fossen/BlueROV2.py:28-33,47-50,342-345 call
//    np.sin/np.cos three times per call on the same angles; here once).  RK4 stages 2-4 get theirs from the
//    first stage's by the addition theorem on the small angle increment (trig_delta), and so does the NEXT step's first
//    stage from the step's total increment (integrate_fast's carry; a full evaluation every 64 steps).
//  * The lag bank is carried in the observer basis of its next three outputs (LagZ::to_observer): stages 1-3
//    of a step read the thrust with one FMA per channel.
//  * The hottest constants live in VGPRs (uniform values, pinned with an empty asm) and the rest is
//    re-read once per step through a laundered constant pointer (scalar loads), so nothing spills to scratch:
//    the first version kept all 150 constants live in SGPRs and paid ~1000 v_readlane per step.
#pragma once
#include "brov2_device.h"

namespace brov {

// ---- constants of the time loop (host: derive_fast() in capi.hip) ------------------------------
struct FastParams {
    // per-stage (hot): pinned in VGPRs
    double E[12];      // Minv-folded Coriolis coefficients, see nu_dot_fast
    double da[6], db[6];
    double G[5];       // G0..G2 = minv_i (W-B), Z3 = minv3 zb B, Z4 = minv4 zb B
    double lc[4][3];   // Cc Ad^s
    double ld[4];      // Cc S_s Bd
    // per-step (warm): scalar loads
    double Tm[6][8];   // Minv T
    double minv[6];    // wrench models: a_cmd = minv . tau
    double poly[5];
    double A1[9], b1[3];   // one lag sample
    double A4[9], b4[3];   // four lag samples
    // the same lag bank in the observer basis w = O z, rows of O = Cc Ad^1..3 (see LagZ): used by the GENERIC = false kernels
    double Ob[9];          // O
    double al4[3];         // Cc Ad^4 O^-1
    double Aw4[9], bw4[3]; // O A4 O^-1, O b4
    double g1[3];          // O Bd
    // rare
    double XY[4];      // Y3 = minv3 yb B, X4 = minv4 xb B, X5 = minv5 xb B, Y5 = minv5 yb B
    double cur[3];
    int has_current;
    int has_xy;
    int tm_dense;      // allocation matrix does not have the reference's zero pattern
    int obs_bad;       // O is (nearly) singular: the observer form is not usable, host selects the GENERIC kernels
};

// FastParams lives in a small device buffer owned by the ctx and is read through a CONSTANT
// address-space pointer: uniform loads from it are scalar (s_load) by construction, whereas a
// by-value kernel argument whose address is taken gets copied to scratch and read per lane.
typedef const FastParams __attribute__((address_space(4)))* CFP;
__device__ __forceinline__ CFP as_constant(const FastParams* g) { return (CFP)(unsigned long long)g; }

// RESIDENT: the stage constants (da, G, ld, al4) are held in SGPRs for the whole launch (kernels whose time loop leaves the
// scalar register file room for them: the body wave of rollout_pair_kernel); otherwise they are re-read through the
// per-step laundered pointer, so that no loop-invariant SGPRs have to be spilled around the sin/cos literals and the
// per-step scalar loads.  E and db are pinned in VGPRs either way (fma(db, |nu|, da) needs a non-scalar second constant).
template <bool RES>
struct HotConstsT {
    static constexpr bool RESIDENT = RES;
    double E[12], db[6];
    double da[6], G[5], ld[4], al4[3];     // RESIDENT only
};
typedef HotConstsT<false> HotConsts;
typedef HotConstsT<true> HotConstsResident;

#ifndef BROV_CLAMP_BRANCH
#define BROV_CLAMP_BRANCH 1
#endif
#ifndef BROV_PSI_FRAME
#define BROV_PSI_FRAME 1
#endif
#define BROV_PIN_V(x) asm volatile("" : "+v"(x))
#define BROV_PIN_S(x) asm volatile("" : "+s"(x))
#define BROV_SC(h, p, f) (HC::RESIDENT ? (h).f : (p)->f)

template <bool RES>
__device__ __forceinline__ void load_hot(CFP pp, HotConstsT<RES>& h) {
    const auto& p = *pp;
#pragma unroll
    for (int i = 0; i < 12; ++i) { h.E[i] = p.E[i]; BROV_PIN_V(h.E[i]); }
#pragma unroll
    for (int i = 0; i < 6; ++i) { h.db[i] = p.db[i]; BROV_PIN_V(h.db[i]); }
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { h.da[i] = p.da[i]; BROV_PIN_S(h.da[i]); }
#pragma unroll
        for (int i = 0; i < 5; ++i) { h.G[i] = p.G[i]; BROV_PIN_S(h.G[i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { h.ld[i] = p.ld[i]; BROV_PIN_S(h.ld[i]); }
#pragma unroll
        for (int i = 0; i < 3; ++i) { h.al4[i] = p.al4[i]; BROV_PIN_S(h.al4[i]); }
    }
}

// pointer laundering: makes the compiler re-issue the (scalar) loads behind `p` at this point
// instead of keeping 100+ SGPRs of loop-invariant constants alive (and spilling them)
__device__ __forceinline__ CFP relaunder(CFP p) {
    asm volatile("" : "+s"(p));
    return p;
}

// ---- sin / cos --------------------------------------------------------------------------------
// Valid (<= ~1 ulp) for |x| < 2^31 pi/2 ~ 3.4e9 rad; NaN/inf give NaN like np.sin/np.cos.  Beyond that
// range the quadrant index saturates (a vehicle attitude angle never gets there).
__device__ __forceinline__ void sincos_fast(double x, double& s, double& c, const double2* __restrict__ qt) {
    const double kf = rint(x * 6.36619772367581382433e-01);
    double r = fma(-kf, 1.57079632673412561417e+00, x);      // pio2_1  (33 bits)
    r = fma(-kf, 6.07710050630396597660e-11, r);             // pio2_2  (33 bits)
    r = fma(-kf, 2.02226624879595063154e-21, r);             // pio2_2t
    const int q = (int)kf;
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    // quadrant rotation: (sin x, cos x) = (sr cn + cr sn, cr cn - sr sn) with (cn, sn) = (cos, sin)(q pi/2) in
    // {(1,0),(0,1),(-1,0),(0,-1)} read from a 4-entry LDS table: 3 integer/LDS + 4 fp64 instructions instead of
    // the 12 of a select + sign-flip sequence; multiplications by 0 and +-1 are exact.
    const double2 t = qt[q & 3];
    s = fma(cr, t.y, sr * t.x);
    c = fma(-sr, t.y, cr * t.x);
}

// the table lives in LDS: call from every thread of the block before the first sincos_fast, then __syncthreads()
__device__ __forceinline__ void init_quadrant_table(double2* qt) {
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        qt[q] = make_double2(q == 0 ? 1.0 : (q == 2 ? -1.0 : 0.0), q == 1 ? 1.0 : (q == 3 ? -1.0 : 0.0));
    }
}

// sin/cos of the three attitude angles of one dynamics() call
struct Trig { double sphi, cphi, sth, cth, spsi, cpsi; };

__device__ __forceinline__ void trig_full(const double* ang, Trig& t, const double2* __restrict__ qt) {
    sincos_fast(ang[0], t.sphi, t.cphi, qt);
    sincos_fast(ang[1], t.sth, t.cth, qt);
    sincos_fast(ang[2], t.spsi, t.cpsi, qt);
}

// RK4 stages 2-4 evaluate the RHS at angles a + d with a = the angles at the start of the step (whose sin/cos the
// first stage computed) and d = c dt k_angle, a few hundredths of a radian: addition theorem on (sin d, cos d - 1) from
// short Taylor-like kernels instead of three fresh range reductions (15 instructions per angle against 27).
//   |d| <= 1/8 : the leading fdlibm coefficients of sincos_fast, truncated -- they differ from the Taylor ones by
//                < 1e-16 relative in the result there, and sharing the literals keeps them in the same SGPRs.
//   larger     : (3 % of the wave-steps of BASELINE config 2: near theta = +-pi/2 the Euler-angle rates blow up)
//                d is halved k times into that range and (sin, cos - 1) doubled back k times, sin 2a = 2 sin a cos a,
//                cos 2a - 1 = -2 sin^2 a: the rounding error doubles per level (|d| = 2 rad: 16 ulp), the code is a
//                dozen instructions, and no second copy of the full evaluation sits in the time loop (the first
//                version had one per stage: its literals cost 36 SGPR spill reloads and 40 AGPR moves per step).
//   |d| >= 2^37 or not finite: NaN (np.sin gives NaN for inf/NaN; for finite arguments of that size the reference's
//                values carry no information about the trajectory either).
// SHORT (BROV_TRIG_SHORT, round 5): the half-step stages' increments are half the full step's, so their kernels stop one term earlier
// and the direct range is |d| <= 1/16 -- dropped terms d^9 / 9! and d^10 / 10! are < 4e-17 and < 3e-19 ABSOLUTE there, against the
// O(1) sines and cosines the increment is combined with; the share of wave-steps that take the halving path stays what it is.
template <bool SHORT = false>
__device__ __forceinline__ void trig_delta(const Trig& b, const double d[3], Trig& t, double* dpsi = nullptr) {
    double dd[3] = {d[0], d[1], d[2]};
    int k = 0;
    const double m = fmax(fmax(fabs(d[0]), fabs(d[1])), fabs(d[2]));
    if (!(m <= (SHORT ? 0.0625 : 0.125))) {                  // also NaN
        int e;
        (void)frexp(m, &e);                                  // m < 2^e
        k = e + (SHORT ? 4 : 3);
        double sc = ldexp(1.0, -k);
        if (!(m < 0x1p37)) { sc = __builtin_nan(""); k = 0; }
#pragma unroll
        for (int i = 0; i < 3; ++i) dd[i] *= sc;
    }
    double sd[3], cm[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double z = dd[i] * dd[i];
        double ps, pc;
        if constexpr (SHORT) {
            ps = fma(z, -1.98412698298579493134e-04, 8.33333333332248946124e-03);
            pc = fma(z, 2.48015872894767294178e-05, -1.38888888888741095749e-03);
        } else {
            ps = fma(z, 2.75573137070700676789e-06, -1.98412698298579493134e-04);
            ps = fma(z, ps, 8.33333333332248946124e-03);
            pc = fma(z, -2.75573143513906633035e-07, 2.48015872894767294178e-05);
            pc = fma(z, pc, -1.38888888888741095749e-03);
        }
        ps = fma(z, ps, -1.66666666666666324348e-01);
        sd[i] = fma(z * dd[i], ps, dd[i]);                   // sin d
        pc = fma(z, pc, 4.16666666666666019037e-02);
        pc = fma(z, pc, -0.5);
        cm[i] = z * pc;                                      // cos d - 1
    }
    for (int j = 0; j < k; ++j) {                            // lane-varying trip count, almost always 0
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double t2 = sd[i] + sd[i];
            const double s2 = fma(t2, cm[i], t2);
            cm[i] = -(t2 * sd[i]);
            sd[i] = s2;
        }
    }
    t.sphi = fma(b.cphi, sd[0], fma(b.sphi, cm[0], b.sphi)); t.cphi = fma(-b.sphi, sd[0], fma(b.cphi, cm[0], b.cphi));
    t.sth = fma(b.cth, sd[1], fma(b.sth, cm[1], b.sth));     t.cth = fma(-b.sth, sd[1], fma(b.cth, cm[1], b.cth));
    t.spsi = fma(b.cpsi, sd[2], fma(b.spsi, cm[2], b.spsi)); t.cpsi = fma(-b.spsi, sd[2], fma(b.cpsi, cm[2], b.cpsi));
    if (dpsi) { dpsi[0] = sd[2]; dpsi[1] = cm[2]; }           // sin(d psi), cos(d psi) - 1 (BROV_PSI_FRAME: see integrate_fast)
}

// 1/x for 1e-7 <= |x| <= 1: v_rcp_f64 seed + two Newton steps  (round 5: one third-order step, y (1 + e + e^2), is an FMA shorter and
// measured no faster: 11.70 against 11.70 ms per config-2 launch -- the reciprocal sits in the shadow of the stage's other chains)
#ifndef BROV_TRIG_SHORT
#define BROV_TRIG_SHORT 1        // shorter sin / cos kernels for the half-step stages of RK4 (trig_delta<SHORT>): 11.70 -> 11.63 ms
#endif
__device__ __forceinline__ double recip_fast(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

// ---- nu_dot = a - Minv C nu - Minv D nu_r - Minv g --------------------------------------------
// sth, ctsp, ctcp = sin(theta), cos(theta) sin(phi), cos(theta) cos(phi)  (or -R20, R21, R22)
// GENERIC = false is the reference's vehicle (no current, xb = yb = 0): one straight-line block per stage.  The
// uniform branches of the generic form cost a scalar load + wait + branch each and stop the scheduler from
// overlapping a stage's scalar constant loads with its sin/cos work.
template <bool GENERIC, class HC>
__device__ __forceinline__ void nu_dot_fast(const HC& h, CFP p, const double R[9], const double nu[6],
                                            const double a[6], double sth, double ctsp, double ctcp, double out[6]) {
    const double u = nu[0], v = nu[1], w = nu[2], pp = nu[3], q = nu[4], r = nu[5];
    double nr0 = u, nr1 = v, nr2 = w;
    if (GENERIC && p->has_current) {   // wave-uniform, rare
        nr0 -= fma(R[6], p->cur[2], fma(R[3], p->cur[1], R[0] * p->cur[0]));
        nr1 -= fma(R[7], p->cur[2], fma(R[4], p->cur[1], R[1] * p->cur[0]));
        nr2 -= fma(R[8], p->cur[2], fma(R[5], p->cur[1], R[2] * p->cur[0]));
    }
    double o0 = fma(-h.E[0], w * q, a[0]);  o0 = fma(h.E[1], v * r, o0);
    double o1 = fma(-h.E[2], u * r, a[1]);  o1 = fma(h.E[3], w * pp, o1);
    double o2 = fma(-h.E[4], v * pp, a[2]); o2 = fma(h.E[5], u * q, o2);
    double o3 = fma(-h.E[6], v * w, a[3]);  o3 = fma(-h.E[7], q * r, o3);
    double o4 = fma(-h.E[8], u * w, a[4]);  o4 = fma(-h.E[9], pp * r, o4);
    double o5 = fma(-h.E[10], u * v, a[5]); o5 = fma(-h.E[11], pp * q, o5);
    o0 = fma(-fma(h.db[0], fabs(nr0), BROV_SC(h, p, da[0])), nr0, o0);
    o1 = fma(-fma(h.db[1], fabs(nr1), BROV_SC(h, p, da[1])), nr1, o1);
    o2 = fma(-fma(h.db[2], fabs(nr2), BROV_SC(h, p, da[2])), nr2, o2);
    o3 = fma(-fma(h.db[3], fabs(pp), BROV_SC(h, p, da[3])), pp, o3);
    o4 = fma(-fma(h.db[4], fabs(q), BROV_SC(h, p, da[4])), q, o4);
    o5 = fma(-fma(h.db[5], fabs(r), BROV_SC(h, p, da[5])), r, o5);
    o0 = fma(-BROV_SC(h, p, G[0]), sth, o0);
    o1 = fma(BROV_SC(h, p, G[1]), ctsp, o1);
    o2 = fma(BROV_SC(h, p, G[2]), ctcp, o2);
    o3 = fma(BROV_SC(h, p, G[3]), ctsp, o3);
    o4 = fma(BROV_SC(h, p, G[4]), sth, o4);
    if (GENERIC && p->has_xy) {        // xb, yb != 0: never for the reference's vehicle
        o3 = fma(-p->XY[0], ctcp, o3);
        o4 = fma(p->XY[1], ctcp, o4);
        o5 = fma(-p->XY[2], ctsp, o5);
        o5 = fma(-p->XY[3], sth, o5);
    }
    out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3; out[4] = o4; out[5] = o5;
}

// xdot for the Euler-angle state; a = Minv tau
// PFRAME: xd[0..1] are left in the frame of the yaw angle (the last of the three plane rotations is not applied)
template <bool GENERIC, class HC, bool PFRAME = false>
__device__ __forceinline__ void rhs_fast_euler(const HC& h, CFP p, const double x[12], const double a[6], double xd[12], const Trig& tg) {
    const double sphi = tg.sphi, cphi = tg.cphi, sth = tg.sth, cth = tg.cth, spsi = tg.spsi, cpsi = tg.cpsi;   // x[3..5] enter only through these
    const double* nu = x + 6;
    const double ctsp = cth * sphi, ctcp = cth * cphi;         // R[7], R[8]
    if constexpr (GENERIC) {
        double R[9];
        const double ss = sth * sphi, sc = sth * cphi;
        R[0] = cpsi * cth; R[1] = fma(cpsi, ss, -(spsi * cphi)); R[2] = fma(cpsi, sc, spsi * sphi);
        R[3] = spsi * cth; R[4] = fma(spsi, ss, cpsi * cphi);    R[5] = fma(spsi, sc, -(cpsi * sphi));
        R[6] = -sth;       R[7] = ctsp;                          R[8] = ctcp;
        nu_dot_fast<true>(h, p, R, nu, a, sth, ctsp, ctcp, xd + 6);
#pragma unroll
        for (int i = 0; i < 3; ++i) xd[i] = fma(R[3 * i + 2], nu[2], fma(R[3 * i + 1], nu[1], R[3 * i] * nu[0]));
    } else {
        nu_dot_fast<false>(h, p, nullptr, nu, a, sth, ctsp, ctcp, xd + 6);
        // p_dot = Rz(psi) Ry(theta) Rx(phi) v applied as three plane rotations: 12 instructions instead of
        // forming the nine entries of R (14) and a 3x3 product (9)
        const double y1 = fma(cphi, nu[1], -(sphi * nu[2])), z1 = fma(sphi, nu[1], cphi * nu[2]);
        const double x2 = fma(cth, nu[0], sth * z1), z2 = fma(cth, z1, -(sth * nu[0]));
        if constexpr (PFRAME) { xd[0] = x2; xd[1] = y1; }
        else {
            xd[0] = fma(cpsi, x2, -(spsi * y1));
            xd[1] = fma(spsi, x2, cpsi * y1);
        }
        xd[2] = z2;
    }
    double cc = cth;
    // fossen/BlueROV2.py:52-54.  As plain code the compiler turns the clamp into nine select / convert instructions per stage;
    // behind a wave-level test (is ANY lane that close to the singularity?) the common path pays one compare and a scalar branch
#if BROV_CLAMP_BRANCH
    if (__builtin_amdgcn_ballot_w64(fabs(cc) < 1e-7) != 0) {
        asm volatile("; cos(theta) clamp, rare" ::: "memory");      // not speculatable: keeps the block behind its branch
        if (fabs(cc) < 1e-7) cc = 1e-7 * ((cc > 0.0) - (cc < 0.0));
    }
#else
    if (fabs(cc) < 1e-7) cc = 1e-7 * ((cc > 0.0) - (cc < 0.0));
#endif
    const double ic = recip_fast(cc);
    const double m = fma(sphi, nu[4], cphi * nu[5]);   // sin(phi) q + cos(phi) r
    xd[3] = fma(sth * ic, m, nu[3]);
    xd[4] = fma(cphi, nu[4], -(sphi * nu[5]));
    xd[5] = ic * m;
}

template <bool GENERIC, class HC>
__device__ __forceinline__ void rhs_fast_quat(const HC& h, CFP p, const double x[13], const double a[6], double xd[13]) {
    double q[4] = {x[3], x[4], x[5], x[6]};
    quat_normalize(q);
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    double R[9];
    R[0] = 1.0 - 2.0 * fma(qy, qy, qz * qz); R[1] = 2.0 * fma(qx, qy, -(qz * qw));    R[2] = 2.0 * fma(qx, qz, qy * qw);
    R[3] = 2.0 * fma(qx, qy, qz * qw);       R[4] = 1.0 - 2.0 * fma(qx, qx, qz * qz); R[5] = 2.0 * fma(qy, qz, -(qx * qw));
    R[6] = 2.0 * fma(qx, qz, -(qy * qw));    R[7] = 2.0 * fma(qy, qz, qx * qw);       R[8] = 1.0 - 2.0 * fma(qx, qx, qy * qy);
    const double* nu = x + 7;
    nu_dot_fast<GENERIC>(h, p, R, nu, a, -R[6], R[7], R[8], xd + 7);
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[i] = fma(R[3 * i + 2], nu[2], fma(R[3 * i + 1], nu[1], R[3 * i] * nu[0]));
    const double wx = nu[3], wy = nu[4], wz = nu[5];
    xd[3] = 0.5 * (-(qx * wx) - qy * wy - qz * wz);
    xd[4] = 0.5 * (qw * wx + qy * wz - qz * wy);
    xd[5] = 0.5 * (qw * wy - qx * wz + qz * wx);
    xd[6] = 0.5 * (qw * wz + qx * wy - qy * wx);
}

// double integrator: dpos = R v, dang = w (Euler angles, "small-angle" in the reference) or q_dot, dnu = a
__device__ __forceinline__ void rhs_di_euler(const double x[12], const double a[6], double xd[12], const Trig& tg) {
    const double sphi = tg.sphi, cphi = tg.cphi, sth = tg.sth, cth = tg.cth, spsi = tg.spsi, cpsi = tg.cpsi;
    const double* v = x + 6;
    const double y1 = fma(cphi, v[1], -(sphi * v[2])), z1 = fma(sphi, v[1], cphi * v[2]);
    const double x2 = fma(cth, v[0], sth * z1);
    xd[0] = fma(cpsi, x2, -(spsi * y1));
    xd[1] = fma(spsi, x2, cpsi * y1);
    xd[2] = fma(cth, z1, -(sth * v[0]));
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[3 + i] = x[9 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) xd[6 + i] = a[i];
}
__device__ __forceinline__ void rhs_di_quat(const double x[13], const double a[6], double xd[13]) {
    double q[4] = {x[3], x[4], x[5], x[6]};
    quat_normalize(q);
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double R0 = 1.0 - 2.0 * fma(qy, qy, qz * qz), R1 = 2.0 * fma(qx, qy, -(qz * qw)), R2 = 2.0 * fma(qx, qz, qy * qw);
    const double R3 = 2.0 * fma(qx, qy, qz * qw), R4 = 1.0 - 2.0 * fma(qx, qx, qz * qz), R5 = 2.0 * fma(qy, qz, -(qx * qw));
    const double R6 = 2.0 * fma(qx, qz, -(qy * qw)), R7 = 2.0 * fma(qy, qz, qx * qw), R8 = 1.0 - 2.0 * fma(qx, qx, qy * qy);
    const double* v = x + 7;
    xd[0] = fma(R2, v[2], fma(R1, v[1], R0 * v[0]));
    xd[1] = fma(R5, v[2], fma(R4, v[1], R3 * v[0]));
    xd[2] = fma(R8, v[2], fma(R7, v[1], R6 * v[0]));
    const double wx = x[10], wy = x[11], wz = x[12];
    xd[3] = 0.5 * (-(qx * wx) - qy * wy - qz * wz);
    xd[4] = 0.5 * (qw * wx + qy * wz - qz * wy);
    xd[5] = 0.5 * (qw * wy - qx * wz + qz * wx);
    xd[6] = 0.5 * (qw * wz + qx * wy - qy * wx);
#pragma unroll
    for (int i = 0; i < 6; ++i) xd[7 + i] = a[i];
}

// tg: sin/cos of x[3..5] for the Euler-angle models (ignored by the quaternion ones)
template <int MODEL, bool GENERIC, class HC, bool PFRAME = false>
__device__ __forceinline__ void rhs_fast(const HC& h, CFP p, const double* x, const double a[6], double* xd, const Trig& tg) {
    if constexpr (MODEL == MODEL_DI_WRENCH_QUAT) rhs_di_quat(x, a, xd);
    else if constexpr (model_is_di(MODEL)) rhs_di_euler(x, a, xd, tg);
    else if constexpr (MODEL == MODEL_WRENCH_QUAT) rhs_fast_quat<GENERIC>(h, p, x, a, xd);
    else rhs_fast_euler<GENERIC, HC, PFRAME>(h, p, x, a, xd, tg);
}

// ---- thruster lag in acceleration space ---------------------------------------------------------
struct LagZ {
    double z[6][3];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 6; ++k) { z[k][0] = 0.0; z[k][1] = 0.0; z[k][2] = 0.0; }
    }
    // Z = (Minv T) X from the per-thruster state X [8][3]
    __device__ __forceinline__ void from_thrusters(CFP p, const double X[8][3]) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double a = p->Tm[k][0] * X[0][j];
#pragma unroll
                for (int i = 1; i < 8; ++i) a = fma(p->Tm[k][i], X[i][j], a);
                z[k][j] = a;
            }
    }
    // acceleration seen by the s-th dynamics() call after the last commit (s = 1..4)
    __device__ __forceinline__ void accel_after(CFP p, int s, const double acmd[6], double a[6]) const {
        const double c0 = p->lc[s - 1][0], c1 = p->lc[s - 1][1], c2 = p->lc[s - 1][2], d = p->ld[s - 1];
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] = fma(c2, z[k][2], fma(c1, z[k][1], fma(c0, z[k][0], d * acmd[k])));
    }
    // ---- observer basis (GENERIC = false kernels) ------------------------------------------------
    // w = O z with rows of O = Cc Ad, Cc Ad^2, Cc Ad^3: the coordinates ARE the zero-input outputs the next three
    // dynamics() calls will see, so stages 1-3 of an RK4 step cost one FMA per channel (w_s + d_s a_cmd), stage 4 four
    // (Cc Ad^4 O^-1 . w), and a one-sample advance is a companion shift.  19 instead of 28 instructions per channel
    // and RK4 step; cond(O) = 9.9 at dt = 0.02 (host refuses the form above 1e4 -> GENERIC kernels).
    __device__ __forceinline__ void to_observer(CFP p) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double a0 = z[k][0], a1 = z[k][1], a2 = z[k][2];
            z[k][0] = fma(p->Ob[2], a2, fma(p->Ob[1], a1, p->Ob[0] * a0));
            z[k][1] = fma(p->Ob[5], a2, fma(p->Ob[4], a1, p->Ob[3] * a0));
            z[k][2] = fma(p->Ob[8], a2, fma(p->Ob[7], a1, p->Ob[6] * a0));
        }
    }
    template <class HC>
    __device__ __forceinline__ void obs_accel_after(const HC& h, CFP p, int s, const double acmd[6], double a[6]) const {
        if (s <= 3) {
            const double d = BROV_SC(h, p, ld[s - 1]);
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] = fma(d, acmd[k], z[k][s - 1]);
        } else {
            const double c0 = BROV_SC(h, p, al4[0]), c1 = BROV_SC(h, p, al4[1]), c2 = BROV_SC(h, p, al4[2]), d = BROV_SC(h, p, ld[3]);
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] = fma(c2, z[k][2], fma(c1, z[k][1], fma(c0, z[k][0], d * acmd[k])));
        }
    }
    __device__ __forceinline__ void obs_advance1(CFP p, const double acmd[6]) {
        const double c0 = p->al4[0], c1 = p->al4[1], c2 = p->al4[2], g0 = p->g1[0], g1 = p->g1[1], g2 = p->g1[2];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double a0 = z[k][0], a1 = z[k][1], a2 = z[k][2];
            z[k][0] = fma(g0, acmd[k], a1);
            z[k][1] = fma(g1, acmd[k], a2);
            z[k][2] = fma(c2, a2, fma(c1, a1, fma(c0, a0, g2 * acmd[k])));
        }
    }
    __device__ __forceinline__ void advance(const double __attribute__((address_space(4)))* A, const double __attribute__((address_space(4)))* b, const double acmd[6]) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double a0 = z[k][0], a1 = z[k][1], a2 = z[k][2];
            z[k][0] = fma(A[2], a2, fma(A[1], a1, fma(A[0], a0, b[0] * acmd[k])));
            z[k][1] = fma(A[5], a2, fma(A[4], a1, fma(A[3], a0, b[1] * acmd[k])));
            z[k][2] = fma(A[8], a2, fma(A[7], a1, fma(A[6], a0, b[2] * acmd[k])));
        }
    }
};

__device__ __forceinline__ void advance_thrusters(const double __attribute__((address_space(4)))* A, const double __attribute__((address_space(4)))* b, const double fcmd[8], double X[8][3]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double a0 = X[i][0], a1 = X[i][1], a2 = X[i][2];
        X[i][0] = fma(A[2], a2, fma(A[1], a1, fma(A[0], a0, b[0] * fcmd[i])));
        X[i][1] = fma(A[5], a2, fma(A[4], a1, fma(A[3], a0, b[1] * fcmd[i])));
        X[i][2] = fma(A[8], a2, fma(A[7], a1, fma(A[6], a0, b[2] * fcmd[i])));
    }
}

// commanded acceleration of one step: Minv T f(u) (thruster model) or Minv tau (wrench models)
// SPARSE: the reference geometry -- horizontal thrusters 0..3 produce no heave, vertical thrusters 4..7 only heave,
// roll and pitch -- so 16 of the 48 allocation products are zero (host-checked, capi.hip: derive_fast)
template <int MODEL, bool SPARSE>
__device__ __forceinline__ void command_accel(CFP p, const double* u, double fcmd[8], double acmd[6]) {
    if constexpr (MODEL == MODEL_THRUSTER_EULER) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double V = u[i], V2 = V * V;
            double hh = fma(V2, p->poly[4], p->poly[3]);
            hh = fma(V2, hh, p->poly[2]);
            hh = fma(V2, hh, p->poly[1]);
            hh = fma(V2, hh, p->poly[0]);
            fcmd[i] = V * hh;
        }
        if constexpr (SPARSE) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int i0 = (k == 2) ? 4 : 0, i1 = (k == 3 || k == 4) ? 8 : (k == 2 ? 8 : 4);
                double a = p->Tm[k][i0] * fcmd[i0];
#pragma unroll
                for (int i = i0 + 1; i < i1; ++i) a = fma(p->Tm[k][i], fcmd[i], a);
                acmd[k] = a;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double a = p->Tm[k][0] * fcmd[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) a = fma(p->Tm[k][i], fcmd[i], a);
                acmd[k] = a;
            }
        }
    } else if constexpr (model_is_di(MODEL)) {
        // a = U K: the [nu][3] gains K_lin / K_ang are stored transposed in Tm (rows 0-2 / 3-5)
        constexpr int NU = Dims<MODEL>::NU;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double a = p->Tm[k][0] * u[0];
#pragma unroll
            for (int i = 1; i < NU; ++i) a = fma(p->Tm[k][i], u[i], a);
            acmd[k] = a;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) acmd[k] = p->minv[k] * u[k];
    }
}

// ---- one integrator step, in two halves ------------------------------------------------------------
// The thruster model's step separates cleanly: the THRUST half (polynomial, allocation, lag bank) never looks at the
// vehicle state, the BODY half (rigid-body right-hand sides + the integrator) sees the thrust only as the acceleration
// a_s = Minv tau of its s-th dynamics() call.  step_fast runs both in one lane; rollout_pair_kernel (rollout.hip) gives
// each half its own wave.
//
// BODY: integrate_fast.  accel(s, a) delivers the commanded acceleration seen by the s-th dynamics() call of the step
// (s = 1..4 for RK4; quirk Q1: the reference's lag filters advance on every call, so the four stages see different thrust).
// RK4 is accumulated as xn = x + dt/6 k1 + dt/3 k2 + dt/3 k3 + dt/6 k4 (four FMAs per state instead of forming
// k1 + 2 k2 + 2 k3 + k4 first; same value up to rounding, 24 instructions fewer per step).
// carry != nullptr: the sin/cos of the angles at the start of the step are carried over from the end of the previous step,
// where they were advanced by the addition theorem on the step's angle increment (trig_delta: 15 instructions per angle
// instead of 27 + a table look-up); `refresh` (wave-uniform) forces the full evaluation, which callers request every 64
// steps so that the accumulated rounding (a few ulp per update) stays ~1e-14.
template <int MODEL, int INTEG, bool GENERIC, class HC, class AccelFn>
__device__ __forceinline__ void integrate_fast(const HC& h, CFP p, double dt, double* x, AccelFn&& accel,
                                               const double2* qt, Trig* carry, bool refresh) {
    constexpr int NX = Dims<MODEL>::NX;
    constexpr bool ANG = !model_is_quat(MODEL);          // x[3..5] are Euler angles
    double a[6];
    Trig tb, ts;
    const bool CARRY = ANG && carry != nullptr;
    if constexpr (ANG) {
        if (!CARRY || refresh) trig_full(x + 3, tb, qt);
        else tb = *carry;
    }
    double dang[3];          // angle increment of the whole step (carry only)
    if constexpr (INTEG == INTEG_EULER) {
        double k[NX];
        accel(1, a);
        rhs_fast<MODEL, GENERIC>(h, p, x, a, k, tb);
        if (CARRY) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if (i >= 3 && i < 6) { dang[i - 3] = dt * k[i]; x[i] += dang[i - 3]; }
                else x[i] = fma(dt, k[i], x[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = fma(dt, k[i], x[i]);
        }
    } else {
        double k[NX], xn[NX], xs[NX], dl[3];
        const double h2 = 0.5 * dt, h6 = dt / 6.0, h3 = dt / 3.0;
        // PSIF (the reference vehicle's Euler-angle models): the horizontal position increment is accumulated in the frame of
        // the step's initial yaw.  A stage's p_dot arrives in ITS yaw frame (rhs_fast_euler<PFRAME>), is turned by the stage's
        // yaw increment -- whose sin and cos - 1 trig_delta has anyway -- and the sum is turned by the initial yaw once at the
        // end: the stages' own sin/cos(psi) are never formed and stage 1 needs no rotation (12 instructions fewer per step).
        constexpr bool PSIF = BROV_PSI_FRAME && ANG && !GENERIC && !model_is_di(MODEL);
        double dps[2] = {0.0, 0.0};
        // stage state xs = x + c k; for the Euler-angle models the angles enter the RHS only through their sin/cos,
        // which come from the angle increments dl = c k[3..5] (trig_delta), so xs[3..5] is never formed
        auto stage_state = [&](double c, auto HALF) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if (ANG && i >= 3 && i < 6) dl[i - 3] = c * k[i];
                else xs[i] = fma(c, k[i], x[i]);
            }
            if constexpr (ANG) trig_delta<BROV_TRIG_SHORT && decltype(HALF)::value>(tb, dl, ts, PSIF ? dps : nullptr);
        };
        // horizontal position increment of a later stage: turn its p_dot by the stage's yaw increment, then weight it
        auto add_turned = [&](double w) {
            const double r0 = fma(dps[1], k[0], fma(-dps[0], k[1], k[0])), r1 = fma(dps[0], k[0], fma(dps[1], k[1], k[1]));
            xn[0] = fma(w, r0, xn[0]);
            xn[1] = fma(w, r1, xn[1]);
        };
        accel(1, a);
        rhs_fast<MODEL, GENERIC, HC, PSIF>(h, p, x, a, k, tb);
        // with a carry the three angles accumulate their INCREMENT in xn (x is added at the end), so that the increment is
        // available for the addition theorem; PSIF does the same with the horizontal position
#pragma unroll
        for (int i = 0; i < NX; ++i) xn[i] = ((CARRY && i >= 3 && i < 6) || (PSIF && i < 2)) ? h6 * k[i] : fma(h6, k[i], x[i]);
        stage_state(h2, std::true_type{});
        accel(2, a);
        rhs_fast<MODEL, GENERIC, HC, PSIF>(h, p, xs, a, k, ts);
#pragma unroll
        for (int i = PSIF ? 2 : 0; i < NX; ++i) xn[i] = fma(h3, k[i], xn[i]);
        if constexpr (PSIF) add_turned(h3);
        stage_state(h2, std::true_type{});
        accel(3, a);
        rhs_fast<MODEL, GENERIC, HC, PSIF>(h, p, xs, a, k, ts);
#pragma unroll
        for (int i = PSIF ? 2 : 0; i < NX; ++i) xn[i] = fma(h3, k[i], xn[i]);
        if constexpr (PSIF) add_turned(h3);
        stage_state(dt, std::false_type{});
        accel(4, a);
        rhs_fast<MODEL, GENERIC, HC, PSIF>(h, p, xs, a, k, ts);
        if constexpr (PSIF) {
            add_turned(h6);
            x[0] = fma(tb.cpsi, xn[0], fma(-tb.spsi, xn[1], x[0]));
            x[1] = fma(tb.spsi, xn[0], fma(tb.cpsi, xn[1], x[1]));
        }
        if (CARRY) {
#pragma unroll
            for (int i = PSIF ? 2 : 0; i < NX; ++i) {
                if (i >= 3 && i < 6) { dang[i - 3] = fma(h6, k[i], xn[i]); x[i] += dang[i - 3]; }
                else x[i] = fma(h6, k[i], xn[i]);
            }
        } else {
#pragma unroll
            for (int i = PSIF ? 2 : 0; i < NX; ++i) x[i] = fma(h6, k[i], xn[i]);
        }
    }
    if constexpr (ANG) { if (CARRY) trig_delta(tb, dang, *carry); }
    if constexpr (model_is_quat(MODEL)) quat_normalize(x + 3);
}

// THRUST: the lag bank after one step's worth of dynamics() calls (RK4 per-call mode: four samples; otherwise one).
// TRACK: also advance the per-thruster lag state X (for lag_io).
template <int INTEG, int LAGMODE, bool TRACK, bool GENERIC>
__device__ __forceinline__ void lag_step_advance(CFP pl, LagZ& lz, const double fcmd[8], const double acmd[6], double X[8][3]) {
    constexpr bool OBS = !GENERIC;                       // lag state held in the observer basis (kernels call to_observer once)
    if constexpr (INTEG == INTEG_RK4 && LAGMODE == 0) {
        if constexpr (OBS) lz.advance(pl->Aw4, pl->bw4, acmd); else lz.advance(pl->A4, pl->b4, acmd);
        if constexpr (TRACK) advance_thrusters(pl->A4, pl->b4, fcmd, X);
    } else {
        if constexpr (OBS) lz.obs_advance1(pl, acmd); else lz.advance(pl->A1, pl->b1, acmd);
        if constexpr (TRACK) advance_thrusters(pl->A1, pl->b1, fcmd, X);
    }
}
// acceleration seen by the s-th dynamics() call since the last lag_step_advance (s = 1 in per-step mode: thrust frozen)
template <int LAGMODE, bool GENERIC, class HC>
__device__ __forceinline__ void lag_stage_accel(const HC& h, CFP p, const LagZ& lz, int s, const double acmd[6], double a[6]) {
    const int se = LAGMODE == 0 ? s : 1;
    if constexpr (!GENERIC) lz.obs_accel_after(h, p, se, acmd, a); else lz.accel_after(p, se, acmd, a);
}

// Both halves in one lane (every model; the thruster model's production rollouts use rollout_pair_kernel instead).
template <int MODEL, int INTEG, int LAGMODE, bool TRACK, bool GENERIC, class HC>
__device__ __forceinline__ void step_fast(const HC& h, CFP p0, double dt, double* x, const double* u,
                                          LagZ& lz, double X[8][3], const double2* qt, Trig* carry = nullptr, bool refresh = true) {
    constexpr bool THR = (MODEL == MODEL_THRUSTER_EULER);
    CFP p = relaunder(p0);
    double fcmd[8], acmd[6];
    if constexpr (MODEL == MODEL_DI_WRENCH_QUAT) quat_normalize(x + 3);   // the reference normalises q before its update (wrench_quat.py:339)
    command_accel<MODEL, !GENERIC>(p, u, fcmd, acmd);
    integrate_fast<MODEL, INTEG, GENERIC>(h, p, dt, x, [&](int s, double* a) {
        if constexpr (THR) lag_stage_accel<LAGMODE, GENERIC>(h, p, lz, s, acmd, a);
        else {
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] = acmd[k];
        }
    }, qt, carry, refresh);
    if constexpr (THR) lag_step_advance<INTEG, LAGMODE, TRACK, GENERIC>(relaunder(p0), lz, fcmd, acmd, X);
}

}  // namespace brov
