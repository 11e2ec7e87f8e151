// qenv_fastmath.hpp -- fp64 elementary functions of the lean handle (QENV_COMPUTE_LEAN): a few ulp, branch-free,
// 25-50 VALU instructions each instead of the 100-270 of the correctly-rounded library versions (ocml), built from the
// hardware seeds v_rcp_f64 / v_rsq_f64 (2^-23 relative) + Newton steps and short near-minimax polynomials
// (tools/gen_minimax.py prints the coefficients and their fit errors).
//
// The per-step closed-loop arithmetic of the reference touches these only a handful of times per env step
// (quat_euler: 2 atan2 + asin, quaternion_euler_utility.py:39-48; LQR / PID laws: 2 atan2, cos, sin, tan,
// lqr_quad.py:128-154, pid_vel_control.py:48-127), but in IEEE form they cost more than the seven RHS evaluations of
// the integrator together (rocprofv3: 3 460 of the 4 990 VALU instructions of a lean LQR step before this file).
//
// HW supplies the two seeds, so the same code is checked on the host against libm with float-precision seeds
// (tests/test_host_cpu.py::test_lean_fastmath_accuracy).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define QENV_FM __device__ __forceinline__
#else
#define QENV_FM static inline
#endif

namespace qenv {
namespace fm {

// ---- coefficients (tools/gen_minimax.py) --------------------------------------------------------------------------
// atan(t)/t in z = t^2: degree 11 on [0, tan^2(pi/8)], max fit error 1.26e-18
static constexpr double ATAN_P[12] = {0.99999999999999999874, -0.33333333333333120948, 0.19999999999940837328,
                                      -0.14285714279244968487, 0.11111110744657099602, -0.090908968014728134434,
                                      0.07692045194561481467,  -0.06662950246036791267, 0.058468667522751389837,
                                      -0.050350497953124393166, 0.037963903161110833033, -0.017803896394812563193};
// (asin(s) - s)/s^3 in z = s^2: degree 11 on [0, 0.25], max fit error 2.35e-16
static constexpr double ASIN_R[12] = {0.16666666666666648529, 0.075000000000208696712, 0.044642857103239559945,
                                      0.030381947379559159502, 0.022372047201808885135, 0.017355268563953518903,
                                      0.013929545950485803228, 0.011876344296279416692, 0.0077986264075622330216,
                                      0.016049124584949317838, -0.010773195325585556887, 0.028187667142318098934};
// (sin(r) - r)/r^3 in z = r^2: degree 5 on [0, (pi/4)^2], max fit error 2.04e-17
static constexpr double SIN_S[6] = {-0.16666666666666664622, 0.0083333333333309472945, -0.00019841269836756772528,
                                    2.7557316101617874437e-6, -2.505113165023518147e-8, 1.5918115263265975194e-10};
// (cos(r) - 1 + z/2)/z^2 in z = r^2: degree 5 on [0, (pi/4)^2], max fit error 1.28e-18
static constexpr double COS_C[6] = {0.041666666666666665388, -0.0013888888888887396492, 0.000024801587298764560771,
                                    -2.7557317271145145962e-7, 2.0876146146558611582e-9, -1.1382623647474604629e-11};
// pi/2 = HI + MID + LO, HI and MID with 33 significant bits: k * HI and k * MID are exact for |k| < 2^20
static constexpr double PIO2_HI = 1.5707963267341256, PIO2_MID = 6.077100506303966e-11, PIO2_LO = 2.0222662487959506e-21;
static constexpr double PI = 3.14159265358979323846, PIO2 = 1.57079632679489661923, PIO4 = 0.78539816339744830962;
static constexpr double TAN_PIO8 = 0.41421356237309504880;

// 12-term polynomial as two interleaved 6-term Horner chains in z^2 (half the dependent latency of one chain)
QENV_FM double poly12(const double* c, double z) {
    const double w = z * z;
    double e = c[10], o = c[11];
    e = fma(e, w, c[8]); o = fma(o, w, c[9]);
    e = fma(e, w, c[6]); o = fma(o, w, c[7]);
    e = fma(e, w, c[4]); o = fma(o, w, c[5]);
    e = fma(e, w, c[2]); o = fma(o, w, c[3]);
    e = fma(e, w, c[0]); o = fma(o, w, c[1]);
    return fma(o, z, e);
}
QENV_FM double poly6(const double* c, double z) {
    const double w = z * z;
    double e = c[4], o = c[5];
    e = fma(e, w, c[2]); o = fma(o, w, c[3]);
    e = fma(e, w, c[0]); o = fma(o, w, c[1]);
    return fma(o, z, e);
}

template <typename HW> QENV_FM double rcp(double x) {
    double r = HW::rcp_seed(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
template <typename HW> QENV_FM double div(double a, double b) {
    const double r = rcp<HW>(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
// 1/sqrt(x) from the 2^-23 seed in ONE third-order step: with e = 1 - x r^2 (|e| <= 2^-22), 1/sqrt(x) = r (1 - e)^(-1/2) =
// r (1 + e/2 + 3 e^2/8 + O(e^3)), and the dropped 5 e^3 / 16 is below 2^-67 -- five multiply-adds after the seed where two
// Newton steps take eight (the quaternion normalisation of every right-hand-side evaluation goes through here).
template <typename HW> QENV_FM double rsqrt(double x) {
    const double r = HW::rsq_seed(x);
    const double t = x * r;
    const double e = fma(-t, r, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(r * e, p, r);
}
template <typename HW> QENV_FM double sqrt(double x) {  // 0 -> 0, negative -> NaN (as ::sqrt)
    const double xs = x == 0.0 ? 1.0 : x;
    const double r = HW::rsq_seed(xs);
    double g = xs * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    g = fma(fma(-g, g, xs), h, g);
    return x == 0.0 ? 0.0 : g;
}

// atan2(y, x): octant reduction to t = min/max in [0, 1], then atan(t) = pi/4 + atan((t - 1)/(t + 1)) above
// tan(pi/8) -- both cases as ONE quotient num/den -- and a 12-term odd polynomial on |t| <= tan(pi/8).
template <typename HW> QENV_FM double atan2(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const bool up = mn > TAN_PIO8 * mx;
    const double num = up ? mn - mx : mn, den = up ? mn + mx : mx;
    double t = div<HW>(num, den);
    t = mx == 0.0 ? 0.0 : t;                    // atan2(+-0, +-0): 0 or pi by the signs below, like libm
    const double z = t * t, w = z * z;
    // atan(t) = t P(z) with P(0) = 1 (to 1e-18): evaluated as t + t z Q(z), Q = (P - 1)/z = c1 + c2 z + ... + c11 z^10,
    // so the leading term stays exact; Q split into its even and odd powers of z (two Horner chains in w = z^2)
    double e = ATAN_P[11], o = ATAN_P[10];
    e = fma(e, w, ATAN_P[9]); o = fma(o, w, ATAN_P[8]);
    e = fma(e, w, ATAN_P[7]); o = fma(o, w, ATAN_P[6]);
    e = fma(e, w, ATAN_P[5]); o = fma(o, w, ATAN_P[4]);
    e = fma(e, w, ATAN_P[3]); o = fma(o, w, ATAN_P[2]);
    e = fma(e, w, ATAN_P[1]);        // c1 + c3 w + ... + c11 w^5
    const double q = fma(o, z, e);   // + z (c2 + c4 w + ... + c10 w^4)
    double a = fma(t * z, q, t);     // in [-(pi/8), pi/8]
    a = up ? a + PIO4 : a;           // atan(min/max) in [0, pi/4]
    a = ay > ax ? PIO2 - a : a;
    a = signbit(x) ? PI - a : a;
    return copysign(a, y);
}

// asin(x): |x| <= 1/2 directly, above through asin(x) = pi/2 - 2 asin(sqrt((1 - x)/2)); one polynomial for both.
template <typename HW> QENV_FM double asin(double x) {
    const double a = fabs(x);
    const bool big = a > 0.5;
    const double z = big ? fma(-0.5, a, 0.5) : a * a;
    const double s = big ? sqrt<HW>(z) : a;            // |x| > 1: z < 0 -> NaN, as the reference's arcsin (quirk Q6)
    const double p = fma(s * z, poly12(ASIN_R, z), s);
    const double r = big ? fma(-2.0, p, PIO2) : p;
    return copysign(r, x);
}

// sin and cos of one argument: Cody-Waite reduction by pi/2 (|x| < ~1e5), two 6-term polynomials
template <typename HW> QENV_FM void sincos(double x, double& sn, double& cs) {
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, PIO2_HI, x);
    r = fma(-k, PIO2_MID, r);
    r = fma(-k, PIO2_LO, r);
    const double z = r * r;
    const double s = fma(r * z, poly6(SIN_S, z), r);
    const double c = fma(z * z, poly6(COS_C, z), fma(-0.5, z, 1.0));
    const int q = (int)k;
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    sn = (q & 2) ? -s1 : s1;
    cs = ((q + 1) & 2) ? -c1 : c1;
}
template <typename HW> QENV_FM double sin(double x) { double s, c; sincos<HW>(x, s, c); return s; }
template <typename HW> QENV_FM double cos(double x) { double s, c; sincos<HW>(x, s, c); return c; }
template <typename HW> QENV_FM double tan(double x) { double s, c; sincos<HW>(x, s, c); return div<HW>(s, c); }

}  // namespace fm
}  // namespace qenv
