// qenv_probe_bound.hpp -- when scipy's select_initial_step does not need its second right-hand-side evaluation.
//
// select_initial_step (scipy/integrate/_ivp/common.py:66-133, restated in rk45_initial_step) evaluates f1 = f(y + h0 f0) only to
// form d2 = ||(f1 - f0) / scale|| / h0, and uses d2 only through h1 = (0.01 / max(d1, d2))^(1/5) in
//     h_abs = min(100 h0, h1, t_bound).
// If max(d1, d2) <= 0.01 / t_bound^5 (1e8 at t_bound = 0.01) then h1 >= t_bound and h_abs = min(100 h0, t_bound) WHATEVER d2 is.
// This file PROVES that inequality from y, f0 and h0 alone, with a Lipschitz bound of the nominal plant's right-hand side
// along the segment y .. y + delta, delta = h0 f0: the evaluation (89 multiply-adds, the stage value, the third norm, a
// division, a square root and a power: ~155 of the 1 330 vector instructions of a dynamics-only env step) is skipped when the
// proof succeeds and made as before when it does not, so h_abs -- hence the sub-step partition and every result -- is scipy's
// either way.  On the states of BASELINE's configs the proof succeeds for > 99.9 % of the (env, step) pairs
// (tests/test_host_cpu.py::test_probe_bound_is_conservative fuzzes the inequality bound >= d2 against the CPU restatement of f).
//
// The bound, with |.|_1 the sum of absolute values (>= the Euclidean norm) and a ' marking the far end of the segment:
//   positions   f = v                         |df_k|  = delta_v,k                                  (exact)
//   quaternion  qh = q / |q|                  |dqh|   <= 2 |delta_q| / |q|  <= 4 |delta_q|_1       (|q| >= 1/2 is checked)
//   rotation    R(qh) x = qh x qh*            ||dR||  <= 2 |dqh|                                   (quaternion product, unit norm)
//   velocities  a = (R fb - [0 0 M G]) / M,  fb = [0 0 F] - KD |vb| vb,  vb = R^T v,   with vn = |v|_1 + |delta_v|_1:
//               |fb'| <= |F| + KD vn^2,   |dvb| <= ||dR|| vn + |delta_v|,   |dfb| <= 2 KD vn |dvb|      (| |a|a - |b|b | <= (|a|+|b|)|a-b|)
//               |da_k| <= (||dR|| (|F| + KD vn^2) + 2 KD vn (||dR|| vn + |delta_v|_1)) / M
//   q rates     qd = Omega(w) qh / 2,  ||Omega(w)|| = |w|:    |dqd_k| <= (|delta_w|_1 + |w|_1 |dqh|) / 2
//   body rates  al_x = Mx/J - wx omr/J - (cm/J) |wx| wx - wy wz (Jz - Jy)/Jx  (y, z alike), with wb_k = |w_k| + delta_w,k:
//               |dal_x| <= (|omr/J| + 2 (cm/J) wb_x) delta_wx + |(Jz - Jy)/Jx| (wb_y delta_wz + wb_z delta_wy)
//                       <= (|omr/J| + 2 (cm/J + c) wb) |delta_w|_1       with wb = |w|_1 + |delta_w|_1 and the largest cm/J, c
// and d2 <= sqrt(mean_i (bound_i / scale_i)^2) / h0.  Every term is an upper bound in exact arithmetic; the comparison keeps a
// factor 2 below the threshold, which covers the rounding of these few operations (and of scipy's own d2) many times over.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define QENV_PB __device__ __forceinline__
#else
#define QENV_PB static inline
#endif

namespace qenv {

// plant constants the bound needs (nominal plant; the caller passes what its right-hand side uses)
template <typename T> struct ProbePlant {
    T F_abs;        // |thrust| held over the step
    T omrj_abs;     // |rotor gyroscopic factor / J_xx|
    T cmj_xy, cmj_z;  // drag-moment coefficients / inertia
    T kd;           // largest translational drag coefficient
    T inv_m;        // 1 / mass
    T cx, cy, cz;   // |(Jz - Jy)/Jx|, |(Jx - Jz)/Jy|, |(Jy - Jx)/Jz|
};

// true: max(d1, d2) <= thr is proven (so h1 >= t_bound); false: not proven (evaluate the probe).  y, f0: state and f(y); isc[i] =
// 1 / (atol + rtol |y_i|); h0, d1 as select_initial_step has them; thr = 0.01 / t_bound^5.
// s' with d2 <= sqrt(s' / 13) for EVERY h0 in (0, h_max]: delta = h0 f0 is linear in h0, so each bound above is h0 times a factor
// that depends on h0 only through the segment's far end (vn, wb), where h0 <= h_max is used -- select_initial_step has clamped
// h0 to t_bound.  The test then needs no h0 at all and its arithmetic is independent of the norms d0, d1 that produce h0 (the
// two chains overlap on a lone wave).  Within a group of components (velocities, quaternion, body rates) ONE bound serves all of
// them (for the body rates: every w_k and every delta_w,k replaced by the group's |.|_1).  The caller checks |q|^2 >= 1/4.
template <typename T>
QENV_PB T probe_bound_sum(const T y[13], const T f0[13], const T isc[13], const ProbePlant<T>& p, T h_max) {
    const T g1 = fabs(f0[1]), g3 = fabs(f0[3]), g5 = fabs(f0[5]);
    const T dv = g1 + g3 + g5;                                             // |delta_v|_1 / h0
    const T dq = (fabs(f0[6]) + fabs(f0[7])) + (fabs(f0[8]) + fabs(f0[9]));   // |delta_q|_1 / h0
    const T dw = fabs(f0[10]) + fabs(f0[11]) + fabs(f0[12]);               // |delta_w|_1 / h0
    const T vn = ((fabs(y[1]) + fabs(y[3])) + fabs(y[5])) + h_max * dv;    // >= |v|_1 anywhere on the segment
    const T wn = fabs(y[10]) + fabs(y[11]) + fabs(y[12]);
    const T dR = T(8) * dq;                       // ||dR|| <= 2 |dqh| <= 8 |delta_q|_1
    const T kv = p.kd * vn;
    const T ba = (dR * (p.F_abs + kv * vn) + (kv + kv) * (dR * vn + dv)) * p.inv_m;
    const T bq = T(0.5) * dw + (wn + wn) * dq;    // (|delta_w|_1 + |w|_1 4 |delta_q|_1) / 2
    const T cmj = p.cmj_xy > p.cmj_z ? p.cmj_xy : p.cmj_z;
    const T cj = p.cx > p.cy ? (p.cx > p.cz ? p.cx : p.cz) : (p.cy > p.cz ? p.cy : p.cz);
    const T bw = (p.omrj_abs + (T(2) * (cmj + cj)) * (wn + h_max * dw)) * dw;
    const T a0 = g1 * isc[0], a2 = g3 * isc[2], a4 = g5 * isc[4];   // positions: df_k = delta_v,k exactly
    const T sp = a0 * a0 + a2 * a2 + a4 * a4;
    const T sv = isc[1] * isc[1] + isc[3] * isc[3] + isc[5] * isc[5];
    const T sq = (isc[6] * isc[6] + isc[7] * isc[7]) + (isc[8] * isc[8] + isc[9] * isc[9]);
    const T sw = isc[10] * isc[10] + isc[11] * isc[11] + isc[12] * isc[12];
    return (sp + (ba * ba) * sv) + ((bq * bq) * sq + (bw * bw) * sw);
}

// true: max(d1, d2) <= thr is proven for the h0 <= h_max that select_initial_step uses (so h1 >= t_bound); false: not proven
// (evaluate the probe).  y, f0: state and f(y); isc[i] = 1 / (atol + rtol |y_i|); d1 as select_initial_step has it; thr = 0.01 /
// t_bound^5.
template <typename T>
QENV_PB bool probe_skippable(const T y[13], const T f0[13], const T isc[13], const ProbePlant<T>& p, T h_max, T d1, T thr) {
    const T half_thr = T(0.5) * thr;
    const T n2 = y[6] * y[6] + y[7] * y[7] + y[8] * y[8] + y[9] * y[9];
    const T s = probe_bound_sum(y, f0, isc, p, h_max);
    // d2 <= sqrt(s / 13) <= thr / 2   <=>   s <= 13 (thr / 2)^2; every test is false on NaN (not proven); one conjunction, no
    // early exits: the lanes of a wave take the same path here
    // (d1 > 2e-15: scipy's other branch needs d1 <= 1e-15;  d1 <= thr / 2: the d1 half of the maximum)
    return (d1 > T(2e-15)) & (d1 <= half_thr) & (n2 >= T(0.25)) & (s <= T(13) * half_thr * half_thr);
}

}  // namespace qenv
