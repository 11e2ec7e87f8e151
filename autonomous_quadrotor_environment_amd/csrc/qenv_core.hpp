// qenv_core.hpp -- device-side math of the quadrotor hot path (gfx950 / CDNA4).
//
// One wavefront lane integrates one quadrotor.  Everything here is written so that
// the whole env-step lives in VGPRs: fully unrolled Butcher tableaus, no local arrays
// indexed at run time, wave-uniform configuration in SGPRs (kernel arguments).
//
// Reference behaviour restated (file:line relative to /root/reference):
//   environment/quadrotor_env.py            f2F :247, f2w :197, drone_eq :274, step :458,
//                                           done_condition :500, reward_function :511
//   environment/quadrotor_env_opt.py        the "opt" variant of the same functions
//   environment/quaternion_euler_utility.py quat_euler :39, deriv_quat :58, quat_rot_mat :71
//   scipy/integrate/_ivp/{rk,common}.py     RK45 / select_initial_step (third party)
//   environment/controller/lqr_quad.py:128-154, pid_vel_control.py:29-127
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qenv_fastmath.hpp"
#include "qenv_probe_bound.hpp"

namespace qenv {

// ---- scalar math per precision --------------------------------------------------------
template <typename T> struct Mth;
template <> struct Mth<float> {
    static __device__ __forceinline__ float sqrt(float x) { return __fsqrt_rn(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return rsqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
    static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
    static __device__ __forceinline__ float min(float a, float b) { return fminf(a, b); }
    static __device__ __forceinline__ float max_abs(float a, float b) { return fmaxf(fabsf(a), fabsf(b)); }
    static __device__ __forceinline__ float atan2(float y, float x) { return atan2f(y, x); }
    static __device__ __forceinline__ float asin(float x) { return asinf(x); }
    static __device__ __forceinline__ float cos(float x) { return cosf(x); }
    static __device__ __forceinline__ float sin(float x) { return sinf(x); }
    static __device__ __forceinline__ float tan(float x) { return tanf(x); }
    static __device__ __forceinline__ float pow(float x, float y) { return powf(x, y); }
    // step-size control only (x > 0): v_log_f32 / v_exp_f32, ~1e-6 relative -- the factor multiplies a
    // step size that is then clamped to the remaining interval, it never touches the state directly
    static __device__ __forceinline__ float pow_ctl(float x, float y) {
        return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
    }
    static __device__ __forceinline__ float div_ctl(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static __device__ __forceinline__ float div(float a, float b) { return a / b; }
    static __device__ __forceinline__ float rcp_ctl(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float sqrt_ctl(float x) { return __fsqrt_rn(x); }
    static constexpr bool lean = false;
    // nextafter(t, +inf) - t for finite t >= 0
    static __device__ __forceinline__ float ulp_up(float t) { return __uint_as_float(__float_as_uint(t) + 1u) - t; }
};
template <> struct Mth<double> {
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double abs(double x) { return fabs(x); }
    static __device__ __forceinline__ double max(double a, double b) { return fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return fmin(a, b); }
    static __device__ __forceinline__ double max_abs(double a, double b) { return fmax(fabs(a), fabs(b)); }
    static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double asin(double x) { return ::asin(x); }
    static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
    static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
    static __device__ __forceinline__ double tan(double x) { return ::tan(x); }
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    static __device__ __forceinline__ double pow_ctl(double x, double y) { return ::pow(x, y); }
    static __device__ __forceinline__ double div_ctl(double a, double b) { return a / b; }
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
    static __device__ __forceinline__ double rcp_ctl(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double sqrt_ctl(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double ulp_up(double t) { return nextafter(t, (double)INFINITY) - t; }
    static constexpr bool lean = false;
};

// "Lean" fp64 (QENV_COMPUTE_LEAN): every operation that touches the STATE is an fp64 multiply-add -- on gfx950 a wave
// alone on its SIMD issues v_fma_f64 at the rate of v_fma_f32 (tools/ubench/valu_rate.hip: 5.1 vs 5.1 cycles) -- while
// what made the plain fp64 handle twice as slow as fp32 is replaced:
//   * IEEE divisions / square roots (15-30 instructions each from the compiler's correctly-rounded expansions) by the
//     hardware seed (v_rcp_f64 / v_rsq_f64, 2^-23 relative) + two Newton steps: <= 2 ulp, 9 instructions;
//   * the step-size CONTROL (scaled error norm, select_initial_step, growth factor) by the raw hardware seeds and fp32
//     log/exp: a control quantity only scales a step size that is then clamped to the rest of the interval; the one
//     decision that selects between different results -- accept/reject at err = 1 -- is re-evaluated with exact
//     divisions whenever the fast norm lands within 1e-5 of the threshold (dp5_step), so the accepted sub-steps are
//     those of the fp64 handle.
struct HwSeed {  // v_rcp_f64 / v_rsq_f64: 2^-23 relative
    static __device__ __forceinline__ double rcp_seed(double x) { return __builtin_amdgcn_rcp(x); }
    static __device__ __forceinline__ double rsq_seed(double x) { return __builtin_amdgcn_rsq(x); }
};
struct MthLean {
    static constexpr bool lean = true;
    static __device__ __forceinline__ double rsqrt(double x) { return fm::rsqrt<HwSeed>(x); }
    static __device__ __forceinline__ double sqrt(double x) { return fm::sqrt<HwSeed>(x); }
    static __device__ __forceinline__ double rcp(double x) { return fm::rcp<HwSeed>(x); }
    static __device__ __forceinline__ double div(double a, double b) { return fm::div<HwSeed>(a, b); }
    static __device__ __forceinline__ double abs(double x) { return fabs(x); }
    static __device__ __forceinline__ double max(double a, double b) { return fmax(a, b); }
    static __device__ __forceinline__ double min(double a, double b) { return fmin(a, b); }
    // max(|a|, |b|) as ONE v_max_f64 with both source modifiers (fmax(fabs, fabs) compiles to the maximum plus a quieting
    // v_max_f64 x, x of an operand: 13 of those per attempted sub-step in the error norm's scale)
    static __device__ __forceinline__ double max_abs(double a, double b) {
        double r;
        asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double atan2(double y, double x) { return fm::atan2<HwSeed>(y, x); }
    static __device__ __forceinline__ double asin(double x) { return fm::asin<HwSeed>(x); }
    static __device__ __forceinline__ double cos(double x) { return fm::cos<HwSeed>(x); }
    static __device__ __forceinline__ double sin(double x) { return fm::sin<HwSeed>(x); }
    static __device__ __forceinline__ double tan(double x) { return fm::tan<HwSeed>(x); }
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    // control only
    static __device__ __forceinline__ double pow_ctl(double x, double y) {
        return (double)__builtin_amdgcn_exp2f((float)y * __builtin_amdgcn_logf((float)x));
    }
    static __device__ __forceinline__ double rcp_ctl(double x) { return __builtin_amdgcn_rcp(x); }
    static __device__ __forceinline__ double div_ctl(double a, double b) { return a * __builtin_amdgcn_rcp(b); }
    static __device__ __forceinline__ double sqrt_ctl(double x) { return __builtin_amdgcn_sqrt(x); }
    // x^(-1/5) to ~1 ulp: the fp32 estimate (2^-23) and two Newton steps on y^-5 = x, y <- y (1.2 - 0.2 x y^5) -- fourteen
    // instructions where ocml's pow is hundreds (inlined into the attempt loop it cost the rollouts 14-22 %).  x = 0 / inf / NaN
    // give inf / NaN, which the callers' fmin / fmax clamp exactly as they clamped the estimate.
    static __device__ __forceinline__ double pow_m02(double x) {
        double y = pow_ctl(x, -0.2);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const double y2 = y * y;
            y = y * fma(-0.2 * x, y2 * y2 * y, 1.2);
        }
        return y;
    }
    static __device__ __forceinline__ double ulp_up(double t) { return nextafter(t, (double)INFINITY) - t; }
};
template <typename T, bool LEAN> struct MathSel { using type = Mth<T>; };
template <> struct MathSel<double, true> { using type = MthLean; };

// Two envs per lane: every add/mul/fma of the integrator becomes ONE packed instruction
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) working on both.  Measured on MI355X: a wave64 fp32
// VALU instruction issues every ~4 cycles per SIMD whatever the occupancy (profiles/), so the only way
// to the chip's fp32 vector peak is two floats per lane per instruction.
using F2 = float __attribute__((ext_vector_type(2)));
template <> struct Mth<F2> {
    static __device__ __forceinline__ F2 sqrt(F2 x) { return F2{__fsqrt_rn(x.x), __fsqrt_rn(x.y)}; }
    static __device__ __forceinline__ F2 rsqrt(F2 x) { return F2{rsqrtf(x.x), rsqrtf(x.y)}; }
    static __device__ __forceinline__ F2 rcp(F2 x) { return F2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
    static __device__ __forceinline__ F2 abs(F2 x) { return __builtin_elementwise_abs(x); }
    static __device__ __forceinline__ F2 max(F2 a, F2 b) { return __builtin_elementwise_max(a, b); }
    static __device__ __forceinline__ F2 min(F2 a, F2 b) { return __builtin_elementwise_min(a, b); }
    static __device__ __forceinline__ F2 max_abs(F2 a, F2 b) { return __builtin_elementwise_max(__builtin_elementwise_abs(a), __builtin_elementwise_abs(b)); }
    static __device__ __forceinline__ F2 rcp_ctl(F2 x) { return rcp(x); }
    static constexpr bool lean = false;
};

// ---- physical constants (quadrotor_env.py:30-81), folded at compile time ----------------
namespace phys {
constexpr double M = 1.03, G = 9.82, RHO = 1.2041, C_D = 1.1;
constexpr double K_F = 1.435e-5, K_M = 2.4086e-7, I_R = 5e-5, T2WR = 2.0;
constexpr double JX = 16.83e-3, JY = 16.83e-3, JZ = 28.34e-3;
constexpr double D = 0.26, BEAM = 0.05;
constexpr double AX = BEAM * 2 * D, AY = BEAM * 2 * D, AZ = BEAM * 2 * D * 2;
constexpr double PI = 3.14159265358979323846;
constexpr double BB_VEL = 10.0, BB_ANG = PI / 2;
// drag force coefficient 0.5 rho C_d A_k (drone_eq :323)
constexpr double KDX = 0.5 * RHO * C_D * AX, KDY = 0.5 * RHO * C_D * AY, KDZ = 0.5 * RHO * C_D * AZ;
// drag moments (drone_eq :328-334): sum over xx_j = D j/9, j=0..9, of xx^3 (main) or
// xx^2 (opt, quadrotor_env_opt.py:260-263)
constexpr double SUM_XX3 = D * D * D * 2025.0 / 729.0;
constexpr double SUM_XX2 = D * D * 285.0 / 81.0;
constexpr double CM0 = RHO * C_D * BEAM * D / 10.0;
constexpr double F_MAX_ROTOR = T2WR * M * G / 4.0;   // 5.0573 N
constexpr double U_MAX = F_MAX_ROTOR / K_F;          // max squared rotor speed
}  // namespace phys

// Wave-uniform configuration, passed by value in the kernel argument segment (SGPRs).
template <typename T> struct Cfg {
    T t_step;
    T cm_xy, cm_z;   // drag-moment coefficients of the selected variant
    T cmj_xy, cmj_z; // ... divided by the nominal body inertia (J_xx = J_yy, J_zz): the nominal right-hand side works in angular accelerations
    T bb[9];         // done box: |v|(3), |ang|(3), |omega|(3)
    T zero_control0; // zero_control = [zc0,0,0,0] (indirect) or [zc0]*4 (direct, zc0=0)
    int32_t n_max;   // n + T
    int32_t warmup;  // T: warm-up steps of reset = window depth
    int32_t training, direct, clipped, variant;
};

// Actuation held constant over one env step: thrust, body moments and the rotor
// gyroscopic term (drone_eq :287-292,:345).  Hoisted out of the RHS: the reference
// recomputes f2F (4 sqrt) in every RHS evaluation although the action is constant.
template <typename T> struct Act {
    T F, Mx, My, Mz, omr;  // omr = (-w1+w2-w3+w4)*I_R
    // nominal plant: the moments and the gyroscopic factor divided by the body inertia, once per env step (act_finish) instead
    // of one multiplication per axis in every right-hand-side evaluation
    T Mxj, Myj, Mzj, omrj;
#ifdef QENV_HOVER_REL   /* experiment (round 5, not kept -- profiles/r05_experiments_not_kept.txt): F = M G (1 + dlt) */
    T dlt; int rel;
#endif
};
static_assert(phys::JX == phys::JY, "act_finish / rhs use one gyroscopic factor omr / J_xx for the x and y axes");
// two envs per lane (k_rollout2): the actuation of env 0 / env 1 in the two halves of the packed type
template <typename T2, typename T1> __device__ __forceinline__ void pack_act(const Act<T1>& a, const Act<T1>& b, Act<T2>& u) {
    u.F = T2{a.F, b.F}; u.Mx = T2{a.Mx, b.Mx}; u.My = T2{a.My, b.My}; u.Mz = T2{a.Mz, b.Mz}; u.omr = T2{a.omr, b.omr};
    u.Mxj = T2{a.Mxj, b.Mxj}; u.Myj = T2{a.Myj, b.Myj}; u.Mzj = T2{a.Mzj, b.Mzj}; u.omrj = T2{a.omrj, b.omrj};
}
template <typename T> __device__ __forceinline__ void act_finish(Act<T>& u) {
    u.Mxj = u.Mx * T(1.0 / phys::JX); u.Myj = u.My * T(1.0 / phys::JY); u.Mzj = u.Mz * T(1.0 / phys::JZ);
    u.omrj = u.omr * T(1.0 / phys::JX);
}

// The plant the RHS integrates.  Nominal: the constants above plus the variant's drag-moment coefficients.
// Robust (robust_control, quadrotor_env.py:84-109, flag :183): the episode's perturbed thrust coefficients, mass,
// rotor inertias and body inertia, and the gust process -- per env, in registers for the whole step.
template <typename T, bool ROBUST, bool LEAN = false> struct Plant;
template <typename T, bool LEAN> struct Plant<T, false, LEAN> {
    static constexpr bool robust = false;
    using math = typename MathSel<T, LEAN>::type;
    T cmj_xy, cmj_z;   // drag-moment coefficients / body inertia
    // select_initial_step's second evaluation skipped where h1 >= t_bound is proven (qenv_probe_bound.hpp).  Set by the kernels
    // it pays for: the proof costs ~5 % of a dynamics-only env step and saves ~12 % (constant-action rollouts: +3..8 %); the
    // closed-loop steps spend their time elsewhere (config 3 lost 3 % with it)
    bool skip_probe = false;
};
constexpr int GUST_PERIOD = 500;  // robust_control.gust_period (:94)
template <typename T, bool LEAN> struct Plant<T, true, LEAN> {
    static constexpr bool robust = true;
    using math = typename MathSel<T, LEAN>::type;
    T cm_xy, cm_z;
    T kf[4];           // episode_kf: fraction of thrust lost per rotor (:98)
    T ir[4];           // I_R (1 + episode_ir) (:342)
    T inv_m, mg;       // 1/quad_m and quad_m G, quad_m = M (1 + episode_m) (:361)
    T ij[3];           // 1 / (J_k + J_k episode_J_k) (:382)
    T gust[3], last[3];  // robust_control.gust / last_gust: end points of linear_wind_change
    T wind[3];         // this step's wind when it is not a redraw step
    bool redraw;       // i % 500 == 1: EVERY RHS evaluation of this step draws a new gust and sees the previous one
    int32_t k;         // RHS evaluations of this step so far = index of the next draw
    int32_t evals;     // draws consumed by the most recent redraw step seen by this lane (-1: none yet)
    // draw source: an injected table [E][K][3][N] (parity runs) or Philox keyed by (seed; env, episode, epoch, k)
    const void* table; int32_t table_f64, table_E, table_K, epoch;
    int64_t N, e;      // row stride and this lane's env
    uint32_t key0, key1, ctr0, ctr1, episode;
};

// the plant's wave-uniform coefficients from the configuration (nominal: already divided by the body inertia; robust: the
// episode's own inertia divides the moments, so the plain coefficients)
template <typename T, bool LEAN> __device__ __forceinline__ void set_plant(Plant<T, false, LEAN>& pl, const Cfg<T>& cfg) {
    pl.cmj_xy = cfg.cmj_xy; pl.cmj_z = cfg.cmj_z;
}
template <typename T, bool LEAN> __device__ __forceinline__ void set_plant(Plant<T, true, LEAN>& pl, const Cfg<T>& cfg) {
    pl.cm_xy = cfg.cm_xy; pl.cm_z = cfg.cm_z;
}

// Philox4x32-10 (Salmon et al. 2011), the same generator qenv_ppo.hip uses for actions and initial states.
__device__ __forceinline__ void philox_core(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float philox_u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ void philox_normal2(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float r = sqrtf(-2.0f * logf(philox_u01(a)));
    float sn, cs;
    sincosf(6.283185307179586f * philox_u01(b), &sn, &cs);
    z0 = r * cs; z1 = r * sn;
}

// Random branch of quad.reset (quadrotor_env.py:440-445) from Philox streams keyed by (seed; global env g, episode):
// Euler angles U(-.5,.5) -> quaternion, position / velocity / body rate N(0,2) clipped to +-2.5 / +-5 / [-15, 7.5].
// One definition for qenv_sample_init_states (qenv_ppo.hip) and the fused masked reset (qenv_env.hip).
__device__ __forceinline__ void sample_init_state(uint64_t seed, uint64_t episode, uint64_t g, float x[13]) {
    float u[4], z[12];
    for (int blk = 0; blk < 4; ++blk) {
        uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)episode, ((uint32_t)(episode >> 32) & 0x0FFFFFFFu) | ((uint32_t)blk << 28)};
        philox_core(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x1417u);
        if (blk == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = philox_u01(c[k]);
        } else {
            philox_normal2(c[0], c[1], z[4 * (blk - 1)], z[4 * (blk - 1) + 1]);
            philox_normal2(c[2], c[3], z[4 * (blk - 1) + 2], z[4 * (blk - 1) + 3]);
        }
    }
    const float phi = u[0] - 0.5f, theta = u[1] - 0.5f, psi = u[2] - 0.5f;
    float sp, cp, st_, ct, ss, cs;
    sincosf(0.5f * phi, &sp, &cp); sincosf(0.5f * theta, &st_, &ct); sincosf(0.5f * psi, &ss, &cs);
    const float q[4] = {cp * ct * cs + sp * st_ * ss, sp * ct * cs - cp * st_ * ss, cp * st_ * cs + sp * ct * ss,
                        cp * ct * ss - sp * st_ * cs};
    const float inv = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    auto clip = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x[2 * k] = clip(2.0f * z[k], -2.5f, 2.5f);          // position
        x[2 * k + 1] = clip(2.0f * z[3 + k], -5.0f, 5.0f);  // velocity
        x[10 + k] = clip(2.0f * z[6 + k], -15.0f, 7.5f);    // body rates (sic: asymmetric, :445)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) x[6 + k] = q[k] * inv;
}

// One gust triple ~ N(0, [5,5,2]) (gust_std, :93): draw k of redraw epoch `epoch`.  Kept out of line on purpose: it
// is needed only in the steps with i % 500 == 1, and inlining Philox + Box-Muller (logf, sincosf) into each of the
// integrator's ~20 RHS call sites bloats the robust kernels into SGPR/VGPR spilling.
template <typename T> struct Gust3 { T x, y, z; };
template <typename T>
__device__ __noinline__ Gust3<T> draw_gust(const void* table, int32_t f64, int32_t E, int32_t K, int32_t epoch, int32_t k,
                                           int64_t N, int64_t e, uint32_t key0, uint32_t key1, uint32_t ctr0, uint32_t ctr1,
                                           uint32_t episode) {
    Gust3<T> g;
    if (table) {
        const int kk = k < K ? k : K - 1;
        const int ee = epoch < E ? epoch : E - 1;
        const int64_t base = ((int64_t)(ee * K + kk) * 3) * N + e;
        if (f64) {
            const double* t = (const double*)table;
            g.x = (T)t[base]; g.y = (T)t[base + N]; g.z = (T)t[base + 2 * N];
        } else {
            const float* t = (const float*)table;
            g.x = (T)t[base]; g.y = (T)t[base + N]; g.z = (T)t[base + 2 * N];
        }
    } else {
        uint32_t c[4] = {ctr0, ctr1, episode, ((uint32_t)epoch << 16) | (uint32_t)(k & 0xFFFF)};
        philox_core(c, key0, key1);
        float z0, z1, z2, z3;
        philox_normal2(c[0], c[1], z0, z1);
        philox_normal2(c[2], c[3], z2, z3);
        g.x = T(5) * (T)z0; g.y = T(5) * (T)z1; g.z = T(2) * (T)z2;
    }
    return g;
}

// robust_control.wind(i) as the k-th RHS evaluation of a step sees it (:102-108)
template <typename T, bool LEAN>
__device__ __forceinline__ void next_wind(Plant<T, true, LEAN>& pl, T wd[3]) {
    if (pl.redraw) {
        const Gust3<T> g = draw_gust<T>(pl.table, pl.table_f64, pl.table_E, pl.table_K, pl.epoch, pl.k, pl.N, pl.e,
                                        pl.key0, pl.key1, pl.ctr0, pl.ctr1, pl.episode);
        pl.last[0] = pl.gust[0]; pl.last[1] = pl.gust[1]; pl.last[2] = pl.gust[2];
        pl.gust[0] = g.x; pl.gust[1] = g.y; pl.gust[2] = g.z;
        wd[0] = pl.last[0]; wd[1] = pl.last[1]; wd[2] = pl.last[2];
        pl.evals = pl.k + 1;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) wd[c] = pl.wind[c];
    }
    pl.k += 1;
}

// Start of a step with step counter i (already incremented, :467): decide redraw / interpolate the wind.
// linear_wind_change = np.linspace(last_gust, gust, 500); wind(i) returns row (i % 500) - 1, row -1 = the end.
template <typename T, bool LEAN>
__device__ __forceinline__ void begin_step_wind(Plant<T, true, LEAN>& pl, int32_t i) {
    const int r = i % GUST_PERIOD;
    pl.redraw = (r == 1);
    pl.k = 0;
    pl.epoch = (i - 1) / GUST_PERIOD;
    const int j = (r == 0) ? GUST_PERIOD - 1 : r - 1;
    T step[3];
    bool any_zero = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        step[c] = (pl.gust[c] - pl.last[c]) / T(GUST_PERIOD - 1);
        any_zero = any_zero || (step[c] == T(0));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // numpy.linspace: j*step + start, or (j/div)*delta + start when any step is 0; the last row is `stop` exactly
        if (j == GUST_PERIOD - 1) pl.wind[c] = pl.gust[c];
        else if (any_zero) pl.wind[c] = (T(j) / T(GUST_PERIOD - 1)) * (pl.gust[c] - pl.last[c]) + pl.last[c];
        else pl.wind[c] = T(j) * step[c] + pl.last[c];
    }
}

// f2F, quadrotor_env.py:247-272 (direct mode).  a = clipped action in [-1,1].
template <typename T, typename P>
__device__ __forceinline__ void mix_direct(const T a[4], const P& pl, T w[4], Act<T>& u) {
    using m = typename P::math;
    const T c = T(phys::M * phys::G) * T(phys::T2WR / 8.0);  // fl(M G)/4: four hover rotors sum to fl(M G) exactly
    T f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = (a[k] + T(1)) * c;
        w[k] = m::sqrt(f[k] * T(1.0 / phys::K_F));
        if constexpr (P::robust) f[k] = f[k] - pl.kf[k] * f[k];  // after w (:265-266)
    }
    u.F = (f[0] + f[1]) + (f[2] + f[3]);
    u.Mx = (f[2] - f[0]) * T(phys::D);
    u.My = (f[1] - f[3]) * T(phys::D);
    u.Mz = ((f[1] - f[0]) + (f[3] - f[2])) * T(phys::K_M / phys::K_F);
    if constexpr (P::robust) u.omr = ((w[1] * pl.ir[1] - w[0] * pl.ir[0]) + (w[3] * pl.ir[3] - w[2] * pl.ir[2]));
    else { u.omr = ((w[1] - w[0]) + (w[3] - w[2])) * T(phys::I_R); act_finish(u); }
#ifdef QENV_HOVER_REL
    u.dlt = ((a[0] + a[1]) + (a[2] + a[3])) * T(0.25); u.rel = P::robust ? 0 : 1;   // T2WR = 2: F = M G (1 + sum a / 4)
#endif
}

// f2w, quadrotor_env.py:197-245 / quadrotor_env_opt.py:144-176 (indirect mode), with the
// closed-form inverse of the 4x4 mixing matrix.
template <typename T, typename P>
__device__ __forceinline__ void mix_indirect(const T act[4], bool clipped, bool opt, const P& pl, T w[4], T effort[4],
                                             T cl[4], Act<T>& u) {
    using m = typename P::math;
    const T a = act[0] * T(1.0 / (4 * phys::K_F));
    const T bx = act[1] * T(1.0 / (2 * phys::D * phys::K_F));
    const T by = act[2] * T(1.0 / (2 * phys::D * phys::K_F));
    const T c = act[3] * T(1.0 / (4 * phys::K_M));
    T s[4] = {a - bx - c, a + by + c, a + bx - c, a - by + c};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (clipped || opt) {
            s[k] = m::min(m::max(s[k], T(0)), T(phys::U_MAX));
            w[k] = m::sqrt(s[k]);
        } else {
            w[k] = s[k] < T(0) ? -m::sqrt(-s[k]) : m::sqrt(s[k]);
        }
        if constexpr (P::robust) s[k] -= s[k] * pl.kf[k];  // u -= u*episode_kf after w (:235-236)
        effort[k] = opt ? s[k] * T(phys::K_F) : s[k] * T(phys::K_F / phys::F_MAX_ROTOR * 2.0) - T(1);
    }
    u.F = T(phys::K_F) * ((s[0] + s[1]) + (s[2] + s[3]));
    u.Mx = T(phys::D * phys::K_F) * (s[2] - s[0]);
    u.My = T(phys::D * phys::K_F) * (s[1] - s[3]);
    u.Mz = T(phys::K_M) * ((s[1] - s[0]) + (s[3] - s[2]));
    if constexpr (P::robust) u.omr = ((w[1] * pl.ir[1] - w[0] * pl.ir[0]) + (w[3] * pl.ir[3] - w[2] * pl.ir[2]));
    else { u.omr = ((w[1] - w[0]) + (w[3] - w[2])) * T(phys::I_R); act_finish(u); }
    cl[0] = u.F; cl[1] = u.Mx; cl[2] = u.My; cl[3] = u.Mz;
#ifdef QENV_HOVER_REL
    u.dlt = T(0); u.rel = 0;
#endif
}

// State layout (quadrotor_env.py:399-400): x vx y vy z vz q0 q1 q2 q3 wx wy wz
// RHS drone_eq, quadrotor_env.py:274-406, closed form.
// `acc` returns the inertial acceleration (self.accel side output).
template <typename T, typename P>
__device__ __forceinline__ void rhs(const T y[13], const Act<T>& u, P& pl, T dy[13]) {
    using m = typename P::math;
    const T vx = y[1], vy = y[3], vz = y[5];
    const T wx = y[10], wy = y[11], wz = y[12];
    const T inv = m::rsqrt(y[6] * y[6] + y[7] * y[7] + y[8] * y[8] + y[9] * y[9]);
    const T a = y[6] * inv, b = y[7] * inv, c = y[8] * inv, d = y[9] * inv;
    // quat_rot_mat of the UNIT quaternion (quaternion_euler_utility.py:77-79): with a^2 + b^2 + c^2 + d^2 = 1 the diagonal
    // a^2 + b^2 - c^2 - d^2 is 1 - 2 (c^2 + d^2), and with the doubled components every entry is one or two multiply-adds:
    // 17 instructions where squares, products, sums and doublings take 31 (the integrator evaluates this seven times per
    // attempted sub-step; the kernels are bound by the number of vector instructions they issue).  An exact hover
    // (q = 1, 0, 0, 0) still gives R = I exactly.
    const T b2 = b + b, c2 = c + c, d2 = d + d;
    const T ub = T(1) - b2 * b, ud = T(1) - d2 * d;
    const T r00 = ud - c2 * c, r11 = ub - d2 * d, r22 = ub - c2 * c;
    const T ad = d2 * a, ac = c2 * a, ab = b2 * a;
    const T r01 = b2 * c - ad, r10 = b2 * c + ad;
    const T r02 = b2 * d + ac, r20 = b2 * d - ac;
    const T r12 = c2 * d - ab, r21 = c2 * d + ab;
    // body velocity and aerodynamic drag (robust: air-relative velocity = v + wind, :317-320)
    T vax = vx, vay = vy, vaz = vz;
    if constexpr (P::robust) {
        T wd[3];
        next_wind(pl, wd);
        vax = vx + wd[0]; vay = vy + wd[1]; vaz = vz + wd[2];
    }
    const T vbx = r00 * vax + r10 * vay + r20 * vaz;
    const T vby = r01 * vax + r11 * vay + r21 * vaz;
    const T vbz = r02 * vax + r12 * vay + r22 * vaz;
    const T fbx = -T(phys::KDX) * (m::abs(vbx) * vbx);
    const T fby = -T(phys::KDY) * (m::abs(vby) * vby);
    const T fbz = u.F - T(phys::KDZ) * (m::abs(vbz) * vbz);
    // inertial acceleration
    T inv_m = T(1.0 / phys::M), mg = T(phys::M * phys::G);
    if constexpr (P::robust) { inv_m = pl.inv_m; mg = pl.mg; }
    const T ax = (r00 * fbx + r01 * fby + r02 * fbz) * inv_m;
    const T ay = (r10 * fbx + r11 * fby + r12 * fbz) * inv_m;
    // (f_z - M G)/M instead of f_z/M - G: the same number, but the weight cancels before the
    // scaling, so an exact hover (f_z == fl(M G)) has az == 0 exactly in fp32 as well
    const T az = ((r20 * fbx + r21 * fby + r22 * fbz) - mg) * inv_m;
    dy[0] = vx; dy[1] = ax; dy[2] = vy; dy[3] = ay; dy[4] = vz; dy[5] = az;
#ifdef QENV_HOVER_REL
    if constexpr (sizeof(T) == 4 && !P::robust) if (u.rel) {
        // thrust relative to hover: a_z = G (dlt - 2 (b^2 + c^2) (1 + dlt)) + (R drag)_z / M -- the weight never meets the thrust
        const T fdz = -T(phys::KDZ) * (m::abs(vbz) * vbz);
        const T g1 = T(phys::G) * (T(1) + u.dlt), s2 = b2 * b + c2 * c;
        dy[1] = (r00 * fbx + r01 * fby + r02 * fdz) * inv_m + r02 * g1;
        dy[3] = (r10 * fbx + r11 * fby + r12 * fdz) * inv_m + r12 * g1;
        dy[5] = (r20 * fbx + r21 * fby + r22 * fdz) * inv_m + (T(phys::G) * u.dlt - s2 * g1);
    }
#endif
    // deriv_quat with the normalised quaternion (quaternion_euler_utility.py:58-69), the 1/2 taken into the body rates
    const T hx = T(0.5) * wx, hy = T(0.5) * wy, hz = T(0.5) * wz;
    dy[6] = -(hx * b) - hy * c - hz * d;
    dy[7] = hx * a + hz * c - hy * d;
    dy[8] = hy * a - hz * b + hx * d;
    dy[9] = hz * a + hy * b - hx * c;
    // moments: action + gyro (sic: x uses -wx, y uses +wy, :347-349) + drag - w x (J w)
    const T gx = m::abs(wx) * wx, gy = m::abs(wy) * wy, gz = m::abs(wz) * wz;
    if constexpr (P::robust) {
        const T mx = u.Mx - wx * u.omr - pl.cm_xy * gx - wy * wz * T(phys::JZ - phys::JY);
        const T my = u.My + wy * u.omr - pl.cm_xy * gy - wz * wx * T(phys::JX - phys::JZ);
        const T mz = u.Mz - pl.cm_z * gz - wx * wy * T(phys::JY - phys::JX);
        dy[10] = mx * pl.ij[0]; dy[11] = my * pl.ij[1]; dy[12] = mz * pl.ij[2];
    } else {
        // nominal plant: every term already divided by the body inertia (act_finish, Cfg::cmj_*): five multiply-adds per axis
        dy[10] = u.Mxj - wx * u.omrj - pl.cmj_xy * gx - (wy * wz) * T((phys::JZ - phys::JY) / phys::JX);
        dy[11] = u.Myj + wy * u.omrj - pl.cmj_xy * gy - (wz * wx) * T((phys::JX - phys::JZ) / phys::JY);
        dy[12] = u.Mzj - pl.cmj_z * gz - (wx * wy) * T((phys::JY - phys::JX) / phys::JZ);
    }
}

// ---- Dormand-Prince 5(4) tableau (scipy rk.py RK45) -----------------------------------
namespace dp {
constexpr double A21 = 1.0 / 5;
constexpr double A31 = 3.0 / 40, A32 = 9.0 / 40;
constexpr double A41 = 44.0 / 45, A42 = -56.0 / 15, A43 = 32.0 / 9;
constexpr double A51 = 19372.0 / 6561, A52 = -25360.0 / 2187, A53 = 64448.0 / 6561, A54 = -212.0 / 729;
constexpr double A61 = 9017.0 / 3168, A62 = -355.0 / 33, A63 = 46732.0 / 5247, A64 = 49.0 / 176,
                 A65 = -5103.0 / 18656;
constexpr double B1 = 35.0 / 384, B3 = 500.0 / 1113, B4 = 125.0 / 192, B5 = -2187.0 / 6784, B6 = 11.0 / 84;
constexpr double E1 = -71.0 / 57600, E3 = 71.0 / 16695, E4 = -71.0 / 1920, E5 = 17253.0 / 339200,
                 E6 = -22.0 / 525, E7 = 1.0 / 40;
}  // namespace dp

// One DP5 step of size h from (y, k1 = f(y)) -> y_new, k7 = f(y_new), and the scaled
// error norm of scipy's _estimate_error_norm (rk.py:108-109,143-144) if WITH_ERR.
// Lean arithmetic sizes a FURTHER sub-step of an env step (after a rejected attempt, or an accepted one short of t_bound) with
// control math good to 2^-23: a 1e-7 different partition of the interval moves the result by up to ~1e-9 relative per env
// step, where a one-sub-step env step stays within 4e-13 of the reference.  Round 5's fuzz campaign found the regime that shows
// it: the unclipped position-PD loop once it tumbles (rates of hundreds of rad/s, three to five sub-steps per env step):
// 1e-6 from the reference after 80 steps where the IEEE handle is at 2e-11.  -DQENV_LEAN_EXACT_CTL=1 forms exactly those
// attempts' norm with full-precision quotients and their factor with a full-precision root and x^-0.2 (1.2e-8 on the same
// case; with -DQENV_LEAN_EXACT_H0=1, select_initial_step again in full precision when its result is below t_bound, 2e-11) --
// measured on one box, product against variant: config 3 8.16 -> 7.62 G env-steps/s (-6.6 %: a wave runs the exact block
// whenever ANY lane continues), config 2 -1 %, configs 4 / 5 unchanged; with H0 (and ocml's pow) 6.4 G, config 2 -14 %.  Not
// the default: the deviation is four orders of magnitude inside the 1e-5 tolerance and the IEEE handle is exact
// (profiles/r05_experiments_not_kept.txt).
#ifndef QENV_LEAN_EXACT_CTL
#define QENV_LEAN_EXACT_CTL 0
#endif
// exact_ctl (lean arithmetic only): another sub-step will follow this one inside the env step, so the error norm goes into a
// step size that enters the arithmetic -- form it with full-precision quotients (see rk45_attempt)
template <typename T, bool WITH_ERR, typename P>
__device__ __forceinline__ T dp5_step(const T y[13], const T k1[13], const Act<T>& u, P& pl, T h,
                                      T yn[13], T k7[13], T inc[13], bool exact_ctl = false) {
    using m = typename P::math;
    T k2[13], k3[13], k4[13], k5[13], k6[13], ys[13];
    // the step size goes into the tableau's coefficients once (15 products per attempt) instead of into every component's
    // stage sum (5 x 13): a stage value is then a chain of multiply-adds on y
    const T a21 = h * T(dp::A21);
#pragma unroll
    for (int i = 0; i < 13; ++i) ys[i] = y[i] + a21 * k1[i];
    rhs(ys, u, pl, k2);
    const T a31 = h * T(dp::A31), a32 = h * T(dp::A32);
#pragma unroll
    for (int i = 0; i < 13; ++i) ys[i] = (y[i] + a31 * k1[i]) + a32 * k2[i];
    rhs(ys, u, pl, k3);
    const T a41 = h * T(dp::A41), a42 = h * T(dp::A42), a43 = h * T(dp::A43);
#pragma unroll
    for (int i = 0; i < 13; ++i) ys[i] = ((y[i] + a41 * k1[i]) + a42 * k2[i]) + a43 * k3[i];
    rhs(ys, u, pl, k4);
    const T a51 = h * T(dp::A51), a52 = h * T(dp::A52), a53 = h * T(dp::A53), a54 = h * T(dp::A54);
#pragma unroll
    for (int i = 0; i < 13; ++i) ys[i] = (((y[i] + a51 * k1[i]) + a52 * k2[i]) + a53 * k3[i]) + a54 * k4[i];
    rhs(ys, u, pl, k5);
    const T a61 = h * T(dp::A61), a62 = h * T(dp::A62), a63 = h * T(dp::A63), a64 = h * T(dp::A64), a65 = h * T(dp::A65);
#pragma unroll
    for (int i = 0; i < 13; ++i)
        ys[i] = ((((y[i] + a61 * k1[i]) + a62 * k2[i]) + a63 * k3[i]) + a64 * k4[i]) + a65 * k5[i];
    rhs(ys, u, pl, k6);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const T sum = T(dp::B1) * k1[i] + T(dp::B3) * k3[i] + T(dp::B4) * k4[i] + T(dp::B5) * k5[i] + T(dp::B6) * k6[i];
        inc[i] = h * sum;
        yn[i] = y[i] + h * sum;  // one fma: a single rounding of the new state
    }
    rhs(yn, u, pl, k7);
    T err2 = T(0);
    if (WITH_ERR) {
        // 13 error_norm^2 = h^2 sum_i ((E . K)_i / scale_i)^2: the step size leaves the sum (one product instead of 13)
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const T e = T(dp::E1) * k1[i] + T(dp::E3) * k3[i] + T(dp::E4) * k4[i] + T(dp::E5) * k5[i] +
                        T(dp::E6) * k6[i] + T(dp::E7) * k7[i];
            const T sc = T(1e-6) + m::max_abs(y[i], yn[i]) * T(1e-3);
            // (fp32 arithmetic keeps h inside every term: with scales down to 1e-6 the un-scaled sum of a diverging state
            // overflows, and inf * h^2 stays inf however far the controller shrinks h -- the attempt could never recover)
            const T r = (sizeof(T) == 4 ? h * e : e) * m::rcp_ctl(sc);
            err2 += r * r;
        }
        if constexpr (sizeof(T) != 4) err2 = err2 * (h * h);
        if constexpr (m::lean) {
            // the hardware reciprocal is good to 2^-23: whenever the fast norm cannot decide err < 1 with a wide margin,
            // form it again with full-precision quotients (rare: the branch is taken by ~1e-5 of the attempts) -- and
            // whenever the norm will size a further sub-step (a rejected attempt, or an accepted one short of t_bound)
            if (m::abs(err2 - T(13)) < T(13e-5) || (QENV_LEAN_EXACT_CTL && (exact_ctl || !(err2 < T(13))))) {
                err2 = T(0);
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    const T e = h * (T(dp::E1) * k1[i] + T(dp::E3) * k3[i] + T(dp::E4) * k4[i] + T(dp::E5) * k5[i] +
                                     T(dp::E6) * k6[i] + T(dp::E7) * k7[i]);
                    const T sc = T(1e-6) + m::max_abs(y[i], yn[i]) * T(1e-3);
                    const T r = m::div(e, sc);
                    err2 += r * r;
                }
            }
        }
    }
    return err2;  // = 13 * error_norm^2
}

// solve_ivp(drone_eq, (0, tb), y, args=(u,)).y[:, -1] with scipy's RK45 defaults, restarted
// every env step as the reference does (quadrotor_env.py:483).  select_initial_step:
// common.py:66-133; accept/reject loop: rk.py:111-167 flattened to one attempted sub-step
// per iteration.  On exit y = state at tb and kq = f(t_end, y_end)
// (FSAL) whose rows 6..9 are the reference's V_q.  Every lane leaves the loop after at
// most MAX_ATTEMPTS attempts.
// scipy has no cap on the attempts of a step; the nominal plant needs 1-4 per env step.  A robust redraw step (a new
// gust at every RHS evaluation) legitimately takes hundreds of rejected sub-steps, so its cap is far higher; either way
// every lane leaves the loop after a bounded number of attempts.
template <typename P> struct Rk45Cap { static constexpr int value = P::robust ? 16384 : 96; };

// select_initial_step (common.py:66-133) from y and k1 = f(y): the first h_abs of the step, NOT yet clamped to t_bound.
// EXACT: full-precision reciprocals / roots / pow instead of the control forms (lean arithmetic: see integrate_rk45).
template <typename M, bool EXACT> struct CtlMath {
    template <typename T> static __device__ __forceinline__ T rcp(T x) { if constexpr (EXACT) return M::rcp(x); else return M::rcp_ctl(x); }
    template <typename T> static __device__ __forceinline__ T sqrt(T x) { if constexpr (EXACT) return M::sqrt(x); else return M::sqrt_ctl(x); }
    template <typename T> static __device__ __forceinline__ T div(T a, T b) { if constexpr (EXACT) return M::div(a, b); else return M::div_ctl(a, b); }
    template <typename T> static __device__ __forceinline__ T pow(T a, T b) { if constexpr (EXACT) return M::pow(a, b); else return M::pow_ctl(a, b); }
};
template <typename T, typename P, bool EXACT = false>
__device__ __forceinline__ T rk45_initial_step(const T y[13], const T k1[13], const Act<T>& u, P& pl, T tb) {
    using m = typename P::math;
    using c = CtlMath<m, EXACT>;
    T d0 = T(0), d1 = T(0);
    T isc[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        isc[i] = c::rcp(T(1e-6) + m::abs(y[i]) * T(1e-3));
        const T a = y[i] * isc[i], b = k1[i] * isc[i];
        d0 += a * a;
        d1 += b * b;
    }
    d0 = c::sqrt(d0 * T(1.0 / 13)); d1 = c::sqrt(d1 * T(1.0 / 13));
    T h0 = (d0 < T(1e-5) || d1 < T(1e-5)) ? T(1e-6) : c::div(T(0.01) * d0, d1);
    h0 = m::min(h0, tb);
    // h1 = (0.01 / max(d1, d2))^(1/5) only matters below t_bound: when max(d1, d2) <= 0.01 / t_bound^5 is PROVEN from y, f0 and
    // h0 alone (qenv_probe_bound.hpp: a Lipschitz bound of the nominal right-hand side along y .. y + h0 f0), the second
    // evaluation f(y + h0 f0) -- made for d2 only -- is skipped and h_abs = min(100 h0, t_bound), the value the full
    // computation would return.  Not proven (or a robust plant, whose gust process counts its evaluations): as before.
    bool proven = false;
#ifndef QENV_PROBE_SKIP
#define QENV_PROBE_SKIP 1   // 0: always evaluate f(y + h0 f0) (A/B builds: tools/build_variant.sh noskip "-DQENV_PROBE_SKIP=0" ...)
#endif
    if constexpr (!P::robust && QENV_PROBE_SKIP) if (pl.skip_probe) {
        ProbePlant<T> pp;
        pp.F_abs = m::abs(u.F); pp.omrj_abs = m::abs(u.omrj); pp.cmj_xy = pl.cmj_xy; pp.cmj_z = pl.cmj_z;
        constexpr double kd_xy = phys::KDY > phys::KDX ? phys::KDY : phys::KDX;   // the largest of the three drag coefficients
        pp.kd = T(phys::KDZ > kd_xy ? phys::KDZ : kd_xy); pp.inv_m = T(1.0 / phys::M);
        pp.cx = T((phys::JZ - phys::JY) / phys::JX < 0 ? -(phys::JZ - phys::JY) / phys::JX : (phys::JZ - phys::JY) / phys::JX);
        pp.cy = T((phys::JX - phys::JZ) / phys::JY < 0 ? -(phys::JX - phys::JZ) / phys::JY : (phys::JX - phys::JZ) / phys::JY);
        pp.cz = T((phys::JY - phys::JX) / phys::JZ < 0 ? -(phys::JY - phys::JX) / phys::JZ : (phys::JY - phys::JX) / phys::JZ);
        const T tb2 = tb * tb;
        proven = probe_skippable(y, k1, isc, pp, tb, d1, T(0.01) / (tb2 * tb2 * tb));   // (h0 <= tb: clamped above)
#if QENV_PROBE_SKIP == 2   /* timing experiment: never evaluate the probe (WRONG results when the proof fails) */
        proven = true;
#elif QENV_PROBE_SKIP == 3 /* timing experiment: the proof computed, the probe evaluated anyway */
        { int keep = proven; asm volatile("" : : "v"(keep)); proven = false; }
#endif
    }
    T h1 = tb;   // (proven: any value >= t_bound leaves the minimum below to 100 h0 and t_bound)
    if (!proven) {
        T y1[13], f1[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) y1[i] = y[i] + h0 * k1[i];
        rhs(y1, u, pl, f1);
        T d2 = T(0);
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const T a = (f1[i] - k1[i]) * isc[i];
            d2 += a * a;
        }
        d2 = c::div(c::sqrt(d2 * T(1.0 / 13)), h0);
        if (d1 <= T(1e-15) && d2 <= T(1e-15)) h1 = m::max(T(1e-6), h0 * T(1e-3));
        else h1 = c::pow(c::div(T(0.01), m::max(d1, d2)), T(0.2));
    }
    return m::min(T(100) * h0, h1);
}

// One attempted sub-step of the accept/reject loop (rk.py:111-167) at time t with k1 = f(t, y).  Returns true when the
// env step is over: t has reached tb, or the solver gave up (TOO_SMALL_STEP: the last accepted y stays).
template <typename T, typename P>
__device__ __forceinline__ bool rk45_attempt(T y[13], T k1[13], T dy[13], const Act<T>& u, P& pl, T tb, T& t, T& h_abs,
                                             bool& rejected) {
    using m = typename P::math;
    const T min_step = T(10) * m::ulp_up(t);
    if (!rejected && h_abs < min_step) h_abs = min_step;
    if (h_abs < min_step) return true;
    T t_new = t + h_abs;
    if (t_new > tb) t_new = tb;
    const T h = t_new - t;
    h_abs = h;
    T yn[13], k7[13], inc[13];
    if constexpr (m::lean && QENV_LEAN_EXACT_CTL) {
        // Lean arithmetic: the control quantities (error norm, growth factor) come from hardware seeds and fp32 log / exp, good
        // to 2^-23 -- harmless while a step size is only compared or clamped to the rest of the interval, which is every env
        // step that one sub-step covers.  When ANOTHER sub-step follows (a rejected attempt, or an accepted one short of
        // t_bound) the factor sizes it, and a 1e-7 different partition of the interval moves the result by ~1e-10 relative:
        // the fuzz campaign's tumbling position-PD loops (unclipped, rates of hundreds of rad/s, three to five sub-steps per
        // env step) drifted 1e-6 from the reference where the IEEE handle stays at 1e-11.  So exactly those attempts form the
        // norm with full-precision quotients and the factor with a full-precision root and x^-0.2; the others skip the factor
        // altogether (its step size is never used: the next env step starts from select_initial_step).
        const bool more = t_new < tb;
        const T err2 = dp5_step<T, true>(y, k1, u, pl, h, yn, k7, inc, more);
        if (err2 < T(13)) {
            if (more) {
                const T err = m::sqrt(err2 * T(1.0 / 13));
                T factor = (err == T(0)) ? T(10) : m::min(T(10), T(0.9) * m::pow_m02(err));
                if (rejected) factor = m::min(T(1), factor);
                h_abs *= factor;
            }
            t = t_new;
            rejected = false;
#pragma unroll
            for (int i = 0; i < 13; ++i) { y[i] = yn[i]; k1[i] = k7[i]; dy[i] += inc[i]; }
        } else {
            const T err = m::sqrt(err2 * T(1.0 / 13));
            h_abs *= m::max(T(0.2), T(0.9) * m::pow_m02(err));
            rejected = true;
        }
        return !(t < tb);
    }
    const T err2 = dp5_step<T, true>(y, k1, u, pl, h, yn, k7, inc);
    const T err = m::sqrt_ctl(err2 * T(1.0 / 13));
    // lean: decided on the (guard-banded) sum itself, sqrt_ctl is a hardware approximation there
    if (m::lean ? (err2 < T(13)) : (err < T(1))) {
        T factor = (err == T(0)) ? T(10) : m::min(T(10), T(0.9) * m::pow_ctl(err, T(-0.2)));
        if (rejected) factor = m::min(T(1), factor);
        h_abs *= factor;
        t = t_new;
        rejected = false;
#pragma unroll
        for (int i = 0; i < 13; ++i) { y[i] = yn[i]; k1[i] = k7[i]; dy[i] += inc[i]; }
    } else {
        h_abs *= m::max(T(0.2), T(0.9) * m::pow_ctl(err, T(-0.2)));
        rejected = true;
    }
    return !(t < tb);
}

template <typename T, typename P>
__device__ __forceinline__ void integrate_rk45(T y[13], const Act<T>& u, P& pl, T tb, T kq[13], T dy[13],
                                               bool have_f0 = false) {
    T k1[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) dy[i] = T(0);
    // have_f0 (wave-uniform): kq already holds f(y) from the previous env step of the same rollout with the same
    // actuation (FSAL across env steps): the same numbers rhs() would return, so results do not change
    if (have_f0) {
#pragma unroll
        for (int i = 0; i < 13; ++i) k1[i] = kq[i];
    } else {
        rhs(y, u, pl, k1);
    }
    // An env AT an equilibrium of the held actuation -- f(y) == 0 in every component, BASELINE configs[1]'s literal "zero-action
    // hover" -- does not move: every stage value y + h a 0 is y, every K_i is f(y) = 0, y_new == y and the error norm is 0, so
    // scipy accepts sub-step after sub-step (h0 = 1e-6 growing tenfold: five of them, 32 right-hand sides) and returns y.  The
    // same result without the attempts, bit for bit (the exact-hover rows of tests/test_gpu_parity.py); the nominal plant only (a
    // robust plant's gust process counts its evaluations).  Two compares in the common case, the other eleven only behind them.
    bool at_rest = false;
    if constexpr (!P::robust) {
        if (k1[5] == T(0) && k1[1] == T(0)) {
            at_rest = true;
#pragma unroll
            for (int i = 0; i < 13; ++i) at_rest = at_rest && (k1[i] == T(0));
        }
    }
    T h_abs = rk45_initial_step(y, k1, u, pl, tb);   // (also for an env at rest: a branch around it costs the others more)
#ifndef QENV_LEAN_EXACT_H0
#define QENV_LEAN_EXACT_H0 0
#endif
    if constexpr (P::math::lean && QENV_LEAN_EXACT_CTL && QENV_LEAN_EXACT_H0) {
        // lean arithmetic: a first step size below t_bound enters the arithmetic (rk45_attempt) -- rare (a violent state): again
        // with full-precision control math; the 1e-6 covers the 2^-23 of the fast form
        if (h_abs < tb * T(1.000001)) h_abs = rk45_initial_step<T, P, true>(y, k1, u, pl, tb);
    }
    h_abs = P::math::min(h_abs, tb);
    T t = at_rest ? tb : T(0);
    bool rejected = false;
    for (int it = 0; it < Rk45Cap<P>::value && t < tb; ++it)
        if (rk45_attempt(y, k1, dy, u, pl, tb, t, h_abs, rejected)) break;
#pragma unroll
    for (int i = 0; i < 13; ++i) kq[i] = k1[i];
}

// The same solve_ivp restatement for the two envs of a lane: the heavy part (RHS evaluations, stage
// combinations, error norm) is packed; the step-size control is scalar per env and drives per-env
// selects.  Each env follows exactly the instruction sequence of the scalar version (an env that has
// reached t_bound idles with h = 0 while its partner finishes), so results are bit-identical to it.
__device__ __forceinline__ void integrate_rk45_x2(F2 y[13], const Act<F2>& u, Plant<F2, false>& pl, float tb, F2 kq[13],
                                                  F2 dy[13], bool have_f0 = false) {
    using m = Mth<float>;
    using m2 = Mth<F2>;
    constexpr int MAX_ATTEMPTS = 96;
    F2 k1[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) dy[i] = F2{0.f, 0.f};
    if (have_f0) {  // see integrate_rk45
#pragma unroll
        for (int i = 0; i < 13; ++i) k1[i] = kq[i];
    } else {
        rhs(y, u, pl, k1);
    }
    float h_abs[2];
    {
        F2 d0 = {0.f, 0.f}, d1 = {0.f, 0.f};
        F2 isc[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            isc[i] = m2::rcp(F2(1e-6f) + m2::abs(y[i]) * F2(1e-3f));
            const F2 a = y[i] * isc[i], b = k1[i] * isc[i];
            d0 += a * a;
            d1 += b * b;
        }
        d0 = m2::sqrt(d0 * F2((float)(1.0 / 13))); d1 = m2::sqrt(d1 * F2((float)(1.0 / 13)));
        float h0[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            h0[c] = (d0[c] < 1e-5f || d1[c] < 1e-5f) ? 1e-6f : m::div_ctl(0.01f * d0[c], d1[c]);
            h0[c] = m::min(h0[c], tb);
        }
        const F2 h0v = {h0[0], h0[1]};
        F2 y1[13], f1[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) y1[i] = y[i] + h0v * k1[i];
        rhs(y1, u, pl, f1);
        F2 d2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const F2 a = (f1[i] - k1[i]) * isc[i];
            d2 += a * a;
        }
        d2 = m2::sqrt(d2 * F2((float)(1.0 / 13)));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float d2c = m::div_ctl(d2[c], h0[c]);
            float h1;
            if (d1[c] <= 1e-15f && d2c <= 1e-15f) h1 = m::max(1e-6f, h0[c] * 1e-3f);
            else h1 = m::pow_ctl(m::div_ctl(0.01f, m::max(d1[c], d2c)), 0.2f);
            h_abs[c] = m::min(m::min(100.f * h0[c], h1), tb);
        }
    }
    float t[2] = {0.f, 0.f};
    bool rejected[2] = {false, false}, failed[2] = {false, false};
    for (int it = 0; it < MAX_ATTEMPTS && ((t[0] < tb && !failed[0]) || (t[1] < tb && !failed[1])); ++it) {
        float h[2], t_new[2];
        bool act[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            act[c] = t[c] < tb && !failed[c];
            const float min_step = 10.f * m::ulp_up(t[c]);
            if (act[c] && !rejected[c] && h_abs[c] < min_step) h_abs[c] = min_step;
            if (act[c] && h_abs[c] < min_step) { failed[c] = true; act[c] = false; }  // TOO_SMALL_STEP
            t_new[c] = t[c] + h_abs[c];
            if (t_new[c] > tb) t_new[c] = tb;
            h[c] = act[c] ? t_new[c] - t[c] : 0.f;
            if (act[c]) h_abs[c] = h[c];
        }
        if (!act[0] && !act[1]) break;
        F2 yn[13], k7[13], inc[13];
        const F2 err2 = dp5_step<F2, true>(y, k1, u, pl, F2{h[0], h[1]}, yn, k7, inc);
        bool accept[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float err = m::sqrt(err2[c] * (float)(1.0 / 13));
            accept[c] = act[c] && err < 1.f;
            if (accept[c]) {
                float factor = (err == 0.f) ? 10.f : m::min(10.f, 0.9f * m::pow_ctl(err, -0.2f));
                if (rejected[c]) factor = m::min(1.f, factor);
                h_abs[c] *= factor;
                t[c] = t_new[c];
                rejected[c] = false;
            } else if (act[c]) {
                h_abs[c] *= m::max(0.2f, 0.9f * m::pow_ctl(err, -0.2f));
                rejected[c] = true;
            }
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            y[i] = F2{accept[0] ? yn[i].x : y[i].x, accept[1] ? yn[i].y : y[i].y};
            k1[i] = F2{accept[0] ? k7[i].x : k1[i].x, accept[1] ? k7[i].y : k1[i].y};
            dy[i] = F2{accept[0] ? dy[i].x + inc[i].x : dy[i].x, accept[1] ? dy[i].y + inc[i].y : dy[i].y};
        }
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) kq[i] = k1[i];
}

template <typename T, typename P>
__device__ __forceinline__ void integrate_dp5(T y[13], const Act<T>& u, P& pl, T tb, T kq[13], T dy[13],
                                              bool have_f0 = false) {
    T k1[13], yn[13];
    if (have_f0) {
#pragma unroll
        for (int i = 0; i < 13; ++i) k1[i] = kq[i];
    } else {
        rhs(y, u, pl, k1);
    }
    dp5_step<T, false>(y, k1, u, pl, tb, yn, kq, dy);
#pragma unroll
    for (int i = 0; i < 13; ++i) y[i] = yn[i];
}

template <typename T, typename P>
__device__ __forceinline__ void integrate_rk4(T y[13], const Act<T>& u, P& pl, T h, T kq[13], T dy[13],
                                              bool have_f0 = false) {
    T k1[13], k2[13], k3[13], k4[13], ys[13];
    if (have_f0) {
#pragma unroll
        for (int i = 0; i < 13; ++i) k1[i] = kq[i];
    } else {
        rhs(y, u, pl, k1);
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) ys[i] = y[i] + T(0.5) * h * k1[i];
    rhs(ys, u, pl, k2);
#pragma unroll
    for (int i = 0; i < 13; ++i) ys[i] = y[i] + T(0.5) * h * k2[i];
    rhs(ys, u, pl, k3);
#pragma unroll
    for (int i = 0; i < 13; ++i) ys[i] = y[i] + h * k3[i];
    rhs(ys, u, pl, k4);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const T sum = T(1.0 / 6) * ((k1[i] + k4[i]) + T(2) * (k2[i] + k3[i]));
        dy[i] = h * sum;
        y[i] = y[i] + h * sum;
    }
    rhs(y, u, pl, kq);
}

enum { INT_RK45 = 0, INT_DP5 = 1, INT_RK4 = 2 };

template <typename T, int INTEG, typename P>
// On exit: y = state at tb, kq = f(t_end, y_end), dy = y_end - y_start as the SUM OF THE
// INTEGRATOR'S INCREMENTS (not the difference of the rounded states): the cancellation-free input
// of the increment-based Euler rates below.
__device__ __forceinline__ void integrate(T y[13], const Act<T>& u, P& pl, T tb, T kq[13], T dy[13],
                                          bool have_f0 = false) {
    if (INTEG == INT_RK45) integrate_rk45(y, u, pl, tb, kq, dy, have_f0);
    else if (INTEG == INT_DP5) integrate_dp5(y, u, pl, tb, kq, dy, have_f0);
    else integrate_rk4(y, u, pl, tb, kq, dy, have_f0);
}

template <int INTEG>
__device__ __forceinline__ void integrate2(F2 y[13], const Act<F2>& u, Plant<F2, false>& pl, float tb, F2 kq[13], F2 dy[13],
                                           bool have_f0 = false) {
    if constexpr (INTEG == INT_RK45) integrate_rk45_x2(y, u, pl, tb, kq, dy, have_f0);
    else if constexpr (INTEG == INT_DP5) integrate_dp5<F2>(y, u, pl, F2{tb, tb}, kq, dy, have_f0);
    else integrate_rk4<F2>(y, u, pl, F2{tb, tb}, kq, dy, have_f0);
}

// ---- per-env bookkeeping carried between steps (registers in fused rollouts) -------------
template <typename T> struct Book {
    T prev_ang[3];
    T ang[3];
    T ang_vel[3];
    T prev_shaping;
    T abs_sum;
    T ep_return;
    T reward;
    int32_t i;
    int32_t flags;  // QENV_FLAG_*
};

constexpr int FLAG_DONE = 1, FLAG_SOLVED = 2, FLAG_HAS_SHAPING = 4, FLAG_ANG_OK = 8;

// quat_euler on the re-normalised end quaternion + finite-difference Euler rates,
// quadrotor_env.py:488-493 (main) / quadrotor_env_opt.py:397-402 (opt: prev_ang = ang
// BEFORE the update).
//
// fp32 only: the reference's rate (ang - prev_ang)/t_step differences two angles that each carry
// ~1e-7 of rounding and divides by 0.01, i.e. it injects ~1e-5 rad/s of noise per step into
// the LQR/PID feedback.  When prev_ang is known to be the angle of the quaternion this step
// started from (inc_ok), the same quantity is formed without cancellation from the integrator's
// quaternion increment dq:  phi = atan2(n, d)  =>  dphi = atan2(dn*d0 - dd*n0, d1*d0 + n1*n0),
// with dn, dd expanded in products of dq (relative accuracy 1e-7 OF THE INCREMENT), plus the
// 2*pi wrap the plain difference would contain.  In fp64 the plain difference is kept.
template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void euler_update(const T q_old[4], const T y[13], const T dy[13], bool inc_ok,
                                             const Cfg<T>& cfg, Book<T>& bk) {
    using m = M;
    const T inv = m::rsqrt(y[6] * y[6] + y[7] * y[7] + y[8] * y[8] + y[9] * y[9]);
    const T q0 = y[6] * inv, q1 = y[7] * inv, q2 = y[8] * inv, q3 = y[9] * inv;
    T ang[3];
    ang[0] = m::atan2(T(2) * (q0 * q1 + q2 * q3), T(1) - T(2) * (q1 * q1 + q2 * q2));
    ang[1] = m::asin(T(2) * (q0 * q2 - q3 * q1));
    ang[2] = m::atan2(T(2) * (q0 * q3 + q1 * q2), T(1) - T(2) * (q2 * q2 + q3 * q3));
    const T inv_dt = m::div(T(1), cfg.t_step);
    T dang[3] = {T(0), T(0), T(0)};
    const bool use_inc = (sizeof(T) == 4) && inc_ok;
    if (use_inc) {
        const T a0 = q_old[0], b0 = q_old[1], c0 = q_old[2], d0 = q_old[3];
        const T a1 = y[6], b1 = y[7], c1 = y[8], d1 = y[9];
        const T da = dy[6], db = dy[7], dc = dy[8], dd = dy[9];
        const T sa = da * (a1 + a0), sb = db * (b1 + b0), sc = dc * (c1 + c0), sd = dd * (d1 + d0);
        // roll: n = 2(ab+cd), d = a^2-b^2-c^2+d^2 (scale-invariant, un-normalised q is fine)
        {
            const T n0 = T(2) * (a0 * b0 + c0 * d0), e0 = (a0 * a0 - b0 * b0) - (c0 * c0 - d0 * d0);
            const T n1 = T(2) * (a1 * b1 + c1 * d1), e1 = (a1 * a1 - b1 * b1) - (c1 * c1 - d1 * d1);
            const T dn = T(2) * ((da * b0 + a1 * db) + (dc * d0 + c1 * dd));
            const T de = (sa - sb) - (sc - sd);
            dang[0] = m::atan2(dn * e0 - de * n0, e1 * e0 + n1 * n0);
        }
        // yaw: n = 2(ad+bc), d = a^2+b^2-c^2-d^2
        {
            const T n0 = T(2) * (a0 * d0 + b0 * c0), e0 = (a0 * a0 + b0 * b0) - (c0 * c0 + d0 * d0);
            const T n1 = T(2) * (a1 * d1 + b1 * c1), e1 = (a1 * a1 + b1 * b1) - (c1 * c1 + d1 * d1);
            const T dn = T(2) * ((da * d0 + a1 * dd) + (db * c0 + b1 * dc));
            const T de = (sa + sb) - (sc + sd);
            dang[2] = m::atan2(dn * e0 - de * n0, e1 * e0 + n1 * n0);
        }
        // pitch: S = sin(theta) = 2(ac-db)/|q|^2
        {
            const T s0 = (a0 * a0 + b0 * b0) + (c0 * c0 + d0 * d0), s1 = (a1 * a1 + b1 * b1) + (c1 * c1 + d1 * d1);
            const T m0 = T(2) * (a0 * c0 - d0 * b0);
            const T dm = T(2) * ((da * c0 + a1 * dc) - (dd * b0 + d1 * db));
            const T ds = (sa + sb) + (sc + sd);
            const T S0 = m0 / s0, S1 = T(2) * (a1 * c1 - d1 * b1) / s1;
            const T dS = (dm * s0 - m0 * ds) / (s0 * s1);
            const T C0 = m::sqrt(m::max(T(0), T(1) - S0 * S0)), C1 = m::sqrt(m::max(T(0), T(1) - S1 * S1));
            const T dC = -dS * (S1 + S0) / m::max(C1 + C0, T(1e-30));
            dang[1] = m::atan2(dS * C0 - S0 * dC, C1 * C0 + S1 * S0);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T prev = cfg.variant ? bk.ang[k] : bk.prev_ang[k];
        const T direct = ang[k] - prev;
        T rate = direct * inv_dt;
        if (use_inc) {
            // the plain difference equals dang + 2 pi j (j != 0 only when the angle wrapped)
            const T j = rintf((float)((direct - dang[k]) * T(0.15915494309189535)));
            rate = (dang[k] + T(6.283185307179586) * j) * inv_dt;
        }
        bk.ang_vel[k] = rate;
        bk.prev_ang[k] = cfg.variant ? prev : ang[k];
        bk.ang[k] = ang[k];
    }
}

// done_condition + reward_function + control_effort, quadrotor_env.py:500-577 /
// quadrotor_env_opt.py:410-484.  `act` is self.action: the clipped
// [-1,1] command (direct) or the RAW [F,M] request (indirect, quirk Q5).
template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void done_reward(const T y[13], const T act[4], const T effort[4], const Cfg<T>& cfg,
                                            Book<T>& bk) {
    using m = M;
    const T v0 = y[1], v1 = y[3], v2 = y[5];
    const T* ang = bk.ang;
    bool done = (bk.flags & FLAG_DONE) != 0;
    const T c[9] = {v0, v1, v2, ang[0], ang[1], ang[2], y[10], y[11], y[12]};
#pragma unroll
    for (int k = 0; k < 9; ++k) done = done || (m::abs(c[k]) >= cfg.bb[k]);
    const bool timeout = bk.i >= cfg.n_max;
    if (cfg.variant) done = done || timeout;

    const T vv = v0 * v0 + v1 * v1 + v2 * v2;
    const T psi = ang[2];
    const T nv = m::sqrt(vv) * T(1.0 / phys::BB_VEL);
    const T npsi = m::abs(psi) * T(0.25);
    const T a01 = ang[0] * ang[0] + ang[1] * ang[1];
    const T nang = m::sqrt(a01) * T(1.0 / phys::BB_ANG);
    const T r_norm = m::sqrt(vv + psi * psi);
    T shaping, pc, tr0;
    if (cfg.variant) {
        shaping = T(-100) * (nv + npsi + T(0.3) * nang);
        pc = T(1); tr0 = T(0.001);
        const T n3 = m::sqrt(a01 + psi * psi);
        const T tr[3] = {T(0.001), T(0.01), T(0.1)}, trp[3] = {T(300), T(100), T(10)};
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool in = !hit && (r_norm < T(2) * tr[k]);
            if (in) { shaping += trp[k]; if (n3 < T(3.4641016151377544) * tr[k]) shaping += trp[k]; }
            hit = hit || in;
        }
    } else {
        shaping = T(-5.0 / 20.0) * (T(15) * nv + T(4) * npsi + nang);
        pc = T(0.003); tr0 = T(0.005);
        const T n2 = m::sqrt(a01);
        const T tr[3] = {T(0.005), T(0.01), T(0.1)}, trp[3] = {T(3), T(2), T(1)};
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool in = !hit && (r_norm < T(2) * tr[k]);
            if (in) { shaping += trp[k]; if (n2 < T(5.656854249492381) * tr[k]) shaping += trp[k]; }
            hit = hit || in;
        }
    }
    T reward = (bk.flags & FLAG_HAS_SHAPING) ? shaping - bk.prev_shaping : T(0);
    bk.prev_shaping = shaping;
    T ss = T(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const T zc = (cfg.direct || k == 0) ? cfg.zero_control0 : T(0);
        const T d = act[k] - zc;
        ss += d * d;
    }
    reward += -ss * pc;
    const T current = vv + a01 + psi * psi + (y[10] * y[10] + y[11] * y[11] + y[12] * y[12]);
    int solved = (bk.flags & FLAG_SOLVED) ? 1 : 0;
    if (current < T(9) * tr0 * tr0) {
        reward = cfg.variant ? T(500) : reward + T(20);
        solved = 1;
        if (cfg.training) done = true;
    } else if (timeout) {
        solved = 0;
        if (!cfg.variant) done = true;
    } else if (done) {
        reward = cfg.variant ? T(-200) : reward + T(-20);
        solved = 0;
    }
    bk.reward = reward;
    // the worker's reward_sum (ppo.py:236,258): zeroed AFTER env.reset(), so the rewards of the reset's own T warm-up
    // steps (i = 1..T) are not part of an episode's return
    if (bk.i > cfg.warmup) bk.ep_return += reward;
    bk.flags = (done ? FLAG_DONE : 0) | (solved ? FLAG_SOLVED : 0) | FLAG_HAS_SHAPING | FLAG_ANG_OK;
    bk.abs_sum += m::sqrt((effort[0] * effort[0] + effort[1] * effort[1]) + (effort[2] * effort[2] + effort[3] * effort[3]));
}

// Action pre-processing of step(), quadrotor_env.py:470-479: returns the held
// actuation `u`, and fills w / effort / clipped_action / self.action.
template <typename T, typename P>
__device__ __forceinline__ void prepare_action(const T raw[4], const Cfg<T>& cfg, const P& pl, T w[4], T effort[4],
                                               T cl[4], T act[4], Act<T>& u) {
    using m = typename P::math;
    if (cfg.direct) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            act[k] = m::min(m::max(raw[k], T(-1)), T(1));
            cl[k] = act[k];
            effort[k] = cfg.variant ? (act[k] + T(1)) * T(phys::T2WR * phys::M * phys::G / 8.0) : act[k];
        }
        mix_direct(act, pl, w, u);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) act[k] = raw[k];
        mix_indirect(raw, cfg.clipped != 0, cfg.variant != 0, pl, w, effort, cl, u);
    }
}

// ---- LQR law, lqr_quad.py:128-154 with the CARE gains in closed form (App. A.8) ---------
template <typename T> struct LqrGains { T kt[3]; T ka[3][2]; };  // velocity gains; attitude (e, e_dot)

template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void lqr_law(const T y[13], const Book<T>& bk, const LqrGains<T>& g, T act[4]) {
    using m = M;
    const T F0 = g.kt[0] * y[1], F1 = g.kt[1] * y[3], F2 = g.kt[2] * y[5];
    // cos(atan2(a, b)) = b / hypot(a, b) (every sign of a and b; atan2(0, 0) = 0): the two cosines of lqr_quad.py:139-141
    // without their range reductions -- a rounding-level difference from evaluating the cosine of the rounded angle
    const T Fz = F2 + T(phys::G);
    const T theta_t = m::atan2(F0, Fz);
    const T r2t = F0 * F0 + Fz * Fz;
    const T ct = r2t > T(0) ? Fz * m::rsqrt(r2t) : T(1);
    const T Fy = -F1 * ct;
    const T phi_t = m::atan2(Fy, Fz);
    const T r2p = Fy * Fy + Fz * Fz;
    const T cp = r2p > T(0) ? Fz * m::rsqrt(r2p) : T(1);
    act[0] = m::div(T(phys::M) * Fz, ct * cp);
    act[1] = g.ka[0][0] * (bk.ang[0] - phi_t) + g.ka[0][1] * bk.ang_vel[0];
    act[2] = g.ka[1][0] * (bk.ang[1] - theta_t) + g.ka[1][1] * bk.ang_vel[1];
    act[3] = g.ka[2][0] * bk.ang[2] + g.ka[2][1] * bk.ang_vel[2];
}

// ---- PID law, pid_vel_control.py:29-127 ------------------------------
template <typename T> struct PidGains { T p[6], i[6], d[6]; T ts; };
template <typename T> struct PidState { T ix[6], x_old[6], ang_d_ant[3]; };

template <typename T, typename M = Mth<T>>
__device__ __forceinline__ T pid_eval(const PidGains<T>& g, PidState<T>& s, int k, T x, T x_d, T dx_d) {
    const T dx = M::div(x - s.x_old[k], g.ts);
    s.x_old[k] = x;
    s.ix[k] = s.ix[k] + (x_d - x) * g.ts;
    return g.p[k] * (x_d - x) + g.d[k] * (dx_d - dx) - g.i[k] * s.ix[k];
}

template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void pid_law(const T y[13], const Book<T>& bk, const PidGains<T>& g, PidState<T>& s,
                                        const T tgt[4], T t_step, T act[4]) {
    using m = M;
    const T u1 = pid_eval<T, M>(g, s, 0, y[1], tgt[0], T(0));
    const T u2 = pid_eval<T, M>(g, s, 1, y[3], tgt[1], T(0));
    const T u3 = pid_eval<T, M>(g, s, 2, y[5], tgt[2], T(0));
    const T theta_d = m::atan2(u1, u3 + T(phys::G));
    const T ctd = m::cos(theta_d);
    const T phi_d = m::atan2(-u2 * ctd, u3 + T(phys::G));
    act[0] = m::div(T(phys::M) * (u3 + T(phys::G)), ctd * m::cos(phi_d));
    const T ang_d[3] = {phi_d, theta_d, tgt[3]};
    T u567[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T v_ang_d = m::div(ang_d[k] - s.ang_d_ant[k], t_step);
        u567[k] = pid_eval<T, M>(g, s, 3 + k, bk.ang[k], ang_d[k], v_ang_d);
        s.ang_d_ant[k] = ang_d[k];
    }
    const T phi = bk.ang[0], theta = bk.ang[1];
    const T sp = m::sin(phi), cp = m::cos(phi), ct = m::cos(theta), tt = m::tan(theta);
    const T b1 = T(1.0 / phys::JX), b2 = tt * sp * T(1.0 / phys::JY), b3 = tt * cp * T(1.0 / phys::JZ);
    const T b4 = cp * T(1.0 / phys::JY), b5 = -sp * T(1.0 / phys::JZ);
    const T b6 = m::div(sp, ct) * T(1.0 / phys::JY), b7 = m::div(cp, ct) * T(1.0 / phys::JZ);
    const T det = b4 * b7 - b5 * b6;
    const T U3 = m::div(b7 * u567[1] - b5 * u567[2], det);
    const T U4 = m::div(-b6 * u567[1] + b4 * u567[2], det);
    act[1] = m::div(u567[0] - b2 * U3 - b3 * U4, b1);
    act[2] = U3;
    act[3] = U4;
}

// ---- position-tracking PD, environment/controller/PID_control.py:24-106,205-216 ---------------
// tgt = {xd, yd, zd, dxd, dyd, dzd, psi_d}; the passed derivatives (state velocity, env.ang_vel) are
// used, arctan instead of arctan2.  Only ix[] and ang_d_ant[] of the PID state are touched.
template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void pidpos_law(const T y[13], const Book<T>& bk, PidState<T>& s, const T tgt[7], T t_step,
                                           T act[4]) {
    using m = M;
    const T P[6] = {T(6), T(6), T(6), T(22), T(22), T(1)}, D[6] = {T(3), T(3), T(3), T(12), T(12), T(0.1)};
    const T ts = T(0.01);
    T u[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T x = y[2 * k], dx = y[2 * k + 1];
        s.ix[k] = s.ix[k] + (tgt[k] - x) * ts;
        u[k] = P[k] * (tgt[k] - x) + D[k] * (tgt[3 + k] - dx);  // I = 0 (PID_control.py:24-27)
    }
    const T theta_d = m::atan2(m::div(u[0], u[2] + T(phys::G)), T(1));  // arctan
    const T ctd = m::cos(theta_d);
    const T phi_d = m::atan2(m::div(-u[1], u[2] + T(phys::G)) * ctd, T(1));
    act[0] = m::div(T(phys::M) * (u[2] + T(phys::G)), ctd * m::cos(phi_d));
    const T ang_d[3] = {phi_d, theta_d, tgt[6]};
    T u567[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const T v_ang_d = m::div(ang_d[k] - s.ang_d_ant[k], t_step);
        s.ix[3 + k] = s.ix[3 + k] + (ang_d[k] - bk.ang[k]) * ts;
        u567[k] = P[3 + k] * (ang_d[k] - bk.ang[k]) + D[3 + k] * (v_ang_d - bk.ang_vel[k]);
        s.ang_d_ant[k] = ang_d[k];
    }
    const T phi = bk.ang[0], theta = bk.ang[1];
    const T sp = m::sin(phi), cp = m::cos(phi), ct = m::cos(theta), tt = m::tan(theta);
    const T b1 = T(1.0 / phys::JX), b2 = tt * sp * T(1.0 / phys::JY), b3 = tt * cp * T(1.0 / phys::JZ);
    const T b4 = cp * T(1.0 / phys::JY), b5 = -sp * T(1.0 / phys::JZ);
    const T b6 = m::div(sp, ct) * T(1.0 / phys::JY), b7 = m::div(cp, ct) * T(1.0 / phys::JZ);
    const T det = b4 * b7 - b5 * b6;
    const T U3 = m::div(b7 * u567[1] - b5 * u567[2], det);
    const T U4 = m::div(-b6 * u567[1] + b4 * u567[2], det);
    act[1] = m::div(u567[0] - b2 * U3 - b3 * U4, b1);
    act[2] = U3;
    act[3] = U4;
}

}  // namespace qenv
