// qenv_sensor.hpp -- the reference's onboard sensor model, `class sensor` of environment/quadrotor_env.py:579-724, one lane
// per env: every method is a device function on the env's sensor state held in registers, so the same arithmetic serves
//   * k_sensor_ops  (qenv_ppo.hip): an arbitrary call sequence -- the N = 1 drop-in class `sensor` issues one method at a time,
//   * k_sensor_read (qenv_ppo.hip): rl_worker.py:164-175's fixed sequence accel_int, gyro_int, gyro, gps, triad (`sensor_sp`),
//   * k_step        (qenv_env_impl.hpp): the same sequence fused into the env step when the policy window is fed from the
//                   SENSED state (environment/position.py:240-242, REAL_CTRL = False).
// Noise enters as standard normals z (injected in the reference's draw order, or Philox): every np.random.normal(loc,
// scale, 3) of the reference is loc + scale * z here.
#pragma once
#include "qenv_core.hpp"

namespace qenv {

// rows of the caller-owned [QENV_SENSOR_ROWS][N] sensor state
// SR_QSAVE / SR_ISAVE: the env quaternion as it was BEFORE an aliased gyro_int advanced it, and the env's step counter + 1
// at that moment: quad.mat_rot is an attribute left behind by the last RHS call of the step, so methods called between
// that gyro_int and the next env step must still see the rotation of the un-advanced quaternion (k_sensor_ops)
// SR_IRESET: the env's step counter + 1 at sensor.reset().  reset() makes quaternion_t0 a VIEW of the array quad.state is
// bound to at that moment (:630); quad.step() rebinds quad.state to a new array, so the first gyro_int advances the env's
// own quaternion in place only if no env step happened in between -- otherwise it updates an orphaned array.
enum { SR_ABD = 0, SR_GBD = 1, SR_MBD = 2, SR_AB = 3, SR_GB = 4, SR_POS = 5, SR_VEL = 8, SR_Q = 11, SR_R = 15, SR_ALIAS = 24,
       SR_QSAVE = 25, SR_ISAVE = 29, SR_IRESET = 30 };
// methods (k_sensor_ops programs) and the number of normals / output rows of each
enum { SOP_ACCEL = 0, SOP_GYRO = 1, SOP_GPS = 2, SOP_TRIAD = 3, SOP_ACCEL_INT = 4, SOP_GYRO_INT = 5, SOP_COUNT = 6 };
__host__ __device__ inline int sop_normals(int op) { return op == SOP_GPS || op == SOP_TRIAD ? 6 : (op == SOP_ACCEL_INT ? 9 : 3); }
__host__ __device__ inline int sop_outputs(int op) {
    return op == SOP_GPS ? 6 : (op == SOP_TRIAD ? 13 : (op == SOP_ACCEL_INT ? 9 : (op == SOP_GYRO_INT ? 4 : 3)));
}

// sensor.__init__'s standard deviations and bias drifts (:588-592), already multiplied by sensor.error
template <typename T> struct SensParams { T std[5]; T bd[3]; };  // accel, gyro, magnet, gps pos, gps vel | accel, gyro, magnet

template <typename T> struct Sens {
    T std[5], a_b_d, g_b_d, a_b, g_b, R[9], pos[3], vel[3], q[4], dt;
    bool alias;  // quaternion_t0 is still the view reset() took (:630) AND quad.state is still that array (no env step since):
                 // the next gyro_int advances the ENV's quaternion too
};
// what the methods read off the env: quad.state, quad.mat_rot, quad.accel + [0, 0, -G] (accelerometer_read = mat_rot^T of
// it, :371), quad.f_in's z component -- all at the state the last step ended in
template <typename T> struct SensEnv { T y[13], Rm[9], a_in[3], f_thrust; };

template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void sens_env(const T y[13], const T accel[3], const T w[4], SensEnv<T>& ev) {
#pragma unroll
    for (int k = 0; k < 13; ++k) ev.y[k] = y[k];
    const T inv = M::rsqrt(y[6] * y[6] + y[7] * y[7] + y[8] * y[8] + y[9] * y[9]);
    const T a = y[6] * inv, b = y[7] * inv, c = y[8] * inv, d = y[9] * inv;
    const T Rm[9] = {a * a + b * b - c * c - d * d, T(2) * (b * c - a * d), T(2) * (b * d + a * c),
                     T(2) * (b * c + a * d), a * a - b * b + c * c - d * d, T(2) * (c * d - a * b),
                     T(2) * (b * d - a * c), T(2) * (c * d + a * b), a * a - b * b - c * c + d * d};
#pragma unroll
    for (int k = 0; k < 9; ++k) ev.Rm[k] = Rm[k];
    ev.a_in[0] = accel[0]; ev.a_in[1] = accel[1]; ev.a_in[2] = accel[2] - T(phys::G);
    T f = T(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) f += w[k] * w[k];
    ev.f_thrust = f * T(phys::K_F);
}

template <typename T>
__device__ __forceinline__ void sens_load(const T* sens, int64_t N, int64_t e, const SensParams<T>& p, T dt, int32_t env_i, Sens<T>& s) {
#pragma unroll
    for (int k = 0; k < 5; ++k) s.std[k] = p.std[k];
    s.a_b_d = sens[(int64_t)SR_ABD * N + e]; s.g_b_d = sens[(int64_t)SR_GBD * N + e];
    s.a_b = sens[(int64_t)SR_AB * N + e]; s.g_b = sens[(int64_t)SR_GB * N + e];
    s.dt = dt;
#pragma unroll
    for (int k = 0; k < 9; ++k) s.R[k] = sens[(int64_t)(SR_R + k) * N + e];
#pragma unroll
    for (int k = 0; k < 3; ++k) { s.pos[k] = sens[(int64_t)(SR_POS + k) * N + e]; s.vel[k] = sens[(int64_t)(SR_VEL + k) * N + e]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) s.q[k] = sens[(int64_t)(SR_Q + k) * N + e];
    s.alias = sens[(int64_t)SR_ALIAS * N + e] != T(0) && sens[(int64_t)SR_IRESET * N + e] == (T)(env_i + 1);
}
template <typename T>
__device__ __forceinline__ void sens_store(T* sens, int64_t N, int64_t e, const Sens<T>& s) {
    sens[(int64_t)SR_AB * N + e] = s.a_b; sens[(int64_t)SR_GB * N + e] = s.g_b;
#pragma unroll
    for (int k = 0; k < 9; ++k) sens[(int64_t)(SR_R + k) * N + e] = s.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { sens[(int64_t)(SR_POS + k) * N + e] = s.pos[k]; sens[(int64_t)(SR_VEL + k) * N + e] = s.vel[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) sens[(int64_t)(SR_Q + k) * N + e] = s.q[k];
}

// accel() :612-620
template <typename T>
__device__ __forceinline__ void sop_accel(Sens<T>& s, const SensEnv<T>& ev, const T* z, T out[3]) {
    s.a_b = s.a_b + s.a_b_d * s.dt;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        out[k] = (ev.Rm[k] * ev.a_in[0] + ev.Rm[3 + k] * ev.a_in[1] + ev.Rm[6 + k] * ev.a_in[2]) + (s.a_b + s.std[0] * z[k]);
}
// gyro() :623-629
template <typename T>
__device__ __forceinline__ void sop_gyro(Sens<T>& s, const SensEnv<T>& ev, const T* z, T out[3]) {
    s.g_b = s.g_b + s.g_b_d * s.dt;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = (s.g_b + s.std[1] * z[k]) + ev.y[10 + k];
}
// gps() :633-638
template <typename T>
__device__ __forceinline__ void sop_gps(const Sens<T>& s, const SensEnv<T>& ev, const T* z, T pos[3], T vel[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pos[k] = s.std[3] * z[k] + ev.y[2 * k];
        vel[k] = s.std[4] * z[3 + k] + ev.y[2 * k + 1];
    }
}
template <typename T, typename M> __device__ __forceinline__ void unit3(T a[3]) {
    const T n = M::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] = M::div(a[0], n); a[1] = M::div(a[1], n); a[2] = M::div(a[2], n);
}
template <typename T> __device__ __forceinline__ void cross3(const T a[3], const T b[3], T c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
// triad() :641-694, including scipy's Rotation.from_matrix(R.T).as_quat() (Shepperd's method); z: accel's 3 normals, then
// the magnetometer's 3
template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void sop_triad(Sens<T>& s, const SensEnv<T>& ev, const T* z, T q[4]) {
    const T mag[3] = {T(-40.47), T(129.11), T(-98.99)};
    T acc[3], gb[3], mb[3];
    const T ind[3] = {s.R[2] * T(phys::G), s.R[5] * T(phys::G), ev.f_thrust * T(1.0 / phys::M) + s.R[8] * T(phys::G)};
    sop_accel(s, ev, z, acc);
    const T mn[3] = {mag[0] + s.std[2] * z[3], mag[1] + s.std[2] * z[4], mag[2] + s.std[2] * z[5]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gb[k] = acc[k] - ind[k];
        mb[k] = ev.Rm[k] * mn[0] + ev.Rm[3 + k] * mn[1] + ev.Rm[6 + k] * mn[2];
    }
    unit3<T, M>(gb); unit3<T, M>(mb);
    T t1b[3] = {gb[0], gb[1], gb[2]}, t2b[3], t3b[3];
    unit3<T, M>(t1b);
    cross3(gb, mb, t2b); unit3<T, M>(t2b);
    cross3(t1b, t2b, t3b); unit3<T, M>(t3b);
    T gi[3] = {T(0), T(0), T(-1)}, mi[3] = {mag[0], mag[1], mag[2]}, t2i[3], t3i[3];
    unit3<T, M>(mi);
    cross3(gi, mi, t2i); unit3<T, M>(t2i);
    cross3(gi, t2i, t3i); unit3<T, M>(t3i);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) s.R[3 * r + c] = t1b[r] * gi[c] + t2b[r] * t2i[c] + t3b[r] * t3i[c];
    // quaternion of Mt = R^T
    const T Mt[9] = {s.R[0], s.R[3], s.R[6], s.R[1], s.R[4], s.R[7], s.R[2], s.R[5], s.R[8]};
    const T tr = Mt[0] + Mt[4] + Mt[8];
    T x[4];
    if (tr > Mt[0] && tr > Mt[4] && tr > Mt[8]) {  // argmax([M00, M11, M22, trace]) takes the FIRST maximum
        x[0] = Mt[7] - Mt[5]; x[1] = Mt[2] - Mt[6]; x[2] = Mt[3] - Mt[1]; x[3] = T(1) + tr;
    } else if (Mt[0] >= Mt[4] && Mt[0] >= Mt[8]) {
        x[0] = T(1) - tr + T(2) * Mt[0]; x[1] = Mt[3] + Mt[1]; x[2] = Mt[6] + Mt[2]; x[3] = Mt[7] - Mt[5];
    } else if (Mt[4] >= Mt[8]) {
        x[1] = T(1) - tr + T(2) * Mt[4]; x[2] = Mt[7] + Mt[5]; x[0] = Mt[1] + Mt[3]; x[3] = Mt[2] - Mt[6];
    } else {
        x[2] = T(1) - tr + T(2) * Mt[8]; x[0] = Mt[2] + Mt[6]; x[1] = Mt[5] + Mt[7]; x[3] = Mt[3] - Mt[1];
    }
    const T n = M::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
    q[0] = M::div(x[3], n); q[1] = M::div(x[0], n); q[2] = M::div(x[1], n); q[3] = M::div(x[2], n);
}
// accel_int() :697-710; z: accel's 3 normals, then triad's 6
template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void sop_accel_int(Sens<T>& s, const SensEnv<T>& ev, const T* z, T acc[3], T vel[3], T pos[3]) {
    T acc_b[3], qd[4];
    sop_accel(s, ev, z, acc_b);
    sop_triad<T, M>(s, ev, z + 3, qd);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        acc[k] = s.R[k] * acc_b[0] + s.R[3 + k] * acc_b[1] + s.R[6 + k] * acc_b[2] + (k == 2 ? T(phys::G) : T(0));
        vel[k] = s.vel[k] + acc[k] * s.dt;
        pos[k] = s.pos[k] + vel[k] * s.dt;
        s.vel[k] = vel[k]; s.pos[k] = pos[k];
    }
}
// deriv_quat(w, q) (quaternion_euler_utility.py:58-69 without its normalisation: sensor passes q as is)
template <typename T>
__device__ __forceinline__ void sens_deriv_quat(const T w3[3], const T qg[4], T vq[4]) {
    vq[0] = T(0.5) * (-w3[0] * qg[1] - w3[1] * qg[2] - w3[2] * qg[3]);
    vq[1] = T(0.5) * (w3[0] * qg[0] + w3[2] * qg[2] - w3[1] * qg[3]);
    vq[2] = T(0.5) * (w3[1] * qg[0] - w3[2] * qg[1] + w3[0] * qg[3]);
    vq[3] = T(0.5) * (w3[2] * qg[0] + w3[1] * qg[1] - w3[0] * qg[2]);
}
// gyro_int() :712-722.  Returns q (un-normalised, as the reference does) and leaves quaternion_t0 = q / |q|.  s.q holds
// quaternion_t0's values: after reset() those of quad.state at that moment.  While the view is live (s.alias) the ENV's
// quaternion advances in place: ev.y[6..9] receives q and wrote_env tells the caller to store it.
template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void sop_gyro_int(Sens<T>& s, SensEnv<T>& ev, const T* z, T q[4], bool& wrote_env) {
    T w3[3], vq[4];
    sop_gyro(s, ev, z, w3);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = s.q[k];
    sens_deriv_quat(w3, q, vq);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = q[k] + vq[k] * s.dt;
    wrote_env = s.alias;
    if (s.alias) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ev.y[6 + k] = q[k];
        s.alias = false;
    }
    const T nq = M::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) s.q[k] = M::div(q[k], nq);
}

// rl_worker.py:164-175 / position.py:226-232: accel_int, gyro_int, gyro, deriv_quat, gps, triad -- 27 normals in that
// order.  out24 = states_sens[14] {x, vx, y, vy, z, vz dead-reckoned; quaternion_gyro; quaternion_vel}, gps_pos[3],
// gps_vel[3], quaternion_triad[4]
template <typename T, typename M = Mth<T>>
__device__ __forceinline__ void sens_sp(Sens<T>& s, SensEnv<T>& ev, const T* z, T out24[24], bool& wrote_env) {
    T acc[3], vel[3], pos[3], qg[4], w3[3], vq[4], gp[3], gv[3], qt[4];
    sop_accel_int<T, M>(s, ev, z, acc, vel, pos);
    sop_gyro_int<T, M>(s, ev, z + 9, qg, wrote_env);
    sop_gyro(s, ev, z + 12, w3);
    sens_deriv_quat(w3, qg, vq);
    sop_gps(s, ev, z + 15, gp, gv);
    sop_triad<T, M>(s, ev, z + 21, qt);
#pragma unroll
    for (int k = 0; k < 3; ++k) { out24[2 * k] = pos[k]; out24[2 * k + 1] = vel[k]; out24[14 + k] = gp[k]; out24[17 + k] = gv[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { out24[6 + k] = qg[k]; out24[10 + k] = vq[k]; out24[20 + k] = qt[k]; }
}

// the 27 (28) standard normals of one sensor_sp call from Philox, keyed by (seed; global env, step)
template <typename T>
__device__ __forceinline__ void sens_philox_normals(uint64_t seed, uint64_t step, uint64_t g, T z[28]) {
#pragma unroll
    for (int blk = 0; blk < 7; ++blk) {
        uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)step, ((uint32_t)(step >> 32) & 0x0FFFFFFFu) | ((uint32_t)blk << 28)};
        philox_core(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5E50u);
        float a0, a1, a2, a3;
        philox_normal2(c[0], c[1], a0, a1); philox_normal2(c[2], c[3], a2, a3);
        z[4 * blk] = (T)a0; z[4 * blk + 1] = (T)a1; z[4 * blk + 2] = (T)a2; z[4 * blk + 3] = (T)a3;
    }
}

}  // namespace qenv
