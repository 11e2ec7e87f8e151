// qenv_env_impl.hpp -- env kernels (reset / step / fused rollout / controllers) and the host templates that launch
// them.  gfx950 only.  Included by qenv_env.hip (C-ABI + fp32), qenv_env_mix.hip, qenv_env_f64.hip, qenv_env_lean.hip.
//
// Data layout in HBM: one caller-provided workspace holding structure-of-arrays rows,
// row r of env e at (row_base + r) * N + e, so a wavefront's 64 lanes touch 64
// consecutive elements of each row (256 B for fp32): every global access is one fully
// coalesced wave-wide transaction.  One lane = one quadrotor; 256-thread workgroups
// (4 waves); the state never leaves VGPRs inside a step (or inside a fused rollout).
#pragma once
#include "qenv_host.hpp"
#include "qenv_sensor.hpp"
#include <hip/hip_runtime.h>
namespace qenv {
// Diagnostic build only (-DQENV_FUSED_STAMPS, tests/tools/fused_stamps.py): s_memtime stamps of a wave's phases in its last step.
#if defined(QENV_FUSED_STAMPS) && defined(QENV_ENV_UNIT_LEAN)
__device__ unsigned long long g_fused_stamps[1024 * 16];
#define FUSED_STAMP(slot)                                                                                      \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                            \
        if ((threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 1024)                             \
            g_fused_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (slot)] = t_;                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#else
#define FUSED_STAMP(slot) do { } while (0)
#endif
// Diagnostic build only (-DQENV_ROLLOUT_STAMPS, tests/tools/rollout_stamps.py): where a k_rollout launch spends its time, per wave
// (s_memtime = shader cycles for the phases, s_memrealtime = the 100 MHz wall clock for entry / exit across the waves).
#if defined(QENV_ROLLOUT_STAMPS) && defined(QENV_ENV_UNIT_LEAN)
__device__ unsigned long long g_rollout_stamps[1024 * 8];
#define ROLLOUT_STAMP(slot, real)                                                                              \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const unsigned long long t_ = (real) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
        const unsigned wv_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                              \
        if ((threadIdx.x & 63) == 0 && wv_ < 1024) g_rollout_stamps[wv_ * 8 + (slot)] = t_;                    \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
// ... and, inside ONE env step in the middle of the launch, the step's phases (tests/tools/rollout_phase_stamps.py): 0 step start,
// 1 controller law + mixing done, 2 integrator done, 3 Euler angles done, 4 done / reward done, 5 end of the step's epilogue
__device__ unsigned long long g_rollout_phase[1024 * 8];
#define ROLLOUT_PHASE(slot, on)                                                                                \
    do {                                                                                                       \
        if (on) {                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                        \
            const unsigned wv_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                          \
            if ((threadIdx.x & 63) == 0 && wv_ < 1024) g_rollout_phase[wv_ * 8 + (slot)] = t_;                 \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }                                                                                                      \
    } while (0)
#else
#define ROLLOUT_STAMP(slot, real) do { } while (0)
#define ROLLOUT_PHASE(slot, on) do { } while (0)
#endif

}  // namespace qenv
#if defined(QENV_FUSED_STAMPS) && defined(QENV_ENV_UNIT_LEAN)
#if defined(QENV_ENVSTEP_STAMPS)   // slots 10-15 inside the env step instead of inside the policy tile
#define ENVSTEP_STAMP(slot) FUSED_STAMP(slot)
#else
#define MLP3_STAMP(slot) FUSED_STAMP(slot)
#endif
#endif
#ifndef ENVSTEP_STAMP
#define ENVSTEP_STAMP(slot) do { } while (0)
#endif
#include "qenv_mlp3.hpp"
#include <type_traits>

namespace qenv {

// ---- kernel argument blocks (live in SGPRs) ------------------------------------------------
// T = arithmetic type, S = storage type of the workspace and of the caller's arrays.  S == T for the fp32 and
// fp64 handles; the mixed handle (fp64 rows, fp32 arithmetic) uses T = float, S = double: the state rows are
// advanced by state += (double)increment, so no per-step rounding of the state enters the closed loop.
using GustSrc = qenv_gust_src;

template <typename T, typename S = T> struct StepArgs {
    GustSrc gust;
    Cfg<T> cfg;
    S* ws;                 // workspace rows of type S
    int32_t* wsi;          // int rows (I, FLAGS)
    int64_t N;
    const S* actions;      // [4][N] or null (zero_control)
    const uint8_t* mask;   // null or [N]: only envs with mask != 0 are stepped
    const int32_t* idx;    // null, or a compacted list of env indices (k_compact_mask) ...
    const int32_t* count;  // ... and its length on the device: lane g steps env idx[g] if g < *count
    S* out_obs;            // [14][N] or null
    S* out_reward;         // [N] or null
    uint8_t* out_done;     // [N] or null
    S* out_mask;           // [N] or null: 1 - done
    float* win;            // observation ring or null
    int32_t win_slot;      // ring slot this step's slice goes to
    int32_t mode;          // QENV_STEP_* (mode bits only)
    int32_t ang_ok;        // every env's prev_ang row is the angle of its state row (see euler_update)
    int32_t outer_kind;    // outer position loop evaluated at the window push (QENV_OUTER_*)
    int32_t outer_rows;
    const S* outer_target; // VELPID: [rows][7] mission table (wave-uniform rows); POSPI: [3][N]
    T outer_gain[3];       // POSPI: P, D, I
    // sensor in the loop (position.py:240-242, REAL_CTRL = False): the window slice of a PUSH_WINDOW step comes from
    // sensor_sp evaluated right here on the state the step ended in (null: the true state)
    S* sens; const S* sens_z; S* sens_out;
    SensParams<S> sens_p;
    uint64_t sens_seed, sens_step;
    int64_t sens_env_offset;
    // overlapped PPO step: envs whose byte in env_skip is set are left to the side stream (null: none) ...
    const uint8_t* env_skip;
    // ... and the auto-reset of the worker loop (ppo.py:232-262) happens in this kernel's epilogue: an env that ends its
    // episode here banks its return, draws its next initial state (the streams of qenv_reset_sampled) and appends
    // itself to the list the warm-up rollout will walk; its byte in ar_flags_next marks it for the next step
    // ar_on = 2: IN-BAND warm-up (round 3).  The env that ends its episode here is reset in the epilogue as above, but no
    // list is kept and no warm-up launch follows: reset() 's T warm-up steps (quadrotor_env.py:448-453) are the env's next T
    // REGULAR steps -- recognised by its own step counter (i < T), they apply zero_control whatever the policy asked for,
    // push their slices into the common ring slot like everybody else, and mark their slot of the rollout invalid
    // (out_valid = 0: excluded from GAE's moments and from the loss).  A rollout step is then two launches, policy + step.
    int32_t ar_on;
    int32_t ar_key_episode;   // 1: the next initial state is keyed by the env's own episode count instead of ar_episode
    uint64_t ar_seed, ar_episode;
    int64_t ar_env_offset;
    int32_t* ar_idx;
    int32_t* ar_count;
    uint8_t* ar_flags_next;
    S* out_valid;             // [N] or null: 0 for an env whose step was one of its reset's warm-up steps (ar_on = 2), else 1
};

template <typename T, typename S = T> struct RolloutArgs {
    GustSrc gust;
    Cfg<T> cfg;
    S* ws;
    int32_t* wsi;
    int64_t N;
    int32_t n_steps;
    int32_t mode;
    int32_t pid_first;
    int32_t ang_ok;
    T const_action[4];
    T pid_target[7];
    const S* target_table;  // null, or [n_steps][7]: PIDPOS target of each step
    const int32_t* idx;     // null, or compacted env list + its device-side length (masked reset warm-up)
    const int32_t* count;
    float* win;             // null, or the observation ring: step s also writes its slice to slot (win_head+s)%win_T
    int32_t win_head, win_T;
    int32_t lane_shift;     // 6: every lane of a wave carries an env; < 6: only lanes 0 .. 2^lane_shift - 1 do (more, emptier
                            // waves: see launch_rollout)
    int32_t list_min_shift; // list mode (idx): >= 0 spreads the list over the launched waves as thinly as they allow, with at
                            // least 2^list_min_shift envs per wave; -1: dense (64 list entries per wave)
    LqrGains<T> lqr;
    PidGains<T> pid;
};

template <typename S> __device__ __forceinline__ S ld(const S* base, int row, int64_t N, int64_t e) {
    return base[(int64_t)row * N + e];
}
template <typename S, typename V> __device__ __forceinline__ void st(S* base, int row, int64_t N, int64_t e, V v) {
    base[(int64_t)row * N + e] = (S)v;
}

// per-env part of a masked quad.reset (quadrotor_env.py:428-445) with a device-drawn initial state: shared by k_reset and
// the auto-reset epilogue of k_step.  S = storage type; the Euler angles of the new state use S's IEEE functions in both.
template <typename S>
__device__ __forceinline__ void reset_env_sampled(S* ws, int32_t* wsi, int64_t N, int64_t e, uint64_t seed, uint64_t episode,
                                                  int64_t env_offset, bool bank, bool key_episode = false) {
    using m = Mth<S>;
    if (bank) {  // an episode ends here: bank its return before the counters are cleared
        st(ws, ROW_LAST_RETURN, N, e, ld(ws, ROW_EP_RETURN, N, e));
        const int32_t n_ep = wsi[IROW_EPISODES * N + e] + 1;
        wsi[IROW_EPISODES * N + e] = n_ep;
        if (key_episode) episode = (uint64_t)n_ep;  // the stream of the env's n-th restart, whenever it happens
    }
    float xf[13];
    sample_init_state(seed, episode, (uint64_t)(env_offset + e), xf);
    S x[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) { x[k] = (S)xf[k]; st(ws, ROW_STATE + k, N, e, x[k]); }
    const S* q = x + 6;
    st(ws, ROW_ANG + 0, N, e, m::atan2(S(2) * (q[0] * q[1] + q[2] * q[3]), S(1) - S(2) * (q[1] * q[1] + q[2] * q[2])));
    st(ws, ROW_ANG + 1, N, e, m::asin(S(2) * (q[0] * q[2] - q[3] * q[1])));
    st(ws, ROW_ANG + 2, N, e, m::atan2(S(2) * (q[0] * q[3] + q[1] * q[2]), S(1) - S(2) * (q[2] * q[2] + q[3] * q[3])));
    st(ws, ROW_PREV_SHAPING, N, e, S(0));
    st(ws, ROW_ABS_SUM, N, e, S(0));
    st(ws, ROW_EP_RETURN, N, e, S(0));
    st(ws, ROW_REWARD, N, e, S(0));
    wsi[IROW_I * N + e] = 0;
    wsi[IROW_FLAGS * N + e] = 0;
}


template <typename T, typename S>
__device__ __forceinline__ void load_book(const S* ws, const int32_t* wsi, int64_t N, int64_t e, bool opt, Book<T>& bk) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bk.prev_ang[k] = (T)ld(ws, ROW_PREV_ANG + k, N, e);
        bk.ang[k] = opt ? (T)ld(ws, ROW_ANG + k, N, e) : T(0);  // only the opt variant reads the old ang
        bk.ang_vel[k] = T(0);
    }
    bk.prev_shaping = (T)ld(ws, ROW_PREV_SHAPING, N, e);
    bk.abs_sum = (T)ld(ws, ROW_ABS_SUM, N, e);
    bk.ep_return = (T)ld(ws, ROW_EP_RETURN, N, e);
    bk.reward = T(0);
    bk.i = wsi[IROW_I * N + e];
    bk.flags = wsi[IROW_FLAGS * N + e];
}

template <typename T, typename S>
__device__ __forceinline__ void store_book(S* ws, int32_t* wsi, int64_t N, int64_t e, const Book<T>& bk, bool aux) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        st(ws, ROW_PREV_ANG + k, N, e, bk.prev_ang[k]);
        st(ws, ROW_ANG + k, N, e, bk.ang[k]);
        if (aux) st(ws, ROW_ANG_VEL + k, N, e, bk.ang_vel[k]);
    }
    st(ws, ROW_PREV_SHAPING, N, e, bk.prev_shaping);
    st(ws, ROW_ABS_SUM, N, e, bk.abs_sum);
    st(ws, ROW_EP_RETURN, N, e, bk.ep_return);
    st(ws, ROW_REWARD, N, e, bk.reward);
    wsi[IROW_I * N + e] = bk.i;
    wsi[IROW_FLAGS * N + e] = bk.flags;
}

// robust handles: the episode's plant parameters and the gust state of env e (rows ROW_ROB.., ROW_GUST..)
template <typename T, typename S, typename P>
__device__ __forceinline__ void load_plant(const S* ws, const int32_t* wsi, int64_t N, int64_t e, const GustSrc& g, P& pl) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pl.kf[k] = (T)ld(ws, ROW_ROB + k, N, e);
        pl.ir[k] = T(phys::I_R) * (T(1) + (T)ld(ws, ROW_ROB + 5 + k, N, e));
    }
    const T qm = T(phys::M) * (T(1) + (T)ld(ws, ROW_ROB + 4, N, e));
    pl.inv_m = T(1) / qm;
    pl.mg = qm * T(phys::G);
    const T J[3] = {T(phys::JX), T(phys::JY), T(phys::JZ)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pl.ij[k] = T(1) / (J[k] + J[k] * (T)ld(ws, ROW_ROB + 9 + k, N, e));
        pl.gust[k] = (T)ld(ws, ROW_GUST + k, N, e);
        pl.last[k] = (T)ld(ws, ROW_GUST + 3 + k, N, e);
        pl.wind[k] = T(0);
    }
    pl.redraw = false; pl.k = 0; pl.epoch = 0; pl.evals = -1;
    pl.table = g.table; pl.table_f64 = g.f64; pl.table_E = g.E; pl.table_K = g.K;
    pl.N = N; pl.e = e;
    const uint64_t ge = (uint64_t)(g.env_offset + e);
    pl.key0 = g.key0; pl.key1 = g.key1; pl.ctr0 = (uint32_t)ge; pl.ctr1 = (uint32_t)(ge >> 32);
    pl.episode = (uint32_t)wsi[IROW_EPISODES * N + e];
}
template <typename S, typename P>
__device__ __forceinline__ void store_gust(S* ws, int32_t* wsi, int64_t N, int64_t e, const P& pl) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { st(ws, ROW_GUST + k, N, e, pl.gust[k]); st(ws, ROW_GUST + 3 + k, N, e, pl.last[k]); }
    if (pl.evals >= 0) wsi[IROW_GUST_K * N + e] = pl.evals;  // draws the last redraw step consumed (host RNG replay)
}

#ifndef QENV_ENV_BLOCK_DEFAULT
#define QENV_ENV_BLOCK_DEFAULT 256
#endif
#ifndef QENV_STEP_MIN_WAVES
#define QENV_STEP_MIN_WAVES 1  // the same for k_step alone (lean rk45 at 2: 256 VGPRs + 236 B scratch; measured: config 5 0.50 ->
                               // 0.48 G env-steps/s, per-step API 5.8 -> 4.8 G/s, config 4 unchanged -- the spills cost more than the overlap gives)
#endif
#ifndef QENV_MIN_WAVES
#define QENV_MIN_WAVES 1  // __launch_bounds__ second argument (waves per SIMD the register allocator must allow)
#endif

// ---- quad.step for N envs: one lane per env ---------------------------------------------------------------------------
// The inputs of one env step (13 state rows, 4 action rows, the bookkeeping rows): what a lane requests from memory BEFORE
// it integrates.  One wave per SIMD is resident (register footprint of the scipy-exact integrator), so this one round trip
// -- ~2 us with every SIMD loading at once -- has nothing to hide behind: 28 % of a k_step wave's life (SQ counters, round
// 2).  k_step<..., TILES = true> caps the grid at the resident waves and has a wave request tile t + 1's inputs before it
// integrates tile t; the loop costs the allocator far more than the ~50 registers of the rows (334 -> 512 + AGPR spills),
// which eats the hidden round trip until a wave owns eight tiles (do_step's launch rule has the measurements).
template <typename T, typename S> struct StepIn {
    S ys[13];
    T raw[4];
    Book<T> bk;
    int64_t e;
    bool active;
};

template <typename T, typename S>
__device__ __forceinline__ void step_load(const StepArgs<T, S>& a, int64_t g, StepIn<T, S>& in) {
    in.active = false;
    in.e = 0;
    int64_t e = g;
    if (e >= a.N) return;
    if (a.idx) {  // masked steps (reset warm-ups) run on the compacted list: only ceil(count/64) waves do any work
        if (e >= *a.count) return;
        e = a.idx[e];
    } else if (a.mask && a.mask[e] == 0) return;
    if (a.env_skip && a.env_skip[e]) return;
    const int64_t N = a.N;
    const Cfg<T>& cfg = a.cfg;
    in.e = e;
    in.active = true;
#pragma unroll
    for (int k = 0; k < 13; ++k) in.ys[k] = ld(a.ws, ROW_STATE + k, N, e);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        in.raw[k] = a.actions ? (T)a.actions[(int64_t)k * N + e] : ((cfg.direct || k == 0) ? cfg.zero_control0 : T(0));
    if (a.mode != QENV_STEP_DYNAMICS_ONLY) load_book(a.ws, a.wsi, N, e, cfg.variant != 0, in.bk);
}

template <typename T, typename S, int INTEG, bool ROBUST, bool LEAN>
__device__ __forceinline__ void step_one(const StepArgs<T, S>& a, const StepIn<T, S>& in) {
    using M = typename MathSel<T, LEAN>::type;
    const int64_t e = in.e;
    const int64_t N = a.N;
    const Cfg<T>& cfg = a.cfg;
    constexpr bool MIXED = !std::is_same<T, S>::value;
    T y[13];
    S ys[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) { ys[k] = in.ys[k]; y[k] = (T)ys[k]; }
    T raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[k] = in.raw[k];
    T w[4], effort[4], cl[4], act[4], kq[13], dy[13];
    const T q_old[4] = {y[6], y[7], y[8], y[9]};
    Act<T> u;
    Plant<T, ROBUST, LEAN> pl;
    set_plant(pl, cfg);
    if constexpr (ROBUST) {
        load_plant<T>(a.ws, a.wsi, N, e, a.gust, pl);
        begin_step_wind(pl, a.wsi[IROW_I * N + e] + 1);  // self.i += 1 comes first in step (:467)
    }
    Book<T> bk = in.bk;   // (requested with the state rows, ahead of the integration)
    bool warming = false;
    if (a.ar_on == 2 && a.mode != QENV_STEP_DYNAMICS_ONLY) {  // in-band warm-up: steps 1..T after a reset apply zero_control (:448-451)
        warming = bk.i < cfg.warmup;
        if (warming) {
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[k] = (cfg.direct || k == 0) ? cfg.zero_control0 : T(0);
        }
    }
    prepare_action(raw, cfg, pl, w, effort, cl, act, u);
    integrate<T, INTEG>(y, u, pl, cfg.t_step, kq, dy);
    if constexpr (ROBUST) store_gust(a.ws, a.wsi, N, e, pl);
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        if (MIXED) { ys[k] += (S)dy[k]; y[k] = (T)ys[k]; st(a.ws, ROW_STATE + k, N, e, ys[k]); }
        else st(a.ws, ROW_STATE + k, N, e, y[k]);
    }
    if (a.mode == QENV_STEP_DYNAMICS_ONLY) return;

    bk.i += 1;
    euler_update<T, M>(q_old, y, dy, a.ang_ok && (bk.flags & FLAG_ANG_OK), cfg, bk);
    done_reward<T, M>(y, act, effort, cfg, bk);
    const bool aux = a.mode != QENV_STEP_NO_AUX;
    store_book(a.ws, a.wsi, N, e, bk, aux);
#pragma unroll
    for (int k = 0; k < 4; ++k) st(a.ws, ROW_VQ + k, N, e, kq[6 + k]);
    if (aux) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            st(a.ws, ROW_W + k, N, e, w[k]);
            st(a.ws, ROW_EFFORT + k, N, e, effort[k]);
            st(a.ws, ROW_CLIPPED + k, N, e, cl[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) st(a.ws, ROW_ACCEL + k, N, e, kq[1 + 2 * k]);
    }
    if (a.out_obs) {
#pragma unroll
        for (int k = 0; k < 10; ++k) a.out_obs[(int64_t)k * N + e] = MIXED ? ys[k] : (S)y[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) a.out_obs[(int64_t)(10 + k) * N + e] = (S)kq[6 + k];
    }
    if (a.out_reward) a.out_reward[e] = (S)bk.reward;
    if (a.out_done) a.out_done[e] = (bk.flags & FLAG_DONE) ? 1 : 0;
    if (a.out_mask) a.out_mask[e] = (bk.flags & FLAG_DONE) ? S(0) : S(1);
    if (a.out_valid) a.out_valid[e] = warming ? S(0) : S(1);
    if (a.win) {  // dl_in_gen.dl_input: [raw action(4), vel(3), q(4), q_dot(4)] (dl_auxiliary.py:27-31)
        float* wv = a.win + (int64_t)a.win_slot * 15 * N + e;
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[(int64_t)k * N] = (float)raw[k];
        T voff[3] = {T(0), T(0), T(0)};
        if (a.outer_kind != QENV_OUTER_NONE) {  // the policy sees state - state_error (position.py:233-244)
            T integ[3], cmd[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) integ[k] = (T)ld(a.ws, ROW_OUTER + k, N, e);
            if (a.outer_kind == QENV_OUTER_VELPID) {  // velocity_pid.py:16-35
                int row = bk.i - cfg.warmup - 1;     // k-th policy step after the T warm-ups
                row = row < 0 ? 0 : (row >= a.outer_rows ? a.outer_rows - 1 : row);
                const S* tg = a.outer_target + (int64_t)row * 7;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T err = (T)tg[k] - y[2 * k];
                    const T prev = (T)ld(a.ws, ROW_OUTER + 3 + k, N, e);
                    integ[k] = integ[k] + err * T(0.01);
                    const T der = (err - prev) / T(0.01);
                    cmd[k] = T(3) * err + T(0) * integ[k] + T(0.05) * der;
                    voff[k] = (T)tg[3 + k] + cmd[k];
                    st(a.ws, ROW_OUTER + 3 + k, N, e, err);
                }
            } else {  // pos_pi, math_trajectory.py:46-62, clip of :123
                const T lo[3] = {T(-0.5), T(-0.5), T(-1.666)}, hi[3] = {T(0.5), T(0.5), T(0.333)};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T err = (T)a.outer_target[(int64_t)k * N + e] - y[2 * k];
                    integ[k] += err * cfg.t_step;
                    const T v = a.outer_gain[0] * err + a.outer_gain[1] * (T(0) - y[2 * k + 1]) + a.outer_gain[2] * integ[k];
                    cmd[k] = v < lo[k] ? lo[k] : (v > hi[k] ? hi[k] : v);
                    voff[k] = cmd[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { st(a.ws, ROW_OUTER + k, N, e, integ[k]); st(a.ws, ROW_OUTER + 6 + k, N, e, cmd[k]); }
        }
        if (a.sens) {
            // states_sens = [dead-reckoned x, vx, y, vy, z, vz; quaternion_gyro; deriv_quat(gyro, quaternion_gyro)] - error_mission
            // (position.py:233-242); the sensor's arithmetic runs in the handle's storage type, as k_sensor_read's does
            S zz[28];
            if (a.sens_z) {
#pragma unroll
                for (int k = 0; k < 27; ++k) zz[k] = a.sens_z[(int64_t)k * N + e];
            } else {
                sens_philox_normals<S>(a.sens_seed, a.sens_step, (uint64_t)(a.sens_env_offset + e), zz);
            }
            Sens<S> sn;
            sens_load(a.sens, N, e, a.sens_p, (S)cfg.t_step, bk.i, sn);  // (bk.i: a step has just happened, the view is dead)
            S yS[13], accS[3], wS[4];
#pragma unroll
            for (int k = 0; k < 13; ++k) yS[k] = MIXED ? ys[k] : (S)y[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) accS[k] = (S)kq[1 + 2 * k];
#pragma unroll
            for (int k = 0; k < 4; ++k) wS[k] = (S)w[k];
            SensEnv<S> ev;
            sens_env<S>(yS, accS, wS, ev);
            S o[24];
            bool wrote_env;
            sens_sp<S>(sn, ev, zz, o, wrote_env);
            if (wrote_env) {  // quaternion_t0 aliased quad.state: the env's own quaternion advances (quirk of :630, :719)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    st(a.ws, ROW_STATE + 6 + k, N, e, ev.y[6 + k]);
                    a.sens[(int64_t)(SR_QSAVE + k) * N + e] = yS[6 + k];  // mat_rot stays that of the un-advanced quaternion
                }
                a.sens[(int64_t)SR_ISAVE * N + e] = (S)(bk.i + 1);
            }
            sens_store(a.sens, N, e, sn);
            a.sens[(int64_t)SR_ALIAS * N + e] = S(0);
            if (a.sens_out) {
#pragma unroll
                for (int k = 0; k < 24; ++k) a.sens_out[(int64_t)k * N + e] = o[k];
            }
            wv[4 * N] = (float)((T)o[1] - voff[0]); wv[5 * N] = (float)((T)o[3] - voff[1]); wv[6 * N] = (float)((T)o[5] - voff[2]);
#pragma unroll
            for (int k = 0; k < 4; ++k) { wv[(int64_t)(7 + k) * N] = (float)o[6 + k]; wv[(int64_t)(11 + k) * N] = (float)o[10 + k]; }
        } else {
        wv[4 * N] = (float)(y[1] - voff[0]); wv[5 * N] = (float)(y[3] - voff[1]); wv[6 * N] = (float)(y[5] - voff[2]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { wv[(int64_t)(7 + k) * N] = (float)y[6 + k]; wv[(int64_t)(11 + k) * N] = (float)kq[6 + k]; }
        }
    }
    if (a.ar_on == 2) {
        // in-band: an episode that ends in a POLICY step restarts here (a done raised by a warm-up step stays sticky, like
        // the done reset() ignores: the first policy step then reports it and the episode has length one)
        if ((bk.flags & FLAG_DONE) != 0 && !warming)
            reset_env_sampled<S>(a.ws, a.wsi, N, e, a.ar_seed, a.ar_episode, a.ar_env_offset, true, a.ar_key_episode != 0);
    } else if (a.ar_on) {  // wave-uniform; the lanes still here are the envs this launch stepped
        const bool fin = (bk.flags & FLAG_DONE) != 0;
        const unsigned long long b = __ballot(fin);
        const int lane = threadIdx.x & 63;
        a.ar_flags_next[e] = fin ? 1 : 0;  // every env is stepped by exactly one launch of a step: all N bytes get written
        if (b) {
            int base = 0;
            const int leader = __ffsll((long long)b) - 1;
            if (lane == leader) base = atomicAdd(a.ar_count, __popcll(b));
            base = __shfl(base, leader);
            if (fin) {
                a.ar_idx[base + __popcll(b & ((1ull << lane) - 1ull))] = (int32_t)e;
                reset_env_sampled<S>(a.ws, a.wsi, N, e, a.ar_seed, a.ar_episode, a.ar_env_offset, true, a.ar_key_episode != 0);
            }
        }
    }
}

// TILES = false: one 64-env tile per wave, the grid covers the batch (every integrator below the chip's resident-wave count,
// and the fixed-step integrators always: they keep 2+ waves per SIMD and the tile loop's registers would cost one).
template <typename T, typename S, int INTEG, bool ROBUST, bool LEAN = false, bool TILES = false>
__global__ __launch_bounds__(256, QENV_STEP_MIN_WAVES) void k_step(const StepArgs<T, S> a) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    StepIn<T, S> cur;
    step_load(a, g, cur);
    if constexpr (!TILES) {
        if (cur.active) step_one<T, S, INTEG, ROBUST, LEAN>(a, cur);
    } else {
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
#pragma unroll 1
        for (;;) {
            const int64_t gn = g + stride;
            const bool more = gn - (int64_t)(threadIdx.x & 63) < a.N;   // wave-uniform: the wave's next tile starts inside the batch
            StepIn<T, S> nxt;
            nxt.active = false;
            if (more) step_load(a, gn, nxt);   // in flight during this tile's integration
            if (cur.active) step_one<T, S, INTEG, ROBUST, LEAN>(a, cur);
            if (!more) break;
            cur = nxt;
            g = gn;
        }
    }
}

// ---- the policy and the env step of a whole rollout in ONE launch --------------------------------------------------------
// n_steps x (policy forward [+ sampling], env step) for the reference's N = 128, T = 5 controller (ppo.py:239-262; the evaluation
// loop ppo_quad_eval.py:44-62 without sampling).  As two launches per step (k_mlp_v3, k_step) a rollout step of 65 536 envs is
// mostly fixed cost: every launch of the forward stages 159 KiB of weights per workgroup (13 k of its ~40 k cycles), every
// launch of the step kernel starts with a load round trip that one wave per SIMD cannot hide, and both pay launch + drain.
// Here a wave OWNS its 64-env blocks for the whole rollout: per step and block it runs mlp3_tile_forward on the block's two
// 32-env tiles and then the env step on the 64 envs -- the same device functions on the same values in the same order as the two
// kernels, so every buffer is bit-identical to the stepwise path (tests) -- and the weights are staged once per launch.
// Nothing crosses waves (an env's window, action and state are touched by its own wave only): no grid-wide synchronisation,
// and since round 4 no memory ordering inside the loop either -- the block's observation windows (75 floats per lane, env
// layout) AND its env state and bookkeeping (fused_env_step below) stay in registers for the whole horizon: the MFMA operands
// of the two tiles come out of v_permlane32_swap, the observation copy is stored from the env layout, the slice a step pushes
// goes straight into the window registers, the action reaches the env step in registers, and the state / bookkeeping rows and
// the ring are written by the launch's last step.  (Round 3 ran the step through memory behind a workgroup-scope fence; an
// agent-scope fence, the first version, wrote back / invalidated the XCD's L2 every step: 25 k cycles for the loads behind it.)
// Both tiles' outputs are finished in one 64-lane pass (exact tanh, stores, sampling).
// ---- the env step of k_ppo_fused: state and bookkeeping of the block's envs stay in REGISTERS for the whole horizon ---------
// step_one's arithmetic, call for call (prepare_action, integrate, euler_update, done_reward on the same values: every buffer
// the fused launch writes is compared bit for bit with the per-step launches), without its memory round trip: per step only what
// the rollout keeps is stored (reward / done / mask / valid rows); the 13 state rows, the bookkeeping rows, V_q and the ring are
// written once, by the launch's last step.  (With the rows going through memory every step a wave issued 25 row loads and ~47
// row stores of 512 B per step on a CU path of ~12 B per clock, plus the 64-bit address arithmetic of each.)
template <typename T, typename S> struct EnvRegs {
    S ys[13];      // state rows (storage type)
    Book<T> bk;    // prev_ang, ang, prev_shaping, abs_sum, ep_return, reward, i, flags
};

// reset_env_sampled on registers (the in-band restart in the step's epilogue): the same draws, the same IEEE angle functions of the
// storage type, the same cleared counters; the banked return and the episode count go to memory (a rare event)
template <typename T, typename S>
__device__ __forceinline__ void reset_env_regs(S* ws, int32_t* wsi, int64_t N, int64_t e, uint64_t seed, uint64_t episode,
                                               int64_t env_offset, bool key_episode, EnvRegs<T, S>& r, S ang_s[3]) {
    using m = Mth<S>;
    st(ws, ROW_LAST_RETURN, N, e, (S)r.bk.ep_return);   // (what the memory version reads back from the row store_book has just written)
    const int32_t n_ep = wsi[IROW_EPISODES * N + e] + 1;
    wsi[IROW_EPISODES * N + e] = n_ep;
    if (key_episode) episode = (uint64_t)n_ep;
    float xf[13];
    sample_init_state(seed, episode, (uint64_t)(env_offset + e), xf);
#pragma unroll
    for (int k = 0; k < 13; ++k) r.ys[k] = (S)xf[k];
    const S* q = r.ys + 6;
    // (ang_s: the angle rows in the STORAGE type, as reset_env_sampled writes them -- on the mixed handle, fp64 rows and fp32
    //  arithmetic, the bookkeeping's copy is the rounded one a later load_book would read)
    ang_s[0] = m::atan2(S(2) * (q[0] * q[1] + q[2] * q[3]), S(1) - S(2) * (q[1] * q[1] + q[2] * q[2]));
    ang_s[1] = m::asin(S(2) * (q[0] * q[2] - q[3] * q[1]));
    ang_s[2] = m::atan2(S(2) * (q[0] * q[3] + q[1] * q[2]), S(1) - S(2) * (q[2] * q[2] + q[3] * q[3]));
#pragma unroll
    for (int k = 0; k < 3; ++k) r.bk.ang[k] = (T)ang_s[k];
    r.bk.prev_shaping = T(0); r.bk.abs_sum = T(0); r.bk.ep_return = T(0); r.bk.reward = T(0);
    r.bk.i = 0; r.bk.flags = 0;
}

// one env step of lane env e on the registers `r` with the raw action `raw_in`; slice[15] = the observation slice this step pushes
// (dl_in_gen.dl_input's row, as floats); `last`: the launch's final step writes the rows back
template <typename T, typename S, int INTEG, bool LEAN>
__device__ __forceinline__ void fused_env_step(const StepArgs<T, S>& a, int64_t e, EnvRegs<T, S>& r, const T raw_in[4], bool last,
                                               float slice[15]) {
    using M = typename MathSel<T, LEAN>::type;
    const int64_t N = a.N;
    const Cfg<T>& cfg = a.cfg;
    constexpr bool MIXED = !std::is_same<T, S>::value;
    T y[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) y[k] = (T)r.ys[k];
    T raw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[k] = raw_in[k];
    T w[4], effort[4], cl[4], act[4], kq[13], dy[13];
    const T q_old[4] = {y[6], y[7], y[8], y[9]};
    Act<T> u;
    Plant<T, false, LEAN> pl;
    set_plant(pl, cfg);
    Book<T>& bk = r.bk;
    bool warming = false;
    if (a.ar_on == 2) {  // in-band warm-up: steps 1..T after a reset apply zero_control (:448-451)
        warming = bk.i < cfg.warmup;
        if (warming) {
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[k] = (cfg.direct || k == 0) ? cfg.zero_control0 : T(0);
        }
    }
    prepare_action(raw, cfg, pl, w, effort, cl, act, u);
    ENVSTEP_STAMP(10);
    integrate<T, INTEG>(y, u, pl, cfg.t_step, kq, dy);
    ENVSTEP_STAMP(11);
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        if (MIXED) { r.ys[k] += (S)dy[k]; y[k] = (T)r.ys[k]; }
        else r.ys[k] = (S)y[k];
    }
    bk.i += 1;
    euler_update<T, M>(q_old, y, dy, a.ang_ok && (bk.flags & FLAG_ANG_OK), cfg, bk);
    ENVSTEP_STAMP(12);
    done_reward<T, M>(y, act, effort, cfg, bk);
    ENVSTEP_STAMP(13);
    if (a.out_reward) a.out_reward[e] = (S)bk.reward;
    if (a.out_done) a.out_done[e] = (bk.flags & FLAG_DONE) ? 1 : 0;
    if (a.out_mask) a.out_mask[e] = (bk.flags & FLAG_DONE) ? S(0) : S(1);
    if (a.out_valid) a.out_valid[e] = warming ? S(0) : S(1);
    // dl_in_gen.dl_input: [raw action(4), vel(3), q(4), q_dot(4)] (dl_auxiliary.py:27-31), as k_step pushes it into the ring
#pragma unroll
    for (int k = 0; k < 4; ++k) slice[k] = (float)raw[k];
    slice[4] = (float)(y[1] - T(0)); slice[5] = (float)(y[3] - T(0)); slice[6] = (float)(y[5] - T(0));
#pragma unroll
    for (int k = 0; k < 4; ++k) { slice[7 + k] = (float)y[6 + k]; slice[11 + k] = (float)kq[6 + k]; }
    if (last) {   // (wave-uniform) what the per-step launches leave in the workspace after this step, before a restart
#pragma unroll
        for (int k = 0; k < 13; ++k) st(a.ws, ROW_STATE + k, N, e, r.ys[k]);
        store_book(a.ws, a.wsi, N, e, bk, false);
#pragma unroll
        for (int k = 0; k < 4; ++k) st(a.ws, ROW_VQ + k, N, e, kq[6 + k]);
    }
    if (a.ar_on == 2 && (bk.flags & FLAG_DONE) != 0 && !warming) {
        // in-band: an episode that ends in a POLICY step restarts here (see k_step's epilogue)
        S ang_s[3];
        reset_env_regs<T, S>(a.ws, a.wsi, N, e, a.ar_seed, a.ar_episode, a.ar_env_offset, a.ar_key_episode != 0, r, ang_s);
        if (last) {   // the rows reset_env_sampled overwrites
#pragma unroll
            for (int k = 0; k < 13; ++k) st(a.ws, ROW_STATE + k, N, e, r.ys[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) st(a.ws, ROW_ANG + k, N, e, ang_s[k]);
            st(a.ws, ROW_PREV_SHAPING, N, e, S(0)); st(a.ws, ROW_ABS_SUM, N, e, S(0));
            st(a.ws, ROW_EP_RETURN, N, e, S(0)); st(a.ws, ROW_REWARD, N, e, S(0));
            a.wsi[IROW_I * N + e] = 0; a.wsi[IROW_FLAGS * N + e] = 0;
        }
    }
}

template <typename T, typename S> struct FusedArgs {
    StepArgs<T, S> st;   // the step kernel's arguments of step 0
    MlpArgs ml;          // the policy forward's arguments of step 0
    int32_t n_steps;
    int32_t obs_slices;  // stored observations as slices: step t > 0 stores its newest slice only (see qenv.h)
    int32_t block_shift; // 6: a wave owns 64-env blocks (two policy tiles + one full-wave env step per step); 5: 32-env blocks (one
                         // tile, the env step on half the lanes) -- small batches spread over twice the waves (do_ppo_fused)
    // bytes by which the per-step arrays advance from one step to the next (0: the array is scratch / not per step)
    int64_t adv_act, adv_logp, adv_noise, adv_obs, adv_row;
};

template <typename T, typename S, int INTEG, bool LEAN, bool SAMPLE>
__global__ __launch_bounds__(256) void k_ppo_fused(const FusedArgs<T, S> fa) {
    using namespace mlp3;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    const int lane = threadIdx.x & 63, half = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t N = fa.ml.N;
    // ---- the weights, once per launch
    if (fa.ml.image) {
        for (int base = threadIdx.x; base < IMAGE_UINT4; base += 8 * 256) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = base + u * 256; v[u] = fa.ml.image[i < IMAGE_UINT4 ? i : IMAGE_UINT4 - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (base + u * 256 < IMAGE_UINT4) lds4[base + u * 256] = v[u];
        }
    } else {
        mlp3_build_image(lds4, fa.ml.W1, fa.ml.b1, fa.ml.W2, fa.ml.b2, fa.ml.W3, fa.ml.b3, fa.ml.n_out, (int)threadIdx.x, 256);
    }
    __syncthreads();
    const int bsh = fa.block_shift, benvs = 1 << bsh;
    const int64_t n_blocks = (N + benvs - 1) >> bsh;
    const int64_t wave_stride = (int64_t)gridDim.x * 4;
#pragma unroll 1
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blocks; blk += wave_stride) {
        // ---- the block's 64 windows live in REGISTERS for the whole horizon: w[k] = element k (time order) of this lane's env.
        // Read from the ring once; per step the slice the env step pushes enters the registers directly and the window shifts by
        // one slice.  (Reading the two tiles' windows from the ring every step was 80 loads of 256 B per wave and step on a CU
        // path of ~12 B per clock: ~2.5 k cycles per tile in front of its first MFMA.)
        const int64_t eb = (blk << bsh) + lane;
        const bool ev = eb < N && lane < benvs;
        const int64_t ebc = ev ? eb : N - 1;
        float w[KW];
#pragma unroll
        for (int sl = 0; sl < TW; ++sl) {
            int slot = fa.ml.head + sl;
            if (slot >= TW) slot -= TW;
            const float* src = fa.ml.win + (int64_t)slot * 15 * N + ebc;
#pragma unroll
            for (int r = 0; r < 15; ++r) w[15 * sl + r] = src[(int64_t)r * N];
        }
        // ---- the block's env state and bookkeeping: read once, carried in registers (fused_env_step), written back by the last step
        EnvRegs<T, S> er;
        if (ev) {
#pragma unroll
            for (int k = 0; k < 13; ++k) er.ys[k] = ld(fa.st.ws, ROW_STATE + k, N, eb);
            load_book(fa.st.ws, fa.st.wsi, N, eb, fa.st.cfg.variant != 0, er.bk);
        }
#pragma unroll 1
        for (int t = 0; t < fa.n_steps; ++t) {
            asm volatile("" ::: "memory");   // (the weights in LDS are loop-invariant: nothing of them is to be hoisted)
            const int head = (fa.ml.head + t) % TW;   // wave-uniform
            // ---- this step's view of the per-step arrays
            MlpArgs m = fa.ml;
            m.step = fa.ml.step + (uint64_t)t;
            if (m.noise) m.noise = (const char*)fa.ml.noise + fa.adv_noise * t;
            if (SAMPLE) {
                m.out_action = (char*)fa.ml.out_action + fa.adv_act * t;
                if (m.out_logprob) m.out_logprob = (char*)fa.ml.out_logprob + fa.adv_logp * t;
            } else {
                m.out = (char*)fa.ml.out + fa.adv_act * t;
            }
            if (m.obs_store) m.obs_store = (float*)((char*)fa.ml.obs_store + fa.adv_obs * t);
            if (fa.obs_slices && t > 0) m.obs_rows_from = 15 * (TW - 1);
            // ---- the observation copy: rows >= obs_rows_from of the window, from the env layout (one 256-byte row store per element)
            if (m.obs_store && ev) {
                float* __restrict__ obs_c = m.obs_store + eb;
#pragma unroll
                for (int k = 0; k < KW; ++k)
                    if (k >= m.obs_rows_from) obs_c[(int64_t)k * N] = w[k];
            }
            // ---- this step's view of the step arguments (the action reaches the step in registers, below)
            StepArgs<T, S> s = fa.st;
            s.actions = nullptr;
            if (s.out_reward) s.out_reward = (S*)((char*)fa.st.out_reward + fa.adv_row * t);
            if (s.out_mask) s.out_mask = (S*)((char*)fa.st.out_mask + fa.adv_row * t);
            if (s.out_valid) s.out_valid = (S*)((char*)fa.st.out_valid + fa.adv_row * t);
            s.win_slot = head;
            s.ar_episode = fa.st.ar_episode + (uint64_t)t;
            if (t > 0) s.ang_ok = 1;   // (as the host sets it after an unmasked full step: every prev_ang row is current from here on)
            // ---- the policy on the block's two 32-env tiles
            FUSED_STAMP(0);
            // the MFMA's B operand wants element 16 sk + 8 half + i of env col in lane (col, half): v_permlane32_swap of
            // (w[16 sk + i], w[16 sk + 8 + i]) leaves tile 0's operand in the first register (lanes 32-63 receive the second
            // register's lower half: envs 0-31, elements + 8) and tile 1's in the second (lanes 0-31 receive the first register's
            // upper half: envs 32-63)
            float o0[4], o1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int tl = 0; tl < 2; ++tl) {
                if ((blk << bsh) + 32 * tl >= N || 32 * tl >= benvs) break;   // (wave-uniform: a ragged last block / a 32-env block has one tile)
                float xw[KS16][8];   // (built per tile: both tiles' operands at once would be 40 more live registers)
#pragma unroll
                for (int sk = 0; sk < KS16; ++sk)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float a0 = 16 * sk + i < KW ? w[16 * sk + i < KW ? 16 * sk + i : 0] : 0.f;
                        const float a1 = 16 * sk + 8 + i < KW ? w[16 * sk + 8 + i < KW ? 16 * sk + 8 + i : 0] : 0.f;
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
                        xw[sk][i] = __uint_as_float(tl ? r[1] : r[0]);
                    }
                FUSED_STAMP(1 + 3 * tl);
                float o[4];
                mlp3_tile_forward(lds4, lane, half, xw, o);
                FUSED_STAMP(2 + 3 * tl);
#pragma unroll
                for (int k = 0; k < 4; ++k) { if (tl == 0) o0[k] = o[k]; else o1[k] = o[k]; }
                FUSED_STAMP(3 + 3 * tl);
            }
            // ---- the outputs of both tiles at once, in the env layout (lane = env of the block): tile 0's are valid in lanes 0-31
            // of o0, tile 1's in lanes 0-31 of o1 -- v_permlane32_swap puts the latter into lanes 32-63.  One pass of the exact
            // tanh, of the stores and of ActorCritic.act's sampling on 64 lanes instead of two passes on 32 (k_mlp_v3's arithmetic
            // per env: the same values)
            {
                float outv[4];
                S act[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o0[k]), __float_as_uint(o1[k]), false, false);
                    float v = __uint_as_float(r[0]);
                    if (m.final_tanh) v = tanhf(v);
                    outv[k] = v;
                    act[k] = (S)v;
                    if (ev && k < m.n_out && m.out) ((S*)m.out)[(int64_t)k * N + eb] = (S)v;
                }
                if constexpr (SAMPLE) {
                    S logp[4];
                    if (ev) {
                        sample_compute<S>(m, eb, outv, act, logp);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            ((S*)m.out_action)[(int64_t)k * N + eb] = act[k];
                            if (m.out_logprob) ((S*)m.out_logprob)[(int64_t)k * N + eb] = logp[k];
                        }
                    }
                }
                // ---- the env step of the block's 64 envs on the actions just computed (what qenv_step would read back from
                // the action buffer: the stored value, converted to the arithmetic type)
                FUSED_STAMP(7);
                T rawT[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) rawT[k] = (T)act[k];
                float nw[15];
#pragma unroll
                for (int r = 0; r < 15; ++r) nw[r] = 0.f;
                if (ev) fused_env_step<T, S, INTEG, LEAN>(s, eb, er, rawT, t + 1 == fa.n_steps, nw);
                FUSED_STAMP(8);
                // the slice this step pushes becomes the window's newest; the window shifts by one slice (registers only: the
                // ring in memory is brought up to date after the last step)
#pragma unroll
                for (int k = 0; k < KW - 15; ++k) w[k] = w[k + 15];
#pragma unroll
                for (int r = 0; r < 15; ++r) w[KW - 15 + r] = nw[r];
            }
            FUSED_STAMP(9);
        }
        // ---- the ring as the per-step launches leave it: time position j of the window lives in slot (head + n_steps + j) mod T
        if (ev) {
            const int head_end = (fa.ml.head + fa.n_steps) % TW;   // the slot of the oldest slice after n_steps pushes
#pragma unroll
            for (int sl = 0; sl < TW; ++sl) {
                int slot = head_end + sl;
                if (slot >= TW) slot -= TW;
                float* dst = const_cast<float*>(fa.ml.win) + (int64_t)slot * 15 * N + eb;
#pragma unroll
                for (int r = 0; r < 15; ++r) dst[(int64_t)r * N] = w[15 * sl + r];
            }
        }
    }
}

// n_steps env steps in one launch, controller on device, state in VGPRs throughout.
template <typename T, typename S, int INTEG, int CTRL, bool ROBUST, bool LEAN = false>
__global__ __launch_bounds__(256, QENV_MIN_WAVES) void k_rollout(const RolloutArgs<T, S> a) {
    using M = typename MathSel<T, LEAN>::type;
    ROLLOUT_STAMP(0, true); ROLLOUT_STAMP(1, false);
    const int wpb = blockDim.x >> 6;  // waves per workgroup (env_block())
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.lane_shift < 6) {
        const int ln = threadIdx.x & 63;
        if (ln >> a.lane_shift) return;
        e = (((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) << a.lane_shift) + ln;
    }
    if (!a.idx && e >= a.N) return;
    if (a.idx) {
        const int64_t cnt = *a.count;  // <= N
        if (a.list_min_shift >= 0) {
            // the fewest envs per wave the launched waves allow: a wave runs every env step as many RK45 attempts as its
            // slowest lane, and the SIMDs a short list leaves idle are free (see do_rollout)
            const int64_t n_waves = (int64_t)gridDim.x * wpb;
            int sh = a.list_min_shift;
            while (sh < 6 && (n_waves << sh) < cnt) ++sh;
            const int ln = threadIdx.x & 63;
            if (ln >> sh) return;
            e = (((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) << sh) + ln;
        }
        if (e >= cnt) return;
        e = a.idx[e];
    }
    const int64_t N = a.N;
    const Cfg<T>& cfg = a.cfg;
    constexpr bool MIXED = !std::is_same<T, S>::value;
    T y[13];
    S ys[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) { ys[k] = ld(a.ws, ROW_STATE + k, N, e); y[k] = (T)ys[k]; }
    const bool dyn_only = a.mode == QENV_STEP_DYNAMICS_ONLY;
    Book<T> bk;
    if (!dyn_only) {
        load_book(a.ws, a.wsi, N, e, true, bk);
#pragma unroll
        for (int k = 0; k < 3; ++k) bk.ang_vel[k] = (T)ld(a.ws, ROW_ANG_VEL + k, N, e);
    }
    PidState<T> ps;
    if (CTRL == QENV_CTRL_PID || CTRL == QENV_CTRL_PIDPOS) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { ps.ix[k] = (T)ld(a.ws, ROW_PID + k, N, e); ps.x_old[k] = (T)ld(a.ws, ROW_PID + 6 + k, N, e); }
#pragma unroll
        for (int k = 0; k < 3; ++k) ps.ang_d_ant[k] = (T)ld(a.ws, ROW_PID + 12 + k, N, e);
    }
    T w[4], effort[4], cl[4], act[4], kq[13], raw[4], dy[13];
    Act<T> u;
    Plant<T, ROBUST, LEAN> pl;
    set_plant(pl, cfg);
    if constexpr (!ROBUST) pl.skip_probe = CTRL == QENV_CTRL_CONST;   // (see Plant: pays in the constant-action rollouts)
    if constexpr (ROBUST) load_plant<T>(a.ws, a.wsi, N, e, a.gust, pl);  // the host refuses dynamics-only robust rollouts
    if (CTRL == QENV_CTRL_CONST) {  // constant action: mix once for the whole rollout
        prepare_action(a.const_action, cfg, pl, w, effort, cl, act, u);
    }
    bool ang_ok = a.ang_ok != 0;
    T q_old[4];
    // start of env step s: the controller's action from the state in registers
    auto begin_step = [&](int s) {
        if (CTRL == QENV_CTRL_LQR) {
            lqr_law<T, M>(y, bk, a.lqr, raw);
            prepare_action(raw, cfg, pl, w, effort, cl, act, u);
        } else if (CTRL == QENV_CTRL_PID) {
            if (s == 0 && a.pid_first) {
                raw[0] = T(phys::M * phys::G); raw[1] = raw[2] = raw[3] = T(0);
            } else {
                pid_law<T, M>(y, bk, a.pid, ps, a.pid_target, cfg.t_step, raw);
            }
            prepare_action(raw, cfg, pl, w, effort, cl, act, u);
        } else if (CTRL == QENV_CTRL_PIDPOS) {
            if (a.target_table) {
                T tg[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) tg[k] = (T)a.target_table[s * 7 + k];
                pidpos_law<T, M>(y, bk, ps, tg, cfg.t_step, raw);
            } else {
                pidpos_law<T, M>(y, bk, ps, a.pid_target, cfg.t_step, raw);
            }
            prepare_action(raw, cfg, pl, w, effort, cl, act, u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) q_old[k] = y[6 + k];
    };
    // end of env step s: y = state at t_step, kq = f(t_end, y_end), dy = the integrator's increments
    auto end_step = [&](int s) {
        if (MIXED) {
#pragma unroll
            for (int k = 0; k < 13; ++k) { ys[k] += (S)dy[k]; y[k] = (T)ys[k]; }
        }
        if (!dyn_only) {
            bk.i += 1;
            euler_update<T, M>(q_old, y, dy, ang_ok && (bk.flags & FLAG_ANG_OK), cfg, bk);
            ROLLOUT_PHASE(3, s == a.n_steps / 2);
            done_reward<T, M>(y, act, effort, cfg, bk);
            ROLLOUT_PHASE(4, s == a.n_steps / 2);
            ang_ok = true;  // from here on every env's prev_ang belongs to the state in its registers
            if (a.win) {  // dl_in_gen.dl_input of this step (reset warm-up: raw action = the constant action)
                int slot = a.win_head + s;
                while (slot >= a.win_T) slot -= a.win_T;
                float* wv = a.win + (int64_t)slot * 15 * N + e;
#pragma unroll
                for (int k = 0; k < 4; ++k) wv[(int64_t)k * N] = (float)(CTRL == QENV_CTRL_CONST ? a.const_action[k] : raw[k]);
                wv[4 * N] = (float)y[1]; wv[5 * N] = (float)y[3]; wv[6 * N] = (float)y[5];
#pragma unroll
                for (int k = 0; k < 4; ++k) { wv[(int64_t)(7 + k) * N] = (float)y[6 + k]; wv[(int64_t)(11 + k) * N] = (float)kq[6 + k]; }
            }
        }
    };
    if constexpr (INTEG == INT_RK45 && !ROBUST && sizeof(T) == 4) {
        // Lane-asynchronous stepping (fp32 arithmetic only, see the end of this comment).  scipy's RK45 is restarted every env step and ~2 % of the (env, step) pairs of a
        // closed loop need a second sub-step (a velocity or body-rate component crossing zero shrinks its error scale to
        // atol, select_initial_step then starts small): with the wave advancing env step by env step, 64 lanes made
        // nearly every wave-step pay two attempts (census of the CPU restatement, config 3: 1.02 attempts per env-step, 1.92 per
        // wave-step).  Here every iteration of ONE loop is one attempted sub-step of whatever env step the lane is in;
        // a lane that finishes its step runs the step epilogue / next prologue on its own and the lanes drift apart by
        // the attempts they needed.  Each lane executes exactly the instruction sequence of integrate_rk45, so results
        // are bit-identical to the per-step kernel; the wave is done after max_lane(sum of attempts) iterations.
        // Measured (config 3, 262 144 envs): fp32 9.3 -> 10.1 G env-steps/s (config 2: 21.4 -> 24.6), but lean fp64
        // 6.0 -> 4.5: a few per cent of the envs need ~1.5 attempts per step PERSISTENTLY (same census: p99 of the
        // per-env mean 1.45), so the slowest lane's total is 1.3-1.55 attempts per step while every iteration now also
        // pays the divergent prologue / epilogue blocks -- no better than advancing in lockstep, and in fp64 worse
        // (register traffic of the larger loop-carried state).  fp64 arithmetic keeps the step-synchronous loop.
        int s = 0, it = 0;
        bool in_step = false, rejected = false;
        T t = T(0), h_abs = T(0);
        while (s < a.n_steps) {
            if (!in_step) {
                begin_step(s);
                // constant actuation: f(y_end) of the previous step IS f0 of this one.  Mixed handle: the step starts
                // from the fp64 row re-rounded to fp32, within one fp32 ulp of the y_end that f was evaluated at -- the
                // size of every other rounding of its fp32 arithmetic; the fused rollout of a mixed handle is therefore
                // close to, not bit-equal with, its per-step path
                if (!(CTRL == QENV_CTRL_CONST && s > 0)) rhs(y, u, pl, kq);
#pragma unroll
                for (int k = 0; k < 13; ++k) dy[k] = T(0);
                h_abs = Plant<T, ROBUST, LEAN>::math::min(rk45_initial_step(y, kq, u, pl, cfg.t_step), cfg.t_step);
                t = T(0); rejected = false; it = 0; in_step = true;
            }
            const bool over = rk45_attempt(y, kq, dy, u, pl, cfg.t_step, t, h_abs, rejected);
            ++it;
            if (over || it >= Rk45Cap<Plant<T, ROBUST, LEAN>>::value) {
                end_step(s);
                ++s;
                in_step = false;
            }
        }
    } else {
        for (int s = 0; s < a.n_steps; ++s) {
            if (s == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ROLLOUT_STAMP(2, false); }
            if (s == 1) ROLLOUT_STAMP(3, false);
            if (s == 2) ROLLOUT_STAMP(4, false);
            ROLLOUT_PHASE(0, s == a.n_steps / 2);
            begin_step(s);
            ROLLOUT_PHASE(1, s == a.n_steps / 2);
            // (a robust plant has no FSAL across steps: the wind changes from step to step)
            if constexpr (ROBUST) begin_step_wind(pl, bk.i + 1);
            integrate<T, INTEG>(y, u, pl, cfg.t_step, kq, dy, CTRL == QENV_CTRL_CONST && !ROBUST && s > 0);
            ROLLOUT_PHASE(2, s == a.n_steps / 2);
            end_step(s);
            ROLLOUT_PHASE(5, s == a.n_steps / 2);
        }
        ROLLOUT_STAMP(5, false);
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        if (MIXED) st(a.ws, ROW_STATE + k, N, e, ys[k]);
        else st(a.ws, ROW_STATE + k, N, e, y[k]);
    }
#if defined(QENV_ROLLOUT_STAMPS) && defined(QENV_ENV_UNIT_LEAN)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ROLLOUT_STAMP(6, false); ROLLOUT_STAMP(7, true);
#endif
    if (dyn_only) return;
    if constexpr (ROBUST) store_gust(a.ws, a.wsi, N, e, pl);
    store_book(a.ws, a.wsi, N, e, bk, true);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        st(a.ws, ROW_VQ + k, N, e, kq[6 + k]);
        st(a.ws, ROW_W + k, N, e, w[k]);
        st(a.ws, ROW_EFFORT + k, N, e, effort[k]);
        st(a.ws, ROW_CLIPPED + k, N, e, cl[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) st(a.ws, ROW_ACCEL + k, N, e, kq[1 + 2 * k]);
    if (CTRL == QENV_CTRL_PID || CTRL == QENV_CTRL_PIDPOS) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { st(a.ws, ROW_PID + k, N, e, ps.ix[k]); st(a.ws, ROW_PID + 6 + k, N, e, ps.x_old[k]); }
#pragma unroll
        for (int k = 0; k < 3; ++k) st(a.ws, ROW_PID + 12 + k, N, e, ps.ang_d_ant[k]);
    }
}

// ---- two envs per lane (fp32, even N): the integrator runs on packed float2 math -----------------
// Lane g owns envs 2g and 2g+1 (adjacent in every row: one 8-byte load/store per row per lane, 512 B
// per wave).  Mixing, Euler angles, done/reward and the controllers are evaluated per env with the
// scalar code; only the integrator -- >90% of the instructions -- is packed.
template <int INTEG, int CTRL>
__global__ __launch_bounds__(256, QENV_MIN_WAVES) void k_rollout2(const RolloutArgs<float> a) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t N = a.N;
    if (2 * g >= N) return;
    const Cfg<float>& cfg = a.cfg;
    const float2* ws2 = reinterpret_cast<const float2*>(a.ws);
    const int64_t N2 = N / 2;
    F2 y[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) { const float2 v = ws2[(int64_t)(ROW_STATE + k) * N2 + g]; y[k] = F2{v.x, v.y}; }
    const bool dyn_only = a.mode == QENV_STEP_DYNAMICS_ONLY;
    Book<float> bk[2];
    PidState<float> ps[2];
    if (!dyn_only) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            load_book(a.ws, a.wsi, N, 2 * g + c, true, bk[c]);
#pragma unroll
            for (int k = 0; k < 3; ++k) bk[c].ang_vel[k] = ld(a.ws, ROW_ANG_VEL + k, N, 2 * g + c);
            if (CTRL == QENV_CTRL_PID || CTRL == QENV_CTRL_PIDPOS) {
#pragma unroll
                for (int k = 0; k < 6; ++k) { ps[c].ix[k] = ld(a.ws, ROW_PID + k, N, 2 * g + c); ps[c].x_old[k] = ld(a.ws, ROW_PID + 6 + k, N, 2 * g + c); }
#pragma unroll
                for (int k = 0; k < 3; ++k) ps[c].ang_d_ant[k] = ld(a.ws, ROW_PID + 12 + k, N, 2 * g + c);
            }
        }
    }
    Plant<F2, false> pl2;
    pl2.cmj_xy = F2{cfg.cmj_xy, cfg.cmj_xy}; pl2.cmj_z = F2{cfg.cmj_z, cfg.cmj_z};
    Plant<float, false> pl1;
    set_plant(pl1, cfg);
    float w[2][4], effort[2][4], cl[2][4], act[2][4], raw[4];
    Act<float> us[2];
    Act<F2> u;
    if (CTRL == QENV_CTRL_CONST) {
        prepare_action(a.const_action, cfg, pl1, w[0], effort[0], cl[0], act[0], us[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { w[1][k] = w[0][k]; effort[1][k] = effort[0][k]; cl[1][k] = cl[0][k]; act[1][k] = act[0][k]; }
        pack_act(us[0], us[0], u);
    }
    bool ang_ok = a.ang_ok != 0;
    F2 kq[13], dy[13];
    for (int s = 0; s < a.n_steps; ++s) {
        if (CTRL != QENV_CTRL_CONST) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float yc[13];
#pragma unroll
                for (int k = 0; k < 13; ++k) yc[k] = y[k][c];
                if (CTRL == QENV_CTRL_LQR) lqr_law(yc, bk[c], a.lqr, raw);
                else if (CTRL == QENV_CTRL_PID) {
                    if (s == 0 && a.pid_first) { raw[0] = float(phys::M * phys::G); raw[1] = raw[2] = raw[3] = 0.f; }
                    else pid_law(yc, bk[c], a.pid, ps[c], a.pid_target, cfg.t_step, raw);
                } else pidpos_law(yc, bk[c], ps[c], a.target_table ? a.target_table + s * 7 : a.pid_target, cfg.t_step, raw);
                prepare_action(raw, cfg, pl1, w[c], effort[c], cl[c], act[c], us[c]);
            }
            pack_act(us[0], us[1], u);
        }
        const F2 q_old[4] = {y[6], y[7], y[8], y[9]};
        integrate2<INTEG>(y, u, pl2, cfg.t_step, kq, dy, CTRL == QENV_CTRL_CONST && s > 0);
        if (!dyn_only) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float yc[13], dyc[13];
#pragma unroll
                for (int k = 0; k < 13; ++k) { yc[k] = y[k][c]; dyc[k] = dy[k][c]; }
                const float qo[4] = {q_old[0][c], q_old[1][c], q_old[2][c], q_old[3][c]};
                bk[c].i += 1;
                euler_update(qo, yc, dyc, ang_ok && (bk[c].flags & FLAG_ANG_OK), cfg, bk[c]);
                done_reward(yc, act[c], effort[c], cfg, bk[c]);
            }
            ang_ok = true;
        }
    }
    float2* wso = reinterpret_cast<float2*>(a.ws);
#pragma unroll
    for (int k = 0; k < 13; ++k) wso[(int64_t)(ROW_STATE + k) * N2 + g] = make_float2(y[k].x, y[k].y);
    if (dyn_only) return;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int64_t e = 2 * g + c;
        store_book(a.ws, a.wsi, N, e, bk[c], true);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            st(a.ws, ROW_VQ + k, N, e, kq[6 + k][c]);
            st(a.ws, ROW_W + k, N, e, w[c][k]);
            st(a.ws, ROW_EFFORT + k, N, e, effort[c][k]);
            st(a.ws, ROW_CLIPPED + k, N, e, cl[c][k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) st(a.ws, ROW_ACCEL + k, N, e, kq[1 + 2 * k][c]);
        if (CTRL == QENV_CTRL_PID || CTRL == QENV_CTRL_PIDPOS) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { st(a.ws, ROW_PID + k, N, e, ps[c].ix[k]); st(a.ws, ROW_PID + 6 + k, N, e, ps[c].x_old[k]); }
#pragma unroll
            for (int k = 0; k < 3; ++k) st(a.ws, ROW_PID + 12 + k, N, e, ps[c].ang_d_ant[k]);
        }
    }
}

// quad.reset front half (quadrotor_env.py:428-445): load initial states, clear counters -- and, for a masked
// reset, everything else the reset needs, in the same launch: the initial states can be drawn here (Philox, the
// qenv_sample_init_states streams) instead of read, and the compacted list of reset envs that the warm-up rollout
// walks is built with one atomic per 1024-thread block (LDS prefix over the 16 wave ballots).  The list length lives
// in one of two ping-pong counters: this launch fills `count` and zeroes `count_next` for the next masked reset (the
// kernel that read it has finished: same stream), so no memset launch is needed.  In config 5 this chain was
// memset + k_sample_init + k_reset + k_compact_mask = 30 us of a 191 us step.
struct ResetSample { uint64_t seed, episode; int64_t env_offset; int32_t on; };
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_reset(T* ws, int32_t* wsi, int64_t N, const T* init, const uint8_t* mask,
                                                 ResetSample smp, int32_t* idx, int32_t* count, int32_t* count_next) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool on = e < N && (!mask || mask[e] != 0);
    if (idx) {  // compaction (uniform branch: every thread of the block takes part in the barriers)
        __shared__ int32_t wcnt[BLOCK / 64], wbase[BLOCK / 64];
        const unsigned long long b = __ballot(on);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wcnt[wv] = __popcll(b);
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < BLOCK / 64; ++w) { wbase[w] = tot; tot += wcnt[w]; }
            const int base = tot ? atomicAdd(count, tot) : 0;
            for (int w = 0; w < BLOCK / 64; ++w) wbase[w] += base;
            if (blockIdx.x == 0) *count_next = 0;
        }
        __syncthreads();
        if (on) idx[wbase[wv] + __popcll(b & ((1ull << lane) - 1ull))] = (int32_t)e;
    }
    if (!on) return;
    using m = Mth<T>;
    if (smp.on) {
        reset_env_sampled<T>(ws, wsi, N, e, smp.seed, smp.episode, smp.env_offset, mask != nullptr);
        return;
    }
    if (mask) {  // an episode ends here: bank its return before the counters are cleared
        st(ws, ROW_LAST_RETURN, N, e, ld(ws, ROW_EP_RETURN, N, e));
        wsi[IROW_EPISODES * N + e] += 1;
    }
    T x[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) x[k] = init[(int64_t)k * N + e];
#pragma unroll
    for (int k = 0; k < 13; ++k) st(ws, ROW_STATE + k, N, e, x[k]);
    const T* q = x + 6;
    // self.ang = quat_euler(q) (:438); only the opt variant reads it back (prev_ang = ang)
    st(ws, ROW_ANG + 0, N, e, m::atan2(T(2) * (q[0] * q[1] + q[2] * q[3]), T(1) - T(2) * (q[1] * q[1] + q[2] * q[2])));
    st(ws, ROW_ANG + 1, N, e, m::asin(T(2) * (q[0] * q[2] - q[3] * q[1])));
    st(ws, ROW_ANG + 2, N, e, m::atan2(T(2) * (q[0] * q[3] + q[1] * q[2]), T(1) - T(2) * (q[2] * q[2] + q[3] * q[3])));
    st(ws, ROW_PREV_SHAPING, N, e, T(0));
    st(ws, ROW_ABS_SUM, N, e, T(0));
    st(ws, ROW_EP_RETURN, N, e, T(0));
    st(ws, ROW_REWARD, N, e, T(0));
    wsi[IROW_I * N + e] = 0;
    wsi[IROW_FLAGS * N + e] = 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_fill_actions(T* out, int64_t N, T a0, T a1, T a2, T a3, const uint8_t* mask) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    (void)mask;
    out[e] = a0; out[N + e] = a1; out[2 * N + e] = a2; out[3 * N + e] = a3;
}

template <typename T>
__global__ __launch_bounds__(256) void k_copy_rows(T* dst, const T* src, int rows, int64_t N) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    for (int r = 0; r < rows; ++r) dst[(int64_t)r * N + e] = src[(int64_t)r * N + e];
}

template <typename T, typename M = Mth<T>>
__global__ __launch_bounds__(256) void k_lqr_action(const T* ws, int64_t N, LqrGains<T> g, T* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    T y[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) y[k] = T(0);
    y[1] = ld(ws, ROW_STATE + 1, N, e); y[3] = ld(ws, ROW_STATE + 3, N, e); y[5] = ld(ws, ROW_STATE + 5, N, e);
    Book<T> bk;
#pragma unroll
    for (int k = 0; k < 3; ++k) { bk.ang[k] = ld(ws, ROW_ANG + k, N, e); bk.ang_vel[k] = ld(ws, ROW_ANG_VEL + k, N, e); }
    T act[4];
    lqr_law<T, M>(y, bk, g, act);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[(int64_t)k * N + e] = act[k];
}

// quad.done_condition() + quad.reward_function() (quadrotor_env.py:500-573, quadrotor_env_opt.py:410-480) on attributes the
// CALLER sets -- state, ang, action, prev_shaping (NaN = None), i, done -- exactly as the reference's methods can be called
// on a hand-set object.  The SAME device function the step kernels run (done_reward); n is the caller's, not the handle's.
template <typename T, typename S, typename M>
__global__ __launch_bounds__(256) void k_reward_function(int64_t n, Cfg<T> cfg, const S* state, const S* ang, const S* action,
                                                         const S* prev_shaping, const int32_t* i_in, const uint8_t* done_in,
                                                         S* out_reward, uint8_t* out_done, int32_t* out_solved, S* out_shaping) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    T y[13], act[4];
#pragma unroll
    for (int k = 0; k < 13; ++k) y[k] = (T)state[(int64_t)k * n + e];
#pragma unroll
    for (int k = 0; k < 4; ++k) act[k] = (T)action[(int64_t)k * n + e];
    Book<T> bk;
#pragma unroll
    for (int k = 0; k < 3; ++k) { bk.ang[k] = (T)ang[(int64_t)k * n + e]; bk.prev_ang[k] = bk.ang[k]; bk.ang_vel[k] = T(0); }
    const S ps = prev_shaping[e];
    const bool has_prev = ps == ps;  // NaN: prev_shaping is None
    bk.prev_shaping = has_prev ? (T)ps : T(0);
    bk.abs_sum = T(0); bk.ep_return = T(0); bk.reward = T(0);
    bk.i = i_in[e];
    bk.flags = (done_in[e] ? FLAG_DONE : 0) | (has_prev ? FLAG_HAS_SHAPING : 0);
    done_reward<T, M>(y, act, act, cfg, bk);
    out_reward[e] = (S)bk.reward;
    out_done[e] = (bk.flags & FLAG_DONE) ? 1 : 0;
    out_solved[e] = (bk.flags & FLAG_SOLVED) ? 1 : 0;
    out_shaping[e] = (S)bk.prev_shaping;
}

// quad.f2F (direct handles, quadrotor_env.py:247-272) / quad.f2w (indirect, :197-245) through prepare_action, the function
// every step kernel starts with: rotor speeds w, the thrust / moments the RHS integrates, step_effort.
template <typename T, typename S, bool LEAN>
__global__ __launch_bounds__(256) void k_mix(int64_t n, Cfg<T> cfg, const S* action, S* out_w, S* out_fm, S* out_effort) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    T raw[4], w[4], effort[4], cl[4], act[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[k] = (T)action[(int64_t)k * n + e];
    Act<T> u;
    Plant<T, false, LEAN> pl;
    set_plant(pl, cfg);
    prepare_action(raw, cfg, pl, w, effort, cl, act, u);
    const T fm[4] = {u.F, u.Mx, u.My, u.Mz};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out_w[(int64_t)k * n + e] = (S)w[k];
        out_fm[(int64_t)k * n + e] = (S)fm[k];
        out_effort[(int64_t)k * n + e] = (S)effort[k];
    }
}

template <typename T, typename M = Mth<T>>
__global__ __launch_bounds__(256) void k_pid_action(T* ws, int64_t N, PidGains<T> g, T t0, T t1, T t2, T t3, T t_step,
                                                    T* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    T y[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) y[k] = T(0);
    y[1] = ld(ws, ROW_STATE + 1, N, e); y[3] = ld(ws, ROW_STATE + 3, N, e); y[5] = ld(ws, ROW_STATE + 5, N, e);
    Book<T> bk;
#pragma unroll
    for (int k = 0; k < 3; ++k) { bk.ang[k] = ld(ws, ROW_ANG + k, N, e); bk.ang_vel[k] = T(0); }
    PidState<T> ps;
#pragma unroll
    for (int k = 0; k < 6; ++k) { ps.ix[k] = ld(ws, ROW_PID + k, N, e); ps.x_old[k] = ld(ws, ROW_PID + 6 + k, N, e); }
#pragma unroll
    for (int k = 0; k < 3; ++k) ps.ang_d_ant[k] = ld(ws, ROW_PID + 12 + k, N, e);
    const T tgt[4] = {t0, t1, t2, t3};
    T act[4];
    pid_law<T, M>(y, bk, g, ps, tgt, t_step, act);
#pragma unroll
    for (int k = 0; k < 6; ++k) { st(ws, ROW_PID + k, N, e, ps.ix[k]); st(ws, ROW_PID + 6 + k, N, e, ps.x_old[k]); }
#pragma unroll
    for (int k = 0; k < 3; ++k) st(ws, ROW_PID + 12 + k, N, e, ps.ang_d_ant[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[(int64_t)k * N + e] = act[k];
}

template <typename T> struct Tgt7 { T v[7]; };

template <typename T, typename M = Mth<T>>
__global__ __launch_bounds__(256) void k_pidpos_action(T* ws, int64_t N, Tgt7<T> tg, T t_step, T* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    T y[13];
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = ld(ws, ROW_STATE + k, N, e);
#pragma unroll
    for (int k = 6; k < 13; ++k) y[k] = T(0);
    Book<T> bk;
#pragma unroll
    for (int k = 0; k < 3; ++k) { bk.ang[k] = ld(ws, ROW_ANG + k, N, e); bk.ang_vel[k] = ld(ws, ROW_ANG_VEL + k, N, e); }
    PidState<T> ps;
#pragma unroll
    for (int k = 0; k < 6; ++k) { ps.ix[k] = ld(ws, ROW_PID + k, N, e); ps.x_old[k] = T(0); }
#pragma unroll
    for (int k = 0; k < 3; ++k) ps.ang_d_ant[k] = ld(ws, ROW_PID + 12 + k, N, e);
    T act[4];
    pidpos_law<T, M>(y, bk, ps, tg.v, t_step, act);
#pragma unroll
    for (int k = 0; k < 6; ++k) st(ws, ROW_PID + k, N, e, ps.ix[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) st(ws, ROW_PID + 12 + k, N, e, ps.ang_d_ant[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[(int64_t)k * N + e] = act[k];
}

template <typename T>
__global__ __launch_bounds__(256) void k_pid_reset(T* ws, int64_t N, const uint8_t* mask) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    if (mask && mask[e] == 0) return;
#pragma unroll
    for (int k = 0; k < 15; ++k) st(ws, ROW_PID + k, N, e, T(0));
}

template <typename S>
__global__ __launch_bounds__(256) void k_outer_reset(S* ws, int64_t N, const uint8_t* mask) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    if (mask && mask[e] == 0) return;
#pragma unroll
    for (int k = 0; k < 9; ++k) st(ws, ROW_OUTER + k, N, e, S(0));
}

// robust_control.reset (quadrotor_env.py:97-101): the 12 episode draws, injected or Philox
template <typename S>
__global__ __launch_bounds__(256) void k_robust_params(S* ws, const int32_t* wsi, int64_t N, const S* params, uint64_t seed,
                                                       int64_t env_offset, const uint8_t* mask) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    if (mask && mask[e] == 0) return;
    if (params) {
#pragma unroll
        for (int k = 0; k < 12; ++k) st(ws, ROW_ROB + k, N, e, params[(int64_t)k * N + e]);
        return;
    }
    const uint64_t ge = (uint64_t)(env_offset + e);
    float v[12];
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        // a masked call precedes the masked qenv_reset that ends the episode and bumps the counter: key the NEXT episode
        uint32_t c[4] = {(uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)(wsi[IROW_EPISODES * N + e] + (mask ? 1 : 0)), (uint32_t)blk};
        philox_core(c, (uint32_t)seed ^ 0x0B0Bu, (uint32_t)(seed >> 32) ^ 0xE915u);
        if (blk == 0) {         // episode_kf = random(4) * D_KF
            for (int k = 0; k < 4; ++k) v[k] = 0.1f * philox_u01(c[k]);
        } else if (blk == 1) {  // episode_ir = random(4) * D_IR
            for (int k = 0; k < 4; ++k) v[5 + k] = 0.1f * philox_u01(c[k]);
        } else {                // episode_m ~ N(0, D_M), episode_J ~ N(0, D_J)
            float z0, z1, z2, z3;
            philox_normal2(c[0], c[1], z0, z1);
            philox_normal2(c[2], c[3], z2, z3);
            v[4] = 0.3f * z0; v[9] = 0.1f * z1; v[10] = 0.1f * z2; v[11] = 0.1f * z3;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) st(ws, ROW_ROB + k, N, e, v[k]);
}

// ---- host-side helpers -------------------------------------------------------------------------
template <typename T> static Cfg<T> make_cfg(const qenv_s* q) {
    Cfg<T> c;
    const qenv_config& f = q->cfg;
    const bool opt = f.variant == QENV_OPT;
    c.t_step = (T)f.t_step;
    c.cm_xy = (T)(phys::CM0 * (opt ? phys::SUM_XX2 : phys::SUM_XX3));
    c.cm_z = (T)(2.0 * phys::CM0 * (opt ? phys::SUM_XX2 : phys::SUM_XX3));
    c.cmj_xy = (T)(phys::CM0 * (opt ? phys::SUM_XX2 : phys::SUM_XX3) / phys::JX);   // (J_xx = J_yy)
    c.cmj_z = (T)(2.0 * phys::CM0 * (opt ? phys::SUM_XX2 : phys::SUM_XX3) / phys::JZ);
    double bb[9];
    if (opt) {
        const double b[9] = {10, 10, 10, phys::BB_ANG, phys::BB_ANG, 4, 30, 30, 30};
        for (int k = 0; k < 9; ++k) bb[k] = f.training ? b[k] : b[k] * 1000;
    } else {
        const double b[9] = {10, 10, 10, phys::BB_ANG, phys::BB_ANG, 0.75 * phys::PI, 20, 20, 20};
        for (int k = 0; k < 9; ++k) bb[k] = b[k];
    }
    for (int k = 0; k < 9; ++k) c.bb[k] = (T)bb[k];
    c.zero_control0 = f.direct_control ? (T)(2.0 / phys::T2WR - 1.0) : (T)(phys::M * phys::G);
    c.n_max = f.n_max + f.T;
    c.warmup = f.T;
    c.training = f.training ? 1 : 0;
    c.direct = f.direct_control ? 1 : 0;
    c.clipped = f.clipped ? 1 : 0;
    c.variant = opt ? 1 : 0;
    return c;
}

// CARE solution for the decoupled double integrators of lqr_quad.py:25-111 (App. A.8)
template <typename T> static LqrGains<T> make_lqr(bool clipped) {
    double q_att[6], q_t[6], r_att = 40, r_t;
    if (clipped) {
        const double qa[6] = {5, 1, 5, 1, 0.05, 0.01}, qt[6] = {1e-8, 1, 1e-8, 1, 1e-8, 0.8};
        for (int k = 0; k < 6; ++k) { q_att[k] = qa[k] * 50; q_t[k] = qt[k] * 10; }
        r_t = 10;
    } else {
        const double qa[6] = {5, 0.3, 5, 0.3, 2, 0.3}, qt[6] = {1e-8, 1, 1e-8, 1, 1e-8, 0.5};
        for (int k = 0; k < 6; ++k) { q_att[k] = qa[k] * 160; q_t[k] = qt[k] * 60; }
        r_t = 5;
    }
    const double b_att[3] = {1 / phys::JX, 1 / phys::JY, 1 / phys::JZ};
    LqrGains<T> g;
    for (int a = 0; a < 3; ++a) {
        double k1 = std::sqrt(q_att[2 * a] / r_att);
        double k2 = std::sqrt(q_att[2 * a + 1] / r_att + 2 * k1 / b_att[a]);
        g.ka[a][0] = (T)(-k1); g.ka[a][1] = (T)(-k2);
        k1 = std::sqrt(q_t[2 * a] / r_t);
        k2 = std::sqrt(q_t[2 * a + 1] / r_t + 2 * k1 * phys::M);
        g.kt[a] = (T)(-k2);  // positions are zeroed in state_t (lqr_quad.py:129), only the velocity gain acts
    }
    return g;
}

// pid_vel_control.py:17-27
template <typename T> static PidGains<T> make_pid(bool clipped) {
    PidGains<T> g;
    const double P = clipped ? 1 : 2, Pz = clipped ? 0.4 : 1, Pa = clipped ? 20 : 180, Da = clipped ? 20 : 50,
                 Pp = clipped ? 5 : 40, Dp = clipped ? 5 : 20;
    const double p[6] = {P, P, Pz, Pa, Pa, Pp}, d[6] = {0, 0, 0, Da, Da, Dp};
    for (int k = 0; k < 6; ++k) { g.p[k] = (T)p[k]; g.i[k] = (T)0; g.d[k] = (T)d[k]; }
    g.ts = (T)0.01;
    return g;
}

static inline dim3 grid_for(int64_t N) { return dim3((unsigned)((N + 255) / 256)); }
// Workgroup size of k_step / k_rollout (one lane per env, nothing shared inside a workgroup): QENV_ENV_BLOCK = 64, 128 or 256.
// Measured (lean handle): config 3 the same at all three (6.06 / 6.06 / 6.08 G env-steps/s: waves of a workgroup finishing
// at different times do not cost SIMD time), config 4 1.604 / 1.624 / 1.632, config 5 0.505 / 0.415 / 0.462 -- 256 stays.
static inline int env_block() {
    static const int b = [] {
        const int v = getenv("QENV_ENV_BLOCK") ? atoi(getenv("QENV_ENV_BLOCK")) : QENV_ENV_BLOCK_DEFAULT;
        return (v == 64 || v == 128 || v == 256) ? v : QENV_ENV_BLOCK_DEFAULT;
    }();
    return b;
}
static inline dim3 env_grid(int64_t N) { return dim3((unsigned)((N + env_block() - 1) / env_block())); }
// k_step's grid: no more workgroups than the chip holds at once (for the scipy-exact integrator one wave per SIMD: 1 024
// waves = 65 536 lanes); beyond that a wave walks several tiles and requests the next one's rows while it integrates the
// current one (step_load / step_one).  QENV_STEP_GRID_LANES overrides the resident-lane count (A/B runs).
template <typename F> static inline dim3 step_grid(F fn, int64_t N) {
    struct Slot { const void* fn; int dev; int64_t lanes; };
    static thread_local Slot cache[32] = {};
    static thread_local int n = 0;
    const int block = env_block();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    int64_t cap = 0;
    for (int i = 0; i < n && !cap; ++i)
        if (cache[i].fn == (const void*)fn && cache[i].dev == dev) cap = cache[i].lanes;
    if (!cap) {
        if (const char* s = getenv("QENV_STEP_GRID_LANES")) cap = atoll(s);
        if (cap <= 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
            if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
            cap = (int64_t)per_cu * block * cus;
        }
        if (dev >= 0 && n < 32) cache[n++] = Slot{(const void*)fn, dev, cap};
    }
    const int64_t lanes = N < cap ? N : cap;
    return dim3((unsigned)((lanes + block - 1) / block));
}

// extras of the overlapped PPO step (see StepArgs)
struct StepExtra {
    const uint8_t* env_skip;
    const int32_t* list_idx; const int32_t* list_count;  // step only the envs of this compacted list
    int32_t ar_on; uint64_t ar_seed, ar_episode; int64_t ar_env_offset;
    int32_t* ar_idx; int32_t* ar_count; uint8_t* ar_flags_next;
    int32_t ar_key_episode; void* out_valid;
};

// the step kernel's argument block for one qenv_step-like call (and the host-side bookkeeping that goes with it: the sensor's
// call counter, the prev_ang flag)
template <typename T, typename S> StepArgs<T, S> make_step_args(qenv_s* q, const void* actions, const uint8_t* mask, void* out_obs,
                                                                 void* out_reward, uint8_t* out_done, void* out_mask, int flags,
                                                                 int win_slot, const StepExtra* ex) {
    StepArgs<T, S> a;
    a.cfg = make_cfg<T>(q);
    a.ws = (S*)q->ws; a.wsi = q->wsi; a.N = q->N;
    a.actions = (const S*)actions; a.mask = mask;
    a.idx = nullptr; a.count = nullptr;
    if (mask && q->compact_valid) { a.idx = q->wsi + IROW_IDX * q->N; a.count = q->counters + q->count_used; }
    a.out_obs = (S*)out_obs; a.out_reward = (S*)out_reward; a.out_done = out_done; a.mode = flags;
    a.out_mask = (S*)out_mask;
    a.win = win_slot >= 0 ? q->window : nullptr;
    a.win_slot = win_slot;
    a.outer_kind = q->outer.kind;
    a.outer_rows = q->outer.rows;
    a.outer_target = (const S*)q->outer.target;
    for (int k = 0; k < 3; ++k) a.outer_gain[k] = (T)q->outer.gain[k];
    a.ang_ok = q->ang_ok;
    a.sens = nullptr; a.sens_z = nullptr; a.sens_out = nullptr; a.sens_seed = a.sens_step = 0; a.sens_env_offset = 0;
    for (int k = 0; k < 5; ++k) a.sens_p.std[k] = S(0);
    for (int k = 0; k < 3; ++k) a.sens_p.bd[k] = S(0);
    if (q->sens.ws && win_slot >= 0 && q->outer.kind != QENV_OUTER_POSPI && !(mask || ex)) {
        // (reset warm-up pushes and the landing baseline's pos_pi loop see the true state, as in the reference)
        static const double dflt[8] = {0.1, 0.035, 15, 1.71, 0.5, 0.0005, 0.00015, 0.075};
        const double* v = q->sens.has_params ? q->sens.params : dflt;
        const double er = q->sens.error ? 1.0 : 0.0;
        for (int k = 0; k < 5; ++k) a.sens_p.std[k] = (S)(v[k] * er);
        for (int k = 0; k < 3; ++k) a.sens_p.bd[k] = (S)(v[5 + k] * er);
        a.sens = (S*)q->sens.ws; a.sens_z = (const S*)q->sens.z27; a.sens_out = (S*)q->sens.out;
        a.sens_seed = q->sens.seed; a.sens_step = q->sens.calls; a.sens_env_offset = q->sens.env_offset;
        q->sens.calls += 1;
    }
    a.env_skip = nullptr; a.ar_on = 0; a.ar_seed = a.ar_episode = 0; a.ar_env_offset = 0;
    a.ar_idx = nullptr; a.ar_count = nullptr; a.ar_flags_next = nullptr; a.ar_key_episode = 0; a.out_valid = nullptr;
    if (ex) {
        a.ar_key_episode = ex->ar_key_episode; a.out_valid = (S*)ex->out_valid;
        a.env_skip = ex->env_skip; a.ar_on = ex->ar_on; a.ar_seed = ex->ar_seed;
        if (ex->list_idx) { a.idx = ex->list_idx; a.count = ex->list_count; }
        a.ar_episode = ex->ar_episode; a.ar_env_offset = ex->ar_env_offset; a.ar_idx = ex->ar_idx; a.ar_count = ex->ar_count;
        a.ar_flags_next = ex->ar_flags_next;
    }
    // host-side half of the prev_ang bookkeeping: a dynamics-only step moves every state without
    // touching the angle rows; an unmasked full step re-synchronises all of them
    if (flags == QENV_STEP_DYNAMICS_ONLY) q->ang_ok = 0;
    else if (!mask) q->ang_ok = 1;
    a.gust = q->gust;
    return a;
}

template <typename T, typename S, bool LEAN = false> int do_step(qenv_s* q, const void* actions, const uint8_t* mask, void* out_obs,
                                                     void* out_reward, uint8_t* out_done, void* out_mask, int flags_in,
                                                     int win_slot, const StepExtra* ex = nullptr) {
    const int flags = flags_in & QENV_STEP_MODE_MASK;
    const StepArgs<T, S> a = make_step_args<T, S>(q, actions, mask, out_obs, out_reward, out_done, out_mask, flags, win_slot, ex);
    const dim3 g = env_grid(q->N), b(env_block());
    if (q->robust) {
        if (flags == QENV_STEP_DYNAMICS_ONLY)
            return fail(QENV_ERR_ARG, "robust_control: the gusts follow the step counter, dynamics-only steps are not available");
        if constexpr (LEAN) {
            return fail(QENV_ERR_ARG, "robust_control needs compute = NATIVE or F32 (the lean fp64 kernels have no robust variant)");
        } else {
            switch (q->cfg.integrator) {
                case QENV_RK45: hipLaunchKernelGGL((k_step<T, S, INT_RK45, true>), g, b, 0, q->stream, a); break;
                case QENV_DP5: hipLaunchKernelGGL((k_step<T, S, INT_DP5, true>), g, b, 0, q->stream, a); break;
                default: hipLaunchKernelGGL((k_step<T, S, INT_RK4, true>), g, b, 0, q->stream, a); break;
            }
            return check_launch("k_step(robust)");
        }
    }
    switch (q->cfg.integrator) {
        case QENV_RK45: {
            // the tile loop (a wave requests tile t + 1's rows before it integrates tile t) costs ~180 more registers than the
            // ~50 the rows need (A/B, tests/tools/time_step_one.py: 131 072 envs 39.8 vs 38.7 us without, 262 144 65.7 vs 65.5,
            // 524 288 121.7..125.7 vs 128.5; fp32 rows 33.1 vs 26.8: the smaller kernel keeps 2 waves per SIMD): from eight
            // tiles per wave on fp64 rows (+2..5 %), never on fp32 rows.  QENV_STEP_TILES=0/1 overrides (A/B).
            bool tiles = false;
            if constexpr (sizeof(S) == 8) {
                auto fn = k_step<T, S, INT_RK45, false, LEAN, true>;
                static const int force = getenv("QENV_STEP_TILES") ? atoi(getenv("QENV_STEP_TILES")) : -1;
                const dim3 gt = step_grid(fn, q->N);
                const int64_t resident = (int64_t)gt.x * b.x;
                tiles = force < 0 ? q->N >= 8 * resident : (force > 0 && q->N > resident);
                if (tiles) hipLaunchKernelGGL(fn, gt, b, 0, q->stream, a);
            }
            if (!tiles) hipLaunchKernelGGL((k_step<T, S, INT_RK45, false, LEAN>), g, b, 0, q->stream, a);
            break;
        }
        case QENV_DP5: hipLaunchKernelGGL((k_step<T, S, INT_DP5, false, LEAN>), g, b, 0, q->stream, a); break;
        default: hipLaunchKernelGGL((k_step<T, S, INT_RK4, false, LEAN>), g, b, 0, q->stream, a); break;
    }
    return check_launch("k_step");
}

// qenv_ppo_rollout (in-band schedule) / qenv_policy_rollout as ONE launch of k_ppo_fused.  The caller (qenv_env.hip) has checked
// that the configuration is one the kernel covers (ppo_fused_ok).
template <typename T, typename S, bool LEAN = false> int do_ppo_fused(qenv_s* q, const qenv_ppo_step_args* p, int n_steps, int deterministic,
                                                                      const void* image) {
    constexpr size_t esz = sizeof(S);
    const size_t N = (size_t)q->N;
    StepExtra ex{};
    if (!deterministic) {
        ex.ar_on = 2; ex.ar_seed = p->seed; ex.ar_episode = p->step; ex.ar_env_offset = p->env_offset;
        ex.ar_key_episode = p->key_by_episode; ex.out_valid = p->out_valid;
    }
    FusedArgs<T, S> fa;
    fa.st = make_step_args<T, S>(q, p->out_action, nullptr, nullptr, p->out_reward, p->out_done, p->out_mask, QENV_STEP_NO_AUX, q->win_head,
                                 deterministic ? nullptr : &ex);
    // fused_env_step restates step_one for the plain in-band case only: whatever else a step can carry -- a mask or an env
    // list, an overlapped-step skip list, an observation output, a mode with auxiliary outputs, an outer loop, a sensor, a
    // robust plant -- is refused HERE, so a regression of the caller's gate (ppo_fused_ok) fails loudly instead of
    // computing a different env
    {
        const auto& st = fa.st;
        if (st.mask || st.idx || st.count || st.env_skip || st.out_obs || st.sens || st.outer_kind != QENV_OUTER_NONE || q->robust ||
            (st.mode & ~QENV_STEP_NO_AUX) != 0 || st.ar_idx || st.ar_count || st.ar_flags_next || !st.win || q->cfg.T != 5 ||
            q->cfg.integrator != QENV_RK45 || !q->cfg.direct_control)
            return fail(QENV_ERR_STATE, "k_ppo_fused: the step carries a feature the fused launch does not implement (mask / env list / "
                                        "skip list / out_obs / aux mode / outer loop / sensor / robust plant / list auto-reset); "
                                        "the separate launches handle it (qenv_set_ppo_fused(q, 0))");
    }
    MlpArgs& m = fa.ml;
    m = MlpArgs{};
    m.win = q->window; m.N = q->N; m.T = q->cfg.T; m.head = q->win_head; m.n_out = 4; m.final_tanh = 1;
    m.W1 = p->W1; m.b1 = p->b1; m.W2 = p->W2; m.b2 = p->b2; m.W3 = p->W3; m.b3 = p->b3;
    m.out = deterministic ? p->out_action : p->out_mean;
    m.out_f64 = q->cfg.dtype == QENV_F64;
    m.obs_store = p->obs_store; m.n_chunks = 1; m.obs_rows_from = p->obs_rows_from;
    m.noise = p->noise; m.seed = p->seed; m.step = p->step; m.env_offset = p->env_offset;
    m.std = p->std; m.log_norm = deterministic ? 0.0 : -std::log(p->std) - 0.9189385332046727;
    m.out_action = p->out_action; m.out_logprob = p->out_logprob;
    m.image = (const uint4*)image;
    fa.n_steps = n_steps;
    fa.obs_slices = p->obs_slices;
    fa.adv_act = (int64_t)(4 * N * esz); fa.adv_logp = (int64_t)(4 * N * esz); fa.adv_noise = (int64_t)(4 * N * esz);
    fa.adv_obs = (int64_t)((p->obs_slices ? (size_t)15 : (size_t)15 * q->cfg.T) * N * sizeof(float));
    fa.adv_row = (int64_t)(N * esz);
    const int cus = device_cu_count();
    // a batch that leaves half of the chip's waves without a 64-env block runs as 32-env blocks: one policy tile and an env step on
    // half the lanes per wave and step -- 36 k instead of 55 k cycles per step, on twice the waves (QENV_FUSED_BLOCK=5/6 forces)
    static const int force_shift = getenv("QENV_FUSED_BLOCK") ? atoi(getenv("QENV_FUSED_BLOCK")) : 0;
    fa.block_shift = (q->ppo_fused_block == 5 || q->ppo_fused_block == 6) ? q->ppo_fused_block   // qenv_set_ppo_fused_block: forced
                     : (force_shift == 5 || force_shift == 6) ? force_shift
                     : (((int64_t)N + 31) / 32 <= (int64_t)cus * 4 ? 5 : 6);
    const int64_t n_blocks = ((int64_t)N + (1 << fa.block_shift) - 1) >> fa.block_shift;
    int64_t blocks = (n_blocks + 3) / 4;
    if (blocks > cus) blocks = cus;   // one resident workgroup per CU (its LDS holds the weights); a wave walks its blocks
    const size_t lds = sizeof(uint4) * (size_t)mlp3::IMAGE_UINT4;
    hipError_t e;
    if (deterministic) {
        auto fn = k_ppo_fused<T, S, INT_RK45, LEAN, false>;
        if ((e = allow_dynamic_lds((const void*)fn, lds)) != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_ppo_fused)");
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), lds, q->stream, fa);
    } else {
        auto fn = k_ppo_fused<T, S, INT_RK45, LEAN, true>;
        if ((e = allow_dynamic_lds((const void*)fn, lds)) != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_ppo_fused)");
        hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), lds, q->stream, fa);
    }
    const int rc = check_launch("k_ppo_fused");
    if (rc == QENV_OK) q->win_head = (q->win_head + n_steps) % q->cfg.T;
    return rc;
}

// Workgroup size of a k_rollout launch.  A batch of several wave rounds (more envs than the 65 536 lanes the chip holds at one
// wave per SIMD) runs as ONE-wave workgroups: a finished wave's SIMD takes the next wave at once instead of waiting for the
// other three of its workgroup, which matters when a launch is short (config 3, 262 144 envs, 20-step launches: 8.02 -> 8.21 G
// env-steps/s; 250-step launches 8.62 -> 8.69).  QENV_ROLLOUT_BLOCK = 64 / 128 / 256 overrides.
static inline int rollout_block(int64_t N) {
    static const int forced = [] {
        const int v = getenv("QENV_ROLLOUT_BLOCK") ? atoi(getenv("QENV_ROLLOUT_BLOCK")) : 0;
        return (v == 64 || v == 128 || v == 256) ? v : 0;
    }();
    return forced ? forced : (N > 65536 ? 64 : env_block());
}

template <typename T, typename S, int INTEG, bool LEAN> static void launch_rollout(qenv_s* q, int controller, const RolloutArgs<T, S>& a) {
    const int blk = rollout_block(q->N);
    const dim3 g((unsigned)((q->N + blk - 1) / blk)), b(blk);
    if constexpr (!LEAN) if (q->robust) {
        switch (controller) {
            case QENV_CTRL_CONST: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_CONST, true>), g, b, 0, q->stream, a); break;
            case QENV_CTRL_LQR: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_LQR, true>), g, b, 0, q->stream, a); break;
            case QENV_CTRL_PIDPOS: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_PIDPOS, true>), g, b, 0, q->stream, a); break;
            default: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_PID, true>), g, b, 0, q->stream, a); break;
        }
        return;
    }
    if constexpr (sizeof(T) == 4 && sizeof(S) == 4) {
        // two envs per lane, packed fp32 integrator: bit-identical results; measured (tools/ab_packed.py) +26..37 %
        // for the fixed-step integrators under a constant action once there are >= 2 lane-pairs-worth of waves per
        // SIMD (N >= 131072), a loss below that, and no gain for rk45 or with a controller in the loop.
        // QENV_PACKED=0/1 overrides the rule.
        const bool auto_packed = q->packed < 0 && INTEG != INT_RK45 && controller == QENV_CTRL_CONST && q->N >= 131072;
        if ((q->packed > 0 || auto_packed) && (q->N % 2) == 0 && a.lane_shift == 6 && !a.idx) {  // the packed kernel has no list mode
            const dim3 g2 = grid_for(q->N / 2), b2(256);
            switch (controller) {
                case QENV_CTRL_CONST: hipLaunchKernelGGL((k_rollout2<INTEG, QENV_CTRL_CONST>), g2, b2, 0, q->stream, a); break;
                case QENV_CTRL_LQR: hipLaunchKernelGGL((k_rollout2<INTEG, QENV_CTRL_LQR>), g2, b2, 0, q->stream, a); break;
                case QENV_CTRL_PIDPOS: hipLaunchKernelGGL((k_rollout2<INTEG, QENV_CTRL_PIDPOS>), g2, b2, 0, q->stream, a); break;
                default: hipLaunchKernelGGL((k_rollout2<INTEG, QENV_CTRL_PID>), g2, b2, 0, q->stream, a); break;
            }
            return;
        }
    }
    if (a.lane_shift < 6) {
        const int64_t per_block = (int64_t)(blk / 64) << a.lane_shift;
        const dim3 gh((unsigned)((q->N + per_block - 1) / per_block));
        switch (controller) {
            case QENV_CTRL_CONST: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_CONST, false, LEAN>), gh, b, 0, q->stream, a); break;
            case QENV_CTRL_LQR: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_LQR, false, LEAN>), gh, b, 0, q->stream, a); break;
            case QENV_CTRL_PIDPOS: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_PIDPOS, false, LEAN>), gh, b, 0, q->stream, a); break;
            default: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_PID, false, LEAN>), gh, b, 0, q->stream, a); break;
        }
        return;
    }
    switch (controller) {
        case QENV_CTRL_CONST: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_CONST, false, LEAN>), g, b, 0, q->stream, a); break;
        case QENV_CTRL_LQR: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_LQR, false, LEAN>), g, b, 0, q->stream, a); break;
        case QENV_CTRL_PIDPOS: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_PIDPOS, false, LEAN>), g, b, 0, q->stream, a); break;
        default: hipLaunchKernelGGL((k_rollout<T, S, INTEG, QENV_CTRL_PID, false, LEAN>), g, b, 0, q->stream, a); break;
    }
}

template <typename T, typename S, bool LEAN = false> int do_rollout(qenv_s* q, int controller, int n_steps, const double* const_action,
                                                       int pid_first, int flags, const void* target_table = nullptr) {
    RolloutArgs<T, S> a;
    a.target_table = (const S*)target_table;
    a.idx = nullptr; a.count = nullptr; a.win = nullptr; a.win_head = 0; a.win_T = q->cfg.T; a.list_min_shift = -1;
    // QENV_HALF_WAVES=1 (experiment of DESIGN.md section 4g): 32 envs per wave everywhere -- 1.7x slower, the idle half is not skipped
    a.lane_shift = (getenv("QENV_HALF_WAVES") && atoi(getenv("QENV_HALF_WAVES")) && !q->robust) ? 5 : 6;
    if (q->compact_valid) {  // called from a masked qenv_reset: warm-up of the reset envs only
        a.idx = q->wsi + (q->compact_valid == 2 ? IROW_IDX2 : IROW_IDX) * q->N; a.count = q->counters + q->count_used;
        if (q->warmup_push) { a.win = q->window; a.win_head = q->win_head; }
        // The warm-up of a reset list is the latency of its slowest wavefront (a handful of waves on an otherwise idle
        // chip), and a wave runs every env step as many RK45 attempts as its slowest lane -- fresh initial states need
        // 2..6.  The kernel therefore spreads the list over the waves of the (unchanged) grid more thinly: 16 envs per wave
        // unless the list needs more (tools/time_masked_reset.py, lean handle, 65 536 envs, 300 finished: 66 us dense,
        // 58 us at 16 per wave; config 5 +9 %).  Measured limits: below 16 active lanes a wave gets SLOWER (4 or 1 env per
        // wave: 100 us -- the kernel itself, rocprofv3), and a larger grid of mostly empty waves serialises behind the
        // kernel's one-wave-per-SIMD register footprint (8 192 waves: +40 us).  QENV_WARMUP_LANES = minimum envs per
        // wave (64 = the dense mapping).
        if (!q->robust && a.lane_shift == 6) {
            static const int lanes = getenv("QENV_WARMUP_LANES") ? atoi(getenv("QENV_WARMUP_LANES")) : 16;
            int sh = 6;
            while (sh > 0 && (1 << sh) > lanes) --sh;
            a.list_min_shift = lanes >= 64 ? -1 : sh;
        }
    }
    a.cfg = make_cfg<T>(q);
    a.gust = q->gust;
    if (q->robust && flags == QENV_STEP_DYNAMICS_ONLY)
        return fail(QENV_ERR_ARG, "robust_control: the gusts follow the step counter, dynamics-only rollouts are not available");
    if (q->robust && LEAN)
        return fail(QENV_ERR_ARG, "robust_control needs compute = NATIVE or F32 (the lean fp64 kernels have no robust variant)");
    a.ws = (S*)q->ws; a.wsi = q->wsi; a.N = q->N; a.n_steps = n_steps; a.mode = flags; a.pid_first = pid_first;
    a.ang_ok = q->ang_ok;
    if (!a.idx) q->ang_ok = flags == QENV_STEP_DYNAMICS_ONLY ? 0 : 1;  // a masked warm-up leaves the global flag alone
    for (int k = 0; k < 7; ++k) a.pid_target[k] = (controller == QENV_CTRL_PIDPOS && const_action) ? (T)const_action[k] : (T)0;
    for (int k = 0; k < 4; ++k)
        a.const_action[k] = (const_action && controller == QENV_CTRL_CONST)
                                ? (T)const_action[k]
                                : ((q->cfg.direct_control || k == 0) ? a.cfg.zero_control0 : (T)0);
    a.lqr = make_lqr<T>(q->cfg.clipped != 0);
    a.pid = make_pid<T>(q->cfg.clipped != 0);
    switch (q->cfg.integrator) {
        case QENV_RK45: launch_rollout<T, S, INT_RK45, LEAN>(q, controller, a); break;
        case QENV_DP5: launch_rollout<T, S, INT_DP5, LEAN>(q, controller, a); break;
        default: launch_rollout<T, S, INT_RK4, LEAN>(q, controller, a); break;
    }
    return check_launch("k_rollout");
}

template <typename T, typename S, bool LEAN = false> int do_reset(qenv_s* q, const void* init, const uint8_t* mask, void* out_states,
                                                     void* out_actions, int push_window, const ResetSample* sample = nullptr) {
    const dim3 g = grid_for(q->N), b(256);
    ResetSample smp{0, 0, 0, 0};
    if (sample) smp = *sample;
    int rc;
    q->compact_valid = 0;
    if (mask) {  // the T warm-up steps touch only the reset envs: they run on the compacted index list built right here
        int32_t* counters = q->counters;  // [0], [1]: ping-pong list lengths (see k_reset)
        const int cur = q->count_cur;
        hipLaunchKernelGGL((k_reset<S, 1024>), dim3((unsigned)((q->N + 1023) / 1024)), dim3(1024), 0, q->stream, (S*)q->ws, q->wsi,
                           q->N, (const S*)init, mask, smp, q->wsi + IROW_IDX * q->N, counters + cur, counters + (cur ^ 1));
        rc = check_launch("k_reset(masked)");
        q->compact_valid = 1;
        q->count_used = cur;
        q->count_cur = cur ^ 1;
    } else {
        hipLaunchKernelGGL((k_reset<S, 256>), g, b, 0, q->stream, (S*)q->ws, q->wsi, q->N, (const S*)init, mask, smp,
                           (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
        rc = check_launch("k_reset");
    }
    const Cfg<S> c = make_cfg<S>(q);
    if (mask && rc == 0 && !out_states && !out_actions) {
        // masked reset inside a rollout: all T warm-up steps of the (few) reset envs in ONE launch -- their wall time
        // is the latency of a single wave's step, so T launches cost T times that; state stays in registers
        q->warmup_push = push_window;
        const int keep = q->packed;
        q->packed = 0;
        rc = do_rollout<T, S, LEAN>(q, QENV_CTRL_CONST, q->cfg.T, nullptr, 0, QENV_STEP_FULL);
        q->packed = keep;
        q->warmup_push = 0;
        q->compact_valid = 0;
        return rc;
    }
    const int outer_keep = q->outer.kind;
    q->outer.kind = QENV_OUTER_NONE;  // the warm-up slices are the plain states (position.py:205-206, math_trajectory.py:106)
    for (int t = 0; t < q->cfg.T && rc == 0; ++t) {
        S* obs = out_states ? (S*)out_states + (int64_t)t * 14 * q->N : nullptr;
        // warm-up slice t goes to ring slot head+t: after T of them the reset env's ring is full and in
        // phase with the (unmoved) head, whatever the other envs did meanwhile
        rc = do_step<T, S, LEAN>(q, nullptr, mask, obs, nullptr, nullptr, nullptr, QENV_STEP_FULL,
                        push_window ? (q->win_head + t) % q->cfg.T : -1);
        if (rc == 0 && out_actions) {
            S* oa = (S*)out_actions + (int64_t)t * 4 * q->N;
            const S z = c.zero_control0;
            hipLaunchKernelGGL((k_fill_actions<S>), g, b, 0, q->stream, oa, q->N, z, c.direct ? z : (S)0,
                               c.direct ? z : (S)0, c.direct ? z : (S)0, mask);
            rc = check_launch("k_fill_actions");
        }
    }
    q->outer.kind = outer_keep;
    q->compact_valid = 0;
    return rc;
}


// qenv_reward_function / qenv_mix: the reference's methods on caller-set attributes (what = 0 / 1)
template <typename T, typename S, bool LEAN> int do_method(qenv_s* q, int what, int64_t n, const void* const* in, void* const* out) {
    using M = typename MathSel<T, LEAN>::type;
    const Cfg<T> c = make_cfg<T>(q);
    const dim3 g = grid_for(n), b(256);
    if (what == 0) {
        hipLaunchKernelGGL((k_reward_function<T, S, M>), g, b, 0, q->stream, n, c, (const S*)in[0], (const S*)in[1], (const S*)in[2],
                           (const S*)in[3], (const int32_t*)in[4], (const uint8_t*)in[5], (S*)out[0], (uint8_t*)out[1],
                           (int32_t*)out[2], (S*)out[3]);
        return check_launch("k_reward_function");
    }
    hipLaunchKernelGGL((k_mix<T, S, LEAN>), g, b, 0, q->stream, n, c, (const S*)in[0], (S*)out[0], (S*)out[1], (S*)out[2]);
    return check_launch("k_mix");
}

// The env kernels are compiled in four translation units (fp32, mixed, fp64, lean fp64), each instantiating the host
// templates of its own (arithmetic, storage) pair; every other unit only sees these declarations.
#define QENV_ENV_INSTANCES(X, T, S, LEAN)                                                                                \
    X template int do_step<T, S, LEAN>(qenv_s*, const void*, const uint8_t*, void*, void*, uint8_t*, void*, int, int,    \
                                       const StepExtra*);                                                                \
    X template int do_rollout<T, S, LEAN>(qenv_s*, int, int, const double*, int, int, const void*);                      \
    X template int do_reset<T, S, LEAN>(qenv_s*, const void*, const uint8_t*, void*, void*, int, const ResetSample*);     \
    X template int do_method<T, S, LEAN>(qenv_s*, int, int64_t, const void* const*, void* const*);                       \
    X template int do_ppo_fused<T, S, LEAN>(qenv_s*, const qenv_ppo_step_args*, int, int, const void*);
#ifndef QENV_ENV_UNIT_F32
QENV_ENV_INSTANCES(extern, float, float, false)
#endif
#ifndef QENV_ENV_UNIT_MIX
QENV_ENV_INSTANCES(extern, float, double, false)
#endif
#ifndef QENV_ENV_UNIT_F64
QENV_ENV_INSTANCES(extern, double, double, false)
#endif
#ifndef QENV_ENV_UNIT_LEAN
QENV_ENV_INSTANCES(extern, double, double, true)
#endif

}  // namespace qenv
