/*
 * qenv.h -- C-ABI of the MI355X-native batched quadrotor environment (libqenv.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference has no
 * FFI layer: the path sits behind the duck-typed Python class `quad`
 * (/root/reference/environment/quadrotor_env.py:111-577) and a handful of controller /
 * PPO helpers.  Each entry point below names the reference interface it replaces.
 * The Python host wrapper (autonomous_quadrotor_environment_amd/) binds these symbols
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - No torch types, no C++ types: plain pointers, sizes and enums.
 *  - All array arguments are DEVICE pointers (HBM) in structure-of-arrays layout:
 *    a "[R][N]" array holds R rows of N contiguous elements (row r, env e at r*N+e),
 *    element type = the handle's dtype (float for QENV_F32, double for QENV_F64)
 *    unless stated otherwise.  Pointers documented "nullable" may be NULL.
 *  - The caller owns every buffer it passes in, including the workspace; getters
 *    return pointers INTO the workspace (valid for the handle's lifetime).
 *  - Every call enqueues work on the handle's HIP stream (qenv_set_stream) and
 *    returns without synchronising; nothing allocates or frees device memory, so
 *    call sequences can be captured into a hipGraph.
 *  - Return value: 0 = ok, negative = error (see qenv_status); no exceptions cross
 *    the ABI.  qenv_last_error() gives a thread-local message.
 *  - One handle per GPU shard; calls on one handle must be serialised by the caller.
 */
#ifndef QENV_H
#define QENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 10: qenv_permute_envs (placement: the envs of a batch in another order); qenv_dense_tail_rows (the dense tail on sample-major features).
 * 9: qenv_mlp_grad_ppo / qenv_mlp_grad_ppo_workspace_bytes (backward + weight gradients of a head in one launch: no activation
 *    round trip through HBM); a plain qenv_set_ppo_fused(q, 0 / 1) leaves a forced env block as it is.
 * 8: qenv_set_ppo_fused_block, qenv_get_ppo_fused; qenv_set_robust refuses the lean fp64 handle (it has no robust kernels) instead of failing at
 *    the first step; qenv_ppo_rollout validates its arguments up front on every path.
 * 7: qenv_policy_rollout; qenv_ppo_rollout (in-band) and qenv_policy_rollout run as one launch where k_ppo_fused applies.
 * 6: qenv_ppo_step_args gains obs_rows_from / obs_slices (the slice layout of stored observations); qenv_dense_tail_chunks_strided.
 * 5: qenv_ppo_step_args gains out_valid / inband / key_by_episode (in-band warm-up); qenv_gae_masked,
 *    qenv_adv_normalize_masked, qenv_mlp_backward_ppo_masked (the `valid` rows of such a rollout); qenv_mlp_prepare /
 *    qenv_mlp_image_bytes; qenv_reward_function, qenv_mix.
 * 4: qenv_mlp_backward_ppo's sums has five entries ([4]: d actor / d std); qenv_ppo_step / qenv_ppo_flush; compute mode LEAN.
 * 3: QENV_FIELD_EP_RETURN no longer counts the rewards of the reset's own warm-up steps; added qenv_mlp_backward_ppo,
 *    qenv_weight_grad_deterministic, qenv_gae_deterministic and their workspace-size queries. */
#define QENV_ABI_VERSION 10

typedef struct qenv_s *qenv_t;

typedef enum {
    QENV_OK = 0,
    QENV_ERR_ARG = -1,      /* bad argument (null, size, enum out of range)        */
    QENV_ERR_WORKSPACE = -2,/* workspace too small or misaligned                   */
    QENV_ERR_HIP = -3,      /* a HIP runtime call failed (message has the detail)  */
    QENV_ERR_STATE = -4     /* call not valid in the handle's configuration        */
} qenv_status;

typedef enum { QENV_F32 = 0, QENV_F64 = 1 } qenv_dtype;

/* QENV_COMPUTE_F32 on a QENV_F64 handle = "mixed": fp64 rows, fp32 arithmetic; the state rows advance by
 * state += (double)increment, so the per-step rounding of an fp32 state never enters the closed loop
 * (DESIGN.md section 5).  Same HBM bytes as fp64, ALU cost of fp32.
 *
 * QENV_COMPUTE_LEAN on a QENV_F64 handle = "lean fp64": every operation on the state is an fp64 multiply-add (on
 * gfx950 a wave alone on its SIMD issues v_fma_f64 at the v_fma_f32 rate); divisions / square roots are the hardware
 * seed + two Newton steps (<= 2 ulp) and the adaptive step-size CONTROL of RK45 runs on hardware approximations with
 * the accept/reject decision re-evaluated exactly inside a guard band, so the accepted sub-steps are those of the
 * NATIVE fp64 handle.  Trajectories within 1e-9 of the reference over closed-loop episodes at >= 1.5x the NATIVE
 * fp64 rate.  No robust_control variant. */
typedef enum { QENV_COMPUTE_NATIVE = 0, QENV_COMPUTE_F32 = 1, QENV_COMPUTE_LEAN = 2 } qenv_compute;

/* quadrotor_env.py (MAIN) vs quadrotor_env_opt.py (OPT) dynamics/reward variant */
typedef enum { QENV_MAIN = 0, QENV_OPT = 1 } qenv_variant;

/* QENV_RK45: scipy.integrate.solve_ivp defaults restarted every step (adaptive
 * Dormand-Prince 5(4), rtol 1e-3, atol 1e-6) = the reference's behaviour
 * (quadrotor_env.py:483).  QENV_DP5: one fixed DP5 step of t_step.  QENV_RK4: classic
 * RK4 (labelled option; BASELINE.json's wording). */
typedef enum { QENV_RK45 = 0, QENV_DP5 = 1, QENV_RK4 = 2 } qenv_integrator;

/* On-device inner-loop controllers for qenv_rollout */
typedef enum {
    QENV_CTRL_CONST = 0, /* one constant action for every env and step              */
    QENV_CTRL_LQR = 1,   /* environment/controller/lqr_quad.py:128-154              */
    QENV_CTRL_PID = 2,   /* environment/controller/pid_vel_control.py:29-127        */
    QENV_CTRL_PIDPOS = 3 /* environment/controller/PID_control.py:29-106 (position PD) */
} qenv_controller;

/* Rows of the per-env workspace exposed through qenv_field (the attributes callers of
 * `quad` read after step(): quadrotor_env.py:485-498). */
typedef enum {
    QENV_FIELD_STATE = 0,          /* [13][N]  state == previous_state               */
    QENV_FIELD_ANG = 1,            /* [3][N]   Euler angles phi,theta,psi            */
    QENV_FIELD_PREV_ANG = 2,       /* [3][N]                                         */
    QENV_FIELD_ANG_VEL = 3,        /* [3][N]   finite-difference Euler rates         */
    QENV_FIELD_VQ = 4,             /* [4][N]   quaternion rate of the last RHS call  */
    QENV_FIELD_W = 5,              /* [4][N]   rotor speeds                          */
    QENV_FIELD_STEP_EFFORT = 6,    /* [4][N]                                         */
    QENV_FIELD_CLIPPED_ACTION = 7, /* [4][N]                                         */
    QENV_FIELD_ACCEL = 8,          /* [3][N]   inertial accel of the last RHS call   */
    QENV_FIELD_PREV_SHAPING = 9,   /* [1][N]                                         */
    QENV_FIELD_REWARD = 10,        /* [1][N]                                         */
    QENV_FIELD_ABS_SUM = 11,       /* [1][N]   control-effort accumulator            */
    QENV_FIELD_EP_RETURN = 12,     /* [1][N]   sum of rewards since reset (the reset's own T warm-up steps excluded: the worker's reward_sum, ppo.py:236,258) */
    QENV_FIELD_PID = 13,           /* [15][N]  ix[6], x_old[6], ang_d_ant[3]         */
    QENV_FIELD_I = 14,             /* [1][N]   int32 step counter                    */
    QENV_FIELD_FLAGS = 15,         /* [1][N]   int32 bit0 done, bit1 solved,
                                               bit2 prev_shaping valid              */
    QENV_FIELD_WINDOW = 16,        /* [15*T][N] float32 observation ring (row block
                                               s = slice written at push s mod T)   */
    QENV_FIELD_LAST_RETURN = 17,   /* [1][N]   return of the last finished episode    */
    QENV_FIELD_EPISODES = 18,      /* [1][N]   int32 episodes finished (masked resets)  */
    QENV_FIELD_ROBUST_PARAMS = 19, /* [12][N]  episode_kf(4), episode_m, episode_ir(4), diag episode_J(3) */
    QENV_FIELD_GUST = 20,          /* [6][N]   robust_control.gust(3), last_gust(3)   */
    QENV_FIELD_GUST_EVALS = 21,    /* [1][N]   int32 RHS evaluations (= gust draws) of the env's last redraw step */
    QENV_FIELD_OUTER = 22,         /* [9][N]   outer loop: integral(3), previous error(3), velocity command(3) */
    QENV_FIELD_COUNT_ = 23
} qenv_field_id;

#define QENV_FLAG_DONE 1
#define QENV_FLAG_SOLVED 2
#define QENV_FLAG_HAS_SHAPING 4
#define QENV_FLAG_ANG_OK 8      /* prev_ang row belongs to the state row (internal bookkeeping) */

/* qenv_step flags */
#define QENV_STEP_FULL 0          /* obs + Euler + done + reward + effort (quad.step)     */
#define QENV_STEP_DYNAMICS_ONLY 1 /* integrate state only (BASELINE config 2)             */
#define QENV_STEP_NO_AUX 2        /* FULL but skip the attribute rows nobody reads in
                                     batched rollouts (w, effort, clipped_action, accel,
                                     ang_vel) -- 224 B/env-step of SURVEY.md section 8d   */
#define QENV_STEP_MODE_MASK 3
#define QENV_STEP_PUSH_WINDOW 4   /* OR-ed in: also append [raw action, vel, q, q_dot] to the
                                     observation ring (dl_in_gen.dl_input fused into the step) */

/* quad.__init__(t_step, n, training, euler, direct_control, T, clipped),
 * quadrotor_env.py:112-187 (euler is unused by the reference as well). */
typedef struct {
    int64_t n_envs;
    double t_step;
    int32_t n_max;          /* `n`; the handle adds T like the reference (:157)        */
    int32_t T;              /* warm-up steps in reset / observation window length      */
    int32_t training;       /* truthy -> solved sets done (:565); OPT: PPO_TRAINING    */
    int32_t direct_control; /* 1: action = 4 rotor commands in [-1,1]; 0: [F,Mx,My,Mz] */
    int32_t clipped;        /* f2w clipping (:218)                                     */
    int32_t variant;        /* qenv_variant                                            */
    int32_t integrator;     /* qenv_integrator                                         */
    int32_t dtype;          /* qenv_dtype: element type of the workspace and all arrays */
    int32_t compute;        /* qenv_compute: arithmetic precision of step / rollout     */
} qenv_config;

/* ---- lifetime ------------------------------------------------------------------- */
int qenv_abi_version(void);
const char *qenv_last_error(void);

/* Bytes of device memory the handle needs for n_envs (256-byte aligned base). */
size_t qenv_workspace_bytes(const qenv_config *cfg);

/* Replaces quad.__init__ (quadrotor_env.py:112).  `workspace` is caller-allocated
 * device memory of at least qenv_workspace_bytes(cfg); it is zeroed on `stream`. */
int qenv_create(const qenv_config *cfg, void *workspace, size_t workspace_bytes, void *stream,
                qenv_t *out);
int qenv_destroy(qenv_t q);
int qenv_set_stream(qenv_t q, void *stream); /* hipStream_t */

/* Device pointer of a workspace row block + its row count (quad attributes). */
int qenv_field(qenv_t q, int field_id, void **dev_ptr, int32_t *rows);

/* ABI 10.  Placement: the envs of the batch in another order -- slot p receives everything slot order[p] held (state,
 * bookkeeping incl. what quad.reset leaves alone -- prev_ang, ang_vel: quadrotor_env.py:408-454 --, PID state, returns,
 * counters, flags, observation ring).  A wavefront runs as many solve_ivp sub-steps per env step as its slowest lane needs
 * (quadrotor_env.py:483: a new solve_ivp per step); envs that a controller does not stabilise take two or three for the rest
 * of their episode, and sorted together (e.g. by |omega|) they stop holding one lane of every wavefront: BASELINE configs[2]
 * +14 % env-steps/s (hard envs first: the slow wavefronts are dispatched first), every env's trajectory bit for bit the same.  order: device, int32[n_envs], a permutation (checked);
 * scratch: device, qenv_workspace_bytes() bytes.  The caller keeps the slot -> env map for its own views (the Python
 * wrapper: BatchedQuad.regroup / logical).  Refused while a masked reset's lists are in use and on handles with per-env
 * state outside the workspace (attached sensor, outer-loop table, robust plant).  Synchronises `stream` once. */
int qenv_permute_envs(qenv_t q, const int32_t *order, void *scratch, size_t scratch_bytes);

/* ---- the hot path ------------------------------------------------------------------ */

/* Replaces quad.reset(det_state) (quadrotor_env.py:408-454): loads the initial
 * 13-states, clears counters/flags/prev_shaping/abs_sum (NOT prev_ang, :172), then runs
 * the T warm-up steps with zero_control.
 *   init_states [13][N]: explicit states (det_state, or the host-side replay of the
 *                        reference's RNG draw order for the random branch)
 *   mask        nullable uint8 [N]: reset only envs with mask != 0
 *   out_states  nullable [T][14][N], out_actions nullable [T][4][N]
 *   push_window nonzero: the T warm-up slices also refill the observation ring of the reset
 *               envs (worker loop ppo.py:234-240) WITHOUT moving the ring head, so envs that
 *               are not reset stay in phase.
 * With a mask, each reset env first banks its episode return (LAST_RETURN, EPISODES). */
int qenv_reset(qenv_t q, const void *init_states, const uint8_t *mask, void *out_states,
               void *out_actions, int push_window);

/* Replaces quad.step(action) (quadrotor_env.py:458-498) for all N envs.
 *   actions    [4][N], nullable = zero_control for every env
 *   out_obs    nullable [14][N]  quat_state = [state[0:10], V_q]
 *   out_reward nullable [N];  out_done nullable uint8 [N]
 *   out_mask   nullable [N]: 1 - done in the handle's dtype (the `masks` GAE consumes)
 *   flags      QENV_STEP_* mode, optionally | QENV_STEP_PUSH_WINDOW                */
int qenv_step(qenv_t q, const void *actions, void *out_obs, void *out_reward, uint8_t *out_done,
              void *out_mask, int flags);

/* quad.done_condition() followed by quad.reward_function() -- quadrotor_env.py:500-573 (MAIN),
 * quadrotor_env_opt.py:410-480 (OPT) -- on attributes the CALLER sets, the way the reference's
 * methods can be invoked on a hand-set object (its reward-branch table is built like that):
 *   state [13][n], ang [3][n], action [4][n] (self.action), prev_shaping [n] (NaN = None) in the
 *   handle's dtype; i int32 [n] (self.i); done_in uint8 [n] (self.done before the calls).
 * The handle supplies the configuration the methods read (variant, training, n + T, the done
 * box, zero_control); n is independent of the handle's n_envs.  Runs the device function the
 * step kernels run.  Outputs: reward [n], done uint8 [n], solved int32 [n], shaping [n]
 * (= self.prev_shaping afterwards). */
int qenv_reward_function(qenv_t q, int64_t n, const void *state, const void *ang, const void *action,
                         const void *prev_shaping, const int32_t *i, const uint8_t *done_in,
                         void *out_reward, uint8_t *out_done, int32_t *out_solved, void *out_shaping);

/* quad.f2F(action) on a direct_control handle (quadrotor_env.py:247-272) / quad.f2w(action[0],
 * action[1:]) on an indirect one (:197-245; quadrotor_env_opt.py:144-176) for n actions [4][n],
 * through the action pre-processing every step kernel starts with (clip to [-1, 1] / rotor
 * saturation per the handle's `clipped`).  out_w [4][n]: rotor speeds; out_fm [4][n]: the
 * [F, Mx, My, Mz] the dynamics integrate (f2w: FM_new); out_effort [4][n]: step_effort. */
int qenv_mix(qenv_t q, int64_t n, const void *action, void *out_w, void *out_fm, void *out_effort);

/* n_steps fused steps in ONE launch with the controller evaluated on device and the
 * state held in registers between steps -- the loop bodies of lqr_quad.py:123-168 and
 * pid_vel_control.py:141-164 (both ignore `done` and keep stepping), or a constant
 * action.  const_action: HOST pointer to 4 doubles (CONST only, nullable =
 * zero_control).  pid_first_action: PID only, nonzero = the first step uses
 * [M*G,0,0,0] like pid_vel_control.py:146.  PIDPOS holds one target (const_action then
 * points to its 7 doubles {xd,yd,zd,dxd,dyd,dzd,psi_d}).  Rewards are summed into EP_RETURN.
 * flags: QENV_STEP_* (DYNAMICS_ONLY skips Euler/done/reward, CONST only). */
int qenv_rollout(qenv_t q, int controller, int n_steps, const double *const_action,
                 int pid_first_action, int flags);

/* Trajectory tracking in one launch: the position PD of PID_control.py with a new target every step, as the
 * loop PID_control.py:245-263 feeds it from mission_control.get_error().  targets: DEVICE [n_steps][7] in the
 * handle's dtype, row s = {xd, yd, zd, dxd, dyd, dzd, psi_d} used at step s (shared by all envs). */
int qenv_rollout_track(qenv_t q, int n_steps, const void *targets, int flags);

/* Stand-alone controller laws writing actions [4][N] from the workspace rows
 * (state, ang, ang_vel) -- for callers that interleave them with qenv_step.
 * LQR: lqr_quad.py:128-154.  PID: pid_control.control(xd, 0, psd, 0),
 * pid_vel_control.py:98-110; targets: HOST pointer to 4 doubles {vx,vy,vz,psi}
 * (nullable = zeros); updates the PID rows of the workspace. */
int qenv_lqr_action(qenv_t q, void *out_actions);
int qenv_pid_action(qenv_t q, const double *targets, void *out_actions);
int qenv_pid_reset(qenv_t q, const uint8_t *mask); /* new pid_control(), :144 */
/* Position-tracking PD of PID_control.py:98-110, pid_control.control(xd, dxd, psi_d, 0):
 * targets: HOST pointer to 7 doubles {xd, yd, zd, dxd, dyd, dzd, psi_d} (nullable = zeros) --
 * one trajectory point shared by all envs, as mission_control.get_error(t) supplies per step
 * (PID_control.py:250-263).  Uses and updates the same PID workspace rows. */
int qenv_pidpos_action(qenv_t q, const double *targets, void *out_actions);

/* ---- PPO inner loop ------------------------------------------------------------------ */

/* Replaces dl_in_gen.dl_input (environment/controller/dl_auxiliary.py:25-32): append
 * one slice [action(4), vel(3), q(4), q_dot(4)] built from obs [14][N] and actions
 * [4][N] to each env's T-slice float32 ring.  reset_first != 0 zeroes the ring first
 * (dl_in_gen.reset). */
int qenv_window_push(qenv_t q, const void *obs, const void *actions, int reset_first);

/* The 75-128-128-n policy (model.py:27-42 with N = 128, T = 5: the reference's checked-in weights) has a PREPARED form:
 * the weights in the LDS layout of the forward kernel (bf16 three-way split fragments of W1 / W2, W3 as [h][4] rows,
 * biases), built once per weight update instead of once per launch and workgroup.
 *   qenv_mlp_image_bytes  size of the image for this shape, 0 if the shape has none.
 *   qenv_mlp_prepare      builds the image (caller-owned device buffer, 16-byte aligned) on the handle's stream and
 *                         registers it for `head` (0 actor, 1 critic) of this handle: qenv_mlp_forward /
 *                         qenv_policy_act / qenv_ppo_step calls whose W1 pointer and n_out equal the registered ones
 *                         then read the image instead of the fp32 weights.  Call it again after changing the weights
 *                         in place (an optimiser step); image = NULL unregisters.  Without it every call stages the
 *                         weights itself, with the same results bit for bit. */
size_t qenv_mlp_image_bytes(int k_in, int hidden);
int qenv_mlp_prepare(qenv_t q, int head, int k_in, int hidden, int n_out, const float *W1, const float *b1,
                     const float *W2, const float *b2, const float *W3, const float *b3, void *image);

/* Replaces ActorCritic.actor / .critic forward (environment/controller/model.py:27-42):
 * y = [tanh](W3 tanh(W2 tanh(W1 x + b1) + b2) + b3) on the observation ring, fp32
 * MFMA.  Weights: float32 device, row-major [out][in] exactly as torch's nn.Linear
 * (W1 [H][k_in], W2 [H][H], W3 [n_out][H]); H in {16,32,64,128,256}; n_out <= 4; k_in = the row length of W1 as the
 * caller allocated it, which must be the handle's 15*T (QENV_ERR_ARG otherwise: a weight set built for another window
 * would be read out of bounds or mis-strided).  out [n_out][N] in the handle's dtype.  obs_store nullable float32 [15*T][N]: the window in
 * time order, as read by the first layer (the `states` a PPO rollout keeps, ppo.py:250). */
int qenv_mlp_forward(qenv_t q, int k_in, int hidden, int n_out, int final_tanh, const float *W1,
                     const float *b1, const float *W2, const float *b2, const float *W3,
                     const float *b3, void *out, float *obs_store);

/* ---- onboard sensor model: class sensor, environment/quadrotor_env.py:579-724 -------
 * Batched IMU (bias drift + white noise), GPS, TRIAD attitude and the dead-reckoning integrators, evaluated the way
 * visual_landing/rl_worker.py:164-175 (`sensor_sp`) calls them after every quad.step.  sens: caller-owned
 * [QENV_SENSOR_ROWS][N] array in the handle's dtype (per-env sensor state).  Noise: injected standard normals
 * (z27 [27][N] in the reference's draw order, u3 [3][N] uniforms for bias_reset) or, if NULL, Philox keyed by
 * (seed; global env index, step).  error = sensor.error (0 disables noise and drift).
 * qenv_sensor_read writes out [24][N] = states_sens[14] {x,vx,y,vy,z,vz dead-reckoned, quaternion_gyro[4],
 * quaternion_vel[4]}, gps_pos[3], gps_vel[3], quaternion_triad[4].  It needs the ACCEL / W rows, i.e. the env must be
 * stepped with QENV_STEP_FULL.  Faithful to the reference's aliasing quirk: sensor.reset() makes quaternion_t0 a view
 * of the array quad.state is bound to (:630), so the first gyro integration after it advances the ENV's own
 * quaternion rows in place (:719) -- but only if the env has not been stepped in between (quad.step rebinds quad.state
 * to a new array; the view then points at an orphan and the env is left alone). */
#define QENV_SENSOR_ROWS 31
/* params: NULL = sensor.__init__'s defaults (:588-592), or eight doubles {accel_std, gyro_std, magnet_std, gps_std_p,
 * gps_std_v, accel_bias_drift, gyro_bias_drift, magnet_bias_drift}; every one is multiplied by `error` (0 / 1). */
int qenv_sensor_reset(qenv_t q, void *sens, const void *u3, uint64_t seed, uint64_t episode,
                      int64_t env_offset, int error, const double *params);
int qenv_sensor_read(qenv_t q, void *sens, const void *z27, uint64_t seed, uint64_t step,
                     int64_t env_offset, int error, const double *params, void *out);
/* The class's methods one by one (the N = 1 drop-in `sensor` of the host package calls them like the reference's
 * scripts do): ops[0..n_ops) with 0 accel (:612), 1 gyro (:623), 2 gps (:633), 3 triad (:641), 4 accel_int (:697),
 * 5 gyro_int (:712).  z: the standard normals the methods draw, consecutive rows [sum][N] in call order (accel 3,
 * gyro 3, gps 3 + 3, triad 3 + 3, accel_int 3 + 6, gyro_int 3); out: consecutive rows [sum][N]: accel 3, gyro 3,
 * gps pos 3 + vel 3, triad q 4 + R 9 (row-major), accel_int acceleration 3 + velocity 3 + position 3, gyro_int q 4.
 * qenv_sensor_bias_reset: bias_reset (:601-609) alone, from three uniforms u3 [3][N]. */
int qenv_sensor_ops(qenv_t q, void *sens, const int32_t *ops, int n_ops, const void *z, int error,
                    const double *params, void *out);
int qenv_sensor_bias_reset(qenv_t q, void *sens, const void *u3, int error, const double *params);
/* Sensor in the loop (environment/position.py:226-242, REAL_CTRL = False): with a sensor attached, every
 * qenv_step(... | QENV_STEP_PUSH_WINDOW) evaluates qenv_sensor_read's sequence inside the step kernel, on the state the
 * step ended in (no ACCEL / W rows needed: QENV_STEP_NO_AUX works), and pushes [action, dead-reckoned velocity - velocity
 * command of the outer loop, quaternion_gyro, quaternion_vel] into the policy window instead of the true state's slice.
 * z27 nullable [27][N] (read at every step: the caller refills it; NULL = Philox keyed by (seed; env, calls + k) for the
 * k-th fused step), out24 nullable [24][N] receiving qenv_sensor_read's outputs.  sens = NULL detaches.  The reset
 * warm-up pushes and the pos_pi landing loop keep the true state, as in the reference. */
int qenv_attach_sensor(qenv_t q, void *sens, const void *z27, uint64_t seed, int64_t env_offset, int error,
                       const double *params, void *out24, uint64_t calls);

/* Dense tail of the legacy landing policy (visual_landing/rl_nn_model.py:58-66 actor, :77-84 critic):
 * Linear(K,H) tanh Linear(H,H) tanh Linear(H,n_out) [tanh] on flattened conv features (K = 800, H = 64,
 * n_out = 3 / 1) -- the GEMM-shaped part of that model; its conv front-end needs rendered camera images
 * and is out of scope.  x float32 [K][N] (feature-major), weights float32 torch layout, H in {16, 32, 64, 128},
 * n_out <= 4, out float32 [n_out][N].  fp32 MFMA, W1 streamed through LDS in K-chunks. */
int qenv_dense_tail(int64_t n, int k_in, int hidden, int n_out, int final_tanh, const float *x,
                    const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                    const float *b3, float *out, void *stream);
/* ABI 10.  The same on the features in the reference's OWN layout: x_rows float32 [N][K], sample-major -- what
 * rl_nn_model.py:58 feeds its first Linear (the conv stack's output flattened to [batch, 800]); out stays [n_out][N].  No
 * transposed copy of the features for the caller to make (419 MB at 131 072 samples); a lane's operand fragment is 32
 * contiguous bytes of its sample's row.  The same time as the feature-major call (the kernel is not bound by how it reads:
 * 132-138 us either way, tools/ubench/read_patterns.hip) and bit-identical to it on the transposed features.  hidden 64,
 * K % 8 == 0, >= 65 536 samples, x_rows 16-byte aligned (else QENV_ERR_ARG: transpose and use qenv_dense_tail). */
int qenv_dense_tail_rows(int64_t n, int k_in, int hidden, int n_out, int final_tanh, const float *x_rows,
                         const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                         const float *b3, float *out, void *stream);
/* The same network over n_chunks independent chunks in ONE launch: x [n_chunks][K][N] -> out [n_chunks][n_out][N]
 * (the layout of a stored time-major rollout: the forward pass of PPO.update, ppo.py:176-183, over all its steps). */
int qenv_dense_tail_chunks(int64_t n, int32_t n_chunks, int k_in, int hidden, int n_out, int final_tanh,
                           const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                           const float *W3, const float *b3, float *out, void *stream);
/* ABI 6.  The same with `chunk_stride` elements between consecutive chunks of x (0 = K*n, dense): K = 75 observations
 * stored as slices (qenv_ppo_step_args.obs_slices) are read in place with chunk_stride = 15*n. */
int qenv_dense_tail_chunks_strided(int64_t n, int32_t n_chunks, int64_t chunk_stride, int k_in, int hidden, int n_out,
                                   int final_tanh, const float *x, const float *W1, const float *b1, const float *W2,
                                   const float *b2, const float *W3, const float *b3, float *out, void *stream);

/* Replaces ActorCritic.act's sampling (model.py:62-68): action = mean + std * eps,
 * logprob = Normal(mean, std).log_prob(action) per dimension.
 *   mean [4][N]; noise nullable [4][N] standard normal in the handle's dtype (injected, so a
 *   CPU run can share it); without it eps comes from Philox4x32-10 keyed by (seed, stream)
 *   with counter = (GLOBAL env index env_offset + e, step) -- shard- and replay-invariant.
 *   out_action [4][N], out_logprob nullable [4][N]. */
int qenv_sample_action(qenv_t q, const void *mean, double std, const void *noise, uint64_t seed,
                       uint64_t step, int64_t env_offset, void *out_action, void *out_logprob);

/* ActorCritic.act in ONE launch (model.py:52-72): qenv_mlp_forward of the actor (75-H-H-4, tanh) with
 * qenv_sample_action in its epilogue -- the same means, the same Philox streams / injected noise, the same
 * arithmetic, bit-identical actions and log-probs.  out_mean nullable [4][N]; out_action [4][N]; out_logprob
 * nullable [4][N]; k_in and obs_store as in qenv_mlp_forward; arrays other than obs_store in the handle's dtype. */
int qenv_policy_act(qenv_t q, int k_in, int hidden, const float *W1, const float *b1, const float *W2, const float *b2,
                    const float *W3, const float *b3, double std, const void *noise, uint64_t seed,
                    uint64_t step, int64_t env_offset, void *out_mean, void *out_action, void *out_logprob,
                    float *obs_store);

/* Random branch of quad.reset (quadrotor_env.py:440-445) on the device: Euler angles U(-.5,.5),
 * pos/vel/body-rate N(0,2) clipped to +-2.5 / +-5 / [-15, 7.5], quaternion by euler_quat.  Same
 * distributions as the reference, Philox streams instead of its MT19937 draw order (use the
 * host replay for seed parity).  out [13][N]; only envs with mask != 0 are written (nullable). */
int qenv_sample_init_states(qenv_t q, uint64_t seed, uint64_t episode, int64_t env_offset,
                            const uint8_t *mask, void *out_states);
/* qenv_sample_init_states + qenv_reset(mask) in one call and, for a masked reset, ONE launch before the warm-up
 * rollout: the masked envs' initial states are drawn inside the reset kernel (the same Philox streams, nothing is
 * written to a staging array), their episode return is banked and the list of reset envs is compacted there too.
 * This is the auto-reset of the rollout worker loop (ppo.py:232-262, `while` episodes until the batch is full).
 * mask nullable (all envs; then the T warm-up steps run as T step launches like qenv_reset). */
int qenv_reset_sampled(qenv_t q, uint64_t seed, uint64_t episode, int64_t env_offset, const uint8_t *mask,
                       int push_window);

/* One step of the rollout worker loop (ppo.py:232-262) with auto-reset: qenv_policy_act, qenv_step(NO_AUX |
 * PUSH_WINDOW) and the masked qenv_reset_sampled of the envs that finished -- the same kernels, the same Philox streams
 * (policy noise keyed by `step`, new initial states by episode = `step`), bit-identical buffers -- scheduled so that the
 * reset envs' T warm-up steps (the latency of ONE wavefront: ~60 us whatever the batch) leave the critical path: an env
 * that finishes in step t resets in that step kernel's epilogue (its return banked, its next initial state drawn) and
 * joins the reset list; during step t + 1 that list is warmed up on a side stream, followed by the policy step of
 * exactly those envs (gathered into dense 32-env tiles), while the main stream steps every other env.  After the call the finished envs
 * hold their new initial state but their warm-up is PENDING: it runs inside the next qenv_ppo_step, or as soon as any
 * other entry point that touches the env is called, or in qenv_ppo_flush (call it before reading the handle's fields).
 * The side-stream schedule is opt-in (QENV_PPO_OVERLAP=1, hidden 64 / 128, T = 5): it pays when few envs finish per step
 * and loses when many do; without it the same launches run in sequence on the handle's stream.
 * Arrays as in qenv_policy_act / qenv_step: out_mean, out_logprob, obs_store, out_mask, noise nullable. */
typedef struct {
    int32_t k_in, hidden;                       /* W1 is [hidden][k_in], k_in = 15*T of the handle */
    const float *W1, *b1, *W2, *b2, *W3, *b3;   /* actor, torch layout */
    double std;
    const void *noise;                          /* null = Philox */
    uint64_t seed, step;
    int64_t env_offset;
    void *out_mean, *out_action, *out_logprob;  /* [4][N] */
    float *obs_store;                           /* [15*T][N] */
    void *out_reward;                           /* [N] */
    uint8_t *out_done;                          /* [N] */
    void *out_mask;                             /* [N]: 1 - done */
    /* ABI 5.  inband != 0: IN-BAND warm-up -- the schedule above without its third part.  An env that finishes in a
     * POLICY step still restarts in that step kernel's epilogue, but the T warm-up steps of its reset (quadrotor_env.py
     * :448-453) are its next T REGULAR steps: recognised by the env's own step counter (i < T) they apply zero_control
     * whatever the policy sampled, push their window slice into the common ring slot, and report out_valid = 0; the
     * caller excludes those rows from GAE's normalisation moments and from the loss (qenv_gae / qenv_adv_normalize /
     * qenv_mlp_backward_ppo take the same array).  No list, no warm-up launch, nothing pending after the call: a
     * rollout step is policy + step.  The valid rows of an env are exactly the samples the schedule above collects for
     * the same episodes (same dynamics, same reward, same windows); they sit T rows later per finished episode.
     * key_by_episode != 0 (either schedule): the next initial state of an env is drawn from the Philox stream of ITS
     * OWN restart count instead of the stream of `step`, so that the n-th episode of an env starts from the same state
     * whenever it starts. */
    void *out_valid;                            /* [N] in the handle's dtype, nullable: 0 for a warm-up row, else 1 */
    int32_t inband, key_by_episode;
    /* ABI 6.  The SLICE layout of stored observations.  The window the policy sees at step t is the T newest 15-float
     * slices {action, v, q, q_dot} (dl_in_gen.dl_input), and every env pushes exactly one slice per step under the in-band
     * schedule -- consecutive windows overlap in T - 1 slices.  obs_rows_from > 0: obs_store receives only rows >=
     * obs_rows_from of the window (15 (T - 1): the newest slice).  obs_slices != 0 (qenv_ppo_rollout): obs_store is an array
     * [(n_steps + T - 1) * 15][N]; step t writes at row 15 t -- step 0 as obs_rows_from of `first` says (0: the whole window,
     * i.e. the rollout's first T slices), every later step only its newest slice -- so that afterwards the window of step t
     * is rows 15 t .. 15 t + 15 T - 1: the [n_steps][15 T][N] layout read with a chunk stride of 15 N
     * (qenv_dense_tail_chunks_strided; a fifth of the store traffic and of the buffer).  inband only (a warm-up LAUNCH pushes
     * T slices in one step); hidden 64 / 128 with T = 5. */
    int32_t obs_rows_from, obs_slices;
} qenv_ppo_step_args;
int qenv_ppo_step(qenv_t q, const qenv_ppo_step_args *args);
int qenv_ppo_flush(qenv_t q);
/* n_steps consecutive qenv_ppo_step calls issued from C: the `for t in range(max_timesteps)` of the worker loop
 * (ppo.py:239-262) for a whole horizon in one call.  `first` describes step 0; step t uses the same struct with
 * step + t, and with every per-step array advanced by t rows of the time-major rollout buffers: out_action /
 * out_logprob [n_steps][4][N], obs_store [n_steps][15*T][N], out_reward / out_mask / out_valid [n_steps][N] (and noise
 * [n_steps][4][N] when injected); out_mean and out_done are scratch and not advanced.  Identical to the n_steps separate
 * calls bit for bit; what it removes is the host's per-step work between the launches (a step is two ~25 us kernels at
 * 65 536 envs: a Python round trip per step leaves the GPU idle for about as long). */
int qenv_ppo_rollout(qenv_t q, const qenv_ppo_step_args *first, int n_steps);
/* ABI 7.  The evaluation loop of ppo_quad_eval.py:44-62 for n_steps steps in one call: per step the DETERMINISTIC policy
 * (action = tanh mean of the actor, model.py:27-33 -- qenv_mlp_forward with final_tanh = 1) and qenv_step(NO_AUX | PUSH_WINDOW)
 * on that action; no sampling, no auto-reset (done is sticky, the envs keep stepping as the reference's loop would if it did
 * not break).  out_actions [n_steps][4][N]; out_reward / out_mask [n_steps][N] nullable; out_done [N] nullable (the last
 * step's), all in the handle's dtype.  Identical to the 2 n_steps separate calls bit for bit.
 * Both this call and the in-band qenv_ppo_rollout run as ONE launch for the reference's controller (hidden 128, T = 5, RK45;
 * no robust_control / sensor / outer loop): a wave keeps its 64-env blocks for the whole horizon, per step the policy forward
 * of the block's two 32-env tiles and then the env step, the weights staged once per launch instead of once per step
 * (k_ppo_fused; qenv_set_ppo_fused(q, 0) keeps the separate launches). */
int qenv_policy_rollout(qenv_t q, int k_in, int hidden, const float *W1, const float *b1, const float *W2, const float *b2,
                        const float *W3, const float *b3, int n_steps, void *out_actions, void *out_reward,
                        uint8_t *out_done, void *out_mask);
/* ABI 7.  Whether this handle's in-band qenv_ppo_rollout and qenv_policy_rollout run as the
 * one fused launch where it applies (1, the default; the environment variable QENV_PPO_FUSED sets the initial value) or as the
 * separate policy / env-step launches (0).  Results are bit-identical either way.  5 / 6: on, with the launch's env blocks forced
 * to 32 / 64 envs per wavefront (by default a batch that leaves half the chip's wavefronts without a 64-env block runs as 32-env
 * blocks: one policy tile and a half-wave env step per wavefront and step; the tests force both).  A plain 0 / 1 leaves a
 * block forced earlier (here or by qenv_set_ppo_fused_block) as it is; qenv_set_ppo_fused_block(q, 0) clears it. */
int qenv_set_ppo_fused(qenv_t q, int enabled);
/* ABI 8.  Only the env block of the fused launch -- 0 (chosen by the batch size), 32 or 64 envs per wavefront -- leaving the
 * on / off state of qenv_set_ppo_fused (and of QENV_PPO_FUSED) as it is: what a caller that runs several handles side by side
 * on their own streams wants (64-env blocks keep a launch to its share of the chip) without overriding an A/B switch. */
int qenv_set_ppo_fused_block(qenv_t q, int block_envs);
/* ABI 8.  The two settings as the handle holds them: bit 0 = fused launch on, bits 8.. = forced block in envs (0, 32 or 64);
 * negative on a null handle. */
int qenv_get_ppo_fused(qenv_t q);

/* ---- robust_control: per-episode plant perturbation + gusts (environment/quadrotor_env.py:84-109) ----------
 * The reference hard-codes `self.robust_control = False` (:183) and a user flips the attribute; this is that
 * switch.  Main variant only (quadrotor_env_opt.py has no such mode).  With it on, every step / rollout of the
 * handle runs the perturbed plant: thrust loss u -= u*episode_kf after w (:235,:265), wind added to the
 * air-relative velocity (:317-320), per-rotor inertia (:341-343), mass (:360-361), body inertia (:381-382).
 * Dynamics-only steps are refused (the gusts follow the step counter i).  Handles: f32, mix and IEEE f64; the lean fp64 handle
 * (QENV_COMPUTE_LEAN) has no robust kernels and is refused here with QENV_ERR_ARG (ABI 8; before, at the first step). */
int qenv_set_robust(qenv_t q, int enabled);

/* robust_control.reset (:97-101): the 12 episode draws of the envs with mask != 0 (mask nullable).
 *   params non-null: [12][N] in the handle's dtype, rows = episode_kf(4) ~ U(0,0.1), episode_m ~ N(0,0.3),
 *     episode_ir(4) ~ U(0,0.1), diag episode_J(3) ~ N(0,0.1) -- the reference's draw order, so its seeded
 *     stream can be replayed on the host;
 *   params null: the same distributions from Philox4x32-10 keyed by (seed; env_offset + e, episode), episode =
 *     episodes[e], or episodes[e] + 1 in a masked call (made BEFORE the masked qenv_reset that bumps the counter). */
int qenv_robust_params(qenv_t q, const void *params, uint64_t seed, int64_t env_offset, const uint8_t *mask);

/* Source of the gust draws of robust_control.wind (:102-108).  In the steps with i % 500 == 1 the reference
 * draws a new gust ~ N(0,[5,5,2]) at EVERY RHS evaluation (8 ... several hundred per step, data dependent)
 * and each evaluation sees the previous draw; in between the wind is linspace(last_gust, gust, 500)[i%500-1].
 *   table non-null: [E][K][3][N] in the handle's dtype, borrowed until replaced: draw k of redraw epoch
 *     (i-1)/500 of the current episode (k, epoch clamped to K-1, E-1) -- parity runs replay recorded draws;
 *   table null: Philox4x32-10 keyed by (seed; env_offset + e, episodes[e], epoch, k) + Box-Muller. */
int qenv_robust_gusts(qenv_t q, const void *table, int32_t E, int32_t K, uint64_t seed, int64_t env_offset);

/* ---- outer position loops around the PPO velocity policy -----------------------------------------------------
 * The reference steers its velocity-stabilising policy by feeding it `state - state_error` (environment/position.py
 * :233-244, visual_landing/math_trajectory.py:118-133): only the three velocity entries of the window slice change.
 * With a loop configured, every qenv_step(... | QENV_STEP_PUSH_WINDOW) evaluates it on the NEW state inside the step
 * kernel, pushes [action, v - v_offset, q, q_dot] and leaves the velocity command in QENV_FIELD_OUTER rows 6..8.
 *
 * QENV_OUTER_VELPID  vel_pid (environment/controller/velocity_pid.py:16-35; P=3, I=0, D=0.05, dt=0.01 as hard-coded
 *   there): table [n_rows][7] = {xd,yd,zd,dxd,dyd,dzd,psi_d} rows of mission.get_error (mission.table()), in the
 *   handle's dtype; an env at step counter i uses row i - T - 1 (its k-th policy step), clamped to the last row.
 *   v_offset = target velocity + PID(position error); command = the PID term (`pid_error`).
 * QENV_OUTER_POSPI   pos_pi (math_trajectory.py:46-62) + the clip of :123 to [-0.5,-0.5,-1.666]..[0.5,0.5,0.333]:
 *   targets [3][N] = s_d per env; v_offset = command = clip(P (s_d - s) + D (0 - ds) + I integral).
 * A null table/targets pointer switches the loop off.  qenv_outer_reset zeroes the loop state of the masked envs
 * (position.py:209-210; math_trajectory.py `pos_pi.reset`). */
enum { QENV_OUTER_NONE = 0, QENV_OUTER_VELPID = 1, QENV_OUTER_POSPI = 2 };
int qenv_outer_velpid(qenv_t q, const void *table, int32_t n_rows);
int qenv_outer_pospi(qenv_t q, const void *targets, double P, double D, double I);
int qenv_outer_reset(qenv_t q, const uint8_t *mask);

/* Replaces PPO.get_advantages (environment/controller/ppo.py:125-141, ddof=0;
 * visual_landing/ppo_trainer.py:133-148, ddof=1) on a time-major rollout:
 *   values [n_steps+1][N], masks [n_steps][N] (1 - terminal), rewards [n_steps][N]
 *   -> returns [n_steps][N], adv [n_steps][N] (un-normalised: returns - values)
 *   -> stats: device double[3] = {count, sum(adv), sum(adv^2)} accumulated (caller
 *      zeroes it; all-reduce it across shards before qenv_adv_normalize).
 * Independent of any handle; n_envs envs scanned in parallel along time. */
int qenv_gae(int dtype, int64_t n_envs, int n_steps, const void *values, const void *masks,
             const void *rewards, void *returns, void *adv, double *stats, void *stream);
/* qenv_gae with run-to-run reproducible moments: each wavefront stores its partial sums in `workspace`
 * (qenv_gae_workspace_bytes(n_envs) bytes, no initialisation needed) and a second launch adds them to stats[1..2] in a
 * fixed order, instead of double-precision atomics whose order varies (the last bit of the normalised advantages). */
size_t qenv_gae_workspace_bytes(int64_t n_envs);
int qenv_gae_deterministic(int dtype, int64_t n_envs, int n_steps, const void *values, const void *masks,
                           const void *rewards, void *returns, void *adv, double *stats, void *workspace,
                           void *stream);
int qenv_adv_normalize(int dtype, int64_t n_elems, void *adv, const double *stats, int ddof,
                       void *stream);
/* The same two for an in-band warm-up rollout (qenv_ppo_step with inband): valid [n_steps][N] in `dtype`, 0 for the rows
 * that were a restarted env's warm-up steps.  Such rows follow a terminal row (mask 0) or open the horizon, so no valid
 * row's recurrence reads them; their advantage is set to 0 (and stays 0 through the normalisation) and they are left out
 * of the moments: stats[0] receives the number of VALID rows.  workspace: null, or qenv_gae_workspace_bytes(n_envs)
 * bytes for the fixed-order sums of qenv_gae_deterministic.  valid = null is qenv_gae / qenv_adv_normalize. */
int qenv_gae_masked(int dtype, int64_t n_envs, int n_steps, const void *values, const void *masks,
                    const void *rewards, const void *valid, void *returns, void *adv, double *stats,
                    void *workspace, void *stream);
int qenv_adv_normalize_masked(int dtype, int64_t n_elems, void *adv, const void *valid, const double *stats,
                              int ddof, void *stream);

/* Replaces the loss of PPO.update (environment/controller/ppo.py:185-203): per sample
 *   logp_k = Normal(mean_k, std).log_prob(action_k); ratio = exp(sum_k logp_k - sum_k old_logprob_k)
 *   actor = -min(ratio*adv, clamp(ratio, 1-eps, 1+eps)*adv); entropy_k = 0.5 + 0.5 log(2 pi) + log std
 *   loss.mean() = mean(actor) + c_value * MSE(value, returns) - c_entropy * mean(sum_k entropy_k)
 * (ppo.py: c_value 0.5, c_entropy 0.006, eps 0.2; ppo_trainer.py:112-122 uses the same terms).
 * Inputs [4][B] / [B] in `dtype`.  sums: device double[4] accumulated (caller zeroes):
 * {B, sum actor, sum (value-returns)^2, sum_k entropy_k * B}.  grad_mean [4][B] / grad_value [B]
 * (nullable): d loss.mean() / d mean, d value -- the seed of the backward pass. */
int qenv_ppo_loss(int dtype, int64_t batch, const void *mean, const void *value, const void *actions,
                  const void *old_logprob, const void *adv, const void *returns, double std,
                  double eps_clip, double c_value, double *sums, void *grad_mean, void *grad_value,
                  void *stream);

/* ---- optimiser half of PPO.update (ppo.py:199-203: loss.mean().backward(); optimizer.step()) -------------
 * All matrices are float32 in the rollout's time-major chunked SoA: sample (c, e) of an R-row matrix at
 * base + (c*R + r)*n_envs + e, c < n_chunks (one chunk per rollout step).
 *
 * qenv_mlp_backward: recomputes the forward of the tanh MLP (model.py:27-42) on obs [c][K][N] and backpropagates
 * seed = d loss / d output [c][n_out][N] (from qenv_ppo_loss) through the activations:
 *   h1, h2 [c][H][N] (post-tanh), dz2, dz1 [c][H][N] (gradients w.r.t. the pre-activations), dz3 [c][4][N].
 * qenv_weight_grad: C[m][n] += sum_samples A[m][s] B[n][s] (C row-major [M][Nn] = torch's weight layout) and, if
 *   rowsum != NULL, rowsum[m] += sum_samples A[m][s] (bias gradient): dW3 = dz3 x h2, dW2 = dz2 x h1, dW1 = dz1 x obs.
 *   The caller zeroes C / rowsum; accumulation order is not deterministic (fp32 atomics).
 * qenv_adam_step: torch.optim.Adam update (ppo.py:105) of n parameters, step counted from 1. */
int qenv_mlp_backward(int64_t n_envs, int n_chunks, int k_in, int hidden, int n_out, int final_tanh,
                      const float *obs, const float *seed, const float *W1, const float *b1,
                      const float *W2, const float *b2, const float *W3, const float *b3, float *h1,
                      float *h2, float *dz2, float *dz1, float *dz3, void *stream);
/* qenv_mlp_backward with the PPO loss of one head fused in (ppo.py:185-199): the seed d loss.mean() / d output is
 * computed inside the backward kernel from the forward it recomputes anyway, so an update needs neither a separate
 * forward launch nor qenv_ppo_loss.  head 0 = actor (75-H-H-4, tanh; clipped surrogate from actions / old_logprob
 * [c][4][N] and adv [c][N]; adds the surrogate sum to sums[1]), head 1 = critic (75-H-H-1; c_value * MSE against
 * returns [c][N]; adds the sum of squared errors to sums[2]).  batch = number of samples the loss is averaged over
 * (n_chunks * n_envs on one learner).  sums: FIVE doubles, [0..3] as in qenv_ppo_loss (caller zeroes; [0] and [3] are
 * the caller's); the actor head also adds sum_s d actor_s / d std to sums[4] -- with the entropy term's -c_entropy * 4 / std
 * that is the gradient of the learned action std (model.py:45-48, fixed_std = False): d loss.mean() / d std =
 * sums[4] / batch - 4 c_entropy / std.  Same arithmetic and operation order as qenv_ppo_loss + qenv_mlp_backward. */
int qenv_mlp_backward_ppo(int64_t n_envs, int n_chunks, int k_in, int hidden, int head, const float *obs,
                          const float *actions, const float *old_logprob, const float *adv, const float *returns,
                          double std, double eps_clip, double c_value, int64_t batch, double *sums, const float *W1,
                          const float *b1, const float *W2, const float *b2, const float *W3, const float *b3,
                          float *h1, float *h2, float *dz2, float *dz1, float *dz3, void *stream);
/* qenv_mlp_backward_ppo with per-sample weights [c][N] (float, nullable = all ones): a sample of weight 0 contributes
 * neither to the loss sums nor to any gradient (its seed is zero) -- the warm-up rows of an in-band rollout; `batch` is
 * then the number of samples of non-zero weight (stats[0] of qenv_gae_masked). */
int qenv_mlp_backward_ppo_masked(int64_t n_envs, int n_chunks, int k_in, int hidden, int head, const float *obs,
                                 const float *actions, const float *old_logprob, const float *adv,
                                 const float *returns, const float *weight, double std, double eps_clip,
                                 double c_value, int64_t batch, double *sums, const float *W1, const float *b1,
                                 const float *W2, const float *b2, const float *W3, const float *b3, float *h1,
                                 float *h2, float *dz2, float *dz1, float *dz3, void *stream);
int qenv_weight_grad(int64_t n_envs, int n_chunks, int m_rows, int n_rows, const float *A, const float *B,
                     float *C, float *rowsum, void *stream);
/* ABI 9.  PPO.update's gradient step for ONE head of the reference's controller (hidden 128, k_in 75: model.py:27-42) as ONE
 * launch: forward recompute, the fused loss of qenv_mlp_backward_ppo_masked (same arguments, same `sums`), the backward
 * pass AND the six weight / bias gradients (environment/controller/ppo.py:185-203's loss.mean().backward()), accumulated
 * INTO gW1 [128][75], gb1 [128], gW2 [128][128], gb2 [128], gW3 [n_out][128], gb3 [n_out] (n_out = 4 actor / 1 critic; the
 * caller zeroes them per step).  No activation is written to memory: the eight wavefronts of a workgroup share a 32-sample
 * tile and contract it over the samples on the spot (qenv_grad5.hpp).  workspace: null = fp32 atomics (summation order
 * varies in the last bits), or qenv_mlp_grad_ppo_workspace_bytes() bytes (no initialisation needed) = every workgroup
 * stores its sums and a second launch adds them in a fixed order (bit-reproducible).  obs_chunk_stride: elements between the
 * [k_in][N] observation blocks of consecutive chunks -- 0 = dense (k_in * N), 15 * N = the slice layout a rollout stores
 * (qenv_ppo_step_args.obs_slices: one 15-float slice per step, consecutive windows overlap), read in place.  Other shapes: QENV_ERR_ARG (use
 * qenv_mlp_backward_ppo + qenv_weight_grad). */
size_t qenv_mlp_grad_ppo_workspace_bytes(void);
int qenv_mlp_grad_ppo(int64_t n_envs, int n_chunks, int k_in, int hidden, int head, const float *obs,
                      int64_t obs_chunk_stride, const float *actions, const float *old_logprob, const float *adv, const float *returns,
                      const float *weight, double std, double eps_clip, double c_value, int64_t batch, double *sums,
                      const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                      const float *b3, float *gW1, float *gb1, float *gW2, float *gb2, float *gW3, float *gb3,
                      void *workspace, void *stream);
/* qenv_weight_grad with a fixed summation order (bit-reproducible gradients, hence training runs): every workgroup
 * stores its partial C / rowsum in `workspace` (qenv_weight_grad_workspace_bytes(m_rows, n_rows) bytes, no
 * initialisation needed) and a second launch adds the slots in slot order. */
size_t qenv_weight_grad_workspace_bytes(int m_rows, int n_rows);
int qenv_weight_grad_deterministic(int64_t n_envs, int n_chunks, int m_rows, int n_rows, const float *A,
                                   const float *B, float *C, float *rowsum, void *workspace, void *stream);
int qenv_adam_step(int64_t n, float *param, const float *grad, float *m, float *v, double lr, double beta1,
                   double beta2, double eps, int step, void *stream);
/* qenv_adam_step with the step count kept in device memory: *step_counter is incremented on the device first (its own
 * one-thread launch) and the bias corrections 1 - beta^step are evaluated there, so a hipGraph that captured this call
 * advances the count on every replay (PpoLearner.update(graph=True)). */
int qenv_adam_step_counted(int64_t n, float *param, const float *grad, float *m, float *v, double lr, double beta1,
                           double beta2, double eps, int32_t *step_counter, void *stream);

/* Replaces visual_reward (visual_landing/rl_reward_fuction.py:3-102), batched.
 * marker,pos,vel,control,ang,v_ang [3][N]; last_shaping [N] (0 = none, like the
 * reference's truthiness test); step int32 [N]; outputs reward, shaping [N],
 * done uint8 [N], n_solved int32 [N]. */
int qenv_visual_reward(int dtype, int64_t n_envs, int total_steps, const void *marker,
                       const void *pos, const void *vel, const void *control,
                       const void *last_shaping, const int32_t *step, const void *ang,
                       const void *v_ang, void *out_reward, void *out_shaping, uint8_t *out_done,
                       int32_t *out_solved, void *stream);

/* Measurement utility (bench.py / profiles): dst[r][e] = src[r][e] with the SAME access shape as
 * the env kernels (one dword or qword per lane, rows N apart), so the rocprofv3 FETCH_SIZE /
 * WRITE_SIZE counters can be calibrated on a known byte count before they are used to price the
 * env kernels' HBM traffic (MI355X_MICROARCH.md, HBM section).  No reference counterpart. */
int qenv_debug_copy_rows(int dtype, void *dst, const void *src, int rows, int64_t n_envs,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QENV_H */
