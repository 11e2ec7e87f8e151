// qenv_env.hip -- C-ABI entry points of the env (include/qenv.h) + the fp32 instantiation of the env kernels
// (qenv_env_impl.hpp).  gfx950 only.
#define QENV_ENV_UNIT_F32
#include "qenv_env_impl.hpp"

namespace qenv {
QENV_ENV_INSTANCES(, float, float, false)
}  // namespace qenv

using namespace qenv;

// (arithmetic, storage) of a handle: fp32 | mixed (fp64 rows, fp32 arithmetic) | lean fp64 | fp64
#define QENV_DISPATCH(q, FN, ...)                                                                     \
    ((q)->cfg.dtype == QENV_F64                                                                       \
         ? ((q)->cfg.compute == QENV_COMPUTE_F32    ? FN<float, double, false>(__VA_ARGS__)           \
            : (q)->cfg.compute == QENV_COMPUTE_LEAN ? FN<double, double, true>(__VA_ARGS__)           \
                                                    : FN<double, double, false>(__VA_ARGS__))         \
         : FN<float, float, false>(__VA_ARGS__))

// ---- C-ABI ----------------------------------------------------------------------------------------
// ---- qenv_permute_envs' kernels (see there)
namespace qenv {
template <typename T>
__global__ __launch_bounds__(256) void k_gather_rows(T* dst, const T* src, const int32_t* order, int rows, int64_t N) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const int64_t s = order[p];
    for (int r = blockIdx.y; r < rows; r += gridDim.y) dst[(int64_t)r * N + p] = src[(int64_t)r * N + s];
}
__global__ __launch_bounds__(256) void k_check_permutation(const int32_t* order, int64_t N, uint8_t* seen, int32_t* bad) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= N) return;
    const int32_t s = order[p];
    if (s < 0 || s >= N) { atomicAdd(bad, 1); return; }
    seen[s] = 1;   // (a repeated source leaves another one unseen: counted by the caller's second pass)
}
__global__ __launch_bounds__(256) void k_count_unseen(const uint8_t* seen, int64_t N, int32_t* bad) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < N && !seen[p]) atomicAdd(bad, 1);
}
}  // namespace qenv

extern "C" {

int qenv_abi_version(void) { return QENV_ABI_VERSION; }
const char* qenv_last_error(void) { return g_err; }

size_t qenv_workspace_bytes(const qenv_config* cfg) {
    if (!cfg || cfg->n_envs <= 0 || cfg->T < 1) return 0;
    return ws_layout(*cfg).total;
}

int qenv_create(const qenv_config* cfg, void* workspace, size_t bytes, void* stream, qenv_t* out) {
    if (!cfg || !workspace || !out) return fail(QENV_ERR_ARG, "qenv_create: null argument");
    if (cfg->n_envs <= 0 || cfg->T < 1 || cfg->T > 64 || cfg->t_step <= 0)
        return fail(QENV_ERR_ARG, "qenv_create: n_envs, T (1..64) and t_step must be positive");
    if (cfg->dtype != QENV_F32 && cfg->dtype != QENV_F64) return fail(QENV_ERR_ARG, "qenv_create: bad dtype");
    if (cfg->compute != QENV_COMPUTE_NATIVE &&
        !((cfg->compute == QENV_COMPUTE_F32 || cfg->compute == QENV_COMPUTE_LEAN) && cfg->dtype == QENV_F64))
        return fail(QENV_ERR_ARG, "qenv_create: compute must be NATIVE, or F32 / LEAN on a QENV_F64 handle");
    if (cfg->integrator < QENV_RK45 || cfg->integrator > QENV_RK4) return fail(QENV_ERR_ARG, "qenv_create: bad integrator");
    if (cfg->variant != QENV_MAIN && cfg->variant != QENV_OPT) return fail(QENV_ERR_ARG, "qenv_create: bad variant");
    const WsLayout L = ws_layout(*cfg);
    if (bytes < L.total) return fail(QENV_ERR_WORKSPACE, "qenv_create: workspace smaller than qenv_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(QENV_ERR_WORKSPACE, "qenv_create: workspace must be 256-byte aligned");
    qenv_s* q = new (std::nothrow) qenv_s;
    if (!q) return fail(QENV_ERR_ARG, "qenv_create: out of host memory");
    q->cfg = *cfg; q->N = cfg->n_envs; q->ws = workspace; q->ws_bytes = bytes; q->stream = (hipStream_t)stream;
    q->wsi = (int32_t*)((char*)workspace + L.int_off);
    q->counters = (int32_t*)((char*)workspace + L.cnt_off);
    q->window = (float*)((char*)workspace + L.win_off);
    q->win_head = 0;
    q->ang_ok = 0;
    q->compact_valid = 0;
    q->warmup_push = 0;
    q->packed = getenv("QENV_PACKED") ? atoi(getenv("QENV_PACKED")) : -1;  // -1: automatic (see launch_rollout)
    q->robust = 0;
    q->count_cur = 0; q->count_used = 0;
    q->sens = qenv_sensor_link{};
    q->side = nullptr; q->ev_main = nullptr; q->ev_side = nullptr;
    q->env_flags[0] = (uint8_t*)workspace + L.flag_off; q->env_flags[1] = q->env_flags[0] + L.flag_stride;
    q->ar_pending = 0; q->ar_parity = 0; q->zero_on_act = nullptr;
    q->prep[0] = {nullptr, nullptr, 0}; q->prep[1] = {nullptr, nullptr, 0};
    q->ppo_fused = getenv("QENV_PPO_FUSED") ? (atoi(getenv("QENV_PPO_FUSED")) != 0) : 1;
    q->ppo_fused_block = 0;
    q->outer = qenv_outer{QENV_OUTER_NONE, nullptr, 0, {0, 0, 0}};
    q->gust = qenv_gust_src{nullptr, cfg->dtype == QENV_F64, 1, 1, 0x0B57u, 0x6057u, 0};
    hipError_t e = hipMemsetAsync(workspace, 0, L.total, q->stream);
    if (e != hipSuccess) { delete q; return fail_hip(e, "hipMemsetAsync(workspace)"); }
    *out = q;
    return QENV_OK;
}

int qenv_destroy(qenv_t q) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_destroy: null handle");
    if (q->ev_main) (void)hipEventDestroy(q->ev_main);
    if (q->ev_side) (void)hipEventDestroy(q->ev_side);
    if (q->side) (void)hipStreamDestroy(q->side);
    delete q;
    return QENV_OK;
}

int qenv_set_stream(qenv_t q, void* stream) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_set_stream: null handle");
    q->stream = (hipStream_t)stream;
    return QENV_OK;
}

int qenv_field(qenv_t q, int field_id, void** dev_ptr, int32_t* rows) {
    if (!q || !dev_ptr) return fail(QENV_ERR_ARG, "qenv_field: null argument");
    static const int row_of[] = {ROW_STATE, ROW_ANG, ROW_PREV_ANG, ROW_ANG_VEL, ROW_VQ, ROW_W, ROW_EFFORT, ROW_CLIPPED,
                                 ROW_ACCEL, ROW_PREV_SHAPING, ROW_REWARD, ROW_ABS_SUM, ROW_EP_RETURN, ROW_PID};
    static const int rows_of[] = {13, 3, 3, 3, 4, 4, 4, 4, 3, 1, 1, 1, 1, 15};
    const size_t esz = q->cfg.dtype == QENV_F64 ? 8 : 4;
    if (field_id >= 0 && field_id <= QENV_FIELD_PID) {
        *dev_ptr = (char*)q->ws + (size_t)row_of[field_id] * q->N * esz;
        if (rows) *rows = rows_of[field_id];
    } else if (field_id == QENV_FIELD_ROBUST_PARAMS || field_id == QENV_FIELD_GUST) {
        *dev_ptr = (char*)q->ws + (size_t)(field_id == QENV_FIELD_GUST ? ROW_GUST : ROW_ROB) * q->N * esz;
        if (rows) *rows = field_id == QENV_FIELD_GUST ? 6 : 12;
    } else if (field_id == QENV_FIELD_OUTER) {
        *dev_ptr = (char*)q->ws + (size_t)ROW_OUTER * q->N * esz;
        if (rows) *rows = 9;
    } else if (field_id == QENV_FIELD_LAST_RETURN) {
        *dev_ptr = (char*)q->ws + (size_t)ROW_LAST_RETURN * q->N * esz;
        if (rows) *rows = 1;
    } else if (field_id == QENV_FIELD_GUST_EVALS) {
        *dev_ptr = q->wsi + IROW_GUST_K * q->N;
        if (rows) *rows = 1;
    } else if (field_id == QENV_FIELD_I || field_id == QENV_FIELD_FLAGS || field_id == QENV_FIELD_EPISODES) {
        *dev_ptr = q->wsi + (field_id == QENV_FIELD_I ? IROW_I : field_id == QENV_FIELD_FLAGS ? IROW_FLAGS : IROW_EPISODES) * q->N;
        if (rows) *rows = 1;
    } else if (field_id == QENV_FIELD_WINDOW) {
        *dev_ptr = q->window;
        if (rows) *rows = 15 * q->cfg.T;
    } else {
        return fail(QENV_ERR_ARG, "qenv_field: unknown field id");
    }
    return QENV_OK;
}

// ---- placement: the envs of the batch in another order (qenv.h, ABI 10) ----------------------------------------------------
// every per-env row of the workspace moves: slot p receives what slot order[p] held -- state, bookkeeping (also what a reset
// does not touch: prev_ang, ang_vel, the PID and outer-loop states, the banked returns, the episode counters), the robust
// plant's parameters and gusts, the flags, the observation ring.  The compacted reset lists hold slot numbers: a reset in
// progress (a masked warm-up waiting on the side stream) is refused.
int qenv_permute_envs(qenv_t q, const int32_t* order, void* scratch, size_t scratch_bytes) {
    if (!q || !order || !scratch) return fail(QENV_ERR_ARG, "qenv_permute_envs: null argument");
    if (scratch_bytes < q->ws_bytes) return fail(QENV_ERR_ARG, "qenv_permute_envs: scratch needs qenv_workspace_bytes() bytes");
    if (q->ar_pending || q->compact_valid)
        return fail(QENV_ERR_ARG, "qenv_permute_envs: a masked reset is in progress (its lists hold slot numbers): qenv_ppo_flush / finish the warm-up first");
    if (q->sens.ws || q->outer.kind != QENV_OUTER_NONE || q->robust)
        return fail(QENV_ERR_ARG, "qenv_permute_envs: the handle has per-env state outside its workspace (an attached sensor, an outer loop's "
                                  "table, a robust plant's gust source): detach it first");
    const int64_t N = q->N;
    const WsLayout L = ws_layout(q->cfg);
    const size_t esz = q->cfg.dtype == QENV_F64 ? 8 : 4;
    char* ws = (char*)q->ws; char* sc = (char*)scratch;
    hipStream_t st = q->stream;
    // is it a permutation?  (a wrong list would duplicate some envs and lose others, silently)
    uint8_t* seen = (uint8_t*)sc; int32_t* bad = (int32_t*)(sc + align256((size_t)N));
    hipError_t e = hipMemsetAsync(sc, 0, align256((size_t)N) + 256, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(qenv_permute_envs)");
    const dim3 g1((unsigned)((N + 255) / 256)), b(256);
    hipLaunchKernelGGL(k_check_permutation, g1, b, 0, st, order, N, seen, bad);
    hipLaunchKernelGGL(k_count_unseen, g1, b, 0, st, (const uint8_t*)seen, N, bad);
    int32_t nbad = 0;
    e = hipMemcpyAsync(&nbad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail_hip(e, "qenv_permute_envs: permutation check");
    if (nbad) return fail(QENV_ERR_ARG, "qenv_permute_envs: order is not a permutation of 0 .. n_envs - 1");
    auto gather = [&](size_t off, int rows, size_t elem) {
        const dim3 g((unsigned)((N + 255) / 256), (unsigned)(rows < 16 ? rows : 16));
        if (elem == 8) hipLaunchKernelGGL((k_gather_rows<double>), g, b, 0, st, (double*)(sc + off), (const double*)(ws + off), order, rows, N);
        else if (elem == 4) hipLaunchKernelGGL((k_gather_rows<float>), g, b, 0, st, (float*)(sc + off), (const float*)(ws + off), order, rows, N);
        else hipLaunchKernelGGL((k_gather_rows<uint8_t>), g, b, 0, st, (uint8_t*)(sc + off), (const uint8_t*)(ws + off), order, rows, N);
        return hipMemcpyAsync(ws + off, sc + off, (size_t)rows * N * elem, hipMemcpyDeviceToDevice, st);
    };
    if ((e = gather(0, ROWS_T, esz)) != hipSuccess) return fail_hip(e, "qenv_permute_envs: rows");
    // the int rows except the two reset lists (slot numbers, not in use: checked above)
    if ((e = gather(L.int_off + (size_t)IROW_I * N * 4, 3, 4)) != hipSuccess) return fail_hip(e, "qenv_permute_envs: int rows");
    if ((e = gather(L.int_off + (size_t)IROW_GUST_K * N * 4, 1, 4)) != hipSuccess) return fail_hip(e, "qenv_permute_envs: int rows");
    if ((e = gather(L.flag_off, 1, 1)) != hipSuccess) return fail_hip(e, "qenv_permute_envs: flags");
    if ((e = gather(L.flag_off + L.flag_stride, 1, 1)) != hipSuccess) return fail_hip(e, "qenv_permute_envs: flags");
    if ((e = gather(L.win_off, 15 * q->cfg.T, 4)) != hipSuccess) return fail_hip(e, "qenv_permute_envs: window");
    return check_launch("k_gather_rows");
}

int qenv_outer_velpid(qenv_t q, const void* table, int32_t n_rows) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_outer_velpid: null handle");
    if (table && n_rows < 1) return fail(QENV_ERR_ARG, "qenv_outer_velpid: a table needs n_rows >= 1");
    q->outer.kind = table ? QENV_OUTER_VELPID : QENV_OUTER_NONE;
    q->outer.target = table; q->outer.rows = table ? n_rows : 0;
    return QENV_OK;
}

int qenv_outer_pospi(qenv_t q, const void* targets, double P, double D, double I) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_outer_pospi: null handle");
    q->outer.kind = targets ? QENV_OUTER_POSPI : QENV_OUTER_NONE;
    q->outer.target = targets; q->outer.rows = 0;
    q->outer.gain[0] = P; q->outer.gain[1] = D; q->outer.gain[2] = I;
    return QENV_OK;
}

int qenv_attach_sensor(qenv_t q, void* sens, const void* z27, uint64_t seed, int64_t env_offset, int error, const double* params,
                       void* out24, uint64_t calls) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_attach_sensor: null handle");
    q->sens = qenv_sensor_link{};
    if (!sens) return QENV_OK;  // detach
    q->sens.ws = sens; q->sens.z27 = z27; q->sens.out = out24; q->sens.seed = seed; q->sens.calls = calls;
    q->sens.env_offset = env_offset; q->sens.error = error ? 1 : 0; q->sens.has_params = params ? 1 : 0;
    if (params) for (int k = 0; k < 8; ++k) q->sens.params[k] = params[k];
    return QENV_OK;
}

int qenv_outer_reset(qenv_t q, const uint8_t* mask) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_outer_reset: null handle");
    const dim3 g = grid_for(q->N), b(256);
    if (q->cfg.dtype == QENV_F64) hipLaunchKernelGGL((k_outer_reset<double>), g, b, 0, q->stream, (double*)q->ws, q->N, mask);
    else hipLaunchKernelGGL((k_outer_reset<float>), g, b, 0, q->stream, (float*)q->ws, q->N, mask);
    return check_launch("k_outer_reset");
}

int qenv_set_ppo_fused(qenv_t q, int enabled) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_set_ppo_fused: null handle");
    q->ppo_fused = enabled ? 1 : 0;
    if (enabled == 5 || enabled == 6) q->ppo_fused_block = enabled;   // a plain 0 / 1 leaves a forced block as it is
    return QENV_OK;
}

int qenv_set_ppo_fused_block(qenv_t q, int block_envs) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_set_ppo_fused_block: null handle");
    if (block_envs != 0 && block_envs != 32 && block_envs != 64)
        return fail(QENV_ERR_ARG, "qenv_set_ppo_fused_block: 0 (chosen by the batch size), 32 or 64 envs per wavefront");
    q->ppo_fused_block = block_envs == 32 ? 5 : block_envs == 64 ? 6 : 0;   // on / off stays as it is
    return QENV_OK;
}

int qenv_get_ppo_fused(qenv_t q) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_get_ppo_fused: null handle");
    return (q->ppo_fused ? 1 : 0) | ((q->ppo_fused_block ? (1 << q->ppo_fused_block) : 0) << 8);
}

int qenv_set_robust(qenv_t q, int enabled) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_set_robust: null handle");
    if (enabled && q->cfg.variant != QENV_MAIN)
        return fail(QENV_ERR_ARG, "qenv_set_robust: only the main variant has robust_control (quadrotor_env.py:84-109)");
    if (enabled && q->cfg.compute == QENV_COMPUTE_LEAN)
        return fail(QENV_ERR_ARG, "qenv_set_robust: the lean fp64 handle has no robust_control kernels; create the handle as f32, "
                                  "mix (fp64 rows, QENV_COMPUTE_F32) or IEEE f64 (compute default)");
    q->robust = enabled ? 1 : 0;
    return QENV_OK;
}

int qenv_robust_params(qenv_t q, const void* params, uint64_t seed, int64_t env_offset, const uint8_t* mask) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_robust_params: null handle");
    const dim3 g = grid_for(q->N), b(256);
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_robust_params<double>), g, b, 0, q->stream, (double*)q->ws, q->wsi, q->N, (const double*)params,
                           seed, env_offset, mask);
    else
        hipLaunchKernelGGL((k_robust_params<float>), g, b, 0, q->stream, (float*)q->ws, q->wsi, q->N, (const float*)params,
                           seed, env_offset, mask);
    return check_launch("k_robust_params");
}

int qenv_robust_gusts(qenv_t q, const void* table, int32_t E, int32_t K, uint64_t seed, int64_t env_offset) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_robust_gusts: null handle");
    if (table && (E < 1 || K < 1)) return fail(QENV_ERR_ARG, "qenv_robust_gusts: a table needs E >= 1 and K >= 1");
    q->gust.table = table;
    q->gust.f64 = q->cfg.dtype == QENV_F64;
    q->gust.E = table ? E : 1; q->gust.K = table ? K : 1;
    q->gust.key0 = (uint32_t)seed ^ 0x0B57u; q->gust.key1 = (uint32_t)(seed >> 32) ^ 0x6057u;
    q->gust.env_offset = env_offset;
    return QENV_OK;
}

int qenv_reset(qenv_t q, const void* init_states, const uint8_t* mask, void* out_states, void* out_actions,
               int push_window) {
    if (!q || !init_states) return fail(QENV_ERR_ARG, "qenv_reset: null argument");
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    return QENV_DISPATCH(q, do_reset, q, init_states, mask, out_states, out_actions, push_window, nullptr);
}

int qenv_reset_sampled(qenv_t q, uint64_t seed, uint64_t episode, int64_t env_offset, const uint8_t* mask, int push_window) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_reset_sampled: null handle");
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    const ResetSample smp{seed, episode, env_offset, 1};
    return QENV_DISPATCH(q, do_reset, q, nullptr, mask, nullptr, nullptr, push_window, &smp);
}

int qenv_step(qenv_t q, const void* actions, void* out_obs, void* out_reward, uint8_t* out_done, void* out_mask,
              int flags) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_step: null handle");
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    const int mode = flags & QENV_STEP_MODE_MASK;
    if (flags < 0 || (flags & ~(QENV_STEP_MODE_MASK | QENV_STEP_PUSH_WINDOW)) || mode > QENV_STEP_NO_AUX)
        return fail(QENV_ERR_ARG, "qenv_step: bad flags");
    const bool push = (flags & QENV_STEP_PUSH_WINDOW) != 0;
    if (push && mode == QENV_STEP_DYNAMICS_ONLY)
        return fail(QENV_ERR_STATE, "qenv_step: PUSH_WINDOW needs a full step (the slice holds q_dot)");
    const int slot = push ? q->win_head : -1;
    const int rc = QENV_DISPATCH(q, do_step, q, actions, nullptr, out_obs, out_reward, out_done, out_mask, mode, slot, nullptr);
    if (push && rc == 0) q->win_head = (q->win_head + 1) % q->cfg.T;
    return rc;
}

}  // extern "C"

// ---- overlapped PPO rollout step -------------------------------------------------------------------------------
// warm-up rollout (T steps, window refilled) of the envs on reset list `parity`, on q->stream
static int ar_warmup(qenv_s* q, int parity) {
    q->compact_valid = parity ? 2 : 1;
    q->count_used = parity;
    q->warmup_push = 1;
    const int keep = q->packed;
    q->packed = 0;
    const int rc = QENV_DISPATCH(q, do_rollout, q, QENV_CTRL_CONST, q->cfg.T, nullptr, 0, QENV_STEP_FULL, nullptr);
    q->packed = keep;
    q->warmup_push = 0;
    q->compact_valid = 0;
    return rc;
}

// k_ppo_fused (policy + env step of a whole rollout in one launch, qenv_env_impl.hpp) covers the reference's controller
// (75-128-128-4, T = 5) on the scipy-exact integrator, without the per-env extras the separate calls support.
// qenv_set_ppo_fused(q, 0) / QENV_PPO_FUSED=0 keep the two launches per step (A/B runs, and what the fused launch is tested against).
static bool ppo_fused_ok(const qenv_s* q, int k_in, int hidden) {
    return q->ppo_fused && hidden == 128 && k_in == 75 && q->cfg.T == 5 && q->cfg.integrator == QENV_RK45 && q->cfg.direct_control && !q->robust &&
           !q->sens.ws && q->outer.kind == QENV_OUTER_NONE;
}
static const void* fused_image(const qenv_s* q, const float* W1) {
    for (int h = 0; h < 2; ++h)
        if (q->prep[h].image && q->prep[h].W1 == W1 && q->prep[h].n_out == 4) return q->prep[h].image;
    return nullptr;
}

// The argument and state checks of a policy-in-the-loop step: ONE list for qenv_ppo_step (every schedule) and for the fused
// launch behind qenv_ppo_rollout, so the two cannot drift apart as checks are added.
static int ppo_args_check(const qenv_s* q, const qenv_ppo_step_args* p) {
    if (!p->W1 || !p->b1 || !p->W2 || !p->b2 || !p->W3 || !p->b3 || !p->out_action) return fail(QENV_ERR_ARG, "qenv_ppo_step: null argument");
    if (!q->cfg.direct_control) return fail(QENV_ERR_STATE, "qenv_ppo_step: the policy commands the four rotors (direct_control=1)");
    if (q->robust) return fail(QENV_ERR_STATE, "qenv_ppo_step: robust_control episodes redraw their plant at reset; use the separate calls");
    if (q->sens.ws) return fail(QENV_ERR_STATE, "qenv_ppo_step: a sensor is attached (qenv_attach_sensor); the fused call neither evaluates nor resets it -- use the separate calls");
    if (q->outer.kind != QENV_OUTER_NONE) return fail(QENV_ERR_STATE, "qenv_ppo_step: an outer position loop is active; its state is not reset with the episode here -- use the separate calls");
    if ((p->obs_rows_from != 0 || p->obs_slices != 0) && !p->inband)
        return fail(QENV_ERR_ARG, "qenv_ppo_step: the slice layout of stored observations needs the in-band schedule (one window slice per env and step)");
    if (p->obs_rows_from < 0 || p->obs_rows_from >= 15 * q->cfg.T) return fail(QENV_ERR_ARG, "qenv_ppo_step: obs_rows_from must be in [0, 15*T)");
    if (!(p->std > 0)) return fail(QENV_ERR_ARG, "qenv_policy_act: std must be positive");
    if (p->k_in != 15 * q->cfg.T) return fail(QENV_ERR_ARG, "qenv_policy_act: k_in (row length of W1) must be 15*T of the handle");
    return QENV_OK;
}

int qenv::ppo_flush(qenv_s* q) {
    if (!q->ar_pending) return QENV_OK;
    q->ar_pending = 0;
    return ar_warmup(q, q->ar_parity);
}

extern "C" {

int qenv_ppo_flush(qenv_t q) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_ppo_flush: null handle");
    return ppo_flush(q);
}

int qenv_ppo_step(qenv_t q, const qenv_ppo_step_args* p) {
    if (!q || !p) return fail(QENV_ERR_ARG, "qenv_ppo_step: null argument");
    if (const int crc = ppo_args_check(q, p)) return crc;
    // the side-stream schedule is opt-in: measured on config 5 (65 536 envs, random-init policy: 1.8 % of the envs finish
    // per step) it is SLOWER than the plain sequence, 234 vs 162 us per step -- the warm-up wavefronts run 1.5x longer when
    // they share their SIMDs with the policy forward, the gathered forward of the reset envs costs 65 us and the two
    // streams meet twice per step; with few resets per step (episodes that run their 500 steps) the side chain is shorter
    // than the main one and it pays (tests/tools/ab_ppo_overlap.py measures both and checks bit-identity)
    if (p->inband) {
        // in-band warm-up: policy + step, the restart in the step kernel's epilogue, the warm-up steps ride the next T steps
        if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
        // (k_ppo_fused with ONE step was measured for this call: 60 us against 53 for the two launches at 65 536 envs -- the
        //  159 KiB of weights per workgroup and the 75 window rows are staged for a single step; it starts at qenv_ppo_rollout)
        StepExtra exi{};
        exi.ar_on = 2; exi.ar_seed = p->seed; exi.ar_episode = p->step; exi.ar_env_offset = p->env_offset;
        exi.ar_key_episode = p->key_by_episode; exi.out_valid = p->out_valid;
        int rci = policy_act_subset(q, p->k_in, p->hidden, p->W1, p->b1, p->W2, p->b2, p->W3, p->b3, p->std, p->noise, p->seed, p->step,
                                    p->env_offset, p->out_mean, p->out_action, p->out_logprob, p->obs_store, nullptr, nullptr, nullptr,
                                    p->obs_rows_from);
        if (rci == QENV_OK)
            rci = QENV_DISPATCH(q, do_step, q, p->out_action, nullptr, nullptr, p->out_reward, p->out_done, p->out_mask, QENV_STEP_NO_AUX,
                                q->win_head, &exi);
        if (rci == QENV_OK) q->win_head = (q->win_head + 1) % q->cfg.T;
        return rci;
    }
    static const int want_overlap = getenv("QENV_PPO_OVERLAP") ? atoi(getenv("QENV_PPO_OVERLAP")) : 0;
    const bool overlap = want_overlap && (p->hidden == 64 || p->hidden == 128) && q->cfg.T == 5;
    const int64_t N = q->N;
    const int cur = q->ar_parity, next = cur ^ 1;
    StepExtra ex{};
    ex.ar_on = 1; ex.ar_seed = p->seed; ex.ar_episode = p->step; ex.ar_env_offset = p->env_offset;
    ex.ar_key_episode = p->key_by_episode; ex.out_valid = p->out_valid;
    ex.ar_idx = q->wsi + (next ? IROW_IDX2 : IROW_IDX) * N;
    ex.ar_count = q->counters + next;
    ex.ar_flags_next = q->env_flags[next];
    hipStream_t M = q->stream;
    // list `next` was walked two steps ago: its length restarts at 0 before this step's kernel appends to it.  In the plain
    // sequence the policy forward that precedes the step kernel on the stream clears the word; with the side stream two
    // step kernels append concurrently, so the clear is its own operation ahead of the fork.
    hipError_t he = hipSuccess;
    if (q->ar_pending && overlap) {
        he = hipMemsetAsync(q->counters + next, 0, sizeof(int32_t), M);
        if (he != hipSuccess) return fail_hip(he, "hipMemsetAsync(reset list length)");
    } else {
        q->zero_on_act = q->counters + next;
    }
    const int slot = q->win_head;
    const int mode = QENV_STEP_NO_AUX;
    int rc = QENV_OK;
    const int32_t* list = q->wsi + (cur ? IROW_IDX2 : IROW_IDX) * N;
    const int32_t* list_n = q->counters + cur;
    auto act = [&](const uint8_t* skip, const int32_t* li, const int32_t* ln) {
        return policy_act_subset(q, p->k_in, p->hidden, p->W1, p->b1, p->W2, p->b2, p->W3, p->b3, p->std, p->noise, p->seed, p->step,
                                 p->env_offset, p->out_mean, p->out_action, p->out_logprob, p->obs_store, skip, li, ln, p->obs_rows_from);
    };
    auto step = [&](const uint8_t* skip, const int32_t* li, const int32_t* ln) {
        ex.env_skip = skip; ex.list_idx = li; ex.list_count = ln;
        return QENV_DISPATCH(q, do_step, q, p->out_action, nullptr, nullptr, p->out_reward, p->out_done, p->out_mask, mode, slot, &ex);
    };
    if (q->ar_pending && overlap) {
        if (!q->side) {
            if ((he = hipStreamCreateWithFlags(&q->side, hipStreamNonBlocking)) != hipSuccess) return fail_hip(he, "hipStreamCreate(side)");
            if ((he = hipEventCreateWithFlags(&q->ev_main, hipEventDisableTiming)) != hipSuccess) return fail_hip(he, "hipEventCreate");
            if ((he = hipEventCreateWithFlags(&q->ev_side, hipEventDisableTiming)) != hipSuccess) return fail_hip(he, "hipEventCreate");
        }
        // side stream: the reset envs' T warm-up steps, then their policy step (dense tiles gathered from the list)
        (void)hipEventRecord(q->ev_main, M);
        (void)hipStreamWaitEvent(q->side, q->ev_main, 0);
        q->stream = q->side;
        rc = ar_warmup(q, cur);
        if (rc == QENV_OK) rc = act(nullptr, list, list_n);
        if (rc == QENV_OK) rc = step(nullptr, list, list_n);
        (void)hipEventRecord(q->ev_side, q->side);
        q->stream = M;
        // main stream, meanwhile: every other env
        if (rc == QENV_OK) rc = act(q->env_flags[cur], nullptr, nullptr);
        if (rc == QENV_OK) rc = step(q->env_flags[cur], nullptr, nullptr);
        (void)hipStreamWaitEvent(M, q->ev_side, 0);
    } else {
        if (q->ar_pending) rc = ar_warmup(q, cur);
        if (rc == QENV_OK) rc = act(nullptr, nullptr, nullptr);
        if (rc == QENV_OK) rc = step(nullptr, nullptr, nullptr);
    }
    q->zero_on_act = nullptr;  // consumed by act(); cleared here too in case an earlier launch of the sequence failed
    if (rc != QENV_OK) return rc;
    q->win_head = (q->win_head + 1) % q->cfg.T;
    q->ar_pending = 1;
    q->ar_parity = next;
    return QENV_OK;
}

int qenv_reward_function(qenv_t q, int64_t n, const void* state, const void* ang, const void* action, const void* prev_shaping,
                         const int32_t* i, const uint8_t* done_in, void* out_reward, uint8_t* out_done, int32_t* out_solved,
                         void* out_shaping) {
    if (!q || !state || !ang || !action || !prev_shaping || !i || !done_in || !out_reward || !out_done || !out_solved || !out_shaping)
        return fail(QENV_ERR_ARG, "qenv_reward_function: null argument");
    if (n <= 0) return n == 0 ? (int)QENV_OK : fail(QENV_ERR_ARG, "qenv_reward_function: negative n");
    const void* in[6] = {state, ang, action, prev_shaping, i, done_in};
    void* out[4] = {out_reward, out_done, out_solved, out_shaping};
    return QENV_DISPATCH(q, do_method, q, 0, n, in, out);
}

int qenv_mix(qenv_t q, int64_t n, const void* action, void* out_w, void* out_fm, void* out_effort) {
    if (!q || !action || !out_w || !out_fm || !out_effort) return fail(QENV_ERR_ARG, "qenv_mix: null argument");
    if (n <= 0) return n == 0 ? (int)QENV_OK : fail(QENV_ERR_ARG, "qenv_mix: negative n");
    const void* in[1] = {action};
    void* out[3] = {out_w, out_fm, out_effort};
    return QENV_DISPATCH(q, do_method, q, 1, n, in, out);
}

int qenv_ppo_rollout(qenv_t q, const qenv_ppo_step_args* first, int n_steps) {
    if (!q || !first) return fail(QENV_ERR_ARG, "qenv_ppo_rollout: null argument");
    if (n_steps < 0) return fail(QENV_ERR_ARG, "qenv_ppo_rollout: negative n_steps");
    if (n_steps == 0) return QENV_OK;
    if (const int crc = ppo_args_check(q, first)) return crc;   // up front, whichever path runs: the per-step calls would say the same at step 0
    if (first->inband && ppo_fused_ok(q, first->k_in, first->hidden)) {
        // the whole horizon as ONE launch (k_ppo_fused)
        if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
        q->zero_on_act = nullptr;   // (the word a list-schedule step asks its policy launch to clear: no list in the in-band schedule)
        return QENV_DISPATCH(q, do_ppo_fused, q, first, n_steps, 0, fused_image(q, first->W1));
    }
    const size_t esz = q->cfg.dtype == QENV_F64 ? 8 : 4;
    const size_t N = (size_t)q->N;
    auto adv = [&](const void* p, size_t bytes_per_step, int t) -> void* {
        return p ? (void*)((char*)const_cast<void*>(p) + bytes_per_step * (size_t)t) : nullptr;
    };
    for (int t = 0; t < n_steps; ++t) {
        qenv_ppo_step_args a = *first;
        a.step = first->step + (uint64_t)t;
        a.noise = adv(first->noise, 4 * N * esz, t);
        a.out_action = adv(first->out_action, 4 * N * esz, t);
        a.out_logprob = adv(first->out_logprob, 4 * N * esz, t);
        if (first->obs_slices) {   // one 15-float slice per step (see qenv.h): step 0 as the caller set it up, then the newest slice only
            a.obs_store = (float*)adv(first->obs_store, (size_t)15 * N * sizeof(float), t);
            if (t > 0) a.obs_rows_from = 15 * (q->cfg.T - 1);
        } else {
            a.obs_store = (float*)adv(first->obs_store, (size_t)15 * q->cfg.T * N * sizeof(float), t);
        }
        a.out_reward = adv(first->out_reward, N * esz, t);
        a.out_mask = adv(first->out_mask, N * esz, t);
        a.out_valid = adv(first->out_valid, N * esz, t);
        const int rc = qenv_ppo_step(q, &a);
        if (rc != QENV_OK) return rc;
    }
    return QENV_OK;
}

int qenv_policy_rollout(qenv_t q, int k_in, int hidden, const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* W3, const float* b3, int n_steps, void* out_actions, void* out_reward, uint8_t* out_done,
                        void* out_mask) {
    if (!q || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out_actions) return fail(QENV_ERR_ARG, "qenv_policy_rollout: null argument");
    if (n_steps < 0) return fail(QENV_ERR_ARG, "qenv_policy_rollout: negative n_steps");
    if (k_in != 15 * q->cfg.T) return fail(QENV_ERR_ARG, "qenv_policy_rollout: k_in (row length of W1) must be 15*T of the handle");
    if (!q->cfg.direct_control) return fail(QENV_ERR_STATE, "qenv_policy_rollout: the policy commands the four rotors (direct_control=1)");
    if (n_steps == 0) return QENV_OK;
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    const size_t esz = q->cfg.dtype == QENV_F64 ? 8 : 4, N = (size_t)q->N;
    if (ppo_fused_ok(q, k_in, hidden)) {
        qenv_ppo_step_args a{};
        a.k_in = k_in; a.hidden = hidden; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
        a.out_action = out_actions; a.out_reward = out_reward; a.out_done = out_done; a.out_mask = out_mask;
        return QENV_DISPATCH(q, do_ppo_fused, q, &a, n_steps, 1, fused_image(q, W1));
    }
    for (int t = 0; t < n_steps; ++t) {   // the same as separate calls
        void* act = (char*)out_actions + 4 * N * esz * (size_t)t;
        int rc = qenv_mlp_forward(q, k_in, hidden, 4, 1, W1, b1, W2, b2, W3, b3, act, nullptr);
        if (rc == QENV_OK)
            rc = qenv_step(q, act, nullptr, out_reward ? (char*)out_reward + N * esz * (size_t)t : nullptr, out_done,
                           out_mask ? (char*)out_mask + N * esz * (size_t)t : nullptr, QENV_STEP_NO_AUX | QENV_STEP_PUSH_WINDOW);
        if (rc != QENV_OK) return rc;
    }
    return QENV_OK;
}

int qenv_rollout(qenv_t q, int controller, int n_steps, const double* const_action, int pid_first_action, int flags) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_rollout: null handle");
    if (controller < QENV_CTRL_CONST || controller > QENV_CTRL_PIDPOS) return fail(QENV_ERR_ARG, "qenv_rollout: bad controller");
    if (n_steps < 0) return fail(QENV_ERR_ARG, "qenv_rollout: negative n_steps");
    if (flags < QENV_STEP_FULL || flags > QENV_STEP_NO_AUX) return fail(QENV_ERR_ARG, "qenv_rollout: bad flags");
    if (flags == QENV_STEP_DYNAMICS_ONLY && controller != QENV_CTRL_CONST)
        return fail(QENV_ERR_STATE, "qenv_rollout: DYNAMICS_ONLY needs the constant-action controller");
    if (controller != QENV_CTRL_CONST && q->cfg.direct_control)
        return fail(QENV_ERR_STATE, "qenv_rollout: LQR/PID command [F,Mx,My,Mz]; create the handle with direct_control=0");
    if (n_steps == 0) return QENV_OK;
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    return QENV_DISPATCH(q, do_rollout, q, controller, n_steps, const_action, pid_first_action, flags, nullptr);
}

int qenv_rollout_track(qenv_t q, int n_steps, const void* targets, int flags) {
    if (!q || !targets) return fail(QENV_ERR_ARG, "qenv_rollout_track: null argument");
    if (n_steps < 0) return fail(QENV_ERR_ARG, "qenv_rollout_track: negative n_steps");
    if (flags != QENV_STEP_FULL && flags != QENV_STEP_NO_AUX) return fail(QENV_ERR_ARG, "qenv_rollout_track: bad flags");
    if (q->cfg.direct_control) return fail(QENV_ERR_STATE, "qenv_rollout_track: the PD commands [F,Mx,My,Mz]; direct_control=0");
    if (n_steps == 0) return QENV_OK;
    return QENV_DISPATCH(q, do_rollout, q, QENV_CTRL_PIDPOS, n_steps, nullptr, 0, flags, targets);
}

int qenv_lqr_action(qenv_t q, void* out_actions) {
    if (!q || !out_actions) return fail(QENV_ERR_ARG, "qenv_lqr_action: null argument");
    const dim3 g = grid_for(q->N), b(256);
    if (q->cfg.dtype == QENV_F64 && q->cfg.compute == QENV_COMPUTE_LEAN)  // the same elementary functions as the fused rollout
        hipLaunchKernelGGL((k_lqr_action<double, MthLean>), g, b, 0, q->stream, (const double*)q->ws, q->N,
                           make_lqr<double>(q->cfg.clipped != 0), (double*)out_actions);
    else if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_lqr_action<double>), g, b, 0, q->stream, (const double*)q->ws, q->N,
                           make_lqr<double>(q->cfg.clipped != 0), (double*)out_actions);
    else
        hipLaunchKernelGGL((k_lqr_action<float>), g, b, 0, q->stream, (const float*)q->ws, q->N,
                           make_lqr<float>(q->cfg.clipped != 0), (float*)out_actions);
    return check_launch("k_lqr_action");
}

int qenv_pid_action(qenv_t q, const double* targets, void* out_actions) {
    if (!q || !out_actions) return fail(QENV_ERR_ARG, "qenv_pid_action: null argument");
    const dim3 g = grid_for(q->N), b(256);
    double t[4] = {0, 0, 0, 0};
    if (targets) for (int k = 0; k < 4; ++k) t[k] = targets[k];
    if (q->cfg.dtype == QENV_F64 && q->cfg.compute == QENV_COMPUTE_LEAN)
        hipLaunchKernelGGL((k_pid_action<double, MthLean>), g, b, 0, q->stream, (double*)q->ws, q->N,
                           make_pid<double>(q->cfg.clipped != 0), t[0], t[1], t[2], t[3], q->cfg.t_step, (double*)out_actions);
    else if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_pid_action<double>), g, b, 0, q->stream, (double*)q->ws, q->N,
                           make_pid<double>(q->cfg.clipped != 0), t[0], t[1], t[2], t[3], q->cfg.t_step, (double*)out_actions);
    else
        hipLaunchKernelGGL((k_pid_action<float>), g, b, 0, q->stream, (float*)q->ws, q->N,
                           make_pid<float>(q->cfg.clipped != 0), (float)t[0], (float)t[1], (float)t[2], (float)t[3],
                           (float)q->cfg.t_step, (float*)out_actions);
    return check_launch("k_pid_action");
}

int qenv_pidpos_action(qenv_t q, const double* targets, void* out_actions) {
    if (!q || !out_actions) return fail(QENV_ERR_ARG, "qenv_pidpos_action: null argument");
    const dim3 g = grid_for(q->N), b(256);
    if (q->cfg.dtype == QENV_F64) {
        Tgt7<double> tg;
        for (int k = 0; k < 7; ++k) tg.v[k] = targets ? targets[k] : 0.0;
        if (q->cfg.compute == QENV_COMPUTE_LEAN)
            hipLaunchKernelGGL((k_pidpos_action<double, MthLean>), g, b, 0, q->stream, (double*)q->ws, q->N, tg, q->cfg.t_step, (double*)out_actions);
        else
            hipLaunchKernelGGL((k_pidpos_action<double>), g, b, 0, q->stream, (double*)q->ws, q->N, tg, q->cfg.t_step, (double*)out_actions);
    } else {
        Tgt7<float> tg;
        for (int k = 0; k < 7; ++k) tg.v[k] = targets ? (float)targets[k] : 0.f;
        hipLaunchKernelGGL((k_pidpos_action<float>), g, b, 0, q->stream, (float*)q->ws, q->N, tg, (float)q->cfg.t_step, (float*)out_actions);
    }
    return check_launch("k_pidpos_action");
}

int qenv_pid_reset(qenv_t q, const uint8_t* mask) {
    if (!q) return fail(QENV_ERR_ARG, "qenv_pid_reset: null handle");
    const dim3 g = grid_for(q->N), b(256);
    if (q->cfg.dtype == QENV_F64) hipLaunchKernelGGL((k_pid_reset<double>), g, b, 0, q->stream, (double*)q->ws, q->N, mask);
    else hipLaunchKernelGGL((k_pid_reset<float>), g, b, 0, q->stream, (float*)q->ws, q->N, mask);
    return check_launch("k_pid_reset");
}

int qenv_debug_copy_rows(int dtype, void* dst, const void* src, int rows, int64_t n_envs, void* stream) {
    if (!dst || !src || rows <= 0 || n_envs <= 0) return fail(QENV_ERR_ARG, "qenv_debug_copy_rows: bad argument");
    const dim3 g = grid_for(n_envs), b(256);
    if (dtype == QENV_F64)
        hipLaunchKernelGGL((k_copy_rows<double>), g, b, 0, (hipStream_t)stream, (double*)dst, (const double*)src, rows, n_envs);
    else
        hipLaunchKernelGGL((k_copy_rows<float>), g, b, 0, (hipStream_t)stream, (float*)dst, (const float*)src, rows, n_envs);
    return check_launch("k_copy_rows");
}

}  // extern "C"
