// qenv_host.hpp -- handle, workspace layout and error plumbing shared by the .hip units.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/qenv.h"
#include "qenv_core.hpp"

namespace qenv {

// Workspace rows of the handle's dtype (row r of env e at (r*N + e)).
enum {
    ROW_STATE = 0,          // 13
    ROW_ANG = 13,           // 3
    ROW_PREV_ANG = 16,      // 3
    ROW_ANG_VEL = 19,       // 3
    ROW_VQ = 22,            // 4
    ROW_W = 26,             // 4
    ROW_EFFORT = 30,        // 4
    ROW_CLIPPED = 34,       // 4
    ROW_ACCEL = 38,         // 3
    ROW_PREV_SHAPING = 41,  // 1
    ROW_REWARD = 42,        // 1
    ROW_ABS_SUM = 43,       // 1
    ROW_EP_RETURN = 44,     // 1
    ROW_PID = 45,           // 15
    ROW_LAST_RETURN = 60,   // 1
    ROW_ROB = 61,           // 12: episode_kf(4), episode_m, episode_ir(4), episode_J diag(3) (robust_control.reset)
    ROW_GUST = 73,          // 6: gust(3), last_gust(3)
    ROW_OUTER = 79,         // 9: outer position loop: integral(3), previous error(3), velocity command(3)
    ROWS_T = 88
};
// IDX: compacted reset list; its two ping-pong length counters live in their own 256-B tail (WsLayout::cnt_off), not in
// a per-env row: a row has only n_envs words, and with n_envs == 1 a second word would alias the next row
// IDX2: second list of the overlapped PPO step (qenv_ppo_step), where step t + 1 appends its finished envs while the
// side stream still walks the list of step t
enum { IROW_I = 0, IROW_FLAGS = 1, IROW_EPISODES = 2, IROW_IDX = 3, IROW_GUST_K = 4, IROW_IDX2 = 5, IROWS = 6 };

struct WsLayout {
    size_t int_off, cnt_off, flag_off, flag_stride, win_off, total;
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline WsLayout ws_layout(const qenv_config& c) {
    const size_t esz = c.dtype == QENV_F64 ? 8 : 4;
    const size_t N = (size_t)c.n_envs;
    WsLayout L;
    L.int_off = align256((size_t)ROWS_T * N * esz);
    L.cnt_off = align256(L.int_off + (size_t)IROWS * N * sizeof(int32_t));
    L.flag_off = L.cnt_off + 256;                   // two byte-per-env flag arrays ("this env was reset in the previous
    L.flag_stride = align256(N);                    //  step and is being warmed up on the side stream": qenv_ppo_step)
    L.win_off = L.flag_off + 2 * L.flag_stride;
    L.total = align256(L.win_off + (size_t)15 * c.T * N * sizeof(float));
    return L;
}

}  // namespace qenv

// where a robust handle's gust draws come from (see Plant<T, true> in qenv_core.hpp)
struct qenv_gust_src {
    const void* table;        // null = Philox; else [E][K][3][N] in the handle's storage type
    int32_t f64, E, K;
    uint32_t key0, key1;
    int64_t env_offset;       // global index of local env 0 (shard-invariant streams)
};

// outer position loop evaluated inside PUSH_WINDOW steps (velocity_pid.py / math_trajectory.py pos_pi)
struct qenv_outer {
    int32_t kind;             // QENV_OUTER_*
    const void* target;       // VELPID: mission table [rows][7]; POSPI: per-env targets [3][N]; storage dtype
    int32_t rows;
    double gain[3];           // POSPI: P, D, I
};

// sensor fused into the PUSH_WINDOW steps (qenv_attach_sensor)
struct qenv_sensor_link {
    void* ws;                 // [QENV_SENSOR_ROWS][N] sensor state, null = not attached
    const void* z27;          // null = Philox
    void* out;                // null or [24][N]
    uint64_t seed, calls;
    int64_t env_offset;
    int32_t error, has_params;
    double params[8];
};

struct qenv_s {
    qenv_config cfg;
    qenv_outer outer;
    qenv_sensor_link sens;
    int32_t robust;           // env.robust_control
    qenv_gust_src gust;
    int64_t N;
    void* ws;
    size_t ws_bytes;
    int32_t* wsi;
    int32_t* counters;  // [0], [1]: ping-pong lengths of the compacted reset list (see k_reset)
    float* window;
    int32_t win_head;  // ring slot the next push writes
    int32_t warmup_push;    // the masked warm-up in progress also refills the observation ring
    int32_t compact_valid;  // the IDX/COUNT rows hold the compaction of the mask of the reset in progress
    int32_t count_cur, count_used;  // ping-pong list-length counters: the one the next masked reset fills / the one in use
    int32_t packed;    // 1/0: force / forbid the two-envs-per-lane packed-fp32 rollout kernels; -1: automatic
    int32_t ang_ok;    // 1: no dynamics-only step since the last unmasked full step (see euler_update)
    hipStream_t stream;
    // overlapped PPO rollout step (qenv_ppo_step): the reset envs' warm-up runs on a side stream beside the policy step
    // of every 32-env tile that holds no reset env
    hipStream_t side;          // created on first use
    hipEvent_t ev_main, ev_side;
    uint8_t* env_flags[2];     // [parity]: 1 for the envs of reset list `parity`
    int32_t ar_pending;        // a reset list (parity ar_parity) still waits for its warm-up
    int32_t ar_parity;
    int32_t* zero_on_act;      // a word the next policy forward clears in its first thread (see MlpArgs::zero_word)
    // qenv_mlp_prepare: per head (0 actor, 1 critic) the caller-owned image of the weights in k_mlp_v3's LDS layout, used by
    // the forward calls whose W1 pointer and n_out match
    struct { const float* W1; const void* image; int32_t n_out; } prep[2];
    // qenv_ppo_rollout (in-band) / qenv_policy_rollout as ONE launch where k_ppo_fused applies
    // (qenv_set_ppo_fused; initial value: QENV_PPO_FUSED, default 1)
    int32_t ppo_fused;
    int32_t ppo_fused_block;   // 0: chosen by the batch size; 5 / 6: 32- / 64-env blocks forced (qenv_set_ppo_fused_block)
};

namespace qenv {

inline thread_local char g_err[256] = "";

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver call (a few microseconds of host time on every launch of the
// small-batch paths): the opt-in is per function and sticky, so it is made only when a launch needs more than was granted
static inline hipError_t allow_dynamic_lds(const void* fn, size_t bytes) {
    // the attribute belongs to the function ON THE CURRENT DEVICE: the device is part of the key (one thread may drive
    // handles on several GPUs)
    struct Slot { const void* fn; int dev; size_t bytes; };
    static thread_local Slot cache[128] = {};
    static thread_local int n = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    for (int i = 0; i < n; ++i)
        if (cache[i].fn == fn && cache[i].dev == dev) {
            if (bytes <= cache[i].bytes) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) cache[i].bytes = bytes;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && n < 128) cache[n++] = Slot{fn, dev, bytes};
    return e;
}

// compute units of the current device, asked once per device and thread (launch sizing of the persistent kernels)
static inline int device_cu_count() {
    static thread_local int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && cached[dev] > 0) return cached[dev];
    int cus = 0;
    if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 256;
    return cached[dev] = cus;
}

static inline int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
static inline int fail_hip(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return QENV_ERR_HIP;
}
static inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? (int)QENV_OK : fail_hip(e, what);
}

// qenv_env.hip: run the warm-up a qenv_ppo_step left pending (no-op otherwise)
int ppo_flush(qenv_s* q);

// qenv_ppo.hip: qenv_policy_act on a subset of the envs (the overlapped PPO step of qenv_env.hip): either every env whose
// byte in env_skip is 0 (the others are computed with their tile but nothing of theirs is stored), or the envs of a
// compacted list (list_idx[0 .. *list_count), gathered into dense 32-env tiles by the tile-per-workgroup kernel)
int policy_act_subset(qenv_s* q, int k_in, int hidden, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                      const float* b3, double std, const void* noise, uint64_t seed, uint64_t step, int64_t env_offset,
                      void* out_mean, void* out_action, void* out_logprob, float* obs_store, const uint8_t* env_skip,
                      const int32_t* list_idx, const int32_t* list_count, int obs_rows_from = 0);

}  // namespace qenv
