// qenv_ppo.hip -- PPO inner-loop kernels: observation ring, actor/critic MLP forward on the
// fp32 MFMA pipe, GAE reverse scan, advantage normalisation, landing reward.  gfx950 only.
//
// Reference behaviour restated (relative to /root/reference):
//   environment/controller/dl_auxiliary.py:15-32   observation window
//   environment/controller/model.py:27-42          actor / critic MLP
//   environment/controller/ppo.py:125-141, visual_landing/ppo_trainer.py:133-148   GAE
//   visual_landing/rl_reward_fuction.py:3-102      visual_reward
#include "qenv_host.hpp"
#include "qenv_mlp_common.hpp"
#include "qenv_mlp3.hpp"
#include "qenv_sensor.hpp"

namespace qenv {

// ---- observation ring -------------------------------------------------------------------------
// dl_in_gen.dl_input shifts a 15*T window left and appends [action(4), vel(3), q(4), q_dot(4)].
// Here the window is a ring of T slices (row block `slot`), so a push writes 60 B/env and
// nothing is shifted; the MLP kernel reads the slices oldest-first starting at `head`.
template <typename T>
__global__ __launch_bounds__(256) void k_window_push(float* win, int64_t N, int slot, const T* obs, const T* act) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    float* w = win + (int64_t)slot * 15 * N + e;
#pragma unroll
    for (int k = 0; k < 4; ++k) w[(int64_t)k * N] = (float)act[(int64_t)k * N + e];
    w[4 * N] = (float)obs[1 * N + e];
    w[5 * N] = (float)obs[3 * N + e];
    w[6 * N] = (float)obs[5 * N + e];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[(int64_t)(7 + k) * N] = (float)obs[(int64_t)(6 + k) * N + e];
}

// ---- MLP forward on v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains) ----------------------------
// Orientation: D[h][e] = sum_k W[h][k] X[k][e]: A = weights (row = hidden unit), B = activations
// (column = env), so a layer's accumulator tile (column on the lane, rows in the 16 registers)
// is directly the next layer's B operand: register j of lane-half p holds hidden row
// 32 t + (j&3) + 8 (j>>2) + 4 p, and the next layer simply pairs the two rows a register holds
// across the lane halves as one k-step, reading the matching two weight columns.  No LDS
// round trip, no shuffles between layers.  Weights are staged once per workgroup into LDS,
// transposed to [k][h] so the 32 lanes of a half read 32 consecutive floats (conflict-free).

// Diagnostic build only (-DQENV_MLP_STAMPS, tools/mlp_stamps.py): s_memtime stamps of k_mlp's phases per wave.  The product
// build compiles none of it.
#ifdef QENV_MLP_STAMPS
__device__ unsigned long long g_mlp_stamps[4096 * 16];
#define MLP_STAMP(slot)                                                                                             \
    do {                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        const unsigned long long t_ = (slot) >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
        if ((threadIdx.x & 63) == 0 && (slot) < 16)                                                                 \
            g_mlp_stamps[((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 16 + (slot)] = t_;             \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    } while (0)
#else
#define MLP_STAMP(slot) do { } while (0)
#endif

// K1 = 15*T window elements; TW > 0: compile-time window length (the whole observation of a tile
// is prefetched into registers with back-to-back global loads before the first MFMA, and every
// loop is unrolled so hipcc hoists the LDS operand reads ahead of the MFMAs); TW == 0: run-time T.
// BF16L2: layer 2 (the H x H one: 54 % of the fp32 MFMA time at H = 128) on the bf16 matrix cores with every fp32
// operand split into three bf16 terms x = hi + mid + lo (8 + 8 + 8 significant bits) and the six products that matter
// (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid: everything down to 2^-24 of the product) accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16: 6 MFMAs of 32 cycles per 16 k instead of 8 fp32 MFMAs of 64 cycles (tools/ubench): 192
// matrix-pipe cycles instead of 512.  (Neither kind overlaps with another wave's VALU work on this chip: mfma_valu.hip.)
// The accumulator tile is still the next operand without an LDS round trip: one 32x32x16 step contracts the 16 hidden
// rows that registers 8 jh .. 8 jh + 7 of the two lane halves hold (MFMA k index 8 half + i  <->  row acc_row(8 jh + i, half)),
// and the weights are staged in exactly that order, already split, as three 16-byte fragments per lane and step
// (NT^2 * 2 * 3 KiB = 96 KiB at H = 128: with fp32 W1 and W3 152 of the 160 KiB).
// BF16 = 2 does the same for layer 1 (the 75-float window padded to 5 steps of 16: lane (col, half) loads the eight
// window elements 16 s + 8 half + i of its env; 4 * 5 * 3 KiB = 60 KiB of fragments) and keeps only the four real
// output rows of W3 in LDS ([H][4] instead of the zero-padded [H][32]): 157 of the 160 KiB.  Layer 3 stays fp32.
template <int H, int TW, int BLOCK, bool W2_LDS, bool SAMPLE = false, int BF16 = 0>
__global__ __launch_bounds__(BLOCK) void k_mlp(const MlpArgs a) {
    constexpr bool BF16L2 = BF16 >= 1, BF16L1 = BF16 >= 2;
    constexpr int HP = H < 32 ? 32 : H;  // hidden rows padded to the 32-row MFMA tile
    constexpr int NT = HP / 32;
    static_assert(!BF16L2 || (W2_LDS && H % 32 == 0), "the split layer 2 keeps its weights in LDS");
    static_assert(!BF16L1 || TW > 0, "the split layer 1 is built for the compile-time window");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0;
    MLP_STAMP(0); MLP_STAMP(14);
    const int K1 = TW ? 15 * TW : 15 * a.T;
    const int K1P = (K1 + 1) & ~1;
    constexpr int KS16 = (15 * (TW ? TW : 1) + 15) / 16;  // 16-wide k-steps of the split layer 1
    float* W1t = lds;                                   // [K1P][HP]; BF16L1: split fragments
    const size_t W1_WORDS = BF16L1 ? (size_t)NT * KS16 * 3 * 64 * 4 : (size_t)K1P * HP;
    float* W2t = W1t + W1_WORDS;                        // [HP][HP]   (k-major), only if W2_LDS; BF16L2: split fragments
    constexpr size_t W2_WORDS = BF16L2 ? (size_t)NT * NT * 2 * 3 * 64 * 4 : (W2_LDS ? (size_t)HP * HP : 0);
    constexpr int W3C = BF16L1 ? 4 : 32;                // output rows of W3 kept in LDS
    float* W3t = W2t + W2_WORDS;                        // [HP][W3C]  (k-major; output rows >= n_out are zero)
    float* b1s = W3t + W3C * HP;
    float* b2s = b1s + HP;
    float* b3s = b2s + HP;  // [32], zero beyond n_out
    // transposing stage-in, 8 x 8 sub-blocks per wave instruction (lane = 8 hl + kl): the global reads fetch 32 B runs
    // of 8 weight rows (a row-per-lane read would touch 64 different cache lines per instruction, 32 times each over
    // the loop), the LDS writes land 8 lanes per bank (k stride = HP words) -- an 8-way write conflict that costs less
    // than the refetching it replaces, and the [k][h] image keeps the MFMA operand reads conflict-free and unpadded
    {
        constexpr int KL = 8, HL = 64 / KL;  // 4 x 16 and 16 x 4 sub-blocks measured within 1.5 % of this
        const int kl = threadIdx.x % KL, hl = (threadIdx.x / KL) % HL, wv = threadIdx.x >> 6;
        constexpr int NW = BLOCK / 64;
        const int kb1 = (K1P + KL - 1) / KL;
        if constexpr (BF16L1) {
            // fragment (t, s, split): lane (col, half) holds W1[32 t + col][16 s + 8 half + i], i = 0..7 (0 beyond K1)
            uint4* frag = reinterpret_cast<uint4*>(W1t);
            for (int c = threadIdx.x; c < NT * KS16 * 64; c += BLOCK) {
                const int ln = c & 63, sk = (c >> 6) % KS16, t = (c >> 6) / KS16;
                const int k0 = 16 * sk + 8 * (ln >> 5);
                const float* src = a.W1 + (size_t)(32 * t + (ln & 31)) * K1;
                float w8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w8[i] = (k0 + i) < K1 ? src[k0 + i] : 0.f;
                bf16x8 sp[3];
                split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
                for (int k = 0; k < 3; ++k) frag[((t * KS16 + sk) * 3 + k) * 64 + ln] = __builtin_bit_cast(uint4, sp[k]);
            }
        } else
        for (int blk = wv; blk < (HP / HL) * kb1; blk += NW) {
            const int h = (blk / kb1) * HL + hl, k = (blk % kb1) * KL + kl;
            if (k < K1P) W1t[k * HP + h] = (k < K1 && h < H) ? a.W1[(size_t)h * K1 + k] : 0.f;
        }
        if constexpr (BF16L2) {
            // fragment (t, ti, jh, split) = 64 lanes x 16 B: lane (col, half) holds W2[32 t + col][32 ti + acc_row(8 jh + i, half)],
            // i = 0..7 -- two runs of four consecutive floats of one weight row
            uint4* frag = reinterpret_cast<uint4*>(W2t);
            for (int c = threadIdx.x; c < NT * NT * 2 * 64; c += BLOCK) {
                const int ln = c & 63, jh = (c >> 6) & 1, ti = (c >> 7) % NT, t = (c >> 7) / NT;
                const float* src = a.W2 + (size_t)(32 * t + (ln & 31)) * H + 32 * ti + 16 * jh + 4 * (ln >> 5);
                const float4 lo4 = *reinterpret_cast<const float4*>(src), hi4 = *reinterpret_cast<const float4*>(src + 8);
                const float w8[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                bf16x8 sp[3];
                split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
                for (int k = 0; k < 3; ++k) frag[(((t * NT + ti) * 2 + jh) * 3 + k) * 64 + ln] = __builtin_bit_cast(uint4, sp[k]);
            }
        } else if (W2_LDS)
            for (int blk = wv; blk < (HP / HL) * (HP / KL); blk += NW) {
                const int h = (blk / (HP / KL)) * HL + hl, k = (blk % (HP / KL)) * KL + kl;
                W2t[k * HP + h] = (k < H && h < H) ? a.W2[(size_t)h * H + k] : 0.f;
            }
    }
    for (int idx = threadIdx.x; idx < W3C * HP; idx += BLOCK) {
        const int h = idx / W3C, o = idx % W3C;
        W3t[idx] = (o < a.n_out && h < H) ? a.W3[(size_t)o * H + h] : 0.f;
    }
    for (int h = threadIdx.x; h < HP; h += BLOCK) {
        b1s[h] = h < H ? a.b1[h] : 0.f;
        b2s[h] = h < H ? a.b2[h] : 0.f;
    }
    if (threadIdx.x < 32) b3s[threadIdx.x] = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;
    __syncthreads();
    MLP_STAMP(1);

    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int wave = threadIdx.x >> 6, waves_per_block = BLOCK / 64;
    const int64_t tiles_per_chunk = (a.N + 31) / 32;
    const int64_t n_tiles = tiles_per_chunk * a.n_chunks;
    [[maybe_unused]] int stamp_tile = 0;
    for (int64_t gtile = (int64_t)blockIdx.x * waves_per_block + wave; gtile < n_tiles;
         gtile += (int64_t)gridDim.x * waves_per_block) {
        MLP_STAMP(2 + 6 * stamp_tile);
        // The weights in LDS and the ring offsets are invariant across tiles, so LICM would hoist hundreds of
        // LDS reads / offsets out of this loop and spill them; an opaque memory clobber per tile stops that.
        asm volatile("" ::: "memory");
        const int64_t chunk = a.n_chunks > 1 ? gtile / tiles_per_chunk : 0;  // wave-uniform
        const int64_t tile = gtile - chunk * tiles_per_chunk;
        const float* __restrict__ winc = a.win + chunk * (int64_t)K1 * a.N;
        const int64_t out_off = chunk * (int64_t)a.n_out * a.N;
        const int64_t e = tile * 32 + col;
        // an env left to the side stream (env_skip) rides along in its tile, but nothing of it is stored
        const bool valid = e < a.N && !(a.env_skip && a.env_skip[e < a.N ? e : 0]);
        f32x16 acc1[NT];
        // Output tiles are the OUTER loop of both layers, so only one 16-register accumulator chain is
        // being fed at a time (a 32x32x2 f32 MFMA chain issues back to back: its 64-cycle dependent latency
        // equals its issue interval) and the live set stays ~110 VGPRs instead of spilling.
        // layer 1: k-step s covers window elements k = 2s (lanes 0-31) and 2s+1 (lanes 32-63)
        if constexpr (BF16L1) {
            uint32_t N32 = (uint32_t)a.N;
            int hsel = half;
            asm volatile("" : "+s"(N32), "+v"(hsel));
            const uint32_t e32 = (uint32_t)(valid ? e : a.N - 1);
            bf16x8 xsp[KS16][3];
#pragma unroll
            for (int sk = 0; sk < KS16; ++sk) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    constexpr int KW = 15 * (TW ? TW : 1);
                    const int k0 = 16 * sk + i, k1 = 16 * sk + 8 + i;  // this element for lane half 0 / 1
                    int sl0 = a.head + (k0 < KW ? k0 : 0) / 15, sl1 = a.head + (k1 < KW ? k1 : 0) / 15;  // wave-uniform
                    if (sl0 >= TW) sl0 -= TW;
                    if (sl1 >= TW) sl1 -= TW;
                    const uint32_t r0 = (uint32_t)(sl0 * 15 + (k0 < KW ? k0 : 0) % 15);
                    const uint32_t r1 = (uint32_t)(sl1 * 15 + (k1 < KW ? k1 : 0) % 15);
                    float x = winc[(hsel ? r1 : r0) * N32 + e32];
                    if (k1 >= KW) x = (hsel || k0 >= KW) ? 0.f : x;
                    v8[i] = x;
                    if (a.obs_store && valid) {
                        const uint32_t k = (uint32_t)(16 * sk + 8 * half + i);
                        if (k < (uint32_t)KW) a.obs_store[k * N32 + e32] = x;
                    }
                }
                split3x8(v8, xsp[sk][0], xsp[sk][1], xsp[sk][2]);
            }
            MLP_STAMP(3 + 6 * stamp_tile);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = b1s[32 * t + acc_row(j, half)];
                const uint4* frag = reinterpret_cast<const uint4*>(W1t) + (size_t)t * KS16 * 3 * 64 + lane;
#pragma unroll
                for (int sk = 0; sk < KS16; ++sk) {
                    const uint4* f = frag + sk * 3 * 64;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                                 al = __builtin_bit_cast(bf16x8, f[128]);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[sk][1], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xsp[sk][0], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][2], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[sk][0], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][1], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][0], acc1[t], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh(acc1[t][j]);
            }
        } else if (TW) {
            constexpr int KS = (15 * (TW ? TW : 1) + 1) / 2;
            float xs[KS];
            // 32-bit element offsets from the (scalar) ring base, no per-load predicates: out-of-range envs
            // read env N-1 (masked at the final store), the one padded k-step is zeroed by a select
            uint32_t N32 = (uint32_t)a.N;
            int hsel = half;
            asm volatile("" : "+s"(N32), "+v"(hsel));  // keeps the 38 row selects/offsets inside the tile loop
            const uint32_t e32 = (uint32_t)(valid ? e : a.N - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                constexpr int dummy = 0; (void)dummy;
                const int k0 = 2 * s, k1 = 2 * s + 1;
                int sl0 = a.head + k0 / 15, sl1 = a.head + k1 / 15;  // wave-uniform (SALU)
                if (sl0 >= TW) sl0 -= TW;
                if (sl1 >= TW) sl1 -= TW;
                const uint32_t r0 = (uint32_t)(sl0 * 15 + k0 % 15);
                const uint32_t r1 = k1 < 15 * TW ? (uint32_t)(sl1 * 15 + k1 % 15) : r0;
                float x = winc[(hsel ? r1 : r0) * N32 + e32];
                if (k1 >= 15 * TW) x = hsel ? 0.f : x;
                xs[s] = x;
            }
            if (a.obs_store && valid) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const uint32_t k = (uint32_t)(2 * s + half);
                    if (2 * s + 1 < 15 * TW || half == 0) a.obs_store[k * N32 + e32] = xs[s];  // each (k, e) is held by exactly one lane
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = b1s[32 * t + acc_row(j, half)];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1t[(2 * s + half) * HP + 32 * t + col], xs[s], acc1[t], 0, 0, 0);
                    // fence the scheduler every 8 k-steps: without it hipcc hoists ALL the LDS operand reads of
                    // the unrolled loop above the first MFMA and spills
                    if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh(acc1[t][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = b1s[32 * t + acc_row(j, half)];
            for (int s = 0; s < K1P / 2; ++s) {
                const int k = 2 * s + half;
                float x = 0.f;
                if (valid && k < K1) {
                    int slice = a.head + k / 15;
                    if (slice >= a.T) slice -= a.T;
                    x = winc[(int64_t)(slice * 15 + k % 15) * a.N + e];
                    if (a.obs_store) a.obs_store[(int64_t)k * a.N + e] = x;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1t[k * HP + 32 * t + col], x, acc1[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh(acc1[t][j]);
        }
        // layer 2 + layer 3, both on the MFMA, one layer-2 output tile at a time.  Register j of input tile ti IS
        // the B operand of the k-step that pairs hidden rows 32 ti + acc_row(j,0) and 32 ti + acc_row(j,1).  Layer 3
        // (n_out <= 4 rows, padded to one 32-row tile) consumes each finished layer-2 tile the same way: 16 more
        // MFMAs per tile instead of 64 dependent VALU fmas + 16 ds_read_b128 -- on this chip fp32 MFMAs and VALU
        // share the vector lanes, and the VALU form of this layer cost 26 us of a 99 us launch (ablation, A/B).
        MLP_STAMP(4 + 6 * stamp_tile);
        f32x16 acc3;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc3[j] = b3s[acc_row(j, half)];
        // BF16L2: the layer-1 activations of this lane, split once per tile (they are the B operand of every output tile)
        bf16x8 bsp[BF16L2 ? NT : 1][2][3];
        if constexpr (BF16L2) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    float v8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v8[i] = acc1[ti][8 * jh + i];
                    split3x8(v8, bsp[ti][jh][0], bsp[ti][jh][1], bsp[ti][jh][2]);
                }
        }
        MLP_STAMP(5 + 6 * stamp_tile);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 acc2;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = b2s[32 * t + acc_row(j, half)];
            if constexpr (BF16L2) {
                const uint4* frag = reinterpret_cast<const uint4*>(W2t) + (size_t)t * NT * 2 * 3 * 64 + lane;
#pragma unroll
                for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
                        const uint4* f = frag + (ti * 2 + jh) * 3 * 64;
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                                     al = __builtin_bit_cast(bf16x8, f[128]);
                        const bf16x8 bh = bsp[ti][jh][0], bm = bsp[ti][jh][1], bl = bsp[ti][jh][2];
                        // smallest products first
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2, 0, 0, 0);
                    }
            } else {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int k = 32 * ti + acc_row(j, half);
                    const float w = W2_LDS ? W2t[k * HP + 32 * t + col]
                                           : ((32 * t + col) < H ? a.W2[(size_t)(32 * t + col) * H + k] : 0.f);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, acc1[ti][j], acc2, 0, 0, 0);
                    if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = fast_tanh(acc2[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float w3 = BF16L1 ? (col < 4 ? W3t[(32 * t + acc_row(j, half)) * 4 + col] : 0.f)
                                        : W3t[(32 * t + acc_row(j, half)) * 32 + col];
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3, acc2[j], acc3, 0, 0, 0);
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        MLP_STAMP(6 + 6 * stamp_tile);
        float outv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = acc3[o];  // output row o = acc_row(o, 0): held by the lower lane half
            if (a.final_tanh) v = tanhf(v);  // the 4 outputs keep the exact tanh
            outv[o] = v;
            if (valid && half == 0 && o < a.n_out && a.out) {
                if (a.out_f64) ((double*)a.out)[out_off + (int64_t)o * a.N + e] = (double)v;
                else ((float*)a.out)[out_off + (int64_t)o * a.N + e] = v;
            }
        }
        if constexpr (SAMPLE) {
            if (valid && half == 0) sample_store(a, e, outv);
        }
        MLP_STAMP(7 + 6 * stamp_tile);
        ++stamp_tile;
    }
    MLP_STAMP(15);
}

// ---- k_mlp_v3: the N = 128, T = 5 policy forward, round 3 ------------------------------------------------------------
// What the in-kernel stamps of k_mlp<128,5,512,.,.,2> showed (tools/mlp_stamps.py, 131 072 envs, 104 k cycles per wave):
// 14 k cycles of weight staging (global fp32 -> three-way split -> LDS), 2 x 9 k cycles in which every wave of the chip
// waits for its window loads at the same time, and layer 3 -- 64 fp32 MFMAs of 64 cycles per tile for 4 real output rows
// in a 32-row tile -- as 29 % of the matrix-pipe time; the bf16 layers themselves run near the pipe's pace for two waves.
// And what tools/ubench/mfma_fill.hip measured: v_mfma_f32_32x32x2_f32 executes on the vector lanes (6 independent v_fma
// after each one: 64 -> 109 cycles, nothing hides), whereas v_mfma_f32_32x32x16_bf16 hides ~5 VALU instructions per MFMA
// in the SAME wave's stream (34 cycles with 4 fillers) -- but an MFMA-only wave next to a VALU-only wave on one SIMD is
// destructive (81 cycles per MFMA, 18 per v_fma).  Hence:
//  * layer 3 leaves the matrix pipe: each lane holds 64 of its env's 128 layer-2 activations in the accumulator layout, so
//    out[o] = sum_h W3[o][h] h2[h] is 4 x 64 exact fp32 FMAs per lane (W3 as [h][4] float4 rows in LDS: one broadcast
//    ds_read_b128 per activation), the two lane halves meet in one cross-half add.  -4 096 pipe cycles per tile, and the
//    FMAs are VALU work of the kind that hides under the bf16 MFMAs of the other wave of the SIMD;
//  * the weights arrive ALREADY in the LDS layout when the caller prepared them (qenv_mlp_prepare: once per weight
//    update, not once per launch and workgroup): staging = a 159 KiB copy; without an image the kernel splits as before;
//  * the first tile's window is requested before the staging, the next tile's right after the current one has been
//    split into its bf16 terms -- the 39 MB of window per launch stream in behind the MFMAs instead of in front of them.
// Layers 1 and 2 are k_mlp<.,.,.,.,.,2>'s arithmetic instruction for instruction (same products, same order).
// k_mlp_small<.,.,.,3> mirrors layer 3 (mlp3_layer3): the two kernels stay bit-identical (shard invariance).

__global__ __launch_bounds__(256) void k_mlp_prepare(uint4* image, const float* W1, const float* b1, const float* W2, const float* b2,
                                                     const float* W3, const float* b3, int n_out) {
    mlp3_build_image(image, W1, b1, W2, b2, W3, b3, n_out, (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256));
}

template <int BLOCK, bool SAMPLE>
__global__ __launch_bounds__(BLOCK) void k_mlp_v3(const MlpArgs a) {
    using namespace mlp3;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0;
    MLP_STAMP(0); MLP_STAMP(14);
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int wave = threadIdx.x >> 6, waves_per_block = BLOCK / 64;
    const int64_t tiles_per_chunk = (a.N + 31) / 32;
    const int64_t n_tiles = tiles_per_chunk * a.n_chunks;
    const int64_t tile_stride = (int64_t)gridDim.x * waves_per_block;
    int64_t gtile = (int64_t)blockIdx.x * waves_per_block + wave;
    uint32_t N32 = (uint32_t)a.N;

    // ---- the window: 40 loads per lane (element 16 sk + 8 half + i of the lane's env), as raw buffer loads whose byte offset
    // is (ring row of that element) * N * 4 + env * 4.  The 80 row offsets depend on the ring head only: a table in LDS,
    // [half][40], read back ten ds_read_b128 per request -- the per-load address arithmetic is ONE v_add (the 64-bit
    // multiply-adds and selects it replaces were ~6 VALU per load, 2 k cycles per request).  The five padded elements
    // (k >= 75) carry an offset beyond the buffer: the hardware range check returns 0 for them, as for the lanes of a
    // ragged last tile.
    uint32_t* rowtab = reinterpret_cast<uint32_t*>(lds4 + IMAGE_UINT4);   // 80 words behind the image
    if (threadIdx.x < 80) {
        const int hf = threadIdx.x / 40, slot = threadIdx.x % 40;        // slot = 8 sk + i
        const int k = 16 * (slot >> 3) + 8 * hf + (slot & 7);
        int sl = a.head + (k < KW ? k : 0) / 15;
        if (sl >= TW) sl -= TW;
        rowtab[threadIdx.x] = k < KW ? (uint32_t)(sl * 15 + k % 15) * N32 * 4u : 0x80000000u;
    }
    __syncthreads();
    float xw[KS16][8];
    const uint4* rowq = reinterpret_cast<const uint4*>(rowtab) + 10 * half;
    auto request_window = [&](int64_t g) {
        const int64_t chunk = a.n_chunks > 1 ? g / tiles_per_chunk : 0;
        const int64_t e = (g - chunk * tiles_per_chunk) * 32 + col;
        const uint32_t e4 = e < a.N ? (uint32_t)e * 4u : 0x80000000u;
        // the descriptor must be PROVABLY wave-uniform (the tile index comes from threadIdx.x >> 6): otherwise hipcc wraps every
        // one of the 40 buffer loads in a waterfall loop (readfirstlane x 4, compare, saveexec, branch)
        const uint64_t base = (uint64_t)(a.win + chunk * (a.win_chunk_stride ? a.win_chunk_stride : (int64_t)KW * a.N));
        const uint32_t base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);          // (the builtin returns int:
        const uint32_t base_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));  //  no sign extension into bit 32+)
        const uint64_t base_u = ((uint64_t)base_hi << 32) | (uint64_t)base_lo;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base_u, 0, (int)((uint32_t)KW * N32 * 4u), 0x00020000);
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const uint4 r = rowq[q];
            const uint32_t off[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                xw[(4 * q + c) >> 3][(4 * q + c) & 7] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off[c] + e4, 0, 0));
        }
    };

    // ---- the weights.  With a prepared image the staging is a copy (W1 / W3 / biases, then W2: every wave takes part); without
    // one every wave splits its share of the fp32 weights.  The upper half of the waves asks for its first window after the
    // barrier, the lower half before it (see below).
    const bool upper = wave >= waves_per_block / 2;
    // image[first .. first + count) -> LDS by nthr threads, eight 16-byte loads in flight per thread: one load per round trip
    // (what a plain loop compiles to) made the 96 KiB of W2 a 20 k-cycle latency chain
    // Every workgroup reads the SAME 159 KiB, and the 32 workgroups of an XCD start together: read in the same order they
    // all ask for the same L2 lines at the same time.  Each workgroup therefore walks its copy from its own starting point
    // (`rot`: 1/32 of the index space per workgroup of the XCD): W2 by four waves 15.9 k -> 9.8 k cycles, W1 14.5 k -> 11.7 k.
    // (A CU pulls ~10 B per clock through its vector L1 whatever the source -- MI355X_MICROARCH.md: prologue bursts ~11
    // B/cyc/CU -- and MORE loads in flight make it worse: all twenty 16-byte loads of a thread at once took 34 k cycles.)
    const int xcd_slot = (int)(blockIdx.x >> 3) & 31;   // workgroups b, b + 8, b + 16 ... share an XCD (round-robin dispatch)
    auto copy_image = [&](int first, int count, int tid, int nthr) {
        const int rot = (count / 32) * xcd_slot;
        for (int base = tid; base < count; base += 8 * nthr) {
            uint4 v[8];
            int idx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * nthr;
                int j = (i < count ? i : count - 1) + rot;
                if (j >= count) j -= count;
                idx[u] = j;
                v[u] = a.image[first + j];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (base + u * nthr < count) lds4[first + idx[u]] = v[u];
        }
    };
    if (a.image) {
        // the image loads go out BEFORE the first tile's window loads: vector-memory results return in issue order, and W1
        // (from L2) should not queue behind the 20 MB of first-tile windows that the whole chip pulls from HBM at once
        static_assert((W2F - W1F) <= 8 * BLOCK && (IMAGE_UINT4 - W3T) <= BLOCK, "one round of nine loads per thread");
        uint4 v[9];
        int idx[8];
        constexpr int C1 = W2F - W1F;
        const int rot = (C1 / 32) * xcd_slot;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = (int)threadIdx.x + u * BLOCK;
            int j = (i < C1 ? i : C1 - 1) + rot;
            if (j >= C1) j -= C1;
            idx[u] = j;
            v[u] = a.image[W1F + j];
        }
        v[8] = a.image[W3T + ((int)threadIdx.x < IMAGE_UINT4 - W3T ? (int)threadIdx.x : 0)];
        if (gtile < n_tiles && !upper) request_window(gtile);   // the lower half's first windows: behind W1, ahead of W2
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if ((int)threadIdx.x + u * BLOCK < C1) lds4[W1F + idx[u]] = v[u];
        if ((int)threadIdx.x < IMAGE_UINT4 - W3T) lds4[W3T + threadIdx.x] = v[8];
        copy_image(W2F, W3T - W2F, (int)threadIdx.x, BLOCK);   // W2 by every wave, eight loads in flight per thread and round
    } else {
        if (gtile < n_tiles) request_window(gtile);   // in flight during the staging
        mlp3_build_image(lds4, a.W1, a.b1, a.W2, a.b2, a.W3, a.b3, a.n_out, (int)threadIdx.x, BLOCK);
    }
    __syncthreads();
    // the upper half asks for its first window only now: two bursts instead of one, and the two waves of every SIMD start
    // their first tile a few thousand cycles apart -- a stagger without an unpaired phase at either end
    if (a.image && upper && gtile < n_tiles) request_window(gtile);
    MLP_STAMP(1);
    const uint4* W1f = lds4 + W1F;
    const uint4* W2f = lds4 + W2F;
    const float4* W3t = reinterpret_cast<const float4*>(lds4 + W3T);
    const float* b1s = reinterpret_cast<const float*>(lds4 + B1);
    const float* b2s = reinterpret_cast<const float*>(lds4 + B2);
    const float* b3s = reinterpret_cast<const float*>(lds4 + B3);

    [[maybe_unused]] int stamp_tile = 0;
    for (; gtile < n_tiles; gtile += tile_stride) {
        MLP_STAMP(2 + 6 * stamp_tile);
        // the weights in LDS and the ring offsets are tile-invariant: without this LICM hoists hundreds of LDS reads and spills
        asm volatile("" ::: "memory");
        const int64_t chunk = a.n_chunks > 1 ? gtile / tiles_per_chunk : 0;  // wave-uniform
        const int64_t tile = gtile - chunk * tiles_per_chunk;
        const int64_t out_off = chunk * (int64_t)a.n_out * a.N;
        const int64_t e = tile * 32 + col;
        const bool valid = e < a.N && !(a.env_skip && a.env_skip[e < a.N ? e : 0]);
        const uint32_t e32 = (uint32_t)(e < a.N ? e : a.N - 1);
        // ---- this tile's window -> bf16 terms; the observation copy
        bf16x8 xsp[KS16][3];
        {
            uint32_t n32 = N32;
            uint32_t kh = 8u * (uint32_t)half;
            asm volatile("" : "+s"(n32), "+v"(kh));  // 32-bit offsets formed per use (hoisted: 75 address pairs, spilled)
            float* __restrict__ obs_c = a.obs_store ? a.obs_store + chunk * (int64_t)KW * a.N : nullptr;
#pragma unroll
            for (int sk = 0; sk < KS16; ++sk) {
                if (obs_c && valid) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t k = (uint32_t)(16 * sk + i) + kh;
                        if (k < (uint32_t)KW && k >= (uint32_t)a.obs_rows_from) obs_c[k * n32 + e32] = xw[sk][i];
                    }
                }
                split3x8(xw[sk], xsp[sk][0], xsp[sk][1], xsp[sk][2]);
            }
        }
        MLP_STAMP(3 + 6 * stamp_tile);
        // ---- layers 1-3 as a software pipeline: every matrix-core chain (30 MFMAs of a layer-1 output tile, 48 of a layer-2
        // one) shares its scheduling region with the vector work of the PREVIOUS output tile (tanh + the bf16 split that
        // turns it into the next layer's B operand, or tanh + layer 3), and sched_group_barrier asks for one MFMA followed
        // by a handful of VALU instructions throughout: ~5 VALU issue slots hide under a v_mfma_f32_32x32x16_bf16 in the
        // same wave's stream (tools/ubench/mfma_fill.hip), while an MFMA-only phase of one wave next to a VALU-only phase
        // of the other wave of the SIMD -- what the un-pipelined loop produced -- costs both (81 cycles per MFMA).
        // (mlp3_tile_forward in qenv_mlp3.hpp is the same arithmetic scheduled for a wave that is alone on its SIMD, for
        // k_ppo_fused -- the two are compared bit for bit by the tests of the fused rollout)
        f32x16 acc1[NT];
        bf16x8 bsp[NT][2][3];
        auto l1_chain = [&](int t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = b1s[32 * t + acc_row(j, half)];
            const uint4* frag = W1f + (size_t)t * KS16 * 3 * 64 + lane;
#pragma unroll
            for (int sk = 0; sk < KS16; ++sk) {
                const uint4* f = frag + sk * 3 * 64;
                const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                             al = __builtin_bit_cast(bf16x8, f[128]);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[sk][1], acc1[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xsp[sk][0], acc1[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][2], acc1[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[sk][0], acc1[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][1], acc1[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][0], acc1[t], 0, 0, 0);
            }
        };
        auto l1_finish = [&](int t) {   // tanh, then the activations as bf16 terms (layer 2's B operand)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh_prescaled(acc1[t][j]);
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = acc1[t][8 * jh + i];
                split3x8(v8, bsp[t][jh][0], bsp[t][jh][1], bsp[t][jh][2]);
            }
        };
        auto l2_chain = [&](int t, f32x16& acc2, int ti0 = 0, int ti1 = NT) {   // input tiles [ti0, ti1) of output tile t
            if (ti0 == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc2[j] = b2s[32 * t + acc_row(j, half)];
            }
            const uint4* frag = W2f + (size_t)t * NT * 2 * 3 * 64 + lane;
#pragma unroll
            for (int ti = ti0; ti < ti1; ++ti)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const uint4* f = frag + (ti * 2 + jh) * 3 * 64;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                                 al = __builtin_bit_cast(bf16x8, f[128]);
                    const bf16x8 bh = bsp[ti][jh][0], bm = bsp[ti][jh][1], bl = bsp[ti][jh][2];
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);   // smallest products first
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2, 0, 0, 0);
                }
        };
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = half ? 0.f : b3s[k];
        {
            __builtin_amdgcn_sched_barrier(0);
            l1_chain(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 1; t < NT; ++t) {   // chain t beside tanh + split of t - 1
                l1_chain(t);
                l1_finish(t - 1);
                sgb_interleave<30, 6>();
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x16 acc2[2];
            l2_chain(0, acc2[0], 0, NT - 1); // its first 36 MFMAs read bsp[0..2] only: beside tanh + split of layer-1 tile 3
            l1_finish(NT - 1);
            sgb_interleave<36, 5>();
            __builtin_amdgcn_sched_barrier(0);
            l2_chain(0, acc2[0], NT - 1, NT);
            // the NEXT tile's window is requested here: its 40 registers are free again (the window lives on as xsp until layer 1
            // ends, the activations as bsp from here), and the rest of layer 2 + layer 3 give the loads ~10 k cycles to land
            if (gtile + tile_stride < n_tiles) request_window(gtile + tile_stride);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 1; t < NT; ++t) {   // chain t beside tanh + layer 3 of t - 1
                l2_chain(t, acc2[t & 1]);
#pragma unroll
                for (int j = 0; j < 16; ++j) acc2[(t - 1) & 1][j] = fast_tanh_prescaled(acc2[(t - 1) & 1][j]);
                mlp3_layer3(W3t, t - 1, half, acc2[(t - 1) & 1], o);
                sgb_interleave<48, 3>();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[(NT - 1) & 1][j] = fast_tanh_prescaled(acc2[(NT - 1) & 1][j]);
            mlp3_layer3(W3t, NT - 1, half, acc2[(NT - 1) & 1], o);
        }
        mlp3_layer3_finish(o);
        MLP_STAMP(6 + 6 * stamp_tile);
        float outv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = o[k];
            if (a.final_tanh) v = tanhf(v);  // the 4 outputs keep the exact tanh
            outv[k] = v;
            if (valid && half == 0 && k < a.n_out && a.out) {
                if (a.out_f64) ((double*)a.out)[out_off + (int64_t)k * a.N + e] = (double)v;
                else ((float*)a.out)[out_off + (int64_t)k * a.N + e] = v;
            }
        }
        if constexpr (SAMPLE) {
            if (valid && half == 0) sample_store(a, e, outv);
        }
        MLP_STAMP(7 + 6 * stamp_tile);
        ++stamp_tile;
    }
    MLP_STAMP(15);
}

// ---- the same MLP for SMALL batches (a handful of 32-env tiles: the N = 1 drop-in, the 256-env training demo) ----
// k_mlp's cost at that size is not arithmetic: one workgroup stages 119 KB of weights (which every wave then uses
// exactly once) and each wave walks the 472 dependent MFMAs of its tile alone: 39 us for 256 envs.  Here a tile
// belongs to NT = HP/32 waves, wave t computing output tile t of layers 1 and 2; the weights are read straight from
// global memory (L2) in the A-operand layout -- no staging, each element is used once per tile -- and the activations
// cross waves through LDS in the lane layout they already have (register j of lane l -> word (16 t + j) 64 + l, so
// every wave reads tile ti's register j as ITS lane's B operand: conflict-free, and the k-pairs and their order are
// those of k_mlp).  Layer 3 stays one 16 NT-MFMA chain on wave 0 so the accumulation order is k_mlp's too: the
// results are bit-identical to k_mlp's.  Dependent MFMAs per tile: 38 + 16 NT + 16 NT instead of NT (38 + 16 NT + 16).
template <int H, int TW, bool SAMPLE = false, int BF16 = 0>
__global__ __launch_bounds__((H < 32 ? 32 : H) * 2) void k_mlp_small(const MlpArgs a) {
    constexpr bool BF16L2 = BF16 >= 1, BF16L1 = BF16 >= 2, L3V = BF16 >= 3;  // 3: k_mlp_v3's arithmetic (layer 3 on the vector lanes)
    constexpr int HP = H < 32 ? 32 : H;
    constexpr int NT = HP / 32;
    constexpr int K1 = 15 * TW, KS = (K1 + 1) / 2, KS16 = (K1 + 15) / 16;
    __shared__ float x1[NT * 16 * 64], x2[NT * 16 * 64];
    __shared__ float4 w3t[L3V ? HP : 1];  // L3V: W3 as [h] float4 rows (outputs 0..3), read by wave 0 behind the tile's barriers
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0;
    if constexpr (L3V) {
        for (int hh = threadIdx.x; hh < HP; hh += HP * 2) {
            float4 w;
            w.x = hh < H ? a.W3[hh] : 0.f;
            w.y = (hh < H && a.n_out > 1) ? a.W3[(size_t)H + hh] : 0.f;
            w.z = (hh < H && a.n_out > 2) ? a.W3[(size_t)2 * H + hh] : 0.f;
            w.w = (hh < H && a.n_out > 3) ? a.W3[(size_t)3 * H + hh] : 0.f;
            w3t[hh] = w;
        }
    }
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, t = threadIdx.x >> 6;
    const int h = 32 * t + col;  // the hidden row this lane supplies as A operand
    const bool h_ok = h < H, o_ok = col < a.n_out;
    const float* __restrict__ w1 = a.W1 + (size_t)(h_ok ? h : 0) * K1;
    const float* __restrict__ w2 = a.W2 + (size_t)(h_ok ? h : 0) * H;
    const float* __restrict__ w3 = a.W3 + (size_t)(o_ok ? col : 0) * H;
    const int64_t tiles_per_chunk = (a.N + 31) / 32;
    const int64_t n_tiles = tiles_per_chunk * a.n_chunks;
    // this lane's A operands of all three layers, fetched ONCE (they do not depend on the tile) with back-to-back
    // loads: a load in front of every MFMA would put an L2 round trip on the dependent chain 38 + 32 NT times
    float a1[BF16L1 ? 1 : KS], a2[NT * 16], a3[NT * 16];
    bf16x8 a1sp[BF16L1 ? KS16 : 1][3];
    if constexpr (BF16L1) {  // k_mlp's split layer 1: this lane's A operand of step sk = W1[h][16 sk + 8 half + i]
#pragma unroll
        for (int sk = 0; sk < KS16; ++sk) {
            float w8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 16 * sk + 8 * half + i;
                const float v = w1[k < K1 ? k : K1 - 1];
                w8[i] = (h_ok && k < K1) ? (L3V ? v * TANH_PRESCALE : v) : 0.f;   // L3V: k_mlp_v3's pre-scaled hidden layers
            }
            split3x8(w8, a1sp[sk][0], a1sp[sk][1], a1sp[sk][2]);
        }
    } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + half;
        const float v = w1[k < K1 ? k : K1 - 1];
        a1[s] = (h_ok && k < K1) ? v : 0.f;
    }
    }
#pragma unroll
    for (int i = 0; i < NT * 16; ++i) {
        const int k = 32 * (i >> 4) + acc_row(i & 15, half);
        const float v = w2[k < H ? k : 0];
        a2[i] = (h_ok && k < H) ? (L3V ? v * TANH_PRESCALE : v) : 0.f;
    }
    if (t == 0 && !L3V) {
#pragma unroll
        for (int i = 0; i < NT * 16; ++i) {
            const int k = 32 * (i >> 4) + acc_row(i & 15, half);
            const float v = w3[k < H ? k : 0];
            a3[i] = (o_ok && k < H) ? v : 0.f;
        }
    }
    // BF16L2 (see k_mlp): the same three-way split operands and the same order of the six products per 16-k step, so
    // the outputs stay bit-identical to k_mlp's
    bf16x8 asp[BF16L2 ? NT : 1][2][3];
    if constexpr (BF16L2) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                float w8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w8[i] = a2[16 * ti + 8 * jh + i];
                split3x8(w8, asp[ti][jh][0], asp[ti][jh][1], asp[ti][jh][2]);
            }
    }
    float bias1[16], bias2[16], bias3[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = 32 * t + acc_row(j, half), r3 = acc_row(j, half);
        bias1[j] = r < H ? (L3V ? a.b1[r] * TANH_PRESCALE : a.b1[r]) : 0.f;
        bias2[j] = r < H ? (L3V ? a.b2[r] * TANH_PRESCALE : a.b2[r]) : 0.f;
        bias3[j] = r3 < a.n_out ? a.b3[r3] : 0.f;
    }
    const int64_t n_list = a.list_idx ? (int64_t)*a.list_count : 0;  // list mode: dense tiles gathered from a compacted list
    for (int64_t gtile = blockIdx.x; gtile < n_tiles; gtile += gridDim.x) {
        const int64_t chunk = a.n_chunks > 1 ? gtile / tiles_per_chunk : 0;
        const int64_t tile = gtile - chunk * tiles_per_chunk;
        if (a.list_idx && tile * 32 >= n_list) break;  // block-uniform: no barrier is skipped by a part of the block
        const float* __restrict__ winc = a.win + chunk * (a.win_chunk_stride ? a.win_chunk_stride : (int64_t)K1 * a.N);
        const int64_t j_list = tile * 32 + col;
        const int64_t e = a.list_idx ? (j_list < n_list ? (int64_t)a.list_idx[j_list] : a.N) : j_list;
        // an env left to the side stream (env_skip: the overlapped PPO step) rides along in its tile, but nothing of it is stored --
        // like k_mlp / k_mlp_v3.  (Round 4: this kernel stored it; the side stream's launch writes the same action / observation
        // rows from the env's warmed-up window, and whichever store landed last stayed: a rare mismatch of the stored rows.)
        const bool valid = e < a.N && !(a.env_skip && a.env_skip[e < a.N ? e : 0]);
        const int64_t ec = valid ? e : a.N - 1;
        f32x16 acc;
        if constexpr (BF16L1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = bias1[j];
#pragma unroll
            for (int sk = 0; sk < KS16; ++sk) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k0 = 16 * sk + i, k1 = 16 * sk + 8 + i;
                    int sl0 = a.head + (k0 < K1 ? k0 : 0) / 15, sl1 = a.head + (k1 < K1 ? k1 : 0) / 15;  // wave-uniform
                    if (sl0 >= TW) sl0 -= TW;
                    if (sl1 >= TW) sl1 -= TW;
                    const int r0 = sl0 * 15 + (k0 < K1 ? k0 : 0) % 15, r1 = sl1 * 15 + (k1 < K1 ? k1 : 0) % 15;
                    float x = winc[(int64_t)(half ? r1 : r0) * a.N + ec];
                    if (k1 >= K1) x = (half || k0 >= K1) ? 0.f : x;
                    v8[i] = x;
                    if (a.obs_store && valid && t == 0) {
                        const int k = 16 * sk + 8 * half + i;
                        if (k < K1 && k >= a.obs_rows_from) a.obs_store[(int64_t)k * a.N + e] = x;
                    }
                }
                bf16x8 xh, xm, xl;
                split3x8(v8, xh, xm, xl);
                const bf16x8 ah = a1sp[sk][0], am = a1sp[sk][1], al = a1sp[sk][2];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, acc, 0, 0, 0);
            }
        } else {
        float xs[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 2 * s, k1 = 2 * s + 1;
            int sl0 = a.head + k0 / 15, sl1 = a.head + k1 / 15;  // wave-uniform
            if (sl0 >= TW) sl0 -= TW;
            if (sl1 >= TW) sl1 -= TW;
            const int r0 = sl0 * 15 + k0 % 15;
            const int r1 = k1 < K1 ? sl1 * 15 + k1 % 15 : r0;
            float x = winc[(int64_t)(half ? r1 : r0) * a.N + ec];
            if (k1 >= K1) x = half ? 0.f : x;
            xs[s] = x;
        }
        if (a.obs_store && valid && t == 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k = 2 * s + half;
                if (k < K1 && k >= a.obs_rows_from) a.obs_store[(int64_t)k * a.N + e] = xs[s];
            }
        }
        // layer 1, output tile t: k-step s pairs window elements 2s (lanes 0-31) and 2s+1 (lanes 32-63)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias1[j];
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], xs[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) x1[(16 * t + j) * 64 + lane] = L3V ? fast_tanh_prescaled(acc[j]) : fast_tanh(acc[j]);
        __syncthreads();
        // layer 2, output tile t, over all NT input tiles
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias2[j];
        if constexpr (BF16L2) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    float v8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v8[i] = x1[(16 * ti + 8 * jh + i) * 64 + lane];
                    bf16x8 bh, bm, bl;
                    split3x8(v8, bh, bm, bl);
                    const bf16x8 ah = asp[ti][jh][0], am = asp[ti][jh][1], al = asp[ti][jh][2];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int i = 0; i < NT * 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[i], x1[i * 64 + lane], acc, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) x2[(16 * t + j) * 64 + lane] = L3V ? fast_tanh_prescaled(acc[j]) : fast_tanh(acc[j]);
        __syncthreads();  // also: every wave is done with x1 before the next tile overwrites it
        if (t == 0) {     // layer 3 (n_out <= 4 rows in one zero-padded 32-row tile): one chain, k_mlp's order
            float o3[4];
            if constexpr (L3V) {  // k_mlp_v3's layer 3: the same FMAs in the same order on the same lanes
#pragma unroll
                for (int k = 0; k < 4; ++k) o3[k] = (half || k >= a.n_out) ? 0.f : a.b3[k];
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    f32x16 h2;
#pragma unroll
                    for (int j = 0; j < 16; ++j) h2[j] = x2[(16 * ti + j) * 64 + lane];
                    mlp3_layer3(w3t, ti, half, h2, o3);
                }
                mlp3_layer3_finish(o3);
            } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = bias3[j];
#pragma unroll
            for (int i = 0; i < NT * 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[i], x2[i * 64 + lane], acc, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) o3[k] = acc[k];
            }
            const int64_t out_off = chunk * (int64_t)a.n_out * a.N;
            float outv[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float v = o3[o];
                if (a.final_tanh) v = tanhf(v);
                outv[o] = v;
                if (valid && half == 0 && o < a.n_out && a.out) {
                    if (a.out_f64) ((double*)a.out)[out_off + (int64_t)o * a.N + e] = (double)v;
                    else ((float*)a.out)[out_off + (int64_t)o * a.N + e] = v;
                }
            }
            if constexpr (SAMPLE) {
                if (valid && half == 0) sample_store(a, e, outv);
            }
        }
        // x2 is rewritten only after the next tile's first barrier, which wave 0 reaches after this layer 3
    }
}

// ---- the same MLP for WIDE hidden layers (N = 256: W2 alone is 256 KiB, more than the LDS) -------------------------
// k_mlp<256> read W2 from global memory inside the MFMA chain: 128 dependent 4-byte loads per output tile, which hipcc
// hoisted until the kernel held 256 VGPRs + 256 AGPRs and still spilled 912 B per lane.  Here the four waves of a
// workgroup march over the eight layer-2 output tiles together (each on its own 32-env tile) and the 32 weight rows of
// the CURRENT output tile -- a contiguous 32 KiB block of W2 -- are staged through LDS by the whole workgroup, rows
// padded by one word so the A-operand reads (lane = weight row) are conflict-free and the coalesced staging writes too.
// NBUF = 2 double-buffers the block (the global loads of tile t+1 are issued before the MFMAs of tile t and written to
// LDS after them: one barrier per output tile); NBUF = 1 when the window is too long for two buffers.  MFMA order, k
// pairing and operands are k_mlp's, so the results are bit-identical to the kernel this one replaces.
template <int H, int TW, int NBUF, bool SAMPLE = false>
__global__ __launch_bounds__(256) void k_mlp_wide(const MlpArgs a) {
    static_assert(H % 32 == 0 && H > 128, "the wide kernel is for hidden layers that do not fit the LDS");
    constexpr int NT = H / 32, PITCH = H + 1, TILE_WORDS = 32 * PITCH, PER_LANE = 32 * H / 256;
    static_assert(NT % 2 == 0 || NBUF == 1, "buffer parity wraps across rounds");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0;
    const int K1 = TW ? 15 * TW : 15 * a.T, K1P = (K1 + 1) & ~1;
    float* W1t = lds;                    // [K1P][H]  (k-major)
    float* W3t = W1t + (size_t)K1P * H;  // [H][4]    (the four real output rows)
    float* b1s = W3t + 4 * H;
    float* b2s = b1s + H;
    float* b3s = b2s + H;                // [32]
    float* W2s = b3s + 32;               // [NBUF][32][PITCH]: output rows 32 t .. 32 t + 31 of W2, k contiguous
    {
        constexpr int KL = 8, HL = 8;
        const int kl = threadIdx.x % KL, hl = (threadIdx.x / KL) % HL, wv = threadIdx.x >> 6;
        const int kb1 = (K1P + KL - 1) / KL;
        for (int blk = wv; blk < (H / HL) * kb1; blk += 4) {
            const int h = (blk / kb1) * HL + hl, k = (blk % kb1) * KL + kl;
            if (k < K1P) W1t[k * H + h] = k < K1 ? a.W1[(size_t)h * K1 + k] : 0.f;
        }
    }
    for (int idx = threadIdx.x; idx < 4 * H; idx += 256) {
        const int h = idx / 4, o = idx % 4;
        W3t[idx] = o < a.n_out ? a.W3[(size_t)o * H + h] : 0.f;
    }
    for (int h = threadIdx.x; h < H; h += 256) {
        b1s[h] = a.b1[h];
        b2s[h] = a.b2[h];
    }
    if (threadIdx.x < 32) b3s[threadIdx.x] = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;

    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    // staging: wave w moves rows 8 w .. 8 w + 7 of the block, 64 consecutive k per instruction
    float st[PER_LANE];
    auto stage_load = [&](int t) {
        // the element offset is made opaque per use: left to itself hipcc computes the 64-bit address of every one of
        // the NT * PER_LANE loads once, outside the round loop, and spills them (180 pairs)
        uint32_t off = (uint32_t)((32 * t + 8 * wave) * H + lane);
        asm volatile("" : "+v"(off));
        const float* src = a.W2 + off;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) st[i] = src[(i / (H / 64)) * H + 64 * (i % (H / 64))];
    };
    auto stage_write = [&](int buf) {
        float* dst = W2s + buf * TILE_WORDS + (8 * wave) * PITCH + lane;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) dst[(i / (H / 64)) * PITCH + 64 * (i % (H / 64))] = st[i];
    };
    stage_load(0);
    stage_write(0);

    const int64_t tiles_per_chunk = (a.N + 31) / 32;
    const int64_t n_tiles = tiles_per_chunk * a.n_chunks;
    const int64_t per_round = (int64_t)gridDim.x * 4;
    const int64_t rounds = (n_tiles + per_round - 1) / per_round;  // block-uniform: every wave reaches every barrier
    for (int64_t round = 0; round < rounds; ++round) {
        asm volatile("" ::: "memory");  // as in k_mlp: keeps the LDS operand reads inside the loop
        const int64_t gtile = round * per_round + (int64_t)blockIdx.x * 4 + wave;
        const bool tile_ok = gtile < n_tiles;
        const int64_t chunk = (tile_ok && a.n_chunks > 1) ? gtile / tiles_per_chunk : 0;
        const int64_t tile = tile_ok ? gtile - chunk * tiles_per_chunk : 0;
        const float* __restrict__ winc = a.win + chunk * (int64_t)K1 * a.N;
        const int64_t out_off = chunk * (int64_t)a.n_out * a.N;
        const int64_t e = tile * 32 + col;
        const bool valid = tile_ok && e < a.N && !(a.env_skip && a.env_skip[e < a.N ? e : 0]);
        f32x16 acc1[NT];
        if (round == 0) __syncthreads();  // W1 / W3 / biases staged (the W2 block has its own barrier below)
        if constexpr (TW > 0) {
            // the reference's window: the whole observation of the tile is in flight before the first MFMA (k_mlp's
            // prefetch; the run-time-T loop below waits for one load per k-step)
            constexpr int KS = (15 * (TW ? TW : 1) + 1) / 2;
            float xs[KS];
            uint32_t N32 = (uint32_t)a.N;
            int hsel = half;
            asm volatile("" : "+s"(N32), "+v"(hsel));
            const uint32_t e32 = (uint32_t)(valid ? e : a.N - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k0 = 2 * s, k1 = 2 * s + 1;
                int sl0 = a.head + k0 / 15, sl1 = a.head + k1 / 15;
                if (sl0 >= TW) sl0 -= TW;
                if (sl1 >= TW) sl1 -= TW;
                const uint32_t r0 = (uint32_t)(sl0 * 15 + k0 % 15);
                const uint32_t r1 = k1 < 15 * TW ? (uint32_t)(sl1 * 15 + k1 % 15) : r0;
                float x = winc[(hsel ? r1 : r0) * N32 + e32];
                if (k1 >= 15 * TW) x = hsel ? 0.f : x;
                xs[s] = x;
            }
            if (a.obs_store && valid) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const uint32_t k = (uint32_t)(2 * s + half);
                    if (2 * s + 1 < 15 * TW || half == 0) a.obs_store[k * N32 + e32] = xs[s];
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = b1s[32 * t + acc_row(j, half)];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1t[(2 * s + half) * H + 32 * t + col], xs[s], acc1[t], 0, 0, 0);
                    if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh(acc1[t][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = b1s[32 * t + acc_row(j, half)];
        for (int s = 0; s < K1P / 2; ++s) {
            const int k = 2 * s + half;
            float x = 0.f;
            if (valid && k < K1) {
                int slice = a.head + k / 15;
                if (slice >= a.T) slice -= a.T;
                x = winc[(int64_t)(slice * 15 + k % 15) * a.N + e];
                if (a.obs_store) a.obs_store[(int64_t)k * a.N + e] = x;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1t[k * H + 32 * t + col], x, acc1[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh(acc1[t][j]);
        }
        f32x16 acc3;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc3[j] = b3s[acc_row(j, half)];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            constexpr int dummy = 0; (void)dummy;
            const int buf = NBUF == 2 ? (t & 1) : 0;
            __syncthreads();  // block t is in LDS; nobody still reads the buffer block t + 1 goes to
            if constexpr (NBUF == 2) stage_load((t + 1) % NT);
            f32x16 acc2;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = b2s[32 * t + acc_row(j, half)];
            const float* wrow = W2s + buf * TILE_WORDS + col * PITCH;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32 * ti + acc_row(j, half)], acc1[ti][j], acc2, 0, 0, 0);
                    if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = fast_tanh(acc2[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float w3 = col < 4 ? W3t[(32 * t + acc_row(j, half)) * 4 + col] : 0.f;
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3, acc2[j], acc3, 0, 0, 0);
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NBUF == 2) {
                stage_write((t + 1) & 1);
            } else {
                __syncthreads();  // every wave is done with block t
                stage_load((t + 1) % NT);
                stage_write(0);
            }
        }
        float outv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = acc3[o];
            if (a.final_tanh) v = tanhf(v);
            outv[o] = v;
            if (valid && half == 0 && o < a.n_out && a.out) {
                if (a.out_f64) ((double*)a.out)[out_off + (int64_t)o * a.N + e] = (double)v;
                else ((float*)a.out)[out_off + (int64_t)o * a.N + e] = v;
            }
        }
        if constexpr (SAMPLE) {
            if (valid && half == 0) sample_store(a, e, outv);
        }
    }
}

// ---- k_mlp_wide_v3: the N = 256, T = 5 policy forward on the bf16 matrix cores -----------------------------------------------
// k_mlp_v3's arithmetic (three-way split operands on v_mfma_f32_32x32x16_bf16, pre-scaled hidden layers, layer 3 on the vector
// lanes) for the reference's 256-wide networks (solved/nn_old_solved_256_*.pth).  Neither weight matrix fits the LDS as
// fragments (W1 120 KiB, W2 384 KiB), so BOTH layers stream through one double-buffered block of 48 KiB, one output tile of 32
// rows at a time (sixteen blocks per tile round: eight of layer 1 with 5 k-steps, eight of layer 2 with 16): every thread
// loads the eight consecutive k of up to four (row, k-step, lane half) fragments from the torch-layout rows, splits them and
// writes the three bf16 terms where the MFMA reads them -- one barrier per block, the next block's rows in flight during the
// current block's MFMAs (k_dense_tail_v3's scheme).  One wave per SIMD: a wave keeps all of H1 as bf16 terms (192 registers).
// 1 008 MFMAs of 32 cycles per 32-env tile instead of 1 324 of 64.
namespace wide3 {
constexpr int H = 256, NT = 8, TW = 5, KW = 75, KS1 = 5, KS2 = 16;
constexpr int BUF = KS2 * 3 * 64;                     // uint4 per weight block buffer (3 072 -> 48 KiB)
constexpr int W3T = 2 * BUF;                          // [256] float4 rows of W3
constexpr int B1 = W3T + H, B2 = B1 + H / 4, B3 = B2 + H / 4, ROWTAB = B3 + 1, LDS_UINT4 = ROWTAB + 20;
}  // namespace wide3

template <bool SAMPLE>
__global__ __launch_bounds__(256) void k_mlp_wide_v3(const MlpArgs a) {
    using namespace wide3;
    constexpr int BLOCK = 256, WPB = 4;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    if (a.zero_word && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_word = 0;
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles_per_chunk = (a.N + 31) / 32;
    const int64_t n_tiles = tiles_per_chunk * a.n_chunks;
    const int64_t rounds = (n_tiles + (int64_t)gridDim.x * WPB - 1) / ((int64_t)gridDim.x * WPB);
    const uint32_t N32 = (uint32_t)a.N;
    // ---- layer 3's rows, the biases, the ring's row offsets (see k_mlp_v3)
    for (int h = threadIdx.x; h < H; h += BLOCK) {
        float4 w;
        w.x = a.W3[h]; w.y = a.n_out > 1 ? a.W3[(size_t)H + h] : 0.f; w.z = a.n_out > 2 ? a.W3[(size_t)2 * H + h] : 0.f;
        w.w = a.n_out > 3 ? a.W3[(size_t)3 * H + h] : 0.f;
        lds4[W3T + h] = __builtin_bit_cast(uint4, w);
        reinterpret_cast<float*>(lds4 + B1)[h] = a.b1[h] * TANH_PRESCALE;
        reinterpret_cast<float*>(lds4 + B2)[h] = a.b2[h] * TANH_PRESCALE;
    }
    if (threadIdx.x < 4) reinterpret_cast<float*>(lds4 + B3)[threadIdx.x] = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;
    uint32_t* rowtab = reinterpret_cast<uint32_t*>(lds4 + ROWTAB);   // [half][40]
    if (threadIdx.x < 80) {
        const int hf = threadIdx.x / 40, slot = threadIdx.x % 40;
        const int k = 16 * (slot >> 3) + 8 * hf + (slot & 7);
        int sl = a.head + (k < KW ? k : 0) / 15;
        if (sl >= TW) sl -= TW;
        rowtab[threadIdx.x] = k < KW ? (uint32_t)(sl * 15 + k % 15) * N32 * 4u : 0x80000000u;
    }
    const float* b3s = reinterpret_cast<const float*>(lds4 + B3);
    __syncthreads();
    const uint4* rowq = reinterpret_cast<const uint4*>(rowtab) + 10 * half;

    // ---- the weight blocks.  Block b < 8: output tile b of layer 1 (KS1 k-steps: 320 fragments); b >= 8: output tile b - 8 of
    // layer 2 (KS2 k-steps: 1 024 fragments).  Fragment f of a block = (k-step f >> 6, lane f & 63); thread i takes f = i + 256 u.
    float4 wq[2][4][2];   // the rows of the next two blocks: set b & 1
    auto request_block = [&](int b, float4 (&wq)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = (int)threadIdx.x + BLOCK * u, ln = f & 63, ks = f >> 6;
            if (b < NT) {
                const int k0 = 16 * ks + 8 * (ln >> 5);
                const float* src = a.W1 + (size_t)(32 * b + (ln & 31)) * KW + k0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {   // (75-float rows: no 16-byte alignment)
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (ks < KS1 && k0 + 4 * q + i < KW) ? src[4 * q + i] : 0.f;
                    wq[u][q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {   // lane (col, half) of k-step (ti, jh) holds W2[32 t + col][32 ti + acc_row(8 jh + i, half)]
                const int ti = ks >> 1, jh = ks & 1;
                const float* src = a.W2 + (size_t)(32 * (b - NT) + (ln & 31)) * H + 32 * ti + 16 * jh + 4 * (ln >> 5);
                wq[u][0] = *reinterpret_cast<const float4*>(src);
                wq[u][1] = *reinterpret_cast<const float4*>(src + 8);
            }
        }
    };
    auto put_block = [&](int b, const float4 (&wq)[4][2]) {
        const int nf = b < NT ? KS1 * 64 : KS2 * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = (int)threadIdx.x + BLOCK * u;
            if (f < nf) {
                const float w8[8] = {wq[u][0].x * TANH_PRESCALE, wq[u][0].y * TANH_PRESCALE, wq[u][0].z * TANH_PRESCALE, wq[u][0].w * TANH_PRESCALE,
                                     wq[u][1].x * TANH_PRESCALE, wq[u][1].y * TANH_PRESCALE, wq[u][1].z * TANH_PRESCALE, wq[u][1].w * TANH_PRESCALE};
                bf16x8 sp[3];
                split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
                for (int k = 0; k < 3; ++k) lds4[(b & 1) * BUF + ((f >> 6) * 3 + k) * 64 + (f & 63)] = __builtin_bit_cast(uint4, sp[k]);
            }
        }
    };
    for (int64_t r = 0; r < rounds; ++r) {
        asm volatile("" ::: "memory");
        const int64_t gtile = (r * gridDim.x + blockIdx.x) * WPB + wave;
        const bool has = gtile < n_tiles;
        const int64_t gt = has ? gtile : n_tiles - 1;
        const int64_t chunk = a.n_chunks > 1 ? gt / tiles_per_chunk : 0;
        const int64_t tile = gt - chunk * tiles_per_chunk;
        const int64_t out_off = chunk * (int64_t)a.n_out * a.N;
        const int64_t e = tile * 32 + col;
        const bool valid = has && e < a.N && !(a.env_skip && a.env_skip[e < a.N ? e : 0]);
        const uint32_t e32 = (uint32_t)(e < a.N ? e : a.N - 1);
        request_block(0, wq[0]);
        request_block(1, wq[1]);
        // ---- the window (raw buffer loads through the row table, see k_mlp_v3) -> bf16 terms; the observation copy
        float xw[KS1][8];
        {
            const uint32_t e4 = (has && e < a.N) ? (uint32_t)e * 4u : 0x80000000u;
            const uint64_t base = (uint64_t)(a.win + chunk * (int64_t)KW * a.N);
            const uint32_t base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            const uint32_t base_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
            const uint64_t base_u = ((uint64_t)base_hi << 32) | (uint64_t)base_lo;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base_u, 0, (int)((uint32_t)KW * N32 * 4u), 0x00020000);
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const uint4 rq = rowq[q];
                const uint32_t off[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    xw[(4 * q + c) >> 3][(4 * q + c) & 7] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off[c] + e4, 0, 0));
            }
        }
        bf16x8 xsp[KS1][3];
        {
            float* __restrict__ obs_c = a.obs_store ? a.obs_store + chunk * (int64_t)KW * a.N : nullptr;
#pragma unroll
            for (int sk = 0; sk < KS1; ++sk) {
                if (obs_c && valid) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t k = (uint32_t)(16 * sk + i) + 8u * (uint32_t)half;
                        if (k < (uint32_t)KW) obs_c[k * N32 + e32] = xw[sk][i];
                    }
                }
                split3x8(xw[sk], xsp[sk][0], xsp[sk][1], xsp[sk][2]);
            }
        }
        bf16x8 bsp[NT][2][3];   // H1 as bf16 terms: layer 2's B operand
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = half ? 0.f : b3s[k];
        // lane parts of the LDS addresses indexed by an accumulator row (32 t + acc_row(j, half) = a literal + 4 half), opaque per round:
        // as loop invariants every base + literal sum is hoisted out of the round loop and spilled (see k_mlp_bwd_v3)
        int w3_base = W3T * 16 + 64 * half, b1_base = B1 * 16 + 16 * half, b2_base = B2 * 16 + 16 * half;
        asm volatile("" : "+v"(w3_base), "+v"(b1_base), "+v"(b2_base));
        const char* ldsb = reinterpret_cast<const char*>(lds4);
        // Block pipeline: when a phase starts block b is staged in buffer b & 1 and the rows of blocks b + 1, b + 2 are in the two
        // register sets (landed / in flight).  The phase issues block b's MFMAs with the split + LDS write of block b + 1 among
        // them (buffer (b + 1) & 1 was last read in phase b - 1, which every wave left before the barrier that opened this phase),
        // then requests block b + 2 into the set it has just emptied... no: into the set block b occupied (free since phase b - 1).
        put_block(0, wq[0]);
        __syncthreads();
        request_block(2, wq[0]);
        // ---- layer 1: eight blocks, output tile b -> tanh -> its bf16 terms
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const uint4* frag = lds4 + (b & 1) * BUF + lane;
            f32x16 acc;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = *reinterpret_cast<const float*>(ldsb + (b1_base + 4 * (32 * b + acc_row(j, 0))));
#pragma unroll
            for (int sk = 0; sk < KS1; ++sk) {
                const uint4* f = frag + sk * 3 * 64;
                const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                             al = __builtin_bit_cast(bf16x8, f[128]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[sk][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xsp[sk][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[sk][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[sk][0], acc, 0, 0, 0);
            }
            put_block(b + 1, wq[(b + 1) & 1]);          // (b + 1 <= 8: the first layer-2 block)
            sgb_interleave<30, 6>();                    // the split among the chain's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            request_block(b + 3, wq[(b + 1) & 1]);      // the set just emptied: two phases until it is put
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fast_tanh_prescaled(acc[j]);
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = acc[8 * jh + i];
                split3x8(v8, bsp[b][jh][0], bsp[b][jh][1], bsp[b][jh][2]);
            }
            __syncthreads();
        }
        // ---- layer 2 + layer 3: eight blocks, two per iteration of a loop that is not unrolled further (the register set of
        // a block is a compile-time index)
        auto l2_phase = [&](int t, auto set_c) {
            constexpr int SET = decltype(set_c)::value;   // = (NT + t) & 1
            const int b = NT + t;
            const uint4* frag = lds4 + (b & 1) * BUF + lane;
            f32x16 acc;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = *reinterpret_cast<const float*>(ldsb + (b2_base + 4 * (32 * t + acc_row(j, 0))));
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const uint4* f = frag + (ti * 2 + jh) * 3 * 64;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                                 al = __builtin_bit_cast(bf16x8, f[128]);
                    const bf16x8 bh = bsp[ti][jh][0], bm = bsp[ti][jh][1], bl = bsp[ti][jh][2];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                }
            if (t + 1 < NT) put_block(b + 1, wq[SET ^ 1]);
            sgb_interleave<96, 2>();
            __builtin_amdgcn_sched_barrier(0);
            if (t + 3 < NT) request_block(b + 3, wq[SET ^ 1]);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float hv = fast_tanh_prescaled(acc[j]);
                const float4 w = *reinterpret_cast<const float4*>(ldsb + (w3_base + 16 * (32 * t + acc_row(j, 0))));
                o[0] = fmaf(w.x, hv, o[0]); o[1] = fmaf(w.y, hv, o[1]); o[2] = fmaf(w.z, hv, o[2]); o[3] = fmaf(w.w, hv, o[3]);
            }
            __syncthreads();
        };
#pragma unroll 1
        for (int t = 0; t < NT; t += 2) {
            l2_phase(t, std::integral_constant<int, 0>{});       // block 8 + t: even
            l2_phase(t + 1, std::integral_constant<int, 1>{});
        }
        mlp3_layer3_finish(o);
        float outv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = o[k];
            if (a.final_tanh) v = tanhf(v);  // the 4 outputs keep the exact tanh
            outv[k] = v;
            if (valid && half == 0 && k < a.n_out && a.out) {
                if (a.out_f64) ((double*)a.out)[out_off + (int64_t)k * a.N + e] = (double)v;
                else ((float*)a.out)[out_off + (int64_t)k * a.N + e] = v;
            }
        }
        if constexpr (SAMPLE) {
            if (valid && half == 0) sample_store(a, e, outv);
        }
    }
}

template <int H, int TW, bool SAMPLE> static int launch_mlp_wide(const MlpArgs& a, int T, hipStream_t stream, bool& fits) {
    const int K1P = (15 * T + 1) & ~1;
    const size_t fixed = sizeof(float) * ((size_t)K1P * H + 4 * H + 2 * H + 32), blockb = sizeof(float) * 32 * (H + 1);
    const int nbuf = fixed + 2 * blockb <= 160 * 1024 ? 2 : (fixed + blockb <= 160 * 1024 ? 1 : 0);
    fits = nbuf > 0 && (TW == 0 || (int64_t)15 * TW * a.N < ((int64_t)1 << 31));
    if (!fits) return QENV_OK;
    const int64_t n_tiles = ((a.N + 31) / 32) * a.n_chunks;
    int64_t blocks = (n_tiles + 3) / 4;
    if (blocks > 256) blocks = 256;
    const size_t lds = fixed + nbuf * blockb;
    const void* fn = nbuf == 2 ? (const void*)k_mlp_wide<H, TW, 2, SAMPLE> : (const void*)k_mlp_wide<H, TW, 1, SAMPLE>;
    hipError_t e = allow_dynamic_lds(fn, lds);
    if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_wide)");
    if (nbuf == 2) hipLaunchKernelGGL((k_mlp_wide<H, TW, 2, SAMPLE>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((k_mlp_wide<H, TW, 1, SAMPLE>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return check_launch("k_mlp_wide");
}

// tiles (32 envs) up to which the split kernel is used: measured crossover (tools/time_mlp_sizes.py)
#ifndef QENV_MLP_SMALL_TILES
#define QENV_MLP_SMALL_TILES 512
#endif

// QENV_MLP_BF16: 3 (default) = k_mlp_v3 for the N = 128, T = 5 policy: layers 1 and 2 on the bf16 matrix cores with three-way
// split operands, layer 3 on the vector lanes, prepared weight image, prefetched windows; 2 = round 2's kernel (layers 1
// and 2 split-bf16, layer 3 on the fp32 MFMA), 1 = layer 2 only, 0 = every layer on the fp32 MFMA (A/B runs)
static inline int mlp_bf16_layer2() {
    static const int on = getenv("QENV_MLP_BF16") ? atoi(getenv("QENV_MLP_BF16")) : 3;
    return on;
}

template <int H, int TW, bool SAMPLE = false> static int launch_mlp_small(const MlpArgs& a, hipStream_t stream) {
    constexpr int HP = H < 32 ? 32 : H;
    const int64_t n_tiles = ((a.N + 31) / 32) * a.n_chunks;
    const dim3 g((unsigned)(n_tiles < 4096 ? n_tiles : 4096)), b(HP * 2);
    if constexpr (H == 128 && TW == 5) {
        if (mlp_bf16_layer2() >= 3) {
            hipLaunchKernelGGL((k_mlp_small<H, TW, SAMPLE, 3>), g, b, 0, stream, a);
            return check_launch("k_mlp_small(bf16 layers 1+2, vector layer 3)");
        }
        if (mlp_bf16_layer2() >= 2) {
            hipLaunchKernelGGL((k_mlp_small<H, TW, SAMPLE, 2>), g, b, 0, stream, a);
            return check_launch("k_mlp_small(bf16 layers 1+2)");
        }
        if (mlp_bf16_layer2() == 1) {
            hipLaunchKernelGGL((k_mlp_small<H, TW, SAMPLE, 1>), g, b, 0, stream, a);
            return check_launch("k_mlp_small(bf16 layer 2)");
        }
    }
    hipLaunchKernelGGL((k_mlp_small<H, TW, SAMPLE>), g, b, 0, stream, a);
    return check_launch("k_mlp_small");
}

template <int H, int TW, int BLOCK, bool W2_LDS, bool SAMPLE = false>
static int launch_mlp(const MlpArgs& a, int T, hipStream_t stream) {
    constexpr int HP = H < 32 ? 32 : H;
    if (a.list_idx && !(H >= 64 && H <= 128 && TW > 0)) return fail(QENV_ERR_ARG, "policy_act_subset: list mode needs hidden 64 / 128 and T = 5");
    if constexpr (H >= 64 && H <= 128 && TW > 0) {  // >= 2 waves per tile to split over, the reference's window
        static const int small_tiles = getenv("QENV_MLP_SMALL_TILES") ? atoi(getenv("QENV_MLP_SMALL_TILES")) : QENV_MLP_SMALL_TILES;
        if (a.list_idx || ((a.N + 31) / 32) * a.n_chunks <= small_tiles) return launch_mlp_small<H, TW, SAMPLE>(a, stream);
    }
    const int K1P = (15 * T + 1) & ~1;
    if (TW && (int64_t)15 * TW * a.N >= ((int64_t)1 << 31))
        return fail(QENV_ERR_ARG, "qenv_mlp_forward: 15*T*n_envs must stay below 2^31 (32-bit ring offsets)");
    const int64_t n_tiles = ((a.N + 31) / 32) * a.n_chunks;
    const int wpb = BLOCK / 64;
    int64_t blocks = (n_tiles + wpb - 1) / wpb;
    if (blocks > 256) blocks = 256;  // one resident workgroup per CU; waves grid-stride over env tiles
    if constexpr (H == 128 && TW == 5 && W2_LDS) {
        // layer 2 on the bf16 matrix cores with three-way split operands (see k_mlp): the default for the reference's
        // N = 128 policy at batch sizes beyond k_mlp_small's; QENV_MLP_BF16=0 keeps the all-fp32-MFMA kernel
        constexpr int NT = HP / 32, KS16 = (15 * TW + 15) / 16;
        if (mlp_bf16_layer2() >= 3 && (int64_t)15 * TW * a.N * 4 < ((int64_t)1 << 31)) {  // round 3's kernel (see k_mlp_v3; its window
            // loads are raw buffer loads: one chunk of the window must stay below 2 GiB)
            const size_t lds = sizeof(uint4) * (size_t)mlp3::IMAGE_UINT4 + 80 * sizeof(uint32_t);
            hipError_t e = allow_dynamic_lds((const void*)k_mlp_v3<BLOCK, SAMPLE>, lds);
            if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_v3)");
            hipLaunchKernelGGL((k_mlp_v3<BLOCK, SAMPLE>), dim3((unsigned)blocks), dim3(BLOCK), lds, stream, a);
            return check_launch("k_mlp_v3");
        }
    }
    if (a.obs_rows_from != 0 || (a.win_chunk_stride != 0 && a.win_chunk_stride != (int64_t)15 * T * a.N))
        return fail(QENV_ERR_ARG, "the slice layout of stored observations (obs_slices / chunk stride) is implemented for hidden 64 / 128 "
                                  "with T = 5 (k_mlp_small, k_mlp_v3)");
    if constexpr (H == 128 && TW == 5 && W2_LDS) {
        constexpr int NT = HP / 32, KS16 = (15 * TW + 15) / 16;
        if (mlp_bf16_layer2() >= 2) {
            const size_t lds = sizeof(float) * ((size_t)NT * KS16 * 3 * 64 * 4 + (size_t)NT * NT * 2 * 3 * 64 * 4 + 4 * HP + 2 * HP + 32);
            if (lds > 160 * 1024) return fail(QENV_ERR_ARG, "qenv_mlp_forward: weights do not fit the 160 KiB LDS");
            hipError_t e = allow_dynamic_lds((const void*)k_mlp<H, TW, BLOCK, W2_LDS, SAMPLE, 2>, lds);
            if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp bf16)");
            hipLaunchKernelGGL((k_mlp<H, TW, BLOCK, W2_LDS, SAMPLE, 2>), dim3((unsigned)blocks), dim3(BLOCK), lds, stream, a);
            return check_launch("k_mlp(bf16 layers 1+2)");
        }
        if (mlp_bf16_layer2() == 1) {
            const size_t lds = sizeof(float) * ((size_t)K1P * HP + (size_t)NT * NT * 2 * 3 * 64 * 4 + 32 * HP + 2 * HP + 32);
            if (lds > 160 * 1024) return fail(QENV_ERR_ARG, "qenv_mlp_forward: weights do not fit the 160 KiB LDS");
            hipError_t e = allow_dynamic_lds((const void*)k_mlp<H, TW, BLOCK, W2_LDS, SAMPLE, 1>, lds);
            if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp bf16)");
            hipLaunchKernelGGL((k_mlp<H, TW, BLOCK, W2_LDS, SAMPLE, 1>), dim3((unsigned)blocks), dim3(BLOCK), lds, stream, a);
            return check_launch("k_mlp(bf16 layer 2)");
        }
    }
    const size_t lds = sizeof(float) * ((size_t)K1P * HP + (W2_LDS ? (size_t)HP * HP : 0) + 32 * HP + 2 * HP + 32);
    if (lds > 160 * 1024) return fail(QENV_ERR_ARG, "qenv_mlp_forward: weights do not fit the 160 KiB LDS (reduce T)");
    hipError_t e = allow_dynamic_lds((const void*)k_mlp<H, TW, BLOCK, W2_LDS, SAMPLE>, lds);
    if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp)");
    hipLaunchKernelGGL((k_mlp<H, TW, BLOCK, W2_LDS, SAMPLE>), dim3((unsigned)blocks), dim3(BLOCK), lds, stream, a);
    return check_launch("k_mlp");
}

// N = 256: the LDS-staged wide kernel whenever W1 and one W2 block fit (T <= 8); QENV_MLP_WIDE=0 keeps the kernel it
// replaced (W2 read from global memory inside the MFMA chain) for A/B runs
template <bool SAMPLE> static int launch_mlp_256(const MlpArgs& a, int T, hipStream_t stream) {
    if (a.obs_rows_from != 0 || (a.win_chunk_stride != 0 && a.win_chunk_stride != (int64_t)15 * T * a.N))
        return fail(QENV_ERR_ARG, "the slice layout of stored observations is implemented for hidden 64 / 128 with T = 5");
    static const int wide = getenv("QENV_MLP_WIDE") ? atoi(getenv("QENV_MLP_WIDE")) : 1;
    // QENV_MLP_WIDE: 1 (default) = k_mlp_wide_v3 for T = 5 (split-bf16 matrix cores), at every batch size (one kernel: results do not
    // depend on how a batch is sharded); 2 = round 2's fp32-MFMA wide kernel; 0 = the kernel that one replaced
    if (wide == 1 && !a.list_idx && T == 5 && (int64_t)15 * 5 * a.N * 4 < ((int64_t)1 << 31)) {
        const int64_t n_tiles = ((a.N + 31) / 32) * a.n_chunks;
        int64_t blocks = (n_tiles + 3) / 4;
        if (blocks > 256) blocks = 256;
        const size_t lds = sizeof(uint4) * (size_t)wide3::LDS_UINT4;
        hipError_t e = allow_dynamic_lds((const void*)k_mlp_wide_v3<SAMPLE>, lds);
        if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_wide_v3)");
        hipLaunchKernelGGL((k_mlp_wide_v3<SAMPLE>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
        return check_launch("k_mlp_wide_v3");
    }
    if (wide && !a.list_idx) {
        bool fits = false;
        const int rc = T == 5 ? launch_mlp_wide<256, 5, SAMPLE>(a, T, stream, fits) : launch_mlp_wide<256, 0, SAMPLE>(a, T, stream, fits);
        if (fits) return rc;
    }
    return launch_mlp<256, 0, 256, false, SAMPLE>(a, T, stream);
}

// ---- dense tail of rl_nn_model.py: K -> H -> H -> n_out with K too large for LDS-resident W1 ------
// Same operand orientation as k_mlp.  W1 is streamed through LDS in chunks of KC input features (rows
// padded by one word so the transposing staging writes are conflict-free); the four waves of a workgroup
// march over the chunks together, each accumulating its own 32-env tile.
struct DenseArgs {
    const float* x;  // [n_chunks][K][N] (feature-major), or [n_chunks][N][K] (rows: sample-major, torch's [batch, features])
    int64_t N;
    int32_t K, n_out, final_tanh, n_chunks, rows;
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    float* out;
};

template <int H>
__global__ __launch_bounds__(256) void k_dense_tail(const DenseArgs a) {
    constexpr int HP = H < 32 ? 32 : H, NT = HP / 32, KC = 64, LDW = HP + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W1c = lds;                   // [KC][H+1]
    float* W2t = W1c + KC * LDW;        // [H][H]  k-major
    float* W3t = W2t + HP * HP;         // [HP][32] k-major, output rows >= n_out zero (layer 3 runs on the MFMA, see k_mlp)
    float* b1s = W3t + 32 * HP;
    float* b2s = b1s + HP;
    float* b3s = b2s + HP;
    for (int idx = threadIdx.x; idx < HP * HP; idx += 256) { const int k = idx / HP, h = idx % HP; W2t[idx] = (k < H && h < H) ? a.W2[(size_t)h * H + k] : 0.f; }
    for (int idx = threadIdx.x; idx < 32 * HP; idx += 256) { const int h = idx / 32, o = idx % 32; W3t[idx] = (o < a.n_out && h < H) ? a.W3[(size_t)o * H + h] : 0.f; }
    for (int h = threadIdx.x; h < HP; h += 256) { b1s[h] = h < H ? a.b1[h] : 0.f; b2s[h] = h < H ? a.b2[h] : 0.f; }
    if (threadIdx.x < 32) b3s[threadIdx.x] = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;

    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles_per_chunk = (a.N + 31) / 32;
    const int64_t n_tiles = tiles_per_chunk * a.n_chunks;  // chunk c: x + c K N -> out + c n_out N (time-major rollouts)
    const int64_t rounds = (n_tiles + (int64_t)gridDim.x * 4 - 1) / ((int64_t)gridDim.x * 4);
    const uint32_t N32 = (uint32_t)a.N;
    for (int64_t r = 0; r < rounds; ++r) {
        const int64_t gtile = (r * gridDim.x + blockIdx.x) * 4 + wave;
        const int64_t chunk = gtile < n_tiles ? gtile / tiles_per_chunk : 0;
        const int64_t tile = gtile - chunk * tiles_per_chunk;
        const float* __restrict__ xc = a.x + chunk * (int64_t)a.K * a.N;
        float* __restrict__ outc = a.out + chunk * (int64_t)a.n_out * a.N;
        const int64_t e = tile * 32 + col;
        const bool valid = gtile < n_tiles && e < a.N;
        const uint32_t e32 = (uint32_t)(valid ? e : a.N - 1);
        f32x16 acc1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = 0.f;
        // the NEXT chunk of W1 travels in registers: its global loads are issued right after the current chunk has been
        // copied to LDS and land while the chunk's MFMAs run (the load latency used to sit between the two barriers)
        constexpr int WPT = KC * HP / 256;
        float wpre[WPT];
        auto fetch_w1 = [&](int kc) {
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int idx = threadIdx.x + 256 * i, h = idx / KC, k = idx % KC;  // global reads run along k (coalesced)
                wpre[i] = ((kc + k) < a.K && h < H) ? a.W1[(size_t)h * a.K + kc + k] : 0.f;
            }
        };
        fetch_w1(0);
        auto load_x = [&](int k0, float (&dst)[8]) {  // 8 k-steps from input feature k0: k = k0 + 2 s + half
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int k = k0 + 2 * s + half;
                const uint32_t kk = (uint32_t)(k < a.K ? k : a.K - 1);
                const float v = xc[kk * N32 + e32];
                dst[s] = k < a.K ? v : 0.f;
            }
        };
        float xs[8];
        load_x(0, xs);
        for (int kc = 0; kc < a.K; kc += KC) {
            __syncthreads();  // every wave is done with the previous chunk (and with the one-off staging above)
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const int idx = threadIdx.x + 256 * i, h = idx / KC, k = idx % KC;
                W1c[k * LDW + h] = wpre[i];  // LDS writes stride HP+1 words
            }
            __syncthreads();
            if (kc + KC < a.K) fetch_w1(kc + KC);
#pragma unroll
            for (int s0 = 0; s0 < KC / 2; s0 += 8) {
                // the activations of the NEXT group of 8 k-steps are requested before this group's MFMAs are issued: a
                // group's 8 NT dependent MFMAs (512 cycles per chain) are shorter than an HBM round trip
                float xn[8];
                load_x(kc + 2 * (s0 + 8), xn);
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1c[(2 * (s0 + s) + half) * LDW + 32 * t + col], xs[s], acc1[t], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 8; ++s) xs[s] = xn[s];
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh(acc1[t][j] + b1s[32 * t + acc_row(j, half)]);
        f32x16 acc3;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc3[j] = b3s[acc_row(j, half)];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 acc2;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = b2s[32 * t + acc_row(j, half)];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(W2t[(32 * ti + acc_row(j, half)) * HP + 32 * t + col], acc1[ti][j], acc2, 0, 0, 0);
                    if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = fast_tanh(acc2[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(W3t[(32 * t + acc_row(j, half)) * 32 + col], acc2[j], acc3, 0, 0, 0);
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = acc3[o];  // output row o is held by the lower lane half
            if (a.final_tanh) v = tanhf(v);
            if (valid && half == 0 && o < a.n_out) outc[(int64_t)o * a.N + e] = v;
        }
    }
}

// ---- k_dense_tail_v3: the H = 64 dense tail (rl_nn_model.py: 800 -> 64 -> 64 -> 3) on the bf16 matrix cores ---------------------
// Three-way split operands (k_mlp_v3's arithmetic: six v_mfma_f32_32x32x16_bf16 of 32 cycles per 16 k and output tile instead of
// eight v_mfma_f32_32x32x2_f32 of 64; pre-scaled hidden layers; layer 3 on the vector lanes).  W1 (K x 64: 200 KiB at K = 800) does
// not fit the LDS in any form: it streams through in chunks of 64 k as READY fragments -- every thread of the 512 loads the eight
// consecutive k of one (row, k-step, lane half) straight from the torch-layout rows (32 contiguous bytes), splits them and writes
// the three bf16 terms where the MFMA reads them; double-buffered, one barrier per chunk, the next chunk's weights and the next
// chunk's features in flight during the current chunk's 48 MFMAs.  One workgroup of eight waves per CU: a W1 chunk is staged
// once for eight 32-sample tiles (the vector L1 passes ~12 B per clock and CU: the features are 100 KiB per tile, W1's share 25).
// Needs K % 8 == 0 (aligned 32-byte fragment loads); other shapes keep k_dense_tail.
namespace dt3 {
constexpr int H = 64, NT = 2, KC = 64, KSC = KC / 16;             // k-steps of 16 per chunk
constexpr int W1B = NT * KSC * 3 * 64;                            // uint4 per W1 chunk buffer (1 536 -> 24 KiB)
constexpr int W2F = 2 * W1B;                                      // W2 fragments: [t][ti][jh][3][64] (1 536 uint4)
constexpr int W3T = W2F + NT * NT * 2 * 3 * 64;                   // [64] float4 rows of W3
constexpr int B1 = W3T + H, B2 = B1 + H / 4, B3 = B2 + H / 4, LDS_UINT4 = B3 + 1;
}  // namespace dt3

// in-kernel stamps (diagnostic builds: tools/build_variant.sh dt3stamps "-DQENV_DT3_STAMPS" qenv_ppo.hip; tests/tools/dense_tail_stamps.py):
// chunk 6 of workgroup 7's first round, s_memtime per wave at: chunk top, operands split, barrier passed, loads issued, products done
#ifdef QENV_DT3_STAMPS
__device__ unsigned long long dt3_stamps[8 * 8];
#define DT3_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (stamp_on && lane == 0) dt3_stamps[wave * 8 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DT3_STAMP(k)
#endif

template <bool ROWS>   // ROWS: the features sample-major (DenseArgs::rows) -- a template parameter: a run-time branch per k-step
__global__ __launch_bounds__(512) void k_dense_tail_v3(const DenseArgs a) {   // would cut the product loop into basic blocks
    using namespace dt3;
    constexpr int BLOCK = 512, WPB = 8;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    // ---- layer 2 / 3 weights and the biases, once per launch and workgroup
    for (int c = threadIdx.x; c < NT * NT * 2 * 64; c += BLOCK) {   // W2: lane (col, half) holds W2[32 t + col][32 ti + acc_row(8 jh + i, half)]
        const int ln = c & 63, jh = (c >> 6) & 1, ti = (c >> 7) % NT, t = (c >> 7) / NT;
        const float* src = a.W2 + (size_t)(32 * t + (ln & 31)) * H + 32 * ti + 16 * jh + 4 * (ln >> 5);
        const float w8[8] = {src[0] * TANH_PRESCALE, src[1] * TANH_PRESCALE, src[2] * TANH_PRESCALE, src[3] * TANH_PRESCALE,
                             src[8] * TANH_PRESCALE, src[9] * TANH_PRESCALE, src[10] * TANH_PRESCALE, src[11] * TANH_PRESCALE};
        bf16x8 sp[3];
        split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) lds4[W2F + (((t * NT + ti) * 2 + jh) * 3 + k) * 64 + ln] = __builtin_bit_cast(uint4, sp[k]);
    }
    for (int h = threadIdx.x; h < H; h += BLOCK) {
        float4 w;
        w.x = a.W3[h]; w.y = a.n_out > 1 ? a.W3[(size_t)H + h] : 0.f; w.z = a.n_out > 2 ? a.W3[(size_t)2 * H + h] : 0.f;
        w.w = a.n_out > 3 ? a.W3[(size_t)3 * H + h] : 0.f;
        lds4[W3T + h] = __builtin_bit_cast(uint4, w);
        reinterpret_cast<float*>(lds4 + B1)[h] = a.b1[h] * TANH_PRESCALE;
        reinterpret_cast<float*>(lds4 + B2)[h] = a.b2[h] * TANH_PRESCALE;
    }
    if (threadIdx.x < 4) reinterpret_cast<float*>(lds4 + B3)[threadIdx.x] = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;
    __syncthreads();   // (the first tile's accumulators start from b1 right below, ahead of the chunk loop's first barrier)
    const float4* W3t = reinterpret_cast<const float4*>(lds4 + W3T);
    const float* b1s = reinterpret_cast<const float*>(lds4 + B1);
    const float* b2s = reinterpret_cast<const float*>(lds4 + B2);
    const float* b3s = reinterpret_cast<const float*>(lds4 + B3);

    // (the wave index as a SCALAR: the feature loads' buffer resource is built from it -- taken from a vector register, hipcc
    // wrapped every one of a tile's 416 loads in a 14-instruction waterfall loop: 5.8 k of the 7.6 k vector instructions per tile
    // that the counters showed; 153 -> 134.5 us per 131 072 samples, round 5)
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tiles_per_chunk = (a.N + 31) / 32;
    const int64_t n_tiles = tiles_per_chunk * a.n_chunks;
    const int64_t rounds = (n_tiles + (int64_t)gridDim.x * WPB - 1) / ((int64_t)gridDim.x * WPB);
    const uint32_t N32 = (uint32_t)a.N;
    const int n_kc = (a.K + KC - 1) / KC;
    // this thread's W1 fragment of a chunk: output row 32 wt + (wl & 31), k-step ws, lane half wl >> 5
    const int wl = threadIdx.x & 63, ws = (threadIdx.x >> 6) & 3, wt = threadIdx.x >> 8;
    const float* w1row = a.W1 + (size_t)(32 * wt + (wl & 31)) * a.K + 16 * ws + 8 * (wl >> 5);
    for (int64_t r = 0; r < rounds; ++r) {
        asm volatile("" ::: "memory");
        const int64_t gtile = (r * gridDim.x + blockIdx.x) * WPB + wave;
        const int64_t chunk = gtile < n_tiles ? gtile / tiles_per_chunk : 0;
        const int64_t tile = gtile - chunk * tiles_per_chunk;
        float* __restrict__ outc = a.out + chunk * (int64_t)a.n_out * a.N;
        const int64_t e = tile * 32 + col;
        const bool valid = gtile < n_tiles && e < a.N;
        // the features of a tile: element (s, i) of a lane = row kc + 16 s + 8 half + i, raw buffer loads (the lane's offset
        // carries the sample and the half's 8 rows; rows >= K fall out of the range check)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + chunk * (int64_t)a.K * a.N), 0,
                                                                            (int)((uint32_t)a.K * N32 * 4u), 0x00020000);
        // rows (the reference's own layout: rl_nn_model.py flattens its conv features to [batch, 800]): a lane's fragment of a
        // k-step -- eight consecutive k of ONE sample -- is 32 contiguous bytes: two 16-byte loads instead of eight 4-byte ones
        // 512 KiB apart, and no transposed copy of the features for the caller to make.  (Bare, the chip streams this pattern
        // at 5.1-5.5 TB/s and the feature-major one at 1.3, tools/ubench/read_patterns.hip -- but this kernel runs 132-138 us
        // on either: how it reads is not what bounds it.)
        const uint32_t voff = !valid ? 0x80000000u
                              : ROWS ? ((uint32_t)e * (uint32_t)a.K + 8u * (uint32_t)half) * 4u
                                       : ((uint32_t)e + 8u * (uint32_t)half * N32) * 4u;
#ifndef QENV_DT3_RING
#define QENV_DT3_RING 1   // feature chunks in flight: chunk ic + RING is requested between the products of chunk ic.  2: 120-126 us against
                          // 130-134 on one box, 122-128 against 122-124 on another (within the boxes' noise); 3 spills (193 us): 1 stays
#endif
        float xring[QENV_DT3_RING][KSC][8];
        auto request_x_step = [&](int kc, int s, float (&xw)[KSC][8]) {   // the eight features of k-step s of the chunk at kc
#ifdef QENV_DT3_RUNTIME_ROWS   /* A/B: the layout as a run-time branch per k-step (basic-block boundaries around the loads) */
            if (a.rows) {
#else
            if constexpr (ROWS) {
#endif
                const bool in = kc + 16 * s + 8 * half < a.K;
                const uint32_t vo = in ? voff : 0x80000000u;
                const auto lo = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (uint32_t)(kc + 16 * s) * 4u, 0);
                const auto hi = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (uint32_t)(kc + 16 * s + 4) * 4u, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) { xw[s][i] = __uint_as_float(lo[i]); xw[s][4 + i] = __uint_as_float(hi[i]); }
                return;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xw[s][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (uint32_t)(kc + 16 * s + i) * N32 * 4u, 0));
        };
        auto request_x = [&](int kc, float (&xw)[KSC][8]) {
            if constexpr (ROWS) {
#pragma unroll
                for (int s = 0; s < KSC; ++s) {
                    const bool in = kc + 16 * s + 8 * half < a.K;   // (K % 8 == 0: a fragment is inside or outside as a whole;
                    const uint32_t vo = in ? voff : 0x80000000u;    //  outside would be the NEXT sample's features, not zeros)
                    const auto lo = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (uint32_t)(kc + 16 * s) * 4u, 0);
                    const auto hi = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (uint32_t)(kc + 16 * s + 4) * 4u, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { xw[s][i] = __uint_as_float(lo[i]); xw[s][4 + i] = __uint_as_float(hi[i]); }
                }
                return;
            }
#pragma unroll
            for (int s = 0; s < KSC; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    xw[s][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (uint32_t)(kc + 16 * s + i) * N32 * 4u, 0));
        };
        float4 wlo, whi;
        auto request_w1 = [&](int kc) {   // (k beyond K: zeros -- K % 8 == 0, so a fragment is inside or outside as a whole)
            const bool in = kc + 16 * ws + 8 * (wl >> 5) < a.K;
            const float4* p4 = reinterpret_cast<const float4*>(w1row + (in ? kc : 0));
            wlo = p4[0]; whi = p4[1];
            if (!in) { wlo = make_float4(0.f, 0.f, 0.f, 0.f); whi = wlo; }
        };
        f32x16 acc1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = b1s[32 * t + acc_row(j, half)];
        request_w1(0);
#pragma unroll
        for (int b = 0; b < QENV_DT3_RING; ++b)
            if (b < n_kc) request_x(b * KC, xring[b]);
        auto do_chunk = [&](int ic, float (&xw)[KSC][8]) {
            const int kc = ic * KC;
            [[maybe_unused]] const bool stamp_on = blockIdx.x == 7 && r == 0 && (ic == 6 || ic == 7);
            if (ic == 6) DT3_STAMP(0);
            {   // this thread's fragment of chunk ic -> its three bf16 terms in buffer ic & 1
                const float w8[8] = {wlo.x * TANH_PRESCALE, wlo.y * TANH_PRESCALE, wlo.z * TANH_PRESCALE, wlo.w * TANH_PRESCALE,
                                     whi.x * TANH_PRESCALE, whi.y * TANH_PRESCALE, whi.z * TANH_PRESCALE, whi.w * TANH_PRESCALE};
                bf16x8 sp[3];
                split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
                for (int k = 0; k < 3; ++k) lds4[(ic & 1) * W1B + ((wt * KSC + ws) * 3 + k) * 64 + wl] = __builtin_bit_cast(uint4, sp[k]);
            }
            bf16x8 xsp[KSC][3];
#pragma unroll
            for (int s = 0; s < KSC; ++s) split3x8(xw[s], xsp[s][0], xsp[s][1], xsp[s][2]);
            if (ic == 6) DT3_STAMP(1);
            __syncthreads();   // chunk ic is staged (and every wave has left chunk ic - 1, whose buffer chunk ic + 1 will overwrite)
            if (ic == 6) DT3_STAMP(2);
#ifndef QENV_DT3_SPREAD
#define QENV_DT3_SPREAD 1   // 1: the next chunk's feature loads are issued k-step by k-step between this chunk's products instead of
#endif                      //    in one burst behind the barrier (the stamps: the burst takes the last waves 5 k cycles to get accepted
                            //    -- the CU's memory path passes ~12.6 B per clock -- and nothing is requested while they multiply)
            if (ic + 1 < n_kc) request_w1(kc + KC);
            if (!QENV_DT3_SPREAD && ic + QENV_DT3_RING < n_kc) request_x(kc + QENV_DT3_RING * KC, xw);
            if (ic == 6) DT3_STAMP(3);
            const uint4* frag = lds4 + (ic & 1) * W1B + lane;
#ifndef QENV_DT3_PIPE
#define QENV_DT3_PIPE 1   // 1: a k-step's six weight fragments are read while the k-step before multiplies, the two output
#endif                    //    tiles' products alternate (A/B: -DQENV_DT3_PIPE=0)
#if QENV_DT3_PIPE
            {
                bf16x8 A[2][NT][3];
                auto ldA = [&](int s_, bf16x8 (&f_)[NT][3]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int k = 0; k < 3; ++k) f_[t][k] = __builtin_bit_cast(bf16x8, frag[((t * KSC + s_) * 3 + k) * 64]);
                };
                __builtin_amdgcn_sched_barrier(0);
                ldA(0, A[0]);
#pragma unroll
                for (int s = 0; s < KSC; ++s) {
                    const int cu = s & 1;
                    if (s + 1 < KSC) ldA(s + 1, A[cu ^ 1]);
                    if (QENV_DT3_SPREAD && ic + QENV_DT3_RING < n_kc) request_x_step(kc + QENV_DT3_RING * KC, s, xw);
                    // (the same six products per tile in the same order: bit-identical sums)
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][0][1], xsp[s][1], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][1][1], xsp[s][1], acc1[1], 0, 0, 0);
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][0][2], xsp[s][0], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][1][2], xsp[s][0], acc1[1], 0, 0, 0);
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][0][0], xsp[s][2], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][1][0], xsp[s][2], acc1[1], 0, 0, 0);
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][0][1], xsp[s][0], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][1][1], xsp[s][0], acc1[1], 0, 0, 0);
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][0][0], xsp[s][1], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][1][0], xsp[s][1], acc1[1], 0, 0, 0);
                    acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][0][0], xsp[s][0], acc1[0], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[cu][1][0], xsp[s][0], acc1[1], 0, 0, 0);
                }
                // the order asked of the scheduler: a k-step's fragment reads and its share of the next chunk's feature loads in
                // front of the k-step's products (left alone hipcc gathers all the loads at the top of the loop: the burst again;
                // in four pieces between the products: 124 us against 116)
                constexpr int NLD = ROWS ? 2 : 8;
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                for (int s = 0; s < KSC; ++s) {
                    if (s + 1 < KSC) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                    if (QENV_DT3_SPREAD && ic + QENV_DT3_RING < n_kc) __builtin_amdgcn_sched_group_barrier(0x020, NLD, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ic == 6) DT3_STAMP(4);
            if (ic == 7) DT3_STAMP(5);   // (the same point one chunk later: a whole chunk period)
#else
#pragma unroll
            for (int s = 0; s < KSC; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const uint4* f = frag + ((t * KSC + s) * 3) * 64;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                                 al = __builtin_bit_cast(bf16x8, f[128]);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[s][1], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xsp[s][0], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[s][2], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xsp[s][0], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[s][1], acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xsp[s][0], acc1[t], 0, 0, 0);
                }
#endif
        };
        for (int ic = 0; ic < n_kc; ic += QENV_DT3_RING) {   // (the ring slot is a compile-time index: the registers of a slot)
#pragma unroll
            for (int b = 0; b < QENV_DT3_RING; ++b)
                if (ic + b < n_kc) do_chunk(ic + b, xring[b]);
        }
        __syncthreads();   // every wave has left the last chunk: the next round restages buffer 0
        // ---- layer 2 on the matrix cores, layer 3 on the vector lanes (see k_mlp_v3)
        bf16x8 bsp[NT][2][3];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = fast_tanh_prescaled(acc1[t][j]);
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = acc1[t][8 * jh + i];
                split3x8(v8, bsp[t][jh][0], bsp[t][jh][1], bsp[t][jh][2]);
            }
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = half ? 0.f : b3s[k];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 acc2;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = b2s[32 * t + acc_row(j, half)];
            const uint4* frag = lds4 + W2F + (size_t)t * NT * 2 * 3 * 64 + lane;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const uint4* f = frag + (ti * 2 + jh) * 3 * 64;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, f[0]), am = __builtin_bit_cast(bf16x8, f[64]),
                                 al = __builtin_bit_cast(bf16x8, f[128]);
                    const bf16x8 bh = bsp[ti][jh][0], bm = bsp[ti][jh][1], bl = bsp[ti][jh][2];
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2, 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc2[j] = fast_tanh_prescaled(acc2[j]);
            mlp3_layer3(W3t, t, half, acc2, o);
        }
        mlp3_layer3_finish(o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = o[k];
            if (a.final_tanh) v = tanhf(v);
            if (valid && half == 0 && k < a.n_out) outc[(int64_t)k * a.N + e] = v;
        }
    }
}

template <int H> static int launch_dense(const DenseArgs& a, hipStream_t stream) {
    constexpr int HP = H < 32 ? 32 : H;
    if constexpr (H == 64) {   // QENV_DENSE_BF16=0 keeps the fp32-MFMA kernel (A/B runs)
        static const int bf16 = getenv("QENV_DENSE_BF16") ? atoi(getenv("QENV_DENSE_BF16")) : 1;
        const int64_t n_tiles3 = ((a.N + 31) / 32) * a.n_chunks;
        if (bf16 && a.K % 8 == 0 && n_tiles3 >= 2048) {
            const size_t lds3 = sizeof(uint4) * (size_t)dt3::LDS_UINT4;
            const void* fn3 = a.rows ? (const void*)k_dense_tail_v3<true> : (const void*)k_dense_tail_v3<false>;
            hipError_t e3 = allow_dynamic_lds(fn3, lds3);
            if (e3 != hipSuccess) return fail_hip(e3, "hipFuncSetAttribute(k_dense_tail_v3)");
            int64_t blocks3 = (n_tiles3 + 7) / 8;
            if (blocks3 > 256) blocks3 = 256;
            if (a.rows) hipLaunchKernelGGL(k_dense_tail_v3<true>, dim3((unsigned)blocks3), dim3(512), lds3, stream, a);
            else hipLaunchKernelGGL(k_dense_tail_v3<false>, dim3((unsigned)blocks3), dim3(512), lds3, stream, a);
            return check_launch("k_dense_tail_v3");
        }
    }
    if (a.rows)
        return fail(QENV_ERR_ARG, "qenv_dense_tail_rows: sample-major features are read by k_dense_tail_v3 only (hidden 64, K % 8 == 0, "
                                  ">= 65 536 samples); other shapes: feature-major x [K][N]");
    const size_t lds = sizeof(float) * (64 * (HP + 1) + HP * HP + 32 * HP + 2 * HP + 32);
    hipError_t e = allow_dynamic_lds((const void*)k_dense_tail<H>, lds);
    if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_dense_tail)");
    const int64_t n_tiles = ((a.N + 31) / 32) * a.n_chunks;
    int64_t blocks = (n_tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL((k_dense_tail<H>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return check_launch("k_dense_tail");
}

// ActorCritic.act sampling (model.py:62-68): a = mean + std*eps, per-dimension Normal log-prob
template <typename T>
__global__ __launch_bounds__(256) void k_sample_action(int64_t N, const T* mean, T std, T log_norm, const T* noise,
                                                       uint64_t seed, uint64_t step, int64_t env_offset, T* out_action,
                                                       T* out_logprob) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    float z[4];
    if (noise) {
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (float)noise[(int64_t)k * N + e];
    } else {
        const uint64_t g = (uint64_t)(env_offset + e);
        uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0xAC710Du);
        box_muller(c[0], c[1], z[0], z[1]);
        box_muller(c[2], c[3], z[2], z[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const T eps = (T)z[k];
        out_action[(int64_t)k * N + e] = mean[(int64_t)k * N + e] + std * eps;
        if (out_logprob) out_logprob[(int64_t)k * N + e] = T(-0.5) * eps * eps + log_norm;
    }
}

// random branch of quad.reset (quadrotor_env.py:440-445) with Philox streams
template <typename T>
__global__ __launch_bounds__(256) void k_sample_init(int64_t N, uint64_t seed, uint64_t episode, int64_t env_offset,
                                                     const uint8_t* mask, T* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    if (mask && mask[e] == 0) return;
    float x[13];
    sample_init_state(seed, episode, (uint64_t)(env_offset + e), x);
#pragma unroll
    for (int k = 0; k < 13; ++k) out[(int64_t)k * N + e] = (T)x[k];
}

// ---- sensor model, quadrotor_env.py:579-724 ------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_sensor_reset(int64_t N, const T* ws, const int32_t* wsi, T* sens, const T* u3, uint64_t seed,
                                                      uint64_t episode, int64_t env_offset, SensParams<T> p) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    T u[3];
    if (u3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = u3[(int64_t)k * N + e];
    } else {
        const uint64_t g = (uint64_t)(env_offset + e);
        uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)episode, (uint32_t)(episode >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5E45u);
        u[0] = (T)u01(c[0]); u[1] = (T)u01(c[1]); u[2] = (T)u01(c[2]);
    }
    // bias_reset :601-609 (p.bd already carries sensor.error)
#pragma unroll
    for (int k = 0; k < 3; ++k) sens[(int64_t)(SR_ABD + k) * N + e] = (u[k] - T(0.5)) * T(2) * p.bd[k];
    sens[(int64_t)SR_AB * N + e] = T(0);
    sens[(int64_t)SR_GB * N + e] = T(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sens[(int64_t)(SR_POS + k) * N + e] = ws[(int64_t)(ROW_STATE + 2 * k) * N + e];
        sens[(int64_t)(SR_VEL + k) * N + e] = ws[(int64_t)(ROW_STATE + 2 * k + 1) * N + e];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sens[(int64_t)(SR_Q + k) * N + e] = ws[(int64_t)(ROW_STATE + 6 + k) * N + e];
    sens[(int64_t)SR_ALIAS * N + e] = T(1);  // quaternion_t0 is a view of quad.state's array until the first gyro_int ...
    sens[(int64_t)SR_IRESET * N + e] = (T)(wsi[IROW_I * N + e] + 1);  // ... which is the env's state only until its next step
    // R keeps its value (the reference's reset() does not touch self.R)
}

// only bias_reset (:601-609): the N = 1 class calls it on its own at construction
template <typename T>
__global__ __launch_bounds__(256) void k_sensor_bias_reset(int64_t N, T* sens, const T* u3, SensParams<T> p) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) sens[(int64_t)(SR_ABD + k) * N + e] = (u3[(int64_t)k * N + e] - T(0.5)) * T(2) * p.bd[k];
}

template <typename T>
__device__ __forceinline__ void sens_env_from_rows(const T* ws, const int32_t* wsi, const T* sens, int64_t N, int64_t e, SensEnv<T>& ev) {
    T y[13], acc[3], w[4];
#pragma unroll
    for (int k = 0; k < 13; ++k) y[k] = ws[(int64_t)(ROW_STATE + k) * N + e];
    // mat_rot of the step's last RHS call: if an aliased gyro_int has advanced the env's quaternion since, rotate with the saved one
    const bool saved = sens[(int64_t)SR_ISAVE * N + e] == (T)(wsi[IROW_I * N + e] + 1);
    T yq[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) yq[k] = (saved && k >= 6 && k < 10) ? sens[(int64_t)(SR_QSAVE + k - 6) * N + e] : y[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = ws[(int64_t)(ROW_ACCEL + k) * N + e];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = ws[(int64_t)(ROW_W + k) * N + e];
    sens_env<T>(yq, acc, w, ev);
#pragma unroll
    for (int k = 6; k < 10; ++k) ev.y[k] = y[k];  // quad.state itself is the current one
}

// sensor_sp of rl_worker.py:164-175: accel_int, gyro_int, gyro, gps, triad -- 27 normals in that order
template <typename T>
__global__ __launch_bounds__(256) void k_sensor_read(int64_t N, T* ws, const int32_t* wsi, T* sens, const T* z27, uint64_t seed,
                                                     uint64_t step, int64_t env_offset, SensParams<T> p, T t_step, T* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    T z[28];
    if (z27) {
#pragma unroll
        for (int k = 0; k < 27; ++k) z[k] = z27[(int64_t)k * N + e];
    } else {
        sens_philox_normals<T>(seed, step, (uint64_t)(env_offset + e), z);
    }
    Sens<T> s;
    sens_load(sens, N, e, p, t_step, wsi[IROW_I * N + e], s);
    SensEnv<T> ev;
    sens_env_from_rows(ws, wsi, sens, N, e, ev);
    T q_before[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q_before[k] = ev.y[6 + k];
    T o[24];
    bool wrote_env;
    sens_sp<T>(s, ev, z, o, wrote_env);
    if (wrote_env) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ws[(int64_t)(ROW_STATE + 6 + k) * N + e] = ev.y[6 + k];
            sens[(int64_t)(SR_QSAVE + k) * N + e] = q_before[k];
        }
        sens[(int64_t)SR_ISAVE * N + e] = (T)(wsi[IROW_I * N + e] + 1);
    }
    sens[(int64_t)SR_ALIAS * N + e] = T(0);  // gyro_int rebinds quaternion_t0 to a new array (q / |q|, :721)
#pragma unroll
    for (int k = 0; k < 24; ++k) out[(int64_t)k * N + e] = o[k];
    sens_store(sens, N, e, s);
}

// any sequence of the class's methods (the N = 1 drop-in `sensor` calls them one at a time): ops[0..n_ops), their normals
// consecutively in z, their outputs consecutively in out (sop_normals / sop_outputs rows each)
struct SensProgram { int32_t n_ops; int32_t ops[8]; };
template <typename T>
__global__ __launch_bounds__(256) void k_sensor_ops(int64_t N, T* ws, const int32_t* wsi, T* sens, const T* zin, SensProgram prog,
                                                    SensParams<T> p, T t_step, T* out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    Sens<T> s;
    sens_load(sens, N, e, p, t_step, wsi[IROW_I * N + e], s);
    SensEnv<T> ev;
    sens_env_from_rows(ws, wsi, sens, N, e, ev);
    int zi = 0, oi = 0;
    for (int k = 0; k < prog.n_ops; ++k) {
        const int op = prog.ops[k];  // wave-uniform
        T z[9], o[13];
        const int nz = sop_normals(op), no = sop_outputs(op);
        for (int j = 0; j < nz; ++j) z[j] = zin[(int64_t)(zi + j) * N + e];
        if (op == SOP_ACCEL) sop_accel(s, ev, z, o);
        else if (op == SOP_GYRO) sop_gyro(s, ev, z, o);
        else if (op == SOP_GPS) sop_gps(s, ev, z, o, o + 3);
        else if (op == SOP_TRIAD) {
            sop_triad<T>(s, ev, z, o);
#pragma unroll
            for (int j = 0; j < 9; ++j) o[4 + j] = s.R[j];
        } else if (op == SOP_ACCEL_INT) sop_accel_int<T>(s, ev, z, o, o + 3, o + 6);
        else {
            bool wrote_env;
            T q_before[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q_before[j] = ev.y[6 + j];
            sop_gyro_int<T>(s, ev, z, o, wrote_env);
            if (wrote_env) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ws[(int64_t)(ROW_STATE + 6 + j) * N + e] = ev.y[6 + j];
                    sens[(int64_t)(SR_QSAVE + j) * N + e] = q_before[j];
                }
                sens[(int64_t)SR_ISAVE * N + e] = (T)(wsi[IROW_I * N + e] + 1);
            }
            sens[(int64_t)SR_ALIAS * N + e] = T(0);  // gyro_int rebinds quaternion_t0 to a new array (q / |q|, :721)
        }
        for (int j = 0; j < no; ++j) out[(int64_t)(oi + j) * N + e] = o[j];
        zi += nz; oi += no;
    }
    sens_store(sens, N, e, s);
}

// ---- GAE ---------------------------------------------------------------------------------------
// One lane per env scans its own time series backwards (rows are time-major, so each step
// is a coalesced row access); the advantage moments for the global normalisation are reduced
// with wavefront shuffles and one double atomicAdd pair per wave.
template <typename T>
__global__ __launch_bounds__(256) void k_gae(int64_t N, int n_steps, const T* values, const T* masks, const T* rewards,
                                             T* returns, T* adv, double* stats, double* partials, const T* valid = nullptr) {
    // valid (in-band warm-up rollouts, qenv_ppo_step): rows with valid == 0 are the warm-up steps of a restarted env.  They
    // follow a terminal row (mask 0) or open the horizon, so no valid row's recurrence reads them; their advantage is set
    // to 0 and they are left out of the moments (count included).
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    if (e < N) {
        const T gmma = T(0.99), lmbda = T(0.99);
        T gae = T(0);
        T v_next = values[(int64_t)n_steps * N + e];
        // the recurrence is serial in time but its operands are not: 8 steps' worth of loads are issued back to back
        // before the first one is consumed (a plain loop exposes one HBM round trip per step: 107 us for 128 steps)
        constexpr int U = 8;
        for (int i0 = n_steps - 1; i0 >= 0; i0 -= U) {
            T v[U], mk[U], rw[U], vl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 - u;
                const int64_t o = (int64_t)(i >= 0 ? i : 0) * N + e;
                v[u] = values[o]; mk[u] = masks[o]; rw[u] = rewards[o];
                vl[u] = valid ? valid[o] : T(1);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 - u;
                if (i >= 0) {
                    const int64_t o = (int64_t)i * N + e;
                    const T delta = rw[u] + gmma * v_next * mk[u] - v[u];
                    gae = delta + gmma * lmbda * mk[u] * gae;
                    const T ret = gae + v[u];
                    returns[o] = ret;
                    const T ad = vl[u] != T(0) ? ret - v[u] : T(0);
                    adv[o] = ad;
                    if (vl[u] != T(0)) {
                        s1 += (double)ad;
                        s2 += (double)ad * (double)ad;
                        cnt += 1.0;
                    }
                    v_next = v[u];
                }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
            if (valid) cnt += __shfl_xor(cnt, off);
        }
        if ((threadIdx.x & 63) == 0) {
            if (partials) {  // deterministic mode: this wave's moments to their own slot, summed in a fixed order afterwards
                const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
                partials[2 * wv] = s1; partials[2 * wv + 1] = s2;
            } else {
                atomicAdd(&stats[1], s1);
                atomicAdd(&stats[2], s2);
            }
            if (valid && cnt != 0.0) atomicAdd(&stats[0], cnt);  // a sum of integers: exact in any order
        }
        if (!valid && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[0], (double)N * (double)n_steps);
    }
}

// out[j] += sum over i of partials[i][j], j < width, always in the same order: thread t of the single workgroup adds
// slots t, t + 1024, ... sequentially, then a fixed LDS tree folds the 1024 partial sums
template <typename A>
__global__ __launch_bounds__(1024) void k_sum_slots(const A* partials, int64_t n_slots, int width, A* out) {
    __shared__ A sh[1024];
    for (int j = 0; j < width; ++j) {
        A acc = A(0);
        for (int64_t i = threadIdx.x; i < n_slots; i += 1024) acc += partials[i * width + j];
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int d = 512; d > 0; d >>= 1) {
            if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[j] += sh[0];
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_adv_normalize(int64_t n, T* adv, const double* stats, int ddof, const T* valid = nullptr) {
    const double cnt = stats[0], mean = stats[1] / cnt;
    double var = (stats[2] - cnt * mean * mean) / (cnt - ddof);
    if (var < 0) var = 0;
    const double inv = 1.0 / (sqrt(var) + 1e-10);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        adv[i] = (valid && valid[i] == T(0)) ? T(0) : (T)(((double)adv[i] - mean) * inv);  // warm-up rows stay at 0
}

// ---- PPO clipped-surrogate loss (ppo.py:185-203) ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_ppo_loss(int64_t B, const T* mean, const T* value, const T* actions,
                                                  const T* old_logprob, const T* adv, const T* returns, T inv_var, T log_norm,
                                                  T eps, T c_value, T entropy4, double* sums, T* grad_mean, T* grad_value) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s_actor = 0.0, s_sq = 0.0;
    if (b < B) {
        T lp = T(0), lp_old = T(0), d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = actions[(int64_t)k * B + b] - mean[(int64_t)k * B + b];
            lp += T(-0.5) * d[k] * d[k] * inv_var + log_norm;
            lp_old += old_logprob[(int64_t)k * B + b];
        }
        const T ratio = (T)exp((double)(lp - lp_old));
        const T A = adv[b];
        const T clipped = Mth<T>::min(Mth<T>::max(ratio, T(1) - eps), T(1) + eps);
        const T s1 = ratio * A, s2 = clipped * A;
        const bool use1 = s1 <= s2;  // torch.min picks surr1 on ties
        s_actor = -(double)(use1 ? s1 : s2);
        const T err = value[b] - returns[b];
        s_sq = (double)err * (double)err;
        const T invB = T(1) / (T)B;
        if (grad_value) grad_value[b] = T(2) * c_value * err * invB;
        if (grad_mean) {
            // d(-min)/d mean_k: -A * ratio * (a_k - mean_k)/var when the un-clipped branch is active
            // (or the clamp is not saturated), else 0
            const bool live = use1 || (ratio > T(1) - eps && ratio < T(1) + eps);
#pragma unroll
            for (int k = 0; k < 4; ++k) grad_mean[(int64_t)k * B + b] = live ? -A * ratio * d[k] * inv_var * invB : T(0);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_actor += __shfl_xor(s_actor, off);
        s_sq += __shfl_xor(s_sq, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[1], s_actor);
        atomicAdd(&sums[2], s_sq);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&sums[0], (double)B);
        atomicAdd(&sums[3], (double)entropy4 * (double)B);
    }
}

// ---- visual_reward -----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_visual_reward(int64_t N, int total_steps, const T* marker, const T* pos,
                                                       const T* vel, const T* control, const T* last_shaping,
                                                       const int32_t* step, const T* ang, const T* v_ang, T* out_reward,
                                                       T* out_shaping, uint8_t* out_done, int32_t* out_solved) {
    using m = Mth<T>;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= N) return;
    const T mx = marker[e], my = marker[N + e], mz = marker[2 * N + e] - T(5);
    const T px = pos[e], py = pos[N + e], pz = pos[2 * N + e];
    const T exy = m::sqrt((mx - px) * (mx - px) + (my - py) * (my - py));
    const T ez = m::abs(mz - pz);
    const T vx = vel[e], vy = vel[N + e], vz = vel[2 * N + e];
    const T nvel = m::sqrt(vx * vx + vy * vy + vz * vz);
    const T c0 = control[e], c1 = control[N + e], c2 = control[2 * N + e];
    const T ce = m::sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    const T shaping = T(-4) * (exy * T(2) + ez);
    const T a0 = ang[e], a1 = ang[N + e];
    const T w0 = v_ang[e], w1 = v_ang[N + e], w2 = v_ang[2 * N + e];
    const bool soft = nvel < T(0.5196152422706632);   // ||ones(3)*0.30||
    const bool landed = pz <= T(-4.95);
    const bool on_target = exy < T(0.14);
    const bool flat = (m::sqrt(a0 * a0 + a1 * a1) < T(0.4937019546242588)) &&  // ||ones(2)*0.3491||
                      (m::sqrt(w0 * w0 + w1 * w1 + w2 * w2) < T(1.4142135623730951));
    const bool astray = (exy > ez / T(7) * T(5) + T(0.2)) || (ez > T(7));
    const bool max_steps = step[e] > total_steps;
    const T last = last_shaping[e];
    T reward = (last != T(0)) ? shaping - last - T(0.1) * ce : -T(0.1) * ce;
    bool done = false;
    int solved = 0;
    if (landed) {
        if (soft) {
            if (flat) { if (on_target) { reward = T(5); solved = 1; } else reward = T(1); }
            else reward = T(0);
        } else reward = T(-1);
        done = true;
    } else if (astray) { reward = T(-5); done = true; }
    if (max_steps) { reward = T(-2); done = true; }
    out_reward[e] = reward;
    out_shaping[e] = shaping;
    out_done[e] = done ? 1 : 0;
    out_solved[e] = solved;
}

}  // namespace qenv

using namespace qenv;

// sensor.__init__'s defaults (:588-592) times sensor.error, or the caller's eight values {accel_std, gyro_std, magnet_std,
// gps_std_p, gps_std_v, accel_bias_drift, gyro_bias_drift, magnet_bias_drift}
template <typename T> static SensParams<T> sens_params(const double* params, int error) {
    static const double dflt[8] = {0.1, 0.035, 15, 1.71, 0.5, 0.0005, 0.00015, 0.075};
    const double* v = params ? params : dflt;
    SensParams<T> p;
    for (int k = 0; k < 5; ++k) p.std[k] = (T)(v[k] * (error ? 1.0 : 0.0));
    for (int k = 0; k < 3; ++k) p.bd[k] = (T)(v[5 + k] * (error ? 1.0 : 0.0));
    return p;
}

extern "C" {

int qenv_window_push(qenv_t q, const void* obs, const void* actions, int reset_first) {
    if (!q || !obs || !actions) return fail(QENV_ERR_ARG, "qenv_window_push: null argument");
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    if (reset_first) {
        hipError_t e = hipMemsetAsync(q->window, 0, sizeof(float) * 15 * (size_t)q->cfg.T * (size_t)q->N, q->stream);
        if (e != hipSuccess) return fail_hip(e, "hipMemsetAsync(window)");
        q->win_head = 0;
    }
    const dim3 g((unsigned)((q->N + 255) / 256)), b(256);
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_window_push<double>), g, b, 0, q->stream, q->window, q->N, q->win_head, (const double*)obs,
                           (const double*)actions);
    else
        hipLaunchKernelGGL((k_window_push<float>), g, b, 0, q->stream, q->window, q->N, q->win_head, (const float*)obs,
                           (const float*)actions);
    q->win_head = (q->win_head + 1) % q->cfg.T;
    return check_launch("k_window_push");
}

size_t qenv_mlp_image_bytes(int k_in, int hidden) {
    return (k_in == 15 * mlp3::TW && hidden == mlp3::H) ? sizeof(uint4) * (size_t)mlp3::IMAGE_UINT4 : 0;
}

int qenv_mlp_prepare(qenv_t q, int head, int k_in, int hidden, int n_out, const float* W1, const float* b1, const float* W2,
                     const float* b2, const float* W3, const float* b3, void* image) {
    if (!q || head < 0 || head > 1) return fail(QENV_ERR_ARG, "qenv_mlp_prepare: null handle or head not 0 (actor) / 1 (critic)");
    if (!image) { q->prep[head] = {nullptr, nullptr, 0}; return QENV_OK; }   // unregister
    if (!W1 || !b1 || !W2 || !b2 || !W3 || !b3) return fail(QENV_ERR_ARG, "qenv_mlp_prepare: null weight pointer");
    if (qenv_mlp_image_bytes(k_in, hidden) == 0 || k_in != 15 * q->cfg.T)
        return fail(QENV_ERR_ARG, "qenv_mlp_prepare: only the 75-128-128-n policy on a T = 5 handle has a prepared form (see qenv_mlp_image_bytes)");
    if (n_out < 1 || n_out > 4) return fail(QENV_ERR_ARG, "qenv_mlp_prepare: n_out must be 1..4");
    if (((uintptr_t)image & 15) != 0) return fail(QENV_ERR_ARG, "qenv_mlp_prepare: the image must be 16-byte aligned");
    hipLaunchKernelGGL(k_mlp_prepare, dim3(16), dim3(256), 0, q->stream, (uint4*)image, W1, b1, W2, b2, W3, b3, n_out);
    const int rc = check_launch("k_mlp_prepare");
    if (rc == QENV_OK) q->prep[head] = {W1, image, n_out};
    return rc;
}

}  // extern "C"

// the prepared image of these weights, if the caller registered one for them on this handle
static const uint4* prepared_image(const qenv_s* q, const float* W1, int hidden, int n_out) {
    if (hidden != mlp3::H || q->cfg.T != mlp3::TW) return nullptr;
    for (int h = 0; h < 2; ++h)
        if (q->prep[h].image && q->prep[h].W1 == W1 && q->prep[h].n_out == n_out) return (const uint4*)q->prep[h].image;
    return nullptr;
}

extern "C" {

int qenv_mlp_forward(qenv_t q, int k_in, int hidden, int n_out, int final_tanh, const float* W1, const float* b1, const float* W2,
                     const float* b2, const float* W3, const float* b3, void* out, float* obs_store) {
    if (!q || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out) return fail(QENV_ERR_ARG, "qenv_mlp_forward: null argument");
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    if (n_out < 1 || n_out > 4) return fail(QENV_ERR_ARG, "qenv_mlp_forward: n_out must be 1..4");
    if (k_in != 15 * q->cfg.T) return fail(QENV_ERR_ARG, "qenv_mlp_forward: k_in (row length of W1) must be 15*T of the handle");
    MlpArgs a{};
    a.win = q->window; a.N = q->N; a.T = q->cfg.T; a.head = q->win_head; a.n_out = n_out; a.final_tanh = final_tanh;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.out = out;
    a.out_f64 = q->cfg.dtype == QENV_F64;
    a.obs_store = obs_store;
    a.n_chunks = 1;
    a.image = prepared_image(q, W1, hidden, n_out);
    const bool t5 = q->cfg.T == 5;  // the reference's window (ppo.py:60, T = 5): fully unrolled variant
    switch (hidden) {
        case 16: return t5 ? launch_mlp<16, 5, 512, true>(a, 5, q->stream) : launch_mlp<16, 0, 512, true>(a, q->cfg.T, q->stream);
        case 32: return t5 ? launch_mlp<32, 5, 512, true>(a, 5, q->stream) : launch_mlp<32, 0, 512, true>(a, q->cfg.T, q->stream);
        case 64: return t5 ? launch_mlp<64, 5, 512, true>(a, 5, q->stream) : launch_mlp<64, 0, 512, true>(a, q->cfg.T, q->stream);
        case 128: return t5 ? launch_mlp<128, 5, 512, true>(a, 5, q->stream) : launch_mlp<128, 0, 512, true>(a, q->cfg.T, q->stream);
        case 256: return launch_mlp_256<false>(a, q->cfg.T, q->stream);
        default: return fail(QENV_ERR_ARG, "qenv_mlp_forward: hidden must be one of 16, 32, 64, 128, 256");
    }
}

int qenv_policy_act(qenv_t q, int k_in, int hidden, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                    const float* b3, double std, const void* noise, uint64_t seed, uint64_t step, int64_t env_offset,
                    void* out_mean, void* out_action, void* out_logprob, float* obs_store) {
    if (q && q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    return qenv::policy_act_subset(q, k_in, hidden, W1, b1, W2, b2, W3, b3, std, noise, seed, step, env_offset, out_mean, out_action,
                                   out_logprob, obs_store, nullptr, nullptr, nullptr);
}

}  // extern "C"

// qenv_policy_act on a subset of the envs (see qenv_host.hpp): the overlapped PPO step of qenv_env.hip
int qenv::policy_act_subset(qenv_s* q, int k_in, int hidden, const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* W3, const float* b3, double std, const void* noise, uint64_t seed, uint64_t step,
                            int64_t env_offset, void* out_mean, void* out_action, void* out_logprob, float* obs_store,
                            const uint8_t* env_skip, const int32_t* list_idx, const int32_t* list_count, int obs_rows_from) {
    if (!q || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out_action) return fail(QENV_ERR_ARG, "qenv_policy_act: null argument");
    if (obs_rows_from < 0 || obs_rows_from >= 15 * q->cfg.T) return fail(QENV_ERR_ARG, "qenv_ppo_step: obs_rows_from must be in [0, 15*T)");
    if (!(std > 0)) return fail(QENV_ERR_ARG, "qenv_policy_act: std must be positive");
    if (k_in != 15 * q->cfg.T) return fail(QENV_ERR_ARG, "qenv_policy_act: k_in (row length of W1) must be 15*T of the handle");
    MlpArgs a{};
    a.win = q->window; a.N = q->N; a.T = q->cfg.T; a.head = q->win_head; a.n_out = 4; a.final_tanh = 1;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.out = out_mean;
    a.out_f64 = q->cfg.dtype == QENV_F64;
    a.obs_store = obs_store; a.n_chunks = 1; a.obs_rows_from = obs_rows_from;
    a.noise = noise; a.seed = seed; a.step = step; a.env_offset = env_offset;
    a.std = std; a.log_norm = -std::log(std) - 0.9189385332046727;  // -log(std) - 0.5 log(2 pi)
    a.out_action = out_action; a.out_logprob = out_logprob;
    a.env_skip = env_skip; a.list_idx = list_idx; a.list_count = list_count;
    a.zero_word = q->zero_on_act; q->zero_on_act = nullptr;  // consumed by this launch (qenv_ppo_step)
    a.image = prepared_image(q, W1, hidden, 4);
    const bool t5 = q->cfg.T == 5;
    switch (hidden) {
        case 16: return t5 ? launch_mlp<16, 5, 512, true, true>(a, 5, q->stream) : launch_mlp<16, 0, 512, true, true>(a, q->cfg.T, q->stream);
        case 32: return t5 ? launch_mlp<32, 5, 512, true, true>(a, 5, q->stream) : launch_mlp<32, 0, 512, true, true>(a, q->cfg.T, q->stream);
        case 64: return t5 ? launch_mlp<64, 5, 512, true, true>(a, 5, q->stream) : launch_mlp<64, 0, 512, true, true>(a, q->cfg.T, q->stream);
        case 128: return t5 ? launch_mlp<128, 5, 512, true, true>(a, 5, q->stream) : launch_mlp<128, 0, 512, true, true>(a, q->cfg.T, q->stream);
        case 256: return launch_mlp_256<true>(a, q->cfg.T, q->stream);
        default: return fail(QENV_ERR_ARG, "qenv_policy_act: hidden must be one of 16, 32, 64, 128, 256");
    }
}

extern "C" {

int qenv_dense_tail_chunks(int64_t n, int32_t n_chunks, int k_in, int hidden, int n_out, int final_tanh, const float* x,
                           const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                           const float* b3, float* out, void* stream) {
    return qenv_dense_tail_chunks_strided(n, n_chunks, 0, k_in, hidden, n_out, final_tanh, x, W1, b1, W2, b2, W3, b3, out, stream);
}

int qenv_dense_tail_chunks_strided(int64_t n, int32_t n_chunks, int64_t chunk_stride, int k_in, int hidden, int n_out, int final_tanh,
                                   const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                   const float* b3, float* out, void* stream) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out) return fail(QENV_ERR_ARG, "qenv_dense_tail: null argument");
    if (n <= 0 || n_chunks < 1 || k_in <= 0 || n_out < 1 || n_out > 4) return fail(QENV_ERR_ARG, "qenv_dense_tail: bad sizes");
    if (chunk_stride < 0) return fail(QENV_ERR_ARG, "qenv_dense_tail_chunks_strided: negative chunk stride");
    if (chunk_stride != 0 && chunk_stride != (int64_t)k_in * n && k_in != 75)
        return fail(QENV_ERR_ARG, "qenv_dense_tail_chunks_strided: a chunk stride other than K*n needs the 75-float observation (K = 75)");
    if ((int64_t)k_in * n >= ((int64_t)1 << 31)) return fail(QENV_ERR_ARG, "qenv_dense_tail: K*n must stay below 2^31");
    if (k_in == 75) {  // the controller's 75-float observation (T = 5): W1 fits the LDS whole -> the k_mlp kernel,
        MlpArgs m{};   // reading each chunk as a window ring with head 0 (row r of a chunk = window element r)
        m.win = x; m.N = n; m.T = 5; m.head = 0; m.n_out = n_out; m.final_tanh = final_tanh;
        m.W1 = W1; m.b1 = b1; m.W2 = W2; m.b2 = b2; m.W3 = W3; m.b3 = b3; m.out = out; m.out_f64 = 0;
        m.obs_store = nullptr; m.n_chunks = n_chunks; m.win_chunk_stride = chunk_stride;
        switch (hidden) {
            case 16: return launch_mlp<16, 5, 512, true>(m, 5, (hipStream_t)stream);
            case 32: return launch_mlp<32, 5, 512, true>(m, 5, (hipStream_t)stream);
            case 64: return launch_mlp<64, 5, 512, true>(m, 5, (hipStream_t)stream);
            case 128: return launch_mlp<128, 5, 512, true>(m, 5, (hipStream_t)stream);
            default: return fail(QENV_ERR_ARG, "qenv_dense_tail: hidden must be 16, 32, 64 or 128");
        }
    }
    DenseArgs a;
    a.x = x; a.N = n; a.K = k_in; a.n_out = n_out; a.final_tanh = final_tanh; a.n_chunks = n_chunks; a.rows = 0;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.out = out;
    switch (hidden) {
        case 16: return launch_dense<16>(a, (hipStream_t)stream);
        case 32: return launch_dense<32>(a, (hipStream_t)stream);
        case 64: return launch_dense<64>(a, (hipStream_t)stream);
        case 128: return launch_dense<128>(a, (hipStream_t)stream);
        default: return fail(QENV_ERR_ARG, "qenv_dense_tail: hidden must be 16, 32, 64 or 128");
    }
}

int qenv_dense_tail_rows(int64_t n, int k_in, int hidden, int n_out, int final_tanh, const float* x_rows, const float* W1,
                         const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, float* out,
                         void* stream) {
    if (!x_rows || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out) return fail(QENV_ERR_ARG, "qenv_dense_tail_rows: null argument");
    if (n <= 0 || k_in <= 0 || n_out < 1 || n_out > 4) return fail(QENV_ERR_ARG, "qenv_dense_tail_rows: bad sizes");
    if ((int64_t)k_in * n >= ((int64_t)1 << 29)) return fail(QENV_ERR_ARG, "qenv_dense_tail_rows: K*n must stay below 2^29 (32-bit byte offsets)");
    if (((uintptr_t)x_rows & 15) != 0) return fail(QENV_ERR_ARG, "qenv_dense_tail_rows: x_rows must be 16-byte aligned");
    if (hidden != 64) return fail(QENV_ERR_ARG, "qenv_dense_tail_rows: hidden must be 64 (rl_nn_model.py's tail); other shapes: qenv_dense_tail");
    DenseArgs a;
    a.x = x_rows; a.N = n; a.K = k_in; a.n_out = n_out; a.final_tanh = final_tanh; a.n_chunks = 1; a.rows = 1;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3; a.out = out;
    return launch_dense<64>(a, (hipStream_t)stream);
}

#ifdef QENV_DT3_STAMPS   /* diagnostic builds only (not declared in qenv.h) */
int qenv_debug_dt3_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(qenv::dt3_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

int qenv_dense_tail(int64_t n, int k_in, int hidden, int n_out, int final_tanh, const float* x, const float* W1,
                    const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, float* out,
                    void* stream) {
    return qenv_dense_tail_chunks(n, 1, k_in, hidden, n_out, final_tanh, x, W1, b1, W2, b2, W3, b3, out, stream);
}

int qenv_sample_action(qenv_t q, const void* mean, double std, const void* noise, uint64_t seed, uint64_t step,
                       int64_t env_offset, void* out_action, void* out_logprob) {
    if (!q || !mean || !out_action) return fail(QENV_ERR_ARG, "qenv_sample_action: null argument");
    if (!(std > 0)) return fail(QENV_ERR_ARG, "qenv_sample_action: std must be positive");
    const dim3 g((unsigned)((q->N + 255) / 256)), b(256);
    const double log_norm = -std::log(std) - 0.9189385332046727;  // -log(std) - 0.5 log(2 pi)
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_sample_action<double>), g, b, 0, q->stream, q->N, (const double*)mean, std, log_norm,
                           (const double*)noise, seed, step, env_offset, (double*)out_action, (double*)out_logprob);
    else
        hipLaunchKernelGGL((k_sample_action<float>), g, b, 0, q->stream, q->N, (const float*)mean, (float)std,
                           (float)log_norm, (const float*)noise, seed, step, env_offset, (float*)out_action,
                           (float*)out_logprob);
    return check_launch("k_sample_action");
}

int qenv_sample_init_states(qenv_t q, uint64_t seed, uint64_t episode, int64_t env_offset, const uint8_t* mask,
                            void* out_states) {
    if (!q || !out_states) return fail(QENV_ERR_ARG, "qenv_sample_init_states: null argument");
    const dim3 g((unsigned)((q->N + 255) / 256)), b(256);
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_sample_init<double>), g, b, 0, q->stream, q->N, seed, episode, env_offset, mask, (double*)out_states);
    else
        hipLaunchKernelGGL((k_sample_init<float>), g, b, 0, q->stream, q->N, seed, episode, env_offset, mask, (float*)out_states);
    return check_launch("k_sample_init");
}

int qenv_sensor_reset(qenv_t q, void* sens, const void* u3, uint64_t seed, uint64_t episode, int64_t env_offset, int error,
                      const double* params) {
    if (!q || !sens) return fail(QENV_ERR_ARG, "qenv_sensor_reset: null argument");
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    const dim3 g((unsigned)((q->N + 255) / 256)), b(256);
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_sensor_reset<double>), g, b, 0, q->stream, q->N, (const double*)q->ws, q->wsi, (double*)sens, (const double*)u3,
                           seed, episode, env_offset, sens_params<double>(params, error));
    else
        hipLaunchKernelGGL((k_sensor_reset<float>), g, b, 0, q->stream, q->N, (const float*)q->ws, q->wsi, (float*)sens, (const float*)u3, seed,
                           episode, env_offset, sens_params<float>(params, error));
    return check_launch("k_sensor_reset");
}

int qenv_sensor_bias_reset(qenv_t q, void* sens, const void* u3, int error, const double* params) {
    if (!q || !sens || !u3) return fail(QENV_ERR_ARG, "qenv_sensor_bias_reset: null argument");
    const dim3 g((unsigned)((q->N + 255) / 256)), b(256);
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_sensor_bias_reset<double>), g, b, 0, q->stream, q->N, (double*)sens, (const double*)u3, sens_params<double>(params, error));
    else
        hipLaunchKernelGGL((k_sensor_bias_reset<float>), g, b, 0, q->stream, q->N, (float*)sens, (const float*)u3, sens_params<float>(params, error));
    return check_launch("k_sensor_bias_reset");
}

int qenv_sensor_read(qenv_t q, void* sens, const void* z27, uint64_t seed, uint64_t step, int64_t env_offset, int error,
                     const double* params, void* out) {
    if (!q || !sens || !out) return fail(QENV_ERR_ARG, "qenv_sensor_read: null argument");
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    const dim3 g((unsigned)((q->N + 255) / 256)), b(256);
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_sensor_read<double>), g, b, 0, q->stream, q->N, (double*)q->ws, q->wsi, (double*)sens, (const double*)z27, seed,
                           step, env_offset, sens_params<double>(params, error), q->cfg.t_step, (double*)out);
    else
        hipLaunchKernelGGL((k_sensor_read<float>), g, b, 0, q->stream, q->N, (float*)q->ws, q->wsi, (float*)sens, (const float*)z27, seed, step,
                           env_offset, sens_params<float>(params, error), (float)q->cfg.t_step, (float*)out);
    return check_launch("k_sensor_read");
}

int qenv_sensor_ops(qenv_t q, void* sens, const int32_t* ops, int n_ops, const void* z, int error, const double* params,
                    void* out) {
    if (!q || !sens || !ops || !z || !out) return fail(QENV_ERR_ARG, "qenv_sensor_ops: null argument");
    if (n_ops < 1 || n_ops > 8) return fail(QENV_ERR_ARG, "qenv_sensor_ops: 1..8 methods per call");
    SensProgram prog;
    prog.n_ops = n_ops;
    for (int k = 0; k < n_ops; ++k) {
        if (ops[k] < 0 || ops[k] >= SOP_COUNT) return fail(QENV_ERR_ARG, "qenv_sensor_ops: unknown method id");
        prog.ops[k] = ops[k];
    }
    if (q->ar_pending) { const int frc = ppo_flush(q); if (frc != QENV_OK) return frc; }
    const dim3 g((unsigned)((q->N + 255) / 256)), b(256);
    if (q->cfg.dtype == QENV_F64)
        hipLaunchKernelGGL((k_sensor_ops<double>), g, b, 0, q->stream, q->N, (double*)q->ws, q->wsi, (double*)sens, (const double*)z, prog,
                           sens_params<double>(params, error), q->cfg.t_step, (double*)out);
    else
        hipLaunchKernelGGL((k_sensor_ops<float>), g, b, 0, q->stream, q->N, (float*)q->ws, q->wsi, (float*)sens, (const float*)z, prog,
                           sens_params<float>(params, error), (float)q->cfg.t_step, (float*)out);
    return check_launch("k_sensor_ops");
}

static int gae_impl(int dtype, int64_t n_envs, int n_steps, const void* values, const void* masks, const void* rewards,
                    void* returns, void* adv, double* stats, double* partials, void* stream, const void* valid = nullptr) {
    if (!values || !masks || !rewards || !returns || !adv) return fail(QENV_ERR_ARG, "qenv_gae: null argument");
    if (n_envs <= 0 || n_steps <= 0) return fail(QENV_ERR_ARG, "qenv_gae: n_envs and n_steps must be positive");
    const dim3 g((unsigned)((n_envs + 255) / 256)), b(256);
    if (dtype == QENV_F64)
        hipLaunchKernelGGL((k_gae<double>), g, b, 0, (hipStream_t)stream, n_envs, n_steps, (const double*)values,
                           (const double*)masks, (const double*)rewards, (double*)returns, (double*)adv, stats, partials,
                           (const double*)valid);
    else if (dtype == QENV_F32)
        hipLaunchKernelGGL((k_gae<float>), g, b, 0, (hipStream_t)stream, n_envs, n_steps, (const float*)values,
                           (const float*)masks, (const float*)rewards, (float*)returns, (float*)adv, stats, partials,
                           (const float*)valid);
    else return fail(QENV_ERR_ARG, "qenv_gae: bad dtype");
    int rc = check_launch("k_gae");
    if (rc == 0 && partials && stats) {
        hipLaunchKernelGGL((k_sum_slots<double>), dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, (int64_t)g.x * 4, 2, stats + 1);
        rc = check_launch("k_sum_slots");
    }
    return rc;
}

int qenv_gae(int dtype, int64_t n_envs, int n_steps, const void* values, const void* masks, const void* rewards,
             void* returns, void* adv, double* stats, void* stream) {
    return gae_impl(dtype, n_envs, n_steps, values, masks, rewards, returns, adv, stats, nullptr, stream);
}

size_t qenv_gae_workspace_bytes(int64_t n_envs) { return n_envs > 0 ? (size_t)((n_envs + 255) / 256) * 4 * 2 * sizeof(double) : 0; }

int qenv_gae_deterministic(int dtype, int64_t n_envs, int n_steps, const void* values, const void* masks,
                           const void* rewards, void* returns, void* adv, double* stats, void* workspace, void* stream) {
    if (!stats || !workspace) return fail(QENV_ERR_ARG, "qenv_gae_deterministic: stats and workspace are required");
    return gae_impl(dtype, n_envs, n_steps, values, masks, rewards, returns, adv, stats, (double*)workspace, stream);
}

int qenv_gae_masked(int dtype, int64_t n_envs, int n_steps, const void* values, const void* masks, const void* rewards,
                    const void* valid, void* returns, void* adv, double* stats, void* workspace, void* stream) {
    if (!stats) return fail(QENV_ERR_ARG, "qenv_gae_masked: stats is required (the count of valid rows is returned there)");
    return gae_impl(dtype, n_envs, n_steps, values, masks, rewards, returns, adv, stats, (double*)workspace, stream, valid);
}

static int adv_normalize_impl(int dtype, int64_t n, void* adv, const void* valid, const double* stats, int ddof, void* stream);

int qenv_adv_normalize_masked(int dtype, int64_t n, void* adv, const void* valid, const double* stats, int ddof, void* stream) {
    return adv_normalize_impl(dtype, n, adv, valid, stats, ddof, stream);
}

int qenv_adv_normalize(int dtype, int64_t n, void* adv, const double* stats, int ddof, void* stream) {
    return adv_normalize_impl(dtype, n, adv, nullptr, stats, ddof, stream);
}

static int adv_normalize_impl(int dtype, int64_t n, void* adv, const void* valid, const double* stats, int ddof, void* stream) {
    if (!adv || !stats) return fail(QENV_ERR_ARG, "qenv_adv_normalize: null argument");
    if (n <= 0) return fail(QENV_ERR_ARG, "qenv_adv_normalize: n must be positive");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const dim3 g((unsigned)blocks), b(256);
    if (dtype == QENV_F64)
        hipLaunchKernelGGL((k_adv_normalize<double>), g, b, 0, (hipStream_t)stream, n, (double*)adv, stats, ddof, (const double*)valid);
    else if (dtype == QENV_F32)
        hipLaunchKernelGGL((k_adv_normalize<float>), g, b, 0, (hipStream_t)stream, n, (float*)adv, stats, ddof, (const float*)valid);
    else return fail(QENV_ERR_ARG, "qenv_adv_normalize: bad dtype");
    return check_launch("k_adv_normalize");
}

int qenv_ppo_loss(int dtype, int64_t batch, const void* mean, const void* value, const void* actions,
                  const void* old_logprob, const void* adv, const void* returns, double std, double eps_clip,
                  double c_value, double* sums, void* grad_mean, void* grad_value, void* stream) {
    if (!mean || !value || !actions || !old_logprob || !adv || !returns || !sums)
        return fail(QENV_ERR_ARG, "qenv_ppo_loss: null argument");
    if (batch <= 0 || !(std > 0)) return fail(QENV_ERR_ARG, "qenv_ppo_loss: batch and std must be positive");
    const dim3 g((unsigned)((batch + 255) / 256)), b(256);
    const double inv_var = 1.0 / (std * std), log_norm = -std::log(std) - 0.9189385332046727;
    const double entropy4 = 4.0 * (0.5 + 0.9189385332046727 + std::log(std));
    if (dtype == QENV_F64)
        hipLaunchKernelGGL((k_ppo_loss<double>), g, b, 0, (hipStream_t)stream, batch, (const double*)mean, (const double*)value,
                           (const double*)actions, (const double*)old_logprob, (const double*)adv, (const double*)returns,
                           inv_var, log_norm, eps_clip, c_value, entropy4, sums, (double*)grad_mean, (double*)grad_value);
    else if (dtype == QENV_F32)
        hipLaunchKernelGGL((k_ppo_loss<float>), g, b, 0, (hipStream_t)stream, batch, (const float*)mean, (const float*)value,
                           (const float*)actions, (const float*)old_logprob, (const float*)adv, (const float*)returns,
                           (float)inv_var, (float)log_norm, (float)eps_clip, (float)c_value, (float)entropy4, sums,
                           (float*)grad_mean, (float*)grad_value);
    else return fail(QENV_ERR_ARG, "qenv_ppo_loss: bad dtype");
    return check_launch("k_ppo_loss");
}

int qenv_visual_reward(int dtype, int64_t n_envs, int total_steps, const void* marker, const void* pos, const void* vel,
                       const void* control, const void* last_shaping, const int32_t* step, const void* ang,
                       const void* v_ang, void* out_reward, void* out_shaping, uint8_t* out_done, int32_t* out_solved,
                       void* stream) {
    if (!marker || !pos || !vel || !control || !last_shaping || !step || !ang || !v_ang || !out_reward || !out_shaping ||
        !out_done || !out_solved)
        return fail(QENV_ERR_ARG, "qenv_visual_reward: null argument");
    if (n_envs <= 0) return fail(QENV_ERR_ARG, "qenv_visual_reward: n_envs must be positive");
    const dim3 g((unsigned)((n_envs + 255) / 256)), b(256);
    if (dtype == QENV_F64)
        hipLaunchKernelGGL((k_visual_reward<double>), g, b, 0, (hipStream_t)stream, n_envs, total_steps,
                           (const double*)marker, (const double*)pos, (const double*)vel, (const double*)control,
                           (const double*)last_shaping, step, (const double*)ang, (const double*)v_ang, (double*)out_reward,
                           (double*)out_shaping, out_done, out_solved);
    else if (dtype == QENV_F32)
        hipLaunchKernelGGL((k_visual_reward<float>), g, b, 0, (hipStream_t)stream, n_envs, total_steps,
                           (const float*)marker, (const float*)pos, (const float*)vel, (const float*)control,
                           (const float*)last_shaping, step, (const float*)ang, (const float*)v_ang, (float*)out_reward,
                           (float*)out_shaping, out_done, out_solved);
    else return fail(QENV_ERR_ARG, "qenv_visual_reward: bad dtype");
    return check_launch("k_visual_reward");
}

}  // extern "C"

#ifdef QENV_MLP_STAMPS
extern "C" int qenv_debug_mlp_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(qenv::g_mlp_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif
