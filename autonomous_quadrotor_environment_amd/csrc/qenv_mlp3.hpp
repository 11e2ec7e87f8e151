// qenv_mlp3.hpp -- the N = 128, T = 5 policy forward on the bf16 matrix cores as device functions: the argument block, the weight
// image, one 32-env tile of layers 1-3 and ActorCritic.act's sampling.  Shared by k_mlp_v3 (qenv_ppo.hip: one launch per
// policy step) and k_ppo_fused (qenv_env_impl.hpp: policy and env step of a whole rollout in one launch) -- one definition, so
// the two produce the same bits.
#pragma once
#include "qenv_mlp_common.hpp"
#include <cstdint>

namespace qenv {

struct MlpArgs {
    const float* win;   // [15*T][N] ring
    int64_t N;
    int32_t T, head, n_out, final_tanh;
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    void* out;          // [n_out][N]
    int32_t out_f64;
    float* obs_store;   // null or [15*T][N]: the window in time order
    int32_t n_chunks;   // > 1: win = [n_chunks][15*T][N] stored observations (head 0), out = [n_chunks][n_out][N]
    // SAMPLE instantiations (qenv_policy_act): ActorCritic.act's sampling in the epilogue of the actor forward
    const void* noise;  // null (Philox) or injected [4][N] standard normals in the handle's dtype
    uint64_t seed, step;
    int64_t env_offset;
    double std, log_norm;
    void *out_action, *out_logprob;  // [4][N] in the handle's dtype; logprob nullable
    // overlapped PPO step: nothing is stored for an env whose byte in env_skip is set (its tile is computed anyway) ...
    const uint8_t* env_skip;
    // ... or: only the envs list_idx[0 .. *list_count) are evaluated, gathered into dense tiles (k_mlp_small only)
    const int32_t* list_idx;
    const int32_t* list_count;
    // cleared by the first thread of the launch: the length counter of the reset list that the step kernel FOLLOWING this
    // policy forward on the stream appends to (qenv_ppo_step: saves a 4-byte memset launch, ~5 us, per rollout step)
    int32_t* zero_word;
    // k_mlp_v3: null, or the weights already in the kernel's LDS layout (qenv_mlp_prepare): staging is then a plain copy
    const uint4* image;
    // The SLICE layout of stored observations (k_mlp_v3 / k_mlp_small; launch_mlp refuses it elsewhere).  The window of step t is
    // the five newest 15-float slices, so consecutive windows overlap in four of them: a rollout stores ONE slice per step into
    // an array [(n_steps + T - 1) * 15][N] and the window of step t is its rows 15 t .. 15 t + 15 T - 1 -- the [n][15 T][N]
    // layout with a chunk stride of 15 N instead of 15 T N.
    int32_t obs_rows_from;      // obs_store: only rows >= this are written (15 (T - 1): the newest slice; 0: the whole window)
    int64_t win_chunk_stride;   // n_chunks > 1: elements between consecutive chunks of `win` (0 = 15 T N, dense)
};

// ---- counter-based RNG: Philox4x32-10 (Salmon et al. 2011) --------------------------------------
// One call = 4 x 32 random bits as a pure function of (key, counter): any env/step can be drawn
// independently, so streams are identical whatever the sharding or launch order.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float r = sqrtf(-2.0f * logf(u01(a)));
    float sn, cs;
    sincosf(6.283185307179586f * u01(b), &sn, &cs);
    z0 = r * cs; z1 = r * sn;
}

// k_sample_action for the env of this lane, from the four means the lane holds (same streams, same arithmetic)
__device__ __forceinline__ void sample_store(const MlpArgs& a, int64_t e, const float mean[4]) {
    float z[4];
    if (a.noise) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            z[k] = a.out_f64 ? (float)((const double*)a.noise)[(int64_t)k * a.N + e] : ((const float*)a.noise)[(int64_t)k * a.N + e];
    } else {
        const uint64_t g = (uint64_t)(a.env_offset + e);
        uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)a.step, (uint32_t)(a.step >> 32)};
        philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32) ^ 0xAC710Du);
        box_muller(c[0], c[1], z[0], z[1]);
        box_muller(c[2], c[3], z[2], z[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (a.out_f64) {
            const double eps = (double)z[k];
            ((double*)a.out_action)[(int64_t)k * a.N + e] = (double)mean[k] + a.std * eps;
            if (a.out_logprob) ((double*)a.out_logprob)[(int64_t)k * a.N + e] = -0.5 * eps * eps + a.log_norm;
        } else {
            const float eps = z[k];
            ((float*)a.out_action)[(int64_t)k * a.N + e] = mean[k] + (float)a.std * eps;
            if (a.out_logprob) ((float*)a.out_logprob)[(int64_t)k * a.N + e] = -0.5f * eps * eps + (float)a.log_norm;
        }
    }
}

// sample_store's arithmetic with the results handed back instead of stored (k_ppo_fused: the action goes straight into the env step):
// act / logp in the handle's storage type S (double iff a.out_f64)
template <typename S>
__device__ __forceinline__ void sample_compute(const MlpArgs& a, int64_t e, const float mean[4], S act[4], S logp[4]) {
    float z[4];
    if (a.noise) {
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (float)((const S*)a.noise)[(int64_t)k * a.N + e];
    } else {
        const uint64_t g = (uint64_t)(a.env_offset + e);
        uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)a.step, (uint32_t)(a.step >> 32)};
        philox4x32_10(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32) ^ 0xAC710Du);
        box_muller(c[0], c[1], z[0], z[1]);
        box_muller(c[2], c[3], z[2], z[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if constexpr (sizeof(S) == 8) {
            const double eps = (double)z[k];
            act[k] = (double)mean[k] + a.std * eps;
            logp[k] = -0.5 * eps * eps + a.log_norm;
        } else {
            const float eps = z[k];
            act[k] = mean[k] + (float)a.std * eps;
            logp[k] = -0.5f * eps * eps + (float)a.log_norm;
        }
    }
}

// the image from the fp32 torch-layout weights, written by `nthr` threads (a workgroup into LDS, or a small grid into
// global memory: qenv_mlp_prepare).  Fragment layouts: see k_mlp.
__device__ __forceinline__ void mlp3_build_image(uint4* img, const float* W1, const float* b1, const float* W2, const float* b2,
                                                 const float* W3, const float* b3, int n_out, int tid, int nthr) {
    using namespace mlp3;
    for (int c = tid; c < NT * KS16 * 64; c += nthr) {   // W1: lane (col, half) holds W1[32 t + col][16 s + 8 half + i]
        const int ln = c & 63, sk = (c >> 6) % KS16, t = (c >> 6) / KS16;
        const int k0 = 16 * sk + 8 * (ln >> 5);
        const float* src = W1 + (size_t)(32 * t + (ln & 31)) * KW;
        float w8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = (k0 + i) < KW ? src[k0 + i] * TANH_PRESCALE : 0.f;   // (see fast_tanh_prescaled)
        bf16x8 sp[3];
        split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) img[W1F + ((t * KS16 + sk) * 3 + k) * 64 + ln] = __builtin_bit_cast(uint4, sp[k]);
    }
    for (int c = tid; c < NT * NT * 2 * 64; c += nthr) { // W2: lane (col, half) holds W2[32 t + col][32 ti + acc_row(8 jh + i, half)]
        const int ln = c & 63, jh = (c >> 6) & 1, ti = (c >> 7) % NT, t = (c >> 7) / NT;
        const float* src = W2 + (size_t)(32 * t + (ln & 31)) * H + 32 * ti + 16 * jh + 4 * (ln >> 5);
        const float4 lo4 = *reinterpret_cast<const float4*>(src), hi4 = *reinterpret_cast<const float4*>(src + 8);
        const float w8[8] = {lo4.x * TANH_PRESCALE, lo4.y * TANH_PRESCALE, lo4.z * TANH_PRESCALE, lo4.w * TANH_PRESCALE,
                             hi4.x * TANH_PRESCALE, hi4.y * TANH_PRESCALE, hi4.z * TANH_PRESCALE, hi4.w * TANH_PRESCALE};
        bf16x8 sp[3];
        split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) img[W2F + (((t * NT + ti) * 2 + jh) * 3 + k) * 64 + ln] = __builtin_bit_cast(uint4, sp[k]);
    }
    for (int h = tid; h < H; h += nthr) {
        float4 w;
        w.x = W3[h]; w.y = n_out > 1 ? W3[(size_t)H + h] : 0.f; w.z = n_out > 2 ? W3[(size_t)2 * H + h] : 0.f;
        w.w = n_out > 3 ? W3[(size_t)3 * H + h] : 0.f;
        img[W3T + h] = __builtin_bit_cast(uint4, w);
    }
    for (int q4 = tid; q4 < H / 4; q4 += nthr) {
        const float4 x1 = *reinterpret_cast<const float4*>(b1 + 4 * q4), x2 = *reinterpret_cast<const float4*>(b2 + 4 * q4);
        const float4 s1 = {x1.x * TANH_PRESCALE, x1.y * TANH_PRESCALE, x1.z * TANH_PRESCALE, x1.w * TANH_PRESCALE};
        const float4 s2 = {x2.x * TANH_PRESCALE, x2.y * TANH_PRESCALE, x2.z * TANH_PRESCALE, x2.w * TANH_PRESCALE};
        img[B1 + q4] = __builtin_bit_cast(uint4, s1);
        img[B2 + q4] = __builtin_bit_cast(uint4, s2);
    }
    if (tid == 0) {
        float4 b;
        b.x = b3[0]; b.y = n_out > 1 ? b3[1] : 0.f; b.z = n_out > 2 ? b3[2] : 0.f; b.w = n_out > 3 ? b3[3] : 0.f;
        img[B3] = __builtin_bit_cast(uint4, b);
    }
}


// ---- layers 1-3 of ONE 32-env tile for a wave that has its SIMD to itself (k_ppo_fused: one wave per SIMD) ---------------
// The arithmetic is k_mlp_v3's, instruction for instruction: per output tile the same chain of v_mfma_f32_32x32x16_bf16 in the
// same order (six products of the three-way split operands per 16 k, smallest first), the same tanh, the same split, layer 3 as
// the same FMAs in the same order -- the two kernels are compared bit for bit by the tests.  What differs is the SCHEDULE.  A
// lone wave has nobody to hide behind: the matrix pipe runs only while this wave's stream feeds it, and an MFMA takes 32 cycles
// of which the issue port is busy for 8 -- the vector work of the PREVIOUS output tile (tanh + split into the next layer's B
// operand, or tanh + layer 3) has to sit BETWEEN the MFMAs of this one.  k_mlp_v3 asks for that with one scheduling region per
// chain (sched_group_barrier); with 512 registers to play with hipcc honours it only in small regions (in a region of 30-48
// MFMAs it issued the chain as a burst and the vector work as a block: 20.5 k cycles per tile, the sum of the two).  So every
// (MFMA, vector work) pair is its own region here, the work of a pair sized by what tools/ubench/mfma_fill_kinds.hip measured
// to fit one MFMA's shadow (see finish1 / finish2 below), and the LDS reads of the NEXT 16-k step's three weight fragments are
// issued a step ahead (a step's MFMAs never wait for their own reads).  20.5 k -> 14.7 k cycles per tile.
// lds4: the image (mlp3 layout); xw: the tile's window, element (sk, i) of a lane = window[16 sk + 8 half + i] of its env (fp32;
// split into bf16 terms here: k-step 0's up front, k-step sk + 1's in the shadows of chain 0's step sk); o: the four outputs
// before the final tanh, valid in the lower lane half.
#ifndef MLP3_STAMP
#define MLP3_STAMP(slot) do { } while (0)   // (diagnostic builds of the including unit define it: s_memtime stamps)
#endif
__device__ __forceinline__ void mlp3_tile_forward(const uint4* lds4, int lane, int half, const float (&xw)[mlp3::KS16][8],
                                                  float (&o)[4]) {
    using namespace mlp3;
    const uint4* W1f = lds4 + W1F;
    const uint4* W2f = lds4 + W2F;
    const float4* W3t = reinterpret_cast<const float4*>(lds4 + W3T);
    const float* b1s = reinterpret_cast<const float*>(lds4 + B1);
    const float* b2s = reinterpret_cast<const float*>(lds4 + B2);
    const float* b3s = reinterpret_cast<const float*>(lds4 + B3);
    struct Frag { bf16x8 h, m, l; };
    auto ld = [&](const uint4* f) {
        Frag r;
        r.h = __builtin_bit_cast(bf16x8, f[0]); r.m = __builtin_bit_cast(bf16x8, f[64]); r.l = __builtin_bit_cast(bf16x8, f[128]);
        return r;
    };
    auto f1 = [&](int t, int sk) { return W1f + (size_t)((t * KS16 + sk) * 3) * 64 + lane; };
    auto f2 = [&](int t, int u) { return W2f + (size_t)((t * NT * 2 + u) * 3) * 64 + lane; };   // u = 2 ti + jh
    auto bias = [&](f32x16& a, const float* b, int t) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = b[32 * t + acc_row(j, half)];
    };
    // one 16-k step: the six products, smallest first, with work(slot) behind each of them -- every (MFMA, work) pair is its own
    // scheduling region, so the vector work stands where it is written: in the 32-cycle shadow of the MFMA in front of it
    auto step6 = [&](f32x16& a, const Frag& f, const bf16x8& bh, const bf16x8& bm, const bf16x8& bl, auto&& work) {
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.m, bm, a, 0, 0, 0); work(0); __builtin_amdgcn_sched_barrier(0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, bh, a, 0, 0, 0); work(1); __builtin_amdgcn_sched_barrier(0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, bl, a, 0, 0, 0); work(2); __builtin_amdgcn_sched_barrier(0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.m, bh, a, 0, 0, 0); work(3); __builtin_amdgcn_sched_barrier(0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, bm, a, 0, 0, 0); work(4); __builtin_amdgcn_sched_barrier(0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, bh, a, 0, 0, 0); work(5); __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc1[NT];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned bspu[NT][2][3][4];   // layer 2's B operand: the tanh of layer 1 as bf16 terms (hi, mid, lo), [tile][k-step half][term][dword]
    auto bsp = [&](int t, int jh, int k) {
        const u32x4 u = {bspu[t][jh][k][0], bspu[t][jh][k][1], bspu[t][jh][k][2], bspu[t][jh][k][3]};
        return __builtin_bit_cast(bf16x8, u);
    };
    // (every piece of vector work ends in an opaque use of what it computed: its results are not needed before layer 2 / the
    // tile's end, and left alone the compiler SINKS the arithmetic out of the place it was written into, down to where the
    // results are used -- the whole of layer 1's vector work then stood in one block in front of layer 2)
    auto pin_f = [](float& x) { asm volatile("" : "+v"(x)); };
    auto pin_u = [](unsigned& x) { asm volatile("" : "+v"(x)); };
    auto cvt_pk = [](float lo, float hi) { unsigned d; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi)); return d; };
    unsigned xspu[KS16][3][4];   // the window as bf16 terms (hi, mid, lo), [k-step][term][dword]
    auto xsp = [&](int sk, int k) {
        const u32x4 u = {xspu[sk][k][0], xspu[sk][k][1], xspu[sk][k][2], xspu[sk][k][3]};
        return __builtin_bit_cast(bf16x8, u);
    };
    auto split_x_pair = [&](int sk, int d) {   // split3x8's arithmetic on elements 2 d, 2 d + 1 of k-step sk
        const float x0 = xw[sk][2 * d], x1 = xw[sk][2 * d + 1];
        const unsigned h = cvt_pk(x0, x1);
        const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cvt_pk(r0, r1);
        const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
        xspu[sk][0][d] = h; xspu[sk][1][d] = m; xspu[sk][2][d] = cvt_pk(s0, s1);
        pin_u(xspu[sk][0][d]); pin_u(xspu[sk][1][d]); pin_u(xspu[sk][2][d]);
    };
    float th1[NT][16];       // tanh of layer 1's accumulators
    float rs1[8][2];         // the split in flight: a pair's remainders after the hi term
    unsigned mp1[8];         // ... and its mid terms
    // The vector work behind layer-1 output tile t, by the slot S = 0, 1, 2 ... of the MFMA chain it rides on (tools/ubench/
    // mfma_fill_kinds.hip: the shadow of one MFMA holds six plain VALU instructions, or two transcendental ones and two plain;
    // v_pk_* and a third transcendental do not fit): slots 0-15 the tanh of register S (v_exp, v_add, v_rcp, v_fma); then the
    // split of the eight register PAIRS into bf16 terms, half a pair per slot -- split3x8's arithmetic per element (hi = bf16(x),
    // r = x - hi, mid = bf16(r), lo = bf16(r - mid)), spelled with v_cvt_pk_bf16_f32 on the pair so that a term's dword is
    // born packed: (hi, r0, r1, mid) = 6 instructions, then (lo) = 5.  n_slots = 30: the last two slots take two halves each.
    auto split_half = [&](int t, int unit) {
        const int pr = unit >> 1, jh = pr >> 2, d = pr & 3;
        if ((unit & 1) == 0) {
            const float x0 = th1[t][2 * pr], x1 = th1[t][2 * pr + 1];
            const unsigned h = cvt_pk(x0, x1);
            rs1[pr][0] = x0 - __uint_as_float(h << 16); rs1[pr][1] = x1 - __uint_as_float(h & 0xffff0000u);
            mp1[pr] = cvt_pk(rs1[pr][0], rs1[pr][1]);
            bspu[t][jh][0][d] = h; bspu[t][jh][1][d] = mp1[pr];
            pin_u(bspu[t][jh][0][d]); pin_u(bspu[t][jh][1][d]); pin_f(rs1[pr][0]); pin_f(rs1[pr][1]);
        } else {
            const float s0 = rs1[pr][0] - __uint_as_float(mp1[pr] << 16), s1 = rs1[pr][1] - __uint_as_float(mp1[pr] & 0xffff0000u);
            bspu[t][jh][2][d] = cvt_pk(s0, s1);
            pin_u(bspu[t][jh][2][d]);
        }
    };
    // tanh as a pipeline over the slots: v_exp, v_add, v_rcp, v_fma of ONE value are a dependent chain of ~50 cycles (two of them
    // transcendental), which a lone wave cannot hide inside one MFMA shadow -- so slot S issues exp(S), add(S - 1), rcp(S - 2),
    // fma(S - 3): four independent instructions, each a whole MFMA behind the one it depends on (fast_tanh_prescaled's
    // arithmetic, stage by stage)
    float tq[3];   // the pipeline registers: exp, exp + 1, rcp of the values in flight (one each)
    auto tanh_stages = [&](const f32x16& a, float* th, int S) {
        if (S >= 3 && S < 19) { th[S - 3] = 1.0f - 2.0f * tq[2]; pin_f(th[S - 3]); }
        if (S >= 2 && S < 18) { tq[2] = __builtin_amdgcn_rcpf(tq[1]); pin_f(tq[2]); }
        if (S >= 1 && S < 17) { tq[1] = tq[0] + 1.0f; pin_f(tq[1]); }
        if (S < 16) { tq[0] = __builtin_amdgcn_exp2f(a[S]); pin_f(tq[0]); }
    };
    // slots 0-18 the tanh pipeline, then the 16 halves of the split, one per slot (n_slots = 30: the last four slots take two)
    auto finish1 = [&](int t, int S, int n_slots) {
        if (S < 19) tanh_stages(acc1[t], th1[t], S);
        if (n_slots == 30) {
            if (S == 18) split_half(t, 0);
            else if (S >= 19 && S < 26) split_half(t, S - 18);
            else if (S >= 26) { split_half(t, 2 * S - 44); split_half(t, 2 * S - 43); }
        } else if (S >= 19 && S < 35) {
            split_half(t, S - 19);
        }
    };
    // The vector work behind layer 2 (tanh + layer 3 of its 64 values V = 16 tile + register, in mlp3_layer3's order) as a
    // pipeline over the slots of the chains 1-3, three slots per value: on the slot clock G = 48 chain + slot, value V issues exp
    // at 48 + 3 V, add one slot later, rcp, fma (= tanh), and its four layer-3 FMAs at 52 + 3 V -- so a slot holds (exp, fma) or
    // (add, 4 FMAs) or (rcp): within one MFMA shadow, every instruction a whole MFMA behind its producer.  Values 48-63 (the last
    // tile) and the last two stages of value 47 have no chain left to sit beside: the tail.  The row of W3 a value needs is read
    // a step EARLIER (a ds_read_b128 followed by its own wait is ~100 cycles that a lone wave cannot hide, 64 times per tile).
    float q2[4];   // exp, exp + 1, rcp, tanh of the values in flight
    float4 w3r[4]; // rows of W3, a ring by V & 3
    auto w3row = [&](int V) { w3r[V & 3] = W3t[32 * (V >> 4) + acc_row(V & 15, half)]; };
    auto l3 = [&](int V) {
        const float4 w = w3r[V & 3];
        o[0] = fmaf(w.x, q2[3], o[0]); o[1] = fmaf(w.y, q2[3], o[1]); o[2] = fmaf(w.z, q2[3], o[2]); o[3] = fmaf(w.w, q2[3], o[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) pin_f(o[k]);
    };
    auto finish2 = [&](const f32x16 (&a2)[2], int G) {   // G >= 48
        const int ph = (G - 48) % 3, V = (G - 48) / 3;
        if (ph == 0) {
            if (V >= 1 && V <= 48) { q2[3] = 1.0f - 2.0f * q2[2]; pin_f(q2[3]); }            // tanh of value V - 1
            if (V < 48) { q2[0] = __builtin_amdgcn_exp2f(a2[(V >> 4) & 1][V & 15]); pin_f(q2[0]); }
        } else if (ph == 1) {
            if (V >= 1 && V <= 48) l3(V - 1);
            if (V < 48) { q2[1] = q2[0] + 1.0f; pin_f(q2[1]); }
        } else if (V < 48) {
            q2[2] = __builtin_amdgcn_rcpf(q2[1]); pin_f(q2[2]);
        }
    };
    // the rows of W3 whose FMAs (slot 52 + 3 V) fall into the step AFTER the one that starts at slot G0 (a step = 6 slots)
    auto w3_prefetch = [&](int G0) {
#pragma unroll
        for (int V = 0; V < 48; ++V)
            if (52 + 3 * V >= G0 + 6 && 52 + 3 * V < G0 + 12) w3row(V);
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = half ? 0.f : b3s[k];
    Frag cur = ld(f1(0, 0));
    bias(acc1[0], b1s, 0);
#pragma unroll
    for (int d = 0; d < 4; ++d) split_x_pair(0, d);
    __builtin_amdgcn_sched_barrier(0);
    // ---- layer 1: chain 0 beside the split of the window's k-steps 1-4 (a pair of elements in slots 0, 1, 3, 4 of step sk for
    // k-step sk + 1); chain t beside tanh + split of tile t - 1
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int sk = 0; sk < KS16; ++sk) {
            if (sk == 0 && t == 1) MLP3_STAMP(13);
            if (sk == 0 && t == 2) MLP3_STAMP(14);
            if (sk == 3 && t == 2) MLP3_STAMP(15);
            const Frag nxt = ld(sk + 1 < KS16 ? f1(t, sk + 1) : (t + 1 < NT ? f1(t + 1, 0) : f2(0, 0)));
            if (sk == 3 && t + 1 < NT) bias(acc1[t + 1], b1s, t + 1);   // the next chain's start values, a step ahead
            __builtin_amdgcn_sched_barrier(0);
            step6(acc1[t], cur, xsp(sk, 0), xsp(sk, 1), xsp(sk, 2), [&](int slot) {
                if (t > 0) finish1(t - 1, 6 * sk + slot, 30);
                else if (sk + 1 < KS16 && slot != 2 && slot != 5) split_x_pair(sk + 1, slot - (slot > 2));
            });
            if (sk + 1 == KS16) asm volatile("" : "+v"(acc1[t]));   // (the chain in VGPRs: an accumulator in AGPRs costs a v_accvgpr_read
                                                                    //  per value, 8 cycles each beside an MFMA)
            cur = nxt;
        }
    // ---- layer 2: chain 0 beside tanh + split of layer-1 tile 3 (its k-steps 6, 7 read that tile's terms: the four parts ride
    // on k-steps 0-3); chain t beside tanh + layer 3 of tile t - 1; the start values of a chain are read a few steps early
    // into `b2n` (its accumulator still holds the tile before last until the previous chain's last step)
    f32x16 acc2[2], b2n;
    bias(acc2[0], b2s, 0);
    __builtin_amdgcn_sched_barrier(0);
    MLP3_STAMP(10);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t > 0) acc2[t & 1] = b2n;
        if (t == 1) MLP3_STAMP(11);
#pragma unroll
        for (int u = 0; u < 2 * NT; ++u) {
            const bool last = t + 1 == NT && u + 1 == 2 * NT;
            const Frag nxt = last ? cur : ld(u + 1 < 2 * NT ? f2(t, u + 1) : f2(t + 1, 0));
            w3_prefetch(48 * t + 6 * u);
            if (u == 5 && t + 1 < NT) bias(b2n, b2s, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            step6(acc2[t & 1], cur, bsp(u >> 1, u & 1, 0), bsp(u >> 1, u & 1, 1), bsp(u >> 1, u & 1, 2), [&](int slot) {
                if (t == 0) finish1(NT - 1, 6 * u + slot, 48);
                else finish2(acc2, 48 * t + 6 * u + slot);
            });
            if (u + 1 == 2 * NT) asm volatile("" : "+v"(acc2[t & 1]));   // (see acc1)
            cur = nxt;
        }
    }
    MLP3_STAMP(12);
    {   // the tail: value 47's last two stages (its row of W3 was read beside the last step), then the last tile's sixteen values
        // with nothing to sit beside -- their rows of W3 at once, tanh four at a time, then the FMAs in order
        finish2(acc2, 192); finish2(acc2, 193);
        float4 wt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) wt[j] = W3t[32 * (NT - 1) + acc_row(j, half)];
        float h2[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) h2[j] = fast_tanh_prescaled(acc2[(NT - 1) & 1][j]);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            o[0] = fmaf(wt[j].x, h2[j], o[0]); o[1] = fmaf(wt[j].y, h2[j], o[1]);
            o[2] = fmaf(wt[j].z, h2[j], o[2]); o[3] = fmaf(wt[j].w, h2[j], o[3]);
        }
    }
    mlp3_layer3_finish(o);
}

}  // namespace qenv
