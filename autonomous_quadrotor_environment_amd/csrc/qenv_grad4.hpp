// qenv_grad4.hpp -- k_mlp_grad_v4: forward recompute, fused PPO loss, backward AND the weight / bias gradients of one head of
// the reference's N = 128, T = 5 networks (environment/controller/ppo.py:172-203, visual_landing/ppo_trainer.py:96-131) in ONE
// launch, with no activation ever written to HBM.
//
// Rounds 1-4 ran the backward pass one wave per 32-sample tile (k_mlp_bwd_v3) and wrote H1, H2, dZ2, dZ1 (4 x 128 rows x 4 B per
// sample) for three split-K weight-gradient launches to read back: 37 GB per head and epoch at config 5's 8.4 M samples, >= 12 x
// the bytes of the samples themselves.  A weight gradient contracts over the SAMPLE index, which an accumulator tile carries on
// its lanes: the operand has to be transposed, and one wave cannot hold the 128 x 128 accumulator of dW2 beside its tile.
// Here the FOUR waves of a workgroup (one per SIMD, 512 registers each) share one 32-sample tile: wave w owns hidden rows
// 32 w .. 32 w + 31 of every layer -- its slice of W2 and of W2^T lives in its registers as ready MFMA fragments for the whole
// launch, its slice of the weight gradients in accumulator registers (dW2: 4 tiles, dW1: 3, dW3 / db2: 1) -- and the waves
// exchange activations through LDS images laid out [sample][hidden] in 16-bit terms.  One image serves both uses of an
// activation: read by rows (ds_read_b128) it is the B operand of the next layer (k = hidden index), read through gfx950's
// transposing ds_read_b64_tr_b16 it is an operand of a weight-gradient product (k = sample index) -- the transpose is free
// (cdna_hip_programming.md T10, image (b); the address maps are the ones tools/ubench/dual_image.hip verifies on the hardware).
//
// Arithmetic.  The FORWARD half is k_mlp_v3's: three-way split bf16 operands, six products per 16-k step smallest first,
// pre-scaled hidden layers (fp32-accurate; the policy's log-probabilities are what the rollout computed, up to the
// association of layer 3's 128-term sum, which four waves now form as 4 x 32).  Everything on the gradient side -- W2^T dZ2 and
// the three weight-gradient products -- runs on TWO-way split operands (hi + mid: 16 significant bits, three products per
// step): a relative 2^-17 per operand on sums over 10^5..10^7 noisy samples (measured against fp64 autograd:
// tests/test_gpu_ppo_rollout.py, tests/tools/fuzz_learner.py).  Bias gradients ride the matrix cores: a ones column in the
// observation image (feature 75) makes db1 column 75 of dW1, a constant ones fragment makes db2 column 4 of the dW3 tile.
//
// Per tile and wave: 154 MFMAs (layer 1: 30, layer 2: 48, dW3 / db2: 10, dW2: 24, W2^T dZ2: 24, dW1: 18) against the 126 + 3 x
// split-K launches of before; HBM traffic = the observations, once.
#pragma once
#define QENV_GRAD4_HPP 1
#include "qenv_mlp_common.hpp"

namespace qenv {
namespace grad4 {
using namespace mlp3;   // H = 128, NT = 4, KW = 75, KS16 = 5
// ---- LDS map (bytes)
constexpr int ROW = 256;                          // a sample's row of a shared image: 128 positions of 16 bits
constexpr int TERM = 32 * ROW;                    // one bf16 term of a [32 samples][128] image
constexpr int X_OFF = 0;                          // observations, 3 terms (positions 0..74 features, 75 = 1, the rest 0), TWO buffers:
constexpr int X_BUF = 3 * TERM;                   // the next tile's are stored while this tile computes
constexpr int HZ_OFF = X_OFF + 2 * X_BUF;         // H1 (3 terms), then dZ2 (2 terms) of the same tile
constexpr int PROW = 64, PTERM = 32 * PROW;       // a wave's private [32 samples][its 32 rows] image, 2 terms
constexpr int PV_OFF = HZ_OFF + 3 * TERM;         // [wave][term]: H2 own, then dZ1 own
constexpr int PV_WAVE = 2 * PTERM;
constexpr int DZ3_OFF = PV_OFF + 4 * PV_WAVE;     // [32 samples][16 positions] (32-byte rows), 2 terms: dz3 in positions 0..3
constexpr int DZ3_TERM = 32 * 32;
constexpr int ZERO_OFF = DZ3_OFF + 2 * DZ3_TERM;  // 1 280 bytes of zeros: the dz3 operand's columns 16..31
constexpr int PART_OFF = ZERO_OFF + 1280;         // layer 3's partial sums [wave][4][32] floats
constexpr int W3_OFF = PART_OFF + 2048;           // float4[128]: W3[0..3][h]
constexpr int B1_OFF = W3_OFF + 2048;             // 128 floats, pre-scaled
constexpr int B2_OFF = B1_OFF + 512;
constexpr int B3_OFF = B2_OFF + 512;              // 4 floats
constexpr int W1F_OFF = B3_OFF + 16;              // W1 fragments [tile][k-step][term][lane] uint4 (k_mlp_v3's layout)
constexpr int LDS_BYTES = W1F_OFF + NT * KS16 * 3 * 64 * 16;
static_assert(W1F_OFF % 16 == 0 && LDS_BYTES <= 160 * 1024, "k_mlp_grad_v4's LDS map");
// a workgroup's slot of partial sums (deterministic mode) = the flat gradient layout of a head: W1, b1, W2, b2, W3, b3
constexpr int O_W1 = 0, O_B1 = O_W1 + H * KW, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + 4 * H, SLOT = O_B3 + 4;

typedef short v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4s trd(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
}
// an MFMA fragment (k = 8 samples) from two transposing reads: rows R .. R + 3 (p0) and R + 8 .. R + 11 (p1) of the lane half.
// On image (b) the XOR partner of a chunk depends on the row's bits 2-3, so the two reads have addresses of their own; on the
// 64- and 32-byte-row images the second is the first + 8 rows.
__device__ __forceinline__ bf16x8 tr_frag2(const char* p0, const char* p1) {
    const v4s lo = trd(p0), hi = trd(p1);
    const short f8 __attribute__((ext_vector_type(8))) = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, f8);
}
__device__ __forceinline__ bf16x8 tr_frag_p(const char* p, int row_bytes) { return tr_frag2(p, p + 8 * row_bytes); }
// x = hi + mid to 16 significant bits
__device__ __forceinline__ void split2x8(const float x[8], bf16x8& hi, bf16x8& mid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h; mid[i] = (__bf16)(x[i] - (float)h);
    }
}
// LDS only: the waves of a workgroup meet; global loads in flight stay in flight (a __syncthreads waits for them too)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace grad4

// in-kernel stamps (diagnostic builds only: tools/build_variant.sh g4stamps "-DQENV_GRAD4_STAMPS" qenv_train.hip): the fourth
// tile of workgroup 7, s_memtime at the phase boundaries per wave, read back by qenv_debug_grad4_stamps
#ifdef QENV_GRAD4_STAMPS
__device__ unsigned long long g4_stamps[4 * 16];
#define G4_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (stamp_on && lane == 0) g4_stamps[wave * 16 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G4_STAMP(k)
#endif

// switches (A/B builds: tools/build_variant.sh NAME "-DG4_...=0/1" qenv_train.hip; tools/ab_grad4.sh, same box, alternating:
// MERGE_PRE 14.7 -> 15.1 ms per epoch, BALANCE_T3 and BUFLOAD_LOSS 14.7 -> 14.5 each)
#ifndef G4_MERGE_PRE
#define G4_MERGE_PRE 0      // 1: no scheduling fence between a phase's preamble and its chain of products
#endif
#ifndef G4_BALANCE_T3
#define G4_BALANCE_T3 1     // 1: the once-per-tile extras of the loss phase spread over the waves instead of all on wave 0
#endif
#ifndef G4_DB2_BRANCH
#define G4_DB2_BRANCH 1     // 1: db2's two products under a scalar branch (t == wave); 0: a zeroed ones fragment for the other tiles (12 more MFMAs: 14.3 against 13.4 ms)
#endif
#ifndef G4_NO_SGB
#define G4_NO_SGB 0         // 1: the chains' read-ahead left to hipcc's own scheduling (no sched_group_barrier sequences)
#endif
#ifndef G4_W3_HOLD
#define G4_W3_HOLD 0        // 1: this wave's 16 rows of W3 read once per tile and held across the loss phase (64 registers)
#endif
#ifndef G4_BUFLOAD_LOSS
#define G4_BUFLOAD_LOSS 1   // 1: the loss's per-sample inputs through raw buffer loads (scalar base, one lane offset)
#endif

struct GradArgs {
    BwdArgs b;                                   // observations, weights, the fused loss's inputs (h1 .. dz3 unused)
    float *gW1, *gb1, *gW2, *gb2, *gW3, *gb3;    // the head's gradients, accumulated into (fp32 atomics) ...
    float* partial;                              // ... or, if set, [gridDim.x][grad4::SLOT]: every workgroup stores its sums (deterministic mode)
    int64_t obs_stride;                          // elements between the observation blocks of consecutive chunks: 75 N (dense) or 15 N (the
};                                               // slice layout of an in-band rollout: the windows of consecutive steps overlap in 4 slices)

__global__ __launch_bounds__(256) void k_mlp_grad_v4(const GradArgs ga) {
    using namespace mlp3;
    using namespace grad4;
    const BwdArgs& a = ga.b;
    constexpr int K1 = KW;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    char* ldsb = reinterpret_cast<char*>(lds4);
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // ---- once per launch: W1 fragments, W3 rows, biases, the constant areas
    for (int c = threadIdx.x; c < NT * KS16 * 64; c += 256) {
        const int ln = c & 63, sk = (c >> 6) % KS16, t = (c >> 6) / KS16;
        const int k0 = 16 * sk + 8 * (ln >> 5);
        const float* src = a.W1 + (size_t)(32 * t + (ln & 31)) * K1;
        float w8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = (k0 + i) < K1 ? src[k0 + i] * TANH_PRESCALE : 0.f;
        bf16x8 sp[3];
        split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) lds4[W1F_OFF / 16 + ((t * KS16 + sk) * 3 + k) * 64 + ln] = __builtin_bit_cast(uint4, sp[k]);
    }
    for (int h = threadIdx.x; h < H; h += 256) {
        float4 w;
        w.x = a.W3[h]; w.y = a.n_out > 1 ? a.W3[(size_t)H + h] : 0.f; w.z = a.n_out > 2 ? a.W3[(size_t)2 * H + h] : 0.f;
        w.w = a.n_out > 3 ? a.W3[(size_t)3 * H + h] : 0.f;
        *reinterpret_cast<float4*>(ldsb + W3_OFF + 16 * h) = w;
        *reinterpret_cast<float*>(ldsb + B1_OFF + 4 * h) = a.b1[h] * TANH_PRESCALE;
        *reinterpret_cast<float*>(ldsb + B2_OFF + 4 * h) = a.b2[h] * TANH_PRESCALE;
    }
    if (threadIdx.x < 4) *reinterpret_cast<float*>(ldsb + B3_OFF + 4 * threadIdx.x) = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;
    for (int i = threadIdx.x; i < (PART_OFF - DZ3_OFF) / 4; i += 256) *reinterpret_cast<uint32_t*>(ldsb + DZ3_OFF + 4 * i) = 0u;
    // the observation image's positions 80 .. 127 are never written by a tile: zero them once (terms 0 .. 2)
    for (int i = threadIdx.x; i < 2 * X_BUF / 16; i += 256) lds4[X_OFF / 16 + i] = uint4{0u, 0u, 0u, 0u};

    // ---- this wave's slices of W2 (rows 32 w + col: layer 2's A operand, k_mlp_v3's fragment) and of W2^T (columns 32 w + col:
    // the A operand of dH1 = W2^T dZ2, k in the order the dZ2 image's row read delivers it), in registers for the whole launch
    bf16x8 A2[8][3], AT[8][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        float w8[8], v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 16 * ks + 8 * (e >> 2) + 4 * half + (e & 3);   // the hidden index of element e of this lane half in k-step ks
            w8[e] = a.W2[(size_t)(32 * wave + col) * H + kk] * TANH_PRESCALE;
            v8[e] = a.W2[(size_t)kk * H + 32 * wave + col];
        }
        split3x8(w8, A2[ks][0], A2[ks][1], A2[ks][2]);
        split2x8(v8, AT[ks][0], AT[ks][1]);
    }
    __syncthreads();

    // ---- lane parts of the LDS addresses
    const int xs = ((col & 3) << 2) | ((col >> 2) & 3);             // image (b): chunk ch of row r sits at ch ^ xs(r)
    const int row_rd = ROW * col + 16 * (half ^ xs);                 // row read of chunk 2 ks + half: row_rd ^ (32 ks)
    const int hz_wr = ROW * col + 16 * ((4 * wave + half) ^ xs);     // this wave's fragment jh of its own rows: hz_wr ^ (32 jh)
    const int q = (lane & 15) >> 2, p = lane & 3, gb = (lane >> 4) & 1;
    // transposed reads (rows 16 ks + 8 jg + 4 half + q; jg = the second read of tr_frag): the chunk's XOR partner is
    // (q << 2) | (2 jg + half), so the address is base + ((lane bits) ^ (tile / jg bits)) + literals
    const int tr_base = ROW * (4 * half + q);
    // sigma-ordered images (written as accumulator fragments: HZ, private): lane 4 q + p supplies chunk 2 blk + (p & 1), + 8 (p >> 1)
    const int trs_x = (q << 6) | ((((2 * gb + (p & 1)) ^ half) & 3) << 4);
    const int trs_b = tr_base + 8 * (p >> 1);
    // natural-order image (X): chunk 2 blk + (p >> 1), + 8 (p & 1)
    const int trn_x = (q << 6) | ((((2 * gb + (p >> 1)) ^ half) & 3) << 4);
    const int trn_b = tr_base + 8 * (p & 1);
    // private images: 64-byte rows, chunk c of row r at c ^ ((r >> 1) & 3): a lane-only term on the transposed reads
    const int pv_wr = PV_OFF + PV_WAVE * wave + PROW * col + 16 * (half ^ ((col >> 1) & 3));          // fragment jh: ^ (32 jh)
    const int pv_tr = PV_OFF + PV_WAVE * wave + PROW * (4 * half + q) + 16 * ((gb * 2 + (p & 1)) ^ ((2 * half + (q >> 1)) & 3)) + 8 * (p >> 1);
    // dz3 operand: columns 0 .. 15 from the dz3 image, columns 16 .. 31 from the zero area (another bank phase)
    const int dz3_tr = (gb ? ZERO_OFF + 128 : DZ3_OFF) + 32 * (4 * half + q) + 8 * p;

    // ---- accumulators of this wave's gradient slices
    f32x16 gw2[NT], gw1[3], g3;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) gw2[t][j] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) gw1[t][j] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) g3[j] = 0.f;
    float db3[4] = {0.f, 0.f, 0.f, 0.f};
    double loss_acc = 0.0, std_acc = 0.0;
    // the constant ones fragment: column 4 of the dW3 tile collects the row sums of dZ2 (db2)
    bf16x8 ones4, zeros8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ones4[i] = (__bf16)(col == 4 ? 1.f : 0.f); zeros8[i] = (__bf16)0.f; }

    const int64_t tiles_per_chunk = (a.N + 31) / 32, n_tiles = tiles_per_chunk * a.n_chunks;
    const uint32_t N32 = (uint32_t)a.N;
    // the observations of a tile: wave w fetches k-step w (rows 16 w + 8 half + i), wave 3 also k-step 4; range-checked raw loads
    float xw[2][8];
    auto request_x = [&](int c_, int r_) {
        const int64_t c = c_;
        const int64_t e = (int64_t)r_ * 32 + col;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.obs + (size_t)c * ga.obs_stride), 0,
                                                                            (int)((uint32_t)K1 * N32 * 4u), 0x00020000);
        const uint32_t voff = e < a.N ? ((uint32_t)e + 8u * (uint32_t)half * N32) * 4u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            xw[0][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (uint32_t)(16 * wave + i) * N32 * 4u, 0));
        if (wave == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xw[1][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (uint32_t)(64 + i) * N32 * 4u, 0));
        }
    };
    auto store_x = [&](int xoff) {   // split, and into the image at xoff: row = sample, chunk 2 sk + half
        bf16x8 sp[3];
        split3x8(xw[0], sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            *reinterpret_cast<uint4*>(ldsb + xoff + k * TERM + (row_rd ^ (32 * wave))) = __builtin_bit_cast(uint4, sp[k]);
        if (wave == 3) {
            if (half) xw[1][3] = 1.0f;   // feature 75: the ones column (W1's fragments are zero there; dW1's column 75 = db1)
            split3x8(xw[1], sp[0], sp[1], sp[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                *reinterpret_cast<uint4*>(ldsb + xoff + k * TERM + (row_rd ^ (32 * 4))) = __builtin_bit_cast(uint4, sp[k]);
        }
    };
    auto six = [&](f32x16& acc, bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    };
    auto three = [&](f32x16& acc, bf16x8 ah, bf16x8 am, bf16x8 bh, bf16x8 bm) {   // two-way split operands, smallest first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    };
    auto ldq = [&](int off) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ldsb + off)); };

    // a tile = (chunk c, position r in the chunk), advanced by gridDim.x without a division (scalar)
    int64_t tile = blockIdx.x;
    int tc = (int)(tile / tiles_per_chunk), tr = (int)(tile % tiles_per_chunk);
    const int tpc = (int)tiles_per_chunk, gstep = (int)gridDim.x;
    auto advance = [&](int& c_, int& r_) {
        r_ += gstep;
        while (r_ >= tpc) { r_ -= tpc; ++c_; }
    };
    if (tile < n_tiles) {
        request_x(tc, tr);
        store_x(X_OFF);
    }
    lds_barrier();
    int iter = 0;
    for (; tile < n_tiles; tile += gridDim.x, ++iter) {
        const int xcur = X_OFF + (iter & 1) * X_BUF, xnext = X_OFF + ((iter & 1) ^ 1) * X_BUF;
        [[maybe_unused]] const bool stamp_on = blockIdx.x == 7 && iter == 3;
        G4_STAMP(0);
        const int64_t c = tc;
        const int64_t e = (int64_t)tr * 32 + col;
        const bool valid = e < a.N;
        const uint32_t ec = (uint32_t)(valid ? e : a.N - 1);
        const bool more = tile + gridDim.x < n_tiles;
        advance(tc, tr);
        if (more) request_x(tc, tr);   // lands while this tile is computed; stored to the image after its last reader
        // the loss's per-sample inputs, requested now and used three phases later (an HBM miss is ~900 cycles)
        float l_ac[4] = {0.f, 0.f, 0.f, 0.f}, l_lo[4] = {0.f, 0.f, 0.f, 0.f}, l_adv = 0.f, l_ret = 0.f;
#if G4_BUFLOAD_LOSS
        auto rsrc = [&](const float* base, uint32_t rows) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(rows * N32 * 4u), 0x00020000);
        };
        const uint32_t lvo = ec * 4u;
        float l_wgt = 1.f;
        if (a.weight) l_wgt = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc(a.weight + (size_t)c * a.N, 1), lvo, 0, 0));
        if (a.loss_mode == 1) {
            const __amdgpu_buffer_rsrc_t ra = rsrc(a.actions + (size_t)c * 4 * a.N, 4), rl = rsrc(a.old_logprob + (size_t)c * 4 * a.N, 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                l_ac[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, lvo, (uint32_t)k * N32 * 4u, 0));
                l_lo[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, lvo, (uint32_t)k * N32 * 4u, 0));
            }
            l_adv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc(a.adv + (size_t)c * a.N, 1), lvo, 0, 0));
        } else {
            l_ret = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc(a.returns + (size_t)c * a.N, 1), lvo, 0, 0));
        }
#else
        const float l_wgt = a.weight ? (a.weight + (size_t)c * a.N)[ec] : 1.f;
        if (a.loss_mode == 1) {
            const float* ac = a.actions + (size_t)c * 4 * a.N;
            const float* lo = a.old_logprob + (size_t)c * 4 * a.N;
#pragma unroll
            for (int k = 0; k < 4; ++k) { l_ac[k] = ac[(uint32_t)k * N32 + ec]; l_lo[k] = lo[(uint32_t)k * N32 + ec]; }
            l_adv = (a.adv + (size_t)c * a.N)[ec];
        } else {
            l_ret = (a.returns + (size_t)c * a.N)[ec];
        }
#endif
        // ---- layer 1: this wave's 32 rows; the fragments of k-step s + 1 are requested before the products of step s
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = *reinterpret_cast<const float*>(ldsb + B1_OFF + 128 * wave + 16 * half + 4 * acc_row(j, 0));
        {
            bf16x8 fa[2][3], fb[2][3];
            auto l1_load = [&](int sk, bf16x8 (&A)[3], bf16x8 (&B)[3]) {
                const int xo = xcur + (row_rd ^ (32 * sk));
                const uint4* f = lds4 + W1F_OFF / 16 + ((wave * KS16 + sk) * 3) * 64 + lane;
#pragma unroll
                for (int k = 0; k < 3; ++k) { A[k] = __builtin_bit_cast(bf16x8, f[64 * k]); B[k] = ldq(xo + k * TERM); }
            };
            if (!G4_MERGE_PRE) __builtin_amdgcn_sched_barrier(0);
            l1_load(0, fa[0], fb[0]);
#pragma unroll
            for (int sk = 0; sk < KS16; ++sk) {
                if (sk + 1 < KS16) l1_load(sk + 1, fa[(sk + 1) & 1], fb[(sk + 1) & 1]);
                six(acc, fa[sk & 1][0], fa[sk & 1][1], fa[sk & 1][2], fb[sk & 1][0], fb[sk & 1][1], fb[sk & 1][2]);
            }
            // the order asked of the scheduler: step 0's six reads, then per step { the next step's reads, six products } -- a
            // step's operands are requested a whole step (192 matrix-pipe cycles) before its first product
#if !G4_NO_SGB
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int sk = 0; sk < KS16; ++sk) {
                if (sk + 1 < KS16) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        G4_STAMP(1);
        float h1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) h1[j] = fast_tanh_prescaled(acc[j]);
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {   // H1 into the shared image (three terms: layer 2's B operand; its own block read back
            bf16x8 sp[3];                  // transposed below: dW2's B operand)
            split3x8(&h1[8 * jh], sp[0], sp[1], sp[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4*>(ldsb + HZ_OFF + k * TERM + (hz_wr ^ (32 * jh))) = __builtin_bit_cast(uint4, sp[k]);
        }
        G4_STAMP(2);
        lds_barrier();   // B1: H1 complete
        G4_STAMP(3);
        // ---- layer 2 (A from registers, B = H1 by rows), layer 3's share of this wave
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = *reinterpret_cast<const float*>(ldsb + B2_OFF + 128 * wave + 16 * half + 4 * acc_row(j, 0));
        // this wave's own H1 block, transposed (hi, mid): dW2's B operand, kept in registers until dZ2 exists (the image is
        // overwritten by then)
        bf16x8 h1t[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int zo = HZ_OFF + ROW * 16 * ks + trs_b + (trs_x ^ (wave << 6)), z1 = HZ_OFF + ROW * (16 * ks + 8) + trs_b + (trs_x ^ ((wave << 6) | 32));
            h1t[ks][0] = tr_frag2(ldsb + zo, ldsb + z1);
            h1t[ks][1] = tr_frag2(ldsb + zo + TERM, ldsb + z1 + TERM);
        }
        {
            bf16x8 fb[2][3];
            auto l2_load = [&](int ks, bf16x8 (&B)[3]) {
                const int ho = HZ_OFF + (row_rd ^ (32 * ks));
#pragma unroll
                for (int k = 0; k < 3; ++k) B[k] = ldq(ho + k * TERM);
            };
            if (more) store_x(xnext);   // the next tile's observations (their buffer's last reader was the previous tile's dW1)
            if (!G4_MERGE_PRE) __builtin_amdgcn_sched_barrier(0);
            l2_load(0, fb[0]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) l2_load(ks + 1, fb[(ks + 1) & 1]);
                six(acc, A2[ks][0], A2[ks][1], A2[ks][2], fb[ks & 1][0], fb[ks & 1][1], fb[ks & 1][2]);
            }
#if !G4_NO_SGB
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        G4_STAMP(4);
        float h2[16];
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#if G4_W3_HOLD
        float4 w3r[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) w3r[j] = *reinterpret_cast<const float4*>(ldsb + W3_OFF + 512 * wave + 64 * half + 16 * acc_row(j, 0));
#endif
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            h2[j] = fast_tanh_prescaled(acc[j]);
#if G4_W3_HOLD
            const float4 w = w3r[j];
#else
            const float4 w = *reinterpret_cast<const float4*>(ldsb + W3_OFF + 512 * wave + 64 * half + 16 * acc_row(j, 0));
#endif
            o[0] = fmaf(w.x, h2[j], o[0]); o[1] = fmaf(w.y, h2[j], o[1]);
            o[2] = fmaf(w.z, h2[j], o[2]); o[3] = fmaf(w.w, h2[j], o[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // the two lane halves' sums meet (v_permlane32_swap: no LDS round trip)
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[k]), __float_as_uint(o[k]), false, false);
            o[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            if (half == 0) *reinterpret_cast<float*>(ldsb + PART_OFF + 512 * wave + 128 * k + 4 * col) = o[k];
        }
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {   // H2 own (dW3's A operand, transposed)
            bf16x8 sp[2];
            split2x8(&h2[8 * jh], sp[0], sp[1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) *reinterpret_cast<uint4*>(ldsb + k * PTERM + (pv_wr ^ (32 * jh))) = __builtin_bit_cast(uint4, sp[k]);
        }
        G4_STAMP(5);
        lds_barrier();   // B2: the partial sums are in; nobody reads H1 by rows any more
        G4_STAMP(6);
        // ---- output, loss, dz3 (every wave, the same bits), dZ2 of this wave's rows
        float outv[4], sd[4], dz3[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = *reinterpret_cast<const float*>(ldsb + B3_OFF + 4 * k);
#pragma unroll
            for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const float*>(ldsb + PART_OFF + 512 * w + 128 * k + 4 * col);
            if (a.final_tanh) v = fast_tanh(v);   // (|error| <= 1.2e-7: a 1e-6 relative change of the probability ratio)
            outv[k] = v;
            sd[k] = 0.f;
        }
        if (a.loss_mode == 1) {   // the actor's clipped surrogate (k_mlp_bwd's block; the ratio's exp in fp32: 1 ulp)
            float lp = 0.f, lp_old = 0.f, d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = l_ac[k] - outv[k];
                lp += -0.5f * d[k] * d[k] * a.inv_var + a.log_norm;
                lp_old += l_lo[k];
            }
            const float ratio = expf(lp - lp_old);
            const float A = l_adv;
            const float clipped = fminf(fmaxf(ratio, 1.f - a.eps_clip), 1.f + a.eps_clip);
            const float s1 = ratio * A, s2 = clipped * A;
            const bool use1 = s1 <= s2;
            const bool live = (use1 || (ratio > 1.f - a.eps_clip && ratio < 1.f + a.eps_clip)) && l_wgt != 0.f && valid;
#pragma unroll
            for (int k = 0; k < 4; ++k) sd[k] = live ? -A * ratio * d[k] * a.inv_var * a.inv_B : 0.f;
            if (wave == 0 && valid && half == 0 && l_wgt != 0.f) {
                loss_acc += -(double)(use1 ? s1 : s2);
                float dls = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) dls += d[k] * d[k] * a.inv_var - 1.f;
                if (live) std_acc += (double)(-A * ratio) * (double)(dls * a.inv_std);
            }
        } else {                  // the critic: c_value (V - ret)^2
            const float err = (l_wgt != 0.f && valid) ? outv[0] - l_ret : 0.f;
            sd[0] = 2.f * a.c_value * err * a.inv_B;
            if (wave == 0 && valid && half == 0) loss_acc += (double)err * (double)err;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dz3[k] = a.final_tanh ? sd[k] * (1.f - outv[k] * outv[k]) : sd[k];
        if (wave == (G4_BALANCE_T3 ? 1 : 0) && half == 0) {   // the dz3 operand image (two terms) and db3
            bf16x8 sp[2];
            float v8[8] = {dz3[0], dz3[1], dz3[2], dz3[3], 0.f, 0.f, 0.f, 0.f};
            split2x8(v8, sp[0], sp[1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint4 u = __builtin_bit_cast(uint4, sp[k]);
                *reinterpret_cast<uint2*>(ldsb + DZ3_OFF + k * DZ3_TERM + 32 * col) = uint2{u.x, u.y};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) db3[k] += dz3[k];
        }
        float dz[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#if G4_W3_HOLD
            const float4 w = w3r[j];
#else
            const float4 w = *reinterpret_cast<const float4*>(ldsb + W3_OFF + 512 * wave + 64 * half + 16 * acc_row(j, 0));
#endif
            const float g = fmaf(w.w, dz3[3], fmaf(w.z, dz3[2], fmaf(w.y, dz3[1], w.x * dz3[0])));
            dz[j] = g * (1.f - h2[j] * h2[j]);
        }
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {   // dZ2 replaces H1 in the shared image (two terms)
            bf16x8 sp[2];
            split2x8(&dz[8 * jh], sp[0], sp[1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) *reinterpret_cast<uint4*>(ldsb + HZ_OFF + k * TERM + (hz_wr ^ (32 * jh))) = __builtin_bit_cast(uint4, sp[k]);
        }
        G4_STAMP(7);
        lds_barrier();   // B3: dZ2 and the dz3 operand complete
        G4_STAMP(8);
        // ---- dW3^T of this wave's rows (columns 0 .. 3 of the g3 tile): A = H2 own, B = dz3, both transposed
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int po = pv_tr + PROW * 16 * ks;
            const bf16x8 ah = tr_frag_p(ldsb + po, PROW), am = tr_frag_p(ldsb + po + PTERM, PROW);
            const int zo = dz3_tr + 32 * 16 * ks;
            const bf16x8 bh = tr_frag_p(ldsb + zo, 32), bm = tr_frag_p(ldsb + zo + (gb ? 0 : DZ3_TERM), 32);
            three(g3, ah, am, bh, bm);
        }
        // ---- dH1 = W2^T dZ2 of this wave's rows (A from registers, B = dZ2 by rows)
        G4_STAMP(9);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        {
            bf16x8 fb[2][2];
            auto bw_load = [&](int ks, bf16x8 (&B)[2]) {
                const int ho = HZ_OFF + (row_rd ^ (32 * ks));
                B[0] = ldq(ho); B[1] = ldq(ho + TERM);
            };
            if (!G4_MERGE_PRE) __builtin_amdgcn_sched_barrier(0);
            bw_load(0, fb[0]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) bw_load(ks + 1, fb[(ks + 1) & 1]);
                three(acc, AT[ks][0], AT[ks][1], fb[ks & 1][0], fb[ks & 1][1]);
            }
#if !G4_NO_SGB
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        G4_STAMP(10);
        // ---- dZ1 of this wave's rows (vector work) beside dW2's column block of this wave and db2 (column 4 of the g3 tile, from
        // this wave's own dZ2 rows): the products read dZ2 and the H1 fragments in registers only
#pragma unroll
        for (int j = 0; j < 16; ++j) dz[j] = acc[j] * (1.f - h1[j] * h1[j]);
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {   // ... replaces H2 in the wave's own image (g3's reads of it were issued earlier: LDS is in order per wave)
            bf16x8 sp[2];
            split2x8(&dz[8 * jh], sp[0], sp[1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) *reinterpret_cast<uint4*>(ldsb + k * PTERM + (pv_wr ^ (32 * jh))) = __builtin_bit_cast(uint4, sp[k]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {           // dZ2 rows 32 t .., transposed
                const int zo = HZ_OFF + ROW * 16 * ks + trs_b + (trs_x ^ (t << 6)), z1 = HZ_OFF + ROW * (16 * ks + 8) + trs_b + (trs_x ^ ((t << 6) | 32));
                const bf16x8 ah = tr_frag2(ldsb + zo, ldsb + z1), am = tr_frag2(ldsb + zo + TERM, ldsb + z1 + TERM);
                three(gw2[t], ah, am, h1t[ks][0], h1t[ks][1]);
                // this wave's own rows (t == wave): their sums into column 4 -- the same two products for every t with the ones
                // fragment zeroed for the others (no branch: the region stays one basic block for the scheduler)
#if G4_DB2_BRANCH
                if (t == wave) {   // (wave-uniform: a scalar branch around the two products)
                    g3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, ones4, g3, 0, 0, 0);
                    g3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ones4, g3, 0, 0, 0);
                }
#else
                const bf16x8 o4 = t == wave ? ones4 : zeros8;
                g3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, o4, g3, 0, 0, 0);
                g3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, o4, g3, 0, 0, 0);
#endif
            }
        }
        G4_STAMP(11);
        // ---- dW1 (and db1 in column 75): A = dZ1 own, B = the observations, both transposed
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int po = pv_tr + PROW * 16 * ks;
            const bf16x8 ah = tr_frag_p(ldsb + po, PROW), am = tr_frag_p(ldsb + po + PTERM, PROW);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int xo = xcur + ROW * 16 * ks + trn_b + (trn_x ^ (t << 6)), x1 = xcur + ROW * (16 * ks + 8) + trn_b + (trn_x ^ ((t << 6) | 32));
                three(gw1[t], ah, am, tr_frag2(ldsb + xo, ldsb + x1), tr_frag2(ldsb + xo + TERM, ldsb + x1 + TERM));
            }
        }
        G4_STAMP(12);
        G4_STAMP(13);
        G4_STAMP(14);
        lds_barrier();   // B0: this tile's images have no reader left; the next tile's observations are in place
        G4_STAMP(15);
    }

    // ---- the workgroup's sums leave: fp32 atomics, or its slot (added up in slot order by k_sum_grad_slots)
    float* slot = ga.partial ? ga.partial + (size_t)blockIdx.x * SLOT : nullptr;
    auto emit = [&](float* dst, int slot_off, float v) {
        if (slot) slot[slot_off] = v;
        else atomicAdd(dst, v);
    };
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int orow = 32 * t + acc_row(j, half), icol = 32 * wave + col;
            emit(ga.gW2 + (size_t)orow * H + icol, O_W2 + orow * H + icol, gw2[t][j]);
        }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int hrow = 32 * wave + acc_row(j, half), k = 32 * t + col;
            if (k < K1) emit(ga.gW1 + (size_t)hrow * K1 + k, O_W1 + hrow * K1 + k, gw1[t][j]);
            else if (k == K1) emit(ga.gb1 + hrow, O_B1 + hrow, gw1[t][j]);
        }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int hrow = 32 * wave + acc_row(j, half);
        if (col < a.n_out) emit(ga.gW3 + (size_t)col * H + hrow, O_W3 + col * H + hrow, g3[j]);
        else if (col == 4) emit(ga.gb2 + hrow, O_B2 + hrow, g3[j]);
    }
    if (wave == (G4_BALANCE_T3 ? 1 : 0)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = half == 0 ? db3[k] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && k < a.n_out) emit(ga.gb3 + k, O_B3 + k, v);
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) loss_acc += __shfl_xor(loss_acc, off);
        if (lane == 0) atomicAdd(&a.sums[a.loss_mode], loss_acc);
        if (a.loss_mode == 1) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) std_acc += __shfl_xor(std_acc, off);
            if (lane == 0) atomicAdd(&a.sums[4], std_acc);
        }
    }
}

// deterministic mode, second pass: element j of the head's gradients += the workgroups' slots in slot order
__global__ __launch_bounds__(256) void k_sum_grad_slots(const float* partial, int n_slots, int n_out, float* gW1, float* gb1, float* gW2,
                                                        float* gb2, float* gW3, float* gb3) {
    using namespace grad4;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= SLOT) return;
    float* dst;
    if (j < O_B1) dst = gW1 + j;
    else if (j < O_W2) dst = gb1 + (j - O_B1);
    else if (j < O_B2) dst = gW2 + (j - O_W2);
    else if (j < O_W3) dst = gb2 + (j - O_B2);
    else if (j < O_B3) { if (j - O_W3 >= n_out * mlp3::H) return; dst = gW3 + (j - O_W3); }
    else { if (j - O_B3 >= n_out) return; dst = gb3 + (j - O_B3); }
    float acc = 0.f;
    for (int b = 0; b < n_slots; ++b) acc += partial[(size_t)b * SLOT + j];
    *dst += acc;
}

}  // namespace qenv
