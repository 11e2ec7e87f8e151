// qenv_grad5.hpp -- k_mlp_grad_v5: forward recompute, fused PPO loss, backward AND the weight / bias gradients of one head of
// the reference's N = 128, T = 5 networks (environment/controller/ppo.py:172-203, visual_landing/ppo_trainer.py:96-131) in ONE
// launch, with no activation ever written to HBM.
//
// Rounds 1-4 ran the backward pass one wave per 32-sample tile (k_mlp_bwd_v3) and wrote H1, H2, dZ2, dZ1 (4 x 128 rows x 4 B per
// sample) for three split-K weight-gradient launches to read back: 37 GB per head and epoch at config 5's 8.4 M samples, >= 12 x
// the bytes of the samples themselves.  A weight gradient contracts over the SAMPLE index, which an accumulator tile carries on
// its lanes: the operand has to be transposed, and one wave cannot hold the 128 x 128 accumulator of dW2 beside its tile.
// Here the EIGHT waves of a workgroup (two per SIMD, 256 registers each) share one 32-sample tile: wave w owns hidden rows
// 16 w .. 16 w + 15 of every layer on v_mfma_f32_16x16x32_bf16 tiles (16 rows x 16 samples x 32 k) -- its slices of W2 and of
// W2^T live in its registers as ready MFMA fragments for the whole launch (48 + 32 registers), its slices of the weight
// gradients in accumulator registers (dW2: 8 tiles, dW1: 5, dW3: 1 -- 56 registers) -- and the waves exchange activations
// through LDS images laid out [sample][hidden] in 16-bit terms.  One image serves both uses of an activation: read by rows
// (ds_read_b128) it is the B operand of the next layer (k = hidden index), read through gfx950's transposing
// ds_read_b64_tr_b16 it is an operand of a weight-gradient product, which contracts over the tile's 32 samples in ONE k-step.
// (The first version of this launch, four waves of 32 rows on 32x32x16 tiles, filled the 512-register file with one wave per
// SIMD: a lone wave issues a vector instruction every ~5 cycles and nothing covers its LDS latencies -- 13.4 ms per epoch at
// config 5's horizon against 12.3 here; git history, round 5.)
//
// The images are plain 256-byte rows; 16-byte chunk ch of row r sits at ch ^ f(r), f(r) = (r & 15) ^ (r & 1 ? 12 : 0):
// conflict-free for the 16x16x32 row reads, for the transposing reads (a half's two blocks 8 rows apart) and for the
// observations' 16-byte stores; the accumulator fragments' 8-byte stores are two-way (16 rows of one chunk: unavoidable).
// tools/lds_layout_grad5.py replays every address formula below on a model of the LDS and of the transposing read and checks
// operands and banks by brute force (MI355X_MICROARCH.md, LDS).
//
// Arithmetic.  The FORWARD half is k_mlp_v3's: three-way split bf16 operands, six products per k-step smallest first,
// pre-scaled hidden layers (fp32-accurate; the policy's log-probabilities are what the rollout computed, up to the
// association of layer 3's 128-term sum, which eight waves now form as 8 x 16).  Everything on the gradient side -- W2^T dZ2 and
// the three weight-gradient products -- runs on TWO-way split operands (hi + mid: 16 significant bits, three products per
// step): a relative 2^-17 per operand on sums over 10^5..10^7 noisy samples (measured against fp64 autograd:
// tests/test_gpu_ppo_rollout.py, tests/tools/fuzz_learner.py).  db1 rides dW1 as a ones column of the observation image
// (feature 75); db2 and db3 are plain fp32 sums on the vector lanes.  The observation image is single-buffered (W1's
// fragments for eight waves take 60 KB of LDS): the next tile's observations are requested after the loss and stored behind
// a barrier of their own, beside the second half of dW2.
//
// Per tile and wave: 152 MFMAs of 16 cycles (layer 1: 36, layer 2: 48, dW3: 3, W2^T dZ2: 24, dW2: 24, dW1: 15; 36 % of the
// SIMD's time at two waves), ~600 vector instructions, 187 LDS instructions; HBM traffic = the observations, once.
#pragma once
#include "qenv_mlp_common.hpp"

namespace qenv {
namespace grad5 {
using namespace mlp3;   // H = 128, KW = 75
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
constexpr int NW = 8;                               // waves of a workgroup; wave w owns hidden rows 16 w .. 16 w + 15
// ---- LDS map (bytes)
constexpr int ROW = 256;                            // a sample's row of a shared image: 128 positions of 16 bits
constexpr int TERM = 32 * ROW;                      // one bf16 term of a [32 samples][128] image
constexpr int X_OFF = 0;                            // observations, 3 terms (positions 0..74 features, 75 = 1, 76..95 zero)
constexpr int H1_OFF = X_OFF + 3 * TERM;            // H1, 3 terms
constexpr int DZ2_OFF = H1_OFF + 3 * TERM;          // dZ2, 2 terms
constexpr int PTERM = 1408;                         // a [32 samples][16 positions] image: 32-byte rows, 128 bytes of padding after
constexpr int PV_OFF = DZ2_OFF + 2 * TERM;          // every 8 rows (see p_off); [wave][term]: H2 own, then dZ1 own
constexpr int DZ3_OFF = PV_OFF + NW * 2 * PTERM;    // the same layout, 2 terms: dz3 in positions 0..3, zeros beyond
constexpr int ZCH_OFF = DZ3_OFF + 2 * PTERM;        // one zero chunk: W1's fragment of positions 80..95
constexpr int PART_OFF = ZCH_OFF + 16;              // layer 3's partial sums [wave][tile half][16] float4
constexpr int W3_OFF = PART_OFF + NW * 2 * 16 * 16; // float4[128]: W3[0..3][h]
constexpr int B3_OFF = W3_OFF + 2048;               // 4 floats
constexpr int B12_OFF = B3_OFF + 16;                // b1, b2 pre-scaled: [layer][128] floats
constexpr int W1F_OFF = B12_OFF + 1024;             // W1 fragments [wave]{[k-step 0..1][term][64 lanes], [k-step 2][term][32 lanes]} uint4
constexpr int W1F_WAVE = 2 * 3 * 1024 + 3 * 512;
constexpr int LDS_BYTES = W1F_OFF + NW * W1F_WAVE;
static_assert(W1F_OFF % 16 == 0 && LDS_BYTES <= 160 * 1024, "k_mlp_grad_v5's LDS map");
// a workgroup's slot of partial sums (deterministic mode) = the flat gradient layout of a head: W1, b1, W2, b2, W3, b3
constexpr int O_W1 = 0, O_B1 = O_W1 + H * KW, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + 4 * H, SLOT = O_B3 + 4;

// the chunk swizzle of the 256-byte-row images
__host__ __device__ constexpr int swz(int row) { return (row & 15) ^ ((row & 1) ? 12 : 0); }
__host__ __device__ constexpr int img_off(int row, int ch) { return ROW * row + 16 * (ch ^ swz(row)); }
// the 32-byte-row images: 8-byte slot s of row r
__host__ __device__ constexpr int p_off(int row, int slot) { return 32 * row + 128 * (row >> 3) + 8 * (slot ^ ((row >> 2) & 3)); }

typedef short v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4s trd(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
}
// an operand fragment whose k index is the SAMPLE (k = 8 g + e for lane group g): two transposing reads, rows 8 g .. 8 g + 3
// (p0) and 8 g + 4 .. 8 g + 7 (p1) of a 16-column block
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    const v4s lo = trd(p0), hi = trd(p1);
    const short f8 __attribute__((ext_vector_type(8))) = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, f8);
}
// x = hi + mid + lo (24 significant bits) / x = hi + mid (16), four values: the 8 bytes a lane stores per term
__device__ __forceinline__ void split3x4(const float x[4], uint2& hi, uint2& mid, uint2& lo) {
    bf16x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 a = (__bf16)x[i];
        const float r1 = x[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        h[i] = a; m[i] = b; l[i] = (__bf16)(r1 - (float)b);
    }
    hi = __builtin_bit_cast(uint2, h); mid = __builtin_bit_cast(uint2, m); lo = __builtin_bit_cast(uint2, l);
}
__device__ __forceinline__ void split2x4(const float x[4], uint2& hi, uint2& mid) {
    bf16x4 h, m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 a = (__bf16)x[i];
        h[i] = a; m[i] = (__bf16)(x[i] - (float)a);
    }
    hi = __builtin_bit_cast(uint2, h); mid = __builtin_bit_cast(uint2, m);
}
__device__ __forceinline__ void split2x8(const float x[8], bf16x8& hi, bf16x8& mid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h; mid[i] = (__bf16)(x[i] - (float)h);
    }
}
// the waves meet (LDS only: global loads in flight stay in flight); nothing is scheduled across -- without the fence hipcc
// consumes the loss's inputs where they are requested, four phases early, and waits for HBM there
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
}  // namespace grad5

#ifdef QENV_GRAD5_STAMPS
__device__ unsigned long long g5_stamps[8 * 16];
#define G5_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (stamp_on && lane == 0) g5_stamps[wave * 16 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G5_STAMP(k)
#endif

struct GradArgs {
    BwdArgs b;                                   // observations, weights, the fused loss's inputs (h1 .. dz3 unused)
    float *gW1, *gb1, *gW2, *gb2, *gW3, *gb3;    // the head's gradients, accumulated into (fp32 atomics) ...
    float* partial;                              // ... or, if set, [gridDim.x][grad5::SLOT]: every workgroup stores its sums (deterministic mode)
    int64_t obs_stride;                          // elements between the observation blocks of consecutive chunks: 75 N (dense) or 15 N (the
};                                               // slice layout of an in-band rollout: the windows of consecutive steps overlap in 4 slices)

__global__ __launch_bounds__(512) void k_mlp_grad_v5(const GradArgs ga) {
    using namespace mlp3;
    using namespace grad5;
    const BwdArgs& a = ga.b;
    constexpr int K1 = KW;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    char* ldsb = reinterpret_cast<char*>(lds4);
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // ---- once per launch: this wave's W1 fragments (A operand of layer 1: row 16 w + c16, k = 32 ks + 8 g + e), W3, b3, the zeros
    for (int ks = 0; ks < 3; ++ks) {
        float w8[8];
        const float* src = a.W1 + (size_t)(16 * wave + c16) * K1;
        const int k0 = 32 * ks + 8 * g;
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = (k0 + i) < K1 ? src[k0 + i] * TANH_PRESCALE : 0.f;
        bf16x8 sp[3];
        split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (ks < 2) *reinterpret_cast<uint4*>(ldsb + W1F_OFF + W1F_WAVE * wave + (ks * 3 + k) * 1024 + 16 * lane) = __builtin_bit_cast(uint4, sp[k]);
            else if (lane < 32) *reinterpret_cast<uint4*>(ldsb + W1F_OFF + W1F_WAVE * wave + 6144 + k * 512 + 16 * lane) = __builtin_bit_cast(uint4, sp[k]);
        }
    }
    for (int h = threadIdx.x; h < H; h += 512) {
        float4 w;
        w.x = a.W3[h]; w.y = a.n_out > 1 ? a.W3[(size_t)H + h] : 0.f; w.z = a.n_out > 2 ? a.W3[(size_t)2 * H + h] : 0.f;
        w.w = a.n_out > 3 ? a.W3[(size_t)3 * H + h] : 0.f;
        *reinterpret_cast<float4*>(ldsb + W3_OFF + 16 * h) = w;
    }
    for (int h = threadIdx.x; h < 2 * H; h += 512)
        *reinterpret_cast<float*>(ldsb + B12_OFF + 4 * h) = (h < H ? a.b1[h] : a.b2[h - H]) * TANH_PRESCALE;
    if (threadIdx.x < 4) *reinterpret_cast<float*>(ldsb + B3_OFF + 4 * threadIdx.x) = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;
    // the observation image's positions 80 .. 127 are never written by a tile, the dz3 image's positions 4 .. 15 neither
    for (int i = threadIdx.x; i < 3 * TERM / 16; i += 512) lds4[X_OFF / 16 + i] = uint4{0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < (PART_OFF - DZ3_OFF) / 4; i += 512) *reinterpret_cast<uint32_t*>(ldsb + DZ3_OFF + 4 * i) = 0u;

    // ---- this wave's slices of W2 (rows 16 w + c16: layer 2's A operand) and of W2^T (columns 16 w + c16: the A operand of
    // dH1 = W2^T dZ2), in registers for the whole launch; its biases in accumulator layout (rows 16 w + 4 g + r)
    bf16x8 A2[4][3], AT[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float w8[8], v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 32 * ks + 8 * g + e;
            w8[e] = a.W2[(size_t)(16 * wave + c16) * H + kk] * TANH_PRESCALE;
            v8[e] = a.W2[(size_t)kk * H + 16 * wave + c16];
        }
        split3x8(w8, A2[ks][0], A2[ks][1], A2[ks][2]);
        split2x8(v8, AT[ks][0], AT[ks][1]);
    }
    __syncthreads();

    // ---- lane parts of the LDS addresses
    const int fc = swz(c16);
    int rr_base = ROW * c16 + 16 * (g ^ fc);                                   // row read, tile half n, k-step ks: 4096 n + (rr_base ^ 64 ks)
    int aw_base = ROW * c16 + 16 * ((2 * wave + (g >> 1)) ^ fc) + 8 * (g & 1);   // an accumulator fragment's 8 bytes, half n: + 4096 n
    const int q = c16 >> 2, p = c16 & 3;
    // transposing reads of column block t: rows 8 g + q (tr0) and 8 g + 4 + q (tr1); the block's bits go in by XOR (32 t)
    int tr0 = ROW * (8 * g + q) + 16 * ((p >> 1) ^ swz(8 * g + q)) + 8 * (p & 1);
    int tr1 = ROW * (8 * g + q + 4) + 16 * ((p >> 1) ^ swz(8 * g + q + 4)) + 8 * (p & 1);
    int pw_base = PV_OFF + 2 * PTERM * wave + p_off(c16, g);                   // the wave's own 32-byte-row image, half n: + 768 n
    int pt0 = p_off(8 * g + q, p), pt1 = p_off(8 * g + q + 4, p);               // its transposing reads (and the dz3 image's)
    int w1f = W1F_OFF + W1F_WAVE * wave + 16 * lane;   // (int: the k-steps' and terms' offsets fold into the reads' immediates)
    int w1f2 = lane < 32 ? W1F_OFF + W1F_WAVE * wave + 6144 + 16 * lane : ZCH_OFF;   // k-step 2: positions 80 .. 95 are
    int w1f2s = lane < 32 ? 512 : 0;                                                        // the zero chunk for every term

    // ---- accumulators of this wave's gradient slices
    f32x4 gw2[8], gw1[5], g3;
#pragma unroll
    for (int t = 0; t < 8; ++t) gw2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 5; ++t) gw1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    g3 = f32x4{0.f, 0.f, 0.f, 0.f};
    float db3[4] = {0.f, 0.f, 0.f, 0.f}, db2[4] = {0.f, 0.f, 0.f, 0.f};
    double loss_acc = 0.0, std_acc = 0.0;

    const int64_t tiles_per_chunk = (a.N + 31) / 32, n_tiles = tiles_per_chunk * a.n_chunks;
    const uint32_t N32 = (uint32_t)a.N;
    // the observations of a tile: waves 0 .. 4, lane = (sample s = lane & 31, chunk 2 w + (lane >> 5)): features 8 chunk + i
    const int xs = lane & 31, xh = lane >> 5;
    float xw[8];
    auto request_x = [&](int c_, int r_) {
        const int64_t e = (int64_t)r_ * 32 + xs;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.obs + (size_t)c_ * ga.obs_stride), 0,
                                                                            (int)((uint32_t)K1 * N32 * 4u), 0x00020000);
        const uint32_t voff = e < a.N ? ((uint32_t)e + 8u * (uint32_t)xh * N32) * 4u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            xw[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (uint32_t)(16 * wave + i) * N32 * 4u, 0));
    };
    auto store_x = [&]() {   // split, and into the image: row = sample, chunk 2 w + xh
        if (wave == 4 && xh) xw[3] = 1.0f;   // feature 75: the ones column (W1's fragments are zero there; dW1's column 75 = db1)
        bf16x8 sp[3];
        split3x8(xw, sp[0], sp[1], sp[2]);
        const int xo = X_OFF + img_off(xs, 2 * wave + xh);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4*>(ldsb + xo + k * TERM) = __builtin_bit_cast(uint4, sp[k]);
    };
    auto six = [&](f32x4& acc, bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    };
    auto three = [&](f32x4& acc, bf16x8 ah, bf16x8 am, bf16x8 bh, bf16x8 bm) {   // two-way split operands, smallest first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    };
    auto ldq = [&](int off) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ldsb + off)); };
    auto st8 = [&](int off, uint2 v) { *reinterpret_cast<uint2*>(ldsb + off) = v; };

    // a tile = (chunk c, position r in the chunk), advanced by gridDim.x without a division (scalar)
    int64_t tile = blockIdx.x;
    int tc = (int)(tile / tiles_per_chunk), tr = (int)(tile % tiles_per_chunk);
    const int tpc = (int)tiles_per_chunk, gstep = (int)gridDim.x;
    auto advance = [&](int& c_, int& r_) {
        r_ += gstep;
        while (r_ >= tpc) { r_ -= tpc; ++c_; }
    };
    if (tile < n_tiles && wave < 5) {
        request_x(tc, tr);
        store_x();
    }
    lds_barrier();
    [[maybe_unused]] int iter = 0;
    for (; tile < n_tiles; tile += gridDim.x, ++iter) {
        [[maybe_unused]] const bool stamp_on = blockIdx.x == 7 && iter == 3;
        G5_STAMP(0);
        // the lane parts are opaque per tile: as loop invariants hipcc hoists every base ^ block and base + literal sum out of the
        // tile loop -- dozens of registers, spilled -- instead of forming it where it is used (one vector instruction, or the
        // read's immediate)
        asm volatile("" : "+v"(rr_base), "+v"(aw_base), "+v"(tr0), "+v"(tr1), "+v"(pw_base), "+v"(pt0), "+v"(pt1), "+v"(w1f), "+v"(w1f2), "+v"(w1f2s));
        const int64_t c = tc;
        const int64_t e = (int64_t)tr * 32 + xs;     // the sample whose loss this lane evaluates: tile half g & 1, column c16 (= xs)
        const bool valid = e < a.N;
        const uint32_t ec = (uint32_t)(valid ? e : a.N - 1);
        const bool more = tile + gridDim.x < n_tiles;
        advance(tc, tr);
        // ---- layer 1: this wave's 16 rows, both tile halves (n: samples 16 n + c16).  Every chain below is written in the order
        // it should issue -- a k-step's operands are requested while the step before multiplies -- and the sched_group_barrier
        // sequence after it pins that order (hipcc otherwise reads a whole step, waits, multiplies: the LDS latency of every step
        // exposed)
        f32x4 acc[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[n] = *reinterpret_cast<const f32x4*>(ldsb + B12_OFF + 64 * wave + 16 * g);   // rows 16 w + 4 g + r
        {
            bf16x8 A[2][3], B[2][2][3];
            auto ldA = [&](int ks, bf16x8 (&f)[3]) {
#pragma unroll
                for (int k = 0; k < 3; ++k) f[k] = ks < 2 ? ldq(w1f + (ks * 3 + k) * 1024) : ldq(w1f2 + k * w1f2s);
            };
            auto ldB = [&](int ks, int n, bf16x8 (&f)[3]) {
#pragma unroll
                for (int k = 0; k < 3; ++k) f[k] = ldq(X_OFF + k * TERM + 4096 * n + (rr_base ^ (64 * ks)));
            };
            ldA(0, A[0]); ldB(0, 0, B[0][0]); ldB(0, 1, B[0][1]);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int cu = ks & 1, nx = cu ^ 1;
                if (ks + 1 < 3) { ldA(ks + 1, A[nx]); ldB(ks + 1, 0, B[nx][0]); }
                six(acc[0], A[cu][0], A[cu][1], A[cu][2], B[cu][0][0], B[cu][0][1], B[cu][0][2]);
                if (ks + 1 < 3) ldB(ks + 1, 1, B[nx][1]);
                six(acc[1], A[cu][0], A[cu][1], A[cu][2], B[cu][1][0], B[cu][1][1], B[cu][1][2]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                if (ks + 1 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                if (ks + 1 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        G5_STAMP(1);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float h1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[r] = fast_tanh_prescaled(acc[n][r]);
            uint2 sp[3];
            split3x4(h1, sp[0], sp[1], sp[2]);
#pragma unroll
            for (int k = 0; k < 3; ++k) st8(H1_OFF + k * TERM + 4096 * n + aw_base, sp[k]);
        }
        G5_STAMP(2);
        lds_barrier();   // B1: H1 complete
        G5_STAMP(3);
        // the loss's per-sample inputs, requested now and used two phases later (an HBM miss is ~900 cycles)
        float l_ac[4] = {0.f, 0.f, 0.f, 0.f}, l_lo[4] = {0.f, 0.f, 0.f, 0.f}, l_adv = 0.f, l_ret = 0.f;
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
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 2 (A from registers, B = H1 by rows)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[n] = *reinterpret_cast<const f32x4*>(ldsb + B12_OFF + 512 + 64 * wave + 16 * g);
        {
            bf16x8 B[2][2][3];
            auto ldB = [&](int ks, int n, bf16x8 (&f)[3]) {
#pragma unroll
                for (int k = 0; k < 3; ++k) f[k] = ldq(H1_OFF + k * TERM + 4096 * n + (rr_base ^ (64 * ks)));
            };
            ldB(0, 0, B[0][0]); ldB(0, 1, B[0][1]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cu = ks & 1, nx = cu ^ 1;
                if (ks + 1 < 4) ldB(ks + 1, 0, B[nx][0]);
                six(acc[0], A2[ks][0], A2[ks][1], A2[ks][2], B[cu][0][0], B[cu][0][1], B[cu][0][2]);
                if (ks + 1 < 4) ldB(ks + 1, 1, B[nx][1]);
                six(acc[1], A2[ks][0], A2[ks][1], A2[ks][2], B[cu][1][0], B[cu][1][1], B[cu][1][2]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                if (ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        G5_STAMP(4);
        // ---- tanh, layer 3's share of this wave (the lane groups' sums meet: v_permlane16_swap, v_permlane32_swap),
        // H2 own into the wave's image (dW3's A operand, transposed)
        float h2[2][4];
        {
            float o[2][4];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int k = 0; k < 4; ++k) o[n][k] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 w = *reinterpret_cast<const float4*>(ldsb + W3_OFF + 256 * wave + 64 * g + 16 * r);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    h2[n][r] = fast_tanh_prescaled(acc[n][r]);
                    o[n][0] = fmaf(w.x, h2[n][r], o[n][0]); o[n][1] = fmaf(w.y, h2[n][r], o[n][1]);
                    o[n][2] = fmaf(w.z, h2[n][r], o[n][2]); o[n][3] = fmaf(w.w, h2[n][r], o[n][3]);
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[n][k]), __float_as_uint(o[n][k]), false, false);
                    const float v = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
                    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    o[n][k] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
                }
                if (g == 0) *reinterpret_cast<float4*>(ldsb + PART_OFF + 512 * wave + 256 * n + 16 * c16) = float4{o[n][0], o[n][1], o[n][2], o[n][3]};
                uint2 sp[2];
                split2x4(h2[n], sp[0], sp[1]);
#pragma unroll
                for (int k = 0; k < 2; ++k) st8(pw_base + k * PTERM + 768 * n, sp[k]);
            }
        }
        G5_STAMP(5);
        lds_barrier();   // B2: the partial sums are in
        G5_STAMP(6);
        // the loss's inputs are first touched HERE (hipcc otherwise starts summing the old log-probabilities where they are
        // requested and waits for HBM there)
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(l_ac[k]), "+v"(l_lo[k]));
        asm volatile("" : "+v"(l_adv), "+v"(l_ret), "+v"(l_wgt));
        // ---- output, loss, dz3 of this lane's sample (tile half g & 1), the other half's from lane ^ 16 (v_permlane16_swap); dZ2 of this wave's rows
        float outv[4], sd[4], dz3[2][4];
        {
            f32x4 s4 = *reinterpret_cast<const f32x4*>(ldsb + B3_OFF);   // (vector adds: v_pk_add_f32, the waves' sums in wave order)
#pragma unroll
            for (int w = 0; w < NW; ++w) s4 += *reinterpret_cast<const f32x4*>(ldsb + PART_OFF + 512 * w + 256 * (g & 1) + 16 * c16);
            outv[0] = s4[0]; outv[1] = s4[1]; outv[2] = s4[2]; outv[3] = s4[3];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (a.final_tanh) outv[k] = fast_tanh(outv[k]);   // (|error| <= 1.2e-7: a 1e-6 relative change of the probability ratio)
            sd[k] = 0.f;
        }
        if (a.loss_mode == 1) {   // the actor's clipped surrogate (k_mlp_bwd's block; the ratio's exp in fp32)
            float lp = 0.f, lp_old = 0.f, d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = l_ac[k] - outv[k];
                lp += -0.5f * d[k] * d[k] * a.inv_var + a.log_norm;
                lp_old += l_lo[k];
            }
            const float ratio = __builtin_amdgcn_exp2f((lp - lp_old) * 1.44269504088896341f);   // (v_exp_f32: 1 ulp; |lp - lp_old| stays far below its range limits)
            const float A = l_adv;
            const float clipped = fminf(fmaxf(ratio, 1.f - a.eps_clip), 1.f + a.eps_clip);
            const float s1 = ratio * A, s2 = clipped * A;
            const bool use1 = s1 <= s2;
            const bool live = (use1 || (ratio > 1.f - a.eps_clip && ratio < 1.f + a.eps_clip)) && l_wgt != 0.f && valid;
#pragma unroll
            for (int k = 0; k < 4; ++k) sd[k] = live ? -A * ratio * d[k] * a.inv_var * a.inv_B : 0.f;
            if (wave == 0 && valid && lane < 32 && l_wgt != 0.f) {
                loss_acc += -(double)(use1 ? s1 : s2);
                float dls = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) dls += d[k] * d[k] * a.inv_var - 1.f;
                if (live) std_acc += (double)(-A * ratio) * (double)(dls * a.inv_std);
            }
        } else {                  // the critic: c_value (V - ret)^2
            const float err = (l_wgt != 0.f && valid) ? outv[0] - l_ret : 0.f;
            sd[0] = 2.f * a.c_value * err * a.inv_B;
            if (wave == 0 && valid && lane < 32) loss_acc += (double)err * (double)err;
        }
        {
            float own[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) own[k] = a.final_tanh ? sd[k] * (1.f - outv[k] * outv[k]) : sd[k];
            if (wave == 1 && lane < 32) {   // the dz3 operand image (two terms; row = sample, slot 0) and db3
                uint2 sp[2];
                split2x4(own, sp[0], sp[1]);
#pragma unroll
                for (int k = 0; k < 2; ++k) st8(DZ3_OFF + k * PTERM + p_off(xs, 0), sp[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) db3[k] += own[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // v_permlane16_swap of a value with itself: [0] = the even lane group's (tile half 0), [1] = the odd one's
                const auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(own[k]), __float_as_uint(own[k]), false, false);
                dz3[0][k] = __uint_as_float(r1[0]);
                dz3[1][k] = __uint_as_float(r1[1]);
            }
        }
        {
            float dz[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 w = *reinterpret_cast<const float4*>(ldsb + W3_OFF + 256 * wave + 64 * g + 16 * r);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float gg = fmaf(w.w, dz3[n][3], fmaf(w.z, dz3[n][2], fmaf(w.y, dz3[n][1], w.x * dz3[n][0])));
                    dz[n][r] = gg * (1.f - h2[n][r] * h2[n][r]);
                }
                db2[r] += dz[0][r] + dz[1][r];   // this lane's two samples of row 4 g + r; the 16 lanes of a group meet at the end
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                uint2 sp[2];
                split2x4(dz[n], sp[0], sp[1]);
#pragma unroll
                for (int k = 0; k < 2; ++k) st8(DZ2_OFF + k * TERM + 4096 * n + aw_base, sp[k]);
            }
        }
        G5_STAMP(7);
        lds_barrier();   // B3: dZ2 and the dz3 operand complete
        G5_STAMP(8);
        if (more && wave < 5) request_x(tc, tr);   // lands during the two phases to the barrier after dW1; no other load is in flight
        __builtin_amdgcn_sched_barrier(0);          // meanwhile (a spilled register's reload would wait for these)
        // ---- dW3^T of this wave's rows (columns 0 .. 3 of the g3 tile): A = H2 own, B = dz3, both transposed
        {
            const char* pv = ldsb + PV_OFF + 2 * PTERM * wave;
            const bf16x8 ah = tr_frag(pv + pt0, pv + pt1), am = tr_frag(pv + PTERM + pt0, pv + PTERM + pt1);
            const char* zv = ldsb + DZ3_OFF;
            const bf16x8 bh = tr_frag(zv + pt0, zv + pt1), bm = tr_frag(zv + PTERM + pt0, zv + PTERM + pt1);
            three(g3, ah, am, bh, bm);
        }
        G5_STAMP(9);
        // ---- dH1 = W2^T dZ2 of this wave's rows (A from registers, B = dZ2 by rows); the halves' products alternate
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            bf16x8 B[2][2][2];
            auto ldB = [&](int ks, bf16x8 (&f)[2][2]) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int k = 0; k < 2; ++k) f[n][k] = ldq(DZ2_OFF + k * TERM + 4096 * n + (rr_base ^ (64 * ks)));
            };
            ldB(0, B[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cu = ks & 1;
                if (ks + 1 < 4) ldB(ks + 1, B[cu ^ 1]);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AT[ks][1], B[cu][0][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AT[ks][1], B[cu][1][0], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AT[ks][0], B[cu][0][1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AT[ks][0], B[cu][1][1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AT[ks][0], B[cu][0][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AT[ks][0], B[cu][1][0], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        G5_STAMP(10);
        // ---- dZ1 of this wave's rows replaces H2 in the wave's own image (g3's reads of it were issued earlier: LDS is in order
        // per wave), then dW1 (and db1 in column 75): A = dZ1 own, B = the observations, both transposed.  That chain is all
        // latency (image reads, vector work, stores, transposing reads, fifteen products): the first half of dW2 -- this wave's
        // column block, A = dZ2 rows 16 t .., B = H1 own, both transposed, independent of dZ1 -- runs in its gaps; two output
        // tiles' products alternate and the pair after them is read meanwhile
        const char* hb = ldsb + H1_OFF;
        const bf16x8 bh = tr_frag(hb + (tr0 ^ (32 * wave)), hb + (tr1 ^ (32 * wave)));
        const bf16x8 bm = tr_frag(hb + TERM + (tr0 ^ (32 * wave)), hb + TERM + (tr1 ^ (32 * wave)));
        bf16x8 Z[2][2][2];   // [stage][tile of the pair][term]
        auto ldZ = [&](int t, bf16x8 (&f)[2]) {
            const char* z0 = ldsb + DZ2_OFF + (tr0 ^ (32 * t));
            const char* z1 = ldsb + DZ2_OFF + (tr1 ^ (32 * t));
            f[0] = tr_frag(z0, z1); f[1] = tr_frag(z0 + TERM, z1 + TERM);
        };
        auto dw2_pair = [&](int t2, bf16x8 (&z)[2][2]) {
            f32x4& ga_ = gw2[2 * t2];
            f32x4& gb_ = gw2[2 * t2 + 1];
            ga_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[0][1], bh, ga_, 0, 0, 0);
            gb_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[1][1], bh, gb_, 0, 0, 0);
            ga_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[0][0], bm, ga_, 0, 0, 0);
            gb_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[1][0], bm, gb_, 0, 0, 0);
            ga_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[0][0], bh, ga_, 0, 0, 0);
            gb_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[1][0], bh, gb_, 0, 0, 0);
        };
        ldZ(0, Z[0][0]); ldZ(1, Z[0][1]);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float dz[4];
            // H1 of this wave's rows back from the image (hi + mid + lo = the fp32 value): eight registers less across five phases
            bf16x4 t3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) t3[k] = __builtin_bit_cast(bf16x4, *reinterpret_cast<const uint2*>(ldsb + H1_OFF + k * TERM + 4096 * n + aw_base));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hh = ((float)t3[0][r] + (float)t3[1][r]) + (float)t3[2][r];
                dz[r] = acc[n][r] * (1.f - hh * hh);
            }
            uint2 sp[2];
            split2x4(dz, sp[0], sp[1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) st8(pw_base + k * PTERM + 768 * n, sp[k]);
        }
        ldZ(2, Z[1][0]); ldZ(3, Z[1][1]);
        dw2_pair(0, Z[0]);
        {
            const char* pv = ldsb + PV_OFF + 2 * PTERM * wave;
            const bf16x8 ah = tr_frag(pv + pt0, pv + pt1), am = tr_frag(pv + PTERM + pt0, pv + PTERM + pt1);
            bf16x8 X[2][2];
            auto ldX = [&](int t, bf16x8 (&f)[2]) {
                const char* x0 = ldsb + X_OFF + (tr0 ^ (32 * t));
                const char* x1 = ldsb + X_OFF + (tr1 ^ (32 * t));
                f[0] = tr_frag(x0, x1); f[1] = tr_frag(x0 + TERM, x1 + TERM);
            };
            ldX(0, X[0]);
            dw2_pair(1, Z[1]);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (t + 1 < 5) ldX(t + 1, X[(t + 1) & 1]);
                three(gw1[t], ah, am, X[t & 1][0], X[t & 1][1]);
            }
        }
        G5_STAMP(11);
        lds_barrier();   // Bx: the observation image has no reader left
        G5_STAMP(12);
        // ---- the next tile's observations into the image, beside the second half of dW2
        ldZ(4, Z[0][0]); ldZ(5, Z[0][1]);
        if (more && wave < 5) store_x();
        ldZ(6, Z[1][0]); ldZ(7, Z[1][1]);
        dw2_pair(2, Z[0]);
        dw2_pair(3, Z[1]);
        G5_STAMP(13);
        lds_barrier();   // B0: this tile's images have no reader left; the next tile's observations are in place
        G5_STAMP(14);
    }

    // ---- the workgroup's sums leave: fp32 atomics, or its slot (added up in slot order by k_sum_grad_slots)
    float* slot = ga.partial ? ga.partial + (size_t)blockIdx.x * SLOT : nullptr;
    auto emit = [&](float* dst, int slot_off, float v) {
        if (slot) slot[slot_off] = v;
        else atomicAdd(dst, v);
    };
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int orow = 16 * t + 4 * g + r, icol = 16 * wave + c16;
            emit(ga.gW2 + (size_t)orow * H + icol, O_W2 + orow * H + icol, gw2[t][r]);
        }
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hrow = 16 * wave + 4 * g + r, k = 16 * t + c16;
            if (k < K1) emit(ga.gW1 + (size_t)hrow * K1 + k, O_W1 + hrow * K1 + k, gw1[t][r]);
            else if (k == K1) emit(ga.gb1 + hrow, O_B1 + hrow, gw1[t][r]);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hrow = 16 * wave + 4 * g + r;
        if (c16 < a.n_out) emit(ga.gW3 + (size_t)c16 * H + hrow, O_W3 + c16 * H + hrow, g3[r]);
        float v = db2[r];   // the group's 16 lanes hold the samples 16 n + c16 of row 4 g + r
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (c16 == 0) emit(ga.gb2 + hrow, O_B2 + hrow, v);
    }
    if (wave == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = lane < 32 ? db3[k] : 0.f;
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
    using namespace grad5;
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
