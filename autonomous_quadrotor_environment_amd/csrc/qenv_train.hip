// qenv_train.hip -- the optimiser half of PPO.update (environment/controller/ppo.py:172-203,
// visual_landing/ppo_trainer.py:96-131) for the 75-N-N-4 / 75-N-N-1 MLPs of model.py: backward pass of the
// tanh MLP on the fp32 MFMA pipe, weight/bias gradients as split-K GEMMs over the samples, Adam.
//
// Sample layout = the rollout's time-major chunked SoA: sample (c, e) of a matrix with R rows lives at
// base + (c*R + r)*N + e (chunk c = one rollout step, N envs).  Every kernel works on 32- or 64-sample
// tiles that never straddle a chunk.
#include "qenv_host.hpp"
#include "qenv_mlp_common.hpp"

namespace qenv {

struct BwdArgs {
    int64_t N;          // envs per chunk
    int32_t n_chunks, K, n_out, final_tanh;
    const float* obs;   // [c][K][N]
    const float* seed;  // [c][n_out][N]  d loss / d output
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    float *h1, *h2, *dz2, *dz1;  // [c][H][N] each
    float* dz3;                  // [c][4][N]
    // loss fused into the backward pass (qenv_mlp_backward_ppo): the seed d loss / d output is computed right here
    // from the recomputed forward output instead of being read -- no separate forward launch, no loss launch
    int32_t loss_mode;           // 0: seed given; 1: clipped surrogate of the actor (ppo.py:185-196); 2: c_value * MSE of the critic
    const float *actions, *old_logprob;  // [c][4][N]
    const float *adv, *returns;          // [c][N]
    float inv_var, inv_std, log_norm, eps_clip, c_value, inv_B;
    double* sums;                // qenv_ppo_loss's {B, sum actor, sum (V - ret)^2, entropy * B}: [1] (mode 1) or [2] (mode 2) accumulated;
                                 // mode 1 also adds sum_s d actor_s / d std to [4] (learned action std, model.py:45-48)
    const float* weight;         // null, or [c][N] sample weights of the fused loss (1 / 0: the `valid` rows of an in-band warm-up
                                 // rollout, qenv_ppo_step); inv_B is then 1 / (number of valid samples)
};

// Stores of a tile go through raw buffer instructions: a lane beyond the chunk (e >= N in the last tile of a chunk) or a
// padded hidden row gets the offset 0xFFFFFFFF and the hardware's range check drops it -- no exec-masked branch per
// store between the MFMAs (the predicated global stores put 130 branches into a tile's instruction stream), and
// no address VGPRs: one per-lane offset register serves every store of the tile, the row offset is scalar.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(float* base, uint32_t rows, uint32_t n32) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(rows * n32 * 4u), 0x00020000);
}
// voff: the lane's byte offset inside a row pair (or 2^31 for a lane that must not store: with rows*N*4 < 2^31 the
// sum stays beyond num_records without wrapping); soff: the row's byte offset, wave-uniform (SALU, no VGPR per store)
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);  // b32 = an integer operand: a float would be CONVERTED
}

// Forward (recomputed) + backward through the activations for one 32-sample tile per wave:
//   H1 = tanh(W1 x + b1), H2 = tanh(W2 H1 + b2), out = [tanh](W3 H2 + b3)
//   dz3 = seed * (1 - out^2 | 1), dZ2 = (W3^T dz3) * (1 - H2^2), dZ1 = (W2^T dZ2) * (1 - H1^2)
// Same operand orientation as k_mlp (accumulator tile = next B operand); W2 is kept ONCE in LDS as
// [k = input][h = output] with rows padded by one word so both the forward read (32 consecutive h) and the
// transposed backward read (stride H+1) are bank-conflict free.
// H = 128 keeps 2 x 64 activation registers + the gradient tile live (223 VGPRs).  BLOCK = 512 (two waves per SIMD, the
// large-batch choice: one wave's tanh / stores / LDS waits under the other's MFMAs) or 256 (one wave per SIMD on twice
// as many CUs when there is at most one tile per wave) -- launch_bwd picks.
template <int H, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_mlp_bwd(const BwdArgs a) {
    constexpr int HP = H < 32 ? 32 : H, NT = HP / 32, LD2 = HP + 1;
    constexpr int WPB = BLOCK / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K1 = a.K, K1P = (K1 + 1) & ~1;
    float* W1t = lds;                        // [K1P][HP]
    float* W2t = W1t + (size_t)K1P * HP;     // [HP][HP+1]: W2t[k*LD2 + h] = W2[h][k]
    float* W3t = W2t + (size_t)HP * LD2;     // [HP][32]: W3t[h*32 + o] = W3[o][h], output rows o >= n_out zero
    float* W3r = W3t + 32 * HP;              // [4][HP]:  W3r[o*HP + h] = W3[o][h] (the transposed product's A operand)
    float* b1s = W3r + 4 * HP;
    float* b2s = b1s + HP;
    float* b3s = b2s + HP;                   // [32]
    for (int idx = threadIdx.x; idx < K1P * HP; idx += BLOCK) {
        const int k = idx / HP, h = idx % HP;
        W1t[idx] = (k < K1 && h < H) ? a.W1[(size_t)h * K1 + k] : 0.f;
    }
    for (int idx = threadIdx.x; idx < HP * HP; idx += BLOCK) {
        const int k = idx / HP, h = idx % HP;
        W2t[k * LD2 + h] = (k < H && h < H) ? a.W2[(size_t)h * H + k] : 0.f;
    }
    for (int idx = threadIdx.x; idx < 32 * HP; idx += BLOCK) {
        const int h = idx / 32, o = idx % 32;
        W3t[idx] = (o < a.n_out && h < H) ? a.W3[(size_t)o * H + h] : 0.f;
    }
    for (int idx = threadIdx.x; idx < 4 * HP; idx += BLOCK) {
        const int o = idx / HP, h = idx % HP;
        W3r[idx] = (o < a.n_out && h < H) ? a.W3[(size_t)o * H + h] : 0.f;
    }
    for (int h = threadIdx.x; h < HP; h += BLOCK) { b1s[h] = h < H ? a.b1[h] : 0.f; b2s[h] = h < H ? a.b2[h] : 0.f; }
    if (threadIdx.x < 32) b3s[threadIdx.x] = (int)threadIdx.x < a.n_out ? a.b3[threadIdx.x] : 0.f;
    __syncthreads();

    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles_per_chunk = (a.N + 31) / 32, n_tiles = tiles_per_chunk * a.n_chunks;
    double loss_acc = 0.0;  // this lane's share of the loss sum (fused-loss modes), folded once after the walk
    double std_acc = 0.0;   // ... and of sum_s d actor_s / d std (actor mode)
    for (int64_t tile = (int64_t)blockIdx.x * WPB + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WPB) {
        asm volatile("" ::: "memory");  // keep the (tile-invariant) LDS reads inside the loop (see k_mlp)
        // chunk index and chunk bases are wave-uniform (SGPRs); everything per lane is a 32-bit element offset
        const int64_t c = __builtin_amdgcn_readfirstlane((int)(tile / tiles_per_chunk));
        const int64_t e = (tile % tiles_per_chunk) * 32 + col;
        const bool valid = e < a.N;
        const uint32_t N32 = (uint32_t)a.N, ec = (uint32_t)(valid ? e : a.N - 1);
        const float* x = a.obs + (size_t)c * K1 * a.N;
        f32x16 h1[NT], h2[NT];
        // ---- layer 1
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) h1[t][j] = b1s[32 * t + acc_row(j, half)];
        auto load_x = [&](int s0, float (&dst)[8]) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int k = 2 * (s0 + s) + half;
                const float v = x[(uint32_t)(k < K1 ? k : K1 - 1) * N32 + ec];
                dst[s] = (k < K1 && 2 * (s0 + s) < K1P) ? v : 0.f;
            }
        };
        float xs[8];
        load_x(0, xs);
        for (int s0 = 0; s0 < K1P / 2; s0 += 8) {
            float xn[8];  // the next group's observations are requested before this group's MFMAs are issued
            load_x(s0 + 8, xn);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int kk = 2 * (s0 + s) + half;
                if (2 * (s0 + s) < K1P) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        h1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1t[kk * HP + 32 * t + col], xs[s], h1[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) xs[s] = xn[s];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) h1[t][j] = fast_tanh(h1[t][j]);
        // ---- layer 2, and layer 3 on the MFMA as one zero-padded 32-row tile (see k_mlp)
        f32x16 acc3;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc3[j] = b3s[acc_row(j, half)];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) h2[t][j] = b2s[32 * t + acc_row(j, half)];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    h2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(W2t[(32 * ti + acc_row(j, half)) * LD2 + 32 * t + col], h1[ti][j], h2[t], 0, 0, 0);
                    if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int j = 0; j < 16; ++j) h2[t][j] = fast_tanh(h2[t][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(W3t[(32 * t + acc_row(j, half)) * 32 + col], h2[t][j], acc3, 0, 0, 0);
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- output and its pre-activation gradient (made available to both lane halves)
        const __amdgpu_buffer_rsrc_t r_dz3 = rows_rsrc(a.dz3 + (size_t)c * 4 * a.N, 4, N32);
        float dz3[4], outv[4], sd[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = half == 0 ? acc3[o] : 0.f;   // output row o lives in register o of the lower lane half
            v += __shfl_xor(v, 32);
            if (a.final_tanh) v = tanhf(v);
            outv[o] = v;
            sd[o] = 0.f;
        }
        if (a.loss_mode == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < a.n_out) sd[o] = (a.seed + (size_t)c * a.n_out * a.N)[(uint32_t)o * N32 + ec];
        } else if (a.loss_mode == 1) {  // k_ppo_loss, actor part, same operation order
            const float* ac = a.actions + (size_t)c * 4 * a.N;
            const float* lo = a.old_logprob + (size_t)c * 4 * a.N;
            float lp = 0.f, lp_old = 0.f, d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = ac[(uint32_t)k * N32 + ec] - outv[k];
                lp += -0.5f * d[k] * d[k] * a.inv_var + a.log_norm;
                lp_old += lo[(uint32_t)k * N32 + ec];
            }
            const float ratio = (float)exp((double)(lp - lp_old));
            const float A = (a.adv + (size_t)c * a.N)[ec];
            const float clipped = fminf(fmaxf(ratio, 1.f - a.eps_clip), 1.f + a.eps_clip);
            const float s1 = ratio * A, s2 = clipped * A;
            const bool use1 = s1 <= s2;  // torch.min picks surr1 on ties
            const float wgt = a.weight ? (a.weight + (size_t)c * a.N)[ec] : 1.f;
            const bool live = (use1 || (ratio > 1.f - a.eps_clip && ratio < 1.f + a.eps_clip)) && wgt != 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sd[k] = live ? -A * ratio * d[k] * a.inv_var * a.inv_B : 0.f;
            if (valid && half == 0 && wgt != 0.f) {
                loss_acc += -(double)(use1 ? s1 : s2);
                // d log N(a; mean, std) / d std summed over the four action dimensions = sum_k (d_k^2 / std^2 - 1) / std
                float dls = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) dls += d[k] * d[k] * a.inv_var - 1.f;
                if (live) std_acc += (double)(-A * ratio) * (double)(dls * a.inv_std);
            }
        } else {  // critic: c_value * (V - ret)^2
            const float wgt = a.weight ? (a.weight + (size_t)c * a.N)[ec] : 1.f;
            const float err = wgt != 0.f ? outv[0] - (a.returns + (size_t)c * a.N)[ec] : 0.f;
            sd[0] = 2.f * a.c_value * err * a.inv_B;
            if (valid && half == 0) loss_acc += (double)err * (double)err;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            dz3[o] = a.final_tanh ? sd[o] * (1.f - outv[o] * outv[o]) : sd[o];
            bstore(r_dz3, (valid && half == 0) ? ec * 4u : 0x80000000u, (uint32_t)o * N32 * 4u, dz3[o]);
        }
        // ---- store H1, H2; dZ2 = (W3^T dz3) (1 - H2^2) in place of H2
        const uint32_t voff = valid ? (ec + 4u * (uint32_t)half * N32) * 4u : 0x80000000u;  // rows r and r + 4 share a register
        // the B operand of the two k-steps of W3^T dz3 (k = output o: lanes 0-31 supply o = 0 | 2, lanes 32-63 o = 1 | 3),
        // picked with compile-time indices: written as dz3[2 s + half] hipcc spilled the four values to scratch and
        // indexed them there (a scratch store + two dependent scratch loads on every tile's critical path)
        const float dzs0 = half ? dz3[1] : dz3[0], dzs1 = half ? dz3[3] : dz3[2];
        const __amdgpu_buffer_rsrc_t h1c = rows_rsrc(a.h1 + (size_t)c * H * a.N, H, N32);
        const __amdgpu_buffer_rsrc_t h2c = rows_rsrc(a.h2 + (size_t)c * H * a.N, H, N32);
        const __amdgpu_buffer_rsrc_t dz2c = rows_rsrc(a.dz2 + (size_t)c * H * a.N, H, N32);
        const __amdgpu_buffer_rsrc_t dz1c = rows_rsrc(a.dz1 + (size_t)c * H * a.N, H, N32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // dH2 tile = W3^T dz3: A[i = h][k = o] = W3[o][h], B[k = o][env] = dz3[o]; k-step s pairs o = 2s | 2s+1
            f32x16 g;
#pragma unroll
            for (int j = 0; j < 16; ++j) g[j] = 0.f;
            g = __builtin_amdgcn_mfma_f32_32x32x2f32(W3r[half * HP + 32 * t + col], dzs0, g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x2f32(W3r[(2 + half) * HP + 32 * t + col], dzs1, g, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int h = 32 * t + acc_row(j, half);
                const bool row_ok = 32 * t + 8 * (j >> 2) < H;  // compile time: a group of 8 rows is padding as a whole
                const uint32_t soff = (uint32_t)(32 * t + 8 * (j >> 2) + (j & 3)) * N32 * 4u;
                if (row_ok) { bstore(h1c, voff, soff, h1[t][j]); bstore(h2c, voff, soff, h2[t][j]); }
                h2[t][j] = g[j] * (1.f - h2[t][j] * h2[t][j]);
                if (row_ok) bstore(dz2c, voff, soff, h2[t][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- dH1 = W2^T dZ2 on the MFMA pipe, one output tile at a time; dZ1 = dH1 (1 - H1^2)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x16 d;
#pragma unroll
            for (int j = 0; j < 16; ++j) d[j] = 0.f;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    // A[i = h1 row][k = h2] = W2[h2][h1] = W2t[h1 * LD2 + h2]
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(W2t[(32 * t + col) * LD2 + 32 * ti + acc_row(j, half)], h2[ti][j], d, 0, 0, 0);
                    if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int h = 32 * t + acc_row(j, half);
                const float g = d[j] * (1.f - h1[t][j] * h1[t][j]);
                if (32 * t + 8 * (j >> 2) < H) bstore(dz1c, voff, (uint32_t)(32 * t + 8 * (j >> 2) + (j & 3)) * N32 * 4u, g);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (a.loss_mode != 0) {  // one atomic per wave
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

// ---- k_mlp_bwd_v3: the same pass for the reference's N = 128, T = 5 networks on the bf16 matrix cores -----------------------
// Three-way split operands as in k_mlp_v3 (qenv_ppo.hip): six v_mfma_f32_32x32x16_bf16 of 32 pipe cycles per 16 k instead of
// eight v_mfma_f32_32x32x2_f32 of 64 (which also occupy the vector lanes), fp32-accurate.  504 MFMAs per 32-sample tile
// (layer 1: 120, layer 2: 192, W2^T dZ2: 192) instead of 736; layer 3 and W3^T dz3 (4 real rows / columns) run on the vector
// lanes.  The forward half is k_mlp_v3's arithmetic instruction for instruction (pre-scaled hidden layers, same product
// order): the probability ratio of the first epoch after a rollout is exactly 1.
// W2 is needed twice: as the A operand of layer 2 (row o, eight k = input indices per lane) and of W2^T dZ2 (column i,
// eight k = output indices per lane).  Two fragment images would be 192 KiB; ONE image of 128 x 128 16-bit elements per
// bf16 term (256 bytes per row and term, 96 KiB) serves both: a row is stored with bits 2 and 3 of the column index swapped, so the
// eight elements a forward fragment needs (the k order of an accumulator tile used as the B operand: 16 s + 8 (j >> 2) +
// 4 half + (j & 3)) are one 16-byte chunk -- one ds_read_b128 -- and the transposed fragment comes from two
// ds_read_b64_tr_b16 (gfx950's transposing LDS read: 4 rows x 16 columns per 16-lane group, delivered column-major); the
// 16-byte chunks of a row are XOR-swizzled by the row index so that both kinds of read are bank-conflict free
// (cdna_hip_programming.md T10 image (b); tools/ubench/dual_image.hip checks the two address maps with exact integer data).
// Registers (two waves per SIMD, 256): the fp32 copy of H1 is not kept across layer 2 and the backward half -- it is stored
// (it has to be, for the weight gradients) and its 16 values per output tile are read back (L2 hits) beside the chain
// that needs them.
namespace bwd3 {
constexpr int W2D = mlp3::W2F * 16;                  // byte offset of the dual-use W2 image: [128 rows][3 terms][256 B]
constexpr int W2D_ROW = 768, W2D_TERM = 256;         // (a row stride of 3 x 64 banks: the bank pattern of 256-byte rows)
constexpr int LDS_BYTES = mlp3::IMAGE_UINT4 * 16;    // the layout around it is k_mlp_v3's (W1 fragments | W2 | W3T | b1 | b2 | b3)
__device__ __forceinline__ int img_off(int row, int ch) { return W2D_ROW * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ int sigma(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }   // bits 2 and 3 swapped
typedef short v4s __attribute__((ext_vector_type(4)));
}  // namespace bwd3

// sgb_interleave with LDS reads in the pattern: left to itself the scheduler lifts every ds_read of a region to its top (the 16
// float4 rows of W3 for layer 3 alone are 64 registers) and spills what it read
template <int N_MFMA, int N_VALU, int N_DS> __device__ __forceinline__ void sgb_interleave_ds() {
#pragma unroll
    for (int m = 0; m < N_MFMA; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, N_DS, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, N_VALU, 0);
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_mlp_bwd_v3(const BwdArgs a) {
    using namespace mlp3;
    using namespace bwd3;
    constexpr int WPB = BLOCK / 64, K1 = KW;
    constexpr bool ONE_WAVE = BLOCK == 256;   // one wave per SIMD, 512 registers: H1 stays in registers, the next tile's observations
                                              // are requested a whole tile ahead
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    char* ldsb = reinterpret_cast<char*>(lds4);
    // ---- the weights, once per launch and workgroup, from the torch layout
    for (int c = threadIdx.x; c < NT * KS16 * 64; c += BLOCK) {   // W1 fragments (k_mlp_v3's layout)
        const int ln = c & 63, sk = (c >> 6) % KS16, t = (c >> 6) / KS16;
        const int k0 = 16 * sk + 8 * (ln >> 5);
        const float* src = a.W1 + (size_t)(32 * t + (ln & 31)) * K1;
        float w8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = (k0 + i) < K1 ? src[k0 + i] * TANH_PRESCALE : 0.f;
        bf16x8 sp[3];
        split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) lds4[W1F + ((t * KS16 + sk) * 3 + k) * 64 + ln] = __builtin_bit_cast(uint4, sp[k]);
    }
    for (int idx = threadIdx.x; idx < H * 16; idx += BLOCK) {     // W2: chunk ch of row o = storage columns 8 ch .. 8 ch + 7
        const int o = idx >> 4, ch = idx & 15;
        float w8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w8[e] = a.W2[(size_t)o * H + sigma(8 * ch + e)] * TANH_PRESCALE;
        bf16x8 sp[3];
        split3x8(w8, sp[0], sp[1], sp[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4*>(ldsb + W2D + k * W2D_TERM + img_off(o, ch)) = __builtin_bit_cast(uint4, sp[k]);   // (img_off: 768-byte rows)
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
    __syncthreads();
    const uint4* W1f = lds4 + W1F;
    const float* b3s = reinterpret_cast<const float*>(lds4 + B3);

    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles_per_chunk = (a.N + 31) / 32, n_tiles = tiles_per_chunk * a.n_chunks;
    const int64_t tile_stride = (int64_t)gridDim.x * WPB;
    const uint32_t N32 = (uint32_t)a.N;
    double loss_acc = 0.0, std_acc = 0.0;
    // ---- the observations of a tile: element (sk, i) of a lane = row 16 sk + 8 half + i, as raw buffer loads -- the lane's
    // offset carries the sample and the half's 8 rows, the row offset 16 sk + i is scalar, rows >= 75 fall out of the range check
    float xw[KS16][8];
    auto request_x = [&](int64_t tile) {
        const int64_t c = __builtin_amdgcn_readfirstlane((int)(tile / tiles_per_chunk));
        const int64_t e = (tile % tiles_per_chunk) * 32 + col;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.obs + (size_t)c * K1 * a.N), 0,
                                                                            (int)((uint32_t)K1 * N32 * 4u), 0x00020000);
        const uint32_t voff = e < a.N ? ((uint32_t)e + 8u * (uint32_t)half * N32) * 4u : 0x80000000u;
#pragma unroll
        for (int sk = 0; sk < KS16; ++sk)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xw[sk][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (uint32_t)(16 * sk + i) * N32 * 4u, 0));
    };
    int64_t tile = (int64_t)blockIdx.x * WPB + wave;
    if (tile < n_tiles) request_x(tile);
    // The lane parts of the W2 read addresses.  The XOR swizzle mixes lane bits with the chunk number, so an address is not
    // base + literal: the forward read of chunk 2 n + half of row 32 t + col is fw_a + 16 ((2 n) ^ fw_x) + 768 * 32 t, the
    // transposed read of rows 32 to + 16 s + 8 jg + 4 half + q, columns of output tile t, is tr_a + 64 (t ^ q) + 32 (g ^ jg)
    // + 768 (32 to + 16 s + 8 jg): one or two vector instructions per 16-k step of six MFMAs, formed where they are used.
    // (Kept as per-(t, n) registers they cost 16; as loop invariants every base + literal sum is hoisted out of the tile loop
    // -- hundreds -- and spilled: the five values are made opaque at the top of every tile.)
    const int tr_q0 = (lane & 15) >> 2, tr_p = lane & 3, tr_g0 = (lane >> 4) & 1;
    int fw_a = W2D + W2D_ROW * col, fw_x = half ^ (((col & 3) << 2) | ((col >> 2) & 3));
    int tr_a = W2D + W2D_ROW * (4 * half + tr_q0) + 8 * (tr_p >> 1) + 16 * ((tr_p & 1) ^ half), tr_q = tr_q0, tr_g = tr_g0;
    for (; tile < n_tiles; tile += tile_stride) {
        asm volatile("" ::: "memory");  // keep the (tile-invariant) LDS reads inside the loop (see k_mlp)
        asm volatile("" : "+v"(fw_a), "+v"(fw_x), "+v"(tr_a), "+v"(tr_q), "+v"(tr_g));
        // the same for the reads indexed by an accumulator row (32 t + acc_row(j, half) = a literal + 4 half): W3's rows, biases
        int w3_base = W3T * 16 + 64 * half, b1_base = B1 * 16 + 16 * half, b2_base = B2 * 16 + 16 * half;
        asm volatile("" : "+v"(w3_base), "+v"(b1_base), "+v"(b2_base));
        auto w3_row = [&](int t, int j) { return *reinterpret_cast<const float4*>(ldsb + (w3_base + 16 * (32 * t + acc_row(j, 0)))); };
        const int64_t c = __builtin_amdgcn_readfirstlane((int)(tile / tiles_per_chunk));
        const int64_t e = (tile % tiles_per_chunk) * 32 + col;
        const bool valid = e < a.N;
        const uint32_t ec = (uint32_t)(valid ? e : a.N - 1);
        const uint32_t voff = valid ? (ec + 4u * (uint32_t)half * N32) * 4u : 0x80000000u;  // rows r and r + 4 share a register
        const __amdgpu_buffer_rsrc_t h1c = rows_rsrc(a.h1 + (size_t)c * H * a.N, H, N32);
        const __amdgpu_buffer_rsrc_t h2c = rows_rsrc(a.h2 + (size_t)c * H * a.N, H, N32);
        const __amdgpu_buffer_rsrc_t dz2c = rows_rsrc(a.dz2 + (size_t)c * H * a.N, H, N32);
        const __amdgpu_buffer_rsrc_t dz1c = rows_rsrc(a.dz1 + (size_t)c * H * a.N, H, N32);
        auto row_soff = [&](int t, int j) { return (uint32_t)(32 * t + 8 * (j >> 2) + (j & 3)) * N32 * 4u; };
        bf16x8 xsp[KS16][3];
#pragma unroll
        for (int sk = 0; sk < KS16; ++sk) split3x8(xw[sk], xsp[sk][0], xsp[sk][1], xsp[sk][2]);
        if (ONE_WAVE && tile + tile_stride < n_tiles) request_x(tile + tile_stride);
        // ---- forward, layers 1 and 2 as k_mlp_v3's software pipeline (a chain beside the vector work of the previous tile)
        f32x16 acc1[NT];
        bf16x8 bsp[NT][2][3];   // H1, then dZ2, as bf16 terms: the B operand of layer 2, then of W2^T dZ2
        auto l1_chain = [&](int t) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc1[t][j] = *reinterpret_cast<const float*>(ldsb + (b1_base + 4 * (32 * t + acc_row(j, 0))));
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
        auto l1_finish = [&](int t) {   // tanh, H1 out, and its bf16 terms
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc1[t][j] = fast_tanh_prescaled(acc1[t][j]);
                if (!ONE_WAVE) bstore(h1c, voff, row_soff(t, j), acc1[t][j]);   // (ONE_WAVE: two per k-step of layer 2, below)
            }
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = acc1[t][8 * jh + i];
                split3x8(v8, bsp[t][jh][0], bsp[t][jh][1], bsp[t][jh][2]);
            }
        };
        // the six products of one 16-k step, smallest first (k_mlp_v3's order)
        auto six = [&](f32x16& acc, bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        };
        auto l2_chain = [&](int t, f32x16& acc2, int ti0 = 0, int ti1 = NT) {   // input tiles [ti0, ti1) of output tile t
            if (ti0 == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc2[j] = *reinterpret_cast<const float*>(ldsb + (b2_base + 4 * (32 * t + acc_row(j, 0))));
            }
#pragma unroll
            for (int ti = ti0; ti < ti1; ++ti)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // row 32 t + col, chunk 2 (2 ti + s) + half: the lane part comes from fw_base, the rest is one literal
                    const char* f = ldsb + (fw_a + 16 * ((2 * (2 * ti + s)) ^ fw_x) + W2D_ROW * 32 * t);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(f)),
                                 am = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(f + W2D_TERM)),
                                 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(f + 2 * W2D_TERM));
                    six(acc2, ah, am, al, bsp[ti][s][0], bsp[ti][s][1], bsp[ti][s][2]);
                }
        };
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = half ? 0.f : b3s[k];
        auto layer3 = [&](int t, const f32x16& hv) {   // mlp3_layer3 with the opaque row base
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 w = w3_row(t, j);
                o[0] = fmaf(w.x, hv[j], o[0]); o[1] = fmaf(w.y, hv[j], o[1]);
                o[2] = fmaf(w.z, hv[j], o[2]); o[3] = fmaf(w.w, hv[j], o[3]);
            }
        };
        f32x16 h2[NT];
        __builtin_amdgcn_sched_barrier(0);
        l1_chain(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 1; t < NT; ++t) {
            l1_chain(t);
            l1_finish(t - 1);
            sgb_interleave_ds<30, 6, 1>();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ONE_WAVE) {
            // Where H1's row stores go.  A CU's vector-memory path passes ~10-13 B per clock in either direction and its four waves
            // share it: a wave that issues stores back to back spends ~105 cycles on each (256 bytes), whatever their width, cache
            // policy or order (measured: the same rows as dwordx4 stores, nt / sc1, XCD-aware tile order -- no change), and with
            // one wave per SIMD nothing else runs meanwhile; hipcc bunches the stores of a scheduling region whatever the
            // sched_group_barrier pattern asks for.  So layer 2 is cut into one region per 16-k step (6 MFMAs): each carries two of the
            // previous output tile's tanh + layer-3 rows and TWO H1 row stores -- a store every ~100 cycles, beside matrix work.
            auto l2_kstep = [&](int t, f32x16& acc2, int r) {   // k-step r = (input tile r >> 1, half-tile r & 1) of output tile t
                const int ti = r >> 1, s2 = r & 1;
                if (r == 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc2[j] = *reinterpret_cast<const float*>(ldsb + (b2_base + 4 * (32 * t + acc_row(j, 0))));
                }
                const char* f = ldsb + (fw_a + 16 * ((2 * (2 * ti + s2)) ^ fw_x) + W2D_ROW * 32 * t);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(f)),
                             am = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(f + W2D_TERM)),
                             al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(f + 2 * W2D_TERM));
                six(acc2, ah, am, al, bsp[ti][s2][0], bsp[ti][s2][1], bsp[ti][s2][2]);
            };
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    l2_kstep(t, h2[t], r);
                    if (t == 0) {   // beside output tile 0: tanh + split of layer 1's last tile (its terms are needed from k-step 6 on)
                        if (r < 4) {
#pragma unroll
                            for (int j = 4 * r; j < 4 * r + 4; ++j) acc1[NT - 1][j] = fast_tanh_prescaled(acc1[NT - 1][j]);
                        }
                        if (r == 2 || r == 4) {
                            const int jh = r == 2 ? 0 : 1;
                            float v8[8];
#pragma unroll
                            for (int i2 = 0; i2 < 8; ++i2) v8[i2] = acc1[NT - 1][8 * jh + i2];
                            split3x8(v8, bsp[NT - 1][jh][0], bsp[NT - 1][jh][1], bsp[NT - 1][jh][2]);
                        }
                    } else {        // beside output tile t: tanh + layer 3 of rows 2 r, 2 r + 1 of tile t - 1
#pragma unroll
                        for (int j = 2 * r; j < 2 * r + 2; ++j) {
                            h2[t - 1][j] = fast_tanh_prescaled(h2[t - 1][j]);
                            const float4 w = w3_row(t - 1, j);
                            o[0] = fmaf(w.x, h2[t - 1][j], o[0]); o[1] = fmaf(w.y, h2[t - 1][j], o[1]);
                            o[2] = fmaf(w.z, h2[t - 1][j], o[2]); o[3] = fmaf(w.w, h2[t - 1][j], o[3]);
                        }
                    }
                    bstore(h1c, voff, row_soff(t, 2 * r), acc1[t][2 * r]);
                    bstore(h1c, voff, row_soff(t, 2 * r + 1), acc1[t][2 * r + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
        l2_chain(0, h2[0], 0, NT - 1);
        l1_finish(NT - 1);
        sgb_interleave_ds<36, 5, 1>();
        __builtin_amdgcn_sched_barrier(0);
        l2_chain(0, h2[0], NT - 1, NT);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 1; t < NT; ++t) {   // chain t beside tanh + layer 3 of t - 1
            l2_chain(t, h2[t]);
#pragma unroll
            for (int j = 0; j < 16; ++j) h2[t - 1][j] = fast_tanh_prescaled(h2[t - 1][j]);
            layer3(t - 1, h2[t - 1]);
            sgb_interleave_ds<48, 3, 1>();
            __builtin_amdgcn_sched_barrier(0);
        }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) h2[NT - 1][j] = fast_tanh_prescaled(h2[NT - 1][j]);
        layer3(NT - 1, h2[NT - 1]);
        // the next tile's observations are requested HERE: vector-memory operations retire in issue order, so a load waits for
        // every store issued before it -- behind the 256 row stores of the backward half it cost 3 ms of a 9.5 ms launch
        // (ablation builds during development); here only H1's 64 stores, issued a layer ago, are ahead of it.  Its 40 registers
        // fit because the backward product below walks the k-steps outermost (dZ2 stays fp32 and is split one step at a time).
        if (!ONE_WAVE && tile + tile_stride < n_tiles) request_x(tile + tile_stride);
        // ---- output and its pre-activation gradient, in both lane halves
        const __amdgpu_buffer_rsrc_t r_dz3 = rows_rsrc(a.dz3 + (size_t)c * 4 * a.N, 4, N32);
        float dz3[4], outv[4], sd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = o[k] + __shfl_xor(o[k], 32);
            if (a.final_tanh) v = tanhf(v);
            outv[k] = v;
            sd[k] = 0.f;
        }
        if (a.loss_mode == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < a.n_out) sd[k] = (a.seed + (size_t)c * a.n_out * a.N)[(uint32_t)k * N32 + ec];
        } else if (a.loss_mode == 1) {  // k_ppo_loss, actor part, same operation order (see k_mlp_bwd)
            const float* ac = a.actions + (size_t)c * 4 * a.N;
            const float* lo = a.old_logprob + (size_t)c * 4 * a.N;
            float lp = 0.f, lp_old = 0.f, d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[k] = ac[(uint32_t)k * N32 + ec] - outv[k];
                lp += -0.5f * d[k] * d[k] * a.inv_var + a.log_norm;
                lp_old += lo[(uint32_t)k * N32 + ec];
            }
            const float ratio = (float)exp((double)(lp - lp_old));
            const float A = (a.adv + (size_t)c * a.N)[ec];
            const float clipped = fminf(fmaxf(ratio, 1.f - a.eps_clip), 1.f + a.eps_clip);
            const float s1 = ratio * A, s2 = clipped * A;
            const bool use1 = s1 <= s2;  // torch.min picks surr1 on ties
            const float wgt = a.weight ? (a.weight + (size_t)c * a.N)[ec] : 1.f;
            const bool live = (use1 || (ratio > 1.f - a.eps_clip && ratio < 1.f + a.eps_clip)) && wgt != 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sd[k] = live ? -A * ratio * d[k] * a.inv_var * a.inv_B : 0.f;
            if (valid && half == 0 && wgt != 0.f) {
                loss_acc += -(double)(use1 ? s1 : s2);
                float dls = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) dls += d[k] * d[k] * a.inv_var - 1.f;
                if (live) std_acc += (double)(-A * ratio) * (double)(dls * a.inv_std);
            }
        } else {  // critic: c_value * (V - ret)^2
            const float wgt = a.weight ? (a.weight + (size_t)c * a.N)[ec] : 1.f;
            const float err = wgt != 0.f ? outv[0] - (a.returns + (size_t)c * a.N)[ec] : 0.f;
            sd[0] = 2.f * a.c_value * err * a.inv_B;
            if (valid && half == 0) loss_acc += (double)err * (double)err;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dz3[k] = a.final_tanh ? sd[k] * (1.f - outv[k] * outv[k]) : sd[k];
            bstore(r_dz3, (valid && half == 0) ? ec * 4u : 0x80000000u, (uint32_t)k * N32 * 4u, dz3[k]);
        }
        // ---- backward.  dZ2 = (W3^T dz3) (1 - H2^2) on the vector lanes (W3 as [h][4] rows: one broadcast read per row; read
        // AGAIN through a base the optimiser cannot equate with layer 3's -- kept from there the rows are 256 registers' worth
        // of live values across the loss block); dH1 = W2^T dZ2 on the matrix cores, the A fragments by transposing reads of
        // the same W2 image.  The 16-k steps are the OUTER loop: step ks feeds the eight dZ2 values of rows 32 to + 16 s + ...
        // to the four output tiles' accumulators (four independent chains), and the vector work of step ks + 1 -- its eight
        // dZ2 values, the H2 / dZ2 row stores, the split into bf16 terms -- sits beside those 24 MFMAs, the stores a few per
        // output tile (ONE_WAVE: see the regions below).  dZ1 = dH1 (1 - H1^2) / (2 log2 e): the image holds the pre-scaled W2.
        int w3_base2 = w3_base;
        asm volatile("" : "+v"(w3_base2));
        bf16x8 bt[2][3];   // the B operand of step ks: bt[ks & 1]
        auto dz2_step = [&](int ks) {   // rows of k-step ks: registers 8 s .. 8 s + 7 of tile to
            const int to = ks >> 1, sj = ks & 1;
            float v8[8];
#pragma unroll
            for (int i2 = 0; i2 < 8; ++i2) {
                const int j2 = 8 * sj + i2;
                const float4 w = *reinterpret_cast<const float4*>(ldsb + (w3_base2 + 16 * (32 * to + acc_row(j2, 0))));
                const float g = fmaf(w.w, dz3[3], fmaf(w.z, dz3[2], fmaf(w.y, dz3[1], w.x * dz3[0])));
                const float hv = h2[to][j2];
                bstore(h2c, voff, row_soff(to, j2), hv);
                v8[i2] = g * (1.f - hv * hv);
                bstore(dz2c, voff, row_soff(to, j2), v8[i2]);
            }
            split3x8(v8, bt[ks & 1][0], bt[ks & 1][1], bt[ks & 1][2]);
        };
        f32x16 dd[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j2 = 0; j2 < 16; ++j2) dd[t][j2] = 0.f;
        dz2_step(0);
        // swizzled chunk of a transposed read = (4 t + 2 g + (p & 1)) ^ (4 q + 2 jg + half)   (row & 3 = q, (row >> 2) & 3 = 2 jg + half)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks) {
            const int to = ks >> 1, sj = ks & 1;
            // one scheduling region per output tile (6 MFMAs): the first computes step ks + 1's eight dZ2 values beside its MFMAs,
            // the other three carry that step's 16 row stores (H2, dZ2) a few at a time -- bunched behind the 24 MFMAs they block
            // the wave for ~1.7 k cycles with nothing in the matrix pipe (see layer 2 above)
            const int ksn = ks + 1, ton = ksn >> 1, sjn = ksn & 1;
            float v8n[8], hvn[8];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tb0 = tr_a + 64 * (t ^ tr_q) + 32 * tr_g, tb1 = tr_a + 64 * (t ^ tr_q) + 32 * (tr_g ^ 1) + W2D_ROW * 8;
                const char* f0 = ldsb + (tb0 + W2D_ROW * (32 * to + 16 * sj));
                const char* f1 = ldsb + (tb1 + W2D_ROW * (32 * to + 16 * sj));
                bf16x8 af[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const v4s lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(f0 + m * W2D_TERM));
                    const v4s hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(f1 + m * W2D_TERM));
                    const short f8 __attribute__((ext_vector_type(8))) = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    af[m] = __builtin_bit_cast(bf16x8, f8);
                }
                six(dd[t], af[0], af[1], af[2], bt[ks & 1][0], bt[ks & 1][1], bt[ks & 1][2]);
                if (ONE_WAVE && ksn < 2 * NT) {
                    if (t == 0) {
#pragma unroll
                        for (int i2 = 0; i2 < 8; ++i2) {
                            const int j2 = 8 * sjn + i2;
                            const float4 w = *reinterpret_cast<const float4*>(ldsb + (w3_base2 + 16 * (32 * ton + acc_row(j2, 0))));
                            const float g = fmaf(w.w, dz3[3], fmaf(w.z, dz3[2], fmaf(w.y, dz3[1], w.x * dz3[0])));
                            hvn[i2] = h2[ton][j2];
                            v8n[i2] = g * (1.f - hvn[i2] * hvn[i2]);
                        }
                    } else {
                        const int lo = t == 1 ? 0 : (t == 2 ? 3 : 6), hi = t == 1 ? 3 : (t == 2 ? 6 : 8);
#pragma unroll
                        for (int i2 = lo; i2 < hi; ++i2) {
                            bstore(h2c, voff, row_soff(ton, 8 * sjn + i2), hvn[i2]);
                            bstore(dz2c, voff, row_soff(ton, 8 * sjn + i2), v8n[i2]);
                        }
                        if (t == NT - 1) split3x8(v8n, bt[ksn & 1][0], bt[ksn & 1][1], bt[ksn & 1][2]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!ONE_WAVE) {
                if (ks + 1 < 2 * NT) dz2_step(ks + 1);
                sgb_interleave_ds<24, 5, 2>();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // H1 as this wave stored it (read at L2: glc), a tile ahead of the tile being finished
        constexpr float INV_C = 1.0f / TANH_PRESCALE;
        float h1r[2][16];
        auto h1_request = [&](int t) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                h1r[t & 1][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h1c, voff, row_soff(t, j), 1));
        };
        if (!ONE_WAVE) h1_request(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!ONE_WAVE && t + 1 < NT) h1_request(t + 1);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float hv = ONE_WAVE ? acc1[t][j] : h1r[t & 1][j];
                bstore(dz1c, voff, row_soff(t, j), (dd[t][j] * INV_C) * (1.f - hv * hv));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (a.loss_mode != 0) {  // one atomic per wave
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

// ---- weight gradients: C[M][Nn] += sum over samples A[m][s] * B[n][s]  (+ row sums of A = bias gradients)
struct WgradArgs {
    int64_t N;
    int32_t n_chunks, M, Nn;
    const float *A, *B;  // chunked [c][M][N], [c][Nn][N]
    float* C;            // [M][Nn] row-major, accumulated with atomics
    float* rowsum;       // [M] or null
    float* partial;      // null, or [blocks][M*Nn + M]: every workgroup stores its sums there (deterministic mode) ...
};                       // ... and k_sum_slots adds the slots in a fixed order

// One workgroup walks a strided set of 64-sample tiles: both operand tiles are staged in LDS ([row][65]: the
// 64-sample rows are written coalesced, the MFMA operands -- 32 rows at a fixed sample -- read stride 65,
// conflict free); each of the NW waves owns up to ceil(MT*NT/NW) of the 32x32 output tiles in registers for the
// whole walk and adds them to C once at the end.  Two workgroups per CU (NW/2 waves per SIMD): a wave's staging,
// barrier and LDS waits are covered by the MFMAs of the other (one workgroup of 4 waves per CU: matrix pipe 44 %
// busy; two: 60 %).
// BF16: the products on the bf16 matrix cores with three-way split operands (k_mlp_v3's arithmetic): the sample index is the k of
// this product and runs along the LDS rows, so a lane's eight consecutive samples are an MFMA fragment -- two ds_read_b128 from
// rows padded to 68 words (16-byte aligned, conflict-free for the 16-lane groups of a 128-bit read), split in registers, six
// v_mfma_f32_32x32x16_bf16 of 32 cycles per 16 samples and output tile instead of eight v_mfma_f32_32x32x2_f32 of 64.  The
// global -> LDS staging is unchanged (fully coalesced row loads; fragments loaded straight from global memory -- 16 bytes per
// lane at a 128-byte stride -- thrashed the vector L1: 25.7 ms per epoch against 24.1).
template <int MT, int NTN, int NW, bool BF16 = false>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_wgrad(const WgradArgs a) {  // NW/2 waves per SIMD = two workgroups per CU
    constexpr int LD = BF16 ? 68 : 65, ROWS = (MT + NTN) * 32, TPW = (MT * NTN + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                 // [MT*32][65]
    float* Bs = As + MT * 32 * LD;   // [NTN*32][65]
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    // Tile ownership from the wave index in an SGPR: as a VGPR value every MFMA of the k loop sat in its own exec-masked
    // basic block with its LDS operand read issued right in front of it (s_waitcnt lgkmcnt(0)).  Worth 0 / 5 / 4 % on
    // dW2 / dW1 / dW3 (the matrix loop is not what limits the kernel); the prefetch / staging addressing keeps the VGPR
    // copy -- with the scalar one its 64 per-row predicates become scalar branches and the kernel is 15 % SLOWER (A/B).
    const int wave = (int)(threadIdx.x >> 6);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    constexpr bool FULL = (MT * NTN) % NW == 0;  // every wave owns exactly TPW tiles
    const int64_t tiles_per_chunk = (a.N + 63) / 64, n_tiles = tiles_per_chunk * a.n_chunks;
    f32x16 acc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;
    // Every thread carries its share of the NEXT 64-sample tile in registers: the global loads are issued right after
    // the current tile has been copied to LDS and land while its MFMAs run.  Element i of a thread is row wave + NW i
    // of the combined [A rows | B rows] image at sample `lane` (the A / B decision is wave-uniform).
    constexpr int PER = (ROWS + NW - 1) / NW, PER_A = (MT * 32 + NW - 1) / NW;
    float pre[PER];
    // bias gradients = row sums of A: accumulated from `pre` in registers (no LDS reads, the waves stay balanced);
    // the 64 samples of a row are folded across the wave once, after the walk
    float rsum[PER_A];
#pragma unroll
    for (int i = 0; i < PER_A; ++i) rsum[i] = 0.f;
    // Range-checked raw buffer loads: one per-lane offset register (the sample; 2^31 for a lane beyond the chunk), the
    // row offset in soffset (scalar), rows >= M / Nn and parked lanes come back as 0 from the hardware's range check --
    // no predicate, no branch and no address VGPR per load (the predicated global loads were 64 exec-masked blocks).
    auto fetch = [&](int64_t tile) {
        const int64_t c = tile / tiles_per_chunk, e0 = (tile % tiles_per_chunk) * 64;  // block-uniform
        uint32_t n32 = (uint32_t)a.N;
        asm volatile("" : "+s"(n32));  // opaque per tile: keeps LICM from carrying per-row values across the walk
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + ((size_t)c * a.M) * a.N), 0,
                                                                            (int)((uint32_t)a.M * n32 * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + ((size_t)c * a.Nn) * a.N), 0,
                                                                            (int)((uint32_t)a.Nn * n32 * 4u), 0x00020000);
        const uint32_t voff = (e0 + lane < a.N) ? (uint32_t)(e0 + lane) * 4u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = wave_s + NW * i;  // scalar
            if (NW * i + NW - 1 < MT * 32)  // compile time: every wave's element i is an A row
                pre[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, voff, (uint32_t)r * n32 * 4u, 0));
            else if (NW * i >= MT * 32)     // ... a B row
                pre[i] = r < ROWS ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, voff, (uint32_t)(r - MT * 32) * n32 * 4u, 0)) : 0.f;
            else if (r < MT * 32)
                pre[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, voff, (uint32_t)r * n32 * 4u, 0));
            else
                pre[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, voff, (uint32_t)(r - MT * 32) * n32 * 4u, 0));
        }
    };
    // the tiles a wave owns are consecutive in row-major order when they then share a row of output tiles (one A
    // operand read per k-step for all of them), else in column-major order when that makes them share a column (B)
    constexpr bool SAME_A = (NTN % TPW) == 0;
    constexpr bool SAME_B = !SAME_A && (MT % TPW) == 0;
    const int tid0 = wave_s * TPW;
    // BF16, 4 x 4 output tiles: a wave owns a 2 x 2 block (two A and two B fragments per 16 samples feed four tiles: four operand
    // splits per 24 MFMAs instead of the five of a row of tiles -- the bf16 variant is bound by the vector work of the splits)
    constexpr bool BLOCK22 = BF16 && MT == 4 && NTN == 4 && NW == 4;
    auto tile_of = [&](int tid, int& tm, int& tn) {
        if (BLOCK22) { const int w = tid / TPW, q = tid % TPW; tm = 2 * (w >> 1) + (q >> 1); tn = 2 * (w & 1) + (q & 1); }
        else if (SAME_B) { tm = tid % MT; tn = tid / MT; }
        else { tm = tid / NTN; tn = tid % NTN; }
    };
    if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int r = wave + NW * i;
            if (r < ROWS) lds[r * LD + lane] = pre[i];  // As and Bs are contiguous: row r of the combined [ROWS][65] image
        }
        __syncthreads();
        if (a.rowsum) {
#pragma unroll
            for (int i = 0; i < PER_A; ++i) rsum[i] += (wave + NW * i < MT * 32) ? pre[i] : 0.f;
        }
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
        int tm0, tn0;
        tile_of(tid0, tm0, tn0);
        if constexpr (BF16) {
            auto frag = [&](const float* rows, int tile32, int s, bf16x8 (&sp)[3]) {   // samples 16 s + 8 half + 0..7 of row 32 tile32 + col
                const float4* p4 = reinterpret_cast<const float4*>(rows + (32 * tile32 + col) * LD + 16 * s + 8 * half);
                const float4 lo = p4[0], hi = p4[1];
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                split3x8(v, sp[0], sp[1], sp[2]);
            };
            if constexpr (BLOCK22) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 a2[2][3], b2[2][3];
                    frag(As, tm0, s, a2[0]); frag(As, tm0 + 1, s, a2[1]);
                    frag(Bs, tn0, s, b2[0]); frag(Bs, tn0 + 1, s, b2[1]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bf16x8 (&a3)[3] = a2[q >> 1];
                        const bf16x8 (&b3)[3] = b2[q & 1];
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[1], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2], b3[0], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[2], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[0], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[1], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0], acc[q], 0, 0, 0);
                    }
                }
            } else
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a3[3], b3[3];
                if (SAME_A) frag(As, tm0, s, a3);
                if (SAME_B) frag(Bs, tn0, s, b3);
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    const int tid = tid0 + q;
                    if (FULL || tid < MT * NTN) {
                        int tm, tn;
                        tile_of(tid, tm, tn);
                        if (!SAME_A) frag(As, tm, s, a3);
                        if (!SAME_B) frag(Bs, tn, s, b3);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[1], acc[q], 0, 0, 0);   // smallest products first
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[2], b3[0], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[2], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], b3[0], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[1], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], b3[0], acc[q], 0, 0, 0);
                    }
                }
            }
        } else
#pragma unroll 4
        for (int ks = 0; ks < 32; ++ks) {
            const int sidx = 2 * ks + half;
            float a0 = 0.f, b0 = 0.f;
            if (SAME_A) a0 = As[(32 * tm0 + col) * LD + sidx];
            if (SAME_B) b0 = Bs[(32 * tn0 + col) * LD + sidx];
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int tid = tid0 + q;
                if (FULL || tid < MT * NTN) {
                    int tm, tn;
                    tile_of(tid, tm, tn);
                    const float av = SAME_A ? a0 : As[(32 * tm + col) * LD + sidx];
                    const float bv = SAME_B ? b0 : Bs[(32 * tn + col) * LD + sidx];
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q], 0, 0, 0);
                }
            }
        }
    }
    const size_t slot = (size_t)a.M * a.Nn + a.M;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int tid = wave_s * TPW + q;
        if (FULL || tid < MT * NTN) {
            int tm, tn;
            tile_of(tid, tm, tn);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = 32 * tm + acc_row(j, half), n = 32 * tn + col;
                if (m < a.M && n < a.Nn) {
                    if (a.partial) a.partial[(size_t)blockIdx.x * slot + (size_t)m * a.Nn + n] = acc[q][j];
                    else atomicAdd(&a.C[(size_t)m * a.Nn + n], acc[q][j]);
                }
            }
        }
    }
    if (a.rowsum) {  // lane i ends up with the total of row wave + NW i: ONE atomic instruction per wave (a single-lane
        float mine = 0.f;  // atomic per row is 32 requests per wave on 128 hot addresses: +50 us on a 25 us launch)
#pragma unroll
        for (int i = 0; i < PER_A; ++i) {
            float v = rsum[i];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
            if (lane == i) mine = v;
        }
        const int r = wave + NW * lane;
        if (lane < PER_A && r < MT * 32 && r < a.M) {
            if (a.partial) a.partial[(size_t)blockIdx.x * slot + (size_t)a.M * a.Nn + r] = mine;
            else atomicAdd(&a.rowsum[r], mine);
        }
    }
}

// deterministic mode, second pass: element j of C (then of rowsum) = sum over the workgroups' slots in slot order
__global__ __launch_bounds__(256) void k_sum_wgrad_slots(const float* partial, int n_slots, int64_t mn, int64_t m, float* C, float* rowsum) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= mn + m) return;
    float acc = 0.f;
    for (int b = 0; b < n_slots; ++b) acc += partial[(size_t)b * (mn + m) + j];
    if (j < mn) C[j] += acc;
    else if (rowsum) rowsum[j - mn] += acc;
}

template <int MT, int NTN> static int launch_wgrad(const WgradArgs& a, hipStream_t stream) {
    // NW = 8 / 6 (4 / 3 waves per SIMD with two workgroups per CU) was measured and is NOT used: under the 128 / 168 VGPR cap
    // hipcc shuttles the accumulator tiles between VGPRs and AGPRs around every MFMA (dW2 3.0 -> 4.9 ms, dW1 2.4 -> 5.8 ms)
    constexpr int NW = 4;
    const int64_t n_tiles = ((a.N + 63) / 64) * a.n_chunks;
    const int64_t blocks = n_tiles < 512 ? n_tiles : 512;
    if constexpr (MT == 4 && NTN == 4) {   // dW2 of the N = 128 networks: 2.78 -> 2.53 ms at 8.4 M samples (dW1, 128 x 75: 2.07 -> 2.18, not
        // used: the staging, not the matrix loop, limits these kernels); QENV_WGRAD_BF16=0 keeps the fp32-MFMA kernel (A/B runs)
        static const int bf16 = getenv("QENV_WGRAD_BF16") ? atoi(getenv("QENV_WGRAD_BF16")) : 1;
        if (bf16 && n_tiles >= 512) {
            const size_t lds3 = sizeof(float) * (size_t)(MT + NTN) * 32 * 68;
            hipError_t e3 = allow_dynamic_lds((const void*)k_wgrad<MT, NTN, NW, true>, lds3);
            if (e3 != hipSuccess) return fail_hip(e3, "hipFuncSetAttribute(k_wgrad bf16)");
            hipLaunchKernelGGL((k_wgrad<MT, NTN, NW, true>), dim3((unsigned)blocks), dim3(NW * 64), lds3, stream, a);
            int rc3 = check_launch("k_wgrad(bf16)");
            if (rc3 == 0 && a.partial) {
                const int64_t mn = (int64_t)a.M * a.Nn, tot = mn + a.M;
                hipLaunchKernelGGL(k_sum_wgrad_slots, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, a.partial, (int)blocks, mn,
                                   (int64_t)a.M, a.C, a.rowsum);
                rc3 = check_launch("k_sum_wgrad_slots");
            }
            return rc3;
        }
    }
    const size_t lds = sizeof(float) * (size_t)(MT + NTN) * 32 * 65;
    hipError_t e = allow_dynamic_lds((const void*)k_wgrad<MT, NTN, NW>, lds);
    if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_wgrad)");
    hipLaunchKernelGGL((k_wgrad<MT, NTN, NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, stream, a);
    int rc = check_launch("k_wgrad");
    if (rc == 0 && a.partial) {
        const int64_t mn = (int64_t)a.M * a.Nn, tot = mn + a.M;
        hipLaunchKernelGGL(k_sum_wgrad_slots, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, a.partial, (int)blocks, mn,
                           (int64_t)a.M, a.C, a.rowsum);
        rc = check_launch("k_sum_wgrad_slots");
    }
    return rc;
}

}  // namespace qenv
#include "qenv_grad5.hpp"
namespace qenv {

// torch.optim.Adam (ppo.py:105: lr 5e-4, betas (0.9, 0.999), eps 1e-8, no weight decay)
__global__ __launch_bounds__(256) void k_adam(int64_t n, float* p, const float* g, float* m, float* v, float lr, float b1,
                                              float b2, float eps, float bc1, float bc2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    }
}

// The same update with the step count in device memory (hipGraph replays must see a new count each time): the counter
// is advanced by its own one-thread launch in front of this kernel, and 1 - beta^step is evaluated here in double, as
// torch does on the host.
__global__ void k_count(int32_t* c) { *c += 1; }
__global__ __launch_bounds__(256) void k_adam_counted(int64_t n, float* p, const float* g, float* m, float* v, float lr, float b1,
                                                      float b2, float eps, double beta1, double beta2, const int32_t* step) {
    const double st = (double)*step;
    const float bc1 = (float)(1.0 - pow(beta1, st)), bc2 = (float)(1.0 - pow(beta2, st));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    }
}

}  // namespace qenv

using namespace qenv;

extern "C" {

static int launch_bwd(BwdArgs& a, int hidden, void* stream);

int qenv_mlp_backward(int64_t n_envs, int n_chunks, int k_in, int hidden, int n_out, int final_tanh, const float* obs,
                      const float* seed, const float* W1, const float* b1, const float* W2, const float* b2,
                      const float* W3, const float* b3, float* h1, float* h2, float* dz2, float* dz1, float* dz3,
                      void* stream) {
    if (!obs || !seed || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !h1 || !h2 || !dz2 || !dz1 || !dz3)
        return fail(QENV_ERR_ARG, "qenv_mlp_backward: null argument");
    if (n_envs <= 0 || n_chunks <= 0 || k_in <= 0 || n_out < 1 || n_out > 4) return fail(QENV_ERR_ARG, "qenv_mlp_backward: bad sizes");
    BwdArgs a;
    a.N = n_envs; a.n_chunks = n_chunks; a.K = k_in; a.n_out = n_out; a.final_tanh = final_tanh;
    a.obs = obs; a.seed = seed; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
    a.h1 = h1; a.h2 = h2; a.dz2 = dz2; a.dz1 = dz1; a.dz3 = dz3;
    a.loss_mode = 0; a.actions = a.old_logprob = a.adv = a.returns = nullptr; a.sums = nullptr;
    a.inv_var = a.inv_std = a.log_norm = a.eps_clip = a.c_value = a.inv_B = 0.f; a.weight = nullptr;
    return launch_bwd(a, hidden, stream);
}

int qenv_mlp_backward_ppo(int64_t n_envs, int n_chunks, int k_in, int hidden, int head, const float* obs,
                          const float* actions, const float* old_logprob, const float* adv, const float* returns,
                          double std, double eps_clip, double c_value, int64_t batch, double* sums, const float* W1,
                          const float* b1, const float* W2, const float* b2, const float* W3, const float* b3, float* h1,
                          float* h2, float* dz2, float* dz1, float* dz3, void* stream) {
    return qenv_mlp_backward_ppo_masked(n_envs, n_chunks, k_in, hidden, head, obs, actions, old_logprob, adv, returns, nullptr, std,
                                        eps_clip, c_value, batch, sums, W1, b1, W2, b2, W3, b3, h1, h2, dz2, dz1, dz3, stream);
}

int qenv_mlp_backward_ppo_masked(int64_t n_envs, int n_chunks, int k_in, int hidden, int head, const float* obs,
                                 const float* actions, const float* old_logprob, const float* adv, const float* returns,
                                 const float* weight, double std, double eps_clip, double c_value, int64_t batch, double* sums,
                                 const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                 const float* b3, float* h1, float* h2, float* dz2, float* dz1, float* dz3, void* stream) {
    if (!obs || !sums || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !h1 || !h2 || !dz2 || !dz1 || !dz3)
        return fail(QENV_ERR_ARG, "qenv_mlp_backward_ppo: null argument");
    if (head != 0 && head != 1) return fail(QENV_ERR_ARG, "qenv_mlp_backward_ppo: head must be 0 (actor) or 1 (critic)");
    if (head == 0 && (!actions || !old_logprob || !adv)) return fail(QENV_ERR_ARG, "qenv_mlp_backward_ppo: the actor needs actions, old_logprob, adv");
    if (head == 1 && !returns) return fail(QENV_ERR_ARG, "qenv_mlp_backward_ppo: the critic needs returns");
    if (n_envs <= 0 || n_chunks <= 0 || k_in <= 0 || batch <= 0) return fail(QENV_ERR_ARG, "qenv_mlp_backward_ppo: bad sizes");
    if (!(std > 0)) return fail(QENV_ERR_ARG, "qenv_mlp_backward_ppo: std must be positive");
    BwdArgs a;
    a.N = n_envs; a.n_chunks = n_chunks; a.K = k_in; a.n_out = head == 0 ? 4 : 1; a.final_tanh = head == 0;
    a.obs = obs; a.seed = nullptr; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
    a.h1 = h1; a.h2 = h2; a.dz2 = dz2; a.dz1 = dz1; a.dz3 = dz3;
    a.loss_mode = head == 0 ? 1 : 2; a.actions = actions; a.old_logprob = old_logprob; a.adv = adv; a.returns = returns;
    a.inv_var = (float)(1.0 / (std * std)); a.inv_std = (float)(1.0 / std); a.log_norm = (float)(-std::log(std) - 0.9189385332046727);
    a.eps_clip = (float)eps_clip; a.c_value = (float)c_value; a.inv_B = 1.f / (float)batch;
    a.sums = sums; a.weight = weight;
    return launch_bwd(a, hidden, stream);
}

static int launch_bwd(BwdArgs& a, int hidden, void* stream) {
    const int64_t n_envs = a.N;
    const int n_chunks = a.n_chunks, k_in = a.K;
    if ((int64_t)(hidden > k_in ? hidden : k_in) * n_envs >= ((int64_t)1 << 29))
        return fail(QENV_ERR_ARG, "qenv_mlp_backward: rows*n_envs per chunk must stay below 2^29 (32-bit byte offsets, range-checked stores)");
    const int HP = hidden < 32 ? 32 : hidden, K1P = (k_in + 1) & ~1;
    const size_t lds = sizeof(float) * ((size_t)K1P * HP + (size_t)HP * (HP + 1) + 32 * HP + 4 * HP + 2 * HP + 32);
    if (lds > 160 * 1024) return fail(QENV_ERR_ARG, "qenv_mlp_backward: weights do not fit the 160 KiB LDS");
    const int64_t n_tiles = ((n_envs + 31) / 32) * n_chunks;
    // two waves per SIMD (512 threads) overlap one wave's tanh / stores / LDS waits with the other's MFMAs once every
    // wave has several tiles; with at most one tile per wave they only share the pipe (a 256-env x 32-step update:
    // 60 us with 256 threads, 84 us with 512), so small batches keep one wave per SIMD on twice as many CUs
    const bool wide = n_tiles > 1024;
    const int wpb = wide ? 8 : 4;
    int64_t blocks = (n_tiles + wpb - 1) / wpb;
    if (blocks > 256) blocks = 256;
    hipError_t e;
    // QENV_BWD_BF16: 1 (default) = k_mlp_bwd_v3 for the N = 128, T = 5 networks, one wave per SIMD; 2 = the same kernel with two
    // waves per SIMD (256 registers: it spills, 36 ms per epoch against 24.3); 0 = the fp32-MFMA kernel (29.3) -- A/B runs
    static const int bwd_bf16 = getenv("QENV_BWD_BF16") ? atoi(getenv("QENV_BWD_BF16")) : 1;
    if (bwd_bf16 && wide && hidden == mlp3::H && k_in == mlp3::KW) {
        if (bwd_bf16 == 2) {
            e = allow_dynamic_lds((const void*)k_mlp_bwd_v3<512>, (size_t)bwd3::LDS_BYTES);
            if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_bwd_v3)");
            hipLaunchKernelGGL((k_mlp_bwd_v3<512>), dim3((unsigned)blocks), dim3(512), (size_t)bwd3::LDS_BYTES, (hipStream_t)stream, a);
            return check_launch("k_mlp_bwd_v3");
        }
        int64_t b4 = (n_tiles + 3) / 4;
        if (b4 > 256) b4 = 256;
        e = allow_dynamic_lds((const void*)k_mlp_bwd_v3<256>, (size_t)bwd3::LDS_BYTES);
        if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_bwd_v3)");
        hipLaunchKernelGGL((k_mlp_bwd_v3<256>), dim3((unsigned)b4), dim3(256), (size_t)bwd3::LDS_BYTES, (hipStream_t)stream, a);
        return check_launch("k_mlp_bwd_v3");
    }
#define QENV_BWD_CASE(HH)                                                                                                   \
    case HH:                                                                                                                 \
        if (wide) {                                                                                                          \
            e = allow_dynamic_lds((const void*)k_mlp_bwd<HH, 512>, lds);  \
            if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_bwd)");                                       \
            hipLaunchKernelGGL((k_mlp_bwd<HH, 512>), dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, a);        \
        } else {                                                                                                             \
            e = allow_dynamic_lds((const void*)k_mlp_bwd<HH, 256>, lds);  \
            if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_bwd)");                                       \
            hipLaunchKernelGGL((k_mlp_bwd<HH, 256>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);        \
        }                                                                                                                    \
        break;
    switch (hidden) {
        QENV_BWD_CASE(16) QENV_BWD_CASE(32) QENV_BWD_CASE(64) QENV_BWD_CASE(128)
        default: return fail(QENV_ERR_ARG, "qenv_mlp_backward: hidden must be 16, 32, 64 or 128");
    }
#undef QENV_BWD_CASE
    return check_launch("k_mlp_bwd");
}

#ifdef QENV_GRAD5_STAMPS   /* diagnostic builds only (not declared in qenv.h) */
int qenv_debug_grad5_stamps(unsigned long long* out128) {
    return hipMemcpyFromSymbol(out128, HIP_SYMBOL(qenv::g5_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

size_t qenv_mlp_grad_ppo_workspace_bytes(void) { return (size_t)256 * grad5::SLOT * sizeof(float); }

int qenv_mlp_grad_ppo(int64_t n_envs, int n_chunks, int k_in, int hidden, int head, const float* obs, int64_t obs_chunk_stride, const float* actions,
                      const float* old_logprob, const float* adv, const float* returns, const float* weight, double std,
                      double eps_clip, double c_value, int64_t batch, double* sums, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* W3, const float* b3, float* gW1, float* gb1, float* gW2,
                      float* gb2, float* gW3, float* gb3, void* workspace, void* stream) {
    if (!obs || !sums || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !gW1 || !gb1 || !gW2 || !gb2 || !gW3 || !gb3)
        return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: null argument");
    if (hidden != mlp3::H || k_in != mlp3::KW)
        return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: the fused gradient kernel covers the reference's controller (hidden 128, 75 inputs); "
                                  "other shapes: qenv_mlp_backward_ppo + qenv_weight_grad");
    if (head != 0 && head != 1) return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: head must be 0 (actor) or 1 (critic)");
    if (head == 0 && (!actions || !old_logprob || !adv)) return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: the actor needs actions, old_logprob, adv");
    if (head == 1 && !returns) return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: the critic needs returns");
    if (n_envs <= 0 || n_chunks <= 0 || batch <= 0) return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: bad sizes");
    if (!(std > 0)) return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: std must be positive");
    if ((int64_t)hidden * n_envs >= ((int64_t)1 << 29))
        return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: 128*n_envs per chunk must stay below 2^29 (32-bit byte offsets, range-checked loads)");
    GradArgs g;
    BwdArgs& a = g.b;
    a.N = n_envs; a.n_chunks = n_chunks; a.K = k_in; a.n_out = head == 0 ? 4 : 1; a.final_tanh = head == 0;
    a.obs = obs; a.seed = nullptr; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
    a.h1 = a.h2 = a.dz2 = a.dz1 = a.dz3 = nullptr;
    a.loss_mode = head == 0 ? 1 : 2; a.actions = actions; a.old_logprob = old_logprob; a.adv = adv; a.returns = returns;
    a.inv_var = (float)(1.0 / (std * std)); a.inv_std = (float)(1.0 / std); a.log_norm = (float)(-std::log(std) - 0.9189385332046727);
    a.eps_clip = (float)eps_clip; a.c_value = (float)c_value; a.inv_B = 1.f / (float)batch;
    a.sums = sums; a.weight = weight;
    g.gW1 = gW1; g.gb1 = gb1; g.gW2 = gW2; g.gb2 = gb2; g.gW3 = gW3; g.gb3 = gb3; g.partial = (float*)workspace;
    if (obs_chunk_stride < 0 || (obs_chunk_stride != 0 && obs_chunk_stride < n_envs))
        return fail(QENV_ERR_ARG, "qenv_mlp_grad_ppo: obs_chunk_stride is 0 (dense: k_in * n_envs) or the element distance of consecutive chunks");
    g.obs_stride = obs_chunk_stride ? obs_chunk_stride : (int64_t)k_in * n_envs;
    const int64_t n_tiles = ((n_envs + 31) / 32) * n_chunks;
    const int cus = device_cu_count();
    int64_t blocks = n_tiles < cus ? n_tiles : cus;   // one resident workgroup per CU (its registers hold W2 twice); at most the workspace's 256 slots
    if (blocks > 256) blocks = 256;
    hipError_t e = allow_dynamic_lds((const void*)k_mlp_grad_v5, (size_t)grad5::LDS_BYTES);
    if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(k_mlp_grad_v5)");
    hipLaunchKernelGGL(k_mlp_grad_v5, dim3((unsigned)blocks), dim3(512), (size_t)grad5::LDS_BYTES, (hipStream_t)stream, g);
    int rc = check_launch("k_mlp_grad_v5");
    if (rc == QENV_OK && workspace) {
        hipLaunchKernelGGL(k_sum_grad_slots, dim3((unsigned)((grad5::SLOT + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)workspace, (int)blocks, a.n_out, gW1, gb1, gW2, gb2, gW3, gb3);
        rc = check_launch("k_sum_grad_slots");
    }
    return rc;
}

static int wgrad_impl(int64_t n_envs, int n_chunks, int m_rows, int n_rows, const float* A, const float* B, float* C,
                      float* rowsum, float* partial, void* stream);

int qenv_weight_grad(int64_t n_envs, int n_chunks, int m_rows, int n_rows, const float* A, const float* B, float* C,
                     float* rowsum, void* stream) {
    return wgrad_impl(n_envs, n_chunks, m_rows, n_rows, A, B, C, rowsum, nullptr, stream);
}

size_t qenv_weight_grad_workspace_bytes(int m_rows, int n_rows) {
    return (m_rows < 1 || n_rows < 1) ? 0 : (size_t)512 * ((size_t)m_rows * n_rows + m_rows) * sizeof(float);
}

int qenv_weight_grad_deterministic(int64_t n_envs, int n_chunks, int m_rows, int n_rows, const float* A, const float* B,
                                   float* C, float* rowsum, void* workspace, void* stream) {
    if (!workspace) return fail(QENV_ERR_ARG, "qenv_weight_grad_deterministic: workspace is required");
    return wgrad_impl(n_envs, n_chunks, m_rows, n_rows, A, B, C, rowsum, (float*)workspace, stream);
}

static int wgrad_impl(int64_t n_envs, int n_chunks, int m_rows, int n_rows, const float* A, const float* B, float* C,
                      float* rowsum, float* partial, void* stream) {
    if (!A || !B || !C) return fail(QENV_ERR_ARG, "qenv_weight_grad: null argument");
    if (n_envs <= 0 || n_chunks <= 0 || m_rows < 1 || m_rows > 128 || n_rows < 1 || n_rows > 128)
        return fail(QENV_ERR_ARG, "qenv_weight_grad: rows must be 1..128");
    if ((int64_t)128 * n_envs >= ((int64_t)1 << 29))
        return fail(QENV_ERR_ARG, "qenv_weight_grad: 128*n_envs must stay below 2^29 (32-bit byte offsets, range-checked loads)");
    WgradArgs a;
    a.N = n_envs; a.n_chunks = n_chunks; a.M = m_rows; a.Nn = n_rows; a.A = A; a.B = B; a.C = C; a.rowsum = rowsum; a.partial = partial;
    const int mt = (m_rows + 31) / 32, nt = (n_rows + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
#define QENV_WG(MT_, NT_) if (mt == MT_ && nt == NT_) return launch_wgrad<MT_, NT_>(a, s);
    QENV_WG(1, 1) QENV_WG(1, 2) QENV_WG(1, 3) QENV_WG(1, 4) QENV_WG(2, 1) QENV_WG(2, 2) QENV_WG(2, 3) QENV_WG(2, 4)
    QENV_WG(3, 1) QENV_WG(3, 2) QENV_WG(3, 3) QENV_WG(3, 4) QENV_WG(4, 1) QENV_WG(4, 2) QENV_WG(4, 3) QENV_WG(4, 4)
#undef QENV_WG
    return fail(QENV_ERR_ARG, "qenv_weight_grad: unsupported tile shape");
}

int qenv_adam_step(int64_t n, float* param, const float* grad, float* m, float* v, double lr, double beta1, double beta2,
                   double eps, int step, void* stream) {
    if (!param || !grad || !m || !v) return fail(QENV_ERR_ARG, "qenv_adam_step: null argument");
    if (n <= 0 || step < 1) return fail(QENV_ERR_ARG, "qenv_adam_step: n must be positive and step >= 1");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const double bc1 = 1.0 - std::pow(beta1, step), bc2 = 1.0 - std::pow(beta2, step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, param, grad, m, v, (float)lr,
                       (float)beta1, (float)beta2, (float)eps, (float)bc1, (float)bc2);
    return check_launch("k_adam");
}

int qenv_adam_step_counted(int64_t n, float* param, const float* grad, float* m, float* v, double lr, double beta1, double beta2,
                           double eps, int32_t* step_counter, void* stream) {
    if (!param || !grad || !m || !v || !step_counter || n <= 0) return fail(QENV_ERR_ARG, "qenv_adam_step_counted: bad argument");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_count, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter);
    hipLaunchKernelGGL(k_adam_counted, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, param, grad, m, v, (float)lr,
                       (float)beta1, (float)beta2, (float)eps, beta1, beta2, (const int32_t*)step_counter);
    return check_launch("k_adam_counted");
}

}  // extern "C"
