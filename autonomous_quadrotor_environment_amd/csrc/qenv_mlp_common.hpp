// qenv_mlp_common.hpp -- device helpers shared by the MLP forward (qenv_ppo.hip) and the PPO update
// kernels (qenv_train.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace qenv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// x = hi + mid + lo with hi, mid, lo in bf16 (round to nearest each time): 24 significant bits, i.e. the fp32 value
// exactly unless its exponent range underflows bf16's (it has fp32's).  Written on PAIRS of values as two-element vectors:
// the residuals are then v_pk_add_f32 (one instruction per pair) whatever registers the values sit in -- on scalars hipcc
// takes the packed form only when the pair happens to be register-aligned and otherwise converts every value twice (once
// alone for its residual, once more beside its neighbour for the packed term: k_dense_tail_v3's chunk loop had 128
// conversions for 40 values).  The same arithmetic, bit for bit.
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ void split3x8(const float x[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 v = {x[i], x[i + 1]};
        const bf16x2 h = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
        const bf16x2 m = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
        const bf16x2 l = __builtin_convertvector(r2, bf16x2);
        hi[i] = h[0]; hi[i + 1] = h[1]; mid[i] = m[0]; mid[i + 1] = m[1]; lo[i] = l[0]; lo[i + 1] = l[1];
    }
}

// C/D layout of v_mfma_f32_32x32x2_f32: register j of lane-half `half` holds row (j&3) + 8 (j>>2) + 4 half
__device__ __forceinline__ int acc_row(int j, int half) { return (j & 3) + 8 * (j >> 2) + 4 * half; }

// tanh(x) = 1 - 2 / (exp(2x) + 1) on v_exp_f32 / v_rcp_f32: 5 VALU instructions (7 issue slots; ocml's tanhf is
// ~40 and the two hidden layers need 256 of them per env).  ABSOLUTE error <= ~1.2e-7 everywhere (near 0 the
// subtraction from 1 rounds at ulp(1) = 6e-8: the relative error of a tiny activation is large, its absolute error is
// not, and only that propagates through the next layer's dot products); exact +-1 saturation (exp -> inf / 0).
// fp32 MFMAs execute on the SIMD's vector FMA lanes (measured, tools/ubench/mfma_interleave.hip: VALU instructions
// placed between dependent v_mfma_f32_32x32x2_f32 are not hidden, each costs its full issue time and more), so every
// VALU instruction of the activation adds directly to the kernel time.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// k_mlp_v3 / k_mlp_small<.,.,.,3>: the hidden layers' weights and biases are multiplied by 2 log2(e) when they are staged
// (once per weight update with a prepared image), so the matrix cores deliver z' = 2 log2(e) z and the activation is
// exp2 / add / rcp / fma: one VALU instruction fewer on each of the 128 activations per env (the kernel is VALU-issue bound).
constexpr float TANH_PRESCALE = 2.8853900817779268f;
__device__ __forceinline__ float fast_tanh_prescaled(float zs) {
    const float e = __builtin_amdgcn_exp2f(zs);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ---- shared by k_mlp_v3 / k_mlp_small (qenv_ppo.hip) and k_mlp_bwd_v3 (qenv_train.hip): the N = 128, T = 5 policy on the bf16
// matrix cores with three-way split operands
namespace mlp3 {
constexpr int H = 128, NT = 4, TW = 5, KW = 75, KS16 = 5;
constexpr int W1F = 0;                               // uint4 offsets into the image
constexpr int W2F = W1F + NT * KS16 * 3 * 64;        // 3840
constexpr int W3T = W2F + NT * NT * 2 * 3 * 64;      // 9984: [128] float4 = W3[0..3][h], rows >= n_out zero
constexpr int B1 = W3T + H;                          // 10112: 128 floats
constexpr int B2 = B1 + H / 4;                       // 10144
constexpr int B3 = B2 + H / 4;                       // 10176: 4 floats, zero beyond n_out
constexpr int IMAGE_UINT4 = B3 + 1;                  // 10177 -> 162 832 B
}  // namespace mlp3

// layer 3 on the vector lanes: o[k] += W3[k][row] * h2[row] over the 16 rows register j of this lane half holds in output
// tile t of layer 2, in register order.  One definition for k_mlp_v3 and k_mlp_small (bit-identical outputs).
__device__ __forceinline__ void mlp3_layer3(const float4* W3t, int t, int half, const f32x16& h2, float o[4]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 w = W3t[32 * t + acc_row(j, half)];
        o[0] = fmaf(w.x, h2[j], o[0]); o[1] = fmaf(w.y, h2[j], o[1]);
        o[2] = fmaf(w.z, h2[j], o[2]); o[3] = fmaf(w.w, h2[j], o[3]);
    }
}
// the two lane halves' partial sums meet: the result is valid in the lower half (which started from the bias)
__device__ __forceinline__ void mlp3_layer3_finish(float o[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = o[k] + __shfl_xor(o[k], 32);
}

// N_MFMA times { one MFMA, then N_VALU VALU instructions }: the interleave a scheduling region is asked for (the region's
// other instructions -- LDS reads, waits -- go where the scheduler likes)
template <int N_MFMA, int N_VALU> __device__ __forceinline__ void sgb_interleave() {
#pragma unroll
    for (int m = 0; m < N_MFMA; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, N_VALU, 0);
    }
}

}  // namespace qenv
