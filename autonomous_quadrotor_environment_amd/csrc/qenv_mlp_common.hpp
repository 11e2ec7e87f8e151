// qenv_mlp_common.hpp -- device helpers shared by the MLP forward (qenv_ppo.hip) and the PPO update
// kernels (qenv_train.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace qenv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// x = hi + mid + lo with hi, mid, lo in bf16 (round to nearest each time): 24 significant bits, i.e. the fp32 value
// exactly unless its exponent range underflows bf16's (it has fp32's)
__device__ __forceinline__ void split3x8(const float x[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
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

}  // namespace qenv
