// qenv_mlp_common.hpp -- device helpers shared by the MLP forward (qenv_ppo.hip) and the PPO update
// kernels (qenv_train.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace qenv {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// C/D layout of v_mfma_f32_32x32x2_f32: register j of lane-half `half` holds row (j&3) + 8 (j>>2) + 4 half
__device__ __forceinline__ int acc_row(int j, int half) { return (j & 3) + 8 * (j >> 2) + 4 * half; }

// tanh with ~1e-7 absolute error in ~12 VALU instructions (ocml's tanhf is ~3x that and the two
// hidden layers need 256 of them per env): odd Taylor polynomial below 0.3 (next term < 2e-8),
// 1 - 2/(exp(2|x|)+1) on v_exp_f32 / v_rcp_f32 above; saturates cleanly for large |x|.
__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    const float small = ax * (1.0f + x2 * (-0.33333333333f + x2 * (0.13333333333f + x2 * (-0.05396825397f + x2 * 0.02186948854f))));
    return copysignf(ax < 0.3f ? small : big, x);
}

}  // namespace qenv
