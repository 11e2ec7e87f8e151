// Does VALU work of one wave overlap with the MFMAs of the other wave on the same SIMD?  fp32 vs bf16 MFMA.
// Block = 512 threads (8 waves: 2 per SIMD), 256 blocks.  mode bits: 1 = waves 0-3 run MFMAs, 2 = waves 4-7 run VALU
// fma chains; kind 0 = v_mfma_f32_32x32x2_f32, 1 = v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, float* out) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc = {0};
            float a = threadIdx.x * 1e-3f, b = 1.0f;
            bf16x8 a8, b8;
            for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)1.0f; }
            for (int i = 0; i < n_mfma; ++i) {
                if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            }
            for (int i = 0; i < 16; ++i) r += acc[i];
        }
    } else if (mode & 2) {
        float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        for (int i = 0; i < n_valu; ++i) {  // 4 independent chains, 4 fma per iteration
            x0 = fmaf(x0, 0.999f, 0.001f); x1 = fmaf(x1, 0.999f, 0.001f);
            x2 = fmaf(x2, 0.999f, 0.001f); x3 = fmaf(x3, 0.999f, 0.001f);
        }
        r = x0 + x1 + x2 + x3;
    }
    if (r == 123.456f) out[0] = r;
}

template <int KIND> float run(int mode, int n_mfma, int n_valu, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, n_mfma, n_valu, d);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, n_mfma, n_valu, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* d; hipMalloc(&d, 4);
    const int nm32 = 4096, nm16 = 8192, nv = 16384;  // 4096 fp32 MFMAs = 262k cycles; 8192 bf16 = 262k; 65536 fma = 262k
    printf("fp32 mfma alone   %.1f us\n", run<0>(1, nm32, nv, d));
    printf("valu alone        %.1f us\n", run<0>(2, nm32, nv, d));
    printf("fp32 mfma + valu  %.1f us\n", run<0>(3, nm32, nv, d));
    printf("bf16 mfma alone   %.1f us\n", run<1>(1, nm16, nv, d));
    printf("bf16 mfma + valu  %.1f us\n", run<1>(3, nm16, nv, d));
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
