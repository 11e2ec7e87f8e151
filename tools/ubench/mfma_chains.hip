// One wave per SIMD: v_mfma_f32_32x32x2_f32 issued on 1, 2 or 4 independent accumulator chains: cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NC>
__global__ __launch_bounds__(256) void k(int n, float* out) {
    f32x16 acc[NC];
    for (int c = 0; c < NC; ++c) acc[c] = f32x16{0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    for (int it = 0; it < n; it += NC) {
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float r = 0.f;
    for (int c = 0; c < NC; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
    if (r == 123.456f) out[0] = r;
}

template <int NC> void run(float* d, int waves) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NC>, dim3(256), dim3(64 * waves), 0, 0, 4096, d);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NC>, dim3(256), dim3(64 * waves), 0, 0, 4096, d);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("chains=%d waves/SIMD=%d  %.1f us  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NC, waves / 4, ms / 5 * 1e3f,
           ms / 5 * 1e-3 * 2.4e9 / (4096.0 * waves / 4));
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    run<1>(d, 4); run<2>(d, 4); run<4>(d, 4); run<1>(d, 8); run<2>(d, 8);
    return hipGetLastError() != hipSuccess;
}
