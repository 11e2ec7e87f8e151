// One wave per SIMD: NV independent v_fma_f32 placed in program order between consecutive DEPENDENT
// v_mfma_f32_32x32x2_f32 of one accumulator chain: how many are hidden under the MFMA's 64 cycles?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k(int n, float* out) {
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    for (int it = 0; it < n; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v & 7] = fmaf(x[v & 7], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i];
    for (int i = 0; i < 8; ++i) r += x[i];
    if (r == 123.456f) out[0] = r;
}

template <int NV> void run(float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 0, 0, 4096, d);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 0, 0, 4096, d);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("NV=%2d  %.1f us  (%.1f cycles per MFMA+fillers at 2.4 GHz)\n", NV, ms / 5 * 1e3f, ms / 5 * 1e-3 * 2.4e9 / 4096);
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    run<0>(d); run<4>(d); run<8>(d); run<12>(d); run<14>(d); run<16>(d); run<24>(d);
    return hipGetLastError() != hipSuccess;
}
