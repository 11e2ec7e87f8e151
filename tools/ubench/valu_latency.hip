// Dependent-issue latency of wave64 VALU instructions on gfx950 with ONE wave per SIMD: cycles per instruction (s_memtime) of an
// unrolled stream in which NCH independent chains are interleaved round-robin (NCH = 1: every instruction waits for the one
// before it; large NCH: the issue rate).  The env kernels run one wave per SIMD, so a chain of dependent fp64 operations (a Newton
// refinement, a Horner polynomial) costs this latency per link unless independent work stands between the links.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_latency tools/ubench/valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

enum { FMA64, MUL64, ADD64, FMA32, RCP64, RSQ64, EXP32, RCP32, N_KIND };
static const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_rcp_f64", "v_rsq_f64", "v_exp_f32", "v_rcp_f32"};

template <int KIND, int NCH>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* cyc, double* sink) {
    double d[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = (float)d[i]; }
    const double cd = 0.999, ad = 1e-3;
    const float cf = 0.999f, af = 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int i = u % NCH;
            if (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(cd), "v"(ad));
            else if (KIND == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            else if (KIND == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(ad));
            else if (KIND == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(cf), "v"(af));
            else if (KIND == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            else if (KIND == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
            else if (KIND == EXP32) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
            else if (KIND == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
    for (int i = 0; i < 8; ++i) r += d[i] + f[i];
    if (r == 123.456) sink[0] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NCH> double run(unsigned long long* dc, double* ds) {
    const int iters = 500, blocks = 256;
    std::vector<unsigned long long> h(blocks * 4);
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(blocks), dim3(256), 0, 0, iters, dc, ds);
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(blocks), dim3(256), 0, 0, iters, dc, ds);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return h[h.size() / 2] / (64.0 * iters);
}
template <int KIND> void row(unsigned long long* dc, double* ds) {
    printf("%-10s cycles per instruction with 1 / 2 / 3 / 4 / 6 / 8 interleaved chains: %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f\n", names[KIND],
           run<KIND, 1>(dc, ds), run<KIND, 2>(dc, ds), run<KIND, 3>(dc, ds), run<KIND, 4>(dc, ds), run<KIND, 6>(dc, ds), run<KIND, 8>(dc, ds));
}
int main() {
    unsigned long long* dc; double* ds;
    hipMalloc(&dc, 8 * 256 * 4); hipMalloc(&ds, 8);
    printf("one wave per SIMD (256 workgroups of 256 threads), s_memtime ticks = shader cycles\n");
    row<FMA64>(dc, ds); row<MUL64>(dc, ds); row<ADD64>(dc, ds); row<FMA32>(dc, ds);
    row<RCP64>(dc, ds); row<RSQ64>(dc, ds); row<EXP32>(dc, ds); row<RCP32>(dc, ds);
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
