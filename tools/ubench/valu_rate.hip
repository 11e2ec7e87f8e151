// Issue cost of wave64 VALU instructions on gfx950, one .. four waves per SIMD: cycles per instruction from s_memtime
// around an unrolled block of 16 independent chains.  Prints cycles per instruction per wave and the aggregate
// instructions per cycle per SIMD.   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { FMA32, PKFMA32, FMA64, MUL64, ADD64, RCP64, RSQ64, SQRT64, RCP32, SQRT32, MOV64, CVT_F64_F32, CVT_F32_F64, N_KIND };
static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_rsq_f64",
                              "v_sqrt_f64", "v_rcp_f32", "v_sqrt_f32", "v_mov_b64", "v_cvt_f64_f32", "v_cvt_f32_f64"};

template <int KIND>
__global__ __launch_bounds__(1024) void k(int iters, unsigned long long* cyc, double* sink) {
    double d[16];
    float f[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
    for (int i = 0; i < 16; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = (float)d[i]; p[i] = f2{f[i], f[i] + 1.f}; }
    const double cd = 0.999, ad = 1e-3;
    const float cf = 0.999f, af = 1e-3f;
    const f2 cp = {0.999f, 0.998f}, ap = {1e-3f, 2e-3f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OP(i)                                                                                               \
    if (KIND == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(cf), "v"(af));            \
    else if (KIND == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cp), "v"(ap));  \
    else if (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(cd), "v"(ad));       \
    else if (KIND == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));                    \
    else if (KIND == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(ad));                    \
    else if (KIND == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));                                  \
    else if (KIND == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));                                  \
    else if (KIND == SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));                                \
    else if (KIND == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));                                  \
    else if (KIND == SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));                                \
    else if (KIND == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) & 15]));           \
    else if (KIND == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));            \
    else if (KIND == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
        REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
    for (int i = 0; i < 16; ++i) r += d[i] + f[i] + p[i].x + p[i].y;
    if (r == 123.456) sink[0] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(int threads, unsigned long long* dc, double* ds) {
    const int iters = 2000, blocks = 256, waves = threads / 64;
    std::vector<unsigned long long> h(blocks * waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, dc, ds);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, dc, ds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = 64.0 * iters;
    const double per_wave = h[h.size() / 2] / n;       // s_memtime ticks (shader cycles? 100 MHz? -> compare with wall)
    const double wall_cyc = ms * 1e-3 * 2.4e9 / n;     // cycles at 2.4 GHz per instruction per wave (wall time)
    printf("%-14s waves/SIMD %d: memtime ticks/inst/wave %.2f | wall: %.2f cyc/inst/wave @2.4GHz -> %.2f cyc per inst per SIMD\n",
           names[KIND], waves / 4, per_wave, wall_cyc, wall_cyc / (waves / 4));
}

int main() {
    unsigned long long* dc; double* ds;
    hipMalloc(&dc, 8 * 256 * 16); hipMalloc(&ds, 8);
    for (int threads : {256, 512, 1024}) {
        run<FMA32>(threads, dc, ds); run<PKFMA32>(threads, dc, ds); run<FMA64>(threads, dc, ds); run<MUL64>(threads, dc, ds);
        run<ADD64>(threads, dc, ds); run<RCP64>(threads, dc, ds); run<RSQ64>(threads, dc, ds); run<SQRT64>(threads, dc, ds);
        run<RCP32>(threads, dc, ds); run<SQRT32>(threads, dc, ds); run<MOV64>(threads, dc, ds);
        run<CVT_F64_F32>(threads, dc, ds); run<CVT_F32_F64>(threads, dc, ds);
    }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
