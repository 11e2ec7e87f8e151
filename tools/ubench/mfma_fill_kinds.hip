// Which vector instructions hide in the shadow of a DEPENDENT chain of v_mfma_f32_32x32x16_bf16 (one wave per SIMD)?
// mfma_fill.hip measured v_fma_f32 fillers with constant MFMA operands; a policy tile (mlp3_tile_forward) puts tanh
// (v_exp_f32 / v_rcp_f32), the bf16 split (v_cvt_pk_bf16_f32, shifts, subtractions), v_accvgpr_read_b32 and LDS reads of the
// next step's weight fragments there.  Per filler kind: cycles per MFMA with NF fillers behind every MFMA of one chain
// (s_memtime, median over the waves; the order is pinned by sched_barrier(0) + volatile asm).
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_fill_kinds tools/ubench/mfma_fill_kinds.hip && tools/ubench/mfma_fill_kinds
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { F_FMA, F_EXP, F_RCP, F_CVT, F_ACCRD, F_SHIFT, F_LDSA, F_PK, N_KIND };
static const char* names[] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_pk_bf16_f32", "v_accvgpr_read_b32", "v_lshlrev_b32",
                              "v_fma_f32 + A operand from LDS (ds_read_b128 x3 per MFMA, one step ahead)", "v_pk_fma_f32"};

template <int KIND, int NF>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ uint4 lds[];   // > 80 KiB requested: one workgroup per CU
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc, other;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; other[i] = 1.f + i; }
    const float fa = threadIdx.x * 1e-3f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(fa + i); b8[i] = (__bf16)1.0f; }
    float x[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
    for (int i = 0; i < 8; ++i) x[i] = fa + i;
    for (int i = 0; i < 4; ++i) p[i] = f2{fa + i, fa - i};
    float c1 = 0.999f, c2 = 0.001f;
    f2 cp = {0.999f, 0.998f};
    asm volatile("" : "+v"(c1), "+v"(c2), "+v"(cp));
    asm volatile("" : "+a"(other));   // sixteen AGPRs the MFMA chain does not touch
    uint4 fr[3];
    const uint4* src = lds + lane;
    fr[0] = src[0]; fr[1] = src[64]; fr[2] = src[128];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (KIND == F_LDSA) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, fr[rep % 3]);
                const uint4* nx = src + 192 * ((it * 8 + rep + 1) & 15);
                fr[0] = nx[0]; fr[1] = nx[64]; fr[2] = nx[128];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b8, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NF; ++v) {
                if (KIND == F_FMA || KIND == F_LDSA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c1), "v"(c2));
                if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[v & 7]));
                if (KIND == F_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[v & 7]));
                if (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(c1));
                if (KIND == F_ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[v & 7]) : "a"(other[v & 15]));
                if (KIND == F_SHIFT) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x[v & 7]));
                if (KIND == F_PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[v & 3]) : "v"(cp));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i] + other[i];
    for (int i = 0; i < 8; ++i) r += x[i];
    for (int i = 0; i < 4; ++i) r += p[i][0] + p[i][1];
    r += __builtin_bit_cast(float, fr[0].x) + __builtin_bit_cast(float, fr[1].x) + __builtin_bit_cast(float, fr[2].x);
    if (r == 123.456f) sink[0] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static unsigned long long* d_cyc; static float* d_sink;

template <int KIND, int NF> int run() {
    const int iters = 256, blocks = 256;
    auto fn = k<KIND, NF>;
    CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 96 * 1024, 0, iters, d_cyc, d_sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (auto c : h) v.push_back((double)c / (8.0 * iters));
    std::sort(v.begin(), v.end());
    printf("  %-80s NF=%d: %6.1f cycles per MFMA\n", names[KIND], NF, v[v.size() / 2]);
    return 0;
}

template <int KIND> int all() {
    if (run<KIND, 0>()) return 1;
    if (run<KIND, 2>()) return 1;
    if (run<KIND, 4>()) return 1;
    if (run<KIND, 6>()) return 1;
    if (run<KIND, 8>()) return 1;
    return 0;
}

int main() {
    CHECK(hipMalloc(&d_cyc, 256 * 4 * 8)); CHECK(hipMalloc(&d_sink, 4));
    printf("one dependent chain of v_mfma_f32_32x32x16_bf16, NF fillers behind every MFMA, one wave per SIMD\n");
    if (all<F_FMA>() || all<F_EXP>() || all<F_RCP>() || all<F_CVT>() || all<F_ACCRD>() || all<F_SHIFT>() || all<F_PK>() || all<F_LDSA>()) return 1;
    return 0;
}
