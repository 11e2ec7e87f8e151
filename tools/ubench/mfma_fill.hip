// How much single-issue VALU work hides under v_mfma_f32_32x32x16_bf16 (and v_mfma_f32_32x32x2_f32) on gfx950?
//
// Round 2's mfma_valu / mfma_interleave concluded "nothing hides"; MI355X_MICROARCH.md says a 32x32x16 bf16 MFMA holds the
// SIMD's vector issue for 8 of its 32 cycles and <= 5 four-cycle fillers per gap are (nearly) free.  This bench settles it
// with IN-KERNEL cycle counts (s_memtime around the loop, per wave, median over waves -- no launch overhead, no
// assumption about the clock), an instruction stream checked in the ISA (sched_barrier(0) + volatile asm pin the order), and the three placements that matter for k_mlp:
//   A  one wave per SIMD: NACC accumulator chains, NF independent v_fma_f32 after every MFMA
//   B  two waves per SIMD (512-thread workgroup), both running program A
//   C  two waves per SIMD, waves 0-3 MFMA only, waves 4-7 VALU only (round 2's experiment, timed per wave)
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_fill tools/ubench/mfma_fill.hip && tools/ubench/mfma_fill
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// KIND 0: v_mfma_f32_32x32x16_bf16 (32 pipe cycles), 1: v_mfma_f32_32x32x2_f32 (64)
// ROLE 0: MFMA + fillers (program A), 1: waves >= 4 run VALU only and waves < 4 MFMA only (program C)
template <int KIND, int NACC, int NF, int ROLE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ float pad[];  // > 80 KiB requested at launch: one workgroup per CU
    const int wave = threadIdx.x >> 6;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const float fa = threadIdx.x * 1e-3f, fb = 1.0f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(fa + i); b8[i] = (__bf16)1.0f; }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = fa + i;
    float c1 = 0.999f, c2 = 0.001f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    const bool do_mfma = ROLE == 0 || wave < 4, do_valu = ROLE == 0 || wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if (do_mfma) {
                    if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[a], 0, 0, 0);
                    else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
                }
                // the order MFMA, NF x v_fma_f32, MFMA, ... is pinned: nothing crosses a sched_barrier(0), and the fillers are
                // volatile asm (the compiler neither merges nor reorders them); eight independent chains
                __builtin_amdgcn_sched_barrier(0);
                if (do_valu) {
#pragma unroll
                    for (int v = 0; v < NF; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(c1), "v"(c2));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) r += acc[a][i];
    for (int i = 0; i < 8; ++i) r += x[i];
    if (r == 123.456f) sink[0] = r + pad[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (BLOCK / 64) + wave] = t1 - t0;
}

static unsigned long long* d_cyc; static float* d_sink;

template <int KIND, int NACC, int NF, int ROLE, int BLOCK> int run(const char* tag) {
    const int iters = 512, waves = BLOCK / 64, blocks = 256;
    auto fn = k<KIND, NACC, NF, ROLE, BLOCK>;
    CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(fn, dim3(blocks), dim3(BLOCK), 96 * 1024, 0, iters, d_cyc, d_sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * waves);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    const double per = 4.0 * NACC * iters;  // MFMAs (or filler groups) per wave
    auto med = [&](int lo, int hi) {
        std::vector<double> v;
        for (int b = 0; b < blocks; ++b) for (int w = lo; w < hi; ++w) v.push_back((double)h[b * waves + w] / per);
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    };
    if (ROLE == 0) printf("%-4s kind=%d waves/SIMD=%d chains=%d fillers/MFMA=%2d : %6.1f cycles per MFMA\n", tag, KIND, BLOCK / 256, NACC, NF,
                          med(0, waves));
    else printf("%-4s kind=%d split roles, %2d fillers per group: MFMA waves %6.1f cycles per MFMA, VALU waves %6.1f cycles per group (%.2f per v_fma)\n",
                tag, KIND, NF, med(0, 4), med(4, 8), NF ? med(4, 8) / NF : 0.0);
    return 0;
}

int main() {
    CHECK(hipMalloc(&d_cyc, 256 * 8 * 8)); CHECK(hipMalloc(&d_sink, 4));
    int rc = 0;
    // A: one wave per SIMD
    rc |= run<0, 1, 0, 0, 256>("A"); rc |= run<0, 1, 2, 0, 256>("A"); rc |= run<0, 1, 4, 0, 256>("A"); rc |= run<0, 1, 5, 0, 256>("A");
    rc |= run<0, 1, 6, 0, 256>("A"); rc |= run<0, 1, 8, 0, 256>("A"); rc |= run<0, 1, 12, 0, 256>("A");
    rc |= run<0, 4, 0, 0, 256>("A"); rc |= run<0, 4, 4, 0, 256>("A"); rc |= run<0, 4, 6, 0, 256>("A"); rc |= run<0, 4, 8, 0, 256>("A");
    rc |= run<1, 1, 0, 0, 256>("A"); rc |= run<1, 1, 6, 0, 256>("A"); rc |= run<1, 1, 12, 0, 256>("A"); rc |= run<1, 1, 16, 0, 256>("A");
    rc |= run<1, 4, 12, 0, 256>("A");
    // B: two waves per SIMD, same program
    rc |= run<0, 1, 0, 0, 512>("B"); rc |= run<0, 1, 3, 0, 512>("B"); rc |= run<0, 1, 6, 0, 512>("B"); rc |= run<0, 4, 3, 0, 512>("B");
    rc |= run<1, 1, 6, 0, 512>("B");
    // C: split roles
    rc |= run<0, 1, 6, 1, 512>("C"); rc |= run<0, 1, 12, 1, 512>("C"); rc |= run<1, 1, 12, 1, 512>("C"); rc |= run<0, 4, 6, 1, 512>("C");
    return rc;
}
