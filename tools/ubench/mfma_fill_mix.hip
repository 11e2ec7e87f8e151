// What the slots of a policy tile (mlp3_tile_forward) cost: cycles per MFMA of a DEPENDENT chain of v_mfma_f32_32x32x16_bf16 with one
// of the tile's instruction MIXES behind every MFMA (one wave per SIMD; the order pinned by sched_barrier(0) + volatile asm).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_fill_mix tools/ubench/mfma_fill_mix.hip && tools/ubench/mfma_fill_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { M_NONE, M_SPLIT_A, M_SPLIT_B, M_SPLIT_A_CVT, M_SPLIT_B_CVT, M_TANH4, M_EXP_FMA, M_ADD_4FMA, M_RCP, M_1T4P, M_1T5P, M_SPLIT_AB, M_6SUB, N_MIX };
static const char* names[] = {"nothing", "split half A: cvt_pk, lshl, and, sub, sub, cvt_pk", "split half B: lshl, and, sub, sub, cvt_pk",
                              "split half A with v_cvt_f32_bf16 (e32 / sdwa WORD_1) for the expansions", "split half B with v_cvt_f32_bf16 expansions",
                              "tanh pipeline slot: exp, add, rcp, fma", "layer-2 slot: exp, fma", "layer-2 slot: add, 4 fma", "layer-2 slot: rcp",
                              "1 transcendental + 4 plain (exp, add, fma, sub, cvt_pk)", "1 transcendental + 5 plain", "a whole pair: halves A + B (11)",
                              "six v_sub_f32"};

template <int MIX>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* cyc, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float fa = threadIdx.x * 1e-3f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(fa + i); b8[i] = (__bf16)1.0f; }
    float x0 = fa + 1.f, x1 = fa + 2.f, r0 = 0.f, r1 = 0.f, e = fa, q = 1.5f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f, w = 0.5f;
    unsigned h = 0, m = 0, t0u = 0, t1u = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MIX == M_SPLIT_A || MIX == M_SPLIT_AB) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0u) : "v"(h));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1u) : "v"(h));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(t0u));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(t1u));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
            }
            if (MIX == M_SPLIT_B || MIX == M_SPLIT_AB) {
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t0u) : "v"(m));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t1u) : "v"(m));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x0) : "v"(r0), "v"(t0u));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x1) : "v"(r1), "v"(t1u));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
            }
            if (MIX == M_SPLIT_A_CVT) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_f32_bf16_e32 %0, %1" : "=v"(t0u) : "v"(h));
                asm volatile("v_cvt_f32_bf16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(t1u) : "v"(h));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(t0u));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(t1u));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
            }
            if (MIX == M_SPLIT_B_CVT) {
                asm volatile("v_cvt_f32_bf16_e32 %0, %1" : "=v"(t0u) : "v"(m));
                asm volatile("v_cvt_f32_bf16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(t1u) : "v"(m));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x0) : "v"(r0), "v"(t0u));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x1) : "v"(r1), "v"(t1u));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
            }
            if (MIX == M_TANH4) {   // four independent stages of the pipeline
                asm volatile("v_fma_f32 %0, %1, -2.0, 1.0" : "=v"(o0) : "v"(q));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(q) : "v"(r1));
                asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(r1) : "v"(e));
                asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x0));
            }
            if (MIX == M_EXP_FMA) {
                asm volatile("v_fma_f32 %0, %1, -2.0, 1.0" : "=v"(o0) : "v"(q));
                asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x0));
            }
            if (MIX == M_ADD_4FMA) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o0) : "v"(w), "v"(q));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o1) : "v"(w), "v"(q));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o2) : "v"(w), "v"(q));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(o3) : "v"(w), "v"(q));
                asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(r1) : "v"(e));
            }
            if (MIX == M_RCP) asm volatile("v_rcp_f32 %0, %1" : "=v"(q) : "v"(r1));
            if (MIX == M_1T4P || MIX == M_1T5P) {
                asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x0));
                asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(r1) : "v"(q));
                asm volatile("v_fma_f32 %0, %1, -2.0, 1.0" : "=v"(o0) : "v"(q));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(x1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
                if (MIX == M_1T5P) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o1) : "v"(x1), "v"(x0));
            }
            if (MIX == M_6SUB) {
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o0) : "v"(x0), "v"(x1));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o1) : "v"(x1), "v"(x0));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o2) : "v"(x0), "v"(w));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o3) : "v"(x1), "v"(w));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(w), "v"(x1));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(w), "v"(x0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = x0 + x1 + r0 + r1 + e + q + o0 + o1 + o2 + o3 + __uint_as_float(h) + __uint_as_float(m);
    for (int i = 0; i < 16; ++i) r += acc[i];
    if (r == 123.456f) sink[0] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

static unsigned long long* d_cyc; static float* d_sink;
template <int MIX> int run() {
    const int iters = 256, blocks = 256;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MIX>, dim3(blocks), dim3(256), 0, 0, iters, d_cyc, d_sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (auto c : h) v.push_back((double)c / (8.0 * iters));
    std::sort(v.begin(), v.end());
    printf("  %-86s %6.1f cycles per MFMA\n", names[MIX], v[v.size() / 2]);
    return 0;
}
int main() {
    CHECK(hipMalloc(&d_cyc, 256 * 4 * 8)); CHECK(hipMalloc(&d_sink, 4));
    printf("one dependent chain of v_mfma_f32_32x32x16_bf16, one instruction mix behind every MFMA, one wave per SIMD\n");
    if (run<M_NONE>() || run<M_SPLIT_A>() || run<M_SPLIT_B>() || run<M_SPLIT_A_CVT>() || run<M_SPLIT_B_CVT>() || run<M_SPLIT_AB>() || run<M_6SUB>() ||
        run<M_TANH4>() || run<M_EXP_FMA>() || run<M_ADD_4FMA>() || run<M_RCP>() || run<M_1T4P>() || run<M_1T5P>()) return 1;
    return 0;
}
