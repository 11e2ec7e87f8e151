// How much straight-line code can a LOOP hold before every iteration pays the instruction fetch again?
// One wave per SIMD on every CU (the occupancy of the env kernels), each running R iterations of a loop whose body is
// KB kilobytes of independent 8-byte VALU instructions (v_fma_f32 in VOP3 encoding).  Prints cycles per instruction for
// the first iteration (cold) and for the steady state -- 4 cycles is the issue rate; anything above it is instruction fetch.
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/icache_loop tools/ubench/icache_loop.hip && tools/ubench/icache_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// 1 KB of code: 128 VOP3 instructions over four independent registers
#define I4 "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
#define R2(x) x x
#define R4(x) R2(R2(x))
#define R32(x) R4(R4(R2(x)))
// (spelled out rather than .rept: the compiler sizes its branches by the lines of an asm statement)
#define KB1(a, b, c, d) asm volatile(R32(I4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k))

template <int KB>
__global__ __launch_bounds__(256) void k_loop(float* out, long long* cyc, int R) {
    float a = threadIdx.x, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    const float k = 0.999f;
    long long t0 = 0, t1 = 0, t2 = 0;
    for (int r = 0; r < R; ++r) {
        if (r == 0) t0 = (long long)__builtin_amdgcn_s_memtime();
        if (r == 1) t1 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < KB; ++i) KB1(a, b, c, d);
    }
    t2 = (long long)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[2 * w] = t1 - t0;
        cyc[2 * w + 1] = R > 1 ? (t2 - t1) / (R - 1) : 0;
    }
}

template <int KB> int run(float* out, long long* cyc, int blocks) {
    const int R = 40;
    for (int rep = 0; rep < 2; ++rep) {   // the second launch is the one reported (module already loaded)
        hipLaunchKernelGGL(k_loop<KB>, dim3(blocks), dim3(256), 0, 0, out, cyc, R);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<long long> h(2 * blocks * 4);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cold, warm;
    for (int w = 0; w < blocks * 4; ++w) { cold.push_back((double)h[2 * w]); warm.push_back((double)h[2 * w + 1]); }
    std::sort(cold.begin(), cold.end()); std::sort(warm.begin(), warm.end());
    const double n_instr = 128.0 * KB;
    printf("body %3d KB (%5.0f instr): first iteration %6.2f cycles/instr (median; %7.0f cycles), later iterations %5.2f (median) %5.2f (max)\n",
           KB, n_instr, cold[cold.size() / 2] / n_instr, cold[cold.size() / 2], warm[warm.size() / 2] / n_instr, warm.back() / n_instr);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount;
    float* out; long long* cyc;
    CHECK(hipMalloc(&out, blocks * 256 * 4)); CHECK(hipMalloc(&cyc, blocks * 4 * 2 * 8));
    printf("%d CUs, one 256-thread workgroup each (one wave per SIMD); s_memtime cycles\n", blocks);
    if (run<8>(out, cyc, blocks)) return 1;
    if (run<16>(out, cyc, blocks)) return 1;
    if (run<24>(out, cyc, blocks)) return 1;
    if (run<32>(out, cyc, blocks)) return 1;
    if (run<48>(out, cyc, blocks)) return 1;
    if (run<56>(out, cyc, blocks)) return 1;
    if (run<64>(out, cyc, blocks)) return 1;
    if (run<72>(out, cyc, blocks)) return 1;
    if (run<96>(out, cyc, blocks)) return 1;
    if (run<128>(out, cyc, blocks)) return 1;
    if (run<192>(out, cyc, blocks)) return 1;
    return 0;
}
