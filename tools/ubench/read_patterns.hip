// read_patterns.hip -- which read pattern of a [800 features] x [N samples] fp32 tensor does the chip stream fastest?  (round 5:
// k_dense_tail_v3 reads 3.1 TB/s of a tensor that torch's sum(0) streams at 5.7 TB/s.)  Every kernel has the dense tail's
// work distribution -- 512-thread workgroups, a wave per 32-sample tile, 64-k chunks with a workgroup barrier per chunk, two
// chunks' loads in flight -- and only sums what it reads.
//   A: feature-major x[k][e] (the shipped layout): lane (sample e, half) reads 8 rows k of 4 bytes per k-step (dword loads)
//   B: sample-major x[e][k] (torch's [batch, features]): lane (sample e, half) reads 32 contiguous bytes per k-step (2 x dwordx4)
//   C: sample-major, a wave reads its 32 samples' 256-byte pieces of a chunk as 16 fully coalesced 512-byte half-rows
//      (lane = 16 bytes of a row: 16 lanes per sample-row piece, 4 samples per instruction)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/read_patterns tools/ubench/read_patterns.hip && tools/ubench/read_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int K = 800, KC = 64;

template <int MODE>
__global__ __launch_bounds__(512) void k_read(const float* __restrict__ x, int64_t N, float* out) {
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const int64_t n_tiles = N / 32;
    float acc = 0.f;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t e = tile * 32 + col;
        for (int kc = 0; kc < K; kc += KC) {
            if (MODE == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int k = kc + 16 * s + 8 * half + i;
                        if (k < K) acc += x[(int64_t)k * N + e];
                    }
            } else if (MODE == 1) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = kc + 16 * s + 8 * half;
                    if (k < K) {
                        const float4 a = *reinterpret_cast<const float4*>(x + e * K + k), b = *reinterpret_cast<const float4*>(x + e * K + k + 4);
                        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
                    }
                }
            } else {
                // 32 samples x 256 bytes = 8 KiB per chunk and wave = 8 instructions of 64 lanes x 16 bytes: instruction j covers samples
                // 4 j .. 4 j + 3, lane = (sample 4 j + (lane >> 4), bytes 16 (lane & 15) of the 256-byte piece)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t es = tile * 32 + 4 * j + (lane >> 4);
                    const int k = kc + 4 * (lane & 15);
                    if (k < K) {
                        const float4 a = *reinterpret_cast<const float4*>(x + es * K + k);
                        acc += a.x + a.y + a.z + a.w;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    for (int64_t N : {131072ll, 1048576ll}) {
        float *x, *out;
        CK(hipMalloc(&x, sizeof(float) * K * N)); CK(hipMalloc(&out, 64));
        CK(hipMemset(x, 0, sizeof(float) * K * N));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const char* names[3] = {"A feature-major dword", "B sample-major 2 x dwordx4 per lane", "C sample-major coalesced 16 B x 64 lanes"};
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(256), dim3(512), 0, 0, x, N, out);
                else if (mode == 1) hipLaunchKernelGGL(k_read<1>, dim3(256), dim3(512), 0, 0, x, N, out);
                else hipLaunchKernelGGL(k_read<2>, dim3(256), dim3(512), 0, 0, x, N, out);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("N %8lld  %-45s %8.1f us  %.2f TB/s\n", (long long)N, names[mode], ms * 100, sizeof(float) * K * N / (ms / 10 * 1e-3) / 1e12);
        }
        CK(hipFree(x)); CK(hipFree(out));
    }
    return 0;
}
