// One LDS image of W2 (128 x 128, 16-bit elements, 256-byte rows) serving BOTH operand reads of the split-bf16 backward kernel:
//   forward  h2 = W2 h1      : A fragment of lane (o = 32 t + col, half), k-step (ti, s) = W2[o][32 ti + 16 s + 8 (j>>2) + 4 half + (j&3)],
//                              j = 0..7 (the k order of an accumulator tile used as the B operand) -- ONE ds_read_b128
//   backward dh1 = W2^T dz2  : A fragment of lane (i = 32 t + col, half), k-step (to, s) = W2[32 to + 16 s + 8 (j>>2) + 4 half + (j&3)][i]
//                              -- TWO ds_read_b64_tr_b16 (rows .. + 4 half + 0..3 and + 8)
// Layout: logical column i of a row is stored at column sigma(i) = i with bits 2 and 3 swapped (so that the eight forward
// elements are one 16-byte chunk), and 16-byte chunk ch of row r lives at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3)))
// (cdna_hip_programming.md T10, image (b): conflict-free for both kinds of read -- checked by brute force against the bank
// rules of MI355X_MICROARCH.md, and on the hardware here with exact integer data).
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dual_image tools/ubench/dual_image.hip && tools/ubench/dual_image
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __host__ inline int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __host__ inline int sigma(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

__global__ __launch_bounds__(64) void k(short* fwd, short* bwd) {
    __shared__ __attribute__((aligned(16))) char img[128 * 256];
    for (int idx = threadIdx.x; idx < 128 * 128; idx += 64) {
        const int o = idx >> 7, i = idx & 127, sc = sigma(i);
        *reinterpret_cast<short*>(img + img_off(o, sc >> 3) + 2 * (sc & 7)) = (short)(o * 128 + i);
    }
    __syncthreads();
    const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
    for (int t = 0; t < 4; ++t)
        for (int ti = 0; ti < 4; ++ti)
            for (int s = 0; s < 2; ++s) {
                const uint4 v = *reinterpret_cast<const uint4*>(img + img_off(32 * t + col, 2 * (2 * ti + s) + half));
                short* dst = fwd + ((((t * 4 + ti) * 2 + s) * 64 + lane) * 8);
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                for (int j = 0; j < 8; ++j) dst[j] = (short)((w[j >> 1] >> (16 * (j & 1))) & 0xffff);
            }
    const int g16 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    for (int t = 0; t < 4; ++t)
        for (int to = 0; to < 4; ++to)
            for (int s = 0; s < 2; ++s)
                for (int jg = 0; jg < 2; ++jg) {
                    const int row = 32 * to + 16 * s + 8 * jg + 4 * half + q;
                    const int addr = img_off(row, 2 * (2 * t + g16) + (p & 1)) + 8 * (p >> 1);
                    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(img + addr));
                    short* dst = bwd + (((((t * 4 + to) * 2 + s) * 2 + jg) * 64 + lane) * 4);
                    for (int e = 0; e < 4; ++e) dst[e] = r[e];
                }
}

int main() {
    short *fwd, *bwd;
    const int nf = 4 * 4 * 2 * 64 * 8, nb = 4 * 4 * 2 * 2 * 64 * 4;
    CHECK(hipMalloc(&fwd, nf * 2)); CHECK(hipMalloc(&bwd, nb * 2));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, fwd, bwd);
    CHECK(hipDeviceSynchronize());
    std::vector<short> hf(nf), hb(nb);
    CHECK(hipMemcpy(hf.data(), fwd, nf * 2, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb.data(), bwd, nb * 2, hipMemcpyDeviceToHost));
    int bad_f = 0, bad_b = 0;
    for (int t = 0; t < 4; ++t) for (int ti = 0; ti < 4; ++ti) for (int s = 0; s < 2; ++s) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
        const int col = lane & 31, half = lane >> 5, o = 32 * t + col, i = 32 * ti + 16 * s + 8 * (j >> 2) + 4 * half + (j & 3);
        if (hf[((((t * 4 + ti) * 2 + s) * 64 + lane) * 8) + j] != (short)(o * 128 + i)) ++bad_f;
    }
    for (int t = 0; t < 4; ++t) for (int to = 0; to < 4; ++to) for (int s = 0; s < 2; ++s) for (int jg = 0; jg < 2; ++jg) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
        const int col = lane & 31, half = lane >> 5, i = 32 * t + col, o = 32 * to + 16 * s + 8 * jg + 4 * half + e;
        if (hb[(((((t * 4 + to) * 2 + s) * 2 + jg) * 64 + lane) * 4) + e] != (short)(o * 128 + i)) ++bad_b;
    }
    printf("forward row reads: %d mismatches of %d;  transposed reads: %d mismatches of %d\n", bad_f, nf, bad_b, nb);
    return (bad_f || bad_b) ? 1 : 0;
}
