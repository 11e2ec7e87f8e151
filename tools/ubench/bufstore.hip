// GPU box: do raw buffer stores with an out-of-range offset get dropped, and which descriptor word works on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int n_valid, int n, unsigned flags, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * 4, flags);
    unsigned off = threadIdx.x < (unsigned)n_valid ? threadIdx.x * 4u : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(1.0f + threadIdx.x), r, off, soff, 0);  // integer operand: bit-cast, or the float is converted
}
int main() {
    float* d; hipMalloc(&d, 4096 * 4);
    unsigned flagsv[3] = {0x00020000u, 0x00027000u, 0u};
    for (unsigned f : flagsv) for (unsigned soff : {0u, 256u}) {
        hipMemset(d, 0, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 40, 1024, f, soff);
        hipError_t e = hipDeviceSynchronize();
        float h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int nz = 0, first = -1, last = -1; for (int i = 0; i < 4096; ++i) if (h[i] != 0) { ++nz; if (first < 0) first = i; last = i; }
        printf("flags %08x soff %u: err %d nonzero %d first %d last %d h[first]=%g\n", f, soff, (int)e, nz, first, last, first >= 0 ? h[first] : 0.f);
    }
    return 0;
}
