// GPU box: raw buffer stores / loads on gfx950 -- is a lane with an out-of-range offset dropped (store) / zero (load),
// does the range check see soffset, and which descriptor word works?  (Findings: DESIGN.md, k_mlp_bwd / k_wgrad.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_store(float* out, int n_valid, int n, unsigned flags, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * 4, flags);
    unsigned off = threadIdx.x < (unsigned)n_valid ? threadIdx.x * 4u : 0x80000000u;
    // b32 = an integer operand: bit-cast, or the float is CONVERTED
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(1.0f + threadIdx.x), r, off, soff, 0);
}
__global__ void k_load(const float* in, float* out, int n_valid, int n, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    unsigned off = threadIdx.x < (unsigned)n_valid ? threadIdx.x * 4u : 0x80000000u;
    out[threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
}
int main() {
    float *d, *o; hipMalloc(&d, 4096 * 4); hipMalloc(&o, 64 * 4);
    float h[4096];
    for (unsigned f : {0x00020000u, 0u}) for (unsigned soff : {0u, 256u, 4000u, 8192u}) {
        hipMemset(d, 0, 4096 * 4);
        hipLaunchKernelGGL(k_store, dim3(1), dim3(64), 0, 0, d, 40, 1024, f, soff);  // 1024 records: bytes 0..4095
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int nz = 0, first = -1, last = -1; for (int i = 0; i < 4096; ++i) if (h[i] != 0) { ++nz; if (first < 0) first = i; last = i; }
        printf("store flags %08x soff %u: err %d nonzero %d first %d last %d h[first]=%g\n", f, soff, (int)e, nz, first, last, first >= 0 ? h[first] : 0.f);
    }
    for (int i = 0; i < 4096; ++i) h[i] = 100.f + i;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (unsigned soff : {0u, 256u, 4000u, 8192u}) {
        hipLaunchKernelGGL(k_load, dim3(1), dim3(64), 0, 0, d, o, 40, 1024, soff);
        hipDeviceSynchronize();
        float g[64]; hipMemcpy(g, o, sizeof(g), hipMemcpyDeviceToHost);
        printf("load soff %u: lane0 %g lane23 %g lane24 %g lane39 %g lane40 %g lane63 %g\n", soff, g[0], g[23], g[24], g[39], g[40], g[63]);
    }
    return 0;
}
