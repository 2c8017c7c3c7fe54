// Cycles per Philox4x32-10 + 2 Box-Muller block (the library's own device functions) for 1..16 waves on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../benchnav_amd/csrc/bn_device_math.h"
__global__ void k(float *out, uint64_t *cyc, int n)
{
    float acc = 0.f;
    __syncthreads();
    const uint64_t c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        float z[4];
        bn::philox_slip_block(42, 7, 0, threadIdx.x, (uint32_t)i, z);
        acc += z[0] + z[1] + z[2] + z[3];
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = c1 - c0;
}
__global__ void k_nobm(float *out, uint64_t *cyc, int n)
{
    uint32_t acc = 0;
    __syncthreads();
    const uint64_t c0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        const bn::u32x4 r = bn::philox4x32_10(bn::u32x4{threadIdx.x, (uint32_t)i, 7u, 0u}, 42u, 0u);
        acc += r.x ^ r.y ^ r.z ^ r.w;
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)acc;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = c1 - c0;
}
int main()
{
    float *out; uint64_t *cyc; hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 1024);
    const int n = 2000; uint64_t hc[16];
    for (int w : {1, 2, 4, 8, 16}) {
        hipLaunchKernelGGL(k, 1, 64 * w, 0, 0, out, cyc, n); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 8 * w, hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < w; ++i) mx = hc[i] > mx ? hc[i] : mx;
        hipLaunchKernelGGL(k_nobm, 1, 64 * w, 0, 0, out, cyc, n); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 8 * w, hipMemcpyDeviceToHost);
        double mx2 = 0; for (int i = 0; i < w; ++i) mx2 = hc[i] > mx2 ? hc[i] : mx2;
        printf("waves=%2d  philox+2BM: %7.1f cycles per block per wave (%6.1f per block per SIMD)   philox only: %7.1f (%6.1f)\n", w, mx / n, mx / n / (w > 4 ? w / 4.0 : 1.0) , mx2 / n, mx2 / n / (w > 4 ? w / 4.0 : 1.0));
    }
    return 0;
}
