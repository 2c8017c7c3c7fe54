// Workgroup dispatch rate on gfx950: N workgroups that each busy-wait ~10 us; time = dispatch ramp + 10 us.
// Varies the workgroup size and its LDS allocation.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void spin(float *out, int cycles)
{
    extern __shared__ float sm[];
    const uint64_t c0 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) sm[0] = 1.0f;
    while ((int64_t)(__builtin_readcyclecounter() - c0) < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && sm[0] < 0.f) out[blockIdx.x] = sm[0];
}
int main()
{
    float *out; hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Cfg { int threads; int lds; } cfgs[] = {{64, 3400}, {320, 39552}, {320, 0}, {64, 0}, {64, 39552}, {256, 39552}, {256, 0}, {1024, 0}, {640, 79104}, {320, 16384}, {320, 65536}};
    for (auto c : cfgs)
        for (int n : {1024, 4352, 6144, 8192}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(spin, dim3(n), dim3(c.threads), c.lds, 0, out, 24000 /* 10 us */);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("threads=%4d lds=%6d blocks=%5d : %7.2f us\n", c.threads, c.lds, n, best * 1e3f);
        }
    return 0;
}
