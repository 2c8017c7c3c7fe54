// When do the wavefronts of ONE workgroup start?  Every wave stamps the shader clock at entry; printed relative to
// the workgroup's first wave, for one workgroup per CU (128 workgroups) and several shapes.  gfx950, hipcc -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
__global__ void stamp(uint64_t *out, int spin)
{
    extern __shared__ float sm[];
    const uint64_t c0 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = c0;
    if (threadIdx.x == 0) sm[0] = 1.f;
    while ((int64_t)(__builtin_readcyclecounter() - c0) < spin) __builtin_amdgcn_s_sleep(8);
}
int main()
{
    uint64_t *out; hipMalloc(&out, 4096 * 16 * 8);
    hipFuncSetAttribute((const void *)stamp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Cfg { int threads, lds, blocks; } cfgs[] = {{1024, 112640, 128}, {512, 112640, 128}, {1024, 0, 128}, {1024, 0, 1}, {320, 39552, 1024}, {1024, 112640, 256}};
    for (auto c : cfgs) {
        const int nw = c.threads / 64;
        std::vector<uint64_t> h(c.blocks * 16);
        std::vector<double> acc(nw, 0.0), mx(nw, 0.0);
        for (int rep = 0; rep < 6; ++rep) {
            hipMemset(out, 0, c.blocks * 16 * 8);
            hipLaunchKernelGGL(stamp, dim3(c.blocks), dim3(c.threads), c.lds, 0, out, 24000);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, c.blocks * 16 * 8, hipMemcpyDeviceToHost);
            if (rep == 0) continue;
            for (int b = 0; b < c.blocks; ++b) {
                uint64_t first = ~0ull;
                for (int w = 0; w < nw; ++w) first = std::min(first, h[b * 16 + w]);
                std::vector<uint64_t> d;
                for (int w = 0; w < nw; ++w) d.push_back(h[b * 16 + w] - first);
                std::sort(d.begin(), d.end());
                for (int w = 0; w < nw; ++w) { acc[w] += d[w] / 2400.0 / (5.0 * c.blocks); mx[w] = std::max(mx[w], d[w] / 2400.0); }
            }
        }
        printf("threads=%4d lds=%6d blocks=%4d  k-th wave start after the first (us, mean over workgroups):", c.threads, c.lds, c.blocks);
        for (int w = 0; w < nw; ++w) printf(" %.2f", acc[w]);
        printf("   | last wave worst case %.2f\n", mx[nw - 1]);
    }
    return 0;
}
