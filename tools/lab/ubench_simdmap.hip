// Which SIMD does each wavefront of a 5-wave workgroup land on when four such workgroups share a CU (the role kernel's
// residency: 320 threads, 38.6 KB of LDS)?  Every wave records HW_ID (SIMD_ID bits 5:4, CU_ID 11:8, SH 12, SE 15:13) and XCC_ID.
// Printed: for each wave index of the workgroup, the histogram of SIMD ids; and per CU the number of "wave 1"s per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <array>
__global__ void probe(uint32_t *out, int spin)
{
    extern __shared__ float sm[];
    const uint64_t c0 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (threadIdx.x == 0) sm[0] = 1.f;
    while ((int64_t)(__builtin_readcyclecounter() - c0) < spin) __builtin_amdgcn_s_sleep(8);
}
int main()
{
    uint32_t *out; hipMalloc(&out, 4096 * 16 * 4);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Cfg { int threads, lds, blocks; } cfgs[] = {{320, 39552, 1024}, {320, 39552, 17}, {320, 83968, 17}, {256, 39552, 1024}};
    for (auto c : cfgs) {
        const int nw = c.threads / 64;
        std::vector<uint32_t> h(c.blocks * 16);
        hipMemset(out, 0, c.blocks * 16 * 4);
        hipLaunchKernelGGL(probe, dim3(c.blocks), dim3(c.threads), c.lds, 0, out, 60000);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe, dim3(c.blocks), dim3(c.threads), c.lds, 0, out, 60000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, c.blocks * 16 * 4, hipMemcpyDeviceToHost);
        printf("threads=%d lds=%d blocks=%d\n", c.threads, c.lds, c.blocks);
        std::map<uint32_t, std::array<int, 4>> per_cu;     // CU key -> count of wave-1s per SIMD
        std::map<uint32_t, int> wg_per_cu;
        for (int w = 0; w < nw; ++w) {
            int hist[4] = {0, 0, 0, 0};
            for (int b = 0; b < c.blocks; ++b) {
                const uint32_t hw = h[(b * 8 + w) * 2], xcc = h[(b * 8 + w) * 2 + 1] & 0xf;
                const int simd = (hw >> 4) & 3;
                hist[simd]++;
                const uint32_t key = (xcc << 16) | (hw & 0xff00);
                if (w == 1) per_cu[key][simd]++;
                if (w == 0) wg_per_cu[key]++;
            }
            printf("  wave %d of the workgroup: SIMD0 %4d  SIMD1 %4d  SIMD2 %4d  SIMD3 %4d\n", w, hist[0], hist[1], hist[2], hist[3]);
        }
        int worst[5] = {0, 0, 0, 0, 0};
        for (auto &kv : per_cu) { int m = 0; for (int s = 0; s < 4; ++s) m = kv.second[s] > m ? kv.second[s] : m; worst[m > 4 ? 4 : m]++; }
        printf("  CUs used %zu; CUs by max number of wave-1s on one SIMD: 1:%d 2:%d 3:%d 4+:%d\n", per_cu.size(), worst[1], worst[2], worst[3], worst[4]);
        // same workgroup: do waves 0 and 4 share a SIMD?
        int share = 0;
        if (nw == 5) { for (int b = 0; b < c.blocks; ++b) share += (((h[(b * 8) * 2] >> 4) & 3) == ((h[(b * 8 + 4) * 2] >> 4) & 3)); printf("  workgroups whose waves 0 and 4 share a SIMD: %d of %d\n", share, c.blocks); }
    }
    return 0;
}
