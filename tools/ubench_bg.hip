// Background memory traffic for an experiment: NB workgroups stream through a buffer for `seconds`, at a rate set by
// `sleep` (s_sleep units between 16-byte loads).  Does keeping the memory system busy change the latency kernel's hand-off?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_bg.hip -o tools/ubench_bg.bin ; run: tools/ubench_bg.bin <workgroups> <seconds> <sleep>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__global__ void k_bg(const float4 *buf, size_t n, float *out, uint64_t ticks, int sleep_units)
{
    const uint64_t t0 = wall_clock64();
    float acc = 0.0f;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        for (int r = 0; r < 64; ++r) {
            const float4 v = buf[i % n];
            acc += v.x + v.w;
            i += (size_t)gridDim.x * blockDim.x;
            if (sleep_units) __builtin_amdgcn_s_sleep(8);
        }
    }
    if (acc == 1.2345f) out[0] = acc;
}
int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 4;
    const double sec = argc > 2 ? atof(argv[2]) : 10.0;
    const int sl = argc > 3 ? atoi(argv[3]) : 0;
    const size_t n = (size_t)1 << 26;              // 1 GiB of float4
    float4 *buf; float *out;
    (void)hipMalloc(&buf, n * sizeof(float4)); (void)hipMalloc(&out, 64);
    (void)hipMemset(buf, 0, n * sizeof(float4));
    hipLaunchKernelGGL(k_bg, nb, 256, 0, 0, buf, n, out, (uint64_t)(sec * 1e8), sl);
    (void)hipDeviceSynchronize();
    printf("background done\n");
    return 0;
}
