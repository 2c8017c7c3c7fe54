// Single-wave issue / latency microbenchmarks for gfx950 (guides the rollout kernel design).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NCHAIN>
__global__ void fma_chain(float *out, uint64_t *cyc, uint64_t *wall, int n, float a, float b)
{
    float v[NCHAIN];
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) v[i] = threadIdx.x * 1e-3f + i;
    uint64_t w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < NCHAIN; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        }
    }
    uint64_t c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0; for (int i = 0; i < NCHAIN; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
}

__global__ void lds_chain(float *out, uint64_t *cyc, int n)
{
    __shared__ int tab[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = (i * 7 + 13) & 1023;
    __syncthreads();
    int idx = threadIdx.x & 1023;
    uint64_t c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) idx = tab[idx];
    }
    uint64_t c1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

// mixed dependent chain resembling a transit step: mul, add, floor, cvt, min/max
__global__ void mixed_chain(float *out, uint64_t *cyc, int n, float a)
{
    float x = threadIdx.x * 0.01f;
    uint64_t c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float q = floorf(x * a);
            int i = (int)q;
            i = min(max(i, 0), 255);
            x = x + (float)i * 1e-3f;
            x = fminf(fmaxf(x, 0.f), 100.f);
        }
    }
    uint64_t c1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

int main()
{
    float *out; uint64_t *cyc, *wall;
    CHECK(hipMalloc(&out, 1 << 22)); CHECK(hipMalloc(&cyc, 8 * 4096)); CHECK(hipMalloc(&wall, 8 * 4096));
    uint64_t hc[4096], hw[4096];
    const int n = 20000;
    auto report = [&](const char *name, int ninstr_per_iter, int blocks) {
        hipDeviceSynchronize();
        hipMemcpy(hc, cyc, 8 * blocks, hipMemcpyDeviceToHost); hipMemcpy(hw, wall, 8 * blocks, hipMemcpyDeviceToHost);
        printf("%-34s blocks=%4d  cycles/instr=%6.2f  (cycles=%llu wall100MHz=%llu -> %.0f MHz)\n", name, blocks,
               (double)hc[0] / ((double)n * ninstr_per_iter), (unsigned long long)hc[0], (unsigned long long)hw[0],
               hw[0] ? (double)hc[0] / (double)hw[0] * 100.0 : 0.0);
    };
    for (int blocks : {1, 16, 1024}) {
        for (int threads : {64, 256}) {
            printf("--- %d threads/block\n", threads);
            hipLaunchKernelGGL(fma_chain<1>, blocks, threads, 0, 0, out, cyc, wall, n, 1.0001f, 0.5f); report("fma dependent x1", 16, blocks);
            hipLaunchKernelGGL(fma_chain<2>, blocks, threads, 0, 0, out, cyc, wall, n, 1.0001f, 0.5f); report("fma 2 chains", 32, blocks);
            hipLaunchKernelGGL(fma_chain<4>, blocks, threads, 0, 0, out, cyc, wall, n, 1.0001f, 0.5f); report("fma 4 chains", 64, blocks);
            hipLaunchKernelGGL(fma_chain<8>, blocks, threads, 0, 0, out, cyc, wall, n, 1.0001f, 0.5f); report("fma 8 chains", 128, blocks);
        }
    }
    hipLaunchKernelGGL(lds_chain, 1, 64, 0, 0, out, cyc, n); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds dependent read: %.1f cycles/read\n", (double)hc[0] / (n * 16.0));
    hipLaunchKernelGGL(mixed_chain, 1, 64, 0, 0, out, cyc, n, 2.0f); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("mixed chain (10 dependent ops incl floor/cvt/minmax): %.1f cycles/iter\n", (double)hc[0] / (n * 4.0));
    // short-kernel clock: time 2000 tiny launches
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(fma_chain<1>, 16, 64, 0, 0, out, cyc, wall, 100, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(hw, wall, 8, hipMemcpyDeviceToHost);
    printf("short kernels (1600 fma): %.2f us/launch, in-kernel %llu cycles, %.0f MHz\n", ms * 1e3 / 200, (unsigned long long)hc[0], (double)hc[0] / hw[0] * 100.0);
    return 0;
}
