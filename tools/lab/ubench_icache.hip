// Straight-line code executed ONCE per launch: how fast does a lone wave get through it, launch after launch, and does a wave that
// ran through the same code earlier in the same workgroup (warming the instruction cache) change that?
// Build: hipcc --offload-arch=gfx950 -O3 ubench_icache.hip -o ubench_icache.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
#define R256(x) R64(x) R64(x) R64(x) R64(x)
#define R1024(x) R256(x) R256(x) R256(x) R256(x)

__device__ __noinline__ float block(float v, float a, float b)
{
    // 2048 dependent FMAs, 8 bytes each: 16 KB of straight-line code
    asm volatile(R1024("v_fma_f32 %0, %0, %1, %2\n") R1024("v_fma_f32 %0, %0, %1, %2\n") : "+v"(v) : "v"(a), "v"(b));
    return v;
}

// mode 0: wave 0 runs the block once.  mode 1: wave 1 runs it first (result discarded), then wave 0.  mode 2: wave 0 runs it twice, second timed
__global__ void k(float *out, uint64_t *cyc, int mode, float a, float b)
{
    __shared__ int flag;
    const int wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    float v = threadIdx.x;
    if (mode == 1 && wid == 1) { v = block(v, a, b); out[threadIdx.x] = v; }
    __syncthreads();
    if (wid == 0) {
        if (mode == 2) v = block(v, a, b);
        const uint64_t c0 = __builtin_readcyclecounter();
        v = block(v, a, b);
        const uint64_t c1 = __builtin_readcyclecounter();
        out[threadIdx.x] = v;
        if (threadIdx.x == 0) cyc[0] = c1 - c0;
    }
}

int main()
{
    float *out; uint64_t *cyc, hc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64);
    for (int mode = 0; mode < 3; ++mode) {
        printf("mode %d (%s):", mode, mode == 0 ? "lone wave, once" : mode == 1 ? "another wave of the workgroup ran it first" : "same wave ran it just before");
        for (int rep = 0; rep < 6; ++rep) {
            hipLaunchKernelGGL(k, 1, 128, 0, 0, out, cyc, mode, 1.0001f, 0.5f);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            printf("  %.2f", (double)hc / 2048.0);
        }
        printf("  cycles/instr per launch\n");
    }
    return 0;
}
