// Consumer B's per-slot work in isolation (one wavefront alone on a CU): the stage cost of a slot -- distance to the goal with the
// correctly rounded square root, the stuck threshold, the fp64 sum in step order -- four slots per look at the LDS ring, as
// rollout_lat.inc's BN_LOOK does it, with parts removed.  In the product this wave runs at ~85 ns a slot; what of it
// is arithmetic, what the look's LDS round trip, what the fp64 chain?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_stagecost.hip -o tools/ubench_stagecost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float sqrt_cr_normal(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = rm <= 0.0f ? sm : s;
    s = rp > 0.0f ? sp : s;
    return s;
}

// MODE bits: 1 no square-root correction (v_sqrt_f32 alone), 2 no fp64 (float sum), 4 no threshold term, 8 slots from registers (no LDS look),
//            16 eight slots per look
template <int MODE>
__global__ void k_b(float *out, uint64_t *cyc, int nslots, float gx, float gy, float thr)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                 // 64 slots x 64 lanes x float4
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 256; i += blockDim.x) ring[i] = 3.0f + 0.37f * (i % 113) + 0.001f * i;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    constexpr int C = (MODE & 16) ? 8 : 4;
    double Sd = 0.0, Sp = 0.0;
    float Sf = 0.0f, term = 0.0f;
    const float4 *slot = reinterpret_cast<const float4 *>(ring) + lane;
    float4 held[C];
#pragma unroll
    for (int i = 0; i < C; ++i) held[i] = slot[64 * i];
    const uint64_t w0 = wall_clock64();
    const uint64_t c0 = __builtin_readcyclecounter();
    for (int t = 0; t < nslots; t += C) {
        float4 rq[C];
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < C; ++i) { rq[i] = held[i]; rq[i].x += 1e-3f * (float)t; asm volatile("" : "+v"(rq[i].x)); }
        } else {
#pragma unroll
            for (int i = 0; i < C; ++i) rq[i] = slot[64 * ((t + i) & 63)];
            asm volatile("" :: "v"(rq[0].z), "v"(rq[1].z), "v"(rq[2].z), "v"(rq[3].z));
        }
        float sc[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const float dx = rq[i].x - gx, dy = rq[i].y - gy;
            const float d2 = dx * dx + dy * dy;
            sc[i] = ((MODE & 1) ? __builtin_amdgcn_sqrtf(d2) : sqrt_cr_normal(d2)) + ((MODE & 4) ? 0.0f : (rq[i].w <= thr ? 1.0e4f : 0.0f));
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            term = sc[i];
            if (MODE & 2) Sf += term; else { Sp = Sd; Sd = Sp + (double)term; }
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    out[lane] = (float)Sp + term + Sf;
    if (lane == 0) { cyc[0] = c1 - c0; cyc[1] = wall_clock64() - w0; }
}

int main()
{
    float *out; uint64_t *cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64);
    const int n = 8000;
    const size_t lds = 64 * 256 * 4;
    uint64_t hc[2];
#define RUN(MODE, label) do { (void)hipFuncSetAttribute((const void *)k_b<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k_b<MODE>, 1, 64, lds, 0, out, cyc, n, 40.0f, 50.0f, 0.3f); (void)hipDeviceSynchronize(); \
        hipLaunchKernelGGL(k_b<MODE>, 1, 64, lds, 0, out, cyc, n, 40.0f, 50.0f, 0.3f); (void)hipDeviceSynchronize(); \
        (void)hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost); \
        printf("%-66s %6.1f cycles/slot  %5.1f ns/slot\n", label, (double)hc[0] / n, (double)hc[1] * 10.0 / n); } while (0)
    RUN(0, "consumer B's slot as in the product (4 slots per look)");
    RUN(16, "eight slots per look");
    RUN(8, "slots from registers (no LDS look)");
    RUN(1, "v_sqrt_f32 without the correction");
    RUN(2, "float sum instead of the fp64 chain");
    RUN(4, "no threshold term");
    RUN(1 | 2 | 4, "distance + v_sqrt_f32 + float add only");
    RUN(1 | 2 | 4 | 8, "the same from registers");
    RUN(2 | 16, "float sum, eight slots per look");
    return 0;
}
