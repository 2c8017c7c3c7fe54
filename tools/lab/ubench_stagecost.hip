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

// The product's loop shape (rollout_lat.inc, consumer B): a progress word polled with the look, the slot count dispatched to
// straight-line code of its own per count, a pointer advanced per look.  `step` = how far the "chain" is ahead when the loop starts:
// prog is preset to nslots + 1 (everything there: n >= 4 at every look).  SHAPE 0: as the product; 1: n >= 4 in a loop of its own
// (one backward branch), the dispatch only behind it; 2: as 1 with the next look's reads issued before this look's arithmetic.
template <int SHAPE>
__global__ void k_prod(float *out, uint64_t *cyc, int T, float gx, float gy, float thr, int reps)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                 // (T + 1) slots x 64 lanes x float4, then the progress word
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < (T + 9) * 256; i += blockDim.x) ring[i] = 3.0f + 0.37f * (i % 113) + 0.001f * i;
    int *prog = reinterpret_cast<int *>(ring + (size_t)(T + 9) * 256);
    if (threadIdx.x == 0) prog[0] = T + 1;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    typedef __attribute__((address_space(3))) volatile int lds_vint;
    lds_vint *cprog = (lds_vint *)prog;
    float acc = 0.0f;
    const uint64_t w0 = wall_clock64();
    const uint64_t c0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        double Sd = 0.0, Sp = 0.0;
        float term = 0.0f;
        int t = 0;
#define BN_LOOK(O, C)                                                                                          \
    do {                                                                                                       \
        float sc[C];                                                                                           \
        _Pragma("unroll") for (int i = 0; i < (C); ++i) {                                                      \
            const float dx = rq[(O) + i].x - gx, dy = rq[(O) + i].y - gy;                                      \
            sc[i] = sqrt_cr_normal(dx * dx + dy * dy) + (rq[(O) + i].w <= thr ? 1.0e4f : 0.0f);               \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < (C); ++i) { term = sc[i]; Sp = Sd; Sd = Sp + (double)term; }     \
    } while (0)
        const float4 *slot = reinterpret_cast<const float4 *>(ring) + lane;
        if (SHAPE == 0) {
            while (t <= T) {
                const int pr = *cprog;
                asm volatile("" ::: "memory");
                float4 rq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) rq[i] = slot[64 * i];
                asm volatile("" :: "v"(rq[0].z), "v"(rq[1].z), "v"(rq[2].z), "v"(rq[3].z));
                const int n = min(__builtin_amdgcn_readfirstlane(pr), T + 1) - t;
                if (n >= 4) { BN_LOOK(0, 4); t += 4; slot += 64 * 4; }
                else if (n == 1) { BN_LOOK(0, 1); t += 1; slot += 64; }
                else if (n == 2) { BN_LOOK(0, 2); t += 2; slot += 128; }
                else if (n == 3) { BN_LOOK(0, 3); t += 3; slot += 192; }
                else if (t + 8 < T) __builtin_amdgcn_s_sleep(2);
            }
        } else if (SHAPE == 1) {
            while (t <= T) {
                int pr = *cprog;
                asm volatile("" ::: "memory");
                float4 rq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) rq[i] = slot[64 * i];
                asm volatile("" :: "v"(rq[0].z), "v"(rq[1].z), "v"(rq[2].z), "v"(rq[3].z));
                int n = min(__builtin_amdgcn_readfirstlane(pr), T + 1) - t;
                while (n >= 4) {
                    BN_LOOK(0, 4); t += 4; slot += 64 * 4;
                    pr = *cprog;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 4; ++i) rq[i] = slot[64 * i];
                    asm volatile("" :: "v"(rq[0].z), "v"(rq[1].z), "v"(rq[2].z), "v"(rq[3].z));
                    n = min(__builtin_amdgcn_readfirstlane(pr), T + 1) - t;
                }
                if (n == 1) { BN_LOOK(0, 1); t += 1; slot += 64; }
                else if (n == 2) { BN_LOOK(0, 2); t += 2; slot += 128; }
                else if (n == 3) { BN_LOOK(0, 3); t += 3; slot += 192; }
                else if (t + 8 < T) __builtin_amdgcn_s_sleep(2);
            }
        } else {
            // the next look's reads go out before this look's arithmetic: their round trip lies under ~90 instructions
            int pr = *cprog;
            asm volatile("" ::: "memory");
            float4 rq[4], nx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rq[i] = slot[64 * i];
            asm volatile("" :: "v"(rq[0].z), "v"(rq[1].z), "v"(rq[2].z), "v"(rq[3].z));
            while (t <= T) {
                const int n = min(__builtin_amdgcn_readfirstlane(pr), T + 1) - t;
                const int take = n >= 4 ? 4 : n > 0 ? n : 0;
                const float4 *ns = slot + 64 * take;
                pr = *cprog;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) nx[i] = ns[64 * i];
                asm volatile("" :: "v"(nx[0].z), "v"(nx[1].z), "v"(nx[2].z), "v"(nx[3].z));
                if (n >= 4) BN_LOOK(0, 4);
                else if (n == 1) BN_LOOK(0, 1);
                else if (n == 2) BN_LOOK(0, 2);
                else if (n == 3) BN_LOOK(0, 3);
                else if (t + 8 < T) __builtin_amdgcn_s_sleep(2);
                t += take; slot = ns;
#pragma unroll
                for (int i = 0; i < 4; ++i) rq[i] = nx[i];
            }
        }
#undef BN_LOOK
        acc += (float)Sp + term;
        asm volatile("" : "+v"(acc));
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    out[lane] = acc;
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
    {
        const int T = 50, reps = 200;
        const size_t lds2 = (size_t)(T + 9) * 256 * 4 + 64;
#define RUNP(SHAPE, label) do { (void)hipFuncSetAttribute((const void *)k_prod<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        hipLaunchKernelGGL(k_prod<SHAPE>, 1, 320, lds2, 0, out, cyc, T, 40.0f, 50.0f, 0.3f, reps); (void)hipDeviceSynchronize(); \
        hipLaunchKernelGGL(k_prod<SHAPE>, 1, 320, lds2, 0, out, cyc, T, 40.0f, 50.0f, 0.3f, reps); (void)hipDeviceSynchronize(); \
        (void)hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost); \
        printf("%-66s %6.1f cycles/slot  %5.1f ns/slot\n", label, (double)hc[0] / (reps * (T + 1)), (double)hc[1] * 10.0 / (reps * (T + 1))); } while (0)
        RUNP(0, "the product's loop shape, all 51 slots there (T = 50)");
        RUNP(1, "n >= 4 in a loop of its own");
        RUNP(2, "next look's reads before this look's arithmetic");
    }
    return 0;
}
