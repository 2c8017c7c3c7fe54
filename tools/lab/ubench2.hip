// Per-instruction issue cost of a single wavefront on gfx950: N independent or dependent copies of one
// instruction (inline asm, so the compiler neither packs nor reorders), cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL_IND(name, INSTR)                                                                                \
__global__ void name(float *out, uint64_t *cyc, int n, float a, float b)                                     \
{                                                                                                            \
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
    int i0 = threadIdx.x;                                                                                    \
    uint64_t c0 = __builtin_readcyclecounter();                                                              \
    for (int it = 0; it < n; ++it) {                                                                         \
        asm volatile(REP4(INSTR) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(i0) : "v"(a), "v"(b) : "vcc"); \
    }                                                                                                        \
    uint64_t c1 = __builtin_readcyclecounter();                                                              \
    out[threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + i0;                                           \
    if (threadIdx.x == 0) cyc[0] = c1 - c0;                                                                  \
}

// 8 independent destinations per group; 4 groups per loop iteration = 32 instructions
KERNEL_IND(k_fma_ind,   "v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %9, %10\n v_fma_f32 %3, %3, %9, %10\n v_fma_f32 %4, %4, %9, %10\n v_fma_f32 %5, %5, %9, %10\n v_fma_f32 %6, %6, %9, %10\n v_fma_f32 %7, %7, %9, %10\n")
KERNEL_IND(k_fma_dep,   "v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %0, %0, %9, %10\n")
KERNEL_IND(k_mul_ind,   "v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n")
KERNEL_IND(k_mul_dep,   "v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %0, %0, %9\n")
KERNEL_IND(k_2chain,    "v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n")
KERNEL_IND(k_3chain,    "v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n")
KERNEL_IND(k_med3_ind,  "v_med3_f32 %0, %0, %9, %10\n v_med3_f32 %1, %1, %9, %10\n v_med3_f32 %2, %2, %9, %10\n v_med3_f32 %3, %3, %9, %10\n v_med3_f32 %4, %4, %9, %10\n v_med3_f32 %5, %5, %9, %10\n v_med3_f32 %6, %6, %9, %10\n v_med3_f32 %7, %7, %9, %10\n")
KERNEL_IND(k_floor_ind, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n")
KERNEL_IND(k_floor_dep, "v_floor_f32 %0, %0\n v_floor_f32 %0, %0\n v_floor_f32 %0, %0\n v_floor_f32 %0, %0\n v_floor_f32 %0, %0\n v_floor_f32 %0, %0\n v_floor_f32 %0, %0\n v_floor_f32 %0, %0\n")
KERNEL_IND(k_cvt_ind,   "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n")
KERNEL_IND(k_rndne_dep, "v_rndne_f32 %0, %0\n v_rndne_f32 %0, %0\n v_rndne_f32 %0, %0\n v_rndne_f32 %0, %0\n v_rndne_f32 %0, %0\n v_rndne_f32 %0, %0\n v_rndne_f32 %0, %0\n v_rndne_f32 %0, %0\n")
KERNEL_IND(k_cnd_dep,   "v_cmp_gt_f32 vcc, %0, %9\n v_cndmask_b32 %0, %0, %10, vcc\n v_cmp_gt_f32 vcc, %0, %9\n v_cndmask_b32 %0, %0, %10, vcc\n v_cmp_gt_f32 vcc, %0, %9\n v_cndmask_b32 %0, %0, %10, vcc\n v_cmp_gt_f32 vcc, %0, %9\n v_cndmask_b32 %0, %0, %10, vcc\n")
KERNEL_IND(k_sqrt_ind,  "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
KERNEL_IND(k_mix_valu_salu, "v_mul_f32 %0, %0, %9\n s_nop 0\n v_mul_f32 %0, %0, %9\n s_nop 0\n v_mul_f32 %0, %0, %9\n s_nop 0\n v_mul_f32 %0, %0, %9\n s_nop 0\n")

__global__ void k_lds_after_valu(float *out, uint64_t *cyc, int n)
{
    __shared__ float tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (float)((i * 7 + 13) & 1023);
    __syncthreads();
    float x = threadIdx.x;
    uint64_t c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x = tab[((int)x) & 1023];   // cvt, and, lshl, ds_read: 3 VALU + LDS dependent
    }
    uint64_t c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}

int main()
{
    float *out; uint64_t *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    const int n = 20000;
    uint64_t hc;
#define RUN(k, per_iter) do { hipLaunchKernelGGL(k, 1, 64, 0, 0, out, cyc, n, 1.0001f, 0.5f); hipDeviceSynchronize(); \
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); printf("%-22s %6.2f cycles/instr\n", #k, (double)hc / ((double)n * (per_iter))); } while (0)
    RUN(k_fma_ind, 32); RUN(k_fma_dep, 32); RUN(k_mul_ind, 32); RUN(k_mul_dep, 32); RUN(k_2chain, 32); RUN(k_3chain, 32);
    RUN(k_med3_ind, 32); RUN(k_floor_ind, 32); RUN(k_floor_dep, 32); RUN(k_cvt_ind, 32); RUN(k_rndne_dep, 32);
    RUN(k_cnd_dep, 32); RUN(k_sqrt_ind, 32); RUN(k_mix_valu_salu, 16);
    hipLaunchKernelGGL(k_lds_after_valu, 1, 64, 0, 0, out, cyc, n); hipDeviceSynchronize(); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds gather chain (cvt+and+lshl+ds_read): %.1f cycles/iter\n", (double)hc / (n * 8.0));
    return 0;
}
