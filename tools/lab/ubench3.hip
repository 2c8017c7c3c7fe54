// Integer-multiply and transcendental issue cost on gfx950 (one wavefront), and VALU throughput of a SIMD shared by
// several wavefronts: what the Philox producers and the batched regime pay.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x

#define KERNEL8(name, I)                                                                                      \
__global__ void name(float *out, uint64_t *cyc, int n, float a, float b)                                     \
{                                                                                                            \
    float v0 = threadIdx.x + 1.5f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
    __syncthreads();                                                                                         \
    uint64_t c0 = __builtin_readcyclecounter();                                                              \
    for (int it = 0; it < n; ++it) {                                                                         \
        asm volatile(REP4(I(%0) I(%1) I(%2) I(%3) I(%4) I(%5) I(%6) I(%7))                                   \
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b) : "vcc"); \
    }                                                                                                        \
    uint64_t c1 = __builtin_readcyclecounter();                                                              \
    out[threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                                \
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = c1 - c0;                                            \
}
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8\n"
#define I_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8\n"
#define I_MULHI24(r) "v_mul_hi_u32_u24 " #r ", " #r ", %8\n"
#define I_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n"
#define I_LOG(r) "v_log_f32 " #r ", " #r "\n"
#define I_SIN(r) "v_sin_f32 " #r ", " #r "\n"
#define I_EXP(r) "v_exp_f32 " #r ", " #r "\n"
#define I_RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define I_PKFMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define I_CVTF64(r) "v_add_f32 " #r ", " #r ", %8\n"
KERNEL8(k_mullo, I_MULLO) KERNEL8(k_mulhi, I_MULHI) KERNEL8(k_mul24, I_MUL24) KERNEL8(k_mulhi24, I_MULHI24)
KERNEL8(k_xor, I_XOR) KERNEL8(k_log, I_LOG) KERNEL8(k_sin, I_SIN) KERNEL8(k_exp, I_EXP) KERNEL8(k_rcp, I_RCP) KERNEL8(k_fma, I_FMA)

__global__ void k_mad64(float *out, uint64_t *cyc, int n, uint32_t a, uint32_t b)
{
    uint64_t v0 = threadIdx.x + 3, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    uint32_t x0 = threadIdx.x * 7 + 1, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    uint64_t c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        asm volatile(REP4("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0\n")
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
    }
    uint64_t c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)(v0 + v1 + v2 + v3);
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = c1 - c0;
}
__global__ void k_f64add(float *out, uint64_t *cyc, int n, double a)
{
    double v0 = threadIdx.x + 3, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    uint64_t c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        asm volatile(REP4("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a));
    }
    uint64_t c1 = __builtin_readcyclecounter();
    out[threadIdx.x] = (float)(v0 + v1 + v2 + v3);
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = c1 - c0;
}

int main()
{
    float *out; uint64_t *cyc;
    hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 1024);
    const int n = 20000;
    uint64_t hc[32];
#define RUN(k, per_iter, waves) do { hipLaunchKernelGGL(k, 1, 64 * (waves), 0, 0, out, cyc, n, 1.0001f, 0.5f); hipDeviceSynchronize(); \
        hipMemcpy(hc, cyc, 8 * (waves), hipMemcpyDeviceToHost); double mx = 0; for (int i = 0; i < (waves); ++i) mx = hc[i] > mx ? hc[i] : mx; \
        printf("%-10s waves=%2d  %6.2f cycles/instr/wave   %6.2f cycles per wave-instr on the CU\n", #k, waves, mx / ((double)n * (per_iter)), mx / ((double)n * (per_iter) * (waves))); } while (0)
    RUN(k_fma, 32, 1); RUN(k_mullo, 32, 1); RUN(k_mulhi, 32, 1); RUN(k_mul24, 32, 1); RUN(k_mulhi24, 32, 1); RUN(k_xor, 32, 1);
    RUN(k_log, 32, 1); RUN(k_sin, 32, 1); RUN(k_exp, 32, 1); RUN(k_rcp, 32, 1);
    for (int w : {4, 8, 16}) { RUN(k_fma, 32, w); RUN(k_mullo, 32, w); RUN(k_log, 32, w); }
    hipLaunchKernelGGL(k_mad64, 1, 64, 0, 0, out, cyc, n, 0xD2511F53u, 0u); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("v_mad_u64_u32 waves=1 %6.2f cycles/instr\n", hc[0] / ((double)n * 16));
    hipLaunchKernelGGL(k_mad64, 1, 512, 0, 0, out, cyc, n, 0xD2511F53u, 0u); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("v_mad_u64_u32 waves=8 %6.2f cycles/instr/wave\n", hc[0] / ((double)n * 16));
    hipLaunchKernelGGL(k_f64add, 1, 64, 0, 0, out, cyc, n, 1.5); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("v_add_f64 waves=1 %6.2f cycles/instr\n", hc[0] / ((double)n * 16));
    hipLaunchKernelGGL(k_f64add, 1, 512, 0, 0, out, cyc, n, 1.5); hipDeviceSynchronize(); hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("v_add_f64 waves=8 %6.2f cycles/instr/wave\n", hc[0] / ((double)n * 16));
    return 0;
}
