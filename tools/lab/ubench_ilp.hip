// The latency kernel's chain step written out in asm, in two instruction ORDERS: "seq" -- the position strand (update, clamp,
// quotient, floor, cell, address, gather) and then the heading strand (rotation) under the gather's latency, the order the product
// uses -- and "ilp" -- the two strands interleaved instruction by instruction, so that (almost) no instruction depends on the one
// issued right before it.  One wavefront alone on a CU, then with four busy companions.  If a dependent VALU instruction cannot
// issue back to back with its producer, "ilp" is the faster one although it executes the same operations.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_ilp.hip -o tools/ubench_ilp.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));

// Fixed registers (the halves of a 64-bit asm operand cannot be named):
//   v[40:43] ring tuple (xn, yn, dth, trav)   v[44:45] clamped position   v[46:47] (cos d, sin d / d)   v[48:49] quotient / cell
//   v[50:51] d^2 | sd, u                      v[52:53] g                  v[54:55] G = g (cs, sn)       v[56:57] (cs, sn)
//   v[58:59] trav copy (seq)                  v60 wq                       v61 gather address            v62 ring address
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62"
// operands: %0 dt (s)  %1 u0 (v)  %2 u1 (v)  %3 ir (s64)  %4 nw (v64)  %5 wn (s)  %6 base (s)  %7 c1 (s64)  %8 c0 (v64)  %9 c2 (s64)
//           %10 lo (s)  %11 hi (v)
#define STEP_SEQ(OFF)                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
    "ds_write_b128 v62, v[40:43] offset:" #OFF "\n\t"                                                  \
    "v_mul_f32 v52, %0, %1\n\t"                                                                         \
    "v_pk_mul_f32 v[54:55], v[52:53], v[56:57] op_sel_hi:[0,1]\n\t"                                     \
    "v_mov_b32 v58, v43\n\t"                                                                            \
    "v_pk_fma_f32 v[40:41], v[58:59], v[54:55], v[44:45] op_sel_hi:[0,1,1]\n\t"                         \
    "v_med3_f32 v44, v40, %10, %11\n\t"                                                                 \
    "v_med3_f32 v45, v41, %10, %11\n\t"                                                                 \
    "v_pk_fma_f32 v[48:49], v[44:45], %3, %4\n\t"                                                       \
    "v_cvt_flr_i32_f32 v48, v48\n\t"                                                                    \
    "v_cvt_flr_i32_f32 v49, v49\n\t"                                                                    \
    "v_mad_u32_u24 v48, v49, %5, v48\n\t"                                                               \
    "v_lshl_add_u32 v61, v48, 2, %6\n\t"                                                                \
    "v_mul_f32 v60, %0, %2\n\t"                                                                         \
    "v_mul_f32 v42, v43, v60\n\t"                                                                       \
    "ds_read_b32 v43, v61\n\t"                                                                          \
    "v_mul_f32 v50, v42, v42\n\t"                                                                       \
    "v_pk_fma_f32 v[46:47], v[50:51], %7, %8 op_sel_hi:[0,1,1]\n\t"                                     \
    "v_pk_fma_f32 v[46:47], v[50:51], v[46:47], %9 op_sel_hi:[0,1,1]\n\t"                               \
    "v_pk_fma_f32 v[46:47], v[50:51], v[46:47], 1.0 op_sel_hi:[0,1,0]\n\t"                              \
    "v_mul_f32 v50, v42, v47\n\t"                                                                       \
    "v_pk_mul_f32 v[50:51], v[56:57], v[50:51] op_sel_hi:[1,0]\n\t"                                     \
    "v_pk_fma_f32 v[56:57], v[56:57], v[46:47], v[50:51] op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1]\n\t"
// the same operations (the position update as two v_fma_f32 -- same roundings as the packed one, and no copy of trav into a pair);
// G of THIS step was formed at the end of the previous one
#define STEP_ILP(OFF)                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
    "ds_write_b128 v62, v[40:43] offset:" #OFF "\n\t"                                                  \
    "v_fma_f32 v40, v43, v54, v44\n\t"                                                                  \
    "v_fma_f32 v41, v43, v55, v45\n\t"                                                                  \
    "v_mul_f32 v42, v43, v60\n\t"                                                                       \
    "v_med3_f32 v44, v40, %10, %11\n\t"                                                                 \
    "v_med3_f32 v45, v41, %10, %11\n\t"                                                                 \
    "v_mul_f32 v50, v42, v42\n\t"                                                                       \
    "v_pk_fma_f32 v[48:49], v[44:45], %3, %4\n\t"                                                       \
    "v_pk_fma_f32 v[46:47], v[50:51], %7, %8 op_sel_hi:[0,1,1]\n\t"                                     \
    "v_mul_f32 v52, %0, %1\n\t"                                                                         \
    "v_cvt_flr_i32_f32 v48, v48\n\t"                                                                    \
    "v_cvt_flr_i32_f32 v49, v49\n\t"                                                                    \
    "v_pk_fma_f32 v[46:47], v[50:51], v[46:47], %9 op_sel_hi:[0,1,1]\n\t"                               \
    "v_mad_u32_u24 v48, v49, %5, v48\n\t"                                                               \
    "v_pk_fma_f32 v[46:47], v[50:51], v[46:47], 1.0 op_sel_hi:[0,1,0]\n\t"                              \
    "v_lshl_add_u32 v61, v48, 2, %6\n\t"                                                                \
    "v_mul_f32 v60, %0, %2\n\t"                                                                         \
    "ds_read_b32 v43, v61\n\t"                                                                          \
    "v_mul_f32 v50, v42, v47\n\t"                                                                       \
    "v_pk_mul_f32 v[50:51], v[56:57], v[50:51] op_sel_hi:[1,0]\n\t"                                     \
    "v_pk_fma_f32 v[56:57], v[56:57], v[46:47], v[50:51] op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1]\n\t" \
    "v_pk_mul_f32 v[54:55], v[52:53], v[56:57] op_sel_hi:[0,1]\n\t"

// MODE 0 seq, 1 ilp; companions (threads > 64) spin on VALU work and LDS reads of the ring like the kernel's other waves
template <int MODE, int BIG = 0>
__global__ void k_chain(float *out, uint64_t *cyc, int nsteps, float dt, float inv_res)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                 // 64 slots x 64 x float4
    float *win = ring + 64 * 256;       // 23 x 23 + guard
    volatile int *stop = (volatile int *)(win + 24 * 24);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 24 * 24; i += blockDim.x) win[i] = 0.3f + 0.001f * (i % 97);
    if (tid == 0) *stop = 0;
    __syncthreads();
    typedef __attribute__((address_space(3))) float lds_f;
    if (wid == 0) {
        const unsigned base = (unsigned)(uintptr_t)(lds_f *)win;
        const unsigned rbase = (unsigned)(uintptr_t)(lds_f *)ring + lane * 16;
        const v2f ir = {inv_res, inv_res}, nw = {-4.0f, -6.0f};
        const v2f c0 = {0.0416666679084300995f, 0.00833333376795053482f}, c1 = {-0.00138888892251998186f, -0.000198412701138295233f}, c2 = {-0.5f, -0.16666667163372040f};
        const float u0 = 0.4f + 0.003f * lane, u1 = 0.3f - 0.004f * lane;
        const float x0 = 5.0f + 0.01f * lane, y0 = 6.0f + 0.02f * lane;
        float r0, r1, r2, r3;
        // set-up of the fixed registers, then the timed loop, all in one asm statement (nothing of the compiler's in between)
        uint64_t w_0 = wall_clock64();
        uint64_t c_0 = __builtin_readcyclecounter();
        asm volatile(
            "v_mov_b32 v44, %3\n\tv_mov_b32 v45, %4\n\tv_mov_b32 v56, 0x3f4ccccd\n\tv_mov_b32 v57, 0x3f19999a\n\t"
            "v_mov_b32 v43, 0.5\n\tv_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v62, %5\n\t"
            "v_mul_f32 v52, %0, %1\n\tv_mul_f32 v60, %0, %2\n\tv_pk_mul_f32 v[54:55], v[52:53], v[56:57] op_sel_hi:[0,1]\n\t"
            "s_mov_b32 s40, %6\n\t"
            :: "s"(dt), "v"(u0), "v"(u1), "v"(x0), "v"(y0), "v"(rbase), "s"(nsteps / 4) : CLOB, "s40", "memory");
        if (MODE == 0)
            asm volatile("1:\n\t" STEP_SEQ(0) STEP_SEQ(1024) STEP_SEQ(2048) STEP_SEQ(3072)
                         "s_sub_u32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\t"
                         :: "s"(dt), "v"(u0), "v"(u1), "s"(ir), "v"(nw), "s"(23), "s"(base), "s"(c1), "v"(c0), "s"(c2), "s"(3.0f), "v"(13.0f) : CLOB, "s40", "scc", "memory");
        else
            asm volatile("1:\n\t" STEP_ILP(0) STEP_ILP(1024) STEP_ILP(2048) STEP_ILP(3072)
                         "s_sub_u32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\t"
                         :: "s"(dt), "v"(u0), "v"(u1), "s"(ir), "v"(nw), "s"(23), "s"(base), "s"(c1), "v"(c0), "s"(c2), "s"(3.0f), "v"(13.0f) : CLOB, "s40", "scc", "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v44\n\tv_mov_b32 %1, v45\n\tv_mov_b32 %2, v56\n\tv_mov_b32 %3, v43" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) :: CLOB);
        uint64_t c_1 = __builtin_readcyclecounter();
        out[lane] = r0;
        out[64 + lane] = r1;
        out[128 + lane] = r2;
        out[192 + lane] = r3;
        if (lane == 0 && blockIdx.x == 0) { cyc[0] = c_1 - c_0; cyc[1] = wall_clock64() - w_0; }
        *stop = 1;
    } else if (BIG) {
        // companions in long straight-line code of their own (BIG x 256 VOP3 instructions = BIG x 2 KB each, a different copy per wave):
        // do five waves streaming through different code slow each other's instruction fetch?
        float a0 = 1.0f + 0.001f * lane, a1 = 0.5f, a2 = 0.25f, a3 = 2.0f;
#define R4 "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1\n\t"
#define R16 R4 R4 R4 R4
#define R64 R16 R16 R16 R16
#define R256 R64 R64 R64 R64
        while (!*stop) {
            if (wid == 1) { for (int r = 0; r < BIG; ++r) asm volatile(R256 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
            else if (wid == 2) { for (int r = 0; r < BIG; ++r) asm volatile(R256 : "+v"(a1), "+v"(a0), "+v"(a2), "+v"(a3)); }
            else if (wid == 3) { for (int r = 0; r < BIG; ++r) asm volatile(R256 : "+v"(a2), "+v"(a1), "+v"(a0), "+v"(a3)); }
            else { for (int r = 0; r < BIG; ++r) asm volatile(R256 : "+v"(a3), "+v"(a1), "+v"(a2), "+v"(a0)); }
        }
        out[256 + tid] = a0 + a1 + a2 + a3;
    } else {
        float acc = 0.0f, a = 1.0f + 0.001f * lane;
        const float4 *slot = reinterpret_cast<const float4 *>(ring) + lane;
        while (!*stop) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float4 o = slot[64 * ((wid + i) & 63)]; a = a * 0.999f + o.x; acc += sqrtf(a * a + o.w); }
        }
        out[256 + tid] = acc;
    }
}

int main()
{
    float *out; uint64_t *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64);
    int n = 4000;
    const size_t lds = (64 * 256 + 24 * 24 + 4) * 4;
    uint64_t hc[2];
    float h[2][256];
#define RUN(MODE, threads, label) RUNG(MODE, 1, threads, label)
#define RUNG(MODE, blocks, threads, label) do { hipFuncSetAttribute((const void *)k_chain<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k_chain<MODE>, blocks, threads, lds, 0, out, cyc, n, 0.1f, 2.0f); hipDeviceSynchronize(); \
        hipLaunchKernelGGL(k_chain<MODE>, blocks, threads, lds, 0, out, cyc, n, 0.1f, 2.0f); hipDeviceSynchronize(); \
        hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost); hipMemcpy(h[MODE], out, 1024, hipMemcpyDeviceToHost); \
        printf("%-44s %7.1f counter ticks/step  %6.1f ns/step (100 MHz wall clock)  -> counter at %.0f MHz\n", label, (double)hc[0] / n, (double)hc[1] * 10.0 / n, (double)hc[0] / ((double)hc[1] * 0.01)); } while (0)
    RUN(0, 64, "lone wave, strands one after the other");
    RUN(1, 64, "lone wave, strands interleaved");
    int same = 1;
    for (int i = 0; i < 256; ++i) same &= h[0][i] == h[1][i];
    printf("final states of the two orders: %s (x[0] = %.7g, cs[0] = %.7g)\n", same ? "bit-identical" : "DIFFERENT", h[0][0], h[0][128]);
    RUN(0, 320, "4 busy companions, one after the other");
    RUN(1, 320, "4 busy companions, interleaved");
    RUNG(0, 17, 320, "17 workgroups x 5 waves, one after the other");
    n = 50;
    RUNG(0, 17, 320, "the same, 50 steps per launch");
    RUNG(0, 1, 64, "lone wave, 50 steps per launch");
    n = 4000;
#define RUNB(BIG, label) do { hipFuncSetAttribute((const void *)k_chain<0, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_chain<0, BIG>), 1, 320, lds, 0, out, cyc, n, 0.1f, 2.0f); hipDeviceSynchronize(); \
        hipLaunchKernelGGL((k_chain<0, BIG>), 1, 320, lds, 0, out, cyc, n, 0.1f, 2.0f); hipDeviceSynchronize(); \
        hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost); \
        printf("%-44s %7.1f counter ticks/step  %6.1f ns/step\n", label, (double)hc[0] / n, (double)hc[1] * 10.0 / n); } while (0)
    RUNB(1, "4 companions, 2 KB of straight-line code each");
    RUNB(4, "4 companions, 8 KB each (unrolled x4 loop)");
    return 0;
}
