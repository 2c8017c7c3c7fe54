// The latency kernel's chain step in isolation (one wavefront, then with companion waves that poll / read LDS like the
// consumers do): cycles per step for the full step and with parts removed.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float med3(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

__device__ __forceinline__ void rotate(float &cs, float &sn, float d)
{
    const float d2 = d * d;
    const v2f dd = {d2, d2};
    v2f pq = __builtin_elementwise_fma(dd, v2f{-0.00138888892251998186f, -0.000198412701138295233f}, v2f{0.0416666679084300995f, 0.00833333376795053482f});
    pq = __builtin_elementwise_fma(dd, pq, v2f{-0.5f, -0.16666667163372040f});
    pq = __builtin_elementwise_fma(dd, pq, v2f{1.0f, 1.0f});
    const float sd = d * pq.y;
    const v2f h = {cs, sn};
    const v2f u = h * v2f{sd, sd};
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1]" : "=v"(r) : "v"(h), "v"(pq), "v"(u));
    cs = r.x; sn = r.y;
}

// MODE bits: 1 no ring write, 2 no rotation, 4 no gather, 8 no progress post, 16 companions poll, 32 companions read ring
template <int MODE>
__global__ void k_chain(float *out, uint64_t *cyc, int nsteps, float dt, float inv_res, float fwm1, float fwn)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                 // 64 slots x 64 x float4
    float *win = ring + 64 * 256;       // 23 x 23
    int *prog = (int *)(win + 23 * 23 + 3);
    float *tile = (float *)(prog + 4);  // 2 x 64 controls x 65
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 23 * 23; i += blockDim.x) win[i] = 0.3f + 0.001f * (i % 97);
    for (int i = tid; i < 128 * 65; i += blockDim.x) tile[i] = 0.4f + 0.003f * (i % 61);
    if (tid == 0) prog[0] = 0;
    __syncthreads();
    typedef __attribute__((address_space(3))) volatile int lds_vint;
    lds_vint *cprog = (lds_vint *)prog;
    if (wid == 0) {
        float x = 5.0f + 0.01f * lane, y = 6.0f + 0.02f * lane, cs = 0.8f, sn = 0.6f, trav = 0.5f;
        float u0 = tile[lane], u1 = tile[65 + lane];
        v2f G = v2f{u0 * dt, u0 * dt} * v2f{cs, sn};
        float wq = u1 * dt;
        uint64_t w0 = wall_clock64();
        uint64_t c0 = __builtin_readcyclecounter();
        for (int t = 0; t < nsteps; t += 4) {
            float ua[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { ua[i][0] = tile[(2 * ((t + i + 1) & 63)) * 65 + lane]; ua[i][1] = tile[(2 * ((t + i + 1) & 63) + 1) * 65 + lane]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dth = trav * wq;
                const v2f pos = __builtin_elementwise_fma(v2f{trav, trav}, G, v2f{x, y});
                const float xn = pos.x, yn = pos.y;
                x = med3(xn, 0.0f, 128.0f); y = med3(yn, 0.0f, 128.0f);
                if (MODE & 4) trav = 0.5f + 0.001f * x;
                else {
                    const v2f q = __builtin_elementwise_fma(v2f{x, y}, v2f{inv_res, inv_res}, v2f{-4.0f, -6.0f});
                    const float li = med3(floorf(q.x), 0.0f, fwm1), lj = med3(floorf(q.y), 0.0f, fwm1);
                    trav = win[(int)__builtin_fmaf(lj, fwn, li)];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(MODE & 2)) rotate(cs, sn, dth); else { cs = cs * 0.999f + dth; }
                { const float g = ua[i][0] * dt; wq = ua[i][1] * dt; G = v2f{g, g} * v2f{cs, sn}; }
                __builtin_amdgcn_sched_barrier(0);
                if (!(MODE & 1)) reinterpret_cast<float4 *>(ring + (size_t)((t + i) & 63) * 256)[lane] = make_float4(xn, yn, dth, trav);
                else { asm volatile("" :: "v"(xn), "v"(yn), "v"(dth)); }
                asm volatile("" ::: "memory");
                if (!(MODE & 8) && i == 3) *cprog = t + i + 1;
            }
        }
        uint64_t c1 = __builtin_readcyclecounter();
        out[lane] = x + y + cs + sn + trav;
        if (lane == 0) { cyc[0] = c1 - c0; cyc[1] = wall_clock64() - w0; }
        *cprog = 1 << 30;
    } else {
        // companions: poll the progress counter like wait_progress (16), and read ring slots like a consumer (32)
        float acc = 0.0f;
        if (MODE & (16 | 32)) {
            int seen = 0;
            while (seen < (1 << 30)) {
                seen = *cprog;
                if (MODE & 32) { const float4 o = reinterpret_cast<const float4 *>(ring + (size_t)(seen & 63) * 256)[lane]; acc += o.x + o.w; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        out[tid] = acc;
    }
}

int main()
{
    float *out; uint64_t *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64);
    const int n = 4000;
    const size_t lds = (64 * 256 + 23 * 23 + 3 + 4 + 128 * 65) * 4;
    uint64_t hc2[2];
#define RUN(MODE, threads, label) do { hipFuncSetAttribute((const void *)k_chain<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k_chain<MODE>, 1, threads, lds, 0, out, cyc, n, 0.1f, 2.0f, 22.0f, 23.0f); hipDeviceSynchronize(); \
        hipLaunchKernelGGL(k_chain<MODE>, 1, threads, lds, 0, out, cyc, n, 0.1f, 2.0f, 22.0f, 23.0f); hipDeviceSynchronize(); \
        hipMemcpy(hc2, cyc, 16, hipMemcpyDeviceToHost); printf("%-52s %7.1f cycles/step  %6.1f ns/step  (%.0f MHz)\n", label, (double)hc2[0] / n, (double)hc2[1] * 10.0 / n, (double)hc2[0] / ((double)hc2[1] * 0.01)); } while (0)
    RUN(0, 64, "lone wave: full step");
    RUN(1, 64, "lone wave: no ring write");
    RUN(2, 64, "lone wave: no rotation");
    RUN(4, 64, "lone wave: no gather");
    RUN(8, 64, "lone wave: no progress post");
    RUN(7, 64, "lone wave: position arithmetic only");
    RUN(16, 320, "4 companions polling the progress counter");
    RUN(48, 320, "4 companions polling + reading ring slots");
    RUN(48, 128, "1 companion polling + reading ring slots");
    return 0;
}
