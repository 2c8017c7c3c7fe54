#include <hip/hip_runtime.h>
#define PARK_CLOBBERS "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111"
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(96))) void k(const float *in, float *out, int T)
{
    const int lane = threadIdx.x;
    float acc = in[lane];
    for (int t = 0; t < T; t += 2) {
        acc = acc * 1.0001f + (float)t;
        const float u0 = acc, u1 = acc * 0.5f, v0 = acc * 0.25f, v1 = acc + 1.0f;
        const int idx = 2 * t;
        asm volatile("s_set_gpr_idx_on %4, 0x8\n\t"
                     "v_mov_b32 v96, %0\n\t"
                     "v_mov_b32 v97, %1\n\t"
                     "v_mov_b32 v98, %2\n\t"
                     "v_mov_b32 v99, %3\n\t"
                     "s_set_gpr_idx_off"
                     :: "v"(u0), "v"(u1), "v"(v0), "v"(v1), "s"(idx) : PARK_CLOBBERS);
    }
    const float e = acc;
    for (int t = 0; t < T; t += 2) {
        float u0, u1, v0, v1;
        const int idx = 2 * t;
        asm volatile("s_set_gpr_idx_on %4, 0x1\n\t"
                     "v_mov_b32 %0, v96\n\t"
                     "v_mov_b32 %1, v97\n\t"
                     "v_mov_b32 %2, v98\n\t"
                     "v_mov_b32 %3, v99\n\t"
                     "s_set_gpr_idx_off"
                     : "=&v"(u0), "=&v"(u1), "=&v"(v0), "=&v"(v1) : "s"(idx));
        out[(2 * t) * 64 + lane] = e * u0;
        out[(2 * t + 1) * 64 + lane] = e * u1;
        out[(2 * t + 2) * 64 + lane] = e * v0;
        out[(2 * t + 3) * 64 + lane] = e * v1;
    }
}
int main(){ float *in,*out; int T=8; hipMalloc(&in,256); hipMalloc(&out,4*2*T*64+1024); float h[64]; for(int i=0;i<64;i++)h[i]=i; hipMemcpy(in,h,256,hipMemcpyHostToDevice); k<<<1,64>>>(in,out,T); float r[64*16]; hipMemcpy(r,out,sizeof(r),hipMemcpyDeviceToHost); printf("%f %f %f %f\n", r[1], r[64+1], r[128+1], r[15*64+1]); return 0; }
