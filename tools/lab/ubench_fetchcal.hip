// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the planner uses:
// coalesced 4-byte-per-lane row reads (window staging, injected noise) vs 16-byte-per-lane streams, over 64 MiB.
// Run: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o c -- ./tools/ubench_fetchcal.bin   (and once with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const float *in, float *out, size_t n)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read16(const float4 *in, float *out, size_t n4)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void write4(float *out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0f;
}
int main()
{
    const size_t n = 16u << 20;                          // 64 MiB of float
    float *a, *o; hipMalloc(&a, n * 4); hipMalloc(&o, n * 4); hipMemset(a, 0, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        read4<<<2048, 256>>>(a, o, n);
        read16<<<2048, 256>>>((const float4 *)a, o, n / 4);
        write4<<<2048, 256>>>(o, n);
    }
    hipDeviceSynchronize();
    printf("each kernel touches %zu bytes\n", n * 4);
    return 0;
}
