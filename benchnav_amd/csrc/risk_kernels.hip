// risk_kernels.hip -- risk-map precompute for the planner's constructor ("next" row N1).
//
// Replaces TraversabilityModel._infer_risk_map (reference traversability_model.py:28-51), which
// materialises (num_samples, G, G) slip samples, runs torch.quantile over dim 0 and a masked nanmean:
//   expected_value   R = mean
//   var              R = quantile_q( mean + std * z_i ),  linear interpolation (torch.quantile default)
//   cvar             R = mean of the samples strictly above that quantile (nanmean of the masked tensor)
// Here one wavefront owns one map cell: its num_samples draws live in LDS, are sorted with a bitonic
// network, and only mean/std in and R out touch HBM (8 B + 4 B per cell instead of 4*num_samples B).
#include "../../include/benchnav_mppi.h"
#include "bn_device_math.h"

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <string>

namespace bn {
namespace {

constexpr int kRiskWaves = 4;            // cells per workgroup (one wave each)

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// P = power-of-two capacity >= num_samples (padding sorts to the end as +inf)
template <int P>
__global__ __launch_bounds__(kRiskWaves * 64) void risk_map_kernel(const float *__restrict__ mean, const float *__restrict__ stdv,
                                                                   const float *__restrict__ z, float *__restrict__ out,
                                                                   int cells, int n, int metric, float qf, uint64_t seed)
{
    __shared__ float smem[kRiskWaves * P];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *buf = smem + wid * P;
    // rank arithmetic of torch.quantile: fp32 q * (n - 1), floor / ceil, weight = fractional part
    const float pos = qf * (float)(n - 1);
    const float lo_f = floorf(pos);
    const int lo = (int)lo_f, hi = (int)ceilf(pos);
    const float wgt = pos - lo_f;

    for (int base = blockIdx.x * kRiskWaves; base < cells; base += gridDim.x * kRiskWaves) {
        const int cell = base + wid;
        const bool live = cell < cells;
        const float mu = live ? mean[cell] : 0.0f, sg = live ? stdv[cell] : 0.0f;
        if (z) {
            for (int i = lane; i < P; i += 64)
                buf[i] = (live && i < n) ? (z[(size_t)i * cells + cell] * sg + mu) : INFINITY;   // Normal.sample: normal_().mul_(std).add_(mean)
        } else {
            for (int i4 = lane; i4 < P / 4; i4 += 64) {
                const u32x4 r = philox4x32_10(u32x4{(uint32_t)cell, (uint32_t)i4, 0x5249534bu, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
                float e[4];
                box_muller(r.x, r.y, e[0], e[1]);
                box_muller(r.z, r.w, e[2], e[3]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int i = 4 * i4 + s;
                    buf[i] = (live && i < n) ? (e[s] * sg + mu) : INFINITY;
                }
            }
        }
        __syncthreads();
        // bitonic sort, ascending
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < P / 2; t += 64) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // lower index of the pair
                    const int l = i | j;
                    const float a = buf[i], b = buf[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { buf[i] = b; buf[l] = a; }
                }
                __syncthreads();
            }
        }
        if (live) {
            const float below = buf[lo], above = buf[hi];
            // at::lerp: |w| < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
            const float var = (fabsf(wgt) < 0.5f) ? below + wgt * (above - below) : above - (above - below) * (1.0f - wgt);
            if (metric == 1) {
                if (lane == 0) out[cell] = var;
            } else {
                float s = 0.0f, c = 0.0f;
                for (int i = lane; i < n; i += 64) {
                    const float x = buf[i];
                    if (x > var) { s += x; c += 1.0f; }
                }
                s = wave_sum_f(s);
                c = wave_sum_f(c);
                if (lane == 0) out[cell] = s / c;            // nanmean: 0/0 = NaN when no sample exceeds the quantile
            }
        }
        __syncthreads();
    }
}

thread_local std::string g_risk_error;

}  // namespace
}  // namespace bn

extern "C" {

const char *bn_risk_last_error(void) { return bn::g_risk_error.c_str(); }

int bn_risk_map_infer(int32_t device_id, void *stream, const float *mean, const float *stdv, bn_mem_kind where_in,
                      int32_t grid_size, bn_risk_metric metric, float confidence, int32_t num_samples, const float *z,
                      bn_mem_kind where_z, uint64_t seed, float *out, bn_mem_kind where_out)
{
    auto fail = [](int code, const std::string &msg) { bn::g_risk_error = msg; return code; };
    if (!mean || !stdv || !out) return fail(BN_ERR_INVALID, "null argument");
    if (grid_size < 1) return fail(BN_ERR_INVALID, "grid_size must be >= 1");
    if (metric != BN_RISK_EXPECTED && metric != BN_RISK_VAR && metric != BN_RISK_CVAR) return fail(BN_ERR_INVALID, "unknown metric");
    if (metric != BN_RISK_EXPECTED && (num_samples < 2 || num_samples > 4096))
        return fail(BN_ERR_INVALID, "num_samples must be in [2, 4096]");
    if (metric != BN_RISK_EXPECTED && !(confidence >= 0.0f && confidence <= 1.0f))
        return fail(BN_ERR_INVALID, "confidence must be in [0, 1]");       // ModelConfig asserts the same (utils.py:27-33)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BN_ERR_NO_DEVICE, "no HIP device visible: no CPU fallback");
    if (hipSetDevice(device_id) != hipSuccess) return fail(BN_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    const size_t cells = (size_t)grid_size * grid_size;
#define RISK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { free_all(); return fail(BN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    float *d_mean = nullptr, *d_std = nullptr, *d_z = nullptr, *d_out = nullptr;
    auto free_all = [&]() {
        if (where_in == BN_MEM_HOST) { if (d_mean) (void)hipFree(d_mean); if (d_std) (void)hipFree(d_std); }
        if (z && where_z == BN_MEM_HOST && d_z) (void)hipFree(d_z);
        if (where_out == BN_MEM_HOST && d_out) (void)hipFree(d_out);
    };
    if (where_in == BN_MEM_HOST) {
        RISK_HIP(hipMalloc((void **)&d_mean, cells * 4));
        RISK_HIP(hipMalloc((void **)&d_std, cells * 4));
        RISK_HIP(hipMemcpyAsync(d_mean, mean, cells * 4, hipMemcpyHostToDevice, s));
        RISK_HIP(hipMemcpyAsync(d_std, stdv, cells * 4, hipMemcpyHostToDevice, s));
    } else { d_mean = const_cast<float *>(mean); d_std = const_cast<float *>(stdv); }
    if (where_out == BN_MEM_HOST) RISK_HIP(hipMalloc((void **)&d_out, cells * 4)); else d_out = out;
    if (metric == BN_RISK_EXPECTED) {
        RISK_HIP(hipMemcpyAsync(d_out, d_mean, cells * 4, hipMemcpyDeviceToDevice, s));     // distributions.mean
    } else {
        if (z) {
            if (where_z == BN_MEM_HOST) {
                RISK_HIP(hipMalloc((void **)&d_z, cells * (size_t)num_samples * 4));
                RISK_HIP(hipMemcpyAsync(d_z, z, cells * (size_t)num_samples * 4, hipMemcpyHostToDevice, s));
            } else d_z = const_cast<float *>(z);
        }
        const int blocks = (int)std::min<size_t>((cells + bn::kRiskWaves - 1) / bn::kRiskWaves, 256 * 8);
        const int m = metric == BN_RISK_VAR ? 1 : 2;
        if (num_samples <= 1024)
            bn::risk_map_kernel<1024><<<blocks, bn::kRiskWaves * 64, 0, s>>>(d_mean, d_std, d_z, d_out, (int)cells, num_samples, m, confidence, seed);
        else if (num_samples <= 2048)
            bn::risk_map_kernel<2048><<<blocks, bn::kRiskWaves * 64, 0, s>>>(d_mean, d_std, d_z, d_out, (int)cells, num_samples, m, confidence, seed);
        else
            bn::risk_map_kernel<4096><<<blocks, bn::kRiskWaves * 64, 0, s>>>(d_mean, d_std, d_z, d_out, (int)cells, num_samples, m, confidence, seed);
        RISK_HIP(hipGetLastError());
    }
    if (where_out == BN_MEM_HOST) RISK_HIP(hipMemcpyAsync(out, d_out, cells * 4, hipMemcpyDeviceToHost, s));
    if (where_in == BN_MEM_HOST || where_out == BN_MEM_HOST || (z && where_z == BN_MEM_HOST)) RISK_HIP(hipStreamSynchronize(s));
    free_all();
#undef RISK_HIP
    return BN_OK;
}

}  // extern "C"
