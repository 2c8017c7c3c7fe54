// risk_kernels.hip -- risk-map precompute for the planner's constructor ("next" row N1).
//
// Replaces TraversabilityModel._infer_risk_map (reference traversability_model.py:28-51), which
// materialises (num_samples, G, G) slip samples, runs torch.quantile over dim 0 and a masked nanmean:
//   expected_value   R = mean
//   var              R = quantile_q( mean + std * z_i ),  linear interpolation (torch.quantile default)
//   cvar             R = mean of the samples strictly above that quantile (nanmean of the masked tensor)
// Here one wavefront owns one map cell and keeps its num_samples draws in REGISTERS (16 per lane at 1000 samples), as
// order-preserving integer keys.  The two order statistics torch.quantile interpolates between are found by a
// radix select over the key bits (32 rounds of compare + ballot + popcount: no sort, no LDS, no barrier), the tail
// mean by one more pass.  Only mean/std in and R out touch HBM (8 B + 4 B per cell instead of 4*num_samples B).
#include "../../include/benchnav_mppi.h"
#include "bn_device_math.h"

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <string>

namespace bn {
namespace {

constexpr int kRiskWaves = 4;            // cells per workgroup (one wave each, independent)

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// float <-> unsigned key with the same order (negative floats reversed, sign bit flipped)
__device__ __forceinline__ uint32_t key_of(float x)
{
    const uint32_t b = __float_as_uint(x);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float value_of(uint32_t k)
{
    return __uint_as_float(k ^ ((k & 0x80000000u) ? 0x80000000u : 0xffffffffu));
}

// R = draws per lane: capacity 64*R >= num_samples (padding = +inf sorts to the end)
template <int R>
__global__ __launch_bounds__(kRiskWaves * 64) void risk_map_kernel(const float *__restrict__ mean, const float *__restrict__ stdv,
                                                                   const float *__restrict__ z, float *__restrict__ out,
                                                                   int cells, int n, int metric, float qf, uint64_t seed)
{
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cell = blockIdx.x * kRiskWaves + wid;
    if (cell >= cells) return;                          // whole wave: no barrier anywhere below
    // rank arithmetic of torch.quantile: fp32 q * (n - 1), floor / ceil, weight = fractional part
    const float pos = qf * (float)(n - 1);
    const float lo_f = floorf(pos);
    const int lo = (int)lo_f, hi = (int)ceilf(pos);
    const float wgt = pos - lo_f;
    const float mu = mean[cell], sg = stdv[cell];

    uint32_t key[R];                                    // element index of key[r]: idx(r), below
    if (z) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            key[r] = key_of(i < n ? (z[(size_t)i * cells + cell] * sg + mu) : INFINITY);       // Normal.sample: normal_().mul_(std).add_(mean)
        }
    } else {
#pragma unroll
        for (int j = 0; j < R / 4; ++j) {
            const int i4 = lane + 64 * j;
            const u32x4 q = philox4x32_10(u32x4{(uint32_t)cell, (uint32_t)i4, 0x5249534bu, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
            float e[4];
            box_muller(q.x, q.y, e[0], e[1]);
            box_muller(q.z, q.w, e[2], e[3]);
#pragma unroll
            for (int s = 0; s < 4; ++s) key[4 * j + s] = key_of((4 * i4 + s) < n ? (e[s] * sg + mu) : INFINITY);
        }
    }
    // radix select of the key of rank lo (0-based, ascending): the largest prefix with #(key < prefix) <= lo
    uint32_t klo = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = klo | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) cnt += __popcll(__ballot(key[r] < cand));
        if (cnt <= lo) klo = cand;
    }
    // rank hi (= lo or lo + 1): the same key if it repeats past rank lo, else the smallest key above it
    uint32_t khi = klo;
    if (hi != lo) {
        int cnt_le = 0;
        uint32_t mn = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            cnt_le += __popcll(__ballot(key[r] <= klo));
            mn = min(mn, key[r] > klo ? key[r] : 0xffffffffu);
        }
        if (cnt_le <= hi) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
            khi = mn;
        }
    }
    const float below = value_of(klo), above = value_of(khi);
    // at::lerp: |w| < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
    const float var = (fabsf(wgt) < 0.5f) ? below + wgt * (above - below) : above - (above - below) * (1.0f - wgt);
    if (metric == 1) {
        if (lane == 0) out[cell] = var;
        return;
    }
    float s = 0.0f, c = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = z ? lane + 64 * r : 4 * (lane + 64 * (r >> 2)) + (r & 3);
        const float x = value_of(key[r]);
        if (i < n && x > var) { s += x; c += 1.0f; }
    }
    s = wave_sum_f(s);
    c = wave_sum_f(c);
    if (lane == 0) out[cell] = s / c;                   // nanmean: 0/0 = NaN when no sample exceeds the quantile
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
    // work on `device_id`, leave the calling thread's current device as it was
    struct Guard {
        int prev = -1; bool changed = false, ok = true;
        explicit Guard(int want) { if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
                                   if (prev != want) { ok = hipSetDevice(want) == hipSuccess; changed = ok; } }
        ~Guard() { if (changed) (void)hipSetDevice(prev); }
    } guard(device_id);
    if (!guard.ok) return fail(BN_ERR_HIP, "hipSetDevice failed");
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
        const int blocks = (int)((cells + bn::kRiskWaves - 1) / bn::kRiskWaves);
        const int m = metric == BN_RISK_VAR ? 1 : 2;
        if (num_samples <= 1024)
            bn::risk_map_kernel<16><<<blocks, bn::kRiskWaves * 64, 0, s>>>(d_mean, d_std, d_z, d_out, (int)cells, num_samples, m, confidence, seed);
        else if (num_samples <= 2048)
            bn::risk_map_kernel<32><<<blocks, bn::kRiskWaves * 64, 0, s>>>(d_mean, d_std, d_z, d_out, (int)cells, num_samples, m, confidence, seed);
        else
            bn::risk_map_kernel<64><<<blocks, bn::kRiskWaves * 64, 0, s>>>(d_mean, d_std, d_z, d_out, (int)cells, num_samples, m, confidence, seed);
        RISK_HIP(hipGetLastError());
    }
    if (where_out == BN_MEM_HOST) RISK_HIP(hipMemcpyAsync(out, d_out, cells * 4, hipMemcpyDeviceToHost, s));
    if (where_in == BN_MEM_HOST || where_out == BN_MEM_HOST || (z && where_z == BN_MEM_HOST)) RISK_HIP(hipStreamSynchronize(s));
    free_all();
#undef RISK_HIP
    return BN_OK;
}

}  // extern "C"
