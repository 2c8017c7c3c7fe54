// mppi_kernels.hip -- gfx950 (CDNA4) kernels of the MPPI solve step.
//
// Layout (DESIGN.md "Kernels"): one wavefront = 64 rollouts, lane = rollout k.
// The T-step recurrence is a serial chain per rollout, so every lane carries its
// own chain; noise, trajectories and controls are stored k-fastest
// ((B,T,2,K) / (B,T+1,3,K)) so each step's loads and stores are 256-byte
// coalesced rows.  The reachable window of the risk map is staged once per
// workgroup into LDS as traversability (1 - clamp(risk,0,1)); per-step gathers
// then hit LDS.  The block-level softmin statistics (max, sum, weighted control
// sums) are produced with wave shuffles plus an LDS control tile; a second,
// one-workgroup-per-instance kernel merges the blocks, writes U*, the weights
// and rolls out X*.
//
// Reference semantics reproduced here (file:line in the BenchNav checkout):
//   sampling   mppi.py:146-157        transit  robot_model.py:59-100 (in-place aliasing :78,86-88)
//   lookup     grid_map.py:145-210    costs    objectives.py:29-65, mppi.py:168-190
//   softmin    mppi.py:193-199        X*, warm start  mppi.py:202-217
#include "mppi_kernels.h"
#include "bn_device_math.h"

#include <math.h>

namespace bn {

namespace {

constexpr int TU = 4;   // time steps per software-pipelined chunk (noise is fetched one chunk ahead)

struct Win { int wx0, wy0; };

template <bool POW2>
__device__ __forceinline__ Win window_origin(const SolveParams &p, float sx, float sy)
{
    const int cx = cell_index<POW2>(sx, p.x0, p.res, p.inv_res, p.G - 1);
    const int cy = cell_index<POW2>(sy, p.y0, p.res, p.inv_res, p.G - 1);
    Win w;
    w.wx0 = min(max(cx - p.reach, 0), p.G - p.WN);
    w.wy0 = min(max(cy - p.reach, 0), p.G - p.WN);
    return w;
}

// Stage the reachable window as traversability: trav = 1 - clamp(risk, 0, 1)
// (reference traversability_model.py:72).  Rows of the window are contiguous
// runs of the map rows, so the loads coalesce per row.
__device__ __forceinline__ void stage_window(float *win, const float *__restrict__ map, const Win w,
                                             int WN, int G, int tid, int nthreads)
{
    const int n = WN * WN;
    for (int e = tid; e < n; e += nthreads) {
        const int r = e / WN;
        const int c = e - r * WN;
        const float risk = map[(size_t)(w.wy0 + r) * G + (w.wx0 + c)];
        win[e] = 1.0f - clampf(risk, 0.0f, 1.0f);
    }
}

template <bool POW2, bool LDSWIN>
__device__ __forceinline__ float trav_lookup(const SolveParams &p, const float *win,
                                             const float *__restrict__ map, const Win w, float x, float y)
{
    const int ix = cell_index<POW2>(x, p.x0, p.res, p.inv_res, p.G - 1);
    const int iy = cell_index<POW2>(y, p.y0, p.res, p.inv_res, p.G - 1);
    if (LDSWIN) {
        const int li = min(max(ix - w.wx0, 0), p.WN - 1);
        const int lj = min(max(iy - w.wy0, 0), p.WN - 1);
        return win[lj * p.WN + li];
    }
    return 1.0f - clampf(map[(size_t)iy * p.G + ix], 0.0f, 1.0f);
}

// One UnicycleModel.transit (robot_model.py:59-100).  (x,y,th) enter as the
// clamped state t and leave as the clamped/wrapped state t+1; (xn,yn,tn) is what
// the reference leaves in slot t (un-clamped, un-wrapped).  `trav` is the
// traversability at (x,y) on entry.
__device__ __forceinline__ void transit_step(const SolveParams &p, float trav, float u0, float u1,
                                             float &x, float &y, float &th, float &xn, float &yn, float &tn)
{
    float sn, cs;
    sincos_spec(th, sn, cs);
    const float tv = trav * u0;          // u0,u1 already lie in [u_min,u_max]: the re-clamp of :82-83 is the identity
    xn = x + (tv * cs) * p.dt;           // :86
    yn = y + (tv * sn) * p.dt;           // :87
    tn = th + (trav * u1) * p.dt;        // :88
    x = clampf(xn, p.x0, p.x_hi);        // :93
    y = clampf(yn, p.y0, p.y_hi);        // :94
    th = wrap_angle(tn);                 // :90
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int EPS>
__device__ __forceinline__ void load_eps_chunk(const SolveParams &p, const float *__restrict__ eps, int b, int kk,
                                               int t0, uint64_t solve, float (&e)[TU][2])
{
    if (EPS == kEpsPhilox) {
#pragma unroll
        for (int i = 0; i < TU / 2; ++i) {
            float z[4];
            philox_eps_pair(p.seed, solve, (uint32_t)b, (uint32_t)kk, (uint32_t)(t0 / 2 + i), z);
            e[2 * i][0] = z[0]; e[2 * i][1] = z[1];
            e[2 * i + 1][0] = z[2]; e[2 * i + 1][1] = z[3];
        }
    } else if (EPS == kEpsKT2) {
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            const int t = min(t0 + i, p.T - 1);
            const float2 v = *reinterpret_cast<const float2 *>(eps + (((size_t)b * p.K + kk) * p.T + t) * 2);
            e[i][0] = v.x; e[i][1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            const int t = min(t0 + i, p.T - 1);
            const size_t row = ((size_t)b * p.T + t) * 2;
            e[i][0] = eps[row * p.K + kk];
            e[i][1] = eps[(row + 1) * p.K + kk];
        }
    }
}

// ------------------------------------------------------------------------------
// Rollout + cost kernel.  grid = (ceil(K/64), B), block = 64 (one wavefront).
// LDS: [ window WN*WN | mean 2T | mean*inv_var 2T | control tile 2T x 65 | e 64 ]
// ------------------------------------------------------------------------------
template <int EPS, bool POW2, bool LDSWIN>
__global__ __launch_bounds__(kRolloutsPerBlock) void rollout_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T, K = p.K;
    float *win = smem;
    float *ml = win + (LDSWIN ? p.WN * p.WN : 0);
    float *mv = ml + 2 * T;
    float *Ul = mv + 2 * T;
    float *el = Ul + 2 * T * kUPad;

    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int k = blockIdx.x * kRolloutsPerBlock + lane;
    const bool active = k < K;
    const int kk = active ? k : K - 1;

    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float *__restrict__ eps = p.eps;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];
    const uint64_t solve = (EPS == kEpsPhilox) ? (uint64_t)*p.counter : 0;

    float ecur[TU][2], enext[TU][2];
    load_eps_chunk<EPS>(p, eps, b, kk, 0, solve, ecur);

    Win w{0, 0};
    if (LDSWIN) {
        w = window_origin<POW2>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, lane, kRolloutsPerBlock);
    }
    for (int j = lane; j < 2 * T; j += kRolloutsPerBlock) {
        const float m = p.mean[(size_t)b * 2 * T + j];
        ml[j] = m;
        mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);      // mean[t] @ inv_cov (diagonal), mppi.py:179-180
    }
    __syncthreads();

    float x = sx, y = sy, th = sth;                   // mppi.py:160
    float trav = trav_lookup<POW2, LDSWIN>(p, win, map, w, x, y);
    double Sd = 0.0, Ad = 0.0;                        // fp64 accumulation of the fp32 terms (Arithmetic spec)
    float *Xk = p.X + (size_t)b * (T + 1) * 3 * K + k;
    float *Uk = p.U ? p.U + (size_t)b * T * 2 * K + k : nullptr;

    for (int t0 = 0; t0 < T; t0 += TU) {
        if (t0 + TU < T) load_eps_chunk<EPS>(p, eps, b, kk, t0 + TU, solve, enext);
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            const int t = t0 + i;
            if (t < T) {
                const float m0 = ml[2 * t], m1 = ml[2 * t + 1];
                // sampling: clamp(mean + sigma*eps, u_min, u_max)   mppi.py:152-157
                const float u0 = clampf(m0 + p.sigma0 * ecur[i][0], p.umin0, p.umax0);
                const float u1 = clampf(m1 + p.sigma1 * ecur[i][1], p.umin1, p.umax1);
                Ul[(2 * t) * kUPad + lane] = u0;
                Ul[(2 * t + 1) * kUPad + lane] = u1;
                if (Uk && active) {
                    Uk[(size_t)(2 * t) * K] = u0;
                    Uk[(size_t)(2 * t + 1) * K] = u1;
                }
                // control cost  mean[t] @ inv_cov @ u   mppi.py:178-182
                const float a = mv[2 * t] * u0 + mv[2 * t + 1] * u1;
                Ad += (double)(p.lambda_ * a);
                float xn, yn, tn;
                transit_step(p, trav, u0, u1, x, y, th, xn, yn, tn);
                if (active) {                          // slot t keeps the un-clamped state (aliasing)
                    Xk[(size_t)(3 * t + 0) * K] = xn;
                    Xk[(size_t)(3 * t + 1) * K] = yn;
                    Xk[(size_t)(3 * t + 2) * K] = tn;
                }
                // The cell of the un-clamped slot equals the cell of the clamped state
                // (index clamp, grid_map.py:209), so one gather serves stage cost t and transit t+1.
                trav = trav_lookup<POW2, LDSWIN>(p, win, map, w, x, y);
                const float dx = xn - gx, dy = yn - gy;
                const float s = sqrtf(dx * dx + dy * dy) + (trav <= p.thr ? 1.0e4f : 0.0f);   // objectives.py:47-53
                Sd += (double)s;
            }
        }
#pragma unroll
        for (int i = 0; i < TU; ++i) { ecur[i][0] = enext[i][0]; ecur[i][1] = enext[i][1]; }
    }
    if (active) {                                      // slot T: clamped / wrapped state
        Xk[(size_t)(3 * T + 0) * K] = x;
        Xk[(size_t)(3 * T + 1) * K] = y;
        Xk[(size_t)(3 * T + 2) * K] = th;
    }
    const float dxT = x - gx, dyT = y - gy;
    const float term = sqrtf(dxT * dxT + dyT * dyT) + (trav <= p.thr ? 1.0e4f : 0.0f);        // mppi.py:184
    const float c = ((float)Sd + term) + (float)Ad;                                            // mppi.py:186-190
    if (active) p.cost[(size_t)b * K + k] = c;

    // block-local softmin statistics   mppi.py:193-199
    const float z = active ? (-c) / p.lambda_ : -INFINITY;
    const float zmax = wave_max(z);
    const float e = active ? expf(z - zmax) : 0.0f;
    const float esum = wave_sum(e);
    el[lane] = e;
    __syncthreads();
    float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
    for (int j = lane; j < 2 * T; j += kRolloutsPerBlock) {
        const float *col = Ul + j * kUPad;
        float acc = 0.0f;
#pragma unroll 16
        for (int q = 0; q < kRolloutsPerBlock; ++q) acc = __builtin_fmaf(el[q], col[q], acc);
        part[2 + j] = acc;
    }
    if (lane == 0) { part[0] = zmax; part[1] = esum; }
}

// ------------------------------------------------------------------------------
// Finish kernel.  grid = B, block = 256.  Merges the per-block statistics,
// writes U* (and the next mean), the normalised weights, and rolls out X*.
// LDS: [ window | ustar 2T | scale nblk | red 256 ]
// ------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, float *red, int tid, bool is_max)
{
    v = is_max ? wave_max(v) : wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < kFinishThreads / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    __syncthreads();
    return r;
}

template <bool POW2, bool LDSWIN>
__global__ __launch_bounds__(kFinishThreads) void finish_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T, K = p.K, nblk = p.nblk, PS = 2 + 2 * p.T;
    float *win = smem;
    float *us = win + (LDSWIN ? p.WN * p.WN : 0);
    float *sc = us + 2 * T;
    float *red = sc + nblk;

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float *part = p.part + (size_t)b * nblk * PS;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];

    Win w{0, 0};
    if (LDSWIN) {
        w = window_origin<POW2>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, tid, kFinishThreads);
    }

    float m = -INFINITY;
    for (int i = tid; i < nblk; i += kFinishThreads) m = fmaxf(m, part[(size_t)i * PS]);
    m = block_reduce(m, red, tid, true);
    float s = 0.0f;
    for (int i = tid; i < nblk; i += kFinishThreads) {
        const float f = expf(part[(size_t)i * PS] - m);
        sc[i] = f;
        s += part[(size_t)i * PS + 1] * f;
    }
    const float S = block_reduce(s, red, tid, false);   // also publishes sc[] (barrier inside)

    for (int j = tid; j < 2 * T; j += kFinishThreads) {
        float acc = 0.0f;
        for (int i = 0; i < nblk; ++i) acc = __builtin_fmaf(part[(size_t)i * PS + 2 + j], sc[i], acc);
        const float u = acc / S;                         // sum_k w_k u_k, mppi.py:196-199
        us[j] = u;
        p.ustar[(size_t)b * 2 * T + j] = u;
        p.mean[(size_t)b * 2 * T + j] = u;               // _previous_action_seq = U*, no shift (mppi.py:217)
    }
    if (tid == 0) {
        p.stats[b * 2 + 0] = m;
        p.stats[b * 2 + 1] = S;
        if (b == 0) *p.counter += 1ull;
    }
    __syncthreads();

    if (tid == 0) {
        // optimal_state_seq: batch-1 rollout of U* with the same aliasing (mppi.py:202-214)
        float x = sx, y = sy, th = sth;
        float trav = trav_lookup<POW2, LDSWIN>(p, win, map, w, x, y);
        float *Xs = p.xstar + (size_t)b * (T + 1) * 3;
        for (int t = 0; t < T; ++t) {
            float xn, yn, tn;
            transit_step(p, trav, us[2 * t], us[2 * t + 1], x, y, th, xn, yn, tn);
            Xs[3 * t + 0] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
            trav = trav_lookup<POW2, LDSWIN>(p, win, map, w, x, y);
        }
        Xs[3 * T + 0] = x; Xs[3 * T + 1] = y; Xs[3 * T + 2] = th;
    } else if (tid >= 64) {
        // _weights = softmax(-costs / lambda)   mppi.py:193
        const float *cost = p.cost + (size_t)b * K;
        float *wout = p.w + (size_t)b * K;
        for (int k = tid - 64; k < K; k += kFinishThreads - 64)
            wout[k] = expf((-cost[k]) / p.lambda_ - m) / S;
    }
}

// ---- layout helpers -----------------------------------------------------------
__global__ void soa_to_aos_kernel(const float *__restrict__ in, float *__restrict__ out, int K, int R)
{   // in (R, K) -> out (K, R)
    const size_t n = (size_t)K * R;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t k = i / R, r = i - k * R;
        out[i] = in[r * K + k];
    }
}

__global__ void gather_states_kernel(const float *__restrict__ X, const int *__restrict__ idx,
                                     float *__restrict__ out, int n, int K, int R)
{   // out (n, R) = X (R, K)[:, idx]
    const size_t tot = (size_t)n * R;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const size_t q = i / R, r = i - q * R;
        out[i] = X[r * K + idx[q]];
    }
}

__global__ void philox_noise_kernel(float *__restrict__ eps, uint64_t seed, uint64_t solve, int b, int K, int T)
{   // eps (K, T, 2) of one instance, exactly the stream rollout_kernel<kEpsPhilox> consumes
    const int npair = (T + 1) / 2;
    const size_t tot = (size_t)K * npair;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / npair), tp = (int)(i - (size_t)k * npair);
        float z[4];
        philox_eps_pair(seed, solve, (uint32_t)b, (uint32_t)k, (uint32_t)tp, z);
        const int t = 2 * tp;
        eps[((size_t)k * T + t) * 2 + 0] = z[0];
        eps[((size_t)k * T + t) * 2 + 1] = z[1];
        if (t + 1 < T) {
            eps[((size_t)k * T + t + 1) * 2 + 0] = z[2];
            eps[((size_t)k * T + t + 1) * 2 + 1] = z[3];
        }
    }
}

template <typename Kern>
hipError_t ensure_lds(Kern kern, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int EPS, bool POW2, bool LDSWIN>
hipError_t launch_rollout_t(const SolveParams &p, hipStream_t s)
{
    const size_t lds = rollout_lds_bytes(p);
    hipError_t e = ensure_lds(rollout_kernel<EPS, POW2, LDSWIN>, lds);
    if (e != hipSuccess) return e;
    rollout_kernel<EPS, POW2, LDSWIN><<<dim3(p.nblk, p.B), dim3(kRolloutsPerBlock), lds, s>>>(p);
    return hipGetLastError();
}

template <int EPS>
hipError_t launch_rollout_e(const SolveParams &p, hipStream_t s)
{
    const bool win = p.WN > 0;
    if (p.pow2) return win ? launch_rollout_t<EPS, true, true>(p, s) : launch_rollout_t<EPS, true, false>(p, s);
    return win ? launch_rollout_t<EPS, false, true>(p, s) : launch_rollout_t<EPS, false, false>(p, s);
}

template <bool POW2, bool LDSWIN>
hipError_t launch_finish_t(const SolveParams &p, hipStream_t s)
{
    const size_t lds = finish_lds_bytes(p);
    hipError_t e = ensure_lds(finish_kernel<POW2, LDSWIN>, lds);
    if (e != hipSuccess) return e;
    finish_kernel<POW2, LDSWIN><<<dim3(p.B), dim3(kFinishThreads), lds, s>>>(p);
    return hipGetLastError();
}

}  // namespace

size_t rollout_lds_bytes(const SolveParams &p)
{
    return sizeof(float) * ((size_t)p.WN * p.WN + 4 * (size_t)p.T + 2 * (size_t)p.T * kUPad + kRolloutsPerBlock);
}

size_t finish_lds_bytes(const SolveParams &p)
{
    return sizeof(float) * ((size_t)p.WN * p.WN + 2 * (size_t)p.T + (size_t)p.nblk + kFinishThreads);
}

hipError_t launch_rollout(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    switch (mode) {
    case kEpsPhilox: return launch_rollout_e<kEpsPhilox>(p, s);
    case kEpsKT2: return launch_rollout_e<kEpsKT2>(p, s);
    default: return launch_rollout_e<kEpsT2K>(p, s);
    }
}

hipError_t launch_finish(const SolveParams &p, hipStream_t s)
{
    const bool win = p.WN > 0;
    if (p.pow2) return win ? launch_finish_t<true, true>(p, s) : launch_finish_t<true, false>(p, s);
    return win ? launch_finish_t<false, true>(p, s) : launch_finish_t<false, false>(p, s);
}

static int grid_for(size_t n) { return (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256); }

hipError_t launch_states_to_reference(const float *X_soa, float *X_aos, int K, int T1, hipStream_t s)
{
    soa_to_aos_kernel<<<grid_for((size_t)K * T1 * 3), 256, 0, s>>>(X_soa, X_aos, K, T1 * 3);
    return hipGetLastError();
}

hipError_t launch_controls_to_reference(const float *U_soa, float *U_aos, int K, int T, hipStream_t s)
{
    soa_to_aos_kernel<<<grid_for((size_t)K * T * 2), 256, 0, s>>>(U_soa, U_aos, K, T * 2);
    return hipGetLastError();
}

hipError_t launch_gather_states(const float *X_soa, const int *idx, float *out, int n, int K, int T1, hipStream_t s)
{
    gather_states_kernel<<<grid_for((size_t)n * T1 * 3), 256, 0, s>>>(X_soa, idx, out, n, K, T1 * 3);
    return hipGetLastError();
}

hipError_t launch_philox_noise(float *eps_kt2, uint64_t seed, uint64_t solve, int b, int K, int T, hipStream_t s)
{
    philox_noise_kernel<<<grid_for((size_t)K * ((T + 1) / 2)), 256, 0, s>>>(eps_kt2, seed, solve, b, K, T);
    return hipGetLastError();
}

}  // namespace bn
