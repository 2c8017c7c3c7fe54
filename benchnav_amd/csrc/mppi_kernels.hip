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

// Timing ablations for tools/ablate.py (never set in the shipped library): bit 0 skip stage cost,
// 1 skip fp64 accumulation, 2 skip X stores, 3 skip control tile + control cost, 4 skip sincos,
// 5 skip the gather, 6 skip the heading wrap.
#ifndef BN_ABLATE
#define BN_ABLATE 0
#endif
#define BN_KEEP(v) asm volatile("" ::"v"(v))
#ifdef BN_TIMING
#define BN_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                          \
            p.stamps[slot] = __builtin_readcyclecounter();                                               \
    } while (0)
#else
#define BN_STAMP(slot) do { } while (0)
#endif

// Geometry specialisations of the cell index ((p - origin) / res).floor().int()  (grid_map.py:195-209):
//   kGeoGeneral  true division            kGeoPow2  res is a power of two: * (1/res) is bit-identical
//   kGeoPow2Origin0  additionally origin == 0, so the subtraction is the identity
enum Geo : int { kGeoGeneral = 0, kGeoPow2 = 1, kGeoPow2Origin0 = 2 };

struct Win { int wx0, wy0; float fx0, fy0, fwn, fwm1; };   // window origin (cells), as floats, edge, edge-1

template <int GEO>
__device__ __forceinline__ int raw_cell(float v, float origin, float res, float inv_res)
{
    const float q = (GEO == kGeoPow2Origin0) ? v * inv_res : (GEO == kGeoPow2) ? (v - origin) * inv_res : (v - origin) / res;
    return (int)floorf(q);                    // v_cvt_i32_f32 saturates
}

template <int GEO>
__device__ __forceinline__ Win window_origin(const SolveParams &p, float sx, float sy)
{
    const int cx = clampi(raw_cell<GEO>(sx, p.x0, p.res, p.inv_res), 0, p.G - 1);
    const int cy = clampi(raw_cell<GEO>(sy, p.y0, p.res, p.inv_res), 0, p.G - 1);
    Win w;
    w.wx0 = min(max(cx - p.reach, 0), p.G - p.WN);
    w.wy0 = min(max(cy - p.reach, 0), p.G - p.WN);
    w.fx0 = (float)w.wx0; w.fy0 = (float)w.wy0; w.fwn = (float)p.WN; w.fwm1 = (float)(p.WN - 1);
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

// Traversability at (x, y): index clamp (grid_map.py:209) then the gather.  With the LDS window the
// two clamps (map, then window) collapse into one: the window lies inside the map, so clamping
// i - wx0 to [0, WN-1] gives the same cell for every i (in, left of, or right of the map).
// SAFE additionally bounds the raw index first, for a caller-supplied start state of any magnitude.
template <int GEO, bool LDSWIN, bool SAFE>
__device__ __forceinline__ float trav_lookup(const SolveParams &p, const float *win,
                                             const float *__restrict__ map, const Win w, float x, float y)
{
    int ix = raw_cell<GEO>(x, p.x0, p.res, p.inv_res);
    int iy = raw_cell<GEO>(y, p.y0, p.res, p.inv_res);
    if (LDSWIN) {
        if (SAFE) { ix = clampi(ix, 0, p.G - 1); iy = clampi(iy, 0, p.G - 1); }
        const int li = clampi(ix - w.wx0, 0, p.WN - 1);
        const int lj = clampi(iy - w.wy0, 0, p.WN - 1);
        return win[(int)__umul24((unsigned)lj, (unsigned)p.WN) + li];
    }
    ix = clampi(ix, 0, p.G - 1);
    iy = clampi(iy, 0, p.G - 1);
    return 1.0f - clampf(map[(size_t)iy * p.G + ix], 0.0f, 1.0f);
}

// In-loop gather: (x, y) already lies inside the map limits.  The window-relative cell is computed in
// the float domain: q = (x - origin)/res as the reference rounds it, then q - wx0 (exact: an integer
// no larger than q is subtracted), floor, clamp to the window, row * WN + col (exact small integers),
// one conversion.  Same cell as trav_lookup<..., false> for every in-limits position.
template <int GEO>
__device__ __forceinline__ float trav_window(const SolveParams &p, const float *win, const Win w, float x, float y)
{
    float qx, qy;
    if (GEO == kGeoPow2Origin0) {
        qx = __builtin_fmaf(x, p.inv_res, -w.fx0);
        qy = __builtin_fmaf(y, p.inv_res, -w.fy0);
    } else if (GEO == kGeoPow2) {
        qx = __builtin_fmaf(x - p.x0, p.inv_res, -w.fx0);
        qy = __builtin_fmaf(y - p.y0, p.inv_res, -w.fy0);
    } else {
        qx = (x - p.x0) / p.res - w.fx0;
        qy = (y - p.y0) / p.res - w.fy0;
    }
    const float li = clampf(floorf(qx), 0.0f, w.fwm1);
    const float lj = clampf(floorf(qy), 0.0f, w.fwm1);
    return win[(int)__builtin_fmaf(lj, w.fwn, li)];
}

// Per-rollout recurrence state: the clamped/wrapped state t, its traversability, sin/cos of its heading.
struct Chain { float x, y, th, sn, cs, trav; };

// One UnicycleModel.transit (robot_model.py:59-100) plus the gather for the next step.
// (xn, yn, tn) is what the reference leaves in slot t (un-clamped, un-wrapped); the chain
// advances to the clamped/wrapped state t+1.  The two dependent strands -- heading (wrap, sincos) and
// position (clamp, cell index, LDS gather) -- are independent after `trav` and overlap in issue.
// u0, u1 already lie in [u_min, u_max]: the re-clamp of robot_model.py:82-83 is the identity.
template <int GEO, bool LDSWIN, bool FIRST>
__device__ __forceinline__ void chain_step(const SolveParams &p, const float *win, const float *__restrict__ map,
                                           const Win w, Chain &c, float u0, float u1, float &xn, float &yn, float &tn)
{
    const float tv = c.trav * u0;
    tn = c.th + (c.trav * u1) * p.dt;                                  // :88
    xn = c.x + (tv * c.cs) * p.dt;                                     // :86
    yn = c.y + (tv * c.sn) * p.dt;                                     // :87
    if (BN_ABLATE & 64) c.th = tn; else
    c.th = FIRST ? wrap_angle(tn) : wrap_angle_near(tn);               // :90
    c.x = clampf(xn, p.x0, p.x_hi);                                    // :93
    c.y = clampf(yn, p.y0, p.y_hi);                                    // :94
    if (BN_ABLATE & 16) { c.sn = c.th * 0.5f; c.cs = 1.0f - c.th; } else
    sincos_spec(c.th, c.sn, c.cs);
    if (BN_ABLATE & 32) { c.trav = 0.5f + 0.001f * c.x; } else
    c.trav = LDSWIN ? trav_window<GEO>(p, win, w, c.x, c.y) : trav_lookup<GEO, false, false>(p, win, map, w, c.x, c.y);
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

// Noise of steps t0 .. t0+TU-1 (t0 odd: step 0 is peeled) for rollout kk of instance b.
template <int EPS>
__device__ __forceinline__ void load_eps_chunk(const SolveParams &p, const float *__restrict__ eps, int b, int kk,
                                               int t0, uint64_t solve, float (&e)[TU][2])
{
    if (EPS == kEpsPhilox) {
#pragma unroll
        for (int i = 0; i < TU / 2; ++i) {
            float z[4];
            philox_eps_pair(p.seed, solve, (uint32_t)b, (uint32_t)kk, (uint32_t)((t0 + 1) / 2 + i), z);
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
            const float *row = eps + ((size_t)b * p.T + t) * 2 * p.K;
            e[i][0] = row[kk];
            e[i][1] = row[p.K + kk];
        }
    }
}

template <int EPS>
__device__ __forceinline__ void load_eps_step0(const SolveParams &p, const float *__restrict__ eps, int b, int kk,
                                               uint64_t solve, float (&e)[2])
{
    if (EPS == kEpsPhilox) {
        float z[4];
        philox_eps_pair(p.seed, solve, (uint32_t)b, (uint32_t)kk, 0u, z);
        e[0] = z[2]; e[1] = z[3];
    } else if (EPS == kEpsKT2) {
        const float2 v = *reinterpret_cast<const float2 *>(eps + ((size_t)b * p.K + kk) * p.T * 2);
        e[0] = v.x; e[1] = v.y;
    } else {
        const float *row = eps + (size_t)b * p.T * 2 * p.K;
        e[0] = row[kk];
        e[1] = row[p.K + kk];
    }
}

// ------------------------------------------------------------------------------
// Rollout + cost kernel.  grid = (ceil(K/64), B), block = 64 (one wavefront).
// LDS: [ window WN*WN | mean 2T | mean*inv_var 2T | control tile 2T x 65 | e 64 ]
// ------------------------------------------------------------------------------
template <int EPS, int GEO, bool LDSWIN, bool STORE_U>
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
    BN_STAMP(0);

    float e0[2];
    float ecur[TU][2], enext[TU][2];
    load_eps_step0<EPS>(p, eps, b, kk, solve, e0);
    load_eps_chunk<EPS>(p, eps, b, kk, 1, solve, ecur);

    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, lane, kRolloutsPerBlock);
    }
    for (int j = lane; j < 2 * T; j += kRolloutsPerBlock) {
        const float m = p.mean[(size_t)b * 2 * T + j];
        ml[j] = m;
        mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);      // mean[t] @ inv_cov (diagonal), mppi.py:179-180
    }
    __syncthreads();
    BN_STAMP(1);

    Chain c;
    c.x = sx; c.y = sy; c.th = sth;                   // mppi.py:160
    sincos_spec(c.th, c.sn, c.cs);
    c.trav = trav_lookup<GEO, LDSWIN, true>(p, win, map, w, c.x, c.y);
    double Sd = 0.0, Ad = 0.0;                        // fp64 accumulation of the fp32 terms (Arithmetic spec)
    float Sf = 0.0f, Af = 0.0f;                       // (ablation builds only)
    // Rows of X and U are pitched to Kp = 64 * nblk floats, so every lane stores unconditionally
    // (lanes past K write into the pad) and the step body stays one basic block.
    const size_t Kp = (size_t)p.Kp;
    float *Xb = p.X + (size_t)b * (T + 1) * 3 * Kp + k;
    float *Ub = STORE_U ? p.U + (size_t)b * T * 2 * Kp + k : nullptr;

    // One time step for this lane's rollout: sample the control, advance the chain, emit slot t,
    // accumulate the control cost and the stage cost of the aliased slot.
#define BN_STEP(FIRST, t, eps0, eps1)                                                                          \
    do {                                                                                                       \
        const float m0 = ml[2 * (t)], m1 = ml[2 * (t) + 1];                                                    \
        const float u0 = clampf(m0 + p.sigma0 * (eps0), p.umin0, p.umax0);   /* mppi.py:152-157 */             \
        const float u1 = clampf(m1 + p.sigma1 * (eps1), p.umin1, p.umax1);                                     \
        float xn, yn, tn;                                                                                      \
        chain_step<GEO, LDSWIN, FIRST>(p, win, map, w, c, u0, u1, xn, yn, tn);                                 \
        if (!(BN_ABLATE & 8)) {                                                                                \
        Ul[(2 * (t)) * kUPad + lane] = u0;                                                                     \
        Ul[(2 * (t) + 1) * kUPad + lane] = u1;                                                                 \
        }                                                                                                      \
        if (!(BN_ABLATE & 4)) {                                                                                \
        float *Xt = Xb + (size_t)(3 * (t)) * Kp;         /* slot t keeps the un-clamped state (aliasing) */     \
        Xt[0] = xn; Xt[Kp] = yn; Xt[2 * Kp] = tn;                                                              \
        } else { BN_KEEP(xn); BN_KEEP(yn); BN_KEEP(tn); }                                                      \
        if (STORE_U) { float *Ut = Ub + (size_t)(2 * (t)) * Kp; Ut[0] = u0; Ut[Kp] = u1; }                     \
        if (!(BN_ABLATE & 8)) {                                                                                \
        const float a = mv[2 * (t)] * u0 + mv[2 * (t) + 1] * u1;            /* mppi.py:178-182 */             \
        if (BN_ABLATE & 2) Af += p.lambda_ * a; else                                                           \
        Ad += (double)(p.lambda_ * a);                                                                         \
        }                                                                                                      \
        /* the cell of the un-clamped slot equals the cell of the clamped state (index clamp,               */ \
        /* grid_map.py:209), so c.trav serves stage cost t and transit t+1                                  */ \
        if (!(BN_ABLATE & 1)) {                                                                                \
        const float dx = xn - gx, dy = yn - gy;                                                                \
        const float sc = sqrtf(dx * dx + dy * dy) + (c.trav <= p.thr ? 1.0e4f : 0.0f);   /* objectives.py:47-53 */ \
        if (BN_ABLATE & 2) Sf += sc; else                                                                      \
        Sd += (double)sc;                                                                                      \
        }                                                                                                      \
    } while (0)

    BN_STEP(true, 0, e0[0], e0[1]);
    BN_STAMP(2);
    int t0 = 1;
    for (; t0 + TU <= T; t0 += TU) {
        load_eps_chunk<EPS>(p, eps, b, kk, t0 + TU, solve, enext);      // indices clamp at T-1: always in bounds
#pragma unroll
        for (int i = 0; i < TU; ++i) BN_STEP(false, t0 + i, ecur[i][0], ecur[i][1]);
#pragma unroll
        for (int i = 0; i < TU; ++i) { ecur[i][0] = enext[i][0]; ecur[i][1] = enext[i][1]; }
    }
#pragma unroll
    for (int i = 0; i < TU - 1; ++i)
        if (t0 + i < T) BN_STEP(false, t0 + i, ecur[i][0], ecur[i][1]);
#undef BN_STEP
    BN_STAMP(3);

    {                                                  // slot T: clamped / wrapped state
        float *Xt = Xb + (size_t)(3 * T) * Kp;
        Xt[0] = c.x; Xt[Kp] = c.y; Xt[2 * Kp] = c.th;
    }
    const float dxT = c.x - gx, dyT = c.y - gy;
    const float term = sqrtf(dxT * dxT + dyT * dyT) + (c.trav <= p.thr ? 1.0e4f : 0.0f);     // mppi.py:184
#if BN_ABLATE
    const float cost = (((float)Sd + Sf) + term) + ((float)Ad + Af);
#else
    const float cost = ((float)Sd + term) + (float)Ad;                                         // mppi.py:186-190
#endif
    if (active) p.cost[(size_t)b * K + k] = cost;

    // block-local softmin statistics   mppi.py:193-199
    const float z = active ? (-cost) / p.lambda_ : -INFINITY;
    const float zmax = wave_max(z);
    const float e = active ? expf(z - zmax) : 0.0f;
    const float esum = wave_sum(e);
    el[lane] = e;
    __syncthreads();
    BN_STAMP(4);
    float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
    for (int j = lane; j < 2 * T; j += kRolloutsPerBlock) {
        const float *col = Ul + j * kUPad;
        float acc = 0.0f;
#pragma unroll 16
        for (int q = 0; q < kRolloutsPerBlock; ++q) acc = __builtin_fmaf(el[q], col[q], acc);
        part[2 + j] = acc;
    }
    if (lane == 0) { part[0] = zmax; part[1] = esum; }
    BN_STAMP(5);
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

template <int GEO, bool LDSWIN>
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
    BN_STAMP(8);

    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, tid, kFinishThreads);
    }
    BN_STAMP(9);

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
    BN_STAMP(10);

    if (tid == 0) {
        // optimal_state_seq: batch-1 rollout of U* with the same aliasing (mppi.py:202-214)
        Chain c;
        c.x = sx; c.y = sy; c.th = sth;
        sincos_spec(c.th, c.sn, c.cs);
        c.trav = trav_lookup<GEO, LDSWIN, true>(p, win, map, w, c.x, c.y);
        float *Xs = p.xstar + (size_t)b * (T + 1) * 3;
        float xn, yn, tn;
        chain_step<GEO, LDSWIN, true>(p, win, map, w, c, us[0], us[1], xn, yn, tn);
        Xs[0] = xn; Xs[1] = yn; Xs[2] = tn;
        for (int t = 1; t < T; ++t) {
            chain_step<GEO, LDSWIN, false>(p, win, map, w, c, us[2 * t], us[2 * t + 1], xn, yn, tn);
            Xs[3 * t + 0] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
        }
        Xs[3 * T + 0] = c.x; Xs[3 * T + 1] = c.y; Xs[3 * T + 2] = c.th;
        BN_STAMP(11);
    } else if (tid >= 64) {
        // _weights = softmax(-costs / lambda)   mppi.py:193
        const float *cost = p.cost + (size_t)b * K;
        float *wout = p.w + (size_t)b * K;
        for (int k = tid - 64; k < K; k += kFinishThreads - 64)
            wout[k] = expf((-cost[k]) / p.lambda_ - m) / S;
    }
}

// ---- layout helpers -----------------------------------------------------------
__global__ void soa_to_aos_kernel(const float *__restrict__ in, float *__restrict__ out, int K, int Kp, int R)
{   // in (R, Kp pitch) -> out (K, R)
    const size_t n = (size_t)K * R;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t k = i / R, r = i - k * R;
        out[i] = in[r * Kp + k];
    }
}

__global__ void gather_states_kernel(const float *__restrict__ X, const int *__restrict__ idx,
                                     float *__restrict__ out, int n, int Kp, int R)
{   // out (n, R) = X (R, Kp pitch)[:, idx]
    const size_t tot = (size_t)n * R;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const size_t q = i / R, r = i - q * R;
        out[i] = X[r * Kp + idx[q]];
    }
}

__global__ void philox_noise_kernel(float *__restrict__ eps, uint64_t seed, uint64_t solve, int b, int K, int T)
{   // eps (K, T, 2) of one instance, exactly the stream rollout_kernel<kEpsPhilox> consumes
    const int npair = T / 2 + 1;                    // pair p = steps (2p-1, 2p)
    const size_t tot = (size_t)K * npair;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / npair), pr = (int)(i - (size_t)k * npair);
        float z[4];
        philox_eps_pair(seed, solve, (uint32_t)b, (uint32_t)k, (uint32_t)pr, z);
        const int t = 2 * pr - 1;
        if (t >= 0) {
            eps[((size_t)k * T + t) * 2 + 0] = z[0];
            eps[((size_t)k * T + t) * 2 + 1] = z[1];
        }
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

int geo_of(const SolveParams &p)
{
    if (!p.pow2) return kGeoGeneral;
    return (p.x0 == 0.0f && p.y0 == 0.0f) ? kGeoPow2Origin0 : kGeoPow2;
}

template <int EPS, int GEO, bool LDSWIN, bool STORE_U>
hipError_t launch_rollout_u(const SolveParams &p, hipStream_t s)
{
    const size_t lds = rollout_lds_bytes(p);
    hipError_t e = ensure_lds(rollout_kernel<EPS, GEO, LDSWIN, STORE_U>, lds);
    if (e != hipSuccess) return e;
    rollout_kernel<EPS, GEO, LDSWIN, STORE_U><<<dim3(p.nblk, p.B), dim3(kRolloutsPerBlock), lds, s>>>(p);
    return hipGetLastError();
}

template <int EPS, int GEO, bool LDSWIN>
hipError_t launch_rollout_t(const SolveParams &p, hipStream_t s)
{
    return p.U ? launch_rollout_u<EPS, GEO, LDSWIN, true>(p, s) : launch_rollout_u<EPS, GEO, LDSWIN, false>(p, s);
}

template <int EPS>
hipError_t launch_rollout_e(const SolveParams &p, hipStream_t s)
{
    const bool win = p.WN > 0;
    switch (geo_of(p)) {
    case kGeoPow2Origin0: return win ? launch_rollout_t<EPS, kGeoPow2Origin0, true>(p, s) : launch_rollout_t<EPS, kGeoPow2Origin0, false>(p, s);
    case kGeoPow2: return win ? launch_rollout_t<EPS, kGeoPow2, true>(p, s) : launch_rollout_t<EPS, kGeoPow2, false>(p, s);
    default: return win ? launch_rollout_t<EPS, kGeoGeneral, true>(p, s) : launch_rollout_t<EPS, kGeoGeneral, false>(p, s);
    }
}

template <int GEO, bool LDSWIN>
hipError_t launch_finish_t(const SolveParams &p, hipStream_t s)
{
    const size_t lds = finish_lds_bytes(p);
    hipError_t e = ensure_lds(finish_kernel<GEO, LDSWIN>, lds);
    if (e != hipSuccess) return e;
    finish_kernel<GEO, LDSWIN><<<dim3(p.B), dim3(kFinishThreads), lds, s>>>(p);
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
    switch (geo_of(p)) {
    case kGeoPow2Origin0: return win ? launch_finish_t<kGeoPow2Origin0, true>(p, s) : launch_finish_t<kGeoPow2Origin0, false>(p, s);
    case kGeoPow2: return win ? launch_finish_t<kGeoPow2, true>(p, s) : launch_finish_t<kGeoPow2, false>(p, s);
    default: return win ? launch_finish_t<kGeoGeneral, true>(p, s) : launch_finish_t<kGeoGeneral, false>(p, s);
    }
}

static int grid_for(size_t n) { return (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256); }

hipError_t launch_states_to_reference(const float *X_soa, float *X_aos, int K, int Kp, int T1, hipStream_t s)
{
    soa_to_aos_kernel<<<grid_for((size_t)K * T1 * 3), 256, 0, s>>>(X_soa, X_aos, K, Kp, T1 * 3);
    return hipGetLastError();
}

hipError_t launch_controls_to_reference(const float *U_soa, float *U_aos, int K, int Kp, int T, hipStream_t s)
{
    soa_to_aos_kernel<<<grid_for((size_t)K * T * 2), 256, 0, s>>>(U_soa, U_aos, K, Kp, T * 2);
    return hipGetLastError();
}

hipError_t launch_gather_states(const float *X_soa, const int *idx, float *out, int n, int Kp, int T1, hipStream_t s)
{
    gather_states_kernel<<<grid_for((size_t)n * T1 * 3), 256, 0, s>>>(X_soa, idx, out, n, Kp, T1 * 3);
    return hipGetLastError();
}

hipError_t launch_philox_noise(float *eps_kt2, uint64_t seed, uint64_t solve, int b, int K, int T, hipStream_t s)
{
    philox_noise_kernel<<<grid_for((size_t)K * (T / 2 + 1)), 256, 0, s>>>(eps_kt2, seed, solve, b, K, T);
    return hipGetLastError();
}

}  // namespace bn
