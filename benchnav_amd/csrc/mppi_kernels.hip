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
#include <algorithm>

namespace bn {

namespace {

constexpr int TU = kChunk;   // time steps per phase (chunk): chain works on chunk c, consumers on c-1, producers on c+2

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
#define BN_STAMP_ANY(slot)                                                                               \
    do {                                                                                                 \
        if (p.stamps && blockIdx.y == 0 && threadIdx.x == 0) p.stamps[slot] = __builtin_readcyclecounter(); \
    } while (0)
// per-workgroup trace (tools/block_trace.py): wall clock (100 MHz, chip-wide) at entry and exit, cycles, HW_ID
#define BN_TRACE_BEGIN()                                                                                 \
    const unsigned long long bn_tr_t0 = wall_clock64(), bn_tr_c0 = __builtin_readcyclecounter()
#define BN_TRACE_END()                                                                                   \
    do {                                                                                                 \
        if (p.stamps && threadIdx.x == 0) {                                                              \
            unsigned long long *r = p.stamps + 64 + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);   \
            r[0] = bn_tr_t0; r[1] = wall_clock64(); r[2] = __builtin_readcyclecounter() - bn_tr_c0;      \
            r[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); \
        }                                                                                                \
    } while (0)
#else
#define BN_STAMP(slot) do { } while (0)
#define BN_STAMP_ANY(slot) do { } while (0)
#define BN_TRACE_BEGIN() do { } while (0)
#define BN_TRACE_END() do { } while (0)
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
    v2f q;
    const v2f xy = {x, y}, ir = {p.inv_res, p.inv_res}, nw = {-w.fx0, -w.fy0};
    if (GEO == kGeoPow2Origin0) {
        q = __builtin_elementwise_fma(xy, ir, nw);                      // one v_pk_fma_f32
    } else if (GEO == kGeoPow2) {
        q = __builtin_elementwise_fma(xy - v2f{p.x0, p.y0}, ir, nw);
    } else {
        q = v2f{(x - p.x0) / p.res, (y - p.y0) / p.res} + nw;
    }
    const float li = clampf(floorf(q.x), 0.0f, w.fwm1);
    const float lj = clampf(floorf(q.y), 0.0f, w.fwm1);
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
    // Position strand first: update, clamp, cell index, and the gather goes out; the heading strand (wrap, sin/cos,
    // ~27 instructions) then runs under the gather's LDS latency.  The scheduling barrier keeps the compiler from
    // interleaving the two again (it used to issue the gather two thirds into the step).
    const float tv = c.trav * u0;
    const float dth = (c.trav * u1) * p.dt;
    // x and y advance in lockstep: packed multiply / multiply / add (same roundings as the scalar form)
    const v2f pos = v2f{c.x, c.y} + (v2f{tv, tv} * v2f{c.cs, c.sn}) * v2f{p.dt, p.dt};   // :86-87
    xn = pos.x;
    yn = pos.y;
    c.x = clampf(xn, p.x0, p.x_hi);                                    // :93
    c.y = clampf(yn, p.y0, p.y_hi);                                    // :94
    if (BN_ABLATE & 32) { c.trav = 0.5f + 0.001f * c.x; } else
    c.trav = LDSWIN ? trav_window<GEO>(p, win, w, c.x, c.y) : trav_lookup<GEO, false, false>(p, win, map, w, c.x, c.y);
    __builtin_amdgcn_sched_barrier(0);
    tn = c.th + dth;                                                   // :88
    if (BN_ABLATE & 64) c.th = tn; else
    c.th = FIRST ? wrap_angle(tn) : wrap_angle_near(tn);               // :90
    if (BN_ABLATE & 16) { c.sn = c.th * 0.5f; c.cs = 1.0f - c.th; } else
    sincos_spec(c.th, c.sn, c.cs);
}

// One PlanetaryEnv.step (planetary_env.py:189-219) for instance b: observation-mode transit with the
// latent slip sampled at the current cell (traversability_model.py:65-69: Normal(mean, std)[cell].sample()
// = z * std + mean), then the goal test.  An instance already within goal_thr of its goal is frozen.
// Every workgroup that needs the next state evaluates this itself: same inputs, same operations.
struct EnvStep { float x, y, th, reward; bool reached, frozen; };

template <int GEO>
__device__ __forceinline__ EnvStep env_advance(const SolveParams &p, int b, float sx, float sy, float sth, float u0, float u1,
                                               const float *z_ptr, uint64_t step)
{
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];
    EnvStep r;
    const float d0x = sx - gx, d0y = sy - gy;
    r.frozen = sqrt_cr(d0x * d0x + d0y * d0y) < p.goal_thr;          // terminated at an earlier step
    const int ix = clampi(raw_cell<GEO>(sx, p.x0, p.res, p.inv_res), 0, p.G - 1);
    const int iy = clampi(raw_cell<GEO>(sy, p.y0, p.res, p.inv_res), 0, p.G - 1);
    const size_t cell = (size_t)b * p.map_stride + (size_t)iy * p.G + ix;
    float z;
    if (z_ptr) {
        z = z_ptr[b];
    } else {
        const u32x4 q = philox4x32_10(u32x4{(uint32_t)b, (uint32_t)step, (uint32_t)(step >> 32), 0x454e5631u},
                                      (uint32_t)p.env_seed, (uint32_t)(p.env_seed >> 32));
        float z1;
        box_muller(q.x, q.y, z, z1);
    }
    const float slip = z * p.lat_std[cell] + p.lat_mean[cell];
    const float trav = 1.0f - clampf(slip, 0.0f, 1.0f);
    const float v = clampf(u0, p.umin0, p.umax0), om = clampf(u1, p.umin1, p.umax1);   // robot_model.py:82-83
    float sn, cs;
    sincos_spec(sth, sn, cs);
    const float xn = sx + ((trav * v) * cs) * p.env_dt;
    const float yn = sy + ((trav * v) * sn) * p.env_dt;
    const float tn = sth + (trav * om) * p.env_dt;
    r.x = r.frozen ? sx : clampf(xn, p.x0, p.x_hi);
    r.y = r.frozen ? sy : clampf(yn, p.y0, p.y_hi);
    r.th = r.frozen ? sth : wrap_angle(tn);
    r.reward = trav;
    const float dx = r.x - gx, dy = r.y - gy;
    r.reached = sqrt_cr(dx * dx + dy * dy) < p.goal_thr;             // planetary_env.py:215-217
    return r;
}

// Sampled-slip helpers (BASELINE config 3, see rollout_sampled_kernel): every lookup evaluates the observation-mode
// traversability 1 - clamp(z*std + mean, 0, 1) (traversability_model.py:65-69) with its own standard normal z.
__device__ __forceinline__ float trav_from_slip(float mu, float sd, float z)
{
    const float slip = z * sd + mu;                   // Normal.sample(): normal_(0,1).mul_(std).add_(mean)
    return 1.0f - clampf(slip, 0.0f, 1.0f);
}

// Cell of a position of any magnitude, as map index (iy * G + ix) or, with the window, window index.
template <int GEO, bool LDSWIN>
__device__ __forceinline__ int slip_cell_safe(const SolveParams &p, const Win w, float x, float y)
{
    const int ix = clampi(raw_cell<GEO>(x, p.x0, p.res, p.inv_res), 0, p.G - 1);
    const int iy = clampi(raw_cell<GEO>(y, p.y0, p.res, p.inv_res), 0, p.G - 1);
    if (LDSWIN) return clampi(iy - w.wy0, 0, p.WN - 1) * p.WN + clampi(ix - w.wx0, 0, p.WN - 1);
    return iy * p.G + ix;
}

// Window cell of a position within (or a step beyond) the map limits, float domain as in trav_window.
template <int GEO>
__device__ __forceinline__ int slip_cell_window(const SolveParams &p, const Win w, float x, float y)
{
    v2f q;
    const v2f xy = {x, y}, ir = {p.inv_res, p.inv_res}, nw = {-w.fx0, -w.fy0};
    if (GEO == kGeoPow2Origin0) q = __builtin_elementwise_fma(xy, ir, nw);
    else if (GEO == kGeoPow2) q = __builtin_elementwise_fma(xy - v2f{p.x0, p.y0}, ir, nw);
    else q = v2f{(x - p.x0) / p.res, (y - p.y0) / p.res} + nw;
    const float li = clampf(floorf(q.x), 0.0f, w.fwm1);
    const float lj = clampf(floorf(q.y), 0.0f, w.fwm1);
    return (int)__builtin_fmaf(lj, w.fwn, li);
}

// One observation-mode transit (robot_model.py:59-100) of the sampled-slip chain on the LDS window of (mean, std)
// pairs: state (x, y, th) with heading (sn, cs) and window cell e advances; (xn, yn, tn) is what slot t keeps.
struct SlipChain { float x, y, th, sn, cs; int e; };

template <int GEO, bool FIRST>
__device__ __forceinline__ void slip_chain_step(const SolveParams &p, const float2 *win2, const Win w, SlipChain &c, float u0,
                                                float u1, float z, float &xn, float &yn, float &tn)
{
    const float2 ms = win2[c.e];
    const float trav = trav_from_slip(ms.x, ms.y, z);                  // robot_model.py:75
    const float tv = trav * u0;
    tn = c.th + (trav * u1) * p.dt;
    xn = c.x + (tv * c.cs) * p.dt;
    yn = c.y + (tv * c.sn) * p.dt;
    c.th = FIRST ? wrap_angle(tn) : wrap_angle_near(tn);
    c.x = clampf(xn, p.x0, p.x_hi);
    c.y = clampf(yn, p.y0, p.y_hi);
    sincos_spec(c.th, c.sn, c.cs);
    c.e = slip_cell_window<GEO>(p, w, c.x, c.y);
}

// Wave-wide butterfly reductions (ds_bpermute).  A DPP row-scan formulation was measured 0.25 us faster
// per launch but hipcc's DPP combiner mis-folds the update_dpp + add pairs inside this kernel (wrong sums
// on hardware, correct in an isolated test kernel), so the shuffle form stays.
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

// ------------------------------------------------------------------------------
// Softmin merge and the tail of a solve (shared by the finish kernel, the aux block of the
// pipelined rollout kernel, and the rollout blocks' own prologue merge).
// ------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ float block_reduce(float v, float *red, int tid, bool is_max)
{
    v = is_max ? wave_max(v) : wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < NT / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    __syncthreads();
    return r;
}

// Device-scope accesses that bypass the per-XCD L2 (sc1): what lets workgroups on different XCDs exchange their
// partials inside one launch without a full L2 write-back / invalidate (see ticket_merge).
__device__ __forceinline__ void store_agent(float *ptr, float v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_agent(const float *ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Merge the nblk per-block statistics (max z, sum e, sum e*u) of one instance into
//   U*[j] = sum_k w_k u_k[j]      mppi.py:193-199
// written to us[0..2T) (LDS).  Deterministic: every caller (256 threads) gets bit-identical values,
// which is what lets each rollout block of the next solve recompute the warm-start mean on its own.
// LDS scratch: sc[nblk], red[4].  Returns (max z, sum exp) for the weights.
constexpr int kMergePrefetch = 16;
struct MergeLoads { float v[kMergePrefetch]; float mi, si; int j; };

// Issue every load of the few-blocks merge (nblk <= 64) without consuming any: lets the caller put
// other memory traffic (the window staging) in flight underneath.
template <bool AGENT = false>
__device__ __forceinline__ MergeLoads merge_issue(const float *__restrict__ part, int nblk, int T, int tid)
{
#define BN_PLD(ix) (AGENT ? load_agent(part + (ix)) : part[(ix)])
    const int PS = 2 + 2 * T;
    const int lane = tid & 63;
    MergeLoads L;
    L.j = tid < 2 * T ? tid : 0;
#pragma unroll
    for (int i = 0; i < kMergePrefetch; ++i) L.v[i] = BN_PLD((size_t)min(i, nblk - 1) * PS + 2 + L.j);
    const bool has = lane < nblk;
    L.mi = has ? BN_PLD((size_t)lane * PS) : -INFINITY;
    L.si = has ? BN_PLD((size_t)lane * PS + 1) : 0.0f;
    return L;
#undef BN_PLD
}

// Two-level merge for more than 64 partial rows (K > 4096): rows are first merged in groups of kGroupRows
// consecutive rows, each relative to its group's max -- by whichever wave(s) get the job: a wave of the stand-alone
// tail, or the waves of the last rollout workgroup of the group to finish (ticket) -- then the group rows are merged
// like ordinary partials.  One definition for every caller, so the result does not depend on who ran it.
constexpr int kGroupRows = 16;

// Rows [row0, row0 + nrows) of `part` -> one row `gout` = (group max, sum e, sum e*u[2T]).  One wave; it handles the
// 64-column blocks cb0, cb0 + cbstep, ... (several waves may share a group: they derive identical scales).
template <bool AGENT, bool AGENT_STORE>
__device__ __forceinline__ void merge_group(const float *__restrict__ part, int row0, int nrows, int T, int lane, int cb0,
                                            int cbstep, float *gout)
{
#define BN_PLD(ix) (AGENT ? load_agent(part + (ix)) : part[(ix)])
#define BN_PST(ptr, val) do { if (AGENT_STORE) store_agent((ptr), (val)); else *(ptr) = (val); } while (0)
    const int PS = 2 + 2 * T;
    const bool has = lane < nrows;
    const float mi = has ? BN_PLD((size_t)(row0 + lane) * PS) : -INFINITY;
    const float si = has ? BN_PLD((size_t)(row0 + lane) * PS + 1) : 0.0f;
    float v[kGroupRows];
    int jj = lane + 64 * cb0;
#pragma unroll
    for (int r = 0; r < kGroupRows; ++r) v[r] = (r < nrows && jj < 2 * T) ? BN_PLD((size_t)(row0 + r) * PS + 2 + jj) : 0.0f;
    const float mg = wave_max(mi);
    const float f = has ? expf(mi - mg) : 0.0f;
    const float sg = wave_sum(si * f);
    const int fb = __float_as_int(f);
    for (int cb = cb0;;) {
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < kGroupRows; ++r) acc = __builtin_fmaf(v[r], __int_as_float(__builtin_amdgcn_readlane(fb, r)), acc);   // f == 0 past nrows
        if (jj < 2 * T) BN_PST(gout + 2 + jj, acc);
        cb += cbstep;
        if (64 * cb >= 2 * T) break;
        jj = lane + 64 * cb;
#pragma unroll
        for (int r = 0; r < kGroupRows; ++r) v[r] = (r < nrows && jj < 2 * T) ? BN_PLD((size_t)(row0 + r) * PS + 2 + jj) : 0.0f;
    }
    if (cb0 == 0 && lane == 0) { BN_PST(gout, mg); BN_PST(gout + 1, sg); }
#undef BN_PLD
#undef BN_PST
}

// BIG = false leaves the two-level code out of callers that never see more than 64 rows (the rollout kernels'
// prologue / aux / ticket merges): it costs them registers.
template <int NT, bool AGENT = false, bool BIG = false>
__device__ __forceinline__ void merge_partials(const float *__restrict__ part, int nblk, int T, float *us, float *sc,
                                               float *red, int tid, float &m_out, float &S_out, const MergeLoads *pre)
{
#define BN_PLD(ix) (AGENT ? load_agent(part + (ix)) : part[(ix)])
    const int PS = 2 + 2 * T;
    const int lane = tid & 63;
    float m, S;
    if (nblk <= 64) {
        // Few blocks (K <= 4096): every wave reduces the nblk (max, sum) pairs itself -- same inputs,
        // same operations, so all waves (and all workgroups) hold identical m, S and scales -- and all
        // loads are issued before the first use: one memory round trip, one barrier.
        const MergeLoads L = pre ? *pre : merge_issue<AGENT>(part, nblk, T, tid);
        m = wave_max(L.mi);
        const float f = lane < nblk ? expf(L.mi - m) : 0.0f;       // scale of block `lane`; 0 past nblk
        S = wave_sum(L.si * f);
        // scale of block i = lane i's f, read with v_readlane (ignores EXEC): only the lanes with jj < 2T enter the
        // loop below, and a ds_bpermute shuffle returns nothing from the lanes that did not
        const int fb = __float_as_int(f);
#define BN_SCALE(i) __int_as_float(__builtin_amdgcn_readlane(fb, (i)))
        for (int jj = tid; jj < 2 * T; jj += NT) {
            float acc = 0.0f;
            if (jj == L.j) {
#pragma unroll
                for (int i = 0; i < kMergePrefetch; ++i) acc = __builtin_fmaf(L.v[i], BN_SCALE(i), acc);   // f == 0 past nblk
                for (int i = kMergePrefetch; i < nblk; ++i) acc = __builtin_fmaf(BN_PLD((size_t)i * PS + 2 + jj), BN_SCALE(i), acc);
            } else {
                for (int i = 0; i < nblk; ++i) acc = __builtin_fmaf(BN_PLD((size_t)i * PS + 2 + jj), BN_SCALE(i), acc);
            }
            us[jj] = acc / S;
        }
#undef BN_SCALE
    } else if (BIG && nblk <= 64 * kGroupRows) {
        // two-level (see merge_group): groups dealt to the waves, group rows in LDS, then the few-rows merge above
        constexpr int NW = NT / 64;
        const int ng = (nblk + kGroupRows - 1) / kGroupRows;
        float *grows = red + 32;                         // ng x PS
        for (int g = tid >> 6; g < ng; g += NW)
            if constexpr (BIG) merge_group<AGENT, false>(part, g * kGroupRows, min(kGroupRows, nblk - g * kGroupRows), T, lane, 0, 1, grows + (size_t)g * PS);
        __syncthreads();
        merge_partials<NT, false, false>(grows, ng, T, us, sc, red, tid, m, S, nullptr);
        m_out = m;
        S_out = S;
        return;
    } else {
        // more than 1024 workgroups (K > 65536): plain column sums in row order (independent of NT as well)
        float mm = -INFINITY;
        for (int i = tid; i < nblk; i += NT) mm = fmaxf(mm, BN_PLD((size_t)i * PS));
        m = block_reduce<NT>(mm, red, tid, true);
        float s = 0.0f;
        for (int i = tid; i < nblk; i += NT) {
            const float f = expf(BN_PLD((size_t)i * PS) - m);
            sc[i] = f;
            s += BN_PLD((size_t)i * PS + 1) * f;
        }
        S = block_reduce<NT>(s, red, tid, false);        // the barrier inside also publishes sc[]
        for (int jj = tid; jj < 2 * T; jj += NT) {
            float acc = 0.0f;
            for (int i = 0; i < nblk; ++i) acc = __builtin_fmaf(BN_PLD((size_t)i * PS + 2 + jj), sc[i], acc);
            us[jj] = acc / S;
        }
    }
    __syncthreads();
    m_out = m;
    S_out = S;
#undef BN_PLD
}

// The tail of one solve of instance b: U* (and the next mean), softmin statistics, normalised weights,
// a stable copy of the costs, and the batch-1 rollout X* of U*.  NT threads (320 as the aux workgroup, 1024 stand-alone for large K).
// LDS: [ window | ustar 2T | scale nblk | red 32 | group rows ceil(nblk/16) x (2+2T) if nblk > 64 | sampled mode: draws, (mean, std) window ]
template <int GEO, bool LDSWIN, int NT, bool BIG = false>
__device__ __forceinline__ void finish_body(const SolveParams &p, int b, const float *part_all, const float *cost_all,
                                            const float *state_all, float *smem)
{
    // p.tail_merged: U* and the softmin statistics of this solve were merged already (ticket merge of the sampled
    // kernel, which also wrote the next mean); they come from (ustar_prev, stats_prev).
    const int T = p.T, K = p.K, nblk = p.nblk, PS = 2 + 2 * p.T;
    float *win = smem;
    float *us = win + (LDSWIN ? p.WN * p.WN : 0);
    float *sc = us + 2 * T;
    float *red = sc + nblk;

    const int tid = threadIdx.x;
    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float *part = part_all + (size_t)b * nblk * PS;
    const float sx = state_all[b * 3 + 0], sy = state_all[b * 3 + 1], sth = state_all[b * 3 + 2];
    BN_STAMP(8);

    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN && !p.slip_on) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, tid, NT);
    }
    BN_STAMP(9);

    float m, S;
    if (p.tail_merged) {
        for (int j = tid; j < 2 * T; j += NT) {
            const float u = p.ustar_prev[(size_t)b * 2 * T + j];
            us[j] = u;
            p.ustar[(size_t)b * 2 * T + j] = u;
        }
        m = p.stats_prev[b * 2 + 0];
        S = p.stats_prev[b * 2 + 1];
        __syncthreads();
    } else {
        merge_partials<NT, false, BIG>(part, nblk, T, us, sc, red, tid, m, S, nullptr);
        for (int j = tid; j < 2 * T; j += NT) {
            p.ustar[(size_t)b * 2 * T + j] = us[j];
            p.mean[(size_t)b * 2 * T + j] = us[j];            // _previous_action_seq = U*, no shift (mppi.py:217)
        }
    }
    if (tid == 0) {
        p.stats[b * 2 + 0] = m;
        p.stats[b * 2 + 1] = S;
    }
    if (p.env_on && tid == 64) {
        // the environment step that follows this solve: apply U*[0], log state, reward and goal arrival
        const EnvStep e = env_advance<GEO>(p, b, sx, sy, sth, us[0], us[1], p.env_z, (uint64_t)p.ep_index);
        const size_t B = p.B;
        float *row = p.ep_states + ((size_t)(p.ep_index + 1) * B + b) * 3;
        row[0] = e.x; row[1] = e.y; row[2] = e.th;
        p.env_state[b * 3 + 0] = e.x; p.env_state[b * 3 + 1] = e.y; p.env_state[b * 3 + 2] = e.th;
        p.ep_reward[(size_t)p.ep_index * B + b] = e.reward;
        p.ep_action[((size_t)p.ep_index * B + b) * 2 + 0] = us[0];
        p.ep_action[((size_t)p.ep_index * B + b) * 2 + 1] = us[1];
        if (p.ep_index == 0) {
            float *row0 = p.ep_states + (size_t)b * 3;
            row0[0] = sx; row0[1] = sy; row0[2] = sth;
        }
        if (e.reached && !e.frozen && p.ep_done[b] < 0) p.ep_done[b] = p.ep_index;
    }
    BN_STAMP(10);

    if (p.slip_on) {
        // sampled-slip mode: the optimal rollout draws a fresh slip per transit as well (mppi.py:202-214 with
        // traversability_model.py:65-69).  Draws and the (mean, std) window are staged by all threads first.
        const float *__restrict__ sg = p.slip_std + (size_t)b * p.map_stride;
        float *zol = red + 32 + ((nblk > 64 && nblk <= 64 * kGroupRows) ? ((nblk + kGroupRows - 1) / kGroupRows) * PS : 0);                      // T + 4 draws
        float2 *win2 = reinterpret_cast<float2 *>((reinterpret_cast<uintptr_t>(zol + ((T + 7) & ~3)) + 7) & ~(uintptr_t)7);
        if (p.zo) {
            for (int t = tid; t < T; t += NT) zol[t] = p.zo[(size_t)b * T + t];
        } else {
            for (int j = tid; 4 * j < T; j += NT) philox_slip_block(p.seed, p.tail_solve, (uint32_t)b, 0xffffffffu, (uint32_t)j, zol + 4 * j);
        }
        if (LDSWIN) {
            w = window_origin<GEO>(p, sx, sy);
            for (int e = tid; e < p.WN * p.WN; e += NT) {
                const int r = e / p.WN, c = e - r * p.WN;
                const size_t g = (size_t)(w.wy0 + r) * p.G + (w.wx0 + c);
                win2[e] = make_float2(map[g], sg[g]);
            }
        }
        __syncthreads();
        BN_STAMP(12);
        if (tid == 0) {
            float *Xs = p.xstar + (size_t)b * (T + 1) * 3;
            float xn, yn, tn;
            if (LDSWIN) {
                SlipChain c;
                c.x = sx; c.y = sy; c.th = sth;
                sincos_spec(c.th, c.sn, c.cs);
                c.e = slip_cell_safe<GEO, true>(p, w, sx, sy);
                slip_chain_step<GEO, true>(p, win2, w, c, us[0], us[1], zol[0], xn, yn, tn);
                Xs[0] = xn; Xs[1] = yn; Xs[2] = tn;
                int t = 1;
                for (; t + 4 <= T; t += 4) {                 // controls and draws of four steps read up front
                    float uq[4][3];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { uq[i][0] = us[2 * (t + i)]; uq[i][1] = us[2 * (t + i) + 1]; uq[i][2] = zol[t + i]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        slip_chain_step<GEO, false>(p, win2, w, c, uq[i][0], uq[i][1], uq[i][2], xn, yn, tn);
                        Xs[3 * (t + i)] = xn; Xs[3 * (t + i) + 1] = yn; Xs[3 * (t + i) + 2] = tn;
                    }
                }
                for (; t < T; ++t) {
                    slip_chain_step<GEO, false>(p, win2, w, c, us[2 * t], us[2 * t + 1], zol[t], xn, yn, tn);
                    Xs[3 * t] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
                }
                Xs[3 * T] = c.x; Xs[3 * T + 1] = c.y; Xs[3 * T + 2] = c.th;
                BN_STAMP(11);
            } else {
                float x = sx, y = sy, th = sth;
                for (int t = 0; t < T; ++t) {
                    const int e = slip_cell_safe<GEO, false>(p, w, x, y);
                    const float trav = trav_from_slip(map[e], sg[e], zol[t]);
                    float sn, cs;
                    sincos_spec(th, sn, cs);
                    xn = x + ((trav * us[2 * t]) * cs) * p.dt; yn = y + ((trav * us[2 * t]) * sn) * p.dt;
                    tn = th + (trav * us[2 * t + 1]) * p.dt;
                    Xs[3 * t] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
                    x = clampf(xn, p.x0, p.x_hi); y = clampf(yn, p.y0, p.y_hi); th = wrap_angle(tn);
                }
                Xs[3 * T] = x; Xs[3 * T + 1] = y; Xs[3 * T + 2] = th;
            }
        }
    }
    if (p.slip_on && tid < 64) {
        // wave 0: thread 0 ran the sampled chain above
    } else if (tid == 0) {
        // optimal_state_seq: batch-1 rollout of U* with the same aliasing (mppi.py:202-214)
        Chain c;
        c.x = sx; c.y = sy; c.th = sth;
        sincos_spec(c.th, c.sn, c.cs);
        c.trav = trav_lookup<GEO, LDSWIN, true>(p, win, map, w, c.x, c.y);
        float *Xs = p.xstar + (size_t)b * (T + 1) * 3;
        float xn, yn, tn;
        chain_step<GEO, LDSWIN, true>(p, win, map, w, c, us[0], us[1], xn, yn, tn);
        Xs[0] = xn; Xs[1] = yn; Xs[2] = tn;
        int t = 1;
        for (; t + 4 <= T; t += 4) {                       // controls of four steps read up front (LDS latency off the chain)
            float uq[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { uq[i][0] = us[2 * (t + i)]; uq[i][1] = us[2 * (t + i) + 1]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                chain_step<GEO, LDSWIN, false>(p, win, map, w, c, uq[i][0], uq[i][1], xn, yn, tn);
                Xs[3 * (t + i) + 0] = xn; Xs[3 * (t + i) + 1] = yn; Xs[3 * (t + i) + 2] = tn;
            }
        }
        for (; t < T; ++t) {
            chain_step<GEO, LDSWIN, false>(p, win, map, w, c, us[2 * t], us[2 * t + 1], xn, yn, tn);
            Xs[3 * t + 0] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
        }
        Xs[3 * T + 0] = c.x; Xs[3 * T + 1] = c.y; Xs[3 * T + 2] = c.th;
        BN_STAMP(11);
    } else if (NT > 64 && tid >= 64) {
        // _weights = softmax(-costs / lambda)   mppi.py:193
        const float *cost = cost_all + (size_t)b * K;
        float *wout = p.w + (size_t)b * K;
        float *cout = p.cost_out + (size_t)b * K;
        for (int k = tid - 64; k < K; k += NT - 64) {
            const float ck = cost[k];
            cout[k] = ck;
            wout[k] = expf((-ck) / p.lambda_ - m) / S;
        }
    }
    if constexpr (NT == 64) {                          // single-wave tail: the weights follow the X* rollout
        const float *cost = cost_all + (size_t)b * K;
        float *wout = p.w + (size_t)b * K;
        float *cout = p.cost_out + (size_t)b * K;
        for (int k = tid; k < K; k += 64) {
            const float ck = cost[k];
            cout[k] = ck;
            wout[k] = expf((-ck) / p.lambda_ - m) / S;
        }
    }
}

// Ticket merge: every workgroup of instance b publishes its partials, takes a ticket, and the one that draws the
// last ticket merges all of them (fixed order: the result does not depend on which workgroup that is) into
// U* = the next mean, plus the softmin statistics the tail needs for the weights.  Saves the merge launch.
// With more than 64 workgroups the merge is the two-level one (merge_group): the last workgroup of each group of 16
// merges its group (its waves split the columns: one memory round trip), the last group to finish merges the groups.
// The partials travel as device-scope sc1 stores / loads (store_agent / load_agent): once every wave has seen its
// stores acknowledged (vmcnt 0) and the workgroup has met at the barrier, the ticket is taken.  No __threadfence:
// that writes back / invalidates the whole L2 once per workgroup (measured: +18 us per launch at 128 workgroups).
// Ticket counters: kTicketStride ints per instance, [0] = groups (or workgroups) done, [1 + g] = workgroups of group g.
// LDS scratch: [ us 2T | sc nblk | red 32 | flag ].  Needs nblk <= 1024 (two levels).
constexpr int kTicketStride = 1 + 64;

template <int NT>
__device__ __forceinline__ void ticket_merge(const SolveParams &p, int b, float *scratch)
{
    const int T = p.T, tid = threadIdx.x, PS = 2 + 2 * p.T;
    float *us = scratch, *sc = us + 2 * T, *red = sc + p.nblk;
    int *flag = reinterpret_cast<int *>(red + 32);       // at most 64 rows reach merge_partials here: no group rows in LDS
    int *ticket = p.ticket + (size_t)b * kTicketStride;
    const float *part = p.part + (size_t)b * p.nblk * PS;
    const float *rows = part;
    int nrows = p.nblk;
    if (p.nblk > 64) {
        const int g = blockIdx.x / kGroupRows, ng = (p.nblk + kGroupRows - 1) / kGroupRows;
        const int in_group = min(kGroupRows, p.nblk - g * kGroupRows);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *flag = (atomicAdd(ticket + 1 + g, 1) == in_group - 1) ? 1 : 0;
        __syncthreads();
        if (!*flag) return;
        if (tid == 0) ticket[1 + g] = 0;
        float *grow = p.gpart + ((size_t)b * 64 + g) * PS;
        merge_group<true, true>(part, g * kGroupRows, in_group, T, tid & 63, tid >> 6, NT / 64, grow);
        rows = p.gpart + (size_t)b * 64 * PS;
        nrows = ng;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *flag = (atomicAdd(ticket, 1) == nrows - 1) ? 1 : 0;
    __syncthreads();
    if (!*flag) return;
    BN_STAMP_ANY(6);
    if (tid == 0) ticket[0] = 0;                       // ready for the next launch (ordered by the stream)
    float m, S;
    merge_partials<NT, true, false>(rows, nrows, T, us, sc, red, tid, m, S, nullptr);
    for (int j = tid; j < 2 * T; j += NT) {
        p.ustar_cur[(size_t)b * 2 * T + j] = us[j];
        p.mean[(size_t)b * 2 * T + j] = us[j];         // _previous_action_seq = U*, no shift (mppi.py:217)
    }
    if (tid == 0) { p.stats_cur[b * 2 + 0] = m; p.stats_cur[b * 2 + 1] = S; }
    BN_STAMP_ANY(7);
}

// Workgroup barrier that only drains LDS traffic: global stores of the consumer wave stay in flight.
#define BN_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Producer: the clamped perturbed controls of steps t and t+1 (t even) of this lane's rollout,
//   u = clamp(mean + sigma * eps, u_min, u_max)          mppi.py:152-157
// written to the LDS control tile (and to HBM when _perturbed_action_seqs is materialised).
template <int EPS, bool STORE_U>
__device__ __forceinline__ void produce_pair(const SolveParams &p, const float *__restrict__ eps, int b, int kk, int t,
                                             uint64_t solve, const float *ml, float *Ul, float *Ub, size_t Kp, int lane)
{
    float e[4];
    const int t1 = min(t + 1, p.T - 1);
    if (EPS == kEpsPhilox) {
        philox_eps_pair(p.seed, solve, (uint32_t)b, (uint32_t)(kk + p.k0), (uint32_t)(t >> 1), e);
    } else if (EPS == kEpsKT2) {
        const float *row = eps + ((size_t)b * p.K + kk) * p.T * 2;
        const float2 v0 = *reinterpret_cast<const float2 *>(row + 2 * t);
        const float2 v1 = *reinterpret_cast<const float2 *>(row + 2 * t1);
        e[0] = v0.x; e[1] = v0.y; e[2] = v1.x; e[3] = v1.y;
    } else {
        const float *r0 = eps + ((size_t)b * p.T + t) * 2 * p.K;
        const float *r1 = eps + ((size_t)b * p.T + t1) * 2 * p.K;
        e[0] = r0[kk]; e[1] = r0[p.K + kk]; e[2] = r1[kk]; e[3] = r1[p.K + kk];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int tt = t + s;
        if (tt < p.T) {
            const float u0 = clampf(ml[2 * tt] + p.sigma0 * e[2 * s], p.umin0, p.umax0);
            const float u1 = clampf(ml[2 * tt + 1] + p.sigma1 * e[2 * s + 1], p.umin1, p.umax1);
            Ul[(2 * tt) * kUPad + lane] = u0;
            Ul[(2 * tt + 1) * kUPad + lane] = u1;
            if (STORE_U) { float *Ut = Ub + (size_t)(2 * tt) * Kp; Ut[0] = u0; Ut[Kp] = u1; }
        }
    }
}

// ------------------------------------------------------------------------------
// Rollout + cost kernel.  grid = (ceil(K/64) [+1 aux], B), block = 320 = 5 wavefronts that all map
// lane -> rollout k = 64*blockIdx.x + lane and split the work of those 64 rollouts by ROLE,
// because one wavefront issues one instruction every ~5 cycles whether or not it depends on the
// previous one (measured, tools/ubench2.hip: no intra-wave overlap) and the T-step recurrence is a
// serial instruction chain:
//   wave 0  chain      the recurrence only: transit + gather (chain_step), ~50 instructions/step
//   wave 1  producer   noise -> clamped controls for the first half of chunk c+2
//   wave 2  producer   ... second half of chunk c+2
//   wave 3  consumer A chunk c-1: trajectory stores, control cost (fp64 accumulation in step order)
//   wave 4  consumer B chunk c-1: stage cost (sqrt, collision flag; fp64 accumulation in step order)
// Chunks are TU = 4 steps; one LDS-only barrier per chunk hands the control tile forward and the
// chain's outputs (ring of 2 chunks) backward.  Each per-rollout sum is accumulated by one wave in
// step order, so the arithmetic is identical to a single sequential loop (Arithmetic spec).
// LDS: [ ring 2 x TU x 64 x float4 | final state + control-cost sum 5 x 64 | e 64 | window WN*WN | mean 2T | mean*inv_var 2T |
//        control tile 2T x 65 ]
// ------------------------------------------------------------------------------
template <int EPS, int GEO, bool LDSWIN, bool STORE_U, bool TICKET>
__global__ __launch_bounds__(kRolloutThreads) void rollout_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BN_TRACE_BEGIN();
    if (blockIdx.x == p.nblk) {
        // pipelined mode: the extra workgroup computes the tail of the PREVIOUS solve (U*, X*, weights)
        // while the other workgroups roll out this one
        finish_body<GEO, LDSWIN, kRolloutThreads>(p, blockIdx.y, p.part_prev, p.cost_prev, p.state_prev, smem);
        BN_TRACE_END();
        return;
    }
    const int T = p.T, K = p.K;
    float *ring = smem;                               // first: 16-byte aligned for the b128 ring accesses
    float *fin = ring + 2 * TU * 4 * 64;
    float *el = fin + 5 * 64;
    float *win = el + 64;
    float *ml = win + (LDSWIN ? p.WN * p.WN : 0);
    float *mv = ml + 2 * T;
    float *Ul = mv + 2 * T;

    const int tid = threadIdx.x;
    const int wid = tid >> 6, lane = tid & 63;
    // A workgroup's waves are dealt to the 4 SIMDs cyclically, so waves 0 and 4 share one: they get the two
    // light consumer roles; the chain (role 0) has a SIMD to itself.  role: 0 chain, 1-2 producers, 3-4 consumers.
    const int wv = (wid == 0) ? 3 : (wid == 4) ? 4 : wid - 1;
    const int b = blockIdx.y;
    const int k = blockIdx.x * kRolloutsPerBlock + lane;
    const bool active = k < K;
    const int kk = active ? k : K - 1;

    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float *__restrict__ eps = p.eps;
    // open loop: the caller's state; closed loop: the previous solve's state, advanced below
    const float *st_src = p.closed_loop ? p.state_prev : p.state;
    float sx = st_src[b * 3 + 0], sy = st_src[b * 3 + 1], sth = st_src[b * 3 + 2];
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];
    const uint64_t solve = p.solve;
    BN_STAMP(0);

    // warm start straight from the previous solve's per-block statistics (bit-identical to the U* the aux
    // block publishes): no kernel boundary between consecutive solves of one instance.  Its loads go out
    // first so that they and the window staging (which waits for the state) share one memory round trip.
    const float *part_prev = p.part_prev + (size_t)b * p.nblk * (2 + 2 * T);
    MergeLoads pre;
    const bool pre_ok = p.mean_from_part && p.nblk <= 64;
    if (pre_ok) pre = merge_issue(part_prev, p.nblk, T, tid);

    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN && !p.closed_loop) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, tid, kRolloutThreads);
    }
    if (p.mean_from_part) {
        float m_unused, S_unused;
        merge_partials<kRolloutThreads>(part_prev, p.nblk, T, ml, ring, ring + p.nblk, tid, m_unused, S_unused, pre_ok ? &pre : nullptr);
        for (int j = tid; j < 2 * T; j += kRolloutThreads) mv[j] = ml[j] * ((j & 1) ? p.iv1 : p.iv0);
    } else {
        for (int j = tid; j < 2 * T; j += kRolloutThreads) {
            const float m = p.mean[(size_t)b * 2 * T + j];
            ml[j] = m;
            mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);      // mean[t] @ inv_cov (diagonal), mppi.py:179-180
        }
    }
    if (p.closed_loop) {
        // PlanetaryEnv.step with the previous solve's first control (ml[0..1] = U*_prev[0], published by the
        // barrier inside merge_partials): every workgroup advances the state itself, no launch in between
        const EnvStep e = env_advance<GEO>(p, b, sx, sy, sth, ml[0], ml[1], p.env_z, (uint64_t)p.ep_index);
        sx = e.x; sy = e.y; sth = e.th;
        if (LDSWIN) {
            w = window_origin<GEO>(p, sx, sy);
            stage_window(win, map, w, p.WN, p.G, tid, kRolloutThreads);
        }
    }
    if (blockIdx.x == 0 && tid == 0) {                 // the state this solve starts from, for its tail and the next solve
        p.state_copy[b * 3 + 0] = sx; p.state_copy[b * 3 + 1] = sy; p.state_copy[b * 3 + 2] = sth;
    }
    BN_BAR();

    // Rows of X and U are pitched to Kp = 64 * nblk floats, so every lane stores unconditionally
    // (lanes past K write into the pad).
    const size_t Kp = (size_t)p.Kp;
    float *Xb = p.X + (size_t)b * (T + 1) * 3 * Kp + k;
    float *Ub = STORE_U ? p.U + (size_t)b * T * 2 * Kp + k : nullptr;

    // controls of chunks 0 and 1 (steps 0 .. 2*TU-1): TU/2 steps per wave (waves 0-3)
    if (wid < 4) {
#pragma unroll
        for (int q = 0; q < TU / 4; ++q) {
            const int t = wid * (TU / 2) + 2 * q;
            if (t < T) produce_pair<EPS, STORE_U>(p, eps, b, kk, t, solve, ml, Ul, Ub, Kp, lane);
        }
    }
    BN_BAR();
    BN_STAMP(1);

    Chain c;                                          // wave 0
    double Sd = 0.0, Ad = 0.0;                        // waves 4 / 3: fp64 accumulation of the fp32 terms (Arithmetic spec)
    if (wv == 0) {
        c.x = sx; c.y = sy; c.th = sth;               // mppi.py:160
        sincos_spec(c.th, c.sn, c.cs);
        c.trav = trav_lookup<GEO, LDSWIN, true>(p, win, map, w, c.x, c.y);
    }

    // wave 0: one chain step; emits what the reference leaves in slot t plus the traversability of state t+1.
    // The controls of the whole chunk are read from the LDS tile up front (uc[]): the compiler cannot hoist
    // those reads over the ring writes itself, and a 64-cycle LDS round trip per step would sit on the chain.
    // The ring writes come last (sched_barrier) so the wait for the gather overlaps the rest of the step.
#define BN_CHAIN(FIRST, i, slot)                                                                               \
    do {                                                                                                       \
        float xn, yn, tn;                                                                                      \
        chain_step<GEO, LDSWIN, FIRST>(p, win, map, w, c, uc[i][0], uc[i][1], xn, yn, tn);                     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        reinterpret_cast<float4 *>(slot)[lane] = make_float4(xn, yn, tn, c.trav);   /* one ds_write_b128 */     \
    } while (0)

    // wave 3: trajectory stores and control cost of one step
#define BN_CONSUME_A(t, o)                                                                                     \
    do {                                                                                                       \
        if (!(BN_ABLATE & 4)) {                                                                                \
        float *Xt = Xb + (size_t)(3 * (t)) * Kp;         /* slot t keeps the un-clamped state (aliasing) */     \
        Xt[0] = o.x; Xt[Kp] = o.y; Xt[2 * Kp] = o.z;                                                           \
        } else { BN_KEEP(o.z); }                                                                               \
        if (!(BN_ABLATE & 8)) {                                                                                \
        const float u0 = Ul[(2 * (t)) * kUPad + lane], u1 = Ul[(2 * (t) + 1) * kUPad + lane];                  \
        const float a = mv[2 * (t)] * u0 + mv[2 * (t) + 1] * u1;            /* mppi.py:178-182 */             \
        Ad += (double)(p.lambda_ * a);                                                                         \
        }                                                                                                      \
    } while (0)

    // wave 4: stage cost of one step.  The cell of the un-clamped slot equals the cell of the clamped
    // state (index clamp, grid_map.py:209), so the chain's gather serves stage cost t and transit t+1.
#define BN_CONSUME_B(t, o)                                                                                     \
    do {                                                                                                       \
        if (BN_ABLATE & 1) { BN_KEEP(o.x); BN_KEEP(o.y); BN_KEEP(o.w); } else {                                \
        const float dx = o.x - gx, dy = o.y - gy;                                                              \
        const float sc = sqrt_cr(dx * dx + dy * dy) + (o.w <= p.thr ? 1.0e4f : 0.0f);   /* objectives.py:47-53 */ \
        Sd += (double)sc;                                                                                      \
        }                                                                                                      \
    } while (0)

    // One phase = one chunk of TU steps for the chain, the previous chunk for the consumer, the
    // chunk after next for the producers.  GUARD handles the ragged last chunk.
#define BN_PHASE(cc, FIRSTCHUNK, GUARD)                                                                        \
    do {                                                                                                       \
        if (wv == 0) {                                                                                         \
            if ((cc) * TU < T) {                                                                               \
                float *slot = ring + (((cc) & 1) * TU) * 256;                                                  \
                /* the NEXT chunk's controls (produced last phase, visible since the barrier) are read now and */ \
                /* land under this chunk's steps: no LDS round trip at the head of a phase                    */ \
                float un[TU][2];                                                                               \
                _Pragma("unroll") for (int i = 0; i < TU; ++i) {                                               \
                    const int t = min(((cc) + 1) * TU + i, T - 1);                                             \
                    un[i][0] = Ul[(2 * t) * kUPad + lane];                                                     \
                    un[i][1] = Ul[(2 * t + 1) * kUPad + lane];                                                 \
                }                                                                                              \
                __builtin_amdgcn_sched_barrier(0);                                                             \
                _Pragma("unroll") for (int i = 0; i < TU; ++i) {                                               \
                    const int t = (cc) * TU + i;                                                               \
                    if (!(GUARD) || t < T) {                                                                   \
                        if ((FIRSTCHUNK) && i == 0) BN_CHAIN(true, i, slot + i * 256);                         \
                        else BN_CHAIN(false, i, slot + i * 256);                                               \
                    }                                                                                          \
                }                                                                                              \
                _Pragma("unroll") for (int i = 0; i < TU; ++i) { uc[i][0] = un[i][0]; uc[i][1] = un[i][1]; }   \
            }                                                                                                  \
        } else if (wv >= 3) {                                                                                  \
            if ((cc) >= 1) {                                                                                   \
                const float *slot = ring + ((((cc) - 1) & 1) * TU) * 256;                                      \
                float4 rq[TU];                        /* the whole chunk in one burst of ds_read_b128 */      \
                _Pragma("unroll") for (int i = 0; i < TU; ++i)                                                 \
                    rq[i] = reinterpret_cast<const float4 *>(slot + i * 256)[lane];                            \
                _Pragma("unroll") for (int i = 0; i < TU; ++i) {                                               \
                    const int t = ((cc) - 1) * TU + i;                                                         \
                    if (!(GUARD) || t < T) {                                                                   \
                        if (wv == 3) BN_CONSUME_A(t, rq[i]);                                                   \
                        else BN_CONSUME_B(t, rq[i]);                                                           \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
        } else {                                                                                               \
            _Pragma("unroll") for (int q = 0; q < TU / 4; ++q) {                                               \
                const int t = ((cc) + 2) * TU + (wv == 1 ? 0 : TU / 2) + 2 * q;                                \
                if (t < T) produce_pair<EPS, STORE_U>(p, eps, b, kk, t, solve, ml, Ul, Ub, Kp, lane);          \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

    const int nfull = T / TU;                         // chunks with all TU steps
    float uc[TU][2];                                  // chain: the controls of its current chunk
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            const int t = min(i, T - 1);
            uc[i][0] = Ul[(2 * t) * kUPad + lane];
            uc[i][1] = Ul[(2 * t + 1) * kUPad + lane];
        }
    }
    BN_PHASE(0, true, true);
    BN_BAR();
    BN_STAMP(2);
    int cc = 1;
#ifdef BN_TIMING
    unsigned long long busy = 0;
#endif
    for (; cc < nfull; ++cc) {                        // steady state: chunks cc (chain) and cc-1 (consumer) are full
#ifdef BN_TIMING
        const unsigned long long t_a = __builtin_readcyclecounter();
#endif
        BN_PHASE(cc, false, false);
#ifdef BN_TIMING
        busy += __builtin_readcyclecounter() - t_a;
#endif
        BN_BAR();
    }
#ifdef BN_TIMING
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) p.stamps[16 + wv] = busy;
#endif
    for (; cc * TU < T + TU; ++cc) {                  // ragged tail and the consumer's drain
        BN_PHASE(cc, false, true);
        BN_BAR();
    }
#undef BN_PHASE
#undef BN_CONSUME_A
#undef BN_CONSUME_B
#undef BN_CHAIN
    BN_STAMP(3);

    if (wv == 0) {                                     // state T (clamped / wrapped) and its traversability
        fin[lane] = c.x; fin[64 + lane] = c.y; fin[128 + lane] = c.th; fin[192 + lane] = c.trav;
    } else if (wv == 3) {
        fin[256 + lane] = (float)Ad;                   // sum_t lambda * control cost, rounded once
    }
    BN_BAR();

    if (wv == 3) {
        float *Xt = Xb + (size_t)(3 * T) * Kp;         // slot T: clamped / wrapped state
        Xt[0] = fin[lane]; Xt[Kp] = fin[64 + lane]; Xt[2 * Kp] = fin[128 + lane];
    } else if (wv == 4) {
        const float xT = fin[lane], yT = fin[64 + lane], trT = fin[192 + lane], A = fin[256 + lane];
        const float dxT = xT - gx, dyT = yT - gy;
        const float term = sqrt_cr(dxT * dxT + dyT * dyT) + (trT <= p.thr ? 1.0e4f : 0.0f);    // mppi.py:184
        const float cost = ((float)Sd + term) + A;                                             // mppi.py:186-190
        if (active) p.cost[(size_t)b * K + k] = cost;
        // block-local softmin statistics   mppi.py:193-199
        const float z = active ? (-cost) / p.lambda_ : -INFINITY;
        const float zmax = wave_max(z);
        const float e = active ? expf(z - zmax) : 0.0f;
        const float esum = wave_sum(e);
        el[lane] = e;
        if (lane == 0) {
            float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
            if (TICKET) { store_agent(part, zmax); store_agent(part + 1, esum); }
            else { part[0] = zmax; part[1] = esum; }
        }
    }
    BN_BAR();
    BN_STAMP(4);
    {   // weighted control sums of this block: column j of the tile, 5 waves x 64 lanes over 2T columns
        float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
        for (int j = tid; j < 2 * T; j += kRolloutThreads) {
            const float *col = Ul + j * kUPad;
            float acc = 0.0f;
#pragma unroll 16
            for (int q = 0; q < kRolloutsPerBlock; ++q) acc = __builtin_fmaf(el[q], col[q], acc);
            if (TICKET) store_agent(part + 2 + j, acc); else part[2 + j] = acc;
        }
    }
    BN_STAMP(5);
    // sizes the pipelined prologue merge does not take (K > 2048): the last workgroup merges, the tail rides in the next launch
    if (TICKET) ticket_merge<kRolloutThreads>(p, b, smem);
    BN_TRACE_END();
}

// ------------------------------------------------------------------------------
// Throughput variant of the rollout kernel: ONE wavefront per 64 rollouts does everything in step order -- per pair
// of steps the noise and controls (in registers), then per step transit + gather, trajectory stores, stage and
// control cost.  No ring, no barriers, no role split, no control tile in LDS: the controls go to HBM (the (T,2,Kp)
// buffer of BN_FLAG_STORE_CONTROLS) and come back, L2-hot, for the weighted control sums -- lane = column there, one
// 256-byte row of the 64 rollouts per column.  3.5 KB of LDS and one wave per workgroup, so a SIMD holds as many
// workgroups as its registers allow (6) and they fill each other's issue gaps and memory waits.  A lone workgroup is
// 2x slower than the role kernel's (the recurrence waits for everything else); with every SIMD full the kernel is
// VALU-bound at ~7000 VALU instructions per workgroup against the role kernel's ~8700 + its skeleton (rocprofv3 SQ
// counters, tools/pmc_sq.sh), and wins by 10-15 % from about 1500 workgroups per launch (96 instances of K=1024) on.
// Same device functions in the same order per rollout: results are bit-identical to the role kernel.
// grid = (ceil(K/64) [+1 aux], B), block = 64.  LDS: [ window | mean 2T | mean*inv_var 2T | e 64 | merge scratch ].
// ------------------------------------------------------------------------------
template <int EPS>
__device__ __forceinline__ void noise_pair(const SolveParams &p, int b, int kk, int t, float e[4])
{
    const int t1 = min(t + 1, p.T - 1);
    if (EPS == kEpsPhilox) {
        philox_eps_pair(p.seed, p.solve, (uint32_t)b, (uint32_t)(kk + p.k0), (uint32_t)(t >> 1), e);
    } else if (EPS == kEpsKT2) {
        const float *row = p.eps + ((size_t)b * p.K + kk) * p.T * 2;
        const float2 v0 = *reinterpret_cast<const float2 *>(row + 2 * t);
        const float2 v1 = *reinterpret_cast<const float2 *>(row + 2 * t1);
        e[0] = v0.x; e[1] = v0.y; e[2] = v1.x; e[3] = v1.y;
    } else {
        const float *r0 = p.eps + ((size_t)b * p.T + t) * 2 * p.K;
        const float *r1 = p.eps + ((size_t)b * p.T + t1) * 2 * p.K;
        e[0] = r0[kk]; e[1] = r0[p.K + kk]; e[2] = r1[kk]; e[3] = r1[p.K + kk];
    }
}

template <int EPS, int GEO, bool LDSWIN>
__global__ __launch_bounds__(64) void rollout_wave_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (blockIdx.x == p.nblk) {
        finish_body<GEO, LDSWIN, 64>(p, blockIdx.y, p.part_prev, p.cost_prev, p.state_prev, smem);
        return;
    }
    const int T = p.T, K = p.K;
    float *win = smem;
    float *ml = win + (LDSWIN ? p.WN * p.WN : 0);
    float *mv = ml + 2 * T;
    float *el = mv + 2 * T;
    float *sc = el + 64;                              // merge scratch: nblk scales + 32
    const int lane = threadIdx.x, b = blockIdx.y;
    const int k = blockIdx.x * kRolloutsPerBlock + lane;
    const bool active = k < K;
    const int kk = active ? k : K - 1;
    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];

    const float *part_prev = p.part_prev + (size_t)b * p.nblk * (2 + 2 * T);
    MergeLoads pre;
    const bool pre_ok = p.mean_from_part && p.nblk <= 64;
    if (pre_ok) pre = merge_issue(part_prev, p.nblk, T, lane);
    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, lane, 64);
    }
    if (p.mean_from_part) {
        float m_unused, S_unused;
        merge_partials<64>(part_prev, p.nblk, T, ml, sc, sc + p.nblk, lane, m_unused, S_unused, pre_ok ? &pre : nullptr);
        for (int j = lane; j < 2 * T; j += 64) mv[j] = ml[j] * ((j & 1) ? p.iv1 : p.iv0);
    } else {
        for (int j = lane; j < 2 * T; j += 64) {
            const float m = p.mean[(size_t)b * 2 * T + j];
            ml[j] = m;
            mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);      // mean[t] @ inv_cov (diagonal), mppi.py:179-180
        }
    }
    if (blockIdx.x == 0 && lane == 0) {
        p.state_copy[b * 3 + 0] = sx; p.state_copy[b * 3 + 1] = sy; p.state_copy[b * 3 + 2] = sth;
    }
    __syncthreads();
    const size_t Kp = (size_t)p.Kp;
    float *Xb = p.X + (size_t)b * (T + 1) * 3 * Kp + k;
    float *Ub = p.U + (size_t)b * T * 2 * Kp + k;

    Chain c;
    c.x = sx; c.y = sy; c.th = sth;                   // mppi.py:160
    sincos_spec(c.th, c.sn, c.cs);
    c.trav = trav_lookup<GEO, LDSWIN, true>(p, win, map, w, c.x, c.y);
    double Sd = 0.0, Ad = 0.0;
    // one step: transit (slot t keeps the un-clamped state), its stores, stage cost on the slot with the traversability
    // of the clamped successor (same cell, grid_map.py:209), control cost; fp64 accumulation in step order
#define BN_WAVE_STEP(FIRST, t, u0, u1)                                                                            \
    do {                                                                                                          \
        float xn, yn, tn;                                                                                         \
        chain_step<GEO, LDSWIN, FIRST>(p, win, map, w, c, (u0), (u1), xn, yn, tn);                                \
        float *Xt = Xb + (size_t)(3 * (t)) * Kp;                                                                  \
        Xt[0] = xn; Xt[Kp] = yn; Xt[2 * Kp] = tn;                                                                 \
        const float dx = xn - gx, dy = yn - gy;                                                                   \
        Sd += (double)(sqrt_cr(dx * dx + dy * dy) + (c.trav <= p.thr ? 1.0e4f : 0.0f));   /* objectives.py:47-53 */  \
        Ad += (double)(p.lambda_ * (mv[2 * (t)] * (u0) + mv[2 * (t) + 1] * (u1)));      /* mppi.py:178-182 */      \
    } while (0)
    for (int t = 0; t < T; t += 2) {
        float e[4];
        noise_pair<EPS>(p, b, kk, t, e);
        const float u0 = clampf(ml[2 * t] + p.sigma0 * e[0], p.umin0, p.umax0);          // mppi.py:152-157
        const float u1 = clampf(ml[2 * t + 1] + p.sigma1 * e[1], p.umin1, p.umax1);
        float *Ut = Ub + (size_t)(2 * t) * Kp;
        Ut[0] = u0; Ut[Kp] = u1;
        if (t == 0) BN_WAVE_STEP(true, t, u0, u1); else BN_WAVE_STEP(false, t, u0, u1);
        if (t + 1 < T) {
            const float v0 = clampf(ml[2 * t + 2] + p.sigma0 * e[2], p.umin0, p.umax0);
            const float v1 = clampf(ml[2 * t + 3] + p.sigma1 * e[3], p.umin1, p.umax1);
            Ut[2 * Kp] = v0; Ut[3 * Kp] = v1;
            BN_WAVE_STEP(false, t + 1, v0, v1);
        }
    }
#undef BN_WAVE_STEP
    {
        float *Xt = Xb + (size_t)(3 * T) * Kp;         // slot T: clamped / wrapped state
        Xt[0] = c.x; Xt[Kp] = c.y; Xt[2 * Kp] = c.th;
    }
    const float dxT = c.x - gx, dyT = c.y - gy;
    const float term = sqrt_cr(dxT * dxT + dyT * dyT) + (c.trav <= p.thr ? 1.0e4f : 0.0f);      // mppi.py:184
    const float cost = ((float)Sd + term) + (float)Ad;                                          // mppi.py:186-190
    if (active) p.cost[(size_t)b * K + k] = cost;
    const float z = active ? (-cost) / p.lambda_ : -INFINITY;
    const float zmax = wave_max(z);
    const float e = active ? expf(z - zmax) : 0.0f;
    const float esum = wave_sum(e);
    el[lane] = e;
    __syncthreads();                                   // e in LDS; this wave's control stores visible to all its lanes
    float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
    if (lane == 0) { part[0] = zmax; part[1] = esum; }
    // weighted control sums: lane = column j, whose 64 rollout values are one contiguous row of the (T,2,Kp) buffer
    const float *Urow0 = p.U + (size_t)b * T * 2 * Kp + (size_t)blockIdx.x * kRolloutsPerBlock;
    for (int j = lane; j < 2 * T; j += 64) {
        const float4 *row = reinterpret_cast<const float4 *>(Urow0 + (size_t)j * Kp);
        float acc = 0.0f;
#pragma unroll
        for (int q4 = 0; q4 < kRolloutsPerBlock / 4; ++q4) {
            const float4 v = row[q4];
            acc = __builtin_fmaf(el[4 * q4 + 0], v.x, acc);
            acc = __builtin_fmaf(el[4 * q4 + 1], v.y, acc);
            acc = __builtin_fmaf(el[4 * q4 + 2], v.z, acc);
            acc = __builtin_fmaf(el[4 * q4 + 3], v.w, acc);
        }
        part[2 + j] = acc;
    }
}

// ------------------------------------------------------------------------------
// Finish kernel.  grid = B, block = 256: finish_body for the latest solve (also the flush of the
// pipelined mode).
// ------------------------------------------------------------------------------
template <int GEO, bool LDSWIN, int NT>
__global__ __launch_bounds__(NT) void finish_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    finish_body<GEO, LDSWIN, NT, true>(p, blockIdx.x, p.part, p.cost, p.state, smem);
}

// ------------------------------------------------------------------------------
// DWA ("next" row N3): reference src/planners/local_planners/dwa.py:116-258.  NA constant-control candidates
// (the dynamic window grid, built by the host exactly as dwa.py:168-199 does) are rolled out with the same
// transit / aliasing as MPPI (dwa.py:224-227), costed with the stage cost against the sub-goal and the
// terminal cost against the goal, accumulated in fp32 in step order like `cost_batch +=` (dwa.py:251-256);
// argmin (first minimum, dwa.py:139), weights = softmax(-cost) (dwa.py:151).
// grid = B, block = 64 * ceil(NA / 64) <= 1024, lane = candidate.  LDS: [ window | red 2*16 ].
// ------------------------------------------------------------------------------
template <int GEO, bool LDSWIN>
__global__ void dwa_kernel(const SolveParams p, const float *__restrict__ actions, const float *__restrict__ stage_goal,
                           int NA, float *__restrict__ Xall, float *__restrict__ cost_out, float *__restrict__ w_out,
                           int *__restrict__ best_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T;
    float *win = smem;
    float *red = win + (LDSWIN ? p.WN * p.WN : 0);
    int *redi = reinterpret_cast<int *>(red + 16);
    const int tid = threadIdx.x, b = blockIdx.x, nthreads = blockDim.x;
    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];                 // terminal cost: the goal
    const float hx = stage_goal[b * 2 + 0], hy = stage_goal[b * 2 + 1];         // stage cost: the sub-goal
    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, tid, nthreads);
    }
    __syncthreads();
    const bool active = tid < NA;
    const int k = active ? tid : NA - 1;
    const float u0 = clampf(actions[((size_t)b * NA + k) * 2 + 0], p.umin0, p.umax0);   // transit re-clamps (robot_model.py:82-83)
    const float u1 = clampf(actions[((size_t)b * NA + k) * 2 + 1], p.umin1, p.umax1);
    Chain c;
    c.x = sx; c.y = sy; c.th = sth;
    sincos_spec(c.th, c.sn, c.cs);
    c.trav = trav_lookup<GEO, LDSWIN, true>(p, win, map, w, c.x, c.y);
    float *Xk = Xall ? Xall + ((size_t)b * NA + k) * (T + 1) * 3 : nullptr;
    float cost = 0.0f;
    for (int t = 0; t < T; ++t) {
        float xn, yn, tn;
        if (t == 0) chain_step<GEO, LDSWIN, true>(p, win, map, w, c, u0, u1, xn, yn, tn);
        else chain_step<GEO, LDSWIN, false>(p, win, map, w, c, u0, u1, xn, yn, tn);
        if (Xk && active) { Xk[3 * t] = xn; Xk[3 * t + 1] = yn; Xk[3 * t + 2] = tn; }
        const float dx = xn - hx, dy = yn - hy;
        cost = cost + (sqrt_cr(dx * dx + dy * dy) + (c.trav <= p.thr ? 1.0e4f : 0.0f));      // objectives.py:47-53
    }
    if (Xk && active) { Xk[3 * T] = c.x; Xk[3 * T + 1] = c.y; Xk[3 * T + 2] = c.th; }
    const float dxT = c.x - gx, dyT = c.y - gy;
    cost = cost + (sqrt_cr(dxT * dxT + dyT * dyT) + (c.trav <= p.thr ? 1.0e4f : 0.0f));       // dwa.py:256
    if (active) cost_out[(size_t)b * NA + tid] = cost;

    // argmin with first-index tie break, then softmax(-cost)
    float cm = active ? cost : INFINITY;
    int im = active ? tid : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float oc = __shfl_xor(cm, o);
        const int oi = __shfl_xor(im, o);
        if (oc < cm || (oc == cm && oi < im)) { cm = oc; im = oi; }
    }
    const int wv = tid >> 6, nw = nthreads >> 6;
    if ((tid & 63) == 0) { red[wv] = cm; redi[wv] = im; }
    __syncthreads();
    float cmin = red[0];
    int imin = redi[0];
    for (int i = 1; i < nw; ++i)
        if (red[i] < cmin || (red[i] == cmin && redi[i] < imin)) { cmin = red[i]; imin = redi[i]; }
    __syncthreads();
    const float e = active ? expf((-cost) - (-cmin)) : 0.0f;
    float es = wave_sum(e);
    if ((tid & 63) == 0) red[wv] = es;
    __syncthreads();
    float tot = 0.0f;
    for (int i = 0; i < nw; ++i) tot += red[i];
    if (active) w_out[(size_t)b * NA + tid] = e / tot;
    if (tid == 0) best_out[b] = imin;
}

// ------------------------------------------------------------------------------
// Sampled-slip rollouts (BASELINE config 3: "GP slip-regressor mean+var sampled per step").  The map holds the slip
// MEAN, slip_std its STD; every get_traversability is the observation-mode branch of traversability_model.py:65-69,
// 1 - clamp(Normal(mean, std)[cell].sample(), 0, 1), with its own draw: T in transit (robot_model.py:75), T+1 in the
// stage / terminal costs (objectives.py:50) per rollout.  The draws do not depend on the state, so they are
// produced up front, in parallel, and only the recurrence itself stays serial:
//   phase 1  8 waves   controls (noise -> clamp) and slip draws of all steps -> LDS tiles (Philox or injected)
//   phase 2  wave 0    the T-step chain on the LDS window of (mean, std) pairs; slot rows -> LDS
//            wave 1    control cost (fp64, step order)
//   phase 3  8 waves   per slot row: trajectory stores, sampled stage cost -> LDS (overwrites its draw)
//   phase 4  wave 0    stage-cost sum (fp64, step order), rollout cost, softmin statistics; all: weighted control sums
// The transit lookup of state t+1 and the stage-cost lookup of slot t hit the same cell (the un-clamped slot and
// its clamped successor index alike, grid_map.py:209), so the chain hands its cell index on with the slot row.
// grid = (ceil(K/64), B), block = 512, lane = rollout.
// LDS: [ slot rows (T+1) x 64 float4 | window WN^2 float2 | Zt TP x 64 | Zc TP x 64 | controls 2T x 65 | mean 2T |
//        mean*inv_var 2T | e 64 | control cost 64 ],  TP = T+1 rounded up to even.
// ------------------------------------------------------------------------------
constexpr int kSampledWaves = 8;
constexpr int kSampledThreads = 64 * kSampledWaves;

__host__ __device__ inline size_t sampled_lds_floats(int T, int WN)
{
    const size_t TP = (size_t)((T + 2) & ~1);
    return 4 * 64 * (size_t)(T + 1) + 2 * (size_t)WN * WN + 2 * 64 * TP + 2 * (size_t)T * kUPad + 4 * (size_t)T + 128;
}

template <int EPS, int GEO, bool STORE_U>
__global__ __launch_bounds__(kSampledThreads) void rollout_sampled_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T, K = p.K, WN2 = p.WN * p.WN;
    const int TP = (T + 2) & ~1;
    float4 *XL = reinterpret_cast<float4 *>(smem);
    float2 *win2 = reinterpret_cast<float2 *>(smem + 4 * 64 * (T + 1));
    float *Zt = reinterpret_cast<float *>(win2 + WN2), *Zc = Zt + 64 * TP;
    float *Ul = Zc + 64 * TP, *ml = Ul + 2 * T * kUPad, *mv = ml + 2 * T, *el = mv + 2 * T, *ad = el + 64;
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y;
    if (blockIdx.x == p.nblk) {
        // aux workgroup: weights, cost copy and X* of the previous solve (merged by its own last workgroup)
        finish_body<GEO, true, kSampledThreads>(p, b, nullptr, p.cost_prev, p.state_prev, smem);
        return;
    }
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = blockIdx.x * 64 + lane;
    const bool active = k < K;
    const int kk = active ? k : K - 1;
    const float *__restrict__ mu = p.map + (size_t)b * p.map_stride;
    const float *__restrict__ sg = p.slip_std + (size_t)b * p.map_stride;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];
    const Win w = window_origin<GEO>(p, sx, sy);
    const size_t Kp = (size_t)p.Kp;
    float *Xb = p.X + (size_t)b * (T + 1) * 3 * Kp + k;
    float *Ub = STORE_U ? p.U + (size_t)b * T * 2 * Kp + k : nullptr;

    BN_STAMP(0);
    // ---- phase 0: window of (mean, std) pairs, warm-start mean ----
    for (int e = tid; e < WN2; e += kSampledThreads) {
        const int r = e / p.WN, c = e - r * p.WN;
        const size_t g = (size_t)(w.wy0 + r) * p.G + (w.wx0 + c);
        win2[e] = make_float2(mu[g], sg[g]);
    }
    for (int j = tid; j < 2 * T; j += kSampledThreads) {
        const float m = p.mean[(size_t)b * 2 * T + j];
        ml[j] = m;
        mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);
    }
    if (blockIdx.x == 0 && tid < 3) p.state_copy[b * 3 + tid] = p.state[b * 3 + tid];
    __syncthreads();

    // ---- phase 1: controls and slip draws of every step ----
    {
        const int nE = (T + 1) >> 1, nS = TP >> 1;
        for (int q = wid; q < nE + nS; q += kSampledWaves) {
            if (q < nE) {
                produce_pair<EPS, STORE_U>(p, p.eps, b, kk, 2 * q, p.solve, ml, Ul, Ub, Kp, lane);
            } else {
                const int r0 = 2 * (q - nE), r1 = r0 + 1;
                float z[4];
                if (p.zt) {
                    z[0] = r0 < T ? p.zt[((size_t)b * T + r0) * K + kk] : 0.0f;
                    z[1] = r1 < T ? p.zt[((size_t)b * T + r1) * K + kk] : 0.0f;
                    z[2] = r0 <= T ? p.zc[((size_t)b * (T + 1) + r0) * K + kk] : 0.0f;
                    z[3] = r1 <= T ? p.zc[((size_t)b * (T + 1) + r1) * K + kk] : 0.0f;
                } else {
                    philox_slip_block(p.seed, p.solve, (uint32_t)b, (uint32_t)(kk + p.k0), (uint32_t)(q - nE), z);
                }
                Zt[r0 * 64 + lane] = z[0]; Zt[r1 * 64 + lane] = z[1];
                Zc[r0 * 64 + lane] = z[2]; Zc[r1 * 64 + lane] = z[3];
            }
        }
    }
    __syncthreads();
    BN_STAMP(1);

    // ---- phase 2: the chain (wave 0) and the control cost (wave 1) ----
    if (wid == 0) {
        SlipChain c;
        c.x = sx; c.y = sy; c.th = sth;                                   // mppi.py:160
        sincos_spec(c.th, c.sn, c.cs);
        c.e = slip_cell_safe<GEO, true>(p, w, sx, sy);
        float xn, yn, tn;
        slip_chain_step<GEO, true>(p, win2, w, c, Ul[lane], Ul[kUPad + lane], Zt[lane], xn, yn, tn);
        XL[lane] = make_float4(xn, yn, tn, __int_as_float(c.e));
        int t = 1;
        for (; t + 4 <= T; t += 4) {                  // controls and draws of four steps read up front: LDS latency off the chain
            float uq[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uq[i][0] = Ul[(2 * (t + i)) * kUPad + lane]; uq[i][1] = Ul[(2 * (t + i) + 1) * kUPad + lane];
                uq[i][2] = Zt[(t + i) * 64 + lane];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                slip_chain_step<GEO, false>(p, win2, w, c, uq[i][0], uq[i][1], uq[i][2], xn, yn, tn);
                __builtin_amdgcn_sched_barrier(0);
                XL[(t + i) * 64 + lane] = make_float4(xn, yn, tn, __int_as_float(c.e));
            }
        }
        for (; t < T; ++t) {
            slip_chain_step<GEO, false>(p, win2, w, c, Ul[(2 * t) * kUPad + lane], Ul[(2 * t + 1) * kUPad + lane], Zt[t * 64 + lane],
                                        xn, yn, tn);
            XL[t * 64 + lane] = make_float4(xn, yn, tn, __int_as_float(c.e));
        }
        XL[T * 64 + lane] = make_float4(c.x, c.y, c.th, __int_as_float(c.e));       // slot T: clamped, wrapped
    } else if (wid == 1) {
        double Ad = 0.0;
        for (int t = 0; t < T; ++t)
            Ad += (double)(p.lambda_ * (mv[2 * t] * Ul[(2 * t) * kUPad + lane] + mv[2 * t + 1] * Ul[(2 * t + 1) * kUPad + lane]));   // mppi.py:175-181
        ad[lane] = (float)Ad;
    }
    __syncthreads();
    BN_STAMP(2);

    // ---- phase 3: slot rows -> trajectory stores and sampled stage / terminal cost ----
    for (int t = wid; t <= T; t += kSampledWaves) {
        const float4 o = XL[t * 64 + lane];
        float *Xt = Xb + (size_t)(3 * t) * Kp;
        Xt[0] = o.x; Xt[Kp] = o.y; Xt[2 * Kp] = o.z;
        const float2 ms = win2[__float_as_int(o.w)];
        const float tc = trav_from_slip(ms.x, ms.y, Zc[t * 64 + lane]);                // objectives.py:50
        const float dx = o.x - gx, dy = o.y - gy;
        Zc[t * 64 + lane] = sqrt_cr(dx * dx + dy * dy) + (tc <= p.thr ? 1.0e4f : 0.0f);  // objectives.py:46-53
    }
    __syncthreads();
    BN_STAMP(3);

    // ---- phase 4: rollout cost and the workgroup's softmin statistics ----
    if (wid == 0) {
        double Sd = 0.0;
        for (int t = 0; t < T; ++t) Sd += (double)Zc[t * 64 + lane];
        const float cost = ((float)Sd + Zc[T * 64 + lane]) + ad[lane];                 // mppi.py:184-190
        if (active) p.cost[(size_t)b * K + k] = cost;
        const float zz = active ? (-cost) / p.lambda_ : -INFINITY;
        const float zmax = wave_max(zz);
        const float e = active ? expf(zz - zmax) : 0.0f;
        const float esum = wave_sum(e);
        el[lane] = e;
        if (lane == 0) {
            float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
            store_agent(part, zmax); store_agent(part + 1, esum);
        }
    }
    __syncthreads();
    float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
    for (int j = tid; j < 2 * T; j += kSampledThreads) {
        const float *col = Ul + j * kUPad;
        float acc = 0.0f;
#pragma unroll 16
        for (int q = 0; q < 64; ++q) acc = __builtin_fmaf(el[q], col[q], acc);
        store_agent(part + 2 + j, acc);
    }
    BN_STAMP(5);
    if (p.ustar_cur) ticket_merge<kSampledThreads>(p, b, smem);   // one-launch mode; the slot rows are dead: their LDS is the merge scratch
}

// The same solve without the LDS window (BN_FLAG_NO_LDS_WINDOW, or a window/horizon too large for the LDS):
// one wave per 64 rollouts, lookups from global memory, draws made in line.
template <int EPS, int GEO, bool STORE_U>
__global__ __launch_bounds__(64) void rollout_sampled_global_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T, K = p.K;
    float *ml = smem, *mv = ml + 2 * T, *Ul = mv + 2 * T, *el = Ul + 2 * T * kUPad;
    const int lane = threadIdx.x, b = blockIdx.y;
    const int k = blockIdx.x * 64 + lane;
    const bool active = k < K;
    const int kk = active ? k : K - 1;
    const float *__restrict__ mu = p.map + (size_t)b * p.map_stride;
    const float *__restrict__ sg = p.slip_std + (size_t)b * p.map_stride;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];
    const Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < 2 * T; j += 64) {
        const float m = p.mean[(size_t)b * 2 * T + j];
        ml[j] = m;
        mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);
    }
    if (blockIdx.x == 0 && lane < 3) p.state_copy[b * 3 + lane] = p.state[b * 3 + lane];
    __syncthreads();
    const size_t Kp = (size_t)p.Kp;
    float *Xb = p.X + (size_t)b * (T + 1) * 3 * Kp + k;
    float *Ub = STORE_U ? p.U + (size_t)b * T * 2 * Kp + k : nullptr;
    for (int t = 0; t < T; t += 2) produce_pair<EPS, STORE_U>(p, p.eps, b, kk, t, p.solve, ml, Ul, Ub, Kp, lane);
    __syncthreads();
    float x = sx, y = sy, th = sth;
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
    double Sd = 0.0, Ad = 0.0;
    for (int t = 0; t < T; ++t) {
        if (p.zt) {
            zq[t & 1] = p.zt[((size_t)b * T + t) * K + kk];
            zq[2 + (t & 1)] = p.zc[((size_t)b * (T + 1) + t) * K + kk];
        } else if ((t & 1) == 0) {
            philox_slip_block(p.seed, p.solve, (uint32_t)b, (uint32_t)(kk + p.k0), (uint32_t)(t >> 1), zq);
        }
        const float u0 = Ul[(2 * t) * kUPad + lane], u1 = Ul[(2 * t + 1) * kUPad + lane];
        const int e = slip_cell_safe<GEO, false>(p, w, x, y);
        const float trav = trav_from_slip(mu[e], sg[e], zq[t & 1]);                                     // robot_model.py:75
        float sn, cs;
        sincos_spec(th, sn, cs);
        const float xn = x + ((trav * u0) * cs) * p.dt, yn = y + ((trav * u0) * sn) * p.dt, tn = th + (trav * u1) * p.dt;
        float *Xt = Xb + (size_t)(3 * t) * Kp;
        Xt[0] = xn; Xt[Kp] = yn; Xt[2 * Kp] = tn;
        x = clampf(xn, p.x0, p.x_hi); y = clampf(yn, p.y0, p.y_hi); th = wrap_angle(tn);
        // stage cost on the aliased slot: its own, independent slip draw (objectives.py:50)
        const int ec = slip_cell_safe<GEO, false>(p, w, xn, yn);
        const float tc = trav_from_slip(mu[ec], sg[ec], zq[2 + (t & 1)]);
        const float dx = xn - gx, dy = yn - gy;
        Sd += (double)(sqrt_cr(dx * dx + dy * dy) + (tc <= p.thr ? 1.0e4f : 0.0f));
        Ad += (double)(p.lambda_ * (mv[2 * t] * u0 + mv[2 * t + 1] * u1));
    }
    {
        float *Xt = Xb + (size_t)(3 * T) * Kp;
        Xt[0] = x; Xt[Kp] = y; Xt[2 * Kp] = th;
    }
    if (p.zt) zq[2 + (T & 1)] = p.zc[((size_t)b * (T + 1) + T) * K + kk];
    else if ((T & 1) == 0) philox_slip_block(p.seed, p.solve, (uint32_t)b, (uint32_t)(kk + p.k0), (uint32_t)(T >> 1), zq);
    const int eT = slip_cell_safe<GEO, false>(p, w, x, y);
    const float tT = trav_from_slip(mu[eT], sg[eT], zq[2 + (T & 1)]);
    const float dxT = x - gx, dyT = y - gy;
    const float term = sqrt_cr(dxT * dxT + dyT * dyT) + (tT <= p.thr ? 1.0e4f : 0.0f);
    const float cost = ((float)Sd + term) + (float)Ad;
    if (active) p.cost[(size_t)b * K + k] = cost;
    const float zz = active ? (-cost) / p.lambda_ : -INFINITY;
    const float zmax = wave_max(zz);
    const float e = active ? expf(zz - zmax) : 0.0f;
    const float esum = wave_sum(e);
    el[lane] = e;
    __syncthreads();
    float *part = p.part + ((size_t)b * p.nblk + blockIdx.x) * (2 + 2 * T);
    for (int j = lane; j < 2 * T; j += 64) {
        const float *col = Ul + j * kUPad;
        float acc = 0.0f;
#pragma unroll 16
        for (int q = 0; q < 64; ++q) acc = __builtin_fmaf(el[q], col[q], acc);
        part[2 + j] = acc;
    }
    if (lane == 0) { part[0] = zmax; part[1] = esum; }
}

// The slip draws of one solve, exactly the stream the sampled kernels consume: zt (K,T), zc (K,T+1), zo (T).
__global__ void philox_slip_kernel(float *__restrict__ zt, float *__restrict__ zc, float *__restrict__ zo, uint64_t seed,
                                   uint64_t solve, int b, int K, int T)
{
    const int nS = ((T + 2) & ~1) >> 1;
    const size_t tot = (size_t)K * nS;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / nS), j = (int)(i - (size_t)k * nS);
        float z[4];
        philox_slip_block(seed, solve, (uint32_t)b, (uint32_t)k, (uint32_t)j, z);
        for (int s = 0; s < 2; ++s) {
            const int r = 2 * j + s;
            if (r < T) zt[(size_t)k * T + r] = z[s];
            if (r <= T) zc[(size_t)k * (T + 1) + r] = z[2 + s];
        }
    }
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; 4 * j < T; j += blockDim.x) {
            float z[4];
            philox_slip_block(seed, solve, (uint32_t)b, 0xffffffffu, (uint32_t)j, z);
            for (int s = 0; s < 4; ++s)
                if (4 * j + s < T) zo[4 * j + s] = z[s];
        }
}

// ------------------------------------------------------------------------------
// PlanetaryEnv mirror for B environments ("next" row N2): step (planetary_env.py:189-219) and collision_check
// (planetary_env.py:221-232) as stand-alone calls, for callers that drive the loop themselves; the fused episode of
// the pipelined solve uses the same env_advance.
// ------------------------------------------------------------------------------
template <int GEO>
__global__ void env_step_kernel(const SolveParams p, const float *__restrict__ actions, float *states, float *reward,
                                int *terminated, const float *z, uint64_t step)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    const EnvStep e = env_advance<GEO>(p, b, states[b * 3 + 0], states[b * 3 + 1], states[b * 3 + 2], actions[b * 2 + 0],
                                       actions[b * 2 + 1], z, step);
    states[b * 3 + 0] = e.x; states[b * 3 + 1] = e.y; states[b * 3 + 2] = e.th;
    reward[b] = e.reward;
    terminated[b] = e.reached ? 1 : 0;
}

// is_collisions[b, n] = (1 - clamp(Normal(mean, std)[cell(states[b, n])].sample(), 0, 1)) <= stuck_threshold: one fresh
// slip draw per position, like the observation-mode get_traversability the reference calls here.
template <int GEO>
__global__ void env_collision_kernel(const SolveParams p, const float *__restrict__ states, int N, float thr,
                                     const float *__restrict__ z, uint64_t draw, unsigned char *out)
{
    const size_t tot = (size_t)p.B * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / N);
        const int ix = clampi(raw_cell<GEO>(states[i * 3 + 0], p.x0, p.res, p.inv_res), 0, p.G - 1);
        const int iy = clampi(raw_cell<GEO>(states[i * 3 + 1], p.y0, p.res, p.inv_res), 0, p.G - 1);
        const size_t cell = (size_t)b * p.map_stride + (size_t)iy * p.G + ix;
        float zz;
        if (z) {
            zz = z[i];
        } else {
            const u32x4 q = philox4x32_10(u32x4{(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)draw, 0x434f4c4cu ^ (uint32_t)(draw >> 32)},
                                          (uint32_t)p.env_seed, (uint32_t)(p.env_seed >> 32));
            float z1;
            box_muller(q.x, q.y, zz, z1);
        }
        const float slip = zz * p.lat_std[cell] + p.lat_mean[cell];
        out[i] = (1.0f - clampf(slip, 0.0f, 1.0f)) <= thr ? 1 : 0;
    }
}

// ---- the library's device math on caller-supplied inputs (test hook: bn_device_math_eval) ----
__global__ void math_eval_kernel(int fn, const float *__restrict__ in, float *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        float sn, cs, r;
        switch (fn) {
        case 0: r = sqrt_cr(x); break;
        case 1: sincos_spec(x, sn, cs); r = sn; break;
        case 2: sincos_spec(x, sn, cs); r = cs; break;
        case 3: r = wrap_angle(x); break;
        default: r = wrap_angle_near(x); break;
        }
        out[i] = r;
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

__global__ void philox_noise_kernel(float *__restrict__ eps, uint64_t seed, uint64_t solve, int b, int K, int T, int k0)
{   // eps (K, T, 2) of one instance, exactly the stream rollout_kernel<kEpsPhilox> consumes
    const int npair = (T + 1) / 2;                  // pair p = steps (2p, 2p+1)
    const size_t tot = (size_t)K * npair;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / npair), pr = (int)(i - (size_t)k * npair);
        float z[4];
        philox_eps_pair(seed, solve, (uint32_t)b, (uint32_t)(k + k0), (uint32_t)pr, z);
        const int t = 2 * pr;
        {
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
    const dim3 grid(p.nblk + (p.have_prev ? 1 : 0), p.B);
    if (p.wave_kernel) {
        const size_t ldsw = wave_lds_bytes(p);
        hipError_t e = ensure_lds(rollout_wave_kernel<EPS, GEO, LDSWIN>, ldsw);
        if (e != hipSuccess) return e;
        rollout_wave_kernel<EPS, GEO, LDSWIN><<<grid, dim3(64), ldsw, s>>>(p);
        return hipGetLastError();
    }
    const size_t lds = rollout_lds_bytes(p);
    if (p.ticket) {
        hipError_t e = ensure_lds(rollout_kernel<EPS, GEO, LDSWIN, STORE_U, true>, lds);
        if (e != hipSuccess) return e;
        rollout_kernel<EPS, GEO, LDSWIN, STORE_U, true><<<grid, dim3(kRolloutThreads), lds, s>>>(p);
    } else {
        hipError_t e = ensure_lds(rollout_kernel<EPS, GEO, LDSWIN, STORE_U, false>, lds);
        if (e != hipSuccess) return e;
        rollout_kernel<EPS, GEO, LDSWIN, STORE_U, false><<<grid, dim3(kRolloutThreads), lds, s>>>(p);
    }
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
    if (p.nblk > 32) {                    // sizes the pipelined mode does not take: a wide tail (merge tiles, weights)
        hipError_t e = ensure_lds(finish_kernel<GEO, LDSWIN, kWideFinishThreads>, lds);
        if (e != hipSuccess) return e;
        finish_kernel<GEO, LDSWIN, kWideFinishThreads><<<dim3(p.B), dim3(kWideFinishThreads), lds, s>>>(p);
    } else {
        hipError_t e = ensure_lds(finish_kernel<GEO, LDSWIN, kFinishThreads>, lds);
        if (e != hipSuccess) return e;
        finish_kernel<GEO, LDSWIN, kFinishThreads><<<dim3(p.B), dim3(kFinishThreads), lds, s>>>(p);
    }
    return hipGetLastError();
}

}  // namespace

bool sampled_fused(const SolveParams &p)
{
    // the multi-wave kernel needs the LDS window and room for its tiles; its LDS also holds the aux tail / merge scratch
    const size_t need = sizeof(float) * sampled_lds_floats(p.T, p.WN);
    const size_t tail = finish_lds_bytes(p) + sizeof(float) * 64;
    return p.slip_on && p.WN > 0 && need <= 160 * 1024 && tail <= need && p.nblk <= 1024;
}

int rollout_blocks_per_cu(const SolveParams &p)
{
    int n = 0;
    const size_t lds = rollout_lds_bytes(p);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rollout_kernel<kEpsPhilox, kGeoPow2Origin0, true, false, false>, kRolloutThreads, lds) != hipSuccess) return -1;
    return n;
}

size_t wave_lds_bytes(const SolveParams &p)
{
    const size_t own = (size_t)p.WN * p.WN + 4 * (size_t)p.T + 64 + (size_t)p.nblk + 32;
    return std::max(sizeof(float) * own, finish_lds_bytes(p) + 256);      // the aux workgroup runs finish_body in the same LDS
}

size_t rollout_lds_bytes(const SolveParams &p)
{
    return sizeof(float) * ((size_t)p.WN * p.WN + 4 * (size_t)p.T + 2 * (size_t)p.T * kUPad + 2 * kChunk * 4 * 64 + 5 * 64 + 64);
}

size_t finish_lds_bytes(const SolveParams &p)
{
    const size_t slip = p.slip_on ? 2 * (size_t)p.WN * p.WN + (size_t)p.T + 16 : 0;    // (mean, std) window + the draws of X*
    const size_t groups = (p.nblk > 64 && p.nblk <= 64 * 16) ? (size_t)((p.nblk + 15) / 16) * (2 + 2 * (size_t)p.T) : 0;   // two-level merge rows
    return sizeof(float) * ((size_t)p.WN * p.WN + 2 * (size_t)p.T + (size_t)p.nblk + 32 + groups + slip);
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

template <int EPS, int GEO>
hipError_t launch_sampled_g(const SolveParams &p, hipStream_t s)
{
    const size_t lds_w = sizeof(float) * sampled_lds_floats(p.T, p.WN);
    const dim3 grid(p.nblk + (sampled_fused(p) && p.have_prev ? 1 : 0), p.B);
    if (sampled_fused(p)) {
#define BN_SL(SU)                                                                                                      \
    do { hipError_t e = ensure_lds(rollout_sampled_kernel<EPS, GEO, SU>, lds_w); if (e != hipSuccess) return e;        \
         rollout_sampled_kernel<EPS, GEO, SU><<<grid, dim3(kSampledThreads), lds_w, s>>>(p); } while (0)
        if (p.U) BN_SL(true); else BN_SL(false);
#undef BN_SL
    } else {
        const size_t lds = sizeof(float) * (4 * (size_t)p.T + 2 * (size_t)p.T * kUPad + 64);
#define BN_SL(SU)                                                                                                      \
    do { hipError_t e = ensure_lds(rollout_sampled_global_kernel<EPS, GEO, SU>, lds); if (e != hipSuccess) return e;   \
         rollout_sampled_global_kernel<EPS, GEO, SU><<<grid, dim3(64), lds, s>>>(p); } while (0)
        if (p.U) BN_SL(true); else BN_SL(false);
#undef BN_SL
    }
    return hipGetLastError();
}

template <int EPS>
hipError_t launch_sampled_e(const SolveParams &p, hipStream_t s)
{
    switch (geo_of(p)) {
    case kGeoPow2Origin0: return launch_sampled_g<EPS, kGeoPow2Origin0>(p, s);
    case kGeoPow2: return launch_sampled_g<EPS, kGeoPow2>(p, s);
    default: return launch_sampled_g<EPS, kGeoGeneral>(p, s);
    }
}

hipError_t launch_rollout_sampled(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    switch (mode) {
    case kEpsPhilox: return launch_sampled_e<kEpsPhilox>(p, s);
    case kEpsKT2: return launch_sampled_e<kEpsKT2>(p, s);
    default: return launch_sampled_e<kEpsT2K>(p, s);
    }
}

hipError_t launch_dwa(const SolveParams &p, const float *actions, const float *stage_goal, int NA, float *Xall, float *cost,
                      float *w, int *best, hipStream_t s)
{
    const int threads = ((NA + 63) / 64) * 64;
    const size_t lds = sizeof(float) * ((size_t)p.WN * p.WN + 32);
    const bool win = p.WN > 0;
#define BN_DWA_LAUNCH(GEO_)                                                                                          \
    do {                                                                                                             \
        if (win) { hipError_t e = ensure_lds(dwa_kernel<GEO_, true>, lds); if (e != hipSuccess) return e;           \
                   dwa_kernel<GEO_, true><<<dim3(p.B), dim3(threads), lds, s>>>(p, actions, stage_goal, NA, Xall, cost, w, best); } \
        else { dwa_kernel<GEO_, false><<<dim3(p.B), dim3(threads), lds, s>>>(p, actions, stage_goal, NA, Xall, cost, w, best); }   \
    } while (0)
    switch (geo_of(p)) {
    case kGeoPow2Origin0: BN_DWA_LAUNCH(kGeoPow2Origin0); break;
    case kGeoPow2: BN_DWA_LAUNCH(kGeoPow2); break;
    default: BN_DWA_LAUNCH(kGeoGeneral); break;
    }
#undef BN_DWA_LAUNCH
    return hipGetLastError();
}

hipError_t launch_env_step(const SolveParams &p, const float *actions, float *states, float *reward, int *terminated, const float *z,
                           uint64_t step, hipStream_t s)
{
    const dim3 grid((p.B + 63) / 64), block(64);
    switch (geo_of(p)) {
    case kGeoPow2Origin0: env_step_kernel<kGeoPow2Origin0><<<grid, block, 0, s>>>(p, actions, states, reward, terminated, z, step); break;
    case kGeoPow2: env_step_kernel<kGeoPow2><<<grid, block, 0, s>>>(p, actions, states, reward, terminated, z, step); break;
    default: env_step_kernel<kGeoGeneral><<<grid, block, 0, s>>>(p, actions, states, reward, terminated, z, step); break;
    }
    return hipGetLastError();
}

hipError_t launch_env_collision(const SolveParams &p, const float *states, int N, float thr, const float *z, uint64_t draw,
                                unsigned char *out, hipStream_t s)
{
    const dim3 grid = grid_for((size_t)p.B * N);
    switch (geo_of(p)) {
    case kGeoPow2Origin0: env_collision_kernel<kGeoPow2Origin0><<<grid, 256, 0, s>>>(p, states, N, thr, z, draw, out); break;
    case kGeoPow2: env_collision_kernel<kGeoPow2><<<grid, 256, 0, s>>>(p, states, N, thr, z, draw, out); break;
    default: env_collision_kernel<kGeoGeneral><<<grid, 256, 0, s>>>(p, states, N, thr, z, draw, out); break;
    }
    return hipGetLastError();
}

hipError_t launch_math_eval(int fn, const float *in, float *out, size_t n, hipStream_t s)
{
    math_eval_kernel<<<grid_for(n), 256, 0, s>>>(fn, in, out, n);
    return hipGetLastError();
}

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

hipError_t launch_philox_slip(float *zt, float *zc, float *zo, uint64_t seed, uint64_t solve, int b, int K, int T, hipStream_t s)
{
    philox_slip_kernel<<<grid_for((size_t)K * (T / 2 + 1)), 256, 0, s>>>(zt, zc, zo, seed, solve, b, K, T);
    return hipGetLastError();
}

hipError_t launch_philox_noise(float *eps_kt2, uint64_t seed, uint64_t solve, int b, int K, int T, int k0, hipStream_t s)
{
    philox_noise_kernel<<<grid_for((size_t)K * ((T + 1) / 2)), 256, 0, s>>>(eps_kt2, seed, solve, b, K, T, k0);
    return hipGetLastError();
}

}  // namespace bn
