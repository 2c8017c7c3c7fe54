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
// This translation unit holds the tail kernel, DWA, the environment mirror and the small helpers; the rollout kernels
// live in rollout_role_*.hip (five-wave role split, one file per noise source), rollout_wave.hip (throughput
// variant) and rollout_sampled.hip (config 3); the device code they share is mppi_device.h.
#include "mppi_device.h"
#include <cstring>

namespace bn {

namespace {

// ------------------------------------------------------------------------------
// Finish kernel.  grid = B, block = NT (320, or 1024 above 32 workgroups): finish_body for the latest solve (also the flush of the
// pipelined mode).
// ------------------------------------------------------------------------------
template <int GEO, bool LDSWIN, int NT, bool REF>
__global__ __launch_bounds__(NT) void finish_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    finish_body<GEO, LDSWIN, NT, true, false, true, REF>(p, blockIdx.x, p.part, p.cost, p.state, smem);
}

// ------------------------------------------------------------------------------
// K-sharded solve: the merge of ALL shards' partial rows on its own -- U* (= the next mean) and the softmin statistics, which is
// all the NEXT solve's rollouts wait for.  The rest of the tail (X*, the shard's weights, the cost copy) follows as
// finish_kernel with p.tail_merged on a second stream, beside the next solve's rollouts (bn_mppi_shard_solve_async).
// The arithmetic is the stand-alone tail's (merge_group per 16 rows, then merge_partials over the group rows; plain merge_partials
// up to 64 rows): the same bits whoever runs it.  More than 64 rows: one workgroup PER GROUP, its four waves splitting the
// columns, so that every row is in flight at once (one workgroup walking 16 groups x 4 column blocks took 7.7 us of a 42 us solve at
// 256 rows); the last group to finish (ticket) merges the group rows.  grid = groups (or 1), block = 256.
// LDS: [ us 2T | scale 64 | red 32 | flag ].
// ------------------------------------------------------------------------------
constexpr int kShardMergeThreads = 256;
__global__ __launch_bounds__(kShardMergeThreads) void shard_merge_kernel(const SolveParams p, float *grows, int *ticket)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T, tid = threadIdx.x, PS = 2 + 2 * p.T;
    float *us = smem, *sc = us + 2 * T, *red = sc + 64;
    int *flag = reinterpret_cast<int *>(red + 32);
    float m, S;
    if (p.nblk > 64) {
        const int g = blockIdx.x, ng = gridDim.x;
        merge_group<false, true>(p.part, g * kGroupRows, min(kGroupRows, p.nblk - g * kGroupRows), T, tid & 63, tid >> 6, kShardMergeThreads / 64,
                                 grows + (size_t)g * PS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *flag = (atomicAdd(ticket, 1) == ng - 1) ? 1 : 0;
        __syncthreads();
        if (!*flag) return;
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next solve (stream-ordered)
        merge_partials<kShardMergeThreads, true, false>(grows, ng, T, us, sc, red, tid, m, S, false, MergeLoads{});
    } else {
        merge_partials<kShardMergeThreads, false, false>(p.part, p.nblk, T, us, sc, red, tid, m, S, false, MergeLoads{});
    }
    for (int j = tid; j < 2 * T; j += kShardMergeThreads) {
        p.ustar_cur[j] = us[j];
        p.ustar[j] = us[j];                             // U* is on the handle's stream from here on (the side-stream tail leaves it alone: ustar_written)
        if (p.mean_used) p.mean_used[j] = p.mean[j];    // what this solve sampled around
        p.mean[j] = us[j];                              // _previous_action_seq = U*, no shift (mppi.py:217)
    }
    if (tid == 0) { p.stats_cur[0] = m; p.stats_cur[1] = S; }
}

// ------------------------------------------------------------------------------
// Re-roll: rows of _state_seq_batch (mppi.py:119-125, 160-165) of the LATEST solve regenerated on demand -- the same noise
// (Philox stream position or the caller's eps), the mean that solve sampled around (mean_used), the state it started from,
// the same device functions in the same order as the rollout kernels: bit-identical to what a full-API solve stored.
// This is what lets lean mode (BN_FLAG_LEAN) skip the 12*K*(T+1)-byte trajectory dump and still serve get_top_samples.
// grid = ceil(n / 256) workgroups of instance b, block = 256; thread i rolls rollout idx[i] (idx == nullptr: rollout i).
// out (n, T+1, 3) in the reference's layout.  LDS: [ window ].
// ------------------------------------------------------------------------------
template <int EPS, int GEO, bool LDSWIN>
__global__ __launch_bounds__(256) void reroll_kernel(const SolveParams p, int b, const int *__restrict__ idx, int n,
                                                      float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T, tid = threadIdx.x;
    float *win = smem;
    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
    const float *ml = p.mean_used + (size_t)b * 2 * T;
    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, tid, 256);
    }
    __syncthreads();
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const int k = idx ? idx[i] : i;
        float *Xo = out + (size_t)i * (T + 1) * 3;
        Chain c;
        c.x = sx; c.y = sy; c.th = sth;                   // mppi.py:160
        sincos_spec(c.th, c.sn, c.cs);
        c.trav = trav_lookup<GEO, LDSWIN, true>(p, win, map, w, c.x, c.y);
        const bool ref = p.ref_order != 0;                    // (uniform; this kernel is not on the solve path)
        for (int t = 0; t < T; t += 2) {
            float e[4], xn, yn, tn;
            noise_pair<EPS>(p, b, k, t, e);
            const float u0 = clampf(ml[2 * t] + p.sigma0 * e[0], p.umin0, p.umax0);          // mppi.py:152-157
            const float u1 = clampf(ml[2 * t + 1] + p.sigma1 * e[1], p.umin1, p.umax1);
            if (ref && t == 0) chain_step<GEO, LDSWIN, true, true, false, 0, true>(p, win, map, w, c, u0, u1, xn, yn, tn);      // (FIRST: the general heading wrap)
            else if (ref) chain_step<GEO, LDSWIN, false, true, false, 0, true>(p, win, map, w, c, u0, u1, xn, yn, tn);
            else if (t == 0) chain_step<GEO, LDSWIN, true>(p, win, map, w, c, u0, u1, xn, yn, tn);
            else chain_step<GEO, LDSWIN, false>(p, win, map, w, c, u0, u1, xn, yn, tn);
            Xo[3 * t] = xn; Xo[3 * t + 1] = yn; Xo[3 * t + 2] = tn;
            if (t + 1 < T) {
                const float v0 = clampf(ml[2 * t + 2] + p.sigma0 * e[2], p.umin0, p.umax0);
                const float v1 = clampf(ml[2 * t + 3] + p.sigma1 * e[3], p.umin1, p.umax1);
                if (ref) chain_step<GEO, LDSWIN, false, true, false, 0, true>(p, win, map, w, c, v0, v1, xn, yn, tn);
                else chain_step<GEO, LDSWIN, false>(p, win, map, w, c, v0, v1, xn, yn, tn);
                Xo[3 * t + 3] = xn; Xo[3 * t + 4] = yn; Xo[3 * t + 5] = tn;
            }
        }
        Xo[3 * T] = c.x; Xo[3 * T + 1] = c.y; Xo[3 * T + 2] = c.th;
    }
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
                           int *__restrict__ best_out, float *__restrict__ best_states, float *__restrict__ best_action)
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
    const bool ref = p.ref_order != 0;
    for (int t = 0; t < T; ++t) {
        float xn, yn, tn;
        if (ref && t == 0) chain_step<GEO, LDSWIN, true, true, false, 0, true>(p, win, map, w, c, u0, u1, xn, yn, tn);
        else if (ref) chain_step<GEO, LDSWIN, false, true, false, 0, true>(p, win, map, w, c, u0, u1, xn, yn, tn);
        else if (t == 0) chain_step<GEO, LDSWIN, true>(p, win, map, w, c, u0, u1, xn, yn, tn);
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
    if (tid == 0) {
        best_out[b] = imin;
        if (best_action) {                             // optimal_action_seq = actions[argmin] (dwa.py:140); also the next window's centre (dwa.py:147)
            best_action[b * 2 + 0] = actions[((size_t)b * NA + imin) * 2 + 0];
            best_action[b * 2 + 1] = actions[((size_t)b * NA + imin) * 2 + 1];
        }
    }
    // optimal_state_seq = the argmin candidate's trajectory (dwa.py:139-143): its stores are complete and visible to
    // the workgroup since the barriers above
    if (Xall && best_states)
        for (int i = tid; i < (T + 1) * 3; i += nthreads) best_states[(size_t)b * (T + 1) * 3 + i] = Xall[((size_t)b * NA + imin) * (T + 1) * 3 + i];
}

// ------------------------------------------------------------------------------
// DWA host geometry on the device (so that DWA.forward needs no host round trip): the dynamic window grid and the
// sub-goal.  grid = B, block = 256.
//   window   dwa.py:168-199: lo = max(u_min, prev - a_lim * dt), hi = min(u_max, prev + a_lim * dt) around the previous first
//            control; vs = linspace(lo_v, hi_v, nv), ws = linspace(lo_w, hi_w, nw); actions = cartesian_prod(vs, ws) (v major).
//            linspace as ATen's scalar kernel computes it: step = (end - start) / (n - 1); element i < n/2 is
//            start + step * i, the others end - step * (n - 1 - i).  (On AVX2 hosts torch's vectorised path evaluates the
//            first 8 elements from `start` alone, so torch itself is machine dependent in the last bit; this is the form
//            AVX-512 hosts and every scalar tail use.)
//   sub-goal dwa.py:240-244 + 260-285, evaluated like the reference on candidate 0's ALIASED slot-0 state (the start state
//            advanced by one un-clamped, un-wrapped step of candidate 0 = (lo_v, lo_w)): nearest path point with
//            |bearing| < pi/2 and distance > lookahead -- the first point at that distance -- else the path's end.
// ------------------------------------------------------------------------------
__device__ __forceinline__ float linspace_at(float start, float end, int n, int i)
{
    if (n == 1) return start;
    const float step = (end - start) / (float)(n - 1);
    return i < n / 2 ? start + step * (float)i : end - step * (float)(n - 1 - i);
}

template <int GEO>
__global__ __launch_bounds__(256) void dwa_window_kernel(const SolveParams p, const float *__restrict__ prev_action, float alim0, float alim1,
                                                         float dwa_dt, int nv, int nw, const float *__restrict__ path, int P, float lookahead,
                                                         float *__restrict__ actions, float *__restrict__ stage_goal)
{
    __shared__ float red[8];
    __shared__ int redi[8];
    __shared__ float sel[3];
    const int b = blockIdx.x, tid = threadIdx.x, NA = nv * nw;
    const float pv = prev_action[b * 2 + 0], pw = prev_action[b * 2 + 1];
    const float lo0 = fmaxf(p.umin0, pv - alim0 * dwa_dt), hi0 = fminf(p.umax0, pv + alim0 * dwa_dt);
    const float lo1 = fmaxf(p.umin1, pw - alim1 * dwa_dt), hi1 = fminf(p.umax1, pw + alim1 * dwa_dt);
    for (int k = tid; k < NA; k += 256) {
        const int iv = k / nw, iw = k - iv * nw;
        actions[((size_t)b * NA + k) * 2 + 0] = linspace_at(lo0, hi0, nv, iv);
        actions[((size_t)b * NA + k) * 2 + 1] = linspace_at(lo1, hi1, nw, iw);
    }
    if (!path || P < 1) {                               // no reference path: the stage cost runs against the goal (dwa.py:243-247)
        if (tid < 2) stage_goal[b * 2 + tid] = p.goal[b * 2 + tid];
        return;
    }
    if (tid == 0) {                                     // candidate 0's slot 0 after the rollouts (aliasing, robot_model.py:86-88)
        const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
        const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
        const Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
        const float trav = trav_lookup<GEO, false, true>(p, nullptr, map, w, sx, sy);
        const float v = clampf(linspace_at(lo0, hi0, nv, 0), p.umin0, p.umax0), om = clampf(linspace_at(lo1, hi1, nw, 0), p.umin1, p.umax1);
        float sn, cs;
        sincos_spec(sth, sn, cs);
        sel[0] = sx + ((trav * v) * cs) * p.dt;
        sel[1] = sy + ((trav * v) * sn) * p.dt;
        sel[2] = sth + (trav * om) * p.dt;
    }
    __syncthreads();
    const float x = sel[0], y = sel[1], th = sel[2];
    float best = INFINITY;
    for (int i = tid; i < P; i += 256) {
        const float dx = path[2 * i] - x, dy = path[2 * i + 1] - y;
        const float dist = sqrt_cr(dx * dx + dy * dy);
        const float ang = atan2f(dy, dx) - th;
        if (fabsf(ang) < kPi / 2.0f && dist > lookahead) best = fminf(best, dist);
    }
    best = -wave_max(-best);
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    best = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    int idx = 0x7fffffff;
    if (best < INFINITY)
        for (int i = tid; i < P; i += 256) {
            const float dx = path[2 * i] - x, dy = path[2 * i + 1] - y;
            if (sqrt_cr(dx * dx + dy * dy) == best) { idx = i; break; }      // torch.where(distances == min)[0][0]: over ALL points
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) idx = min(idx, __shfl_xor(idx, o));
    if ((tid & 63) == 0) redi[tid >> 6] = idx;
    __syncthreads();
    if (tid == 0) {
        idx = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
        if (best == INFINITY || idx >= P) idx = P - 1;                       // nothing ahead: the path's last point
        stage_goal[b * 2 + 0] = path[2 * idx];
        stage_goal[b * 2 + 1] = path[2 * idx + 1];
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
        case 5: r = sqrt_cr_normal(x); break;
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

template <int GEO, bool LDSWIN, bool REF>
hipError_t launch_finish_r(const SolveParams &p, hipStream_t s)
{
    const size_t lds = finish_lds_bytes(p);
    if (p.nblk > 32) {                    // sizes the pipelined mode does not take: a wide tail (merge tiles, weights)
        hipError_t e = ensure_lds(finish_kernel<GEO, LDSWIN, kWideFinishThreads, REF>, lds);
        if (e != hipSuccess) return e;
        finish_kernel<GEO, LDSWIN, kWideFinishThreads, REF><<<dim3(p.B), dim3(kWideFinishThreads), lds, s>>>(p);
    } else {
        hipError_t e = ensure_lds(finish_kernel<GEO, LDSWIN, kFinishThreads, REF>, lds);
        if (e != hipSuccess) return e;
        finish_kernel<GEO, LDSWIN, kFinishThreads, REF><<<dim3(p.B), dim3(kFinishThreads), lds, s>>>(p);
    }
    return hipGetLastError();
}

template <int GEO, bool LDSWIN>
hipError_t launch_finish_t(const SolveParams &p, hipStream_t s)
{
    return p.ref_order ? launch_finish_r<GEO, LDSWIN, true>(p, s) : launch_finish_r<GEO, LDSWIN, false>(p, s);
}

}  // namespace

size_t wave_lds_bytes(const SolveParams &p)
{
    // [ A = max(window + (mean, mean @ inv_cov) rows, the epilogue's tile over them) | mean 2T | e 64 | merge scratch | parked chunk's tile ]
    const size_t wn_f = ((size_t)p.WN * p.WN + 3) & ~(size_t)3;
    const size_t own = std::max(wn_f + 4 * ((size_t)p.T + 1), (size_t)16 * kUPad + 4) + 2 * (size_t)p.T + 4 + 64 + (size_t)p.nblk + 32 + 4 +
                       (p.lds_park ? (size_t)16 * kUPad : 0);
#if defined(BN_EXPERIMENTS) && defined(BN_WAVE_LDS_PAD)      // measurement builds (csrc/experiments.h): fewer workgroups per CU
    return std::max(sizeof(float) * own, finish_lds_bytes(p) + 256) + BN_WAVE_LDS_PAD;
#endif
    return std::max(sizeof(float) * own, finish_lds_bytes(p) + 256);      // the aux workgroup runs finish_body in the same LDS
}

size_t lat_lds_bytes(const SolveParams &p)
{
    // [ ring (T + 1) x 64 x float4 | fin 5 x 64 | e 64 | progress 4 | window | mean 2T | mean*inv_var 2T | control tile 2T x 65 ]
    if (p.WN <= 0) return 0;
    const size_t wcap = (size_t)p.WN + 2 * (size_t)p.spec_extra;
    // (+ 2 x kChunk rows of slack behind the tile: the chain wave reads the controls of "the next chunk" with immediate offsets)
    const size_t own = ((size_t)p.T + 1) * 256 + 5 * 64 + 64 + 8 + wcap * wcap + 4 * (size_t)p.T + 2 * ((size_t)p.T + 2 * kChunk) * kUPad;
    const size_t bytes = std::max(sizeof(float) * own, finish_lds_bytes(p) + 256);      // the aux workgroup runs finish_body in the same LDS
    return bytes <= 160 * 1024 ? bytes : 0;
}

size_t rollout_lds_bytes(const SolveParams &p)
{
    return sizeof(float) * ((size_t)p.WN * p.WN + 4 * (size_t)p.T + 2 * (size_t)p.T * kUPad + 2 * kChunk * 4 * 64 + 5 * 64 + 64);
}

size_t finish_lds_bytes(const SolveParams &p)
{
    const size_t slip = p.slip_on ? 2 * (size_t)p.WN * p.WN + (size_t)p.T + 16 : 0;    // (mean, std) window + the draws of X*
    const size_t groups = (p.nblk > 64 && p.nblk <= 64 * 16) ? (size_t)((p.nblk + 15) / 16) * (2 + 2 * (size_t)p.T) : 0;   // two-level merge rows
    return sizeof(float) * ((size_t)p.WN * p.WN + 2 * (size_t)p.T + 3 * ((size_t)p.T + 1) + (size_t)p.nblk + 32 + groups + slip);
}

size_t finish_lds_bytes_for(SolveParams p, bool sampled)
{
    p.slip_on = sampled ? 1 : 0;
    return finish_lds_bytes(p);
}

hipError_t launch_rollout(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    if (p.wave_kernel) return launch_rollout_wave(p, mode, s);
    if (p.ref_order)
        switch (mode) {
        case kEpsPhilox: return launch_rollout_role_ref_philox(p, s);
        case kEpsKT2: return launch_rollout_role_ref_kt2(p, s);
        default: return launch_rollout_role_ref_t2k(p, s);
        }
    switch (mode) {
    case kEpsPhilox: return launch_rollout_role_philox(p, s);
    case kEpsKT2: return launch_rollout_role_kt2(p, s);
    default: return launch_rollout_role_t2k(p, s);
    }
}

hipError_t launch_rollout_lat_self(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    if (p.ref_order)
        switch (mode) {
        case kEpsPhilox: return launch_rollout_lat_self_ref_philox(p, s);
        case kEpsKT2: return launch_rollout_lat_self_ref_kt2(p, s);
        default: return launch_rollout_lat_self_ref_t2k(p, s);
        }
    switch (mode) {
    case kEpsPhilox: return launch_rollout_lat_self_philox(p, s);
    case kEpsKT2: return launch_rollout_lat_self_kt2(p, s);
    default: return launch_rollout_lat_self_t2k(p, s);
    }
}

hipError_t launch_shard_merge(const SolveParams &p, float *group_rows, int *ticket, hipStream_t s)
{
    if (p.nblk > 64 * kGroupRows) return hipErrorInvalidValue;       // (K > 65536: not sharded through the library's exchange)
    const size_t lds = sizeof(float) * (2 * (size_t)p.T + 64 + 32 + 4);
    const unsigned ng = p.nblk > 64 ? (unsigned)((p.nblk + kGroupRows - 1) / kGroupRows) : 1u;
    hipError_t e = ensure_lds(shard_merge_kernel, lds);
    if (e != hipSuccess) return e;
    shard_merge_kernel<<<dim3(ng), dim3(kShardMergeThreads), lds, s>>>(p, group_rows, ticket);
    return hipGetLastError();
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

namespace {
template <int EPS, int GEO>
hipError_t launch_reroll_g(const SolveParams &p, int b, const int *idx, int n, float *out, hipStream_t s)
{
    const size_t lds = sizeof(float) * ((size_t)p.WN * p.WN + 4);
    const dim3 grid((unsigned)std::min(256, (n + 255) / 256));
    if (p.WN > 0) {
        hipError_t e = ensure_lds(reroll_kernel<EPS, GEO, true>, lds);
        if (e != hipSuccess) return e;
        reroll_kernel<EPS, GEO, true><<<grid, dim3(256), lds, s>>>(p, b, idx, n, out);
    } else {
        reroll_kernel<EPS, GEO, false><<<grid, dim3(256), lds, s>>>(p, b, idx, n, out);
    }
    return hipGetLastError();
}
template <int EPS>
hipError_t launch_reroll_e(const SolveParams &p, int b, const int *idx, int n, float *out, hipStream_t s)
{
    switch (geo_of(p)) {
    case kGeoPow2Origin0: return launch_reroll_g<EPS, kGeoPow2Origin0>(p, b, idx, n, out, s);
    case kGeoPow2: return launch_reroll_g<EPS, kGeoPow2>(p, b, idx, n, out, s);
    default: return launch_reroll_g<EPS, kGeoGeneral>(p, b, idx, n, out, s);
    }
}
}  // namespace

hipError_t launch_reroll(const SolveParams &p, EpsMode mode, int b, const int *idx, int n, float *out, hipStream_t s)
{
    switch (mode) {
    case kEpsPhilox: return launch_reroll_e<kEpsPhilox>(p, b, idx, n, out, s);
    case kEpsKT2: return launch_reroll_e<kEpsKT2>(p, b, idx, n, out, s);
    default: return launch_reroll_e<kEpsT2K>(p, b, idx, n, out, s);
    }
}

hipError_t launch_dwa_window(const SolveParams &p, const float *prev_action, const float a_lim[2], float dwa_dt, int nv, int nw,
                             const float *path, int P, float lookahead, float *actions, float *stage_goal, hipStream_t s)
{
    switch (geo_of(p)) {
    case kGeoPow2Origin0: dwa_window_kernel<kGeoPow2Origin0><<<dim3(p.B), dim3(256), 0, s>>>(p, prev_action, a_lim[0], a_lim[1], dwa_dt, nv, nw, path, P, lookahead, actions, stage_goal); break;
    case kGeoPow2: dwa_window_kernel<kGeoPow2><<<dim3(p.B), dim3(256), 0, s>>>(p, prev_action, a_lim[0], a_lim[1], dwa_dt, nv, nw, path, P, lookahead, actions, stage_goal); break;
    default: dwa_window_kernel<kGeoGeneral><<<dim3(p.B), dim3(256), 0, s>>>(p, prev_action, a_lim[0], a_lim[1], dwa_dt, nv, nw, path, P, lookahead, actions, stage_goal); break;
    }
    return hipGetLastError();
}

hipError_t launch_dwa(const SolveParams &p, const float *actions, const float *stage_goal, int NA, float *Xall, float *cost,
                      float *w, int *best, float *best_states, float *best_action, hipStream_t s)
{
    const int threads = ((NA + 63) / 64) * 64;
    const size_t lds = sizeof(float) * ((size_t)p.WN * p.WN + 32);
    const bool win = p.WN > 0;
#define BN_DWA_LAUNCH(GEO_)                                                                                          \
    do {                                                                                                             \
        if (win) { hipError_t e = ensure_lds(dwa_kernel<GEO_, true>, lds); if (e != hipSuccess) return e;           \
                   dwa_kernel<GEO_, true><<<dim3(p.B), dim3(threads), lds, s>>>(p, actions, stage_goal, NA, Xall, cost, w, best, best_states, best_action); } \
        else { dwa_kernel<GEO_, false><<<dim3(p.B), dim3(threads), lds, s>>>(p, actions, stage_goal, NA, Xall, cost, w, best, best_states, best_action); }   \
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

__global__ void quotient_check_kernel(float res, float inv_res, uint32_t last, unsigned long long *bad)
{
    SolveParams p{};
    p.res = res; p.inv_res = inv_res;
    unsigned long long mine = 0;
    for (uint64_t u = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u <= last; u += (uint64_t)gridDim.x * blockDim.x) {
        const float d = __uint_as_float((uint32_t)u);
        const v2f q = quotient_general(p, v2f{d, d});
        const float t = d / res;
        if (floorf(q.x) != floorf(t) || (d >= 1.0e-30f && q.x != t) || q.y != q.x) ++mine;
    }
    if (mine) atomicAdd(bad, mine);
}

hipError_t launch_quotient_check(float res, float inv_res, float d_max, unsigned long long *bad, hipStream_t s)
{
    uint32_t last;
    std::memcpy(&last, &d_max, 4);                      // non-negative floats order like their bit patterns
    quotient_check_kernel<<<4096, 256, 0, s>>>(res, inv_res, last, bad);
    return hipGetLastError();
}

__global__ void stamp_kernel(int *word, int value)
{
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void echo64_kernel(const unsigned long long *src, unsigned long long *dst)
{
    __hip_atomic_store(dst, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_echo64(const unsigned long long *src, unsigned long long *dst, hipStream_t s)
{
    echo64_kernel<<<1, 1, 0, s>>>(src, dst);
    return hipGetLastError();
}

hipError_t launch_stamp(int *word, int value, hipStream_t s)
{
    stamp_kernel<<<1, 1, 0, s>>>(word, value);
    return hipGetLastError();
}

hipError_t launch_math_eval(int fn, const float *in, float *out, size_t n, hipStream_t s)
{
    math_eval_kernel<<<grid_for(n), 256, 0, s>>>(fn, in, out, n);
    return hipGetLastError();
}

// ---- can two streams DISPATCH concurrently? (bn_mppi_create: the extra stream of overlapped launches) ----
// HIP deals its streams onto a handful of hardware queues in creation order; two streams on ONE queue serialise (seen: a process that
// had brought up RCCL first, 15.2 instead of 9.6 us per dependent solve).  The probe asks for a little more than "not the same queue":
// a waiter grid larger than the chip holds (one workgroup per CU through its LDS request) whose workgroups stay until a flag is set,
// and a one-thread setter on the candidate stream.  Workgroup 0 reports whether it saw the flag before its (bounded) patience ran out:
// only if the setter could be dispatched while the waiter grid was still being placed -- which is what an overlapped launch of more
// workgroups than the chip holds (64 instances x 17) needs from its successor's queue.
__global__ void queue_probe_wait_kernel(int *flag, int *seen)
{
    if (threadIdx.x != 0) return;
    int ok = 0;
    for (int it = 0; it < 2000 && !ok; ++it) {          // ~0.75 ms at most
        ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ok) __builtin_amdgcn_s_sleep(14);
    }
    if (blockIdx.x == 0) *seen = ok;
}
__global__ void queue_probe_set_kernel(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

hipError_t launch_queue_probe(int *flag_and_seen, hipStream_t waiter, hipStream_t setter, int n_cus)
{
    const size_t lds = 96 * 1024;                       // one workgroup per CU
    hipError_t e = ensure_lds(queue_probe_wait_kernel, lds);
    if (e != hipSuccess) return e;
    queue_probe_wait_kernel<<<dim3((unsigned)(n_cus + n_cus / 2)), dim3(64), lds, waiter>>>(flag_and_seen, flag_and_seen + 1);
    queue_probe_set_kernel<<<1, 1, 0, setter>>>(flag_and_seen);
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

hipError_t launch_philox_noise(float *eps_kt2, uint64_t seed, uint64_t solve, int b, int K, int T, int k0, hipStream_t s)
{
    philox_noise_kernel<<<grid_for((size_t)K * ((T + 1) / 2)), 256, 0, s>>>(eps_kt2, seed, solve, b, K, T, k0);
    return hipGetLastError();
}

}  // namespace bn

