// rollout_sampled.hip -- the sampled-slip rollout kernels of BASELINE config 3 and their launchers.
#include "mppi_device.h"

namespace bn {

namespace {

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
// grid = rollout_grid, block = 512, lane = rollout.
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
    const WgId wg = decode_wg(p);
    if (wg.idle) return;
    const int tid = threadIdx.x, lane = tid & 63, b = wg.b;
    if (wg.aux) {
        // aux workgroup: weights, cost copy and X* of the previous solve (merged by its own last workgroup)
        if (p.overlap) finish_body<GEO, true, kSampledThreads, false, true>(p, b, nullptr, p.cost_prev, p.state_prev, smem);
        else finish_body<GEO, true, kSampledThreads>(p, b, nullptr, p.cost_prev, p.state_prev, smem);
        return;
    }
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k = wg.blk * 64 + lane;
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
        const size_t g = (size_t)min(w.wy0 + r, p.G - 1) * p.G + min(w.wx0 + c, p.G - 1);      // (guard row / column: window_origin_wide)
        win2[e] = make_float2(mu[g], sg[g]);
    }
    // Overlapped launch on the ticket path (round 3): the previous solve may still be running on the other stream; its last workgroup
    // merges the partials and counts that in.  The window and the slip draws do not need the mean: they go first, the wait sits
    // between them and the controls.
    const bool ovs = p.overlap != 0;
    const bool tpub = p.flag_part != nullptr;          // member of an overlapped batch: costs and start state as device-scope stores
    if (!ovs)
    for (int j = tid; j < 2 * T; j += kSampledThreads) {
        const float m = p.mean[(size_t)b * 2 * T + j];
        ml[j] = m;
        mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);
    }
    if (wg.blk == 0 && tid < 3) { if (tpub) store_agent(p.state_copy + b * 3 + tid, p.state[b * 3 + tid]); else p.state_copy[b * 3 + tid] = p.state[b * 3 + tid]; }
    if (p.state_snap && wg.blk == 0 && tid < 3) p.state_snap[b * 3 + tid] = p.state[b * 3 + tid];   // first launch of a journalled batch
    __syncthreads();

    // ---- phase 1: slip draws, then controls, of every step ----
    {
        const int nE = (T + 1) >> 1, nS = TP >> 1;
        for (int q = nE + wid; q < nE + nS; q += kSampledWaves) {
            {
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
        if (ovs) {
            if (tid == 0) wait_counter<16>(flag_ctr(p.flag_part, p.prev_slot * p.B + b), p.wait_part, p);
            __syncthreads();
            for (int j = tid; j < 2 * T; j += kSampledThreads) {
                const float m = load_agent(p.mean + (size_t)b * 2 * T + j);
                ml[j] = m;
                mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);
            }
            __syncthreads();
        }
        if (p.mean_snap && wg.blk == 0 && wid == 0) snapshot_mean(p, b, ml, lane);
        for (int q = wid; q < nE; q += kSampledWaves) produce_pair<EPS, STORE_U>(p, p.eps, b, kk, 2 * q, p.solve, ml, Ul, Ub, Kp, lane);
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
        Zc[t * 64 + lane] = sqrt_cr_normal(dx * dx + dy * dy) + (tc <= p.thr ? 1.0e4f : 0.0f);  // objectives.py:46-53
    }
    __syncthreads();
    BN_STAMP(3);

    // ---- phase 4: rollout cost and the workgroup's softmin statistics ----
    if (wid == 0) {
        double Sd = 0.0;
#pragma unroll 16
        for (int t = 0; t < T; ++t) Sd += (double)Zc[t * 64 + lane];          // reads batched: one LDS round trip per 16 steps
        const float cost = ((float)Sd + Zc[T * 64 + lane]) + ad[lane];                 // mppi.py:184-190
        if (active) { if (tpub) store_agent(p.cost + (size_t)b * K + k, cost); else p.cost[(size_t)b * K + k] = cost; }
        const float zz = active ? (-cost) / p.lambda_ : -INFINITY;
        const float zmax = wave_max(zz);
        const float e = active ? expf(zz - zmax) : 0.0f;
        const float esum = wave_sum(e);
        el[lane] = e;
        if (lane == 0) {
            float *part = p.part + ((size_t)b * p.nblk + wg.blk) * (2 + 2 * T);
            store_agent(part, zmax); store_agent(part + 1, esum);
        }
    }
    __syncthreads();
    column_sums<kSampledThreads, true>(Ul, el, T, tid, p.part + ((size_t)b * p.nblk + wg.blk) * (2 + 2 * T));
    BN_STAMP(5);
    if (p.ustar_cur) ticket_merge<kSampledThreads>(p, b, wg.blk, smem);   // one-launch mode; the slot rows are dead: their LDS is the merge scratch
}

// The same solve without the LDS window (BN_FLAG_NO_LDS_WINDOW, or a window/horizon too large for the LDS):
// one wave per 64 rollouts, lookups from global memory, draws made in line.
template <int EPS, int GEO, bool STORE_U>
__global__ __launch_bounds__(64) void rollout_sampled_global_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.T, K = p.K;
    float *ml = smem, *mv = ml + 2 * T, *Ul = mv + 2 * T, *el = Ul + 2 * T * kUPad;
    const WgId wg = decode_wg(p);
    if (wg.idle) return;
    const int lane = threadIdx.x, b = wg.b;
    const int k = wg.blk * 64 + lane;
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
    if (wg.blk == 0 && lane < 3) p.state_copy[b * 3 + lane] = p.state[b * 3 + lane];
    if (p.state_snap && wg.blk == 0 && lane < 3) p.state_snap[b * 3 + lane] = p.state[b * 3 + lane];
    __syncthreads();
    const size_t Kp = (size_t)p.Kp;
    float *Xb = p.X + (size_t)b * (T + 1) * 3 * Kp + k;
    float *Ub = STORE_U ? p.U + (size_t)b * T * 2 * Kp + k : nullptr;
    for (int t = 0; t < T; t += 2) produce_pair<EPS, STORE_U>(p, p.eps, b, kk, t, p.solve, ml, Ul, Ub, Kp, lane);
    __syncthreads();
    float x = sx, y = sy, th = sth;
    float sn, cs;
    sincos_spec(th, sn, cs);
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
        float xn, yn, tn;
        if (p.ref_order) {                                                                              // BN_FLAG_REFERENCE_ORDER (chain_step<..., REF>)
            sincos_spec(th, sn, cs);
            const float tv = trav * u0;
            xn = x + (tv * cs) * p.dt; yn = y + (tv * sn) * p.dt;
            tn = th + (trav * u1) * p.dt;
            th = wrap_angle(tn);
        } else {
            const float g = u0 * p.dt, dth = trav * (u1 * p.dt);                                       // the transit arithmetic of chain_step
            xn = __builtin_fmaf(trav, g * cs, x); yn = __builtin_fmaf(trav, g * sn, y);
            tn = theta_step(th, dth, t == 0);
            rotate_spec(cs, sn, dth);                                                                   // carried heading vector
        }
        float *Xt = Xb + (size_t)(3 * t) * Kp;
        Xt[0] = xn; Xt[Kp] = yn; Xt[2 * Kp] = tn;
        x = clampf(xn, p.x0, p.x_hi); y = clampf(yn, p.y0, p.y_hi);
        // stage cost on the aliased slot: its own, independent slip draw (objectives.py:50)
        const int ec = slip_cell_safe<GEO, false>(p, w, xn, yn);
        const float tc = trav_from_slip(mu[ec], sg[ec], zq[2 + (t & 1)]);
        const float dx = xn - gx, dy = yn - gy;
        Sd += (double)(sqrt_cr_normal(dx * dx + dy * dy) + (tc <= p.thr ? 1.0e4f : 0.0f));
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
    const float term = sqrt_cr_normal(dxT * dxT + dyT * dyT) + (tT <= p.thr ? 1.0e4f : 0.0f);
    const float cost = ((float)Sd + term) + (float)Ad;
    if (active) p.cost[(size_t)b * K + k] = cost;
    const float zz = active ? (-cost) / p.lambda_ : -INFINITY;
    const float zmax = wave_max(zz);
    const float e = active ? expf(zz - zmax) : 0.0f;
    const float esum = wave_sum(e);
    el[lane] = e;
    __syncthreads();
    float *part = p.part + ((size_t)b * p.nblk + wg.blk) * (2 + 2 * T);
    column_sums<64, false>(Ul, el, T, lane, part);
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


}  // namespace

size_t sampled_resident_per_cu(const SolveParams &p)
{
    // workgroups of the fused sampled-slip kernel one CU holds at once: LDS (its slot rows and draw tiles) and 32 wave slots
    const size_t need = sizeof(float) * sampled_lds_floats(p.T, p.WN);
    return std::max<size_t>(1, std::min<size_t>(160 * 1024 / std::max<size_t>(need, 1), 32 / kSampledWaves));
}

bool sampled_fused(const SolveParams &p)
{
    // the multi-wave kernel needs the LDS window and room for its tiles; its LDS also holds the aux tail / merge scratch
    const size_t need = sizeof(float) * sampled_lds_floats(p.T, p.WN);
    const size_t tail = finish_lds_bytes(p) + sizeof(float) * 64;
    return p.slip_on && !p.ref_order && p.WN > 0 && need <= 160 * 1024 && tail <= need && p.nblk <= 1024;
}


namespace {

template <int EPS, int GEO>
hipError_t launch_sampled_g(const SolveParams &p, hipStream_t s)
{
    const size_t lds_w = sizeof(float) * sampled_lds_floats(p.T, p.WN);
    const dim3 grid = rollout_grid(p, sampled_fused(p) && p.have_prev);
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


}  // namespace

hipError_t launch_rollout_sampled(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    switch (mode) {
    case kEpsPhilox: return launch_sampled_e<kEpsPhilox>(p, s);
    case kEpsKT2: return launch_sampled_e<kEpsKT2>(p, s);
    default: return launch_sampled_e<kEpsT2K>(p, s);
    }
}

hipError_t launch_philox_slip(float *zt, float *zc, float *zo, uint64_t seed, uint64_t solve, int b, int K, int T, hipStream_t s)
{
    philox_slip_kernel<<<grid_for((size_t)K * (T / 2 + 1)), 256, 0, s>>>(zt, zc, zo, seed, solve, b, K, T);
    return hipGetLastError();
}


}  // namespace bn
