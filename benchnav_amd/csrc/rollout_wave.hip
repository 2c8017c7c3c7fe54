// rollout_wave.hip -- the one-wave-per-64-rollouts throughput variant of the rollout kernel.
#include "mppi_device.h"

namespace bn {

namespace {

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
// grid = rollout_grid, block = 64.  LDS: [ window | mean 2T | mean*inv_var 2T | e 64 | merge scratch ].
// ------------------------------------------------------------------------------
constexpr int kRegenCols = 8;            // columns (= 4 steps) of the control tile the noise-regenerating epilogue works on at a time

template <int EPS, int GEO, bool LDSWIN>
__global__ __launch_bounds__(64) void rollout_wave_kernel(const SolveParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WgId wg = decode_wg(p);
    if (wg.idle) return;
    if (wg.aux) {
        if (p.aux_prio) __builtin_amdgcn_s_setprio(3);
        if (p.overlap) finish_body<GEO, LDSWIN, 64, false, true>(p, wg.b, p.part_prev, p.cost_prev, p.state_prev, smem);
        else finish_body<GEO, LDSWIN, 64>(p, wg.b, p.part_prev, p.cost_prev, p.state_prev, smem);
        return;
    }
    const int T = p.T, K = p.K;
    float *win = smem;
    float *ml = win + (LDSWIN ? p.WN * p.WN : 0);
    float *mv = ml + 2 * T;
    float *el = mv + 2 * T;
    float *sc = el + 64;                              // merge scratch: nblk scales + 32
    // REGEN (the library's own noise): the epilogue draws the noise again instead of reading the controls back from HBM.  The round
    // trip -- (T,2,Kp) floats out and back per instance, 105 MB + 105 MB at 256 instances -- made this kernel bandwidth-bound in
    // both modes (392 MB per launch against 229 MB algorithmic with the trajectory dump, 231 MB against 69 MB without); a Philox
    // block costs VALU slots.  Measured, 256 instances, overlapped launches: 71.7 -> 64.3 us per launch with the trajectory dump
    // (3.57 -> 3.98 M solves/s; traffic 392 -> ~230 MB), but 49.2 -> 55.4 us in lean mode, where the trajectory stores are not there to
    // compete for the bandwidth and the launch becomes VALU-bound -- so lean launches keep the round trip, and so does injected noise
    // (reading eps again is the same bytes).
    const bool REGEN = EPS == kEpsPhilox && p.lean == 0 && !(BN_VAR_SKIP & 32);
    float *ut = sc + p.nblk + 32;                     // REGEN: control tile of one chunk, kRegenCols columns x kUPad
    const int lane = threadIdx.x, b = wg.b;
    const int k = wg.blk * kRolloutsPerBlock + lane;
    const bool active = k < K;
    const int kk = active ? k : K - 1;
    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float sx = p.state[b * 3 + 0], sy = p.state[b * 3 + 1], sth = p.state[b * 3 + 2];
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];

    const float *part_prev = p.part_prev + (size_t)b * p.nblk * (2 + 2 * T);
    // overlapped launch: the previous solve of this instance may still run in a launch on the other stream (rollout_role.inc)
    const bool ov = p.overlap && p.mean_from_part;
    MergeLoads pre{};
    const bool pre_ok = p.mean_from_part && p.nblk <= 64 && !ov;
    if (pre_ok) pre = merge_issue(part_prev, p.nblk, T, lane);
    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN) {
        w = window_origin<GEO>(p, sx, sy);
        stage_window(win, map, w, p.WN, p.G, lane, 64);
    }
    if (ov) {
        if (lane == 0) wait_counter<16>(flag_ctr(p.flag_part, p.prev_slot * p.B + b), p.wait_part, p.err);
        __syncthreads();
        float m_unused, S_unused;
        merge_partials<64, true>(part_prev, p.nblk, T, ml, sc, sc + p.nblk, lane, m_unused, S_unused, false, pre);
        for (int j = lane; j < 2 * T; j += 64) mv[j] = ml[j] * ((j & 1) ? p.iv1 : p.iv0);
    } else if (p.mean_from_part) {
        float m_unused, S_unused;
        merge_partials<64>(part_prev, p.nblk, T, ml, sc, sc + p.nblk, lane, m_unused, S_unused, pre_ok, pre);
        for (int j = lane; j < 2 * T; j += 64) mv[j] = ml[j] * ((j & 1) ? p.iv1 : p.iv0);
    } else {
        for (int j = lane; j < 2 * T; j += 64) {
            const float m = p.mean[(size_t)b * 2 * T + j];
            ml[j] = m;
            mv[j] = m * ((j & 1) ? p.iv1 : p.iv0);      // mean[t] @ inv_cov (diagonal), mppi.py:179-180
        }
    }
    const bool pub = p.flag_part != nullptr;           // member of an overlapped batch: what a successor reads goes out as device-scope stores
    if (wg.blk == 0 && lane == 0) {
        if (pub) { store_agent(p.state_copy + b * 3 + 0, sx); store_agent(p.state_copy + b * 3 + 1, sy); store_agent(p.state_copy + b * 3 + 2, sth); }
        else { p.state_copy[b * 3 + 0] = sx; p.state_copy[b * 3 + 1] = sy; p.state_copy[b * 3 + 2] = sth; }
    }
    __syncthreads();
    if (p.mean_snap && wg.blk == 0) snapshot_mean(p, b, ml, lane);
    const size_t Kp = (size_t)p.Kp;
    const bool lean = p.lean != 0;                    // lean mode: no trajectory batch (p.X is null)
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
        if (!lean) {                                                                                              \
            float *Xt = Xb + (size_t)(3 * (t)) * Kp;                                                              \
            Xt[0] = xn; Xt[Kp] = yn; Xt[2 * Kp] = tn;                                                             \
        }                                                                                                         \
        const float dx = xn - gx, dy = yn - gy;                                                                   \
        Sd += (double)(sqrt_cr_normal(dx * dx + dy * dy) + (c.trav <= p.thr ? 1.0e4f : 0.0f));   /* objectives.py:47-53 */  \
        Ad += (double)(p.lambda_ * (mv[2 * (t)] * (u0) + mv[2 * (t) + 1] * (u1)));      /* mppi.py:178-182 */      \
    } while (0)
    for (int t = 0; t < T; t += 2) {
        float e[4];
        noise_pair<EPS>(p, b, kk, t, e);
        const float u0 = clampf(ml[2 * t] + p.sigma0 * e[0], p.umin0, p.umax0);          // mppi.py:152-157
        const float u1 = clampf(ml[2 * t + 1] + p.sigma1 * e[1], p.umin1, p.umax1);
        float *Ut = Ub + (size_t)(2 * t) * Kp;
        if (!REGEN || p.store_u) { Ut[0] = u0; Ut[Kp] = u1; }
        if (t == 0) BN_WAVE_STEP(true, t, u0, u1); else BN_WAVE_STEP(false, t, u0, u1);
        if (t + 1 < T) {
            const float v0 = clampf(ml[2 * t + 2] + p.sigma0 * e[2], p.umin0, p.umax0);
            const float v1 = clampf(ml[2 * t + 3] + p.sigma1 * e[3], p.umin1, p.umax1);
            if (!REGEN || p.store_u) { Ut[2 * Kp] = v0; Ut[3 * Kp] = v1; }
            BN_WAVE_STEP(false, t + 1, v0, v1);
        }
    }
#undef BN_WAVE_STEP
    if (!lean) {
        float *Xt = Xb + (size_t)(3 * T) * Kp;         // slot T: clamped / wrapped state
        Xt[0] = c.x; Xt[Kp] = c.y; Xt[2 * Kp] = c.th;
    }
    const float dxT = c.x - gx, dyT = c.y - gy;
    const float term = sqrt_cr_normal(dxT * dxT + dyT * dyT) + (c.trav <= p.thr ? 1.0e4f : 0.0f);      // mppi.py:184
    const float cost = ((float)Sd + term) + (float)Ad;                                          // mppi.py:186-190
    if (active) { if (pub) store_agent(p.cost + (size_t)b * K + k, cost); else p.cost[(size_t)b * K + k] = cost; }
    const float z = active ? (-cost) / p.lambda_ : -INFINITY;
    const float zmax = wave_max(z);
    const float e = active ? expf(z - zmax) : 0.0f;
    const float esum = wave_sum(e);
    el[lane] = e;
    __syncthreads();                                   // e in LDS; this wave's control stores visible to all its lanes
    float *part = p.part + ((size_t)b * p.nblk + wg.blk) * (2 + 2 * T);
    if (lane == 0) {
        if (pub) { store_agent(part, zmax); store_agent(part + 1, esum); }
        else { part[0] = zmax; part[1] = esum; }
    }
    if (REGEN) {
        // weighted control sums, a chunk of kRegenCols / 2 steps at a time: every lane draws its controls of the chunk again (same
        // Philox blocks, same clamp: the same bits), the chunk goes through a small LDS tile, and column_sums' own arithmetic --
        // quarters of 16 rollouts summed in rollout order with fma, (q0 + q1) + (q2 + q3) by two DPP quad permutes -- adds it up
        for (int t0 = 0; t0 < T; t0 += kRegenCols / 2) {
#pragma unroll
            for (int q = 0; q < kRegenCols / 4; ++q) {
                const int t = t0 + 2 * q;
                if (t < T) {
                    float e2[4];
                    noise_pair<EPS>(p, b, kk, t, e2);
                    ut[(4 * q + 0) * kUPad + lane] = clampf(ml[2 * t] + p.sigma0 * e2[0], p.umin0, p.umax0);
                    ut[(4 * q + 1) * kUPad + lane] = clampf(ml[2 * t + 1] + p.sigma1 * e2[1], p.umin1, p.umax1);
                    if (t + 1 < T) {
                        ut[(4 * q + 2) * kUPad + lane] = clampf(ml[2 * t + 2] + p.sigma0 * e2[2], p.umin0, p.umax0);
                        ut[(4 * q + 3) * kUPad + lane] = clampf(ml[2 * t + 3] + p.sigma1 * e2[3], p.umin1, p.umax1);
                    }
                }
            }
            __syncthreads();
            const int ncol = min(kRegenCols, 2 * (T - t0));
            if (pub) column_sums<64, true>(ut, el, ncol / 2, lane, part + 2 * t0);
            else column_sums<64, false>(ut, el, ncol / 2, lane, part + 2 * t0);
            __syncthreads();                           // the tile is free again
        }
        if (pub) publish_counter(flag_ctr(p.flag_part, p.cur_slot * p.B + b), lane);
        return;
    }
    // weighted control sums: lane = column j, whose 64 rollout values are one contiguous row of the (T,2,Kp) buffer
    const float *Urow0 = p.U + (size_t)b * T * 2 * Kp + (size_t)wg.blk * kRolloutsPerBlock;
    for (int j = lane; j < 2 * T; j += 64) {
        const float4 *row = reinterpret_cast<const float4 *>(Urow0 + (size_t)j * Kp);
        // the summation order of column_sums: quarters of 16 rollouts, (q0 + q1) + (q2 + q3).  A rolled loop over the
        // quarters on purpose: unrolled, the four independent sums are scheduled side by side and cost the kernel half its
        // occupancy (127-143 VGPRs instead of 80).
        float acc = 0.0f, pair = 0.0f;
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            float a_ = 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = row[4 * r + q4];
                a_ = __builtin_fmaf(el[16 * r + 4 * q4 + 0], v.x, a_);
                a_ = __builtin_fmaf(el[16 * r + 4 * q4 + 1], v.y, a_);
                a_ = __builtin_fmaf(el[16 * r + 4 * q4 + 2], v.z, a_);
                a_ = __builtin_fmaf(el[16 * r + 4 * q4 + 3], v.w, a_);
            }
            if (r & 1) { pair = pair + a_; acc = (r == 1) ? pair : acc + pair; }
            else pair = a_;
        }
        if (pub) store_agent(part + 2 + j, acc); else part[2 + j] = acc;
    }
    if (pub) publish_counter(flag_ctr(p.flag_part, p.cur_slot * p.B + b), lane);
}


template <int EPS, int GEO>
hipError_t launch_wave_g(const SolveParams &p, hipStream_t s)
{
    const dim3 grid = rollout_grid(p, p.have_prev != 0);
    const size_t lds = wave_lds_bytes(p);
    if (p.WN > 0) {
        hipError_t e = ensure_lds(rollout_wave_kernel<EPS, GEO, true>, lds);
        if (e != hipSuccess) return e;
        rollout_wave_kernel<EPS, GEO, true><<<grid, dim3(64), lds, s>>>(p);
    } else {
        hipError_t e = ensure_lds(rollout_wave_kernel<EPS, GEO, false>, lds);
        if (e != hipSuccess) return e;
        rollout_wave_kernel<EPS, GEO, false><<<grid, dim3(64), lds, s>>>(p);
    }
    return hipGetLastError();
}

template <int EPS>
hipError_t launch_wave_e(const SolveParams &p, hipStream_t s)
{
    switch (geo_of(p)) {
    case kGeoPow2Origin0: return launch_wave_g<EPS, kGeoPow2Origin0>(p, s);
    case kGeoPow2: return launch_wave_g<EPS, kGeoPow2>(p, s);
    default: return launch_wave_g<EPS, kGeoGeneral>(p, s);
    }
}

}  // namespace

hipError_t launch_rollout_wave(const SolveParams &p, EpsMode mode, hipStream_t s)
{
    switch (mode) {
    case kEpsPhilox: return launch_wave_e<kEpsPhilox>(p, s);
    case kEpsKT2: return launch_wave_e<kEpsKT2>(p, s);
    default: return launch_wave_e<kEpsT2K>(p, s);
    }
}

}  // namespace bn
