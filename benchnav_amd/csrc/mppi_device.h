// mppi_device.h -- device code shared by the kernel translation units of the MPPI planner: geometry and lookups,
// the transit step, the environment step, softmin merges, the tail of a solve, the ticket merge, control production.
// Everything here has internal linkage (anonymous namespace); each .hip file instantiates what it launches.
#pragma once
#include "mppi_kernels.h"
#include "bn_device_math.h"

#include <math.h>
#include <algorithm>

namespace bn {

namespace {


constexpr int TU = kChunk;   // time steps per phase (chunk): chain works on chunk c, consumers on c-1, producers on c+2

// In-kernel time stamps and per-workgroup traces (BN_STAMP, BN_WSTAMP, BN_TRACE_*, BN_TIMING_DO(statements)): compiled in by the measurement builds only
// (-DBN_EXPERIMENTS -DBN_TIMING, tools/stamps*.py, tools/block_trace*.py); csrc/experiments.h holds them, the shipped library sees
// empty statements.  Round 5 took everything else of the lab bench out of the kernels: the ablation and critical-path switches
// (BN_ABLATE, BN_VAR_*: compile-time bit tests inside the arithmetic) and the alternative cuts they selected are gone from the
// sources -- the measurements they produced are in DESIGN_NOTEBOOK.md, the code that produced them in the history (round 4's head).
#ifdef BN_EXPERIMENTS
#include "experiments.h"
#else
#define BN_STAMP(slot) do { } while (0)
#define BN_STAMP_ANY(slot) do { } while (0)
#define BN_WSTAMP(i) do { } while (0)
#define BN_TRACE_BEGIN() do { } while (0)
#define BN_TRACE_END() do { } while (0)
#define BN_TIMING_DO(...)
#define BN_SSTAMP(i) do { } while (0)
#endif

// Geometry specialisations of the cell index ((p - origin) / res).floor().int()  (grid_map.py:195-209):
//   kGeoGeneral  true division            kGeoPow2  res is a power of two: * (1/res) is bit-identical
//   kGeoPow2Origin0  additionally origin == 0, so the subtraction is the identity
enum Geo : int { kGeoGeneral = 0, kGeoPow2 = 1, kGeoPow2Origin0 = 2 };

struct Win { int wx0, wy0; float fx0, fy0, fwn, fwm1; int wn;
             // the latency kernel's chain wave: constants it keeps in vector registers of its own -- left to the compiler
             // they are re-materialised from scalar registers in every chunk
             float xhi_v, yhi_v; v2f c0_v; };   // window origin (cells), as floats, edge, edge-1, edge

// Which workgroup of which instance this is (see rollout_grid): rollout workgroup `blk` of instance `b`, or the aux
// workgroup of instance `b` (tail of the previous solve).  Rows past the instances hold the aux workgroups, so they are
// dispatched last; with xs = 3 the workgroups of one instance share an XCD (and its L2: window rows, partials).
struct WgId { int b, blk; bool aux, idle, self; };   // self: the aux workgroup that writes the tail of THIS launch's solve (SolveParams::self_tail)

__device__ __forceinline__ WgId decode_wg(const SolveParams &p)
{
    const int rows = (p.B + (1 << p.xs) - 1) >> p.xs;
    WgId r;
    r.self = false;
    if ((int)blockIdx.y >= rows) {
        r.aux = true;
        r.blk = p.nblk;
        r.b = ((int)blockIdx.y - rows) * (int)gridDim.x + (int)blockIdx.x;
        if (p.self_tail && r.b >= (p.have_prev ? p.B : 0)) {            // behind the previous solve's tails: this solve's own
            r.self = true;
            r.b -= p.have_prev ? p.B : 0;
        }
    } else {
        r.aux = false;
        r.blk = (int)blockIdx.x >> p.xs;
        r.b = ((int)blockIdx.y << p.xs) + ((int)blockIdx.x & ((1 << p.xs) - 1));
    }
    r.idle = r.b >= p.B;
    return r;
}

template <int GEO>
__device__ __forceinline__ int raw_cell(float v, float origin, float res, float inv_res)
{
    const float q = (GEO == kGeoPow2Origin0) ? v * inv_res : (GEO == kGeoPow2) ? (v - origin) * inv_res : (v - origin) / res;
    return (int)floorf(q);                    // v_cvt_i32_f32 saturates
}

// (p - origin) / res for a resolution that is not a power of two, as the reference rounds it (a true division, grid_map.py:203), for
// the in-loop lookups: d >= 0 is a clamped position minus the lower limit.  With r = RN(1 / res) (the host's correctly rounded
// reciprocal) q0 = d r, e = fma(-q0, res, d) (the exact residual), q = fma(e, r, q0) is the correctly rounded quotient (Markstein's
// correction step): three packed instructions where the IEEE expansion of two divisions is twenty-two.  No branch back to the
// division here (it cost the chain 1.4 us per solve, tools/res_rate.py): bn_mppi_create checks the form EXHAUSTIVELY on the device
// -- every float d in [0, upper limit - lower limit]: same floor as d / res, same bits above the denormal range -- and refuses a
// resolution that fails (none known: a host sweep over resolutions incl. all-ones significands, the theorem's exception, found no
// mismatch above 1e-30).  K=1024, T=50 at res 0.3: 13.1 (division) -> 10.0 us per solve; res 0.5 (exact multiply): 8.7.
__device__ __forceinline__ v2f quotient_general(const SolveParams &p, v2f d)
{
    const v2f r = {p.inv_res, p.inv_res}, b = {p.res, p.res};
    const v2f q0 = d * r;
    const v2f e = __builtin_elementwise_fma(-q0, b, d);
    return __builtin_elementwise_fma(e, r, q0);
}

// Window of edge wn covering `reach` cells either side of the cell of (sx, sy) and one more above, shifted into the map PLUS ONE
// GUARD ROW / COLUMN at index G (staged as a copy of row / column G-1): a position on the upper map limit has the raw cell G, and
// the reference's index clamp (grid_map.py:209) is then built into the window instead of costing the chain two clamps a step.
template <int GEO>
__device__ __forceinline__ Win window_origin_wide(const SolveParams &p, float sx, float sy, int reach, int wn)
{
    const int cx = clampi(raw_cell<GEO>(sx, p.x0, p.res, p.inv_res), 0, p.G - 1);
    const int cy = clampi(raw_cell<GEO>(sy, p.y0, p.res, p.inv_res), 0, p.G - 1);
    Win w;
    w.wx0 = min(max(cx - reach, 0), p.G + 1 - wn);
    w.wy0 = min(max(cy - reach, 0), p.G + 1 - wn);
    w.fx0 = (float)w.wx0; w.fy0 = (float)w.wy0; w.fwn = (float)wn; w.fwm1 = (float)(wn - 1); w.wn = wn;
    return w;
}

template <int GEO>
__device__ __forceinline__ Win window_origin(const SolveParams &p, float sx, float sy)
{
    return window_origin_wide<GEO>(p, sx, sy, p.reach, p.WN);
}

// Stage the reachable window as traversability: trav = 1 - clamp(risk, 0, 1)
// (reference traversability_model.py:72).  Rows of the window are contiguous
// runs of the map rows, so the loads coalesce per row.
__device__ __forceinline__ void stage_window(float *win, const float *__restrict__ map, const Win w,
                                             int WN, int G, int tid, int nthreads)
{
    // Four cells per thread and round trip: written as one cell per iteration the loop is load - wait - store, a full memory round trip
    // per 320 cells, in the prologue of every launch and of every tail (round 5: the headline's 24 x 24 window took two).
    const int n = WN * WN;
    for (int e0 = tid; e0 < n; e0 += 4 * nthreads) {
        float risk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = min(e0 + i * nthreads, n - 1);       // (behind the window: the last cell again, not stored)
            const int r = e / WN;
            const int c = e - r * WN;
            risk[i] = map[(size_t)min(w.wy0 + r, G - 1) * G + min(w.wx0 + c, G - 1)];      // (the guard row / column: index G -> G-1)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i * nthreads;
            if (e < n) win[e] = 1.0f - clampf(risk[i], 0.0f, 1.0f);
        }
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
    int ix, iy;
    if (GEO == kGeoGeneral && !SAFE) {                 // in-loop lookup of a clamped position: the validated three-instruction quotient
        const v2f q = quotient_general(p, v2f{x, y} - v2f{p.x0, p.y0});
        ix = (int)floorf(q.x); iy = (int)floorf(q.y);
    } else {
        ix = raw_cell<GEO>(x, p.x0, p.res, p.inv_res);
        iy = raw_cell<GEO>(y, p.y0, p.res, p.inv_res);
    }
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

// In-loop gather: (x, y) already lies inside the map limits and within `reach` cells of the start, so its raw cell lies inside
// the window (guard row / column included: window_origin_wide) and nothing has to be clamped.  q = (x - origin)/res as the
// reference rounds it, then q - wx0 (exact: an integer no larger than q is subtracted), floor, row * WN + col in the float domain
// (exact small integers), one conversion.  Same cell as trav_lookup<..., false> for every such position; six instructions before
// the ds_read where floor, clamp, floor, clamp, fma, convert, shift-add were eight -- on the chain wave they sit on the solve's
// critical path T times.
template <int GEO, int ASMIDX = 0>
__device__ __forceinline__ float trav_window(const SolveParams &p, const float *win, const Win w, float x, float y)
{
    v2f q;
    const v2f xy = {x, y}, ir = {p.inv_res, p.inv_res}, nw = {-w.fx0, -w.fy0};
    if (GEO == kGeoPow2Origin0) {
        q = __builtin_elementwise_fma(xy, ir, nw);                      // one v_pk_fma_f32
    } else if (GEO == kGeoPow2) {
        q = __builtin_elementwise_fma(xy - v2f{p.x0, p.y0}, ir, nw);
    } else {
        q = quotient_general(p, xy - v2f{p.x0, p.y0}) + nw;
    }
    if (ASMIDX == 2 && GEO != kGeoGeneral) {
        // The latency kernel's chain wave (ASMIDX = 2; the tail's X* rollout keeps ASMIDX = 1 below: fixed registers at the top of
        // a 128-register budget would cost the role kernel, whose aux workgroup runs that tail, its occupancy).  The same five instructions -- quotient (packed FMA), floor-and-convert x 2, row * WN + col, byte address -- as ONE asm
        // block whose only result is consumed by the LDS read.  Why: LLVM's hazard recogniser (gfx940+ "dst_sel forwarding") puts
        // an s_nop between (a) any VALU consumer and an inline asm that defines its operand and (b) an inline asm and a packed
        // instruction with op_sel_hi[0] = 1 that defines one of its operands (the bit shares its position with VOP3's dst op_sel:
        // a false positive for VOP3P).  The old cut -- packed FMA | asm(cvt, cvt, mad) | shift-add -- paid both, two issue slots
        // of the chain wave per step.  Inside one block there is no hazard to assume, and a DS consumer is not a VALU consumer.
        // v126 / v127: scratch (the halves of a 64-bit asm operand cannot be named, so the quotient lives in fixed registers).
        typedef __attribute__((address_space(3))) const float lds_cf;
        const unsigned base = (unsigned)(uintptr_t)(lds_cf *)win;
        const v2f xyo = GEO == kGeoPow2Origin0 ? xy : xy - v2f{p.x0, p.y0};
        unsigned addr;
        asm("v_pk_fma_f32 v[126:127], %1, %2, %3\n\t"
            "v_cvt_flr_i32_f32 v126, v126\n\t"
            "v_cvt_flr_i32_f32 v127, v127\n\t"
            "v_mad_u32_u24 v126, v127, %4, v126\n\t"
            "v_lshl_add_u32 %0, v126, 2, %5"
            : "=v"(addr) : "v"(xyo), "s"(ir), "v"(nw), "s"(w.wn), "s"(base) : "v126", "v127");
        return *(lds_cf *)(uintptr_t)addr;
    }
    if (ASMIDX == 1 || ASMIDX == 3) {
        // The latency kernel's chain wave: floor-and-convert in one instruction, row * WN + col as one v_mad_u32_u24 (left to itself
        // the compiler spreads the *4 of the byte address over both terms).  Same integers.  8.65 -> 8.57 us per dependent solve;
        // NOT for the role kernel (64 instances: +3 %): the compiler pads both ends of an asm block with a wait state and cannot
        // schedule across it.
        int cell, lj;
        asm("v_cvt_flr_i32_f32 %0, %2\n\tv_cvt_flr_i32_f32 %1, %3\n\tv_mad_u32_u24 %0, %1, %4, %0"
            : "=&v"(cell), "=&v"(lj) : "v"(q.x), "v"(q.y), "s"(w.wn));
        return win[cell];
    }
    return win[(int)__builtin_fmaf(floorf(q.y), w.fwn, floorf(q.x))];
}

// Per-rollout recurrence state: the clamped/wrapped state t, its traversability, sin/cos of its heading.
// G, wq: what the next step can know before the traversability of its start cell arrives (chain_prepare).
struct Chain { float x, y, th, sn, cs, trav; v2f G; float wq; };

// The heading of a rollout as an OUTPUT: theta_{t+1} = wrap(theta_t + d_t) (robot_model.py:88,90), tn = the un-wrapped sum slot t
// keeps.  The first step takes the general wrap (a caller's start heading may be anything), later ones the near form.
__device__ __forceinline__ float theta_step(float &th, float dth, bool first)
{
    const float tn = th + dth;                                         // :88
    th = first ? wrap_angle(tn) : wrap_angle_near(tn);                 // :90
    return tn;
}

// One UnicycleModel.transit (robot_model.py:59-100) plus the gather for the next step.
// (xn, yn, tn) is what the reference leaves in slot t (un-clamped, un-wrapped); the chain
// advances to the clamped/wrapped state t+1.  The heading enters the positions only through (cos, sin), which are CARRIED by a
// rotation (rotate_spec); theta is integrated beside it when THETA is set (kernels whose one wave does everything).  With
// THETA = false (the role kernels' chain wave) tn returns the step's heading increment d_t instead and c.th is not touched:
// the wave that stores the trajectory integrates theta from the increments (theta_step).
// u0, u1 already lie in [u_min, u_max]: the re-clamp of robot_model.py:82-83 is the identity.
// Arithmetic of one transit (DESIGN.md "Arithmetic spec"): with g = v dt and w = omega dt (the controls, pre-scaled),
//   x~ = fma(trav, g cos, x)   y~ = fma(trav, g sin, y)   d = trav w   theta~ = theta + d
// -- the reference's ((trav v) cos) dt + x with one rounding fewer and everything but `trav` folded into factors that are ready
// before the gather returns: ONE dependent instruction between the traversability and the new position instead of four.
__device__ __forceinline__ void chain_prepare(const SolveParams &p, Chain &c, float u0, float u1)
{
    const float g = u0 * p.dt;
    c.wq = u1 * p.dt;
    c.G = v2f{g, g} * v2f{c.cs, c.sn};
}

// PREP: the caller's loop hands over the NEXT step's controls (u0n, u1n) and chain_prepare runs for them at the end of this step,
// under the gather's latency (the role kernels' chain wave); otherwise the step prepares itself from its own controls first.
// REF (BN_FLAG_REFERENCE_ORDER): the transit as the reference writes it -- sin / cos of THIS step's heading (sincos_spec), then
// x + ((trav v) cos) dt, y + ((trav v) sin) dt, theta + (trav omega) dt in that operation order (robot_model.py:86-88), the general
// heading wrap at every step (no bound on dt |omega|).  c.sn / c.cs / c.G / c.wq are not used.  The oracle's trig = 2.
template <int GEO, bool LDSWIN, bool FIRST, bool THETA = true, bool PREP = false, int ASMIDX = 0, bool REF = false>
__device__ __forceinline__ void chain_step(const SolveParams &p, const float *win, const float *__restrict__ map,
                                           const Win w, Chain &c, float u0, float u1, float &xn, float &yn, float &tn,
                                           float u0n = 0.0f, float u1n = 0.0f)
{
    if constexpr (REF) {
        static_assert(THETA && !PREP, "the reference-order step integrates its own heading");
        // (c.sn, c.cs) = sincos_spec(c.th) on entry and on exit: every caller sets them with the start heading, and the step ends with
        // the sine and cosine of the NEXT heading, issued behind the next gather.  After the first step the heading wrap takes the
        // branch-free near form where it is valid (p.wrap_near: dt max|omega| < 3; bit-identical to the general form there,
        // tests/test_gpu_device_math.py).  Same operations per value as before (round 5) -- and the same 11.7 us per
        // dependent single-instance solve: the chain wave is alone on its SIMD and issues one instruction per ~5 cycles whatever
        // they wait for, so its 46 instructions per step ARE the step (the default arithmetic has 25); the gather's latency was
        // already covered.  The price of this arithmetic on the latency path stays 1.36x; `value_reference_order` reports it.
        const float tv = c.trav * u0;
        xn = c.x + (tv * c.cs) * p.dt;                                 // :86
        yn = c.y + (tv * c.sn) * p.dt;                                 // :87
        tn = c.th + (c.trav * u1) * p.dt;                              // :88
        c.x = clampf(xn, p.x0, ASMIDX >= 2 ? w.xhi_v : p.x_hi);        // :93
        c.y = clampf(yn, p.y0, ASMIDX >= 2 ? w.yhi_v : p.y_hi);        // :94
        c.trav = LDSWIN ? trav_window<GEO, ASMIDX>(p, win, w, c.x, c.y) : trav_lookup<GEO, false, false>(p, win, map, w, c.x, c.y);
        __builtin_amdgcn_sched_barrier(0);
        c.th = (!FIRST && p.wrap_near) ? wrap_angle_near(tn) : wrap_angle(tn);      // :90
        sincos_spec(c.th, c.sn, c.cs);
        return;
    }
    if (!PREP) chain_prepare(p, c, u0, u1);
    // Position strand first: update, clamp, cell index, and the gather goes out; the heading strand (rotation, 7
    // instructions) then runs under the gather's LDS latency.  The scheduling barrier keeps the compiler from
    // interleaving the two again (it used to issue the gather two thirds into the step).
    const float dth = c.trav * c.wq;                                   // :88
    const v2f pos = __builtin_elementwise_fma(v2f{c.trav, c.trav}, c.G, v2f{c.x, c.y});   // :86-87, x and y in lockstep
    xn = pos.x;
    yn = pos.y;
    c.x = clampf(xn, p.x0, ASMIDX >= 2 ? w.xhi_v : p.x_hi);            // :93   (ASMIDX 2, 3: the upper limits in vector registers the caller keeps)
    c.y = clampf(yn, p.y0, ASMIDX >= 2 ? w.yhi_v : p.y_hi);            // :94
    c.trav = LDSWIN ? trav_window<GEO, ASMIDX>(p, win, w, c.x, c.y) : trav_lookup<GEO, false, false>(p, win, map, w, c.x, c.y);
    __builtin_amdgcn_sched_barrier(0);
    if (THETA) tn = theta_step(c.th, dth, FIRST); else tn = dth;
    rotate_spec<ASMIDX == 2>(c.cs, c.sn, dth, w.c0_v);
    if (PREP) chain_prepare(p, c, u0n, u1n);
}

// One PlanetaryEnv.step (planetary_env.py:189-219) for instance b: observation-mode transit with the
// latent slip sampled at the current cell (traversability_model.py:65-69: Normal(mean, std)[cell].sample()
// = z * std + mean), then the goal test.  Like the reference, a terminated environment keeps moving when stepped again;
// with p.env_freeze (opt-in, for fixed-length batched episodes) an instance already within goal_thr of its goal stays put.
// Every workgroup that needs the next state evaluates this itself: same inputs, same operations.
struct EnvStep { float x, y, th, reward; bool reached, frozen; };

// The latent slip model at the cell of (sx, sy): the two loads of env_advance, issued on their own so that a caller who knows the
// state early can have them in flight while it waits for the control.
struct EnvCell { float mean, std; };

template <int GEO>
__device__ __forceinline__ EnvCell env_fetch(const SolveParams &p, int b, float sx, float sy)
{
    const int ix = clampi(raw_cell<GEO>(sx, p.x0, p.res, p.inv_res), 0, p.G - 1);
    const int iy = clampi(raw_cell<GEO>(sy, p.y0, p.res, p.inv_res), 0, p.G - 1);
    const size_t cell = (size_t)b * p.map_stride + (size_t)iy * p.G + ix;
    return EnvCell{p.lat_mean[cell], p.lat_std[cell]};
}

template <int GEO>
__device__ __forceinline__ EnvStep env_advance_with(const SolveParams &p, int b, float sx, float sy, float sth, float u0, float u1,
                                                    const float *z_ptr, uint64_t step, const EnvCell lc)
{
    const float gx = p.goal[b * 2 + 0], gy = p.goal[b * 2 + 1];
    EnvStep r;
    const float d0x = sx - gx, d0y = sy - gy;
    r.frozen = p.env_freeze && sqrt_cr(d0x * d0x + d0y * d0y) < p.goal_thr;      // terminated at an earlier step
    float z;
    if (z_ptr) {
        z = z_ptr[b];
    } else {
        const u32x4 q = philox4x32_10(u32x4{(uint32_t)b, (uint32_t)step, (uint32_t)(step >> 32), 0x454e5631u},
                                      (uint32_t)p.env_seed, (uint32_t)(p.env_seed >> 32));
        float z1;
        box_muller(q.x, q.y, z, z1);
    }
    const float slip = z * lc.std + lc.mean;
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

template <int GEO>
__device__ __forceinline__ EnvStep env_advance(const SolveParams &p, int b, float sx, float sy, float sth, float u0, float u1,
                                               const float *z_ptr, uint64_t step)
{
    return env_advance_with<GEO>(p, b, sx, sy, sth, u0, u1, z_ptr, step, env_fetch<GEO>(p, b, sx, sy));
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
    else q = quotient_general(p, xy - v2f{p.x0, p.y0}) + nw;
    const float li = clampf(floorf(q.x), 0.0f, w.fwm1);
    const float lj = clampf(floorf(q.y), 0.0f, w.fwm1);
    return (int)__builtin_fmaf(lj, w.fwn, li);
}

// One observation-mode transit (robot_model.py:59-100) of the sampled-slip chain on the LDS window of (mean, std)
// pairs: state (x, y, th) with heading (sn, cs) and window cell e advances; (xn, yn, tn) is what slot t keeps.
struct SlipChain { float x, y, th, sn, cs; int e; };

template <int GEO, bool FIRST, bool REF = false>
__device__ __forceinline__ void slip_chain_step(const SolveParams &p, const float2 *win2, const Win w, SlipChain &c, float u0,
                                                float u1, float z, float &xn, float &yn, float &tn)
{
    const float2 ms = win2[c.e];
    const float trav = trav_from_slip(ms.x, ms.y, z);                  // robot_model.py:75
    if constexpr (REF) {                                               // the reference's operation order (chain_step<..., REF>)
        float sn, cs;
        sincos_spec(c.th, sn, cs);
        const float tv = trav * u0;
        xn = c.x + (tv * cs) * p.dt;
        yn = c.y + (tv * sn) * p.dt;
        tn = c.th + (trav * u1) * p.dt;
        c.th = wrap_angle(tn);
        c.x = clampf(xn, p.x0, p.x_hi);
        c.y = clampf(yn, p.y0, p.y_hi);
        c.e = slip_cell_window<GEO>(p, w, c.x, c.y);
        return;
    }
    const float g = u0 * p.dt;                                         // the transit arithmetic of chain_step
    const float dth = trav * (u1 * p.dt);
    xn = __builtin_fmaf(trav, g * c.cs, c.x);
    yn = __builtin_fmaf(trav, g * c.sn, c.y);
    tn = theta_step(c.th, dth, FIRST);
    c.x = clampf(xn, p.x0, p.x_hi);
    c.y = clampf(yn, p.y0, p.y_hi);
    rotate_spec(c.cs, c.sn, dth);                                      // carried heading vector (bn_device_math.h)
    c.e = slip_cell_window<GEO>(p, w, c.x, c.y);
}

// Wave-wide reductions with DPP row shifts / row broadcasts: six VALU instructions instead of six ds_bpermute round trips
// through the LDS crossbar (~70 cycles each for a lone wave).  Written as inline assembly on purpose: expressed with
// __builtin_amdgcn_update_dpp, hipcc's DPP combiner mis-folded the update_dpp + add pairs inside the rollout kernel (wrong
// sums on hardware, correct in an isolated kernel).  The assembler adds no hazard padding inside an asm block, so the
// two wait states a DPP read needs after the VALU write of its source are spelled out (s_nop 1).
// Lanes whose DPP source does not exist are disabled (no bound_ctrl) and keep their accumulator.  After the six steps
// lane 63 holds the reduction of all 64 lanes; v_readlane hands it to everyone.  The order of the additions is fixed
// (inclusive scan within rows of 16, then across rows): every kernel and every caller gets the same bits.
#define BN_DPP_REDUCE(OP, v)                                                                                   \
    asm volatile("s_nop 1\n\t"                                                                                 \
                 OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                          \
                 OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                          \
                 OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                          \
                 OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"                          \
                 OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"                       \
                 OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"                            \
                 : "+v"(v))

__device__ __forceinline__ float wave_max(float v)
{
    BN_DPP_REDUCE("v_max_f32_dpp", v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum(float v)
{
    BN_DPP_REDUCE("v_add_f32_dpp", v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Device-scope accesses that bypass the per-XCD L2 (sc1): what lets workgroups on different XCDs exchange their
// partials inside one launch without a full L2 write-back / invalidate (see ticket_merge).
__device__ __forceinline__ void store_agent(float *ptr, float v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_agent(const float *ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool AGENT>
__device__ __forceinline__ float ld(const float *ptr) { return AGENT ? load_agent(ptr) : *ptr; }
template <bool AGENT>
__device__ __forceinline__ void st(float *ptr, float v) { if (AGENT) store_agent(ptr, v); else *ptr = v; }

// Overlapped launches: a counter another launch advances (device-scope atomics) reaches `need`.  ONE lane polls, with sc1 loads
// and s_sleep in between; the caller puts a workgroup barrier behind it.  Bounded: a wait that does not end within ~2 s sets
// *err and gives up -- a wrong result that the host sees at its next synchronising call (and repairs: recover_overlap in mppi_capi.cpp), never a hung GPU.
// SLEEP: s_sleep argument between polls (x 64 cycles).  1 where the wait is short and on the critical path (latency kernel); the
// role kernel's workgroups may hold a slot for many microseconds with hundreds of them polling at once -- at s_sleep 1 their
// loads saturate the memory channel the counters live in and everything else that touches it (measured: 40 instances, 31 us
// per launch instead of 21) -- so they poll every ~0.5 us, and the counters are kFlagStride apart (one channel each).
__device__ __forceinline__ unsigned long long *flag_ctr(unsigned long long *base, int i) { return base + (size_t)i * kFlagStride; }

// The error word lives in pinned host memory (SolveParams::err): a system-scope store, so that the host sees it without a copy.
// ... and in a device word (SolveParams::err_dev) that every later WRITER of the warm-start mean looks at first: a launch whose wait
// expired goes on with incomplete partials, and whatever is merged from them must not reach `mean` -- it is the one piece of state
// the re-run starts from (mean_snap is taken from it by the first launch of a stretch, which in exactly this situation may run
// LATE: round 4 saw NaN survive a repair that way).  The flag is raised before the workgroup publishes anything, so whoever sees
// its counts sees the flag.
__device__ __forceinline__ void raise_wait_expired(const SolveParams &p, const unsigned long long *ctr = nullptr, unsigned long long need = 0)
{
#ifdef BN_EXPERIMENTS                                  // which wait it was, for the host's message (tools/_repro*.py)
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        p.err[8] = ctr ? (int)((ctr - p.flag_part) / kFlagStride) : -1; p.err[9] = (int)need;
        p.err[10] = ctr ? (int)__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1; p.err[11] = (int)p.solve;
        p.err[12] = blockIdx.x; p.err[13] = blockIdx.y; p.err[14] = p.have_prev * 100 + p.overlap * 10 + p.cur_slot; p.err[15] = (int)p.wait_part * 1000 + (int)p.wait_tail;
    }
#endif
    __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p.err_dev) __hip_atomic_store(p.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool batch_spoiled(const SolveParams &p)
{
    return p.err_dev != nullptr && __hip_atomic_load(p.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// First launch of the batches enqueued since the host last checked the error word: workgroup 0 of every instance keeps the mean
// this solve samples around (ml, in LDS), so that the host can re-run those batches on one stream should a wait expire.
__device__ __forceinline__ void snapshot_mean(const SolveParams &p, int b, const float *ml, int lane)
{
    for (int j = lane; j < 2 * p.T; j += 64) p.mean_snap[(size_t)b * 2 * p.T + j] = ml[j];
}

template <int SLEEP = 1>
__device__ __forceinline__ void wait_counter(const unsigned long long *ctr, unsigned long long need, const SolveParams &p)
{
    for (int it = 0; it < (1 << 23) / SLEEP + 1024; ++it) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return;
        __builtin_amdgcn_s_sleep(SLEEP);
    }
    raise_wait_expired(p, ctr, need);
}

// Publish: every wave has seen its own sc1 stores acknowledged (vmcnt 0), the workgroup meets, one lane counts it in.
__device__ __forceinline__ void publish_counter(unsigned long long *ctr, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Granules: an 8-byte {value, tag} pair written by ONE device-scope store, so a reader that sees the tag sees the value.
// The partial rows an overlapped successor's prologue needs are published this way as well (tag = producing solve's
// index + 1): it can poll the rows themselves -- one memory round trip -- instead of a counter and then the rows.
__device__ __forceinline__ void store_granule(unsigned long long *ptr, float v, uint32_t tag)
{
    __hip_atomic_store(ptr, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... to pinned host memory (system scope): what the host polls in bn_mppi_first_action
__device__ __forceinline__ void store_granule_host(unsigned long long *ptr, float v, uint32_t tag)
{
    __hip_atomic_store(ptr, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float load_granule(const unsigned long long *ptr, uint32_t tag, bool &ok)
{
    const unsigned long long g = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = ok && (uint32_t)(g >> 32) == tag;
    return __uint_as_float((uint32_t)g);
}

// The workgroup's weighted control sums  sum_k e_k u_k[j]  (mppi.py:196-199, before the cross-workgroup merge) for the
// 2T columns j of the LDS control tile (pitch kUPad).  One definition of the summation order for every rollout kernel,
// so that all of them produce the same bits: the 64 rollouts in four quarters of 16, each summed in rollout order with
// fma, combined as (q0 + q1) + (q2 + q3).  Here four adjacent lanes take the quarters of one column (2T x 4 work items
// over NT threads; 64 sequential fma per column on a quarter of the threads was 0.55 us of a 13 us solve) and the
// combination is two DPP quad permutes.  NT and the item count are multiples of 4: a quad is active as a whole.
template <int NT, bool AGENT>
__device__ __forceinline__ void column_sums(const float *Ul, const float *el, int T, int tid, float *part,
                                            unsigned long long *grow = nullptr, uint32_t tag = 0)
{
    for (int it = tid; it < 8 * T; it += NT) {
        const int j = it >> 2, r = it & 3;
        const float *col = Ul + j * kUPad + 16 * r, *e = el + 16 * r;
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = __builtin_fmaf(e[q], col[q], acc);
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                     : "+v"(acc));
        if (r == 0) {
            if (grow) store_granule(grow + 2 + j, acc, tag);           // first: what the successor's prologue polls
            if (AGENT) store_agent(part + 2 + j, acc); else part[2 + j] = acc;
        }
    }
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

// Merge the nblk per-block statistics (max z, sum e, sum e*u) of one instance into
//   U*[j] = sum_k w_k u_k[j]      mppi.py:193-199
// written to us[0..2T) (LDS).  Deterministic: every caller (any thread count NT) gets bit-identical values,
// which is what lets each rollout block of the next solve recompute the warm-start mean on its own.
// LDS scratch: sc[nblk], red[4].  Returns (max z, sum exp) for the weights.
constexpr int kMergePrefetch = 16;
struct MergeLoads { float v[kMergePrefetch]; float mi, si; int j; };

// Which column of U* a thread of an NT-thread workgroup merges: its own index, rotated by c0 threads.  The latency kernel gives the
// columns to the waves that have a SIMD to themselves (c0 = 128: waves 2 and 3 take 2T <= 128 columns; waves 0 and 4 share a SIMD and
// were the last to reach the barrier behind the merge by 0.5 us, tools/stamps_overlap.py).  Who merges a column does not change its bits.
template <int NT>
__device__ __forceinline__ int merge_column(int tid, int c0) { return tid >= c0 ? tid - c0 : tid + NT - c0; }

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

// The same loads from the granule copy of the rows (nblk <= kMergePrefetch), each checked against the producing solve's tag:
// `ok` stays true only if every granule this thread needs carried it.
// `col`: the column this thread merges (merge_column: not necessarily its thread index).
__device__ __forceinline__ MergeLoads merge_issue_granules(const unsigned long long *__restrict__ grows, int nblk, int T, int tid, uint32_t tag, bool &ok, int col)
{
    const int PS = 2 + 2 * T;
    const int lane = tid & 63;
    MergeLoads L;
    L.j = col < 2 * T ? col : 0;
    ok = true;
#pragma unroll
    for (int i = 0; i < kMergePrefetch; ++i) L.v[i] = load_granule(grows + (size_t)min(i, nblk - 1) * PS + 2 + L.j, tag, ok);
    const bool has = lane < nblk;
    bool ok2 = true;
    const float mi = load_granule(grows + (size_t)min(lane, nblk - 1) * PS, tag, ok2);
    const float si = load_granule(grows + (size_t)min(lane, nblk - 1) * PS + 1, tag, ok2);
    ok = ok && ok2;
    L.mi = has ? mi : -INFINITY;
    L.si = has ? si : 0.0f;
    return L;
}

// Split-phase form of merge_issue_granules: issue() puts the 18 loads of one poll on the wire, take() looks at them (the compiler's
// vmcnt wait lands there, not at the issue).
struct GranulePoll {
    unsigned long long v[kMergePrefetch], mi, si;
    __device__ __forceinline__ void issue(const unsigned long long *__restrict__ grows, int nblk, int T, int lane, int col)
    {
        const int PS = 2 + 2 * T;
        const int j = col < 2 * T ? col : 0;
#pragma unroll
        for (int i = 0; i < kMergePrefetch; ++i)
            v[i] = __hip_atomic_load(grows + (size_t)min(i, nblk - 1) * PS + 2 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mi = __hip_atomic_load(grows + (size_t)min(lane, nblk - 1) * PS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        si = __hip_atomic_load(grows + (size_t)min(lane, nblk - 1) * PS + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // all of this lane's granules carry `tag`?  L: the values, laid out as merge_issue_granules leaves them
    __device__ __forceinline__ bool take(uint32_t tag, int nblk, int T, int lane, int col, MergeLoads &L) const
    {
        bool ok = (uint32_t)(mi >> 32) == tag && (uint32_t)(si >> 32) == tag;
#pragma unroll
        for (int i = 0; i < kMergePrefetch; ++i) { ok = ok && (uint32_t)(v[i] >> 32) == tag; L.v[i] = __uint_as_float((uint32_t)v[i]); }
        const bool has = lane < nblk;
        L.mi = has ? __uint_as_float((uint32_t)mi) : -INFINITY;
        L.si = has ? __uint_as_float((uint32_t)si) : 0.0f;
        L.j = col < 2 * T ? col : 0;
        return ok;
    }
};

// U*[column L.j] from one thread's prefetched rows (nblk <= kMergePrefetch): the arithmetic of merge_partials' few-blocks branch,
// operation for operation, for a caller that wants the value in a register instead of in LDS.  Every lane of the wave must call it.
__device__ __forceinline__ float merge_one(const MergeLoads &L, int nblk, int lane)
{
    const float m = wave_max(L.mi);
    const float f = lane < nblk ? expf(L.mi - m) : 0.0f;
    const float S = wave_sum(L.si * f);
    const int fb = __float_as_int(f);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < kMergePrefetch; ++i) acc = __builtin_fmaf(L.v[i], __int_as_float(__builtin_amdgcn_readlane(fb, i)), acc);   // f == 0 past nblk
    return acc / S;
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
// SYNC = false leaves out the closing barrier: for a caller whose threads go on to use only the us[] entries they wrote
// themselves (jj = tid, tid + NT, ...) and that has a barrier of its own before anybody reads somebody else's.
// WIDE picks how the rows beyond the prefetched 16 are loaded (same arithmetic either way): sixteen at a time through a second
// register array (kernels that own a CU: the latency kernel, the stand-alone tail), or eight at a time from one address register
// (the role and one-wave kernels, whose occupancy hangs on ~80 VGPRs and no scratch segment: tests/test_build_artifacts.py).
template <int NT, bool AGENT = false, bool BIG = false, bool SYNC = true, bool WIDE = false>
__device__ __forceinline__ void merge_partials(const float *__restrict__ part, int nblk, int T, float *us, float *sc,
                                               float *red, int tid, float &m_out, float &S_out, bool have_pre, const MergeLoads &pre,
                                               int c0 = 0)
{
    const int col = merge_column<NT>(tid, c0);
#define BN_PLD(ix) (AGENT ? load_agent(part + (ix)) : part[(ix)])
    const int PS = 2 + 2 * T;
    const int lane = tid & 63;
    float m, S;
    if (nblk <= 64) {
        // Few blocks (K <= 4096): every wave reduces the nblk (max, sum) pairs itself -- same inputs,
        // same operations, so all waves (and all workgroups) hold identical m, S and scales -- and all
        // loads are issued before the first use: one memory round trip, one barrier.
        const MergeLoads L = have_pre ? pre : merge_issue<AGENT>(part, nblk, T, tid);
        m = wave_max(L.mi);
        const float f = lane < nblk ? expf(L.mi - m) : 0.0f;       // scale of block `lane`; 0 past nblk
        S = wave_sum(L.si * f);
        // scale of block i = lane i's f, read with v_readlane (ignores EXEC): only the lanes with jj < 2T enter the
        // loop below, and a ds_bpermute shuffle returns nothing from the lanes that did not
        const int fb = __float_as_int(f);
#define BN_SCALE(i) __int_as_float(__builtin_amdgcn_readlane(fb, (i)))
        for (int jj = col; jj < 2 * T; jj += NT) {
            float acc = 0.0f;
            int i0 = 0;
            if (jj == L.j) {
#pragma unroll
                for (int i = 0; i < kMergePrefetch; ++i) acc = __builtin_fmaf(L.v[i], BN_SCALE(i), acc);   // f == 0 past nblk
                i0 = kMergePrefetch;
            }
            // Rows beyond the prefetched ones with several loads in flight (one at a time, each behind the previous row's FMA, 48
            // device-scope loads cost a K=4096 solve 13 us).  A second 16-entry array costs the role kernel 25 VGPRs, reusing the
            // prefetch array a scratch segment, eight scalars with 64-bit addresses 8 VGPRs too many -- each 17-35 % of its
            // throughput at four workgroups per CU (tools/config_rate.py) --, hence the two forms.
            if constexpr (WIDE) {
                for (; i0 < nblk; i0 += kMergePrefetch) {
                    float v[kMergePrefetch];
#pragma unroll
                    for (int q = 0; q < kMergePrefetch; ++q) v[q] = BN_PLD((size_t)min(i0 + q, nblk - 1) * PS + 2 + jj);
#pragma unroll
                    for (int q = 0; q < kMergePrefetch; ++q)
                        if (i0 + q < nblk) acc = __builtin_fmaf(v[q], BN_SCALE(i0 + q), acc);
                }
            } else {
                for (; i0 < nblk; i0 += 8) {
                    // one 32-bit offset from the (uniform) row base, the eight rows at constant distances: one address register
                    const float *row = part + (unsigned)(i0 * PS + 2 + jj);
                    const int left = nblk - i0;
                    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f;
    #define BN_ROW(k) (AGENT ? load_agent(row + (k) * PS) : row[(k) * PS])
                    v0 = BN_ROW(0);
                    if (left > 1) v1 = BN_ROW(1);
                    if (left > 2) v2 = BN_ROW(2);
                    if (left > 3) v3 = BN_ROW(3);
                    if (left > 4) v4 = BN_ROW(4);
                    if (left > 5) v5 = BN_ROW(5);
                    if (left > 6) v6 = BN_ROW(6);
                    if (left > 7) v7 = BN_ROW(7);
    #undef BN_ROW
                    acc = __builtin_fmaf(v0, BN_SCALE(i0), acc);
                    if (left > 1) acc = __builtin_fmaf(v1, BN_SCALE(i0 + 1), acc);
                    if (left > 2) acc = __builtin_fmaf(v2, BN_SCALE(i0 + 2), acc);
                    if (left > 3) acc = __builtin_fmaf(v3, BN_SCALE(i0 + 3), acc);
                    if (left > 4) acc = __builtin_fmaf(v4, BN_SCALE(i0 + 4), acc);
                    if (left > 5) acc = __builtin_fmaf(v5, BN_SCALE(i0 + 5), acc);
                    if (left > 6) acc = __builtin_fmaf(v6, BN_SCALE(i0 + 6), acc);
                    if (left > 7) acc = __builtin_fmaf(v7, BN_SCALE(i0 + 7), acc);
                }
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
        merge_partials<NT, false, false>(grows, ng, T, us, sc, red, tid, m, S, false, MergeLoads{});
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
    if (SYNC) __syncthreads();
    m_out = m;
    S_out = S;
#undef BN_PLD
}

// The tail of one solve of instance b: U* (and the next mean), softmin statistics, normalised weights,
// a stable copy of the costs, and the batch-1 rollout X* of U*.  NT threads (320 as the aux workgroup, 1024 stand-alone for large K).
// LDS: [ window | ustar 2T | X* 3(T+1) | scale nblk | red 32 | group rows ceil(nblk/16) x (2+2T) if nblk > 64 | sampled mode: draws, (mean, std) window ]
// AGENT: the launch overlaps its predecessor (other stream): wait for the counters first, read what the predecessor wrote with
// device-scope loads, and write EVERY output with device-scope (sc1, write-through) stores: the tails of consecutive solves run in
// different kernels, possibly on different XCDs, and write the same addresses -- the tail counter orders the stores themselves, but
// a plain store may sit dirty in its XCD's L2 until its kernel ends, and which kernel ends last is not ordered by anything.
// SELF (round 6): the tail of the solve whose rollouts run in THIS launch (SolveParams::self_tail) -- a synchronous forward() as one
// launch, or the last launch of a batch.  Nothing it needs before the rows is unknown: the state is the caller's (or travels in the kernel
// arguments), so the window is staged and in LDS long before the rollout workgroups are through.  With `gran_self` (the granule copies of
// this solve's partial rows, K <= 1024) every thread then polls the granules IT merges -- one memory round trip from "stored" to "merged",
// no counter, no second fetch -- and thread 0 posts U*[0] to the host's mailbox the moment the merge is done.  The costs travel as plain
// device-scope stores, not as granules: the waves that turn them into weights wait for the workgroups' counts (publish_counter: every
// store acknowledged) first, off the critical path.  Without granules: the counter, then the rows (as AGENT).
template <int GEO, bool LDSWIN, int NT, bool BIG = false, bool AGENT = false, bool WIDE = false, bool REF = false, bool SELF = false>
__device__ __forceinline__ void finish_body(const SolveParams &p, int b, const float *part_all, const float *cost_all,
                                            const float *state_all, float *smem, const unsigned long long *gran_self = nullptr)
{
    static_assert(!SELF || AGENT, "the tail of a launch's own solve reads what the rollout workgroups of that launch publish");
    // Tails of consecutive overlapped solves write the same output buffers, so they are ordered by a counter.  What has to be ordered are
    // the STORES: with the wait in front of everything (rounds 3-4) a tail started when its predecessor was through, and at ~8.5 us a
    // tail (two dependent fetches, the merge, the 50-step rollout of U*) the chain of tails, not the rollouts, set the period of a batch
    // of dependent solves (tools/block_trace_lat.py: the aux workgroup left its launch 4 us after the rollout workgroups).  Open loop,
    // the waves that write (1 ..) wait for the tail before this one just before their first store, behind the merge; wave 0 rolls U*
    // out into LDS meanwhile and meets them at the barrier in front of its own stores.  (Episodes: the environment step reads what the
    // previous tail wrote -- the wait stays in front.)
    // (Reference order: its tail -- a sine and a cosine per step of the serial rollout -- is as long as its period; with the window
    // staged early, below, the deferred wait brings it from 11.05 to 10.8 us per dependent solve.  Alone it had measured 13.1 against
    // 11.5: the phase between the successor's polls and the publication, DESIGN_NOTEBOOK.md R5.4.)
    const bool defer_tail_wait = AGENT && NT > 64 && !p.env_on && !p.slip_on && !p.tail_merged;
    // Reference order, whose tail is as long as its period: while the solve whose
    // tail this is still runs, the window is staged around the state as it reads NOW -- two dependent fetches that were the first
    // 1.4 us of the tail after the rows: 11.5 -> 10.9 us per dependent solve.  The state is written early in that solve's prologue
    // and almost always there; a thread that read something else (checked below against the load behind the wait) stages its cells
    // again.  (The default arithmetic's period is the rollouts' path and its instantiation is left exactly as it was: the same
    // lines there -- even as dead code that only moved declarations -- measured 7.93 -> 8.11 us.)
    float sx0 = 0.0f, sy0 = 0.0f;
    bool staged = false;
    if constexpr (AGENT && REF && LDSWIN && !SELF) {
        if (!p.slip_on && !p.env_on) {
            sx0 = ld<AGENT>(state_all + b * 3 + 0); sy0 = ld<AGENT>(state_all + b * 3 + 1);
            stage_window(smem, p.map + (size_t)b * p.map_stride, window_origin<GEO>(p, sx0, sy0), p.WN, p.G, (int)threadIdx.x, NT);
            staged = true;
        }
    }
    const bool self_gran = SELF && gran_self != nullptr && !p.tail_merged;
    if (AGENT && !SELF) {
        if (threadIdx.x == 0) {
            wait_counter(flag_ctr(p.flag_part, p.prev_slot * p.B + b), p.wait_part, p);   // the solve whose tail this is has published everything
            if (!defer_tail_wait) wait_counter(flag_ctr(p.flag_tail, b), p.wait_tail, p);   // and the tail before it has left the output buffers
        }
        __syncthreads();
    }
    // p.tail_merged: U* and the softmin statistics of this solve were merged already (ticket merge of the sampled
    // kernel, which also wrote the next mean); they come from (ustar_prev, stats_prev).
    const int T = p.T, K = p.K, nblk = p.nblk, PS = 2 + 2 * p.T;
    float *win = smem;
    float *us = win + (LDSWIN ? p.WN * p.WN : 0);
    float *xl = us + 2 * T;                             // X* rows: the serial rollout's lane stages them here
    float *sc = xl + 3 * (T + 1);
    float *red = sc + nblk;

    const int tid = threadIdx.x;
    const float *__restrict__ map = p.map + (size_t)b * p.map_stride;
    const float *part = part_all + (size_t)b * nblk * PS;
    float sx, sy, sth;
    BN_SSTAMP(0);
    if constexpr (SELF) {                              // the state this launch's rollouts start from: the caller's, known from the start
        if (p.state_inline) { sx = p.sv0; sy = p.sv1; sth = p.sv2; }
        else { sx = p.state[b * 3 + 0]; sy = p.state[b * 3 + 1]; sth = p.state[b * 3 + 2]; }
    } else {
        sx = ld<AGENT>(state_all + b * 3 + 0); sy = ld<AGENT>(state_all + b * 3 + 1); sth = ld<AGENT>(state_all + b * 3 + 2);
    }
    BN_STAMP(8);

    // the merge's loads go out before the window staging (which waits for the state): one memory round trip for both
    MergeLoads pre{};
    const bool pre_ok = !p.tail_merged && nblk <= 64;
    if (pre_ok && !SELF) pre = merge_issue<AGENT>(part, nblk, T, tid);
    Win w{0, 0, 0.f, 0.f, 0.f, 0.f};
    if (LDSWIN && !p.slip_on) {
        w = window_origin<GEO>(p, sx, sy);
        if constexpr (AGENT && REF && !SELF) { if (!(staged && sx == sx0 && sy == sy0)) stage_window(win, map, w, p.WN, p.G, tid, NT); }   // (per thread: no barrier inside)
        else stage_window(win, map, w, p.WN, p.G, tid, NT);
    }
    BN_SSTAMP(1);
    if constexpr (SELF) {
        // ... and only now the wait for this launch's own rollout workgroups: they were dispatched before this workgroup (grid order) and
        // wait for nobody that comes after them, so the wait ends; it is bounded all the same (the error word, see bn_mppi_forward_async)
        if (self_gran) {
            const unsigned long long *grows = gran_self + (size_t)b * nblk * PS;
            const uint32_t tag = (uint32_t)p.tail_solve + 1u;
            const bool early = p.host_paced && !p.no_early_mail;
            if (tid < 64 && !p.have_prev && p.mail != nullptr && early) {
                // The host waits for U*[0] -- two columns of the sixteen rows.  Wave 0 polls just those (and the rows' statistics): a light
                // poll sees the rows sooner than the full one (the lines that are being written are what makes a poll slow: notebook R5.4),
                // and its lanes 0 and 1 post the first action -- merge_one: the arithmetic of merge_partials, the same bits -- before the
                // workgroup has merged anything else.  Then it joins the full poll below (the rows are there by then).  Host-paced launches
                // only (same box, C loop: 15.2 -> 14.6 us per step): the tail ends 1.2 us later for it, and a one-launch forward that is
                // called back to back waits for exactly that end (22.7 -> 23.7 us per step there).
                GranulePoll pa;
                MergeLoads L{};
                bool got = false;
                for (int it = 0; it < (1 << 21) && !got; ++it) {
                    pa.issue(grows, nblk, T, tid, tid & 1);
                    got = __builtin_amdgcn_ballot_w64(!pa.take(tag, nblk, T, tid, tid & 1, L)) == 0;
                }
                if (!got) raise_wait_expired(p, nullptr, 902);
                const float val = merge_one(L, nblk, tid);
                if (tid < 2) store_granule_host(p.mail + 2 * b + tid, val, (uint32_t)p.tail_solve + 1u);
                BN_SSTAMP(3);
            }
            bool ok = false;
            for (int it = 0; it < (1 << 22) && !ok; ++it) {
                pre = merge_issue_granules(grows, nblk, T, tid, tag, ok, tid);
                if (!ok) __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) raise_wait_expired(p, nullptr, 901);
        } else {
            if (tid == 0) wait_counter(flag_ctr(p.flag_part, p.prev_slot * p.B + b), p.wait_part, p);
            __syncthreads();
            if (pre_ok) pre = merge_issue<AGENT>(part, nblk, T, tid);
        }
        if (!defer_tail_wait) {                        // (not reached by today's callers: the latency kernel's self tail always defers)
            if (tid == 0) {
                if (self_gran) wait_counter(flag_ctr(p.flag_part, p.prev_slot * p.B + b), p.wait_part, p);
                wait_counter(flag_ctr(p.flag_tail, b), p.wait_tail, p);
            }
            __syncthreads();
        }
    }
    BN_STAMP(9);
    BN_SSTAMP(2);

    float m, S;
    if (p.tail_merged) {
        for (int j = tid; j < 2 * T; j += NT) {
            const float u = ld<AGENT>(p.ustar_prev + (size_t)b * 2 * T + j);
            us[j] = u;
            if (!(BIG && p.ustar_written)) st<AGENT>(p.ustar + (size_t)b * 2 * T + j, u);      // (BIG: the stand-alone tail, the only one that follows a merge kernel)
            if (p.out_copy) st<AGENT>(p.out_copy + (size_t)b * 2 * T + j, u);
        }
        m = ld<AGENT>(p.stats_prev + b * 2 + 0);
        S = ld<AGENT>(p.stats_prev + b * 2 + 1);
        __syncthreads();
    } else {
        if constexpr (SELF) {
            // No tail in front of this one (a synchronous forward(): every earlier tail was ordered before the launch by the stream): the
            // first control goes to the host's mailbox at once -- by the two threads that merged its columns, each its own, in front of the
            // barrier the other waves' columns are waited for at -- and before the wait for the costs, the X* rollout and the weights.
            merge_partials<NT, AGENT, BIG, false, WIDE>(part, nblk, T, us, sc, red, tid, m, S, pre_ok, pre);
            if ((!self_gran || !p.host_paced || p.no_early_mail) && !p.have_prev && p.mail != nullptr && nblk <= 64 && tid < 2) store_granule_host(p.mail + 2 * b + tid, us[tid], (uint32_t)p.tail_solve + 1u);   // (self_gran: posted above)
            __syncthreads();
        } else {
            merge_partials<NT, AGENT, BIG, true, WIDE>(part, nblk, T, us, sc, red, tid, m, S, pre_ok, pre);
        }
    }
    const bool mail_early = SELF && !p.have_prev && p.mail != nullptr && (p.tail_merged || nblk <= 64);
    if (mail_early && p.tail_merged && tid == 0) {
        store_granule_host(p.mail + 2 * b, us[0], (uint32_t)p.tail_solve + 1u);
        store_granule_host(p.mail + 2 * b + 1, us[1], (uint32_t)p.tail_solve + 1u);
    }
    if (!self_gran || !p.host_paced || p.no_early_mail) BN_SSTAMP(3);
    // U*, the next mean, the statistics: threads j0, j0 + step, .. (all of them, or -- deferred wait -- the writing waves behind their wait)
    auto store_ustar = [&](int j0, int step, bool first) {
        if (!p.tail_merged) {
            const bool spoiled = batch_spoiled(p);             // a wait of this stretch expired: nothing merged since may reach `mean`
            for (int j = j0; j < 2 * T; j += step) {
                st<AGENT>(p.ustar + (size_t)b * 2 * T + j, us[j]);
                if (p.out_copy) st<AGENT>(p.out_copy + (size_t)b * 2 * T + j, us[j]);
                if (spoiled) continue;
                if (p.mean_used) st<AGENT>(p.mean_used + (size_t)b * 2 * T + j, ld<AGENT>(p.mean + (size_t)b * 2 * T + j));   // what this solve sampled around
                st<AGENT>(p.mean + (size_t)b * 2 * T + j, us[j]);  // _previous_action_seq = U*, no shift (mppi.py:217)
            }
        }
        if (first) {
            if (p.mail && !mail_early) {               // the first control of U*, for the host that waits for it: out before anything else
                store_granule_host(p.mail + 2 * b, us[0], (uint32_t)p.tail_solve + 1u);
                store_granule_host(p.mail + 2 * b + 1, us[1], (uint32_t)p.tail_solve + 1u);
            }
            st<AGENT>(p.stats + b * 2 + 0, m);
            st<AGENT>(p.stats + b * 2 + 1, S);
        }
    };
    if (!defer_tail_wait) store_ustar(tid, NT, tid == 0);
    if (p.env_on && tid == 64) {
        // the environment step that follows this solve: apply U*[0], log state, reward and goal arrival
        const EnvStep e = env_advance<GEO>(p, b, sx, sy, sth, us[0], us[1], p.env_z, (uint64_t)p.ep_index);
        const size_t B = p.B;
        float *row = p.ep_states + ((size_t)(p.ep_index + 1) * B + b) * 3;
        row[0] = e.x; row[1] = e.y; row[2] = e.th;
        st<AGENT>(p.env_state + b * 3 + 0, e.x); st<AGENT>(p.env_state + b * 3 + 1, e.y); st<AGENT>(p.env_state + b * 3 + 2, e.th);
        p.ep_reward[(size_t)p.ep_index * B + b] = e.reward;
        p.ep_action[((size_t)p.ep_index * B + b) * 2 + 0] = us[0];
        p.ep_action[((size_t)p.ep_index * B + b) * 2 + 1] = us[1];
        if (p.ep_index == 0) {
            float *row0 = p.ep_states + (size_t)b * 3;
            row0[0] = sx; row0[1] = sy; row0[2] = sth;
        }
        // (tails of consecutive solves may run in different launches in flight at once: device-scope accesses, like the mean)
        if (e.reached && !e.frozen) {
            const int seen = AGENT ? __hip_atomic_load(p.ep_done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.ep_done[b];
            if (seen < 0) { if (AGENT) __hip_atomic_store(p.ep_done + b, p.ep_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else p.ep_done[b] = p.ep_index; }
        }
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
                const size_t g = (size_t)min(w.wy0 + r, p.G - 1) * p.G + min(w.wx0 + c, p.G - 1);
                win2[e] = make_float2(map[g], sg[g]);
            }
        }
        __syncthreads();
        BN_STAMP(12);
        if (tid == 0) {
            float *Xs = xl;                              // staged in LDS, written out coalesced below
            float xn, yn, tn;
            if (LDSWIN) {
                SlipChain c;
                c.x = sx; c.y = sy; c.th = sth;
                sincos_spec(c.th, c.sn, c.cs);
                c.e = slip_cell_safe<GEO, true>(p, w, sx, sy);
                slip_chain_step<GEO, true, REF>(p, win2, w, c, us[0], us[1], zol[0], xn, yn, tn);
                Xs[0] = xn; Xs[1] = yn; Xs[2] = tn;
                int t = 1;
                for (; t + 4 <= T; t += 4) {                 // controls and draws of four steps read up front
                    float uq[4][3];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { uq[i][0] = us[2 * (t + i)]; uq[i][1] = us[2 * (t + i) + 1]; uq[i][2] = zol[t + i]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        slip_chain_step<GEO, false, REF>(p, win2, w, c, uq[i][0], uq[i][1], uq[i][2], xn, yn, tn);
                        Xs[3 * (t + i)] = xn; Xs[3 * (t + i) + 1] = yn; Xs[3 * (t + i) + 2] = tn;
                    }
                }
                for (; t < T; ++t) {
                    slip_chain_step<GEO, false, REF>(p, win2, w, c, us[2 * t], us[2 * t + 1], zol[t], xn, yn, tn);
                    Xs[3 * t] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
                }
                Xs[3 * T] = c.x; Xs[3 * T + 1] = c.y; Xs[3 * T + 2] = c.th;
                BN_STAMP(11);
            } else {
                float x = sx, y = sy, th = sth;
                float sn, cs;
                sincos_spec(th, sn, cs);
                for (int t = 0; t < T; ++t) {
                    const int e = slip_cell_safe<GEO, false>(p, w, x, y);
                    const float trav = trav_from_slip(map[e], sg[e], zol[t]);
                    if constexpr (REF) {
                        sincos_spec(th, sn, cs);
                        const float tv = trav * us[2 * t];
                        xn = x + (tv * cs) * p.dt; yn = y + (tv * sn) * p.dt;
                        tn = th + (trav * us[2 * t + 1]) * p.dt;
                        th = wrap_angle(tn);
                    } else {
                    const float g = us[2 * t] * p.dt, dth = trav * (us[2 * t + 1] * p.dt);
                    xn = __builtin_fmaf(trav, g * cs, x); yn = __builtin_fmaf(trav, g * sn, y);
                    tn = theta_step(th, dth, t == 0);
                    rotate_spec(cs, sn, dth);
                    }
                    Xs[3 * t] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
                    x = clampf(xn, p.x0, p.x_hi); y = clampf(yn, p.y0, p.y_hi);
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
        float *Xs = xl;                              // staged in LDS, written out coalesced below
        float xn, yn, tn;
        if constexpr (REF) {
            chain_step<GEO, LDSWIN, true, true, false, 0, true>(p, win, map, w, c, us[0], us[1], xn, yn, tn);
            Xs[0] = xn; Xs[1] = yn; Xs[2] = tn;
            int t = 1;
            for (; t + 4 <= T; t += 4) {                   // controls of four steps read up front (LDS latency off the chain)
                float uq[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) { uq[i][0] = us[2 * (t + i)]; uq[i][1] = us[2 * (t + i) + 1]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    chain_step<GEO, LDSWIN, false, true, false, 0, true>(p, win, map, w, c, uq[i][0], uq[i][1], xn, yn, tn);
                    Xs[3 * (t + i) + 0] = xn; Xs[3 * (t + i) + 1] = yn; Xs[3 * (t + i) + 2] = tn;
                }
            }
            for (; t < T; ++t) {
                chain_step<GEO, LDSWIN, false, true, false, 0, true>(p, win, map, w, c, us[2 * t], us[2 * t + 1], xn, yn, tn);
                Xs[3 * t + 0] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
            }
        } else {
            // The chain form of the latency kernel (round 4): the next step's controls are folded into its factors at the end of each
            // step, under the gather's latency (PREP), and the window index is the three-instruction form (ASMIDX with the LDS window).
            // Same operations per step, so the same bits; this serial rollout is the last thing between a batch's final solve and
            // its results (4.5 -> ~3.8 us at T = 50), and part of every synchronous forward().
            constexpr int AI = LDSWIN ? 1 : 0;
            chain_prepare(p, c, us[0], us[1]);
            chain_step<GEO, LDSWIN, true, true, true, AI>(p, win, map, w, c, us[0], us[1], xn, yn, tn, T > 1 ? us[2] : 0.0f, T > 1 ? us[3] : 0.0f);
            Xs[0] = xn; Xs[1] = yn; Xs[2] = tn;
            int t = 1;
            for (; t + 4 <= T; t += 4) {                   // controls of four steps (and the first of the next four) read up front
                float uq[5][2];
#pragma unroll
                for (int i = 0; i < 5; ++i) { const int tt = min(t + i, T - 1); uq[i][0] = us[2 * tt]; uq[i][1] = us[2 * tt + 1]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    chain_step<GEO, LDSWIN, false, true, true, AI>(p, win, map, w, c, uq[i][0], uq[i][1], xn, yn, tn, uq[i + 1][0], uq[i + 1][1]);
                    Xs[3 * (t + i) + 0] = xn; Xs[3 * (t + i) + 1] = yn; Xs[3 * (t + i) + 2] = tn;
                }
            }
            for (; t < T; ++t) {
                const int tt = min(t + 1, T - 1);
                chain_step<GEO, LDSWIN, false, true, true, AI>(p, win, map, w, c, us[2 * t], us[2 * t + 1], xn, yn, tn, us[2 * tt], us[2 * tt + 1]);
                Xs[3 * t + 0] = xn; Xs[3 * t + 1] = yn; Xs[3 * t + 2] = tn;
            }
        }
        Xs[3 * T + 0] = c.x; Xs[3 * T + 1] = c.y; Xs[3 * T + 2] = c.th;
        BN_STAMP(11);
    } else if (NT > 64 && tid >= 64) {
        if (defer_tail_wait) {                         // every writing wave waits for itself: wave 0 is on the serial rollout, no barrier to meet at
            if ((tid & 63) == 0) {
                // (granule-polling self tail: the rows are here, the per-rollout costs -- plain device-scope stores -- only once every
                // rollout workgroup has counted itself in behind its acknowledged stores)
                if (self_gran) wait_counter(flag_ctr(p.flag_part, p.prev_slot * p.B + b), p.wait_part, p);
                wait_counter(flag_ctr(p.flag_tail, b), p.wait_tail, p);
            }
            store_ustar(tid - 64, NT - 64, tid == 64);
        }
        // _weights = softmax(-costs / lambda)   mppi.py:193
        const float *cost = cost_all + (size_t)b * K;
        float *wout = p.w + (size_t)b * K;
        float *cout = p.cost_out + (size_t)b * K;
        for (int k = tid - 64; k < K; k += NT - 64) {
            const float ck = ld<AGENT>(cost + k);
            st<AGENT>(cout + k, ck);
            st<AGENT>(wout + k, expf((-ck) / p.lambda_ - m) / S);
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
    // optimal_state_seq out of LDS: coalesced, and to the caller's copy of the packed U* | X* block as well
    BN_SSTAMP(4);
    __syncthreads();
    BN_SSTAMP(5);
    {
        float *Xg = p.xstar + (size_t)b * (T + 1) * 3;
        float *Xc = p.out_copy ? p.out_copy + (size_t)p.B * 2 * T + (size_t)b * (T + 1) * 3 : nullptr;
        for (int i = tid; i < 3 * (T + 1); i += NT) {
            const float v = xl[i];
            st<AGENT>(Xg + i, v);
            if (Xc) st<AGENT>(Xc + i, v);
        }
    }
    // one more tail done (counted per instance): what an overlapped successor's tail waits for before it takes the output buffers
    if (p.flag_tail) publish_counter(flag_ctr(p.flag_tail, b), tid);
    BN_SSTAMP(6);
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
__device__ __forceinline__ void ticket_merge(const SolveParams &p, int b, int blk, float *scratch)
{
    const int T = p.T, tid = threadIdx.x, PS = 2 + 2 * p.T;
    float *us = scratch, *sc = us + 2 * T, *red = sc + p.nblk;
    int *flag = reinterpret_cast<int *>(red + 32);       // at most 64 rows reach merge_partials here: no group rows in LDS
    int *ticket = p.ticket + (size_t)b * kTicketStride;
    const float *part = p.part + (size_t)b * p.nblk * PS;
    const float *rows = part;
    int nrows = p.nblk;
    if (p.nblk > 64) {
        const int g = blk / kGroupRows, ng = (p.nblk + kGroupRows - 1) / kGroupRows;
        const int in_group = min(kGroupRows, p.nblk - g * kGroupRows);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *flag = (atomicAdd(ticket + 1 + g, 1) == in_group - 1) ? 1 : 0;
        __syncthreads();
        if (!*flag) return;
        if (tid == 0) __hip_atomic_store(ticket + 1 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (an overlapped successor takes its tickets behind the count below)
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
    if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (ordered by the stream, or by the count below)
    float m, S;
    merge_partials<NT, true, false>(rows, nrows, T, us, sc, red, tid, m, S, false, MergeLoads{});
    // Member of an overlapped batch (p.flag_part): the next launch may be running already -- on the other stream, its workgroups
    // waiting for exactly this -- so what it reads goes out as device-scope stores and the merge counts itself in behind them.
    const bool pubm = p.flag_part != nullptr;
    const bool spoiled = pubm && batch_spoiled(p);         // (see raise_wait_expired)
    for (int j = tid; j < 2 * T; j += NT) {
        const size_t at = (size_t)b * 2 * T + j;
        if (pubm) {
            store_agent(p.ustar_cur + at, us[j]);
            if (spoiled) continue;
            if (p.mean_used) store_agent(p.mean_used + at, load_agent(p.mean + at));
            store_agent(p.mean + at, us[j]);
        } else {
            p.ustar_cur[at] = us[j];
            if (p.mean_used) p.mean_used[at] = p.mean[at];   // what this solve sampled around
            p.mean[at] = us[j];                        // _previous_action_seq = U*, no shift (mppi.py:217)
        }
    }
    if (tid == 0) {
        if (pubm) { store_agent(p.stats_cur + b * 2 + 0, m); store_agent(p.stats_cur + b * 2 + 1, S); }
        else { p.stats_cur[b * 2 + 0] = m; p.stats_cur[b * 2 + 1] = S; }
    }
    if (pubm) publish_counter(flag_ctr(p.flag_part, p.cur_slot * p.B + b), tid);
    BN_STAMP_ANY(7);
}

// Workgroup barrier that only drains LDS traffic: global stores of the consumer wave stay in flight.
#define BN_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Producer: the clamped perturbed controls of steps t and t+1 (t even) of this lane's rollout,
//   u = clamp(mean + sigma * eps, u_min, u_max)          mppi.py:152-157
// written to the LDS control tile (and to HBM when _perturbed_action_seqs is materialised).
// FRESH: the Philox round keys are derived at the call (philox_eps_pair<true>) instead of living in scalar registers across the caller's
// loop -- the role kernel (round 6): its loops reload spilled scalars with v_readlane, a vector instruction each.
template <int EPS, bool STORE_U, bool FRESH = false>
__device__ __forceinline__ void produce_pair(const SolveParams &p, const float *__restrict__ eps, int b, int kk, int t,
                                             uint64_t solve, const float *ml, float *Ul, float *Ub, size_t Kp, int lane)
{
    float e[4];
    const int t1 = min(t + 1, p.T - 1);
    if (EPS == kEpsPhilox) {
        philox_eps_pair<FRESH>(p.seed, solve, (uint32_t)b, (uint32_t)(kk + p.k0), (uint32_t)(t >> 1), e);
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


// The noise of steps t and t+1 (t even) of rollout kk: the library's Philox stream or the caller's arrays.
template <int EPS, bool FRESH_KEYS = false>
__device__ __forceinline__ void noise_pair(const SolveParams &p, int b, int kk, int t, float e[4])
{
    const int t1 = min(t + 1, p.T - 1);
    if (EPS == kEpsPhilox) {
        philox_eps_pair<FRESH_KEYS>(p.seed, p.solve, (uint32_t)b, (uint32_t)(kk + p.k0), (uint32_t)(t >> 1), e);
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

static int grid_for(size_t n) { return (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256); }

}  // namespace

// launchers that live in their own translation units (one per kernel family)
hipError_t launch_rollout_role_philox(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_role_kt2(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_role_t2k(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_lat_self_philox(const SolveParams &p, hipStream_t s);    // rollout_lat_self_*.hip
hipError_t launch_rollout_lat_self_kt2(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_lat_self_t2k(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_lat_self_ref_philox(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_lat_self_ref_kt2(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_lat_self_ref_t2k(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_role_ref_philox(const SolveParams &p, hipStream_t s);   // rollout_role_ref_*.hip: BN_FLAG_REFERENCE_ORDER
hipError_t launch_rollout_role_ref_kt2(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_role_ref_t2k(const SolveParams &p, hipStream_t s);
hipError_t launch_rollout_wave(const SolveParams &p, EpsMode mode, hipStream_t s);
hipError_t launch_rollout_wave_ref(const SolveParams &p, EpsMode mode, hipStream_t s);   // rollout_wave_ref.hip

}  // namespace bn
