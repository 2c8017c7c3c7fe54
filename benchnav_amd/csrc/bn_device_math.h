// bn_device_math.h -- fp32 arithmetic of the MPPI hot path for gfx950 device code.
//
// Implements DESIGN.md "Arithmetic spec": strict evaluation order, no implicit
// contraction (the library is compiled with -ffp-contract=off; every FMA is an
// explicit __builtin_fmaf), correctly rounded division and square root.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn {

// float32 casts of torch.pi and 2*torch.pi (reference robot_model.py:90)
constexpr float kPi = 3.14159274101257324f;
constexpr float kTwoPi = 6.28318548202514648f;

// ---- sincos spec -------------------------------------------------------------
// n = rint(x/pi) by the add-magic trick (one fma; |x| < 1.3e7); r = x - n*pi by a 3-term Cody-Waite
// split with fma, r in [-pi/2, pi/2]; sin(r) = r + r*s*P(s) (degree 9) and cos(r) = 1 + s*Q(s)
// (degree 10) evaluated as one packed stream; sin(x) = (-1)^n sin(r), cos(x) = (-1)^n cos(r): one
// shared sign flip.  Absolute error <= 1.2e-7 (positions integrate the absolute error).
// 16 instructions.  The arithmetic (DESIGN.md "Arithmetic spec") is restated independently by the test oracle.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void sincos_spec(float x, float &sn, float &cs)
{
    const float t = __builtin_fmaf(x, 0.318309886183790672f, 12582912.0f);
    const float fn = t - 12582912.0f;
    float r = __builtin_fmaf(-fn, 3.140625f, x);
    r = __builtin_fmaf(-fn, 9.67502593994140625e-4f, r);
    r = __builtin_fmaf(-fn, 1.509957990978376432e-7f, r);
    const float s = r * r;
    const v2f s2 = {s, s};
    // lane x: sine coefficients (one fewer: starts at SC4), lane y: cosine coefficients
    v2f pq = {2.599125082269893e-06f, __builtin_fmaf(-2.6072027026202704e-07f, s, 2.476157715136651e-05f)};
    pq = __builtin_elementwise_fma(pq, s2, v2f{-0.0001980613305931911f, -0.001388839678838849f});
    pq = __builtin_elementwise_fma(pq, s2, v2f{0.008333009667694569f, 0.04166664183139801f});
    pq = __builtin_elementwise_fma(pq, s2, v2f{-0.16666656732559204f, -0.5f});
    // two plain FMAs, pinned: left to itself the vectoriser packs them into one v_pk_fma_f32 whose operand pairs cost two
    // register moves and a hazard slot -- four issue slots instead of two on the one wave whose slots are the solve's latency
    float S, C;
    const float ps = pq.x * s;
    asm("v_fma_f32 %0, %1, %2, %2" : "=v"(S) : "v"(ps), "v"(r));
    asm("v_fma_f32 %0, %1, %2, 1.0" : "=v"(C) : "v"(pq.y), "v"(s));
    const uint32_t sign = __float_as_uint(t) << 31;          // parity of n
    sn = __uint_as_float(__float_as_uint(S) ^ sign);
    cs = __uint_as_float(__float_as_uint(C) ^ sign);
}

// ---- carried heading ---------------------------------------------------------
// Within a rollout the heading vector (cos, sin) is set once from the start heading (sincos_spec) and then carried by a
// rotation per step, (cs, sn) <- R(d)(cs, sn) with d the step's heading increment (robot_model.py:88), instead of being
// re-evaluated from theta: 7 instructions (two of them scalar, five packed) where wrap + sincos take 21, on the one wave
// whose issue slots are a solve's latency -- and theta itself drops out of the chain (it is an output only; whoever stores
// the trajectory integrates and wraps it).  cos d and sin d / d are degree-6 Taylor polynomials in d (truncation d^8/40320:
// 3e-13 at the reference's |d| <= 0.1, 1e-7 at 0.5, which is where bn_mppi_create draws the line); the rounding of a
// step is carried along, ~sqrt(T) * 4e-8 after T steps.  DESIGN.md "Arithmetic spec"; restated independently by the
// test oracle (bn_rotate_spec).  Scalar statement of what the packed stream computes:
//   d2 = d*d; cd = fma(d2, fma(d2, fma(d2, C6, C4), C2), 1); sp = fma(d2, fma(d2, fma(d2, S7, S5), S3), 1); sd = d*sp;
//   cs' = fma(cs, cd, -(sn*sd)); sn' = fma(sn, cd, cs*sd)
template <bool ONEBLOCK = false>
__device__ __forceinline__ void rotate_spec(float &cs, float &sn, float d, v2f c0_pinned = v2f{0.0f, 0.0f})
{
    if (ONEBLOCK) {
        // The latency kernel's chain wave: the seven instructions below as the compiler emits them for the statements that
        // follow, in ONE asm block -- between the packed multiply (op_sel_hi[0] = 1) and the asm of the last FMA the hazard
        // recogniser put an s_nop (trav_window has the story).  v[124:127]: scratch; v127 is read as the unused half of a pair.
        v2f h = {cs, sn};
        const v2f c0 = c0_pinned;                      // {1/24, 1/120}, in vector registers the caller keeps (Win::c0_v)
        const v2f c1 = {-0.00138888892251998186f, -0.000198412701138295233f};
        const v2f c2 = {-0.5f, -0.16666667163372040f};
        asm("v_mul_f32 v126, %1, %1\n\t"
            "v_pk_fma_f32 v[124:125], v[126:127], %2, %3 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 v[124:125], v[126:127], v[124:125], %4 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 v[124:125], v[126:127], v[124:125], 1.0 op_sel_hi:[0,1,0]\n\t"
            "v_mul_f32 v126, %1, v125\n\t"
            "v_pk_mul_f32 v[126:127], %0, v[126:127] op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %0, v[124:125], v[126:127] op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1]"
            : "+v"(h) : "v"(d), "s"(c1), "v"(c0), "s"(c2) : "v124", "v125", "v126", "v127");
        cs = h.x;
        sn = h.y;
        return;
    }
    const float d2 = d * d;
    const v2f dd = {d2, d2};
    v2f pq = __builtin_elementwise_fma(dd, v2f{-0.00138888892251998186f, -0.000198412701138295233f}, v2f{0.0416666679084300995f, 0.00833333376795053482f});
    pq = __builtin_elementwise_fma(dd, pq, v2f{-0.5f, -0.16666667163372040f});
    pq = __builtin_elementwise_fma(dd, pq, v2f{1.0f, 1.0f});                  // (cos d, sin d / d)
    const float sd = d * pq.y;
    const v2f h = {cs, sn};
    const v2f u = h * v2f{sd, sd};                                             // (cs*sd, sn*sd)
    // (cs, sn) * cd + (-(sn*sd), cs*sd): the swap and the (exact) negation of the addend ride on the packed FMA's operand
    // selectors -- spelled out, the compiler forms (-sn, cs) with a v_xor and a v_mov first, two more issue slots per step
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1]" : "=v"(r) : "v"(h), "v"(pq), "v"(u));
    cs = r.x;
    sn = r.y;
}

// torch.remainder(a, b), b > 0: fmod (exact) then the divisor-sign fix.  The
// fast paths return exactly what fmodf would (Sterbenz / identity).
__device__ __forceinline__ float py_mod_pos(float a, float b)
{
    if (a >= 0.0f && a < b) return a;
    if (a >= b && a < 2.0f * b) return a - b;
    if (a < 0.0f && a > -b) return a + b;
    float m = fmodf(a, b);
    if (m != 0.0f && m < 0.0f) m += b;
    return m;
}

// (theta + pi) % (2 pi) - pi     reference robot_model.py:90
__device__ __forceinline__ float wrap_angle(float th)
{
    return py_mod_pos(th + kPi, kTwoPi) - kPi;
}

// Same value, branch-free, valid for theta + pi in (-2 pi, 4 pi): every step after the first,
// because the previous wrap left theta in [-pi, pi] and |trav * omega * dt| < pi (checked at create).
// q = floor(a / 2pi) is -1, 0 or 1 there; with the multiplier fl(1/2pi_f) the rounded product lands on
// the right side of 1 at a = 2pi_f and its predecessor, and rounding is monotone, so q is exact for every
// float a in the interval (tests: test_near_wrap_identity).  fma(q, -2pi, a) is then
// a - 2pi (exact, Sterbenz), a, or a + 2pi rounded once: exactly fmod plus torch.remainder's sign fix.
__device__ __forceinline__ float wrap_angle_near(float th)
{
    const float a = th + kPi;
    const float q = floorf(a * 0.15915493667125702f);
    return __builtin_fmaf(q, -kTwoPi, a) - kPi;
}

// min(max(v, lo), hi) in one v_med3_f32 (lo <= hi)
__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    return __builtin_amdgcn_fmed3f(v, lo, hi);
}

// Correctly rounded sqrt (== sqrtf of the oracle's libm): the compiler's IEEE expansion of sqrtf under -fno-fast-math
// (v_sqrt_f32 + two fma residuals that pick the correctly rounded neighbour, 2^32 scaling for inputs v_sqrt_f32 would
// flush).  A hand-written form of the same fix-up was measured at the same speed (selects) or slower (a branch for the
// tiny inputs splits the consumer's four-step block), so the compiler's stays.  tests/test_gpu_device_math.py checks it
// value by value against IEEE sqrt, denormals included.
__device__ __forceinline__ float sqrt_cr(float x) { return sqrtf(x); }

// The same value for x == 0 and 2^-96 <= x < inf -- squared distances between float positions on a map are zero or at least
// (one ulp of a coordinate)^2 >= 2^-68 --, without the compiler's scaling of small arguments and its inf / nan pass-through: v_sqrt_f32
// (at most one ulp off), then the neighbour below / above is taken when the residual x - s' * s says it lies on the correct side.
// Nine instructions instead of seventeen: consumer B's SIMD is the one that runs out of issue slots behind a fast chain.
// tests/test_gpu_device_math.py checks it value by value against IEEE sqrt.
__device__ __forceinline__ float sqrt_cr_normal(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = rm <= 0.0f ? sm : s;
    s = rp > 0.0f ? sp : s;
    return s;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return min(max(v, lo), hi);
}

// ---- Philox4x32-10 + Box-Muller ------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint64_t mul_wide_u32(uint32_t a, uint32_t b)
{
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "vcc");
    return r;
}

template <int ROUNDS>
__device__ __forceinline__ u32x4 philox4x32(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // one v_mad_u64_u32 per product (both halves; measured full rate, tools/ubench3.hip) instead of
        // v_mul_lo_u32 + v_mul_hi_u32 (each ~4.5 SIMD cycles per wavefront)
        const uint64_t p0 = mul_wide_u32(0xD2511F53u, c.x), p1 = mul_wide_u32(0xCD9E8D57u, c.z);
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // hi ^ c ^ key as ONE v_bitop3_b32 (gfx950; truth table 0x96 = three-way xor) instead of two v_xor_b32: 16 of a block's 48
        // instructions.  Round 0 keeps the plain form: its counter words and the key are wave-uniform there and fold into scalars.
        if (r == 0) c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        else c = u32x4{(uint32_t)__builtin_amdgcn_bitop3_b32(hi1, c.y, k0, 0x96), lo1, (uint32_t)__builtin_amdgcn_bitop3_b32(hi0, c.w, k1, 0x96), lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) { return philox4x32<10>(c, k0, k1); }
// The per-rollout streams (control noise, slip draws: one block per rollout and step pair, the hot ones) take EIGHT rounds:
// Philox4x32-7 is the smallest variant that passes BigCrush (Salmon et al., SC'11: "Philox4x32-7 ... Crush-resistant"; ten is
// the authors' default with a safety margin), and the noise is 35 % of the throughput kernel's VALU instructions and what the
// latency kernel's producers spend their time on.  (Eight, not seven: with seven the register allocator gives one role-kernel
// variant a 20-byte scratch segment -- the emergency slot tests/test_build_artifacts.py exists to catch: 23 -> 28 us per
// 64-instance launch.)  The library's own stream is not part of the parity spec;
// bn_mppi_get_philox_noise / bn_mppi_get_slip_noise regenerate exactly what the kernels consume.
constexpr int kStreamRounds = 8;

// Two independent standard normals from two 32-bit words (Box-Muller).  This is the library's own noise
// stream, not part of the parity spec, so it uses the hardware transcendentals: v_log_f32, v_sqrt_f32 and
// v_sin_f32 / v_cos_f32 (which take their argument in revolutions, so 2*pi*u2 is never formed).  ~12
// instructions per pair instead of ~50 with the precise library forms; bn_mppi_get_philox_noise regenerates
// the identical values with the same instructions.
// (round 5) The words enter whole: u1 = fl(a) 2^-32 + 2^-33 in (0, 1] (fl = the conversion's round-to-nearest; the product is
// exact, so one fma), u2 = fl(b) 2^-32 in [0, 1] revolutions (1 is the angle 0 again).  Round 4 cut both to 24 bits first (a shift
// each: 4 of a block's 72 instructions); now the small u1 keep all their bits (largest radius 6.76 instead of 5.9).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1)
{
    const float u1 = __builtin_fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float u2 = (float)b * 2.3283064365386963e-10f;
    // -2 ln u1 = (-2 ln 2) * log2(u1)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z0 = rad * __builtin_amdgcn_cosf(u2);
    z1 = rad * __builtin_amdgcn_sinf(u2);
}

// The library's own noise stream (BN_NOISE_PHILOX): one Philox block per (instance b, rollout k,
// step pair p) of solve number `solve`, key = seed.  Pair p holds the (v, omega) noise of steps
// 2p and 2p+1.
// FRESH_KEYS: the key words are made opaque at the call, so the round keys (key + r * Weyl constant) are derived by scalar adds at
// every call instead of living in fourteen scalar registers across the caller's loop -- for a kernel whose loop is short of scalar
// registers (the one-wave kernel reloaded eight spilled ones per step pair with v_readlane, a vector instruction each).
template <bool FRESH_KEYS = false>
__device__ __forceinline__ void philox_eps_pair(uint64_t seed, uint64_t solve, uint32_t b, uint32_t k,
                                                uint32_t pair, float e[4])
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (FRESH_KEYS) asm volatile("" : "+s"(k0), "+s"(k1));
    const u32x4 r = philox4x32<kStreamRounds>(u32x4{k, pair, (uint32_t)solve ^ (b << 20), (uint32_t)(solve >> 32) ^ (b >> 12)}, k0, k1);
    box_muller(r.x, r.y, e[0], e[1]);   // step 2*pair: (v, omega) noise
    box_muller(r.z, r.w, e[2], e[3]);   // step 2*pair+1
}

// Slip stream of the sampled-slip mode (BASELINE config 3): own key, same counter layout.  Block j of rollout k
// holds the transit draws of steps 2j, 2j+1 and the cost draws of slots 2j, 2j+1; rollout index 0xffffffff is the
// optimal rollout, whose block j holds the transit draws of steps 4j .. 4j+3.
__device__ __forceinline__ void philox_slip_block(uint64_t seed, uint64_t solve, uint32_t b, uint32_t k, uint32_t j, float z[4])
{
    const u32x4 r = philox4x32<kStreamRounds>(u32x4{k, j, (uint32_t)solve ^ (b << 20), (uint32_t)(solve >> 32) ^ (b >> 12)},
                                              (uint32_t)seed ^ 0x534c4950u, (uint32_t)(seed >> 32));
    box_muller(r.x, r.y, z[0], z[1]);
    box_muller(r.z, r.w, z[2], z[3]);
}

}  // namespace bn
