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
// n = rint(x*2/pi); 3-term Cody-Waite reduction by pi/2 with fma; Cephes single
// precision kernels on [-pi/4, pi/4]; compensated 1 - s/2 for the cosine.
// Max error 1.5 ulp on |x| <= pi + 0.1 (measured against fp64).
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void sincos_spec(float x, float &sn, float &cs)
{
    const float fn = __builtin_rintf(x * 0.636619772367581343f);
    float r = __builtin_fmaf(-fn, 1.5703125f, x);
    r = __builtin_fmaf(-fn, 4.837512969970703125e-4f, r);
    r = __builtin_fmaf(-fn, 7.54978995489188216e-8f, r);
    const float s = r * r;
    // the sine (x) and cosine (y) polynomials share their shape: evaluate them as one packed stream
    const v2f s2 = {s, s};
    v2f pq = __builtin_elementwise_fma(v2f{-1.9515295891e-4f, 2.443315711809948e-5f}, s2,
                                       v2f{8.3321608736e-3f, -1.388731625493765e-3f});
    pq = __builtin_elementwise_fma(pq, s2, v2f{-1.6666654611e-1f, 4.166664568298827e-2f});
    pq = pq * s2;
    const float S = __builtin_fmaf(pq.x, r, r);
    const float hz = 0.5f * s;
    const float w = 1.0f - hz;
    const float C = w + __builtin_fmaf(pq.y, s, (1.0f - w) - hz);
    const int n = (int)fn;
    const bool odd = (n & 1) != 0;
    const float a = odd ? C : S;              // |sin|-side value
    const float b = odd ? S : C;              // |cos|-side value
    // quadrant signs as sign-bit flips: sin negative for n&3 in {2,3}; cos negative for n&3 in {1,2}
    const uint32_t h = (uint32_t)n << 30;     // bit 1 of n -> bit 31
    sn = __uint_as_float(__float_as_uint(a) ^ (h & 0x80000000u));
    cs = __uint_as_float(__float_as_uint(b) ^ ((h + 0x40000000u) & 0x80000000u));
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
__device__ __forceinline__ float wrap_angle_near(float th)
{
    const float a = th + kPi;
    float m = (a >= kTwoPi) ? (a - kTwoPi) : a;      // exact (Sterbenz), == fmod
    m = (a < 0.0f) ? (a + kTwoPi) : m;               // the divisor-sign fix of torch.remainder
    return m - kPi;
}

// min(max(v, lo), hi) in one v_med3_f32 (lo <= hi)
__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    return __builtin_amdgcn_fmed3f(v, lo, hi);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
    return min(max(v, lo), hi);
}

// ---- Philox4x32-10 + Box-Muller ------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// Two independent standard normals from two 32-bit words.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float &z0, float &z1)
{
    const float u1 = (float)(a >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;  // (0,1)
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;                            // [0,1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincos_spec(kTwoPi * u2, sn, cs);
    z0 = rad * cs;
    z1 = rad * sn;
}

// The library's own noise stream (BN_NOISE_PHILOX): one Philox block per (instance b, rollout k,
// step pair p) of solve number `solve`, key = seed.  Pair p holds the (v, omega) noise of steps
// 2p and 2p+1.
__device__ __forceinline__ void philox_eps_pair(uint64_t seed, uint64_t solve, uint32_t b, uint32_t k,
                                                uint32_t pair, float e[4])
{
    const u32x4 r = philox4x32_10(u32x4{k, pair, (uint32_t)solve ^ (b << 20), (uint32_t)(solve >> 32) ^ (b >> 12)},
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller(r.x, r.y, e[0], e[1]);   // step 2*pair: (v, omega) noise
    box_muller(r.z, r.w, e[2], e[3]);   // step 2*pair+1
}

}  // namespace bn
