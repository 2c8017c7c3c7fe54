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
__device__ __forceinline__ void sincos_spec(float x, float &sn, float &cs)
{
    const float fn = __builtin_rintf(x * 0.636619772367581343f);
    float r = __builtin_fmaf(-fn, 1.5703125f, x);
    r = __builtin_fmaf(-fn, 4.837512969970703125e-4f, r);
    r = __builtin_fmaf(-fn, 7.54978995489188216e-8f, r);
    const float s = r * r;
    float p = __builtin_fmaf(-1.9515295891e-4f, s, 8.3321608736e-3f);
    p = __builtin_fmaf(p, s, -1.6666654611e-1f);
    const float S = __builtin_fmaf(p * s, r, r);
    float q = __builtin_fmaf(2.443315711809948e-5f, s, -1.388731625493765e-3f);
    q = __builtin_fmaf(q, s, 4.166664568298827e-2f);
    const float hz = 0.5f * s;
    const float w = 1.0f - hz;
    const float C = w + __builtin_fmaf(q * s, s, (1.0f - w) - hz);
    const int n = (int)fn;
    const float a = (n & 1) ? C : S;          // |sin|-side value
    const float b = (n & 1) ? S : C;          // |cos|-side value
    // quadrant signs: sin negative for n&3 in {2,3}; cos negative for n&3 in {1,2}
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -b : b;
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

__device__ __forceinline__ float clampf(float v, float lo, float hi)
{
    return fminf(fmaxf(v, lo), hi);
}

// ((p - origin) / res).floor().int().clamp(0, G-1)    reference grid_map.py:195-209
// POW2: res is a power of two, so multiplying by 1/res is bit-identical to dividing.
template <bool POW2>
__device__ __forceinline__ int cell_index(float p, float origin, float res, float inv_res, int gmax)
{
    const float q = POW2 ? (p - origin) * inv_res : (p - origin) / res;
    const int i = (int)floorf(q);             // v_cvt_i32_f32 saturates
    return min(max(i, 0), gmax);
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

// eps[b][k][t][0..1] for solve number `solve` of the stream keyed by `seed`.
// Counter = (k, t | b<<16 .. , solve_lo, solve_hi): one Philox block per (k, t-pair).
__device__ __forceinline__ void philox_eps_pair(uint64_t seed, uint64_t solve, uint32_t b, uint32_t k,
                                                uint32_t tpair, float e[4])
{
    const u32x4 r = philox4x32_10(u32x4{k, tpair, (uint32_t)solve ^ (b << 20), (uint32_t)(solve >> 32) ^ (b >> 12)},
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller(r.x, r.y, e[0], e[1]);   // step 2*tpair:   (v, omega) noise
    box_muller(r.z, r.w, e[2], e[3]);   // step 2*tpair+1
}

}  // namespace bn
