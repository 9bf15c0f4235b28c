// sim_math.h -- device arithmetic of sim_spec_v0 (binary32).
//
// Every function is an explicit sequence of IEEE-754 +,-,*,/,sqrt and fma operations (the
// translation unit is compiled with -ffp-contract=off and HIP's default correctly-rounded
// f32 divide/sqrt), so results do not depend on compiler scheduling and can be reproduced
// bit for bit on a CPU.  No v_sin/v_cos/v_rcp approximations: the arguments here are robot
// headings in (-pi-0.2, pi+0.2) and the whole step is a latency-bound dependent chain, so a
// short polynomial on the VALU is both exact-to-spec and as fast as the hardware
// transcendentals plus their range fix-ups.
#pragma once
#include <hip/hip_runtime.h>

namespace rg {

// sin and cos of t, |t| < ~8: quadrant reduction by pi/2 (two-term Cody-Waite) + degree-7/8
// minimax polynomials on [-pi/4, pi/4] (Cephes sinf/cosf coefficients).  |err| < 1e-7.
__device__ __forceinline__ void sincos_spec(float t, float &s, float &c) {
    const float q = __builtin_rintf(t * 0.63661977236758134f);
    float r = __builtin_fmaf(q, -1.57079625129699707031f, t);
    r = __builtin_fmaf(q, -7.54978941586159635335e-08f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(z, sp, -1.6666654611e-1f);
    const float sr = __builtin_fmaf(r * z, sp, r);
    float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(z, cp, 4.166664568298827e-2f);
    const float cr = __builtin_fmaf(z * z, cp, __builtin_fmaf(z, -0.5f, 1.0f));
    const int k = static_cast<int>(q) & 3;
    float ss = (k & 1) ? cr : sr;
    float cc = (k & 1) ? sr : cr;
    if (k == 1 || k == 2) cc = -cc;
    if (k >= 2) ss = -ss;
    s = ss;
    c = cc;
}

// sin/cos of the per-sub-step heading increment dt*w (|.| <= 0.12 rad with the rps constants): Taylor
// polynomials, no range reduction; valid for |t| <= 0.25 (the caller falls back to sincos_spec above)
__device__ __forceinline__ void sincos_small_spec(float t, float &s, float &c) {
    const float z = t * t;
    const float sp = __builtin_fmaf(z, 8.33333377e-3f, -1.66666672e-1f);
    s = __builtin_fmaf(t * z, sp, t);
    float cp = __builtin_fmaf(z, -1.38888892e-3f, 4.16666679e-2f);
    cp = __builtin_fmaf(z, cp, -0.5f);
    c = __builtin_fmaf(z, cp, 1.0f);
}

// rps wraps headings with atan2(sin t, cos t): the identity on (-pi, pi].  The float spec
// subtracts 2*pi (hi + lo) only when |t| exceeds pi (see oracle/oracle_core.h).
__device__ __forceinline__ float wrap_spec(float t) {
    const float dn = (t - 6.283185482025146484375f) - (-1.74845553146951715462e-07f);
    const float up = (t + 6.283185482025146484375f) + (-1.74845553146951715462e-07f);
    const float r = t < -3.1415927410125732421875f ? up : t;   // branch-free: selects, not exec masks
    return t > 3.1415927410125732421875f ? dn : r;
}

__device__ __forceinline__ float norm2_spec(float dx, float dy) { return __builtin_sqrtf(dx * dx + dy * dy); }

__device__ __forceinline__ float clamp_spec(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// natural log for the reset sampler's Box-Muller (Cephes logf), x in (0, 1]
__device__ __forceinline__ float log_spec(float x) {
    int e;
    float m = __builtin_frexpf(x, &e);
    if (m < 0.707106781186547524f) {
        e = e - 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = __builtin_fmaf(m, 7.0376836292e-2f, -1.1514610310e-1f);
    y = __builtin_fmaf(m, y, 1.1676998740e-1f);
    y = __builtin_fmaf(m, y, -1.2420140846e-1f);
    y = __builtin_fmaf(m, y, 1.4249322787e-1f);
    y = __builtin_fmaf(m, y, -1.6668057665e-1f);
    y = __builtin_fmaf(m, y, 2.0000714765e-1f);
    y = __builtin_fmaf(m, y, -2.4999993993e-1f);
    y = __builtin_fmaf(m, y, 3.3333331174e-1f);
    y = (y * m) * z;
    const float fe = static_cast<float>(e);
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// Philox4x32-10 (Salmon et al. 2011): counter-based, so the reset of env e in episode k draws
// the same numbers however envs are sharded over GPUs.
struct Philox {
    uint32_t c[4];
    uint32_t k0, k1;
};
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c[0];
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c[2];
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
    uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0];
    out[1] = c[1];
    out[2] = c[2];
    out[3] = c[3];
}

}  // namespace rg
