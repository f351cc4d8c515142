// march.h -- exact O(#binades) replacement for the reference's serial marching loop.
//
// The reference advances a ray in whole steps with a serial fp32 accumulation
//     while (t + dt*0.5f < target) t += dt;            (cuda/csrc/grid.cu:153-163, :196-205)
// e.g. ~650 dependent adds for a camera 2.2 units away from the grid at step 2*sqrt(3)/1024.
// Sample positions must stay BIT-IDENTICAL to that accumulation (the sample count of a ray
// depends on comparisons of these values), so t0 + k*dt is not an option.
//
// Observation: while t stays inside one binade [2^e, 2^(e+1)) it is a multiple of u = ulp(t),
// and fl(t + dt) = t + RN_u(dt) adds the SAME multiple of u on every step (round-to-nearest of
// dt to a multiple of u does not depend on t, except in the exact-tie case dt/u = k + 1/2, where
// ties-to-even makes the increment constant from the second step on).  So after OBSERVING two
// consecutive equal in-binade increments we may jump n steps ahead with exact arithmetic
// (t + n*inc is a multiple of u below 2^(e+1): exactly representable), as long as every skipped
// step's exact sum stays below the binade's upper bound and the loop condition stays true.
// The function below is the plain serial loop plus that shortcut; it never changes a result.
//
// Compiles as HIP device code and as plain host C++ (tests/test_march_cpu.py builds it with g++).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NFA_HD __host__ __device__ __forceinline__
#else
#define NFA_HD static inline
#endif

namespace nfa {

NFA_HD uint32_t f32_bits(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    return u;
}
NFA_HD float bits_f32(uint32_t u)
{
    float x;
    memcpy(&x, &u, 4);
    return x;
}

NFA_HD float calc_dt(float t, float cone_angle, float step)
{
    return fmaxf(step, fminf(t * cone_angle, 1e10f));  // grid.cu:23-28, utils_math.cuh:1167
}

// Plain serial loop: the semantics (reference + our no-progress guard).
NFA_HD float fast_forward_serial(float t, float target, float dt)
{
    const float half = dt * 0.5f;
    while (t + half < target) {
        const float tn = t + dt;
        if (tn == t) return target;  // no progress: the reference would spin forever
        t = tn;
    }
    return t;
}

// Same result as fast_forward_serial, O(number of binades crossed), integer/fp32 only.
// Inside a binade consecutive floats have consecutive bit patterns, so "t + n*inc" is
// bits(t) + n*q with q the observed bit-pattern increment.
NFA_HD float fast_forward_exact(float t, float target, float dt)
{
    const float half = dt * 0.5f;
    uint32_t prev_q = 0;  // bit-pattern increment of the previous step if it stayed inside its binade
    for (;;) {
        if (!(t + half < target)) return t;
        float tn = t + dt;
        if (tn == t) return target;
        const uint32_t bt = f32_bits(t), bn = f32_bits(tn);
        uint32_t q = 0;
        if ((bt >> 23) == (bn >> 23) && (int32_t)bt > 0 && (bt >> 23) != 0) {  // same binade, positive, normal
            q = bn - bt;
            const uint32_t Bb = (bn | 0x7FFFFFu) + 1u;  // bit pattern of the binade's upper bound 2^(e+1)
            if (q == prev_q && (Bb >> 23) < 255u) {
                // (a) steps i whose exact sum T_(i-1) + dt stays below 2^(e+1): |dt/ulp - q| <= 1/2, so
                //     i*q <= Bb - bn - 1 suffices;  (b) steps for which the loop condition is still true:
                //     the first false one is at n_true >= (target - half - tn)/inc - 1.
                // Both estimates are fp32; the 1e-5 relative and 4-step absolute margins cover their
                // rounding, the tail is marched serially.
                const float nx = (float)(Bb - bn - 1u) / (float)q;
                const float ny = ((target - half) - tn) / (tn - t);
                const float nf = fminf(nx, ny) * 0.99999f - 4.0f;
                if (nf >= 1.0f) tn = bits_f32(bn + (uint32_t)nf * q);
            }
        }
        prev_q = q;
        t = tn;
    }
}

// t_last after marching to `target` (step <= 0: jump there, grid.cu:155,198).
NFA_HD float fast_forward(float t_last, float target, float step, float cone_angle)
{
    if (step <= 0.0f) return target;
    return fast_forward_exact(t_last, target, calc_dt(t_last, cone_angle, step));
}

}  // namespace nfa
