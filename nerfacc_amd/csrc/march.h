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
#define NFA_HDM __host__ __device__ __forceinline__   /* member functions */
#else
#define NFA_HD static inline
#define NFA_HDM inline
#endif
// reciprocal for ESTIMATES that are verified afterwards (1 ulp hardware rcp on the device)
#if defined(__HIP_DEVICE_COMPILE__)
#define NFA_RCP(x) __builtin_amdgcn_rcpf(x)
#else
#define NFA_RCP(x) (1.0f / (x))
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

// ------------------------------------------------------------------------------------------
// Stepper: the same shortcut with memory.  The stable increment of a binade depends on (dt, binade)
// only -- not on where in the binade the march currently is -- so once it has been observed it can be
// reused by every later march of the same ray in that binade (a ray alternates between skipping and
// emitting a dozen times but crosses a binade boundary once or twice).  Two things are tracked:
//   q_stable / q_binade : the increment observed twice in a row inside binade q_binade;
//   aligned             : the current t was reached by a step with that increment (or by a jump), so the
//                         NEXT step has it too.  In the exact-tie case (dt/ulp = k + 1/2) the first step
//                         after an arbitrary t may differ by one ulp, hence a freshly assigned t or a
//                         new binade is not aligned until one matching step has been observed.
// A jump takes n = min(steps left in the binade, steps for which the loop condition holds, max_steps)
// steps at once; the condition count comes from an fp32 estimate that is then VERIFIED on the actual
// float (the condition is monotone in t), so there is no safety margin and no serial tail: a march is
// one jump plus the final failed test.  Under-estimating n is always safe (the caller loops).
struct Stepper {
    uint32_t q_stable, q_binade;  // q_binade = biased exponent (bits >> 23), 0 = none
    uint32_t obs_q, obs_binade;   // the previous observed in-binade step
    bool aligned;
};
NFA_HD void stepper_init(Stepper &s) { s.q_stable = 0; s.q_binade = 0; s.obs_q = 0; s.obs_binade = 0; s.aligned = false; }
// after t has been assigned a value that is not the result of a step
NFA_HD void stepper_reset(Stepper &s) { s.aligned = false; s.obs_binade = 0; }

// The stable increment of binade e (biased exponent of a positive normal t) without observing it: dt = (k + f) ulp
// exactly (ulp is a power of two), and every in-binade step adds RN(k + f) ulps whatever t is -- unless f == 1/2
// (ties-to-even depends on t's parity for the first step; that case keeps the observation protocol).
NFA_HD bool stepper_align(Stepper &s, uint32_t e, float dt)
{
    if (!(e >= 1u && e < 254u)) return false;
    const float r = ldexpf(dt, 150 - (int)e);  // dt / ulp(t), exact (a power-of-two scaling) when finite
    if (!(r >= 0.5f && r < 8388608.0f)) return false;
    const float k = floorf(r), f = r - k;      // exact: r < 2^23 has at least one fractional bit
    if (f == 0.5f) return false;
    s.q_stable = (uint32_t)k + (f > 0.5f ? 1u : 0u);
    s.q_binade = e;
    s.aligned = true;
    s.obs_q = s.q_stable; s.obs_binade = e;
    return true;
}

// Precondition: t + half < thr (the serial loop would take a step).  Takes n >= 1 steps exactly as the serial
// loop `while (t + half < thr) t += dt` would (never past the point where the condition turns false, never
// more than max_steps >= 1), all with the same exact increment *inc, and returns n; returns 0 (t unchanged)
// when t + dt == t (no progress).
NFA_HD uint32_t stepper_advance(Stepper &s, float &t, float dt, float half, float thr, uint32_t max_steps, float *inc)
{
    const uint32_t bt = f32_bits(t);
    const uint32_t e = bt >> 23;  // sign must be 0 for a jump: e in [1, 254]
    if (!(s.aligned && e == s.q_binade) && max_steps > 1u) stepper_align(s, e, dt);
    if (s.aligned && e == s.q_binade && max_steps > 1u) {
        const uint32_t q = s.q_stable;
        const uint32_t room = (bt | 0x7FFFFFu) - bt;                 // bit patterns left in the binade (Bb - 1 - bt)
        if (q <= room) {
            const float step_val = bits_f32(bt + q) - t;               // exact value of one step
            // steps for which the loop condition holds: t_i + half < thr for i = 0..n-1 (true for i = 0).  An
            // ESTIMATE (fast reciprocal on the device), verified below.
            const float est = ((thr - half) - t) * NFA_RCP(step_val);
            uint32_t n = est < 8388608.0f ? (uint32_t)fmaxf(est, 0.0f) + 2u : 0x800000u;
            if (n > max_steps) n = max_steps;
            // steps that keep the result inside the binade: floor(room / q); only computed when it binds
            if ((uint64_t)n * q > (uint64_t)room) {
                n = (uint32_t)((float)room / (float)q);  // both < 2^24: the quotient is off by at most one
                if (n * q > room) n--;
            }
            // verify the last condition test of the jump; walk back at most 4 steps, else a single step
            int tries = 0;
            while (n > 1u && !(bits_f32(bt + (n - 1u) * q) + half < thr)) {
                n--;
                if (++tries == 4 && n > 1u) { n = 1u; break; }
            }
            if (n >= 1u) {
                t = bits_f32(bt + n * q);
                *inc = step_val;
                s.obs_q = q; s.obs_binade = e;
                return n;
            }
        }
    }
    const float tn = t + dt;
    if (tn == t) return 0u;
    const uint32_t bn = f32_bits(tn);
    if ((bn >> 23) == e && (int32_t)bt > 0 && e != 0u && e < 254u) {  // same binade, positive, normal
        const uint32_t q = bn - bt;
        if (e == s.q_binade && q == s.q_stable) {
            s.aligned = true;
        } else if (s.obs_binade == e && s.obs_q == q) {
            s.q_stable = q; s.q_binade = e; s.aligned = true;
        } else {
            s.aligned = false;
        }
        s.obs_q = q; s.obs_binade = e;
    } else {
        s.aligned = false;
        s.obs_binade = 0;
    }
    *inc = tn - t;  // exact (Sterbenz: tn/2 <= t <= 2 tn whenever the loop runs with dt <= t; otherwise still the value used)
    t = tn;
    return 1u;
}

// ------------------------------------------------------------------------------------------
// Approach table.  Every ray of a batch that starts at the same near plane with the same step walks
// the SAME sequence near, near + dt, ... until it reaches its grid (~650 steps, ~10 binades, for a camera
// 2.2 units away); only where it stops differs.  The table holds, per binade, the first point of that
// sequence at which the stepper is aligned (T, with its stable increment q): a ray looks up the binade
// its target falls into, continues from that point and is one jump away from its stop.  Built on the host
// (it needs only near and dt), passed by value; exact because the stop condition is monotone in t and T
// is a point of the very sequence the serial loop would visit.
constexpr int APPROACH_MAX = 40;
struct ApproachTable {
    uint32_t near_bits;   // bit pattern of the near plane the table was built for
    uint32_t e_lo, n;     // entry i belongs to binade (biased exponent) e_lo + i; n == 0: no table
    float T[APPROACH_MAX];
    uint32_t q[APPROACH_MAX];  // 0 = no entry for this binade
};

NFA_HD void approach_table_build(ApproachTable &tb, float near, float dt)
{
    tb.near_bits = f32_bits(near);
    tb.e_lo = 0; tb.n = 0;
    for (int i = 0; i < APPROACH_MAX; ++i) { tb.T[i] = 0.0f; tb.q[i] = 0u; }
    if (!(dt > 0.0f) || !(near >= 0.0f)) return;
    const float half = dt * 0.5f;
    const float inf = bits_f32(0x7F800000u);
    Stepper s;
    stepper_init(s);
    float t = near;
    for (int it = 0; it < 64 * APPROACH_MAX; ++it) {
        float inc;
        if (stepper_advance(s, t, dt, half, inf, 0xFFFFFFFFu, &inc) == 0u) break;
        const uint32_t e = f32_bits(t) >> 23;
        if (s.aligned && e == s.q_binade) {
            if (tb.n == 0) tb.e_lo = e;
            const uint32_t i = e - tb.e_lo;
            if (i >= (uint32_t)APPROACH_MAX) break;
            if (tb.q[i] == 0u) { tb.T[i] = t; tb.q[i] = s.q_stable; if (i + 1 > tb.n) tb.n = i + 1; }
        }
        if (e >= 127u + 40u) break;
    }
}

// Continue a SKIPPING march that still stands on the near plane from the table: moves t (and the
// stepper) to the furthest tabulated point that the serial loop would pass on its way to thr.
NFA_HD void approach_table_apply(const ApproachTable &tb, Stepper &s, float &t, float half, float thr)
{
    if (tb.n == 0u || f32_bits(t) != tb.near_bits) return;
    const float c = thr - half;
    if (!(c > 0.0f)) return;
    uint32_t ec = f32_bits(c) >> 23;
    if (ec > tb.e_lo + tb.n - 1u) ec = tb.e_lo + tb.n - 1u;  // targets beyond the table: its last binade
    for (int d = 0; d < 2; ++d) {
        const uint32_t e = ec - (uint32_t)d;
        if (e < tb.e_lo || e > ec) return;
        const uint32_t i = e - tb.e_lo;
        const float T = tb.T[i];
        if (tb.q[i] != 0u && T > t && T + half < thr) {
            t = T;
            s.q_stable = tb.q[i]; s.q_binade = e; s.obs_q = tb.q[i]; s.obs_binade = e; s.aligned = true;
            return;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The lattice.  With a constant step every ray that starts at the same near plane stands, at every moment of its march, on
// a point of ONE sequence L[0] = near, L[j + 1] = fl(L[j] + dt): skipping and sampling both move t_last by whole steps
// (grid.cu:159-160, 201-202, 214-245).  The state of a march is therefore an INDEX, and "march while t + dt/2 < thr" is
//      J(thr) = min { j : !(L[j] + dt/2 < thr) }          (the march stands at max(j, J(thr)) afterwards)
// a pure function of the threshold, monotone in it; the samples of an occupied stretch of cells are the lattice points
// between two such indices.  Nothing is carried from one threshold to the next: no stable increment to refresh, no binade
// end to cross, no first march from the near plane -- the cases the marcher of round 2 (walk.hip: marcher_run) serves by
// stopping a lane and re-running the loop, 2.4 trips per list entry on BASELINE cfg 2.
// The table lists the sequence as ROWS: maximal runs of consecutive points inside one binade whose neighbours differ by the
// same number q of ulps (n >= 1 points A, A + q ulp, ...).  Rows are contiguous in j; the step from a row's last point to the
// next row's first has an increment of its own (it crosses a binade end, or is the odd first step of an exact-tie binade).
// Built on the host from (near, dt) alone by running the serial accumulation with exact jumps (the argument of this file's
// header); tests/test_march_cpu.py compares tables and J against the serial loop.
constexpr int LATTICE_MAX_ROWS = 64;
constexpr int LATTICE_MAX_BINADES = 48;
constexpr uint32_t LATTICE_FAIL = 0xFFFFFFFFu;       // lattice_J_fast: not served (the exact search decides)
constexpr uint32_t LATTICE_OFF_TABLE = 0xFFFFFFFEu;  // the march leaves the tabulated part of the sequence
constexpr int LATTICE_ROW_BITS = 6;                   // packed position: index << 6 | row
constexpr uint32_t LATTICE_MAX_INDEX = (1u << (32 - LATTICE_ROW_BITS)) - 2u;
static_assert(LATTICE_MAX_ROWS <= (1 << LATTICE_ROW_BITS), "row index bits");
struct LatticeTable {
    uint32_t near_bits, dt_bits;
    uint32_t n_rows;                          // 0: no table
    uint32_t e_lo, n_binades;                 // main_row[i]: the last row of binade (biased exponent) e_lo + i
    uint32_t j_end;                           // indices [0, j_end) are tabulated
    uint32_t A[LATTICE_MAX_ROWS];             // bit pattern of the row's first point
    uint32_t jA[LATTICE_MAX_ROWS];            // its index
    uint32_t q[LATTICE_MAX_ROWS];             // ulps between neighbours (0: single point)
    uint32_t n[LATTICE_MAX_ROWS];             // points
    uint8_t main_row[LATTICE_MAX_BINADES];    // 0xFF: none
};

NFA_HD void lattice_table_build(LatticeTable &tb, float near, float dt)
{
    tb.near_bits = f32_bits(near); tb.dt_bits = f32_bits(dt);
    tb.n_rows = 0; tb.e_lo = 0; tb.n_binades = 0; tb.j_end = 0;
    for (int i = 0; i < LATTICE_MAX_ROWS; ++i) { tb.A[i] = 0u; tb.jA[i] = 0u; tb.q[i] = 0u; tb.n[i] = 0u; }
    for (int i = 0; i < LATTICE_MAX_BINADES; ++i) tb.main_row[i] = 0xFFu;
    if (!(dt > 0.0f) || !(near >= 0.0f) || !(near < 3.0e38f) || !(dt < 3.0e38f)) return;
    float t = near;
    uint32_t j = 0u;
    int r = 0;
    tb.A[0] = f32_bits(t); tb.jA[0] = 0u; tb.q[0] = 0u; tb.n[0] = 1u;
    int equal_steps = 0;   // consecutive equal in-binade increments that ended at t
    for (;;) {
        const float tn = t + dt;
        if (tn == t || !(tn < 3.0e38f)) break;                       // no progress: the sequence ends here
        const uint32_t bt = f32_bits(t), bn = f32_bits(tn);
        const bool same = (bt >> 23) == (bn >> 23) && (bt >> 23) != 0u;   // same binade, normal
        const uint32_t inc = bn - bt;
        if (same && (tb.n[r] == 1u || inc == tb.q[r])) {
            equal_steps = (tb.n[r] == 1u) ? 1 : equal_steps + 1;
            tb.q[r] = inc; tb.n[r] += 1u; t = tn; j += 1u;
            if (equal_steps >= 2) {
                // two equal in-binade increments in a row: every further step that stays inside the binade adds the same
                // (this file's header); take them all at once
                const uint32_t room = (bn | 0x7FFFFFu) - bn;
                const uint32_t m = room / inc;
                if (m > 0u) {
                    if (j + m > LATTICE_MAX_INDEX) break;
                    tb.n[r] += m; j += m; t = bits_f32(bn + m * inc);
                }
            }
            if (j > LATTICE_MAX_INDEX) break;
            continue;
        }
        // the step leaves the row: tn opens the next one
        if (r + 1 >= LATTICE_MAX_ROWS || j + 1u > LATTICE_MAX_INDEX) break;
        ++r;
        tb.A[r] = bn; tb.jA[r] = j + 1u; tb.q[r] = 0u; tb.n[r] = 1u;
        equal_steps = 0;
        t = tn; j += 1u;
    }
    tb.n_rows = (uint32_t)(r + 1);
    tb.j_end = tb.jA[r] + tb.n[r];
    // the last row of each binade, for the rows of normal numbers
    bool have = false;
    for (int i = 0; i <= r; ++i) {
        const uint32_t e = tb.A[i] >> 23;
        if (e == 0u) continue;
        if (!have) { tb.e_lo = e; have = true; }
        const uint32_t b = e - tb.e_lo;
        if (b < (uint32_t)LATTICE_MAX_BINADES) { tb.main_row[b] = (uint8_t)i; if (b + 1u > tb.n_binades) tb.n_binades = b + 1u; }
    }
}

// point j of the sequence (j < j_end), by its row
NFA_HD float lattice_point_in_row(const LatticeTable &tb, uint32_t r, uint32_t j) { return bits_f32(tb.A[r] + (j - tb.jA[r]) * tb.q[r]); }
NFA_HD float lattice_point(const LatticeTable &tb, uint32_t j)
{
    uint32_t r = 0u;
    while (r + 1u < tb.n_rows && tb.jA[r + 1u] <= j) ++r;
    return lattice_point_in_row(tb, r, j);
}
NFA_HD uint32_t lattice_pack(uint32_t j, uint32_t row) { return (j << LATTICE_ROW_BITS) | row; }

// J(thr) inside ONE row, when the answer is an interior point of it: an fp32 estimate of the step count, then the condition
// itself on three consecutive points (true, ?, false pins the answer; anything else is declined).  rcp = 1 / (q ulp) serves
// the estimate only.  Returns the packed position or LATTICE_FAIL.
NFA_HD uint32_t lattice_J_fast(uint32_t A, uint32_t jA, uint32_t q, uint32_t n, float rcp, uint32_t row, float thr, float half)
{
    const float c = thr - half;
    const float est = (c - bits_f32(A)) * rcp;
    if (!(est >= 1.0f && est < 4194304.0f)) return LATTICE_FAIL;
    const uint32_t a = (uint32_t)est - 1u;                       // floor(est) - 1 >= 0
    if (n < 3u || a > n - 3u) return LATTICE_FAIL;               // the three points a, a + 1, a + 2 must exist
    const uint32_t b0 = A + a * q;                               // (a q < 2^23: a < n, and the row stays inside its binade)
    const bool f0 = bits_f32(b0) + half < thr, f1 = bits_f32(b0 + q) + half < thr, f2 = bits_f32(b0 + 2u * q) + half < thr;
    if (!f0 || f2) return LATTICE_FAIL;
    const uint32_t j = jA + a + 1u + (f1 ? 1u : 0u);
    return lattice_pack(j, row);
}

// J(thr) exactly, for any threshold: the first row whose last point fails the condition (the condition is monotone along
// the sequence), then the first such point inside it.  LATTICE_OFF_TABLE when even the last tabulated point passes.
// ROWS: anything with n_rows() and row(r) -> {A, jA, q, n} (the host table, or the kernels' copy in LDS).
struct LatticeRow { uint32_t A, jA, q, n; };
struct LatticeRowsHost {
    const LatticeTable &tb;
    NFA_HDM uint32_t n_rows() const { return tb.n_rows; }
    NFA_HDM LatticeRow row(uint32_t r) const { return LatticeRow{tb.A[r], tb.jA[r], tb.q[r], tb.n[r]}; }
};
template <class ROWS>
NFA_HD uint32_t lattice_J_search_rows(const ROWS &rows, float thr, float half)
{
    const uint32_t nr = rows.n_rows();
    if (nr == 0u) return LATTICE_OFF_TABLE;
    uint32_t lo = 0u, hi = nr - 1u;
    {
        const LatticeRow z = rows.row(hi);
        if (bits_f32(z.A + (z.n - 1u) * z.q) + half < thr) return LATTICE_OFF_TABLE;
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const LatticeRow z = rows.row(mid);
        if (bits_f32(z.A + (z.n - 1u) * z.q) + half < thr) lo = mid + 1u; else hi = mid;
    }
    const LatticeRow rr = rows.row(lo);
    uint32_t klo = 0u, khi = rr.n - 1u;                          // the condition fails at khi
    while (klo < khi) {
        const uint32_t mid = (klo + khi) >> 1;
        if (bits_f32(rr.A + mid * rr.q) + half < thr) klo = mid + 1u; else khi = mid;
    }
    return lattice_pack(rr.jA + klo, lo);
}
NFA_HD uint32_t lattice_J_search(const LatticeTable &tb, float thr, float half) { return lattice_J_search_rows(LatticeRowsHost{tb}, thr, half); }

// the row whose binade holds thr - dt/2 (the estimate's starting point); rows beyond the binade table: its last entry
NFA_HD uint32_t lattice_main_row(const LatticeTable &tb, float thr, float half)
{
    const int32_t e = (int32_t)f32_bits(thr - half) >> 23;      // (a negative value: below every row)
    int32_t b = e - (int32_t)tb.e_lo;
    if (b < 0) b = 0;
    if (b >= (int32_t)tb.n_binades) b = (int32_t)tb.n_binades - 1;
    return b >= 0 ? (uint32_t)tb.main_row[b] : 0xFFu;
}
NFA_HD uint32_t lattice_J(const LatticeTable &tb, float thr, float half)
{
    const uint32_t r = lattice_main_row(tb, thr, half);
    if (r < tb.n_rows && tb.q[r] != 0u) {
        const float rcp = 1.0f / ldexpf((float)tb.q[r], (int)(tb.A[r] >> 23) - 150);
        const uint32_t w = lattice_J_fast(tb.A[r], tb.jA[r], tb.q[r], tb.n[r], rcp, r, thr, half);
        if (w != LATTICE_FAIL) return w;
    }
    return lattice_J_search(tb, thr, half);
}

// The whole march of one event in one shot, straight-line code for the cases that make up > 99.9 % of the events of a
// ray: t inside a binade whose stable increment is known (or computable), the threshold reached inside this binade
// or inside the next one (one binade boundary crossed).  On success (true) t has advanced by the steps listed in
// seg0 / seg1 / seg2 -- n (possibly 0) steps from t0 with exact increment inc each -- and the loop condition is known to be FALSE at
// the new t; on false NOTHING has changed and the general loop has to do the event (exact ties, a sample budget,
// denormal / huge t, thresholds sitting exactly at a binade end, estimates that were no upper bound).
// Same principle as the jump of stepper_advance: fp32 estimates, then the condition is evaluated on the actual floats.
struct StepSeg { float t0, inc; uint32_t n; };

// One binade of stepper_run_event.  Returns 0: the event is finished (condition false at lt), 1: lt stands on the last
// point of the binade and the condition still holds, 2: give up (the general loop does the event).
NFA_HD int stepper_event_stage(Stepper &ls, float &lt, float dt, float half, float thr, StepSeg &seg)
{
    const uint32_t bt = f32_bits(lt), e = bt >> 23;
    if (!(ls.aligned && e == ls.q_binade) && !stepper_align(ls, e, dt)) return 2;
    if (!(lt + half < thr)) return 0;
    const uint32_t q = ls.q_stable, room = (bt | 0x7FFFFFu) - bt;
    if (q > room) return 1;
    const float step_val = bits_f32(bt + q) - lt;
    const float est = ((thr - half) - lt) * NFA_RCP(step_val);
    const uint32_t n0 = est < 4194304.0f ? (uint32_t)fmaxf(est, 0.0f) + 2u : 0x400002u;
    if ((float)n0 * (float)q < (float)room) {
        // threshold inside this binade: largest m in [1, n0] with cond(m - 1); cond(0) holds
        uint32_t m = n0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 4; ++i)
            if (m > 1u && !(bits_f32(bt + (m - 1u) * q) + half < thr)) m--;
        if (m > 1u && !(bits_f32(bt + (m - 1u) * q) + half < thr)) return 2;  // four probes were not enough
        if (m == n0) return 2;  // the estimate was no upper bound
        seg.t0 = lt; seg.inc = step_val; seg.n = m;
        lt = bits_f32(bt + m * q);  // cond(m - 1) true, cond(m) false (that probe failed)
        return 0;
    }
    // threshold at or beyond the binade's end: all the steps that stay inside it
    uint32_t n_b = (uint32_t)((float)room * NFA_RCP((float)q));  // floor(room / q), estimated with a 1-ulp reciprocal and corrected
    while (n_b * q > room) n_b--;             // (at most two corrections either way: relative error of the
    while ((n_b + 1u) * q <= room) n_b++;     //  estimate < 3e-7, quotient < 2^23)
    if (!(bits_f32(bt + (n_b - 1u) * q) + half < thr)) return 2;  // it sits right at the end: general loop
    seg.t0 = lt; seg.inc = step_val; seg.n = n_b;
    lt = bits_f32(bt + n_b * q);
    return (lt + half < thr) ? 1 : 0;
}

NFA_HD bool stepper_run_event(Stepper &s, float &t, float dt, float half, float thr, StepSeg &seg0, StepSeg &seg1, StepSeg &seg2)
{
    // seg0: steps inside the first binade, seg1: the step across its end, seg2: steps inside the next binade (n == 0: none)
    Stepper ls = s;
    float lt = t;
    seg0.n = seg1.n = seg2.n = 0u;
    seg0.t0 = seg1.t0 = seg2.t0 = 0.0f;
    seg0.inc = seg1.inc = seg2.inc = 0.0f;
    int r = stepper_event_stage(ls, lt, dt, half, thr, seg0);
    if (r == 2) return false;
    if (r == 1) {
        const uint32_t e = f32_bits(lt) >> 23;
        const float tn = lt + dt;  // the step across the boundary (its increment is its own)
        if (tn == lt || (f32_bits(tn) >> 23) == e) return false;
        seg1.t0 = lt; seg1.inc = tn - lt; seg1.n = 1u;
        ls.aligned = false; ls.obs_binade = 0;
        lt = tn;
        r = stepper_event_stage(ls, lt, dt, half, thr, seg2);
        if (r != 0) return false;  // a second binade end inside one event, or give up
    }
    s = ls; t = lt;
    return true;
}

// fast_forward_serial through the stepper (one-shot state): used by tests and by the serial traversal
NFA_HD float fast_forward_stepper(float t, float target, float dt)
{
    const float half = dt * 0.5f;
    Stepper s;
    stepper_init(s);
    for (;;) {
        if (!(t + half < target)) return t;
        float inc;
        if (stepper_advance(s, t, dt, half, target, 0xFFFFFFFFu, &inc) == 0u) return target;
    }
}

// t_last after marching to `target` (step <= 0: jump there, grid.cu:155,198).
NFA_HD float fast_forward(float t_last, float target, float step, float cone_angle)
{
    if (step <= 0.0f) return target;
    const float dt = calc_dt(t_last, cone_angle, step);
    // short skips (a few steps from one cell to the next: the common case inside a grid) stay on the plain loop
    const float half = dt * 0.5f;
    float t = t_last;
    for (int i = 0; i < 6; ++i) {
        if (!(t + half < target)) return t;
        const float tn = t + dt;
        if (tn == t) return target;
        t = tn;
    }
    return fast_forward_exact(t, target, dt);
}

}  // namespace nfa
