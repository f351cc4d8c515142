// walk.hip -- the run-length walk of the constant-step traversal (cone_angle == 0): one DDA pass per ray that leaves
// RUN RECORDS (n consecutive samples with one exact fp32 increment) for traverse2.hip's coalesced expansions.
//
// Reference semantics: cuda/csrc/grid.cu:68-282 (kernel), include/utils_grid.cuh:58-142 (setup_traversal,
// single_traversal); the reference walks every ray twice (count + fill) with one scattered 1-byte grid load, a
// per-sample loop and three-way divergence per cell.  The walk moves almost no memory; what it costs is a wave's time
// per cell (instructions of every kind, and the round trips it waits for), so it is built around those:
//
//   phase 1  the cell loop does ONLY the DDA: the three boundary distances (one v_min3, lane masks from sign bits,
//            t += delta & mask), an interleaved cell index that moves by a masked increment and ends the span when the
//            stepped axis' bits reach the end cell's, one 4-byte load per cell from the 1-bit-per-cell copy of the grid --
//            requested TWO steps before it is looked at: the cell sequence does not depend on occupancy --, and one LDS
//            store: the exit distance of a cell goes to the slot of the ray's open list entry, and the slot index moves
//            on when the occupancy flips.  No marching, no branches besides the loop's own.  26 vector instructions.
//   phase 2  a ray's list is a handful of thresholds of alternating kind (skip to / emit to).  Inside one binade every
//            step of the serial accumulation t += dt adds the same number q of ulps (march.h).  With ONE near plane for
//            the launch every ray stands on one lattice of sample positions and an entry is the index J(thr) of a lattice
//            point (lattice_run: no state, no branches in the conversion; samples are index differences); with per-ray near
//            planes the march to a threshold is "the smallest J with fl(t + J q ulp + dt/2) >= thr": an fp32 estimate and
//            four exact probes, straight-line code (marcher_run).  Events that leave the binade (or stand on the near plane,
//            or meet an exact tie) are left for a general path that the whole wave runs together.
//
// Results are bit-identical to the serial accumulation (oracle/nerfacc_oracle.c; tests/test_march_cpu.py checks
// the same marching code against the serial loop on the CPU).
#include "common.hip.h"
#include "march.h"
#include "walk_layout.h"

namespace nfa {

#ifndef NFA_WK_EV
#define NFA_WK_EV 16
#endif
constexpr int WK_EV = NFA_WK_EV;           // list slots per ray: 17 x 4 bytes of LDS per lane
#ifndef NFA_WALK_WAVES
#define NFA_WALK_WAVES 5   /* 100 -> 95 registers (one 8-byte spill outside the loops): 5 waves per SIMD instead of 4 fill a part of the
                              time a slot waits for its next workgroup (scripts/walk_timeline.py): 168 -> 160 us on cfg 2, 384 -> 351 at 256^3 */
#endif
#if NFA_WALK_WAVES > 0
#define NFA_WALK_OCC __attribute__((amdgpu_waves_per_eu(NFA_WALK_WAVES, NFA_WALK_WAVES)))
#else
#define NFA_WALK_OCC
#endif
static_assert(NFA_WK_EV == 16 || NFA_WK_EV == 8, "the list-full test is one bit of the slot address: a power of two");
#ifndef NFA_WALK_LG
#define NFA_WALK_LG 8    /* log2 of the bytes of one list slot = 4 bytes x threads per workgroup: 8 = 64 threads, one wave.  A workgroup's
                            wave slots and LDS are handed on when its LAST wave is done; the waves of a 256-thread workgroup differ by
                            tens of microseconds (wave time stamps, scripts/walk_timeline.py).  256 / 128 / 64 threads: 146.9 / 144.7 /
                            142.0 us on cfg 2, 286.5 / 285.0 / 273.4 at 256^3 (one box) */
#endif
constexpr int WK_LG = NFA_WALK_LG;
constexpr int WK_THREADS = 1 << (WK_LG - 2);
constexpr uint32_t WK_FULL = (uint32_t)WK_EV << WK_LG;  // slot address bit that says "the open entry sits in slot WK_EV"
constexpr uint32_t WK_GUARD = (1u << 9) | (1u << 19) | (1u << 29);
constexpr int WK_MAX_RES = 512;            // 9-bit step counters

// one thread per 32-bit word of the grid copy
__global__ __launch_bounds__(256) void pack_walk_bits_kernel(const uint8_t *__restrict__ binaries, int32_t n_grids, int32_t rx, int32_t ry,
                                                             int32_t rz, WalkLayout L, uint32_t *__restrict__ bits)
{
    const int64_t n_words = ((int64_t)n_grids << L.bits) >> 5;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += (int64_t)blockDim.x * gridDim.x) {
        const uint32_t base = (uint32_t)(wi << 5);
        const uint32_t lvl = base >> L.bits, in_lvl = base & ((1u << L.bits) - 1u);
        uint32_t w = 0u;
        for (uint32_t b = 0; b < 32u; ++b) {
            const uint32_t pidx = in_lvl | b;
            const uint32_t x = bit_extract(pidx, L.mask[0]), y = bit_extract(pidx, L.mask[1]), z = bit_extract(pidx, L.mask[2]);
            const uint32_t rest = pidx & ~(L.mask[0] | L.mask[1] | L.mask[2]);
            if (rest == 0u && x < (uint32_t)rx && y < (uint32_t)ry && z < (uint32_t)rz &&
                binaries[(((int64_t)lvl * rx + x) * ry + y) * rz + z])
                w |= 1u << b;
        }
        bits[wi] = w;
    }
}

struct WalkParams {
    const uint32_t *bits;        // 1 bit per cell in the bit-interleaved order of WalkLayout (nfa_pack_walk_bits)
    WalkLayout lay;
    int32_t *run_cnts;           // [n_rays]
    unsigned long long *runs;    // [max_runs, n_rays] slot-major
    int64_t n_rays;
    int32_t max_runs;
    int32_t *overflow;           // [1] rays with more runs than max_runs
    const int32_t *order;        // lane -> ray assignment or NULL
    int64_t n_order;             // entries of `order` (< n_rays: only the listed rays are walked; the others keep their outputs)
    ApproachTable approach;      // march.h; n == 0: none
    LatticeTable lat;            // march.h: the lattice of the launch's near plane and step (walk_kernel<.., LATTICE = true>)
#ifdef NFA_WALK_STAMPS
    unsigned long long *stamps;  // debugging aid (scripts/walk_timeline.py): [waves][4] = {start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID}
    const int32_t *tile_order;   // experiment: workgroup b walks tile tile_order[b]
#endif
};

enum { WK_EMPTY = 0, WK_OCC = 1, WK_SPAN = 2 };
__device__ __forceinline__ bool is_span_entry(uint32_t ev_span, int32_t k) { return (ev_span >> k) & 1u; }

// Marcher state of one ray (phase 2)
struct Marcher {
    float t_last;
    int32_t continuous;          // 0 / 1
    int32_t n_samples, n_chains, n_runs;
    float run_inc;               // increment of the open run (0: none)
    float span_tmax;             // this_tmax of the span the list entries belong to (thresholds are clamped here)
    int32_t ptype;               // kind of the next cell entry (kinds alternate inside a span)
    int32_t at_near;             // t_last is still the near plane
    // stable increment of the binade t_last is in: fe = biased exponent it was derived for, fq = 0: none (exact tie)
    uint32_t fe, fq;
    float fstep, frcp;
};

__device__ __forceinline__ float vmin_f32(float a, float b)
{
    float m;
    asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));  // fminf costs two canonicalisations more; no NaN reaches it for finite rays
    return m;
}

__device__ __forceinline__ void marcher_refresh(Marcher &s, float dt)
{
    const uint32_t e = f32_bits(s.t_last) >> 23;
    if (e == s.fe) return;
    s.fe = e; s.fq = 0u; s.fstep = 0.f; s.frcp = 0.f;
    if (e >= 1u && e < 254u) {
        const float r = ldexpf(dt, 150 - (int)e);   // dt / ulp(t): exact
        if (r >= 0.5f && r < 8388608.0f) {
            const float k = floorf(r), f = r - k;
            if (f != 0.5f) {                         // an exact tie depends on t's parity: general path
                s.fq = (uint32_t)k + (f > 0.5f ? 1u : 0u);
                s.fstep = ldexpf((float)s.fq, (int)e - 150);  // q ulps, exact
                s.frcp = NFA_RCP(s.fstep);   // for estimates only
            }
        }
    }
}

// A ray's run records go to the slot-major global array runs[slot][ray] (a slot of 32 consecutive rays is one 256-byte line
// for the expansion)
__device__ __forceinline__ void store_run(const WalkParams &p, int32_t slot, int64_t tid, unsigned long long rec)
{
    // (slot < 2^5 and n_rays < 2^31, checked on the host: one v_mad_u64_u32 instead of a 64 x 64-bit product -- two quarter-rate
    //  multiplies and an add3 more per record)
    if (slot < p.max_runs) (p.runs + tid)[(uint64_t)(uint32_t)slot * (uint64_t)(uint32_t)p.n_rays] = rec;
}

// n samples t0, t0 + inc, ... join the ray's run list; a run record {t_first : f32 | k_start : 31, continues_previous : 1}
// is written when a run starts (its length is the next record's k_start, or the ray's count)
__device__ __forceinline__ void marcher_emit(Marcher &s, float t0, float inc, uint32_t n, const WalkParams &p, int64_t tid)
{
    if (!(s.continuous && inc == s.run_inc)) {
        store_run(p, s.n_runs, tid,
                      (unsigned long long)f32_bits(t0) |
                          ((unsigned long long)((uint32_t)s.n_samples | (s.continuous ? 0x80000000u : 0u)) << 32));
        s.n_runs++;
        s.n_chains += s.continuous ? 0 : 1;
        s.run_inc = inc;
    }
    s.n_samples += (int32_t)n;
    s.continuous = 1;
}

// The approach table (march.h) staged in LDS: a per-lane index into a kernel argument would be a waterfall of scalar loads
struct ApproachLds {
    float T[APPROACH_MAX];
    uint32_t q[APPROACH_MAX];
};
// approach_table_apply without the stepper: moves t to the furthest tabulated point of the common sequence near, near + dt, ...
// that the serial loop passes on its way to thr
__device__ __forceinline__ void marcher_approach(const WalkParams &p, const ApproachLds &tb, float &t, float half, float thr)
{
    const uint32_t n = p.approach.n, e_lo = p.approach.e_lo;
    if (n == 0u || f32_bits(t) != p.approach.near_bits) return;
    const float c = thr - half;
    if (!(c > 0.0f)) return;
    uint32_t ec = f32_bits(c) >> 23;
    if (ec > e_lo + n - 1u) ec = e_lo + n - 1u;
    for (int d = 0; d < 2; ++d) {
        const uint32_t e = ec - (uint32_t)d;
        if (e < e_lo || e > ec) return;
        const uint32_t i = e - e_lo;
        const float T = tb.T[i];
        if (tb.q[i] != 0u && T > t && T + half < thr) { t = T; return; }
    }
}

// The marcher's way across the end of a binade, for a lane the lock-step loop could not serve: the first march of the
// ray (the way from the near plane is tabulated), a stale stable increment, a march that reaches the end of the binade.
// Straight-line code; afterwards the lock-step loop looks at the same list entry again (an entry is "march until the
// threshold", so progress never has to be remembered).  Returns 2: progress was made (call again if the loop declines the
// entry again), 1: nothing to do here (if the loop declines again, the general path is next), 0: march.h's general stepper
// has to do the entry.
__device__ __forceinline__ int marcher_cross(Marcher &s, float thr, int type, float dt, float half, int32_t limit, const WalkParams &p,
                                             const ApproachLds &tb, int64_t tid)
{
    int progress = 0;
    if (s.at_near) {
        s.at_near = 0;
        progress = 1;
        if (type != WK_OCC) marcher_approach(p, tb, s.t_last, half, thr);
    }
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {   // (a march across several binades: rays that start near t = 0)
        const uint32_t fe_before = s.fe;
        marcher_refresh(s, dt);
        if (s.fe != fe_before) progress = 1;
        if (s.fq == 0u) return 0;
        const float t = s.t_last;
        if (!(t + half < thr)) break;
        const uint32_t bt = f32_bits(t), q = s.fq;
        const uint32_t room = (bt | 0x7FFFFFu) - bt;          // bit patterns left in the binade
        uint32_t budget = 0xFFFFFFFFu;
        if (type == WK_OCC && limit > 0) {
            if (s.n_samples >= limit) break;
            budget = (uint32_t)(limit - s.n_samples);
        }
        if (q <= room) {
            // all the steps that stay inside the binade, if the condition holds for every one of them
            uint32_t n_b = (uint32_t)((float)room * NFA_RCP((float)q));   // floor(room / q): estimate (both < 2^23), then corrected
            if (n_b * q > room) n_b--;
            if (n_b * q > room) n_b--;
            if ((n_b + 1u) * q <= room) n_b++;
            if ((n_b + 1u) * q <= room) n_b++;
            if (n_b == 0u || n_b * q > room || (n_b + 1u) * q <= room) return 0;
            if (!(bits_f32(bt + (n_b - 1u) * q) + half < thr)) break;      // it stops inside this binade: the lock-step loop's case
            n_b = min(n_b, budget);
            if (type == WK_OCC) marcher_emit(s, t, s.fstep, n_b, p, tid);
            s.t_last = bits_f32(bt + n_b * q);
            progress = 1;
            if (n_b == budget) break;
            budget -= n_b;
        }
        // the step across the end of the binade: its increment is its own
        const float t1 = s.t_last;
        if (!(t1 + half < thr)) break;
        const float tn = t1 + dt;
        if (tn == t1 || (f32_bits(tn) >> 23) == (f32_bits(t1) >> 23)) return 0;
        if (type == WK_OCC) marcher_emit(s, t1, tn - t1, 1u, p, tid);
        s.t_last = tn;
        progress = 1;
    }
    marcher_refresh(s, dt);
    return progress ? 2 : 1;
}

// The same entry through march.h's stepper: binade boundaries, exact ties, the way from the near plane (tabulated),
// denormal / huge distances, no-progress steps.
__device__ __forceinline__ void marcher_general(Marcher &s, float thr, int type, float dt, float half, int32_t limit,
                                                const WalkParams &p, int64_t tid)
{
    Stepper stp;
    stepper_init(stp);
    const bool emit = type == WK_OCC;
    for (;;) {
        if (!(s.t_last + half < thr)) break;
        uint32_t budget = 0xFFFFFFFFu;
        if (emit && limit > 0) {
            if (s.n_samples >= limit) break;
            budget = (uint32_t)(limit - s.n_samples);
        }
        const float t = s.t_last;
        float tn = t, inc;
        const uint32_t n = stepper_advance(stp, tn, dt, half, thr, budget, &inc);
        if (n == 0u) {  // no progress (see oracle): skipping jumps to the target, emission stops
            if (!emit) s.t_last = thr;
            break;
        }
        if (emit) marcher_emit(s, t, inc, n, p, tid);
        s.t_last = tn;
    }
    if (type == WK_EMPTY) s.continuous = 0;
    marcher_refresh(s, dt);
}

// Phase 2: the closed entries [0, cnt) of this lane's list.  ev_span bit k: slots k, k + 1 hold (this_tmin, this_tmax) of a
// span start; bit 16 + k: kind of the first cell entry of that span (slot k + 2).
//
// The lock-step loop serves every entry whose march stays inside the binade of t_last: the steps until the condition
// t + dt/2 < thr fails are estimated in fp32 and the estimate is PROBED on the actual floats (four consecutive step
// counts; the condition is monotone), so the result is the serial loop's.  One code path for the three kinds of entries,
// flags as integers, the only branches are the run-record store and the loop itself.  A lane that cannot be served
// (binade end, first march, no stable increment) stops consuming entries; when no lane can go on, those lanes cross
// together (marcher_cross, or march.h's general stepper) and the loop resumes.
template <bool HAS_LIMIT>
__device__ __forceinline__ void marcher_run(Marcher &s, const char *col /* LDS column of this lane */, int32_t cnt, uint32_t ev_span,
                                            float dt, int32_t limit_arg, const WalkParams &p, const ApproachLds &tb, int64_t tid)
{
    const int32_t limit = HAS_LIMIT ? limit_arg : 0;   // (traverse_steps_limit: the test-mode loop only)
    const float half = dt * 0.5f;
    int32_t k = 0;
    int32_t tried = 0;   // marcher_cross has been run for entry k and the loop declined it again: the general path is next
    for (;;) {
        int32_t blocked = 0;
        float thr = 0.f;
        int32_t type = 0, adv = 1, next_ptype = 0;
        while (k < cnt && !blocked) {
            if (limit > 0 && s.n_samples >= limit) { k = cnt; break; }  // grid.cu:184: nothing moves once the limit is hit
            const float v0 = *reinterpret_cast<const float *>(col + (k << WK_LG));
            const float v1 = *reinterpret_cast<const float *>(col + ((k + 1) << WK_LG));   // (slot k + 1 <= WK_EV exists)
            const int32_t is_span = (int32_t)((ev_span >> k) & 1u);
            const float tmax_k = is_span ? v1 : s.span_tmax;
            thr = is_span ? v0 : vmin_f32(v0, tmax_k);
            type = is_span ? WK_SPAN : s.ptype;
            adv = 1 + is_span;
            next_ptype = is_span ? (int32_t)((ev_span >> (16 + k)) & 1u) : (s.ptype ^ 1);
            const int32_t skip = (type == WK_SPAN) & s.continuous;           // grid.cu:153: `if (!continuous)`
            const float t = s.t_last;
            const int32_t stepping = (int32_t)(t + half < thr) & (skip ^ 1);   // the serial loop would take a step
            // steps until the condition fails: estimate, window of four probes
            const uint32_t bt = f32_bits(t), q = s.fq;
            const float est = ((thr - half) - t) * s.frcp;
            const uint32_t c = (uint32_t)fminf(fmaxf(est, 1.0f), 4194304.0f);  // (NaN -> 1)
            const uint32_t a = c - 1u;
            const uint32_t room = (bt | 0x7FFFFFu) - bt;                      // bit patterns left in the binade
            // (a + 3) q <= room, exactly: products below 2^24 are exact in fp32, larger ones exceed room < 2^23 anyway
            const int32_t fits = (int32_t)((float)(a + 3u) * (float)q <= (float)room) & (int32_t)((bt >> 23) == s.fe) &
                                 (int32_t)(q != 0u) & (s.at_near ^ 1);
            const uint32_t b0 = mad_u24(a, q, bt);                            // (garbage when !fits: unused)
            const int32_t f0 = bits_f32(b0) + half < thr, f1 = bits_f32(b0 + q) + half < thr;
            const int32_t f2 = bits_f32(b0 + 2u * q) + half < thr, f3 = bits_f32(b0 + 3u * q) + half < thr;
            const int32_t ok = fits & f0 & (f3 ^ 1);                          // cond(a) true, cond(a + 3) false: the window holds the answer
            blocked = stepping & (ok ^ 1);
            const int32_t go = stepping & ok;                                 // this lane marches J steps now
            uint32_t J = a + 1u + (uint32_t)f1 + (uint32_t)f2;                // cond(J - 1) true, cond(J) false
            const int32_t emit = go & (int32_t)(type == WK_OCC);
            if (limit > 0 && emit) J = min(J, (uint32_t)(limit - s.n_samples));
            // run records: a new one unless the samples continue the open run (same increment, no gap)
            const int32_t new_run = emit & ((s.continuous & (int32_t)(s.fstep == s.run_inc)) ^ 1);
            if (new_run)
                store_run(p, s.n_runs, tid,
                              (unsigned long long)bt | ((unsigned long long)((uint32_t)s.n_samples | ((uint32_t)s.continuous << 31)) << 32));
            s.n_runs += new_run;
            s.n_chains += new_run & (s.continuous ^ 1);
            s.run_inc = new_run ? s.fstep : s.run_inc;
            s.n_samples += emit ? (int32_t)J : 0;
            s.t_last = go ? bits_f32(bt + J * q) : t;
            const int32_t commit = blocked ^ 1;
            // continuous: set by emitted samples, cleared by an EMPTY entry (whether or not it marched)
            s.continuous = emit ? 1 : ((commit & (int32_t)(type == WK_EMPTY) & (skip ^ 1)) ? 0 : s.continuous);
            k += commit ? adv : 0;
            s.ptype = commit ? next_ptype : s.ptype;
            s.span_tmax = commit ? tmax_k : s.span_tmax;
            tried = commit ? 0 : tried;
        }
        if (!__any(blocked)) break;
        if (blocked) {
            int r = 0;
            if (!tried) r = marcher_cross(s, thr, type, dt, half, limit, p, tb, tid);
            tried = r == 1;
            if (r == 0) {
                marcher_general(s, thr, type, dt, half, limit, p, tid);
                if (is_span_entry(ev_span, k)) s.span_tmax = *reinterpret_cast<const float *>(col + ((k + 1) << WK_LG));
                k += adv; s.ptype = next_ptype;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Phase 2 on the lattice (march.h: LatticeTable).  Every ray of the launch starts at the same near plane and marches with
// the same step, so it always stands on a point of ONE sequence; a list entry "march until the threshold" is the index
// J(thr) of a lattice point, whatever came before it.  Pass A turns the lane's list into such indices, slot by slot, with
// no state: the row of the binade the threshold falls into, an fp32 estimate of the step count, the condition itself on
// three consecutive points (lattice_J_fast); what that declines (answers next to a row end, the first steps behind the
// near plane) goes to the exact search, once per call, for all the lanes that have any.  Pass B walks the list with
// integers: samples of an occupied entry = difference of two indices, run records cut where the increment changes (the
// end of a row: the step across a binade end is a record of its own).
struct LatticeLds {
    uint4 row[LATTICE_MAX_ROWS];          // {A, jA, q, n}
    uint4 main_a[LATTICE_MAX_BINADES];    // the same for the last row of binade e_lo + i (n = 0: none)
    uint2 main_b[LATTICE_MAX_BINADES];    // {bits of 1 / (q ulp), row}
};
struct LatticeRowsLds {
    const LatticeLds &t;
    uint32_t nr;
    __device__ __forceinline__ uint32_t n_rows() const { return nr; }
    __device__ __forceinline__ LatticeRow row(uint32_t r) const { const uint4 v = t.row[r]; return LatticeRow{v.x, v.y, v.z, v.w}; }
};
// position of one ray on the lattice and what its samples add up to
struct LatState {
    uint32_t jr;                 // index << 6 | row
    uint32_t jmax;               // the same for this_tmax of the span the entries belong to
    int32_t cont, ptype;
    int32_t n_samples, n_chains, n_runs;
};
__device__ __forceinline__ float lattice_value(const LatticeLds &lt, uint32_t w)
{
    const uint4 R = lt.row[w & (LATTICE_MAX_ROWS - 1)];
    return bits_f32(R.x + ((w >> LATTICE_ROW_BITS) - R.y) * R.z);
}
// the packed position of index j, which lies in row r or behind it
__device__ __forceinline__ uint32_t lattice_word_of(const LatticeLds &lt, uint32_t j, uint32_t r, uint32_t n_rows)
{
    while (r + 1u < n_rows && j >= lt.row[r].y + lt.row[r].w) ++r;
    return lattice_pack(j, r);
}
// `count` samples from the position w on: one record per stretch of equal increments
__device__ __forceinline__ void lattice_emit(LatState &s, const LatticeLds &lt, uint32_t w, uint32_t count, const WalkParams &p, int64_t tid)
{
    uint32_t j0 = w >> LATTICE_ROW_BITS, r = w & (LATTICE_MAX_ROWS - 1);
    const uint32_t j1 = j0 + count;
    uint32_t k_start = (uint32_t)s.n_samples, cflag = (uint32_t)s.cont;
    do {
        const uint4 R = lt.row[r];
        const uint32_t cross = R.y + R.w - 1u;     // the sample that starts on the row's last point has an increment of its own
        if (j0 < cross) {
            const uint32_t cut = min(j1, cross);
            store_run(p, s.n_runs, tid, (unsigned long long)(R.x + (j0 - R.y) * R.z) | ((unsigned long long)(k_start | (cflag << 31)) << 32));
            s.n_runs++;
            k_start += cut - j0; j0 = cut; cflag = 1u;
        }
        if (j0 < j1) {
            store_run(p, s.n_runs, tid, (unsigned long long)(R.x + (R.w - 1u) * R.z) | ((unsigned long long)(k_start | (cflag << 31)) << 32));
            s.n_runs++;
            k_start += 1u; j0 += 1u; cflag = 1u; ++r;
        }
    } while (j0 < j1);
}

template <bool HAS_LIMIT>
__device__ __forceinline__ void lattice_run(LatState &s, char *col /* LDS column of this lane */, int32_t cnt, uint32_t ev_span, float half,
                                            int32_t limit_arg, const WalkParams &p, const LatticeLds &lt, int64_t tid)
{
    const int32_t limit = HAS_LIMIT ? limit_arg : 0;
    const uint32_t n_rows = p.lat.n_rows;
    const int32_t e_lo = (int32_t)p.lat.e_lo, b_hi = (int32_t)p.lat.n_binades - 1;
    const float near_f = bits_f32(p.lat.near_bits);
    // ---- pass A: thresholds -> lattice positions.  lattice_J_fast (march.h) without a branch, in the instructions that issue
    // at the full rate (common.hip.h): conditions are lane masks from the sign of a difference, selects are v_bitop3.
    uint32_t failed = 0u;
    const float near_half = near_f + half;
    for (int32_t k = 0; k < cnt; ++k) {
        float *slot = reinterpret_cast<float *>(col + (k << WK_LG));
        const float v = *slot;
        const float c = v - half;
        int32_t b = ((int32_t)f32_bits(c) >> 23) - e_lo;            // (a negative value: below every row)
        b = max(0, min(b, b_hi));
        const uint4 ra = lt.main_a[b];                              // {A, jA, q, n}
        const uint2 rb = lt.main_b[b];                              // {1 / (q ulp), row}
        const float est = (c - bits_f32(ra.x)) * bits_f32(rb.x);
        const uint32_t a = (uint32_t)__builtin_amdgcn_fmed3f(est, 0.0f, 4194304.0f) - 1u;   // floor(est) - 1 (est < 1, NaN: wraps, declined below)
        // the three points a, a + 1, a + 2 exist: a <= n - 3 as unsigned numbers (n < 3: n - 3 wraps, so test n too)
        const uint32_t in_row = ~(uint32_t)((int32_t)((ra.w - 3u - a) | a | (ra.w - 3u)) >> 31);
        const uint32_t b0 = mad_u24(a, ra.z, ra.x), b1 = b0 + ra.z, b2 = b1 + ra.z;   // (a q < 2^23 when the row is right: a < n)
        const uint32_t m0 = mask_less(bits_f32(b0) + half, v), m1 = mask_less(bits_f32(b1) + half, v), m2 = mask_less(bits_f32(b2) + half, v);
        const uint32_t ok = in_row & m0 & ~m2;                      // the condition holds at a, fails at a + 2: the answer is a + 1 or a + 2
        const uint32_t j = ra.y + a + 1u - m1;                      // (m1 is 0 or -1)
        const uint32_t w_ok = (j << LATTICE_ROW_BITS) | rb.y;
        const uint32_t stays = ~mask_less(near_half, v);            // no step at all: the ray stays on the near plane, position (0, row 0)
        const uint32_t served = ok | stays;
        const uint32_t w = sel_mask(stays, 0u, w_ok);
        failed |= (~served & 1u) << k;
        *reinterpret_cast<uint32_t *>(slot) = sel_mask(served, w, f32_bits(v));
    }
    if (failed != 0u) {
        const LatticeRowsLds rows{lt, n_rows};
        do {
            const int32_t k = __builtin_ctz(failed);
            failed &= failed - 1u;
            float *slot = reinterpret_cast<float *>(col + (k << WK_LG));
            uint32_t w = lattice_J_search_rows(rows, *slot, half);
            if (w == LATTICE_OFF_TABLE) {   // beyond the tabulated sequence: the caller repeats the traversal without the table
                atomicOr(reinterpret_cast<unsigned int *>(p.overflow + 1), 1u);
                w = lattice_pack(p.lat.j_end - 1u, n_rows - 1u);
            }
            *reinterpret_cast<uint32_t *>(slot) = w;
        } while (failed != 0u);
    }
#ifdef NFA_WALK_NO_PASSB   /* timing experiment: thresholds converted, nothing emitted (wrong results) */
    return;
#endif
    // ---- pass B: the list, in integers (grid.cu:153-163 span start, :193-206 empty cells, :207-262 occupied cells)
    int32_t k = 0;
    while (k < cnt) {
        if (limit > 0 && s.n_samples >= limit) break;                // grid.cu:184: nothing moves once the limit is hit
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(col + (k << WK_LG));
        const uint32_t w1 = *reinterpret_cast<const uint32_t *>(col + ((k + 1) << WK_LG));   // (slot k + 1 <= WK_EV exists)
        const bool is_span = (ev_span >> k) & 1u;
        const uint32_t tgt = is_span ? w0 : min(w0, s.jmax);          // thresholds are clamped to this_tmax: J is monotone
        const int32_t type = is_span ? WK_SPAN : s.ptype;
        const bool skip = is_span && s.cont != 0;                     // grid.cu:153: `if (!continuous)`
        uint32_t target = skip ? s.jr : max(s.jr, tgt);
        if (type == WK_OCC) {
            uint32_t count = (target >> LATTICE_ROW_BITS) - (s.jr >> LATTICE_ROW_BITS);
            if (limit > 0 && count > (uint32_t)(limit - s.n_samples)) {
                count = (uint32_t)(limit - s.n_samples);
                target = lattice_word_of(lt, (s.jr >> LATTICE_ROW_BITS) + count, s.jr & (LATTICE_MAX_ROWS - 1), n_rows);
            }
            if (count > 0u) {
                lattice_emit(s, lt, s.jr, count, p, tid);
                s.n_samples += (int32_t)count;
                s.n_chains += s.cont ^ 1;
                s.cont = 1;
            }
        } else if (type == WK_EMPTY) {
            s.cont = 0;                                               // whether or not it marched
        }
        s.jr = target;
        if (is_span) { s.jmax = w1; s.ptype = (int32_t)((ev_span >> (16 + k)) & 1u); k += 2; }
        else { s.ptype ^= 1; k += 1; }
    }
}

// DDA state of the span being walked
struct WalkSpan {
    float tx, ty, tz, dx, dy, dz;
    uint32_t mx, my, mz;      // bits of each axis in the interleaved cell index (0: the ray does not move along the axis)
    uint32_t rem;             // steps left per axis (9 bits + guard each)
    uint32_t widx;            // interleaved index of the current cell, every axis counted in the ray's direction of travel
    uint32_t end;             // the same for the coordinates (per axis) at which the span is over: start + steps, modulo the axis' range
    uint32_t over;            // 0: the DDA has left the span's last cell (dda_step_lds_end)
    uint32_t flip;            // widx ^ flip = bit index in the grid copy (axes walked downwards reflected, level bits)
};

// setup_traversal (include/utils_grid.cuh:58-114) in the reference's operation order
__device__ __forceinline__ void walk_span_setup(const nfa_traverse_args &a, const WalkLayout &lay, const float o[3], const float d[3],
                                                int32_t level, float this_tmin, float this_tmax, WalkSpan &sp)
{
    const float eps = 1e-6f;
    const float *bmin = a.aabbs + 6 * level, *bmax = bmin + 3;
    float tdist[3], delta[3];
    int32_t cur[3], stepi[3], nst[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float inv = 1.0f / d[ax];
        const float resf = (float)a.res[ax];
        const float extent = bmax[ax] - bmin[ax];
        const float voxel = extent / resf;
        const float ray_start = o[ax] + d[ax] * (this_tmin + eps);
        const float ray_end = o[ax] + d[ax] * (this_tmax - eps);
        int32_t c = (int32_t)(((ray_start - bmin[ax]) / extent) * resf);
        int32_t f = (int32_t)(((ray_end - bmin[ax]) / extent) * resf);
        c = max(0, min(c, a.res[ax] - 1));
        f = max(0, min(f, a.res[ax] - 1));
        const int32_t start_index = c + (d[ax] > 0.0f ? 1 : 0);
        const float tmax_ax = ((bmin[ax] + (((float)start_index * voxel) - ray_start)) * inv) + this_tmin;
        const float step_f = (d[ax] == 0.0f) ? 0.0f : (d[ax] > 0.0f ? 1.0f : -1.0f);
        tdist[ax] = (d[ax] == 0.0f) ? this_tmax : tmax_ax;
        const int32_t st = (int32_t)step_f;
        const float delta_tmp = voxel * inv * step_f;
        delta[ax] = (d[ax] == 0.0f) ? this_tmax : delta_tmp;
        cur[ax] = c;
        stepi[ax] = st;
        // The reference leaves the loop when cur == final + step (utils_grid.cuh:138).  Steps along this axis until then:
        // |f - c| + 1 when the final cell lies in the direction of travel; otherwise that test never fires and the ray
        // would walk out of the grid (undefined upstream): stop where it would leave the grid.
        const int32_t ahead = (f - c) * st;
        int32_t n = ahead + 1;
        if (st == 0) n = (f == c) ? 1 : WK_MAX_RES;
        else if (ahead < 0) n = (st > 0 ? a.res[ax] - 1 - c : c) + 1;
        nst[ax] = max(1, min(n, WK_MAX_RES));
    }
    // (+0: a distance of -0 becomes +0, so that m - t == +0 singles out the minimum in dda_step; -0 < +0 is false in the
    //  reference's comparisons too, the value is the same)
    sp.tx = tdist[0] + 0.0f; sp.ty = tdist[1] + 0.0f; sp.tz = tdist[2] + 0.0f;
    sp.dx = delta[0]; sp.dy = delta[1]; sp.dz = delta[2];
    sp.rem = (uint32_t)(nst[0] - 1) | ((uint32_t)(nst[1] - 1) << 10) | ((uint32_t)(nst[2] - 1) << 20) | WK_GUARD;
    // an axis walked downwards counts its reflected coordinate (2^nb - 1 - c = c ^ (2^nb - 1)) upwards: every step is "+1"
    uint32_t widx = 0u, flip = (uint32_t)level << lay.bits, mk[3], end = 0u;
    const bool regular = walk_layout_regular(lay);   // (wave-uniform: the layout is a kernel argument)
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const uint32_t M = lay.mask[ax];
        // (regular: the three masks are the plain rotation z, x, y -- walk_layout.h -- and axis ax sits at bit (ax + 1) % 3)
        const uint32_t dep = regular ? (spread_by_3((uint32_t)cur[ax]) << ((ax + 1) % 3)) : bit_deposit((uint32_t)cur[ax], M);
        const uint32_t w_ax = stepi[ax] < 0 ? (dep ^ M) : dep;
        widx |= w_ax;
        flip |= stepi[ax] < 0 ? M : 0u;
        mk[ax] = stepi[ax] != 0 ? M : 0u;
        // the coordinate nst steps of +1 further, modulo the axis' range (dda_step_lds_end; nst <= 2^bits, so the first time
        // the axis' bits agree with these is after exactly nst steps; the deposit drops the bits beyond the mask's)
        const uint32_t c_max = (1u << __builtin_popcount(M)) - 1u;
        const uint32_t c_ref = stepi[ax] < 0 ? c_max - (uint32_t)cur[ax] : (uint32_t)cur[ax];
        const uint32_t c_end = (c_ref + (uint32_t)nst[ax]) & c_max;
        end |= regular ? (spread_by_3(c_end) << ((ax + 1) % 3)) : bit_deposit(c_end, M);
    }
    sp.mx = mk[0]; sp.my = mk[1]; sp.mz = mk[2];
    sp.widx = widx; sp.flip = flip; sp.end = end;
}

// single_traversal (include/utils_grid.cuh:116-142) on the walk's state; the step functions below return the exit distance m
// of the cell the ray leaves.  The reference steps x if tx < ty && tx < tz, else y if ty < tz, else z: that is "z if tz is
// the minimum, else y if ty is, else x" (ties go z over y over x either way), and the chosen axis' distance IS m, so its
// update is m + delta.  Written against the issue rates of common.hip.h: ONE half-rate instruction (v_min3); "is the
// minimum" comes from the sign of m - t, every select is a v_bitop3 on those masks (the first form -- two v_min, two v_cmp,
// ten v_cndmask, two v_bfi -- was 15 half-rate + 15 full-rate instructions per cell).  The distances are never -0
// (walk_span_setup adds +0), so m - t is +0 exactly for the minimum.
//
// The same step with the per-axis constants in LDS (the constant-step walk's cell loop): row a of the lane's table holds
// {axis a's delta in component a, +0 in the others; the axis' bits of the interleaved index}, so the chosen axis' row IS the
// update of the three distances (t + 0 = t exactly: the distances are never -0) and no constant is selected in registers:
// 5 half-rate + 22 full-rate instructions and one 16-byte LDS read per cell (registers: 5 + 27, and six registers more).
// Rows of 16 bytes, lanes 16 bytes apart, axes 4096 bytes apart: a quad of lanes reads four different bank groups
// whichever rows its lanes pick.
#ifndef NFA_CONE_THREADS
#define NFA_CONE_THREADS 256   /* lanes per workgroup of the cone-angle kernels (their LDS tables are laid out for that many).  Unlike the
                                  constant-step walk they gain nothing from smaller workgroups (their rays are binned by length, their waves own
                                  chunks): 256 / 128 / 64 lanes: cfg 5's count pass 7.20 / 7.23 / 7.26 ms, its test-mode image 98.8 / 101.4 / 100.8 ms */
#endif
constexpr int CONE_THREADS = NFA_CONE_THREADS;
constexpr uint32_t WK_TAB_AXIS = CONE_THREADS * 16;   // (the cone kernels' workgroups, whatever the constant-step walk's are)
static_assert((WK_TAB_AXIS & (WK_TAB_AXIS - 1)) == 0, "the row address is formed by OR-ing the axis offset into the lane's");
__device__ __forceinline__ float dda_step_lds(const char *lds, uint32_t ax /* the lane's x row */, uint32_t az /* its z row */, float &tx,
                                              float &ty, float &tz, uint32_t &rem, uint32_t &widx)
{
    const float m = min3_f32(tx, ty, tz);
    const uint32_t kz = mask_less(m, tz), ky = mask_less(m, ty);   // ~0: that axis is NOT the minimum
    uint32_t addr, dec;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xce" : "=v"(addr) : "v"(ky), "v"(ax), "s"(WK_TAB_AXIS));   // ky ? x row : y row  (b | (~a & c))
    addr = sel_mask(kz, addr, az);
    const nfa_v4f row = *reinterpret_cast<const nfa_v4f *>(lds + addr);
    // (plain adds: left to itself the compiler pairs two of them into a half-rate v_pk_add_f32 behind two moves)
    asm("v_add_f32 %0, %0, %1" : "+v"(tx) : "v"(row.x));
    asm("v_add_f32 %0, %0, %1" : "+v"(ty) : "v"(row.y));
    asm("v_add_f32 %0, %0, %1" : "+v"(tz) : "v"(row.z));
    asm("v_bitop3_b32 %0, %1, 1, %2 bitop3:0xca" : "=v"(dec) : "v"(ky), "s"(1u << 10));          // ky ? 1 : 1 << 10
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(dec) : "v"(kz), "v"(dec), "s"(1u << 20));  // kz ? that : 1 << 20
    rem -= dec;
    const uint32_t M = f32_bits(row.w);
    uint32_t filled;
    asm("v_bitop3_b32 %0, %1, %2, %2 bitop3:0xcf" : "=v"(filled) : "v"(M), "v"(widx));   // (M & widx) | ~M
    filled += 1u;
    widx = sel_mask(M, filled, widx);                                                   // (M & sum) | (~M & widx)
    return m;
}

// The constant-step walk's form of the same step: no step counters.  The span is over when the coordinate of the axis just
// stepped reaches the one behind the span's last cell (utils_grid.cuh:138: `cur == final + step`); with every axis counted
// in the direction of travel that is "the axis' bits of widx equal those of `end`" (walk_span_setup), one v_bitop3 on the
// mask the row holds anyway: `over` = 0 when the span has ended (an axis that cannot move has no bits: the first step
// along it ends the span, as a counter of one step did).  3 full-rate instructions less than counters + guard bits.
__device__ __forceinline__ float dda_step_lds_end(const char *lds, uint32_t ax /* the lane's x row */, uint32_t az /* its z row */, float &tx,
                                                  float &ty, float &tz, uint32_t end, uint32_t &widx, uint32_t &over)
{
    const float m = min3_f32(tx, ty, tz);
    const uint32_t kz = (uint32_t)((int32_t)f32_bits(m - tz) >> 31), ky = (uint32_t)((int32_t)f32_bits(m - ty) >> 31);   // ~0: that axis is NOT the minimum
    const uint32_t addr = __builtin_amdgcn_bitop3_b32(kz, __builtin_amdgcn_bitop3_b32(ky, ax, WK_TAB_AXIS, 0xce), az, 0xca);   // kz ? (ky ? x row : y row) : z row
    const nfa_v4f row = *reinterpret_cast<const nfa_v4f *>(lds + addr);
    asm("v_add_f32 %0, %0, %1" : "+v"(tx) : "v"(row.x));
    asm("v_add_f32 %0, %0, %1" : "+v"(ty) : "v"(row.y));
    asm("v_add_f32 %0, %0, %1" : "+v"(tz) : "v"(row.z));
    const uint32_t M = f32_bits(row.w);
    const uint32_t filled = __builtin_amdgcn_bitop3_b32(M, widx, widx, 0xcf) + 1u;   // ((M & widx) | ~M) + 1
    widx = __builtin_amdgcn_bitop3_b32(M, filled, widx, 0xca);                       // (M & sum) | (~M & widx)
    over = __builtin_amdgcn_bitop3_b32(widx, end, M, 0x28);                          // (widx ^ end) & M
    return m;
}

// The same step with the per-axis constants in registers and no table: the distances are updated as t += delta & mask, so no
// LDS read stands between one cell's minimum and the next one's (26 vector instructions instead of 23 and a read).
__device__ __forceinline__ float dda_step_reg_end(float dx, float dy, float dz, uint32_t mx, uint32_t my, uint32_t mz, float &tx, float &ty,
                                                  float &tz, uint32_t end, uint32_t &widx, uint32_t &over)
{
    // (compiler builtins where there is one: between two `asm` statements the compiler puts an s_nop whenever the second reads
    //  what the first wrote -- it cannot know that the first is no transcendental -- and seven of those per cell cost as much
    //  as seven instructions at this occupancy)
    const float m = min3_f32(tx, ty, tz);
    const uint32_t kz = (uint32_t)((int32_t)f32_bits(m - tz) >> 31), ky = (uint32_t)((int32_t)f32_bits(m - ty) >> 31);   // ~0: that axis is NOT the minimum
    const float ix = bits_f32(__builtin_amdgcn_bitop3_b32(f32_bits(dx), kz, ky, 0x80));   // dx & kz & ky
    const float iy = bits_f32(__builtin_amdgcn_bitop3_b32(f32_bits(dy), kz, ky, 0x40));   // dy & kz & ~ky
    const float iz = bits_f32(__builtin_amdgcn_bitop3_b32(f32_bits(dz), kz, kz, 0x30));   // dz & ~kz
    asm("v_add_f32 %0, %0, %1" : "+v"(tx) : "v"(ix));   // (t + 0 = t exactly: the distances are never -0; plain adds would be paired into a half-rate v_pk_add_f32)
    asm("v_add_f32 %0, %0, %1" : "+v"(ty) : "v"(iy));
    asm("v_add_f32 %0, %0, %1" : "+v"(tz) : "v"(iz));
    const uint32_t M = __builtin_amdgcn_bitop3_b32(kz, __builtin_amdgcn_bitop3_b32(ky, mx, my, 0xca), mz, 0xca);
    const uint32_t filled = __builtin_amdgcn_bitop3_b32(M, widx, widx, 0xcf) + 1u;   // ((M & widx) | ~M) + 1: the carry runs up to the axis' lowest bit
    widx = __builtin_amdgcn_bitop3_b32(M, filled, widx, 0xca);
    over = __builtin_amdgcn_bitop3_b32(widx, end, M, 0x28);                          // (widx ^ end) & M
    return m;
}

// the lane's three rows of the table, from the span's DDA state
__device__ __forceinline__ void dda_table_write(char *tab_lds, const WalkSpan &sp)
{
    char *row = tab_lds + 16u * threadIdx.x;
    *reinterpret_cast<nfa_v4f *>(row) = nfa_v4f{sp.dx, 0.0f, 0.0f, bits_f32(sp.mx)};
    *reinterpret_cast<nfa_v4f *>(row + WK_TAB_AXIS) = nfa_v4f{0.0f, sp.dy, 0.0f, bits_f32(sp.my)};
    *reinterpret_cast<nfa_v4f *>(row + 2 * WK_TAB_AXIS) = nfa_v4f{0.0f, 0.0f, sp.dz, bits_f32(sp.mz)};
}

// One cell of the walk in three pieces, because they run one cell apart (walk_ray's cell loop):
//   dda_step_lds_end  steps the DDA out of the cell the ray is in and returns that cell's exit distance;
//   walk_request      requests the word of the grid copy that holds the occupancy bit of the cell the ray is in now
//                     ((i, w): the bit's index and the word);
//   walk_record       looks at a cell's bit once its word has arrived, closes the ray's open list entry when the occupancy
//                     flips and records the cell's exit distance in the open entry's slot (`open`: the kind of the open entry).
// The cell sequence does not depend on the occupancy, so the DDA may run ahead: a word is requested TWO steps before it is
// looked at (the request used to be one cell old and the wave waited for it in every cell: 43 % of a wave's life was spent in
// s_waitcnt, SQ_WAIT_ANY).
// The request is written as an instruction the compiler does not track and is waited for by hand (walk_arrived): its own
// wait-count pass has to assume that a wave may go from the loop's first half straight to the next trip (the lanes leave the
// loop one by one, so the structured loop has that edge) and would wait for EVERY request at the first use of a word, the one
// made half a trip ago included.  Requests return in order: "at most one in flight" means the older word has arrived.
__device__ __forceinline__ void walk_request(uint32_t widx, uint32_t flip, const uint32_t *__restrict__ bits, uint32_t &i, uint32_t &w)
{
    i = widx ^ flip;
#if defined(NFA_WALK_EXP) && NFA_WALK_EXP == 2  /* timing experiment: every load hits one 128-byte line */
    const uint32_t off = ((i >> 5) & 31u) << 2;
#else
    const uint32_t off = (i >> 5) << 2;
#endif
    asm volatile("global_load_dword %0, %1, %2" : "=v"(w) : "v"(off), "s"(bits) : "memory");
}
// the older of the two words in flight has arrived / both have
// (`after`: a value the wait is to follow in the instruction order -- left alone it is scheduled to the top of the cell)
__device__ __forceinline__ void walk_arrived(uint32_t &w, uint32_t after) { asm volatile("s_waitcnt vmcnt(1)" : "+v"(w) : "v"(after) : "memory"); }
__device__ __forceinline__ void walk_arrived_all(uint32_t &w0, uint32_t &w1) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0), "+v"(w1) : : "memory"); }
__device__ __forceinline__ void walk_record(float m, uint32_t w, uint32_t i, int32_t &open, uint32_t &ev_addr, char *ev_lds)
{
    const uint32_t changed = __builtin_amdgcn_ubfe(w, i, 1u) ^ (uint32_t)open;   // bit (i & 31) of the cell's word
    open ^= (int32_t)changed;                    // = the cell's occupancy
    ev_addr += changed << WK_LG;                 // the open entry is complete when the occupancy flips
    *reinterpret_cast<float *>(ev_lds + ev_addr) = m;
}
// stop when the span is over (dda_step_lds_end) or the open entry sits in the last slot
__device__ __forceinline__ bool walk_stop(uint32_t over, uint32_t ev_addr)
{
    return (over == 0u) | (ev_addr >= WK_FULL);
}

// The approach table of the launch, read from the kernel-argument segment as memory (indexed by thread: as an argument in
// registers it would occupy 80 scalar registers and be selected entry by entry).  The offset of the second by-value
// argument is the ABI's: arguments are laid out in order, each at its natural alignment.
static_assert(alignof(WalkParams) == 8 && sizeof(nfa_traverse_args) % 8 == 0, "walk kernels: WalkParams follows nfa_traverse_args in the kernarg segment");
__device__ __forceinline__ void approach_to_lds(ApproachLds &tb, const WalkParams &p)
{
    const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t p_off = (sizeof(nfa_traverse_args) + alignof(WalkParams) - 1) / alignof(WalkParams) * alignof(WalkParams);
    const WalkParams *pk = reinterpret_cast<const WalkParams *>(ka + p_off);
    if (threadIdx.x < APPROACH_MAX) { tb.T[threadIdx.x] = pk->approach.T[threadIdx.x]; tb.q[threadIdx.x] = pk->approach.q[threadIdx.x]; }
#ifdef NFA_WALK_DEBUG
    if (threadIdx.x == 0 && (pk->n_rays != p.n_rays || pk->approach.n != p.approach.n)) __builtin_trap();
#endif
}

// The lattice table of the launch, staged the same way; the rows of the binade table with what lattice_J_fast needs
__device__ __forceinline__ void lattice_to_lds(LatticeLds &lt)
{
    const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t p_off = (sizeof(nfa_traverse_args) + alignof(WalkParams) - 1) / alignof(WalkParams) * alignof(WalkParams);
    const LatticeTable *tk = &reinterpret_cast<const WalkParams *>(ka + p_off)->lat;
    const uint32_t t = threadIdx.x;
    if (t < (uint32_t)LATTICE_MAX_ROWS) lt.row[t] = make_uint4(tk->A[t], tk->jA[t], tk->q[t], tk->n[t]);
    if (t < (uint32_t)LATTICE_MAX_BINADES) {
        const uint32_t r = tk->main_row[t];
        uint4 a = make_uint4(0u, 0u, 0u, 0u);
        uint2 b = make_uint2(0u, 0u);
        if (r < (uint32_t)LATTICE_MAX_ROWS && tk->q[r] != 0u) {
            a = make_uint4(tk->A[r], tk->jA[r], tk->q[r], tk->n[r]);
            b = make_uint2(f32_bits(NFA_RCP(ldexpf((float)a.z, (int)(a.x >> 23) - 150))), r);
        }
        lt.main_a[t] = a; lt.main_b[t] = b;
    }
}

// One ray through the grid(s): phases 1 and 2 alternate until the event walk is over.  Leaves the marcher's state (sample
// count, run count, chain count, last distance); run records go to store_run.
struct WalkOut {
    float t_last;
    int32_t n_samples, n_chains, n_runs;
};
#define WALK_DDA(tx_, ty_, tz_, widx_, over_) dda_step_reg_end(sp.dx, sp.dy, sp.dz, sp.mx, sp.my, sp.mz, tx_, ty_, tz_, sp.end, widx_, over_)
template <bool FUSED, bool HAS_LIMIT, bool LATTICE>
__device__ __forceinline__ void walk_ray(const nfa_traverse_args &a, const WalkParams &p, int64_t tid, char *ev_lds,
                                         uint32_t lane_off, const ApproachLds &tb, const LatticeLds &lt, int32_t steps_limit, WalkOut &out)
{
    Marcher s;
    LatState ls;
    char *const col = ev_lds + lane_off;
    const float dt = a.step_size;
    const int32_t limit = HAS_LIMIT ? steps_limit : 0;
    const uint32_t *__restrict__ bits = p.bits;
    const float near_plane = a.near_planes[tid], far_plane = a.far_planes[tid];
    const float o[3] = {a.rays_o[3 * tid], a.rays_o[3 * tid + 1], a.rays_o[3 * tid + 2]};
    const float d[3] = {a.rays_d[3 * tid], a.rays_d[3 * tid + 1], a.rays_d[3 * tid + 2]};
    s.t_last = near_plane; s.continuous = 0; s.n_samples = 0; s.n_chains = 0; s.n_runs = 0; s.run_inc = 0.f;
    s.span_tmax = 0.f; s.ptype = 0; s.at_near = 1; s.fe = 0xFFFFFFFFu; s.fq = 0u; s.fstep = 0.f; s.frcp = 0.f;
    ls.jr = lattice_pack(0u, 0u); ls.jmax = 0u; ls.cont = 0; ls.ptype = 0; ls.n_samples = 0; ls.n_chains = 0; ls.n_runs = 0;

    float f_tmin = 0.f, f_tmax = 0.f;
    bool f_pending = false;
    if (FUSED) {  // slab test (include/utils_grid.cuh:10-55) with near = -inf, far = +inf (grid.py:158)
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        float tmin, tmax, lo, hi;
        bool hit = true;
        const float *bmin = a.aabbs, *bmax = a.aabbs + 3;
        if (inv[0] >= 0) { tmin = (bmin[0] - o[0]) * inv[0]; tmax = (bmax[0] - o[0]) * inv[0]; }
        else             { tmin = (bmax[0] - o[0]) * inv[0]; tmax = (bmin[0] - o[0]) * inv[0]; }
#pragma unroll
        for (int ax = 1; ax < 3; ++ax) {
            if (inv[ax] >= 0) { lo = (bmin[ax] - o[ax]) * inv[ax]; hi = (bmax[ax] - o[ax]) * inv[ax]; }
            else              { lo = (bmax[ax] - o[ax]) * inv[ax]; hi = (bmin[ax] - o[ax]) * inv[ax]; }
            if (tmin > hi || lo > tmax) hit = false;
            if (lo > tmin) tmin = lo;
            if (hi < tmax) tmax = hi;
        }
        if (tmax <= 0) hit = false;
        f_tmin = fmaxf(tmin, near_plane); f_tmax = fminf(tmax, far_plane);
        f_pending = hit && f_tmin < f_tmax;
    }
    const int32_t G = a.n_grids;
    int32_t next_i = 0;  // next entry of the event walk over the sorted intersections (non-fused)
    // A ray with a non-finite origin or direction has no geometry: upstream its NaN planes survive fmaxf / fminf as
    // [near, far] and the ray is sampled all the way to the far plane (1e10 by default).  Here it gets no samples.
    if (!(isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(d[0]) && isfinite(d[1]) && isfinite(d[2]))) {
        f_pending = false;
        next_i = 2 * G;
    }
    if (LATTICE && f32_bits(near_plane) != p.lat.near_bits) {
        // not a ray of this lattice (near_hint did not hold for it): the caller repeats the traversal without the table
        atomicOr(reinterpret_cast<unsigned int *>(p.overflow + 1), 1u);
        f_pending = false;
        next_i = 2 * G;
    }

    WalkSpan sp;
    sp.tx = sp.ty = sp.tz = sp.dx = sp.dy = sp.dz = 0.f; sp.mx = sp.my = sp.mz = 0u; sp.rem = 0u; sp.widx = 0u; sp.flip = 0u; sp.end = 0u; sp.over = 0u;
    int32_t in_span = 0, has_open = 0, open_type = 0;
    // the two cells in flight (see the cell loop): P has been stepped out of (exit distance m_p) and waits for its word, Q is
    // the cell the ray is in; (i, w): the index of a cell's occupancy bit and the word of the grid copy that holds it
    float m_p = 0.f, m_q = 0.f;
    uint32_t i_p = 0u, w_p = 0u, i_q = 0u, w_q = 0u;
    int32_t par = 0;              // 1: the roles of P and Q are exchanged (the cell loop was left after its first half)
    uint32_t ev_addr = lane_off;  // byte offset of the open entry's slot: slot << 10 | lane offset
    uint32_t ev_span = 0u;
    float m_last = 0.f;

    for (;;) {
        // ---------------- phase 1
        int32_t finished = 0;
        for (;;) {
            if (!in_span) {
                const uint32_t used = (ev_addr >> WK_LG) + (uint32_t)has_open;
                if (used > (uint32_t)(WK_EV - 2)) break;  // a span start needs two slots and one for its first entry: flush first
                float this_tmin = 0.f, this_tmax = 0.f;
                int32_t level = 0;
                bool found = false;
                if (FUSED) {
                    found = f_pending; f_pending = false;
                    this_tmin = f_tmin; this_tmax = f_tmax;
                } else {  // grid.cu:125-150
                    const uint8_t *hits = a.hits + tid * G;
                    const float *ts = a.t_sorted + tid * 2 * G;
                    const int64_t *ti = a.t_indices + tid * 2 * G;
                    while (!found && next_i < 2 * G - 1) {
                        const int32_t i = next_i++;
                        const int64_t idx = ti[i];
                        level = event_level(idx, G);
                        if ((uint32_t)level >= (uint32_t)G || !hits[level]) continue;
                        if (!(idx < G)) {
                            const int64_t nidx = ti[i + 1];
                            if (nidx < G) continue;
                            level = event_level(nidx, G);
                            if ((uint32_t)level >= (uint32_t)G || !hits[level]) continue;
                        }
                        this_tmin = fmaxf(ts[i], near_plane); this_tmax = fminf(ts[i + 1], far_plane);
                        if (this_tmin >= this_tmax) continue;
                        found = true;
                    }
                }
                if (!found) { finished = 1; break; }
                // the open entry of the previous span is complete; then the span start: (this_tmin, this_tmax)
                uint32_t kk = used;
                *reinterpret_cast<float *>(col + (kk << WK_LG)) = this_tmin;
                *reinterpret_cast<float *>(col + ((kk + 1u) << WK_LG)) = this_tmax;
                ev_span |= 1u << kk;
                walk_span_setup(a, p.lay, o, d, level, this_tmin, this_tmax, sp);
                const uint32_t idx0 = sp.widx ^ sp.flip;
                w_p = bits[idx0 >> 5]; i_p = idx0;
                open_type = (int32_t)((w_p >> (idx0 & 31u)) & 1u);
                ev_span |= (uint32_t)open_type << (16u + kk);
                ev_addr = ((kk + 2u) << WK_LG) | lane_off;
                has_open = 1;
                in_span = 1;
                // the first cell's step: its exit distance, the request for the second cell's word
                __builtin_amdgcn_s_waitcnt(0xC07F);
                m_p = WALK_DDA(sp.tx, sp.ty, sp.tz, sp.widx, sp.over);
                walk_request(sp.widx, sp.flip, bits, i_q, w_q);
                par = 0;
            }
            // The reference's cell loop (grid.cu:184-272) reduced to the DDA.  At its head: cell P (m_p, i_p, w_p) has been
            // stepped out of but not recorded, cell Q (i_q, w_q) is the one the ray is in, its word requested.  Two cells per
            // trip, P's and Q's registers trading roles, so that nothing is moved from one to the other.
            float tx = sp.tx, ty = sp.ty, tz = sp.tz;
            const uint32_t flip = sp.flip;
            uint32_t over = sp.over;
            uint32_t widx = sp.widx;
            // Nothing in the cell loop but the table read uses LDS results or scalar memory.  Without this the compiler's wait-count
            // pass, which merges the loop header's state with the preheader's, puts an `s_waitcnt lgkmcnt(0)` in front of every
            // ds_write whenever some path into the loop leaves an LDS read or a kernel-argument load in flight.
            // The same pass decides how many requests may stay in flight at the loop's first use of a word from ALL the paths into
            // the loop (behind phase 2 its stores are in flight too) and settles for "none": the request made half a trip ago
            // would be waited for.  With nothing in flight at the loop's head it finds vmcnt(1) in both halves.
            // vmcnt(0) lgkmcnt(0), expcnt untouched:
            __builtin_amdgcn_s_waitcnt(0x0070);
#if defined(NFA_WALK_EXP) && NFA_WALK_EXP == 5   /* timing experiment: everything but the cell loop */
            over = 0u;
#else
            if (over != 0u) {
                // (`par`: whose registers hold the unrecorded cell when the loop is left -- the lanes leave it one by one and
                //  nothing is moved on the way out: a move of a register with a request in flight waits for the request.  A lane
                //  that left after the first half resumes with the second.)
                bool go = true;
                if (par) {
                    m_p = WALK_DDA(tx, ty, tz, widx, over);
                    walk_arrived(w_q, over);
                    walk_record(m_q, w_q, i_q, open_type, ev_addr, ev_lds);
                    walk_request(widx, flip, bits, i_q, w_q);
                    par = 0;
                    go = !walk_stop(over, ev_addr);
                }
                if (go) for (;;) {
                    m_q = WALK_DDA(tx, ty, tz, widx, over);   // out of Q
                    walk_arrived(w_p, over);
                    walk_record(m_p, w_p, i_p, open_type, ev_addr, ev_lds);
                    walk_request(widx, flip, bits, i_p, w_p);                               // P's registers: the cell behind Q
                    if (walk_stop(over, ev_addr)) { par = 1; break; }
                    m_p = WALK_DDA(tx, ty, tz, widx, over);   // out of the cell in P's registers
                    walk_arrived(w_q, over);
                    walk_record(m_q, w_q, i_q, open_type, ev_addr, ev_lds);
                    walk_request(widx, flip, bits, i_q, w_q);
                    if (walk_stop(over, ev_addr)) break;
                }
                asm volatile("" : "+v"(par));   // (otherwise "the cell recorded last" is tracked with a move in every cell)
                m_last = par ? m_p : m_q;   // the cell recorded last
            }
#endif
            walk_arrived_all(w_p, w_q);
            sp.tx = tx; sp.ty = ty; sp.tz = tz; sp.widx = widx; sp.over = over;
            if (ev_addr >= WK_FULL) break;   // list full: phase 2 first (the unrecorded cell stays so)
            // the span's DDA is over: its last cell is recorded, and the next span (or the end) follows
            m_last = par ? m_q : m_p;
            walk_record(m_last, par ? w_q : w_p, par ? i_q : i_p, open_type, ev_addr, ev_lds);
            in_span = 0;
        }
        // ---------------- phase 2
        int32_t cnt = (int32_t)(ev_addr >> WK_LG);
        if (finished && has_open) { cnt += 1; has_open = 0; }
#ifndef NFA_WALK_NO_PHASE2
        if (LATTICE) lattice_run<HAS_LIMIT>(ls, col, cnt, ev_span, dt * 0.5f, limit, p, lt, tid);
        else marcher_run<HAS_LIMIT>(s, col, cnt, ev_span, dt, limit, p, tb, tid);
#endif
        if (finished || (limit > 0 && (LATTICE ? ls.n_samples : s.n_samples) >= limit)) break;
        // the open entry moves to slot 0
        ev_span = 0u;
        ev_addr = lane_off;
        if (has_open) *reinterpret_cast<float *>(col) = m_last;
    }
    if (LATTICE) { out.t_last = lattice_value(lt, ls.jr); out.n_samples = ls.n_samples; out.n_chains = ls.n_chains; out.n_runs = ls.n_runs; }
    else { out.t_last = s.t_last; out.n_samples = s.n_samples; out.n_chains = s.n_chains; out.n_runs = s.n_runs; }
}

// LATTICE: every ray starts at near_hint and the table of that near plane is in p.lat (phase 2 = lattice_run); otherwise
// the per-ray marcher (per-ray near planes: the API's traverse_grids with a tensor of them, the chunks of the test-mode loop)
template <bool FUSED, bool HAS_LIMIT, bool LATTICE>
__device__ __forceinline__ void walk_body(const nfa_traverse_args &a, const WalkParams &p)
{
    __shared__ __attribute__((aligned(16))) char ev_lds[(WK_EV + 1) << WK_LG];   // [WK_EV + 1][256] floats
    __shared__ ApproachLds tb;
    __shared__ LatticeLds lt;
#ifdef NFA_WALK_STAMPS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (LATTICE) lattice_to_lds(lt);
    else approach_to_lds(tb, p);
    __syncthreads();
    const uint32_t lane_off = 4u * threadIdx.x;
    // device-side controls (include/nerfacc_hip.h: steps_limit_dev, n_listed_dev): a loop that does not wait for the host
    const int32_t steps_limit = (HAS_LIMIT && a.steps_limit_dev) ? *a.steps_limit_dev : a.traverse_steps_limit;
    if (HAS_LIMIT && steps_limit <= 0) return;
    int64_t n_walk = p.order ? p.n_order : a.n_rays;
    if (p.order && a.n_listed_dev) n_walk = min(n_walk, *a.n_listed_dev);
#ifdef NFA_WALK_STAMPS
    const int64_t bid = p.tile_order ? (int64_t)p.tile_order[blockIdx.x] : xcd_fair_block(blockIdx.x, gridDim.x);
#else
    const int64_t bid = xcd_fair_block(blockIdx.x, gridDim.x);
#endif
    for (int64_t slot_i = bid * blockDim.x + threadIdx.x; slot_i < n_walk;
         slot_i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t tid = p.order ? (int64_t)p.order[slot_i] : slot_i;
        if (a.mode == 2 && a.rays_mask != nullptr && !a.rays_mask[tid]) {
            if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
            a.sm_cnts[tid] = 0;
            if (a.iv_cnts) a.iv_cnts[tid] = 0;
            p.run_cnts[tid] = 0;
            continue;
        }
        WalkOut s;
        walk_ray<FUSED, HAS_LIMIT, LATTICE>(a, p, tid, ev_lds, lane_off, tb, lt, steps_limit, s);
        if (a.terminate_planes) a.terminate_planes[tid] = s.t_last;
        a.sm_cnts[tid] = s.n_samples;
        if (a.iv_cnts) a.iv_cnts[tid] = s.n_samples + s.n_chains;  // edges = samples + one leading edge per chain
        // rays with > 2^21 samples go to the serial fill too (the expansion packs a 27-bit batch offset)
        int32_t n_runs = s.n_runs;
        if (s.n_samples > (1 << 21) && n_runs <= p.max_runs) n_runs = p.max_runs + 1;
        p.run_cnts[tid] = n_runs;
        if (n_runs > p.max_runs) atomicAdd(p.overflow, 1);
    }
#ifdef NFA_WALK_STAMPS
    if (p.stamps && (threadIdx.x & 63) == 0) {
        unsigned long long *o = p.stamps + (bid * (WK_THREADS / 64) + (threadIdx.x >> 6)) * 4;
        o[0] = st_t0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        o[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
#endif
}
// (two kernels: the register budget of the lattice form is held to 5 waves per SIMD, the per-ray marcher keeps what it needs)
template <bool FUSED, bool HAS_LIMIT>
NFA_WALK_OCC __global__ __launch_bounds__(WK_THREADS) void walk_lattice_kernel(const nfa_traverse_args a, const WalkParams p)
{
    walk_body<FUSED, HAS_LIMIT, true>(a, p);
}
template <bool FUSED, bool HAS_LIMIT>
__global__ __launch_bounds__(WK_THREADS) void walk_kernel(const nfa_traverse_args a, const WalkParams p)
{
    walk_body<FUSED, HAS_LIMIT, false>(a, p);
}

// ------------------------------------------------------------------------------------------------------------------
// The cone-angle walk (step_size > 0 and cone_angle > 0; ref grid.cu:207-262 recomputes dt = max(step, t * cone) at every
// sample and at every empty cell, so samples are no arithmetic runs and phase 2 above does not apply: the march is the
// reference's loop).  The DDA is phase 1's -- boundary distances, the packed step counter, the interleaved bit index, one
// 4-byte load per cell from the 1-bit grid copy requested a cell ahead -- instead of grid.hip's (brick index from three
// coordinates, end test on three overflow indices, a brick word cached in registers): 104 -> ~45 vector instructions per
// cell outside the march.  Output: the counts and run records of grid.hip's EMIT_RUNS pass (nfa_expand_cone_runs).
struct ConeParams {
    const uint32_t *bits;
    WalkLayout lay;
    int32_t *run_cnts;           // [n_rays]
    unsigned long long *runs;    // [max_runs, n_rays] slot-major
    int32_t max_runs;
    int32_t *overflow;           // [1]
    const int32_t *order;        // lane -> ray assignment or NULL
    int64_t n_order;
    int32_t chunk, min_busy;     // cone_refill_kernel: entries of the ray list per wave; lanes that keep the cell loop going
    // Records that do not fit the ray's max_runs slots go to an ARENA instead of sending the ray to the serial fill pass (a
    // second full walk: 1.2 ms per step on cfg 5 for 0.1 % of the rays): the ray keeps max_runs - 1 records, its last slot
    // holds a sentinel {NaN, samples so far} that makes the expansion skip the rest of its range, and the others are
    // appended to arena[] as {t_first, k_start | continues << 31, ray, samples}, CONE_ARENA_BLOCK at a time (one atomic per
    // block); nfa_expand_cone_arena writes their samples.  NULL: no arena (rays with too many records overflow as before).
    char *tab_lds;               // (set by the kernels) LDS: [3][256] rows of the DDA's per-axis constants (dda_step_lds)
    uint4 *arena;                // [arena_cap], zeroed by the caller (samples == 0: unused entry)
    int32_t arena_cap;
    int32_t *arena_count;        // [1] entries handed out (whole blocks), zeroed by the call
#ifdef NFA_CONE_PROFILE
    unsigned long long *profile; // [8] debugging aid: cycles outside / inside the cell loop, trips, lane-trips, rounds
#endif
};
#ifndef NFA_CONE_WALK_SPLIT
#define NFA_CONE_WALK_SPLIT 0
#endif
#ifndef NFA_CONE_REFILL_SPLIT
#define NFA_CONE_REFILL_SPLIT 1
#endif
constexpr uint32_t CONE_ARENA_BLOCK = 16u;          // arena entries a ray takes per atomic (a power of two)
constexpr uint32_t CONE_ARENA_NONE = 0xFFFFFFFFu;    // the lane's arena state: the arena was full when this ray asked
constexpr uint32_t CONE_ARENA_FRESH = 0xFFFFFFFEu;   //                         no entry yet
constexpr int CONE_WALK_RUN_CAP = 64;   // = grid.hip's CONE_RUN_CAP: the expansion iterates the recurrence at most this often
struct ConeRay {
    float t_last;
    int32_t continuous, n_samples, n_runs, run_len;
};

// the span's first cell
__device__ __forceinline__ void cone_span_begin(const nfa_traverse_args &a, const ConeParams &p, const float o[3], const float d[3],
                                                int32_t level, float this_tmin, float this_tmax, ConeRay &st, WalkSpan &sp,
                                                unsigned long long &w_cur, uint32_t &i_cur)
{
    if (!st.continuous) st.t_last = fast_forward(st.t_last, this_tmin, a.step_size, a.cone_angle);  // grid.cu:151-163
    walk_span_setup(a, p.lay, o, d, level, this_tmin, this_tmax, sp);
    dda_table_write(p.tab_lds, sp);
    i_cur = sp.widx ^ sp.flip;
    w_cur = reinterpret_cast<const unsigned long long *>(p.bits)[i_cur >> 6];
}

// One cell (grid.cu:184-272); true when the span is over (the step left its last cell, or the sample budget is spent).
template <bool SPLIT>
__device__ __forceinline__ bool cone_cell(const nfa_traverse_args &a, const ConeParams &p, int64_t tid, float this_tmax, WalkSpan &sp,
                                          unsigned long long &w_cur, uint32_t &i_cur, ConeRay &st, uint32_t *arena_slot /* LDS, this lane's */)
{
    const float step_size = a.step_size, cone = a.cone_angle;
    const int32_t limit = a.traverse_steps_limit;
    const uint32_t w_half = (i_cur & 32u) ? (uint32_t)(w_cur >> 32) : (uint32_t)w_cur;
    const bool occupied = __builtin_amdgcn_ubfe(w_half, i_cur, 1u) != 0u;   // bit (i_cur & 31) of the half
    const uint32_t tab_x = 16u * threadIdx.x;
    // (the end of the span: by the cell index in the one-ray-per-lane kernel -- cfg 5's count pass 7.30 -> 7.20 ms --, by the
    //  step counters in the refilling one, where the index form measured 2 % slower per image: 100.6 -> 102.6 ms)
    float m;
    bool done;
    if constexpr (SPLIT) {
        m = dda_step_lds(p.tab_lds, tab_x, tab_x + 2u * WK_TAB_AXIS, sp.tx, sp.ty, sp.tz, sp.rem, sp.widx);
        done = (sp.rem & WK_GUARD) != WK_GUARD;
    } else {
        m = dda_step_lds_end(p.tab_lds, tab_x, tab_x + 2u * WK_TAB_AXIS, sp.tx, sp.ty, sp.tz, sp.end, sp.widx, sp.over);
        done = sp.over == 0u;
    }
    const float t_traverse = vmin_f32(m, this_tmax);
    // The next cell's bit: the low six index bits are two of each coordinate, so an aligned 64-bit word of the copy is a
    // 4 x 4 x 4 brick; it stays in registers while the ray is inside it (with unrelated rays every load of a wave is 64
    // cache lines: one load per cell instead of one per brick cost the unlimited walk of cfg 5 2 ms of 12) and the load,
    // when there is one, is taken in after the march, whose instructions hide its latency.
    const uint32_t i_next = sp.widx ^ sp.flip;
    const bool fetch = !done && ((i_next ^ i_cur) >> 6) != 0u;
    unsigned long long w_next = 0ull;
    if (fetch) w_next = reinterpret_cast<const unsigned long long *>(p.bits)[i_next >> 6];

    // one sample [t_last, t_next) (grid.cu:219-258): counted, and a run record at the head of a chain and every 64 samples
    auto emit = [&](float t_next) {
        const bool cut = !st.continuous || st.run_len == CONE_WALK_RUN_CAP;
        if (cut) {
            const uint32_t kc = (uint32_t)st.n_samples | (st.continuous ? 0x80000000u : 0u);
            const int32_t inl = p.arena ? p.max_runs - 1 : p.max_runs;   // records the ray keeps in its own slots
            if (st.n_runs < inl) {
                p.runs[(int64_t)st.n_runs * a.n_rays + tid] = (unsigned long long)f32_bits(st.t_last) | ((unsigned long long)kc << 32);
            } else if (p.arena) {
                uint32_t e = *arena_slot;                                // the ray's previous arena entry (CONE_ARENA_NONE: gave up)
                if (st.n_runs == inl) {
                    p.runs[(int64_t)inl * a.n_rays + tid] = 0x7FC00000ull | ((unsigned long long)(uint32_t)st.n_samples << 32);
                    e = CONE_ARENA_FRESH;
                } else if (e < CONE_ARENA_NONE) {
                    reinterpret_cast<uint32_t *>(p.arena + e)[3] = (uint32_t)st.run_len;   // the previous entry is complete: its samples
                }
                if (e != CONE_ARENA_NONE) {
                    e = (e == CONE_ARENA_FRESH || ((e + 1u) & (CONE_ARENA_BLOCK - 1u)) == 0u) ? (uint32_t)atomicAdd(p.arena_count, (int32_t)CONE_ARENA_BLOCK)
                                                                                             : e + 1u;
                    if (e + CONE_ARENA_BLOCK <= (uint32_t)p.arena_cap || (e & (CONE_ARENA_BLOCK - 1u)) != 0u)
                        p.arena[e] = make_uint4(f32_bits(st.t_last), kc, (uint32_t)tid, 0u);
                    else
                        e = CONE_ARENA_NONE;                             // the arena is full: this ray goes to the serial fill pass
                }
                *arena_slot = e;
            }
            st.n_runs++;
        }
        st.run_len = cut ? 1 : st.run_len + 1;
        st.n_samples++;
        st.continuous = 1;
        st.t_last = t_next;
    };
    // March to t_traverse.  An empty cell skips with the dt of its first step (grid.cu:193-206), an occupied one emits with
    // dt recomputed per sample (grid.cu:207-262): one loop, so that a wave whose lanes sit in cells of both kinds runs it once.
    float dt = calc_dt(st.t_last, cone, step_size);
    if (SPLIT) {
        // The same two marches for walks that spend their time in empty cells (limited walks).  The empty cell's is
        // straight-line code: eight select steps cover a cell of the finest level at the smallest step, the loop behind them
        // runs only for what is left (a step without progress leaves t_last unchanged; the loop then sees it and the jump
        // applies, as in the merged loop); the sampling loop runs only when some lane of the wave has an occupied cell.
        if (!occupied) {
            if (t_traverse - st.t_last > 8.0f * dt) st.t_last = fast_forward_exact(st.t_last, t_traverse, dt);
            const float half = dt * 0.5f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float t_next = st.t_last + dt;
                st.t_last = (st.t_last + half < t_traverse) ? t_next : st.t_last;
            }
            for (;;) {
                const float t_next = st.t_last + dt;
                if (!((st.t_last + half < t_traverse) && (t_next != st.t_last))) break;
                st.t_last = t_next;
            }
            if (st.t_last + half < t_traverse) st.t_last = t_traverse;
            st.continuous = 0;
        } else {
            for (;;) {
                const float t_next = st.t_last + dt;
                const bool budget = !(limit > 0 && st.n_samples >= limit);
                if (!((st.t_last + dt * 0.5f < t_traverse) && (t_next != st.t_last) && budget)) break;
                emit(t_next);
                dt = calc_dt(t_next, cone, step_size);
            }
        }
    } else {
        // a skip of many steps (cell much larger than the step): closed form (march.h), same result as the loop
        if (!occupied && t_traverse - st.t_last > 8.0f * dt) st.t_last = fast_forward_exact(st.t_last, t_traverse, dt);
        for (;;) {
            const float t_next = st.t_last + dt;
            const bool budget = !(occupied && limit > 0 && st.n_samples >= limit);
            if (!((st.t_last + dt * 0.5f < t_traverse) && (t_next != st.t_last) && budget)) break;
            if (occupied) {
                emit(t_next);
                dt = calc_dt(t_next, cone, step_size);
            } else {
                st.t_last = t_next;
            }
        }
        if (!occupied) {
            // left the loop before the target without progress (ours: the reference would spin): jump there
            if (st.t_last + dt * 0.5f < t_traverse) st.t_last = t_traverse;
            st.continuous = 0;
        }
    }
    i_cur = i_next;
    w_cur = fetch ? w_next : w_cur;
    return done || (limit > 0 && st.n_samples >= limit);
}

// The ray's next span from its event list (grid.cu:125-150), or from the in-kernel slab test (FUSED: one grid).  `ev`: the
// next event to look at (FUSED: 0 = the span not taken yet).
template <bool FUSED>
__device__ __forceinline__ bool cone_next_span(const nfa_traverse_args &a, const ConeParams &p, int64_t tid, const float o[3],
                                               const float d[3], float near_plane, float far_plane, int32_t &ev, ConeRay &st,
                                               WalkSpan &sp, float &span_tmax, unsigned long long &w_cur, uint32_t &i_cur)
{
    const int32_t G = a.n_grids;
    if (FUSED) {
        if (ev != 0) return false;
        ev = 1;
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        float tmin, tmax, lo, hi;
        bool hit = true;
        const float *bmin = a.aabbs, *bmax = a.aabbs + 3;
        if (inv[0] >= 0) { tmin = (bmin[0] - o[0]) * inv[0]; tmax = (bmax[0] - o[0]) * inv[0]; }
        else             { tmin = (bmax[0] - o[0]) * inv[0]; tmax = (bmin[0] - o[0]) * inv[0]; }
#pragma unroll
        for (int ax = 1; ax < 3; ++ax) {
            if (inv[ax] >= 0) { lo = (bmin[ax] - o[ax]) * inv[ax]; hi = (bmax[ax] - o[ax]) * inv[ax]; }
            else              { lo = (bmax[ax] - o[ax]) * inv[ax]; hi = (bmin[ax] - o[ax]) * inv[ax]; }
            if (tmin > hi || lo > tmax) hit = false;
            if (lo > tmin) tmin = lo;
            if (hi < tmax) tmax = hi;
        }
        if (tmax <= 0) hit = false;
        const float this_tmin = fmaxf(tmin, near_plane), this_tmax = fminf(tmax, far_plane);
        if (!(hit && this_tmin < this_tmax)) return false;
        span_tmax = this_tmax;
        cone_span_begin(a, p, o, d, 0, this_tmin, this_tmax, st, sp, w_cur, i_cur);
        return true;
    } else {
        const uint8_t *hits = a.hits + tid * G;
        const float *ts = a.t_sorted + tid * 2 * G;
        const int64_t *ti = a.t_indices + tid * 2 * G;
        while (ev < 2 * G - 1) {
            const int32_t i = ev++;
            const int64_t idx = ti[i];
            int32_t level = event_level(idx, G);
            bool ok = (uint32_t)level < (uint32_t)G && hits[level] != 0;
            if (ok && idx >= G) {  // leaving: inside the next grid?
                const int64_t nidx = ti[i + 1];
                level = event_level(nidx, G);
                ok = nidx >= G && (uint32_t)level < (uint32_t)G && hits[level] != 0;
            }
            const float this_tmin = fmaxf(ts[i], near_plane);
            const float this_tmax = fminf(ts[i + 1], far_plane);
            if (ok && this_tmin < this_tmax) {
                span_tmax = this_tmax;
                cone_span_begin(a, p, o, d, level, this_tmin, this_tmax, st, sp, w_cur, i_cur);
                return true;
            }
        }
        return false;
    }
}

// The same walk over a ray's event list held by the lane: the (at most eight) argsort indices packed four bits each, the hit
// flags as a bit mask, the sorted distances in the lane's LDS column -- loaded in one go when the lane takes the ray.  Read from
// memory event by event (an index, then the flag it points at, then two distances: a chain of dependent loads per event,
// three to six events before a fresh ray's first span) the list was most of a refill round's 19 k cycles.
constexpr int CONE_EV_MAX = 8;   // 2 * n_grids entries: up to four levels
__device__ __forceinline__ void cone_stage_events(const nfa_traverse_args &a, int64_t tid, uint32_t &ti_pack, uint32_t &hit_mask, float *ts_col)
{
    const int32_t G = a.n_grids;
    const int64_t *ti = a.t_indices + tid * 2 * G;
    const float *ts = a.t_sorted + tid * 2 * G;
    const uint8_t *hits = a.hits + tid * G;
    uint32_t pk = 0u, hm = 0u;
#pragma unroll
    for (int i = 0; i < CONE_EV_MAX; ++i)
        if (i < 2 * G) {
            const int64_t v = ti[i];
            pk |= ((uint64_t)v < (uint64_t)(2 * G) ? (uint32_t)v : 15u) << (4 * i);   // (out of range: 15, a level nobody hits)
            ts_col[i * CONE_THREADS] = ts[i];
        }
#pragma unroll
    for (int g = 0; g < CONE_EV_MAX / 2; ++g)
        if (g < G) hm |= (hits[g] != 0 ? 1u : 0u) << g;
    ti_pack = pk; hit_mask = hm;
}

__device__ __forceinline__ bool cone_next_span_staged(const nfa_traverse_args &a, const ConeParams &p, const float o[3], const float d[3],
                                                      float near_plane, float far_plane, int32_t &ev, uint32_t ti_pack, uint32_t hit_mask,
                                                      const float *ts_col, ConeRay &st, WalkSpan &sp, float &span_tmax,
                                                      unsigned long long &w_cur, uint32_t &i_cur)
{
    const int32_t G = a.n_grids;
    while (ev < 2 * G - 1) {  // grid.cu:125-150
        const int32_t i = ev++;
        const int32_t idx = (int32_t)((ti_pack >> (4 * i)) & 15u);
        int32_t level = idx >= G ? idx - G : idx;
        bool ok = level < G && ((hit_mask >> level) & 1u) != 0u;
        if (ok && idx >= G) {  // leaving: inside the next grid?
            const int32_t nidx = (int32_t)((ti_pack >> (4 * (i + 1))) & 15u);
            level = nidx >= G ? nidx - G : nidx;
            ok = nidx >= G && level < G && ((hit_mask >> level) & 1u) != 0u;
        }
        const float this_tmin = fmaxf(ts_col[i * CONE_THREADS], near_plane);
        const float this_tmax = fminf(ts_col[(i + 1) * CONE_THREADS], far_plane);
        if (ok && this_tmin < this_tmax) {
            span_tmax = this_tmax;
            cone_span_begin(a, p, o, d, level, this_tmin, this_tmax, st, sp, w_cur, i_cur);
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ void cone_ray_out(const nfa_traverse_args &a, const ConeParams &p, int64_t tid, const ConeRay &st,
                                             const uint32_t *arena_slot)
{
    if (a.terminate_planes) a.terminate_planes[tid] = st.t_last;
    a.sm_cnts[tid] = st.n_samples;
    // rays with > 2^21 samples go to the serial fill (the expansion packs a 27-bit batch offset)
    int32_t n_runs = st.n_runs;
    if (p.arena && n_runs >= p.max_runs) {   // the ray's last records are in the arena: its slots hold max_runs - 1 of them + the sentinel
        const uint32_t e = *arena_slot;
        if (e < CONE_ARENA_FRESH) { reinterpret_cast<uint32_t *>(p.arena + e)[3] = (uint32_t)st.run_len; n_runs = p.max_runs; }
        else n_runs = p.max_runs + 1;
    }
    if (st.n_samples > (1 << 21) && n_runs <= p.max_runs) n_runs = p.max_runs + 1;
    p.run_cnts[tid] = n_runs;
    if (n_runs > p.max_runs) atomicAdd(p.overflow, 1);
}

// a ray masked out by rays_mask (grid.cu:100; the reference leaves its outputs uninitialised, we define them)
__device__ __forceinline__ bool cone_ray_masked(const nfa_traverse_args &a, const ConeParams &p, int64_t tid)
{
    if (!(a.mode == 2 && a.rays_mask != nullptr && !a.rays_mask[tid])) return false;
    if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
    a.sm_cnts[tid] = 0;
    p.run_cnts[tid] = 0;
    return true;
}

// one ray per lane, from its first span to its last
#ifndef NFA_CONE_WALK_WAVES
#define NFA_CONE_WALK_WAVES 6   /* 89 -> 80 registers: 6 waves per SIMD instead of 5; cfg 5's traversal 10.6 -> 10.1 ms (4 waves 11.4, 8 spill: 10.8) */
#endif
template <bool FUSED>
__attribute__((amdgpu_waves_per_eu(NFA_CONE_WALK_WAVES, NFA_CONE_WALK_WAVES)))
__global__ __launch_bounds__(CONE_THREADS) void cone_walk_kernel(const nfa_traverse_args a, const ConeParams p_in)
{
    __shared__ __attribute__((aligned(16))) char s_tab[3 * WK_TAB_AXIS];
    ConeParams p = p_in;
    p.tab_lds = s_tab;
    __shared__ uint32_t s_arena[CONE_THREADS];        // per lane: the ray's current arena entry (cone_cell: emit)
    uint32_t *const arena_slot = s_arena + threadIdx.x;
    const int64_t n_walk = p.order ? p.n_order : a.n_rays;
    const int32_t limit = a.traverse_steps_limit;
    for (int64_t slot_i = xcd_fair_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; slot_i < n_walk;
         slot_i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t tid = p.order ? (int64_t)p.order[slot_i] : slot_i;
        if (cone_ray_masked(a, p, tid)) continue;
        const float near_plane = a.near_planes[tid], far_plane = a.far_planes[tid];
        const float o[3] = {a.rays_o[3 * tid], a.rays_o[3 * tid + 1], a.rays_o[3 * tid + 2]};
        const float d[3] = {a.rays_d[3 * tid], a.rays_d[3 * tid + 1], a.rays_d[3 * tid + 2]};
        ConeRay st;
        st.t_last = near_plane; st.continuous = 0; st.n_samples = 0; st.n_runs = 0; st.run_len = 0;
        // (a ray with a non-finite origin or direction has no geometry: no samples, see grid.hip's traverse_kernel)
        const bool ray_ok = isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(d[0]) && isfinite(d[1]) && isfinite(d[2]);
        int32_t ev = ray_ok ? 0 : 2 * a.n_grids;
        WalkSpan sp;
        float span_tmax = 0.f;
        unsigned long long w_cur = 0ull;
        uint32_t i_cur = 0u;
        while (cone_next_span<FUSED>(a, p, tid, o, d, near_plane, far_plane, ev, st, sp, span_tmax, w_cur, i_cur)) {
            while (!cone_cell<NFA_CONE_WALK_SPLIT != 0>(a, p, tid, span_tmax, sp, w_cur, i_cur, st, arena_slot)) {}
            // The budget is spent: the last thing that happened was a sample (continuous), so the spans still to come would
            // change nothing (grid.cu:151,185: no fast-forward, no cell visited).
            if (limit > 0 && st.n_samples >= limit) break;
        }
        cone_ray_out(a, p, tid, st, arena_slot);
    }
}

// Limited walks (traverse_steps_limit > 0: one iteration of the test-mode loop, examples/utils.py:252-425) stop after a
// handful of samples, i.e. after a number of cells that is geometric in the local occupancy; with one ray per lane a wave
// lasts as long as its unluckiest ray (cfg 5, 2 % scattered occupancy: 50 cells to the first sample on average, ~240 for
// the worst of 64 lanes, lanes busy a fifth of the time).  Here a wave owns `chunk` consecutive entries of the ray list
// and a lane that has finished its ray is given the next one: the wave leaves its cell loop when fewer than `min_busy`
// lanes are still walking, sets up new rays (and the next spans of rays that crossed into another level) on the free
// lanes, and re-enters.  Per ray the same functions as cone_walk_kernel: identical results.
template <bool FUSED, bool STAGED /* the event list travels with the lane (n_grids <= 4) */>
#ifdef NFA_CONE_REFILL_WAVES
__attribute__((amdgpu_waves_per_eu(NFA_CONE_REFILL_WAVES, NFA_CONE_REFILL_WAVES)))
#endif
__global__ __launch_bounds__(CONE_THREADS) void cone_refill_kernel(const nfa_traverse_args a, const ConeParams p_in)
{
    static_assert(!(FUSED && STAGED), "a fused walk has no event list");
    __shared__ __attribute__((aligned(16))) char s_tab[3 * WK_TAB_AXIS];
    ConeParams p = p_in;
    p.tab_lds = s_tab;
    __shared__ float ts_lds[STAGED ? CONE_EV_MAX * CONE_THREADS : 1];
    __shared__ uint32_t s_arena[CONE_THREADS];        // per lane: the ray's current arena entry (cone_cell: emit)
    uint32_t *const arena_slot = s_arena + threadIdx.x;
    float *const ts_col = ts_lds + (STAGED ? threadIdx.x : 0);
    uint32_t ti_pack = 0u, hit_mask = 0u;
    enum { IDLE = 0, SPAN = 1, WALK = 2, FINISH = 3 };
    const int lane = lane_id();
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const int64_t n_walk = p.order ? p.n_order : a.n_rays;
    const int64_t wave = xcd_fair_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t next = wave * p.chunk;  // (wave-uniform) first entry not handed out yet
    const int64_t end = next + p.chunk < n_walk ? next + p.chunk : n_walk;
    const int32_t limit = a.traverse_steps_limit;

    int32_t phase = IDLE, ev = 0;
    int64_t tid = 0;
    float near_plane = 0.0f, far_plane = 0.0f, span_tmax = 0.0f;
    float o[3] = {0.0f, 0.0f, 0.0f}, d[3] = {0.0f, 0.0f, 0.0f};
    ConeRay st;
    st.t_last = 0.0f; st.continuous = 0; st.n_samples = 0; st.n_runs = 0; st.run_len = 0;
    WalkSpan sp;
    sp.tx = sp.ty = sp.tz = sp.dx = sp.dy = sp.dz = 0.f; sp.mx = sp.my = sp.mz = 0u; sp.rem = 0u; sp.widx = 0u; sp.flip = 0u; sp.end = 0u; sp.over = 0u;
    unsigned long long w_cur = 0ull;
    uint32_t i_cur = 0u;

#ifdef NFA_CONE_PROFILE
    unsigned long long pf_setup = 0, pf_cells = 0, pf_trips = 0, pf_lanes = 0, pf_rounds = 0, pf_t = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        // Two passes: rays that left a span in the cell loop (their next span, or their end), then the rays handed to the
        // lanes that are free after that.
#pragma nounroll
        for (int pass = 0; pass < 2; ++pass) {
            if (phase == SPAN) {
                bool found;
                if (STAGED) found = cone_next_span_staged(a, p, o, d, near_plane, far_plane, ev, ti_pack, hit_mask, ts_col, st, sp, span_tmax, w_cur, i_cur);
                else found = cone_next_span<FUSED>(a, p, tid, o, d, near_plane, far_plane, ev, st, sp, span_tmax, w_cur, i_cur);
                phase = found ? WALK : FINISH;
            }
            if (phase == FINISH) {
                cone_ray_out(a, p, tid, st, arena_slot);
                phase = IDLE;
            }
            if (pass == 1) break;
            const unsigned long long idle = __ballot(phase == IDLE);
            if (idle != 0ull && next < end) {
                if (phase == IDLE) {
                    const int64_t slot = next + __popcll(idle & lanes_below);
                    if (slot < end) {
                        tid = p.order ? (int64_t)p.order[slot] : slot;
                        if (!cone_ray_masked(a, p, tid)) {
                            near_plane = a.near_planes[tid]; far_plane = a.far_planes[tid];
#pragma unroll
                            for (int ax = 0; ax < 3; ++ax) { o[ax] = a.rays_o[3 * tid + ax]; d[ax] = a.rays_d[3 * tid + ax]; }
                            if (STAGED) cone_stage_events(a, tid, ti_pack, hit_mask, ts_col);
                            st.t_last = near_plane; st.continuous = 0; st.n_samples = 0; st.n_runs = 0; st.run_len = 0;
                            const bool ray_ok = isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(d[0]) && isfinite(d[1]) && isfinite(d[2]);
                            ev = ray_ok ? 0 : 2 * a.n_grids;
                            phase = SPAN;
                        }
                    }
                }
                next += __popcll(idle);
            }
        }
        const unsigned long long walking = __ballot(phase == WALK);
#ifdef NFA_CONE_PROFILE
        { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); pf_setup += t1 - pf_t; pf_t = t1; pf_rounds++; }
#endif
        if (walking == 0ull) {
            if (next >= end) break;  // (every lane is IDLE here: SPAN and FINISH were resolved above)
            continue;
        }
        // ---- cells, for as long as enough lanes have one to visit
        const int32_t n_walking = __popcll(walking);
        const int32_t need = next < end ? p.min_busy : (n_walking * 3 >> 2) > 1 ? (n_walking * 3 >> 2) : 1;
        do {
#ifdef NFA_CONE_PROFILE
            pf_trips++; pf_lanes += __popcll(__ballot(phase == WALK));
#endif
            if (phase == WALK) {
                if (cone_cell<NFA_CONE_REFILL_SPLIT != 0>(a, p, tid, span_tmax, sp, w_cur, i_cur, st, arena_slot))
                    phase = (limit > 0 && st.n_samples >= limit) ? FINISH : SPAN;  // budget spent: nothing after it changes the ray
            }
        } while (__popcll(__ballot(phase == WALK)) >= need);
#ifdef NFA_CONE_PROFILE
        { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); pf_cells += t1 - pf_t; pf_t = t1; }
#endif
    }
#ifdef NFA_CONE_PROFILE
    if (p.profile && lane == 0) {
        unsigned long long *pr = p.profile + 8 * (wave & 127);
        atomicAdd(pr + 0, pf_setup); atomicAdd(pr + 1, pf_cells); atomicAdd(pr + 2, pf_trips); atomicAdd(pr + 3, pf_lanes);
        atomicAdd(pr + 4, pf_rounds); atomicAdd(pr + 5, 1ull);
    }
#endif
}

}  // namespace nfa

using namespace nfa;

extern "C" {

int64_t nfa_walk_bits_words(int32_t n_grids, const int32_t *res)
{
    const WalkLayout L = walk_layout(res);
    return (((int64_t)n_grids << L.bits) + 31) / 32;
}

int nfa_pack_walk_bits(const uint8_t *binaries, int32_t n_grids, const int32_t *res, uint32_t *bits, nfa_stream_t stream)
{
    NFA_REQUIRE(binaries && res && bits && n_grids >= 1, "pack_walk_bits: bad arguments");
    NFA_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && res[0] <= WK_MAX_RES && res[1] <= WK_MAX_RES && res[2] <= WK_MAX_RES,
                "pack_walk_bits: 1..512 cells per axis");
    const WalkLayout L = walk_layout(res);
    NFA_REQUIRE(L.bits >= 5 && ((int64_t)n_grids << L.bits) < ((int64_t)1 << 31), "pack_walk_bits: grid too large");
    const int64_t n_words = ((int64_t)n_grids << L.bits) >> 5;
    hipLaunchKernelGGL(pack_walk_bits_kernel, dim3(grid_1d(n_words, 256)), dim3(256), 0, as_stream(stream), binaries, n_grids, res[0],
                       res[1], res[2], L, bits);
    NFA_CHECK_LAUNCH("pack_walk_bits");
    return NFA_OK;
}

int nfa_traverse_runs(const nfa_traverse_args *pa, const uint32_t *bits, int32_t *run_cnts, uint64_t *runs, int32_t max_runs,
                      int32_t *overflow_count, float near_hint, const int32_t *ray_order, int64_t n_order, nfa_stream_t stream)
{
    NFA_REQUIRE(pa != nullptr, "traverse_runs: null args");
    const nfa_traverse_args &a = *pa;
    NFA_REQUIRE(a.n_rays >= 0 && a.n_rays < (int64_t)1 << 31, "traverse_runs: n_rays out of range");
    NFA_REQUIRE(overflow_count, "traverse_runs: overflow_count is null");
    hipStream_t s = as_stream(stream);
    // (device-driven calls -- steps_limit_dev set -- come with overflow_count zeroed by nfa_testmode_begin: no memset node in their graph)
    if (!a.steps_limit_dev && hipMemsetAsync(overflow_count, 0, 2 * sizeof(int32_t), s) != hipSuccess) { set_error("traverse_runs: memset failed"); return NFA_EHIP; }
    if (a.n_rays == 0) return NFA_OK;
    NFA_REQUIRE(a.step_size > 0.0f && a.cone_angle == 0.0f, "traverse_runs: needs step_size > 0 and cone_angle == 0");
    NFA_REQUIRE(a.mode == 0 || a.mode == 2, "traverse_runs: mode must be 0 (all rays) or 2 (rays_mask + limit)");
    NFA_REQUIRE(a.mode != 2 || a.traverse_steps_limit > 0, "traverse_steps_limit must be > 0 when over_allocate is true");
    NFA_REQUIRE(a.rays_o && a.rays_d && a.aabbs && a.near_planes && a.far_planes && a.sm_cnts && bits && run_cnts && runs,
                "traverse_runs: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs <= 32, "traverse_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(a.n_grids >= 1 && a.res[0] > 0 && a.res[1] > 0 && a.res[2] > 0, "traverse_runs: bad grid shape");
    NFA_REQUIRE(a.res[0] <= WK_MAX_RES && a.res[1] <= WK_MAX_RES && a.res[2] <= WK_MAX_RES,
                "traverse_runs: at most 512 cells per axis (use nfa_traverse_grids beyond)");
    const bool fused = !a.t_sorted && !a.t_indices && !a.hits;
    NFA_REQUIRE(fused || (a.t_sorted && a.t_indices && a.hits), "traverse_runs: t_sorted, t_indices and hits must be given together");
    NFA_REQUIRE(!fused || a.n_grids == 1, "traverse_runs: in-kernel intersection supports one grid");
    WalkParams p;
    p.bits = bits;
    p.lay = walk_layout(a.res);
    NFA_REQUIRE(p.lay.bits >= 5 && ((int64_t)a.n_grids << p.lay.bits) < ((int64_t)1 << 31), "traverse_runs: grid too large");
    p.run_cnts = run_cnts;
    p.runs = reinterpret_cast<unsigned long long *>(runs);
    p.max_runs = max_runs;
    p.n_rays = a.n_rays;
    p.overflow = overflow_count;
    p.order = ray_order;
    p.n_order = ray_order ? n_order : a.n_rays;
    NFA_REQUIRE(!ray_order || (n_order >= 0 && n_order <= a.n_rays), "traverse_runs: n_order out of range");
    if (ray_order && n_order == 0) return NFA_OK;
    // near_hint: the value most (or all) entries of near_planes hold, NaN if unknown.  Rays whose near plane
    // differs bit-wise simply do not use the table.
    // With a near plane that holds for every ray, phase 2 runs on the lattice of (near, step): its table is built here, from
    // the two numbers.  Rays the table does not serve (a near plane that differs, a march beyond the tabulated sequence)
    // raise overflow_count[1], and the caller repeats the call with near_hint = NaN: the per-ray marcher.
    p.approach.n = 0;
    p.lat.n_rows = 0;
    if (near_hint == near_hint && !tuning_env("NFA_WALK_NO_LATTICE")) lattice_table_build(p.lat, near_hint, a.step_size);
    const bool lattice = p.lat.n_rows >= 1 && p.lat.n_binades >= 1;
    if (!lattice && near_hint == near_hint) approach_table_build(p.approach, near_hint, a.step_size);
#ifdef NFA_WALK_STAMPS
    { const char *e = getenv("NFA_WALK_STAMPS_PTR"); p.stamps = e ? reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0)) : nullptr; }
    { const char *e = getenv("NFA_WALK_ORDER_PTR"); p.tile_order = e ? reinterpret_cast<const int32_t *>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
    const size_t shmem = 0;  // the lists are static LDS
    const unsigned grid = grid_1d(p.n_order, WK_THREADS, 1 << 20);
    const bool lim = a.traverse_steps_limit > 0;
#define NFA_WALK_LAUNCH(K, F, L) hipLaunchKernelGGL((K<F, L>), dim3(grid), dim3(WK_THREADS), shmem, s, a, p)
    if (lattice) {
        if (fused && !lim)      NFA_WALK_LAUNCH(walk_lattice_kernel, true, false);
        else if (fused)         NFA_WALK_LAUNCH(walk_lattice_kernel, true, true);
        else if (!lim)          NFA_WALK_LAUNCH(walk_lattice_kernel, false, false);
        else                    NFA_WALK_LAUNCH(walk_lattice_kernel, false, true);
    } else {
        if (fused && !lim)      NFA_WALK_LAUNCH(walk_kernel, true, false);
        else if (fused)         NFA_WALK_LAUNCH(walk_kernel, true, true);
        else if (!lim)          NFA_WALK_LAUNCH(walk_kernel, false, false);
        else                    NFA_WALK_LAUNCH(walk_kernel, false, true);
    }
#undef NFA_WALK_LAUNCH
    NFA_CHECK_LAUNCH("traverse_runs");
    return NFA_OK;
}

// nfa_expand_cone_arena: the samples of the arena's entries (ConeParams::arena): one wave per entry, lane i re-runs i steps of
// the recurrence t <- t + max(step, t * cone) from the entry's first distance (grid.cu:213-216, as expand_runs_kernel<EXP_CONE>
// does for the records a ray keeps in its own slots) and writes sample k_start + i of the entry's ray
__global__ __launch_bounds__(256) void expand_cone_arena_kernel(const uint4 *__restrict__ arena, int32_t n_entries, float step, float cone,
                                                                const longlong2 *__restrict__ packed_info, float *__restrict__ t_starts,
                                                                float *__restrict__ t_ends, int64_t *__restrict__ ray_indices)
{
    const int32_t e = (int32_t)(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (e >= n_entries) return;
    const uint4 rec = arena[e];
    const uint32_t i = threadIdx.x & 63u;
    if (i >= rec.w) return;                                   // (rec.w == 0: an unused entry of a ray's block)
    float t = bits_f32(rec.x);
    for (uint32_t k = 0; k < i; ++k) t = t + calc_dt(t, cone, step);
    const int64_t at = packed_info[rec.z].x + (int64_t)(rec.y & 0x7FFFFFFFu) + i;
    t_starts[at] = t;
    t_ends[at] = t + calc_dt(t, cone, step);
    ray_indices[at] = (int64_t)rec.z;
}

int nfa_expand_cone_arena(const uint32_t *arena, int32_t n_entries, float step_size, float cone_angle, const int64_t *packed_info,
                          float *t_starts, float *t_ends, int64_t *ray_indices, nfa_stream_t stream)
{
    NFA_REQUIRE(n_entries >= 0, "expand_cone_arena: negative n_entries");
    if (n_entries == 0) return NFA_OK;
    NFA_REQUIRE(arena && packed_info && t_starts && t_ends && ray_indices, "expand_cone_arena: null pointer");
    NFA_REQUIRE(step_size > 0.0f && cone_angle > 0.0f, "expand_cone_arena: step_size and cone_angle must be > 0");
    hipLaunchKernelGGL(expand_cone_arena_kernel, dim3((unsigned)((n_entries + 3) / 4)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const uint4 *>(arena), n_entries, step_size, cone_angle, reinterpret_cast<const longlong2 *>(packed_info),
                       t_starts, t_ends, ray_indices);
    NFA_CHECK_LAUNCH("expand_cone_arena");
    return NFA_OK;
}

int nfa_traverse_cone_walk(const nfa_traverse_args *pa, const uint32_t *bits, int32_t *run_cnts, uint64_t *runs, int32_t max_runs,
                           int32_t *overflow_count, uint32_t *arena, int32_t arena_capacity, const int32_t *ray_order, int64_t n_order,
                           nfa_stream_t stream)
{
    NFA_REQUIRE(pa != nullptr, "traverse_cone_walk: null args");
    const nfa_traverse_args &a = *pa;
    NFA_REQUIRE(a.n_rays >= 0 && a.n_rays < (int64_t)1 << 31, "traverse_cone_walk: n_rays out of range");
    NFA_REQUIRE(overflow_count, "traverse_cone_walk: overflow_count is null");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(overflow_count, 0, 2 * sizeof(int32_t), s) != hipSuccess) { set_error("traverse_cone_walk: memset failed"); return NFA_EHIP; }
    NFA_REQUIRE(arena_capacity >= 0 && (arena == nullptr || arena_capacity % (int32_t)CONE_ARENA_BLOCK == 0), "traverse_cone_walk: arena_capacity must be a multiple of 16");
    if (a.n_rays == 0) return NFA_OK;
    NFA_REQUIRE(a.step_size > 0.0f && a.cone_angle > 0.0f, "traverse_cone_walk: needs step_size > 0 and cone_angle > 0");
    NFA_REQUIRE(a.mode == 0 || a.mode == 2, "traverse_cone_walk: mode must be 0 (all rays) or 2 (rays_mask + traverse_steps_limit)");
    NFA_REQUIRE(a.mode != 2 || a.traverse_steps_limit > 0, "traverse_steps_limit must be > 0 when over_allocate is true");
    NFA_REQUIRE(a.rays_o && a.rays_d && a.aabbs && a.near_planes && a.far_planes && a.sm_cnts && !a.iv_cnts && bits && run_cnts && runs,
                "traverse_cone_walk: null pointer (or interval outputs requested)");
    NFA_REQUIRE(max_runs >= 1 && max_runs <= 32, "traverse_cone_walk: max_runs must be in [1, 32]");
    NFA_REQUIRE(a.n_grids >= 1 && a.res[0] > 0 && a.res[1] > 0 && a.res[2] > 0, "traverse_cone_walk: bad grid shape");
    NFA_REQUIRE(a.res[0] <= WK_MAX_RES && a.res[1] <= WK_MAX_RES && a.res[2] <= WK_MAX_RES,
                "traverse_cone_walk: at most 512 cells per axis (use nfa_traverse_cone_runs beyond)");
    const bool fused = !a.t_sorted && !a.t_indices && !a.hits;
    NFA_REQUIRE(fused || (a.t_sorted && a.t_indices && a.hits), "traverse_cone_walk: t_sorted, t_indices and hits must be given together");
    NFA_REQUIRE(!fused || a.n_grids == 1, "traverse_cone_walk: in-kernel intersection supports one grid");
    ConeParams p;
    p.bits = bits;
    p.lay = walk_layout(a.res);
    NFA_REQUIRE(p.lay.bits >= 6 && ((int64_t)a.n_grids << p.lay.bits) < ((int64_t)1 << 31),
                "traverse_cone_walk: a level of the grid copy must be a whole number of 64-bit words (at least 4 cells per axis) and the copy below 2^31 bits");
    p.run_cnts = run_cnts;
    p.runs = reinterpret_cast<unsigned long long *>(runs);
    p.max_runs = max_runs;
    p.overflow = overflow_count;
    p.arena = (arena && arena_capacity > 0 && max_runs >= 2) ? reinterpret_cast<uint4 *>(arena) : nullptr;
    p.arena_cap = arena_capacity;
    p.arena_count = overflow_count + 1;
    p.order = ray_order;
    p.n_order = ray_order ? n_order : a.n_rays;
    NFA_REQUIRE(!ray_order || (n_order >= 0 && n_order <= a.n_rays), "traverse_cone_walk: n_order out of range");
    if (ray_order && n_order == 0) return NFA_OK;
#ifdef NFA_CONE_PROFILE
    {   // debugging aid: env NFA_CONE_PROFILE_PTR = address of a zeroed device buffer of 128 x 8 uint64
        const char *e = getenv("NFA_CONE_PROFILE_PTR");
        p.profile = e ? reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 0)) : nullptr;
    }
#endif
    const char *refill_env = tuning_env("NFA_REFILL");  // "0": one ray per lane also for limited walks; "chunk,min_busy": tuning
    const char *refill_all = tuning_env("NFA_REFILL_ALL");   // "1": the refilling kernel for unlimited walks too (measurements)
    if ((a.traverse_steps_limit > 0 || (refill_all && refill_all[0] == '1')) && !(refill_env && refill_env[0] == '0')) {
        // entries per wave: enough of them that a lane is refilled several times, as long as the launch still fills the chip
        int64_t chunk = ((p.n_order + 4095) / 4096 + 63) / 64 * 64;
        chunk = chunk < 64 ? 64 : (chunk > 1024 ? 1024 : chunk);
        int min_busy = 48;
        if (refill_env) { long c = 0; int m = 0; if (sscanf(refill_env, "%ld,%d", &c, &m) == 2 && c >= 64 && m >= 1 && m <= 64) { chunk = c / 64 * 64; min_busy = m; } }
        p.chunk = (int32_t)chunk; p.min_busy = min_busy;
        const int64_t n_waves = (p.n_order + chunk - 1) / chunk;
        constexpr int wpb = CONE_THREADS / 64;
        const unsigned grid = (unsigned)((n_waves + wpb - 1) / wpb);
        const char *staged_env = tuning_env("NFA_CONE_STAGED");   // "0": event lists read from memory (A/B)
        const bool staged = !fused && 2 * a.n_grids <= CONE_EV_MAX && !(staged_env && staged_env[0] == '0');
        if (fused)       hipLaunchKernelGGL((cone_refill_kernel<true, false>), dim3(grid), dim3(CONE_THREADS), 0, s, a, p);
        else if (staged) hipLaunchKernelGGL((cone_refill_kernel<false, true>), dim3(grid), dim3(CONE_THREADS), 0, s, a, p);
        else             hipLaunchKernelGGL((cone_refill_kernel<false, false>), dim3(grid), dim3(CONE_THREADS), 0, s, a, p);
    } else {
        p.chunk = 64; p.min_busy = 64;
        const unsigned grid = grid_1d(p.n_order, CONE_THREADS, 1 << 20);
        if (fused) hipLaunchKernelGGL((cone_walk_kernel<true>), dim3(grid), dim3(CONE_THREADS), 0, s, a, p);
        else       hipLaunchKernelGGL((cone_walk_kernel<false>), dim3(grid), dim3(CONE_THREADS), 0, s, a, p);
    }
    NFA_CHECK_LAUNCH("traverse_cone_walk");
    return NFA_OK;
}

}  // extern "C"
