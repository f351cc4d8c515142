// walk.hip -- the run-length walk of the constant-step traversal (cone_angle == 0): one DDA pass per ray that leaves
// RUN RECORDS (n consecutive samples with one exact fp32 increment) for traverse2.hip's coalesced expansions.
//
// Reference semantics: cuda/csrc/grid.cu:68-282 (kernel), include/utils_grid.cuh:58-142 (setup_traversal,
// single_traversal); the reference walks every ray twice (count + fill) with one scattered 1-byte grid load, a
// per-sample loop and three-way divergence per cell.  The walk is bound by instruction issue, not by memory, so it
// is built around the instruction count per cell:
//
//   phase 1  the cell loop does ONLY the DDA: the three boundary distances, a packed step counter that ends the span
//            (three 10-bit fields with a guard bit each: the step that would leave the span's last cell clears a guard),
//            a linear bit index that moves by a per-axis stride, one 4-byte load per cell from the 1-bit-per-cell copy of
//            the grid (issued for the NEXT cell: the cell sequence does not depend on occupancy), and one LDS store: the
//            exit distance of the current cell goes to the slot of the ray's open list entry, and the slot index moves
//            on when the occupancy flips.  No marching, no branches besides the loop's own.  ~25 vector instructions.
//   phase 2  a ray's list is a handful of thresholds of alternating kind (skip to / emit to).  Inside one binade every
//            step of the serial accumulation t += dt adds the same number q of ulps (march.h), so the march to a
//            threshold is "the smallest J with fl(t + J q ulp + dt/2) >= thr": an fp32 estimate and four exact probes,
//            straight-line code, no loop.  Events that leave the binade (or stand on the near plane, or meet an exact
//            tie) are left for a general path that the whole wave runs together once the lock-step loop has drained.
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
constexpr int WK_EV = NFA_WK_EV;           // list slots per ray: 16 KiB of LDS per 256 rays -> 8 workgroups per CU
#ifndef NFA_WALK_WAVES
#define NFA_WALK_WAVES 0
#endif
#if NFA_WALK_WAVES > 0
#define NFA_WALK_OCC __attribute__((amdgpu_waves_per_eu(NFA_WALK_WAVES, NFA_WALK_WAVES)))
#else
#define NFA_WALK_OCC
#endif
static_assert(NFA_WK_EV == 16, "the list-full test is bit 14 of the slot address: 16 slots of 1 KiB");
constexpr uint32_t WK_FULL = (uint32_t)WK_EV << 10;  // slot address bit that says "the open entry sits in slot WK_EV"
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
    ApproachTable approach;      // march.h; n == 0: none
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

// n samples t0, t0 + inc, ... join the ray's run list; a run record {t_first : f32 | k_start : 31, continues_previous : 1}
// is written when a run starts (its length is the next record's k_start, or the ray's count)
__device__ __forceinline__ void marcher_emit(Marcher &s, float t0, float inc, uint32_t n, const WalkParams &p, int64_t tid)
{
    if (!(s.continuous && inc == s.run_inc)) {
        if (s.n_runs < p.max_runs)
            p.runs[(int64_t)s.n_runs * p.n_rays + tid] =
                (unsigned long long)f32_bits(t0) |
                ((unsigned long long)((uint32_t)s.n_samples | (s.continuous ? 0x80000000u : 0u)) << 32);
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
            const float v0 = *reinterpret_cast<const float *>(col + (k << 10));
            const float v1 = *reinterpret_cast<const float *>(col + ((k + 1) << 10));   // (slot k + 1 <= WK_EV exists)
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
            if (new_run && s.n_runs < p.max_runs)
                (p.runs + tid)[(int64_t)s.n_runs * p.n_rays] =
                    (unsigned long long)bt | ((unsigned long long)((uint32_t)s.n_samples | ((uint32_t)s.continuous << 31)) << 32);
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
                if (is_span_entry(ev_span, k)) s.span_tmax = *reinterpret_cast<const float *>(col + ((k + 1) << 10));
                k += adv; s.ptype = next_ptype;
            }
        }
    }
}

// DDA state of the span being walked
struct WalkSpan {
    float tx, ty, tz, dx, dy, dz;
    uint32_t mx, my, mz;      // bits of each axis in the interleaved cell index (0: the ray does not move along the axis)
    uint32_t rem;             // steps left per axis (9 bits + guard each)
    uint32_t widx;            // interleaved index of the current cell, every axis counted in the ray's direction of travel
    uint32_t flip;            // widx ^ flip = bit index in the grid copy (axes walked downwards reflected, level bits)
};

// setup_traversal (include/utils_grid.cuh:58-114) in the reference's operation order
__device__ __forceinline__ void walk_span_setup(const nfa_traverse_args &a, const WalkParams &p, const float o[3], const float d[3],
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
    sp.tx = tdist[0]; sp.ty = tdist[1]; sp.tz = tdist[2];
    sp.dx = delta[0]; sp.dy = delta[1]; sp.dz = delta[2];
    sp.rem = (uint32_t)(nst[0] - 1) | ((uint32_t)(nst[1] - 1) << 10) | ((uint32_t)(nst[2] - 1) << 20) | WK_GUARD;
    // an axis walked downwards counts its reflected coordinate (2^nb - 1 - c = c ^ (2^nb - 1)) upwards: every step is "+1"
    uint32_t widx = 0u, flip = (uint32_t)level << p.lay.bits, mk[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const uint32_t M = p.lay.mask[ax];
        const uint32_t dep = bit_deposit((uint32_t)cur[ax], M);
        widx |= stepi[ax] < 0 ? (dep ^ M) : dep;
        flip |= stepi[ax] < 0 ? M : 0u;
        mk[ax] = stepi[ax] != 0 ? M : 0u;
    }
    sp.mx = mk[0]; sp.my = mk[1]; sp.mz = mk[2];
    sp.widx = widx; sp.flip = flip;
}

// One cell of the walk.  (w_cur, i_cur): the word of the grid copy that holds the occupancy bit of the cell the ray is in
// and the bit's index -- requested when the ray entered the cell, one cell's worth of instructions ago; `open`: the kind of
// the ray's open list entry.  Steps the DDA, looks at the current cell's bit, requests the next cell's word into the same
// registers, closes the open entry when the occupancy flips and records the cell's exit distance in the open entry's slot.
__device__ __forceinline__ void walk_cell(float dx, float dy, float dz, uint32_t mx, uint32_t my, uint32_t mz, uint32_t flip, float &tx,
                                          float &ty, float &tz, uint32_t &rem, uint32_t &widx, uint32_t &ev_addr, float &m_out,
                                          uint32_t &w_cur, uint32_t &i_cur, int32_t &open, const uint32_t *__restrict__ bits, char *ev_lds)
{
    const float n = vmin_f32(ty, tz);
    const float m = vmin_f32(tx, n);          // exit distance of this cell (clamped to this_tmax by phase 2)
    // single_traversal (include/utils_grid.cuh:116-142): x if tx < ty && tx < tz, else y if ty < tz, else z.
    // The chosen axis' distance IS m, so its update is m + delta.
    const bool s0 = tx < n;
    const bool s1 = ty < tz;
    const float dsel = s0 ? dx : (s1 ? dy : dz);
    const float nm = m + dsel;
    const float ty1 = s1 ? nm : ty, tz1 = s1 ? tz : nm;
    tx = s0 ? nm : tx;
    ty = s0 ? ty : ty1;
    tz = s0 ? tz : tz1;
    rem -= s0 ? 1u : (s1 ? (1u << 10) : (1u << 20));
    // +1 on the chosen axis' bits of the interleaved index: fill the other bits with ones so that the carry runs through
    // them, add the axis' lowest bit, keep the axis' bits of the sum (a carry out of the top bit is dropped: one step
    // outside the grid wraps to a valid cell, which is never used)
    const uint32_t M = s0 ? mx : (s1 ? my : mz);
    uint32_t filled;
    asm("v_bfi_b32 %0, %1, %2, -1" : "=v"(filled) : "v"(M), "v"(widx));       // (M & widx) | ~M
    filled += M & 7u;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(widx) : "v"(M), "v"(filled), "v"(widx));   // (M & sum) | (~M & widx)
    const uint32_t changed = __builtin_amdgcn_ubfe(w_cur, i_cur, 1u) ^ (uint32_t)open;   // bit (i_cur & 31) of the current cell's word
    i_cur = widx ^ flip;
#if defined(NFA_WALK_EXP) && NFA_WALK_EXP == 1   /* timing experiment: no load at all */
    w_cur = i_cur >> 2;
#elif defined(NFA_WALK_EXP) && NFA_WALK_EXP == 2  /* timing experiment: every load hits one 128-byte line */
    w_cur = bits[(i_cur >> 5) & 31u];
#else
    w_cur = bits[i_cur >> 5];
#endif
    open ^= (int32_t)changed;                 // = the current cell's occupancy
    ev_addr += changed << 10;                 // the open entry is complete when the occupancy flips
    *reinterpret_cast<float *>(ev_lds + ev_addr) = m;
    m_out = m;
}
// stop when a step counter has run out (a guard bit is gone: the span ends) or the open entry sits in the last slot
__device__ __forceinline__ bool walk_stop(uint32_t rem, uint32_t ev_addr)
{
    return ((rem & WK_GUARD) | (ev_addr & WK_FULL)) != WK_GUARD;
}

template <bool FUSED, bool HAS_LIMIT>
NFA_WALK_OCC __global__ __launch_bounds__(256) void walk_kernel(const nfa_traverse_args a, const WalkParams p)
{
    __shared__ __attribute__((aligned(16))) char ev_lds[(WK_EV + 1) * 1024];   // [WK_EV + 1][256] floats
    __shared__ ApproachLds tb;
    {   // read the table from the kernel-argument segment as memory (indexed by thread: as an argument in registers it would
        // occupy 80 scalar registers and be selected entry by entry)
        const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr size_t p_off = (sizeof(nfa_traverse_args) + alignof(WalkParams) - 1) / alignof(WalkParams) * alignof(WalkParams);
        const WalkParams *pk = reinterpret_cast<const WalkParams *>(ka + p_off);
        if (threadIdx.x < APPROACH_MAX) { tb.T[threadIdx.x] = pk->approach.T[threadIdx.x]; tb.q[threadIdx.x] = pk->approach.q[threadIdx.x]; }
        __syncthreads();
    }
    const uint32_t lane_off = 4u * threadIdx.x;
    char *const col = ev_lds + lane_off;
    const float dt = a.step_size;
    const int32_t limit = HAS_LIMIT ? a.traverse_steps_limit : 0;
    const uint32_t *__restrict__ bits = p.bits;
    for (int64_t slot_i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot_i < a.n_rays;
         slot_i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t tid = p.order ? (int64_t)p.order[slot_i] : slot_i;
        if (a.mode == 2 && a.rays_mask != nullptr && !a.rays_mask[tid]) {
            if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
            a.sm_cnts[tid] = 0;
            if (a.iv_cnts) a.iv_cnts[tid] = 0;
            p.run_cnts[tid] = 0;
            continue;
        }
        const float near_plane = a.near_planes[tid], far_plane = a.far_planes[tid];
        const float o[3] = {a.rays_o[3 * tid], a.rays_o[3 * tid + 1], a.rays_o[3 * tid + 2]};
        const float d[3] = {a.rays_d[3 * tid], a.rays_d[3 * tid + 1], a.rays_d[3 * tid + 2]};
        Marcher s;
        s.t_last = near_plane; s.continuous = 0; s.n_samples = 0; s.n_chains = 0; s.n_runs = 0; s.run_inc = 0.f;
        s.span_tmax = 0.f; s.ptype = 0; s.at_near = 1; s.fe = 0xFFFFFFFFu; s.fq = 0u; s.fstep = 0.f; s.frcp = 0.f;

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

        WalkSpan sp;
        sp.tx = sp.ty = sp.tz = sp.dx = sp.dy = sp.dz = 0.f; sp.mx = sp.my = sp.mz = 0u; sp.rem = 0u; sp.widx = 0u; sp.flip = 0u;
        int32_t in_span = 0, has_open = 0, open_type = 0;
        uint32_t w_cur = 0u, i_cur = 0u;   // the word of the grid copy with the current cell's bit, and the bit's index
        uint32_t ev_addr = lane_off;  // byte offset of the open entry's slot: slot << 10 | lane offset
        uint32_t ev_span = 0u;
        float m_last = 0.f;

        for (;;) {
            // ---------------- phase 1
            int32_t finished = 0;
            for (;;) {
                if (!in_span) {
                    const uint32_t used = (ev_addr >> 10) + (uint32_t)has_open;
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
                            level = (int32_t)(idx % G);
                            if (!hits[level]) continue;
                            if (!(idx < G)) {
                                const int64_t nidx = ti[i + 1];
                                if (nidx < G) continue;
                                level = (int32_t)(nidx % G);
                                if (!hits[level]) continue;
                            }
                            this_tmin = fmaxf(ts[i], near_plane); this_tmax = fminf(ts[i + 1], far_plane);
                            if (this_tmin >= this_tmax) continue;
                            found = true;
                        }
                    }
                    if (!found) { finished = 1; break; }
                    // the open entry of the previous span is complete; then the span start: (this_tmin, this_tmax)
                    uint32_t kk = used;
                    *reinterpret_cast<float *>(col + (kk << 10)) = this_tmin;
                    *reinterpret_cast<float *>(col + ((kk + 1u) << 10)) = this_tmax;
                    ev_span |= 1u << kk;
                    walk_span_setup(a, p, o, d, level, this_tmin, this_tmax, sp);
                    const uint32_t idx0 = sp.widx ^ sp.flip;
                    w_cur = bits[idx0 >> 5]; i_cur = idx0;
                    open_type = (int32_t)((w_cur >> (idx0 & 31u)) & 1u);
                    ev_span |= (uint32_t)open_type << (16u + kk);
                    ev_addr = ((kk + 2u) << 10) | lane_off;
                    has_open = 1;
                    in_span = 1;
                }
                // the reference's cell loop (grid.cu:184-272) reduced to the DDA
                float tx = sp.tx, ty = sp.ty, tz = sp.tz;
                const float dx = sp.dx, dy = sp.dy, dz = sp.dz;
                const uint32_t mx = sp.mx, my = sp.my, mz = sp.mz, flip = sp.flip;
                uint32_t rem = sp.rem;
                uint32_t widx = sp.widx;
                do {
                    walk_cell(dx, dy, dz, mx, my, mz, flip, tx, ty, tz, rem, widx, ev_addr, m_last, w_cur, i_cur, open_type, bits, ev_lds);
                } while (!walk_stop(rem, ev_addr));
                sp.tx = tx; sp.ty = ty; sp.tz = tz; sp.rem = rem; sp.widx = widx;
                if ((rem & WK_GUARD) != WK_GUARD) in_span = 0;
                else break;  // list full
            }
            // ---------------- phase 2
            int32_t cnt = (int32_t)(ev_addr >> 10);
            if (finished && has_open) { cnt += 1; has_open = 0; }
#ifndef NFA_WALK_NO_PHASE2
            marcher_run<HAS_LIMIT>(s, col, cnt, ev_span, dt, limit, p, tb, tid);
#endif
            if (finished || (limit > 0 && s.n_samples >= limit)) break;
            // the open entry moves to slot 0
            ev_span = 0u;
            ev_addr = lane_off;
            if (has_open) *reinterpret_cast<float *>(col) = m_last;
        }
        if (a.terminate_planes) a.terminate_planes[tid] = s.t_last;
        a.sm_cnts[tid] = s.n_samples;
        if (a.iv_cnts) a.iv_cnts[tid] = s.n_samples + s.n_chains;  // edges = samples + one leading edge per chain
        // rays with > 2^21 samples go to the serial fill too (the expansion packs a 27-bit batch offset)
        int32_t n_runs = s.n_runs;
        if (s.n_samples > (1 << 21) && n_runs <= p.max_runs) n_runs = p.max_runs + 1;
        p.run_cnts[tid] = n_runs;
        if (n_runs > p.max_runs) atomicAdd(p.overflow, 1);
    }
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
                      int32_t *overflow_count, float near_hint, const int32_t *ray_order, nfa_stream_t stream)
{
    NFA_REQUIRE(pa != nullptr, "traverse_runs: null args");
    const nfa_traverse_args &a = *pa;
    NFA_REQUIRE(a.n_rays >= 0 && a.n_rays < (int64_t)1 << 31, "traverse_runs: n_rays out of range");
    NFA_REQUIRE(overflow_count, "traverse_runs: overflow_count is null");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(overflow_count, 0, sizeof(int32_t), s) != hipSuccess) { set_error("traverse_runs: memset failed"); return NFA_EHIP; }
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
    // near_hint: the value most (or all) entries of near_planes hold, NaN if unknown.  Rays whose near plane
    // differs bit-wise simply do not use the table.
    if (near_hint == near_hint) approach_table_build(p.approach, near_hint, a.step_size);
    else p.approach.n = 0;
    const size_t shmem = 0;  // the lists are static LDS
    const unsigned grid = grid_1d(a.n_rays, 256, 1 << 20);
    const bool lim = a.traverse_steps_limit > 0;
    if (fused && !lim)      hipLaunchKernelGGL((walk_kernel<true, false>), dim3(grid), dim3(256), shmem, s, a, p);
    else if (fused)         hipLaunchKernelGGL((walk_kernel<true, true>), dim3(grid), dim3(256), shmem, s, a, p);
    else if (!lim)          hipLaunchKernelGGL((walk_kernel<false, false>), dim3(grid), dim3(256), shmem, s, a, p);
    else                    hipLaunchKernelGGL((walk_kernel<false, true>), dim3(grid), dim3(256), shmem, s, a, p);
    NFA_CHECK_LAUNCH("traverse_runs");
    return NFA_OK;
}

}  // extern "C"
