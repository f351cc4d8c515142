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
#ifndef NFA_WALK_OP_RL
#define NFA_WALK_OP_RL 6
#endif
static_assert(NFA_WK_EV == 16, "the list-full test is one bit of the slot address: 16 slots");
#ifndef NFA_WALK_LG
#define NFA_WALK_LG 10   /* log2 of the bytes of one list slot = 4 bytes x threads per workgroup: 10 = 256 threads */
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

// Where a ray's run records go: the first RL of them into the lane's LDS column (the one-pass kernel expands them from
// there), the others -- all of them with RL == 0 -- into the slot-major global array runs[slot][ray].
constexpr int WK_REC_STRIDE = WK_THREADS * 8;   // bytes between two slots of one lane: [RL][threads] records of 8 bytes
// RL == WK_RL_AGENT + k: k LDS slots, and the global array written with agent-scope (write-through) 8-byte stores: the
// records are read by a wave of another compute unit, possibly behind another L2, while this launch is still running
// (expand_units_kernel).
constexpr int WK_RL_AGENT = 64;
template <int RL>
__device__ __forceinline__ void store_run(const WalkParams &p, char *rec_col, int32_t slot, int64_t tid, unsigned long long rec)
{
    constexpr int LDS_SLOTS = RL >= WK_RL_AGENT ? RL - WK_RL_AGENT : RL;
    if (LDS_SLOTS > 0 && slot < LDS_SLOTS) *reinterpret_cast<unsigned long long *>(rec_col + slot * WK_REC_STRIDE) = rec;
    else if (slot < p.max_runs) {
        if (RL >= WK_RL_AGENT) __hip_atomic_store((p.runs + tid) + (int64_t)slot * p.n_rays, rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (p.runs + tid)[(int64_t)slot * p.n_rays] = rec;
    }
}

// n samples t0, t0 + inc, ... join the ray's run list; a run record {t_first : f32 | k_start : 31, continues_previous : 1}
// is written when a run starts (its length is the next record's k_start, or the ray's count)
template <int RL>
__device__ __forceinline__ void marcher_emit(Marcher &s, float t0, float inc, uint32_t n, const WalkParams &p, int64_t tid, char *rec_col)
{
    if (!(s.continuous && inc == s.run_inc)) {
        store_run<RL>(p, rec_col, s.n_runs, tid,
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
template <int RL>
__device__ __forceinline__ int marcher_cross(Marcher &s, float thr, int type, float dt, float half, int32_t limit, const WalkParams &p,
                                             const ApproachLds &tb, int64_t tid, char *rec_col)
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
            if (type == WK_OCC) marcher_emit<RL>(s, t, s.fstep, n_b, p, tid, rec_col);
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
        if (type == WK_OCC) marcher_emit<RL>(s, t1, tn - t1, 1u, p, tid, rec_col);
        s.t_last = tn;
        progress = 1;
    }
    marcher_refresh(s, dt);
    return progress ? 2 : 1;
}

// The same entry through march.h's stepper: binade boundaries, exact ties, the way from the near plane (tabulated),
// denormal / huge distances, no-progress steps.
template <int RL>
__device__ __forceinline__ void marcher_general(Marcher &s, float thr, int type, float dt, float half, int32_t limit,
                                                const WalkParams &p, int64_t tid, char *rec_col)
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
        if (emit) marcher_emit<RL>(s, t, inc, n, p, tid, rec_col);
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
template <bool HAS_LIMIT, int RL>
__device__ __forceinline__ void marcher_run(Marcher &s, const char *col /* LDS column of this lane */, int32_t cnt, uint32_t ev_span,
                                            float dt, int32_t limit_arg, const WalkParams &p, const ApproachLds &tb, int64_t tid,
                                            char *rec_col /* LDS column of this lane's run records (RL > 0) */)
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
                store_run<RL>(p, rec_col, s.n_runs, tid,
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
            if (!tried) r = marcher_cross<RL>(s, thr, type, dt, half, limit, p, tb, tid, rec_col);
            tried = r == 1;
            if (r == 0) {
                marcher_general<RL>(s, thr, type, dt, half, limit, p, tid, rec_col);
                if (is_span_entry(ev_span, k)) s.span_tmax = *reinterpret_cast<const float *>(col + ((k + 1) << WK_LG));
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
    sp.tx = tdist[0]; sp.ty = tdist[1]; sp.tz = tdist[2];
    sp.dx = delta[0]; sp.dy = delta[1]; sp.dz = delta[2];
    sp.rem = (uint32_t)(nst[0] - 1) | ((uint32_t)(nst[1] - 1) << 10) | ((uint32_t)(nst[2] - 1) << 20) | WK_GUARD;
    // an axis walked downwards counts its reflected coordinate (2^nb - 1 - c = c ^ (2^nb - 1)) upwards: every step is "+1"
    uint32_t widx = 0u, flip = (uint32_t)level << lay.bits, mk[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const uint32_t M = lay.mask[ax];
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
    ev_addr += changed << WK_LG;                 // the open entry is complete when the occupancy flips
    *reinterpret_cast<float *>(ev_lds + ev_addr) = m;
    m_out = m;
}
// stop when a step counter has run out (a guard bit is gone: the span ends) or the open entry sits in the last slot
__device__ __forceinline__ bool walk_stop(uint32_t rem, uint32_t ev_addr)
{
    return ((rem & WK_GUARD) | (ev_addr & WK_FULL)) != WK_GUARD;
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

// One ray through the grid(s): phases 1 and 2 alternate until the event walk is over.  Leaves the marcher's state (sample
// count, run count, chain count, last distance); run records go to store_run<RL>.
template <bool FUSED, bool HAS_LIMIT, int RL>
__device__ __forceinline__ void walk_ray(const nfa_traverse_args &a, const WalkParams &p, int64_t tid, char *ev_lds, uint32_t lane_off,
                                         char *rec_col, const ApproachLds &tb, Marcher &s)
{
    char *const col = ev_lds + lane_off;
    const float dt = a.step_size;
    const int32_t limit = HAS_LIMIT ? a.traverse_steps_limit : 0;
    const uint32_t *__restrict__ bits = p.bits;
    const float near_plane = a.near_planes[tid], far_plane = a.far_planes[tid];
    const float o[3] = {a.rays_o[3 * tid], a.rays_o[3 * tid + 1], a.rays_o[3 * tid + 2]};
    const float d[3] = {a.rays_d[3 * tid], a.rays_d[3 * tid + 1], a.rays_d[3 * tid + 2]};
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
                w_cur = bits[idx0 >> 5]; i_cur = idx0;
                open_type = (int32_t)((w_cur >> (idx0 & 31u)) & 1u);
                ev_span |= (uint32_t)open_type << (16u + kk);
                ev_addr = ((kk + 2u) << WK_LG) | lane_off;
                has_open = 1;
                in_span = 1;
            }
            // the reference's cell loop (grid.cu:184-272) reduced to the DDA
            float tx = sp.tx, ty = sp.ty, tz = sp.tz;
            const float dx = sp.dx, dy = sp.dy, dz = sp.dz;
            const uint32_t mx = sp.mx, my = sp.my, mz = sp.mz, flip = sp.flip;
            uint32_t rem = sp.rem;
            uint32_t widx = sp.widx;
            // Nothing in the cell loop reads LDS or scalar memory.  Without this the compiler's wait-count pass, which
            // merges the loop header's state with the preheader's, puts an `s_waitcnt lgkmcnt(0)` INSIDE the loop whenever
            // some path into it leaves an LDS read or a kernel-argument load in flight (in the single-launch forms of the one-pass traversal: every cell
            // then waited for its own ds_write, the walk took 3x as long).  lgkmcnt(0), vmcnt / expcnt untouched:
            __builtin_amdgcn_s_waitcnt(0xC07F);
            do {
                walk_cell(dx, dy, dz, mx, my, mz, flip, tx, ty, tz, rem, widx, ev_addr, m_last, w_cur, i_cur, open_type, bits, ev_lds);
            } while (!walk_stop(rem, ev_addr));
            sp.tx = tx; sp.ty = ty; sp.tz = tz; sp.rem = rem; sp.widx = widx;
            if ((rem & WK_GUARD) != WK_GUARD) in_span = 0;
            else break;  // list full
        }
        // ---------------- phase 2
        int32_t cnt = (int32_t)(ev_addr >> WK_LG);
        if (finished && has_open) { cnt += 1; has_open = 0; }
#ifndef NFA_WALK_NO_PHASE2
        marcher_run<HAS_LIMIT, RL>(s, col, cnt, ev_span, dt, limit, p, tb, tid, rec_col);
#endif
        if (finished || (limit > 0 && s.n_samples >= limit)) break;
        // the open entry moves to slot 0
        ev_span = 0u;
        ev_addr = lane_off;
        if (has_open) *reinterpret_cast<float *>(col) = m_last;
    }
}

template <bool FUSED, bool HAS_LIMIT>
NFA_WALK_OCC __global__ __launch_bounds__(WK_THREADS) void walk_kernel(const nfa_traverse_args a, const WalkParams p)
{
    __shared__ __attribute__((aligned(16))) char ev_lds[(WK_EV + 1) << WK_LG];   // [WK_EV + 1][256] floats
    __shared__ ApproachLds tb;
    approach_to_lds(tb, p);
    __syncthreads();
    const uint32_t lane_off = 4u * threadIdx.x;
    const int64_t n_walk = p.order ? p.n_order : a.n_rays;
    for (int64_t slot_i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot_i < n_walk;
         slot_i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t tid = p.order ? (int64_t)p.order[slot_i] : slot_i;
        if (a.mode == 2 && a.rays_mask != nullptr && !a.rays_mask[tid]) {
            if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
            a.sm_cnts[tid] = 0;
            if (a.iv_cnts) a.iv_cnts[tid] = 0;
            p.run_cnts[tid] = 0;
            continue;
        }
        Marcher s;
        walk_ray<FUSED, HAS_LIMIT, 0>(a, p, tid, ev_lds, lane_off, nullptr, tb, s);
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
    i_cur = sp.widx ^ sp.flip;
    w_cur = reinterpret_cast<const unsigned long long *>(p.bits)[i_cur >> 6];
}

// One cell (grid.cu:184-272); true when the span is over (the step left its last cell, or the sample budget is spent).
template <bool SPLIT>
__device__ __forceinline__ bool cone_cell(const nfa_traverse_args &a, const ConeParams &p, int64_t tid, float this_tmax, WalkSpan &sp,
                                          unsigned long long &w_cur, uint32_t &i_cur, ConeRay &st)
{
    const float step_size = a.step_size, cone = a.cone_angle;
    const int32_t limit = a.traverse_steps_limit;
    const float n = vmin_f32(sp.ty, sp.tz);
    const float m = vmin_f32(sp.tx, n);
    const float t_traverse = vmin_f32(m, this_tmax);
    const uint32_t w_half = (i_cur & 32u) ? (uint32_t)(w_cur >> 32) : (uint32_t)w_cur;
    const bool occupied = __builtin_amdgcn_ubfe(w_half, i_cur, 1u) != 0u;   // bit (i_cur & 31) of the half
    // single_traversal (include/utils_grid.cuh:116-142), as in walk_cell
    const bool s0 = sp.tx < n;
    const bool s1 = sp.ty < sp.tz;
    const float dsel = s0 ? sp.dx : (s1 ? sp.dy : sp.dz);
    const float nm = m + dsel;
    const float ty1 = s1 ? nm : sp.ty, tz1 = s1 ? sp.tz : nm;
    sp.tx = s0 ? nm : sp.tx;
    sp.ty = s0 ? sp.ty : ty1;
    sp.tz = s0 ? sp.tz : tz1;
    sp.rem -= s0 ? 1u : (s1 ? (1u << 10) : (1u << 20));
    const uint32_t M = s0 ? sp.mx : (s1 ? sp.my : sp.mz);
    uint32_t filled;
    asm("v_bfi_b32 %0, %1, %2, -1" : "=v"(filled) : "v"(M), "v"(sp.widx));
    filled += M & 7u;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sp.widx) : "v"(M), "v"(filled), "v"(sp.widx));
    const bool done = (sp.rem & WK_GUARD) != WK_GUARD;
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
            if (st.n_runs < p.max_runs)
                p.runs[(int64_t)st.n_runs * a.n_rays + tid] =
                    (unsigned long long)f32_bits(st.t_last) |
                    ((unsigned long long)((uint32_t)st.n_samples | (st.continuous ? 0x80000000u : 0u)) << 32);
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
            ts_col[i * 256] = ts[i];
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
        const float this_tmin = fmaxf(ts_col[i * 256], near_plane);
        const float this_tmax = fminf(ts_col[(i + 1) * 256], far_plane);
        if (ok && this_tmin < this_tmax) {
            span_tmax = this_tmax;
            cone_span_begin(a, p, o, d, level, this_tmin, this_tmax, st, sp, w_cur, i_cur);
            return true;
        }
    }
    return false;
}

__device__ __forceinline__ void cone_ray_out(const nfa_traverse_args &a, const ConeParams &p, int64_t tid, const ConeRay &st)
{
    if (a.terminate_planes) a.terminate_planes[tid] = st.t_last;
    a.sm_cnts[tid] = st.n_samples;
    // rays with > 2^21 samples go to the serial fill (the expansion packs a 27-bit batch offset)
    int32_t n_runs = st.n_runs;
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
__global__ __launch_bounds__(256) void cone_walk_kernel(const nfa_traverse_args a, const ConeParams p)
{
    const int64_t n_walk = p.order ? p.n_order : a.n_rays;
    const int32_t limit = a.traverse_steps_limit;
    for (int64_t slot_i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot_i < n_walk;
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
            while (!cone_cell<NFA_CONE_WALK_SPLIT != 0>(a, p, tid, span_tmax, sp, w_cur, i_cur, st)) {}
            // The budget is spent: the last thing that happened was a sample (continuous), so the spans still to come would
            // change nothing (grid.cu:151,185: no fast-forward, no cell visited).
            if (limit > 0 && st.n_samples >= limit) break;
        }
        cone_ray_out(a, p, tid, st);
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
__global__ __launch_bounds__(256) void cone_refill_kernel(const nfa_traverse_args a, const ConeParams p)
{
    static_assert(!(FUSED && STAGED), "a fused walk has no event list");
    __shared__ float ts_lds[STAGED ? CONE_EV_MAX * 256 : 1];
    float *const ts_col = ts_lds + (STAGED ? threadIdx.x : 0);
    uint32_t ti_pack = 0u, hit_mask = 0u;
    enum { IDLE = 0, SPAN = 1, WALK = 2, FINISH = 3 };
    const int lane = lane_id();
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const int64_t n_walk = p.order ? p.n_order : a.n_rays;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
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
    sp.tx = sp.ty = sp.tz = sp.dx = sp.dy = sp.dz = 0.f; sp.mx = sp.my = sp.mz = 0u; sp.rem = 0u; sp.widx = 0u; sp.flip = 0u;
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
                cone_ray_out(a, p, tid, st);
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
                if (cone_cell<NFA_CONE_REFILL_SPLIT != 0>(a, p, tid, span_tmax, sp, w_cur, i_cur, st))
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

// ------------------------------------------------------------------------------------------------------------------
// traverse_grids without the serial chain walk -> cumsum -> expansion (the sampler's form of it: ray_indices, t_starts,
// t_ends, packed_info; ref: cuda/csrc/grid.cu:405-471 runs count pass, cumsums + host read, fill pass).
//
// The walk is bound by instruction issue and touches almost no memory; the expansion of its run records is a pure
// write stream.  One after the other they take 182 + 20 + 142 us on BASELINE cfg 2.  Here they are two launches that
// run AT THE SAME TIME on two streams:
//
//   walk_publish_kernel  the walk, one workgroup per 256 rays as before; each wave (a UNIT of 64 consecutive rays) leaves
//                        its run records and {samples, runs} per ray in global memory with agent-scope 8-byte stores,
//                        waits for those stores and publishes status[unit] = {AGG, total of the 64 counts};
//   expand_units_kernel  a few persistent waves (launched first, on the second stream) that take units in order from a
//                        counter: a look-back over the status words (nearest published inclusive prefix + the totals in
//                        between: rocPRIM's decoupled look-back, run by whoever expands the unit rather than by whoever
//                        walked it) gives the number of samples in front of the unit as soon as every unit up to it
//                        has been walked; the wave publishes {PFX, inclusive prefix}, writes the unit's rows of
//                        packed_info and expands its records (staged in LDS) into the outputs.
//
// Why two kernels and not one: all waves of a launch get the same register budget.  Three single-launch forms were
// built and measured (bit-exact, all slower than the two serial launches' 345 us): every wave expands the unit it has
// just walked, 418 us (waves idle behind the slowest walk in front of them); persistent waves taking walk and
// expansion jobs from two counters, never waiting while a walk job is left, 612 us (16 unrelated walks per compute unit:
// the four waves of a workgroup must walk neighbouring rays at the same time to share the grid copy's lines in L1 -- the
// walk alone 314 us instead of 203); the same in workgroup-synchronous rounds of 4 adjacent units, 429 us (at 128
// registers only 4 waves fit a SIMD, so every expanding wave displaces a walking one, and the walk needs its 4 waves
// to hide its own latencies).  Across launches the hardware does mix register budgets: 4 walking waves + an expanding one.
//
// Integers only: packed_info is the exact exclusive cumsum of the counts in ray order whatever the order of
// completion (determinism).  The only data that crosses waves are 8-byte words written and read with
// agent-scope atomics (status words, {samples, runs} per ray, run records), each published behind a full wait for the
// producing wave's stores.  The expander waits only for walks, which never wait for anything; it holds a bounded
// number of wave slots, so the walk's workgroups always find room; should the walk not run at all (a queue that is not
// serviced) the expander gives up after a bounded number of polls and reports it, and the host takes the serial form.
// The outputs were sized by the host before the total is known (capacity): nothing is written beyond them, the total
// is reported, and the host falls back to the serial form when it was too small.
// Register budgets: a SIMD has 512 registers per lane; the walk takes 109 (112 allocated), so 4 walking waves leave 64 for
// one expanding wave beside them.  The expander wants 76; held to 64 (8 waves per SIMD) it spills a few values that are
// reloaded once per unit, outside its chunk loop (checked in the generated code: no scratch access and no wait for
// memory inside the loop -- see onepass_expand_chunks).
#ifndef NFA_OP_EXP_WAVES
#define NFA_OP_EXP_WAVES 8
#endif
#ifndef NFA_OP_WALK_WAVES
#define NFA_OP_WALK_WAVES 5
#endif
#if NFA_OP_WALK_WAVES > 0
#define NFA_OP_WALK_OCC __attribute__((amdgpu_waves_per_eu(NFA_OP_WALK_WAVES, NFA_OP_WALK_WAVES)))
#else
#define NFA_OP_WALK_OCC
#endif
#if NFA_OP_EXP_WAVES > 0
#define NFA_OP_EXP_OCC __attribute__((amdgpu_waves_per_eu(NFA_OP_EXP_WAVES, NFA_OP_EXP_WAVES)))
#else
#define NFA_OP_EXP_OCC
#endif
constexpr int OP_RL = NFA_WALK_OP_RL;       // run records per ray staged in LDS by the expander; later ones are read from global memory
constexpr int OP_SENTINEL = 32;             // entry index of "this ray's samples are filled by the serial kernel"
constexpr int OP_WAVES = 4;                 // waves per expander workgroup (they never cooperate)
constexpr unsigned long long ST_AGG = 1ull << 62, ST_PFX = 2ull << 62, ST_VAL = (1ull << 62) - 1ull;

struct OnePassParams {
    unsigned long long *status;   // [n_units], zeroed
    uint32_t *gave_up;            // [1], zeroed: set when the expander stopped waiting for the walk
    unsigned long long *meta;     // [3], zeroed: total samples, sum of wave maxima (sampled), sum of counts (sampled)
    unsigned long long *cnt_runs; // [n_rays] {samples : 32 | runs : 32}, handed from the walk to the expander
    longlong2 *packed_info;       // [n_rays] {start, count}
    float *t_starts, *t_ends;     // [capacity]
    int64_t *ray_indices;         // [capacity]
    int64_t capacity;
    int64_t n_units;              // ceil(n_rays / 64)
    int32_t vec;                  // outputs 16-byte aligned
    uint32_t max_polls;           // patience of a waiting expander wave
    long long *profile;           // NFA_OP_PROFILE builds: 10 words per expander wave, else NULL
};

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// a value every lane holds, moved to scalar registers (the compiler cannot know that a shuffle result is uniform)
__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t wave_ticket(uint32_t *counter)
{
    uint32_t t = 0;
    if (lane_id() == 0) t = atomicAdd(counter, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}

template <bool FUSED, bool HAS_LIMIT>
NFA_OP_WALK_OCC __global__ __launch_bounds__(WK_THREADS) void walk_publish_kernel(const nfa_traverse_args a, const WalkParams p, const OnePassParams q)
{
    __shared__ __attribute__((aligned(16))) char ev_lds[(WK_EV + 1) << WK_LG];   // [WK_EV + 1][256] floats
    __shared__ ApproachLds tb;
    // The rays' first OP_RL run records wait in LDS and leave together when the unit is done.  (Stored one by one from the
    // marcher with write-through stores they stalled the walk: the cell loop's wait for its grid word is a wait for
    // every earlier memory operation of the wave, and a write-through store is acknowledged by memory -- microseconds
    // while the expander's write stream saturates it: the walk took 850 us beside the expander instead of 181.)
    __shared__ __attribute__((aligned(16))) unsigned long long rec_lds[OP_RL * WK_THREADS];   // [OP_RL][threads]
    approach_to_lds(tb, p);
    __syncthreads();
    const int lane = lane_id();
    const uint32_t lane_off = 4u * threadIdx.x;
    char *const rec_col = reinterpret_cast<char *>(rec_lds) + 8u * threadIdx.x;
    const int64_t unit = (int64_t)blockIdx.x * (WK_THREADS / 64) + (threadIdx.x >> 6);
    if (unit >= q.n_units) return;   // (the whole wave)
    const int64_t tid = unit * 64 + lane;
    const bool active = tid < a.n_rays;
    int32_t n = 0, c_real = 0;
    if (active) {
        if (a.mode == 2 && a.rays_mask != nullptr && !a.rays_mask[tid]) {
            if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
        } else {
            Marcher s;
            walk_ray<FUSED, HAS_LIMIT, WK_RL_AGENT + OP_RL>(a, p, tid, ev_lds, lane_off, rec_col, tb, s);
            if (a.terminate_planes) a.terminate_planes[tid] = s.t_last;
            n = s.n_samples;
            c_real = s.n_runs;
            if (n > (1 << 21) && c_real <= p.max_runs) c_real = p.max_runs + 1;   // (27-bit positions inside a unit's window)
        }
    }
    const int64_t total = wave_sum_i64((int64_t)n);
    if (total >= ((int64_t)1 << 30)) c_real = n > 0 ? p.max_runs + 1 : c_real;   // rays of millions of samples: all of them to the serial fill
    if (active) {
        __hip_atomic_store(&q.cnt_runs[tid], (unsigned long long)(uint32_t)n | ((unsigned long long)(uint32_t)c_real << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c_real > p.max_runs) atomicAdd(p.overflow, 1);
    }
    {   // the records kept in LDS: slot-major, one 512-byte line per slot and wave
        const int32_t c_st = (active && c_real <= p.max_runs) ? min(c_real, OP_RL) : 0;
        int32_t c_max = c_st;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c_max = max(c_max, __shfl_xor(c_max, off, 64));
        for (int32_t i = 0; i < c_max; ++i)
            if (i < c_st)
                __hip_atomic_store((p.runs + tid) + (int64_t)i * p.n_rays, *reinterpret_cast<const unsigned long long *>(rec_col + i * WK_REC_STRIDE),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((unit & 7) == 0) {   // coherence sample for the caller's next batch (nfa_exclusive_cumsum_pairs_stats_i64's measure)
        int32_t m = n;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        if (lane == 0) { atomicAdd(&q.meta[1], (unsigned long long)m); atomicAdd(&q.meta[2], (unsigned long long)total); }
    }
    // every store of this wave (run records, counts) has left before the unit is announced
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
        __hip_atomic_store(&q.status[unit], ST_AGG | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A waiting expander wave's patience: after max_polls polls of its own (or, looked at every 1024 polls, once another wave has
// run out of patience) it stops: the walk is not running.
__device__ __forceinline__ bool onepass_give_up(const OnePassParams &q, uint32_t polls)
{
    if (polls > q.max_polls) {
        if (lane_id() == 0) __hip_atomic_store(q.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    if ((polls & 1023u) == 0u) {
        uint32_t g = 0u;
        if (lane_id() == 0) g = __hip_atomic_load(q.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return __builtin_amdgcn_readfirstlane((int)g) != 0;
    }
    return false;
}

// Number of samples in front of unit e, once every unit up to e has been walked; publishes the unit's inclusive prefix.
// false: gave up waiting (q.max_polls polls without the walk making the needed progress).
__device__ __forceinline__ bool onepass_resolve(const OnePassParams &q, int64_t e, int64_t &base)
{
    const int lane = lane_id();
    uint32_t polls = 0;
    // first the unit itself, one 8-byte load per poll (a thousand waves polling 64-word windows is half a TB/s of loads that
    // go past every cache: the walk took 6x as long)
    for (;;) {
        unsigned long long st = 0ull;
        if (lane == 0) st = __hip_atomic_load(&q.status[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(st >> 32)) != 0) break;
        if (onepass_give_up(q, ++polls)) return false;
        __builtin_amdgcn_s_sleep(127);
    }
    for (;;) {
        int64_t acc = 0, total_e = 0;
        int64_t win = e;          // lane l looks at unit win - l; lane 0 of the first window is unit e itself
        bool ready = true;
        for (int hop = 0; ; ++hop) {
            const int64_t idx = win - lane;
            unsigned long long st = ST_PFX;   // (in front of unit 0: prefix 0)
            if (idx >= 0) st = __hip_atomic_load(&q.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t fl = (uint32_t)(st >> 62);
            const bool own = hop == 0 && lane == 0;
            if (hop == 0) total_e = (int64_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)st)) |
                                    ((int64_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(st >> 32)) & 0x3FFFFFFFu) << 32);
            const unsigned long long pfx = __ballot(fl == 2u && !own), inv = __ballot(fl == 0u);
            const int fp = pfx ? __builtin_ctzll(pfx) : 64;                 // nearest unit in front whose inclusive prefix is known
            const unsigned long long need = fp == 64 ? ~0ull : ((2ull << fp) - 1ull);   // lanes 0 .. fp
            if (inv & need) { ready = false; break; }                        // a unit in between is still being walked
            acc += wave_sum_i64((lane <= fp && !own) ? (int64_t)(st & ST_VAL) : 0);
            if (fp < 64) break;
            win -= 64;
        }
        if (ready) {
            acc = uniform_i64(acc);
            base = acc;
            if (lane == 0)
                __hip_atomic_store(&q.status[e], ST_PFX | (unsigned long long)(acc + total_e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        if (onepass_give_up(q, ++polls)) return false;
        __builtin_amdgcn_s_sleep(64);
    }
}

// The chunk loop of an expansion job: 256 outputs per step; the entries that start inside the chunk are scattered into an LDS
// line by a cursor per ray, a "most recent entry" scan gives every output its run (traverse2.hip: expand_runs_kernel).
// SPILL: some ray of the unit has more records than the OP_RL staged in LDS, the others are read from global memory.
// Two instances because a global load anywhere in the loop makes the loop wait for memory: vmcnt counts loads and stores
// together, so the wait for a (never executed) load is a wait for the previous chunk's output stores to be acknowledged --
// the first version ran at 0.5 TB/s (1017 us for cfg 2's 516 MB) until the common case had no load in its loop.
template <bool SPILL>
__device__ __forceinline__ void onepass_expand_chunks(const WalkParams &p, const OnePassParams &q, float dt, int lane, int32_t *slot,
                                                      const unsigned long long *recs, int64_t r0, int64_t tid, int64_t W0, int64_t W1,
                                                      int32_t s_rel, int32_t c, bool ovf, int32_t n)
{
        // cursor over the lane's own entries: entry id = lane | index << 6; its position = s_rel + k_start
    int32_t cur = 0;
    int32_t next_pos = (c > 0 && n > 0) ? s_rel : 0x7FFFFFFF;   // (the first record of a ray starts at its sample 0)
    const int64_t c_first = (W0 / 256) * 256;
    int32_t carry_j = -1;
    for (int64_t cb = c_first; cb < W1; cb += 256) {
        const int32_t lo = (int32_t)(cb - W0), hi = lo + 256;
        {   // (the -1 is made here: as a loop invariant the four registers of the vector were spilled under the kernel's register
            //  budget and reloaded -- a load, hence a wait for the previous chunk's stores -- in every chunk)
            int32_t m1 = -1;
            asm volatile("" : "+v"(m1));
            *reinterpret_cast<int4 *>(slot + 4 * lane) = make_int4(m1, m1, m1, m1);
        }
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            const bool take = next_pos < hi;
            if (!__any(take)) break;
            if (take) {
                slot[next_pos - lo] = lane | ((ovf ? OP_SENTINEL : cur) << 6);
                ++cur;
                next_pos = 0x7FFFFFFF;
                if (cur < c) {
                    const uint32_t hi32 = (!SPILL || cur < OP_RL) ? (uint32_t)(recs[cur * 64 + lane] >> 32)
                                                      : (uint32_t)(__hip_atomic_load((p.runs + tid) + (int64_t)cur * p.n_rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
                    next_pos = s_rel + (int32_t)(hi32 & 0x7FFFFFFFu);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int4 o4 = *reinterpret_cast<const int4 *>(slot + 4 * lane);
        __builtin_amdgcn_wave_barrier();
        // most recent entry at or before each element
        int32_t j4[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int k = 1; k < 4; ++k) j4[k] = j4[k] >= 0 ? j4[k] : j4[k - 1];
        int32_t ah = j4[3];
        { int32_t u = dpp_step<0>(-1, ah); ah = ah >= 0 ? ah : u; }
        { int32_t u = dpp_step<1>(-1, ah); ah = ah >= 0 ? ah : u; }
        { int32_t u = dpp_step<2>(-1, ah); ah = ah >= 0 ? ah : u; }
        { int32_t u = dpp_step<3>(-1, ah); ah = ah >= 0 ? ah : u; }
        { int32_t u = dpp_step<4>(-1, ah); ah = ah >= 0 ? ah : u; }
        { int32_t u = dpp_step<5>(-1, ah); ah = ah >= 0 ? ah : u; }
        int32_t pj = dpp_prev_lane(-1, ah);
        if (pj < 0) pj = carry_j;
#pragma unroll
        for (int k = 0; k < 4; ++k) j4[k] = j4[k] >= 0 ? j4[k] : pj;
        carry_j = nfa::last_lane(j4[3]);

        const int64_t p0 = cb + 4 * lane;
        bool valid[4];
        float ts4[4], te4[4];
        int64_t ri4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t pa = p0 + k;
            const int32_t j = j4[k];
            valid[k] = false;
            ts4[k] = te4[k] = 0.f; ri4[k] = 0;
            const int32_t rl = j & 63, i = j >> 6;
            const int32_t srl = __shfl(s_rel, rl, 64);        // (all lanes take part)
            if (pa < W0 || pa >= W1 || j < 0 || i == OP_SENTINEL) continue;   // sentinel: filled by the serial kernel
            const unsigned long long rec = (!SPILL || i < OP_RL) ? recs[i * 64 + rl]
                                                     : __hip_atomic_load(p.runs + (int64_t)i * p.n_rays + r0 + rl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float t0 = bits_f32((uint32_t)rec);
            const uint32_t kk = (uint32_t)((int32_t)(pa - W0) - srl) - ((uint32_t)(rec >> 32) & 0x7FFFFFFFu);
            const float inc = (t0 + dt) - t0;  // the run's exact per-step increment
            // t0 + k * inc is exactly representable for every sample of a run: one fused multiply-add reproduces the serial sums
            ts4[k] = __builtin_fmaf((float)kk, inc, t0);
            te4[k] = __builtin_fmaf((float)(kk + 1), inc, t0);
            ri4[k] = r0 + rl;
            valid[k] = true;
        }
        if (q.vec && valid[0] && valid[1] && valid[2] && valid[3]) {
            store_f4<true>(q.t_starts + p0, ts4[0], ts4[1], ts4[2], ts4[3]);
            store_f4<true>(q.t_ends + p0, te4[0], te4[1], te4[2], te4[3]);
            store_l2<true>(q.ray_indices + p0, ri4[0], ri4[1]);
            store_l2<true>(q.ray_indices + p0 + 2, ri4[2], ri4[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (valid[k]) { q.t_starts[p0 + k] = ts4[k]; q.t_ends[p0 + k] = te4[k]; q.ray_indices[p0 + k] = ri4[k]; }
        }
    }
}

// persistent: every wave takes units in order until none is left
NFA_OP_EXP_OCC __global__ __launch_bounds__(64 * OP_WAVES) void expand_units_kernel(int64_t n_rays, float dt, const WalkParams p, const OnePassParams q,
                                                                    int64_t *__restrict__ sm_cnts)
{
    __shared__ __attribute__((aligned(16))) unsigned long long rec_lds[OP_WAVES][OP_RL * 64];   // staged run records, [slot][lane]
    __shared__ __attribute__((aligned(16))) int32_t slot_lds[OP_WAVES][256];                    // entry that starts at each output of a chunk
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int32_t *const slot = slot_lds[wave];
    unsigned long long *const recs = rec_lds[wave];
#ifndef NFA_OP_NOPRIO
    __builtin_amdgcn_s_setprio(3);   // few waves beside many walking ones: they must not starve for issue slots
#endif
#ifdef NFA_OP_PROFILE
    long long pf[6] = {0, 0, 0, 0, 0, 0};   // ticket, resolve, header, chunks, units, -
    const long long pf_t0 = wall_clock64();
#define PF_T() clock64()
#define PF_ADD(i, t) pf[i] += clock64() - (t)
#else
#define PF_T() 0
#define PF_ADD(i, t) (void)(t)
#endif
    // Units are dealt round-robin: wave w of the grid takes units w, w + n_waves, ...  (No counter: a returning atomic on ONE
    // address, and likewise a write-through load of one address, is served by memory at about 60 ns apiece whoever asks; one
    // ticket per unit -- 16 384 of them on cfg 2 -- put a floor of 1 ms under the kernel however many waves it had, the atomic
    // itself took 18 us and every other memory operation queued behind it.  Tickets for blocks of 8 units removed the floor
    // but a wave then sat on its block until the walk reached it and needed 200 us for it afterwards.)  Neighbouring units go
    // to different waves, so the dense part of an image is spread over all of them.  A workgroup of the grid that is not
    // resident delays its own units only: nobody waits for an expander wave.
    const int32_t n_waves = (int32_t)gridDim.x * OP_WAVES, n_units = (int32_t)q.n_units;       // (n_units < 2^25; scalars)
    int32_t e = __builtin_amdgcn_readfirstlane((int32_t)blockIdx.x * OP_WAVES + wave);
    for (; e < n_units;) {
        int64_t base = 0;
        const long long t_b = PF_T();
        if (!onepass_resolve(q, e, base)) break;
        PF_ADD(1, t_b);
        const long long t_c = PF_T();
        const int64_t unit = e;
        e += n_waves;
        const int64_t r0 = unit * 64, tid = r0 + lane;
        const bool active = tid < n_rays;
        unsigned long long cr = 0ull;
        if (active) cr = __hip_atomic_load(&q.cnt_runs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t n = (int32_t)(uint32_t)cr, c_real = (int32_t)(uint32_t)(cr >> 32);
        const int64_t incl = wave_incl_sum_i64((int64_t)n);
        const int64_t total = uniform_i64(__shfl(incl, 63, 64));
        if (active) {
            q.packed_info[tid] = make_longlong2(base + incl - n, (int64_t)n);
            p.run_cnts[tid] = c_real;
            if (sm_cnts) sm_cnts[tid] = n;
        }
        if (unit == q.n_units - 1 && lane == 0) q.meta[0] = (unsigned long long)(base + total);
        const int64_t W0 = base, W1 = min(base + total, q.capacity);
        if (total >= ((int64_t)1 << 30) || W1 <= W0) continue;
        const bool ovf = c_real > p.max_runs;
        const int32_t c = ovf ? 1 : c_real;
        {   // stage the rays' first OP_RL records in LDS (slot-major in memory: one 512-byte line per slot, independent loads)
            const int32_t c_st = ovf ? 0 : min(c, OP_RL);
            int32_t c_max = c_st;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c_max = max(c_max, __shfl_xor(c_max, off, 64));
            for (int i0 = 0; i0 < c_max; i0 += 4) {   // (four loads in flight per lane)
                unsigned long long rec[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    rec[u] = (i0 + u < OP_RL && i0 + u < c_st) ? __hip_atomic_load((p.runs + tid) + (int64_t)(i0 + u) * p.n_rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < OP_RL && i0 + u < c_max) recs[(i0 + u) * 64 + lane] = rec[u];
            }
        }
        const int32_t s_rel = (int32_t)(incl - n);          // the ray's first output, relative to W0
        PF_ADD(2, t_c);
        const long long t_d = PF_T();
        if (__any(c > OP_RL)) onepass_expand_chunks<true>(p, q, dt, lane, slot, recs, r0, tid, W0, W1, s_rel, c, ovf, n);
        else onepass_expand_chunks<false>(p, q, dt, lane, slot, recs, r0, tid, W0, W1, s_rel, c, ovf, n);
        PF_ADD(3, t_d);
#ifdef NFA_OP_PROFILE
        pf[4]++;
#endif
        __builtin_amdgcn_wave_barrier();
    }
#ifdef NFA_OP_PROFILE
    if (lane == 0 && q.profile) {
        long long *o = q.profile + ((int64_t)blockIdx.x * OP_WAVES + wave) * 10;
        for (int i = 0; i < 6; ++i) o[i] = pf[i];
        o[8] = pf_t0; o[9] = wall_clock64();
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
    p.n_order = ray_order ? n_order : a.n_rays;
    NFA_REQUIRE(!ray_order || (n_order >= 0 && n_order <= a.n_rays), "traverse_runs: n_order out of range");
    if (ray_order && n_order == 0) return NFA_OK;
    // near_hint: the value most (or all) entries of near_planes hold, NaN if unknown.  Rays whose near plane
    // differs bit-wise simply do not use the table.
    if (near_hint == near_hint) approach_table_build(p.approach, near_hint, a.step_size);
    else p.approach.n = 0;
    const size_t shmem = 0;  // the lists are static LDS
    const unsigned grid = grid_1d(p.n_order, WK_THREADS, 1 << 20);
    const bool lim = a.traverse_steps_limit > 0;
    if (fused && !lim)      hipLaunchKernelGGL((walk_kernel<true, false>), dim3(grid), dim3(WK_THREADS), shmem, s, a, p);
    else if (fused)         hipLaunchKernelGGL((walk_kernel<true, true>), dim3(grid), dim3(WK_THREADS), shmem, s, a, p);
    else if (!lim)          hipLaunchKernelGGL((walk_kernel<false, false>), dim3(grid), dim3(WK_THREADS), shmem, s, a, p);
    else                    hipLaunchKernelGGL((walk_kernel<false, true>), dim3(grid), dim3(WK_THREADS), shmem, s, a, p);
    NFA_CHECK_LAUNCH("traverse_runs");
    return NFA_OK;
}

int nfa_traverse_cone_walk(const nfa_traverse_args *pa, const uint32_t *bits, int32_t *run_cnts, uint64_t *runs, int32_t max_runs,
                           int32_t *overflow_count, const int32_t *ray_order, int64_t n_order, nfa_stream_t stream)
{
    NFA_REQUIRE(pa != nullptr, "traverse_cone_walk: null args");
    const nfa_traverse_args &a = *pa;
    NFA_REQUIRE(a.n_rays >= 0 && a.n_rays < (int64_t)1 << 31, "traverse_cone_walk: n_rays out of range");
    NFA_REQUIRE(overflow_count, "traverse_cone_walk: overflow_count is null");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(overflow_count, 0, sizeof(int32_t), s) != hipSuccess) { set_error("traverse_cone_walk: memset failed"); return NFA_EHIP; }
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
    const char *refill_env = getenv("NFA_REFILL");  // "0": one ray per lane also for limited walks; "chunk,min_busy": tuning
    const char *refill_all = getenv("NFA_REFILL_ALL");   // "1": the refilling kernel for unlimited walks too (measurements)
    if ((a.traverse_steps_limit > 0 || (refill_all && refill_all[0] == '1')) && !(refill_env && refill_env[0] == '0')) {
        // entries per wave: enough of them that a lane is refilled several times, as long as the launch still fills the chip
        int64_t chunk = ((p.n_order + 4095) / 4096 + 63) / 64 * 64;
        chunk = chunk < 64 ? 64 : (chunk > 1024 ? 1024 : chunk);
        int min_busy = 48;
        if (refill_env) { long c = 0; int m = 0; if (sscanf(refill_env, "%ld,%d", &c, &m) == 2 && c >= 64 && m >= 1 && m <= 64) { chunk = c / 64 * 64; min_busy = m; } }
        p.chunk = (int32_t)chunk; p.min_busy = min_busy;
        const int64_t n_waves = (p.n_order + chunk - 1) / chunk;
        const unsigned grid = (unsigned)((n_waves + 3) / 4);
        const char *staged_env = getenv("NFA_CONE_STAGED");   // "0": event lists read from memory (A/B)
        const bool staged = !fused && 2 * a.n_grids <= CONE_EV_MAX && !(staged_env && staged_env[0] == '0');
        if (fused)       hipLaunchKernelGGL((cone_refill_kernel<true, false>), dim3(grid), dim3(256), 0, s, a, p);
        else if (staged) hipLaunchKernelGGL((cone_refill_kernel<false, true>), dim3(grid), dim3(256), 0, s, a, p);
        else             hipLaunchKernelGGL((cone_refill_kernel<false, false>), dim3(grid), dim3(256), 0, s, a, p);
    } else {
        p.chunk = 64; p.min_busy = 64;
        const unsigned grid = grid_1d(p.n_order, 256, 1 << 20);
        if (fused) hipLaunchKernelGGL((cone_walk_kernel<true>), dim3(grid), dim3(256), 0, s, a, p);
        else       hipLaunchKernelGGL((cone_walk_kernel<false>), dim3(grid), dim3(256), 0, s, a, p);
    }
    NFA_CHECK_LAUNCH("traverse_cone_walk");
    return NFA_OK;
}

int64_t nfa_traverse_onepass_scratch_words(int64_t n_rays) { return 8 + (n_rays + 63) / 64 + n_rays; }

// scratch: [0] total samples, [1], [2] coherence sums, [3] rays with too many runs, [5] "the expander gave
// up", [8 ...] one status word per unit of 64 rays (all of that zeroed by _begin), then {samples, runs} per ray
static void onepass_params(OnePassParams &q, int64_t n_rays, int64_t *scratch)
{
    const int64_t n_units = (n_rays + 63) / 64;
    q.meta = reinterpret_cast<unsigned long long *>(scratch);
    q.gave_up = reinterpret_cast<uint32_t *>(scratch + 5);
    q.status = reinterpret_cast<unsigned long long *>(scratch + 8);
    q.cnt_runs = reinterpret_cast<unsigned long long *>(scratch + 8 + n_units);
    q.packed_info = nullptr; q.t_starts = q.t_ends = nullptr; q.ray_indices = nullptr;
    q.capacity = 0; q.n_units = n_units; q.vec = 0; q.max_polls = 0;
    q.profile = nullptr;
#ifdef NFA_OP_PROFILE
    {   // debugging aid: env NFA_OP_PROFILE_PTR = address of a device buffer of 10 int64 per expander wave
        const char *e = getenv("NFA_OP_PROFILE_PTR");
        if (e) q.profile = reinterpret_cast<long long *>(strtoull(e, nullptr, 0));
    }
#endif
}

int nfa_traverse_onepass_begin(int64_t n_rays, int64_t *scratch, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_rays < (int64_t)1 << 31 && scratch, "traverse_onepass_begin: bad arguments");
    if (hipMemsetAsync(scratch, 0, sizeof(int64_t) * (size_t)(8 + (n_rays + 63) / 64), as_stream(stream)) != hipSuccess) {
        set_error("traverse_onepass_begin: memset failed");
        return NFA_EHIP;
    }
    return NFA_OK;
}

int nfa_traverse_onepass_expand(int64_t n_rays, float step_size, int32_t *run_cnts, const uint64_t *runs, int32_t max_runs,
                                int64_t *packed_info, float *t_starts, float *t_ends, int64_t *ray_indices, int64_t capacity,
                                int64_t *sm_cnts, int64_t *scratch, int32_t n_workgroups, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_rays < (int64_t)1 << 31 && scratch, "traverse_onepass_expand: bad arguments");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(run_cnts && runs && packed_info, "traverse_onepass_expand: null pointer");
    NFA_REQUIRE(capacity >= 0 && (capacity == 0 || (t_starts && t_ends && ray_indices)), "traverse_onepass_expand: outputs missing");
    NFA_REQUIRE(max_runs >= 1 && max_runs <= 32, "traverse_onepass_expand: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f, "traverse_onepass_expand: step_size must be > 0");
    NFA_REQUIRE(n_workgroups >= 1 && n_workgroups <= 4096, "traverse_onepass_expand: 1..4096 workgroups");
    WalkParams p;
    memset(&p, 0, sizeof(p));
    p.run_cnts = run_cnts;
    p.runs = reinterpret_cast<unsigned long long *>(const_cast<uint64_t *>(runs));
    p.max_runs = max_runs;
    p.n_rays = n_rays;
    OnePassParams q;
    onepass_params(q, n_rays, scratch);
    q.packed_info = reinterpret_cast<longlong2 *>(packed_info);
    q.t_starts = t_starts; q.t_ends = t_ends; q.ray_indices = ray_indices;
    q.capacity = capacity;
    q.vec = ((reinterpret_cast<uintptr_t>(t_starts) | reinterpret_cast<uintptr_t>(t_ends) | reinterpret_cast<uintptr_t>(ray_indices)) & 15) == 0;
    q.max_polls = 50000u;   // ~0.1 s of waiting for one unit: the walk is not running
    const unsigned grid = (unsigned)min((int64_t)n_workgroups, (q.n_units + OP_WAVES - 1) / OP_WAVES);
    hipLaunchKernelGGL(expand_units_kernel, dim3(grid), dim3(64 * OP_WAVES), 0, as_stream(stream), n_rays, step_size, p, q, sm_cnts);
    NFA_CHECK_LAUNCH("traverse_onepass_expand");
    return NFA_OK;
}

int nfa_traverse_onepass_walk(const nfa_traverse_args *pa, const uint32_t *bits, uint64_t *runs, int32_t max_runs, float near_hint,
                              int64_t *scratch, nfa_stream_t stream)
{
    NFA_REQUIRE(pa != nullptr, "traverse_onepass_walk: null args");
    const nfa_traverse_args &a = *pa;
    NFA_REQUIRE(a.n_rays >= 0 && a.n_rays < (int64_t)1 << 31, "traverse_onepass_walk: n_rays out of range");
    NFA_REQUIRE(scratch, "traverse_onepass_walk: scratch is null");
    if (a.n_rays == 0) return NFA_OK;
    NFA_REQUIRE(a.step_size > 0.0f && a.cone_angle == 0.0f, "traverse_onepass_walk: needs step_size > 0 and cone_angle == 0");
    NFA_REQUIRE(a.mode == 0 || a.mode == 2, "traverse_onepass_walk: mode must be 0 (all rays) or 2 (rays_mask + limit)");
    NFA_REQUIRE(a.mode != 2 || a.traverse_steps_limit > 0, "traverse_steps_limit must be > 0 when over_allocate is true");
    NFA_REQUIRE(a.rays_o && a.rays_d && a.aabbs && a.near_planes && a.far_planes && bits && runs, "traverse_onepass_walk: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs <= 32, "traverse_onepass_walk: max_runs must be in [1, 32]");
    NFA_REQUIRE(a.n_grids >= 1 && a.res[0] > 0 && a.res[1] > 0 && a.res[2] > 0, "traverse_onepass_walk: bad grid shape");
    NFA_REQUIRE(a.res[0] <= WK_MAX_RES && a.res[1] <= WK_MAX_RES && a.res[2] <= WK_MAX_RES,
                "traverse_onepass_walk: at most 512 cells per axis (use nfa_traverse_grids beyond)");
    const bool fused = !a.t_sorted && !a.t_indices && !a.hits;
    NFA_REQUIRE(fused || (a.t_sorted && a.t_indices && a.hits), "traverse_onepass_walk: t_sorted, t_indices and hits must be given together");
    NFA_REQUIRE(!fused || a.n_grids == 1, "traverse_onepass_walk: in-kernel intersection supports one grid");
    WalkParams p;
    p.bits = bits;
    p.lay = walk_layout(a.res);
    NFA_REQUIRE(p.lay.bits >= 5 && ((int64_t)a.n_grids << p.lay.bits) < ((int64_t)1 << 31), "traverse_onepass_walk: grid too large");
    p.run_cnts = nullptr;
    p.runs = reinterpret_cast<unsigned long long *>(runs);
    p.max_runs = max_runs;
    p.n_rays = a.n_rays;
    p.overflow = reinterpret_cast<int32_t *>(scratch + 3);
    p.order = nullptr;
    p.n_order = a.n_rays;
    if (near_hint == near_hint) approach_table_build(p.approach, near_hint, a.step_size);
    else p.approach.n = 0;
    OnePassParams q;
    onepass_params(q, a.n_rays, scratch);
    hipStream_t s = as_stream(stream);
    constexpr int UPW = WK_THREADS / 64;
    const unsigned grid = (unsigned)((q.n_units + UPW - 1) / UPW);
    const bool lim = a.traverse_steps_limit > 0;
    if (fused && !lim)      hipLaunchKernelGGL((walk_publish_kernel<true, false>), dim3(grid), dim3(WK_THREADS), 0, s, a, p, q);
    else if (fused)         hipLaunchKernelGGL((walk_publish_kernel<true, true>), dim3(grid), dim3(WK_THREADS), 0, s, a, p, q);
    else if (!lim)          hipLaunchKernelGGL((walk_publish_kernel<false, false>), dim3(grid), dim3(WK_THREADS), 0, s, a, p, q);
    else                    hipLaunchKernelGGL((walk_publish_kernel<false, true>), dim3(grid), dim3(WK_THREADS), 0, s, a, p, q);
    NFA_CHECK_LAUNCH("traverse_onepass_walk");
    return NFA_OK;
}

}  // extern "C"
