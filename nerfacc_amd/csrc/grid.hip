// grid.hip -- ray/AABB intersection, occupancy-grid traversal, int64 cumsum, pack_info.
//
// Hand-written for gfx950 (wave64).  Semantics follow the reference kernels cited below
// (paths relative to /root/reference/nerfacc/cuda/csrc/); the structure does not: one
// templated marcher shared by the count / fill / over-allocated launches, optional in-kernel
// intersection, direct (t_starts, t_ends, ray_indices) emission for the sampler, a device-side
// int64 scan instead of torch::cumsum, and launch errors are reported.
//
// Floating-point contract: this TU is compiled with -ffp-contract=off; every expression is
// un-fused IEEE fp32 in the reference's source order so that sample counts are bit-identical
// to oracle/nerfacc_oracle.c.
#include <stdarg.h>

#include "common.hip.h"
#include "march.h"

namespace nfa {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// knobs of the A/B tests and measurement scripts (nfa_set_tuning): a handful of {name, value} pairs
struct TuningKnob { char name[32]; char value[64]; };
static TuningKnob g_knobs[8];
static int g_n_knobs = 0;
const char *tuning_env(const char *name)
{
    for (int i = 0; i < g_n_knobs; ++i)
        if (strcmp(g_knobs[i].name, name) == 0) return g_knobs[i].value[0] ? g_knobs[i].value : nullptr;
    return nullptr;
}

// ------------------------------------------------------------------------------------------
// Slab test, grid.cu:284-313 / include/utils_grid.cuh:10-55.
__device__ __forceinline__ bool slab_test(const float o[3], const float inv[3], const float *bmin,
                                          const float *bmax, float near, float far, float &tmin,
                                          float &tmax)
{
    float lo, hi;
    if (inv[0] >= 0) { tmin = (bmin[0] - o[0]) * inv[0]; tmax = (bmax[0] - o[0]) * inv[0]; }
    else             { tmin = (bmax[0] - o[0]) * inv[0]; tmax = (bmin[0] - o[0]) * inv[0]; }
#pragma unroll
    for (int a = 1; a < 3; ++a) {
        if (inv[a] >= 0) { lo = (bmin[a] - o[a]) * inv[a]; hi = (bmax[a] - o[a]) * inv[a]; }
        else             { lo = (bmax[a] - o[a]) * inv[a]; hi = (bmin[a] - o[a]) * inv[a]; }
        if (tmin > hi || lo > tmax) return false;
        if (lo > tmin) tmin = lo;
        if (hi < tmax) tmax = hi;
    }
    if (tmax <= 0) return false;
    tmin = fmaxf(tmin, near);
    tmax = fminf(tmax, far);
    return true;
}

__global__ __launch_bounds__(256) void ray_aabb_kernel(const float *__restrict__ rays_o,
                                                       const float *__restrict__ rays_d, int64_t n_rays,
                                                       const float *__restrict__ aabbs, int32_t n_aabbs,
                                                       float near, float far, float miss,
                                                       float *__restrict__ t_mins, float *__restrict__ t_maxs,
                                                       uint8_t *__restrict__ hits)
{
    const int64_t numel = n_rays * n_aabbs;
    for (int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < numel;
         tid += (int64_t)blockDim.x * gridDim.x) {
        const int64_t r = tid / n_aabbs;
        const int32_t g = (int32_t)(tid - r * n_aabbs);
        const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const float inv[3] = {1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2]};
        float tmin, tmax;
        const bool hit = slab_test(o, inv, aabbs + 6 * g, aabbs + 6 * g + 3, near, far, tmin, tmax);
        t_mins[tid] = hit ? tmin : miss;
        t_maxs[tid] = hit ? tmax : miss;
        hits[tid] = hit ? 1 : 0;
    }
}

// The event list the traversal walks (grid.py:156-162 of the reference: ray_aabb_intersect with its defaults near = -inf,
// far = +inf, miss = +inf, then torch.sort(torch.cat([t_mins, t_maxs], -1), -1)) in one pass: a thread intersects its
// ray with the G boxes and sorts the 2G distances in registers.  The sort is STABLE (an insertion sort that moves an
// element only past strictly larger ones): ties keep the order of cat([t_mins, t_maxs]) -- the reference's torch.sort
// leaves their order unspecified; ties are the +inf pairs of the boxes a ray misses, which the event walk skips whatever
// their order (grid.cu:129-150), or measure-zero geometry.  With torch: a cat, a generic bitonic sort of 2G-element rows
// and three temporaries -- 0.6 ms per 2 M rays and 4 levels (cfg 5's step), this kernel 0.07 ms.
template <int G>
__global__ __launch_bounds__(256) void ray_events_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                         int64_t n_rays, const float *__restrict__ aabbs,
                                                         float *__restrict__ t_sorted, int64_t *__restrict__ t_indices,
                                                         uint8_t *__restrict__ hits)
{
    const float inf = __builtin_inff();
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)blockDim.x * gridDim.x) {
        const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const float inv[3] = {1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2]};
        float v[2 * G];
        int32_t id[2 * G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float tmin, tmax;
            const bool hit = slab_test(o, inv, aabbs + 6 * g, aabbs + 6 * g + 3, -inf, inf, tmin, tmax);
            v[g] = hit ? tmin : inf;
            v[G + g] = hit ? tmax : inf;
            hits[r * G + g] = hit ? 1 : 0;
        }
#pragma unroll
        for (int k = 0; k < 2 * G; ++k) id[k] = k;
        // insertion sort as a network of compare-exchanges on neighbours (fully unrolled: everything stays in registers);
        // an exchange happens only when the left element is strictly larger, so equal elements never pass each other
#pragma unroll
        for (int i = 1; i < 2 * G; ++i) {
#pragma unroll
            for (int j = i; j > 0; --j) {
                const bool sw = v[j - 1] > v[j];
                const float a = v[j - 1], b = v[j];
                const int32_t ia = id[j - 1], ib = id[j];
                v[j - 1] = sw ? b : a; v[j] = sw ? a : b;
                id[j - 1] = sw ? ib : ia; id[j] = sw ? ia : ib;
            }
        }
#pragma unroll
        for (int k = 0; k < 2 * G; ++k) {
            t_sorted[r * (2 * G) + k] = v[k];
            t_indices[r * (2 * G) + k] = (int64_t)id[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Traversal, grid.cu:68-282; helpers include/utils_grid.cuh:58-142.

// calc_dt / fast_forward: march.h (exact O(#binades) form of the loops at grid.cu:153-163, :196-205)

enum { EMIT_NONE = 0, EMIT_API = 1, EMIT_DIRECT = 2, EMIT_RUNS = 3 };

// EMIT_RUNS: a count pass that also leaves run records (same format as traverse2.hip's: {t_first : f32 | k_start : 31,
// continues_previous : 1}, slot-major).  With a cone angle a chain of samples is the recurrence t <- t + max(step,
// t * cone) from its first distance, so {t_first, k_start} determines every sample of it; chains are cut every
// CONE_RUN_CAP samples so that the expansion (expand_runs_kernel<EXP_CONE>) iterates the recurrence at most that often
// per output.  The second walk of the fill pass becomes a coalesced expansion.
#ifndef NFA_REFILL_SPLIT
#define NFA_REFILL_SPLIT 1
#endif
#ifndef NFA_TRAVERSE_SPLIT
#define NFA_TRAVERSE_SPLIT 0
#endif
constexpr int CONE_RUN_CAP = 64;
struct RunOut {
    int32_t *run_cnts;          // [n_rays]
    unsigned long long *runs;   // [max_runs, n_rays]
    int32_t max_runs;
    int32_t *overflow;          // [1]
    const int32_t *order;       // lane -> ray assignment (nfa_bin_rays / nfa_bin_rays_levels) or NULL
    int64_t n_order;            // its entries (< n_rays: only the listed rays are walked; the others keep their outputs)
};

struct RayState {
    float t_last;
    int32_t continuous;     // 0 / 1 (an integer: a bool carried across the loops lives in scalar lane masks)
    int32_t n_intervals;
    int32_t n_samples;
    int32_t brick_id;       // brick cache (when the brick-packed grid is given)
    uint32_t brick_lo, brick_hi;
    int32_t n_runs, run_len;  // EMIT_RUNS: records written, samples in the open record
};

// The walk of one [this_tmin, this_tmax) span inside grid `level`, as a resumable pair: span_begin() sets the DDA up,
// span_cell() visits one cell and says whether the span is over.  traverse_span() runs them back to back (one ray per lane
// from start to end); traverse_refill_kernel interleaves the cells of different rays on one lane.
struct SpanState {
    float tdist[3], delta[3];
    int32_t step[3], cur[3], overflow[3];
    float this_tmax;
    int32_t level;
    int32_t cells_left;  // safety cap, never binding for a valid DDA
};

// 4x4x4 bricks: bit ((x&3)<<4 | (y&3)<<2 | (z&3)) of word ((x>>2)*by + (y>>2))*bz + (z>>2)
__device__ __forceinline__ int32_t brick_index(const nfa_traverse_args &a, const SpanState &sp)
{
    const int32_t by = (a.res[1] + 3) >> 2, bz = (a.res[2] + 3) >> 2, bx = (a.res[0] + 3) >> 2;
    const int32_t brick_base = sp.level * bx * by * bz;
    return brick_base + (int32_t)mad_u24(mad_u24((uint32_t)sp.cur[0] >> 2, (uint32_t)by, (uint32_t)sp.cur[1] >> 2), (uint32_t)bz, (uint32_t)sp.cur[2] >> 2);
}

__device__ __forceinline__ void span_begin(const nfa_traverse_args &a, const float o[3], const float d[3], const float inv[3],
                                           int32_t level, float this_tmin, float this_tmax, RayState &st, SpanState &sp)
{
    const float eps = 1e-6f;  // grid.cu:95
    if (!st.continuous) st.t_last = fast_forward(st.t_last, this_tmin, a.step_size, a.cone_angle);

    const float *bmin = a.aabbs + 6 * level, *bmax = bmin + 3;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float resf = (float)a.res[ax];
        const float extent = bmax[ax] - bmin[ax];
        const float voxel = extent / resf;
        const float ray_start = o[ax] + d[ax] * (this_tmin + eps);
        const float ray_end = o[ax] + d[ax] * (this_tmax - eps);
        int32_t c = (int32_t)(((ray_start - bmin[ax]) / extent) * resf);  // v_cvt_i32_f32 saturates
        int32_t f = (int32_t)(((ray_end - bmin[ax]) / extent) * resf);
        c = max(0, min(c, a.res[ax] - 1));
        f = max(0, min(f, a.res[ax] - 1));
        const int32_t start_index = c + (d[ax] > 0.0f ? 1 : 0);
        const float tmax_ax = ((bmin[ax] + (((float)start_index * voxel) - ray_start)) * inv[ax]) + this_tmin;
        const float step_f = (d[ax] == 0.0f) ? 0.0f : (d[ax] > 0.0f ? 1.0f : -1.0f);
        sp.tdist[ax] = (d[ax] == 0.0f) ? this_tmax : tmax_ax;
        sp.step[ax] = (int32_t)step_f;
        const float delta_tmp = voxel * inv[ax] * step_f;
        sp.delta[ax] = (d[ax] == 0.0f) ? this_tmax : delta_tmp;
        sp.cur[ax] = c;
        sp.overflow[ax] = f + sp.step[ax];
    }
    sp.this_tmax = this_tmax;
    sp.level = level;
    sp.cells_left = a.res[0] + a.res[1] + a.res[2] + 3;
    if (a.bricks != nullptr) {
        const int32_t bid = brick_index(a, sp);
        if (bid != st.brick_id) {
            st.brick_id = bid;
            const unsigned long long w = a.bricks[bid];  // (the 1-bit mask would be a second, dependent access)
            st.brick_lo = (uint32_t)w; st.brick_hi = (uint32_t)(w >> 32);
        }
    }
}

// One cell of the span; true when the span is over (its last cell, or the sample budget spent).
//
// The body is written for the scalar unit as much as for the vector ALUs: with unrelated rays every `if`, `break` and bool
// that lives across a branch becomes a handful of 64-bit mask operations on the CU's single scalar pipe, and the first
// version of this loop issued more scalar than vector instructions (8.1 G vs 6.1 G per launch on cfg 5: the scalar pipe was
// the bottleneck).  Hence: one exit, flags in integer registers, selects instead of branches, and the brick word of the NEXT
// cell requested before the current cell is marched (the DDA does not depend on the march), so that the load's latency
// overlaps the march.
template <int EMIT, bool HAS_IV, bool HAS_SM, bool SPLIT = false>
__device__ __forceinline__ bool span_cell(const nfa_traverse_args &a, int64_t tid, int64_t iv_base, int64_t sm_base,
                                          SpanState &sp, RayState &st, const RunOut &ro)
{
    const float step_size = a.step_size, cone = a.cone_angle;
    const int32_t limit = a.traverse_steps_limit;
    float (&tdist)[3] = sp.tdist;
    int32_t (&cur)[3] = sp.cur;
    const float this_tmax = sp.this_tmax;

    // one sample [t_last, t_next) (grid.cu:219-258)
    auto emit = [&](float t_next) {
        if (EMIT == EMIT_RUNS) {
            const bool cut = !st.continuous || st.run_len == CONE_RUN_CAP;
            if (cut) {
                if (st.n_runs < ro.max_runs)
                    ro.runs[(int64_t)st.n_runs * a.n_rays + tid] =
                        (unsigned long long)f32_bits(st.t_last) |
                        ((unsigned long long)((uint32_t)st.n_samples | (st.continuous ? 0x80000000u : 0u)) << 32);
                st.n_runs++;
            }
            st.run_len = cut ? 1 : st.run_len + 1;
        }
        if (HAS_IV) {
            if (EMIT == EMIT_API) {
                // Both mask bytes of every edge are written (the reference zero-fills the
                // arrays first and only sets the true ones, data_spec.hpp:66-71).
                const int64_t idx = iv_base + st.n_intervals;
                if (!st.continuous) {
                    a.iv_vals[idx] = st.t_last; a.iv_ray_indices[idx] = tid;
                    a.iv_is_left[idx] = 1; a.iv_is_right[idx] = 0;
                    a.iv_vals[idx + 1] = t_next; a.iv_ray_indices[idx + 1] = tid;
                    a.iv_is_left[idx + 1] = 0; a.iv_is_right[idx + 1] = 1;
                } else {
                    a.iv_vals[idx] = t_next; a.iv_ray_indices[idx] = tid;
                    a.iv_is_left[idx - 1] = 1; a.iv_is_left[idx] = 0; a.iv_is_right[idx] = 1;
                }
            }
            st.n_intervals += st.continuous ? 1 : 2;
        }
        if (HAS_SM) {
            const int64_t idx = sm_base + st.n_samples;
            if (EMIT == EMIT_API) {
                a.sm_vals[idx] = (t_next + st.t_last) * 0.5f;
                a.sm_ray_indices[idx] = tid;
                a.sm_is_valid[idx] = 1;
            } else if (EMIT == EMIT_DIRECT) {
                a.sm_t_starts[idx] = st.t_last;
                a.sm_t_ends[idx] = t_next;
                if (a.sm_ray_indices) a.sm_ray_indices[idx] = tid;
            }
        }
        st.n_samples++;
        st.continuous = 1;
        st.t_last = t_next;
    };

    const bool use_bricks = a.bricks != nullptr;
    const float t_traverse = fminf(fminf(tdist[0], fminf(tdist[1], tdist[2])), this_tmax);
    bool occupied;
    if (use_bricks) {
        const uint32_t half_w = (cur[0] & 2) ? st.brick_hi : st.brick_lo;
        occupied = (half_w >> (((cur[0] & 1) << 4) | ((cur[1] & 3) << 2) | (cur[2] & 3))) & 1u;
    } else {
        const int64_t level_base = (int64_t)sp.level * a.res[0] * a.res[1] * a.res[2];
        occupied = a.binaries[level_base + (int64_t)(cur[0] * a.res[1] * a.res[2] + cur[1] * a.res[2] + cur[2])] != 0;
    }
    // single_traversal, utils_grid.cuh:116-142 (branch-free)
    const bool s0 = (tdist[0] < tdist[1]) && (tdist[0] < tdist[2]);
    const bool s1 = !s0 && (tdist[1] < tdist[2]);
    const bool s2 = !s0 && !s1;
    cur[0] += s0 ? sp.step[0] : 0; tdist[0] += s0 ? sp.delta[0] : 0.0f;
    cur[1] += s1 ? sp.step[1] : 0; tdist[1] += s1 ? sp.delta[1] : 0.0f;
    cur[2] += s2 ? sp.step[2] : 0; tdist[2] += s2 ? sp.delta[2] : 0.0f;
    // (the stepped axis reached its overflow index; as one select chain: `||` of `&&`s compiles to nested branches)
    const int32_t off_end = s0 ? (cur[0] ^ sp.overflow[0]) : (s1 ? (cur[1] ^ sp.overflow[1]) : (cur[2] ^ sp.overflow[2]));
    const bool done = off_end == 0;
    unsigned long long w_next = 0ull;
    const int32_t bid_next = use_bricks ? brick_index(a, sp) : st.brick_id;
    const bool fetch = use_bricks && !done && bid_next != st.brick_id;
    if (fetch) w_next = a.bricks[bid_next];

    if (step_size <= 0.0f) {  // one interval per occupied cell (grid.cu:155,198,212)
        if (occupied) emit(t_traverse);
        else { st.t_last = t_traverse; st.continuous = 0; }
    } else if (SPLIT) {
        // The same two marches for walks that spend their time in empty cells (limited walks): the empty cell's is
        // straight-line code -- eight select steps cover a cell of the finest level at the smallest step, the loop behind
        // them runs only for what is left (a step that makes no progress leaves t_last unchanged, the loop then sees it and
        // the jump below applies, as in the merged loop) -- and the sampling loop runs only when some lane has an occupied cell.
        float dt = calc_dt(st.t_last, cone, step_size);
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
        // March to t_traverse.  An empty cell skips with the dt of its first step (grid.cu:193-206), an occupied
        // one emits with dt recomputed per sample (grid.cu:207-262): one loop, so that a wave whose lanes sit in
        // cells of both kinds runs it once.  (Measured: two loops, and empty cells walked ahead with the occupied
        // ones sampled in batches, are both slower -- 14.0 and 22-24 ms against 12.8 ms on cfg 5.)
        float dt = calc_dt(st.t_last, cone, step_size);
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
    st.brick_id = fetch ? bid_next : st.brick_id;
    st.brick_lo = fetch ? (uint32_t)w_next : st.brick_lo;
    st.brick_hi = fetch ? (uint32_t)(w_next >> 32) : st.brick_hi;
    --sp.cells_left;
    return done || sp.cells_left <= 0 || (limit > 0 && st.n_samples >= limit);
}

template <int EMIT, bool HAS_IV, bool HAS_SM>
__device__ __forceinline__ void traverse_span(const nfa_traverse_args &a, int64_t tid, const float o[3],
                                              const float d[3], const float inv[3], int32_t level,
                                              float this_tmin, float this_tmax, int64_t iv_base,
                                              int64_t sm_base, RayState &st, const RunOut &ro)
{
    SpanState sp;
    span_begin(a, o, d, inv, level, this_tmin, this_tmax, st, sp);
    if (a.traverse_steps_limit > 0 && st.n_samples >= a.traverse_steps_limit) return;
    while (!span_cell<EMIT, HAS_IV, HAS_SM, EMIT == EMIT_RUNS && NFA_TRAVERSE_SPLIT != 0>(a, tid, iv_base, sm_base, sp, st, ro)) {}
}

// EMIT_NONE  : count pass (mode 0)
// EMIT_API   : interval edges + masks, sample centres + is_valid (modes 1, 2)
// EMIT_DIRECT: (t_starts, t_ends, ray_indices) per sample (modes 1, 2)
// FUSED      : single grid, intersection computed here (t_sorted/t_indices/hits are NULL)
template <int EMIT, bool HAS_IV, bool HAS_SM, bool FUSED>
#ifdef NFA_TRAVERSE_WAVES
__attribute__((amdgpu_waves_per_eu(NFA_TRAVERSE_WAVES, 8)))
#endif
__global__ __launch_bounds__(256) void traverse_kernel(const nfa_traverse_args a_in, const RunOut ro)
{
    if (a_in.run_if_nonzero && *a_in.run_if_nonzero == 0) return;   // (a fill pass launched without the host knowing whether it is needed)
    nfa_traverse_args a = a_in;
    if (a.steps_limit_dev) {   // the step limit of a loop that does not wait for the host (0: this iteration does nothing)
        a.traverse_steps_limit = *a.steps_limit_dev;
        if (a.traverse_steps_limit <= 0) return;
    }
    const int64_t n_walk = (EMIT == EMIT_RUNS && ro.order) ? ro.n_order : a.n_rays;
    for (int64_t slot_i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot_i < n_walk;
         slot_i += (int64_t)blockDim.x * gridDim.x) {
        // (a wave runs as long as its longest ray: with unrelated rays the lanes of a wave get rays of similar length)
        const int64_t tid = (EMIT == EMIT_RUNS && ro.order) ? (int64_t)ro.order[slot_i] : slot_i;
        const bool overalloc = a.mode == 2;
        if (overalloc && a.rays_mask != nullptr && !a.rays_mask[tid]) {  // grid.cu:100
            // (the reference leaves these entries uninitialised; we define them)
            if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
            if (HAS_IV) a.iv_cnts[tid] = 0;
            if (HAS_SM) a.sm_cnts[tid] = 0;
            if (EMIT == EMIT_RUNS) ro.run_cnts[tid] = 0;
            continue;
        }
        if (a.ray_filter != nullptr && a.ray_filter[tid] <= a.ray_filter_min) continue;
        int64_t iv_base = 0, sm_base = 0;
        if (EMIT != EMIT_NONE && EMIT != EMIT_RUNS) {
            if (a.mode == 1) {  // grid.cu:103-106: nothing to fill for empty rays
                if (HAS_IV && a.iv_cnts[tid] == 0) continue;
                if (HAS_SM && a.sm_cnts[tid] == 0) continue;
            }
            if (HAS_IV) iv_base = a.iv_starts[tid];
            if (HAS_SM) sm_base = a.sm_starts[tid];
        }
        const float near_plane = a.near_planes[tid], far_plane = a.far_planes[tid];
        const float o[3] = {a.rays_o[3 * tid], a.rays_o[3 * tid + 1], a.rays_o[3 * tid + 2]};
        const float d[3] = {a.rays_d[3 * tid], a.rays_d[3 * tid + 1], a.rays_d[3 * tid + 2]};
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};

        RayState st;
        st.t_last = near_plane;
        st.continuous = 0;
        st.n_intervals = 0;
        st.n_samples = 0;
        st.brick_id = -1; st.brick_lo = st.brick_hi = 0u;
        st.n_runs = 0; st.run_len = 0;

        // A ray with a non-finite origin or direction has no geometry: upstream its NaN planes survive fmaxf / fminf as
        // [near, far] and the ray is sampled all the way to the far plane (1e10 by default).  Here it gets no samples.
        const bool ray_ok = isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(d[0]) && isfinite(d[1]) && isfinite(d[2]);
        if (FUSED) {
            // grid.py:158-162 with one grid: events are (t_min: enter 0), (t_max: leave 0).
            float tmin, tmax;
            const bool hit = ray_ok && slab_test(o, inv, a.aabbs, a.aabbs + 3, -INFINITY, INFINITY, tmin, tmax);
            if (hit) {
                const float this_tmin = fmaxf(tmin, near_plane);
                const float this_tmax = fminf(tmax, far_plane);
                if (this_tmin < this_tmax)
                    traverse_span<EMIT, HAS_IV, HAS_SM>(a, tid, o, d, inv, 0, this_tmin, this_tmax, iv_base,
                                                        sm_base, st, ro);
            }
        } else {
            const int32_t G = a.n_grids;
            const uint8_t *hits = a.hits + tid * G;
            const float *ts = a.t_sorted + tid * 2 * G;
            const int64_t *ti = a.t_indices + tid * 2 * G;
            for (int32_t i = 0; i < (ray_ok ? 2 * G - 1 : 0); ++i) {  // grid.cu:125-150
                const int64_t idx = ti[i];
                const bool is_entering = idx < G;
                int32_t level = event_level(idx, G);
                if ((uint32_t)level >= (uint32_t)G || !hits[level]) continue;
                if (!is_entering) {
                    const int64_t nidx = ti[i + 1];
                    if (nidx < G) continue;
                    level = event_level(nidx, G);
                    if ((uint32_t)level >= (uint32_t)G || !hits[level]) continue;
                }
                const float this_tmin = fmaxf(ts[i], near_plane);
                const float this_tmax = fminf(ts[i + 1], far_plane);
                if (this_tmin >= this_tmax) continue;
                traverse_span<EMIT, HAS_IV, HAS_SM>(a, tid, o, d, inv, level, this_tmin, this_tmax, iv_base,
                                                    sm_base, st, ro);
                // The budget is spent: the last thing that happened was a sample (continuous), so the spans still to
                // come would change nothing (grid.cu:151,185: no fast-forward, no cell visited) -- skip their set-up.
                if (a.traverse_steps_limit > 0 && st.n_samples >= a.traverse_steps_limit) break;
            }
        }
        if (a.terminate_planes) a.terminate_planes[tid] = st.t_last;
        if (EMIT == EMIT_NONE || EMIT == EMIT_RUNS || overalloc) {
            if (HAS_IV) a.iv_cnts[tid] = st.n_intervals;
            if (HAS_SM) a.sm_cnts[tid] = st.n_samples;
        }
        if (EMIT == EMIT_RUNS) {
            // rays with > 2^21 samples go to the serial fill (the expansion packs a 27-bit batch offset)
            if (st.n_samples > (1 << 21) && st.n_runs <= ro.max_runs) st.n_runs = ro.max_runs + 1;
            ro.run_cnts[tid] = st.n_runs;
            if (st.n_runs > ro.max_runs) atomicAdd(ro.overflow, 1);
        }
    }
}

template <int EMIT, bool HAS_IV, bool HAS_SM>
static void launch_traverse(const nfa_traverse_args &a, bool fused, hipStream_t s, const RunOut ro = RunOut{nullptr, nullptr, 0, nullptr, nullptr, 0})
{
    const unsigned grid = grid_1d((EMIT == EMIT_RUNS && ro.order) ? ro.n_order : a.n_rays, 256, 1 << 20);
    if (fused) hipLaunchKernelGGL((traverse_kernel<EMIT, HAS_IV, HAS_SM, true>), dim3(grid), dim3(256), 0, s, a, ro);
    else       hipLaunchKernelGGL((traverse_kernel<EMIT, HAS_IV, HAS_SM, false>), dim3(grid), dim3(256), 0, s, a, ro);
}

// ------------------------------------------------------------------------------------------
// Limited walks (traverse_steps_limit > 0: one iteration of the test-mode loop, examples/utils.py:252-425) stop after a
// handful of samples, i.e. after a number of cells that is geometric in the local occupancy.  With one ray per lane from
// start to end a wave lasts as long as its unluckiest ray: on cfg 5 (2 % scattered occupancy) the mean is 50 cells to the
// first sample, the maximum over 64 lanes about 240, and the lanes are busy a fifth of the time (2.5 ms per call for
// 2 M rays where the cells themselves are worth 0.3 ms).  Here a wave owns `chunk` consecutive slots of the ray list and
// a lane that has finished its ray is given the next one: the wave leaves its cell loop when fewer than `min_busy` lanes
// are still walking, sets up new rays (and the next spans of rays that crossed into another level) on the free lanes, and
// re-enters.  Per ray the arithmetic is span_begin / span_cell, the same code as traverse_kernel: results are identical.
template <bool FUSED>
__global__ __launch_bounds__(256) void traverse_refill_kernel(const nfa_traverse_args a, const RunOut ro, const int32_t chunk,
                                                              const int32_t min_busy)
{
    enum { IDLE = 0, SPAN = 1, WALK = 2, FINISH = 3 };
    const int lane = lane_id();
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    const int64_t n_walk = ro.order ? ro.n_order : a.n_rays;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t next = wave * chunk;  // (wave-uniform) first slot not handed out yet
    const int64_t end = next + chunk < n_walk ? next + chunk : n_walk;
    const int32_t limit = a.traverse_steps_limit;
    const int32_t G = a.n_grids;

    int32_t phase = IDLE, ev = 0;
    int64_t tid = 0;
    float near_plane = 0.0f, far_plane = 0.0f;
    float o[3] = {0.0f, 0.0f, 0.0f}, d[3] = {0.0f, 0.0f, 0.0f}, inv[3] = {0.0f, 0.0f, 0.0f};
    RayState st;
    SpanState sp;
    st.t_last = 0.0f; st.continuous = 0; st.n_intervals = 0; st.n_samples = 0;
    st.brick_id = -1; st.brick_lo = st.brick_hi = 0u; st.n_runs = 0; st.run_len = 0;

    for (;;) {
        // Two passes: rays that left a span in the cell loop (their next span, or their end), then the rays handed to
        // the lanes that are free after that.
#pragma nounroll
        for (int pass = 0; pass < 2; ++pass) {
            // ---- lanes between spans: the ray's next span, or its end
            if (phase == SPAN) {
                bool found = false;
                if (FUSED) {
                    if (ev == 0) {
                        float tmin, tmax;
                        if (slab_test(o, inv, a.aabbs, a.aabbs + 3, -INFINITY, INFINITY, tmin, tmax)) {
                            const float this_tmin = fmaxf(tmin, near_plane);
                            const float this_tmax = fminf(tmax, far_plane);
                            if (this_tmin < this_tmax) { span_begin(a, o, d, inv, 0, this_tmin, this_tmax, st, sp); found = true; }
                        }
                    }
                    ev = 1;
                } else {
                    const uint8_t *hits = a.hits + tid * G;
                    const float *ts = a.t_sorted + tid * 2 * G;
                    const int64_t *ti = a.t_indices + tid * 2 * G;
                    while (ev < 2 * G - 1 && !found) {  // grid.cu:125-150
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
                        if (ok && this_tmin < this_tmax) { span_begin(a, o, d, inv, level, this_tmin, this_tmax, st, sp); found = true; }
                    }
                }
                phase = found ? WALK : FINISH;
            }
            if (phase == FINISH) {
                if (a.terminate_planes) a.terminate_planes[tid] = st.t_last;
                a.sm_cnts[tid] = st.n_samples;
                if (st.n_samples > (1 << 21) && st.n_runs <= ro.max_runs) st.n_runs = ro.max_runs + 1;
                ro.run_cnts[tid] = st.n_runs;
                if (st.n_runs > ro.max_runs) atomicAdd(ro.overflow, 1);
                phase = IDLE;
            }
            if (pass == 1) break;
            // ---- free lanes take the next rays of the wave's chunk
            const unsigned long long idle = __ballot(phase == IDLE);
            if (idle != 0ull && next < end) {
                if (phase == IDLE) {
                    const int64_t slot = next + __popcll(idle & lanes_below);
                    if (slot < end) {
                        tid = ro.order ? (int64_t)ro.order[slot] : slot;
                        if (a.mode == 2 && a.rays_mask != nullptr && !a.rays_mask[tid]) {  // grid.cu:100 (outputs defined, as in traverse_kernel)
                            if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
                            a.sm_cnts[tid] = 0;
                            ro.run_cnts[tid] = 0;
                        } else if (!(a.ray_filter != nullptr && a.ray_filter[tid] <= a.ray_filter_min)) {
                            near_plane = a.near_planes[tid]; far_plane = a.far_planes[tid];
    #pragma unroll
                            for (int ax = 0; ax < 3; ++ax) {
                                o[ax] = a.rays_o[3 * tid + ax];
                                d[ax] = a.rays_d[3 * tid + ax];
                                inv[ax] = 1.0f / d[ax];
                            }
                            st.t_last = near_plane; st.continuous = 0; st.n_samples = 0;
                            st.brick_id = -1; st.n_runs = 0; st.run_len = 0;
                            const bool ray_ok = isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(d[0]) && isfinite(d[1]) && isfinite(d[2]);
                            ev = ray_ok ? 0 : 2 * G;  // (no geometry: no spans, see traverse_kernel)
                            phase = SPAN;
                        }
                    }
                }
                next += __popcll(idle);
            }
        }
        const unsigned long long walking = __ballot(phase == WALK);
        if (walking == 0ull) {
            if (next >= end) break;  // (every lane is IDLE here: SPAN and FINISH were resolved above)
            continue;
        }
        // ---- cells, for as long as enough lanes have one to visit
        const int32_t n_walking = __popcll(walking);
        const int32_t need = next < end ? min_busy : (n_walking * 3 >> 2) > 1 ? (n_walking * 3 >> 2) : 1;
        do {
            if (phase == WALK) {
                if (span_cell<EMIT_RUNS, false, true, NFA_REFILL_SPLIT != 0>(a, tid, 0, 0, sp, st, ro))
                    phase = (limit > 0 && st.n_samples >= limit) ? FINISH : SPAN;  // budget spent: nothing after it changes the ray (see traverse_kernel)
            }
        } while (__popcll(__ballot(phase == WALK)) >= need);
    }
}

// ------------------------------------------------------------------------------------------
// int64 exclusive cumsum (replaces torch::cumsum + the .item() of data_spec.hpp:86-96;
// the total stays on the device, the caller decides when to read it).
constexpr int CS_THREADS = 256;
constexpr int CS_ITEMS = 8;
constexpr int CS_BLOCK = CS_THREADS * CS_ITEMS;  // 2048 elements per workgroup

__device__ __forceinline__ int64_t block_excl_scan_i64(int64_t v, int64_t &block_total, int64_t *lds /*[4+1]*/)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t incl = wave_incl_sum_i64(v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int64_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CS_THREADS / 64; ++w) {
        const int64_t x = lds[w];
        if (w < wave) wave_off += x;
        tot += x;
    }
    __syncthreads();
    block_total = tot;
    return wave_off + incl - v;
}

__global__ __launch_bounds__(CS_THREADS) void cumsum_partials_kernel(const int64_t *__restrict__ in, int64_t n,
                                                                      int64_t in_stride, int64_t *__restrict__ partials)
{
    __shared__ int64_t lds[8];
    const int64_t base = (int64_t)blockIdx.x * CS_BLOCK + (int64_t)threadIdx.x * CS_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < CS_ITEMS; ++k)
        if (base + k < n) s += in[(base + k) * in_stride];
    int64_t tot;
    block_excl_scan_i64(s, tot, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the per-block partials in place; total -> *total
__global__ __launch_bounds__(CS_THREADS) void cumsum_spine_kernel(int64_t *partials, int64_t n_blocks, int64_t *total)
{
    __shared__ int64_t lds[8];
    int64_t carry = 0;
    for (int64_t base = 0; base < n_blocks; base += CS_THREADS) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n_blocks ? partials[i] : 0;
        int64_t tot;
        const int64_t ex = block_excl_scan_i64(v, tot, lds);
        if (i < n_blocks) partials[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

// out_starts[i*out_stride] = exclusive prefix; optionally copies the counts next to it
// (pairs != 0 writes packed_info rows {start, count}).
// SPINE: `partials` holds the raw per-block sums and every workgroup adds up the ones before it by itself (a few
// loads per thread from L2) instead of waiting for a one-workgroup spine kernel in between: two launches, not three.
// stats (optional, [2], zeroed by the caller): += sum over groups of 64 consecutive inputs of the group's maximum,
// += sum of all inputs.  64 * stats[0] / stats[1] = how much longer a wave of 64 neighbouring rays runs than its
// average ray: the coherence measure behind OccGridEstimator's automatic ray binning.
template <bool SPINE>
__global__ __launch_bounds__(CS_THREADS) void cumsum_final_kernel(const int64_t *__restrict__ in, int64_t n,
                                                                   int64_t in_stride,
                                                                   const int64_t *__restrict__ partials,
                                                                   int64_t *__restrict__ out, int pairs,
                                                                   int64_t *__restrict__ total,
                                                                   unsigned long long *__restrict__ stats = nullptr)
{
    __shared__ int64_t lds[8];
    int64_t block_off = 0;
    if (SPINE) {
        int64_t before = 0;
        for (int64_t j = threadIdx.x; j < (int64_t)blockIdx.x; j += CS_THREADS) before += partials[j];
        block_excl_scan_i64(before, block_off, lds);  // block_off = sum over the workgroup
    } else {
        block_off = partials[blockIdx.x];
    }
    const int64_t base = (int64_t)blockIdx.x * CS_BLOCK + (int64_t)threadIdx.x * CS_ITEMS;
    int64_t v[CS_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < CS_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[(base + k) * in_stride] : 0;
        s += v[k];
    }
    int64_t tot;
    int64_t run = block_off + block_excl_scan_i64(s, tot, lds);
    if (SPINE && total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = block_off + tot;
    if (SPINE && stats && (blockIdx.x & 7) == 0) {  // a sample (one workgroup in 8) keeps the atomics off the critical path;
                                                  // all lanes are here: the loads above are bounds-checked, not skipped
        int64_t m = 0;
#pragma unroll
        for (int k = 0; k < CS_ITEMS; ++k) m = v[k] > m ? v[k] : m;
        // 8 lanes x 8 items = 64 consecutive inputs
        for (int off = 1; off < 8; off <<= 1) { const int64_t u = __shfl_xor(m, off, 64); m = u > m ? u : m; }
        int64_t gm = (lane_id() & 7) == 0 ? m : 0;
        for (int off = 8; off < 64; off <<= 1) gm += __shfl_xor(gm, off, 64);
        if (lane_id() == 0) atomicAdd(&stats[0], (unsigned long long)gm);
        if (threadIdx.x == 0) atomicAdd(&stats[1], (unsigned long long)tot);
    }
#pragma unroll
    for (int k = 0; k < CS_ITEMS; ++k) {
        if (base + k < n) {
            if (pairs) { out[2 * (base + k)] = run; out[2 * (base + k) + 1] = v[k]; }
            else out[base + k] = run;
        }
        run += v[k];
    }
}

static int run_cumsum(const int64_t *in, int64_t n, int64_t in_stride, int64_t *out, int pairs, int64_t *total,
                      void *scratch, hipStream_t s, unsigned long long *stats = nullptr)
{
    const int64_t n_blocks = ceil_div64(n > 0 ? n : 1, CS_BLOCK);
    int64_t *partials = reinterpret_cast<int64_t *>(scratch);
    hipLaunchKernelGGL(cumsum_partials_kernel, dim3((unsigned)n_blocks), dim3(CS_THREADS), 0, s, in, n, in_stride, partials);
    if (n_blocks <= 2048) {  // up to 4 M elements: 8 partials per thread at most
        hipLaunchKernelGGL(cumsum_final_kernel<true>, dim3((unsigned)n_blocks), dim3(CS_THREADS), 0, s, in, n, in_stride, partials,
                           out, pairs, total, stats);
    } else {
        hipLaunchKernelGGL(cumsum_spine_kernel, dim3(1), dim3(CS_THREADS), 0, s, partials, n_blocks, total);
        hipLaunchKernelGGL(cumsum_final_kernel<false>, dim3((unsigned)n_blocks), dim3(CS_THREADS), 0, s, in, n, in_stride, partials,
                           out, pairs, nullptr, nullptr);
    }
    NFA_CHECK_LAUNCH("exclusive_cumsum_i64");
    return NFA_OK;
}

// ------------------------------------------------------------------------------------------
// pack_info, pack.py:38-46.  Histogram with one atomic per (wave, run of equal indices);
// for a ray-sorted stream that is ~1 atomic per ray instead of one per sample.
// cnts of ray r lives at packed[2r+1].
__global__ __launch_bounds__(256) void pack_hist_kernel(const int64_t *__restrict__ ray_indices, int64_t n,
                                                        int64_t n_rays, unsigned long long *__restrict__ packed,
                                                        int32_t *__restrict__ flags)
{
    const int lane = lane_id();
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n;
         base += (int64_t)blockDim.x * gridDim.x) {
        const int64_t i = base + lane;
        const bool valid = i < n;
        const int64_t r = valid ? ray_indices[i] : -1;
        int64_t prev = __shfl_up(r, 1, NFA_WAVE);
        if (lane == 0) prev = (i > 0 && valid) ? ray_indices[i - 1] : r;
        const bool in_range = valid && r >= 0 && r < n_rays;
        if (valid && (!in_range || r < prev)) atomicOr(flags, 1);
        const bool head = valid && (lane == 0 || r != prev);
        const unsigned long long heads = __ballot(head);
        const unsigned long long live = __ballot(valid);
        if (head && in_range) {
            const unsigned long long later = (lane == 63) ? 0ull : (heads >> (lane + 1));
            const int next = later ? (lane + 1 + __builtin_ctzll(later)) : (int)__builtin_popcountll(live);
            atomicAdd(&packed[2 * r + 1], (unsigned long long)(next - lane));
        }
    }
}

__global__ void zero_i64_kernel(int64_t *p, int64_t n, int32_t *flags)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) p[i] = 0;
    if (flags && blockIdx.x == 0 && threadIdx.x == 0) *flags = 0;
}

__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *__restrict__ b, int64_t n, uint32_t *__restrict__ bits)
{
    const int lane = lane_id();
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n;
         base += (int64_t)blockDim.x * gridDim.x) {
        const int64_t i = base + lane;
        const unsigned long long m = __ballot(i < n && b[i] != 0);
        if (lane == 0) bits[base >> 5] = (uint32_t)m;
        if (lane == 32 && base + 32 < n) bits[(base >> 5) + 1] = (uint32_t)(m >> 32);
    }
}

}  // namespace nfa

using namespace nfa;

extern "C" {

const char *nfa_last_error(void) { return g_err; }
int nfa_version(void) { return NFA_VERSION; }
int nfa_set_tuning(const char *name, const char *value)
{
    NFA_REQUIRE(name && strlen(name) > 0 && strlen(name) < sizeof(g_knobs[0].name), "set_tuning: bad name");
    NFA_REQUIRE(!value || strlen(value) < sizeof(g_knobs[0].value), "set_tuning: value too long");
    int i = 0;
    while (i < g_n_knobs && strcmp(g_knobs[i].name, name) != 0) ++i;
    if (i == g_n_knobs) {
        NFA_REQUIRE(g_n_knobs < (int)(sizeof(g_knobs) / sizeof(g_knobs[0])), "set_tuning: too many knobs");
        strcpy(g_knobs[g_n_knobs++].name, name);
    }
    strcpy(g_knobs[i].value, value ? value : "");
    return NFA_OK;
}

int nfa_device_arch(char *buf, int buflen)
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_error("no HIP device");
        return NFA_EHIP;
    }
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return NFA_OK;
}

int64_t nfa_cumsum_scratch_bytes(int64_t n) { return (ceil_div64(n > 0 ? n : 1, CS_BLOCK) + 1) * 8; }

int nfa_exclusive_cumsum_i64(const int64_t *cnts, int64_t n, int64_t *starts, int64_t *total, void *scratch,
                             nfa_stream_t stream)
{
    NFA_REQUIRE(n >= 0 && scratch && (n == 0 || (cnts && starts)), "exclusive_cumsum_i64: bad arguments");
    return run_cumsum(cnts, n, 1, starts, 0, total, scratch, as_stream(stream));
}

int nfa_exclusive_cumsum_pairs_i64(const int64_t *cnts, int64_t n, int64_t *packed_info, int64_t *total, void *scratch,
                                   nfa_stream_t stream)
{
    NFA_REQUIRE(n >= 0 && scratch && (n == 0 || (cnts && packed_info)), "exclusive_cumsum_pairs_i64: bad arguments");
    return run_cumsum(cnts, n, 1, packed_info, 1, total, scratch, as_stream(stream));
}

int nfa_exclusive_cumsum_pairs_stats_i64(const int64_t *cnts, int64_t n, int64_t *packed_info, int64_t *total_and_stats,
                                         void *scratch, nfa_stream_t stream)
{
    NFA_REQUIRE(n >= 0 && scratch && total_and_stats && (n == 0 || (cnts && packed_info)),
                "exclusive_cumsum_pairs_stats_i64: bad arguments");
    return run_cumsum(cnts, n, 1, packed_info, 1, total_and_stats, scratch, as_stream(stream),
                      reinterpret_cast<unsigned long long *>(total_and_stats + 1));
}

int nfa_pack_info(const int64_t *ray_indices, int64_t n, int64_t n_rays, int64_t *packed_info, int32_t *flags,
                  void *scratch, nfa_stream_t stream)
{
    NFA_REQUIRE(n >= 0 && n_rays >= 0 && flags && scratch, "pack_info: bad arguments");
    NFA_REQUIRE(n_rays == 0 || packed_info, "pack_info: packed_info is null");
    NFA_REQUIRE(n == 0 || ray_indices, "pack_info: ray_indices is null");
    hipStream_t s = as_stream(stream);
    // Counts are accumulated in the count slots packed_info[2r+1], then scanned in place.
    hipLaunchKernelGGL(zero_i64_kernel, dim3(grid_1d(2 * n_rays + 1, 256)), dim3(256), 0, s, packed_info, 2 * n_rays, flags);
    if (n > 0 && n_rays > 0)
        hipLaunchKernelGGL(pack_hist_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, s, ray_indices, n, n_rays,
                           reinterpret_cast<unsigned long long *>(packed_info), flags);
    NFA_CHECK_LAUNCH("pack_info");
    if (n_rays == 0) return NFA_OK;
    return run_cumsum(packed_info + 1, n_rays, 2, packed_info, 1, nullptr, scratch, s);
}

int nfa_pack_bits(const uint8_t *binaries, int64_t n_cells, uint32_t *bits, nfa_stream_t stream)
{
    NFA_REQUIRE(n_cells >= 0 && (n_cells == 0 || (binaries && bits)), "pack_bits: bad arguments");
    if (n_cells == 0) return NFA_OK;
    hipLaunchKernelGGL(pack_bits_kernel, dim3(grid_1d(n_cells, 256)), dim3(256), 0, as_stream(stream), binaries, n_cells, bits);
    NFA_CHECK_LAUNCH("pack_bits");
    return NFA_OK;
}

int nfa_ray_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays, const float *aabbs,
                           int32_t n_aabbs, float near_plane, float far_plane, float miss_value, float *t_mins,
                           float *t_maxs, uint8_t *hits, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_aabbs >= 0, "ray_aabb_intersect: negative size");
    if (n_rays * (int64_t)n_aabbs == 0) return NFA_OK;
    NFA_REQUIRE(rays_o && rays_d && aabbs && t_mins && t_maxs && hits, "ray_aabb_intersect: null pointer");
    hipLaunchKernelGGL(ray_aabb_kernel, dim3(grid_1d(n_rays * n_aabbs, 256)), dim3(256), 0, as_stream(stream), rays_o,
                       rays_d, n_rays, aabbs, n_aabbs, near_plane, far_plane, miss_value, t_mins, t_maxs, hits);
    NFA_CHECK_LAUNCH("ray_aabb_intersect");
    return NFA_OK;
}

int nfa_ray_events(const float *rays_o, const float *rays_d, int64_t n_rays, const float *aabbs, int32_t n_aabbs,
                   float *t_sorted, int64_t *t_indices, uint8_t *hits, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_aabbs >= 1 && n_aabbs <= NFA_MAX_EVENT_LEVELS, "ray_events: 1..NFA_MAX_EVENT_LEVELS boxes");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(rays_o && rays_d && aabbs && t_sorted && t_indices && hits, "ray_events: null pointer");
    const dim3 grid(grid_1d(n_rays, 256)), block(256);
    hipStream_t s = as_stream(stream);
#define NFA_EV(G) case G: hipLaunchKernelGGL(ray_events_kernel<G>, grid, block, 0, s, rays_o, rays_d, n_rays, aabbs, t_sorted, t_indices, hits); break
    switch (n_aabbs) { NFA_EV(1); NFA_EV(2); NFA_EV(3); NFA_EV(4); NFA_EV(5); NFA_EV(6); NFA_EV(7); NFA_EV(8); }
#undef NFA_EV
    NFA_CHECK_LAUNCH("ray_events");
    return NFA_OK;
}

int nfa_traverse_grids(const nfa_traverse_args *pa, nfa_stream_t stream)
{
    NFA_REQUIRE(pa != nullptr, "traverse_grids: null args");
    const nfa_traverse_args &a = *pa;
    NFA_REQUIRE(a.n_rays >= 0 && a.n_rays < (int64_t)1 << 31, "traverse_grids: n_rays out of range");
    if (a.n_rays == 0) return NFA_OK;
    NFA_REQUIRE(a.mode >= 0 && a.mode <= 2, "traverse_grids: mode must be 0, 1 or 2");
    NFA_REQUIRE(a.rays_o && a.rays_d && a.binaries && a.aabbs && a.near_planes && a.far_planes,
                "traverse_grids: null input pointer");
    NFA_REQUIRE(a.n_grids >= 1 && a.res[0] > 0 && a.res[1] > 0 && a.res[2] > 0, "traverse_grids: bad grid shape");
    NFA_REQUIRE((int64_t)a.res[0] * a.res[1] * a.res[2] < (int64_t)1 << 31, "traverse_grids: grid level too large");
    const bool fused = !a.t_sorted && !a.t_indices && !a.hits;
    NFA_REQUIRE(fused || (a.t_sorted && a.t_indices && a.hits),
                "traverse_grids: t_sorted, t_indices and hits must be given together");
    NFA_REQUIRE(!fused || a.n_grids == 1, "traverse_grids: in-kernel intersection supports one grid; pass t_sorted/t_indices/hits");
    NFA_REQUIRE(a.mode != 2 || a.traverse_steps_limit > 0,
                "traverse_steps_limit must be > 0 when over_allocate is true");  // grid.cu:345
    const bool has_iv = a.iv_cnts != nullptr, has_sm = a.sm_cnts != nullptr;
    NFA_REQUIRE(has_iv || has_sm, "traverse_grids: nothing to compute");
    const bool direct = a.sm_t_starts != nullptr;
    hipStream_t s = as_stream(stream);
    if (a.mode == 0) {
        if (has_iv && has_sm) launch_traverse<EMIT_NONE, true, true>(a, fused, s);
        else if (has_sm) launch_traverse<EMIT_NONE, false, true>(a, fused, s);
        else launch_traverse<EMIT_NONE, true, false>(a, fused, s);
    } else {
        NFA_REQUIRE(!has_iv || (a.iv_vals && a.iv_ray_indices && a.iv_is_left && a.iv_is_right && a.iv_starts),
                    "traverse_grids: interval outputs missing");
        NFA_REQUIRE(!has_sm || a.sm_starts, "traverse_grids: sample starts missing");
        if (direct) {
            NFA_REQUIRE(!has_iv && has_sm && a.sm_t_ends,
                        "traverse_grids: direct emission needs sm_t_starts, sm_t_ends and no intervals");
            launch_traverse<EMIT_DIRECT, false, true>(a, fused, s);
        } else {
            NFA_REQUIRE(!has_sm || (a.sm_vals && a.sm_ray_indices && a.sm_is_valid), "traverse_grids: sample outputs missing");
            if (has_iv && has_sm) launch_traverse<EMIT_API, true, true>(a, fused, s);
            else if (has_sm) launch_traverse<EMIT_API, false, true>(a, fused, s);
            else launch_traverse<EMIT_API, true, false>(a, fused, s);
        }
    }
    NFA_CHECK_LAUNCH("traverse_grids");
    return NFA_OK;
}

int nfa_traverse_cone_runs(const nfa_traverse_args *pa, int32_t *run_cnts, uint64_t *runs, int32_t max_runs,
                           int32_t *overflow_count, const int32_t *ray_order, int64_t n_order, nfa_stream_t stream)
{
    NFA_REQUIRE(pa != nullptr, "traverse_cone_runs: null args");
    const nfa_traverse_args &a = *pa;
    NFA_REQUIRE(a.n_rays >= 0 && a.n_rays < (int64_t)1 << 31, "traverse_cone_runs: n_rays out of range");
    NFA_REQUIRE(overflow_count, "traverse_cone_runs: overflow_count is null");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(overflow_count, 0, sizeof(int32_t), s) != hipSuccess) { set_error("traverse_cone_runs: memset failed"); return NFA_EHIP; }
    if (a.n_rays == 0) return NFA_OK;
    NFA_REQUIRE(a.step_size > 0.0f && a.cone_angle > 0.0f, "traverse_cone_runs: needs step_size > 0 and cone_angle > 0");
    NFA_REQUIRE(a.mode == 0 || a.mode == 2, "traverse_cone_runs: mode must be 0 (all rays) or 2 (rays_mask + traverse_steps_limit)");
    NFA_REQUIRE(a.mode != 2 || a.traverse_steps_limit > 0, "traverse_steps_limit must be > 0 when over_allocate is true");
    NFA_REQUIRE(a.rays_o && a.rays_d && a.binaries && a.aabbs && a.near_planes && a.far_planes && a.sm_cnts && !a.iv_cnts &&
                    run_cnts && runs, "traverse_cone_runs: null pointer (or interval outputs requested)");
    NFA_REQUIRE(max_runs >= 1 && max_runs <= 32, "traverse_cone_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(a.n_grids >= 1 && a.res[0] > 0 && a.res[1] > 0 && a.res[2] > 0, "traverse_cone_runs: bad grid shape");
    NFA_REQUIRE((int64_t)a.res[0] * a.res[1] * a.res[2] < (int64_t)1 << 31, "traverse_cone_runs: grid level too large");
    const bool fused = !a.t_sorted && !a.t_indices && !a.hits;
    NFA_REQUIRE(fused || (a.t_sorted && a.t_indices && a.hits), "traverse_cone_runs: t_sorted, t_indices and hits must be given together");
    NFA_REQUIRE(!fused || a.n_grids == 1, "traverse_cone_runs: in-kernel intersection supports one grid");
    RunOut ro;
    ro.run_cnts = run_cnts;
    ro.runs = reinterpret_cast<unsigned long long *>(runs);
    ro.max_runs = max_runs;
    ro.overflow = overflow_count;
    ro.order = ray_order;
    ro.n_order = ray_order ? n_order : a.n_rays;
    NFA_REQUIRE(!ray_order || (n_order >= 0 && n_order <= a.n_rays), "traverse_cone_runs: n_order out of range");
    if (ray_order && n_order == 0) return NFA_OK;
    const char *refill_env = tuning_env("NFA_REFILL");  // "0": one ray per lane; "chunk,min_busy": tuning
    const char *refill_all = tuning_env("NFA_REFILL_ALL");
    if ((a.traverse_steps_limit > 0 || (refill_all && refill_all[0] == '1')) && !(refill_env && refill_env[0] == '0')) {
        // slots per wave: enough of them that a lane is refilled several times, as long as the launch still fills the chip
        const int64_t n_walk = ro.n_order;
        int64_t chunk = ((n_walk + 4095) / 4096 + 63) / 64 * 64;
        chunk = chunk < 64 ? 64 : (chunk > 1024 ? 1024 : chunk);
        int min_busy = 48;
        if (refill_env) { long c = 0; int m = 0; if (sscanf(refill_env, "%ld,%d", &c, &m) == 2 && c >= 64 && m >= 1 && m <= 64) { chunk = c / 64 * 64; min_busy = m; } }
        const int64_t n_waves = (n_walk + chunk - 1) / chunk;
        const unsigned grid = (unsigned)((n_waves + 3) / 4);
        if (fused) hipLaunchKernelGGL((traverse_refill_kernel<true>), dim3(grid), dim3(256), 0, s, a, ro, (int32_t)chunk, (int32_t)min_busy);
        else       hipLaunchKernelGGL((traverse_refill_kernel<false>), dim3(grid), dim3(256), 0, s, a, ro, (int32_t)chunk, (int32_t)min_busy);
    } else {
        launch_traverse<EMIT_RUNS, false, true>(a, fused, s, ro);
    }
    NFA_CHECK_LAUNCH("traverse_cone_runs");
    return NFA_OK;
}

}  // extern "C"
