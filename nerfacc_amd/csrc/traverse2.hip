// traverse2.hip -- run-length traversal (constant step, cone_angle == 0): the sampler's and the API's fast path.
//
// The reference walks every ray twice with the same divergent DDA kernel (count pass, fill pass;
// cuda/csrc/grid.cu:405-471), one 1-byte scattered grid load per cell and per-thread strided
// output writes.  Restructured for MI355X:
//
//   pack_bricks   binaries (torch.bool, 1 B/cell)  ->  4x4x4-cell bricks, one 64-bit word each,
//                 plus a 1-bit-per-brick "any occupied" mask (128^3: 256 KiB + 4 KiB).
//   runs pass     ONE DDA walk per ray (thread per ray).  The brick mask lives in LDS, a brick
//                 word is fetched (8 B, L2) only when the ray enters a non-empty brick, and the
//                 current word is cached in registers, so most cells cost no memory access.
//                 The walk does no marching: it records one typed threshold per run of cells of
//                 one kind (a handful per ray, in LDS).  A second, lock-step phase marches through
//                 those thresholds with march.h's Stepper (remembered stable increment per binade,
//                 verified jump counts, tabulated approach from a common near plane) -- skipping and
//                 sample emission are the same code path, emission being "count the steps" -- so
//                 there is no per-sample loop and no per-cell divergence.  Instead of samples the
//                 pass emits RUN RECORDS: n consecutive samples with one exact fp32 increment,
//                 typically 2-10 per ray (8 B each, slot-major).
//   expand passes after the device-side cumsum of the counts, every output element is computed
//                 independently: t = fma(k, inc, t_first) (exact: every sample of a run is t_first + k*inc).
//                 A wave stages the records of 32 rays in LDS; per 256-output chunk the run starts are
//                 scattered into an LDS line and a "most recent entry" DPP scan tells every output its
//                 run; 16 B-per-lane stores.  expand_runs: (t_starts, t_ends, ray_indices) or the API's
//                 sample centres; expand_intervals: the API's edge stream with is_left / is_right.
// Results are bit-identical to the reference's serial accumulation (oracle/nerfacc_oracle.c).
#include "common.hip.h"
#include "march.h"

namespace nfa {

constexpr int EXP_RPW = 32;        // rays per wave batch in the expansion
constexpr int EXP_QMAX = 1024;     // runs staged per batch (EXP_RPW * max_runs)
#ifndef NFA_EXP_WPB
#define NFA_EXP_WPB 2  /* measured on cfg 2: 1 wave 176 us, 2 waves 160 us, 4 waves 175 us */
#endif
constexpr int EXP_WPB = NFA_EXP_WPB;  // waves per workgroup of the expansion kernels (they never cooperate)
constexpr int COARSE_LDS_WORDS = 2048;  // 8 KiB: up to 40^3 bricks (160^3 cells).  A larger mask in LDS costs occupancy: 256^3 (32 KiB) 1103 us in LDS, 634 us from L2

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_bricks_kernel(const uint8_t *__restrict__ binaries, int32_t n_grids,
                                                          int32_t rx, int32_t ry, int32_t rz, int32_t bx, int32_t by,
                                                          int32_t bz, unsigned long long *__restrict__ bricks,
                                                          uint32_t *__restrict__ coarse)
{
    const int64_t n_bricks = (int64_t)n_grids * bx * by * bz;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_bricks;
         b += (int64_t)blockDim.x * gridDim.x) {
        int64_t r = b;
        const int32_t kz = (int32_t)(r % bz); r /= bz;
        const int32_t ky = (int32_t)(r % by); r /= by;
        const int32_t kx = (int32_t)(r % bx); r /= bx;
        const int32_t lvl = (int32_t)r;
        unsigned long long w = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const int32_t x = 4 * kx + i, y = 4 * ky + j;
                if (x >= rx || y >= ry) continue;
                const uint8_t *row = binaries + (((int64_t)lvl * rx + x) * ry + y) * rz + 4 * kz;
                for (int k = 0; k < 4; ++k)
                    if (4 * kz + k < rz && row[k]) w |= 1ull << ((i << 4) | (j << 2) | k);
            }
        bricks[b] = w;
        if (w) atomicOr(&coarse[b >> 5], 1u << (b & 31));
    }
}

// ------------------------------------------------------------------------------------------
struct RunsParams {
    const unsigned long long *bricks;
    const uint32_t *coarse;      // global copy of the brick mask
    int32_t n_coarse_words;
    int32_t bx, by, bz;
    int32_t *run_cnts;           // [n_rays]
    unsigned long long *runs;    // [max_runs, n_rays] (slot-major: a wave reads one slot of 32 rays as one 256 B line)
    int64_t n_rays;
    int32_t max_runs;
    int32_t *overflow;           // [1] number of rays with more runs than max_runs
    const int32_t *order;        // [n_rays] lane -> ray assignment (a permutation) or NULL
    ApproachTable approach;      // shared start of every ray's march (march.h); n == 0: none
};

// Events recorded by the cell walk (phase 1) and consumed by the marcher (phase 2).  Along a ray
// the thresholds are non-decreasing and "advance while the step's mid-point is before the
// threshold" is monotone in the threshold, so consecutive cells of one kind collapse into one
// event carrying the last cell's exit distance:
//   EV_EMPTY(thr)  skip steps while mid < thr              (grid.cu:193-206, continuous = false)
//   EV_OCC(thr)    emit steps while mid < thr              (grid.cu:207-262)
//   EV_SPAN(thr)   start of a grid span: skip to thr only if the previous step was not emitted
//                  (grid.cu:153-163, `if (!continuous)`)
#ifndef NFA_EV_MAX
#define NFA_EV_MAX 24
#endif
constexpr int EV_MAX = NFA_EV_MAX;
// Occupancy of the walk: 24 list entries (24 KiB per 256 rays) + the brick mask leave room for 5 workgroups per CU, and
// the register allocator is asked for 5 waves per SIMD (96 VGPRs; 20 bytes of cold state spill).  Measured on cfg 2
// (1 M rays = 16 waves per SIMD): 32 entries / 4 waves 330 us, 24 / 5: 307 us, 20 / 6: 311 us, 16 / 8: 348 us.
#ifndef NFA_RUNS_WAVES
#define NFA_RUNS_WAVES 5
#endif
enum { EV_EMPTY = 0, EV_OCC = 1, EV_SPAN = 2, EV_NONE = 3 };

struct RunState {
    float t_last;
    bool continuous;
    int32_t n_samples, n_runs, n_chains;  // n_chains: samples emitted while not continuous (each adds one edge)
    Stepper stp;  // remembered stable increment of the current binade (march.h)
    bool at_near; // t_last is still the near plane
    // open run
    bool open, run_cont;
    float run_t0, run_inc;
    int32_t run_n;
    // brick cache
    int32_t brick_id;
    uint32_t brick_lo, brick_hi;
    // event list: open (unmerged) entry in registers, closed entries in LDS
    int32_t ev_cnt;
    uint32_t ev_occ, ev_span;  // kind of closed entry k: bit k of ev_span -> EV_SPAN, else bit k of ev_occ -> EV_OCC / EV_EMPTY
    int32_t open_type;
    float open_thr;
};

// Run record: {t_first : f32 | k_start : 31, continues_previous : 1}, k_start = number of samples of
// this ray before the run (the run's length is the next record's k_start, or the ray's count).
__device__ __forceinline__ void close_run(RunState &st, const RunsParams &p, int64_t tid)
{
    if (!st.open) return;
    if (st.n_runs < p.max_runs) {
        const uint32_t k_start = (uint32_t)(st.n_samples - st.run_n);
        p.runs[(int64_t)st.n_runs * p.n_rays + tid] =
            (unsigned long long)f32_bits(st.run_t0) | ((unsigned long long)(k_start | (st.run_cont ? 0x80000000u : 0u)) << 32);
    }
    st.n_runs++;
    st.open = false;
}

// n samples t, t + inc, ... (exact sums) join the ray's run list
__device__ __forceinline__ void emit_steps(RunState &st, float t, float inc, uint32_t n, const RunsParams &p, int64_t tid)
{
    if (st.open && st.continuous && inc == st.run_inc) {
        st.run_n += (int32_t)n;
    } else {
        close_run(st, p, tid);
        st.open = true; st.run_t0 = t; st.run_inc = inc; st.run_n = (int32_t)n; st.run_cont = st.continuous;
        st.n_chains += st.continuous ? 0 : 1;
    }
    st.n_samples += (int32_t)n;
    st.continuous = true;
}

// Advance t_last while the step's mid-point is before `thr`; with `emit` every step is a sample
// and is appended to the ray's run list.  march.h's Stepper does the arithmetic: one exact jump per
// binade (the stable increment is remembered across the marches of a ray), plus the sample budget of
// traverse_steps_limit.
__device__ __forceinline__ void march(RunState &st, float thr, float dt, float half, bool emit, int32_t limit,
                                      const RunsParams &p, int64_t tid)
{
    if (st.at_near) {  // first march of the ray: the way from the near plane is the same for all rays
        st.at_near = false;
        if (!emit) approach_table_apply(p.approach, st.stp, st.t_last, half, thr);
    }
    for (;;) {
        if (!(st.t_last + half < thr)) return;
        uint32_t budget = 0xFFFFFFFFu;
        if (emit && limit > 0) {
            if (st.n_samples >= limit) return;
            budget = (uint32_t)(limit - st.n_samples);
        }
        const float t = st.t_last;
        float tn = t, inc;
        const uint32_t n = stepper_advance(st.stp, tn, dt, half, thr, budget, &inc);
        if (n == 0u) {  // no progress (see oracle): skipping jumps to the target, emission stops
            if (!emit) { st.t_last = thr; stepper_reset(st.stp); }
            return;
        }
        if (emit) emit_steps(st, t, inc, n, p, tid);
        st.t_last = tn;
    }
}

// Phase 2: consume the closed entries of this lane's list (all lanes loop over the entry index in
// lock-step; one code path for all three kinds).
__device__ __forceinline__ void process_events(RunState &st, const float *ev_thr /*LDS, [EV_MAX][256]*/, float dt,
                                               int32_t limit, const RunsParams &p, int64_t tid)
{
    const float half = dt * 0.5f;
    for (int k = 0; k < st.ev_cnt; ++k) {
        const int type = ((st.ev_span >> k) & 1u) ? EV_SPAN : (int)((st.ev_occ >> k) & 1u);
        const float thr = ev_thr[k * 256 + threadIdx.x];
        if (limit > 0 && st.n_samples >= limit) break;  // grid.cu:184: nothing moves once the limit is hit
        if (type == EV_SPAN && st.continuous) continue;
        // common case in one shot (aligned inside the binade, threshold well before its end, no sample budget);
        // everything else -- and the rare under-estimate -- goes through the general loop
        bool handled = false;
        if (limit <= 0 && !st.at_near) {
            float t = st.t_last;
            StepSeg g0, g1, g2;
            if (stepper_run_event(st.stp, t, dt, half, thr, g0, g1, g2)) {
                if (type == EV_OCC) {
                    if (g0.n > 0u) emit_steps(st, g0.t0, g0.inc, g0.n, p, tid);
                    if (g1.n > 0u) emit_steps(st, g1.t0, g1.inc, g1.n, p, tid);
                    if (g2.n > 0u) emit_steps(st, g2.t0, g2.inc, g2.n, p, tid);
                }
                st.t_last = t;
                handled = true;
            }
        }
        if (!handled) march(st, thr, dt, half, type == EV_OCC, limit, p, tid);
        if (type == EV_EMPTY) st.continuous = false;
    }
    st.ev_cnt = 0;
    st.ev_occ = 0u; st.ev_span = 0u;
}

__device__ __forceinline__ void push_closed(RunState &st, float *ev_thr, int type, float thr)
{
    ev_thr[st.ev_cnt * 256 + threadIdx.x] = thr;
    st.ev_occ |= (uint32_t)(type & 1) << st.ev_cnt;
    st.ev_span |= (uint32_t)(type >> 1) << st.ev_cnt;
    st.ev_cnt++;
}

// DDA state of the span being walked (setup_traversal, include/utils_grid.cuh:58-114)
struct SpanDDA {
    float tdist[3], delta[3], this_tmax;
    int32_t step[3], cur[3], overflow[3];
    int32_t lvl_brick_base, cells_left;
};

__device__ __forceinline__ void span_setup(const nfa_traverse_args &a, const RunsParams &p, const float o[3], const float d[3],
                                           const float inv[3], int32_t level, float this_tmin, float this_tmax, SpanDDA &sp)
{
    const float eps = 1e-6f;
    const float *bmin = a.aabbs + 6 * level, *bmax = bmin + 3;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
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
    sp.lvl_brick_base = level * p.bx * p.by * p.bz;
    sp.cells_left = a.res[0] + a.res[1] + a.res[2] + 3;
}

// min of four finite-or-inf floats as two instructions (fminf's IEEE canonicalisation of signalling NaNs costs
// three more per cell; the DDA's distances are never NaN: d == 0 axes carry this_tmax)
__device__ __forceinline__ float min4_f32(float a, float b, float c, float d)
{
    float m;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(d));
    return m;
}

// One ray = a loop of (phase 1: walk cells, opening the next span when one ends, until the event list is full or
// the ray has no span left) + (phase 2: consume the list).  There is exactly ONE copy of each phase in the code.
template <bool FUSED, bool COARSE_LDS>
__attribute__((amdgpu_waves_per_eu(NFA_RUNS_WAVES, NFA_RUNS_WAVES)))
__global__ __launch_bounds__(256) void runs_kernel(const nfa_traverse_args a, const RunsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_raw[];
    float *ev_thr = reinterpret_cast<float *>(lds_raw);          // [EV_MAX][256]
    uint32_t *coarse_lds = lds_raw + EV_MAX * 256;          // [n_coarse_words] (COARSE_LDS only)
    if (COARSE_LDS) {
        for (int i = threadIdx.x; i < p.n_coarse_words; i += blockDim.x) coarse_lds[i] = p.coarse[i];
        __syncthreads();
    }
    float *ev_col = ev_thr + threadIdx.x;
    const float dt = a.step_size;
    const int32_t limit = a.traverse_steps_limit;
    for (int64_t slot_i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot_i < a.n_rays;
         slot_i += (int64_t)blockDim.x * gridDim.x) {
        // which ray this lane walks: the rays of a training batch are unrelated and a wave runs as long as its longest
        // ray, so the caller may pass an assignment that puts rays of similar length side by side (nfa_bin_rays);
        // everything the walk writes stays indexed by the ray itself
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
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        RunState st;
        st.t_last = near_plane; st.continuous = false; stepper_init(st.stp); st.at_near = true;
        st.n_samples = 0; st.n_runs = 0; st.n_chains = 0; st.open = false; st.run_cont = false; st.run_t0 = 0.f; st.run_inc = 0.f;
        st.run_n = 0; st.brick_id = -1; st.brick_lo = st.brick_hi = 0u;
        st.ev_cnt = 0; st.ev_occ = 0u; st.ev_span = 0u; st.open_type = EV_NONE; st.open_thr = 0.f;

        // the ray's spans: one (slab test here) or the event walk over the sorted intersections (grid.cu:125-150)
        float f_tmin = 0.f, f_tmax = 0.f;
        bool f_pending = false;
        if (FUSED) {
            float tmin, tmax, lo, hi;
            bool hit = true;
            {   // slab test (include/utils_grid.cuh:10-55) with near = -inf, far = +inf (grid.py:158)
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
            }
            f_tmin = fmaxf(tmin, near_plane); f_tmax = fminf(tmax, far_plane);
            f_pending = hit && f_tmin < f_tmax;
        }
        const int32_t G = a.n_grids;
        int32_t next_i = 0;  // next entry of the event walk (non-fused)

        SpanDDA sp;
        sp.this_tmax = 0.f; sp.lvl_brick_base = 0; sp.cells_left = 0;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) { sp.tdist[ax] = sp.delta[ax] = 0.f; sp.step[ax] = sp.cur[ax] = sp.overflow[ax] = 0; }
        bool in_span = false;

        for (;;) {
            // ---------------- phase 1: the reference's cell loop (grid.cu:184-272) reduced to the DDA and one event
            // per run of cells of one kind, straight-line predicated code, 32-bit integer ops only.  The open entry's
            // threshold is written to slot ev_cnt on EVERY cell; the slot becomes a closed entry when the kind changes.
            bool finished = false;
            for (;;) {
                if (!in_span) {
                    if (st.ev_cnt >= EV_MAX - 1) break;  // a span start needs two entries: flush first
                    float this_tmin = 0.f, this_tmax = 0.f;
                    int32_t level = 0;
                    bool found = false;
                    if (FUSED) {
                        found = f_pending; f_pending = false;
                        this_tmin = f_tmin; this_tmax = f_tmax;
                    } else {
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
                    if (!found) { finished = true; break; }
                    // span start: the (conditional) skip to this_tmin becomes the open entry; the first cell closes it
                    if (st.open_type != EV_NONE) push_closed(st, ev_thr, st.open_type, st.open_thr);
                    st.open_type = EV_SPAN;
                    st.open_thr = this_tmin;
                    span_setup(a, p, o, d, inv, level, this_tmin, this_tmax, sp);
                    st.ev_span |= 1u << st.ev_cnt;  // the first cell always closes the EV_SPAN entry: its kind bit
                    in_span = true;
                }
                for (;;) {
                    const int32_t bid = sp.lvl_brick_base +
                                        (int32_t)mad_u24(mad_u24((uint32_t)sp.cur[0] >> 2, (uint32_t)p.by, (uint32_t)sp.cur[1] >> 2), (uint32_t)p.bz, (uint32_t)sp.cur[2] >> 2);
                    if (bid != st.brick_id) {
                        st.brick_id = bid;
                        // the 1-bit mask saves the 8-byte load for empty bricks when it sits in LDS; read from
                        // global memory it would be a second, dependent access in front of the brick
                        unsigned long long w;
                        if (COARSE_LDS) w = ((coarse_lds[bid >> 5] >> (bid & 31)) & 1u) ? p.bricks[bid] : 0ull;
                        else w = p.bricks[bid];
                        st.brick_lo = (uint32_t)w; st.brick_hi = (uint32_t)(w >> 32);
                    }
                    // bit ((x&3)<<4 | (y&3)<<2 | (z&3)) of the 64-bit brick word, on 32-bit halves
                    const uint32_t half_w = (sp.cur[0] & 2) ? st.brick_hi : st.brick_lo;
                    const int sh = ((sp.cur[0] & 1) << 4) | ((sp.cur[1] & 3) << 2) | (sp.cur[2] & 3);
                    const int type = (int)((half_w >> sh) & 1u);  // EV_EMPTY / EV_OCC
                    const bool changed = type != st.open_type;
                    ev_col[st.ev_cnt * 256] = st.open_thr;
                    st.ev_occ |= (changed ? (uint32_t)(st.open_type & 1) : 0u) << st.ev_cnt;
                    st.ev_cnt += changed ? 1 : 0;
                    st.open_type = type;
                    st.open_thr = min4_f32(sp.tdist[0], sp.tdist[1], sp.tdist[2], sp.this_tmax);  // t_traverse, non-decreasing
                    // single_traversal (include/utils_grid.cuh:116-142), branch-free
                    const bool s0 = (sp.tdist[0] < sp.tdist[1]) && (sp.tdist[0] < sp.tdist[2]);
                    const bool s1 = !s0 && (sp.tdist[1] < sp.tdist[2]);
                    const bool s2 = !s0 && !s1;
                    sp.cur[0] += s0 ? sp.step[0] : 0; sp.tdist[0] += s0 ? sp.delta[0] : 0.0f;
                    sp.cur[1] += s1 ? sp.step[1] : 0; sp.tdist[1] += s1 ? sp.delta[1] : 0.0f;
                    sp.cur[2] += s2 ? sp.step[2] : 0; sp.tdist[2] += s2 ? sp.delta[2] : 0.0f;
                    const bool hit_end = (s0 && sp.cur[0] == sp.overflow[0]) || (s1 && sp.cur[1] == sp.overflow[1]) ||
                                         (s2 && sp.cur[2] == sp.overflow[2]);
                    // the span ends when cells_left reaches 0 (one integer carries the exit reason out of the loop:
                    // a bool would live in scalar lane masks and cost mask algebra on every cell)
                    sp.cells_left = hit_end ? 0 : sp.cells_left - 1;
                    if (sp.cells_left <= 0 || st.ev_cnt == EV_MAX) break;
                }
                if (sp.cells_left <= 0) in_span = false;
                if (st.ev_cnt == EV_MAX) break;
            }
            if (finished && st.open_type != EV_NONE) {
                if (st.ev_cnt == EV_MAX) finished = false;  // no slot for the last open entry: flush and come back
                else { push_closed(st, ev_thr, st.open_type, st.open_thr); st.open_type = EV_NONE; }
            }
            // ---------------- phase 2
            process_events(st, ev_thr, dt, limit, p, tid);
            if (finished) break;
        }
        close_run(st, p, tid);
        if (a.terminate_planes) a.terminate_planes[tid] = st.t_last;
        a.sm_cnts[tid] = st.n_samples;
        if (a.iv_cnts) a.iv_cnts[tid] = st.n_samples + st.n_chains;  // edges = samples + one leading edge per chain
        // rays with > 2^21 samples go to the serial fill too (the expansion packs a 27-bit batch offset)
        if (st.n_samples > (1 << 21) && st.n_runs <= p.max_runs) st.n_runs = p.max_runs + 1;
        p.run_cnts[tid] = st.n_runs;
        if (st.n_runs > p.max_runs) atomicAdd(p.overflow, 1);
    }
}

// ------------------------------------------------------------------------------------------
// Lane -> ray assignment for batches of unrelated rays: a counting sort of the ray ids by the length of the ray's
// path through the outermost grid box (BIN_COUNT bins), so that the 64 rays of a wave take about the same number of
// cells.  Three small launches; the order inside a bin is whatever the atomics give (results do not depend on it).
constexpr int BIN_COUNT = 256;

__device__ __forceinline__ int ray_bin(const float *__restrict__ rays_o, const float *__restrict__ rays_d, int64_t r,
                                       const float *__restrict__ box)
{
    const float ex = box[3] - box[0], ey = box[4] - box[1], ez = box[5] - box[2];
    const float inv_diag = 1.0f / sqrtf(ex * ex + ey * ey + ez * ez);
    float tmin = -INFINITY, tmax = INFINITY;
    bool hit = true;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float o = rays_o[3 * r + ax], inv = 1.0f / rays_d[3 * r + ax];
        float lo = (box[ax] - o) * inv, hi = (box[3 + ax] - o) * inv;
        if (lo > hi) { const float t = lo; lo = hi; hi = t; }
        if (!(lo <= hi)) hit = false;  // NaN: direction component 0 and origin on the slab plane
        tmin = fmaxf(tmin, lo); tmax = fminf(tmax, hi);
    }
    tmin = fmaxf(tmin, 0.0f);
    if (!hit || !(tmin < tmax)) return 0;
    const int b = 1 + (int)((tmax - tmin) * inv_diag * (float)(BIN_COUNT - 1));
    return b < 1 ? 1 : (b > BIN_COUNT - 1 ? BIN_COUNT - 1 : b);
}

// Every workgroup owns one contiguous range of rays in both passes, so the global histogram / cursors see 256 atomics
// per workgroup (not per 256 rays), and inside a workgroup the counting is LDS atomics.
__global__ __launch_bounds__(256) void bin_count_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                        int64_t n_rays, int64_t per_block, const float *__restrict__ box,
                                                        uint8_t *__restrict__ bins, int32_t *__restrict__ hist)
{
    __shared__ int32_t h[BIN_COUNT];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r_lo = (int64_t)blockIdx.x * per_block, r_hi = min(r_lo + per_block, n_rays);
    for (int64_t r = r_lo + threadIdx.x; r < r_hi; r += blockDim.x) {
        const int b = ray_bin(rays_o, rays_d, r, box);
        bins[r] = (uint8_t)b;
        atomicAdd(&h[b], 1);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// one workgroup: exclusive scan of the histogram in place (-> first output slot of every bin)
__global__ __launch_bounds__(256) void bin_scan_kernel(int32_t *__restrict__ hist)
{
    __shared__ int32_t h[BIN_COUNT];
    h[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t run = 0;
        for (int i = 0; i < BIN_COUNT; ++i) { const int32_t c = h[i]; h[i] = run; run += c; }
    }
    __syncthreads();
    hist[threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(const uint8_t *__restrict__ bins, int64_t n_rays, int64_t per_block,
                                                          int32_t *__restrict__ cursor, int32_t *__restrict__ order)
{
    __shared__ int32_t h[BIN_COUNT];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r_lo = (int64_t)blockIdx.x * per_block, r_hi = min(r_lo + per_block, n_rays);
    for (int64_t r = r_lo + threadIdx.x; r < r_hi; r += blockDim.x) atomicAdd(&h[bins[r]], 1);
    __syncthreads();
    const int32_t mine = h[threadIdx.x];
    __syncthreads();
    h[threadIdx.x] = mine ? atomicAdd(&cursor[threadIdx.x], mine) : 0;  // the workgroup's slots of this bin; then its local cursor
    __syncthreads();
    for (int64_t r = r_lo + threadIdx.x; r < r_hi; r += blockDim.x) order[atomicAdd(&h[bins[r]], 1)] = (int32_t)r;
}

// ------------------------------------------------------------------------------------------
// Expansion: runs -> (t_starts, t_ends, ray_indices).  One wave per batch of EXP_RPW rays.
// Staging: each half-wave lane copies alternate run records of "its" ray (slot-major records: one 256 B
// line per slot and batch, independent loads) to LDS entry {pos : 27 | local ray : 5, t_first}, pos = the
// run's first output relative to the batch; a ray whose runs overflowed gets one sentinel entry
// (t_first = NaN) so that its output range is skipped (the serial kernel fills it).  Outputs are then
// produced chunk by chunk (256 per step, 16 B per lane and array) with a scatter + "most recent entry"
// scan that tells every output its run.
enum { EXP_STARTS_ENDS = 0, EXP_MIDS = 1 /* the API's sample centres */, EXP_RAY_INDICES = 2 /* ray_indices only, from packed_info alone */,
       EXP_CONE = 3 /* (t_starts, t_ends) of cone-angle chains: the recurrence t <- t + max(step, t * cone) from the record's t_first */ };
template <int MODE>
__global__ __launch_bounds__(64 * EXP_WPB) void expand_runs_kernel(int64_t n_rays, float dt, const int32_t *__restrict__ run_cnts,
                                                          const unsigned long long *__restrict__ runs, int32_t max_runs,
                                                          const longlong2 *__restrict__ packed_info,
                                                          float *__restrict__ t_starts, float *__restrict__ t_ends,
                                                          float *__restrict__ t_mids, int64_t *__restrict__ ray_indices, int vec, float cone)
{
    __shared__ uint32_t s_pos[EXP_WPB][EXP_QMAX];
    __shared__ float s_t0[EXP_WPB][EXP_QMAX];
    __shared__ __attribute__((aligned(16))) int32_t s_own[EXP_WPB][256];  // entry that starts at each output of the current chunk
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t *pos = s_pos[wave];
    float *t0s = s_t0[wave];
    int32_t *slot = s_own[wave];
    const int64_t n_batches = ceil_div64(n_rays, EXP_RPW);
    for (int64_t batch = (int64_t)blockIdx.x * EXP_WPB + wave; batch < n_batches; batch += (int64_t)gridDim.x * EXP_WPB) {
        const int64_t r0 = batch * EXP_RPW;
        const int64_t ray = r0 + lane;
        const bool own = lane < EXP_RPW && ray < n_rays;
        int32_t c = 0, c_real = 0;
        int64_t s = 0, n = 0;
        if (own) {
            const longlong2 row = packed_info[ray];
            s = row.x;
            n = row.y;
            c_real = MODE == EXP_RAY_INDICES ? (n > 0 ? 1 : 0) : run_cnts[ray];  // (ray indices: the ray is one "run")
            c = (c_real > max_runs) ? 1 : c_real;  // overflowed ray: one sentinel entry
        }
        const int64_t W0 = __shfl(s, 0, 64);
        const int last_lane = (int)min((int64_t)EXP_RPW, n_rays - r0) - 1;
        const int64_t W1 = __shfl(s + n, last_lane, 64);
        int32_t incl = c;
#pragma unroll
        for (int off = 1; off < EXP_RPW; off <<= 1) {
            const int32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        const int32_t Q = __shfl(incl, EXP_RPW - 1, 64);
        // overflow flag per local ray as a wave-uniform mask
        const unsigned long long ovf_mask = __ballot(own && c_real > max_runs);
        if (MODE == EXP_RAY_INDICES && W1 - W0 >= ((int64_t)1 << 27)) {
            // the packed 27-bit positions below cannot address this window (rays of millions of samples): plain
            // cooperative fill, still coalesced
            for (int rl = 0; rl <= last_lane; ++rl) {
                const int64_t s_r = __shfl(s, rl, 64), n_r = __shfl(n, rl, 64);
                for (int64_t i = lane; i < n_r; i += 64) ray_indices[s_r + i] = r0 + rl;
            }
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        {   // staging: lane -> (ray rl, half); the two half-waves take alternate slots of every ray.  The loads of
            // different slots do not depend on each other (slot-major records: one 256 B line per slot and batch).
            const int rl = lane & (EXP_RPW - 1), half = lane >> 5;
            const int32_t c_rl = __shfl(c, rl, 64);
            const int32_t base_rl = __shfl(incl - c, rl, 64);
            const uint32_t rel_rl = (uint32_t)(__shfl(s, rl, 64) - W0);
            const bool ovf = (ovf_mask >> rl) & 1ull;
            int32_t c_max = c_rl;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) c_max = max(c_max, __shfl_xor(c_max, off, 64));
            const unsigned long long *col = runs + r0 + rl;
            for (int32_t i0 = 0; i0 < c_max; i0 += 8) {
                unsigned long long rec[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t i = i0 + 2 * u + half;
                    rec[u] = (MODE != EXP_RAY_INDICES && i < c_rl && !ovf) ? col[(int64_t)i * n_rays] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int32_t i = i0 + 2 * u + half;
                    if (i < c_rl) {
                        const uint32_t k_start = ovf ? 0u : ((uint32_t)(rec[u] >> 32) & 0x7FFFFFFFu);
                        const float t0 = ovf ? __builtin_nanf("") : bits_f32((uint32_t)rec[u]);
                        pos[base_rl + i] = ((rel_rl + k_start) & 0x7FFFFFFu) | ((uint32_t)rl << 27);
                        t0s[base_rl + i] = t0;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (Q > 0 && W1 > W0) {
            // Each 256-output chunk: scatter the indices of the entries that start in it into a 1 KiB LDS
            // line, then a "most recent entry" scan (4 elements per lane + 6 DPP steps, carry across chunks)
            // gives every output its run -- no per-element search.
            const int64_t c_first = (W0 / 256) * 256;
            int32_t q_next = 0;       // first entry not yet scattered (entries are sorted by position)
            int32_t carry_j = -1;     // entry covering the end of the previous chunk
            for (int64_t cb = c_first; cb < W1; cb += 256) {
                const int64_t lo64 = cb - W0;                           // chunk start relative to the window (may be < 0)
                *reinterpret_cast<int4 *>(slot + 4 * lane) = make_int4(-1, -1, -1, -1);
                __builtin_amdgcn_wave_barrier();
                for (;;) {
                    const int32_t q = q_next + lane;
                    bool take = false;
                    if (q < Q) {
                        const int64_t rel = (int64_t)(pos[q] & 0x7FFFFFFu) - lo64;
                        take = rel < 256;                                // rel >= 0: earlier entries were consumed
                        if (take) slot[(int)rel] = q;
                    }
                    const int cnt = __builtin_popcountll(__ballot(take));
                    q_next += cnt;
                    if (cnt < 64) break;
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
                    if (pa < W0 || pa >= W1 || j < 0) continue;
                    const float t0 = t0s[j];
                    if (t0 != t0) continue;  // sentinel: overflowed ray, filled by the serial kernel
                    const uint32_t e = pos[j];
                    const uint32_t kk = (uint32_t)(pa - W0) - (e & 0x7FFFFFFu);
                    if (MODE == EXP_CONE) {
                        // the serial loop's own arithmetic (grid.cu:213-216: dt = calc_dt(t_last), t_next = t_last + dt);
                        // inside a record the previous element's end is this element's start
                        if (k > 0 && valid[k - 1] && j4[k - 1] == j) {
                            ts4[k] = te4[k - 1];
                        } else {
                            float t = t0;
                            for (uint32_t i = 0; i < kk; ++i) t = t + calc_dt(t, cone, dt);
                            ts4[k] = t;
                        }
                        te4[k] = ts4[k] + calc_dt(ts4[k], cone, dt);
                    } else {
                        const float inc = (t0 + dt) - t0;  // the run's exact per-step increment
                        // t0 + k * inc is exactly representable for every sample of a run (that is what makes it a run),
                        // so one fused multiply-add (single rounding of the exact value) reproduces the serial sums
                        ts4[k] = __builtin_fmaf((float)kk, inc, t0);
                        te4[k] = __builtin_fmaf((float)(kk + 1), inc, t0);
                    }
                    ri4[k] = r0 + (e >> 27);
                    valid[k] = true;
                }
                if (MODE == EXP_MIDS) {  // API form of the samples (ref grid.cu:244: vals = (t_next + t_last) * 0.5f)
#pragma unroll
                    for (int k = 0; k < 4; ++k) ts4[k] = (te4[k] + ts4[k]) * 0.5f;
                }
                if (vec && valid[0] && valid[1] && valid[2] && valid[3]) {
                    if (MODE == EXP_MIDS) {
                        *reinterpret_cast<float4 *>(t_mids + p0) = make_float4(ts4[0], ts4[1], ts4[2], ts4[3]);
                    } else if (MODE == EXP_STARTS_ENDS || MODE == EXP_CONE) {
                        *reinterpret_cast<float4 *>(t_starts + p0) = make_float4(ts4[0], ts4[1], ts4[2], ts4[3]);
                        *reinterpret_cast<float4 *>(t_ends + p0) = make_float4(te4[0], te4[1], te4[2], te4[3]);
                    }
                    longlong2 *rp = reinterpret_cast<longlong2 *>(ray_indices + p0);
                    rp[0] = make_longlong2(ri4[0], ri4[1]);
                    rp[1] = make_longlong2(ri4[2], ri4[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (valid[k]) {
                            if (MODE == EXP_MIDS) t_mids[p0 + k] = ts4[k];
                            else if (MODE == EXP_STARTS_ENDS || MODE == EXP_CONE) { t_starts[p0 + k] = ts4[k]; t_ends[p0 + k] = te4[k]; }
                            ray_indices[p0 + k] = ri4[k];
                        }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// Expansion of the INTERVAL stream of the API's traverse_grids (ref grid.cu:219-262): a chain of
// continuous samples has one edge more than samples.  Run j of a ray (k_start_j samples before it, L_j chain
// starts before it) owns the edges [k_start_j + L_j, ...): its leading edge t_first if it starts a chain,
// then the end t_first + m * inc of each of its samples.  is_right is false exactly at chain starts, is_left
// is false exactly before a chain start (or the end of the ray's edges).  Same batch / chunk structure as
// expand_runs_kernel; one lane per ray stages that ray's records (it needs the running count of chain starts).
__global__ __launch_bounds__(64 * EXP_WPB) void expand_intervals_kernel(int64_t n_rays, float dt, const int32_t *__restrict__ run_cnts,
                                                               const unsigned long long *__restrict__ runs, int32_t max_runs,
                                                               const longlong2 *__restrict__ iv_packed_info,
                                                               float *__restrict__ vals, int64_t *__restrict__ ray_indices,
                                                               uint8_t *__restrict__ is_left, uint8_t *__restrict__ is_right, int vec)
{
    __shared__ uint32_t s_pos[EXP_WPB][EXP_QMAX];
    __shared__ float s_t0[EXP_WPB][EXP_QMAX];
    __shared__ uint8_t s_cont[EXP_WPB][EXP_QMAX];
    __shared__ __attribute__((aligned(16))) int32_t s_own[EXP_WPB][256 + 4];  // + look-ahead slot for the chunk's last edge
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t *pos = s_pos[wave];
    float *t0s = s_t0[wave];
    uint8_t *conts = s_cont[wave];
    int32_t *slot = s_own[wave];
    const int64_t n_batches = ceil_div64(n_rays, EXP_RPW);
    for (int64_t batch = (int64_t)blockIdx.x * EXP_WPB + wave; batch < n_batches; batch += (int64_t)gridDim.x * EXP_WPB) {
        const int64_t r0 = batch * EXP_RPW;
        const int64_t ray = r0 + lane;
        const bool mine = lane < EXP_RPW && ray < n_rays;
        int32_t c = 0, c_real = 0;
        int64_t s = 0, n = 0;
        if (mine) {
            c_real = run_cnts[ray];
            const longlong2 row = iv_packed_info[ray];
            s = row.x;
            n = row.y;
            c = (c_real > max_runs) ? 1 : c_real;  // overflowed ray: one sentinel entry
        }
        const int64_t W0 = __shfl(s, 0, 64);
        const int last = (int)min((int64_t)EXP_RPW, n_rays - r0) - 1;
        const int64_t W1 = __shfl(s + n, last, 64);
        int32_t incl = c;
#pragma unroll
        for (int off = 1; off < EXP_RPW; off <<= 1) {
            const int32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        const int32_t Q = __shfl(incl, EXP_RPW - 1, 64);
        __builtin_amdgcn_wave_barrier();
        if (mine && c > 0) {  // staging, one lane per ray
            const int32_t base = incl - c;
            const uint32_t rel = (uint32_t)(s - W0);
            if (c_real > max_runs) {
                pos[base] = (rel & 0x7FFFFFFu) | ((uint32_t)lane << 27);
                t0s[base] = __builtin_nanf("");
                conts[base] = 0;
            } else {
                const unsigned long long *col = runs + ray;
                uint32_t chains = 0;
                for (int32_t i0 = 0; i0 < c; i0 += 4) {
                    unsigned long long rec[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) rec[u] = (i0 + u < c) ? col[(int64_t)(i0 + u) * n_rays] : 0ull;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (i0 + u < c) {
                            const uint32_t hi = (uint32_t)(rec[u] >> 32);
                            const uint32_t k_start = hi & 0x7FFFFFFFu, cont = hi >> 31;
                            pos[base + i0 + u] = ((rel + k_start + chains) & 0x7FFFFFFu) | ((uint32_t)lane << 27);
                            t0s[base + i0 + u] = bits_f32((uint32_t)rec[u]);
                            conts[base + i0 + u] = (uint8_t)cont;
                            chains += cont ? 0u : 1u;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (Q > 0 && W1 > W0) {
            const int64_t c_first = (W0 / 256) * 256;
            int32_t q_next = 0, carry_j = -1;
            for (int64_t cb = c_first; cb < W1; cb += 256) {
                const int64_t lo64 = cb - W0;
                *reinterpret_cast<int4 *>(slot + 4 * lane) = make_int4(-1, -1, -1, -1);
                if (lane == 0) slot[256] = -1;
                __builtin_amdgcn_wave_barrier();
                for (;;) {
                    const int32_t q = q_next + lane;
                    bool take = false;
                    if (q < Q) {
                        const int64_t relq = (int64_t)(pos[q] & 0x7FFFFFFu) - lo64;
                        take = relq < 256;
                        if (relq <= 256) slot[(int)relq] = q;   // relq == 256: look-ahead only, consumed by the next chunk
                    }
                    const int cnt = __builtin_popcountll(__ballot(take));
                    q_next += cnt;
                    if (cnt < 64) break;
                }
                __builtin_amdgcn_wave_barrier();
                const int4 o4 = *reinterpret_cast<const int4 *>(slot + 4 * lane);
                const int32_t o_next_chunk = slot[256];
                __builtin_amdgcn_wave_barrier();
                const int32_t own4[4] = {o4.x, o4.y, o4.z, o4.w};
                // chain start flags of p .. p+3 and of p+4 (next lane's first edge, or the look-ahead slot)
                bool cs[5];
#pragma unroll
                for (int k = 0; k < 4; ++k) cs[k] = own4[k] >= 0 && conts[own4[k]] == 0;
                {
                    const int32_t nxt = __shfl_down((int32_t)(cs[0] ? 1 : 0), 1, 64);
                    cs[4] = lane == 63 ? (o_next_chunk >= 0 && conts[o_next_chunk] == 0) : (nxt != 0);
                }
                int32_t j4[4] = {own4[0], own4[1], own4[2], own4[3]};
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
                float v4[4];
                int64_t ri4[4];
                uint8_t l4[4], r4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t pa = p0 + k;
                    const int32_t j = j4[k];
                    valid[k] = false;
                    v4[k] = 0.f; ri4[k] = 0; l4[k] = r4[k] = 0;
                    if (pa < W0 || pa >= W1 || j < 0) continue;
                    const float t0 = t0s[j];
                    if (t0 != t0) continue;  // sentinel: overflowed ray, filled by the serial kernel
                    const uint32_t e = pos[j];
                    const uint32_t m = (uint32_t)(pa - W0) - (e & 0x7FFFFFFu) + (uint32_t)conts[j];
                    const float inc = (t0 + dt) - t0;
                    v4[k] = __builtin_fmaf((float)m, inc, t0);
                    ri4[k] = r0 + (e >> 27);
                    r4[k] = cs[k] ? 0 : 1;
                    l4[k] = (cs[k + 1] || pa + 1 >= W1) ? 0 : 1;
                    valid[k] = true;
                }
                if (vec && valid[0] && valid[1] && valid[2] && valid[3]) {
                    *reinterpret_cast<float4 *>(vals + p0) = make_float4(v4[0], v4[1], v4[2], v4[3]);
                    longlong2 *rp = reinterpret_cast<longlong2 *>(ray_indices + p0);
                    rp[0] = make_longlong2(ri4[0], ri4[1]);
                    rp[1] = make_longlong2(ri4[2], ri4[3]);
                    *reinterpret_cast<uchar4 *>(is_left + p0) = make_uchar4(l4[0], l4[1], l4[2], l4[3]);
                    *reinterpret_cast<uchar4 *>(is_right + p0) = make_uchar4(r4[0], r4[1], r4[2], r4[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (valid[k]) { vals[p0 + k] = v4[k]; ray_indices[p0 + k] = ri4[k]; is_left[p0 + k] = l4[k]; is_right[p0 + k] = r4[k]; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace nfa

using namespace nfa;

extern "C" {

int64_t nfa_bricks_words(int32_t n_grids, const int32_t *res)
{
    return (int64_t)n_grids * ((res[0] + 3) / 4) * ((res[1] + 3) / 4) * ((res[2] + 3) / 4);
}

int nfa_pack_bricks(const uint8_t *binaries, int32_t n_grids, const int32_t *res, uint64_t *bricks, uint32_t *coarse,
                    nfa_stream_t stream)
{
    NFA_REQUIRE(binaries && res && bricks && coarse && n_grids >= 1, "pack_bricks: bad arguments");
    NFA_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0, "pack_bricks: bad resolution");
    const int32_t bx = (res[0] + 3) / 4, by = (res[1] + 3) / 4, bz = (res[2] + 3) / 4;
    const int64_t nb = (int64_t)n_grids * bx * by * bz;
    NFA_REQUIRE(nb < ((int64_t)1 << 31), "pack_bricks: grid too large");
    NFA_REQUIRE((int64_t)bx * by < ((int64_t)1 << 24) && bz < (1 << 24),
                "pack_bricks: more than 2^24 bricks in an x-y slice (brick indices are formed with 24-bit multiplies)");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(coarse, 0, (size_t)((nb + 31) / 32) * 4, s) != hipSuccess) { set_error("pack_bricks: memset failed"); return NFA_EHIP; }
    hipLaunchKernelGGL(pack_bricks_kernel, dim3(grid_1d(nb, 256)), dim3(256), 0, s, binaries, n_grids, res[0], res[1], res[2],
                       bx, by, bz, reinterpret_cast<unsigned long long *>(bricks), coarse);
    NFA_CHECK_LAUNCH("pack_bricks");
    return NFA_OK;
}

int nfa_bin_rays(const float *rays_o, const float *rays_d, int64_t n_rays, const float *box, int32_t *order,
                 void *scratch, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), "bin_rays: n_rays out of range");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(rays_o && rays_d && box && order && scratch, "bin_rays: null pointer");
    hipStream_t s = as_stream(stream);
    int32_t *hist = reinterpret_cast<int32_t *>(scratch);                      // [BIN_COUNT]
    uint8_t *bins = reinterpret_cast<uint8_t *>(hist + BIN_COUNT);             // [n_rays]
    if (hipMemsetAsync(hist, 0, sizeof(int32_t) * BIN_COUNT, s) != hipSuccess) { set_error("bin_rays: memset failed"); return NFA_EHIP; }
    int64_t per_block = ceil_div64(n_rays, 512);  // up to 512 workgroups (two per CU), at least 1024 rays each
    if (per_block < 1024) per_block = 1024;
    const unsigned grid = (unsigned)ceil_div64(n_rays, per_block);
    hipLaunchKernelGGL(bin_count_kernel, dim3(grid), dim3(256), 0, s, rays_o, rays_d, n_rays, per_block, box, bins, hist);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(256), 0, s, hist);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(grid), dim3(256), 0, s, bins, n_rays, per_block, hist, order);
    NFA_CHECK_LAUNCH("bin_rays");
    return NFA_OK;
}

int nfa_traverse_runs(const nfa_traverse_args *pa, const uint64_t *bricks, const uint32_t *coarse, int32_t *run_cnts,
                      uint64_t *runs, int32_t max_runs, int32_t *overflow_count, float near_hint,
                      const int32_t *ray_order, nfa_stream_t stream)
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
    NFA_REQUIRE(a.rays_o && a.rays_d && a.aabbs && a.near_planes && a.far_planes && a.sm_cnts && bricks && coarse &&
                    run_cnts && runs, "traverse_runs: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs * EXP_RPW <= EXP_QMAX, "traverse_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(a.n_grids >= 1 && a.res[0] > 0 && a.res[1] > 0 && a.res[2] > 0, "traverse_runs: bad grid shape");
    const bool fused = !a.t_sorted && !a.t_indices && !a.hits;
    NFA_REQUIRE(fused || (a.t_sorted && a.t_indices && a.hits), "traverse_runs: t_sorted, t_indices and hits must be given together");
    NFA_REQUIRE(!fused || a.n_grids == 1, "traverse_runs: in-kernel intersection supports one grid");
    RunsParams p;
    p.bricks = reinterpret_cast<const unsigned long long *>(bricks);
    p.coarse = coarse;
    p.bx = (a.res[0] + 3) / 4; p.by = (a.res[1] + 3) / 4; p.bz = (a.res[2] + 3) / 4;
    const int64_t nb = (int64_t)a.n_grids * p.bx * p.by * p.bz;
    NFA_REQUIRE(nb < ((int64_t)1 << 31), "traverse_runs: grid too large");
    p.n_coarse_words = (int32_t)((nb + 31) / 32);
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
    int64_t lds_words = COARSE_LDS_WORDS;
    if (const char *e = getenv("NFA_COARSE_LDS_WORDS")) lds_words = atol(e);  // tuning experiments
    const bool lds = p.n_coarse_words <= lds_words;
    const size_t shmem = (size_t)EV_MAX * 256 * 4 + (lds ? (size_t)p.n_coarse_words * 4 : 0);
    const unsigned grid = grid_1d(a.n_rays, 256, 1 << 20);
    if (fused) {
        if (lds) hipLaunchKernelGGL((runs_kernel<true, true>), dim3(grid), dim3(256), shmem, s, a, p);
        else     hipLaunchKernelGGL((runs_kernel<true, false>), dim3(grid), dim3(256), shmem, s, a, p);
    } else {
        if (lds) hipLaunchKernelGGL((runs_kernel<false, true>), dim3(grid), dim3(256), shmem, s, a, p);
        else     hipLaunchKernelGGL((runs_kernel<false, false>), dim3(grid), dim3(256), shmem, s, a, p);
    }
    NFA_CHECK_LAUNCH("traverse_runs");
    return NFA_OK;
}

int nfa_expand_runs(int64_t n_rays, float step_size, const int32_t *run_cnts, const uint64_t *runs, int32_t max_runs,
                    const int64_t *packed_info, float *t_starts, float *t_ends, float *t_mids,
                    int64_t *ray_indices, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0, "expand_runs: negative n_rays");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(run_cnts && runs && packed_info && ray_indices && (t_mids || (t_starts && t_ends)),
                "expand_runs: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs * EXP_RPW <= EXP_QMAX, "expand_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f, "expand_runs: step_size must be > 0");
    const int vec = ((reinterpret_cast<uintptr_t>(t_starts) | reinterpret_cast<uintptr_t>(t_ends) |
                      reinterpret_cast<uintptr_t>(t_mids) | reinterpret_cast<uintptr_t>(ray_indices)) & 15) == 0;
    const unsigned grid = grid_1d(ceil_div64(n_rays, EXP_RPW) * 64, 64 * EXP_WPB, 1 << 22);
    if (t_mids)
        hipLaunchKernelGGL(expand_runs_kernel<EXP_MIDS>, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                           reinterpret_cast<const unsigned long long *>(runs), max_runs,
                           reinterpret_cast<const longlong2 *>(packed_info), t_starts, t_ends, t_mids, ray_indices, vec, 0.0f);
    else
        hipLaunchKernelGGL(expand_runs_kernel<EXP_STARTS_ENDS>, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                           reinterpret_cast<const unsigned long long *>(runs), max_runs,
                           reinterpret_cast<const longlong2 *>(packed_info), t_starts, t_ends, t_mids, ray_indices, vec, 0.0f);
    NFA_CHECK_LAUNCH("expand_runs");
    return NFA_OK;
}

int nfa_expand_cone_runs(int64_t n_rays, float step_size, float cone_angle, const int32_t *run_cnts, const uint64_t *runs,
                         int32_t max_runs, const int64_t *packed_info, float *t_starts, float *t_ends, int64_t *ray_indices,
                         nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0, "expand_cone_runs: negative n_rays");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(run_cnts && runs && packed_info && ray_indices && t_starts && t_ends, "expand_cone_runs: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs * EXP_RPW <= EXP_QMAX, "expand_cone_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f && cone_angle > 0.0f, "expand_cone_runs: step_size and cone_angle must be > 0");
    const int vec = ((reinterpret_cast<uintptr_t>(t_starts) | reinterpret_cast<uintptr_t>(t_ends) |
                      reinterpret_cast<uintptr_t>(ray_indices)) & 15) == 0;
    const unsigned grid = grid_1d(ceil_div64(n_rays, EXP_RPW) * 64, 64 * EXP_WPB, 1 << 22);
    hipLaunchKernelGGL(expand_runs_kernel<EXP_CONE>, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                       reinterpret_cast<const unsigned long long *>(runs), max_runs,
                       reinterpret_cast<const longlong2 *>(packed_info), t_starts, t_ends, nullptr, ray_indices, vec, cone_angle);
    NFA_CHECK_LAUNCH("expand_cone_runs");
    return NFA_OK;
}

int nfa_fill_ray_indices(int64_t n_rays, const int64_t *packed_info, int64_t *ray_indices, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0, "fill_ray_indices: negative n_rays");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(packed_info && ray_indices, "fill_ray_indices: null pointer");
    const int vec = (reinterpret_cast<uintptr_t>(ray_indices) & 15) == 0;
    const unsigned grid = grid_1d(ceil_div64(n_rays, EXP_RPW) * 64, 64 * EXP_WPB, 1 << 22);
    hipLaunchKernelGGL(expand_runs_kernel<EXP_RAY_INDICES>, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, 0.0f,
                       nullptr, nullptr, 1, reinterpret_cast<const longlong2 *>(packed_info), nullptr, nullptr, nullptr,
                       ray_indices, vec, 0.0f);
    NFA_CHECK_LAUNCH("fill_ray_indices");
    return NFA_OK;
}

int nfa_expand_intervals(int64_t n_rays, float step_size, const int32_t *run_cnts, const uint64_t *runs, int32_t max_runs,
                         const int64_t *iv_packed_info, float *vals, int64_t *ray_indices,
                         uint8_t *is_left, uint8_t *is_right, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0, "expand_intervals: negative n_rays");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(run_cnts && runs && iv_packed_info && vals && ray_indices && is_left && is_right,
                "expand_intervals: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs * EXP_RPW <= EXP_QMAX, "expand_intervals: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f, "expand_intervals: step_size must be > 0");
    const int vec = ((reinterpret_cast<uintptr_t>(vals) | reinterpret_cast<uintptr_t>(ray_indices)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(is_left) | reinterpret_cast<uintptr_t>(is_right)) & 3) == 0;
    const unsigned grid = grid_1d(ceil_div64(n_rays, EXP_RPW) * 64, 64 * EXP_WPB, 1 << 22);
    hipLaunchKernelGGL(expand_intervals_kernel, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                       reinterpret_cast<const unsigned long long *>(runs), max_runs,
                       reinterpret_cast<const longlong2 *>(iv_packed_info), vals, ray_indices, is_left, is_right, vec);
    NFA_CHECK_LAUNCH("expand_intervals");
    return NFA_OK;
}

}  // extern "C"
