// traverse2.hip -- run-length traversal for the sampler (constant step, cone_angle == 0).
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
//                 Marching is the exact O(#binades) fast-forward of march.h, deferred across runs
//                 of empty cells (legal for a constant step: see flush_pending).  Instead of
//                 samples the pass emits RUNS: (t_first, n) for n consecutive samples with one
//                 exact fp32 increment -- typically 3-10 per ray (8 B each).
//   expand pass   after the device-side cumsum of the sample counts, every output element is
//                 computed independently: t_start = t_first + k*inc, t_end = t_first + (k+1)*inc
//                 (exact: multiples of one ulp inside a binade), ray index from the run.  A wave
//                 stages the runs of 32 rays in LDS and streams its contiguous output range with
//                 16 B-per-lane stores: the 16 B/sample of the sampler's output are written once,
//                 fully coalesced.
// Results are bit-identical to the reference's serial accumulation (oracle/nerfacc_oracle.c).
#include "common.hip.h"
#include "march.h"

namespace nfa {

constexpr int EXP_RPW = 32;        // rays per wave batch in the expansion
constexpr int EXP_QMAX = 1024;     // runs staged per batch (EXP_RPW * max_runs)
constexpr int COARSE_LDS_WORDS = 8192;  // 32 KiB: up to 64^3 bricks (256^3 cells) per level set

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
    unsigned long long *runs;    // [n_rays, max_runs]
    int32_t max_runs;
    int32_t *overflow;           // [1] number of rays with more runs than max_runs
};

struct RunState {
    float t_last;
    float pend;          // deferred fast-forward target (valid when has_pend)
    bool has_pend;
    bool continuous;
    int32_t n_samples, n_runs;
    // open run
    bool open, run_cont;
    float run_t0, run_inc;
    int32_t run_n;
    // brick cache
    int32_t brick_id;
    unsigned long long brick_word;
};

__device__ __forceinline__ void close_run(RunState &st, const RunsParams &p, int64_t tid)
{
    if (!st.open) return;
    if (st.n_runs < p.max_runs)
        p.runs[tid * p.max_runs + st.n_runs] =
            (unsigned long long)f32_bits(st.run_t0) | ((unsigned long long)((uint32_t)st.run_n | (st.run_cont ? 0x80000000u : 0u)) << 32);
    st.n_runs++;
    st.open = false;
}

// Consecutive fast-forwards with non-decreasing targets and one constant dt collapse into a single
// one to the last target (the loop `while (t + dt/2 < target) t += dt` is monotone in target), so
// empty cells only record the target and the marching happens once per run of empty cells.
__device__ __forceinline__ void flush_pending(RunState &st, float dt)
{
    if (st.has_pend) {
        st.t_last = fast_forward_exact(st.t_last, st.pend, dt);
        st.has_pend = false;
    }
}

template <bool COARSE_LDS>
__device__ __forceinline__ void runs_span(const nfa_traverse_args &a, const RunsParams &p, const uint32_t *coarse_lds,
                                          int64_t tid, const float o[3], const float d[3], const float inv[3],
                                          int32_t level, float this_tmin, float this_tmax, RunState &st)
{
    const float eps = 1e-6f;
    const float dt = a.step_size, half = dt * 0.5f;
    const int32_t limit = a.traverse_steps_limit;
    if (!st.continuous) {  // grid.cu:153-163, deferred
        st.pend = st.has_pend ? fmaxf(st.pend, this_tmin) : this_tmin;
        st.has_pend = true;
    }
    const float *bmin = a.aabbs + 6 * level, *bmax = bmin + 3;
    float tdist[3], delta[3];
    int32_t step[3], cur[3], overflow[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {  // setup_traversal, include/utils_grid.cuh:58-114
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
        tdist[ax] = (d[ax] == 0.0f) ? this_tmax : tmax_ax;
        step[ax] = (int32_t)step_f;
        const float delta_tmp = voxel * inv[ax] * step_f;
        delta[ax] = (d[ax] == 0.0f) ? this_tmax : delta_tmp;
        cur[ax] = c;
        overflow[ax] = f + step[ax];
    }
    const int32_t lvl_brick_base = level * p.bx * p.by * p.bz;
    int32_t cells_left = a.res[0] + a.res[1] + a.res[2] + 3;

    auto occupied = [&]() -> bool {  // occupancy of the current cell through the brick cache
        const int32_t bid = lvl_brick_base + ((cur[0] >> 2) * p.by + (cur[1] >> 2)) * p.bz + (cur[2] >> 2);
        if (bid != st.brick_id) {
            st.brick_id = bid;
            const uint32_t cw = COARSE_LDS ? coarse_lds[bid >> 5] : p.coarse[bid >> 5];
            st.brick_word = ((cw >> (bid & 31)) & 1u) ? p.bricks[bid] : 0ull;
        }
        const int bit = ((cur[0] & 3) << 4) | ((cur[1] & 3) << 2) | (cur[2] & 3);
        return (st.brick_word >> bit) & 1ull;
    };
    auto advance = [&]() -> bool {  // single_traversal, include/utils_grid.cuh:116-142; false at the end
        const int ax = (tdist[0] < tdist[1] && tdist[0] < tdist[2]) ? 0 : (tdist[1] < tdist[2] ? 1 : 2);
        bool done;
        if (ax == 0)      { cur[0] += step[0]; tdist[0] += delta[0]; done = cur[0] == overflow[0]; }
        else if (ax == 1) { cur[1] += step[1]; tdist[1] += delta[1]; done = cur[1] == overflow[1]; }
        else              { cur[2] += step[2]; tdist[2] += delta[2]; done = cur[2] == overflow[2]; }
        return !(done || --cells_left <= 0);
    };

    // The reference's cell loop (grid.cu:184-272) in lock-step phases, so that a wave runs the
    // expensive part (marching + emission) once per RUN of occupied cells instead of once per cell
    // iteration in which any lane happens to need it:
    //   A: walk empty cells (cheap DDA steps), only recording the marching target;
    //   B: march once, then emit samples through consecutive occupied cells.
    bool alive = true;
    while (alive && (limit <= 0 || st.n_samples < limit)) {
        // ---- phase A
        bool occ = false;
        while (alive) {
            occ = occupied();
            if (occ) break;
            st.pend = fminf(fminf(tdist[0], fminf(tdist[1], tdist[2])), this_tmax);  // non-decreasing along the ray
            st.has_pend = true;
            st.continuous = false;
            alive = advance();
        }
        if (!occ) break;
        // ---- phase B
        flush_pending(st, dt);
        while (limit <= 0 || st.n_samples < limit) {
            const float t_traverse = fminf(fminf(tdist[0], fminf(tdist[1], tdist[2])), this_tmax);
            while (limit <= 0 || st.n_samples < limit) {  // grid.cu:208-261
                if (st.t_last + half >= t_traverse) break;
                const float t_next = st.t_last + dt;
                if (t_next == st.t_last) break;
                const float inc = t_next - st.t_last;  // exact (Sterbenz)
                if (st.open && st.continuous && inc == st.run_inc) {
                    st.run_n++;
                } else {
                    close_run(st, p, tid);
                    st.open = true; st.run_t0 = st.t_last; st.run_inc = inc; st.run_n = 1; st.run_cont = st.continuous;
                }
                st.n_samples++;
                st.continuous = true;
                st.t_last = t_next;
                if (t_next >= t_traverse) break;
            }
            alive = advance();
            if (!alive || !occupied()) break;  // an empty cell: back to phase A (it re-tests the cell)
        }
    }
}

template <bool FUSED, bool COARSE_LDS>
__global__ __launch_bounds__(256) void runs_kernel(const nfa_traverse_args a, const RunsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t coarse_lds[];
    if (COARSE_LDS) {
        for (int i = threadIdx.x; i < p.n_coarse_words; i += blockDim.x) coarse_lds[i] = p.coarse[i];
        __syncthreads();
    }
    for (int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < a.n_rays;
         tid += (int64_t)blockDim.x * gridDim.x) {
        if (a.mode == 2 && a.rays_mask != nullptr && !a.rays_mask[tid]) {
            if (a.terminate_planes) a.terminate_planes[tid] = a.near_planes[tid];
            a.sm_cnts[tid] = 0;
            p.run_cnts[tid] = 0;
            continue;
        }
        const float near_plane = a.near_planes[tid], far_plane = a.far_planes[tid];
        const float o[3] = {a.rays_o[3 * tid], a.rays_o[3 * tid + 1], a.rays_o[3 * tid + 2]};
        const float d[3] = {a.rays_d[3 * tid], a.rays_d[3 * tid + 1], a.rays_d[3 * tid + 2]};
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        RunState st;
        st.t_last = near_plane; st.has_pend = false; st.pend = 0.f; st.continuous = false;
        st.n_samples = 0; st.n_runs = 0; st.open = false; st.run_cont = false; st.run_t0 = 0.f; st.run_inc = 0.f;
        st.run_n = 0; st.brick_id = -1; st.brick_word = 0ull;
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
            if (hit) {
                const float this_tmin = fmaxf(tmin, near_plane), this_tmax = fminf(tmax, far_plane);
                if (this_tmin < this_tmax) runs_span<COARSE_LDS>(a, p, coarse_lds, tid, o, d, inv, 0, this_tmin, this_tmax, st);
            }
        } else {
            const int32_t G = a.n_grids;
            const uint8_t *hits = a.hits + tid * G;
            const float *ts = a.t_sorted + tid * 2 * G;
            const int64_t *ti = a.t_indices + tid * 2 * G;
            for (int32_t i = 0; i < 2 * G - 1; ++i) {  // grid.cu:125-150
                const int64_t idx = ti[i];
                int32_t level = (int32_t)(idx % G);
                if (!hits[level]) continue;
                if (!(idx < G)) {
                    const int64_t nidx = ti[i + 1];
                    if (nidx < G) continue;
                    level = (int32_t)(nidx % G);
                    if (!hits[level]) continue;
                }
                const float this_tmin = fmaxf(ts[i], near_plane), this_tmax = fminf(ts[i + 1], far_plane);
                if (this_tmin >= this_tmax) continue;
                runs_span<COARSE_LDS>(a, p, coarse_lds, tid, o, d, inv, level, this_tmin, this_tmax, st);
            }
        }
        flush_pending(st, a.step_size);
        close_run(st, p, tid);
        if (a.terminate_planes) a.terminate_planes[tid] = st.t_last;
        a.sm_cnts[tid] = st.n_samples;
        p.run_cnts[tid] = st.n_runs;
        if (st.n_runs > p.max_runs) atomicAdd(p.overflow, 1);
    }
}

// ------------------------------------------------------------------------------------------
// Expansion: runs -> (t_starts, t_ends, ray_indices).  One wave per batch of EXP_RPW rays.
__global__ __launch_bounds__(256) void expand_runs_kernel(int64_t n_rays, float dt, const int32_t *__restrict__ run_cnts,
                                                          const unsigned long long *__restrict__ runs, int32_t max_runs,
                                                          const int64_t *__restrict__ sm_starts,
                                                          const int64_t *__restrict__ sm_cnts,
                                                          float *__restrict__ t_starts, float *__restrict__ t_ends,
                                                          int64_t *__restrict__ ray_indices, int vec)
{
    __shared__ uint32_t s_pos[4][EXP_QMAX];
    __shared__ float s_t0[4][EXP_QMAX];
    __shared__ uint32_t s_meta[4][EXP_QMAX];  // n (24 bits) | local ray (8 bits)
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t *pos = s_pos[wave];
    float *t0s = s_t0[wave];
    uint32_t *meta = s_meta[wave];
    const int64_t n_batches = ceil_div64(n_rays, EXP_RPW);
    for (int64_t batch = (int64_t)blockIdx.x * 4 + wave; batch < n_batches; batch += (int64_t)gridDim.x * 4) {
        const int64_t r0 = batch * EXP_RPW;
        const int64_t ray = r0 + lane;
        const bool own = lane < EXP_RPW && ray < n_rays;
        int32_t c = 0;
        int64_t s = 0, n = 0;
        if (own) {
            c = run_cnts[ray];
            if (c > max_runs) c = 0;  // overflowed ray: filled by the serial kernel instead
            s = sm_starts[ray];
            n = sm_cnts[ray];
        }
        const int64_t W0 = __shfl(s, 0, 64);
        // end of the batch's output range = max over owned rays of (s + n): lane of the last ray
        const int last_lane = (int)min((int64_t)EXP_RPW, n_rays - r0) - 1;
        const int64_t W1 = __shfl(s + n, last_lane, 64);
        // exclusive scan of the run counts over the 32 owning lanes
        int32_t incl = c;
#pragma unroll
        for (int off = 1; off < EXP_RPW; off <<= 1) {
            const int32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        const int32_t Q = __shfl(incl, EXP_RPW - 1, 64);
        int32_t q = incl - c;
        uint32_t rel = (uint32_t)(s - W0);
        for (int32_t j = 0; j < c; ++j) {
            const unsigned long long rec = runs[ray * max_runs + j];
            const uint32_t nn = (uint32_t)(rec >> 32) & 0x7FFFFFFFu;
            pos[q] = rel;
            t0s[q] = bits_f32((uint32_t)rec);
            meta[q] = (nn & 0xFFFFFFu) | ((uint32_t)lane << 24);
            rel += nn;
            ++q;
        }
        __builtin_amdgcn_wave_barrier();
        if (Q > 0 && W1 > W0) {
            const int64_t c_first = (W0 / 256) * 256;
            for (int64_t cb = c_first; cb < W1; cb += 256) {
                const int64_t p0 = cb + 4 * lane;
                bool valid[4];
                float ts4[4], te4[4];
                int64_t ri4[4];
                // run containing the first in-range position of this lane
                int32_t j = -1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t pa = p0 + k;
                    valid[k] = false;
                    ts4[k] = te4[k] = 0.f; ri4[k] = 0;
                    if (pa < W0 || pa >= W1) continue;
                    const uint32_t pr = (uint32_t)(pa - W0);
                    if (j < 0) {  // upper_bound(pos, pr) - 1
                        int32_t lo = 0, hi = Q;
                        while (lo < hi) {
                            const int32_t mid = (lo + hi) >> 1;
                            if (pos[mid] <= pr) lo = mid + 1; else hi = mid;
                        }
                        j = lo - 1;
                    } else {
                        while (j + 1 < Q && pos[j + 1] <= pr) ++j;
                    }
                    if (j < 0) continue;  // before the first run (an overflowed ray's range)
                    const uint32_t m = meta[j];
                    const uint32_t kk = pr - pos[j];
                    if (kk >= (m & 0xFFFFFFu)) continue;  // not covered by a run (overflowed ray)
                    const float t0 = t0s[j];
                    const float inc = (t0 + dt) - t0;  // the run's exact per-step increment
                    ts4[k] = (float)((double)t0 + (double)kk * (double)inc);
                    te4[k] = (float)((double)t0 + (double)(kk + 1) * (double)inc);
                    ri4[k] = r0 + (m >> 24);
                    valid[k] = true;
                }
                if (vec && valid[0] && valid[1] && valid[2] && valid[3]) {
                    *reinterpret_cast<float4 *>(t_starts + p0) = make_float4(ts4[0], ts4[1], ts4[2], ts4[3]);
                    *reinterpret_cast<float4 *>(t_ends + p0) = make_float4(te4[0], te4[1], te4[2], te4[3]);
                    longlong2 *rp = reinterpret_cast<longlong2 *>(ray_indices + p0);
                    rp[0] = make_longlong2(ri4[0], ri4[1]);
                    rp[1] = make_longlong2(ri4[2], ri4[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (valid[k]) { t_starts[p0 + k] = ts4[k]; t_ends[p0 + k] = te4[k]; ray_indices[p0 + k] = ri4[k]; }
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
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(coarse, 0, (size_t)((nb + 31) / 32) * 4, s) != hipSuccess) { set_error("pack_bricks: memset failed"); return NFA_EHIP; }
    hipLaunchKernelGGL(pack_bricks_kernel, dim3(grid_1d(nb, 256)), dim3(256), 0, s, binaries, n_grids, res[0], res[1], res[2],
                       bx, by, bz, reinterpret_cast<unsigned long long *>(bricks), coarse);
    NFA_CHECK_LAUNCH("pack_bricks");
    return NFA_OK;
}

int nfa_traverse_runs(const nfa_traverse_args *pa, const uint64_t *bricks, const uint32_t *coarse, int32_t *run_cnts,
                      uint64_t *runs, int32_t max_runs, int32_t *overflow_count, nfa_stream_t stream)
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
    p.overflow = overflow_count;
    const bool lds = p.n_coarse_words <= COARSE_LDS_WORDS;
    const size_t shmem = lds ? (size_t)p.n_coarse_words * 4 : 0;
    const unsigned grid = grid_1d(a.n_rays, 256, 1 << 20);
    if (fused) {
        if (lds) hipLaunchKernelGGL((runs_kernel<true, true>), dim3(grid), dim3(256), shmem, s, a, p);
        else     hipLaunchKernelGGL((runs_kernel<true, false>), dim3(grid), dim3(256), 0, s, a, p);
    } else {
        if (lds) hipLaunchKernelGGL((runs_kernel<false, true>), dim3(grid), dim3(256), shmem, s, a, p);
        else     hipLaunchKernelGGL((runs_kernel<false, false>), dim3(grid), dim3(256), 0, s, a, p);
    }
    NFA_CHECK_LAUNCH("traverse_runs");
    return NFA_OK;
}

int nfa_expand_runs(int64_t n_rays, float step_size, const int32_t *run_cnts, const uint64_t *runs, int32_t max_runs,
                    const int64_t *sm_starts, const int64_t *sm_cnts, float *t_starts, float *t_ends,
                    int64_t *ray_indices, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0, "expand_runs: negative n_rays");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(run_cnts && runs && sm_starts && sm_cnts && t_starts && t_ends && ray_indices, "expand_runs: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs * EXP_RPW <= EXP_QMAX, "expand_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f, "expand_runs: step_size must be > 0");
    const int vec = ((reinterpret_cast<uintptr_t>(t_starts) | reinterpret_cast<uintptr_t>(t_ends) |
                      reinterpret_cast<uintptr_t>(ray_indices)) & 15) == 0;
    const int64_t n_batches = ceil_div64(n_rays, EXP_RPW);
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(n_batches, 4), 1 << 20);
    hipLaunchKernelGGL(expand_runs_kernel, dim3(grid), dim3(256), 0, as_stream(stream), n_rays, step_size, run_cnts,
                       reinterpret_cast<const unsigned long long *>(runs), max_runs, sm_starts, sm_cnts, t_starts,
                       t_ends, ray_indices, vec);
    NFA_CHECK_LAUNCH("expand_runs");
    return NFA_OK;
}

}  // extern "C"
