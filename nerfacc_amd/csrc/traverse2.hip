// traverse2.hip -- what surrounds the run-length walk (walk.hip) of the constant-step traversal: the brick-packed grid
// copy used by the serial kernels, the ray binning, and the coalesced EXPANSIONS of the run records.
//
//   pack_bricks   binaries (torch.bool, 1 B/cell)  ->  4x4x4-cell bricks, one 64-bit word each,
//                 plus a 1-bit-per-brick "any occupied" mask (grid.hip's serial kernels read occupancy from them).
//   expand passes after the device-side cumsum of the counts, every output element is computed
//                 independently: t = fma(k, inc, t_first) (exact: every sample of a run is t_first + k*inc).
//                 A wave stages the records of 32 rays in LDS; per 256-output chunk the run starts are
//                 scattered into an LDS line and a "most recent entry" DPP scan tells every output its
//                 run; 16 B-per-lane stores.  expand_runs: (t_starts, t_ends, ray_indices) or the API's
//                 sample centres; expand_intervals: the API's edge stream with is_left / is_right.
// Results are bit-identical to the reference's serial accumulation (oracle/nerfacc_oracle.c).
#include "common.hip.h"
#ifndef NFA_NT_EXPAND
#define NFA_NT_EXPAND 1   /* sample arrays are written once and read by later kernels long after they left L2: non-temporal stores, expand 160 -> 152 us and the whole cfg-2 step -40 us (A/B on one box); the engine ops lose with them */
#endif
#include "march.h"

namespace nfa {

constexpr int EXP_RPW = 32;        // rays per wave batch in the expansion
#ifndef NFA_EXP_QMAX
#define NFA_EXP_QMAX 1024
#endif
constexpr int EXP_QMAX = NFA_EXP_QMAX;  // runs staged per batch (EXP_RPW * max_runs)
#ifndef NFA_EXP_IV_QMAX
#define NFA_EXP_IV_QMAX 256
#endif
#ifndef NFA_EXP_RUNS_QMAX
#define NFA_EXP_RUNS_QMAX 1024  /* one group; 256 (2 more waves per SIMD) measured 2 % slower here (156 vs 152.5 us), where it made expand_intervals 26 % faster */
#endif
constexpr int EXP_RUNS_QMAX = NFA_EXP_RUNS_QMAX;  // expand_runs stages a batch's records this many at a time (>= 32: one ray's)
static_assert(EXP_RUNS_QMAX >= 32 && EXP_RUNS_QMAX <= EXP_QMAX, "expand_runs: a ray's records must fit the staging area");
constexpr int EXP_IV_QMAX = NFA_EXP_IV_QMAX;  // expand_intervals stages a batch's records this many at a time (>= 32: one ray's)
static_assert(EXP_IV_QMAX >= 32 && EXP_IV_QMAX <= EXP_QMAX, "expand_intervals: a ray's records must fit the staging area");
#ifndef NFA_EXP_WPB
#define NFA_EXP_WPB 2  /* measured on cfg 2: 1 wave 176 us, 2 waves 160 us, 4 waves 175 us */
#endif
constexpr int EXP_WPB = NFA_EXP_WPB;  // waves per workgroup of the expansion kernels (they never cooperate)

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

// The same for nested levels (rays that start inside the finest box and leave through the coarser ones): the key is the
// number of cell boundaries the ray crosses, sum over levels of (length inside level l but outside level l - 1) x
// sum_k |d_k| res_k / extent_k, from the near plane on.  box: [n_grids][6], finest first.
__device__ __forceinline__ int ray_bin_levels(const float *__restrict__ rays_o, const float *__restrict__ rays_d, int64_t r,
                                              const float *__restrict__ boxes, int32_t n_grids, const int32_t *res3, float near,
                                              float inv_cells_max)
{
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    float cells = 0.f, inner = 0.f;     // inner: length of the ray inside the previous (finer) box
    for (int32_t l = 0; l < n_grids; ++l) {
        const float *box = boxes + 6 * l;
        float tmin = -INFINITY, tmax = INFINITY, dens = 0.f;
        bool hit = true;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float inv = 1.0f / d[ax];
            float lo = (box[ax] - o[ax]) * inv, hi = (box[3 + ax] - o[ax]) * inv;
            if (lo > hi) { const float t = lo; lo = hi; hi = t; }
            if (!(lo <= hi)) hit = false;
            tmin = fmaxf(tmin, lo); tmax = fminf(tmax, hi);
            dens += fabsf(d[ax]) * (float)res3[ax] / (box[3 + ax] - box[ax]);
        }
        tmin = fmaxf(tmin, near);
        const float len = (hit && tmin < tmax) ? tmax - tmin : 0.f;
        cells += fmaxf(len - inner, 0.f) * dens;
        inner = fmaxf(len, inner);
    }
    if (!(cells > 0.f)) return 0;
    const int b = 1 + (int)(cells * inv_cells_max * (float)(BIN_COUNT - 1));
    return b < 1 ? 1 : (b > BIN_COUNT - 1 ? BIN_COUNT - 1 : b);
}

struct BinLevels { const float *boxes; int32_t n_grids; int32_t res[3]; float near; float inv_cells_max; unsigned long long *stats; };

// Every workgroup owns one contiguous range of rays in both passes, so the global histogram / cursors see 256 atomics
// per workgroup (not per 256 rays), and inside a workgroup the counting is LDS atomics.
template <bool LEVELS>
__global__ __launch_bounds__(256) void bin_count_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                        int64_t n_rays, int64_t per_block, const float *__restrict__ box,
                                                        uint8_t *__restrict__ bins, int32_t *__restrict__ hist, const BinLevels lv)
{
    __shared__ int32_t h[BIN_COUNT];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r_lo = (int64_t)blockIdx.x * per_block, r_hi = min(r_lo + per_block, n_rays);
    uint32_t w_max = 0u, w_sum = 0u;   // coherence of the key: per 64 consecutive rays, the largest bin and the sum of the bins
    for (int64_t r0 = r_lo; r0 < r_hi; r0 += blockDim.x) {
        const int64_t r = r0 + threadIdx.x;
        int b = 0;
        if (r < r_hi) {
            b = LEVELS ? ray_bin_levels(rays_o, rays_d, r, lv.boxes, lv.n_grids, lv.res, lv.near, lv.inv_cells_max)
                       : ray_bin(rays_o, rays_d, r, box);
            bins[r] = (uint8_t)b;
            atomicAdd(&h[b], 1);
        }
        if (LEVELS && lv.stats) {
            int32_t m = b, t = b;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { m = max(m, __shfl_xor(m, off, 64)); t += __shfl_xor(t, off, 64); }
            w_max += (uint32_t)m; w_sum += (uint32_t)t;
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
    if (LEVELS && lv.stats && lane_id() == 0) {
        atomicAdd(&lv.stats[0], (unsigned long long)w_max);
        atomicAdd(&lv.stats[1], (unsigned long long)w_sum);
    }
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
                                                          float *__restrict__ t_mids, int64_t *__restrict__ ray_indices, int vec, float cone,
                                                          int64_t capacity)
{
    __shared__ uint32_t s_pos[EXP_WPB][EXP_RUNS_QMAX];
    __shared__ float s_t0[EXP_WPB][EXP_RUNS_QMAX];
    __shared__ __attribute__((aligned(16))) int32_t s_own[EXP_WPB][256];  // entry that starts at each output of the current chunk
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t *pos = s_pos[wave];
    float *t0s = s_t0[wave];
    int32_t *slot = s_own[wave];
    const int64_t n_batches = ceil_div64(n_rays, EXP_RPW);
    for (int64_t batch = (int64_t)blockIdx.x * EXP_WPB + wave; batch < n_batches; batch += (int64_t)gridDim.x * EXP_WPB) {
        const int64_t r0 = batch * EXP_RPW;
        const int64_t ray = r0 + lane;
        const int n_loc = (int)min((int64_t)EXP_RPW, n_rays - r0);
        const bool own_all = lane < n_loc;
        int32_t c_all = 0, c_real = 0;
        int64_t s = 0, n = 0;
        if (own_all) {
            const longlong2 row = packed_info[ray];
            s = row.x;
            n = row.y;
            c_real = MODE == EXP_RAY_INDICES ? (n > 0 ? 1 : 0) : run_cnts[ray];  // (ray indices: the ray is one "run")
            c_all = (c_real > max_runs) ? 1 : c_real;  // overflowed ray: one sentinel entry
        }
        int32_t incl_all = c_all;
#pragma unroll
        for (int off = 1; off < EXP_RPW; off <<= 1) {
            const int32_t u = __shfl_up(incl_all, off, 64);
            if (lane >= off) incl_all += u;
        }
        if (MODE == EXP_RAY_INDICES) {
            const int64_t W0b = __shfl(s, 0, 64);
            const int64_t W1b = min((int64_t)__shfl(s + n, n_loc - 1, 64), capacity);
            if (W1b - W0b >= ((int64_t)1 << 27)) {
                // the packed 27-bit positions below cannot address this window (rays of millions of samples): plain
                // cooperative fill, still coalesced
                for (int rl = 0; rl < n_loc; ++rl) {
                    const int64_t s_r = __shfl(s, rl, 64), n_r = __shfl(n, rl, 64);
                    for (int64_t i = lane; i < n_r; i += 64) ray_indices[s_r + i] = r0 + rl;
                }
                continue;
            }
        }
        // The batch's records are staged EXP_RUNS_QMAX at a time: the rays [a, b) whose records fit (with the default, the
        // worst case of 32 rays x 32 records, always one group: the smaller staging area that lifted expand_intervals from 14
        // to 24 waves per CU does nothing for this kernel, which is bound by its write stream).
        for (int a = 0; a < n_loc;) {
        const int32_t before = __shfl(incl_all - c_all, a, 64);
        const int b = a + __popcll(__ballot(lane >= a && lane < n_loc && incl_all - before <= EXP_RUNS_QMAX));
        const bool own = lane >= a && lane < b;
        const int32_t c = own ? c_all : 0;
        const int32_t incl = incl_all - before;   // (lanes a .. b-1)
        const int64_t W0 = __shfl(s, a, 64);
        const int last_lane = b - 1;
        // (capacity: the output arrays may have been allocated before the total was known to the host; nothing is written
        //  beyond them, the caller re-runs the expansion in that case)
        const int64_t W1 = min((int64_t)__shfl(s + n, last_lane, 64), capacity);
        const int32_t Q = __shfl(incl, last_lane, 64);
        // overflow flag per local ray as a wave-uniform mask
        const unsigned long long ovf_mask = __ballot(own && c_real > max_runs);
        a = b;
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
                        store_f4<NFA_NT_EXPAND>(t_mids + p0, ts4[0], ts4[1], ts4[2], ts4[3]);
                    } else if (MODE == EXP_STARTS_ENDS || MODE == EXP_CONE) {
                        store_f4<NFA_NT_EXPAND>(t_starts + p0, ts4[0], ts4[1], ts4[2], ts4[3]);
                        store_f4<NFA_NT_EXPAND>(t_ends + p0, te4[0], te4[1], te4[2], te4[3]);
                    }
                    store_l2<NFA_NT_EXPAND>(ray_indices + p0, ri4[0], ri4[1]);
                    store_l2<NFA_NT_EXPAND>(ray_indices + p0 + 2, ri4[2], ri4[3]);
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
        }  // groups of rays
    }
}

// ------------------------------------------------------------------------------------------
// Expansion of the INTERVAL stream of the API's traverse_grids (ref grid.cu:219-262): a chain of
// continuous samples has one edge more than samples.  Run j of a ray (k_start_j samples before it, L_j chain
// starts before it) owns the edges [k_start_j + L_j, ...): its leading edge t_first if it starts a chain,
// then the end t_first + m * inc of each of its samples.  is_right is false exactly at chain starts, is_left
// is false exactly before a chain start (or the end of the ray's edges).  Same batch / chunk structure as
// expand_runs_kernel; one lane per ray stages that ray's records (it needs the running count of chain starts).
#ifdef NFA_EXP_IV_WAVES
__attribute__((amdgpu_waves_per_eu(NFA_EXP_IV_WAVES, NFA_EXP_IV_WAVES)))
#endif
__global__ __launch_bounds__(64 * EXP_WPB) void expand_intervals_kernel(int64_t n_rays, float dt, const int32_t *__restrict__ run_cnts,
                                                               const unsigned long long *__restrict__ runs, int32_t max_runs,
                                                               const longlong2 *__restrict__ iv_packed_info,
                                                               float *__restrict__ vals, int64_t *__restrict__ ray_indices,
                                                               uint8_t *__restrict__ is_left, uint8_t *__restrict__ is_right, int vec)
{
    __shared__ uint32_t s_pos[EXP_WPB][EXP_IV_QMAX];
    __shared__ float s_t0[EXP_WPB][EXP_IV_QMAX];
    __shared__ uint8_t s_cont[EXP_WPB][EXP_IV_QMAX];
    __shared__ __attribute__((aligned(16))) int32_t s_own[EXP_WPB][256 + 4];  // + look-ahead slot for the chunk's last edge
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[EXP_WPB][128];   // the chunk's is_left / is_right bytes, 4 per lane
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t *pos = s_pos[wave];
    float *t0s = s_t0[wave];
    uint8_t *conts = s_cont[wave];
    int32_t *slot = s_own[wave];
    uint32_t *mbytes = s_mask[wave];
    const int64_t n_batches = ceil_div64(n_rays, EXP_RPW);
    for (int64_t batch = (int64_t)blockIdx.x * EXP_WPB + wave; batch < n_batches; batch += (int64_t)gridDim.x * EXP_WPB) {
        const int64_t r0 = batch * EXP_RPW;
        const int64_t ray = r0 + lane;
        const int n_loc = (int)min((int64_t)EXP_RPW, n_rays - r0);
        int32_t c_all = 0, c_real = 0;
        int64_t s = 0, n = 0;
        if (lane < n_loc) {
            c_real = run_cnts[ray];
            const longlong2 row = iv_packed_info[ray];
            s = row.x;
            n = row.y;
            c_all = (c_real > max_runs) ? 1 : c_real;  // overflowed ray: one sentinel entry
        }
        int32_t incl_all = c_all;
#pragma unroll
        for (int off = 1; off < EXP_RPW; off <<= 1) {
            const int32_t u = __shfl_up(incl_all, off, 64);
            if (lane >= off) incl_all += u;
        }
        // The batch's records are staged EXP_IV_QMAX at a time: the rays [a, b) whose records fit (a typical batch has a
        // hundred and is one group; 32 rays with 32 records each would be four).  The staging area decides how many waves a
        // CU holds: sized for the worst case (1024 entries, 10.8 KB per wave) the kernel ran at 2.4 TB/s, with 256 entries
        // (3.8 KB) at 3.3 TB/s.
        for (int a = 0; a < n_loc;) {
        const int32_t before = __shfl(incl_all - c_all, a, 64);
        const int b = a + __popcll(__ballot(lane >= a && lane < n_loc && incl_all - before <= EXP_IV_QMAX));
        const bool mine = lane >= a && lane < b;
        const int32_t c = mine ? c_all : 0;
        const int32_t incl = incl_all - before;   // (lanes a .. b-1)
        const int64_t W0 = __shfl(s, a, 64);
        const int64_t W1 = __shfl(s + n, b - 1, 64);
        const int32_t Q = __shfl(incl, b - 1, 64);
        a = b;
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
                // The two mask streams are 1 byte per edge: written 4 bytes per lane they were 256-byte store instructions and
                // the kernel ran at 2.0 TB/s (244 us for cfg 2's 34 M edges).  The lanes' mask bytes go through LDS and leave
                // 16 bytes per lane: lanes 0-15 store is_left, lanes 16-31 is_right, a 16-edge group each -- where all 16 edges
                // of the group belong to this batch (a group cut by the batch's window is written byte by byte).
                const bool all4 = valid[0] && valid[1] && valid[2] && valid[3];
                const unsigned long long full4 = __ballot(all4);
                const bool group_full = vec && ((full4 >> (4 * (lane >> 2))) & 0xFull) == 0xFull;
                mbytes[lane] = (uint32_t)l4[0] | ((uint32_t)l4[1] << 8) | ((uint32_t)l4[2] << 16) | ((uint32_t)l4[3] << 24);
                mbytes[64 + lane] = (uint32_t)r4[0] | ((uint32_t)r4[1] << 8) | ((uint32_t)r4[2] << 16) | ((uint32_t)r4[3] << 24);
                __builtin_amdgcn_wave_barrier();
                if (vec && lane < 32) {
                    const int g = lane & 15, which = lane >> 4;
                    if (((full4 >> (4 * g)) & 0xFull) == 0xFull) {
                        const uint4 mv = *reinterpret_cast<const uint4 *>(mbytes + 64 * which + 4 * g);
                        typedef unsigned int nfa_v4u __attribute__((ext_vector_type(4)));
                        nfa_v4u v = {mv.x, mv.y, mv.z, mv.w};
                        __builtin_nontemporal_store(v, reinterpret_cast<nfa_v4u *>((which ? is_right : is_left) + cb + 16 * g));
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (vec && all4) {
                    store_f4<NFA_NT_EXPAND>(vals + p0, v4[0], v4[1], v4[2], v4[3]);
                    store_l2<NFA_NT_EXPAND>(ray_indices + p0, ri4[0], ri4[1]);
                    store_l2<NFA_NT_EXPAND>(ray_indices + p0 + 2, ri4[2], ri4[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (valid[k]) { vals[p0 + k] = v4[k]; ray_indices[p0 + k] = ri4[k]; }
                }
                if (!group_full) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (valid[k]) { is_left[p0 + k] = l4[k]; is_right[p0 + k] = r4[k]; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        }  // groups of rays
    }
}

// ------------------------------------------------------------------------------------------
// The test-mode loop's bookkeeping between two iterations (ref examples/utils.py:409-414:
// ray_mask = (opacity <= 1 - early_stop_eps) & (samples taken == n_samples)) as ONE launch: the mask, the list of the alive
// rays (what the next traversal walks) and their number.  The torch composition was ten launches (<=, ==, &, sum, the four of
// nonzero_static, a cast, a copy) -- on a small scene more than the iteration's kernels.  A workgroup owns ALIVE_PER_WG
// consecutive rays and appends its alive ones, in ascending order, at an offset taken from one atomic counter: the list is
// ascending inside a workgroup's stretch and the stretches come in the order the workgroups finish (the walk's results do not
// depend on the list's order).
constexpr int ALIVE_THREADS = 1024, ALIVE_PER_THREAD = 8, ALIVE_PER_WG = ALIVE_THREADS * ALIVE_PER_THREAD;
// state (optional): the device-side schedule of the test-mode loop (include/nerfacc_hip.h: nfa_testmode_begin); n_samples is
// then state[0], and nothing happens when that is 0 (*count stays 0: the host zeroed it)
__global__ __launch_bounds__(ALIVE_THREADS) void alive_rays_kernel(const float *__restrict__ opacity, const longlong2 *__restrict__ packed_info,
                                                                  int64_t n_samples_arg, float thre, int64_t n_rays, uint8_t *__restrict__ mask,
                                                                  int32_t *__restrict__ alive, unsigned long long *__restrict__ count,
                                                                  int32_t *__restrict__ state, int32_t count_samples)
{
    __shared__ int32_t s_wave[ALIVE_THREADS / 64];
    __shared__ unsigned long long s_base;
    const int64_t n_samples = state ? (int64_t)state[0] : n_samples_arg;
    if (state) {
        if (n_samples == 0) return;
        if (count_samples && blockIdx.x == 0 && threadIdx.x == 0 && n_rays > 0) {
            const longlong2 last = packed_info[n_rays - 1];
            *reinterpret_cast<long long *>(state + 4) += last.x + last.y;
        }
    }
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * ALIVE_PER_WG + (int64_t)threadIdx.x * ALIVE_PER_THREAD;
    uint32_t bits = 0u;
#pragma unroll
    for (int k = 0; k < ALIVE_PER_THREAD; ++k) {
        const int64_t r = r0 + k;
        if (r < n_rays) {
            const bool a = (opacity[r] <= thre) && (packed_info[r].y == n_samples);
            mask[r] = a ? 1 : 0;
            bits |= (a ? 1u : 0u) << k;
        }
    }
    const int32_t c = __builtin_popcount(bits);
    int32_t incl = c;   // inclusive scan over the wave, then over the workgroup's waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < ALIVE_THREADS / 64; ++w) {
        const int32_t x = s_wave[w];
        before += w < wave ? x : 0;
        total += x;
    }
    if (threadIdx.x == 0) s_base = total > 0 ? atomicAdd(count, (unsigned long long)total) : 0ull;
    __syncthreads();
    int64_t o = (int64_t)s_base + before + incl - c;
#pragma unroll
    for (int k = 0; k < ALIVE_PER_THREAD; ++k)
        if ((bits >> k) & 1u) alive[o++] = (int32_t)(r0 + k);
}

// nfa_testmode_begin: the schedule of one iteration (thread 0) and the zeroing of what the iteration accumulates into
__global__ __launch_bounds__(256) void testmode_begin_kernel(int64_t *__restrict__ alive_count, int32_t *__restrict__ state, int64_t n_rays,
                                                            int32_t min_samples, int32_t max_samples, int64_t *__restrict__ sm_cnts,
                                                            int32_t *__restrict__ run_cnts, int64_t *__restrict__ zero_words, int32_t n_zero)
{
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)blockDim.x * gridDim.x;
    for (int64_t i = i0; i < n_rays; i += stride) { sm_cnts[i] = 0; run_cnts[i] = 0; }
    for (int64_t i = i0; i < n_zero; i += stride) zero_words[i] = 0;
    if (i0 == 0) {   // examples/utils.py:330-340
        const int64_t n_alive = alive_count[0];
        alive_count[1] = n_alive;    // what this iteration's walk lists (n_listed_dev)
        alive_count[0] = 0;          // nfa_testmode_alive counts into it
        int32_t n = 0;
        if (n_alive > 0 && state[1] < max_samples) {
            const int64_t q = n_rays / n_alive;
            n = (int32_t)(q < 64 ? q : 64);
            n = n > min_samples ? n : min_samples;
            state[1] += n;
            state[2] += 1;
        }
        state[0] = n;
    }
}

}  // namespace nfa

using namespace nfa;

extern "C" {

int nfa_testmode_begin(int64_t *alive_count, int32_t *state, int64_t n_rays, int32_t min_samples, int32_t max_samples,
                       int64_t *sm_cnts, int32_t *run_cnts, int64_t *zero_words, int32_t n_zero, nfa_stream_t stream)
{
    NFA_REQUIRE(alive_count && state && n_rays >= 0 && min_samples >= 1 && max_samples >= 1 && (n_rays == 0 || (sm_cnts && run_cnts)) &&
                n_zero >= 0 && (n_zero == 0 || zero_words), "testmode_begin: bad arguments");
    hipLaunchKernelGGL(testmode_begin_kernel, dim3(grid_1d(n_rays > 0 ? n_rays : 1, 256, 2048)), dim3(256), 0, as_stream(stream), alive_count,
                       state, n_rays, min_samples, max_samples, sm_cnts, run_cnts, zero_words, n_zero);
    NFA_CHECK_LAUNCH("testmode_begin");
    return NFA_OK;
}

int nfa_testmode_alive(const float *opacity, const int64_t *packed_info, int32_t *state, float opacity_max, int64_t n_rays,
                       uint8_t *mask, int32_t *alive, int64_t *count, int32_t count_samples, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31) && count && state, "testmode_alive: bad arguments");
    hipStream_t s = as_stream(stream);
    if (n_rays == 0) return NFA_OK;      // (*count was zeroed by nfa_testmode_begin: no memset node in the iteration's graph)
    NFA_REQUIRE(opacity && packed_info && mask && alive, "testmode_alive: null pointer");
    const unsigned grid = (unsigned)((n_rays + ALIVE_PER_WG - 1) / ALIVE_PER_WG);
    hipLaunchKernelGGL(alive_rays_kernel, dim3(grid), dim3(ALIVE_THREADS), 0, s, opacity, reinterpret_cast<const longlong2 *>(packed_info),
                       (int64_t)0, opacity_max, n_rays, mask, alive, reinterpret_cast<unsigned long long *>(count), state, count_samples);
    NFA_CHECK_LAUNCH("testmode_alive");
    return NFA_OK;
}

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
    hipLaunchKernelGGL(bin_count_kernel<false>, dim3(grid), dim3(256), 0, s, rays_o, rays_d, n_rays, per_block, box, bins, hist, BinLevels{});
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(256), 0, s, hist);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(grid), dim3(256), 0, s, bins, n_rays, per_block, hist, order);
    NFA_CHECK_LAUNCH("bin_rays");
    return NFA_OK;
}

int nfa_alive_rays(const float *opacity, const int64_t *packed_info, int64_t n_samples, float opacity_max, int64_t n_rays,
                   uint8_t *mask, int32_t *alive, int64_t *count, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31) && count, "alive_rays: bad arguments");
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(count, 0, sizeof(int64_t), s) != hipSuccess) { set_error("alive_rays: memset failed"); return NFA_EHIP; }
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(opacity && packed_info && mask && alive, "alive_rays: null pointer");
    const unsigned grid = (unsigned)((n_rays + ALIVE_PER_WG - 1) / ALIVE_PER_WG);
    hipLaunchKernelGGL(alive_rays_kernel, dim3(grid), dim3(ALIVE_THREADS), 0, s, opacity, reinterpret_cast<const longlong2 *>(packed_info),
                       n_samples, opacity_max, n_rays, mask, alive, reinterpret_cast<unsigned long long *>(count), nullptr, 0);
    NFA_CHECK_LAUNCH("alive_rays");
    return NFA_OK;
}

int nfa_bin_rays_levels(const float *rays_o, const float *rays_d, int64_t n_rays, const float *aabbs, int32_t n_grids,
                        const int32_t *res, float near_plane, int32_t *order, void *scratch, uint64_t *coherence, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_rays < ((int64_t)1 << 31), "bin_rays_levels: n_rays out of range");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(rays_o && rays_d && aabbs && res && order && scratch && n_grids >= 1, "bin_rays_levels: bad arguments");
    hipStream_t s = as_stream(stream);
    int32_t *hist = reinterpret_cast<int32_t *>(scratch);                      // [BIN_COUNT]
    uint8_t *bins = reinterpret_cast<uint8_t *>(hist + BIN_COUNT);             // [n_rays]
    if (hipMemsetAsync(hist, 0, sizeof(int32_t) * BIN_COUNT, s) != hipSuccess) { set_error("bin_rays_levels: memset failed"); return NFA_EHIP; }
    int64_t per_block = ceil_div64(n_rays, 512);
    if (per_block < 1024) per_block = 1024;
    const unsigned grid = (unsigned)ceil_div64(n_rays, per_block);
    BinLevels lv;
    lv.boxes = aabbs; lv.n_grids = n_grids; lv.res[0] = res[0]; lv.res[1] = res[1]; lv.res[2] = res[2];
    lv.near = near_plane;
    lv.stats = reinterpret_cast<unsigned long long *>(coherence);
    lv.inv_cells_max = 1.0f / ((float)n_grids * (float)(res[0] + res[1] + res[2]));   // a ray crosses at most that many boundaries
    hipLaunchKernelGGL(bin_count_kernel<true>, dim3(grid), dim3(256), 0, s, rays_o, rays_d, n_rays, per_block, nullptr, bins, hist, lv);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(256), 0, s, hist);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(grid), dim3(256), 0, s, bins, n_rays, per_block, hist, order);
    NFA_CHECK_LAUNCH("bin_rays_levels");
    return NFA_OK;
}

int nfa_expand_runs(int64_t n_rays, float step_size, const int32_t *run_cnts, const uint64_t *runs, int32_t max_runs,
                    const int64_t *packed_info, float *t_starts, float *t_ends, float *t_mids,
                    int64_t *ray_indices, int64_t capacity, nfa_stream_t stream)
{
    NFA_REQUIRE(capacity >= 0, "expand_runs: negative capacity");
    NFA_REQUIRE(n_rays >= 0, "expand_runs: negative n_rays");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(run_cnts && runs && packed_info && ray_indices && (t_mids || (t_starts && t_ends)),
                "expand_runs: null pointer");
    NFA_REQUIRE(max_runs >= 1 && max_runs <= EXP_RUNS_QMAX && max_runs * EXP_RPW <= EXP_QMAX, "expand_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f, "expand_runs: step_size must be > 0");
    const int vec = ((reinterpret_cast<uintptr_t>(t_starts) | reinterpret_cast<uintptr_t>(t_ends) |
                      reinterpret_cast<uintptr_t>(t_mids) | reinterpret_cast<uintptr_t>(ray_indices)) & 15) == 0;
    const unsigned grid = grid_1d(ceil_div64(n_rays, EXP_RPW) * 64, 64 * EXP_WPB, 1 << 22);
    if (t_mids)
        hipLaunchKernelGGL(expand_runs_kernel<EXP_MIDS>, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                           reinterpret_cast<const unsigned long long *>(runs), max_runs,
                           reinterpret_cast<const longlong2 *>(packed_info), t_starts, t_ends, t_mids, ray_indices, vec, 0.0f, capacity);
    else
        hipLaunchKernelGGL(expand_runs_kernel<EXP_STARTS_ENDS>, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                           reinterpret_cast<const unsigned long long *>(runs), max_runs,
                           reinterpret_cast<const longlong2 *>(packed_info), t_starts, t_ends, t_mids, ray_indices, vec, 0.0f, capacity);
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
    NFA_REQUIRE(max_runs >= 1 && max_runs <= EXP_RUNS_QMAX && max_runs * EXP_RPW <= EXP_QMAX, "expand_cone_runs: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f && cone_angle > 0.0f, "expand_cone_runs: step_size and cone_angle must be > 0");
    const int vec = ((reinterpret_cast<uintptr_t>(t_starts) | reinterpret_cast<uintptr_t>(t_ends) |
                      reinterpret_cast<uintptr_t>(ray_indices)) & 15) == 0;
    const unsigned grid = grid_1d(ceil_div64(n_rays, EXP_RPW) * 64, 64 * EXP_WPB, 1 << 22);
    hipLaunchKernelGGL(expand_runs_kernel<EXP_CONE>, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                       reinterpret_cast<const unsigned long long *>(runs), max_runs,
                       reinterpret_cast<const longlong2 *>(packed_info), t_starts, t_ends, nullptr, ray_indices, vec, cone_angle,
                       (int64_t)1 << 62);
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
                       ray_indices, vec, 0.0f, (int64_t)1 << 62);
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
    NFA_REQUIRE(max_runs >= 1 && max_runs <= EXP_IV_QMAX && max_runs * EXP_RPW <= EXP_QMAX, "expand_intervals: max_runs must be in [1, 32]");
    NFA_REQUIRE(step_size > 0.0f, "expand_intervals: step_size must be > 0");
    const int vec = ((reinterpret_cast<uintptr_t>(vals) | reinterpret_cast<uintptr_t>(ray_indices) | reinterpret_cast<uintptr_t>(is_left) |
                      reinterpret_cast<uintptr_t>(is_right)) & 15) == 0;
    const unsigned grid = grid_1d(ceil_div64(n_rays, EXP_RPW) * 64, 64 * EXP_WPB, 1 << 22);
    hipLaunchKernelGGL(expand_intervals_kernel, dim3(grid), dim3(64 * EXP_WPB), 0, as_stream(stream), n_rays, step_size, run_cnts,
                       reinterpret_cast<const unsigned long long *>(runs), max_runs,
                       reinterpret_cast<const longlong2 *>(iv_packed_info), vals, ray_indices, is_left, is_right, vec);
    NFA_CHECK_LAUNCH("expand_intervals");
    return NFA_OK;
}

}  // extern "C"
