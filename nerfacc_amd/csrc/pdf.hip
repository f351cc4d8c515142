// pdf.hip -- inverse-CDF importance sampling and per-ray searchsorted.
//
// Semantics: /root/reference/nerfacc/cuda/csrc/pdf.cu:98-167 (importance_sampling_kernel),
// :169-241 (compute_intervels_kernel), :245-286 (searchsorted_kernel), :43-63 (upper_bound).
// Structure (ours): the two reference kernels are fused; a power-of-two lane group owns one ray,
// each lane inverts the CDF for one sample, neighbouring samples are exchanged with wave
// shuffles to form the S+1 interval edges, and a wave writes whole [rays_per_wave, S+1] rows
// (coalesced).  The per-ray jitter is one Philox4x32-10 draw per RAY (the reference initialises
// a curand state per sample for the same value, pdf.cu:139-144).
#include "common.hip.h"

namespace nfa {

__device__ __forceinline__ int64_t upper_bound_f(const float *__restrict__ data, int64_t start, int64_t end, float val)
{
    while (start < end) {
        const int64_t mid = start + ((end - start) >> 1);
        if (!(data[mid] > val)) start = mid + 1;
        else end = mid;
    }
    return start;
}
__device__ __forceinline__ int64_t clamp64(int64_t v, int64_t lo, int64_t hi)
{
    const int64_t m = v < hi ? v : hi;
    return m > lo ? m : lo;
}

// Philox4x32-10 (Salmon et al. 2011), counter layout of curand_init(seed, subsequence, offset).
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t subsequence, uint64_t offset)
{
    const uint64_t blk = offset >> 2;
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = (uint32_t)subsequence, c3 = (uint32_t)(subsequence >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t out[4] = {c0, c1, c2, c3};
    // curand_uniform: x * 2^-32 + 2^-33  (in (0, 1])
    return (float)out[offset & 3] * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

__device__ __forceinline__ int upper_bound_lds(const float *data, int start, int end, float val)
{
    while (start < end) {
        const int mid = start + ((end - start) >> 1);
        if (!(data[mid] > val)) start = mid + 1;
        else end = mid;
    }
    return start;
}

__device__ __forceinline__ int64_t uniform64_pdf(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

constexpr int IS_STAGE_MAX = 512;  // CDF entries a wave may stage in LDS (x 2 arrays x 4 waves = 16 KiB per workgroup)

// L lanes per ray (power of two <= 64); 64 / L rays per wave.
// STAGED (batched input whose rays fit): the wave's rays are consecutive rows of one contiguous block, which is
// copied to LDS with coalesced loads; the S binary searches of a ray then run on LDS (7 dependent ~64-cycle reads
// instead of 7 dependent global loads per sample).
template <bool STAGED>
__global__ __launch_bounds__(256) void importance_sampling_kernel(
    const float *__restrict__ in_vals, const float *__restrict__ cdfs, const int64_t *__restrict__ in_packed,
    int64_t n_rays, int64_t n_edges_per_ray, int64_t S, int L, int RB, int stratified, uint64_t seed, uint64_t offset,
    float *__restrict__ out_iv, float *__restrict__ out_sm, int transform, float t_a, float t_b,
    float *__restrict__ out_ts, float *__restrict__ out_te)
{
    // optional s -> t mapping of the S+1 edges (ref: estimators/prop_net.py:215-229), written as contiguous
    // t_starts / t_ends rows: uniform t = s*t_b + (1-s)*t_a (t_a = t_min, t_b = t_max);
    // lindisp t = 1 / (s*t_b + (1-s)*t_a) (t_a = 1/t_min, t_b = 1/t_max)
    auto emit_edge = [&](int64_t ray, int64_t k, float e) {
        out_iv[ray * (S + 1) + k] = e;
        if (transform) {
            const float lin = e * t_b + (1.0f - e) * t_a;
            const float t = transform == 2 ? 1.0f / lin : lin;
            if (k < S) out_ts[ray * S + k] = t;
            if (k > 0) out_te[ray * S + k - 1] = t;
        }
    };
    __shared__ float s_cdf[4][STAGED ? IS_STAGE_MAX : 1];
    __shared__ float s_val[4][STAGED ? IS_STAGE_MAX : 1];
    const int lane = lane_id();
    const int gl = lane & (L - 1);            // lane within the ray's group
    const int rays_per_wave = 64 / L;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float *lc = s_cdf[threadIdx.x >> 6], *lv = s_val[threadIdx.x >> 6];
    // A wave owns blocks of RB consecutive rays (RB a multiple of 64 / L, at most 64).  The jitter is one Philox draw per RAY
    // (ten rounds of four 32-bit multiplies, quarter rate): lane i draws it for ray i of the block once, the groups fetch
    // it by shuffle -- with one ray per wave (S = 64) every lane would repeat the same 100+ instructions for every ray.
    // STAGED: the rows of the NEXT group of rays are requested (into registers) before the current group is searched, and
    // written to LDS when their turn comes: without that every group is one exposed memory latency (load -> LDS -> search
    // -> store, ~3 us) and the kernel runs at the rate occupancy x group / latency.
    constexpr int SLOTS = STAGED ? IS_STAGE_MAX / 64 : 1;
    float pc[SLOTS], pv[SLOTS];
    auto rows_here = [&](int64_t r0) { return (int)(min((int64_t)rays_per_wave, n_rays - r0) * n_edges_per_ray); };
    auto prefetch = [&](int64_t r0) {
        const int64_t blk = r0 * n_edges_per_ray;
        const int n_h = rows_here(r0);
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (64 * k < n_h) {  // wave-uniform
                const int f = lane + 64 * k;
                pc[k] = cdfs[blk + (f < n_h ? f : 0)];
                pv[k] = in_vals[blk + (f < n_h ? f : 0)];
            }
    };
    int64_t rb = wave * RB;
    int sub = 0;
    float lane_bias = 0.5f;
    if (STAGED && rb < n_rays) prefetch(rb);
    while (rb < n_rays) {
        if (sub == 0) {
            lane_bias = 0.5f;
            if (stratified && lane < RB && rb + lane < n_rays) lane_bias = philox_uniform(seed, (uint64_t)(rb + lane), offset);
        }
        const int64_t r0 = rb + sub;
        int64_t next_rb = rb;
        int next_sub = sub + rays_per_wave;
        if (next_sub >= RB || rb + next_sub >= n_rays) { next_sub = 0; next_rb = rb + n_waves * RB; }
        const int cur_sub = sub;
        rb = next_rb; sub = next_sub;
        const int64_t ray = r0 + lane / L;
        const bool ray_ok = ray < n_rays;
        int64_t base = 0, last = 0;
        if (ray_ok) {
            if (in_packed) { base = in_packed[2 * ray]; last = base + in_packed[2 * ray + 1] - 1; }
            else { base = ray * n_edges_per_ray; last = base + n_edges_per_ray - 1; }
        }
        int lbase = 0;  // the ray's first entry in LDS
        if (STAGED) {
            const int n_h = rows_here(r0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < SLOTS; ++k)
                if (64 * k < n_h) {
                    const int f = lane + 64 * k;
                    if (f < n_h) { lc[f] = pc[k]; lv[f] = pv[k]; }
                }
            __builtin_amdgcn_wave_barrier();
            if (next_rb < n_rays) prefetch(next_rb + next_sub);
            lbase = (lane / L) * (int)n_edges_per_ray;
        }
        const int llast = lbase + (int)n_edges_per_ray - 1;
        float u_floor = 0.f, u_step = 0.f, t_min = 0.f, t_max = 0.f;
        const float bias = __shfl(lane_bias, cur_sub + lane / L, 64);
        if (ray_ok) {
            u_floor = STAGED ? lc[lbase] : cdfs[base];
            const float u_ceil = STAGED ? lc[llast] : cdfs[last];
            u_step = (u_ceil - u_floor) / S;
            t_min = STAGED ? lv[lbase] : in_vals[base];
            t_max = STAGED ? lv[llast] : in_vals[last];
        }
        float t_carry = 0.f;  // last sample of the previous block of L samples
        for (int64_t s0 = 0; s0 < S; s0 += L) {
            const int64_t sid = s0 + gl;
            const bool ok = ray_ok && sid < S;
            float t = 0.f;
            if (ok) {  // pdf.cu:133-166
                const float u = u_floor + (sid + bias) * u_step;
                float u_lower, u_upper, t_lower, t_upper;
                if (STAGED) {
                    const int p = upper_bound_lds(lc, lbase, llast, u);
                    const int p0 = min(max(p - 1, lbase), llast), p1 = min(max(p, lbase), llast);
                    u_lower = lc[p0]; u_upper = lc[p1]; t_lower = lv[p0]; t_upper = lv[p1];
                } else {
                    const int64_t p = upper_bound_f(cdfs, base, last, u);
                    const int64_t p0 = clamp64(p - 1, base, last), p1 = clamp64(p, base, last);
                    u_lower = cdfs[p0]; u_upper = cdfs[p1]; t_lower = in_vals[p0]; t_upper = in_vals[p1];
                }
                if (u_upper - u_lower < 1e-10f) t = (t_lower + t_upper) * 0.5f;
                else {
                    const float scaling = (t_upper - t_lower) / (u_upper - u_lower);
                    t = (u - u_lower) * scaling + t_lower;
                }
                if (out_sm) out_sm[ray * S + sid] = t;
            }
            // neighbours (pdf.cu:209-239)
            float t_prev = __shfl_up(t, 1, L);
            if (gl == 0) t_prev = t_carry;
            const float t_next = __shfl_down(t, 1, L);
            if (ok) {
                if (S == 1) {  // one sample: its interval is the ray's whole range (the reference reads t_1 out of bounds, pdf.cu:211)
                    emit_edge(ray, 0, t_min);
                    emit_edge(ray, 1, t_max);
                } else if (sid == 0) {
                    const float half_width = (t_next - t) * 0.5f;  // S >= 2 and L >= 2 guarantee lane 1 holds t_1
                    emit_edge(ray, 0, fmaxf(t - half_width, t_min));
                } else {
                    emit_edge(ray, sid, (t + t_prev) * 0.5f);
                    if (sid == S - 1) {
                        const float half_width = (t - t_prev) * 0.5f;
                        emit_edge(ray, sid + 1, fminf(t + half_width, t_max));
                    }
                }
            }
            t_carry = __shfl(t, L - 1, L);
        }
    }
}

// The batched case that matters (PropNetEstimator: rows of <= 64 samples whose CDF rows fit the LDS stage), written
// without divergent control flow: the general kernel above spends ~300 instructions per group of rays on exec-mask
// branches (the search loop, four kinds of edges, 64-bit index arithmetic) and is issue-bound at 2.4 TB/s.  Same
// arithmetic, same results:
//   * the upper bound is the same bisection run for a wave-uniform number of rounds with selects (a finished lane idles);
//   * every lane forms ITS edge k = sample id with selects (first edge / middle edge), the ray's last lane also the edge S;
//   * the s -> t mapping is applied to the lane's edge and to the next one (a shuffle), so t_starts / t_ends are written as
//     two plain coalesced rows;
//   * rows are addressed as a scalar 64-bit base per group plus 32-bit lane offsets.
template <int LL, int NB /* blocks of LL samples per ray: sample i * LL + lane-in-group, so that S <= LL * NB */>
__global__ __launch_bounds__(256) void importance_sampling_rows_kernel(
    const float *__restrict__ in_vals, const float *__restrict__ cdfs, int64_t n_rays, int n_edges, int S, int n_rounds, int RB,
    int stratified, uint64_t seed, uint64_t offset, float *__restrict__ out_iv, float *__restrict__ out_sm, int transform,
    float t_a, float t_b, float *__restrict__ out_ts, float *__restrict__ out_te)
{
    constexpr int RPW = 64 / LL;  // rays per group
    constexpr int SLOTS = IS_STAGE_MAX / 64;
    __shared__ float s_cdf[4][IS_STAGE_MAX];
    __shared__ float s_val[4][IS_STAGE_MAX];
    const int lane = lane_id();
    const int gl = lane & (LL - 1), grp = lane / LL;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float *lc = s_cdf[threadIdx.x >> 6], *lv = s_val[threadIdx.x >> 6];
    float pc[SLOTS], pv[SLOTS];
    auto rows_here = [&](int64_t r0) { return (int)min((int64_t)RPW, n_rays - r0) * n_edges; };
    auto prefetch = [&](int64_t r0) {
        const float *c = cdfs + r0 * n_edges, *v = in_vals + r0 * n_edges;
        const int n_h = rows_here(r0);
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (64 * k < n_h) {  // wave-uniform
                const int f = lane + 64 * k;
                pc[k] = c[f < n_h ? f : 0];
                pv[k] = v[f < n_h ? f : 0];
            }
    };
    int64_t rb = uniform64_pdf(wave * RB);
    int sub = 0;
    float lane_bias = 0.5f;
    if (rb < n_rays) prefetch(rb);
    while (rb < n_rays) {
        if (sub == 0) {
            lane_bias = 0.5f;
            if (stratified && lane < RB && rb + lane < n_rays) lane_bias = philox_uniform(seed, (uint64_t)(rb + lane), offset);
        }
        const int64_t r0 = rb + sub;
        int64_t next_rb = rb;
        int next_sub = sub + RPW;
        if (next_sub >= RB || rb + next_sub >= n_rays) { next_sub = 0; next_rb = rb + n_waves * RB; }
        const float bias = __shfl(lane_bias, sub + grp, 64);
        rb = next_rb; sub = next_sub;
        {
            const int n_h = rows_here(r0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < SLOTS; ++k)
                if (64 * k < n_h) {
                    const int f = lane + 64 * k;
                    if (f < n_h) { lc[f] = pc[k]; lv[f] = pv[k]; }
                }
            __builtin_amdgcn_wave_barrier();
            if (next_rb < n_rays) prefetch(next_rb + next_sub);
        }
        const bool ray_ok = r0 + grp < n_rays;
        const int lbase = ray_ok ? grp * n_edges : 0, llast = lbase + n_edges - 1;
        const float u_floor = lc[lbase], u_ceil = lc[llast], t_min = lv[lbase], t_max = lv[llast];
        const float u_step = (u_ceil - u_floor) / (float)S;
        // the NB samples of the lane: their bisections run together (upper_bound over [lbase, llast), pdf.cu:43-63, 133-137)
        float u[NB];
        int start[NB], end[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            u[i] = u_floor + ((float)(i * LL + gl) + bias) * u_step;
            start[i] = lbase; end[i] = llast;
        }
        for (int it = 0; it < n_rounds; ++it) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bool act = start[i] < end[i];
                const int mid = start[i] + ((end[i] - start[i]) >> 1);
                const bool right = !(lc[act ? mid : lbase] > u[i]);
                start[i] = (act && right) ? mid + 1 : start[i];
                end[i] = (act && !right) ? mid : end[i];
            }
        }
        float t[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int p0 = min(max(start[i] - 1, lbase), llast), p1 = min(max(start[i], lbase), llast);
            const float u_lower = lc[p0], u_upper = lc[p1], t_lower = lv[p0], t_upper = lv[p1];
            const float du = u_upper - u_lower;
            t[i] = du < 1e-10f ? (t_lower + t_upper) * 0.5f : (u[i] - u_lower) * ((t_upper - t_lower) / du) + t_lower;
        }
        // edges (pdf.cu:205-239): the lane forms edge k = sample id of each of its samples, the ray's last sample also edge S
        float *iv = out_iv + r0 * (S + 1) + grp * (S + 1);
        float e[NB], te0[NB];
        float e_last = 0.0f;
        auto map = [&](float x) { const float lin = x * t_b + (1.0f - x) * t_a; return transform == 2 ? 1.0f / lin : lin; };
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int sid = i * LL + gl;
            float t_prev = __shfl_up(t[i], 1, LL);
            const float t_carry = __shfl(t[i > 0 ? i - 1 : 0], LL - 1, LL);   // (every lane takes part in a shuffle)
            if (i > 0 && gl == 0) t_prev = t_carry;
            const float t_next = __shfl_down(t[i], 1, LL);   // only the first sample of a ray looks at it (S >= 2: lane 1 holds t_1)
            e[i] = sid == 0 ? fmaxf(t[i] - (t_next - t[i]) * 0.5f, t_min) : (t[i] + t_prev) * 0.5f;
            if (S == 1) e[i] = t_min;  // one sample: its interval is the ray's whole range
            const bool ok = ray_ok && sid < S;
            if (sid == S - 1) e_last = S == 1 ? t_max : fminf(t[i] + (t[i] - t_prev) * 0.5f, t_max);
            if (out_sm && ok) (out_sm + r0 * S)[grp * S + sid] = t[i];
            if (ok) iv[sid] = e[i];
            if (ok && sid == S - 1) iv[S] = e_last;
            te0[i] = transform ? map(e[i]) : 0.0f;
        }
        if (transform) {
            const float te_last = map(e_last);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int sid = i * LL + gl;
                float te1 = __shfl_down(te0[i], 1, LL);
                const float te_first = __shfl(te0[i + 1 < NB ? i + 1 : i], 0, LL);
                if (i + 1 < NB && gl == LL - 1) te1 = te_first;
                if (sid == S - 1) te1 = te_last;
                if (ray_ok && sid < S) { (out_ts + r0 * S)[grp * S + sid] = te0[i]; (out_te + r0 * S)[grp * S + sid] = te1; }
            }
        }
    }
}

// Per-ray sample counts (the reference's Tensor overload, pdf.cu:294-355, which allocates zero samples upstream --
// memalloc_data(false, false) at :324 -- and therefore never worked): ray r is resampled into cnt[r] samples and
// cnt[r] + 1 edges (0 edges when cnt[r] == 0), written PACKED: sample j of ray r at sm_starts[r] + j, edge j at
// sm_starts[r] + r' + j with r' = number of non-empty rays before r (= iv_starts[r]).  Same arithmetic per sample as the
// batched kernel with n = cnt[r]; one 16-lane group per ray, neighbours by shuffle.  is_left / is_right as
// compute_intervels_kernel (:205-238) sets them: every edge but a ray's last is a left edge, every edge but its first a
// right edge.  cnt[r] == 1: the single interval is the ray's whole range.
constexpr int ISP_L = 16;
__global__ __launch_bounds__(256) void importance_sampling_packed_kernel(
    const float *__restrict__ in_vals, const float *__restrict__ cdfs, const int64_t *__restrict__ in_packed,
    int64_t n_rays, int64_t n_edges_per_ray, const int64_t *__restrict__ sm_packed /*[n_rays,2]*/,
    const int64_t *__restrict__ iv_packed /*[n_rays,2]*/, int stratified, uint64_t seed, uint64_t offset,
    float *__restrict__ sm_vals, int64_t *__restrict__ sm_ray_indices, float *__restrict__ iv_vals,
    int64_t *__restrict__ iv_ray_indices, uint8_t *__restrict__ iv_left, uint8_t *__restrict__ iv_right)
{
    const int lane = lane_id(), gl = lane & (ISP_L - 1);
    const int64_t group = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * (64 / ISP_L) + lane / ISP_L;
    const int64_t n_groups = (((int64_t)gridDim.x * blockDim.x) >> 6) * (64 / ISP_L);
    const int64_t n_iter = ceil_div64(n_rays, n_groups);
    for (int64_t it = 0; it < n_iter; ++it) {   // (uniform trip count: the shuffles below need every lane of the wave)
        const int64_t ray = group + it * n_groups;
        const bool ray_ok = ray < n_rays;
        int64_t base = 0, last = 0, S = 0, o_sm = 0, o_iv = 0;
        if (ray_ok) {
            if (in_packed) { base = in_packed[2 * ray]; last = base + in_packed[2 * ray + 1] - 1; }
            else { base = ray * n_edges_per_ray; last = base + n_edges_per_ray - 1; }
            o_sm = sm_packed[2 * ray]; S = sm_packed[2 * ray + 1];
            o_iv = iv_packed[2 * ray];
        }
        const bool has = ray_ok && S > 0 && last >= base;
        float u_floor = 0.f, u_step = 0.f, bias = 0.5f, t_min = 0.f, t_max = 0.f;
        if (has) {
            u_floor = cdfs[base];
            u_step = (cdfs[last] - u_floor) / S;
            if (stratified) bias = philox_uniform(seed, (uint64_t)ray, offset);
            t_min = in_vals[base]; t_max = in_vals[last];
        }
        // the groups of a wave loop together over the longest of their rays
        int64_t S_max = has ? S : 0;
#pragma unroll
        for (int off = ISP_L; off < 64; off <<= 1) S_max = max(S_max, (int64_t)__shfl_xor(S_max, off, 64));
        float t_carry = 0.f;
        for (int64_t s0 = 0; s0 < S_max; s0 += ISP_L) {
            const int64_t sid = s0 + gl;
            const bool ok = has && sid < S;
            float t = 0.f;
            if (ok) {  // pdf.cu:133-166
                const float u = u_floor + (sid + bias) * u_step;
                const int64_t p = upper_bound_f(cdfs, base, last, u);
                const int64_t p0 = clamp64(p - 1, base, last), p1 = clamp64(p, base, last);
                const float u_lower = cdfs[p0], u_upper = cdfs[p1], t_lower = in_vals[p0], t_upper = in_vals[p1];
                if (u_upper - u_lower < 1e-10f) t = (t_lower + t_upper) * 0.5f;
                else {
                    const float scaling = (t_upper - t_lower) / (u_upper - u_lower);
                    t = (u - u_lower) * scaling + t_lower;
                }
                sm_vals[o_sm + sid] = t;
                sm_ray_indices[o_sm + sid] = ray;
            }
            float t_prev = __shfl_up(t, 1, ISP_L);
            if (gl == 0) t_prev = t_carry;
            const float t_next = __shfl_down(t, 1, ISP_L);   // (sid 0 with S >= 2: lane 1 of the same block holds t_1)
            if (ok) {
                auto edge = [&](int64_t j, float e) {
                    iv_vals[o_iv + j] = e;
                    iv_ray_indices[o_iv + j] = ray;
                    iv_left[o_iv + j] = j < S ? 1 : 0;
                    iv_right[o_iv + j] = j > 0 ? 1 : 0;
                };
                if (S == 1) { edge(0, t_min); edge(1, t_max); }
                else if (sid == 0) edge(0, fmaxf(t - (t_next - t) * 0.5f, t_min));
                else {
                    edge(sid, (t + t_prev) * 0.5f);
                    if (sid == S - 1) edge(sid + 1, fminf(t + (t - t_prev) * 0.5f, t_max));
                }
            }
            t_carry = __shfl(t, ISP_L - 1, ISP_L);
        }
    }
}

// STAGED (query and key both batched, the keys of the rays a wave touches fit): a wave's 64 consecutive queries
// belong to a few consecutive rays whose key rows are one contiguous block; it is copied to LDS and searched there.
template <bool STAGED>
__global__ __launch_bounds__(256) void searchsorted_kernel(
    const float *__restrict__ q_vals, const int64_t *__restrict__ q_packed, const int64_t *__restrict__ q_ray_indices,
    int64_t q_n_rays, int64_t q_per_ray, int64_t q_total, const float *__restrict__ k_vals,
    const int64_t *__restrict__ k_packed, int64_t k_per_ray, int64_t *__restrict__ ids_left,
    int64_t *__restrict__ ids_right)
{
    __shared__ float s_key[4][STAGED ? IS_STAGE_MAX : 1];
    const bool q_batched = q_packed == nullptr;
    const int lane = lane_id();
    float *lk = s_key[threadIdx.x >> 6];
    for (int64_t tid0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; tid0 < q_total;
         tid0 += (int64_t)blockDim.x * gridDim.x) {
        const int64_t tid = tid0 + lane;
        if (STAGED) {
            const int64_t r_first = tid0 / q_per_ray;
            const int64_t r_last = min(tid0 + 63, q_total - 1) / q_per_ray;
            const int64_t blk0 = r_first * k_per_ray, n_here = (r_last - r_first + 1) * k_per_ray;
            __builtin_amdgcn_wave_barrier();
            for (int f = lane; f < n_here; f += 64) lk[f] = k_vals[blk0 + f];
            __builtin_amdgcn_wave_barrier();
            if (tid < q_total) {
                const int64_t ray_id = tid / q_per_ray;
                const int lbase = (int)((ray_id - r_first) * k_per_ray), llast = lbase + (int)k_per_ray - 1;
                const int p = upper_bound_lds(lk, lbase, llast, q_vals[tid]);
                ids_left[tid] = min(max(p - 1, lbase), llast) - lbase;
                ids_right[tid] = min(max(p, lbase), llast) - lbase;
            }
            continue;
        }
        if (tid >= q_total) continue;
        int64_t ray_id;
        if (q_batched) ray_id = tid / q_per_ray;
        else if (q_ray_indices) ray_id = q_ray_indices[tid];
        else {  // binary_search_chunk_id(tid) - 1, pdf.cu:65-80
            int64_t s = 0, e = q_n_rays;
            while (s < e) {
                const int64_t m = s + ((e - s) >> 1);
                if (!(q_packed[2 * m] > tid)) s = m + 1; else e = m;
            }
            ray_id = s - 1;
        }
        int64_t base, last;
        if (k_packed) { base = k_packed[2 * ray_id]; last = base + k_packed[2 * ray_id + 1] - 1; }
        else { base = ray_id * k_per_ray; last = base + k_per_ray - 1; }
        const int64_t p = upper_bound_f(k_vals, base, last, q_vals[tid]);
        const int64_t l = clamp64(p - 1, base, last), r = clamp64(p, base, last);
        ids_left[tid] = q_batched ? l - base : l;
        ids_right[tid] = q_batched ? r - base : r;
    }
}

// ------------------------------------------------------------------------------------------
// Interlevel ("proposal") loss of Mip-NeRF 360 for batched rays, forward and backward, one kernel each
// (ref: estimators/prop_net.py:232-256 = searchsorted + two gathers + five elementwise ops, and their autograd
// backward = two scatter_adds + ~10 elementwise ops):
//     l_j = max(w_j - wo_j, 0)^2 / (w_j + eps),  w_j = cq[j+1] - cq[j],  wo_j = ck[right_j] - ck[left_j],
//     left_j = clamp(upper_bound(kv, qv[j]) - 1), right_j = clamp(upper_bound(kv, qv[j+1]))
// L lanes per ray (power of two), key rows staged in LDS; the backward accumulates a ray's key-CDF gradient row
// in LDS (only the ray's own lane group touches it) and writes it once.
constexpr int PL_STAGE_MAX = 1024;  // key entries (vals + cdfs [+ grad row]) a wave may stage

// The mean form (PropNetEstimator.compute_loss takes `.mean()` of the loss, ref prop_net.py:151): the forward leaves one partial
// sum per wave instead of the loss array (the caller adds them up: deterministic, the wave -> rows assignment is fixed), the
// backward takes the scalar gradient of the mean.  Saves writing and re-reading the loss array and its expanded gradient.
__device__ __forceinline__ float pl_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void pdf_loss_fwd_kernel(const float *__restrict__ q_vals, const float *__restrict__ q_cdfs,
                                                           const float *__restrict__ k_vals, const float *__restrict__ k_cdfs,
                                                           int64_t n_rays, int Q1, int K1, int L, float eps,
                                                           float *__restrict__ loss /* or null */, uint32_t *__restrict__ ids /* left | right << 16, or null */,
                                                           float *__restrict__ partials /* [waves] or null */)
{
    float acc = 0.0f;
    __shared__ float s_kv[4][PL_STAGE_MAX];
    __shared__ float s_kc[4][PL_STAGE_MAX];
    const int lane = lane_id(), gl = lane & (L - 1), rpw = 64 / L, Q = Q1 - 1;
    float *kv = s_kv[threadIdx.x >> 6], *kc = s_kc[threadIdx.x >> 6];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r0 = wave * rpw; r0 < n_rays; r0 += n_waves * rpw) {
        const int n_here = (int)min((int64_t)rpw, n_rays - r0);
        __builtin_amdgcn_wave_barrier();
        for (int f = lane; f < n_here * K1; f += 64) { kv[f] = k_vals[r0 * K1 + f]; kc[f] = k_cdfs[r0 * K1 + f]; }
        __builtin_amdgcn_wave_barrier();
        const int slot = lane / L;
        const int64_t ray = r0 + slot;
        if (ray < n_rays) {
            const int kb = slot * K1, kl = kb + K1 - 1;
            const float *qv = q_vals + ray * Q1, *qc = q_cdfs + ray * Q1;
            for (int j = gl; j < Q; j += L) {
                const int pl = upper_bound_lds(kv, kb, kl, qv[j]);
                const int pr = upper_bound_lds(kv, kb, kl, qv[j + 1]);
                const int left = min(max(pl - 1, kb), kl), right = min(max(pr, kb), kl);
                const float w = qc[j + 1] - qc[j];
                const float wo = kc[right] - kc[left];
                const float d = fmaxf(w - wo, 0.0f);
                const float l = (d * d) / (w + eps);
                if (loss) loss[ray * Q + j] = l;
                acc += l;
                if (ids) ids[ray * Q + j] = (uint32_t)(left - kb) | ((uint32_t)(right - kb) << 16);
            }
        }
    }
    if (partials) {
        const float tot = pl_wave_sum(acc);
        if (lane == 0) partials[wave] = tot;
    }
}

// The same pass for rows of at most 64 query intervals whose key rows fit half the stage, in the style of
// importance_sampling_rows_kernel: one query interval per lane, the NEXT group's key rows and query edges requested before
// the current group is searched, both bisections run together for a wave-uniform number of rounds with selects.
template <int LL>
__global__ __launch_bounds__(256) void pdf_loss_fwd_rows_kernel(const float *__restrict__ q_vals, const float *__restrict__ q_cdfs,
                                                                const float *__restrict__ k_vals, const float *__restrict__ k_cdfs,
                                                                int64_t n_rays, int Q1, int K1, int n_rounds, float eps,
                                                                float *__restrict__ loss /* or null */, uint32_t *__restrict__ ids,
                                                                float *__restrict__ partials /* [waves] or null */)
{
    float acc = 0.0f;
    constexpr int RPW = 64 / LL;
    constexpr int STAGE = PL_STAGE_MAX / 2, SLOTS = STAGE / 64;
    __shared__ float s_kv[4][STAGE];
    __shared__ float s_kc[4][STAGE];
    const int lane = lane_id(), gl = lane & (LL - 1), grp = lane / LL, Q = Q1 - 1;
    float *kv = s_kv[threadIdx.x >> 6], *kc = s_kc[threadIdx.x >> 6];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float pkv[SLOTS], pkc[SLOTS], pq[4];
    auto rows_here = [&](int64_t r0) { return (int)min((int64_t)RPW, n_rays - r0) * K1; };
    auto prefetch = [&](int64_t r0) {
        const float *v = k_vals + r0 * K1, *c = k_cdfs + r0 * K1;
        const int n_h = rows_here(r0);
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (64 * k < n_h) {  // wave-uniform
                const int f = lane + 64 * k;
                pkv[k] = v[f < n_h ? f : 0];
                pkc[k] = c[f < n_h ? f : 0];
            }
        const int qo = (r0 + grp < n_rays && gl < Q) ? grp * Q1 + gl : 0;   // rows of >= 2 edges: qo + 1 is inside the first row
        const float *qv = q_vals + r0 * Q1, *qc = q_cdfs + r0 * Q1;
        pq[0] = qv[qo]; pq[1] = qv[qo + 1]; pq[2] = qc[qo]; pq[3] = qc[qo + 1];
    };
    int64_t r0 = uniform64_pdf(wave * RPW);
    if (r0 < n_rays) prefetch(r0);
    while (r0 < n_rays) {
        const int64_t r_next = r0 + n_waves * RPW;
        const int n_h = rows_here(r0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (64 * k < n_h) {
                const int f = lane + 64 * k;
                if (f < n_h) { kv[f] = pkv[k]; kc[f] = pkc[k]; }
            }
        const float qa = pq[0], qb = pq[1], w = pq[3] - pq[2];
        __builtin_amdgcn_wave_barrier();
        if (r_next < n_rays) prefetch(r_next);
        const bool ok = (r0 + grp < n_rays) && gl < Q;
        const int kb = (r0 + grp < n_rays) ? grp * K1 : 0, kl = kb + K1 - 1;
        int s0 = kb, e0 = kl, s1 = kb, e1 = kl;   // upper_bound(kv, qa), upper_bound(kv, qb) over [kb, kl)
        for (int it = 0; it < n_rounds; ++it) {
            const bool a0 = s0 < e0, a1 = s1 < e1;
            const int m0 = s0 + ((e0 - s0) >> 1), m1 = s1 + ((e1 - s1) >> 1);
            const bool g0 = !(kv[a0 ? m0 : kb] > qa), g1 = !(kv[a1 ? m1 : kb] > qb);
            s0 = (a0 && g0) ? m0 + 1 : s0; e0 = (a0 && !g0) ? m0 : e0;
            s1 = (a1 && g1) ? m1 + 1 : s1; e1 = (a1 && !g1) ? m1 : e1;
        }
        const int left = min(max(s0 - 1, kb), kl), right = min(max(s1, kb), kl);
        const float wo = kc[right] - kc[left];
        const float d = fmaxf(w - wo, 0.0f);
        const int o = grp * Q + gl;
        if (ok) {
            const float l = (d * d) / (w + eps);
            if (loss) (loss + r0 * Q)[o] = l;
            acc += l;
            if (ids) (ids + r0 * Q)[o] = (uint32_t)(left - kb) | ((uint32_t)(right - kb) << 16);
        }
        r0 = r_next;
    }
    if (partials) {
        const float tot = pl_wave_sum(acc);
        if (lane == 0) partials[wave] = tot;
    }
}

// Backward from the key indices saved by the forward pass: no searches, no key rows; a ray's key-CDF gradient row is
// accumulated in LDS by the ray's own lane group and written once.
__global__ __launch_bounds__(256) void pdf_loss_bwd_kernel(const float *__restrict__ q_cdfs, const float *__restrict__ k_cdfs,
                                                           const uint32_t *__restrict__ ids, int64_t n_rays, int Q1, int K1,
                                                           int L, float eps, const float *__restrict__ g_loss /* or null: the mean form */,
                                                           float *__restrict__ g_k_cdfs, float *__restrict__ g_q_cdfs,
                                                           const float *__restrict__ g_mean, float count)
{
    const float g_all = g_loss ? 0.0f : g_mean[0] / count;   // d mean / d loss_i (torch: grad / numel)
    __shared__ float s_g[4][PL_STAGE_MAX];
    __shared__ float s_gq[4][PL_STAGE_MAX];
    const int lane = lane_id(), gl = lane & (L - 1), rpw = 64 / L, Q = Q1 - 1;
    float *gk = s_g[threadIdx.x >> 6], *gq = s_gq[threadIdx.x >> 6];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r0 = wave * rpw; r0 < n_rays; r0 += n_waves * rpw) {
        const int n_here = (int)min((int64_t)rpw, n_rays - r0);
        __builtin_amdgcn_wave_barrier();
        for (int f = lane; f < n_here * K1; f += 64) gk[f] = 0.0f;
        if (g_q_cdfs) for (int f = lane; f < n_here * Q1; f += 64) gq[f] = 0.0f;
        __builtin_amdgcn_wave_barrier();
        const int slot = lane / L;
        const int64_t ray = r0 + slot;
        if (ray < n_rays) {
            const int kb = slot * K1;
            const float *qc = q_cdfs + ray * Q1, *kc = k_cdfs + ray * K1;
            for (int j = gl; j < Q; j += L) {
                const uint32_t id = ids[ray * Q + j];
                const int left = (int)(id & 0xFFFFu), right = (int)(id >> 16);
                const float w = qc[j + 1] - qc[j];
                const float d = fmaxf(w - (kc[right] - kc[left]), 0.0f);
                if (d > 0.0f) {
                    // d l / d wo = -2 d / (w + eps);  d l / d w = 2 d / (w + eps) - d^2 / (w + eps)^2
                    const float g = g_loss ? g_loss[ray * Q + j] : g_all, inv = 1.0f / (w + eps);
                    const float gwo = -2.0f * d * inv * g;
                    atomicAdd(&gk[kb + right], gwo);
                    atomicAdd(&gk[kb + left], -gwo);
                    if (g_q_cdfs) {
                        const float gw = (2.0f * d * inv - d * d * inv * inv) * g;
                        atomicAdd(&gq[slot * Q1 + j + 1], gw);
                        atomicAdd(&gq[slot * Q1 + j], -gw);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int f = lane; f < n_here * K1; f += 64) g_k_cdfs[r0 * K1 + f] = gk[f];
        if (g_q_cdfs) for (int f = lane; f < n_here * Q1; f += 64) g_q_cdfs[r0 * Q1 + f] = gq[f];
    }
}

// The backward for short rows, in the style of the forward: one query interval per lane, the key-CDF rows of the group in
// LDS (coalesced loads, requested one group ahead together with the lane's id / query weights / loss gradient), so that
// nothing in a group waits for a gather that depends on another load of the same group.
template <int LL>
__global__ __launch_bounds__(256) void pdf_loss_bwd_rows_kernel(const float *__restrict__ q_cdfs, const float *__restrict__ k_cdfs,
                                                                const uint32_t *__restrict__ ids, int64_t n_rays, int Q1, int K1,
                                                                float eps, const float *__restrict__ g_loss /* or null: the mean form */,
                                                                float *__restrict__ g_k_cdfs, float *__restrict__ g_q_cdfs,
                                                                const float *__restrict__ g_mean, float count)
{
    const float g_all = g_loss ? 0.0f : g_mean[0] / count;   // d mean / d loss_i (torch: grad / numel)
    constexpr int RPW = 64 / LL;
    constexpr int STAGE = PL_STAGE_MAX / 2, SLOTS = STAGE / 64;
    __shared__ float s_kc[4][STAGE];
    __shared__ float s_gk[4][STAGE];
    __shared__ float s_gq[4][64 + RPW];      // RPW rows of Q1 <= LL + 1 entries
    const int lane = lane_id(), gl = lane & (LL - 1), grp = lane / LL, Q = Q1 - 1;
    float *kc = s_kc[threadIdx.x >> 6], *gk = s_gk[threadIdx.x >> 6], *gq = s_gq[threadIdx.x >> 6];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float pkc[SLOTS], pq0, pq1, pg;
    uint32_t pid;
    auto rows_here = [&](int64_t r0) { return (int)min((int64_t)RPW, n_rays - r0); };
    auto prefetch = [&](int64_t r0) {
        const float *c = k_cdfs + r0 * K1;
        const int n_h = rows_here(r0) * K1;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (64 * k < n_h) {  // wave-uniform
                const int f = lane + 64 * k;
                pkc[k] = c[f < n_h ? f : 0];
            }
        const bool ok = (r0 + grp < n_rays) && gl < Q;
        const int qo = ok ? grp * Q1 + gl : 0, lo = ok ? grp * Q + gl : 0;
        const float *qc = q_cdfs + r0 * Q1;
        pq0 = qc[qo]; pq1 = qc[qo + 1];
        pid = (ids + r0 * Q)[lo];
        pg = g_loss ? (g_loss + r0 * Q)[lo] : g_all;
    };
    int64_t r0 = uniform64_pdf(wave * RPW);
    if (r0 < n_rays) prefetch(r0);
    while (r0 < n_rays) {
        const int64_t r_next = r0 + n_waves * RPW;
        const int n_r = rows_here(r0), n_h = n_r * K1;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (64 * k < n_h) {
                const int f = lane + 64 * k;
                if (f < n_h) { kc[f] = pkc[k]; gk[f] = 0.0f; }
            }
        if (g_q_cdfs) for (int f = lane; f < n_r * Q1; f += 64) gq[f] = 0.0f;
        const uint32_t id = pid;
        const float w = pq1 - pq0, g = pg;
        __builtin_amdgcn_wave_barrier();
        if (r_next < n_rays) prefetch(r_next);
        const bool ok = (r0 + grp < n_rays) && gl < Q;
        const int kb = (r0 + grp < n_rays) ? grp * K1 : 0;
        const int left = min((int)(id & 0xFFFFu), K1 - 1), right = min((int)(id >> 16), K1 - 1);
        const float d = fmaxf(w - (kc[kb + right] - kc[kb + left]), 0.0f);
        if (ok && d > 0.0f) {
            // d l / d wo = -2 d / (w + eps);  d l / d w = 2 d / (w + eps) - d^2 / (w + eps)^2
            const float inv = 1.0f / (w + eps);
            const float gwo = -2.0f * d * inv * g;
            atomicAdd(&gk[kb + right], gwo);
            atomicAdd(&gk[kb + left], -gwo);
            if (g_q_cdfs) {
                const float gw = (2.0f * d * inv - d * d * inv * inv) * g;
                atomicAdd(&gq[grp * Q1 + gl + 1], gw);
                atomicAdd(&gq[grp * Q1 + gl], -gw);
            }
        }
        __builtin_amdgcn_wave_barrier();
        float *ok_rows = g_k_cdfs + r0 * K1;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (64 * k < n_h) {
                const int f = lane + 64 * k;
                if (f < n_h) ok_rows[f] = gk[f];
            }
        if (g_q_cdfs) for (int f = lane; f < n_r * Q1; f += 64) (g_q_cdfs + r0 * Q1)[f] = gq[f];
        r0 = r_next;
    }
}

}  // namespace nfa

using namespace nfa;

extern "C" {

static int launch_importance_sampling(const float *in_vals, const float *cdfs, const int64_t *in_packed_info, int64_t n_rays,
                                      int64_t n_edges_per_ray, int64_t n_samples, int stratified, uint64_t seed,
                                      uint64_t offset, float *out_intervals, float *out_samples, int transform, float t_a,
                                      float t_b, float *out_ts, float *out_te, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0, "importance_sampling: negative n_rays");
    NFA_REQUIRE(n_samples >= 1, "importance_sampling: n_intervals_per_ray must be >= 1");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(in_vals && cdfs && out_intervals, "importance_sampling: null pointer");
    NFA_REQUIRE(in_packed_info || n_edges_per_ray >= 1, "importance_sampling: need packed_info or n_edges_per_ray >= 1");
    NFA_REQUIRE(transform == 0 || (out_ts && out_te), "importance_sampling: t_starts / t_ends missing");
    int L = 2;
    while (L < 64 && L < n_samples) L <<= 1;
    const int64_t rays_per_wave = 64 / L;
    // rays per wave block (the Philox draws of a block are shared out over the lanes): 64 when that still leaves >= 4096 waves
    int RB = (int)rays_per_wave;
    while (RB < 64 && n_rays / (2 * RB) >= 4096) RB <<= 1;
    const int64_t n_waves = ceil_div64(n_rays, (int64_t)RB);
    const unsigned grid = grid_1d(n_waves * 64, 256, 1 << 16);
    const bool staged = !in_packed_info && rays_per_wave * n_edges_per_ray <= IS_STAGE_MAX;
    if (staged && n_samples <= 64 && n_edges_per_ray < (1 << 20)) {
        int n_rounds = 0;
        while ((1 << n_rounds) <= (int)n_edges_per_ray - 1) ++n_rounds;   // rounds until a range of n_edges - 1 entries is empty
        // more than 16 samples per ray: 16 lanes per ray and 2 or 4 samples per lane (4 rays per group instead of 1-2:
        // the per-group work is shared and a lane's bisections overlap), if four CDF rows fit the stage
        const bool blocks = n_samples > 16 && 4 * n_edges_per_ray <= IS_STAGE_MAX;
        const int LL = blocks ? 16 : L, NB = blocks ? (n_samples > 32 ? 4 : 2) : 1;
        const int64_t rpw = 64 / LL;
        int RBk = (int)rpw;
        while (RBk < 64 && n_rays / (2 * RBk) >= 4096) RBk <<= 1;
        const unsigned gridk = grid_1d(ceil_div64(n_rays, (int64_t)RBk) * 64, 256, 1 << 16);
#define NFA_IS_ROWS(L_, NB_)                                                                                              \
    hipLaunchKernelGGL((importance_sampling_rows_kernel<L_, NB_>), dim3(gridk), dim3(256), 0, as_stream(stream), in_vals, cdfs, n_rays, \
                       (int)n_edges_per_ray, (int)n_samples, n_rounds, RBk, stratified, seed, offset, out_intervals, out_samples, \
                       transform, t_a, t_b, out_ts, out_te)
        if (NB == 4) NFA_IS_ROWS(16, 4);
        else if (NB == 2) NFA_IS_ROWS(16, 2);
        else switch (LL) {
            case 2: NFA_IS_ROWS(2, 1); break;
            case 4: NFA_IS_ROWS(4, 1); break;
            case 8: NFA_IS_ROWS(8, 1); break;
            case 16: NFA_IS_ROWS(16, 1); break;
            case 32: NFA_IS_ROWS(32, 1); break;
            default: NFA_IS_ROWS(64, 1); break;
        }
#undef NFA_IS_ROWS
    } else if (staged)
        hipLaunchKernelGGL(importance_sampling_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), in_vals, cdfs,
                           in_packed_info, n_rays, n_edges_per_ray, n_samples, L, RB, stratified, seed, offset, out_intervals,
                           out_samples, transform, t_a, t_b, out_ts, out_te);
    else
        hipLaunchKernelGGL(importance_sampling_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), in_vals, cdfs,
                           in_packed_info, n_rays, n_edges_per_ray, n_samples, L, RB, stratified, seed, offset, out_intervals,
                           out_samples, transform, t_a, t_b, out_ts, out_te);
    NFA_CHECK_LAUNCH("importance_sampling");
    return NFA_OK;
}

int nfa_importance_sampling(const float *in_vals, const float *cdfs, const int64_t *in_packed_info, int64_t n_rays,
                            int64_t n_edges_per_ray, int64_t n_samples, int stratified, uint64_t seed,
                            uint64_t offset, float *out_intervals, float *out_samples, nfa_stream_t stream)
{
    return launch_importance_sampling(in_vals, cdfs, in_packed_info, n_rays, n_edges_per_ray, n_samples, stratified, seed,
                                      offset, out_intervals, out_samples, 0, 0.f, 0.f, nullptr, nullptr, stream);
}

int nfa_importance_sampling_t(const float *in_vals, const float *cdfs, const int64_t *in_packed_info, int64_t n_rays,
                              int64_t n_edges_per_ray, int64_t n_samples, int stratified, uint64_t seed, uint64_t offset,
                              float *out_intervals, float *out_samples, int transform, float t_a, float t_b,
                              float *out_t_starts, float *out_t_ends, nfa_stream_t stream)
{
    NFA_REQUIRE(transform == 1 || transform == 2, "importance_sampling_t: transform must be 1 (uniform) or 2 (lindisp)");
    return launch_importance_sampling(in_vals, cdfs, in_packed_info, n_rays, n_edges_per_ray, n_samples, stratified, seed,
                                      offset, out_intervals, out_samples, transform, t_a, t_b, out_t_starts, out_t_ends, stream);
}

int nfa_importance_sampling_packed(const float *in_vals, const float *cdfs, const int64_t *in_packed_info, int64_t n_rays,
                                   int64_t n_edges_per_ray, const int64_t *sm_packed_info, const int64_t *iv_packed_info,
                                   int stratified, uint64_t seed, uint64_t offset, float *sm_vals, int64_t *sm_ray_indices,
                                   float *iv_vals, int64_t *iv_ray_indices, uint8_t *iv_is_left, uint8_t *iv_is_right,
                                   nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0, "importance_sampling_packed: negative n_rays");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(in_vals && cdfs && sm_packed_info && iv_packed_info, "importance_sampling_packed: null pointer");
    NFA_REQUIRE(in_packed_info || n_edges_per_ray >= 1, "importance_sampling_packed: need packed_info or n_edges_per_ray >= 1");
    // (output pointers may be NULL only when every count is zero, which the caller knows from the totals)
    const int64_t n_waves = ceil_div64(n_rays, 64 / ISP_L);
    const unsigned grid = grid_1d(n_waves * 64, 256, 1 << 16);
    hipLaunchKernelGGL(importance_sampling_packed_kernel, dim3(grid), dim3(256), 0, as_stream(stream), in_vals, cdfs, in_packed_info,
                       n_rays, n_edges_per_ray, sm_packed_info, iv_packed_info, stratified, seed, offset, sm_vals, sm_ray_indices,
                       iv_vals, iv_ray_indices, iv_is_left, iv_is_right);
    NFA_CHECK_LAUNCH("importance_sampling_packed");
    return NFA_OK;
}

int nfa_searchsorted(const float *q_vals, const int64_t *q_packed_info, const int64_t *q_ray_indices, int64_t q_n_rays,
                     int64_t q_per_ray, int64_t q_total, const float *k_vals, const int64_t *k_packed_info,
                     int64_t k_per_ray, int64_t *ids_left, int64_t *ids_right, nfa_stream_t stream)
{
    NFA_REQUIRE(q_total >= 0, "searchsorted: negative size");
    if (q_total == 0) return NFA_OK;
    NFA_REQUIRE(q_vals && k_vals && ids_left && ids_right, "searchsorted: null pointer");
    NFA_REQUIRE(q_packed_info || q_per_ray >= 1, "searchsorted: batched query needs q_per_ray");
    NFA_REQUIRE(k_packed_info || k_per_ray >= 1, "searchsorted: batched key needs k_per_ray");
    // rays touched by 64 consecutive queries: at most 63 / q_per_ray + 2
    const bool staged = !q_packed_info && !k_packed_info && (63 / q_per_ray + 2) * k_per_ray <= IS_STAGE_MAX;
    if (staged)
        hipLaunchKernelGGL(searchsorted_kernel<true>, dim3(grid_1d(q_total, 256)), dim3(256), 0, as_stream(stream), q_vals,
                           q_packed_info, q_ray_indices, q_n_rays, q_per_ray, q_total, k_vals, k_packed_info, k_per_ray,
                           ids_left, ids_right);
    else
        hipLaunchKernelGGL(searchsorted_kernel<false>, dim3(grid_1d(q_total, 256)), dim3(256), 0, as_stream(stream), q_vals,
                           q_packed_info, q_ray_indices, q_n_rays, q_per_ray, q_total, k_vals, k_packed_info, k_per_ray,
                           ids_left, ids_right);
    NFA_CHECK_LAUNCH("searchsorted");
    return NFA_OK;
}

static int pdf_loss_lanes(int Q, int K1, int Q1)
{
    int L = 2;
    while (L < 64 && L < Q) L <<= 1;
    while (L < 64 && (64 / L) * (K1 > Q1 ? K1 : Q1) > PL_STAGE_MAX) L <<= 1;  // fewer rays per wave until the rows fit
    return L;
}

static unsigned pdf_loss_grid(int64_t n_rays, int32_t n_query_edges, int32_t n_key_edges)
{
    const int L = pdf_loss_lanes(n_query_edges - 1, n_key_edges, n_query_edges);
    return grid_1d(ceil_div64(n_rays, 64 / L) * 64, 256, 1 << 16);
}

static int pdf_loss_fwd_impl(const float *q_vals, const float *q_cdfs, const float *k_vals, const float *k_cdfs, int64_t n_rays,
                             int32_t n_query_edges, int32_t n_key_edges, float eps, float *loss, uint32_t *key_ids,
                             float *partials, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_query_edges >= 2 && n_key_edges >= 1, "pdf_loss_fwd: bad sizes");
    NFA_REQUIRE(n_key_edges <= PL_STAGE_MAX && n_query_edges <= PL_STAGE_MAX, "pdf_loss_fwd: rows longer than 1024 edges are not supported");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(q_vals && q_cdfs && k_vals && k_cdfs && (loss || partials), "pdf_loss_fwd: null pointer");
    const int L = pdf_loss_lanes(n_query_edges - 1, n_key_edges, n_query_edges);
    const unsigned grid = pdf_loss_grid(n_rays, n_query_edges, n_key_edges);
    if (n_query_edges - 1 <= L && (64 / L) * n_key_edges <= PL_STAGE_MAX / 2) {
        int n_rounds = 0;
        while ((1 << n_rounds) <= (int)n_key_edges - 1) ++n_rounds;
#define NFA_PL_ROWS(LL)                                                                                                   \
    hipLaunchKernelGGL(pdf_loss_fwd_rows_kernel<LL>, dim3(grid), dim3(256), 0, as_stream(stream), q_vals, q_cdfs, k_vals, k_cdfs, \
                       n_rays, (int)n_query_edges, (int)n_key_edges, n_rounds, eps, loss, key_ids, partials)
        switch (L) {
            case 2: NFA_PL_ROWS(2); break;
            case 4: NFA_PL_ROWS(4); break;
            case 8: NFA_PL_ROWS(8); break;
            case 16: NFA_PL_ROWS(16); break;
            case 32: NFA_PL_ROWS(32); break;
            default: NFA_PL_ROWS(64); break;
        }
#undef NFA_PL_ROWS
        NFA_CHECK_LAUNCH("pdf_loss_fwd");
        return NFA_OK;
    }
    hipLaunchKernelGGL(pdf_loss_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), q_vals, q_cdfs, k_vals, k_cdfs,
                       n_rays, (int)n_query_edges, (int)n_key_edges, L, eps, loss, key_ids, partials);
    NFA_CHECK_LAUNCH("pdf_loss_fwd");
    return NFA_OK;
}

int nfa_pdf_loss_fwd(const float *q_vals, const float *q_cdfs, const float *k_vals, const float *k_cdfs, int64_t n_rays,
                     int32_t n_query_edges, int32_t n_key_edges, float eps, float *loss, uint32_t *key_ids, nfa_stream_t stream)
{
    NFA_REQUIRE(loss || n_rays == 0, "pdf_loss_fwd: loss is null");
    return pdf_loss_fwd_impl(q_vals, q_cdfs, k_vals, k_cdfs, n_rays, n_query_edges, n_key_edges, eps, loss, key_ids, nullptr, stream);
}

int64_t nfa_pdf_loss_partials(int64_t n_rays, int32_t n_query_edges, int32_t n_key_edges)
{
    if (n_rays <= 0 || n_query_edges < 2 || n_key_edges < 1) return 0;
    return (int64_t)pdf_loss_grid(n_rays, n_query_edges, n_key_edges) * 4;   // waves of the launch
}

int nfa_pdf_loss_sum_fwd(const float *q_vals, const float *q_cdfs, const float *k_vals, const float *k_cdfs, int64_t n_rays,
                         int32_t n_query_edges, int32_t n_key_edges, float eps, float *partials, uint32_t *key_ids, nfa_stream_t stream)
{
    NFA_REQUIRE(partials || n_rays == 0, "pdf_loss_sum_fwd: partials is null");
    return pdf_loss_fwd_impl(q_vals, q_cdfs, k_vals, k_cdfs, n_rays, n_query_edges, n_key_edges, eps, nullptr, key_ids, partials, stream);
}

static int pdf_loss_bwd_impl(const float *q_cdfs, const float *k_cdfs, const uint32_t *key_ids, int64_t n_rays,
                             int32_t n_query_edges, int32_t n_key_edges, float eps, const float *g_loss, const float *g_mean,
                             float *g_k_cdfs, float *g_q_cdfs, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_query_edges >= 2 && n_key_edges >= 1, "pdf_loss_bwd: bad sizes");
    NFA_REQUIRE(n_key_edges <= PL_STAGE_MAX && n_query_edges <= PL_STAGE_MAX, "pdf_loss_bwd: rows longer than 1024 edges are not supported");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(q_cdfs && k_cdfs && key_ids && (g_loss || g_mean) && g_k_cdfs, "pdf_loss_bwd: null pointer");
    const int L = pdf_loss_lanes(n_query_edges - 1, n_key_edges, n_query_edges);
    const unsigned grid = pdf_loss_grid(n_rays, n_query_edges, n_key_edges);
    const float count = (float)((double)n_rays * (double)(n_query_edges - 1));
    if (n_query_edges - 1 <= L && (64 / L) * n_key_edges <= PL_STAGE_MAX / 2) {
#define NFA_PLB_ROWS(LL)                                                                                                  \
    hipLaunchKernelGGL(pdf_loss_bwd_rows_kernel<LL>, dim3(grid), dim3(256), 0, as_stream(stream), q_cdfs, k_cdfs, key_ids, n_rays, \
                       (int)n_query_edges, (int)n_key_edges, eps, g_loss, g_k_cdfs, g_q_cdfs, g_mean, count)
        switch (L) {
            case 2: NFA_PLB_ROWS(2); break;
            case 4: NFA_PLB_ROWS(4); break;
            case 8: NFA_PLB_ROWS(8); break;
            case 16: NFA_PLB_ROWS(16); break;
            case 32: NFA_PLB_ROWS(32); break;
            default: NFA_PLB_ROWS(64); break;
        }
#undef NFA_PLB_ROWS
        NFA_CHECK_LAUNCH("pdf_loss_bwd");
        return NFA_OK;
    }
    hipLaunchKernelGGL(pdf_loss_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), q_cdfs, k_cdfs, key_ids, n_rays,
                       (int)n_query_edges, (int)n_key_edges, L, eps, g_loss, g_k_cdfs, g_q_cdfs, g_mean, count);
    NFA_CHECK_LAUNCH("pdf_loss_bwd");
    return NFA_OK;
}

int nfa_pdf_loss_bwd(const float *q_cdfs, const float *k_cdfs, const uint32_t *key_ids, int64_t n_rays,
                     int32_t n_query_edges, int32_t n_key_edges, float eps, const float *g_loss, float *g_k_cdfs,
                     float *g_q_cdfs, nfa_stream_t stream)
{
    NFA_REQUIRE(g_loss || n_rays == 0, "pdf_loss_bwd: g_loss is null");
    return pdf_loss_bwd_impl(q_cdfs, k_cdfs, key_ids, n_rays, n_query_edges, n_key_edges, eps, g_loss, nullptr, g_k_cdfs, g_q_cdfs, stream);
}

int nfa_pdf_loss_mean_bwd(const float *q_cdfs, const float *k_cdfs, const uint32_t *key_ids, int64_t n_rays,
                          int32_t n_query_edges, int32_t n_key_edges, float eps, const float *g_mean, float *g_k_cdfs,
                          float *g_q_cdfs, nfa_stream_t stream)
{
    NFA_REQUIRE(g_mean || n_rays == 0, "pdf_loss_mean_bwd: g_mean is null");
    return pdf_loss_bwd_impl(q_cdfs, k_cdfs, key_ids, n_rays, n_query_edges, n_key_edges, eps, nullptr, g_mean, g_k_cdfs, g_q_cdfs, stream);
}

}  // extern "C"
