/*
 * oracle/nerfacc_oracle.c -- CPU restatement of nerfacc's native hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (nerfacc_amd/) never does.
 *
 * Every function restates, in plain scalar C and in source order, what one of
 * the reference's CUDA kernels computes.  Citations are relative to
 * /root/reference/nerfacc/cuda/csrc/.  Build with
 *     gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (no FMA contraction: the canonical semantics are un-fused IEEE fp32 in
 * source order, see DESIGN.md "Floating-point contract").
 *
 * Parity pinning: checked against the reference's own hard-coded test
 * vectors and against the importable pure-torch twins of the reference
 * (oracle/gen_golden.py, tests/test_oracle_golden.py).  The traversal has no
 * golden vectors in the reference; it is pinned by the reference's own
 * property tests (tests/test_grid.py) evaluated through the reference's
 * `_query`.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* float -> int conversion with the saturating behaviour of the GPU's
 * v_cvt_i32_f32 (and of CUDA's cvt.rzi.s32.f32); plain C casts are undefined
 * out of range. */
static inline int32_t f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi)
{
    int32_t m = v < hi ? v : hi;
    return lo > m ? lo : m;
}

/* ------------------------------------------------------------------------ */
/* Slab test.  include/utils_grid.cuh:10-55, include/data_spec_packed.cuh:43-59
 */
static inline int slab_test(const float o[3], const float d[3], const float inv[3],
                            const float bmin[3], const float bmax[3],
                            float near, float far, float *tmin_out, float *tmax_out)
{
    float tmin, tmax, lo, hi;
    (void)d;
    if (inv[0] >= 0) { tmin = (bmin[0] - o[0]) * inv[0]; tmax = (bmax[0] - o[0]) * inv[0]; }
    else             { tmin = (bmax[0] - o[0]) * inv[0]; tmax = (bmin[0] - o[0]) * inv[0]; }
    for (int a = 1; a < 3; ++a) {
        if (inv[a] >= 0) { lo = (bmin[a] - o[a]) * inv[a]; hi = (bmax[a] - o[a]) * inv[a]; }
        else             { lo = (bmax[a] - o[a]) * inv[a]; hi = (bmin[a] - o[a]) * inv[a]; }
        if (tmin > hi || lo > tmax) return 0;
        if (lo > tmin) tmin = lo;
        if (hi < tmax) tmax = hi;
    }
    if (tmax <= 0) return 0;
    *tmin_out = fmaxf(tmin, near);
    *tmax_out = fminf(tmax, far);
    return 1;
}

/* grid.cu:284-313 (kernel), grid.cu:477-519 (host) */
ORC_API int orc_ray_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays,
                                   const float *aabbs, int64_t n_aabbs, float near, float far,
                                   float miss_value, float *t_mins, float *t_maxs, uint8_t *hits)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        for (int64_t g = 0; g < n_aabbs; ++g) {
            float tmin, tmax;
            int hit = slab_test(o, d, inv, aabbs + 6 * g, aabbs + 6 * g + 3, near, far, &tmin, &tmax);
            int64_t k = r * n_aabbs + g;
            t_mins[k] = hit ? tmin : miss_value;
            t_maxs[k] = hit ? tmax : miss_value;
            hits[k] = (uint8_t)hit;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Traversal.  grid.cu:23-28 (_calc_dt), grid.cu:68-282 (kernel),
 * include/utils_grid.cuh:58-142 (setup_traversal / single_traversal).       */

typedef struct {
    float *vals;          /* [n] */
    int64_t *ray_indices; /* [n] */
    uint8_t *is_left;     /* [n] or NULL */
    uint8_t *is_right;    /* [n] or NULL */
    uint8_t *is_valid;    /* [n] or NULL */
    int64_t *chunk_starts;/* [n_rays] (read in fill pass) */
    int64_t *chunk_cnts;  /* [n_rays] (written)  NULL => spec disabled */
    float *t_starts;      /* [n] or NULL: samples only -- the values the sampler takes from the interval stream, */
    float *t_ends;        /*   intervals.vals[is_left] / [is_right] (estimators/occ_grid.py:174-175), written directly */
} orc_segments;

static inline float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    return fmaxf(dt_min, fminf(t * cone_angle, dt_max)); /* utils_math.cuh:1167-1170 */
}

/* march t_last forward in whole steps until the step's midpoint is at or past
 * `target` (grid.cu:153-163 and :196-205).  The no-progress guard is ours: the
 * reference spins forever when t_last + dt == t_last. */
static inline float fast_forward(float t_last, float target, float step_size, float cone_angle)
{
    if (step_size <= 0.0f) return target;
    float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
    for (;;) {
        if (t_last + dt * 0.5f >= target) break;
        float t_new = t_last + dt;
        if (t_new == t_last) { t_last = target; break; }
        t_last = t_new;
    }
    return t_last;
}

static void traverse_one_ray(
    int64_t tid, const float *rays_o, const float *rays_d,
    int32_t n_grids, const int32_t res[3], const uint8_t *binaries, const float *aabbs,
    const uint8_t *hits, const float *t_sorted, const int64_t *t_indices,
    float near_plane, float far_plane, float step_size, float cone_angle, int32_t limit,
    int first_pass, orc_segments *iv, orc_segments *sm, float *terminate_planes)
{
    const float eps = 1e-6f; /* grid.cu:95 */
    const int has_iv = iv->chunk_cnts != NULL, has_sm = sm->chunk_cnts != NULL;
    if (has_iv && !first_pass && iv->chunk_cnts[tid] == 0) return; /* grid.cu:103-106 */
    if (has_sm && !first_pass && sm->chunk_cnts[tid] == 0) return;

    int64_t chunk_start = 0, chunk_start_bin = 0;
    if (!first_pass) {
        if (has_iv) chunk_start = iv->chunk_starts[tid];
        if (has_sm) chunk_start_bin = sm->chunk_starts[tid];
    }
    const float o[3] = {rays_o[3 * tid], rays_o[3 * tid + 1], rays_o[3 * tid + 2]};
    const float d[3] = {rays_d[3 * tid], rays_d[3 * tid + 1], rays_d[3 * tid + 2]};
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float resf[3] = {(float)res[0], (float)res[1], (float)res[2]};

    const int64_t base_hits = tid * n_grids, base_t = tid * n_grids * 2;
    int64_t n_intervals = 0, n_samples = 0;
    float t_last = near_plane;
    int continuous = 0;

    for (int64_t i = base_t; i < base_t + 2 * n_grids - 1; ++i) { /* grid.cu:125 */
        int is_entering = t_indices[i] < n_grids;
        int64_t level = t_indices[i] % n_grids;
        if (!hits[base_hits + level]) continue;
        if (!is_entering) { /* leaving this grid: are we still inside the next one? */
            if (t_indices[i + 1] < n_grids) continue;
            level = t_indices[i + 1] % n_grids;
            if (!hits[base_hits + level]) continue;
        }
        float this_tmin = fmaxf(t_sorted[i], near_plane);
        float this_tmax = fminf(t_sorted[i + 1], far_plane);
        if (this_tmin >= this_tmax) continue;

        if (!continuous) t_last = fast_forward(t_last, this_tmin, step_size, cone_angle);

        /* --- setup_traversal, utils_grid.cuh:58-114 --- */
        const float *bmin = aabbs + 6 * level, *bmax = aabbs + 6 * level + 3;
        float tdist[3], delta[3];
        int32_t step[3], cur[3], fin[3], overflow[3];
        for (int a = 0; a < 3; ++a) {
            float voxel = (bmax[a] - bmin[a]) / resf[a];
            float ray_start = o[a] + d[a] * (this_tmin + eps);
            float ray_end = o[a] + d[a] * (this_tmax - eps);
            cur[a] = clampi(f2i(((ray_start - bmin[a]) / (bmax[a] - bmin[a])) * resf[a]), 0, res[a] - 1);
            fin[a] = clampi(f2i(((ray_end - bmin[a]) / (bmax[a] - bmin[a])) * resf[a]), 0, res[a] - 1);
            int32_t start_index = cur[a] + (d[a] > 0 ? 1 : 0);
            float tmax_a = ((bmin[a] + (((float)start_index * voxel) - ray_start)) * inv[a]) + this_tmin;
            float step_f = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
            tdist[a] = (d[a] == 0.0f) ? this_tmax : tmax_a;
            step[a] = (int32_t)step_f;
            float delta_tmp = voxel * inv[a] * step_f;
            delta[a] = (d[a] == 0.0f) ? this_tmax : delta_tmp;
            overflow[a] = fin[a] + step[a];
        }

        /* Safety cap (ours): a DDA cannot legitimately take more cell steps. */
        int64_t cells_left = (int64_t)res[0] + res[1] + res[2] + 3;
        while (limit <= 0 || n_samples < limit) { /* grid.cu:184 */
            float t_traverse = fminf(fminf(tdist[0], fminf(tdist[1], tdist[2])), this_tmax);
            int64_t cell_id = (int64_t)(cur[0] * res[1] * res[2] + cur[1] * res[2] + cur[2])
                              + level * (int64_t)res[0] * res[1] * res[2];
            if (!binaries[cell_id]) {
                t_last = fast_forward(t_last, t_traverse, step_size, cone_angle);
                continuous = 0;
            } else {
                while (limit <= 0 || n_samples < limit) { /* grid.cu:208 */
                    float t_next;
                    if (step_size <= 0.0f) {
                        t_next = t_traverse;
                    } else {
                        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_traverse) break;
                        t_next = t_last + dt;
                        if (t_next == t_last) break; /* no-progress guard (ours) */
                    }
                    if (has_iv) {
                        if (!continuous) {
                            if (!first_pass) {
                                int64_t idx = chunk_start + n_intervals;
                                iv->vals[idx] = t_last; iv->ray_indices[idx] = tid; iv->is_left[idx] = 1;
                            }
                            n_intervals++;
                            if (!first_pass) {
                                int64_t idx = chunk_start + n_intervals;
                                iv->vals[idx] = t_next; iv->ray_indices[idx] = tid; iv->is_right[idx] = 1;
                            }
                            n_intervals++;
                        } else {
                            if (!first_pass) {
                                int64_t idx = chunk_start + n_intervals;
                                iv->vals[idx] = t_next; iv->ray_indices[idx] = tid;
                                iv->is_left[idx - 1] = 1; iv->is_right[idx] = 1;
                            }
                            n_intervals++;
                        }
                    }
                    if (has_sm && !first_pass) {
                        int64_t idx = chunk_start_bin + n_samples;
                        if (sm->vals) sm->vals[idx] = (t_next + t_last) * 0.5f;
                        sm->ray_indices[idx] = tid;
                        if (sm->is_valid) sm->is_valid[idx] = 1;
                        if (sm->t_starts) { sm->t_starts[idx] = t_last; sm->t_ends[idx] = t_next; }
                    }
                    n_samples++;
                    continuous = 1;
                    t_last = t_next;
                    if (t_next >= t_traverse) break;
                }
            }
            /* --- single_traversal, utils_grid.cuh:116-142 --- */
            int a = (tdist[0] < tdist[1] && tdist[0] < tdist[2]) ? 0 : (tdist[1] < tdist[2] ? 1 : 2);
            cur[a] += step[a];
            tdist[a] += delta[a];
            if (cur[a] == overflow[a]) break;
            if (--cells_left <= 0) break;
        }
    }
    if (terminate_planes) terminate_planes[tid] = t_last;
    if (has_iv) iv->chunk_cnts[tid] = n_intervals;
    if (has_sm) sm->chunk_cnts[tid] = n_samples;
}

/* One launch of traverse_grids_kernel (grid.cu:68).  `first_pass` selects the
 * count pass.  Pointers inside iv/sm may be NULL exactly where the reference
 * passes undefined tensors.  iv_* / sm_* are passed flat for ctypes. */
ORC_API int orc_traverse_grids_pass(
    int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *rays_mask,
    int32_t n_grids, const int32_t *res, const uint8_t *binaries, const float *aabbs,
    const uint8_t *hits, const float *t_sorted, const int64_t *t_indices,
    const float *near_planes, const float *far_planes, float step_size, float cone_angle,
    int32_t limit, int first_pass,
    float *iv_vals, int64_t *iv_ray_indices, uint8_t *iv_is_left, uint8_t *iv_is_right,
    int64_t *iv_chunk_starts, int64_t *iv_chunk_cnts,
    float *sm_vals, int64_t *sm_ray_indices, uint8_t *sm_is_valid,
    int64_t *sm_chunk_starts, int64_t *sm_chunk_cnts,
    float *terminate_planes)
{
    orc_segments iv = {iv_vals, iv_ray_indices, iv_is_left, iv_is_right, NULL, iv_chunk_starts, iv_chunk_cnts, NULL, NULL};
    orc_segments sm = {sm_vals, sm_ray_indices, NULL, NULL, sm_is_valid, sm_chunk_starts, sm_chunk_cnts, NULL, NULL};
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t tid = 0; tid < n_rays; ++tid) {
        if (rays_mask && !rays_mask[tid]) continue; /* grid.cu:100 */
        traverse_one_ray(tid, rays_o, rays_d, n_grids, res, binaries, aabbs, hits, t_sorted,
                         t_indices, near_planes[tid], far_planes[tid], step_size, cone_angle,
                         limit, first_pass, &iv, &sm, terminate_planes);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Packed-segment scans.  include/utils_scan.cuh:28-112 (inclusive),
 * :153-239 (exclusive); hosts scan.cu:9-125.  The association order of the
 * reference is reproduced exactly: 32-element tiles, previous tiles' total
 * folded into element 0, Blelloch up-sweep + down-sweep.                     */

#define ORC_TILE 32 /* 2 * num_threads_x, scan.cu:38 dim3(16, 32) */

static inline float scan_op(float a, float b, int is_prod) { return is_prod ? a * b : a + b; }

static void tile_scan(float *buf, int is_prod)
{
    const int nx = ORC_TILE / 2;
    /* up-sweep, utils_scan.cuh:74-80 */
    for (int s = nx, dd = 1; s >= 1; s >>= 1, dd <<= 1)
        for (int t = 0; t < s; ++t) {
            int off = (2 * t + 1) * dd - 1;
            buf[off + dd] = scan_op(buf[off], buf[off + dd], is_prod);
        }
    /* down-sweep, utils_scan.cuh:83-89 */
    for (int s = 2, dd = nx / 2; dd >= 1; s <<= 1, dd >>= 1)
        for (int t = 0; t < s - 1; ++t) {
            int off = 2 * (t + 1) * dd - 1;
            buf[off + dd] = scan_op(buf[off], buf[off + dd], is_prod);
        }
}

/* One row (ray).  `src`/`tgt` are accessed through (base, stride) so the
 * reverse-iterator launches of the backward pass are the same code with
 * stride -1 (scan.cu:41-51). */
static void row_scan(const float *src, float *tgt, int64_t stride, int64_t row_size,
                     int exclusive, int is_prod, int normalize)
{
    const float init = is_prod ? 1.0f : 0.0f;
    float block_total = init;
    float buf[ORC_TILE];
    if (row_size == 0) return;
    if (exclusive) tgt[0] = init; /* utils_scan.cuh:172 */
    for (int64_t c0 = 0; c0 < row_size; c0 += ORC_TILE) {
        for (int k = 0; k < ORC_TILE; ++k)
            buf[k] = (c0 + k < row_size) ? src[(c0 + k) * stride] : init;
        buf[0] = scan_op(buf[0], block_total, is_prod);
        tile_scan(buf, is_prod);
        for (int k = 0; k < ORC_TILE; ++k) {
            int64_t col = c0 + k;
            if (!exclusive) { if (col < row_size) tgt[col * stride] = buf[k]; }
            else            { if (col < row_size - 1) tgt[(col + 1) * stride] = buf[k]; }
        }
        block_total = buf[ORC_TILE - 1];
    }
    if (normalize) { /* utils_scan.cuh:102-110, :229-237 */
        float den = fmaxf(block_total, 1e-10f);
        if (!exclusive) for (int64_t c = 0; c < row_size; ++c) tgt[c * stride] /= den;
        else for (int64_t c = 0; c + 1 < row_size; ++c) tgt[(c + 1) * stride] /= den;
    }
}

/* kind: 0 inclusive_sum, 1 exclusive_sum, 2 inclusive_prod, 3 exclusive_prod.
 * backward!=0 runs the scan over each chunk from its last element to its
 * first, which is what the reverse-iterator launch computes. */
ORC_API int orc_packed_scan(int kind, const int64_t *chunk_starts, const int64_t *chunk_cnts,
                            int64_t n_rays, const float *inputs, int64_t n_edges,
                            int normalize, int backward, float *outputs)
{
    const int exclusive = kind & 1, is_prod = kind >> 1;
    (void)n_edges;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t s = chunk_starts[r], n = chunk_cnts[r];
        if (n <= 0) continue;
        if (!backward) row_scan(inputs + s, outputs + s, 1, n, exclusive, is_prod, normalize);
        else row_scan(inputs + s + n - 1, outputs + s + n - 1, -1, n, exclusive, is_prod, normalize);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* pack_info.  pack.py:38-46: histogram (index_add_) + cumsum.              */
ORC_API int orc_pack_info(const int64_t *ray_indices, int64_t n, int64_t n_rays, int64_t *packed_info)
{
    for (int64_t r = 0; r < n_rays; ++r) { packed_info[2 * r] = 0; packed_info[2 * r + 1] = 0; }
    for (int64_t i = 0; i < n; ++i) packed_info[2 * ray_indices[i] + 1] += 1;
    int64_t acc = 0;
    for (int64_t r = 0; r < n_rays; ++r) { packed_info[2 * r] = acc; acc += packed_info[2 * r + 1]; }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11), as used by curand_init /
 * curand_uniform at pdf.cu:139-144: key = seed, counter = {offset/4 (64 bit),
 * subsequence (64 bit)}, first output word -> (x * 2^-32 + 2^-33).            */
static inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof(c));
}
static inline float philox_uniform_per_ray(uint64_t seed, uint64_t subsequence, uint64_t offset)
{
    uint64_t blk = offset / 4;
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)subsequence, (uint32_t)(subsequence >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    orc_philox4x32_10(ctr, key, out);
    return (float)out[offset & 3] * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}
ORC_API float orc_philox_uniform(uint64_t seed, uint64_t subsequence, uint64_t offset)
{
    return philox_uniform_per_ray(seed, subsequence, offset);
}

/* ------------------------------------------------------------------------ */
/* upper_bound over [start, end), pdf.cu:43-63 */
static inline int64_t upper_bound_f(const float *data, int64_t start, int64_t end, float val)
{
    while (start < end) {
        int64_t mid = start + ((end - start) >> 1);
        if (!(data[mid] > val)) start = mid + 1; else end = mid;
    }
    return start;
}
static inline int64_t clamp64(int64_t v, int64_t lo, int64_t hi)
{
    int64_t m = v < hi ? v : hi;
    return m > lo ? m : lo;
}

/* importance_sampling, int overload (pdf.cu:359-421): every ray gets
 * n_samples samples and n_samples+1 edges, batched outputs.  The input
 * segments are batched (in_starts == NULL, n_edges_per_ray edges each) or
 * packed (in_starts/in_cnts).  Kernels: pdf.cu:98-167 and :169-241.         */
ORC_API int orc_importance_sampling(
    const float *in_vals, const float *cdfs, const int64_t *in_starts, const int64_t *in_cnts,
    int64_t n_rays, int64_t n_edges_per_ray, int64_t n_samples, int stratified,
    uint64_t seed, uint64_t offset, float *out_intervals, float *out_samples)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t base = in_starts ? in_starts[r] : r * n_edges_per_ray;
        int64_t last = base + (in_cnts ? in_cnts[r] : n_edges_per_ray) - 1;
        float u_floor = cdfs[base], u_ceil = cdfs[last];
        float u_step = (u_ceil - u_floor) / n_samples;
        float bias = 0.5f;
        if (stratified) bias = philox_uniform_per_ray(seed, (uint64_t)r, offset);
        float *smp = out_samples + r * n_samples;
        for (int64_t sid = 0; sid < n_samples; ++sid) {
            float u = u_floor + (sid + bias) * u_step;
            int64_t p = upper_bound_f(cdfs, base, last, u);
            int64_t p0 = clamp64(p - 1, base, last), p1 = clamp64(p, base, last);
            float u_lower = cdfs[p0], u_upper = cdfs[p1];
            float t_lower = in_vals[p0], t_upper = in_vals[p1];
            float t;
            if (u_upper - u_lower < 1e-10f) t = (t_lower + t_upper) * 0.5f;
            else {
                float scaling = (t_upper - t_lower) / (u_upper - u_lower);
                t = (u - u_lower) * scaling + t_lower;
            }
            smp[sid] = t;
        }
        /* compute_intervels_kernel, pdf.cu:169-241 (n_samples >= 2) */
        float t_min = in_vals[base], t_max = in_vals[last];
        float *edge = out_intervals + r * (n_samples + 1);
        for (int64_t sid = 0; sid < n_samples; ++sid) {
            float t = smp[sid];
            if (sid == 0) {
                float half_width = (smp[1] - t) * 0.5f;
                edge[0] = fmaxf(t - half_width, t_min);
            } else {
                float t_prev = smp[sid - 1];
                edge[sid] = (t + t_prev) * 0.5f;
                if (sid == n_samples - 1) {
                    float half_width = (t - t_prev) * 0.5f;
                    edge[sid + 1] = fminf(t + half_width, t_max);
                }
            }
        }
    }
    return 0;
}

/* searchsorted, pdf.cu:245-286 + :426-456.  query/key are batched (starts ==
 * NULL) or packed.  Batched query => ray-relative ids.                       */
ORC_API int orc_searchsorted(
    const float *q_vals, const int64_t *q_starts, const int64_t *q_cnts, const int64_t *q_ray_indices,
    int64_t q_n_rays, int64_t q_per_ray, int64_t q_total,
    const float *k_vals, const int64_t *k_starts, const int64_t *k_cnts, int64_t k_per_ray,
    int64_t *ids_left, int64_t *ids_right)
{
    const int q_batched = (q_starts == NULL);
#pragma omp parallel for schedule(static)
    for (int64_t tid = 0; tid < q_total; ++tid) {
        int64_t ray_id;
        if (q_batched) ray_id = tid / q_per_ray;
        else if (q_ray_indices) ray_id = q_ray_indices[tid];
        else { /* binary_search_chunk_id(tid) - 1, pdf.cu:65-80 */
            int64_t s = 0, e = q_n_rays;
            while (s < e) { int64_t m = s + ((e - s) >> 1); if (!(q_starts[m] > tid)) s = m + 1; else e = m; }
            ray_id = s - 1;
        }
        (void)q_cnts;
        int64_t base = k_starts ? k_starts[ray_id] : ray_id * k_per_ray;
        int64_t last = base + (k_cnts ? k_cnts[ray_id] : k_per_ray) - 1;
        int64_t p = upper_bound_f(k_vals, base, last, q_vals[tid]);
        int64_t l = clamp64(p - 1, base, last), rr = clamp64(p, base, last);
        if (q_batched) { ids_left[tid] = l - base; ids_right[tid] = rr - base; }
        else { ids_left[tid] = l; ids_right[tid] = rr; }
    }
    return 0;
}


/* ------------------------------------------------------------------------ */
/* bench.py's CPU baseline: one whole step of the hot path with every stage parallel over rays (the numpy front end's
 * elementwise passes -- sin, exp, boolean indexing over tens of millions of samples -- run on one thread whatever
 * OMP_NUM_THREADS says).  Same semantics as the composition oracle.occgrid_sampling + oracle.rendering + the analytic
 * backward of bench._oracle_step, with the per-ray scans done serially in fp32 instead of in the reference's 32-element
 * tiles (tests/test_oracle_golden.py checks the two against each other: identical samples outside the visibility
 * threshold's guard band, colours and gradients within 1e-5).  Three calls, the two prefix sums in between are the
 * caller's:
 *   orc_step_count    traversal count pass, samples only (grid.cu:405-431)
 *   orc_step_fill     fill pass writing (t_start, t_end) directly (grid.cu:432-471 + occ_grid.py:174-175), the bench
 *                     field's density sigma = scale * 4 (1/2 + 1/2 sin(20 (ts + te))), visibility mask
 *                     T >= early_stop_eps (volrend.py:474-480; alpha_thre = 0) and the kept count per ray
 *   orc_step_render   compaction (occ_grid.py:216-220), weights (volrend.py:358-362), colours with rgb = t_start
 *                     (volrend.py:483-547), and the backward of colours.sum() to the densities      */
ORC_API int orc_step_count(
    int64_t n_rays, const float *rays_o, const float *rays_d, int32_t n_grids, const int32_t *res, const uint8_t *binaries,
    const float *aabbs, const uint8_t *hits, const float *t_sorted, const int64_t *t_indices, const float *near_planes,
    const float *far_planes, float step_size, float cone_angle, int64_t *sm_cnts)
{
    orc_segments iv = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
    orc_segments sm = {NULL, NULL, NULL, NULL, NULL, NULL, sm_cnts, NULL, NULL};
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t tid = 0; tid < n_rays; ++tid)
        traverse_one_ray(tid, rays_o, rays_d, n_grids, res, binaries, aabbs, hits, t_sorted, t_indices, near_planes[tid],
                         far_planes[tid], step_size, cone_angle, -1, 1, &iv, &sm, NULL);
    return 0;
}

ORC_API int orc_step_fill(
    int64_t n_rays, const float *rays_o, const float *rays_d, int32_t n_grids, const int32_t *res, const uint8_t *binaries,
    const float *aabbs, const uint8_t *hits, const float *t_sorted, const int64_t *t_indices, const float *near_planes,
    const float *far_planes, float step_size, float cone_angle, int64_t *sm_starts, int64_t *sm_cnts,
    float sigma_scale, float early_stop_eps,
    int64_t *ray_indices, float *t_starts, float *t_ends, uint8_t *vis, int64_t *kept_cnts)
{
    orc_segments iv = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
    orc_segments sm = {NULL, ray_indices, NULL, NULL, NULL, sm_starts, sm_cnts, t_starts, t_ends};
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t tid = 0; tid < n_rays; ++tid) {
        traverse_one_ray(tid, rays_o, rays_d, n_grids, res, binaries, aabbs, hits, t_sorted, t_indices, near_planes[tid],
                         far_planes[tid], step_size, cone_angle, -1, 0, &iv, &sm, NULL);
        const int64_t s = sm_starts[tid], n = sm_cnts[tid];
        float acc = 0.0f;   /* exclusive sum of sigma * delta */
        int64_t kept = 0;
        for (int64_t k = s; k < s + n; ++k) {
            const float sig = sigma_scale * (4.0f * (0.5f + 0.5f * sinf(20.0f * (t_starts[k] + t_ends[k]))));
            const float T = expf(-acc);
            const uint8_t v = T >= early_stop_eps;
            vis[k] = v; kept += v;
            acc += sig * (t_ends[k] - t_starts[k]);
        }
        kept_cnts[tid] = kept;
    }
    return 0;
}

ORC_API int orc_step_render(
    int64_t n_rays, const int64_t *sm_starts, const int64_t *sm_cnts, const float *t_starts, const float *t_ends,
    const uint8_t *vis, const int64_t *kept_starts, float sigma_scale,
    int64_t *k_ray_indices, float *k_t_starts, float *k_t_ends, float *colors /*[n_rays,3]*/, float *g_sigma)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t tid = 0; tid < n_rays; ++tid) {
        const int64_t s = sm_starts[tid], n = sm_cnts[tid];
        int64_t o = kept_starts[tid];
        const int64_t o0 = o;
        for (int64_t k = s; k < s + n; ++k)
            if (vis[k]) { k_ray_indices[o] = tid; k_t_starts[o] = t_starts[k]; k_t_ends[o] = t_ends[k]; ++o; }
        /* forward: w = T alpha, colour += w * rgb with rgb = (ts, ts, ts) */
        float acc = 0.0f, c = 0.0f;
        for (int64_t k = o0; k < o; ++k) {
            const float ts = k_t_starts[k], te = k_t_ends[k];
            const float sig = sigma_scale * (4.0f * (0.5f + 0.5f * sinf(20.0f * (ts + te))));
            const float x = sig * (te - ts);
            const float T = expf(-acc), alpha = 1.0f - expf(-x);
            c += T * alpha * ts;
            acc += x;
        }
        colors[3 * tid] = colors[3 * tid + 1] = colors[3 * tid + 2] = c;
        /* backward of sum(colours): g_w = 3 ts; g_sigma_k = delta (g_w T (1 - alpha) - sum_{i > k} g_w_i w_i) */
        float suffix = 0.0f;
        float acc_b = acc;
        for (int64_t k = o - 1; k >= o0; --k) {
            const float ts = k_t_starts[k], te = k_t_ends[k];
            const float sig = sigma_scale * (4.0f * (0.5f + 0.5f * sinf(20.0f * (ts + te))));
            const float x = sig * (te - ts);
            acc_b -= x;                       /* sum over j < k (recomputed backwards; forward order would need a stack) */
            const float T = expf(-acc_b), em = expf(-x);
            const float gw = 3.0f * ts;
            g_sigma[k] = (te - ts) * (gw * T * em - suffix);
            suffix += gw * T * (1.0f - em);
        }
    }
    return 0;
}
