/*
 * nerfacc_hip.h -- C ABI of libnerfacc_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for nerfacc's native hot path.  The reference
 * binds its native layer through pybind11/ATen (nerfacc/cuda/csrc/nerfacc.cpp:
 * 100-129, resolved lazily by nerfacc/cuda/__init__.py:8-41); there is no C ABI
 * upstream.  Each entry point below replaces one of those pybind symbols (or
 * one ATen composite the Python layer builds around them) and takes only plain
 * device pointers, sizes and a stream: no torch types cross this line.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - fp32 data, int64 indices/counts, uint8 for bool tensors (torch.bool);
 *   - all work is enqueued on `stream` (hipStream_t passed as void*); nothing
 *     synchronises unless stated;
 *   - return value: 0 on success, otherwise a negative NFA_E* code; the text
 *     of the last error on the calling thread is nfa_last_error();
 *   - outputs are caller-allocated (the Python host layer allocates them with
 *     torch so they live in the caching allocator like the reference's).
 *
 * Citations "ref:" are relative to /root/reference/nerfacc/.
 */
#ifndef NERFACC_HIP_H
#define NERFACC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFA_OK 0
#define NFA_EINVAL (-1)   /* bad argument (shape, null pointer, unsupported mode) */
#define NFA_EHIP (-2)     /* HIP runtime error (launch failure, ...) */

#define NFA_MAX_GRID_LEVELS 8 /* fused in-kernel intersection+sort supports up to this many levels */

typedef void *nfa_stream_t; /* hipStream_t */

const char *nfa_last_error(void);
/* The version of THIS header.  Bumped whenever an entry point changes its arguments or what it expects of them; a caller
 * built against another value must not call the library (nerfacc_amd/_backend.py refuses to load it). */
#define NFA_VERSION 401
int nfa_version(void);          /* NFA_VERSION of the header the library was built from */
/* Knobs of the A/B tests and measurement scripts (which of two equivalent kernels a call takes, tile sizes); value NULL
 * or "" unsets.  Names: NFA_REFILL, NFA_REFILL_ALL, NFA_CONE_STAGED, NFA_SEG_TILE, NFA_WALK_NO_LATTICE.  Results never
 * depend on them.  Not thread-safe: set before the calls that should see them. */
int nfa_set_tuning(const char *name, const char *value);
int nfa_device_arch(char *buf, int buflen); /* gcnArchName of the current device */

/* ------------------------------------------------------------------ utilities */

/* starts[i] = sum(cnts[0..i)), *total = sum(cnts).  `scratch` needs
 * nfa_cumsum_scratch_bytes(n) bytes.  Replaces torch::cumsum in
 * ref: cuda/csrc/include/data_spec.hpp:86-105. */
int64_t nfa_cumsum_scratch_bytes(int64_t n);
int nfa_exclusive_cumsum_i64(const int64_t *cnts, int64_t n, int64_t *starts, int64_t *total,
                             void *scratch, nfa_stream_t stream);
/* Same scan, written as packed_info rows {start, count} (ref: data_specs.py:68-69 stacks them afterwards). */
int nfa_exclusive_cumsum_pairs_i64(const int64_t *cnts, int64_t n, int64_t *packed_info /*[n,2]*/, int64_t *total,
                                   void *scratch, nfa_stream_t stream);
/* The same with a coherence measure for the caller's next batch: total_and_stats[0] = total; over a sample of the input
 * (every 8th block of 2048 counts) [1] += sum over groups of 64 consecutive counts of the group's maximum and [2] += sum
 * of the counts ([1], [2] zeroed by the caller; n <= 4 M, else left untouched).  64 * [1] / [2] is how much longer a
 * wave of 64 neighbouring rays runs than its average ray. */
int nfa_exclusive_cumsum_pairs_stats_i64(const int64_t *cnts, int64_t n, int64_t *packed_info /*[n,2]*/,
                                         int64_t *total_and_stats /*[3]*/, void *scratch, nfa_stream_t stream);

/* packed_info[r] = {start, count} of ray r in a ray-sorted index stream; also
 * reports (flags[0]) whether ray_indices is non-decreasing and in range.
 * ref: pack.py:38-46 (index_add_ histogram + cumsum). scratch as above (n_rays). */
int nfa_pack_info(const int64_t *ray_indices, int64_t n, int64_t n_rays, int64_t *packed_info /*[n_rays,2]*/,
                  int32_t *flags /*[1], set to 1 if unsorted/out of range*/, void *scratch, nfa_stream_t stream);

/* 1 bit per cell copy of a torch.bool grid [n_cells] (derived cache, never serialised). */
int nfa_pack_bits(const uint8_t *binaries, int64_t n_cells, uint32_t *bits /*[(n_cells+31)/32]*/, nfa_stream_t stream);

/* ------------------------------------------------------------------ grid */

/* ref: cuda/csrc/grid.cu:477-519 (ray_aabb_intersect). */
int nfa_ray_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays,
                           const float *aabbs, int32_t n_aabbs, float near_plane, float far_plane,
                           float miss_value, float *t_mins, float *t_maxs, uint8_t *hits,
                           nfa_stream_t stream);

/* The sorted enter / exit events of every ray against the n_aabbs nested boxes, as the reference's Python forms them
 * (grid.py:156-162: ray_aabb_intersect(rays_o, rays_d, aabbs) with near = -inf, far = +inf, miss = +inf, then
 * torch.sort(torch.cat([t_mins, t_maxs], -1), -1)) in one pass: t_sorted f32 [n_rays, 2 n_aabbs], t_indices int64
 * [n_rays, 2 n_aabbs] (k < n_aabbs: entering box k, else leaving box k - n_aabbs), hits bool [n_rays, n_aabbs].
 * Stable: ties keep the order of the concatenation (the reference leaves it unspecified).  n_aabbs <= NFA_MAX_EVENT_LEVELS. */
#define NFA_MAX_EVENT_LEVELS 8
int nfa_ray_events(const float *rays_o, const float *rays_d, int64_t n_rays, const float *aabbs, int32_t n_aabbs,
                   float *t_sorted, int64_t *t_indices, uint8_t *hits, nfa_stream_t stream);

/* One launch of the traversal, ref: cuda/csrc/grid.cu:68-282 + host :320-474.
 *
 * mode 0  count pass  : writes iv_cnts / sm_cnts (whichever is non-null) and terminate_planes.
 * mode 1  fill pass   : reads iv_starts/iv_cnts, sm_starts/sm_cnts, writes the data arrays.
 * mode 2  one pass into over-allocated chunks (reads *_starts as the over-allocated offsets,
 *         honours rays_mask, writes the actual counts to *_cnts and terminate_planes).
 *
 * Intersections: pass t_sorted/t_indices/hits as produced by nfa_ray_aabb_intersect + sort,
 * or all three NULL to have the kernel intersect (and, for n_grids > 1, sort) in registers
 * (n_grids <= NFA_MAX_GRID_LEVELS).
 *
 * Sample-only fast path used by OccGridEstimator.sampling: pass iv_* NULL and
 * sm_t_starts/sm_t_ends non-null; the kernel then emits (t_starts, t_ends, ray_indices)
 * per sample directly instead of interval edges + masks (same values as
 * intervals.vals[is_left] / [is_right], ref: estimators/occ_grid.py:174-177).
 */
typedef struct nfa_traverse_args {
    int64_t n_rays;
    const float *rays_o;        /* [n_rays,3] */
    const float *rays_d;        /* [n_rays,3] */
    const uint8_t *rays_mask;   /* [n_rays] or NULL (mode 2 only) */
    int32_t n_grids;
    int32_t res[3];
    const uint8_t *binaries;    /* [n_grids,res0,res1,res2] torch.bool */
    const float *aabbs;         /* [n_grids,6] */
    const uint8_t *hits;        /* [n_rays,n_grids] or NULL */
    const float *t_sorted;      /* [n_rays,2*n_grids] or NULL */
    const int64_t *t_indices;   /* [n_rays,2*n_grids] or NULL */
    const float *near_planes;   /* [n_rays] */
    const float *far_planes;    /* [n_rays] */
    float step_size;
    float cone_angle;
    int32_t traverse_steps_limit; /* <= 0: none */
    int32_t mode;
    /* intervals (all NULL => not computed) */
    float *iv_vals; int64_t *iv_ray_indices; uint8_t *iv_is_left; uint8_t *iv_is_right;
    int64_t *iv_starts; int64_t *iv_cnts;
    /* samples */
    float *sm_vals; int64_t *sm_ray_indices; uint8_t *sm_is_valid;
    float *sm_t_starts; float *sm_t_ends;       /* sample-only fast path (sm_ray_indices may be NULL: see nfa_fill_ray_indices) */
    int64_t *sm_starts; int64_t *sm_cnts;
    float *terminate_planes;    /* [n_rays] or NULL */
    /* optional ray filter (mode 1): only rays with ray_filter[r] > ray_filter_min are processed */
    const int32_t *ray_filter;
    int32_t ray_filter_min;
    /* optional accelerator: the brick-packed copy of binaries made by nfa_pack_bricks (both or neither).
     * Occupancy is then read from 8-byte bricks (one load per 4x4x4 cells, 1/8 of the bool grid's footprint);
     * results are identical. */
    const uint64_t *bricks;
    const uint32_t *coarse;
    /* Optional DEVICE-side controls, for loops that must not wait for the host (the test-mode loop captured into a hipGraph):
     *   steps_limit_dev  (nfa_traverse_runs, mode 2) the step limit of this call, read from the device instead of
     *                    traverse_steps_limit (which must still be > 0: it selects the kernel); 0 = nothing to do, the
     *                    call leaves every output as it is;
     *   n_listed_dev     (nfa_traverse_runs with ray_order) the number of entries of ray_order to walk (<= n_order);
     *   run_if_nonzero   (nfa_traverse_grids) the launch does nothing when *run_if_nonzero == 0 (the fill pass for rays
     *                    with too many run records, whose count the host has not seen). */
    const int32_t *steps_limit_dev;
    const int64_t *n_listed_dev;
    const int32_t *run_if_nonzero;
} nfa_traverse_args;
int nfa_traverse_grids(const nfa_traverse_args *args, nfa_stream_t stream);

/* Run-length traversal used by the sampler and by traverse_grids when step_size > 0 and cone_angle == 0 (same
 * results as nfa_traverse_grids, one DDA walk instead of the reference's count + fill passes, ref: cuda/csrc/grid.cu:
 * 405-471; coalesced output):
 *   nfa_pack_walk_bits 1-bit-per-cell copy of the torch.bool grid in the walk's own cell order (the coordinate bits of
 *                      x, y, z interleaved: a 128-byte line is a 16 x 8 x 8 block of cells); nfa_walk_bits_words() uint32.
 *   nfa_traverse_runs  per ray: sample count (args->sm_cnts), edge count (args->iv_cnts, optional), terminate plane,
 *                      run count and up to max_runs (<= 32) run records {t_first:f32 | k_start:31, continues_previous:1}
 *                      in runs[max_runs][n_rays] (slot-major: record i of ray r at runs[i * n_rays + r]); rays with
 *                      more runs are counted in overflow_count[0] and must be filled with
 *                      nfa_traverse_grids(mode 1, ray_filter = run_cnts, ray_filter_min = max_runs).
 *                      args->mode: 0 = all rays, 2 = honour rays_mask (+ traverse_steps_limit).  At most 512 cells
 *                      per axis.  near_hint: the value EVERY entry of args->near_planes holds, NaN if there is no such
 *                      value.  With it the march runs on the lattice near, near + step, ... that all rays share (one
 *                      table built on the host from the two numbers); same results.  overflow_count is int32[2]:
 *                      overflow_count[1] != 0 reports rays the table did not serve (a near plane that differs from
 *                      near_hint, a march beyond the tabulated part of the sequence): the outputs are then
 *                      incomplete and the caller repeats the call with near_hint = NaN (the per-ray marcher).
 *   nfa_pack_bricks    torch.bool grid -> 4x4x4-cell 64-bit bricks + 1 bit per brick ("coarse") for the serial
 *                      kernels of nfa_traverse_grids (args->bricks / args->coarse);
 *                      bricks has nfa_bricks_words() entries, coarse (words+31)/32 uint32.
 *   nfa_expand_runs    runs + exclusive cumsum of the counts -> (t_starts, t_ends) or, when t_mids is given,
 *                      the API's sample values (t_start + t_end) / 2; and ray_indices. */
int64_t nfa_bricks_words(int32_t n_grids, const int32_t *res);
int nfa_pack_bricks(const uint8_t *binaries, int32_t n_grids, const int32_t *res, uint64_t *bricks,
                    uint32_t *coarse, nfa_stream_t stream);
int64_t nfa_walk_bits_words(int32_t n_grids, const int32_t *res);
int nfa_pack_walk_bits(const uint8_t *binaries, int32_t n_grids, const int32_t *res, uint32_t *bits, nfa_stream_t stream);
/* ray_order[n_order] (NULL: every ray, in index order): the rays to walk and the lane each gets.  n_order < n_rays walks
 * the listed rays only -- the others' sm_cnts / run_cnts / terminate_planes are left as the caller initialised them (the
 * test-mode loop lists its alive rays, so that dead rays cost no lanes). */
int nfa_traverse_runs(const nfa_traverse_args *args, const uint32_t *bits, int32_t *run_cnts, uint64_t *runs,
                      int32_t max_runs, int32_t *overflow_count, float near_hint, const int32_t *ray_order,
                      int64_t n_order, nfa_stream_t stream);
/* Lane -> ray assignment for nfa_traverse_runs (ray_order; NULL = identity): order[n_rays] = the ray ids sorted into 256
 * bins by the length of the ray's path through box[6] = {min xyz, max xyz} (the outermost grid box), so that the rays a
 * wave walks together are of similar length.  For batches of unrelated rays (training) the walk is ~1.6x faster;
 * image-ordered rays are coherent already.  Everything the walk writes stays indexed by ray id: results do not depend on
 * the order.  scratch: 1024 + n_rays bytes. */
int nfa_bin_rays(const float *rays_o, const float *rays_d, int64_t n_rays, const float *box, int32_t *order,
                 void *scratch, nfa_stream_t stream);
/* Between two iterations of the test-mode loop (ref: examples/utils.py:409-414): mask[r] = opacity[r] <= opacity_max &&
 * packed_info[r].count == n_samples (the ray is not opaque yet and used its whole budget), alive[0 .. *count) = the ids of
 * the rays with mask 1 (ascending inside stretches of 8192 rays, the stretches in arbitrary order: a ray_order for
 * nfa_traverse_runs / nfa_traverse_cone_walk with n_order = *count), *count = their number.  One launch. */
int nfa_alive_rays(const float *opacity, const int64_t *packed_info /*[n_rays,2]*/, int64_t n_samples, float opacity_max,
                   int64_t n_rays, uint8_t *mask, int32_t *alive, int64_t *count, nfa_stream_t stream);
/* The test-mode loop without the host in it (ref examples/utils.py:330-414; nerfacc_amd/marching.py, padded form): the
 * iteration schedule lives in state[8] (int32, zeroed before the first iteration):
 *   state[0] samples per ray of the CURRENT iteration, 0 = the loop is over (no ray alive, or max_samples handed out),
 *   state[1] samples per ray handed out so far, state[2] iterations that did something, state[4..5] (one int64) samples
 *   that entered the accumulation so far.
 * alive_count is int64[2]: [0] the alive rays (written by nfa_testmode_alive, n_rays before the first iteration), [1] its
 * value at the start of the current iteration (the walk's n_listed_dev).
 * nfa_testmode_begin   starts an iteration: state[0] = max(min(n_rays / alive_count[0], 64), min_samples) while rays are alive
 *                      and state[1] < max_samples (then state[1] += state[0]), else 0; alive_count[1] = alive_count[0],
 *                      alive_count[0] = 0; zeroes sm_cnts[n_rays], run_cnts[n_rays] and zero_words[n_zero] (int64: the
 *                      caller's counters of the iteration, e.g. nfa_traverse_runs' overflow_count -- a call with
 *                      steps_limit_dev set does not zero it itself, so that the iteration has no memset node);
 * nfa_testmode_alive   ends it: nfa_alive_rays with n_samples = state[0] (state[0] == 0: nothing happens, count[0] stays 0)
 *                      into count = alive_count, and state[4..5] += the iteration's samples (the last row of packed_info)
 *                      when count_samples != 0. */
int nfa_testmode_begin(int64_t *alive_count, int32_t *state, int64_t n_rays, int32_t min_samples, int32_t max_samples,
                       int64_t *sm_cnts, int32_t *run_cnts, int64_t *zero_words, int32_t n_zero, nfa_stream_t stream);
int nfa_testmode_alive(const float *opacity, const int64_t *packed_info, int32_t *state, float opacity_max, int64_t n_rays,
                       uint8_t *mask, int32_t *alive, int64_t *count, int32_t count_samples, nfa_stream_t stream);
/* The same for nested levels (aabbs[n_grids][6], finest first; e.g. rays that start inside the finest box): the key is the
 * number of cell boundaries the ray crosses from near_plane on, summed over the levels (the length inside level l but
 * outside level l - 1, times sum_k |d_k| res_k / extent_k).  ray_order of nfa_traverse_cone_runs / nfa_traverse_runs. */
int nfa_bin_rays_levels(const float *rays_o, const float *rays_d, int64_t n_rays, const float *aabbs, int32_t n_grids,
                        const int32_t *res, float near_plane, int32_t *order, void *scratch,
                        uint64_t *coherence /* [2] or NULL, zeroed by the caller: += sum over groups of 64 consecutive rays of
                        the group's largest key, += sum of the keys (64 * [0] / [1]: how much longer a wave of neighbouring
                        rays walks than its average ray) */, nfa_stream_t stream);
/* capacity: number of elements the output arrays hold; nothing is written at or beyond it (a caller that allocated the
 * outputs before the total was known to the host re-runs the expansion if the total turns out larger). */
int nfa_expand_runs(int64_t n_rays, float step_size, const int32_t *run_cnts, const uint64_t *runs,
                    int32_t max_runs, const int64_t *packed_info /*[n_rays,2] {start, count}*/, float *t_starts,
                    float *t_ends, float *t_mids, int64_t *ray_indices, int64_t capacity, nfa_stream_t stream);
/* ray_indices[k] = r for every k in [packed_info[r].start, packed_info[r].start + packed_info[r].count): the inverse of
 * nfa_pack_info for contiguous, ray-ordered segments (what the traversal produces), written as coalesced 32-byte
 * stores.  Used after nfa_traverse_grids' direct fill pass with sm_ray_indices == NULL, so that the per-ray serial
 * kernel does not scatter 8-byte indices (ref: grid.cu:247 writes them from the marching loop). */
int nfa_fill_ray_indices(int64_t n_rays, const int64_t *packed_info /*[n_rays,2]*/, int64_t *ray_indices,
                         nfa_stream_t stream);
/* Sampler path for distance-dependent steps (step_size > 0 and cone_angle > 0; ref grid.cu:207-262 recomputes
 * dt = max(step, t * cone) for every sample, so samples are not arithmetic runs):
 *   nfa_traverse_cone_runs  the count pass of nfa_traverse_grids (samples only; args->mode 0, or 2 = honour rays_mask and
 *                           traverse_steps_limit: the test-mode loop's compact form of over_allocate) that also leaves
 *                           run records {t_first:f32 | k_start:31, continues_previous:1} in runs[max_runs][n_rays]
 *                           (slot-major), one per chain of continuous samples and at least one per 64 samples; rays
 *                           with more records are counted in *overflow_count and must be filled with
 *                           nfa_traverse_grids(mode 1, ray_filter = run_cnts, ray_filter_min = max_runs).
 *                           With traverse_steps_limit > 0 the lanes of a wave are not bound to one ray each: a wave owns a
 *                           chunk of consecutive entries of the ray list and hands the next one to a lane whose ray is
 *                           finished (limited walks end after a geometric number of cells; results are the same).
 *                           Environment, for measurements: NFA_REFILL=0 (one ray per lane), NFA_REFILL=<chunk>,<min_busy>;
 *   nfa_expand_cone_runs    records + exclusive cumsum of the counts -> (t_starts, t_ends, ray_indices): every output
 *                           re-runs the serial recurrence t <- t + max(step, t * cone) from its record's t_first (at
 *                           most 63 steps), so the values are bit-identical to the marching loop's, and the second DDA
 *                           walk of the fill pass is replaced by coalesced stores. */
int nfa_traverse_cone_runs(const nfa_traverse_args *args, int32_t *run_cnts, uint64_t *runs, int32_t max_runs,
                           int32_t *overflow_count, const int32_t *ray_order /* as for nfa_traverse_runs, or NULL */,
                           int64_t n_order, nfa_stream_t stream);
/* nfa_traverse_cone_runs over the 1-bit grid copy of nfa_pack_walk_bits (grids of at most 512 cells per axis, as for
 * nfa_traverse_runs): the same outputs, bit for bit, with the constant-step walk's DDA (packed step counters, interleaved
 * bit index, one 4-byte load per cell) -- about half the instructions per cell.  args->bricks is not read.
 * (csrc/walk.hip: cone_walk_kernel, cone_refill_kernel; ref grid.cu:68-282.) */
/* arena (optional; NULL / 0: none): where the records go that do not fit a ray's max_runs slots, instead of sending the ray
 * to the serial fill pass (ref grid.cu:405-471: a second full walk).  arena_capacity entries of 16 bytes {t_first:f32,
 * k_start:31 | continues:1, ray:u32, samples:u32}, a multiple of 16, handed in ZEROED.  Such a ray keeps max_runs - 1 records
 * and a sentinel {NaN, samples so far} in its last slot (nfa_expand_cone_runs skips the rest of its range; run_cnts = max_runs).
 * overflow_count is int32[2]: [0] rays for the fill pass (without an arena: every ray with more records than slots; with one:
 * only rays that found it full), [1] arena entries handed out (16 at a time; entries with samples == 0 are unused) --
 * nfa_expand_cone_arena(arena, min(overflow_count[1], arena_capacity), ...) writes their samples. */
int nfa_traverse_cone_walk(const nfa_traverse_args *args, const uint32_t *bits, int32_t *run_cnts, uint64_t *runs,
                           int32_t max_runs, int32_t *overflow_count, uint32_t *arena, int32_t arena_capacity,
                           const int32_t *ray_order, int64_t n_order, nfa_stream_t stream);
int nfa_expand_cone_arena(const uint32_t *arena, int32_t n_entries, float step_size, float cone_angle,
                          const int64_t *packed_info /*[n_rays,2]*/, float *t_starts, float *t_ends, int64_t *ray_indices,
                          nfa_stream_t stream);
int nfa_expand_cone_runs(int64_t n_rays, float step_size, float cone_angle, const int32_t *run_cnts,
                         const uint64_t *runs, int32_t max_runs, const int64_t *packed_info, float *t_starts,
                         float *t_ends, int64_t *ray_indices, nfa_stream_t stream);
/* The interval stream of the API's traverse_grids from the same run records (ref: grid.cu:219-262; edge
 * values, ray_indices, is_left, is_right); iv_cnts as written by nfa_traverse_runs when args->iv_cnts is set
 * (edges = samples + one leading edge per chain of continuous samples), iv_packed_info its {start, count} rows. */
int nfa_expand_intervals(int64_t n_rays, float step_size, const int32_t *run_cnts, const uint64_t *runs,
                         int32_t max_runs, const int64_t *iv_packed_info /*[n_rays,2]*/, float *vals,
                         int64_t *ray_indices, uint8_t *is_left, uint8_t *is_right, nfa_stream_t stream);

/* ------------------------------------------------------------------ grid maintenance */

/* OccGridEstimator._update (ref: estimators/occ_grid.py:368-404) as kernels.
 *   nfa_grid_cell_points  x[i] = aabb_lo + (coords(indices[i]) + jitter[i]) / res * (aabb_hi - aabb_lo)   (ref :383-391;
 *                         indices are cell ids inside one level, x slowest; aabb = the level's 6 floats on the device)
 *   nfa_grid_ema_update   occs[cell_base + indices[i]] = max(occs[...] * ema_decay, occ[i])   (ref :393-398); a cell
 *                         listed several times gets max(occs * decay, max_i occ_i) -- deterministic, one of the values
 *                         the reference's index_put may leave.  scratch: n floats.
 *   nfa_grid_rebinarize   thre = min(mean(occs[occs >= 0]), occ_thre) reduced on the device, binaries = occs > thre
 *                         (ref :403-404) written as the torch.bool buffer AND as the walk's 1-bit grid copy
 *                         (nfa_pack_walk_bits layout); scratch: nfa_grid_rebinarize_scratch_bytes(), on return its last
 *                         two floats hold {thre, mean}. */
int nfa_grid_cell_points(const int64_t *indices, const float *jitter /*[n,3]*/, int64_t n, const int32_t *res,
                         const float *aabb /*[6]*/, float *x /*[n,3]*/, nfa_stream_t stream);
int nfa_grid_ema_update(float *occs, int64_t cell_base, const int64_t *indices, int64_t n, const float *occ,
                        float ema_decay, float *scratch /*[n]*/, nfa_stream_t stream);
int64_t nfa_grid_rebinarize_scratch_bytes(void);
int nfa_grid_rebinarize(const float *occs, int32_t n_grids, const int32_t *res, float occ_thre, uint8_t *binaries,
                        uint32_t *walk_bits, void *scratch, nfa_stream_t stream);

/* ------------------------------------------------------------------ packed segments */

/* Ownership table for the flat segmented kernels: the element range is cut into n_tiles tiles of
 * tile_elems element offsets, and after every 256 rays (nfa_seg_plan picks tile_elems and n_tiles); a tile OWNS the rays whose chunk starts
 * inside it.  tiles holds nfa_seg_table_rows(n_tiles) int64 pairs: the (n_tiles + 1) {first ray, first element} entries.  Requires
 * contiguous chunks (starts[r+1] == starts[r] + cnts[r]); flags[0] is set to 1 when they are not,
 * in which case the caller must use the *_generic entry points (flags may be NULL for a packed_info the
 * caller knows to be contiguous: no check result, and no memset launch). */
void nfa_seg_plan(int64_t n_elems, int64_t n_rays, int64_t *tile_elems, int64_t *n_tiles);
/* int64 PAIRS the caller allocates for `tiles` (n_tiles + 1 today; ask instead of assuming). */
int64_t nfa_seg_table_rows(int64_t n_tiles);
int nfa_seg_build_tiles(const int64_t *packed_info /*[n_rays,2]*/, int64_t n_rays, int64_t n_elems,
                        int64_t tile_elems, int64_t n_tiles, int64_t *tiles /*[2*(n_tiles+1)]*/,
                        int32_t *flags, nfa_stream_t stream);

/* kind: 0 inclusive_sum 1 exclusive_sum 2 inclusive_prod 3 exclusive_prod.
 * reverse != 0 scans each chunk from its last element to its first: the reverse-iterator
 * launches of ref: cuda/csrc/scan.cu:41-51,100-110 (backward of the sums).
 * ref: cuda/csrc/scan.cu:9-165,217-257; kernels include/utils_scan.cuh:28-263. */
int nfa_packed_scan(int kind, int reverse, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                    int64_t n_rays, int64_t n_elems, const float *inputs, float *outputs,
                    nfa_stream_t stream);
/* Any (start,count) chunks (overlapping, unordered, gaps), one wave per ray; also implements
 * `normalize` (ref: include/utils_scan.cuh:102-110,229-237). */
int nfa_packed_scan_generic(int kind, int reverse, int normalize, const int64_t *packed_info,
                            int64_t n_rays, int64_t n_elems, const float *inputs, float *outputs,
                            nfa_stream_t stream);
/* grad_in = reverse_{incl|excl}_sum(grad_out * outputs) / max(inputs, 1e-10)
 * ref: cuda/csrc/scan.cu:169-214 (inclusive), :259-304 (exclusive). kind: 2 or 3. */
int nfa_packed_prod_backward(int kind, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                             int64_t n_rays, int64_t n_elems, const float *inputs,
                             const float *outputs, const float *grad_outputs, float *grad_inputs,
                             nfa_stream_t stream);

/* Fused transmittance / weights.  ref: volrend.py:256-264,358-362 (density) and
 * :200-206,305-309 (alpha).  Any of weights/trans/alphas may be NULL. */
int nfa_render_from_density_fwd(const float *t_starts, const float *t_ends, const float *sigmas,
                                const float *prefix_trans /*NULL ok*/, const int64_t *packed_info,
                                const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                                float *weights, float *trans, float *alphas, nfa_stream_t stream);
int nfa_render_from_alpha_fwd(const float *alphas, const float *prefix_trans, const int64_t *packed_info,
                              const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                              float *weights, float *trans, nfa_stream_t stream);
/* Backward of the fused density op given the forward's saved trans/alphas:
 *   B_k = g_w_k T_k (1-a_k) + g_a_k (1-a_k) - sum_{i>k}(g_w_i w_i + g_T_i T_i)
 *   grad_sigmas = (t_ends - t_starts) * B,  grad_x = B  (either may be NULL). */
int nfa_render_from_density_bwd(const float *t_starts, const float *t_ends, const float *trans,
                                const float *alphas, const float *g_weights, const float *g_trans,
                                const float *g_alphas, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                                int64_t n_rays, int64_t n_elems, float *grad_sigmas, float *grad_x,
                                nfa_stream_t stream);
/* PropNetEstimator's level loop for batched rows of row_len samples (ref: estimators/prop_net.py:96-107, 139-142):
 * transmittance from the proposal density AND the resampler's CDF rows `1 - cat([T, 0], -1)` ([n_rays, row_len + 1]) in
 * one pass, and the backward from the gradient at those rows (g_T = -g_cdfs[:, :-1]) to the densities.  packed_info /
 * tiles describe n_rays chunks of exactly row_len elements.  Same arithmetic as nfa_render_from_density_{fwd,bwd};
 * alphas may be NULL in both (with only g_T arriving, alpha drops out of the backward). */
int nfa_density_cdf_rows_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const int64_t *packed_info,
                             const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems, int32_t row_len,
                             float *trans, float *alphas, float *cdfs /*[n_rays, row_len + 1]*/, nfa_stream_t stream);
int nfa_density_cdf_rows_bwd(const float *t_starts, const float *t_ends, const float *trans, const float *alphas,
                             const float *g_cdfs /*[n_rays, row_len + 1]*/, const int64_t *packed_info, const int64_t *tiles,
                             int64_t n_tiles, int64_t n_rays, int64_t n_elems, int32_t row_len, float *grad_sigmas,
                             nfa_stream_t stream);
/*   grad_alphas_k = g_w_k T_k - sum_{i>k}(g_w_i w_i + g_T_i T_i) / max(1 - a_k, 1e-10) */
int nfa_render_from_alpha_bwd(const float *alphas, const float *trans, const float *g_weights,
                              const float *g_trans, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                              int64_t n_rays, int64_t n_elems, float *grad_alphas, nfa_stream_t stream);

/* Visibility mask (ref: volrend.py:412-418,474-480) fused with the per-ray visible count that
 * the compaction of OccGridEstimator.sampling needs (ref: estimators/occ_grid.py:216-220).
 * sigmas_or_alphas is sigma when t_starts != NULL, alpha otherwise. vis_cnts may be NULL. */
int nfa_render_visibility(const float *t_starts, const float *t_ends, const float *sigmas_or_alphas,
                          const float *prefix_trans, float early_stop_eps, float alpha_thre,
                          const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays,
                          int64_t n_elems, uint8_t *vis, int64_t *vis_cnts, nfa_stream_t stream);
/* Boolean-mask compaction of (ray_indices, t_starts, t_ends) with known per-ray output offsets (ref: estimators/
 * occ_grid.py:216-220, three boolean-index gathers).  capacity: elements the output arrays hold; nothing is written at or
 * beyond it (a caller that sized them from the previous batch, before this batch's total reached the host, repeats the call
 * into larger arrays when the total turns out larger). */
int nfa_compact_samples(const uint8_t *vis, const float *t_starts, const float *t_ends,
                        const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, const int64_t *out_starts,
                        int64_t n_rays, int64_t n_elems, int64_t *out_ray_indices,
                        float *out_t_starts, float *out_t_ends, int64_t capacity, nfa_stream_t stream);

/* out[r, :] (+)= sum_i w_i * values[i, :] over ray r's chunk, deterministic order.
 * values NULL => D = 1 and out = sum w.  accumulate != 0 adds to `out` (accumulate_along_rays_).
 * ref: volrend.py:483-573 (index_add_). */
int nfa_accumulate_along_rays(const float *weights, const float *values, int32_t D,
                              const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays,
                              int64_t n_elems, int accumulate, float *out, nfa_stream_t stream);
/* Fallback for unsorted ray_indices: atomics, same contract as index_add_. out must be initialised. */
int nfa_accumulate_along_rays_atomic(const float *weights, const float *values, int32_t D,
                                     const int64_t *ray_indices, int64_t n_rays, int64_t n_elems,
                                     float *out, nfa_stream_t stream);
/* g_w[i] = sum_c g_out[ray(i),c] v[i,c];  g_v[i,c] = g_out[ray(i),c] w[i]. */
int nfa_accumulate_along_rays_bwd(const float *weights, const float *values, int32_t D,
                                  const float *g_out, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                                  int64_t n_rays, int64_t n_elems, float *g_weights, float *g_values,
                                  nfa_stream_t stream);
/* The three accumulations of `rendering` in one pass (ref: volrend.py:140-156):
 * colors[r,3] = sum w rgb, opacities[r] = sum w, depths[r] = sum w (ts+te)/2 (un-normalised). */
int nfa_render_accumulate_fwd(const float *weights, const float *rgbs, const float *t_starts,
                              const float *t_ends, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                              int64_t n_rays, int64_t n_elems, float *colors, float *opacities,
                              float *depths, nfa_stream_t stream);
int nfa_render_accumulate_bwd(const float *weights, const float *rgbs, const float *t_starts,
                              const float *t_ends, const float *g_colors, const float *g_opacities,
                              const float *g_depths, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                              int64_t n_rays, int64_t n_elems, float *g_weights, float *g_rgbs,
                              nfa_stream_t stream);

/* `rendering` with a density callback in one pass (ref: volrend.py:109-156 = render_weight_from_density
 * + three accumulate_along_rays): per-sample weights / trans / alphas (each may be NULL) and the per-ray
 * colors[r,3], opacities[r], un-normalised depths[r].  Results are bit-identical to
 * nfa_render_from_density_fwd followed by nfa_render_accumulate_fwd. */
int nfa_render_fused_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                         const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays,
                         int64_t n_elems, float *weights, float *trans, float *alphas, float *colors,
                         float *opacities, float *depths, nfa_stream_t stream);
/* Its backward in one reverse pass.  g_colors[r,3] / g_opacities[r] / g_depths[r]: gradients of the per-ray
 * outputs (NULL = zero); g_weights / g_trans / g_alphas: gradients arriving at the per-sample outputs
 * (NULL = zero).  Writes grad_sigmas[n] and/or grad_rgbs[n,3]. */
int nfa_render_fused_bwd(const float *t_starts, const float *t_ends, const float *rgbs, const float *trans,
                         const float *alphas, const float *g_colors, const float *g_opacities,
                         const float *g_depths, const float *g_weights, const float *g_trans,
                         const float *g_alphas, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                         int64_t n_rays, int64_t n_elems, float *grad_sigmas, float *grad_rgbs,
                         nfa_stream_t stream);

/* One iteration of the test-mode marching loop (ref: examples/utils.py:370-405) in one pass: weights from
 * densities with prefix_trans = 1 - opacities[ray], samples with alpha < alpha_thre dropped (alpha_thre <= 0:
 * none), and colors[r,3] / opacities[r] / depths[r] accumulated in place (+=).  Deterministic (a ray is owned by
 * one wave); replaces render_weight_from_density + boolean masks + 3 accumulate_along_rays_ launches.
 * n_visible (may be NULL): NFA_VISIBLE_SLOTS counters, zeroed by the caller before the first iteration; their sum is the number
 * of samples that passed the threshold so far (the loop's running sample count; one counter would serialise the launch on
 * its address). */
#define NFA_VISIBLE_SLOTS 1024
int nfa_render_step_accumulate(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                               const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays,
                               int64_t n_elems, float alpha_thre, float *colors, float *opacities, float *depths,
                               int64_t *n_visible, nfa_stream_t stream);

/* ------------------------------------------------------------------ pdf */

/* ref: cuda/csrc/pdf.cu:359-421 (int overload): S >= 1 samples + S+1 edges per ray, batched outputs (S == 1, which reads
 * out of bounds upstream (:211), is defined here: the single interval is the ray's whole range).
 * Input segments batched (in_packed_info NULL, n_edges_per_ray each) or packed.
 * stratified: one Philox4x32-10 uniform per ray, subsequence = ray id (pdf.cu:139-144). */
int nfa_importance_sampling(const float *in_vals, const float *cdfs, const int64_t *in_packed_info,
                            int64_t n_rays, int64_t n_edges_per_ray, int64_t n_samples,
                            int stratified, uint64_t seed, uint64_t offset,
                            float *out_intervals /*[n_rays,S+1]*/, float *out_samples /*[n_rays,S] or NULL*/,
                            nfa_stream_t stream);
/* The same resampling with the s -> t mapping of PropNetEstimator.sampling fused in (ref: estimators/prop_net.py:215-229):
 * besides the s-space intervals, the S+1 edges mapped to metric distance are written as contiguous rows
 * t_starts[n_rays,S] / t_ends[n_rays,S].  transform 1 = uniform: t = s*t_b + (1-s)*t_a with (t_a, t_b) = (t_min, t_max);
 * 2 = lindisp: t = 1 / (s*t_b + (1-s)*t_a) with (t_a, t_b) = (1/t_min, 1/t_max).  Same fp32 operation order as the
 * reference's tensor expression. */
int nfa_importance_sampling_t(const float *in_vals, const float *cdfs, const int64_t *in_packed_info, int64_t n_rays,
                              int64_t n_edges_per_ray, int64_t n_samples, int stratified, uint64_t seed,
                              uint64_t offset, float *out_intervals, float *out_samples, int transform, float t_a,
                              float t_b, float *out_t_starts, float *out_t_ends, nfa_stream_t stream);
/* The Tensor-count overload (ref: cuda/csrc/pdf.cu:294-355, non-functional upstream: :324 allocates zero samples): ray r is
 * resampled into sm_packed_info[r].count samples and count + 1 edges (none when count == 0), PACKED at the starts given by
 * sm_packed_info / iv_packed_info ({exclusive cumsum, count} rows made by the caller: iv count = (count + 1) * (count > 0),
 * :341-342).  Per-sample arithmetic of the int overload with n = count; a single sample's interval is the ray's whole
 * range.  iv_is_left / iv_is_right as compute_intervels_kernel sets them (:205-238). */
int nfa_importance_sampling_packed(const float *in_vals, const float *cdfs, const int64_t *in_packed_info, int64_t n_rays,
                                   int64_t n_edges_per_ray, const int64_t *sm_packed_info, const int64_t *iv_packed_info,
                                   int stratified, uint64_t seed, uint64_t offset, float *sm_vals, int64_t *sm_ray_indices,
                                   float *iv_vals, int64_t *iv_ray_indices, uint8_t *iv_is_left, uint8_t *iv_is_right,
                                   nfa_stream_t stream);
/* ref: cuda/csrc/pdf.cu:245-286,426-456. Batched query => ray-relative ids. */
int nfa_searchsorted(const float *q_vals, const int64_t *q_packed_info, const int64_t *q_ray_indices,
                     int64_t q_n_rays, int64_t q_per_ray, int64_t q_total,
                     const float *k_vals, const int64_t *k_packed_info, int64_t k_per_ray,
                     int64_t *ids_left, int64_t *ids_right, nfa_stream_t stream);

/* Interlevel (proposal) loss for batched rays (ref: estimators/prop_net.py:232-256): loss[r,j] =
 * max(w - w_outer, 0)^2 / (w + eps) with w = q_cdfs[r,j+1] - q_cdfs[r,j] and w_outer the key CDF mass between the
 * key edges that enclose query interval j (the reference's searchsorted + gathers), j < n_query_edges - 1.
 * key_ids[r,j] (optional) = left | right << 16, the key edge indices, kept for the backward pass.
 * Backward: g_k_cdfs[r, n_key_edges] (required) and g_q_cdfs[r, n_query_edges] (optional); rows of <= 1024 edges. */
int nfa_pdf_loss_fwd(const float *q_vals, const float *q_cdfs, const float *k_vals, const float *k_cdfs,
                     int64_t n_rays, int32_t n_query_edges, int32_t n_key_edges, float eps, float *loss,
                     uint32_t *key_ids, nfa_stream_t stream);
int nfa_pdf_loss_bwd(const float *q_cdfs, const float *k_cdfs, const uint32_t *key_ids, int64_t n_rays,
                     int32_t n_query_edges, int32_t n_key_edges, float eps, const float *g_loss, float *g_k_cdfs,
                     float *g_q_cdfs, nfa_stream_t stream);
/* The same loss as its MEAN over all (ray, interval) pairs, which is what PropNetEstimator.compute_loss uses (ref:
 * estimators/prop_net.py:151 `.mean()`): the forward writes nfa_pdf_loss_partials(...) per-wave partial sums instead of the
 * loss array (mean = sum of partials / (n_rays * (n_query_edges - 1)); the wave -> rows assignment is fixed, so the sum is
 * deterministic), the backward takes the gradient of the mean (one device float).  Saves the loss array's write + read and
 * its expanded gradient's write + read. */
int64_t nfa_pdf_loss_partials(int64_t n_rays, int32_t n_query_edges, int32_t n_key_edges);
int nfa_pdf_loss_sum_fwd(const float *q_vals, const float *q_cdfs, const float *k_vals, const float *k_cdfs,
                         int64_t n_rays, int32_t n_query_edges, int32_t n_key_edges, float eps, float *partials,
                         uint32_t *key_ids, nfa_stream_t stream);
int nfa_pdf_loss_mean_bwd(const float *q_cdfs, const float *k_cdfs, const uint32_t *key_ids, int64_t n_rays,
                          int32_t n_query_edges, int32_t n_key_edges, float eps, const float *g_mean, float *g_k_cdfs,
                          float *g_q_cdfs, nfa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFACC_HIP_H */
