// segscan.hip -- packed-segment ("flattened ray") kernels: scans, fused transmittance /
// weights forward+backward, visibility + compaction, per-ray accumulation.
//
// Design (MI355X, wave64).  The reference scans one ray per 16 threads through a 32-element
// shared-memory Blelloch tile with ~10 __syncthreads per tile (include/utils_scan.cuh) and
// builds everything else out of ATen elementwise launches.  Here a single engine streams the
// flat sample arrays once:
//   * the flat element range is cut into tiles of NFA_SEG_TILE element offsets and at most SEG_TILE_ROWS rays; a
//     tile OWNS the rays whose chunk starts inside it (ownership table built once per packed_info), so every
//     ray is scanned start-to-end by exactly one wave: no cross-workgroup carry, no atomics,
//     deterministic results that do not depend on the tiling, load balance independent of the ray-length
//     distribution (empty rays included); the tiles that own many rays are taken by a launch's first waves;
//   * a wave walks its element range in 256-element steps, 16 B per lane per array (coalesced
//     1 KiB wave loads/stores); everything that is uniform over the wave (tile bounds, step base,
//     loop control) lives in scalar registers;
//   * segment heads are scattered into a 1 KiB per-wave LDS line from a register-cached window
//     of packed_info rows (16 B/ray read once); the ray id of every element is resolved ONCE per
//     step (lane-local + 6 DPP steps: row_shr 1/2/4/8, row_bcast 15/31), value scans reuse that
//     structure (one DPP add + one select per step and channel) and carry the open ray across
//     steps; waves never synchronise with each other;
//   * a step may run a second scan stage whose inputs are the first stage's results (per-ray
//     totals of w*rgb after the transmittance scan) and a pre-scan hook that sees the ray id
//     (gradient of a per-ray reduction), which is how `rendering()` becomes one pass each way.
// The op-specific arithmetic (exp, alpha, weights, gradients, masks, compaction, per-ray sums)
// is fused into the same pass through small functor structs.
//
// Reverse scans (the reference's reverse-iterator launches, scan.cu:41-51) are the same engine
// with the lane/element order mirrored (DIR = -1).
#include <stdlib.h>

#include "common.hip.h"

namespace nfa {

#ifndef NFA_SEG_OCC_HINTS
#define NFA_SEG_OCC_HINTS 0  /* A/B on one box: 5-6 waves instead of 4-5 for the fused passes is within run-to-run noise */
#endif
#ifndef NFA_SEG_E
#define NFA_SEG_E 4
#endif
// Elements per lane and step (4 or 8, consecutive).  The cross-lane part of a step (head resolution, 6 DPP steps per scan
// channel, the window of packed_info rows) costs the same for 4 or 8 elements per lane, so 8 halves it per element; the
// registers it costs lower the occupancy.  Measured on cfg 2 (fused fwd / bwd / visibility, us, one box): 4 -> 292 / 344 /
// 139, 8 -> 405 / 518 / 168 (156 / 171 / 101 VGPRs: 3 / 2 / 5 waves per SIMD) -- occupancy is worth more than instructions.
constexpr int SE = NFA_SEG_E;
static_assert(SE == 4 || SE == 8, "NFA_SEG_E must be 4 or 8");
constexpr int SQ = SE / 4;              // 16-byte quads per lane
constexpr int SEG_CHUNK = 64 * SE;      // elements per wave step
// non-temporal loads per op (A/B switches): visibility -7 %, fused forward -2 %, but the backward pass that re-reads the
// same arrays later loses as much (+9 us): the step does not move, so they stay off
#ifndef NFA_NT_VIS
#define NFA_NT_VIS false
#endif
#ifndef NFA_NT_FWD
#define NFA_NT_FWD false
#endif
#ifndef NFA_SEG_PIPE
#define NFA_SEG_PIPE 0
#endif
#ifndef NFA_SEG_ANCHOR
#define NFA_SEG_ANCHOR 0
#endif
#ifndef NFA_SEG_TILE_ROWS
#define NFA_SEG_TILE_ROWS 256
#endif
constexpr int64_t SEG_TILE_ROWS = NFA_SEG_TILE_ROWS;
#ifndef NFA_SEG_WINDOW_PREFETCH
#define NFA_SEG_WINDOW_PREFETCH 0   /* measured: neutral on cfg 2 (the row-heavy tiles are dispatched first instead), compaction 10 % slower on cfg 5 */
#endif
#ifndef NFA_SEG_EARLY_FETCH
#define NFA_SEG_EARLY_FETCH 0   /* measured: neutral on cfg 2, compaction 10 % slower on cfg 5 */
#endif
#ifndef NFA_BWD_PIPE
#define NFA_BWD_PIPE 1
#endif
#ifndef NFA_VIS_EXP_FREE
#define NFA_VIS_EXP_FREE 1
#endif
#ifndef NFA_VIS_PIPE
#define NFA_VIS_PIPE 0
#endif
#ifndef NFA_SEG_WAVES_PER_BLOCK
#define NFA_SEG_WAVES_PER_BLOCK 4
#endif
// waves never cooperate, so the workgroup size is only a dispatch granularity; measured on cfg 2 (fused bwd / fwd /
// visibility, us): 1 wave 333 / 271 / 135, 2 waves 335 / 273 / 137, 4 waves 343 / 275 / 138, 8 waves 369 / 288 / 144 --
// but the whole step (and the pipelined loop) is not faster with 1 than with 4, so 4 stays
constexpr int SEG_WAVES_PER_BLOCK = NFA_SEG_WAVES_PER_BLOCK;

// ------------------------------------------------------------------------------------------
// tile ownership table
// tiles[b] = {first ray owned by tile b, its first element}; tiles[n_tiles] is the end sentinel
// {n_rays, end of the last ray}.  Tile b covers element offsets [b*tile_elems, (b+1)*tile_elems).
// (Uniform tiles: cutting the last sixth of the range into quarter-size tiles, to shorten a launch's emptying last
// round of waves, was measured slower -- fused fwd / bwd 290 / 352 -> 304 / 373 us.)
// A tile boundary is also drawn every SEG_TILE_ROWS rays: tile(ray r) = start[r] / tile_elems + r / SEG_TILE_ROWS (monotone in
// r), so that a region of short and empty rays is cut into tiles of at most SEG_TILE_ROWS rows instead of one tile with
// thousands (SS4 (10)): n_tiles = n_elems / tile_elems + n_rays / SEG_TILE_ROWS + 1.
__global__ __launch_bounds__(256) void seg_build_tiles_kernel(const int64_t *__restrict__ packed_info, int64_t n_rays,
                                                              int64_t n_elems, int64_t tile_elems, int64_t n_tiles,
                                                              longlong2 *__restrict__ tiles, int32_t *__restrict__ flags)
{
    // thread r (0..n_rays): ray r is the first ray of every tile b with
    // floor(start[r-1]/T) < b <= floor(start[r]/T); r == n_rays is the sentinel.
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rays;
         r += (int64_t)blockDim.x * gridDim.x) {
        int64_t b_lo, b_hi, e_first;
        bool bad = false;
        if (r < n_rays) {
            const int64_t s = packed_info[2 * r], n = packed_info[2 * r + 1];
            int64_t s_prev = -1;
            if (r > 0) {
                const int64_t ps = packed_info[2 * r - 2], pn = packed_info[2 * r - 1];
                s_prev = ps;
                bad |= (ps + pn != s);
            }
            bad |= (n < 0) || (s < 0) || (s + n > n_elems);
            if (bad) { if (flags) atomicOr(flags, 1); continue; }
            b_lo = (s_prev < 0) ? 0 : s_prev / tile_elems + (r - 1) / SEG_TILE_ROWS + 1;
            b_hi = s / tile_elems + r / SEG_TILE_ROWS;
            e_first = s;
        } else {
            const int64_t s_prev = n_rays > 0 ? packed_info[2 * n_rays - 2] : -1;
            const int64_t n_prev = n_rays > 0 ? packed_info[2 * n_rays - 1] : 0;
            b_lo = (s_prev < 0) ? 0 : s_prev / tile_elems + (n_rays - 1) / SEG_TILE_ROWS + 1;
            b_hi = n_tiles;
            e_first = (s_prev < 0) ? 0 : s_prev + n_prev;
            if (s_prev > n_elems) { if (flags) atomicOr(flags, 1); continue; }
        }
        for (int64_t b = b_lo; b <= b_hi && b <= n_tiles; ++b) tiles[b] = make_longlong2(r, e_first);
    }
}

// (A list of the tiles that own many rays, taken by a launch's first waves so that none of them starts last, was the first
// remedy for row-heavy tiles -- fused fwd 264 -> 237 us on cfg 2 -- and became useless, slightly harmful, once a tile could
// not own more than SEG_TILE_ROWS rays; removed.)
__host__ __device__ inline int64_t seg_table_rows(int64_t n_tiles) { return n_tiles + 1; }

// ------------------------------------------------------------------------------------------
// 16-byte vector helpers (addresses are 16 B aligned when VEC is true)
// Loads are UNCONDITIONAL and RAW: a lane without valid elements reads the step's base address `ps`
// (always inside the array), and the validity selects are applied where the values are consumed
// (sel4).  A load inside an `if`, or a select right behind it, makes the compiler wait for that one
// load on the spot, which serialises the 3-7 array loads of a step (one memory latency each) and
// defeats the one-step-ahead prefetch.
struct F4 { float v[SE]; };  // one lane's elements of a step (the name predates SE)

// Where a lane stands in the current step.  `c` (step base, multiple of 256) and `safe` are wave-uniform
// and live in scalar registers; only `off` is per lane, so element addresses are scalar base + 32-bit
// lane offset and the range checks are 32-bit compares against scalars.
struct Pos {
    int64_t c;      // first element offset of the step
    int32_t off;    // SE * (lane in address order)
    int32_t safe;   // offset from c of an in-range, 16 B aligned quad every lane may read
    bool valid[SE]; // element c + off + j belongs to the tile's element range
    bool any, all;  // over the lane's SE elements
    int32_t d_lo, d_hi;  // the step's valid element offsets [d_lo, d_hi) from c (wave-uniform)
    bool qany[SQ], qall[SQ];  // per 16-byte quad
    __device__ __forceinline__ int64_t p0() const { return c + off; }
};

template <bool VEC, bool NT = false>
__device__ __forceinline__ void ld4(const float *__restrict__ p, const Pos &q, F4 &out)
{
    const float *b = p + q.c;
    if (VEC) {
#pragma unroll
        for (int h = 0; h < SQ; ++h) {
            const nfa_v4f *src = reinterpret_cast<const nfa_v4f *>(b + (q.qany[h] ? q.off + 4 * h : q.safe));
            const nfa_v4f v = NT ? __builtin_nontemporal_load(src) : *src;
            out.v[4 * h] = v.x; out.v[4 * h + 1] = v.y; out.v[4 * h + 2] = v.z; out.v[4 * h + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SE; ++j) out.v[j] = b[q.valid[j] ? q.off + j : q.safe];
    }
}
__device__ __forceinline__ float sel(const F4 &r, int j, const bool valid[SE], float fill) { return valid[j] ? r.v[j] : fill; }

// Full lanes store one 16-byte vector.  The per-element path (a lane straddling a range end) starts with an opaque asm
// statement: without it the compiler if-converts both paths into dwordx3 + dword stores for EVERY lane, which halves the
// store rate.  (It used to go through a volatile pointer instead -- which compiles to system-scope flat stores with an
// s_waitcnt vmcnt(0) after EACH of them: sixteen serial round trips to memory in the first and last step of every tile
// of the fused backward, 22 us of its 344.)
#define NFA_ELEMENTWISE_PATH() asm volatile("; element-wise path" ::: "memory")
#if defined(NFA_VOLATILE_ELEMENTWISE) && NFA_VOLATILE_ELEMENTWISE   /* the old form, for A/B runs */
#define NFA_PV volatile
#else
#define NFA_PV
#endif
template <bool VEC>
__device__ __forceinline__ void store4(float *__restrict__ p, const Pos &q, const float v[SE])
{
    float *b = p + q.c;
#pragma unroll
    for (int h = 0; h < SQ; ++h) {
        if (VEC && q.qall[h]) {
            store_f4(b + q.off + 4 * h, v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
        } else {
            NFA_ELEMENTWISE_PATH();
            NFA_PV float *pv = b;
#pragma unroll
            for (int j = 4 * h; j < 4 * h + 4; ++j)
                if (q.valid[j]) pv[q.off + j] = v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Wave-level segmented scan primitives (256 elements per step: 4 per lane in scan order k).
struct StepHeads {
    int32_t lh[SE];       // most recent head ray id over the lane's elements 0..k (-1: none)
    uint32_t acc;         // bit s: at Hillis-Steele step s (offset 2^s) this lane still accumulates
    bool open_prefix;     // no head in any earlier lane of this step: the carry of previous steps applies
    bool carry_on_lane_end;  // (wave-uniform) the range's last element is that lane's last element
    int32_t carry_lane;   // (wave-uniform) last lane in scan order that holds an element of the tile's range: carries are read
                          // THERE -- a ray that ends at the range's end gets the same scan tree as a ray that ends anywhere else
    int32_t ph;           // ray id in front of the lane's first element (carry folded in)
    int32_t rid[SE], prev_rid[SE];
    bool is_head[SE];
};

template <int S>
__device__ __forceinline__ void heads_step(int32_t &ah, uint32_t &acc)
{
    const int32_t uh = dpp_step<S>(-1, ah);
    // (a lane without a source reads -1: it keeps ah < 0 and its acc bit combines with the identity)
    if (ah < 0) { acc |= 1u << S; ah = uh; }
}

template <int DIR>
__device__ __forceinline__ void resolve_heads(const int32_t hj[SE], const bool valid[SE], int32_t carry_rid, StepHeads &hd)
{
#pragma unroll
    for (int k = 0; k < SE; ++k) {
        const int j = DIR > 0 ? k : SE - 1 - k;
        const int32_t h = valid[j] ? hj[j] : -1;
        hd.is_head[k] = h >= 0;
        hd.lh[k] = (k == 0 || h >= 0) ? h : hd.lh[k - 1 < 0 ? 0 : k - 1];  // most recent head (ids fall in reverse scans)
    }
    int32_t ah = hd.lh[SE - 1];
    uint32_t acc = 0;
    heads_step<0>(ah, acc); heads_step<1>(ah, acc); heads_step<2>(ah, acc);
    heads_step<3>(ah, acc); heads_step<4>(ah, acc); heads_step<5>(ah, acc);
    // keep only the steps at which this lane has a source lane (so that x + identity is never formed:
    // -0.0 would come back as +0.0)
    const int lane = lane_id(), r = lane & 15;
    const uint32_t has_src = (r >= 1 ? 1u : 0u) | (r >= 2 ? 2u : 0u) | (r >= 4 ? 4u : 0u) | (r >= 8 ? 8u : 0u) |
                             ((lane & 16) ? 16u : 0u) | (lane >= 32 ? 32u : 0u);
    hd.acc = acc & has_src;
    int32_t ph = dpp_prev_lane(-1, ah);
    hd.open_prefix = ph < 0;
    if (ph < 0) ph = carry_rid;
    hd.ph = ph;
    int32_t pr = ph;
#pragma unroll
    for (int k = 0; k < SE; ++k) {
        hd.prev_rid[k] = pr;
        hd.rid[k] = hd.lh[k] >= 0 ? hd.lh[k] : ph;
        pr = hd.rid[k];
    }
}

template <int S, int N, class FI, class FC>
__device__ __forceinline__ void values_step(const StepHeads &hd, float av[N], FI identity, FC comb)
{
    const bool take = (hd.acc >> S) & 1u;
#pragma unroll
    for (int ch = 0; ch < N; ++ch) {
        const float uv = dpp_step<S>(identity(ch), av[ch]);
        if (take) av[ch] = comb(ch, uv, av[ch]);
    }
}

// Inclusive segmented scan of x (scan order) given the resolved heads; `prev[k]` is the inclusive
// value of the element before k (in that element's own ray).  `carry` is updated to the state after
// the step's last element.
template <int N, class FI, class FC>
__device__ __forceinline__ void scan_values(const StepHeads &hd, const float x[SE][N], float carry[N], float incl[SE][N],
                                            float prev[SE][N], FI identity, FC comb)
{
    float li[SE][N];
#pragma unroll
    for (int k = 0; k < SE; ++k)
#pragma unroll
        for (int ch = 0; ch < N; ++ch)
            li[k][ch] = (k == 0 || hd.is_head[k]) ? x[k][ch] : comb(ch, li[k - 1 < 0 ? 0 : k - 1][ch], x[k][ch]);
    float av[N];
#pragma unroll
    for (int ch = 0; ch < N; ++ch) av[ch] = li[SE - 1][ch];
    values_step<0, N>(hd, av, identity, comb); values_step<1, N>(hd, av, identity, comb);
    values_step<2, N>(hd, av, identity, comb); values_step<3, N>(hd, av, identity, comb);
    values_step<4, N>(hd, av, identity, comb); values_step<5, N>(hd, av, identity, comb);
    float pv[N];
#pragma unroll
    for (int ch = 0; ch < N; ++ch) {
        pv[ch] = dpp_prev_lane(identity(ch), av[ch]);
        if (hd.open_prefix) pv[ch] = comb(ch, carry[ch], pv[ch]);
    }
#pragma unroll
    for (int k = 0; k < SE; ++k)
#pragma unroll
        for (int ch = 0; ch < N; ++ch) {
            prev[k][ch] = (k == 0) ? pv[ch] : incl[k - 1 < 0 ? 0 : k - 1][ch];
            incl[k][ch] = hd.lh[k] >= 0 ? li[k][ch] : comb(ch, pv[ch], li[k][ch]);
        }
    // The state after the range's last element, formed exactly as the NEXT element would see it (so that a ray's total does
    // not depend on whether the ray ends inside a tile or at its end): behind a lane's last element that is the scanned lane
    // aggregate (what the next lane reads as `pv`), inside a lane the element's inclusive value.
    const bool open_next = hd.open_prefix && hd.lh[SE - 1] < 0;
#pragma unroll
    for (int ch = 0; ch < N; ++ch) {
        const float nxt = open_next ? comb(ch, carry[ch], av[ch]) : av[ch];
        carry[ch] = lane_value(hd.carry_on_lane_end ? nxt : incl[SE - 1][ch], hd.carry_lane);
    }
}

// Per-ray totals: a ray is finished where the next head appears; (prev_rid, prev) there is its id and
// its inclusive total.  A lane has at most 4 such heads and almost always at most one, so the first is
// handled in one predicated block and further ones behind a wave-uniform (rarely taken) branch.
template <int N, class F>
__device__ __forceinline__ void flush_totals(const StepHeads &hd, const float prev[SE][N], F &&done)
{
    bool f[SE];
    int nf = 0;
#pragma unroll
    for (int k = 0; k < SE; ++k) { f[k] = hd.is_head[k] && hd.prev_rid[k] >= 0; nf += f[k] ? 1 : 0; }
    if (nf > 0) {
        int32_t rid = hd.prev_rid[SE - 1];   // the FIRST finished ray of the lane (lowest k wins)
        float t[N];
#pragma unroll
        for (int ch = 0; ch < N; ++ch) t[ch] = prev[SE - 1][ch];
#pragma unroll
        for (int k = SE - 2; k >= 0; --k) {
            if (f[k]) rid = hd.prev_rid[k];
#pragma unroll
            for (int ch = 0; ch < N; ++ch) if (f[k]) t[ch] = prev[k][ch];
        }
        done(rid, t);
    }
    if (__ballot(nf > 1) != 0ull) {
        bool seen = false;
#pragma unroll
        for (int k = 0; k < SE; ++k) {
            if (f[k] && seen) done(hd.prev_rid[k], prev[k]);
            seen = seen || f[k];
        }
    }
}

// Stage-B flavour of scan_values + flush_totals: only the per-ray totals are wanted, so the scan runs
// channel by channel (about ten live registers per channel instead of N x 12) and every finished ray's
// value goes straight to done(rid, ch, total).  xb(k, ch) yields the input of element k (scan order).
template <int N, class FX, class FD>
__device__ __forceinline__ void scan_totals(const StepHeads &hd, FX &&xb, float carry[N], FD &&done)
{
    bool f[SE];
    int nf = 0;
#pragma unroll
    for (int k = 0; k < SE; ++k) { f[k] = hd.is_head[k] && hd.prev_rid[k] >= 0; nf += f[k] ? 1 : 0; }
    int32_t rid1 = hd.prev_rid[SE - 1];   // the first finished ray of the lane
#pragma unroll
    for (int k = SE - 2; k >= 0; --k) if (f[k]) rid1 = hd.prev_rid[k];
    const bool more = __ballot(nf > 1) != 0ull;  // wave-uniform, rare: a lane closing two or more rays
#pragma unroll
    for (int ch = 0; ch < N; ++ch) {
        float li[SE];
#pragma unroll
        for (int k = 0; k < SE; ++k) {
            const float x = xb(k, ch);
            li[k] = (k == 0 || hd.is_head[k]) ? x : li[k - 1 < 0 ? 0 : k - 1] + x;
        }
        float av[1] = {li[SE - 1]};
        auto ident = [](int) { return 0.0f; };
        auto add = [](int, float u, float v) { return u + v; };
        values_step<0, 1>(hd, av, ident, add); values_step<1, 1>(hd, av, ident, add);
        values_step<2, 1>(hd, av, ident, add); values_step<3, 1>(hd, av, ident, add);
        values_step<4, 1>(hd, av, ident, add); values_step<5, 1>(hd, av, ident, add);
        float pv = dpp_prev_lane(0.0f, av[0]);
        if (hd.open_prefix) pv = carry[ch] + pv;
        // prev[k]: inclusive value of the element before k
        float prev[SE], incl = pv;
#pragma unroll
        for (int k = 0; k < SE; ++k) {
            prev[k] = incl;
            incl = hd.lh[k] >= 0 ? li[k] : pv + li[k];
        }
        float t1 = prev[SE - 1];
#pragma unroll
        for (int k = SE - 2; k >= 0; --k) if (f[k]) t1 = prev[k];
        if (nf > 0) done(rid1, ch, t1);
        if (more) {
            bool seen = false;
#pragma unroll
            for (int k = 0; k < SE; ++k) {
                if (f[k] && seen) done(hd.prev_rid[k], ch, prev[k]);
                seen = seen || f[k];
            }
        }
        const float nxt = (hd.open_prefix && hd.lh[SE - 1] < 0) ? carry[ch] + av[0] : av[0];   // (see scan_values)
        carry[ch] = lane_value(hd.carry_on_lane_end ? nxt : incl, hd.carry_lane);
    }
}

__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------------
// The engine.  Op interface (all __device__):
//   static constexpr int NCH;                       scan channels of stage A
//   static constexpr int NCHB;                      channels of the optional stage B (additive; 0 = none):
//                                                   per-ray totals of values derived from stage A's results
//   static constexpr bool NEEDS_RID;                op.pre(j, pos, valid, rid) is called before stage A's
//                                                   inputs are read (the ray id is known before any value scan)
//   static constexpr bool TOTALS;                   op.ray_done(rid, total[NCH]) for EVERY finished ray
//   float identity(int ch); float comb(int ch, float a, float b);   a = earlier in scan order
//   struct Raw;                                      registers filled straight from memory
//   void  fetch(const Pos &q, Raw &r) const;         loads only (unconditional, raw)
//   void  load(const Raw &r, const Pos &q);          derive scan inputs
//   float x(int j, int ch);                          stage-A input of element j (address order)
//   void  emit(int j, int64_t pos, bool valid, bool is_head, int rid, int prev_rid,
//              const float incl[NCH], const float prev[NCH]);
//        incl = inclusive scan value at this element; prev = inclusive value of the previous
//        element in scan order (in that element's own ray; exclusive value = is_head ? identity : prev)
//   float xb(int j, int ch);  void ray_done_b(int rid, int ch, float total);          (stage B)
//   void  store(const Pos &q);
//   void  empty_ray(int rid);
// Everything that is the same for the whole wave (tile bounds, step base, loop control) is kept in
// scalar registers (the tile index is made uniform with readfirstlane).
template <int DIR, int PIPE /* 0: none, 1: next step's loads before this step's compute, 2: before this step's stores */, class Op>
__device__ __forceinline__ void seg_run_tile(Op &op, const int64_t *__restrict__ packed_info,
                                             const longlong2 *__restrict__ tiles, int64_t n_rays, int64_t tile,
                                             int32_t *__restrict__ hid /* LDS, SEG_CHUNK ints, wave private */,
                                             float *__restrict__ ray_lds /* LDS, Op::RAY_LDS_FLOATS floats, wave private */)
{
    constexpr int NCH = Op::NCH;
    const int lane = lane_id();
    const int alane = DIR > 0 ? lane : 63 - lane;  // lane in address order
    const longlong2 t_lo = tiles[tile], t_hi = tiles[tile + 1];
    const int32_t r_lo = __builtin_amdgcn_readfirstlane((int32_t)t_lo.x), r_hi = __builtin_amdgcn_readfirstlane((int32_t)t_hi.x);
    if (r_lo >= r_hi) return;
    const int32_t n_own = r_hi - r_lo;
    // every element this wave touches belongs to one of the rays r_lo .. r_hi - 1: an op may stage their per-ray data
    if constexpr (Op::RAY_LDS_FLOATS > 0) op.tile_begin(r_lo, r_hi, ray_lds);
    const int64_t e_lo = uniform64(t_lo.y), e_hi = uniform64(t_hi.y);  // chunks are contiguous: the last owned ray ends where the next tile begins

    // window of packed_info rows, in walk order v = 0..n_own-1: ray(v) = r_lo + v (fwd) / r_hi-1-v (rev)
    int32_t v_next = 0, win_base = 0;
    // The window FOLLOWING the current one is requested as soon as the current one is in place (nxt_*): a tile in a region
    // of short and empty rays owns hundreds of rows, and every window used to be a dependent load the wave waited for
    // (the bench's 30 % empty rays cost the fused passes 8 %, at 256^3 -- the same number of rays, interleaved with short
    // ones instead of lying in long runs -- 20 %).
    int64_t win_s = 0, win_n = 0, nxt_s = 0, nxt_n = 0;
    int32_t nxt_base = -1;   // window base the prefetched rows belong to (-1: none)
    auto fetch_rows = [&](int32_t base, int64_t &rs, int64_t &rn) {
        const int32_t v = base + lane;
        if (v < n_own) {
            const int64_t ray = DIR > 0 ? (int64_t)r_lo + v : (int64_t)r_hi - 1 - v;
            const longlong2 row = *reinterpret_cast<const longlong2 *>(packed_info + 2 * ray);
            rs = row.x; rn = row.y;
        }
    };
    auto load_window = [&]() {
        if (NFA_SEG_WINDOW_PREFETCH && nxt_base == win_base) { win_s = nxt_s; win_n = nxt_n; }
        else fetch_rows(win_base, win_s, win_n);
        nxt_base = -1;
        if (NFA_SEG_WINDOW_PREFETCH && win_base + 64 < n_own) { nxt_base = win_base + 64; fetch_rows(nxt_base, nxt_s, nxt_n); }
    };
    load_window();

    float carry[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) carry[ch] = op.identity(ch);
    constexpr int NCB = Op::NCHB > 0 ? Op::NCHB : 1;
    float carry_b[NCB];
#pragma unroll
    for (int ch = 0; ch < NCB; ++ch) carry_b[ch] = 0.0f;
    int32_t carry_rid = -1;

    // Where the steps are anchored.  NFA_SEG_ANCHOR = 0: at multiples of the step size (a range of ~1000 elements at an
    // arbitrary offset then touches FIVE 256-element chunks); n > 0 (a multiple of 4, which keeps every 16-byte access
    // aligned; 32 = one 128-byte line of floats): at the tile's own range rounded to n elements -- four steps for a range of
    // <= 1024 - n elements.  Measured (cfg 2): 32 makes the visibility pass 4 % faster (102 vs 106 us) and leaves the fused
    // passes where they are, 4 makes the fused passes 5 % slower (every wave access then straddles one more 128-byte line);
    // with anchored steps a ray is cut into steps relative to its TILE, so results would depend on the tiling again: off.
#if NFA_SEG_ANCHOR == 0
    const int64_t c_first = DIR > 0 ? (e_lo / SEG_CHUNK) * SEG_CHUNK : ((e_hi - 1) / SEG_CHUNK) * SEG_CHUNK;
    const int64_t n_chunks = e_hi > e_lo ? ((e_hi - 1) / SEG_CHUNK - e_lo / SEG_CHUNK + 1) : 0;
#else
    constexpr int64_t AG = NFA_SEG_ANCHOR;
    const int64_t a_lo = (e_lo / AG) * AG, a_hi = ((e_hi + AG - 1) / AG) * AG;
    const int64_t c_first = DIR > 0 ? a_lo : a_hi - SEG_CHUNK;
    const int64_t n_chunks = e_hi > e_lo ? (DIR > 0 ? (e_hi - a_lo + SEG_CHUNK - 1) / SEG_CHUNK : (a_hi - e_lo + SEG_CHUNK - 1) / SEG_CHUNK) : 0;
#endif

    auto chunk_base = [&](int64_t ci) { return c_first + (DIR > 0 ? ci : -ci) * SEG_CHUNK; };
    auto make_pos = [&](int64_t c, Pos &q) {
        // range of the step in element offsets from c, clamped to [0, 256]: scalar
        const int64_t lo64 = e_lo - c, hi64 = e_hi - c;
        const int32_t d_lo = lo64 < 0 ? 0 : (lo64 > SEG_CHUNK ? SEG_CHUNK : (int32_t)lo64);
        const int32_t d_hi = hi64 < 0 ? 0 : (hi64 > SEG_CHUNK ? SEG_CHUNK : (int32_t)hi64);
        q.c = c;
        q.d_lo = d_lo; q.d_hi = d_hi;
        q.off = SE * alane;
        q.safe = (d_lo / 4) * 4;  // first in-range multiple of 4 (every step holds at least one element)
#pragma unroll
        for (int j = 0; j < SE; ++j) q.valid[j] = (q.off + j >= d_lo) && (q.off + j < d_hi);
        q.any = (q.off + SE - 1 >= d_lo) && (q.off < d_hi);
        q.all = (q.off >= d_lo) && (q.off + SE - 1 < d_hi);
#pragma unroll
        for (int h = 0; h < SQ; ++h) {
            q.qany[h] = (q.off + 4 * h + 3 >= d_lo) && (q.off + 4 * h < d_hi);
            q.qall[h] = (q.off + 4 * h >= d_lo) && (q.off + 4 * h + 3 < d_hi);
        }
    };
    // Software pipeline (PIPE): the loads of step i+1 are issued before step i is computed and stored
    // (vmcnt retires in order: loads issued BEFORE the stores can be waited for without them).
    typename Op::Raw raw_cur, raw_next;
    if (n_chunks > 0) {
        Pos q0;
        make_pos(chunk_base(0), q0);
        op.fetch(q0, raw_cur);
    }
    for (int64_t ci = 0; ci < n_chunks; ++ci) {
        const int64_t c = chunk_base(ci);
        if (PIPE == 1 && ci + 1 < n_chunks) {
            Pos qn;
            make_pos(chunk_base(ci + 1), qn);
            op.fetch(qn, raw_next);
        }
        if (PIPE == 0 && ci > 0 && NFA_SEG_EARLY_FETCH) {
            // this step's inputs are requested BEFORE its segment heads are resolved (the window of packed_info rows, the
            // LDS scatter and its read-back: ~150 instructions and two LDS round trips that need none of the data)
            Pos qf;
            make_pos(c, qf);
            op.fetch(qf, raw_cur);
        }
        // ---- segment heads of this chunk -> LDS
#pragma unroll
        for (int h = 0; h < SQ; ++h) *reinterpret_cast<int4 *>(hid + 256 * h + 4 * lane) = make_int4(-1, -1, -1, -1);
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            const int32_t v = win_base + lane;
            const bool live = v >= v_next && v < n_own;
            const int64_t key = DIR > 0 ? win_s : win_s + win_n - 1;
            const bool take = live && (DIR > 0 ? key < c + SEG_CHUNK : key >= c);
            const int32_t ray = DIR > 0 ? r_lo + v : r_hi - 1 - v;
            if (take) {
                if (win_n > 0) hid[(int)(key - c)] = ray;
                else op.empty_ray(ray);
            }
            const int cnt = __builtin_popcountll(__ballot(take));
            v_next += cnt;
            if (v_next == win_base + 64 && v_next < n_own) {
                // (A tile owns at most SEG_TILE_ROWS rows, i.e. a few windows: long runs of empty rays -- the background of an
                //  image, the finished rays of the test-mode loop -- are spread over many tiles and walked in parallel.  The
                //  64-ary search that used to skip such runs inside one tile is gone with the tiles that needed it.)
                win_base = v_next;
                load_window();
                continue;
            }
            break;
        }
        __builtin_amdgcn_wave_barrier();
        int32_t hj[SE];
#pragma unroll
        for (int h = 0; h < SQ; ++h) {
            const int4 h4 = *reinterpret_cast<const int4 *>(hid + SE * alane + 4 * h);
            hj[4 * h] = h4.x; hj[4 * h + 1] = h4.y; hj[4 * h + 2] = h4.z; hj[4 * h + 3] = h4.w;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- this step's data
        Pos q;
        make_pos(c, q);
        if (PIPE == 0 && ci > 0 && !NFA_SEG_EARLY_FETCH) op.fetch(q, raw_cur);   // (the old place, for A/B runs)
        op.load(raw_cur, q);

        // ---- segment structure of this step: ray id of every element (scan order k, address j = DIR>0 ? k : 3-k)
        StepHeads hd;
        resolve_heads<DIR>(hj, q.valid, carry_rid, hd);
        hd.carry_lane = DIR > 0 ? (q.d_hi - 1) / SE : 63 - q.d_lo / SE;
        hd.carry_on_lane_end = DIR > 0 ? (q.d_hi % SE == 0) : (q.d_lo % SE == 0);
        if constexpr (Op::NEEDS_RID) {
#pragma unroll
            for (int k = 0; k < SE; ++k) {
                const int j = DIR > 0 ? k : SE - 1 - k;
                op.pre(j, q.p0() + j, q.valid[j], hd.rid[k]);
            }
            op.store_pre(q);
        }
        // ---- stage A: scan of op.x, results to op.emit
        {
            float xa[SE][NCH], incl[SE][NCH], prev[SE][NCH];
#pragma unroll
            for (int k = 0; k < SE; ++k) {
                const int j = DIR > 0 ? k : SE - 1 - k;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) xa[k][ch] = q.valid[j] ? op.x(j, ch) : op.identity(ch);
            }
            scan_values<NCH>(hd, xa, carry, incl, prev, [&](int ch) { return op.identity(ch); },
                             [&](int ch, float u, float v) { return op.comb(ch, u, v); });
#pragma unroll
            for (int k = 0; k < SE; ++k) {
                const int j = DIR > 0 ? k : SE - 1 - k;
                op.emit(j, q.p0() + j, q.valid[j], hd.is_head[k], hd.rid[k], hd.prev_rid[k], incl[k], prev[k]);
            }
            if constexpr (Op::TOTALS) flush_totals<NCH>(hd, prev, [&](int32_t rid, const float *t) { op.ray_done(rid, t); });
        }
        // ---- stage B (optional): per-ray totals of values derived from stage A's results.  The per-element
        //      outputs are complete after stage A: they are stored first (their registers are free for stage B
        //      and the stores are in flight while it runs).
        if (PIPE == 2 && ci + 1 < n_chunks) {
            // the results are in registers, most temporaries are dead: request the next step's inputs BEFORE the stores
            // (vmcnt retires in order, so the next step can wait for its loads without waiting for these stores)
            Pos qn;
            make_pos(chunk_base(ci + 1), qn);
            op.fetch(qn, raw_next);
        }
        op.store(q);
        if constexpr (Op::NCHB > 0) {
            scan_totals<NCB>(hd,
                             [&](int k, int ch) { const int j = DIR > 0 ? k : SE - 1 - k; return q.valid[j] ? op.xb(j, ch) : 0.0f; },
                             carry_b, [&](int32_t rid, int ch, float t) { op.ray_done_b(rid, ch, t); });
        }
        carry_rid = lane_value(hd.rid[SE - 1], hd.carry_lane);
        if (PIPE) raw_cur = raw_next;
    }
    // remaining owned rays are all empty (their start equals e_hi / e_lo)
    for (;;) {
        const int32_t v = win_base + lane;
        const bool live = v >= v_next && v < n_own;
        if (live && win_n <= 0) op.empty_ray(DIR > 0 ? r_lo + v : r_hi - 1 - v);
        v_next = min(win_base + 64, n_own);
        if (v_next < n_own) { win_base = v_next; load_window(); continue; }
        break;
    }
    if (carry_rid >= 0 && lane == 0) {
        if constexpr (Op::TOTALS) op.ray_done(carry_rid, carry);
        if constexpr (Op::NCHB > 0) {
#pragma unroll
            for (int ch = 0; ch < NCB; ++ch) op.ray_done_b(carry_rid, ch, carry_b[ch]);
        }
    }
}

template <int DIR, int PIPE, class Op>
__global__ __launch_bounds__(64 * SEG_WAVES_PER_BLOCK, Op::MIN_WAVES_PER_EU) void seg_kernel(Op op, const int64_t *__restrict__ packed_info,
                                                                       const longlong2 *__restrict__ tiles,
                                                                       int64_t n_rays, int64_t n_tiles)
{
    __shared__ __attribute__((aligned(16))) int32_t hid_all[SEG_WAVES_PER_BLOCK * SEG_CHUNK];
    constexpr int RL = Op::RAY_LDS_FLOATS > 0 ? Op::RAY_LDS_FLOATS : 4;
    __shared__ __attribute__((aligned(16))) float ray_all[SEG_WAVES_PER_BLOCK * RL];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * SEG_WAVES_PER_BLOCK + wave;
    if (tile >= n_tiles) return;
    seg_run_tile<DIR, PIPE>(op, packed_info, tiles, n_rays, tile, hid_all + wave * SEG_CHUNK, ray_all + wave * RL);
}

template <int DIR, class Op>
static void launch_seg(const Op &op, const int64_t *packed_info, const int64_t *tiles_raw, int64_t n_rays, int64_t n_tiles,
                       hipStream_t s)
{
    const longlong2 *tiles = reinterpret_cast<const longlong2 *>(tiles_raw);
    const unsigned grid = (unsigned)ceil_div64(n_tiles, SEG_WAVES_PER_BLOCK);
    // NFA_SEG_PIPE (compile time): 0 = a step's loads are requested when the step starts; 1 = one step ahead, before the
    // previous step's compute (its registers cost occupancy: slower on every op); 2 = one step ahead, between the previous
    // step's compute and its stores.
    hipLaunchKernelGGL((seg_kernel<DIR, Op::PIPE, Op>), dim3(grid), dim3(64 * SEG_WAVES_PER_BLOCK), 0, s, op, packed_info, tiles,
                       n_rays, n_tiles);
}

// ------------------------------------------------------------------------------------------
// Ops.  `Raw` holds what one step loads (fetched one step ahead by the engine); the per-step
// working registers are members (fully unrolled, register resident).

struct OpBase1 {  // one additive channel
    static constexpr int NCH = 1;
    static constexpr int NCHB = 0;
    static constexpr bool NEEDS_RID = false;
    static constexpr bool TOTALS = false;
    static constexpr int MIN_WAVES_PER_EU = 1;  // occupancy floor asked of the register allocator (1 = none)
    static constexpr int RAY_LDS_FLOATS = 0;    // per-wave LDS floats for per-ray data staged at tile start (tile_begin)
    static constexpr int PIPE = NFA_SEG_PIPE;   // when the next step's loads are requested (seg_run_tile)
    __device__ __forceinline__ float identity(int) const { return 0.0f; }
    __device__ __forceinline__ float comb(int, float a, float b) const { return a + b; }
    __device__ __forceinline__ void ray_done(int, const float *) const {}
    __device__ __forceinline__ void empty_ray(int) const {}
};

// ---- plain scans: scan.cu:9-165 (sum), :127-165 / :217-257 (prod)
template <bool EXCL, bool PROD, bool VEC>
struct ScanOp {
    static constexpr int NCH = 1;
    static constexpr int NCHB = 0;
    static constexpr bool NEEDS_RID = false;
    static constexpr bool TOTALS = false;
    static constexpr int MIN_WAVES_PER_EU = 1;  // occupancy floor asked of the register allocator (1 = none)
    static constexpr int RAY_LDS_FLOATS = 0;    // per-wave LDS floats for per-ray data staged at tile start (tile_begin)
    static constexpr int PIPE = NFA_SEG_PIPE;   // when the next step's loads are requested (seg_run_tile)
    struct Raw { F4 x; };
    const float *in;
    float *out;
    float xin[SE], res[SE];
    __device__ __forceinline__ float identity(int) const { return PROD ? 1.0f : 0.0f; }
    __device__ __forceinline__ float comb(int, float a, float b) const { return PROD ? a * b : a + b; }
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const { ld4<VEC>(in, q, r.x); }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) xin[j] = sel(r.x, j, valid, identity(0));
    }
    __device__ __forceinline__ float x(int j, int) const { return xin[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int, int, const float incl[1], const float prev[1])
    {
        res[j] = EXCL ? (is_head ? identity(0) : prev[0]) : incl[0];
    }
    __device__ __forceinline__ void store(const Pos &q) { store4<VEC>(out, q, res); }
    __device__ __forceinline__ void ray_done(int, const float *) const {}
    __device__ __forceinline__ void empty_ray(int) const {}
};

// ---- prod backward: reverse {incl,excl} sum of g*out, divided by clamp_min(in, 1e-10)
//      scan.cu:169-214, :259-304
template <bool EXCL, bool VEC>
struct ProdBwdOp : OpBase1 {
    struct Raw { F4 o, g, in; };
    const float *in, *outv, *g;
    float *gin;
    float q[SE], den[SE], res[SE];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(outv, q, r.o);
        ld4<VEC>(g, q, r.g);
        ld4<VEC>(in, q, r.in);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) { q[j] = sel(r.g, j, valid, 0.0f) * sel(r.o, j, valid, 0.0f); den[j] = sel(r.in, j, valid, 1.0f); }
    }
    __device__ __forceinline__ float x(int j, int) const { return q[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int, int, const float incl[1], const float prev[1])
    {
        const float sres = EXCL ? (is_head ? 0.0f : prev[0]) : incl[0];
        res[j] = sres / fmaxf(den[j], 1e-10f);
    }
    __device__ __forceinline__ void store(const Pos &q) { store4<VEC>(gin, q, res); }
};

// ---- transmittance / alpha / weights from density, volrend.py:256-264, :358-362
template <bool VEC>
struct DensityFwdOp : OpBase1 {
    struct Raw { F4 a, b, s, pf; };
    const float *ts, *te, *sig, *prefix;
    float *w, *tr, *al;
    // batched rows of row_len samples (PropNetEstimator): the resampler's CDF rows `1 - cat([T, 0])` (row_len + 1 entries,
    // ref estimators/prop_net.py:104-107) written by the same pass; element p of ray r lands at p + r
    float *cdf = nullptr;
    int32_t row_len = 0;
    int32_t crid[SE];
    float xs[SE], pf[SE], rw[SE], rt[SE], ra[SE];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(ts, q, r.a);
        ld4<VEC>(te, q, r.b);
        ld4<VEC>(sig, q, r.s);
        if (prefix) ld4<VEC>(prefix, q, r.pf);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            xs[j] = valid[j] ? r.s.v[j] * (r.b.v[j] - r.a.v[j]) : 0.0f;
            pf[j] = prefix ? r.pf.v[j] : 1.0f;
        }
    }
    __device__ __forceinline__ float x(int j, int) const { return xs[j]; }
    __device__ __forceinline__ void emit(int j, int64_t pos, bool valid, bool is_head, int rid, int, const float *, const float prev[1])
    {
        const float S = is_head ? 0.0f : prev[0];
        float T = expf(-S);
        if (prefix) T *= pf[j];
        const float a = 1.0f - expf(-xs[j]);
        rt[j] = T; ra[j] = a; rw[j] = T * a;
        if (cdf) {
            crid[j] = rid;
            if (valid && is_head) cdf[pos + rid + row_len] = 1.0f;   // the row's last entry, 1 - 0
        }
    }
    __device__ __forceinline__ void store(const Pos &q)
    {
        const bool *valid = q.valid;
        const int64_t p0 = q.p0();
        (void)valid; (void)p0;
        if (w) store4<VEC>(w, q, rw);
        if (tr) store4<VEC>(tr, q, rt);
        if (al) store4<VEC>(al, q, ra);
        if (cdf) {
            // a quad inside one row is one 16-byte store at a 4-byte aligned address (rows are shifted by their index)
            struct __attribute__((packed, aligned(4))) Q4 { float x, y, z, w; };
            float *b = cdf + q.c;
#pragma unroll
            for (int h = 0; h < SQ; ++h) {
                if (q.qall[h] && crid[4 * h] == crid[4 * h + 3]) {
                    Q4 v = {1.0f - rt[4 * h], 1.0f - rt[4 * h + 1], 1.0f - rt[4 * h + 2], 1.0f - rt[4 * h + 3]};
                    *reinterpret_cast<Q4 *>(b + q.off + 4 * h + crid[4 * h]) = v;
                } else {
                    NFA_ELEMENTWISE_PATH();
            NFA_PV float *pv = b;
#pragma unroll
                    for (int j = 4 * h; j < 4 * h + 4; ++j)
                        if (q.valid[j]) pv[q.off + j + crid[j]] = 1.0f - rt[j];
                }
            }
        }
    }
};

// ---- transmittance / weights from alpha, volrend.py:200-206, :305-309
template <bool VEC>
struct AlphaFwdOp {
    static constexpr int NCH = 1;
    static constexpr int NCHB = 0;
    static constexpr bool NEEDS_RID = false;
    static constexpr bool TOTALS = false;
    static constexpr int MIN_WAVES_PER_EU = 1;  // occupancy floor asked of the register allocator (1 = none)
    static constexpr int RAY_LDS_FLOATS = 0;    // per-wave LDS floats for per-ray data staged at tile start (tile_begin)
    static constexpr int PIPE = NFA_SEG_PIPE;   // when the next step's loads are requested (seg_run_tile)
    struct Raw { F4 a, pf; };
    const float *al, *prefix;
    float *w, *tr;
    float a4[SE], pf[SE], rw[SE], rt[SE];
    __device__ __forceinline__ float identity(int) const { return 1.0f; }
    __device__ __forceinline__ float comb(int, float a, float b) const { return a * b; }
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(al, q, r.a);
        if (prefix) ld4<VEC>(prefix, q, r.pf);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) { a4[j] = sel(r.a, j, valid, 0.0f); pf[j] = prefix ? r.pf.v[j] : 1.0f; }
    }
    __device__ __forceinline__ float x(int j, int) const { return 1.0f - a4[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int, int, const float *, const float prev[1])
    {
        float T = is_head ? 1.0f : prev[0];
        if (prefix) T *= pf[j];
        rt[j] = T; rw[j] = T * a4[j];
    }
    __device__ __forceinline__ void store(const Pos &q)
    {
        const bool *valid = q.valid;
        const int64_t p0 = q.p0();
        (void)valid; (void)p0;
        if (w) store4<VEC>(w, q, rw);
        if (tr) store4<VEC>(tr, q, rt);
    }
    __device__ __forceinline__ void ray_done(int, const float *) const {}
    __device__ __forceinline__ void empty_ray(int) const {}
};

// ---- backward of the fused density op (reverse scan), SURVEY App. A.7
template <bool VEC, bool CDF = false /* the gradient arrives at the CDF rows of DensityFwdOp::cdf: g_T[p] = -g_cdf[p + ray] */>
struct DensityBwdOp : OpBase1 {
    static constexpr bool NEEDS_RID = CDF;
    struct Raw { F4 a, b, T, A, gw, gt, ga; };
    const float *ts, *te, *tr, *al, *gw, *gt, *ga;
    const float *gcdf = nullptr;
    float *gsig, *gx;
    float T[SE], A[SE], GW[SE], GA[SE], dlt[SE], q[SE], rs[SE], rx[SE];
    int32_t crid[SE];
    __device__ __forceinline__ void pre(int j, int64_t, bool, int rid) { crid[j] = rid; }
    // (called after the ray ids of all the lane's elements are known: a quad inside one row is one 16-byte load)
    __device__ __forceinline__ void store_pre(const Pos &pq)
    {
        struct __attribute__((packed, aligned(4))) Q4 { float x, y, z, w; };
        const float *b = gcdf + pq.c;
#pragma unroll
        for (int h = 0; h < SQ; ++h) {
            float g4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (pq.qall[h] && crid[4 * h] == crid[4 * h + 3]) {
                const Q4 v = *reinterpret_cast<const Q4 *>(b + pq.off + 4 * h + crid[4 * h]);
                g4[0] = v.x; g4[1] = v.y; g4[2] = v.z; g4[3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (pq.valid[4 * h + j]) g4[j] = b[pq.off + 4 * h + j + crid[4 * h + j]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float GT = -g4[j];
                q[4 * h + j] = GW[4 * h + j] * (T[4 * h + j] * A[4 * h + j]) + GT * T[4 * h + j];
            }
        }
    }
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(ts, q, r.a);
        ld4<VEC>(te, q, r.b);
        ld4<VEC>(tr, q, r.T);
        if (!CDF) ld4<VEC>(al, q, r.A);   // (with only g_T arriving alpha drops out: B = -E, q = g_T T)
        if (gw) ld4<VEC>(gw, q, r.gw);
        if (gt) ld4<VEC>(gt, q, r.gt);
        if (ga) ld4<VEC>(ga, q, r.ga);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            T[j] = sel(r.T, j, valid, 0.0f); A[j] = CDF ? 0.0f : sel(r.A, j, valid, 0.0f);
            GW[j] = (gw && valid[j]) ? r.gw.v[j] : 0.0f;
            const float GT = (gt && valid[j]) ? r.gt.v[j] : 0.0f;
            GA[j] = (ga && valid[j]) ? r.ga.v[j] : 0.0f;
            dlt[j] = r.b.v[j] - r.a.v[j];
            q[j] = GW[j] * (T[j] * A[j]) + GT * T[j];
        }
    }
    __device__ __forceinline__ float x(int j, int) const { return q[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int, int, const float *, const float prev[1])
    {
        const float E = is_head ? 0.0f : prev[0];
        const float om = 1.0f - A[j];
        const float B = GW[j] * T[j] * om + GA[j] * om - E;
        rx[j] = B; rs[j] = dlt[j] * B;
    }
    __device__ __forceinline__ void store(const Pos &q)
    {
        const bool *valid = q.valid;
        const int64_t p0 = q.p0();
        (void)valid; (void)p0;
        if (gsig) store4<VEC>(gsig, q, rs);
        if (gx) store4<VEC>(gx, q, rx);
    }
};

// ---- backward of the fused alpha op: g_a = g_w T - sum_{i>k}(g_w_i w_i + g_T_i T_i) / max(1-a, 1e-10)
template <bool VEC>
struct AlphaBwdOp : OpBase1 {
    struct Raw { F4 T, A, gw, gt; };
    const float *al, *tr, *gw, *gt;
    float *galpha;
    float T[SE], A[SE], GW[SE], q[SE], res[SE];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(tr, q, r.T);
        ld4<VEC>(al, q, r.A);
        if (gw) ld4<VEC>(gw, q, r.gw);
        if (gt) ld4<VEC>(gt, q, r.gt);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            T[j] = sel(r.T, j, valid, 0.0f); A[j] = sel(r.A, j, valid, 0.0f);
            GW[j] = (gw && valid[j]) ? r.gw.v[j] : 0.0f;
            const float GT = (gt && valid[j]) ? r.gt.v[j] : 0.0f;
            q[j] = (GW[j] * A[j] + GT) * T[j];
        }
    }
    __device__ __forceinline__ float x(int j, int) const { return q[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int, int, const float *, const float prev[1])
    {
        const float E = is_head ? 0.0f : prev[0];
        res[j] = GW[j] * T[j] - E / fmaxf(1.0f - A[j], 1e-10f);
    }
    __device__ __forceinline__ void store(const Pos &q) { store4<VEC>(galpha, q, res); }
};

// ---- visibility mask, volrend.py:412-418 / :474-480.  COUNT adds the per-ray number of visible
//      samples (what the sampler's compaction needs) as a stage-B scan of the mask just computed.
template <bool DENSITY, bool VEC, bool COUNT>
struct VisibilityOp {
    static constexpr int NCH = 1;
    static constexpr int NCHB = COUNT ? 1 : 0;
    static constexpr bool NEEDS_RID = false;
    static constexpr bool TOTALS = false;
    static constexpr int MIN_WAVES_PER_EU = 1;  // occupancy floor asked of the register allocator (1 = none)
    static constexpr int RAY_LDS_FLOATS = 0;    // per-wave LDS floats for per-ray data staged at tile start (tile_begin)
    static constexpr int PIPE = NFA_VIS_PIPE;   // (measured per op: 0, 1 and 2 are within noise here)
    struct Raw { F4 s, pf, a, b; };
    const float *ts, *te, *val, *prefix;
    float eps, thre;
    // Density without a prefix: T = exp(-S) >= eps is decided on S (the scanned sum) wherever S is clearly on one side of
    // -ln(eps); only inside a band of a few ulps around it is exp evaluated (same result as evaluating it everywhere: exp
    // is computed with the library's own expf there).  s_lo / s_hi come from the host.
    float s_lo, s_hi;
    uint8_t *vis;
    int64_t *cnts;
    float x0[SE], a4[SE], pf[SE];
    uint8_t m[SE];
    __device__ __forceinline__ float identity(int) const { return DENSITY ? 0.0f : 1.0f; }
    __device__ __forceinline__ float comb(int, float a, float b) const { return DENSITY ? a + b : a * b; }
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC, NFA_NT_VIS>(val, q, r.s);
        if (prefix) ld4<VEC, NFA_NT_VIS>(prefix, q, r.pf);
        if (DENSITY) {
            ld4<VEC, NFA_NT_VIS>(ts, q, r.a);
            ld4<VEC, NFA_NT_VIS>(te, q, r.b);
        }
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            pf[j] = prefix ? r.pf.v[j] : 1.0f;
            const float sv = sel(r.s, j, valid, 0.0f);
            if (DENSITY) { x0[j] = valid[j] ? sv * (r.b.v[j] - r.a.v[j]) : 0.0f; a4[j] = (thre > 0.0f) ? 1.0f - expf(-x0[j]) : 1.0f; }  // alpha only when it is tested
            else { a4[j] = sv; x0[j] = 1.0f - sv; }
        }
    }
    __device__ __forceinline__ float x(int j, int) const { return x0[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool valid, bool is_head, int, int, const float *, const float prev[1])
    {
        bool v;
        if (DENSITY && NFA_VIS_EXP_FREE && !prefix) {
            const float S = is_head ? 0.0f : prev[0];
            v = S <= s_lo;
            const bool band = S > s_lo && S < s_hi;
            if (__ballot(band) != 0ull) {   // wave-uniform, rare
                asm volatile("; transmittance near the threshold" ::: "memory");
                if (band) v = expf(-S) >= eps;
            }
        } else {
            float T = DENSITY ? expf(-(is_head ? 0.0f : prev[0])) : (is_head ? 1.0f : prev[0]);
            if (prefix) T *= pf[j];
            v = T >= eps;
        }
        if (thre > 0.0f) v = v && (a4[j] >= thre);
        m[j] = (valid && v) ? 1 : 0;
    }
    __device__ __forceinline__ float xb(int j, int) const { return (float)m[j]; }
    __device__ __forceinline__ void ray_done_b(int rid, int, float tot) const { cnts[rid] = (int64_t)tot; }
    __device__ __forceinline__ void store(const Pos &q)
    {
        const bool *valid = q.valid;
        const int64_t p0 = q.p0();
        (void)valid; (void)p0;
        uint8_t *b = vis + q.c;
#pragma unroll
        for (int h = 0; h < SQ; ++h) {
            if (VEC && q.qall[h]) {
                *reinterpret_cast<uchar4 *>(b + q.off + 4 * h) = make_uchar4(m[4 * h], m[4 * h + 1], m[4 * h + 2], m[4 * h + 3]);
            } else {
                NFA_ELEMENTWISE_PATH();
                NFA_PV uint8_t *pv = b;
#pragma unroll
                for (int j = 4 * h; j < 4 * h + 4; ++j)
                    if (q.valid[j]) pv[q.off + j] = m[j];
            }
        }
    }
    __device__ __forceinline__ void ray_done(int, const float *) const {}
    __device__ __forceinline__ void empty_ray(int rid) const
    {
        if (COUNT) cnts[rid] = 0;
    }
};

struct U4 { uint32_t w[SQ]; };  // the lane's SE mask bytes, raw
__device__ __forceinline__ void load_mask4(const uint8_t *vis, bool vec, const Pos &q, U4 &m)
{
    const uint8_t *b = vis + q.c;
#pragma unroll
    for (int h = 0; h < SQ; ++h) {
        if (vec) {
            // (written as arithmetic: the plain select became a two-entry table in scratch memory)
            const int32_t o = q.safe + (q.qany[h] ? 1 : 0) * (q.off + 4 * h - q.safe);
            m.w[h] = *reinterpret_cast<const uint32_t *>(b + o);
        } else {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) w |= (uint32_t)b[q.valid[4 * h + j] ? q.off + 4 * h + j : q.safe] << (8 * j);
            m.w[h] = w;
        }
    }
}
__device__ __forceinline__ float mask_sel(const U4 &m, int j, const bool valid[SE])
{
    return (valid[j] && ((m.w[j / 4] >> (8 * (j % 4))) & 0xFFu)) ? 1.0f : 0.0f;
}

// ---- compaction of the visible samples (per-ray output offsets = cumsum of VisibilityOp's counts)
template <bool VEC>
struct CompactOp : OpBase1 {
    // The kept samples of a step go to CONSECUTIVE output positions (out_starts is the running sum of the per-ray counts and
    // a ray's kept samples are numbered by the scan), so the wave packs them in LDS and writes them out as 16-byte vectors,
    // 4 outputs per lane, instead of three predicated 4/8-byte scatters per element.  A caller whose out_starts are not that
    // running sum still gets every sample written (element-wise fallback, decided per step).
    // The output offsets of the tile's rays (consecutive rays) are staged in LDS when the tile starts, like the fused
    // backward's per-ray gradients: per element they would be a gather that depends on the ray id.
    static constexpr int RAY_CAP = 192;
    static constexpr int RAY_LDS_FLOATS = 3 * SEG_CHUNK + 2 * RAY_CAP;
    struct Raw { U4 m; F4 a, b; };
    const uint8_t *vis;
    int vis_vec;
    const float *ts, *te;
    const int64_t *out_starts;
    int64_t *o_ri;
    float *o_ts, *o_te;
    int64_t cap = (int64_t)1 << 62;   // elements the output arrays hold (a caller that sized them before the total was known)
    float *stage = nullptr;
    const int64_t *s_start = nullptr;
    int32_t g_lo = 0, g_n = 0;
    float m[SE], a[SE], b[SE];
    int64_t dst[SE];
    int32_t er[SE];
    float rank[SE];
    __device__ __forceinline__ void tile_begin(int32_t r_lo, int32_t r_hi, float *lds)
    {
        stage = lds;
        int64_t *st = reinterpret_cast<int64_t *>(lds + 3 * SEG_CHUNK);   // (8-byte aligned: the per-wave block is 16-byte aligned)
        s_start = st; g_lo = r_lo; g_n = min(r_hi - r_lo, RAY_CAP);
        __builtin_amdgcn_wave_barrier();
        for (int32_t i = lane_id(); i < g_n; i += 64) st[i] = out_starts[(int64_t)r_lo + i];
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        load_mask4(vis, vis_vec != 0, q, r.m);
        ld4<VEC>(ts, q, r.a);
        ld4<VEC>(te, q, r.b);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) { m[j] = mask_sel(r.m, j, valid); a[j] = r.a.v[j]; b[j] = r.b.v[j]; }
    }
    __device__ __forceinline__ float x(int j, int) const { return m[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int rid, int, const float *, const float prev[1])
    {
        er[j] = rid;
        rank[j] = is_head ? 0.0f : prev[0];   // kept samples of the ray in front of this one
    }
    __device__ __forceinline__ void store(const Pos &)
    {
        struct __attribute__((packed, aligned(4))) Q4 { float x, y, z, w; };
        struct __attribute__((packed, aligned(8))) L2 { int64_t x, y; };
        bool keep[SE];
        bool any = false;
        int n_mine = 0;
#pragma unroll
        for (int j = 0; j < SE; ++j) { keep[j] = m[j] != 0.0f; any = any || keep[j]; n_mine += keep[j] ? 1 : 0; }
        const unsigned long long lanes = __ballot(any);
        if (lanes == 0ull) return;  // wave-uniform
        // output offsets of the elements' rays: the staged ones are four independent LDS reads; rays beyond the stage
        // (a tile owning more than RAY_CAP rays) read global memory behind a wave-uniform branch.  (Written as a plain
        // select of the two sources the compiler forms ONE flat load through a select of the two pointers.)
        bool far = false;
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            const uint32_t slot = (uint32_t)(er[j] - g_lo);
            const bool staged = slot < (uint32_t)g_n;
            dst[j] = s_start[staged ? slot : 0u];
            far = far || (keep[j] && !staged);
        }
        if (__ballot(far) != 0ull) {
            asm volatile("; output offsets beyond the staged rays" ::: "memory");
#pragma unroll
            for (int j = 0; j < SE; ++j)
                if (keep[j] && (uint32_t)(er[j] - g_lo) >= (uint32_t)g_n) dst[j] = out_starts[er[j]];
        }
#pragma unroll
        for (int j = 0; j < SE; ++j) dst[j] = keep[j] ? dst[j] + (int64_t)rank[j] : 0;
        const int lane = lane_id();
        const int first = __builtin_ctzll(lanes), last = 63 - __builtin_clzll(lanes);
        int64_t my_first = dst[SE - 1], my_last = dst[0];
#pragma unroll
        for (int j = SE - 2; j >= 0; --j) if (keep[j]) my_first = dst[j];
#pragma unroll
        for (int j = 1; j < SE; ++j) if (keep[j]) my_last = dst[j];
        const int64_t base = uniform64(__shfl(my_first, first, 64));
        const int64_t span = uniform64(__shfl(my_last, last, 64)) - base + 1;
        // number kept in the step (sum over lanes): equals span exactly when the outputs are consecutive
        int cnt = n_mine;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (span == (int64_t)cnt && span <= SEG_CHUNK) {
            float *s_ts = stage, *s_te = stage + SEG_CHUNK;
            int32_t *s_ri = reinterpret_cast<int32_t *>(stage + 2 * SEG_CHUNK);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < SE; ++j)
                if (keep[j]) {
                    const uint32_t o = (uint32_t)(dst[j] - base);
                    if (o < (uint32_t)SEG_CHUNK) { s_ts[o] = a[j]; s_te[o] = b[j]; s_ri[o] = er[j]; }
                }
            __builtin_amdgcn_wave_barrier();
            const int n = (int)max((int64_t)0, min(span, cap - base));   // nothing at or beyond the capacity
            float *g_ts = o_ts + base, *g_te = o_te + base;
            int64_t *g_ri = o_ri + base;
            for (int o = 4 * lane; o < n; o += 256) {
                if (o + 3 < n) {
                    const Q4 vt = {s_ts[o], s_ts[o + 1], s_ts[o + 2], s_ts[o + 3]};
                    const Q4 ve = {s_te[o], s_te[o + 1], s_te[o + 2], s_te[o + 3]};
                    *reinterpret_cast<Q4 *>(g_ts + o) = vt;
                    *reinterpret_cast<Q4 *>(g_te + o) = ve;
                    const L2 r0 = {(int64_t)s_ri[o], (int64_t)s_ri[o + 1]}, r1 = {(int64_t)s_ri[o + 2], (int64_t)s_ri[o + 3]};
                    *reinterpret_cast<L2 *>(g_ri + o) = r0;
                    *reinterpret_cast<L2 *>(g_ri + o + 2) = r1;
                } else {
                    for (int i = o; i < n; ++i) { g_ts[i] = s_ts[i]; g_te[i] = s_te[i]; g_ri[i] = (int64_t)s_ri[i]; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < SE; ++j)
                if (keep[j] && dst[j] < cap) { o_ri[dst[j]] = er[j]; o_ts[dst[j]] = a[j]; o_te[dst[j]] = b[j]; }
        }
    }
};

// ---- per-ray accumulation of w * values[:, d0:d0+C], volrend.py:532-547
template <int C, bool VEC>
struct AccumOp {
    static constexpr int NCH = C;
    static constexpr int NCHB = 0;
    static constexpr bool NEEDS_RID = false;
    static constexpr bool TOTALS = true;
    static constexpr int MIN_WAVES_PER_EU = 1;
    static constexpr int RAY_LDS_FLOATS = 0;    // per-wave LDS floats for per-ray data staged at tile start (tile_begin)
    static constexpr int PIPE = NFA_SEG_PIPE;   // when the next step's loads are requested (seg_run_tile)
    struct Raw { F4 w; float v[SE][C]; };
    const float *w, *vals;  // vals may be null (C == 1): accumulate w
    int32_t D, d0;
    float *out;
    int accumulate;
    float xv[SE][C];
    __device__ __forceinline__ float identity(int) const { return 0.0f; }
    __device__ __forceinline__ float comb(int, float a, float b) const { return a + b; }
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(w, q, r.w);
        if (vals) {
#pragma unroll
            for (int j = 0; j < SE; ++j)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) r.v[j][ch] = vals[(q.c + (q.valid[j] ? q.off + j : q.safe)) * D + d0 + ch];
        }
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
#pragma unroll
        for (int j = 0; j < SE; ++j)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) xv[j][ch] = valid[j] ? (vals ? r.w.v[j] * r.v[j][ch] : r.w.v[j]) : 0.0f;
    }
    __device__ __forceinline__ float x(int j, int ch) const { return xv[j][ch]; }
    __device__ __forceinline__ void put(int rid, const float *tot) const
    {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            float *o = out + (int64_t)rid * D + d0 + ch;
            *o = accumulate ? *o + tot[ch] : tot[ch];
        }
    }
    __device__ __forceinline__ void emit(int, int64_t, bool, bool, int, int, const float *, const float *) const {}
    __device__ __forceinline__ void store(const Pos &) const {}
    __device__ __forceinline__ void ray_done(int rid, const float tot[C]) const { put(rid, tot); }
    __device__ __forceinline__ void empty_ray(int rid) const
    {
        if (!accumulate)
#pragma unroll
            for (int ch = 0; ch < C; ++ch) out[(int64_t)rid * D + d0 + ch] = 0.0f;
    }
};

template <int C, bool VEC>
struct AccumBwdOp : OpBase1 {
    struct Raw { F4 w, g; float v[SE][C]; };
    const float *w, *vals, *gout;
    int32_t D, d0;
    int first;  // first channel group: g_w is written, later groups add to it
    float *gw, *gv;
    float ww[SE], res[SE], vv[SE][C];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(w, q, r.w);
        if (gw && !first) ld4<VEC>(gw, q, r.g);
        if (vals) {
#pragma unroll
            for (int j = 0; j < SE; ++j)
#pragma unroll
                for (int ch = 0; ch < C; ++ch) r.v[j][ch] = vals[(q.c + (q.valid[j] ? q.off + j : q.safe)) * D + d0 + ch];
        }
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &)
    {
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            ww[j] = r.w.v[j];
            res[j] = (gw && !first) ? r.g.v[j] : 0.0f;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) vv[j][ch] = vals ? r.v[j][ch] : 0.0f;
        }
    }
    __device__ __forceinline__ float x(int, int) const { return 0.0f; }
    __device__ __forceinline__ void emit(int j, int64_t pos, bool valid, bool, int rid, int, const float *, const float *)
    {
        if (!valid) return;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const float go = gout[(int64_t)rid * D + d0 + ch];
            if (vals) {
                res[j] += go * vv[j][ch];
                if (gv) gv[pos * D + d0 + ch] = go * ww[j];
            } else {
                res[j] += go;
            }
        }
    }
    __device__ __forceinline__ void store(const Pos &q)
    {
        const bool *valid = q.valid;
        const int64_t p0 = q.p0();
        (void)valid; (void)p0;
        if (gw) store4<VEC>(gw, q, res);
    }
};

// 4 x rgb (12 consecutive floats at 3*p), raw.  With VEC the 48 bytes are three aligned 16 B loads from
// p0 when all 4 elements are valid, else from the step's base (always inside the array); the few
// lanes that straddle a range end re-read their valid elements one by one in fix_rgb12.
__device__ __forceinline__ void load_rgb12(const float *rgb, bool vec, const Pos &q, float c[3 * SE])
{
    const float *b = rgb + 3 * q.c;
    if (vec) {
#pragma unroll
        for (int h = 0; h < SQ; ++h) {
            const float4 *v = reinterpret_cast<const float4 *>(b + 3 * (q.qall[h] ? q.off + 4 * h : q.safe));
            const float4 q0 = v[0], q1 = v[1], q2 = v[2];
            float *o = c + 12 * h;
            o[0] = q0.x; o[1] = q0.y; o[2] = q0.z; o[3] = q0.w; o[4] = q1.x; o[5] = q1.y;
            o[6] = q1.z; o[7] = q1.w; o[8] = q2.x; o[9] = q2.y; o[10] = q2.z; o[11] = q2.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SE; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) c[3 * j + k] = b[3 * (q.valid[j] ? q.off + j : q.safe) + k];
    }
}
__device__ __forceinline__ void fix_rgb12(const float *rgb, bool vec, const Pos &q, const float raw[3 * SE], float c[3 * SE])
{
#pragma unroll
    for (int k = 0; k < 3 * SE; ++k) c[k] = raw[k];
    if (vec && q.any && !q.all) {  // rare: first / last lane of a range
        const float *b = rgb + 3 * q.c;
#pragma unroll
        for (int j = 0; j < SE; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (!q.qall[j / 4]) c[3 * j + k] = q.valid[j] ? b[3 * (q.off + j) + k] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < SE; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) c[3 * j + k] = q.valid[j] ? c[3 * j + k] : 0.0f;
}

// SE x rgb out: three 16 B stores per full quad, element-wise (see store4) where a range ends
__device__ __forceinline__ void store_rgb12(float *rgb, bool vec, const Pos &q, const float g[3 * SE])
{
    float *b = rgb + 3 * q.c;
#pragma unroll
    for (int h = 0; h < SQ; ++h) {
        if (vec && q.qall[h]) {
            float *o = b + 3 * (q.off + 4 * h);
            const float *s = g + 12 * h;
            store_f4(o, s[0], s[1], s[2], s[3]);
            store_f4(o + 4, s[4], s[5], s[6], s[7]);
            store_f4(o + 8, s[8], s[9], s[10], s[11]);
        } else {
            NFA_ELEMENTWISE_PATH();
            NFA_PV float *pv = b;
#pragma unroll
            for (int j = 4 * h; j < 4 * h + 4; ++j)
                if (q.valid[j]) {
                    pv[3 * (q.off + j)] = g[3 * j]; pv[3 * (q.off + j) + 1] = g[3 * j + 1]; pv[3 * (q.off + j) + 2] = g[3 * j + 2];
                }
        }
    }
}

// ---- the three accumulations of `rendering` fused: colours(3), opacity, depth  (volrend.py:140-151)
template <bool VEC>
struct RenderAccumOp {
    static constexpr int NCH = 5;
    static constexpr int NCHB = 0;
    static constexpr bool NEEDS_RID = false;
    static constexpr bool TOTALS = true;
    static constexpr int MIN_WAVES_PER_EU = 1;
    static constexpr int RAY_LDS_FLOATS = 0;    // per-wave LDS floats for per-ray data staged at tile start (tile_begin)
    static constexpr int PIPE = NFA_SEG_PIPE;   // when the next step's loads are requested (seg_run_tile)
    struct Raw { F4 w, a, b; float c[3 * SE]; };
    const float *w, *rgb, *ts, *te;
    float *colors, *opac, *depth;
    float xv[SE][5];
    __device__ __forceinline__ float identity(int) const { return 0.0f; }
    __device__ __forceinline__ float comb(int, float a, float b) const { return a + b; }
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(w, q, r.w);
        ld4<VEC>(ts, q, r.a);
        ld4<VEC>(te, q, r.b);
        load_rgb12(rgb, VEC, q, r.c);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
        float c[3 * SE];
        fix_rgb12(rgb, VEC, pos, r.c, c);
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            const float ww = sel(r.w, j, valid, 0.0f);
            xv[j][0] = ww * c[3 * j]; xv[j][1] = ww * c[3 * j + 1]; xv[j][2] = ww * c[3 * j + 2];
            xv[j][3] = ww;
            xv[j][4] = ww * ((r.a.v[j] + r.b.v[j]) / 2.0f);
        }
    }
    __device__ __forceinline__ float x(int j, int ch) const { return xv[j][ch]; }
    __device__ __forceinline__ void put(int rid, const float *t) const
    {
        colors[3 * (int64_t)rid] = t[0]; colors[3 * (int64_t)rid + 1] = t[1]; colors[3 * (int64_t)rid + 2] = t[2];
        opac[rid] = t[3]; depth[rid] = t[4];
    }
    __device__ __forceinline__ void emit(int, int64_t, bool, bool, int, int, const float *, const float *) const {}
    __device__ __forceinline__ void store(const Pos &) const {}
    __device__ __forceinline__ void ray_done(int rid, const float t[5]) const { put(rid, t); }
    __device__ __forceinline__ void empty_ray(int rid) const
    {
        const float z[5] = {0, 0, 0, 0, 0};
        put(rid, z);
    }
};

template <bool VEC>
struct RenderAccumBwdOp : OpBase1 {
    struct Raw { F4 w, a, b; float c[3 * SE]; };
    const float *w, *rgb, *ts, *te, *gc, *go, *gd;
    float *gw, *grgb;
    float ww[SE], mid[SE], res[SE], c[3 * SE], gr[3 * SE];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(w, q, r.w);
        ld4<VEC>(ts, q, r.a);
        ld4<VEC>(te, q, r.b);
        load_rgb12(rgb, VEC, q, r.c);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
        (void)valid;
#pragma unroll
        for (int j = 0; j < SE; ++j) { ww[j] = r.w.v[j]; mid[j] = (r.a.v[j] + r.b.v[j]) / 2.0f; }
        fix_rgb12(rgb, VEC, pos, r.c, c);
    }
    __device__ __forceinline__ float x(int, int) const { return 0.0f; }
    __device__ __forceinline__ void emit(int j, int64_t, bool valid, bool, int rid, int, const float *, const float *)
    {
        float g = 0.0f;
        gr[3 * j] = gr[3 * j + 1] = gr[3 * j + 2] = 0.0f;
        if (valid) {
            if (gc) {
                const float g0 = gc[3 * (int64_t)rid], g1 = gc[3 * (int64_t)rid + 1], g2 = gc[3 * (int64_t)rid + 2];
                g += g0 * c[3 * j] + g1 * c[3 * j + 1] + g2 * c[3 * j + 2];
                gr[3 * j] = g0 * ww[j]; gr[3 * j + 1] = g1 * ww[j]; gr[3 * j + 2] = g2 * ww[j];
            }
            if (go) g += go[rid];
            if (gd) g += gd[rid] * mid[j];
        }
        res[j] = g;
    }
    __device__ __forceinline__ void store(const Pos &q)
    {
        const bool *valid = q.valid;
        const int64_t p0 = q.p0();
        (void)valid; (void)p0;
        if (gw) store4<VEC>(gw, q, res);
        if (grgb) store_rgb12(grgb, VEC, q, gr);
    }
};

// ---- `rendering` with a density callback in ONE pass (volrend.py:109-151): stage A scans
//      sigma*delta into transmittance -> (w, T, alpha); stage B scans w*rgb, w, w*mid into the per-ray
//      colour / opacity / un-normalised depth.  Bit-identical to DensityFwdOp followed by
//      RenderAccumOp (same expressions, same scan tree), 12 B/sample less traffic.
template <bool VEC>
struct RenderFusedFwdOp : OpBase1 {
    static constexpr int NCHB = 5;
    static constexpr int MIN_WAVES_PER_EU = NFA_SEG_OCC_HINTS ? 6 : 1;  // 81 VGPRs without the hint: one over the 6-wave budget
    struct Raw { F4 a, b, s; float c[3 * SE]; };
    const float *ts, *te, *sig, *rgb;
    float *w, *tr, *al, *colors, *opac, *depth;
    float xs[SE], mid[SE], rw[SE], rt[SE], ra[SE], c[3 * SE];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC, NFA_NT_FWD>(ts, q, r.a);
        ld4<VEC, NFA_NT_FWD>(te, q, r.b);
        ld4<VEC, NFA_NT_FWD>(sig, q, r.s);
        load_rgb12(rgb, VEC, q, r.c);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
        fix_rgb12(rgb, VEC, pos, r.c, c);
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            xs[j] = valid[j] ? r.s.v[j] * (r.b.v[j] - r.a.v[j]) : 0.0f;
            mid[j] = (r.a.v[j] + r.b.v[j]) / 2.0f;
        }
    }
    __device__ __forceinline__ float x(int j, int) const { return xs[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int, int, const float *, const float prev[1])
    {
        const float S = is_head ? 0.0f : prev[0];
        const float T = expf(-S);
        const float a = 1.0f - expf(-xs[j]);
        rt[j] = T; ra[j] = a; rw[j] = T * a;
    }
    __device__ __forceinline__ float xb(int j, int ch) const
    {
        return ch < 3 ? rw[j] * c[3 * j + ch] : (ch == 3 ? rw[j] : rw[j] * mid[j]);
    }
    __device__ __forceinline__ void put(int rid, const float *t) const
    {
        colors[3 * (int64_t)rid] = t[0]; colors[3 * (int64_t)rid + 1] = t[1]; colors[3 * (int64_t)rid + 2] = t[2];
        opac[rid] = t[3]; depth[rid] = t[4];
    }
    __device__ __forceinline__ void ray_done_b(int rid, int ch, float t) const
    {
        if (ch < 3) colors[3 * (int64_t)rid + ch] = t;
        else if (ch == 3) opac[rid] = t;
        else depth[rid] = t;
    }
    __device__ __forceinline__ void empty_ray(int rid) const
    {
        const float z[5] = {0, 0, 0, 0, 0};
        put(rid, z);
    }
    __device__ __forceinline__ void store(const Pos &q)
    {
        const bool *valid = q.valid;
        const int64_t p0 = q.p0();
        (void)valid; (void)p0;
        if (w) store4<VEC>(w, q, rw);
        if (tr) store4<VEC>(tr, q, rt);
        if (al) store4<VEC>(al, q, ra);
    }
};

// ---- one iteration of the test-mode marching loop (ref: examples/utils.py:370-405) in one pass: weights with
//      prefix_trans = 1 - opacity[ray] (the ray id is known before the scan), samples with alpha < alpha_thre dropped,
//      and rgb / opacity / depth accumulated IN PLACE into the per-ray image buffers.  Nothing per sample is written.
//      A ray is owned by one wave, which reads its old opacity for all the ray's samples before it adds the total.
template <bool VEC>
struct RenderStepOp : OpBase1 {
    static constexpr int NCHB = 5;
    static constexpr bool NEEDS_RID = true;
    struct Raw { F4 a, b, s; float c[3 * SE]; };
    const float *ts, *te, *sig, *rgb;
    float thre;
    float *colors, *opac, *depth;  // [R,3], [R], [R] in/out
    unsigned long long *n_visible;  // [NFA_VISIBLE_SLOTS] += samples that pass the alpha threshold (the loop's sample count), or null
    float xs[SE], mid[SE], pf[SE], rw[SE], c[3 * SE];
    bool keep[SE];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(ts, q, r.a);
        ld4<VEC>(te, q, r.b);
        ld4<VEC>(sig, q, r.s);
        load_rgb12(rgb, VEC, q, r.c);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
        fix_rgb12(rgb, VEC, pos, r.c, c);
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            xs[j] = valid[j] ? r.s.v[j] * (r.b.v[j] - r.a.v[j]) : 0.0f;
            mid[j] = (r.a.v[j] + r.b.v[j]) / 2.0f;
        }
    }
    __device__ __forceinline__ void pre(int j, int64_t, bool valid, int rid) { pf[j] = valid ? 1.0f - opac[rid] : 1.0f; }
    __device__ __forceinline__ void store_pre(const Pos &) const {}
    __device__ __forceinline__ float x(int j, int) const { return xs[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool valid, bool is_head, int, int, const float *, const float prev[1])
    {
        const float S = is_head ? 0.0f : prev[0];
        const float T = expf(-S) * pf[j];
        const float a = 1.0f - expf(-xs[j]);
        keep[j] = valid && !(thre > 0.0f && !(a >= thre));
        rw[j] = keep[j] ? T * a : 0.0f;
    }
    __device__ __forceinline__ float xb(int j, int ch) const
    {
        return ch < 3 ? rw[j] * c[3 * j + ch] : (ch == 3 ? rw[j] : rw[j] * mid[j]);
    }
    __device__ __forceinline__ void ray_done_b(int rid, int ch, float t) const
    {
        if (ch < 3) colors[3 * (int64_t)rid + ch] += t;
        else if (ch == 3) opac[rid] += t;
        else depth[rid] += t;
    }
    __device__ __forceinline__ void store(const Pos &) const
    {
        if (n_visible) {
            int c = 0;
#pragma unroll
            for (int j = 0; j < SE; ++j) c += __builtin_popcountll(__ballot(keep[j]));
            // One counter per step was one ADDRESS for the whole launch: atomics on one address are served one after the
            // other by its L2 channel (~15 ns each), and 33 k steps (8 M samples) made that 0.5 ms -- the whole pass.  The count
            // is spread over NFA_VISIBLE_SLOTS counters by wave; the caller adds them up.
            if (c > 0 && lane_id() == 0)
                atomicAdd(n_visible + ((blockIdx.x * SEG_WAVES_PER_BLOCK + (threadIdx.x >> 6)) & (NFA_VISIBLE_SLOTS - 1)), (unsigned long long)c);
        }
    }
};

// ---- its backward in one reverse pass: the gradient of the three accumulations w.r.t. w is formed
//      from the per-ray output gradients (needs the ray id before the scan: NEEDS_RID), added to the
//      gradients arriving at extras' weights / trans / alphas, and pushed through the transmittance
//      chain.  Same expressions as RenderAccumBwdOp followed by DensityBwdOp.
#ifndef NFA_BWD_RAY_CAP
#define NFA_BWD_RAY_CAP 192   // 0: no staging (A/B switch)
#endif
template <bool VEC, bool EXTRA /* gradients arrive at weights / trans / alphas too */>
struct RenderFusedBwdOp : OpBase1 {
    static constexpr bool NEEDS_RID = true;
    static constexpr int MIN_WAVES_PER_EU = (NFA_SEG_OCC_HINTS && !EXTRA) ? 5 : 1;  // 98 VGPRs without the hint: two over the 5-wave budget
    // The per-ray output gradients (colour, opacity, depth: 5 floats per ray) are needed per ELEMENT, after the ray id is
    // known: as global gathers they were a second, dependent memory latency in every step (12 gather instructions per lane
    // and step; SQ counters: the pass waits 80 % of its wave-cycles, VALU 36 % busy).  The rays of a tile are consecutive,
    // so the wave stages their gradients in LDS with coalesced loads at tile start (up to RAY_CAP rays, the rest falls
    // back to the gathers) and the per-element reads are LDS reads.
    static constexpr int RAY_CAP = NFA_BWD_RAY_CAP;
    static constexpr int RAY_LDS_FLOATS = 8 * RAY_CAP;   // {g_r, g_g, g_b, g_opacity, g_depth, -, -, -} per ray
    static constexpr int PIPE = NFA_BWD_PIPE;   // measured per op: a full step ahead is 2-4 % faster here (313 -> 300 us), 2 % slower on the forward pass
    struct Raw { F4 a, b, T, A, gw, gt, ga; float c[3 * SE]; };
    const float *ts, *te, *rgb, *tr, *al, *gc, *go, *gd, *gw, *gt, *ga;
    float *gsig, *grgb;
    const float *g_lds = nullptr;
    int32_t g_lo = 0, g_n = 0;
    __device__ __forceinline__ void tile_begin(int32_t r_lo, int32_t r_hi, float *lds)
    {
        // (the pass runs from the tile's last ray to its first: when the tile owns more than RAY_CAP rays the stage holds
        //  the LAST ones, which most of its elements belong to)
        g_lds = lds; g_n = min(r_hi - r_lo, RAY_CAP); g_lo = r_hi - g_n;
        __builtin_amdgcn_wave_barrier();
        for (int32_t i = lane_id(); i < g_n; i += 64) {
            const int64_t r = (int64_t)g_lo + i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gc) { v.x = gc[3 * r]; v.y = gc[3 * r + 1]; v.z = gc[3 * r + 2]; }
            if (go) v.w = go[r];
            *reinterpret_cast<float4 *>(lds + 8 * i) = v;
            lds[8 * i + 4] = gd ? gd[r] : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
    }
    float T[SE], A[SE], GW[SE], GT[SE], GA[SE], dlt[SE], mid[SE], q[SE], rs[SE], c[3 * SE], gr[3 * SE];
    __device__ __forceinline__ void fetch(const Pos &q, Raw &r) const
    {
        ld4<VEC>(ts, q, r.a);
        ld4<VEC>(te, q, r.b);
        ld4<VEC>(tr, q, r.T);
        ld4<VEC>(al, q, r.A);
        if (EXTRA) {
            if (gw) ld4<VEC>(gw, q, r.gw);
            if (gt) ld4<VEC>(gt, q, r.gt);
            if (ga) ld4<VEC>(ga, q, r.ga);
        }
        load_rgb12(rgb, VEC, q, r.c);
    }
    __device__ __forceinline__ void load(const Raw &r, const Pos &pos)
    {
        const bool *valid = pos.valid;
        fix_rgb12(rgb, VEC, pos, r.c, c);
#pragma unroll
        for (int j = 0; j < SE; ++j) {
            T[j] = sel(r.T, j, valid, 0.0f); A[j] = sel(r.A, j, valid, 0.0f);
            GW[j] = (EXTRA && gw && valid[j]) ? r.gw.v[j] : 0.0f;
            GT[j] = (EXTRA && gt && valid[j]) ? r.gt.v[j] : 0.0f;
            GA[j] = (EXTRA && ga && valid[j]) ? r.ga.v[j] : 0.0f;
            dlt[j] = r.b.v[j] - r.a.v[j];
            mid[j] = (r.a.v[j] + r.b.v[j]) / 2.0f;
        }
    }
    __device__ __forceinline__ void pre(int j, int64_t, bool valid, int rid)
    {
        float g = 0.0f;
        gr[3 * j] = gr[3 * j + 1] = gr[3 * j + 2] = 0.0f;
        const float wj = T[j] * A[j];
        if (valid) {
            const uint32_t slot = (uint32_t)(rid - g_lo);
            float g0, g1, g2, g3, g4;
            if (slot < (uint32_t)g_n) {   // staged (the usual case)
                const float4 v = *reinterpret_cast<const float4 *>(g_lds + 8 * slot);
                g0 = v.x; g1 = v.y; g2 = v.z; g3 = v.w; g4 = g_lds[8 * slot + 4];
            } else {
                g0 = gc ? gc[3 * (int64_t)rid] : 0.0f; g1 = gc ? gc[3 * (int64_t)rid + 1] : 0.0f; g2 = gc ? gc[3 * (int64_t)rid + 2] : 0.0f;
                g3 = go ? go[rid] : 0.0f; g4 = gd ? gd[rid] : 0.0f;
            }
            if (gc) {
                g += g0 * c[3 * j] + g1 * c[3 * j + 1] + g2 * c[3 * j + 2];
                gr[3 * j] = g0 * wj; gr[3 * j + 1] = g1 * wj; gr[3 * j + 2] = g2 * wj;
            }
            if (go) g += g3;
            if (gd) g += g4 * mid[j];
        }
        GW[j] = g + GW[j];
        q[j] = GW[j] * wj + GT[j] * T[j];
    }
    // g_rgb is complete before the scan: stored first (frees its registers, stores in flight during the scan)
    __device__ __forceinline__ void store_pre(const Pos &pq)
    {
        if (grgb) store_rgb12(grgb, VEC, pq, gr);
    }
    __device__ __forceinline__ float x(int j, int) const { return q[j]; }
    __device__ __forceinline__ void emit(int j, int64_t, bool, bool is_head, int, int, const float *, const float prev[1])
    {
        const float E = is_head ? 0.0f : prev[0];
        const float om = 1.0f - A[j];
        const float Bv = GW[j] * T[j] * om + GA[j] * om - E;
        rs[j] = dlt[j] * Bv;
    }
    __device__ __forceinline__ void store(const Pos &pq)
    {
        if (gsig) store4<VEC>(gsig, pq, rs);
    }
};

// ------------------------------------------------------------------------------------------
// Generic fallback: arbitrary (start, count) chunks, one wave per ray (semantics of
// include/utils_scan.cuh incl. `normalize`).
template <bool EXCL, bool PROD>
__global__ __launch_bounds__(256) void generic_scan_kernel(const int64_t *__restrict__ packed_info, int64_t n_rays,
                                                           const float *__restrict__ in, float *__restrict__ out,
                                                           int reverse, int normalize)
{
    const int lane = lane_id();
    const float init = PROD ? 1.0f : 0.0f;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n_rays;
         r += ((int64_t)blockDim.x * gridDim.x) >> 6) {
        const int64_t s = packed_info[2 * r], n = packed_info[2 * r + 1];
        if (n <= 0) continue;
        float den = 1.0f;
        if (normalize) {  // utils_scan.cuh:102-110 / :229-237: divide by the row's inclusive total
            float tot = init;
            for (int64_t k = lane; k < n; k += 64) tot = PROD ? tot * in[s + k] : tot + in[s + k];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float u = __shfl_xor(tot, off, 64);
                tot = PROD ? tot * u : tot + u;
            }
            den = fmaxf(tot, 1e-10f);
        }
        float carry = init;
        for (int64_t c = 0; c < n; c += 64) {
            const int64_t k = c + lane;
            const int64_t pos = reverse ? s + n - 1 - k : s + k;
            float v = k < n ? in[pos] : init;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float u = __shfl_up(v, off, 64);
                if (lane >= off) v = PROD ? u * v : u + v;
            }
            v = PROD ? carry * v : carry + v;
            float prevv = __shfl_up(v, 1, 64);
            if (lane == 0) prevv = carry;
            if (k < n) {
                float o = EXCL ? prevv : v;
                if (normalize && !(EXCL && k == 0)) o /= den;
                out[pos] = o;
            }
            carry = __shfl(v, 63, 64);
        }
    }
}

__global__ __launch_bounds__(256) void accumulate_atomic_kernel(const float *__restrict__ w, const float *__restrict__ vals,
                                                                int32_t D, const int64_t *__restrict__ ri, int64_t n_rays,
                                                                int64_t n, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * D; i += (int64_t)blockDim.x * gridDim.x) {
        const int64_t e = i / D;
        const int32_t ch = (int32_t)(i - e * D);
        const int64_t r = ri[e];
        if (r < 0 || r >= n_rays) continue;
        const float v = vals ? w[e] * vals[i] : w[e];
        atomicAdd(out + r * D + ch, v);
    }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
template <typename... P>
static inline bool all_aligned16(P... p) { return (aligned16(p) && ...); }

}  // namespace nfa

using namespace nfa;

#define SEG_COMMON_CHECKS(name)                                                                         \
    NFA_REQUIRE(n_rays >= 0 && n_elems >= 0, name ": negative size");                                   \
    NFA_REQUIRE(n_rays < ((int64_t)1 << 31) - 64, name ": too many rays");                              \
    if (n_elems == 0 && n_rays == 0) return NFA_OK;                                                      \
    NFA_REQUIRE(packed_info && tiles && n_tiles >= 1, name ": packed_info/tiles is null")

extern "C" {

void nfa_seg_plan(int64_t n_elems, int64_t n_rays, int64_t *tile_elems, int64_t *n_tiles)
{
    // One wave per tile.  Measured on MI355X (scripts/sweep_seg.sh, 32 M samples): 1024-element tiles
    // (4 steps per wave, ~120 waves per CU) are fastest; longer tiles lose to the tail of the last
    // wave round, shorter ones to the per-tile prologue.  A tile also ends after SEG_TILE_ROWS rays.
    const int64_t t_env = tuning_env("NFA_SEG_TILE") ? atoll(tuning_env("NFA_SEG_TILE")) : 0;  // tuning knob (multiple of 4), read once
    const int64_t t = t_env > 0 ? t_env : 1024;
    *tile_elems = t;
    *n_tiles = n_elems / t + (n_rays > 0 ? n_rays : 0) / SEG_TILE_ROWS + 1;
}

int64_t nfa_seg_table_rows(int64_t n_tiles) { return seg_table_rows(n_tiles); }

int nfa_seg_build_tiles(const int64_t *packed_info, int64_t n_rays, int64_t n_elems, int64_t tile_elems, int64_t n_tiles,
                        int64_t *tiles, int32_t *flags, nfa_stream_t stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_elems >= 0 && tiles, "seg_build_tiles: bad arguments");
    NFA_REQUIRE(n_rays == 0 || packed_info, "seg_build_tiles: packed_info is null");
    NFA_REQUIRE(n_rays < ((int64_t)1 << 31) - 64, "seg_build_tiles: too many rays");
    NFA_REQUIRE(tile_elems >= 64 && tile_elems % 4 == 0 && n_tiles == n_elems / tile_elems + n_rays / SEG_TILE_ROWS + 1,
                "seg_build_tiles: tile_elems must be a multiple of 4 (>= 64) and n_tiles what nfa_seg_plan returns for (n_elems, n_rays)");
    hipStream_t s = as_stream(stream);
    if (flags && hipMemsetAsync(flags, 0, sizeof(int32_t), s) != hipSuccess) { set_error("seg_build_tiles: memset failed"); return NFA_EHIP; }
    hipLaunchKernelGGL(seg_build_tiles_kernel, dim3(grid_1d(n_rays + 1, 256)), dim3(256), 0, s, packed_info, n_rays,
                       n_elems, tile_elems, n_tiles, reinterpret_cast<longlong2 *>(tiles), flags);
    NFA_CHECK_LAUNCH("seg_build_tiles");
    return NFA_OK;
}

int nfa_packed_scan(int kind, int reverse, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays,
                    int64_t n_elems, const float *inputs, float *outputs, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("packed_scan");
    NFA_REQUIRE(kind >= 0 && kind <= 3, "packed_scan: kind must be 0..3");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(inputs && outputs, "packed_scan: null data pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(inputs, outputs);
#define NFA_SCAN_CASE(EX, PR)                                                                             \
    do {                                                                                                  \
        if (vec) { ScanOp<EX, PR, true> op; op.in = inputs; op.out = outputs;                            \
            if (reverse) launch_seg<-1>(op, packed_info, tiles, n_rays, n_tiles, s);                      \
            else launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); }                             \
        else { ScanOp<EX, PR, false> op; op.in = inputs; op.out = outputs;                               \
            if (reverse) launch_seg<-1>(op, packed_info, tiles, n_rays, n_tiles, s);                      \
            else launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); }                             \
    } while (0)
    switch (kind) {
        case 0: NFA_SCAN_CASE(false, false); break;
        case 1: NFA_SCAN_CASE(true, false); break;
        case 2: NFA_SCAN_CASE(false, true); break;
        default: NFA_SCAN_CASE(true, true); break;
    }
#undef NFA_SCAN_CASE
    NFA_CHECK_LAUNCH("packed_scan");
    return NFA_OK;
}

int nfa_packed_scan_generic(int kind, int reverse, int normalize, const int64_t *packed_info, int64_t n_rays,
                            int64_t n_elems, const float *inputs, float *outputs, nfa_stream_t stream)
{
    NFA_REQUIRE(kind >= 0 && kind <= 3 && n_rays >= 0 && n_elems >= 0, "packed_scan_generic: bad arguments");
    if (n_elems == 0 || n_rays == 0) return NFA_OK;
    NFA_REQUIRE(packed_info && inputs && outputs, "packed_scan_generic: null pointer");
    hipStream_t s = as_stream(stream);
    const unsigned grid = grid_1d(n_rays * 64, 256, 1 << 16);
    switch (kind) {
        case 0: hipLaunchKernelGGL((generic_scan_kernel<false, false>), dim3(grid), dim3(256), 0, s, packed_info, n_rays, inputs, outputs, reverse, normalize); break;
        case 1: hipLaunchKernelGGL((generic_scan_kernel<true, false>), dim3(grid), dim3(256), 0, s, packed_info, n_rays, inputs, outputs, reverse, normalize); break;
        case 2: hipLaunchKernelGGL((generic_scan_kernel<false, true>), dim3(grid), dim3(256), 0, s, packed_info, n_rays, inputs, outputs, reverse, normalize); break;
        default: hipLaunchKernelGGL((generic_scan_kernel<true, true>), dim3(grid), dim3(256), 0, s, packed_info, n_rays, inputs, outputs, reverse, normalize); break;
    }
    NFA_CHECK_LAUNCH("packed_scan_generic");
    return NFA_OK;
}

int nfa_packed_prod_backward(int kind, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays,
                             int64_t n_elems, const float *inputs, const float *outputs, const float *grad_outputs,
                             float *grad_inputs, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("packed_prod_backward");
    NFA_REQUIRE(kind == 2 || kind == 3, "packed_prod_backward: kind must be 2 or 3");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(inputs && outputs && grad_outputs && grad_inputs, "packed_prod_backward: null data pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(inputs, outputs, grad_outputs, grad_inputs);
#define NFA_PB(EX, V)                                                                                      \
    do { ProdBwdOp<EX, V> op; op.in = inputs; op.outv = outputs; op.g = grad_outputs; op.gin = grad_inputs; \
         launch_seg<-1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (kind == 2) { if (vec) NFA_PB(false, true); else NFA_PB(false, false); }
    else           { if (vec) NFA_PB(true, true); else NFA_PB(true, false); }
#undef NFA_PB
    NFA_CHECK_LAUNCH("packed_prod_backward");
    return NFA_OK;
}

int nfa_render_from_density_fwd(const float *t_starts, const float *t_ends, const float *sigmas,
                                const float *prefix_trans, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                                int64_t n_rays, int64_t n_elems, float *weights, float *trans, float *alphas,
                                nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_from_density_fwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(t_starts && t_ends && sigmas, "render_from_density_fwd: null input");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends, sigmas, prefix_trans, weights, trans, alphas);
#define NFA_DF(V)                                                                                          \
    do { DensityFwdOp<V> op; op.ts = t_starts; op.te = t_ends; op.sig = sigmas; op.prefix = prefix_trans;   \
         op.w = weights; op.tr = trans; op.al = alphas; launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_DF(true); else NFA_DF(false);
#undef NFA_DF
    NFA_CHECK_LAUNCH("render_from_density_fwd");
    return NFA_OK;
}

int nfa_render_from_alpha_fwd(const float *alphas, const float *prefix_trans, const int64_t *packed_info,
                              const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems, float *weights, float *trans,
                              nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_from_alpha_fwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(alphas, "render_from_alpha_fwd: null input");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(alphas, prefix_trans, weights, trans);
#define NFA_AF(V)                                                                                          \
    do { AlphaFwdOp<V> op; op.al = alphas; op.prefix = prefix_trans; op.w = weights; op.tr = trans;          \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_AF(true); else NFA_AF(false);
#undef NFA_AF
    NFA_CHECK_LAUNCH("render_from_alpha_fwd");
    return NFA_OK;
}

int nfa_render_from_density_bwd(const float *t_starts, const float *t_ends, const float *trans, const float *alphas,
                                const float *g_weights, const float *g_trans, const float *g_alphas,
                                const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                                float *grad_sigmas, float *grad_x, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_from_density_bwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(t_starts && t_ends && trans && alphas && (grad_sigmas || grad_x), "render_from_density_bwd: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends, trans, alphas, g_weights, g_trans, g_alphas, grad_sigmas, grad_x);
#define NFA_DB(V)                                                                                          \
    do { DensityBwdOp<V> op; op.ts = t_starts; op.te = t_ends; op.tr = trans; op.al = alphas; op.gw = g_weights; \
         op.gt = g_trans; op.ga = g_alphas; op.gsig = grad_sigmas; op.gx = grad_x;                            \
         launch_seg<-1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_DB(true); else NFA_DB(false);
#undef NFA_DB
    NFA_CHECK_LAUNCH("render_from_density_bwd");
    return NFA_OK;
}

int nfa_density_cdf_rows_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const int64_t *packed_info,
                             const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems, int32_t row_len,
                             float *trans, float *alphas, float *cdfs, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("density_cdf_rows_fwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(t_starts && t_ends && sigmas && cdfs, "density_cdf_rows_fwd: null pointer");
    NFA_REQUIRE(row_len >= 1 && n_rays * (int64_t)row_len == n_elems, "density_cdf_rows_fwd: n_elems must be n_rays * row_len");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends, sigmas, trans, alphas);
#define NFA_DC(V)                                                                                          \
    do { DensityFwdOp<V> op; op.ts = t_starts; op.te = t_ends; op.sig = sigmas; op.prefix = nullptr;         \
         op.w = nullptr; op.tr = trans; op.al = alphas; op.cdf = cdfs; op.row_len = row_len;                \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_DC(true); else NFA_DC(false);
#undef NFA_DC
    NFA_CHECK_LAUNCH("density_cdf_rows_fwd");
    return NFA_OK;
}

int nfa_density_cdf_rows_bwd(const float *t_starts, const float *t_ends, const float *trans, const float *alphas,
                             const float *g_cdfs, const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles,
                             int64_t n_rays, int64_t n_elems, int32_t row_len, float *grad_sigmas, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("density_cdf_rows_bwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(t_starts && t_ends && trans && g_cdfs && grad_sigmas, "density_cdf_rows_bwd: null pointer");
    (void)alphas;  // not needed: only the transmittance carries a gradient
    NFA_REQUIRE(row_len >= 1 && n_rays * (int64_t)row_len == n_elems, "density_cdf_rows_bwd: n_elems must be n_rays * row_len");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends, trans, alphas, grad_sigmas);
#define NFA_DCB(V)                                                                                         \
    do { DensityBwdOp<V, true> op; op.ts = t_starts; op.te = t_ends; op.tr = trans; op.al = alphas; op.gw = nullptr; \
         op.gt = nullptr; op.ga = nullptr; op.gcdf = g_cdfs; op.gsig = grad_sigmas; op.gx = nullptr;          \
         launch_seg<-1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_DCB(true); else NFA_DCB(false);
#undef NFA_DCB
    NFA_CHECK_LAUNCH("density_cdf_rows_bwd");
    return NFA_OK;
}

int nfa_render_from_alpha_bwd(const float *alphas, const float *trans, const float *g_weights, const float *g_trans,
                              const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                              float *grad_alphas, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_from_alpha_bwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(alphas && trans && grad_alphas, "render_from_alpha_bwd: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(alphas, trans, g_weights, g_trans, grad_alphas);
#define NFA_AB(V)                                                                                          \
    do { AlphaBwdOp<V> op; op.al = alphas; op.tr = trans; op.gw = g_weights; op.gt = g_trans; op.galpha = grad_alphas; \
         launch_seg<-1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_AB(true); else NFA_AB(false);
#undef NFA_AB
    NFA_CHECK_LAUNCH("render_from_alpha_bwd");
    return NFA_OK;
}

int nfa_render_visibility(const float *t_starts, const float *t_ends, const float *sigmas_or_alphas,
                          const float *prefix_trans, float early_stop_eps, float alpha_thre,
                          const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                          uint8_t *vis, int64_t *vis_cnts, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_visibility");
    hipStream_t s = as_stream(stream);
    if (n_elems > 0) {
        NFA_REQUIRE(sigmas_or_alphas && vis, "render_visibility: null pointer");
        const bool density = t_starts != nullptr;
        NFA_REQUIRE(!density || t_ends, "render_visibility: t_ends is null");
        // S = -ln(eps) is where exp(-S) crosses eps; expf is good to a couple of ulps, the band is +-2e-5 relative (~20 ulps)
        float s_lo, s_hi;
        if (early_stop_eps > 0.0f) {
            const double L = -log((double)early_stop_eps);
            const double w = 2e-5 * (L > 1.0 ? L : 1.0);
            s_lo = (float)(L - w); s_hi = (float)(L + w);
        } else {   // every transmittance >= eps (exp(-S) is never negative; a NaN sum stays invisible as before)
            s_lo = INFINITY; s_hi = INFINITY;
        }
        // the uchar4 mask store needs 4-byte alignment of vis, the float loads 16
        const bool vec = all_aligned16(t_starts, t_ends, sigmas_or_alphas, prefix_trans) &&
                         (reinterpret_cast<uintptr_t>(vis) & 3) == 0;
#define NFA_VIS(DN, V, CN)                                                                                 \
    do { VisibilityOp<DN, V, CN> op; op.ts = t_starts; op.te = t_ends; op.val = sigmas_or_alphas; op.prefix = prefix_trans; \
         op.eps = early_stop_eps; op.thre = alpha_thre; op.vis = vis; op.cnts = vis_cnts; op.s_lo = s_lo; op.s_hi = s_hi; \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
#define NFA_VIS2(DN, V) do { if (vis_cnts) NFA_VIS(DN, V, true); else NFA_VIS(DN, V, false); } while (0)
        if (density) { if (vec) NFA_VIS2(true, true); else NFA_VIS2(true, false); }
        else         { if (vec) NFA_VIS2(false, true); else NFA_VIS2(false, false); }
#undef NFA_VIS2
#undef NFA_VIS
    } else if (vis_cnts && n_rays > 0) {
        if (hipMemsetAsync(vis_cnts, 0, sizeof(int64_t) * n_rays, s) != hipSuccess) { set_error("render_visibility: memset failed"); return NFA_EHIP; }
    }
    NFA_CHECK_LAUNCH("render_visibility");
    return NFA_OK;
}

int nfa_compact_samples(const uint8_t *vis, const float *t_starts, const float *t_ends, const int64_t *packed_info,
                        const int64_t *tiles, int64_t n_tiles, const int64_t *out_starts, int64_t n_rays, int64_t n_elems,
                        int64_t *out_ray_indices, float *out_t_starts, float *out_t_ends, int64_t capacity, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("compact_samples");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(vis && t_starts && t_ends && out_starts, "compact_samples: null input");
    NFA_REQUIRE(capacity >= 0, "compact_samples: negative capacity");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends);
#define NFA_CP(V)                                                                                          \
    do { CompactOp<V> op; op.vis = vis; op.vis_vec = (reinterpret_cast<uintptr_t>(vis) & 3) == 0; op.ts = t_starts; op.te = t_ends; op.out_starts = out_starts;       \
         op.o_ri = out_ray_indices; op.o_ts = out_t_starts; op.o_te = out_t_ends; op.cap = capacity;        \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_CP(true); else NFA_CP(false);
#undef NFA_CP
    NFA_CHECK_LAUNCH("compact_samples");
    return NFA_OK;
}

int nfa_accumulate_along_rays(const float *weights, const float *values, int32_t D, const int64_t *packed_info,
                              const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems, int accumulate, float *out,
                              nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("accumulate_along_rays");
    NFA_REQUIRE(D >= 1 && (values || D == 1), "accumulate_along_rays: bad D");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(out && (n_elems == 0 || weights), "accumulate_along_rays: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(weights);
    for (int32_t d0 = 0; d0 < D;) {
        const int32_t c = (D - d0 >= 4) ? 4 : (D - d0);
#define NFA_ACC(C, V)                                                                                      \
    do { AccumOp<C, V> op; op.w = weights; op.vals = values; op.D = D; op.d0 = d0; op.out = out;             \
         op.accumulate = accumulate; launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
        if (vec) { if (c == 4) NFA_ACC(4, true); else if (c == 3) NFA_ACC(3, true); else if (c == 2) NFA_ACC(2, true); else NFA_ACC(1, true); }
        else     { if (c == 4) NFA_ACC(4, false); else if (c == 3) NFA_ACC(3, false); else if (c == 2) NFA_ACC(2, false); else NFA_ACC(1, false); }
#undef NFA_ACC
        d0 += c;
    }
    NFA_CHECK_LAUNCH("accumulate_along_rays");
    return NFA_OK;
}

int nfa_accumulate_along_rays_atomic(const float *weights, const float *values, int32_t D, const int64_t *ray_indices,
                                     int64_t n_rays, int64_t n_elems, float *out, nfa_stream_t stream)
{
    NFA_REQUIRE(D >= 1 && (values || D == 1) && n_rays >= 0 && n_elems >= 0, "accumulate_along_rays_atomic: bad arguments");
    if (n_elems == 0 || n_rays == 0) return NFA_OK;
    NFA_REQUIRE(weights && ray_indices && out, "accumulate_along_rays_atomic: null pointer");
    hipLaunchKernelGGL(accumulate_atomic_kernel, dim3(grid_1d(n_elems * D, 256)), dim3(256), 0, as_stream(stream), weights,
                       values, D, ray_indices, n_rays, n_elems, out);
    NFA_CHECK_LAUNCH("accumulate_along_rays_atomic");
    return NFA_OK;
}

int nfa_accumulate_along_rays_bwd(const float *weights, const float *values, int32_t D, const float *g_out,
                                  const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                                  float *g_weights, float *g_values, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("accumulate_along_rays_bwd");
    NFA_REQUIRE(D >= 1 && (values || D == 1), "accumulate_along_rays_bwd: bad D");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(weights && g_out && (g_weights || g_values), "accumulate_along_rays_bwd: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(weights, g_weights);
    for (int32_t d0 = 0; d0 < D;) {
        const int32_t c = (D - d0 >= 4) ? 4 : (D - d0);
#define NFA_ACB(C, V)                                                                                      \
    do { AccumBwdOp<C, V> op; op.w = weights; op.vals = values; op.gout = g_out; op.D = D; op.d0 = d0;       \
         op.first = (d0 == 0); op.gw = g_weights; op.gv = g_values;                                          \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
        if (vec) { if (c == 4) NFA_ACB(4, true); else if (c == 3) NFA_ACB(3, true); else if (c == 2) NFA_ACB(2, true); else NFA_ACB(1, true); }
        else     { if (c == 4) NFA_ACB(4, false); else if (c == 3) NFA_ACB(3, false); else if (c == 2) NFA_ACB(2, false); else NFA_ACB(1, false); }
#undef NFA_ACB
        d0 += c;
    }
    NFA_CHECK_LAUNCH("accumulate_along_rays_bwd");
    return NFA_OK;
}

int nfa_render_accumulate_fwd(const float *weights, const float *rgbs, const float *t_starts, const float *t_ends,
                              const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                              float *colors, float *opacities, float *depths, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_accumulate_fwd");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(colors && opacities && depths && (n_elems == 0 || (weights && rgbs && t_starts && t_ends)),
                "render_accumulate_fwd: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(weights, rgbs, t_starts, t_ends);
#define NFA_RA(V)                                                                                          \
    do { RenderAccumOp<V> op; op.w = weights; op.rgb = rgbs; op.ts = t_starts; op.te = t_ends;               \
         op.colors = colors; op.opac = opacities; op.depth = depths;                                         \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_RA(true); else NFA_RA(false);
#undef NFA_RA
    NFA_CHECK_LAUNCH("render_accumulate_fwd");
    return NFA_OK;
}

int nfa_render_accumulate_bwd(const float *weights, const float *rgbs, const float *t_starts, const float *t_ends,
                              const float *g_colors, const float *g_opacities, const float *g_depths,
                              const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                              float *g_weights, float *g_rgbs, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_accumulate_bwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(weights && rgbs && t_starts && t_ends && (g_weights || g_rgbs), "render_accumulate_bwd: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(weights, rgbs, t_starts, t_ends, g_weights, g_rgbs);
#define NFA_RB(V)                                                                                          \
    do { RenderAccumBwdOp<V> op; op.w = weights; op.rgb = rgbs; op.ts = t_starts; op.te = t_ends;            \
         op.gc = g_colors; op.go = g_opacities; op.gd = g_depths; op.gw = g_weights; op.grgb = g_rgbs;       \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_RB(true); else NFA_RB(false);
#undef NFA_RB
    NFA_CHECK_LAUNCH("render_accumulate_bwd");
    return NFA_OK;
}

int nfa_render_fused_fwd(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                         const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays, int64_t n_elems,
                         float *weights, float *trans, float *alphas, float *colors, float *opacities, float *depths,
                         nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_fused_fwd");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(colors && opacities && depths && (n_elems == 0 || (t_starts && t_ends && sigmas && rgbs)),
                "render_fused_fwd: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends, sigmas, rgbs, weights, trans, alphas);
#define NFA_FF(V)                                                                                          \
    do { RenderFusedFwdOp<V> op; op.ts = t_starts; op.te = t_ends; op.sig = sigmas; op.rgb = rgbs; op.w = weights; \
         op.tr = trans; op.al = alphas; op.colors = colors; op.opac = opacities; op.depth = depths;          \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_FF(true); else NFA_FF(false);
#undef NFA_FF
    NFA_CHECK_LAUNCH("render_fused_fwd");
    return NFA_OK;
}

int nfa_render_fused_bwd(const float *t_starts, const float *t_ends, const float *rgbs, const float *trans, const float *alphas,
                         const float *g_colors, const float *g_opacities, const float *g_depths, const float *g_weights,
                         const float *g_trans, const float *g_alphas, const int64_t *packed_info, const int64_t *tiles,
                         int64_t n_tiles, int64_t n_rays, int64_t n_elems, float *grad_sigmas, float *grad_rgbs,
                         nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_fused_bwd");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(t_starts && t_ends && rgbs && trans && alphas && (grad_sigmas || grad_rgbs), "render_fused_bwd: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends, rgbs, trans, alphas, g_weights, g_trans, g_alphas, grad_sigmas, grad_rgbs);
#define NFA_FB(V, X)                                                                                       \
    do { RenderFusedBwdOp<V, X> op; op.ts = t_starts; op.te = t_ends; op.rgb = rgbs; op.tr = trans; op.al = alphas; \
         op.gc = g_colors; op.go = g_opacities; op.gd = g_depths; op.gw = g_weights; op.gt = g_trans; op.ga = g_alphas; \
         op.gsig = grad_sigmas; op.grgb = grad_rgbs;                                                         \
         launch_seg<-1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    const bool extra = g_weights || g_trans || g_alphas;
    if (vec) { if (extra) NFA_FB(true, true); else NFA_FB(true, false); }
    else     { if (extra) NFA_FB(false, true); else NFA_FB(false, false); }
#undef NFA_FB
    NFA_CHECK_LAUNCH("render_fused_bwd");
    return NFA_OK;
}

int nfa_render_step_accumulate(const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
                               const int64_t *packed_info, const int64_t *tiles, int64_t n_tiles, int64_t n_rays,
                               int64_t n_elems, float alpha_thre, float *colors, float *opacities, float *depths,
                               int64_t *n_visible, nfa_stream_t stream)
{
    SEG_COMMON_CHECKS("render_step_accumulate");
    if (n_elems == 0) return NFA_OK;
    NFA_REQUIRE(t_starts && t_ends && sigmas && rgbs && colors && opacities && depths, "render_step_accumulate: null pointer");
    hipStream_t s = as_stream(stream);
    const bool vec = all_aligned16(t_starts, t_ends, sigmas, rgbs);
#define NFA_RS(V)                                                                                          \
    do { RenderStepOp<V> op; op.ts = t_starts; op.te = t_ends; op.sig = sigmas; op.rgb = rgbs; op.thre = alpha_thre; \
         op.colors = colors; op.opac = opacities; op.depth = depths;                                         \
         op.n_visible = reinterpret_cast<unsigned long long *>(n_visible);                                   \
         launch_seg<1>(op, packed_info, tiles, n_rays, n_tiles, s); } while (0)
    if (vec) NFA_RS(true); else NFA_RS(false);
#undef NFA_RS
    NFA_CHECK_LAUNCH("render_step_accumulate");
    return NFA_OK;
}

}  // extern "C"
