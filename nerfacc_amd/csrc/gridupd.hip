// gridupd.hip -- occupancy-grid maintenance (ref: nerfacc/estimators/occ_grid.py:345-404, OccGridEstimator._update) as
// kernels: the step that produces the grid the traversal consumes.
//
// The reference builds one update out of ~15 ATen launches per level (gather of the cell coordinates, rand, add, div,
// mul, add; gather of occs, mul, maximum, index_put; a boolean-index mean; a compare) and leaves a torch.bool grid that
// the traversal reads one byte per cell.  Here:
//   cell_points      cell ids + jitter -> world positions (one pass, the reference's fp32 operation order)
//   ema_update       occs[cell] = max(occs[cell] * decay, occ) for the sampled cells.  Cells drawn more than once (the
//                    uniform and the occupied samples overlap, randint draws with replacement) get
//                    max(occs * decay, max_i occ_i): one of the values the reference's racing index_put may leave,
//                    chosen deterministically (the reference's own comment asks for a scatter-max)
//   rebinarize       thre = min(mean(occs[occs >= 0]), occ_thre) reduced on the device (no host read), then
//                    binaries = occs > thre written BOTH as the torch.bool buffer (the serialised view, same layout as
//                    the reference's state_dict) and as the walk's bit-interleaved 1-bit copy (walk.hip), so the next
//                    traversal needs no packing pass.
#include "common.hip.h"
#include "walk_layout.h"

namespace nfa {

__global__ __launch_bounds__(256) void cell_points_kernel(const int64_t *__restrict__ indices, const float *__restrict__ jitter,
                                                          int64_t n, int32_t rx, int32_t ry, int32_t rz,
                                                          const float *__restrict__ aabb, float *__restrict__ x)
{
    const float lo[3] = {aabb[0], aabb[1], aabb[2]};
    const float ext[3] = {aabb[3] - aabb[0], aabb[4] - aabb[1], aabb[5] - aabb[2]};
    const float resf[3] = {(float)rx, (float)ry, (float)rz};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) {
        int64_t c = indices[i];
        const int32_t cz = (int32_t)(c % rz); c /= rz;
        const int32_t cy = (int32_t)(c % ry); c /= ry;
        const int32_t cx = (int32_t)c;
        const float cc[3] = {(float)cx, (float)cy, (float)cz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float u = (cc[a] + jitter[3 * i + a]) / resf[a];   // occ_grid.py:385-387
            x[3 * i + a] = lo[a] + u * ext[a];                        // :389-391
        }
    }
}

__global__ __launch_bounds__(256) void ema_gather_kernel(const float *__restrict__ occs, int64_t cell_base,
                                                         const int64_t *__restrict__ indices, int64_t n, float decay,
                                                         float *__restrict__ dec)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x)
        dec[i] = occs[cell_base + indices[i]] * decay;
}
__global__ __launch_bounds__(256) void ema_decay_kernel(float *__restrict__ occs, int64_t cell_base,
                                                        const int64_t *__restrict__ indices, int64_t n,
                                                        const float *__restrict__ dec)
{
    // duplicates of a cell all carry the same value (they read the same old occupancy): a benign race
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x)
        occs[cell_base + indices[i]] = dec[i];
}
__device__ __forceinline__ void atomic_max_f32(float *addr, float v)
{
    if (v != v) { atomicExch(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v)); return; }  // torch.maximum propagates NaN
    // order-preserving integer views: non-negative floats compare like signed ints, negative ones like reversed unsigned
    if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}
__global__ __launch_bounds__(256) void ema_max_kernel(float *__restrict__ occs, int64_t cell_base,
                                                      const int64_t *__restrict__ indices, int64_t n,
                                                      const float *__restrict__ occ)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x)
        atomic_max_f32(&occs[cell_base + indices[i]], occ[i]);
}

// ---- threshold: mean of the non-negative entries (cells marked invisible hold -1), fixed-shape reduction
constexpr int THR_BLOCKS = 1024;
__global__ __launch_bounds__(256) void occ_sum_kernel(const float *__restrict__ occs, int64_t n, double *__restrict__ partial /* [THR_BLOCKS][2] */)
{
    double s = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)blockDim.x * gridDim.x) {
        const float v = occs[i];
        if (v >= 0.0f) { s += (double)v; c += 1.0; }
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh[0][0]; partial[2 * blockIdx.x + 1] = sh[1][0]; }
}
__global__ __launch_bounds__(256) void occ_thre_kernel(const double *__restrict__ partial, float occ_thre, float *__restrict__ thre_out)
{
    __shared__ double sh[2][256];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < THR_BLOCKS; i += 256) { s += partial[2 * i]; c += partial[2 * i + 1]; }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(sh[0][0] / sh[1][0]);            // 0 / 0 = NaN, as torch's mean of an empty selection
        thre_out[0] = mean != mean ? mean : fminf(mean, occ_thre);   // torch.clamp(mean, max=occ_thre) keeps NaN
        thre_out[1] = mean;
    }
}

// binaries = occs > thre: one thread per 32-bit word of the walk's grid copy; the same 32 cells are written to the
// torch.bool buffer (4 z-consecutive cells = one 4-byte store where the level has at least 4 cells along z)
__global__ __launch_bounds__(256) void binarize_kernel(const float *__restrict__ occs, int32_t n_grids, int32_t rx, int32_t ry, int32_t rz,
                                                       WalkLayout L, const float *__restrict__ thre_p, uint8_t *__restrict__ binaries,
                                                       uint32_t *__restrict__ bits)
{
    const float thre = thre_p[0];
    const int64_t n_words = ((int64_t)n_grids << L.bits) >> 5;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += (int64_t)blockDim.x * gridDim.x) {
        const uint32_t base = (uint32_t)(wi << 5);
        const uint32_t lvl = base >> L.bits, in_lvl = base & ((1u << L.bits) - 1u);
        uint32_t w = 0u;
        for (uint32_t b = 0; b < 32u; ++b) {
            const uint32_t pidx = in_lvl | b;
            const uint32_t x = bit_extract(pidx, L.mask[0]), y = bit_extract(pidx, L.mask[1]), z = bit_extract(pidx, L.mask[2]);
            const uint32_t rest = pidx & ~(L.mask[0] | L.mask[1] | L.mask[2]);
            if (rest == 0u && x < (uint32_t)rx && y < (uint32_t)ry && z < (uint32_t)rz) {
                const int64_t cell = (((int64_t)lvl * rx + x) * ry + y) * rz + z;
                const bool on = occs[cell] > thre;
                binaries[cell] = on ? 1 : 0;
                w |= on ? (1u << b) : 0u;
            }
        }
        bits[wi] = w;
    }
}

}  // namespace nfa

using namespace nfa;

extern "C" {

int nfa_grid_cell_points(const int64_t *indices, const float *jitter, int64_t n, const int32_t *res, const float *aabb,
                         float *x, nfa_stream_t stream)
{
    NFA_REQUIRE(n >= 0 && res, "grid_cell_points: bad arguments");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(indices && jitter && aabb && x && res[0] > 0 && res[1] > 0 && res[2] > 0, "grid_cell_points: null pointer / bad resolution");
    hipLaunchKernelGGL(cell_points_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, as_stream(stream), indices, jitter, n, res[0], res[1],
                       res[2], aabb, x);
    NFA_CHECK_LAUNCH("grid_cell_points");
    return NFA_OK;
}

int nfa_grid_ema_update(float *occs, int64_t cell_base, const int64_t *indices, int64_t n, const float *occ, float ema_decay,
                        float *scratch, nfa_stream_t stream)
{
    NFA_REQUIRE(n >= 0 && cell_base >= 0, "grid_ema_update: bad arguments");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(occs && indices && occ && scratch, "grid_ema_update: null pointer");
    hipStream_t s = as_stream(stream);
    const unsigned g = grid_1d(n, 256);
    hipLaunchKernelGGL(ema_gather_kernel, dim3(g), dim3(256), 0, s, occs, cell_base, indices, n, ema_decay, scratch);
    hipLaunchKernelGGL(ema_decay_kernel, dim3(g), dim3(256), 0, s, occs, cell_base, indices, n, scratch);
    hipLaunchKernelGGL(ema_max_kernel, dim3(g), dim3(256), 0, s, occs, cell_base, indices, n, occ);
    NFA_CHECK_LAUNCH("grid_ema_update");
    return NFA_OK;
}

int64_t nfa_grid_rebinarize_scratch_bytes(void) { return (int64_t)THR_BLOCKS * 2 * sizeof(double) + 16; }

int nfa_grid_rebinarize(const float *occs, int32_t n_grids, const int32_t *res, float occ_thre, uint8_t *binaries,
                        uint32_t *walk_bits, void *scratch, nfa_stream_t stream)
{
    NFA_REQUIRE(occs && res && binaries && walk_bits && scratch && n_grids >= 1, "grid_rebinarize: bad arguments");
    NFA_REQUIRE(res[0] > 0 && res[1] > 0 && res[2] > 0 && res[0] <= 512 && res[1] <= 512 && res[2] <= 512,
                "grid_rebinarize: 1..512 cells per axis");
    const WalkLayout L = walk_layout(res);
    NFA_REQUIRE(L.bits >= 5 && ((int64_t)n_grids << L.bits) < ((int64_t)1 << 31), "grid_rebinarize: grid too large");
    hipStream_t s = as_stream(stream);
    double *partial = reinterpret_cast<double *>(scratch);
    float *thre = reinterpret_cast<float *>(partial + 2 * THR_BLOCKS);
    const int64_t n = (int64_t)n_grids * res[0] * res[1] * res[2];
    hipLaunchKernelGGL(occ_sum_kernel, dim3(THR_BLOCKS), dim3(256), 0, s, occs, n, partial);
    hipLaunchKernelGGL(occ_thre_kernel, dim3(1), dim3(256), 0, s, partial, occ_thre, thre);
    const int64_t n_words = ((int64_t)n_grids << L.bits) >> 5;
    hipLaunchKernelGGL(binarize_kernel, dim3(grid_1d(n_words, 256)), dim3(256), 0, s, occs, n_grids, res[0], res[1], res[2], L, thre,
                       binaries, walk_bits);
    NFA_CHECK_LAUNCH("grid_rebinarize");
    return NFA_OK;
}

}  // extern "C"
