// common.hip.h -- shared device/host helpers for libnerfacc_hip.so (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nerfacc_hip.h"

#define NFA_WAVE 64

namespace nfa {

void set_error(const char *fmt, ...);

#define NFA_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) {                                 \
            nfa::set_error(__VA_ARGS__);               \
            return NFA_EINVAL;                         \
        }                                              \
    } while (0)

// Launch errors are not swallowed (the reference discards cudaGetLastError, scan.cu:64).
#define NFA_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            nfa::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return NFA_EHIP;                                                         \
        }                                                                            \
    } while (0)

static inline hipStream_t as_stream(nfa_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for 1-D grid-stride kernels: enough workgroups to fill 256 CUs x 8 blocks.
static inline unsigned grid_1d(int64_t n, int block, int64_t cap = 256 * 16)
{
    int64_t g = ceil_div64(n, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------- wave64 primitives
__device__ __forceinline__ int lane_id() { return __lane_id(); }

template <typename T>
__device__ __forceinline__ T wave_shfl_up(T v, int delta) { return __shfl_up(v, delta, NFA_WAVE); }
template <typename T>
__device__ __forceinline__ T wave_shfl(T v, int src) { return __shfl(v, src, NFA_WAVE); }

__device__ __forceinline__ int64_t wave_incl_sum_i64(int64_t v)
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < NFA_WAVE; off <<= 1) {
        int64_t u = __shfl_up(v, off, NFA_WAVE);
        if (lane >= off) v += u;
    }
    return v;
}

}  // namespace nfa
