// common.hip.h -- shared device/host helpers for libnerfacc_hip.so (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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

// Knobs of the A/B tests and measurement scripts, set through nfa_set_tuning (grid.hip); the entry points do not read the
// environment.  NULL: not set.
const char *tuning_env(const char *name);

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

// Cross-lane moves as DPP modifiers (one VALU instruction each, no LDS crossbar round trip):
// step s < 4 shifts by 2^s inside each row of 16 lanes, step 4 broadcasts lane 15 of rows 0 / 2 to rows 1 / 3,
// step 5 broadcasts lane 31 to rows 2 and 3.  Lanes without a source keep `old`.
template <int S>
__device__ __forceinline__ int32_t dpp_step(int32_t old, int32_t src)
{
    static_assert(S >= 0 && S < 6, "dpp_step");
    if constexpr (S == 0) return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xf, 0xf, false);       // row_shr:1
    else if constexpr (S == 1) return __builtin_amdgcn_update_dpp(old, src, 0x112, 0xf, 0xf, false);  // row_shr:2
    else if constexpr (S == 2) return __builtin_amdgcn_update_dpp(old, src, 0x114, 0xf, 0xf, false);  // row_shr:4
    else if constexpr (S == 3) return __builtin_amdgcn_update_dpp(old, src, 0x118, 0xf, 0xf, false);  // row_shr:8
    else if constexpr (S == 4) return __builtin_amdgcn_update_dpp(old, src, 0x142, 0xa, 0xf, false);  // row_bcast:15
    else return __builtin_amdgcn_update_dpp(old, src, 0x143, 0xc, 0xf, false);                        // row_bcast:31
}
template <int S>
__device__ __forceinline__ float dpp_step(float old, float src)
{
    return __int_as_float(dpp_step<S>(__float_as_int(old), __float_as_int(src)));
}
// value of the previous lane (lane 0 keeps `old`)
__device__ __forceinline__ int32_t dpp_prev_lane(int32_t old, int32_t src)
{
    return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ float dpp_prev_lane(float old, float src)
{
    return __int_as_float(dpp_prev_lane(__float_as_int(old), __float_as_int(src)));
}
// a * b + c on 24-bit unsigned operands as ONE full-rate instruction (hipcc turns __umul24(a, b) + c into the
// quarter-rate 64-bit v_mad_u64_u32)
// Level of an event of a ray's sorted intersection list.  t_indices holds argsort indices in [0, 2G) (grid.py:158-162: entries
// below G enter level idx, the others leave level idx - G), so the reference's `idx % G` (grid.cu:127,137) is a conditional
// subtraction; as a 64-bit modulo by a run-time divisor it was ~200 instructions, twice per event -- a fifth of a limited
// walk's launch (the test-mode loop reads its list from the start in every iteration).  Out-of-range input: level >= G,
// which callers treat as "not hit".
__device__ __forceinline__ int32_t event_level(int64_t idx, int32_t G)
{
    const int32_t i = (int32_t)idx;
    return i >= G ? i - G : i;
}

__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8: MI355X_MICROARCH.md), and an XCD only
// takes the blocks dealt to it.  A cost that is periodic in the block index with a period that divides 8 therefore lands
// on a few XCDs: the rays of a 1024-pixel image row are four 256-ray blocks, the two middle ones cross the most cells, and
// XCDs 1, 2, 5, 6 were still walking them while the other four had been idle for the last third of the launch (wave
// time stamps, scripts/walk_timeline.py: 4096 busy waves, then 2000).  Within each group of 8 consecutive blocks the work
// items are rotated by the group's number, so every XCD sees every residue; a bijection on [0, n_blocks).
__device__ __forceinline__ int64_t xcd_fair_block(uint32_t b, uint32_t n_blocks)
{
    const uint32_t group = b >> 3;
    if (((group + 1u) << 3) > n_blocks) return b;   // (the last, partial group)
    return (int64_t)((b & ~7u) | ((b + group) & 7u));
}

// ---------------------------------------------------------------- issue rates (gfx950, scripts/valu_probe.hip)
// One SIMD issues a wave64 v_add / v_sub / v_mul / v_fma (f32), v_add / v_sub (u32), v_and / v_or / v_xor, v_lshrrev,
// v_ashrrev, v_mov, v_cndmask (VOP2 form, condition in vcc) and v_bitop3 every 2 cycles, and every 4 cycles: v_min / v_max
// (f32 and u32), v_min3 / v_med3, v_cmp, v_cndmask with the condition in another scalar pair (VOP3 form), v_bfi, v_bfe,
// v_lshlrev, v_mul_u32_u24, v_cvt and all three-operand integer instructions (v_lshl_add, v_and_or, v_mad_u32_u24, v_add3,
// v_perm, v_alignbit); v_pk_add_f32 too (profiles/r04_valu_probe.txt).  The issue-bound kernels are written against that
// table: a select whose condition is a lane MASK in a register ((k & x) | (~k & y)) is one v_bitop3 at the full rate where
// v_cndmask / v_bfi run at half of it, and "x is the minimum m of the values" is the sign of m - x (two full-rate
// instructions for a mask that serves any number of selects) instead of a compare plus a select each.
// (The compiler's builtin where there is one: between two `asm` statements it puts an s_nop whenever the second reads what the
// first wrote -- it cannot know that the first is no transcendental -- and an s_nop costs an issue slot like an instruction.)
__device__ __forceinline__ uint32_t sel_mask(uint32_t k, uint32_t x, uint32_t y)   // k ? x : y, bit by bit
{
    return __builtin_amdgcn_bitop3_b32(k, x, y, 0xca);
}
__device__ __forceinline__ float sel_mask(uint32_t k, float x, float y)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(k, __builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, y), 0xca));
}
// ~0 where a - b is negative (a < b for finite a, b; a == b gives +0: no bits), else 0
__device__ __forceinline__ uint32_t mask_less(float a, float b)
{
    return (uint32_t)(__builtin_bit_cast(int32_t, a - b) >> 31);   // v_sub_f32, v_ashrrev_i32 (checked in the generated code)
}
__device__ __forceinline__ float min3_f32(float a, float b, float c)
{
    float m;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}

__device__ __forceinline__ int32_t last_lane(int32_t v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float last_lane(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
// value of lane `l` (wave-uniform index)
__device__ __forceinline__ int32_t lane_value(int32_t v, int32_t l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float lane_value(float v, int32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }


// 16-byte stores of streamed outputs.  NFA_NT_STORES=1 marks them non-temporal (A/B switch).
#ifndef NFA_NT_STORES
#define NFA_NT_STORES 0
#endif
typedef float nfa_v4f __attribute__((ext_vector_type(4)));
typedef long long nfa_v2l __attribute__((ext_vector_type(2)));
template <bool NT = (NFA_NT_STORES != 0)>
__device__ __forceinline__ void store_f4(float *p, float a, float b, float c, float d)
{
    nfa_v4f v = {a, b, c, d};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<nfa_v4f *>(p));
    else *reinterpret_cast<nfa_v4f *>(p) = v;
}
template <bool NT = (NFA_NT_STORES != 0)>
__device__ __forceinline__ void store_l2(int64_t *p, int64_t a, int64_t b)
{
    nfa_v2l v = {a, b};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<nfa_v2l *>(p));
    else *reinterpret_cast<nfa_v2l *>(p) = v;
}

}  // namespace nfa
