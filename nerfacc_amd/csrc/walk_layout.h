// walk_layout.h -- cell order of the 1-bit-per-cell grid copy read by the run-length walk (walk.hip) and written by the
// grid maintenance kernels (gridupd.hip).
#pragma once
#include <stdint.h>

namespace nfa {

// Cell (x, y, z) of level l sits at bit  l << bits | dep(x, mask[0]) | dep(y, mask[1]) | dep(z, mask[2])  of the grid copy
// (dep = deposit the coordinate's bits at the mask's positions): z0 x0 y0 in bits 0..2, then the remaining coordinate
// bits interleaved.  A 128-byte line is a 16 x 8 x 8 block of cells (z, x, y) whatever the ray directions are: with the
// reference's z-fastest order every lane of a wave of neighbouring rays sits in a line of its own, none of which
// survives in the 32 KiB L1 (measured: +100 us at 128^3, +330 us at 256^3 on 1 M rays); blocked, the walk runs from L1.
struct WalkLayout {
    uint32_t mask[3];
    int32_t bits;                // index bits per level
};

__host__ __device__ inline WalkLayout walk_layout(const int32_t res[3])
{
    WalkLayout L;
    int nb[3];
    for (int ax = 0; ax < 3; ++ax) {
        nb[ax] = 0;
        while ((1 << nb[ax]) < res[ax]) nb[ax]++;
        L.mask[ax] = 0u;
    }
    // bits 0, 1, 2 belong to z, x, y (an axis of one cell leaves its bit unused), so that the lowest bit of an axis' mask is
    // mask & 7; the higher coordinate bits follow in the same rotation, exhausted axes skipped
    const int order[3] = {2, 0, 1};
    int pos = 0;
    for (int round = 0; round < 10; ++round)
        for (int i = 0; i < 3; ++i) {
            const int ax = order[i];
            if (round < nb[ax]) L.mask[ax] |= 1u << pos;
            if (round == 0 || round < nb[ax]) pos++;
        }
    L.bits = pos < 5 ? 5 : pos;   // a level is a whole number of 32-bit words
    return L;
}

__host__ __device__ inline uint32_t bit_deposit(uint32_t v, uint32_t mask)
{
    uint32_t out = 0u;
    for (uint32_t m = mask; m != 0u; m &= m - 1u, v >>= 1) out |= (v & 1u) ? (m & (0u - m)) : 0u;
    return out;
}
// When the three axes have the same number of index bits (any cubic grid) the masks are the regular rotation z, x, y:
// mask[2] = 0b...001001, mask[0] = mask[2] << 1, mask[1] = mask[2] << 2, and a deposit is the classic "one bit in three"
// spread (8 instructions) shifted by the axis' position instead of a loop over the mask's bits (~4 instructions per bit).
__host__ __device__ inline bool walk_layout_regular(const WalkLayout &L)
{
    const uint32_t mz = L.mask[2];
    uint32_t nb = 0;
    for (uint32_t m = mz; m != 0u; m &= m - 1u) ++nb;
    const uint32_t want = nb >= 11u ? 0u : (0x09249249u & ((1u << (3u * nb)) - 1u));
    return nb >= 1u && mz == want && L.mask[0] == (mz << 1) && L.mask[1] == (mz << 2);
}
__host__ __device__ inline uint32_t spread_by_3(uint32_t x)   // bit k of x (k < 10) -> bit 3k
{
    x &= 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__host__ __device__ inline uint32_t bit_extract(uint32_t v, uint32_t mask)
{
    uint32_t out = 0u, k = 0u;
    for (uint32_t m = mask; m != 0u; m &= m - 1u, ++k) out |= (v & (m & (0u - m))) ? (1u << k) : 0u;
    return out;
}

}  // namespace nfa
