// loop_probe.hip -- where the walk's cell loop (walk.hip: walk_cell, 32 vector instructions per cell) loses issue slots.
// The loop body as the compiler emits it, replayed in a synthetic kernel (no flips, one-line table), with parts
// switched off one at a time.  ns per cell per SIMD at W waves per SIMD; the vector pipe alone would need
// 5 half-rate + 27 full-rate instructions = 5 * 4 + 27 * 2 = 74 cycles.
//   hipcc -O3 --offload-arch=gfx950 scripts/loop_probe.hip -o build/loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define VALU_A                                                                                     \
    "v_mov_b32_e32 v7, v82\n"                                                                      \
    "v_min3_f32 v67, v72, v5, v4\n"
#define WAIT "s_waitcnt vmcnt(0)\n"
#define VALU_B                                                                                     \
    "v_bfe_u32 v82, v70, v69, 1\n"                                                                 \
    "v_sub_f32 v69, v67, v5\n"                                                                     \
    "v_sub_f32 v20, v67, v4\n"                                                                     \
    "v_xor_b32_e32 v7, v82, v7\n"                                                                  \
    "v_ashrrev_i32 v89, 31, v69\n"                                                                 \
    "v_ashrrev_i32 v20, 31, v20\n"                                                                 \
    "v_lshl_add_u32 v19, v7, 10, v19\n"                                                            \
    "v_bitop3_b32 v69, v89, v77, v78 bitop3:0xca\n"                                                \
    "v_and_b32_e32 v7, v89, v20\n"                                                                 \
    "v_bitop3_b32 v69, v20, v69, v79 bitop3:0xca\n"                                                \
    "v_bitop3_b32 v91, v89, v76, v74 bitop3:0xca\n"
#define SALU_1 "s_movk_i32 s26, 0x4000\n"
#define VALU_C                                                                                     \
    "v_bitop3_b32 v70, v69, v75, v75 bitop3:0xcf\n"                                                \
    "v_and_b32_e32 v90, 7, v69\n"                                                                  \
    "v_add_u32_e32 v70, v70, v90\n"                                                                \
    "v_bitop3_b32 v75, v69, v70, v75 bitop3:0xca\n"                                                \
    "v_bitop3_b32 v90, v20, v89, v20 bitop3:0x30\n"                                                \
    "v_xor_b32_e32 v69, v75, v80\n"                                                                \
    "v_lshrrev_b32_e32 v70, 3, v69\n"                                                              \
    "v_and_b32_e32 v70, 0x7c, v70\n"
#define LOAD "global_load_dword v70, v70, %[tab]\n"
#define NOLOAD "v_mov_b32 v70, 0\n"
#define VALU_D                                                                                     \
    "v_bitop3_b32 v89, v89, v30, v31 bitop3:0xca\n"                                                \
    "v_bitop3_b32 v91, v20, v91, v73 bitop3:0xca\n"
#define STORE "ds_write_b32 v19, v67\n"
#define VALU_E                                                                                     \
    "v_bitop3_b32 v89, v20, v89, v32 bitop3:0xca\n"                                                \
    "v_add_f32_e32 v91, v67, v91\n"                                                                \
    "v_sub_u32_e32 v71, v71, v89\n"                                                                \
    "v_bitop3_b32 v72, v7, v91, v72 bitop3:0xca\n"                                                 \
    "v_and_b32_e32 v7, 0x20080200, v71\n"                                                          \
    "v_and_or_b32 v89, v19, s26, v7\n"                                                             \
    "v_cmp_ne_u32_e32 vcc, s59, v89\n"
#define SALU_2 "s_or_b64 s[24:25], vcc, s[24:25]\n"
#define VALU_F                                                                                     \
    "v_bitop3_b32 v5, v90, v91, v5 bitop3:0xca\n"                                                  \
    "v_bitop3_b32 v4, v20, v4, v91 bitop3:0xca\n"
#define SALU_3 "s_andn2_b64 s[28:29], exec, s[24:25]\n"

#define CLOBBERS "v4", "v5", "v7", "v19", "v20", "v30", "v31", "v32", "v67", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", \
                 "v77", "v78", "v79", "v80", "v82", "v89", "v90", "v91", "s24", "s25", "s26", "s28", "s29", "s30", "s59", "vcc", "memory"

#define INIT                                                                                                         \
    "v_lshlrev_b32 v19, 2, %[lane]\n v_cvt_f32_u32 v72, %[lane]\n v_add_f32 v72, 1.0, v72\n v_add_f32 v5, 0.5, v72\n" \
    "v_add_f32 v4, 0.25, v72\n v_mov_b32 v77, 0.5\n v_mov_b32 v78, 1.0\n v_mov_b32 v79, 2.0\n v_mov_b32 v76, 0x249249\n"  \
    "v_mov_b32 v74, 0x492492\n v_mov_b32 v73, 0x924924\n v_mov_b32 v30, 1\n v_mov_b32 v31, 0x400\n v_mov_b32 v32, 0x100000\n" \
    "v_mov_b32 v75, %[lane]\n v_mov_b32 v80, 0\n v_mov_b32 v71, 0x3fffffff\n v_mov_b32 v82, 0\n v_mov_b32 v70, 0\n v_mov_b32 v69, 0\n" \
    "s_mov_b32 s59, 0x1234\n s_mov_b64 s[24:25], 0\n s_mov_b32 s30, %[iters]\n"

#define LOOP_HEAD "1:\n"
#define LOOP_TAIL "s_sub_u32 s30, s30, 1\n s_cmp_lg_u32 s30, 0\n s_cbranch_scc1 1b\n"

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, const uint32_t *tab, int iters)
{
    __shared__ float lds[256 * 4];
    const uint32_t lane = threadIdx.x;
    lds[lane] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define RUN(BODY) asm volatile(INIT LOOP_HEAD BODY LOOP_TAIL "s_waitcnt vmcnt(0) lgkmcnt(0)\n" :: [lane] "v"(lane), [tab] "s"(tab), [iters] "s"(iters) : CLOBBERS)
    if (KIND == 0) RUN(VALU_A WAIT VALU_B SALU_1 VALU_C LOAD VALU_D STORE VALU_E SALU_2 VALU_F SALU_3);            // everything
    if (KIND == 1) RUN(VALU_A VALU_B VALU_C NOLOAD VALU_D VALU_E VALU_F);                                          // vector only
    if (KIND == 2) RUN(VALU_A VALU_B SALU_1 VALU_C NOLOAD VALU_D VALU_E SALU_2 VALU_F SALU_3);                     // + scalar
    if (KIND == 3) RUN(VALU_A VALU_B SALU_1 VALU_C NOLOAD VALU_D STORE VALU_E SALU_2 VALU_F SALU_3);               // + LDS store
    if (KIND == 4) RUN(VALU_A WAIT VALU_B SALU_1 VALU_C LOAD VALU_D VALU_E SALU_2 VALU_F SALU_3);                  // + load, no store
    if (KIND == 5) RUN(VALU_A WAIT VALU_B SALU_1 VALU_C LOAD VALU_D STORE VALU_E SALU_2 VALU_F SALU_3              // everything, two cells per trip
                       VALU_A WAIT VALU_B SALU_1 VALU_C LOAD VALU_D STORE VALU_E SALU_2 VALU_F SALU_3);
    if (KIND == 6) RUN(VALU_A VALU_B VALU_C NOLOAD VALU_D VALU_E VALU_F VALU_A VALU_B VALU_C NOLOAD VALU_D VALU_E VALU_F   // vector only, four cells per trip
                       VALU_A VALU_B VALU_C NOLOAD VALU_D VALU_E VALU_F VALU_A VALU_B VALU_C NOLOAD VALU_D VALU_E VALU_F);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int cells_per_trip)
{
    unsigned long long *d_out;
    uint32_t *d_tab;
    (void)hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 8 * 4);
    (void)hipMalloc(&d_tab, 4096);
    (void)hipMemset(d_tab, 0, 4096);
    const int iters = 4000 / cells_per_trip;
    printf("%-40s", name);
    for (int w : {1, 2, 4, 5, 6, 8}) {
        const int blocks = 256 * w;
        probe<KIND><<<blocks, 256>>>(d_out, d_tab, iters);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        probe<KIND><<<blocks, 256>>>(d_out, d_tab, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  W=%d %6.1f", w, (double)ms * 1e6 / ((double)iters * cells_per_trip * w));
    }
    printf("   ns per cell per SIMD\n");
    (void)hipFree(d_out); (void)hipFree(d_tab);
}

int main()
{
    run<1>("vector only (32)", 1);
    run<6>("vector only, 4 cells per trip", 4);
    run<2>("+ 3 scalar", 1);
    run<3>("+ 3 scalar + ds_write", 1);
    run<4>("+ 3 scalar + load + waitcnt", 1);
    run<0>("everything", 1);
    run<5>("everything, 2 cells per trip", 2);
    return 0;
}
