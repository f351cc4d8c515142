// valu_probe.hip -- what one SIMD of gfx950 issues per cycle for the instructions the walk's cell loop is made of.
// Each wave runs ITER trips of a body of 32 instructions of one kind (4 independent chains), timed with s_memtime;
// W waves per SIMD (blocks of 256 threads = one wave per SIMD, W blocks per CU, 256 CUs).
// Output: cycles per wave-instruction as seen by ONE wave, and cycles per instruction per SIMD (= that / W).
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

#define BODY_2OP(op)                                                                                              \
    REP8(asm volatile(op " %0, %4, %0\n" op " %1, %4, %1\n" op " %2, %4, %2\n" op " %3, %4, %3\n"                 \
                      : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
#define BODY_3OP(op)                                                                                              \
    REP8(asm volatile(op " %0, %4, %0, %5\n" op " %1, %4, %1, %5\n" op " %2, %4, %2, %5\n" op " %3, %4, %3, %5\n" \
                      : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, int iters, float seed)
{
    float a = seed + threadIdx.x, b = a * 1.5f, c = a * 2.5f, d = a * 3.5f, e = seed * 0.25f, f = seed * 0.125f;
    __shared__ float lds[256 * 4 + 16];
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 q4 = {0, 0, 0, 0};
    uint32_t la16 = threadIdx.x * 16;
    uint32_t la = threadIdx.x * 4, g = threadIdx.x * 7u, h = threadIdx.x * 13u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { BODY_2OP("v_add_f32") }
        if (KIND == 1) { BODY_2OP("v_min_f32") }
        if (KIND == 2) { BODY_2OP("v_and_b32") }
        if (KIND == 3) { BODY_2OP("v_add_u32") }
        if (KIND == 4) { BODY_2OP("v_cndmask_b32") }   // e32: condition in vcc
        if (KIND == 5) { BODY_3OP("v_bfi_b32") }
        if (KIND == 6) { BODY_3OP("v_lshl_add_u32") }
        if (KIND == 7) { BODY_3OP("v_and_or_b32") }
        if (KIND == 8) { BODY_3OP("v_min3_f32") }
        if (KIND == 9) { BODY_3OP("v_fma_f32") }
        if (KIND == 10) { BODY_3OP("v_bfe_u32") }
        if (KIND == 11) {  // v_cndmask e64 with an SGPR-pair condition
            REP8(asm volatile("v_cndmask_b32 %0, %4, %0, s[20:21]\nv_cndmask_b32 %1, %4, %1, s[20:21]\n"
                              "v_cndmask_b32 %2, %4, %2, s[20:21]\nv_cndmask_b32 %3, %4, %3, s[20:21]\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");)
        }
        if (KIND == 12) {  // compares into vcc / sgpr pairs
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\nv_cmp_lt_f32 s[20:21], %1, %4\n"
                              "v_cmp_lt_f32 vcc, %2, %4\nv_cmp_lt_f32 s[20:21], %3, %4\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc", "s20", "s21");)
        }
        if (KIND == 13) {  // packed fp32 add: two adds per lane per instruction (register pairs)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p = {a, b}, q = {c, d}, r = {e, f};
            REP8(asm volatile("v_pk_add_f32 %0, %2, %0\nv_pk_add_f32 %1, %2, %1\nv_pk_add_f32 %0, %2, %0\nv_pk_add_f32 %1, %2, %1\n"
                              : "+v"(p), "+v"(q) : "v"(r));)
            a = p.x; b = p.y; c = q.x; d = q.y;
        }
        if (KIND == 14) {  // ONE dependent chain
            REP8(asm volatile("v_add_f32 %0, %1, %0\nv_add_f32 %0, %1, %0\nv_add_f32 %0, %1, %0\nv_add_f32 %0, %1, %0\n" : "+v"(a) : "v"(e));)
        }

        if (KIND == 20) { BODY_2OP("v_xor_b32") }
        if (KIND == 21) { BODY_2OP("v_or_b32") }
        if (KIND == 22) { BODY_2OP("v_lshlrev_b32") }
        if (KIND == 23) { BODY_2OP("v_lshrrev_b32") }
        if (KIND == 24) { BODY_2OP("v_ashrrev_i32") }
        if (KIND == 25) { BODY_2OP("v_sub_u32") }
        if (KIND == 26) { BODY_2OP("v_mul_f32") }
        if (KIND == 27) { BODY_2OP("v_max_f32") }
        if (KIND == 28) { BODY_2OP("v_sub_f32") }
        if (KIND == 29) { BODY_2OP("v_min_u32") }
        if (KIND == 30) { BODY_2OP("v_mul_u32_u24") }
        if (KIND == 31) { BODY_3OP("v_mad_u32_u24") }
        if (KIND == 32) { BODY_3OP("v_add3_u32") }
        if (KIND == 33) { BODY_3OP("v_xad_u32") }
        if (KIND == 34) { BODY_3OP("v_or3_b32") }
        if (KIND == 35) { BODY_3OP("v_lshl_or_b32") }
        if (KIND == 36) { BODY_3OP("v_med3_f32") }
        if (KIND == 37) { BODY_3OP("v_alignbit_b32") }
        if (KIND == 38) { BODY_3OP("v_perm_b32") }
        if (KIND == 39) {
            REP8(asm volatile("v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %0\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        }
        if (KIND == 40) {
            REP8(asm volatile("v_cvt_f32_u32 %0, %0\nv_cvt_u32_f32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_u32_f32 %3, %3\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        }
        if (KIND == 41) {  // v_cndmask e32 with vcc written once per group
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\nv_cndmask_b32 %0, %4, %0, vcc\nv_cndmask_b32 %1, %4, %1, vcc\nv_cndmask_b32 %2, %4, %2, vcc\nv_cndmask_b32 %3, %4, %3, vcc\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
        }
        if (KIND == 42) {  // v_bitop3 (a ^ b ^ c = 0x96)
            REP8(asm volatile("v_bitop3_b32 %0, %4, %0, %5 bitop3:0x96\nv_bitop3_b32 %1, %4, %1, %5 bitop3:0x96\nv_bitop3_b32 %2, %4, %2, %5 bitop3:0x96\nv_bitop3_b32 %3, %4, %3, %5 bitop3:0x96\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        }
        if (KIND == 43) {  // cndmask with inline constants (0 / 1.0)
            REP8(asm volatile("v_cndmask_b32 %0, 0, 1.0, s[20:21]\nv_cndmask_b32 %1, 0, 1.0, s[20:21]\nv_cndmask_b32 %2, 0, 1.0, s[20:21]\nv_cndmask_b32 %3, 0, 1.0, s[20:21]\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "s20", "s21");)
        }
        if (KIND == 44) {  // LDS: b128 read + b32 write per 8 VALU
            REP8(asm volatile("ds_read_b128 %0, %2\nv_add_f32 %1, %1, %1\nv_add_f32 %1, %1, %1\nv_add_f32 %1, %1, %1\nv_add_f32 %1, %1, %1\n"
                              "ds_write_b32 %3, %1\nv_add_f32 %1, %1, %1\nv_add_f32 %1, %1, %1\nv_add_f32 %1, %1, %1\nv_add_f32 %1, %1, %1\ns_waitcnt lgkmcnt(0)\n"
                              : "=&v"(q4), "+v"(a) : "v"(la16), "v"(la) : "memory");)
            b += q4.x;
        }

        if (KIND == 50) {  // one e32 cndmask (vcc never written) among adds
            REP8(asm volatile("v_cndmask_b32 %0, %4, %0, vcc\nv_add_f32 %1, %4, %1\nv_add_f32 %2, %4, %2\nv_add_f32 %3, %4, %3\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        }
        if (KIND == 51) {  // e32 cndmask with a destination that is not a source
            REP8(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\nv_cndmask_b32 %1, %4, %5, vcc\nv_cndmask_b32 %2, %4, %5, vcc\nv_cndmask_b32 %3, %4, %5, vcc\n"
                              : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(e), "v"(f));)
        }
        if (KIND == 52) {  // VOP3 encoding reading vcc
            REP8(asm volatile("v_cndmask_b32_e64 %0, %4, %0, vcc\nv_cndmask_b32_e64 %1, %4, %1, vcc\nv_cndmask_b32_e64 %2, %4, %2, vcc\nv_cndmask_b32_e64 %3, %4, %3, vcc\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        }
        if (KIND == 53) {  // two e32 cndmask, two adds
            REP8(asm volatile("v_cndmask_b32 %0, %4, %0, vcc\nv_cndmask_b32 %1, %4, %1, vcc\nv_add_f32 %2, %4, %2\nv_add_f32 %3, %4, %3\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        }
        if (KIND == 54) {  // sgpr-pair cndmask, 4 different pairs
            REP8(asm volatile("v_cndmask_b32 %0, %4, %0, s[20:21]\nv_cndmask_b32 %1, %4, %1, s[22:23]\nv_cndmask_b32 %2, %4, %2, s[24:25]\nv_cndmask_b32 %3, %4, %3, s[26:27]\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        }
        if (KIND == 55) {  // bitop3 select with masks: (m & x) | (~m & y) = 0xCA-like
            REP8(asm volatile("v_bitop3_b32 %0, %5, %4, %0 bitop3:0xca\nv_bitop3_b32 %1, %5, %4, %1 bitop3:0xca\nv_bitop3_b32 %2, %5, %4, %2 bitop3:0xca\nv_bitop3_b32 %3, %5, %4, %3 bitop3:0xca\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        }
        if (KIND == 56) {  // bitop3 with an SGPR and an inline constant
            REP8(asm volatile("v_bitop3_b32 %0, s20, %4, %0 bitop3:0xca\nv_bitop3_b32 %1, %4, -1, %1 bitop3:0xca\nv_bitop3_b32 %2, s20, %4, %2 bitop3:0xca\nv_bitop3_b32 %3, %4, -1, %3 bitop3:0xca\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20");)
        }
        if (KIND == 15) {  // the mix of a cell step: 2 min, 2 cmp, 10 cndmask, 2 bfi, rest integer (30 VALU) + one LDS store
            REP4(asm volatile(
                "v_cmp_lt_f32 vcc, %1, %2\nv_min_f32 %3, %1, %2\nv_cmp_lt_f32 s[20:21], %0, %3\nv_cndmask_b32 %4, %5, %4, vcc\n"
                "v_bfe_u32 %6, %6, %7, 1\nv_cndmask_b32 %4, %4, %5, s[20:21]\nv_bfi_b32 %7, %4, %6, -1\nv_and_b32 %6, 7, %4\n"
                "v_add_u32 %7, %6, %7\nv_bfi_b32 %6, %4, %7, %6\nv_cndmask_b32 %4, %5, %4, vcc\nv_xor_b32 %7, %6, %5\n"
                "v_lshrrev_b32 %7, 3, %7\nv_and_b32 %7, 0x1ffc, %7\nv_min_f32 %3, %0, %3\nv_cndmask_b32 %4, %5, %4, vcc\n"
                "v_xor_b32 %6, %6, %7\nv_cndmask_b32 %4, %4, %5, s[20:21]\nv_cndmask_b32 %5, %5, %4, s[20:21]\nv_lshl_add_u32 %8, %6, 10, %8\n"
                "v_add_f32 %3, %3, %4\nv_add_u32 %6, %5, %6\nv_cndmask_b32 %1, %1, %3, vcc\nv_cndmask_b32 %0, %0, %3, s[20:21]\n"
                "v_cndmask_b32 %2, %3, %2, s[20:21]\nv_and_b32 %6, 0x20080200, %6\nv_and_or_b32 %7, %8, 64, %6\nv_cmp_ne_u32 vcc, 0x1234, %7\n"
                "v_cndmask_b32 %1, %1, %3, s[20:21]\nv_mov_b32 %5, %6\n"
                "v_and_b32 %8, 0x3fc, %8\nds_write_b32 %8, %3\n"
                : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(la)
                :: "vcc", "s20", "s21", "memory");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (a + b + c + d + e + f + (float)(g + h + la) == 12345.678f) lds[threadIdx.x] = a;   // keep the values alive
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int n_per_trip)
{
    unsigned long long *d_out;
    const int max_blocks = 256 * 8;
    hipMalloc(&d_out, sizeof(unsigned long long) * max_blocks * 4);
    const int iters = 2000;
    printf("%-28s", name);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = 256 * w;
        probe<KIND><<<blocks, 256>>>(d_out, iters, 1.0f);   // warm-up
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<KIND><<<blocks, 256>>>(d_out, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[h.size() / 2] / ((double)iters * n_per_trip);   // s_memtime ticks (100 MHz?) or cycles: see header line
        const double wall_per_instr_ns = (double)ms * 1e6 / ((double)iters * n_per_trip * w);   // per instruction per SIMD
        printf("  W=%d: %6.2f tick/inst/wave  %6.3f ns/inst/SIMD", w, cyc, wall_per_instr_ns);
    }
    printf("\n");
    hipFree(d_out);
}

int main()
{
    printf("ticks are s_memtime units; ns/inst/SIMD = kernel wall time / (instructions of one wave x waves per SIMD)\n");
    run<0>("v_add_f32", 32);
    run<1>("v_min_f32", 32);
    run<2>("v_and_b32", 32);
    run<3>("v_add_u32", 32);
    run<4>("v_cndmask_b32 (vcc)", 32);
    run<11>("v_cndmask_b32 (sgpr pair)", 32);
    run<5>("v_bfi_b32", 32);
    run<6>("v_lshl_add_u32", 32);
    run<7>("v_and_or_b32", 32);
    run<8>("v_min3_f32", 32);
    run<9>("v_fma_f32", 32);
    run<10>("v_bfe_u32", 32);
    run<12>("v_cmp_lt_f32", 32);
    run<13>("v_pk_add_f32", 32);
    run<14>("v_add_f32 dependent", 32);
    run<20>("v_xor_b32", 32);
    run<21>("v_or_b32", 32);
    run<22>("v_lshlrev_b32", 32);
    run<23>("v_lshrrev_b32", 32);
    run<24>("v_ashrrev_i32", 32);
    run<25>("v_sub_u32", 32);
    run<26>("v_mul_f32", 32);
    run<27>("v_max_f32", 32);
    run<28>("v_sub_f32", 32);
    run<29>("v_min_u32", 32);
    run<30>("v_mul_u32_u24", 32);
    run<31>("v_mad_u32_u24", 32);
    run<32>("v_add3_u32", 32);
    run<33>("v_xad_u32", 32);
    run<34>("v_or3_b32", 32);
    run<35>("v_lshl_or_b32", 32);
    run<36>("v_med3_f32", 32);
    run<37>("v_alignbit_b32", 32);
    run<38>("v_perm_b32", 32);
    run<39>("v_mov_b32", 32);
    run<40>("v_cvt f32<->u32", 32);
    run<41>("v_cmp + 4 cndmask e32", 40);
    run<42>("v_bitop3_b32", 32);
    run<43>("v_cndmask consts (sgpr)", 32);
    run<44>("8 add + ds_read_b128 + ds_write_b32", 80);
    run<50>("1 cndmask e32 + 3 add", 32);
    run<51>("cndmask e32 dst != src", 32);
    run<52>("cndmask e64 vcc", 32);
    run<53>("2 cndmask e32 + 2 add", 32);
    run<54>("cndmask 4 sgpr pairs", 32);
    run<55>("bitop3 select", 32);
    run<56>("bitop3 sgpr / const", 32);
    run<15>("cell-step mix (31 VALU+ds)", 4 * 32);
    return 0;
}
