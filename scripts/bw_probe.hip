// Scratch: what bounds the segmented engine?  3 arrays in, 3 out, 32M floats each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_flat(const float4* a, const float4* b, const float4* c, float4* x, float4* y, float4* z, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 va = a[i], vb = b[i], vc = c[i];
        float4 o1 = make_float4(va.x * vb.x, va.y * vb.y, va.z * vb.z, va.w * vb.w);
        x[i] = o1; y[i] = vc; z[i] = make_float4(o1.x + vc.x, o1.y + vc.y, o1.z + vc.z, o1.w + vc.w);
    }
}
// wave-tile: each wave owns CH consecutive 256-element chunks
template <int CH, int NSHFL, bool LDSOPS>
__global__ __launch_bounds__(256) void k_tile(const float* a, const float* b, const float* c, float* x, float* y, float* z, long n) {
    __shared__ int lds[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long tile = (long)blockIdx.x * 4 + wave;
    long base = tile * CH * 256;
    float carry = 0.f;
    for (int ci = 0; ci < CH; ++ci) {
        long p0 = base + ci * 256 + 4 * lane;
        if (p0 + 3 >= n) break;
        float4 va = *(const float4*)(a + p0), vb = *(const float4*)(b + p0), vc = *(const float4*)(c + p0);
        int h = -1;
        if (LDSOPS) {
            *(int4*)(lds + wave * 256 + 4 * lane) = make_int4(-1, -1, -1, -1);
            __builtin_amdgcn_wave_barrier();
            if ((lane & 7) == 0) lds[wave * 256 + lane * 3] = lane;
            __builtin_amdgcn_wave_barrier();
            int4 hh = *(const int4*)(lds + wave * 256 + 4 * lane);
            h = hh.x | hh.y | hh.z | hh.w;
        }
        float v = va.x * vb.x + va.y * vb.y + va.z * vb.z + va.w * vb.w;
#pragma unroll
        for (int s = 0; s < NSHFL; ++s) {
            float u = __shfl_up(v, 1 << (s % 6), 64);
            if (lane >= (1 << (s % 6)) && h < 0) v += u;
        }
        v += carry;
        carry = __shfl(v, 63, 64);
        *(float4*)(x + p0) = make_float4(v, v + va.y, v + va.z, v + va.w);
        *(float4*)(y + p0) = vc;
        *(float4*)(z + p0) = make_float4(v * vc.x, v * vc.y, v * vc.z, v * vc.w);
    }
}
// the fused backward's stream mix without any scan: 4 x 4 B + 12 B in, 4 B + 12 B out per element
template <bool STRIDED>
__global__ __launch_bounds__(256) void k_bwdmix(const float4* a, const float4* b, const float4* c, const float4* d, const float4* rgb,
                                                float4* gs, float4* grgb, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 va = a[i], vb = b[i], vc = c[i], vd = d[i];
        float4 r0, r1, r2;
        if (STRIDED) { r0 = rgb[3 * i]; r1 = rgb[3 * i + 1]; r2 = rgb[3 * i + 2]; }
        else { const long w = (i / 64) * 192 + (i % 64); r0 = rgb[w]; r1 = rgb[w + 64]; r2 = rgb[w + 128]; }
        const float4 o = make_float4(va.x * vb.x + vc.x * vd.x + r0.x, va.y * vb.y + vc.y * vd.y + r1.y, va.z * vb.z + vc.z * vd.z + r2.z, va.w * vb.w + vc.w * vd.w);
        gs[i] = o;
        const float4 g0 = make_float4(r0.x * o.x, r0.y * o.x, r0.z * o.x, r0.w * o.y), g1 = make_float4(r1.x * o.y, r1.y * o.y, r1.z * o.z, r1.w * o.z),
                     g2 = make_float4(r2.x * o.z, r2.y * o.w, r2.z * o.w, r2.w * o.w);
        if (STRIDED) { grgb[3 * i] = g0; grgb[3 * i + 1] = g1; grgb[3 * i + 2] = g2; }
        else { const long w = (i / 64) * 192 + (i % 64); grgb[w] = g0; grgb[w + 64] = g1; grgb[w + 128] = g2; }
    }
}
// tile kernel whose tile start comes from a table in memory (one dependent load) and whose first step also waits for a
// second table row addressed by the first (two dependent loads before any data is requested), like seg_run_tile
template <int CH, int DEP>
__global__ __launch_bounds__(256) void k_tile_dep(const float* a, const float* b, const float* c, float* x, float* y, float* z, long n,
                                                  const long* tab, const long* tab2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 4 + wave;
    long base = tile * CH * 256;
    if (DEP >= 1) base = tab[tile];
    if (DEP >= 2) base = tab2[base / (CH * 256)] + (lane > 64 ? 1 : 0);
    float carry = 0.f;
    for (int ci = 0; ci < CH; ++ci) {
        long p0 = base + ci * 256 + 4 * lane;
        if (p0 + 3 >= n) break;
        float4 va = *(const float4*)(a + p0), vb = *(const float4*)(b + p0), vc = *(const float4*)(c + p0);
        float v = va.x * vb.x + va.y * vb.y + va.z * vb.z + va.w * vb.w;
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            float u = __shfl_up(v, 1 << (s % 6), 64);
            if (lane >= (1 << (s % 6))) v += u;
        }
        v += carry;
        carry = __shfl(v, 63, 64);
        *(float4*)(x + p0) = make_float4(v, v + va.y, v + va.z, v + va.w);
        *(float4*)(y + p0) = vc;
        *(float4*)(z + p0) = make_float4(v * vc.x, v * vc.y, v * vc.z, v * vc.w);
    }
}
// tile kernel + NV extra VALU instructions per 256-element step (ILP-way independent chains)
template <int NV, int ILP>
__global__ __launch_bounds__(256) void k_tile_valu(const float* a, const float* b, const float* c, float* x, float* y, float* z, long n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long base = ((long)blockIdx.x * 4 + wave) * 1024;
    float carry = 0.f;
    for (int ci = 0; ci < 4; ++ci) {
        long p0 = base + ci * 256 + 4 * lane;
        if (p0 + 3 >= n) break;
        float4 va = *(const float4*)(a + p0), vb = *(const float4*)(b + p0), vc = *(const float4*)(c + p0);
        float acc[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = va.x + i;
#pragma unroll
        for (int s = 0; s < NV / ILP; ++s)
#pragma unroll
            for (int i = 0; i < ILP; ++i) acc[i] = __builtin_fmaf(acc[i], vb.y, vc.z);
        float v = carry;
#pragma unroll
        for (int i = 0; i < ILP; ++i) v += acc[i];
        carry = __shfl(v, 63, 64);
        *(float4*)(x + p0) = make_float4(v, v + va.y, v + va.z, v + va.w);
        *(float4*)(y + p0) = vc;
        *(float4*)(z + p0) = make_float4(v * vc.x, v * vc.y, v * vc.z, v * vc.w);
    }
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10 * 1e3f;
}
int main() {
    const long n = 32244949 / 2048 * 2048;
    float *p[6];
    for (int i = 0; i < 6; ++i) { CHECK(hipMalloc(&p[i], n * 4)); CHECK(hipMemset(p[i], 0, n * 4)); }
    double bytes = 6.0 * n * 4;
    auto rep = [&](const char* name, float us) { printf("%-44s %8.1f us  %7.2f TB/s\n", name, us, bytes / us / 1e6); };
    rep("flat grid-stride 2048 blocks", timeit([&] { hipLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, (float4*)p[0], (float4*)p[1], (float4*)p[2], (float4*)p[3], (float4*)p[4], (float4*)p[5], n / 4); }));
    rep("flat one-elem-per-thread", timeit([&] { hipLaunchKernelGGL(k_flat, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (float4*)p[0], (float4*)p[1], (float4*)p[2], (float4*)p[3], (float4*)p[4], (float4*)p[5], n / 4); }));
#define TILE(CH, NS, L) rep("tile CH=" #CH " shfl=" #NS " lds=" #L, timeit([&] { hipLaunchKernelGGL((k_tile<CH, NS, L>), dim3((unsigned)(n / (CH * 256) / 4)), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], n); }))
    TILE(8, 0, false); TILE(8, 6, false); TILE(8, 12, false); TILE(8, 12, true); TILE(8, 36, true);
    TILE(1, 0, false); TILE(1, 12, true); TILE(2, 12, true); TILE(4, 12, true); TILE(16, 12, true); TILE(32, 12, true);
#define VAL(NV, ILP) rep("tile CH=4 + " #NV " VALU/step, ILP " #ILP, timeit([&] { hipLaunchKernelGGL((k_tile_valu<NV, ILP>), dim3((unsigned)(n / 1024 / 4)), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], n); }))
    VAL(200, 1); VAL(400, 1); VAL(800, 1); VAL(400, 4); VAL(800, 4); VAL(1600, 4);
    {   // dependent loads at tile start
        const long nt = n / 1024;
        std::vector<long> h(nt);
        for (long i = 0; i < nt; ++i) h[i] = i * 1024;
        long *tab, *tab2;
        CHECK(hipMalloc(&tab, nt * 8)); CHECK(hipMalloc(&tab2, nt * 8));
        CHECK(hipMemcpy(tab, h.data(), nt * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(tab2, h.data(), nt * 8, hipMemcpyHostToDevice));
        rep("tile CH=4, start known", timeit([&] { hipLaunchKernelGGL((k_tile_dep<4, 0>), dim3((unsigned)(nt / 4)), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], n, tab, tab2); }));
        rep("tile CH=4, start from a table (1 dep. load)", timeit([&] { hipLaunchKernelGGL((k_tile_dep<4, 1>), dim3((unsigned)(nt / 4)), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], n, tab, tab2); }));
        rep("tile CH=4, 2 dependent loads first", timeit([&] { hipLaunchKernelGGL((k_tile_dep<4, 2>), dim3((unsigned)(nt / 4)), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], n, tab, tab2); }));
    }
    {   // fused-backward stream mix: 44 B per element
        float *q[4], *rgb, *gs, *grgb;
        for (int i = 0; i < 4; ++i) { CHECK(hipMalloc(&q[i], n * 4)); CHECK(hipMemset(q[i], 0, n * 4)); }
        CHECK(hipMalloc(&rgb, n * 12)); CHECK(hipMemset(rgb, 0, n * 12)); CHECK(hipMalloc(&gs, n * 4)); CHECK(hipMalloc(&grgb, n * 12));
        const double b2 = 44.0 * n;
        auto rep2 = [&](const char* name, float us) { printf("%-44s %8.1f us  %7.2f TB/s\n", name, us, b2 / us / 1e6); };
        for (unsigned g : {2048u, 8192u, (unsigned)(n / 4 / 256)}) {
            char nm[96];
            snprintf(nm, 96, "bwd mix strided-48B, %u blocks", g);
            rep2(nm, timeit([&] { hipLaunchKernelGGL(k_bwdmix<true>, dim3(g), dim3(256), 0, 0, (float4*)q[0], (float4*)q[1], (float4*)q[2], (float4*)q[3], (float4*)rgb, (float4*)gs, (float4*)grgb, n / 4); }));
            snprintf(nm, 96, "bwd mix coalesced, %u blocks", g);
            rep2(nm, timeit([&] { hipLaunchKernelGGL(k_bwdmix<false>, dim3(g), dim3(256), 0, 0, (float4*)q[0], (float4*)q[1], (float4*)q[2], (float4*)q[3], (float4*)rgb, (float4*)gs, (float4*)grgb, n / 4); }));
        }
    }
    return 0;
}
