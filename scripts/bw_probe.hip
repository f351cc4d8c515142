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
    return 0;
}
