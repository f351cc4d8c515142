// Scratch: the write-only ceiling for the expansion's stream mix (two f32 arrays + one i64 array, 16 B per element, 32 M elements).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef long long v2l __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void k_stride(float *a, float *b, long long *c, long n)
{
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
        v4f va = {(float)i, 1.f, 2.f, 3.f}, vb = {(float)i + 1.f, 2.f, 3.f, 4.f};
        v2l c0 = {i, i + 1}, c1 = {i + 2, i + 3};
        if (NT) {
            __builtin_nontemporal_store(va, (v4f *)(a + i)); __builtin_nontemporal_store(vb, (v4f *)(b + i));
            __builtin_nontemporal_store(c0, (v2l *)(c + i)); __builtin_nontemporal_store(c1, (v2l *)(c + i + 2));
        } else {
            *(v4f *)(a + i) = va; *(v4f *)(b + i) = vb; *(v2l *)(c + i) = c0; *(v2l *)(c + i + 2) = c1;
        }
    }
}
// wave-tile: each wave owns CH consecutive 256-element chunks (like the expansion's batches)
template <bool NT, int CH>
__global__ __launch_bounds__(128) void k_tile(float *a, float *b, long long *c, long n)
{
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 2 + (threadIdx.x >> 6);
    for (int ci = 0; ci < CH; ++ci) {
        const long i = (tile * CH + ci) * 256 + 4 * lane;
        if (i + 3 >= n) break;
        v4f va = {(float)i, 1.f, 2.f, 3.f}, vb = {(float)i + 1.f, 2.f, 3.f, 4.f};
        v2l c0 = {i, i + 1}, c1 = {i + 2, i + 3};
        if (NT) {
            __builtin_nontemporal_store(va, (v4f *)(a + i)); __builtin_nontemporal_store(vb, (v4f *)(b + i));
            __builtin_nontemporal_store(c0, (v2l *)(c + i)); __builtin_nontemporal_store(c1, (v2l *)(c + i + 2));
        } else {
            *(v4f *)(a + i) = va; *(v4f *)(b + i) = vb; *(v2l *)(c + i) = c0; *(v2l *)(c + i + 2) = c1;
        }
    }
}
template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / 10;
}
int main()
{
    const long n = 32244992;  // multiple of 256
    float *a, *b; long long *c;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4)); CHECK(hipMalloc(&c, n * 8));
    const double bytes = (double)n * 16;
    for (int g : {2048, 8192, 32768}) {
        float t0 = timeit([&] { hipLaunchKernelGGL(k_stride<false>, dim3(g), dim3(256), 0, 0, a, b, c, n); });
        float t1 = timeit([&] { hipLaunchKernelGGL(k_stride<true>, dim3(g), dim3(256), 0, 0, a, b, c, n); });
        printf("grid-stride %5d WGs: plain %6.1f us %5.2f TB/s   nontemporal %6.1f us %5.2f TB/s\n", g, t0 * 1e3, bytes / t0 / 1e9, t1 * 1e3, bytes / t1 / 1e9);
    }
    {
        const long tiles4 = n / (4 * 256), tiles16 = n / (16 * 256);
        float t0 = timeit([&] { hipLaunchKernelGGL((k_tile<false, 4>), dim3((tiles4 + 1) / 2), dim3(128), 0, 0, a, b, c, n); });
        float t1 = timeit([&] { hipLaunchKernelGGL((k_tile<true, 4>), dim3((tiles4 + 1) / 2), dim3(128), 0, 0, a, b, c, n); });
        float t2 = timeit([&] { hipLaunchKernelGGL((k_tile<true, 16>), dim3((tiles16 + 1) / 2), dim3(128), 0, 0, a, b, c, n); });
        printf("wave tiles of 4 chunks: plain %6.1f us %5.2f TB/s   nontemporal %6.1f us %5.2f TB/s;  16 chunks nontemporal %6.1f us %5.2f TB/s\n",
               t0 * 1e3, bytes / t0 / 1e9, t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9);
    }
    float tm = timeit([&] { CHECK(hipMemsetAsync(a, 0, n * 4, 0)); CHECK(hipMemsetAsync(b, 0, n * 4, 0)); CHECK(hipMemsetAsync(c, 0, n * 8, 0)); });
    printf("hipMemsetAsync x3: %6.1f us %5.2f TB/s\n", tm * 1e3, bytes / tm / 1e9);
    return 0;
}
