// bench_csrc/field.hip -- the synthetic radiance field of bench.py (HARNESS code, not part of the
// product library and not behind its C ABI).
//
// nerfacc's API takes the radiance field as a user callback (an MLP in real use).  bench.py needs a field
// that costs as little as possible next to the hot path it measures, with a gradient flowing back into
// "network" parameters.  Written with torch elementwise ops the field below is ~25 kernel launches and
// 1.3 ms per step on 32 M samples, a third of the step; as three small kernels it streams each array once:
//     sigma_base(t) = 4 (1/2 + 1/2 sin(20 (ts + te)))
//     sigma = p0 * sigma_base,  rgb = (p1 ts, p1 ts, p1 ts)          (two scalar parameters p0, p1)
//     (the sampler's no-grad density callback is scale * sigma_base with the scale passed by value)
//     d loss / d p0 = sum g_sigma * sigma_base,   d loss / d p1 = sum (g_r + g_g + g_b) * ts
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ float sigma_base(float ts, float te) { return 4.0f * (0.5f + 0.5f * sinf(20.0f * (ts + te))); }

__global__ __launch_bounds__(256) void field_sigma_kernel(const float *__restrict__ ts, const float *__restrict__ te, int64_t n,
                                                          float scale, float *__restrict__ sigma)
{
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(ts)[i], b = reinterpret_cast<const float4 *>(te)[i];
        reinterpret_cast<float4 *>(sigma)[i] = make_float4(sigma_base(a.x, b.x) * scale, sigma_base(a.y, b.y) * scale,
                                                           sigma_base(a.z, b.z) * scale, sigma_base(a.w, b.w) * scale);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) { const int64_t i = 4 * n4 + threadIdx.x; sigma[i] = sigma_base(ts[i], te[i]) * scale; }
}

__global__ __launch_bounds__(256) void field_fwd_kernel(const float *__restrict__ ts, const float *__restrict__ te, int64_t n,
                                                        const float *__restrict__ params, float *__restrict__ sigma,
                                                        float *__restrict__ rgb)
{
    const float p0 = params[0], p1 = params[1];
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(ts)[i], b = reinterpret_cast<const float4 *>(te)[i];
        reinterpret_cast<float4 *>(sigma)[i] = make_float4(sigma_base(a.x, b.x) * p0, sigma_base(a.y, b.y) * p0,
                                                           sigma_base(a.z, b.z) * p0, sigma_base(a.w, b.w) * p0);
        const float c0 = a.x * p1, c1 = a.y * p1, c2 = a.z * p1, c3 = a.w * p1;
        float4 *o = reinterpret_cast<float4 *>(rgb) + 3 * i;
        o[0] = make_float4(c0, c0, c0, c1); o[1] = make_float4(c1, c1, c2, c2); o[2] = make_float4(c2, c3, c3, c3);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
        const int64_t i = 4 * n4 + threadIdx.x;
        sigma[i] = sigma_base(ts[i], te[i]) * p0;
        const float c = ts[i] * p1;
        rgb[3 * i] = c; rgb[3 * i + 1] = c; rgb[3 * i + 2] = c;
    }
}

// per-block partial sums (deterministic: fixed grid, tree reduction), finished by the host with a sum over blocks
__global__ __launch_bounds__(256) void field_bwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                        const float *__restrict__ g_sigma, const float *__restrict__ g_rgb,
                                                        int64_t n, float *__restrict__ partial /* [gridDim.x, 2] */)
{
    float s0 = 0.f, s1 = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(ts)[i];
        if (g_sigma) {
            const float4 b = reinterpret_cast<const float4 *>(te)[i], g = reinterpret_cast<const float4 *>(g_sigma)[i];
            s0 += g.x * sigma_base(a.x, b.x) + g.y * sigma_base(a.y, b.y) + g.z * sigma_base(a.z, b.z) + g.w * sigma_base(a.w, b.w);
        }
        if (g_rgb) {
            const float4 *q = reinterpret_cast<const float4 *>(g_rgb) + 3 * i;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2];
            s1 += (q0.x + q0.y + q0.z) * a.x + (q0.w + q1.x + q1.y) * a.y + (q1.z + q1.w + q2.x) * a.z + (q2.y + q2.z + q2.w) * a.w;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
        const int64_t i = 4 * n4 + threadIdx.x;
        if (g_sigma) s0 += g_sigma[i] * sigma_base(ts[i], te[i]);
        if (g_rgb) s1 += (g_rgb[3 * i] + g_rgb[3 * i + 1] + g_rgb[3 * i + 2]) * ts[i];
    }
    __shared__ float sh0[256], sh1[256];
    sh0[threadIdx.x] = s0; sh1[threadIdx.x] = s1;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sh0[threadIdx.x] += sh0[threadIdx.x + off]; sh1[threadIdx.x] += sh1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh0[0]; partial[2 * blockIdx.x + 1] = sh1[0]; }
}

// ---- cfg 3 (PropNetEstimator): proposal density  sigma = p0 exp(-(mid - p1)^2), mid = (ts + te) / 2, two parameters;
//      fine density 5 exp(-2 (mid - 4)^2) (no parameters).  Same role as the field above: the user's networks, kept cheap.
__global__ __launch_bounds__(256) void prop_fwd_kernel(const float *__restrict__ ts, const float *__restrict__ te, int64_t n,
                                                       const float *__restrict__ params, float a, float b, float c, float *__restrict__ sigma)
{
    // params != nullptr: sigma = params[0] * exp(-(mid - params[1])^2); else sigma = a * exp(-b (mid - c)^2)
    const float p0 = params ? params[0] : a, p1 = params ? params[1] : c, k = params ? 1.0f : b;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float u = (ts[i] + te[i]) * 0.5f - p1;
        sigma[i] = expf(-(u * u) * k) * p0;
    }
}
__global__ __launch_bounds__(256) void prop_bwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                       const float *__restrict__ g, int64_t n, const float *__restrict__ params,
                                                       float *__restrict__ partial /* [gridDim.x, 2] */)
{
    const float p0 = params[0], p1 = params[1];
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float u = (ts[i] + te[i]) * 0.5f - p1;
        const float e = expf(-(u * u));
        s0 += g[i] * e;                       // d sigma / d p0
        s1 += g[i] * (p0 * e * 2.0f * u);     // d sigma / d p1
    }
    __shared__ float sh0[256], sh1[256];
    sh0[threadIdx.x] = s0; sh1[threadIdx.x] = s1;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sh0[threadIdx.x] += sh0[threadIdx.x + off]; sh1[threadIdx.x] += sh1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh0[0]; partial[2 * blockIdx.x + 1] = sh1[0]; }
}

}  // namespace

extern "C" {

// one wave busy-waits for `ticks` of the 100 MHz wall clock: bench.py's per-kernel timing puts it in front of the first event
// of a timed native call, so that the call and both events are queued by the time the GPU gets to them (an event pair
// around a launch the host has not issued yet would measure the launch latency as well)
__global__ void spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
int bf_spin(long long ticks, void *stream)
{
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int bf_prop_fwd(const float *ts, const float *te, int64_t n, const float *params, float a, float b, float c, float *sigma, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(prop_fwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, ts, te, n, params, a, b, c, sigma);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int bf_prop_bwd(const float *ts, const float *te, const float *g_sigma, int64_t n, const float *params, float *partial, void *stream)
{
    hipLaunchKernelGGL(prop_bwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, ts, te, g_sigma, n, params, partial);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}


int bf_grid_blocks(void) { return 2048; }

int bf_field_sigma(const float *ts, const float *te, int64_t n, float scale, float *sigma, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(field_sigma_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, ts, te, n, scale, sigma);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int bf_field_fwd(const float *ts, const float *te, int64_t n, const float *params, float *sigma, float *rgb, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(field_fwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, ts, te, n, params, sigma, rgb);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int bf_field_bwd(const float *ts, const float *te, const float *g_sigma, const float *g_rgb, int64_t n, float *partial,
                 void *stream)
{
    hipLaunchKernelGGL(field_bwd_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, ts, te, g_sigma, g_rgb, n, partial);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
