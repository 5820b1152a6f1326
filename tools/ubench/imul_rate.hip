// Micro-benchmark (development aid): issue rate of 32-bit and 64-bit integer multiplies against
// fp32 and fp64 fused multiply-adds.  Every lane runs eight independent dependency chains of one
// operation, so the rate measured is the unit's throughput, not its latency.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/imul_rate.hip -o /tmp/imul_rate && /tmp/imul_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)

template <class T> __device__ __forceinline__ T step(T a, T b);
template <> __device__ __forceinline__ float step(float a, float b) { return __builtin_fmaf(a, b, b); }
template <> __device__ __forceinline__ double step(double a, double b) { return __builtin_fma(a, b, b); }
template <> __device__ __forceinline__ uint32_t step(uint32_t a, uint32_t b) { return a * b + 1u; }
template <> __device__ __forceinline__ uint64_t step(uint64_t a, uint64_t b) { return a * b + 1ull; }

template <class T>
__global__ __launch_bounds__(256) void k(T seed, int iters, T *out)
{
    T a[8];
    for (int j = 0; j < 8; j++) a[j] = seed + (T)(threadIdx.x + j);
    const T b = seed + (T)3;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = step<T>(a[j], b);
    T s = 0;
    for (int j = 0; j < 8; j++) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class T> static int run(const char *name, T seed)
{
    const int blocks = 256 * 8, iters = 4096;
    T *out = nullptr;
    CHECK(hipMalloc(&out, sizeof(T) * blocks * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, seed, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, seed, iters, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 256 * iters * 8;
    printf("%-10s %8.3f ms  %8.2f T lane-op/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    if (run<float>("fp32 fma", 1.0001f)) return 1;
    if (run<double>("fp64 fma", 1.0001)) return 1;
    if (run<uint32_t>("u32 mul", 2654435761u)) return 1;
    if (run<uint64_t>("u64 mul", 0x9E3779B97F4A7C15ull)) return 1;
    return 0;
}
