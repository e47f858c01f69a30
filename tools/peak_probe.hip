// On-device peak probe for MI355X (gfx950): what the matrix cores and HBM deliver to the simplest possible kernels,
// to set beside the nominal figures (2.5 PFLOP/s dense bf16, 8 TB/s) the roofline fractions are quoted against.
//   hipcc -O3 --offload-arch=gfx950 tools/peak_probe.hip -o tools/build/peak_probe && tools/build/peak_probe
// Prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

// NACC independent accumulators per wave, operands held in registers: nothing but v_mfma_f32_32x32x16_bf16.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.001f * (float)((threadIdx.x + e) & 15));
        b[e] = (__bf16)(0.002f * (float)((threadIdx.x * 3 + e) & 15));
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ src, float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s += src[i];
    const float t = s[0] + s[1] + s[2] + s[3];
    if (t == 12345.678f) out[0] = t;   // never true for the zero-filled buffer: keeps the loads alive
}

template <class F>
static float best_ms(F&& launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;

    // ---- matrix cores: 4 waves per workgroup (one per SIMD), wpc workgroups per CU
    float* out;
    const int max_blocks = cus * 4;
    CK(hipMalloc(&out, (size_t)max_blocks * 256 * sizeof(float)));
    const int iters = 20000;
    const double flop_per_mfma = 2.0 * 32 * 32 * 16;
    double tf[3];
    int cfg = 0;
    for (int wpc : {1, 2, 4}) {
        const int blocks = cus * wpc;
        const float ms = best_ms([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
        tf[cfg++] = (double)blocks * 4 * iters * 4 * flop_per_mfma / (ms * 1e-3) / 1e12;
    }
    CK(hipFree(out));

    // ---- HBM: 2 GiB source and destination (far beyond the 256 MB last-level cache)
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(f32x4);
    f32x4 *src, *dst;
    float* flag;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMalloc(&flag, 256));
    CK(hipMemset(src, 0, bytes));
    CK(hipMemset(dst, 0, bytes));
    const int grid = cus * 16;
    const float copy_ms = best_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, src, dst, n); }, 5);
    const float read_ms = best_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, src, flag, n); }, 5);
    const float d2d_ms = best_ms([&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); }, 5);
    CK(hipFree(src));
    CK(hipFree(dst));
    CK(hipFree(flag));

    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, "
           "\"mfma_bf16_tflops\": {\"1_wave_per_simd\": %.1f, \"2_waves_per_simd\": %.1f, \"4_waves_per_simd\": %.1f}, "
           "\"hbm_gbs\": {\"copy_kernel_read_plus_write\": %.1f, \"read_kernel\": %.1f, \"memcpy_d2d_read_plus_write\": %.1f}, "
           "\"buffer_gib\": 2}\n",
           prop.gcnArchName, cus, prop.clockRate / 1000, tf[0], tf[1], tf[2], 2.0 * bytes / (copy_ms * 1e-3) / 1e9,
           (double)bytes / (read_ms * 1e-3) / 1e9, 2.0 * bytes / (d2d_ms * 1e-3) / 1e9);
    return 0;
}
