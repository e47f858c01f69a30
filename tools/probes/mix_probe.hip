// Probe (tool, not shipped): one or two waves per SIMD, each running the same stream of {1 MFMA, KV vector instructions,
// optionally 1 ds_read_b128}.  Reports cycles per MFMA step per wave and the SIMD's aggregate MFMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KV, int LDS, int TRANS, int RING = 9>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int n, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
    __syncthreads();
    float res = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < nwaves) {
        f32x16 acc0 = {0}, acc1 = {0};
        bf16x8 w = *reinterpret_cast<const bf16x8*>(smem + lane * 16);
        bf16x8 f[RING];
        int addr = lane * 16 + (wave & 3) * 8192;
#pragma unroll
        for (int i = 0; i < RING; ++i) f[i] = *reinterpret_cast<const bf16x8*>(smem + addr + (i & 7) * 1024);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.01f * (lane + j);
        for (int it = 0; it < n; it += 2 * RING) {
#pragma unroll
            for (int i = 0; i < 2 * RING; ++i) {
                if (LDS) {
                    int a2 = addr ^ ((i & 3) << 5);
                    asm volatile("" : "+v"(a2));
                    f[(i + 8) % RING] = *reinterpret_cast<const bf16x8*>(smem + a2 + ((i * 1024) & 4095));
                }
#pragma unroll
                for (int j = 0; j < KV; ++j) {
                    const int q = (i * KV + j) & 7;
                    if (TRANS && (j % 3) == 2) v[q] = __builtin_amdgcn_exp2f(v[q]);
                    else v[q] = fmaf(v[q], 0.999f, 0.001f);
                    asm volatile("" : "+v"(v[q]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(w), "v"(f[i % RING]));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(w), "v"(f[i % RING]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7");
        for (int i = 0; i < 16; ++i) res += acc0[i] + acc1[i];
        for (int j = 0; j < 8; ++j) res += v[j];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + tid] = res;
}

template <int KV, int LDS, int TRANS, int RING = 9>
void run(float* out, unsigned long long* cyc, int n) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KV, LDS, TRANS, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int nw = 4; nw <= 8; nw += 4) {
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<KV, LDS, TRANS, RING>), dim3(256), dim3(512), 65536, 0, out, cyc, n, nw);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0;
        for (int i = 0; i < 256; ++i) for (int w = 0; w < nw; ++w) a += h[i * 8 + w];
        a /= 256.0 * nw;
        printf("ring %d  KV=%d vector instr per MFMA%s%s, %d wave(s) per SIMD: %.1f cycles per step per wave, SIMD: one MFMA per %.1f cycles\n", RING, KV,
               LDS ? " + ds_read_b128" : "", TRANS ? " (1/3 transcendental)" : "", nw / 4, a / n, a / n / (nw / 4));
    }
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int n = 1980;
    run<0, 0, 0>(out, cyc, n);
    run<2, 0, 0>(out, cyc, n);
    run<4, 0, 0>(out, cyc, n);
    run<6, 0, 0>(out, cyc, n);
    run<8, 0, 0>(out, cyc, n);
    run<12, 0, 0>(out, cyc, n);
    run<0, 1, 0>(out, cyc, n);
    run<0, 1, 0, 10>(out, cyc, n);
    run<0, 1, 0, 11>(out, cyc, n);
    run<6, 1, 0>(out, cyc, n);
    run<6, 1, 0, 10>(out, cyc, n);
    run<6, 1, 0, 11>(out, cyc, n);
    run<6, 1, 1, 11>(out, cyc, n);
    run<6, 1, 1>(out, cyc, n);
    run<8, 1, 1>(out, cyc, n);
    run<12, 1, 1>(out, cyc, n);
    return 0;
}
