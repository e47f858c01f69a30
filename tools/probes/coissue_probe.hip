// Probe (tool, not shipped): how fast does a VALU-bound wave run beside an MFMA-bound wave on the same SIMD?
// 512-thread workgroups, one per CU.  Waves 0-3 ("A") run NA steps of {optional ds_read_b128 + address VALU, MFMA,
// optional s_nop}; waves 4-7 ("B") run NB steps of an epilogue-like VALU body.  Each wave reports its cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int AMODE, int BMODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int na, int nb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    float res = 0.f;
    if (wave < 4) {
        if (AMODE > 0) {
            f32x16 acc0 = {0}, acc1 = {0};
            bf16x8 w = *reinterpret_cast<const bf16x8*>(smem + lane * 16);
            bf16x8 wa = w;
            asm volatile("" : "+a"(wa));
            bf16x8 f[9];
            int addr = lane * 16 + wave * 8192;
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const bf16x8*>(smem + addr + i * 1024);
            for (int it = 0; it < na; it += 18) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    if (AMODE >= 2) {
                        int a2 = addr ^ ((i & 3) << 5);
                        asm volatile("" : "+v"(a2));
                        f[(i + 8) % 9] = *reinterpret_cast<const bf16x8*>(smem + a2 + ((i * 1024) & 4095));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (AMODE == 5) {      // inline asm, A operand in an accumulation register
                        if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "a"(wa), "v"(f[i % 9]));
                        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "a"(wa), "v"(f[i % 9]));
                    } else if (AMODE == 6) {   // inline asm, A operand in a vector register
                        if (i & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(w), "v"(f[i % 9]));
                        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(w), "v"(f[i % 9]));
                    } else if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, f[i % 9], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, f[i % 9], acc0, 0, 0, 0);
                    if (AMODE == 3) asm volatile("s_nop 3");
                    if (AMODE == 4) asm volatile("s_nop 15");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (int i = 0; i < 16; ++i) res += acc0[i] + acc1[i];
        }
    } else {
        if (BMODE > 0) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.01f * (lane + j);
            for (int it = 0; it < nb; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float y = fmaf(v[j], 0.999f, 0.001f);
                    if (BMODE == 2) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * y));
                    else y = fmaf(y, 1.0001f, 0.5f) * 0.7f + 0.1f;
                    v[j] = y;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) res += v[j];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + tid] = res;
}

template <int A, int B>
void run(const char* name, float* out, unsigned long long* cyc, int na, int nb) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<A, B>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 65536, 0, out, cyc, na, nb);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < 256; ++i) { for (int w = 0; w < 4; ++w) a += h[i * 8 + w]; for (int w = 4; w < 8; ++w) b += h[i * 8 + w]; }
    printf("%-52s A waves %8.0f cycles (%.1f / MFMA step)   B waves %8.0f cycles (%.1f / body of 8)\n", name, a / 1024, a / 1024 / na,
           b / 1024, b / 1024 / nb);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int na = 1800, nb = 600;
    run<1, 0>("A: MFMA only, alone", out, cyc, na, nb);
    run<2, 0>("A: MFMA + ds_read + xor, alone", out, cyc, na, nb);
    run<5, 0>("A: asm MFMA (A operand in AGPR) + ds_read, alone", out, cyc, na, nb);
    run<6, 0>("A: asm MFMA (A operand in VGPR) + ds_read, alone", out, cyc, na, nb);
    run<0, 1>("B: fma body, alone", out, cyc, na, nb);
    run<0, 2>("B: silu body, alone", out, cyc, na, nb);
    run<1, 1>("A MFMA only | B fma", out, cyc, na, nb);
    run<1, 2>("A MFMA only | B silu", out, cyc, na, nb);
    run<2, 1>("A MFMA+ds_read | B fma", out, cyc, na, nb);
    run<2, 2>("A MFMA+ds_read | B silu", out, cyc, na, nb);
    run<3, 2>("A MFMA+ds_read+s_nop 3 | B silu", out, cyc, na, nb);
    run<4, 2>("A MFMA+ds_read+s_nop 15 | B silu", out, cyc, na, nb);
    return 0;
}
