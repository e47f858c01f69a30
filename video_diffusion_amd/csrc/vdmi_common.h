// Shared device/host helpers for the vdmi kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include "../../include/vdmi.h"

// Precision build.  The default library stores activations and weights as bfloat16 and multiplies them with
// v_mfma_f32_32x32x16_bf16.  Built with -DVDMI_F16 (libvdmi_f16.so) every "bf16" in these sources is IEEE half
// instead: same storage bytes, same matrix-core rate, same fp32 accumulation, but 11 significant bits where bfloat16
// carries 8 (and a range of 65504, which the normalised activations of the denoiser stay far inside).  The type and
// the MFMA are switched here, once, for every kernel and for the host-side packers.
#ifdef VDMI_F16
#define __bf16 _Float16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#endif

// HOST: float -> the 16 bits of the library's storage type, round to nearest even (NaN stays NaN).
static inline uint16_t vdmi_host_half_bits(float f) {
#ifdef VDMI_F16
    const _Float16 hv = (_Float16)f;
    uint16_t r;
    __builtin_memcpy(&r, &hv, 2);
    return r;
#else
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
#endif
}
// HOST: the float value of those 16 bits
static inline float vdmi_host_half_value(uint16_t h) {
#ifdef VDMI_F16
    _Float16 hv;
    __builtin_memcpy(&hv, &h, 2);
    return (float)hv;
#else
    const uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

// Largest power of two that the site-mask trick of the fused temporal attention may put into a 16-bit operand: a key of
// another site scores -(VDMI_MASK_BIG^2), which exp2 turns into an exact zero (2^50 in bfloat16; IEEE half ends at 65504).
#ifdef VDMI_F16
#define VDMI_MASK_BIG 32768.f
#else
#define VDMI_MASK_BIG 1125899906842624.f
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VDMI_WAVE 64

__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float v) { return (__bf16)v; }   // v_cvt_pk_bf16_f32, RNE

// y * sigmoid(y) with the hardware exp2 / rcp (1 ulp-class approximations; the result is rounded to bf16)
__device__ __forceinline__ float silu_f(float y) {
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
}

// Sum / maximum of a value with the lane 32 away (the other half of the wave64), the same in both lanes.
// v_permlane32_swap is a vector instruction; __shfl_xor(x, 32) compiles to ds_bpermute_b32, an LDS round trip whose
// wait (lgkmcnt) also waits for every LDS read in flight.
__device__ __forceinline__ float xhalf_add(float v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ float xhalf_max(float v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// Sum over the aligned group of N consecutive lanes (N = 2, 4, 8 or 16), the same value in all of them, with DPP operand
// modifiers (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of the ds_bpermute round
// trips of __shfl_xor.  After two steps a quad holds one value, so mirroring within 8 and within 16 lanes pairs the same
// two partial sums a butterfly with lane ^ 4 / lane ^ 8 would: the result has the same bits.
template <int CTRL>
__device__ __forceinline__ float vdmi_dpp(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float vdmi_row_sum(float v) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16, "lanes of one DPP row");
    if constexpr (N >= 2) v += vdmi_dpp<0xB1>(v);
    if constexpr (N >= 4) v += vdmi_dpp<0x4E>(v);
    if constexpr (N >= 8) v += vdmi_dpp<0x141>(v);
    if constexpr (N >= 16) v += vdmi_dpp<0x140>(v);
    return v;
}

// The same over an aligned group of N = 2 .. 64 lanes: DPP inside a row of 16, shuffles / the half swap across rows.
template <int N>
__device__ __forceinline__ float vdmi_group_sum(float v) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "power of two");
    v = vdmi_row_sum<(N < 16 ? N : 16)>(v);
    if constexpr (N >= 32) v += __shfl_xor(v, 16);
    if constexpr (N >= 64) v = xhalf_add(v);
    return v;
}

// hipFuncSetAttribute acts on the current device only, so "already done" is one bit per device in a mask that each
// call site keeps for its kernel.  Returns 0 when the kernel may use `bytes` of dynamic LDS on the current device.
static inline int vdmi_set_max_lds(const void* fn, std::atomic<uint64_t>& done_mask, int bytes) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    const uint64_t bit = 1ull << dev;
    if (done_mask.load(std::memory_order_acquire) & bit) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
    done_mask.fetch_or(bit, std::memory_order_release);
    return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define VDMI_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return VDMI_E_LAUNCH;          \
    } while (0)
