// Cross-attention from the f*h*w positions of a sample to that sample's previous-frame feature tokens
// (reference video_diffusion_pytorch/video_diffusion_pytorch.py:480-528; Unet3D wiring :803-805, :824-826).
//
// The keys and values depend only on the context, not on the activations, so they are projected once
// per context (one row GEMM for all layers, done by the caller with vdmi_tapgemm) and repacked here into
// MFMA operand order.  Per denoise step a layer then needs:  q = LayerNorm(x) Wq  (vdmi_tapgemm, QKV
// epilogue with q_cols = n),  this file's core kernel,  and the out-projection + bias + residual
// (vdmi_tapgemm again).
//
// Core kernel: one wave owns 32 query rows of one sample and walks the heads; per head
//   S^T [key x query] = K (A operand, from the pack) * Q^T (B operand, 16-byte loads of the q rows)
//   online softmax over the key tiles in the log2 domain (q is pre-multiplied by scale * log2 e)
//   O^T [d x query]  += V^T (A operand, keys in accumulator-row order) * P^T (the S^T registers themselves)
// so no LDS and no barrier is needed; the K/V pack of a sample (heads * ceil(L/32) * 4 KB) stays in L2.
// A lane ends with 16 features of its query row for the head; they are stored as 32 contiguous bytes at
// column head*32 + half*16, i.e. in a permuted feature order that the out-projection's packed weights undo.
#include "vdmi_common.h"

namespace {

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f2bf(a[8 * s + j]);
    return f;
}

// ctx fp32 [b][l][d] (times keep[b]) -> bf16 rows [b*l][d]; one thread per 8 values
__global__ __launch_bounds__(256) void xattn_ctx_kernel(const float* __restrict__ ctx, const float* __restrict__ keep,
                                                        __bf16* __restrict__ rows, long long n8, long long per_sample8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float k = keep ? keep[i / per_sample8] : 1.0f;
    const f32x4 a = *reinterpret_cast<const f32x4*>(ctx + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(ctx + i * 8 + 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = f2bf(a[e] * k); o[4 + e] = f2bf(b[e] * k); }
    *reinterpret_cast<bf16x8*>(rows + i * 8) = o;
}

// kv rows bf16 [b*l][ld] (per layer: hid K columns then hid V columns) -> operand fragments
// [layer][b][K,V][head][key tile][k-step][64 lanes][8]; keys >= l are zero.  One thread per fragment piece.
__global__ __launch_bounds__(256) void xattn_pack_kv_kernel(const __bf16* __restrict__ rows, int ld, __bf16* __restrict__ out,
                                                            int nlayers, int b, int l, int heads, int nt, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), s = (int)((i >> 6) & 1);
    long long r = i >> 7;
    const int t = (int)(r % nt); r /= nt;
    const int head = (int)(r % heads); r /= heads;
    const int kv = (int)(r & 1); r >>= 1;
    const int bi = (int)(r % b);
    const int layer = (int)(r / b);
    const int half = lane >> 5, l31 = lane & 31, hid = heads * 32;
    const long long col0 = (long long)layer * 2 * hid + (long long)kv * hid + head * 32;
    bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kv == 0) {
        // K as the A operand [key x d]: this lane's key, features s*16 + 8*half .. +7
        const int key = t * 32 + l31;
        if (key < l) o = *reinterpret_cast<const bf16x8*>(rows + ((long long)bi * l + key) * ld + col0 + s * 16 + 8 * half);
    } else {
        // V^T as the A operand [d x key]: this lane's feature, keys in the accumulator-row order of the scores
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = t * 32 + (j & 3) + 16 * s + 8 * (j >> 2) + 4 * half;
            if (key < l) o[j] = rows[((long long)bi * l + key) * ld + col0 + l31];
        }
    }
    *reinterpret_cast<bf16x8*>(out + i * 8) = o;
}

struct XCoreArgs {
    const __bf16* q;
    __bf16* o;
    const __bf16* kv;       // this layer: [b][2][heads][nt][2][64][8]
    long long rows_per_sample;
    int ldq, heads, l, nt, groups_per_sample, ngroups;
};

__global__ __launch_bounds__(256) void xattn_core_kernel(const XCoreArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int grp = blockIdx.x * 4 + wave;
    if (grp >= a.ngroups) return;                       // whole waves leave; the kernel has no barrier
    const int b = grp / a.groups_per_sample;
    const long long row = (long long)(grp - b * a.groups_per_sample) * 32 + l31;
    const bool valid = row < a.rows_per_sample;
    const long long grow = (long long)b * a.rows_per_sample + (valid ? row : 0);
    const int hid = a.heads * 32;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tail = a.l - (a.nt - 1) * 32;             // keys in the last tile (1..32)
    unsigned tailmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (acc_row(r, half) < tail) tailmask |= 1u << r;

    for (int head = 0; head < a.heads; ++head) {
        const __bf16* qp = a.q + grow * a.ldq + head * 32 + 8 * half;
        bf16x8 q0 = {0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0;
        if (valid) {
            q0 = *reinterpret_cast<const bf16x8*>(qp);
            q1 = *reinterpret_cast<const bf16x8*>(qp + 16);
        }
        const __bf16* kp = a.kv + (((long long)(b * 2 + 0) * a.heads + head) * a.nt) * 1024 + lane * 8;
        const __bf16* vp = a.kv + (((long long)(b * 2 + 1) * a.heads + head) * a.nt) * 1024 + lane * 8;
        float m = -1e30f, lsum = 0.f;
        f32x16 o = zero;
        for (int t = 0; t < a.nt; ++t) {
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kp + t * 1024);
            const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kp + t * 1024 + 512);
            const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(vp + t * 1024);
            const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(vp + t * 1024 + 512);
            f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, zero, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, st, 0, 0, 0);
            if (t == a.nt - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!((tailmask >> r) & 1u)) st[r] = -1e30f;      // padding keys: exp2 gives exact zeros
            }
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[r] - mnew);
                st[r] = p;
                ps += p;
            }
            lsum = lsum * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, acc_frag(st, 0), o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, acc_frag(st, 1), o, 0, 0, 0);
        }
        lsum += __shfl_xor(lsum, 32);
        const float inv = 1.0f / lsum;                  // every query sees at least one real key (l >= 1)
        if (valid) {
            __bf16* op = a.o + grow * hid + head * 32 + half * 16;
            *reinterpret_cast<bf16x8*>(op) = acc_frag(o * inv, 0);
            *reinterpret_cast<bf16x8*>(op + 8) = acc_frag(o * inv, 1);
        }
    }
}

}   // namespace

extern "C" size_t vdmi_xattn_kv_elems(int nlayers, int b, int l, int heads) {
    if (nlayers <= 0 || b <= 0 || l <= 0 || heads <= 0) return 0;
    return (size_t)nlayers * b * 2 * heads * ((l + 31) / 32) * 1024;
}

extern "C" int vdmi_xattn_context(const float* ctx, const float* keep, void* rows, int b, int l, int d, vdmi_stream stream) {
    if (!ctx || !rows) return VDMI_E_ARG;
    if (b <= 0 || l <= 0 || d <= 0 || d % 8) return VDMI_E_SHAPE;
    if (!aligned16(ctx) || !aligned16(rows)) return VDMI_E_ALIGN;
    const long long per8 = (long long)l * d / 8, n8 = per8 * b;
    hipLaunchKernelGGL(xattn_ctx_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       ctx, keep, static_cast<__bf16*>(rows), n8, per8);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_xattn_pack_kv(const void* kvrows, int ldkv, void* kvpack, int nlayers, int b, int l, int heads,
                                  vdmi_stream stream) {
    if (!kvrows || !kvpack) return VDMI_E_ARG;
    if (nlayers <= 0 || b <= 0 || l <= 0 || heads <= 0 || ldkv < nlayers * 2 * heads * 32 || ldkv % 8) return VDMI_E_SHAPE;
    if (!aligned16(kvrows) || !aligned16(kvpack)) return VDMI_E_ALIGN;
    const int nt = (l + 31) / 32;
    const long long total = (long long)nlayers * b * 2 * heads * nt * 128;
    if (total > 0x7fffffffLL * 256) return VDMI_E_SHAPE;
    hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const __bf16*>(kvrows), ldkv,
                       static_cast<__bf16*>(kvpack), nlayers, b, l, heads, nt, total);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_xattn_core(const void* q, int ldq, void* o, const void* kvpack_layer, int b, int64_t rows_per_sample,
                               int l, int heads, vdmi_stream stream) {
    if (!q || !o || !kvpack_layer) return VDMI_E_ARG;
    if (b <= 0 || rows_per_sample <= 0 || l <= 0 || heads <= 0 || ldq < heads * 32 || ldq % 8) return VDMI_E_SHAPE;
    if (!aligned16(q) || !aligned16(o) || !aligned16(kvpack_layer)) return VDMI_E_ALIGN;
    XCoreArgs a;
    a.q = static_cast<const __bf16*>(q);
    a.o = static_cast<__bf16*>(o);
    a.kv = static_cast<const __bf16*>(kvpack_layer);
    a.rows_per_sample = rows_per_sample;
    a.ldq = ldq;
    a.heads = heads;
    a.l = l;
    a.nt = (l + 31) / 32;
    const long long gps = (rows_per_sample + 31) / 32;
    if (gps * b > 0x7fffffffLL) return VDMI_E_SHAPE;
    a.groups_per_sample = (int)gps;
    a.ngroups = (int)(gps * b);
    hipLaunchKernelGGL(xattn_core_kernel, dim3((unsigned)((a.ngroups + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
