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
// A workgroup is 8 such waves of one sample; it stages the keys/values of a (head, 8 key tiles) step in LDS once
// (double buffered, one barrier per step), so the pack is read from L2 once per 256 queries, not once per 32.
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
// kinds = 4 (training): two more operand forms follow K and V^T per sample - V as an A operand [key x feature] whose
// contraction index runs over the features in the order the core kernel stores them (column 16*hh + r of a head holds
// feature (r&3) + 8*(r>>2) + 4*hh), and K^T as an A operand [feature x key] with the keys in accumulator-row order.
__global__ __launch_bounds__(256) void xattn_pack_kv_kernel(const __bf16* __restrict__ rows, int ld, __bf16* __restrict__ out,
                                                            int nlayers, int b, int l, int heads, int nt, long long total,
                                                            int kinds) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63), s = (int)((i >> 6) & 1);
    long long r = i >> 7;
    const int t = (int)(r % nt); r /= nt;
    const int head = (int)(r % heads); r /= heads;
    const int kind = (int)(r % kinds); r /= kinds;
    const int kv = (kind == 1 || kind == 2) ? 1 : 0;        // which half of the row: keys or values
    const int bi = (int)(r % b);
    const int layer = (int)(r / b);
    const int half = lane >> 5, l31 = lane & 31, hid = heads * 32;
    const long long col0 = (long long)layer * 2 * hid + (long long)kv * hid + head * 32;
    bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kind == 0) {
        // K as the A operand [key x d]: this lane's key, features s*16 + 8*half .. +7
        const int key = t * 32 + l31;
        if (key < l) o = *reinterpret_cast<const bf16x8*>(rows + ((long long)bi * l + key) * ld + col0 + s * 16 + 8 * half);
    } else if (kind == 2) {
        // V as the A operand [key x stored column]: contraction index s*16 + 8*half + j = stored column 16*s + (8*half + j)
        const int key = t * 32 + l31;
        if (key < l) {
            const __bf16* src = rows + ((long long)bi * l + key) * ld + col0 + 16 * half + 4 * s;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = src[(j & 3) + 8 * (j >> 2)];
        }
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
    int ldq, heads, l, nt, groups_per_sample, blocks_per_sample;
    int kinds;              // operand forms per sample in the pack: 2 (K, V^T) or 4 (the training pack)
    float* lse;             // [b][heads][lse_stride] or null: log2-domain log-sum-exp of every (query, head) for the backward
    long long lse_stride;
};

constexpr int XC_WAVES = 8;          // waves (= 32-query groups) per workgroup, all of one sample
constexpr int XC_CH = 8;             // key tiles staged per step: 8 * 4 KB, double buffered
constexpr int XC_PP = XC_CH * 256 / (XC_WAVES * 64);   // 16-byte pieces per thread per step

__global__ __launch_bounds__(XC_WAVES * 64) void xattn_core_kernel(const XCoreArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 kvbuf[2][XC_CH * 2048];   // per tile: K (2 fragments), V (2 fragments)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / a.blocks_per_sample;
    const int grp = (blockIdx.x - b * a.blocks_per_sample) * XC_WAVES + wave;      // 32-row group inside the sample
    const long long row = (long long)grp * 32 + l31;
    const bool valid = grp < a.groups_per_sample && row < a.rows_per_sample;
    const long long grow = (long long)b * a.rows_per_sample + (valid ? row : 0);
    const int hid = a.heads * 32;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tail = a.l - (a.nt - 1) * 32;             // keys in the last tile (1..32)
    unsigned tailmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (acc_row(r, half) < tail) tailmask |= 1u << r;

    // staging: step = (head, chunk of XC_CH key tiles); the keys/values of a step are read from L2 once per
    // workgroup (not once per wave), global -> registers during the previous step, registers -> LDS after it
    const int nchunks = (a.nt + XC_CH - 1) / XC_CH, nsteps = a.heads * nchunks;
    bf16x8 pre[XC_PP];
    auto prefetch = [&](int step) {
        const int head = step / nchunks, t0 = (step - head * nchunks) * XC_CH;
        const int cnt = min(XC_CH, a.nt - t0);
        const __bf16* kb = a.kv + (((long long)(b * a.kinds + 0) * a.heads + head) * a.nt + t0) * 1024;
        const __bf16* vb = a.kv + (((long long)(b * a.kinds + 1) * a.heads + head) * a.nt + t0) * 1024;
#pragma unroll
        for (int p = 0; p < XC_PP; ++p) {
            const int idx = tid + p * (XC_WAVES * 64);
            const int tt = idx >> 8, within = idx & 255;
            if (tt < cnt)
                pre[p] = *reinterpret_cast<const bf16x8*>((within < 128 ? kb : vb - 1024) + tt * 1024 + within * 8);
        }
    };
    auto commit = [&](int step) {
        const int head = step / nchunks, t0 = (step - head * nchunks) * XC_CH;
        const int cnt = min(XC_CH, a.nt - t0);
        __bf16* dst = kvbuf[step & 1];
#pragma unroll
        for (int p = 0; p < XC_PP; ++p) {
            const int idx = tid + p * (XC_WAVES * 64);
            if ((idx >> 8) < cnt) *reinterpret_cast<bf16x8*>(dst + idx * 8) = pre[p];
        }
    };

    prefetch(0);
    int step = 0;
    for (int head = 0; head < a.heads; ++head) {
        const __bf16* qp = a.q + grow * a.ldq + head * 32 + 8 * half;
        bf16x8 q0 = {0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0;
        if (valid) {
            q0 = *reinterpret_cast<const bf16x8*>(qp);
            q1 = *reinterpret_cast<const bf16x8*>(qp + 16);
        }
        float m = 0.f, lsum = 0.f;          // the first tile of a head always sets m to its own maximum
        f32x16 o = zero, negm = zero;
        for (int c = 0; c < nchunks; ++c, ++step) {
            commit(step);
            __syncthreads();      // this step's tiles are visible; every wave is done with the buffer the NEXT commit overwrites
            if (step + 1 < nsteps) prefetch(step + 1);
            const int t0 = c * XC_CH, cnt = min(XC_CH, a.nt - t0);
            const __bf16* buf = kvbuf[step & 1] + lane * 8;
            for (int tt = 0; tt < cnt; ++tt) {
                const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(buf + tt * 2048);
                const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(buf + tt * 2048 + 512);
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(buf + tt * 2048 + 1024);
                const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(buf + tt * 2048 + 1536);
                // scores relative to the running maximum: the accumulator starts at -m (negm is refreshed only when m moves)
                f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, negm, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, st, 0, 0, 0);
                if (t0 + tt == a.nt - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (!((tailmask >> r) & 1u)) st[r] = -1e30f;      // padding keys: exp2 gives exact zeros
                }
                float mx = st[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
                mx = xhalf_max(mx);
                // some query's maximum grew (rare after the first tiles; wave-uniform branch): move m, rescale
                const bool first = (t0 + tt == 0);
                if (first || __builtin_amdgcn_ballot_w64(mx > 0.f) != 0) {
                    const float delta = first ? mx : fmaxf(mx, 0.f);
                    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);      // o and lsum are still zero on the first tile
                    m += delta;
                    lsum *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o[r] *= alpha;
                        st[r] -= delta;
                        negm[r] = -m;
                    }
                }
                float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(st[r]), p1 = __builtin_amdgcn_exp2f(st[r + 1]);
                    st[r] = p0;
                    st[r + 1] = p1;
                    ps0 += p0;
                    ps1 += p1;
                }
                lsum += ps0 + ps1;
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, acc_frag(st, 0), o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, acc_frag(st, 1), o, 0, 0, 0);
            }
        }
        lsum = xhalf_add(lsum);
        const float inv = 1.0f / lsum;                  // every query sees at least one real key (l >= 1)
        if (valid) {
            __bf16* op = a.o + grow * hid + head * 32 + half * 16;
            *reinterpret_cast<bf16x8*>(op) = acc_frag(o * inv, 0);
            *reinterpret_cast<bf16x8*>(op + 8) = acc_frag(o * inv, 1);
            if (a.lse && half == 0) a.lse[((long long)b * a.heads + head) * a.lse_stride + row] = m + __builtin_amdgcn_logf(lsum);
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

static int xattn_pack(const void* kvrows, int ldkv, void* kvpack, int nlayers, int b, int l, int heads, int kinds,
                      vdmi_stream stream);

extern "C" int vdmi_xattn_pack_kv(const void* kvrows, int ldkv, void* kvpack, int nlayers, int b, int l, int heads,
                                  vdmi_stream stream) {
    return xattn_pack(kvrows, ldkv, kvpack, nlayers, b, l, heads, 2, stream);
}

// Training pack of one layer: 2 * vdmi_xattn_kv_elems(1, b, l, heads) elements, [b][K, V^T, V, K^T][head][tile][2][64][8].
extern "C" int vdmi_xattn_pack_kv_train(const void* kvrows, int ldkv, void* kvpack, int b, int l, int heads,
                                        vdmi_stream stream) {
    return xattn_pack(kvrows, ldkv, kvpack, 1, b, l, heads, 4, stream);
}

static int xattn_pack(const void* kvrows, int ldkv, void* kvpack, int nlayers, int b, int l, int heads, int kinds,
                      vdmi_stream stream) {
    if (!kvrows || !kvpack) return VDMI_E_ARG;
    if (nlayers <= 0 || b <= 0 || l <= 0 || heads <= 0 || ldkv < nlayers * 2 * heads * 32 || ldkv % 8) return VDMI_E_SHAPE;
    if (!aligned16(kvrows) || !aligned16(kvpack)) return VDMI_E_ALIGN;
    const int nt = (l + 31) / 32;
    const long long total = (long long)nlayers * b * kinds * heads * nt * 128;
    if (total > 0x7fffffffLL * 256) return VDMI_E_SHAPE;
    hipLaunchKernelGGL(xattn_pack_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const __bf16*>(kvrows), ldkv,
                       static_cast<__bf16*>(kvpack), nlayers, b, l, heads, nt, total, kinds);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

static int xattn_core_launch(const void* q, int ldq, void* o, const void* kvpack_layer, int b, int64_t rows_per_sample,
                             int l, int heads, int kinds, float* lse, int64_t lse_stride, vdmi_stream stream);

extern "C" int vdmi_xattn_core(const void* q, int ldq, void* o, const void* kvpack_layer, int b, int64_t rows_per_sample,
                               int l, int heads, vdmi_stream stream) {
    return xattn_core_launch(q, ldq, o, kvpack_layer, b, rows_per_sample, l, heads, 2, nullptr, 0, stream);
}

// The same core over a training pack (vdmi_xattn_pack_kv_train); also writes lse float [b][heads][lse_stride]
// (lse_stride >= rows_per_sample): log2 of the softmax denominator plus the running maximum, in the log2 domain of q.
extern "C" int vdmi_xattn_core_train_fwd(const void* q, int ldq, void* o, const void* kvpack_train, float* lse,
                                         int64_t lse_stride, int b, int64_t rows_per_sample, int l, int heads,
                                         vdmi_stream stream) {
    if (!lse || lse_stride < rows_per_sample) return VDMI_E_ARG;
    return xattn_core_launch(q, ldq, o, kvpack_train, b, rows_per_sample, l, heads, 4, lse, lse_stride, stream);
}

static int xattn_core_launch(const void* q, int ldq, void* o, const void* kvpack_layer, int b, int64_t rows_per_sample,
                             int l, int heads, int kinds, float* lse, int64_t lse_stride, vdmi_stream stream) {
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
    a.kinds = kinds;
    a.lse = lse;
    a.lse_stride = lse_stride;
    const long long gps = (rows_per_sample + 31) / 32;
    const long long bps = (gps + XC_WAVES - 1) / XC_WAVES;
    if (gps > 0x7fffffffLL || bps * b > 0x7fffffffLL) return VDMI_E_SHAPE;
    a.groups_per_sample = (int)gps;
    a.blocks_per_sample = (int)bps;
    hipLaunchKernelGGL(xattn_core_kernel, dim3((unsigned)(bps * b)), dim3(XC_WAVES * 64), 0,
                       static_cast<hipStream_t>(stream), a);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
