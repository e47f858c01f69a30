// Attention cores for gfx950.
//  * vdmi_attn_core   — softmax attention over a small token axis (frames, or the 8x8 bottleneck
//                       positions); one lane per (row, head), online softmax in fp32.
//  * vdmi_sla_context — SpatialLinearAttention pass 1: softmax of k over positions, 32x32 context
//                       per (frame, head) on the matrix cores, folded into to_out's weights.
#include "vdmi_common.h"

namespace {

__global__ __launch_bounds__(256) void attn_core_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ out,
                                                        long long rows, int heads, int n_tok, long long tok_stride,
                                                        const float* __restrict__ bias, const uint8_t* __restrict__ focus,
                                                        long long rows_per_sample) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long m = gid / heads;
    const int hd = (int)(gid - m * heads);
    if (m >= rows) return;
    const int hid = heads * 32, ld = 3 * hid;
    const long long span = (long long)n_tok * tok_stride;
    const long long outer = m / span, rem = m - outer * span;
    const int i = (int)(rem / tok_stride);
    const long long base = outer * span + (rem - (long long)i * tok_stride);

    float q[32], o[32];
    {
        const bf16x8* qp = reinterpret_cast<const bf16x8*>(qkv + m * ld + hd * 32);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bf16x8 t = qp[v];
#pragma unroll
            for (int e = 0; e < 8; ++e) q[v * 8 + e] = (float)t[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) o[e] = 0.f;
    float mx = -INFINITY, l = 0.f;
    const float* brow = bias ? bias + ((long long)hd * n_tok + i) * n_tok : nullptr;
    const bool self_only = focus && focus[m / rows_per_sample];   // vdp.py:590-600: every other key is masked out
    for (int j = 0; j < n_tok; ++j) {
        if (self_only && j != i) continue;
        const long long mj = base + (long long)j * tok_stride;
        const bf16x8* kp = reinterpret_cast<const bf16x8*>(qkv + mj * ld + hid + hd * 32);
        const bf16x8* vp = reinterpret_cast<const bf16x8*>(qkv + mj * ld + 2 * hid + hd * 32);
        float s = brow ? brow[j] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bf16x8 t = kp[v];
#pragma unroll
            for (int e = 0; e < 8; ++e) s += q[v * 8 + e] * (float)t[e];
        }
        const float mn = fmaxf(mx, s);
        const float corr = __expf(mx - mn), p = __expf(s - mn);
        l = l * corr + p;
        mx = mn;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bf16x8 t = vp[v];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[v * 8 + e] = o[v * 8 + e] * corr + p * (float)t[e];
        }
    }
    const float inv = 1.0f / l;
    bf16x8* op = reinterpret_cast<bf16x8*>(out + m * hid + hd * 32);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = f2bf(o[v * 8 + e] * inv);
        op[v] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Same attention with the keys and values of a set of token groups staged once in LDS (the gather
// version above re-reads every k / v row n_tok times through L2).  A workgroup owns `gpb` groups
// (spatial attention: one frame's h*w tokens; temporal: gpb neighbouring sites x F frames); set row
// r = g * n_tok + i.  Thread = (set row, head): lanes of a wave walk consecutive rows of one head, so
// the k / v reads of a group are LDS broadcasts.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_core_lds_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ out,
                                                            long long rows, int heads, int n_tok, long long tok_stride,
                                                            const float* __restrict__ bias, int gpb, long long ngroups,
                                                            const uint8_t* __restrict__ focus, long long rows_per_sample) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int hid = heads * 32, ld = 3 * hid;
    const int krow = 2 * hid + 8;                        // padded k|v row (elements)
    __bf16* kv = reinterpret_cast<__bf16*>(smem);         // [gpb * n_tok][krow]
    const int tid = threadIdx.x;
    const long long g0 = (long long)blockIdx.x * gpb;
    const int ng = (int)min((long long)gpb, ngroups - g0);
    const int R = ng * n_tok;
    // group index -> first global row: groups are (outer, inner) with inner = position inside a tok_stride run
    auto group_base = [&](long long g) -> long long {
        const long long outer = g / tok_stride, inner = g - outer * tok_stride;
        return outer * n_tok * tok_stride + inner;
    };
    const int ppr = 2 * hid / 8;                          // 16-byte pieces of a k|v row
    for (int idx = tid; idx < R * ppr; idx += 256) {
        const int r = idx / ppr, pc = idx - r * ppr;
        const int g = r / n_tok, i = r - g * n_tok;
        const long long m = group_base(g0 + g) + (long long)i * tok_stride;
        *reinterpret_cast<bf16x8*>(kv + r * krow + pc * 8) = *reinterpret_cast<const bf16x8*>(qkv + m * ld + hid + pc * 8);
    }
    __syncthreads();
    for (int p = tid; p < R * heads; p += 256) {
        const int hd = p / R, r = p - hd * R;
        const int g = r / n_tok, i = r - g * n_tok;
        const long long m = group_base(g0 + g) + (long long)i * tok_stride;
        float q[32], o[32];
        {
            const bf16x8* qp = reinterpret_cast<const bf16x8*>(qkv + m * ld + hd * 32);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bf16x8 t = qp[v];
#pragma unroll
                for (int e = 0; e < 8; ++e) q[v * 8 + e] = (float)t[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) o[e] = 0.f;
        float mx = -INFINITY, l = 0.f;
        const float* brow = bias ? bias + ((long long)hd * n_tok + i) * n_tok : nullptr;
        const __bf16* kb = kv + (g * n_tok) * krow + hd * 32;
        const bool self_only = focus && focus[m / rows_per_sample];
        for (int j = 0; j < n_tok; ++j) {
            if (self_only && j != i) continue;
            const bf16x8* kp = reinterpret_cast<const bf16x8*>(kb + j * krow);
            const bf16x8* vp = reinterpret_cast<const bf16x8*>(kb + j * krow + hid);
            float sc = brow ? brow[j] : 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bf16x8 t = kp[v];
#pragma unroll
                for (int e = 0; e < 8; ++e) sc += q[v * 8 + e] * (float)t[e];
            }
            const float mn = fmaxf(mx, sc);
            const float corr = __expf(mx - mn), pw = __expf(sc - mn);
            l = l * corr + pw;
            mx = mn;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bf16x8 t = vp[v];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[v * 8 + e] = o[v * 8 + e] * corr + pw * (float)t[e];
            }
        }
        const float inv = 1.0f / l;
        bf16x8* op = reinterpret_cast<bf16x8*>(out + m * hid + hd * 32);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = f2bf(o[v * 8 + e] * inv);
            op[v] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// SLA context.  One workgroup per (frame, head); 4 waves split the positions.
// ------------------------------------------------------------------------------------------
constexpr int SLA_CH = 256;            // positions per chunk
constexpr int SLA_ROW = SLA_CH + 8;    // padded LDS row (elements)

__global__ __launch_bounds__(256) void sla_context_kernel(const __bf16* __restrict__ kvT, int heads, int hw,
                                                          const float* __restrict__ w_out, int c, int cpad, int bkc,
                                                          __bf16* __restrict__ w_folded) {
    __shared__ __attribute__((aligned(16))) __bf16 Ek[32 * SLA_ROW];
    __shared__ __attribute__((aligned(16))) __bf16 Vv[32 * SLA_ROW];
    __shared__ __attribute__((aligned(16))) float ctx[4][32][33];
    __shared__ float rowmax[32], rowsum[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frame = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int hid = heads * 32;
    const int r = tid >> 3, part = tid & 7;        // loader: row r, 8 threads per row
    const __bf16* krow = kvT + ((long long)frame * 2 * hid + hd * 32 + r) * hw;
    const __bf16* vrow = kvT + ((long long)frame * 2 * hid + hid + hd * 32 + r) * hw;
    const bool vec = (hw % 8) == 0;

    // pass A: max over positions of each k row
    float mx = -INFINITY;
    if (vec) {
        for (int n = part * 8; n < hw; n += 64) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(krow + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)t[e]);
        }
    } else {
        for (int n = part; n < hw; n += 8) mx = fmaxf(mx, (float)krow[n]);
    }
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) mx = fmaxf(mx, __shfl_xor(mx, s));

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float lsum = 0.f;

    for (int n0 = 0; n0 < hw; n0 += SLA_CH) {
        __syncthreads();
        // each loader thread fills 32 consecutive positions of its row in both tiles
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int nl = part * 32 + v * 8, n = n0 + nl;
            bf16x8 kk, vv;
            if (vec && n + 8 <= hw) {
                kk = *reinterpret_cast<const bf16x8*>(krow + n);
                vv = *reinterpret_cast<const bf16x8*>(vrow + n);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = n + e < hw;
                    kk[e] = ok ? krow[n + e] : (__bf16)(-INFINITY);
                    vv[e] = ok ? vrow[n + e] : (__bf16)0.f;
                }
            }
            bf16x8 ee;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 x = f2bf(__expf((float)kk[e] - mx));
                ee[e] = x;
                lsum += (float)x;
            }
            *reinterpret_cast<bf16x8*>(Ek + r * SLA_ROW + nl) = ee;
            *reinterpret_cast<bf16x8*>(Vv + r * SLA_ROW + nl) = vv;
        }
        __syncthreads();
        // ctx[d][e] += sum_n E[d][n] * V[e][n]; this wave takes positions [wave*64, wave*64+64)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = (lane & 31) * SLA_ROW + wave * 64 + kk * 16 + 8 * (lane >> 5);
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ek + off);
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(Vv + off);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) lsum += __shfl_xor(lsum, s);
    if (part == 0) { rowsum[r] = lsum; rowmax[r] = mx; }
#pragma unroll
    for (int i = 0; i < 16; ++i) ctx[wave][(i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)][lane & 31] = acc[i];
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        const int d = idx >> 5, e = idx & 31;
        ctx[0][d][e] = (ctx[0][d][e] + ctx[1][d][e] + ctx[2][d][e] + ctx[3][d][e]) / rowsum[d];
    }
    __syncthreads();

    // Fold into the output projection, W'[co][h*32+d] = sum_e Wout[co][h*32+e] * ctx[d][e], on the matrix
    // cores and write it in vdmi_pack_taps order (1 tap, cin = heads*32).  Both factors are split into
    // bf16 high and low parts (three products), so the fold stays within ~2^-16 of fp32 before its one
    // rounding to bf16.  Orientation: ctx is the A operand (rows d), Wout the B operand (columns co): a
    // lane then holds 4 consecutive d of one output channel per register quad = one 8-byte store.
    {
        const int l31 = lane & 31, hf = lane >> 5;
        bf16x8 ch[2], cl[2];     // A fragments: row d = l31, k = e = 16 s + 8 hf + j
#pragma unroll
        for (int sk = 0; sk < 2; ++sk)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = ctx[0][l31][16 * sk + 8 * hf + j];
                const __bf16 hi = f2bf(v);
                ch[sk][j] = hi;
                cl[sk][j] = f2bf(v - (float)hi);
            }
        __bf16* wf = w_folded + (long long)frame * hid * cpad;
        const int ksper = bkc / 16, ctn = cpad / 32;
        for (int ct = wave; ct < c / 32; ct += 4) {
            const int co = ct * 32 + l31;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc = zero, acc1 = zero, acc2 = zero;
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                const float* wr = w_out + (long long)co * hid + hd * 32 + 16 * sk + 8 * hf;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
                bf16x8 wh, wl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = j < 4 ? w0[j & 3] : w1[j & 3];
                    const __bf16 hi = f2bf(v);
                    wh[j] = hi;
                    wl[j] = f2bf(v - (float)hi);
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[sk], wh, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[sk], wl, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[sk], wh, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc1[r] + acc2[r];
            // registers 4u..4u+3 hold d = 8u + 4 hf + {0..3} of channel co
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = hd * 32 + 8 * u + 4 * hf;
                const int chunk = k / bkc, kin = k % bkc, ks = kin >> 4, half = (kin >> 3) & 1;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[4 * u + e]);
                *reinterpret_cast<bf16x4*>(wf + ((((long long)chunk * ctn + ct) * ksper + ks) * 64 + half * 32 + l31) * 8 +
                                           (kin & 7)) = o;
            }
        }
    }
}

}   // namespace

extern "C" int vdmi_attn_core(const void* qkv, void* out, int64_t rows, int heads, int n_tok, int64_t tok_stride,
                              const float* bias, const uint8_t* focus, int64_t rows_per_sample, vdmi_stream stream) {
    if (!qkv || !out) return VDMI_E_ARG;
    if (focus && (rows_per_sample <= 0 || rows % rows_per_sample)) return VDMI_E_SHAPE;
    if (!focus) rows_per_sample = rows > 0 ? rows : 1;
    if (rows <= 0 || heads <= 0 || n_tok <= 0 || tok_stride <= 0 || rows % ((int64_t)n_tok * tok_stride))
        return VDMI_E_SHAPE;
    if (!aligned16(qkv) || !aligned16(out)) return VDMI_E_ALIGN;
    {   // keys / values staged in LDS when a useful number of groups fits
        const long long ngroups = rows / n_tok;
        const size_t row_bytes = (size_t)(2 * heads * 32 + 8) * 2;
        // one pass of the 256 threads over a workgroup's (row, head) pairs: each thread walks its keys serially, so
        // the launch is as long as a thread's passes; more, smaller workgroups finish sooner (8x8 level: 86 -> 171)
        int gpb = (256 / heads) / n_tok;
        if (gpb < 1) gpb = 1;
        const size_t lds = (size_t)gpb * n_tok * row_bytes;
        if (lds <= 150 * 1024) {
            static std::atomic<uint64_t> attr_mask{0};
            if (vdmi_set_max_lds(reinterpret_cast<const void*>(attn_core_lds_kernel), attr_mask, 160 * 1024))
                return VDMI_E_LAUNCH;
            const long long nb = (ngroups + gpb - 1) / gpb;
            if (nb > 0x7fffffffLL) return VDMI_E_SHAPE;
            hipLaunchKernelGGL(attn_core_lds_kernel, dim3((unsigned)nb), dim3(256), lds, static_cast<hipStream_t>(stream),
                               static_cast<const __bf16*>(qkv), static_cast<__bf16*>(out), (long long)rows, heads, n_tok,
                               (long long)tok_stride, bias, gpb, ngroups, focus, (long long)rows_per_sample);
            VDMI_LAUNCH_CHECK();
            return VDMI_OK;
        }
    }
    const long long threads = (long long)rows * heads;
    const long long blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return VDMI_E_SHAPE;
    hipLaunchKernelGGL(attn_core_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(qkv), static_cast<__bf16*>(out), (long long)rows, heads, n_tok,
                       (long long)tok_stride, bias, focus, (long long)rows_per_sample);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_sla_context(const void* kvT, int frames, int heads, int hw, const float* w_out, int c, int cpad,
                                int bkc, void* w_folded, vdmi_stream stream) {
    if (!kvT || !w_out || !w_folded) return VDMI_E_ARG;
    if (frames <= 0 || heads <= 0 || hw <= 0 || c <= 0 || cpad < c || cpad % 32 || (bkc != 32 && bkc != 64) ||
        (heads * 32) % bkc)
        return VDMI_E_SHAPE;
    if (!aligned16(kvT) || !aligned16(w_folded) || !aligned16(w_out)) return VDMI_E_ALIGN;
    hipLaunchKernelGGL(sla_context_kernel, dim3((unsigned)(frames * heads)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const __bf16*>(kvT), heads, hw, w_out, c, cpad,
                       bkc, static_cast<__bf16*>(w_folded));
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
