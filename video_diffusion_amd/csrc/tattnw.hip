// Fused temporal attention for the wide layers (C = 256, 512) of small images: Residual(PreNorm(Attention over
// frames)) of the reference (vdp.py:331-337, 380-408, 531-612, 636) in ONE launch instead of three (QKV GEMM,
// attention core, output GEMM).
//
// Same arithmetic and the same packed weights as tattn.hip (vdmi_pack_tattn); what differs is who holds what.
// At these widths a row of x does not fit a wave's registers next to the output accumulators, and a head's weight
// slice (96 x C) does not fit LDS twice, so:
//   * a workgroup owns NT = 2 groups of 32 rows (a group = floor(32 / F) sites x F frames, as in tattn.hip) and has
//     one wave per head;
//   * phase 0: the 64 rows are loaded once, LayerNormed (8 lanes per row) and kept in LDS as bf16;
//   * phase 1: wave h streams ITS head's Wq, Wk, Wv fragments straight from global memory (L2) into registers, a
//     few k-steps ahead - weights never pass through LDS and no barrier separates the k-steps - and multiplies
//     them with the row fragments read from LDS: Q^T, K^T (rows on lanes) and V (d on lanes) of both groups;
//   * phase 2: rotary, scores, softmax over the frames of each site, P.V per group, exactly the register chain of
//     tattn.hip; the normalised O^T goes to LDS in the k-slot order the packed to_out fragments expect;
//   * phase 3 (after the one barrier of the kernel's main part): wave w multiplies all heads' O with the to_out
//     fragments of ITS output-channel tiles (again streamed from global), adds the residual and stores 16 B per lane.
#include "vdmi_common.h"
#include <cstring>

namespace {

struct TWArgs {
    const __bf16* x;
    __bf16* y;
    const __bf16* w;        // vdmi_pack_tattn: [heads][3*KS + 2*CT][64][8]
    const float* bias;      // [heads][F][F] or null
    const float* rot_cos;   // [F][16] or null
    const float* rot_sin;
    const uint8_t* focus;   // [B] or null
    const float* b_out;     // [C] or null: bias of to_out (the linear-attention layer has one)
    int F, HW, heads;
    int sites_per_group, groups_per_sample, ngroups;
    float q_scale, eps;
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f2bf(a[8 * s + j]);
    return f;
}

constexpr int TW_NT = 2;        // 32-row groups per workgroup
constexpr int TW_HEADS = 8;     // one wave per head
constexpr int TW_PD = 4;        // k-steps of weight fragments in flight per wave

// MODE 0: attention over the frames of each site (temporal).  MODE 1: full softmax attention over the positions of one
// frame (the mid-block spatial attention, vdp.py:708-710: no rotary, no bias, no focus flags): a workgroup is one frame
// of at most 64 positions, its two 32-row tiles are the frame's key / query halves.  MODE 2: SpatialLinearAttention
// (vdp.py:454-479) over such a frame: q softmax over the head's 32 features, k softmax over the positions,
// context = k^T v (32 x 32, in registers), out = q context, to_out with bias.
template <int C, bool FOLD, int MODE = 0>
__global__ __launch_bounds__(TW_HEADS * 64) void tattn_wide_kernel(const TWArgs a) {
    constexpr int KS = C / 16, CT = C / 32, NFRAG = 3 * KS + 2 * CT, SLICE = NFRAG * 512;
    constexpr int XROW = C + 8, OROW = TW_HEADS * 32 + 8, NT = TW_NT, NTHR = TW_HEADS * 64;
    constexpr int PPR = C / 8, PPT = PPR / 8;      // 16-byte pieces per row / per loader thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* xs = reinterpret_cast<__bf16*>(smem);                        // [NT * 32][XROW] LayerNormed rows
    __bf16* os = xs + NT * 32 * XROW;                                    // [NT * 32][OROW] attention output
    float* biasl = reinterpret_cast<float*>(os + NT * 32 * OROW);
    __bf16* kxl = reinterpret_cast<__bf16*>(biasl);                      // FOLD: [head][32 key rows][16 features]
    constexpr float LOG2E = 1.4426950408889634f;

    const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int F = a.F, g = a.sites_per_group;

    if constexpr (MODE != 0) {
    } else if constexpr (FOLD) {
        for (int i = tid; i < TW_HEADS * 512; i += NTHR) {
            const int slot = i & 15, row = (i >> 4) & 31, hd = i >> 9;
            const bool valid = row < g * F;
            const int kf = row % F, site = valid ? row / F : g;
            float v = 0.f;
            if (slot < F) v = (valid && a.bias) ? a.bias[(hd * F + slot) * F + kf] * LOG2E : 0.f;
            else if (slot < F + g + 1) v = (slot - F == site) ? VDMI_MASK_BIG : 0.f;      // 2^50
            kxl[i] = f2bf(v);
        }
    } else {
        for (int i = tid; i < TW_HEADS * F * 32; i += NTHR) {
            const int row = i & 31, hq = i >> 5;
            biasl[i] = a.bias ? a.bias[hq * F + row % F] * LOG2E : 0.f;
        }
    }

    // ---- phase 0: rows -> LayerNorm -> LDS.  Thread t loads pieces (t & 7) + 8 i of row t >> 3.
    {
        const int rr = tid >> 3, tile = rr >> 5, rl = rr & 31;
        bool ok;
        long long srow;
        if constexpr (MODE != 0) {
            ok = rr < a.HW;
            srow = (long long)blockIdx.x * a.HW + rr;
        } else {
            const int grp = blockIdx.x * NT + tile;
            const int b = grp / a.groups_per_sample;
            const int site = rl / F, fr = rl - site * F;
            const int s = (grp - b * a.groups_per_sample) * g + site;
            ok = grp < a.ngroups && site < g && s < a.HW;
            srow = ((long long)b * F + fr) * a.HW + s;
        }
        const __bf16* src = a.x + (ok ? srow : 0) * C;
        bf16x8 pv[PPT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const bf16x8*>(src + ((tid & 7) + 8 * i) * 8);
            pv[i] = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1 += f; s2 += f * f; }
        }
        s1 = vdmi_group_sum<8>(s1);
        s2 = vdmi_group_sum<8>(s2);
        const float mean = s1 * (1.0f / C);
        const float rstd = rsqrtf(fmaxf(s2 * (1.0f / C) - mean * mean, 0.f) + a.eps);
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(((float)pv[i][e] - mean) * rstd);
            *reinterpret_cast<bf16x8*>(xs + rr * XROW + ((tid & 7) + 8 * i) * 8) = o;
        }
    }

    // per-lane constants: this lane's row as a query, per group
    const int q_site = l31 / F, q_f = l31 - q_site * F;
    const float qs = a.q_scale * LOG2E;
    bool rowvalid[NT];
    long long grow[NT];
    unsigned kmask[NT];
    {
        unsigned kmask_all = 0, kmask_self = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ks = acc_row(r, half) / F;
            if (ks == q_site && ks < g) kmask_all |= 1u << r;
            if (acc_row(r, half) == l31 && ks < g) kmask_self |= 1u << r;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (MODE != 0) {
                rowvalid[t] = t * 32 + l31 < a.HW;
                grow[t] = rowvalid[t] ? (long long)blockIdx.x * a.HW + t * 32 + l31 : 0;
                kmask[t] = 0;
            } else {
                const int grp = blockIdx.x * NT + t;
                const int b = grp / a.groups_per_sample;
                const int s = (grp - b * a.groups_per_sample) * g + q_site;
                rowvalid[t] = grp < a.ngroups && q_site < g && s < a.HW;
                grow[t] = rowvalid[t] ? ((long long)b * F + q_f) * a.HW + s : 0;
                kmask[t] = (a.focus && grp < a.ngroups && a.focus[b]) ? kmask_self : kmask_all;
            }
        }
    }
    bf16x8 qx = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (FOLD) {
        const int qs_idx = l31 < g * F ? q_site : g;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int slot = 8 * half + j;
            float v = 0.f;
            if (slot < F) v = (slot == q_f) ? 1.f : 0.f;
            else if (slot < F + g + 1) v = (slot - F == qs_idx) ? 0.f : -VDMI_MASK_BIG;
            qx[j] = f2bf(v);
        }
    }
    // packed rotary factors of register pairs (2i, 2i+1): plain for k, times the q scale for q
    f16x2 rcp[4], rsp[4], rcq[4], rsq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rg = i >> 1, e = 2 * (i & 1);
        float c0 = 1.f, c1 = 1.f, s0 = 0.f, s1 = 0.f;
        if (a.rot_cos) {
            const float* cp = a.rot_cos + q_f * 16 + 8 * rg + 4 * half + e;
            const float* sp = a.rot_sin + q_f * 16 + 8 * rg + 4 * half + e;
            c0 = cp[0]; c1 = cp[1]; s0 = sp[0]; s1 = sp[1];
        }
        rcp[i] = f16x2{(_Float16)c0, (_Float16)c1};
        rsp[i] = f16x2{(_Float16)s0, (_Float16)s1};
        rcq[i] = f16x2{(_Float16)(c0 * qs), (_Float16)(c1 * qs)};
        rsq[i] = f16x2{(_Float16)(s0 * qs), (_Float16)(s1 * qs)};
    }
    auto rotate16 = [](const f32x16& t, const f16x2 (&cc)[4], const f16x2 (&ss)[4], f16x8& f0, f16x8& f1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x2 lo = f16x2{(_Float16)t[2 * i], (_Float16)t[2 * i + 1]};
            const f16x2 hi = f16x2{(_Float16)t[2 * i + 8], (_Float16)t[2 * i + 9]};
            const f16x2 l2 = lo * cc[i] - hi * ss[i], h2 = hi * cc[i] + lo * ss[i];
            f0[2 * i] = l2[0]; f0[2 * i + 1] = l2[1];
            f1[2 * i] = h2[0]; f1[2 * i + 1] = h2[1];
        }
    };

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const __bf16* wh = a.w + (long long)head * SLICE + lane * 8;      // this head's fragments, this lane's 16 bytes

    // ---- phase 1: projections of this wave's head for both groups; weight fragments TW_PD k-steps ahead
    f32x16 qT[NT], kT[NT], vv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { qT[t] = zero; kT[t] = zero; vv[t] = zero; }
    bf16x8 wq[TW_PD], wk[TW_PD], wv[TW_PD];
#pragma unroll
    for (int u = 0; u < TW_PD; ++u) {
        wq[u] = *reinterpret_cast<const bf16x8*>(wh + (0 * KS + u) * 512);
        wk[u] = *reinterpret_cast<const bf16x8*>(wh + (1 * KS + u) * 512);
        wv[u] = *reinterpret_cast<const bf16x8*>(wh + (2 * KS + u) * 512);
    }
    __syncthreads();      // the LayerNormed rows (and the bias tables) are in LDS
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int u = ks % TW_PD;
        bf16x8 xb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            xb[t] = *reinterpret_cast<const bf16x8*>(xs + (t * 32 + l31) * XROW + ks * 16 + 8 * half);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            qT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[u], xb[t], qT[t], 0, 0, 0);
            // linear attention wants K like V: positions in the registers, the head's feature on the lane
            if constexpr (MODE == 2) kT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[t], wk[u], kT[t], 0, 0, 0);
            else kT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[u], xb[t], kT[t], 0, 0, 0);
            vv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[t], wv[u], vv[t], 0, 0, 0);
        }
        if (ks + TW_PD < KS) {
            wq[u] = *reinterpret_cast<const bf16x8*>(wh + (0 * KS + ks + TW_PD) * 512);
            wk[u] = *reinterpret_cast<const bf16x8*>(wh + (1 * KS + ks + TW_PD) * 512);
            wv[u] = *reinterpret_cast<const bf16x8*>(wh + (2 * KS + ks + TW_PD) * 512);
        }
    }

    // ---- phase 2: attention of this head; O^T -> LDS in the to_out fragments' k-slot order
    if constexpr (MODE == 2) {
        // k: softmax over the frame's positions, per feature d = this lane's column (rows past the frame weigh nothing)
        float mx = -1e30f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (t * 32 + acc_row(r, half) >= a.HW) kT[t][r] = -1e30f;
                mx = fmaxf(mx, kT[t][r]);
            }
        mx = xhalf_max(mx);
        const float mxl = mx * LOG2E;
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(kT[t][r], LOG2E, -mxl));
                kT[t][r] = e;
                lsum += e;
            }
        lsum = xhalf_add(lsum);
        const float kinv = 1.0f / lsum;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) kT[t][r] *= kinv;
        // context[d][e] = sum_n k[n][d] v[n][e]: d in the registers, e on the lanes (one accumulator per tile, so that
        // neighbouring MFMAs are independent)
        f32x16 ctx = zero, ctx1 = zero;
        ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kT[0], 0), acc_frag(vv[0], 0), ctx, 0, 0, 0);
        ctx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kT[1], 0), acc_frag(vv[1], 0), ctx1, 0, 0, 0);
        ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kT[0], 1), acc_frag(vv[0], 1), ctx, 0, 0, 0);
        ctx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kT[1], 1), acc_frag(vv[1], 1), ctx1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) ctx[r] += ctx1[r];
        const bf16x8 c0 = acc_frag(ctx, 0), c1 = acc_frag(ctx, 1);
        // position of feature e = l31 inside a row's 32 k-slots of this head (see phase 3's fragment order)
        const int ee = l31 & 15, epos = (l31 >> 4) * 16 + ((ee >> 2) & 1) * 8 + (ee & 3) + 4 * (ee >> 3);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // q: softmax over the head's 32 features (16 registers here, 16 in lane ^ 32), times the scale
            float qm = qT[t][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) qm = fmaxf(qm, qT[t][r]);
            qm = xhalf_max(qm);
            const float qml = qm * LOG2E;
            float qsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { qT[t][r] = __builtin_amdgcn_exp2f(fmaf(qT[t][r], LOG2E, -qml)); qsum += qT[t][r]; }
            qsum = xhalf_add(qsum);
            const float qinv = a.q_scale / qsum;
#pragma unroll
            for (int r = 0; r < 16; ++r) qT[t][r] *= qinv;
            // out[n][e] = sum_d q[n][d] context[d][e]: rows n in the registers, e on the lanes
            f32x16 o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(qT[t], 0), c0, zero, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(qT[t], 1), c1, o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) os[(t * 32 + acc_row(r, half)) * OROW + head * 32 + epos] = f2bf(o[r]);
        }
    } else if constexpr (MODE == 1) {
        // operands of the score product in f16, q times scale * log2 e
        const f16x2 qs2 = f16x2{(_Float16)qs, (_Float16)qs};
        f16x8 qf[NT][2], kf[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f16x2 q2 = f16x2{(_Float16)qT[t][8 * s + 2 * i], (_Float16)qT[t][8 * s + 2 * i + 1]} * qs2;
                    const f16x2 k2 = f16x2{(_Float16)kT[t][8 * s + 2 * i], (_Float16)kT[t][8 * s + 2 * i + 1]};
                    qf[t][s][2 * i] = q2[0]; qf[t][s][2 * i + 1] = q2[1];
                    kf[t][s][2 * i] = k2[0]; kf[t][s][2 * i + 1] = k2[1];
                }
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            f32x16 st[NT];
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][0], qf[qt][0], zero, 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][1], qf[qt][1], st[kt], 0, 0, 0);
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (kt * 32 + acc_row(r, half) >= a.HW) st[kt][r] = -1e30f;      // keys past the frame
                    mx = fmaxf(mx, st[kt][r]);
                }
            mx = xhalf_max(mx);
            float lsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(st[kt][r] - mx);
                    st[kt][r] = p;
                    lsum += p;
                }
            lsum = xhalf_add(lsum);
            f32x16 o = zero, o1 = zero;      // two accumulators: neighbouring MFMAs stay independent
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vv[0], 0), acc_frag(st[0], 0), o, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vv[1], 0), acc_frag(st[1], 0), o1, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vv[0], 1), acc_frag(st[0], 1), o, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vv[1], 1), acc_frag(st[1], 1), o1, 0, 0, 0);
            const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = (o[r] + o1[r]) * inv;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                *reinterpret_cast<bf16x8*>(os + (qt * 32 + l31) * OROW + head * 32 + s * 16 + 8 * half) = acc_frag(o, s);
        }
    } else
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f16x8 qf0, qf1, kf0, kf1;
        rotate16(qT[t], rcq, rsq, qf0, qf1);
        rotate16(kT[t], rcp, rsp, kf0, kf1);
        f32x16 st;
        float mx = -1e30f;
        if constexpr (FOLD) {
            const bf16x8 kx = *reinterpret_cast<const bf16x8*>(kxl + (head * 32 + l31) * 16 + 8 * half);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx, qx, zero, 0, 0, 0);          // bias + mask
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf0, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf1, st, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
        } else {
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf0, zero, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf1, st, 0, 0, 0);
            const float* bl = biasl + (head * F + q_f) * 32 + 4 * half;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + 8 * u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * u + e;
                    const float sv = st[r] + ((kmask[t] >> r) & 1u ? b4[e] : -1e30f);
                    st[r] = sv;
                    mx = fmaxf(mx, sv);
                }
            }
        }
        mx = xhalf_max(mx);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(st[r] - mx);
            st[r] = p;
            lsum += p;
        }
        lsum = xhalf_add(lsum);
        f32x16 o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vv[t], 0), acc_frag(st, 0), zero, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vv[t], 1), acc_frag(st, 1), o, 0, 0, 0);
        const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= inv;
        // registers 8s .. 8s+7 of lane (row l31, half) are the 8 values of k-step s of that row's fragment
#pragma unroll
        for (int s = 0; s < 2; ++s)
            *reinterpret_cast<bf16x8*>(os + (t * 32 + l31) * OROW + head * 32 + s * 16 + 8 * half) = acc_frag(o, s);
    }
    __syncthreads();

    // ---- phase 3: to_out over all heads for this wave's channel tiles, + residual, store
    for (int ct = head; ct < CT; ct += TW_HEADS) {
        f32x16 yacc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) yacc[t] = zero;
#pragma unroll
        for (int hs = 0; hs < TW_HEADS * 2; ++hs) {
            const int hd = hs >> 1, s = hs & 1;
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(a.w + (long long)hd * SLICE + (3 * KS + 2 * ct + s) * 512 + lane * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 ob = *reinterpret_cast<const bf16x8*>(os + (t * 32 + l31) * OROW + hd * 32 + s * 16 + 8 * half);
                yacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, ob, yacc[t], 0, 0, 0);
            }
        }
        // registers (4u+v) of tile ct hold channel 32ct + 16(u>>1) + 8 half + 4(u&1) + v
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (rowvalid[t]) {
#pragma unroll
                for (int kl = 0; kl < 2; ++kl) {
                    const long long off = grow[t] * C + (2 * ct + kl) * 16 + 8 * half;
                    const bf16x8 xr = *reinterpret_cast<const bf16x8*>(a.x + off);
                    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
                    if (MODE == 2 && a.b_out) {
                        b0 = *reinterpret_cast<const f32x4*>(a.b_out + (2 * ct + kl) * 16 + 8 * half);
                        b1 = *reinterpret_cast<const f32x4*>(a.b_out + (2 * ct + kl) * 16 + 8 * half + 4);
                    }
                    bf16x8 out;
#pragma unroll
                    for (int e = 0; e < 8; ++e) out[e] = f2bf(yacc[t][8 * kl + e] + (e < 4 ? b0[e & 3] : b1[e & 3]) + (float)xr[e]);
                    *reinterpret_cast<bf16x8*>(a.y + off) = out;
                }
            }
        }
    }
}


// SpatialLinearAttention over frames of MORE than 64 positions (the 16x16 level), still one launch and one workgroup per
// frame: two sweeps over the frame in pairs of 32-row tiles.  Sweep A projects k and v and accumulates the context with
// a running maximum per feature (online softmax over the positions); sweep B projects q, multiplies with the context
// and runs the output projection per pair.  The rows are read and LayerNormed twice and the weight fragments are
// streamed from L2 once per pair, which is still far less than the three launches it replaces.
template <int C>
__global__ __launch_bounds__(TW_HEADS * 64) void sla_wide_long_kernel(const TWArgs a) {
    constexpr int KS = C / 16, CT = C / 32, NFRAG = 3 * KS + 2 * CT, SLICE = NFRAG * 512;
    constexpr int XROW = C + 8, OROW = TW_HEADS * 32 + 8, NT = TW_NT;
    constexpr int PPT = C / 64;      // 16-byte pieces per loader thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* xs = reinterpret_cast<__bf16*>(smem);                        // [NT * 32][XROW]
    __bf16* os = xs + NT * 32 * XROW;                                    // [NT * 32][OROW]
    float* xch = reinterpret_cast<float*>(os + NT * 32 * OROW);          // [heads][32]: per-feature factors, lane -> register order
    constexpr float LOG2E = 1.4426950408889634f;
    const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const long long frow0 = (long long)blockIdx.x * a.HW;
    const int npairs = (a.HW + NT * 32 - 1) / (NT * 32);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const __bf16* wh = a.w + (long long)head * SLICE + lane * 8;

    // rows p0 .. p0 + 63 of the frame -> LayerNorm -> xs (thread t: pieces (t & 7) + 8 i of row t >> 3)
    auto load_rows = [&](int p0) {
        const int rr = tid >> 3;
        const bool ok = p0 + rr < a.HW;
        const __bf16* src = a.x + (frow0 + (ok ? p0 + rr : 0)) * C;
        bf16x8 pv[PPT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const bf16x8*>(src + ((tid & 7) + 8 * i) * 8);
            pv[i] = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1 += f; s2 += f * f; }
        }
        s1 = vdmi_group_sum<8>(s1);
        s2 = vdmi_group_sum<8>(s2);
        const float mean = s1 * (1.0f / C);
        const float rstd = rsqrtf(fmaxf(s2 * (1.0f / C) - mean * mean, 0.f) + a.eps);
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(((float)pv[i][e] - mean) * rstd);
            *reinterpret_cast<bf16x8*>(xs + rr * XROW + ((tid & 7) + 8 * i) * 8) = o;
        }
    };
    // a per-feature value held by lane d (both halves agree) -> the 16 values of this lane's context registers
    // (register r is feature acc_row(r, half)); through this wave's 32 floats of LDS, no barrier needed
    auto to_register_order = [&](float v, float (&out)[16]) {
        float* xc = xch + head * 32;
        if (half == 0) xc[l31] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(xc + 8 * u + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[4 * u + e] = q4[e];
        }
        __builtin_amdgcn_wave_barrier();
    };

    // ---- sweep A: context[d][e] = sum_n softmax_n(k)[n][d] v[n][e]
    f32x16 ctx = zero, ctx1 = zero;
    float m_run = -1e30f, l_run = 0.f;
    for (int pi = 0; pi < npairs; ++pi) {
        __syncthreads();
        load_rows(pi * NT * 32);
        bf16x8 wk[TW_PD], wv[TW_PD];
#pragma unroll
        for (int u = 0; u < TW_PD; ++u) {
            wk[u] = *reinterpret_cast<const bf16x8*>(wh + (1 * KS + u) * 512);
            wv[u] = *reinterpret_cast<const bf16x8*>(wh + (2 * KS + u) * 512);
        }
        __syncthreads();
        f32x16 kk[NT], vv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { kk[t] = zero; vv[t] = zero; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int u = ks % TW_PD;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xs + (t * 32 + l31) * XROW + ks * 16 + 8 * half);
                kk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, wk[u], kk[t], 0, 0, 0);
                vv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, wv[u], vv[t], 0, 0, 0);
            }
            if (ks + TW_PD < KS) {
                wk[u] = *reinterpret_cast<const bf16x8*>(wh + (1 * KS + ks + TW_PD) * 512);
                wv[u] = *reinterpret_cast<const bf16x8*>(wh + (2 * KS + ks + TW_PD) * 512);
            }
        }
        float tmax = -1e30f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (pi * NT * 32 + t * 32 + acc_row(r, half) >= a.HW) kk[t][r] = -1e30f;
                tmax = fmaxf(tmax, kk[t][r]);
            }
        tmax = xhalf_max(tmax);
        if (__any(tmax > m_run)) {      // some feature's maximum moved: rescale what has been accumulated
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            m_run = m_new;
            l_run *= alpha;
            float ar[16];
            to_register_order(alpha, ar);
#pragma unroll
            for (int r = 0; r < 16; ++r) { ctx[r] *= ar[r]; ctx1[r] *= ar[r]; }
        }
        const float mxl = m_run * LOG2E;
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(kk[t][r], LOG2E, -mxl));
                kk[t][r] = e;
                lsum += e;
            }
        l_run += xhalf_add(lsum);
        ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kk[0], 0), acc_frag(vv[0], 0), ctx, 0, 0, 0);
        ctx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kk[1], 0), acc_frag(vv[1], 0), ctx1, 0, 0, 0);
        ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kk[0], 1), acc_frag(vv[0], 1), ctx, 0, 0, 0);
        ctx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(kk[1], 1), acc_frag(vv[1], 1), ctx1, 0, 0, 0);
    }
    bf16x8 c0, c1;
    {
        float lr[16];
        to_register_order(1.0f / l_run, lr);
#pragma unroll
        for (int r = 0; r < 16; ++r) ctx[r] = (ctx[r] + ctx1[r]) * lr[r];
        c0 = acc_frag(ctx, 0);
        c1 = acc_frag(ctx, 1);
    }
    const int ee = l31 & 15, epos = (l31 >> 4) * 16 + ((ee >> 2) & 1) * 8 + (ee & 3) + 4 * (ee >> 3);

    // ---- sweep B: q, out = q context, to_out + bias + residual, pair by pair
    for (int pi = 0; pi < npairs; ++pi) {
        __syncthreads();      // the previous pair's readers of xs / os are done
        load_rows(pi * NT * 32);
        bf16x8 wq[TW_PD];
#pragma unroll
        for (int u = 0; u < TW_PD; ++u) wq[u] = *reinterpret_cast<const bf16x8*>(wh + (0 * KS + u) * 512);
        __syncthreads();
        f32x16 qT[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) qT[t] = zero;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int u = ks % TW_PD;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xs + (t * 32 + l31) * XROW + ks * 16 + 8 * half);
                qT[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[u], xb, qT[t], 0, 0, 0);
            }
            if (ks + TW_PD < KS) wq[u] = *reinterpret_cast<const bf16x8*>(wh + (0 * KS + ks + TW_PD) * 512);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float qm = qT[t][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) qm = fmaxf(qm, qT[t][r]);
            qm = xhalf_max(qm);
            const float qml = qm * LOG2E;
            float qsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { qT[t][r] = __builtin_amdgcn_exp2f(fmaf(qT[t][r], LOG2E, -qml)); qsum += qT[t][r]; }
            qsum = xhalf_add(qsum);
            const float qinv = a.q_scale / qsum;
#pragma unroll
            for (int r = 0; r < 16; ++r) qT[t][r] *= qinv;
            f32x16 o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(qT[t], 0), c0, zero, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(qT[t], 1), c1, o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) os[(t * 32 + acc_row(r, half)) * OROW + head * 32 + epos] = f2bf(o[r]);
        }
        __syncthreads();
        for (int ct = head; ct < CT; ct += TW_HEADS) {
            f32x16 yacc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) yacc[t] = zero;
#pragma unroll
            for (int hs = 0; hs < TW_HEADS * 2; ++hs) {
                const int hd = hs >> 1, s = hs & 1;
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(a.w + (long long)hd * SLICE + (3 * KS + 2 * ct + s) * 512 + lane * 8);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 ob = *reinterpret_cast<const bf16x8*>(os + (t * 32 + l31) * OROW + hd * 32 + s * 16 + 8 * half);
                    yacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, ob, yacc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int pos = pi * NT * 32 + t * 32 + l31;
                if (pos < a.HW) {
#pragma unroll
                    for (int kl = 0; kl < 2; ++kl) {
                        const int ch = (2 * ct + kl) * 16 + 8 * half;
                        const long long off = (frow0 + pos) * C + ch;
                        const bf16x8 xr = *reinterpret_cast<const bf16x8*>(a.x + off);
                        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
                        if (a.b_out) {
                            b0 = *reinterpret_cast<const f32x4*>(a.b_out + ch);
                            b1 = *reinterpret_cast<const f32x4*>(a.b_out + ch + 4);
                        }
                        bf16x8 out;
#pragma unroll
                        for (int e = 0; e < 8; ++e) out[e] = f2bf(yacc[t][8 * kl + e] + (e < 4 ? b0[e & 3] : b1[e & 3]) + (float)xr[e]);
                        *reinterpret_cast<bf16x8*>(a.y + off) = out;
                    }
                }
            }
        }
    }
}

}   // namespace

static int frame_attention_wide(int mode, const void* x, void* y, const void* wpack, const float* b_out, int frames, int hw,
                                int c, int heads, float q_scale, float eps, vdmi_stream stream_) {
    if (!x || !y || !wpack) return VDMI_E_ARG;
    if (frames <= 0 || hw <= 0 || (c != 256 && c != 512) || heads != TW_HEADS) return VDMI_E_SHAPE;
    if (hw > TW_NT * 32 && mode != 2) return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(y) || !aligned16(wpack)) return VDMI_E_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TWArgs a;
    memset(&a, 0, sizeof(a));
    a.x = static_cast<const __bf16*>(x);
    a.y = static_cast<__bf16*>(y);
    a.w = static_cast<const __bf16*>(wpack);
    a.b_out = b_out;
    a.F = 1;
    a.HW = hw;
    a.heads = heads;
    a.sites_per_group = 32;
    a.groups_per_sample = 1;
    a.ngroups = frames;
    a.q_scale = q_scale;
    a.eps = eps;
    const size_t lds = (size_t)TW_NT * 32 * (c + 8) * 2 + (size_t)TW_NT * 32 * (TW_HEADS * 32 + 8) * 2 + TW_HEADS * 32 * 4;
    if (hw > TW_NT * 32) {      // linear attention over a larger frame: the two-sweep kernel
#define VDMI_SL_CASE(C_)                                                                                          \
    if (c == C_) {                                                                                                \
        static std::atomic<uint64_t> attr_mask{0};                                                                \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(sla_wide_long_kernel<C_>), attr_mask, 160 * 1024))     \
            return VDMI_E_LAUNCH;                                                                                 \
        hipLaunchKernelGGL((sla_wide_long_kernel<C_>), dim3(frames), dim3(TW_HEADS * 64), lds, stream, a);        \
        VDMI_LAUNCH_CHECK();                                                                                      \
        return VDMI_OK;                                                                                           \
    }
        VDMI_SL_CASE(256)      // (at 512 channels this kernel needs more registers than a wave has; those layers keep the three launches)
#undef VDMI_SL_CASE
        return VDMI_E_SHAPE;
    }
#define VDMI_SW_CASE(C_, MODE_)                                                                                         \
    if (c == C_ && mode == MODE_) {                                                                                     \
        static std::atomic<uint64_t> attr_mask{0};                                                                      \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(tattn_wide_kernel<C_, false, MODE_>), attr_mask, 160 * 1024)) \
            return VDMI_E_LAUNCH;                                                                                       \
        hipLaunchKernelGGL((tattn_wide_kernel<C_, false, MODE_>), dim3(frames), dim3(TW_HEADS * 64), lds, stream, a);   \
        VDMI_LAUNCH_CHECK();                                                                                            \
        return VDMI_OK;                                                                                                 \
    }
    VDMI_SW_CASE(256, 1)
    VDMI_SW_CASE(512, 1)
    VDMI_SW_CASE(256, 2)
    VDMI_SW_CASE(512, 2)
#undef VDMI_SW_CASE
    return VDMI_E_SHAPE;
}

extern "C" int vdmi_sattn_fused_wide(const void* x, void* y, const void* wpack, int frames, int hw, int c, int heads,
                                     float q_scale, float eps, vdmi_stream stream) {
    return frame_attention_wide(1, x, y, wpack, nullptr, frames, hw, c, heads, q_scale, eps, stream);
}

extern "C" int vdmi_sla_fused_wide(const void* x, void* y, const void* wpack, const float* b_out, int frames, int hw, int c,
                                   int heads, float q_scale, float eps, vdmi_stream stream) {
    if (b_out && !aligned16(b_out)) return VDMI_E_ALIGN;
    return frame_attention_wide(2, x, y, wpack, b_out, frames, hw, c, heads, q_scale, eps, stream);
}

extern "C" int vdmi_tattn_fused_wide(const void* x, void* y, const void* wpack, const float* bias, const float* rot_cos,
                                     const float* rot_sin, int b, int f, int hw, int c, int heads, float q_scale,
                                     float eps, const uint8_t* focus, vdmi_stream stream_) {
    if (!x || !y || !wpack) return VDMI_E_ARG;
    if (b <= 0 || f <= 0 || f > 32 || hw <= 0 || (rot_cos && !rot_sin)) return VDMI_E_SHAPE;
    if ((c != 256 && c != 512) || heads != TW_HEADS) return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(y) || !aligned16(wpack)) return VDMI_E_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    TWArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.y = static_cast<__bf16*>(y);
    a.w = static_cast<const __bf16*>(wpack);
    a.bias = bias;
    a.rot_cos = rot_cos;
    a.rot_sin = rot_sin;
    a.focus = focus;
    a.b_out = nullptr;
    a.F = f;
    a.HW = hw;
    a.heads = heads;
    a.sites_per_group = 32 / f;
    a.groups_per_sample = (hw + a.sites_per_group - 1) / a.sites_per_group;
    a.ngroups = b * a.groups_per_sample;
    a.q_scale = q_scale;
    a.eps = eps;
    const bool fold = !focus && f + 32 / f + 1 <= 16;
    const size_t tables = fold ? (size_t)heads * 1024 : (size_t)heads * f * 32 * 4;
    const size_t lds = (size_t)TW_NT * 32 * (c + 8) * 2 + (size_t)TW_NT * 32 * (TW_HEADS * 32 + 8) * 2 + tables;
    if (lds > 160 * 1024) return VDMI_E_LDS;
    const int grid = (a.ngroups + TW_NT - 1) / TW_NT;
#define VDMI_TW_CASE(C_, FOLD_)                                                                                  \
    if (c == C_ && fold == FOLD_) {                                                                              \
        static std::atomic<uint64_t> attr_mask{0};                                                               \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(tattn_wide_kernel<C_, FOLD_>), attr_mask, 160 * 1024)) \
            return VDMI_E_LAUNCH;                                                                                \
        hipLaunchKernelGGL((tattn_wide_kernel<C_, FOLD_>), dim3(grid), dim3(TW_HEADS * 64), lds, stream, a);     \
        VDMI_LAUNCH_CHECK();                                                                                     \
        return VDMI_OK;                                                                                          \
    }
    VDMI_TW_CASE(256, true)
    VDMI_TW_CASE(256, false)
    VDMI_TW_CASE(512, true)
    VDMI_TW_CASE(512, false)
#undef VDMI_TW_CASE
    return VDMI_E_SHAPE;
}
