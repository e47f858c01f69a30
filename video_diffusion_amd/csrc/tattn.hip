// Fused temporal attention for gfx950: Residual(PreNorm(Attention over frames)) in ONE kernel,
// activations never leave registers.
//
// Unit of work: a "group" of 32 rows = floor(32 / F) spatial sites x F frames (+ padding rows).
// One wave owns one group.  With v_mfma_f32_32x32x16_bf16 a 32x32 accumulator holds its column
// on the lane and its 16 rows in registers; a product that contracts over the accumulator's ROW
// index can take the accumulator registers directly as its 8-element operand fragments
// (registers 8s..8s+7 for k-step s), in whatever row order they are in, provided the other
// operand uses the same order.  The chain below is arranged so that every product contracts over
// the previous accumulator's row index:
//
//   x (rows on lanes, channels in fragments)  -- LayerNorm in registers -->  xh
//   Q^T, K^T [d x row]   = Wq/Wk (A) * xh (B)          rows on lanes, d in registers
//   V   [row x d]        = xh (A) * Wv (B)             d on lanes, rows in registers
//   S^T [key x query]    = K^T regs (A) * Q^T regs (B) contraction over d
//   P^T = softmax over keys (registers + one cross-half shuffle), block-diagonal site mask, bias
//   O^T [d x query]      = V regs (A) * P^T regs (B)   contraction over keys
//   Y^T [c x query]     += Wout_h (A, k permuted to the register order) * O^T regs (B)
//
// Output channels are permuted (through Wout's row order) so that a lane ends up with 8
// consecutive channels per register pair: 16-byte stores, and the residual x is re-read with the
// same 16-byte pattern.  Weights are the only LDS traffic.  When all heads' slices fit in LDS at once
// (RES: C <= 64 with 8 heads) a persistent workgroup loads them once and its waves then run with no
// barrier at all; otherwise one head's slice is double buffered with a barrier per head.
// Scores are kept in the log2 domain (q scale and bias table pre-multiplied by log2 e) so the
// softmax is one subtract and one v_exp_f32 per element.
#include "vdmi_common.h"
#include <cstring>

namespace {

struct TAttnArgs {
    const __bf16* x;
    __bf16* y;
    const __bf16* w;        // [heads][3*KS + 2*CT][64][8]
    const float* bias;      // [heads][F][F] or null
    const float* rot_cos;   // [F][16] or null
    const float* rot_sin;
    const uint8_t* focus;   // [B] or null: samples whose frames attend only to themselves (vdp.py:590-600)
    int F, HW, heads;
    int sites_per_group;    // floor(32 / F)
    int groups_per_sample;  // ceil(HW / sites_per_group): groups never straddle samples, so a sample's
                            // result does not depend on what else is in the batch
    int ngroups;            // B * groups_per_sample
    float q_scale, eps;
};

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// Rotary and the score product in half precision (VDMI_TATTN_ROT16, default on): the projected q and k go to packed
// f16 (11 significant bits against the 8 of the bf16 they were rounded to before), the rotation runs on pairs
// (v_pk_mul_f16 / v_pk_fma_f16: 16 instructions per tensor instead of 32) and its results ARE the operands of
// v_mfma_f32_32x32x16_f16 - no second conversion.  48 vector instructions per head instead of 80.
#ifndef VDMI_TATTN_ROT16
#define VDMI_TATTN_ROT16 1
#endif
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f2bf(a[8 * s + j]);
    return f;
}

// FOLD: the relative-position bias and the block-diagonal site mask are a 16-feature extension of the score
// product (one more MFMA into the same accumulator) instead of 32 vector instructions per head:
//   key side   [bias[h][0..F-1][frame(key)] * log2 e,  2^50 * onehot(site(key)) over g+1 sites (the last = padding rows)]
//   query side [onehot(frame(query)),                  -2^50 * (1 - onehot(site(query)))]
// so a key of another site (or a padding row) gets -2^100 and a key of the query's own site gets exactly its bias
// (in bf16).  Needs F + floor(32/F) + 1 <= 16 and no focus-present flags.
template <int C, bool RES, int NWAVES, bool FOLD>
__global__ __launch_bounds__(NWAVES * 64) void tattn_fused_kernel(const TAttnArgs a) {
    constexpr int KS = C / 16, CT = C / 32, NFRAG = 3 * KS + 2 * CT, SLICE = NFRAG * 512;
    constexpr int NTHR = NWAVES * 64;
    constexpr int WP = (SLICE / 8 + NTHR - 1) / NTHR;    // 16-byte pieces per thread per slice
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* wbuf = reinterpret_cast<__bf16*>(smem);      // RES: [heads][SLICE], else [2][SLICE]
    float* biasl = reinterpret_cast<float*>(smem + (RES ? a.heads : 2) * SLICE * 2);
    constexpr float LOG2E = 1.4426950408889634f;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int F = a.F, g = a.sites_per_group;

    __bf16* kxl = reinterpret_cast<__bf16*>(biasl);      // FOLD: [head][32 key rows][16 features]
    if constexpr (FOLD) {
        for (int i = tid; i < a.heads * 512; i += NTHR) {
            const int slot = i & 15, row = (i >> 4) & 31, head = i >> 9;
            const bool valid = row < g * F;
            const int kf = row % F, site = valid ? row / F : g;
            float v = 0.f;
            if (slot < F) v = (valid && a.bias) ? a.bias[(head * F + slot) * F + kf] * LOG2E : 0.f;
            else if (slot < F + g + 1) v = (slot - F == site) ? VDMI_MASK_BIG : 0.f;      // 2^50
            kxl[i] = f2bf(v);
        }
    } else {
        // bias table expanded to the 32 key rows of a group: [head][query frame][key row], times log2 e
        for (int i = tid; i < a.heads * F * 32; i += NTHR) {
            const int row = i & 31, hq = i >> 5;
            biasl[i] = a.bias ? a.bias[hq * F + row % F] * LOG2E : 0.f;
        }
    }
    if constexpr (RES) {
        const int pieces = a.heads * SLICE / 8;
        for (int i = tid; i < pieces; i += NTHR)
            *reinterpret_cast<bf16x8*>(wbuf + i * 8) = *reinterpret_cast<const bf16x8*>(a.w + i * 8);
        __syncthreads();      // the only barrier of the kernel
    }
    const float qs = a.q_scale * LOG2E;

    // per-lane constants: this lane's row as a query / row of x, and the 16 key rows of its registers
    const int q_site = l31 / F, q_f = l31 - q_site * F;
    unsigned kmask_all = 0, kmask_self = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ks = acc_row(r, half) / F;
        if (ks == q_site && ks < g) kmask_all |= 1u << r;
        if (acc_row(r, half) == l31 && ks < g) kmask_self |= 1u << r;
    }
    // FOLD: this lane's query-side features (k-slots 8*half .. 8*half+7 of the extra step)
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
    f32x4 rc[2], rs[2];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
        if (a.rot_cos) {
            rc[rg] = *reinterpret_cast<const f32x4*>(a.rot_cos + q_f * 16 + 8 * rg + 4 * half);
            rs[rg] = *reinterpret_cast<const f32x4*>(a.rot_sin + q_f * 16 + 8 * rg + 4 * half);
        } else {
            rc[rg] = f32x4{1.f, 1.f, 1.f, 1.f};
            rs[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // packed rotary factors of register pairs (2i, 2i+1), i = 0..3: plain for k, times the q scale for q
    [[maybe_unused]] f16x2 rcp[4], rsp[4], rcq[4], rsq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rg = i >> 1, e = 2 * (i & 1);
        rcp[i] = f16x2{(_Float16)rc[rg][e], (_Float16)rc[rg][e + 1]};
        rsp[i] = f16x2{(_Float16)rs[rg][e], (_Float16)rs[rg][e + 1]};
        rcq[i] = f16x2{(_Float16)(rc[rg][e] * qs), (_Float16)(rc[rg][e + 1] * qs)};
        rsq[i] = f16x2{(_Float16)(rs[rg][e] * qs), (_Float16)(rs[rg][e + 1] * qs)};
    }
    // registers 0..7 (k-step 0) and 8..15 (k-step 1) of a projected tensor, rotated: the two f16 operand fragments
    [[maybe_unused]] auto rotate16 = [](const f32x16& t, const f16x2 (&cc)[4], const f16x2 (&ss)[4], f16x8& f0, f16x8& f1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f16x2 lo = f16x2{(_Float16)t[2 * i], (_Float16)t[2 * i + 1]};
            const f16x2 hi = f16x2{(_Float16)t[2 * i + 8], (_Float16)t[2 * i + 9]};
            const f16x2 l2 = lo * cc[i] - hi * ss[i], h2 = hi * cc[i] + lo * ss[i];
            f0[2 * i] = l2[0]; f0[2 * i + 1] = l2[1];
            f1[2 * i] = h2[0]; f1[2 * i + 1] = h2[1];
        }
    };

    bf16x8 wreg[WP];
    auto load_slice = [&](int head) {
        const __bf16* p = a.w + (long long)head * SLICE;
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            const int idx = tid + k * NTHR;
            if (idx * 8 < SLICE) wreg[k] = *reinterpret_cast<const bf16x8*>(p + idx * 8);
        }
    };
    auto store_slice = [&](__bf16* dst) {
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            const int idx = tid + k * NTHR;
            if (idx * 8 < SLICE) *reinterpret_cast<bf16x8*>(dst + idx * 8) = wreg[k];
        }
    };

    const int rounds = (a.ngroups + NWAVES - 1) / NWAVES;
    int step = 0;   // counts head iterations over the whole kernel (buffer parity)
    if constexpr (!RES) load_slice(0);
    for (int round = blockIdx.x; round < rounds; round += gridDim.x) {
        const int grp = round * NWAVES + wave;
        const int b = grp / a.groups_per_sample;
        const int s = (grp - b * a.groups_per_sample) * g + q_site;      // site within the sample
        const bool rowvalid = grp < a.ngroups && q_site < g && s < a.HW;
        const unsigned kmask = (a.focus && grp < a.ngroups && a.focus[b]) ? kmask_self : kmask_all;
        const long long grow = rowvalid ? ((long long)b * F + q_f) * a.HW + s : 0;
        // ---- load this lane's row (8*KS values) and LayerNorm it in registers -------------------
        bf16x8 xh[KS];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (rowvalid) v = *reinterpret_cast<const bf16x8*>(a.x + grow * C + ks * 16 + 8 * half);
            xh[ks] = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1 += f; s2 += f * f; }
        }
        s1 = xhalf_add(s1);
        s2 = xhalf_add(s2);
        const float mean = s1 * (1.0f / C);
        const float rstd = rsqrtf(fmaxf(s2 * (1.0f / C) - mean * mean, 0.f) + a.eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) xh[ks][e] = f2bf(((float)xh[ks][e] - mean) * rstd);

        f32x16 oacc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[ct][r] = 0.f;

        for (int head = 0; head < a.heads; ++head, ++step) {
            const __bf16* wb;
            if constexpr (RES) {
                wb = wbuf + head * SLICE;
            } else {
                __bf16* wd = wbuf + (step & 1) * SLICE;
                store_slice(wd);
                __syncthreads();
                // prefetch the next slice (next head, or head 0 of this block's next round)
                const int nh = head + 1 < a.heads ? head + 1 : 0;
                if (head + 1 < a.heads || round + (int)gridDim.x < rounds) load_slice(nh);
                wb = wd;
            }
            const __bf16* wfrag = wb + lane * 8;

            // ---- projections ----------------------------------------------------------------------
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 q = zero, k = zero, v = zero;
            // long contractions: v is not needed until the scores are done, so the scheduler sinks its products below
            // the rotary of q and k, where they would run as one chain of dependent MFMAs; two accumulators (even / odd
            // k-steps) keep neighbouring MFMAs independent wherever they end up
            [[maybe_unused]] f32x16 v1 = zero;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wq = *reinterpret_cast<const bf16x8*>(wfrag + (0 * KS + ks) * 512);
                const bf16x8 wk = *reinterpret_cast<const bf16x8*>(wfrag + (1 * KS + ks) * 512);
                const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wfrag + (2 * KS + ks) * 512);
                q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq, xh[ks], q, 0, 0, 0);
                k = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk, xh[ks], k, 0, 0, 0);
                if (C >= 256 && (ks & 1)) v1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[ks], wv, v1, 0, 0, 0);
                else v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[ks], wv, v, 0, 0, 0);
            }
            if constexpr (C >= 256) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += v1[r];
            }
            // q scale (times log2 e) + half-split rotary (feature d pairs with d+16 = register r with r+8)
#if VDMI_TATTN_ROT16
            f16x8 qf0, qf1, kf0, kf1;
            rotate16(q, rcq, rsq, qf0, qf1);
            rotate16(k, rcp, rsp, kf0, kf1);
#else
#pragma unroll
            for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rg + e;
                    const float cq = rc[rg][e] * qs, sq = rs[rg][e] * qs;
                    const float q1 = q[r], q2 = q[r + 8];
                    q[r] = q1 * cq - q2 * sq;
                    q[r + 8] = q2 * cq + q1 * sq;
                    const float k1 = k[r], k2 = k[r + 8];
                    k[r] = k1 * rc[rg][e] - k2 * rs[rg][e];
                    k[r + 8] = k2 * rc[rg][e] + k1 * rs[rg][e];
                }
#endif
            // ---- S^T = K Q^T, softmax over keys (this lane's 16 registers + the other half) --------
            f32x16 st;
            float mx = -1e30f;
            if constexpr (FOLD) {
                const bf16x8 kx = *reinterpret_cast<const bf16x8*>(kxl + (head * 32 + l31) * 16 + 8 * half);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx, qx, zero, 0, 0, 0);          // bias + mask
#if VDMI_TATTN_ROT16
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf0, st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf1, st, 0, 0, 0);
#else
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(k, 0), acc_frag(q, 0), st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(k, 1), acc_frag(q, 1), st, 0, 0, 0);
#endif
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
            } else {
#if VDMI_TATTN_ROT16
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qf0, zero, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qf1, st, 0, 0, 0);
#else
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(k, 0), acc_frag(q, 0), zero, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(k, 1), acc_frag(q, 1), st, 0, 0, 0);
#endif
                const float* bl = biasl + (head * F + q_f) * 32 + 4 * half;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + 8 * u);     // key rows of registers 4u..4u+3
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * u + e;
                        // keys of other sites (and padding rows) get -1e30: exp2 below turns them into exact zeros
                        const float sv = st[r] + ((kmask >> r) & 1u ? b4[e] : -1e30f);
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
            // ---- O^T = V^T P^T ---------------------------------------------------------------------
            f32x16 o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(v, 0), acc_frag(st, 0), zero, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(v, 1), acc_frag(st, 1), o, 0, 0, 0);
            const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= inv;
            // ---- output projection, accumulated over heads ------------------------------------------
            const bf16x8 o0 = acc_frag(o, 0), o1 = acc_frag(o, 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wfrag + (3 * KS + 2 * ct) * 512);
                const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wfrag + (3 * KS + 2 * ct + 1) * 512);
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, o0, oacc[ct], 0, 0, 0);
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, o1, oacc[ct], 0, 0, 0);
            }
        }

        // ---- residual + store: registers (4u+v) of tile ct hold channel 32ct + 16(u>>1) + 8 half + 4(u&1) + v
#ifdef VDMI_TA_ABL_NOSTORE      // timing-only ablation (wrong results): neither the residual reads nor the stores
        if (rowvalid && a.eps < -1.f) {
#else
        if (rowvalid) {
#endif
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int kl = 0; kl < 2; ++kl) {
                    const long long off = grow * C + (2 * ct + kl) * 16 + 8 * half;
                    const bf16x8 xr = *reinterpret_cast<const bf16x8*>(a.x + off);
                    bf16x8 out;
#pragma unroll
                    for (int e = 0; e < 8; ++e) out[e] = f2bf(oacc[ct][8 * kl + e] + (float)xr[e]);
                    *reinterpret_cast<bf16x8*>(a.y + off) = out;
                }
        }
    }
}

}   // namespace

// k-slot -> accumulator-row order of v_mfma_f32_32x32x16_bf16 (k-step s, element j, lane half h)
static inline int slot_row(int s, int j, int h) { return (j & 3) + 16 * s + 8 * (j >> 2) + 4 * h; }

static inline uint16_t host_bf16(float f) { return vdmi_host_half_bits(f); }

extern "C" size_t vdmi_pack_tattn_elems(int c, int heads) {
    return (size_t)heads * (3 * (c / 16) + 2 * (c / 32)) * 512;
}

// HOST.  wqkv: float [3*heads*32][c] (LayerNorm gamma already folded in), wout: float [c][heads*32].
extern "C" int vdmi_pack_tattn(const float* wqkv, const float* wout, int c, int heads, uint16_t* out) {
    if (!wqkv || !wout || !out) return VDMI_E_ARG;
    if (c % 32 || c <= 0 || heads <= 0) return VDMI_E_SHAPE;
    const int KS = c / 16, CT = c / 32, hid = heads * 32;
    size_t o = 0;
    for (int h = 0; h < heads; ++h) {
        for (int t = 0; t < 3; ++t)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        out[o++] = host_bf16(wqkv[(size_t)(t * hid + h * 32 + (lane & 31)) * c + ks * 16 + 8 * (lane >> 5) + j]);
        for (int ct = 0; ct < CT; ++ct)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane) {
                    const int arow = lane & 31, u = arow >> 3, hh = (arow >> 2) & 1, vv = arow & 3;
                    const int chan = 32 * ct + 16 * (u >> 1) + 8 * hh + 4 * (u & 1) + vv;
                    for (int j = 0; j < 8; ++j)
                        out[o++] = host_bf16(wout[(size_t)chan * hid + h * 32 + slot_row(s, j, lane >> 5)]);
                }
    }
    return VDMI_OK;
}

extern "C" int vdmi_tattn_fused(const void* x, void* y, const void* wpack, const float* bias, const float* rot_cos,
                                const float* rot_sin, int b, int f, int hw, int c, int heads, float q_scale,
                                float eps, const uint8_t* focus, vdmi_stream stream) {
    if (!x || !y || !wpack) return VDMI_E_ARG;
    if (b <= 0 || f <= 0 || f > 32 || hw <= 0 || heads <= 0 || (rot_cos && !rot_sin)) return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(y) || !aligned16(wpack)) return VDMI_E_ALIGN;
    TAttnArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.y = static_cast<__bf16*>(y);
    a.w = static_cast<const __bf16*>(wpack);
    a.bias = bias;
    a.rot_cos = rot_cos;
    a.rot_sin = rot_sin;
    a.focus = focus;
    a.F = f;
    a.HW = hw;
    a.heads = heads;
    a.sites_per_group = 32 / f;
    a.groups_per_sample = (hw + a.sites_per_group - 1) / a.sites_per_group;
    a.ngroups = b * a.groups_per_sample;
    a.q_scale = q_scale;
    a.eps = eps;
    const int nfrag = 3 * (c / 16) + 2 * (c / 32);
    const bool fold = !focus && f + 32 / f + 1 <= 16;
    const size_t bias_bytes = fold ? (size_t)heads * 1024 : (size_t)heads * f * 32 * 4;
    // all heads resident (and 12 waves per workgroup, 3 per SIMD) when they fit beside the bias table
    const size_t lds_res = (size_t)heads * nfrag * 1024 + bias_bytes, lds_dbuf = (size_t)2 * nfrag * 1024 + bias_bytes;
    const bool res = c <= 64 && lds_res <= 150 * 1024;
    const size_t lds = res ? lds_res : lds_dbuf;
    if (lds > 160 * 1024) return VDMI_E_LDS;
    const int nwaves = res ? 12 : (c >= 256 ? 4 : 8);
    const int rounds = (a.ngroups + nwaves - 1) / nwaves;
    const int grid = rounds < 256 ? rounds : 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define VDMI_TATTN_CASE_(C_, RES_, NW_, FOLD_)                                                                \
    if (c == C_ && res == RES_ && fold == FOLD_) {                                                            \
        static std::atomic<uint64_t> attr_mask{0};                                                            \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(tattn_fused_kernel<C_, RES_, NW_, FOLD_>), attr_mask, \
                             160 * 1024))                                                                     \
            return VDMI_E_LAUNCH;                                                                             \
        hipLaunchKernelGGL((tattn_fused_kernel<C_, RES_, NW_, FOLD_>), dim3(grid), dim3(NW_ * 64), lds, st, a); \
        VDMI_LAUNCH_CHECK();                                                                                  \
        return VDMI_OK;                                                                                       \
    }
#define VDMI_TATTN_CASE(C_, RES_, NW_) VDMI_TATTN_CASE_(C_, RES_, NW_, true) VDMI_TATTN_CASE_(C_, RES_, NW_, false)
    VDMI_TATTN_CASE(32, true, 12)
    VDMI_TATTN_CASE(64, true, 12)
    VDMI_TATTN_CASE(32, false, 8)
    VDMI_TATTN_CASE(64, false, 8)
    VDMI_TATTN_CASE(128, false, 8)
    VDMI_TATTN_CASE(256, false, 4)      // one wave per SIMD: the row and the output accumulators alone are 192 registers
#undef VDMI_TATTN_CASE
#undef VDMI_TATTN_CASE_
    return VDMI_E_SHAPE;
}
