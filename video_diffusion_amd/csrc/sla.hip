// Fused SpatialLinearAttention for gfx950: Residual(PreNorm(SpatialLinearAttention)) of the
// reference (vdp.py:454-479 with 331-337, 380-408) in two passes over x plus a small fold, with
// every activation tile living in MFMA registers.
//
//  pass 1  per (frame, part): channel LayerNorm -> K, V projections -> softmax of K over the
//          positions (online max / sum) -> context^T += V^T E.  One wave per head: the head's Wk, Wv
//          fragments stay in that wave's registers; the LayerNormed x tile is staged once per
//          32-row chunk in LDS and shared by the 8 waves.  K and V are produced "rows in registers,
//          channel on the lane", so the per-channel max/sum over positions is in-lane and the
//          accumulators are directly the operands of the context product (contraction over rows).
//  fold    per (frame, head): combine the parts, normalise, fold the 32x32 context into to_out:
//          W'[c][h*32+d] = sum_e Wout[c][h*32+e] * ctx[d][e], written in pass 2's fragment order.
//  pass 2  per 32-row chunk (one wave): LayerNorm -> Q projection -> softmax over the head's 32
//          features (in-lane) -> y = W' q + bias + x.  Wq and the frame's W' live in LDS.
#include "vdmi_common.h"
#include <cstdlib>
#include <cstring>

namespace {

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f2bf(a[8 * s + j]);
    return f;
}

struct SlaArgs {
    const __bf16* x;
    __bf16* y;
    const __bf16* wkv;      // [heads][2][KS][64][8]
    const __bf16* wq;       // [heads][KS][64][8]
    const float* w_out;     // [C][heads*32]
    const float* b_out;     // [C]
    float* part_m;          // [frames][nparts][heads][32]
    float* part_l;
    float* part_ctx;        // [frames][nparts][heads][32 e][32 d]
    __bf16* wf;             // [frames][heads][CT][2][64][8]
    int frames, hw, heads, nparts, chunks, chunks_per_part;
    float q_scale, eps;
};

// ------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512, 2) void sla_pass1_kernel(const SlaArgs a) {
    constexpr int KS = C / 16, PPR = C / 8, XROW = C + 8;
    constexpr int SUB = 4;                                    // 32-row chunks staged per barrier
    constexpr int PIECES = SUB * 32 * PPR, LP = (PIECES + 511) / 512;   // 16-byte pieces per stage / per thread
    __shared__ __attribute__((aligned(16))) __bf16 xs[2][SUB * 32 * XROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int frame = blockIdx.x / a.nparts, part = blockIdx.x - frame * a.nparts;
    const int c0 = part * a.chunks_per_part;
    const int c1 = min(c0 + a.chunks_per_part, a.chunks);
    const int head = wave;
    const bool active = head < a.heads;

    // Wk rows carry a factor log2 e (host packing), so exp2 of the projected k is exp of the reference's k
    bf16x8 wk[KS], wv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (active) {
            wk[ks] = *reinterpret_cast<const bf16x8*>(a.wkv + (((long long)head * 2 + 0) * KS + ks) * 512 + lane * 8);
            wv[ks] = *reinterpret_cast<const bf16x8*>(a.wkv + (((long long)head * 2 + 1) * KS + ks) * 512 + lane * 8);
        } else {
            wk[ks] = wv[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    // loader role: piece p = tid + k * 512 of a stage is 8 channels of row p / PPR
    const long long fbase = (long long)frame * a.hw;
    bf16x8 pre[LP];
    auto load_stage = [&](int c) {
#pragma unroll
        for (int k = 0; k < LP; ++k) {
            const int p = tid + k * 512, n = c * 32 + p / PPR;
            pre[k] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (p < PIECES && n < a.hw && n < c1 * 32)
                pre[k] = *reinterpret_cast<const bf16x8*>(a.x + (fbase + n) * C + (p % PPR) * 8);
        }
    };
    auto norm_store = [&](__bf16* dst) {      // channel LayerNorm of each row (its PPR pieces sit in adjacent lanes)
#pragma unroll
        for (int k = 0; k < LP; ++k) {
            const int p = tid + k * 512;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)pre[k][e]; s1 += f; s2 += f * f; }
            s1 = vdmi_group_sum<PPR>(s1);
            s2 = vdmi_group_sum<PPR>(s2);
            const float mean = s1 * (1.0f / C);
            const float rstd = rsqrtf(fmaxf(s2 * (1.0f / C) - mean * mean, 0.f) + a.eps);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(((float)pre[k][e] - mean) * rstd);
            if (p < PIECES) *reinterpret_cast<bf16x8*>(dst + (p / PPR) * XROW + (p % PPR) * 8) = o;
        }
    };

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 ctx = zero;
    float m_run = -1e30f, l_run = 0.f;      // running max (log2 domain) and sum of this lane's feature

    load_stage(c0);
    int buf = 0;
    for (int cs = c0; cs < c1; cs += SUB, buf ^= 1) {
        __bf16* xb = xs[buf];
        norm_store(xb);
        __syncthreads();
        if (cs + SUB < c1) load_stage(cs + SUB);
        const int nsub = min(SUB, c1 - cs);
        for (int u = 0; u < nsub; ++u) {
            const int c = cs + u;
            const __bf16* xr = xb + u * 32 * XROW;
            f32x16 kacc = zero, vacc = zero;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 xa = *reinterpret_cast<const bf16x8*>(xr + l31 * XROW + ks * 16 + 8 * half);
                kacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, wk[ks], kacc, 0, 0, 0);
                vacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, wv[ks], vacc, 0, 0, 0);
            }
            // rows (positions) are in the registers, the head's feature d is this lane's column
            if (c * 32 + 32 > a.hw) {     // ragged last chunk: rows past the frame never win the max, weigh nothing
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (c * 32 + acc_row(r, half) >= a.hw) kacc[r] = -1e30f;
            }
            float tmax = kacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, kacc[r]);
            tmax = xhalf_max(tmax);
            if (__any(tmax > m_run)) {    // rescale only when some feature's maximum moved (rare after the first chunks)
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) ctx[r] *= alpha;
            }
            float lsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(kacc[r] - m_run);
                kacc[r] = e;
                lsum += e;
            }
            lsum = xhalf_add(lsum);
            l_run += lsum;
            // ctx^T[e][d] += sum_n V[n][e] * E[n][d]
            ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vacc, 0), acc_frag(kacc, 0), ctx, 0, 0, 0);
            ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag(vacc, 1), acc_frag(kacc, 1), ctx, 0, 0, 0);
        }
    }

    if (active) {
        const long long pbase = ((long long)frame * a.nparts + part) * a.heads + head;
        if (half == 0) {
            a.part_m[pbase * 32 + l31] = m_run;
            a.part_l[pbase * 32 + l31] = l_run;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a.part_ctx[(pbase * 32 + acc_row(r, half)) * 32 + l31] = ctx[r];
    }
}

// k-slot -> accumulator-row order of v_mfma_f32_32x32x16_bf16 (k-step s, element j, lane half h)
__host__ __device__ inline int slot_row(int s, int j, int h) { return (j & 3) + 16 * s + 8 * (j >> 2) + 4 * h; }
// accumulator row a of channel tile ct -> output channel (8 consecutive channels per register pair)
__host__ __device__ inline int out_chan(int ct, int arow) {
    const int u = arow >> 3, hh = (arow >> 2) & 1, vv = arow & 3;
    return 32 * ct + 16 * (u >> 1) + 8 * hh + 4 * (u & 1) + vv;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sla_fold_kernel(const SlaArgs a, int c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ctxn = reinterpret_cast<float*>(smem);      // [32 d][33]
    float* wsl = ctxn + 32 * 33;                       // [c][33]: this head's 32 columns of to_out
    const int tid = threadIdx.x;
    const int frame = blockIdx.x / a.heads, head = blockIdx.x - frame * a.heads;
    const int CT = c / 32, hid = a.heads * 32;
    for (int idx = tid; idx < c * 32; idx += 256) {
        const int co = idx >> 5, e = idx & 31;
        wsl[co * 33 + e] = a.w_out[(long long)co * hid + head * 32 + e];
    }
    // combine the parts of this (frame, head), per feature d = tid & 31 (maxima are in the log2 domain)
    {
        const int d = tid & 31;
        float m = -1e30f;
        for (int p = 0; p < a.nparts; ++p)
            m = fmaxf(m, a.part_m[(((long long)frame * a.nparts + p) * a.heads + head) * 32 + d]);
        float l = 0.f, sacc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < a.nparts; ++p) {
            const long long pb = ((long long)frame * a.nparts + p) * a.heads + head;
            const float w = __builtin_amdgcn_exp2f(a.part_m[pb * 32 + d] - m);
            l += w * a.part_l[pb * 32 + d];
#pragma unroll
            for (int q = 0; q < 4; ++q) sacc[q] += w * a.part_ctx[(pb * 32 + (tid >> 5) + 8 * q) * 32 + d];
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int q = 0; q < 4; ++q) ctxn[d * 33 + (tid >> 5) + 8 * q] = sacc[q] * inv;
    }
    __syncthreads();
    __bf16* wf = a.wf + (((long long)frame * a.heads + head) * CT) * 2 * 512;
    // element (ct, s, lane, j) = W'[out_chan(ct, lane&31)][head*32 + slot_row(s, j, lane>>5)].
    // On the matrix cores: the context is the A operand (rows d), to_out the B operand (column l31 carries
    // channel out_chan(ct, l31)); accumulator registers 8s..8s+7 of lane (l31, half) are then exactly the 8
    // values j of fragment s.  Both factors are split into bf16 high and low parts (three products), which
    // keeps the fold within ~2^-16 of fp32 before its single rounding to bf16.
    const int lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    bf16x8 ch[2], cl[2];
#pragma unroll
    for (int sk = 0; sk < 2; ++sk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = ctxn[l31 * 33 + 16 * sk + 8 * hf + j];
            const __bf16 hi = f2bf(v);
            ch[sk][j] = hi;
            cl[sk][j] = f2bf(v - (float)hi);
        }
    for (int ct = wave; ct < CT; ct += 4) {
        const float* wr = wsl + out_chan(ct, l31) * 33;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc = zero, acc1 = zero, acc2 = zero;
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
            bf16x8 wh, wl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = wr[16 * sk + 8 * hf + j];
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
#pragma unroll
        for (int sf = 0; sf < 2; ++sf) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(acc[8 * sf + j]);
            *reinterpret_cast<bf16x8*>(wf + ((ct * 2 + sf) * 64 + lane) * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512, 2) void sla_pass2_kernel(const SlaArgs a) {
    constexpr int KS = C / 16, CT = C / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* wq = reinterpret_cast<__bf16*>(smem);                 // [heads][KS][512]
    __bf16* wf = wq + a.heads * KS * 512;                         // [heads][CT][2][512]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int frame = blockIdx.x / a.nparts, part = blockIdx.x - frame * a.nparts;
    const int c0 = part * a.chunks_per_part;
    const int c1 = min(c0 + a.chunks_per_part, a.chunks);

    {
        const int nq = a.heads * KS * 512 / 8, nf = a.heads * CT * 2 * 512 / 8;
        const __bf16* wfg = a.wf + (long long)frame * a.heads * CT * 2 * 512;
        for (int i = tid; i < nq; i += 512)
            *reinterpret_cast<bf16x8*>(wq + i * 8) = *reinterpret_cast<const bf16x8*>(a.wq + i * 8);
        for (int i = tid; i < nf; i += 512)
            *reinterpret_cast<bf16x8*>(wf + i * 8) = *reinterpret_cast<const bf16x8*>(wfg + i * 8);
    }
    __syncthreads();

    for (int c = c0 + wave; c < c1; c += 8) {
        const int n = c * 32 + l31;
        const bool rowvalid = n < a.hw;
        const long long grow = (long long)frame * a.hw + n;
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

        for (int head = 0; head < a.heads; ++head) {
            f32x16 q = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(wq + (head * KS + ks) * 512 + lane * 8);
                q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xh[ks], q, 0, 0, 0);
            }
            // softmax over the head's 32 features: 16 registers here + 16 in lane ^ 32
            float mx = q[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, q[r]);
            mx = xhalf_max(mx);
            float sum = 0.f;      // Wq rows carry a factor log2 e (host packing)
#pragma unroll
            for (int r = 0; r < 16; ++r) { q[r] = __builtin_amdgcn_exp2f(q[r] - mx); sum += q[r]; }
            sum = xhalf_add(sum);
            const float inv = a.q_scale / sum;
#pragma unroll
            for (int r = 0; r < 16; ++r) q[r] *= inv;
            const bf16x8 q0 = acc_frag(q, 0), q1 = acc_frag(q, 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wf + ((head * CT + ct) * 2 + 0) * 512 + lane * 8);
                const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wf + ((head * CT + ct) * 2 + 1) * 512 + lane * 8);
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, q0, oacc[ct], 0, 0, 0);
                oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, q1, oacc[ct], 0, 0, 0);
            }
        }
#ifdef VDMI_SLA_ABL_NOSTORE     // timing-only ablation (wrong results): neither the residual reads nor the stores
        if (rowvalid && a.eps < -1.f) {
#else
        if (rowvalid) {
#endif
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int kl = 0; kl < 2; ++kl) {
                    const int ch = (2 * ct + kl) * 16 + 8 * half;
                    const long long off = grow * C + ch;
                    const bf16x8 xr = *reinterpret_cast<const bf16x8*>(a.x + off);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b_out + ch);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.b_out + ch + 4);
                    bf16x8 out;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        out[e] = f2bf(oacc[ct][8 * kl + e] + (e < 4 ? b0[e] : b1[e - 4]) + (float)xr[e]);
                    *reinterpret_cast<bf16x8*>(a.y + off) = out;
                }
        }
    }
}

inline uint16_t host_bf16(float f) { return vdmi_host_half_bits(f); }

// The split of a frame's positions into parts depends on hw only (never on the batch), so a
// sample's result is bit-identical however the batch is sharded.
void sla_geometry(int frames, int hw, int* nparts, int* chunks, int* cpp) {
    (void)frames;
    *chunks = (hw + 31) / 32;
    static const int cap = [] { const char* e = std::getenv("VDMI_SLA_PARTS"); const int v = e ? std::atoi(e) : 4; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
    int np = *chunks / 16;
    if (np > cap) np = cap;
    if (np < 1) np = 1;
    *cpp = (*chunks + np - 1) / np;
    *nparts = (*chunks + *cpp - 1) / *cpp;
}

}   // namespace

extern "C" size_t vdmi_pack_sla_elems(int c, int heads) { return (size_t)heads * 3 * (c / 16) * 512; }

// HOST.  wqkv: float [3*heads*32][c] (LayerNorm gamma folded in).  out: k,v fragments
// [heads][2][c/16][64][8] followed by q fragments [heads][c/16][64][8].
// The q and k rows, which only ever feed a softmax, are scaled by log2 e so the kernels exponentiate with
// exp2 alone.
extern "C" int vdmi_pack_sla(const float* wqkv, int c, int heads, uint16_t* out) {
    if (!wqkv || !out) return VDMI_E_ARG;
    if (c <= 0 || c % 32 || heads <= 0) return VDMI_E_SHAPE;
    const int KS = c / 16, hid = heads * 32;
    const float log2e = 1.4426950408889634f;
    size_t o = 0;
    for (int h = 0; h < heads; ++h)
        for (int t = 1; t <= 2; ++t)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        out[o++] = host_bf16((t == 1 ? log2e : 1.0f) *
                                             wqkv[(size_t)(t * hid + h * 32 + (lane & 31)) * c + ks * 16 + 8 * (lane >> 5) + j]);
    for (int h = 0; h < heads; ++h)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    out[o++] = host_bf16(log2e * wqkv[(size_t)(h * 32 + (lane & 31)) * c + ks * 16 + 8 * (lane >> 5) + j]);
    return VDMI_OK;
}

extern "C" size_t vdmi_sla_workspace_bytes(int frames, int hw, int c, int heads) {
    int nparts, chunks, cpp;
    sla_geometry(frames, hw, &nparts, &chunks, &cpp);
    const size_t parts = (size_t)frames * nparts * heads;
    size_t bytes = parts * 32 * 4 * 2 + parts * 32 * 32 * 4;
    bytes = (bytes + 255) / 256 * 256;
    bytes += (size_t)frames * heads * (c / 32) * 2 * 512 * 2;
    return bytes;
}

extern "C" int vdmi_sla_fused(const void* x, void* y, const void* wpack, const float* w_out, const float* b_out,
                              void* workspace, int frames, int hw, int c, int heads, float q_scale, float eps,
                              vdmi_stream stream) {
    if (!x || !y || !wpack || !w_out || !b_out || !workspace) return VDMI_E_ARG;
    if (frames <= 0 || hw <= 0 || heads <= 0 || heads > 8 || (c != 32 && c != 64 && c != 128)) return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(y) || !aligned16(wpack) || !aligned16(workspace) || !aligned16(b_out))
        return VDMI_E_ALIGN;
    SlaArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.y = static_cast<__bf16*>(y);
    a.wkv = static_cast<const __bf16*>(wpack);
    a.wq = a.wkv + (size_t)heads * 2 * (c / 16) * 512;
    a.w_out = w_out;
    a.b_out = b_out;
    a.frames = frames;
    a.hw = hw;
    a.heads = heads;
    sla_geometry(frames, hw, &a.nparts, &a.chunks, &a.chunks_per_part);
    const size_t parts = (size_t)frames * a.nparts * heads;
    char* ws = static_cast<char*>(workspace);
    a.part_m = reinterpret_cast<float*>(ws);
    a.part_l = a.part_m + parts * 32;
    a.part_ctx = a.part_l + parts * 32;
    size_t off = parts * 32 * 4 * 2 + parts * 32 * 32 * 4;
    off = (off + 255) / 256 * 256;
    a.wf = reinterpret_cast<__bf16*>(ws + off);
    a.q_scale = q_scale;
    a.eps = eps;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(frames * a.nparts));
    const size_t lds2 = ((size_t)heads * (c / 16) + (size_t)heads * (c / 32) * 2) * 1024;
    if (lds2 > 160 * 1024) return VDMI_E_LDS;
#define VDMI_SLA_CASE(C_)                                                                                     \
    if (c == C_) {                                                                                            \
        static std::atomic<uint64_t> attr_mask{0};                                                            \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(sla_pass2_kernel<C_>), attr_mask, 160 * 1024))     \
            return VDMI_E_LAUNCH;                                                                             \
        hipLaunchKernelGGL(sla_pass1_kernel<C_>, grid, dim3(512), 0, st, a);                                  \
        VDMI_LAUNCH_CHECK();                                                                                  \
        hipLaunchKernelGGL(sla_fold_kernel, dim3((unsigned)(frames * heads)), dim3(256),                      \
                           (size_t)(32 + C_) * 33 * 4, st, a, c);                                             \
        VDMI_LAUNCH_CHECK();                                                                                  \
        hipLaunchKernelGGL(sla_pass2_kernel<C_>, grid, dim3(512), lds2, st, a);                               \
        VDMI_LAUNCH_CHECK();                                                                                  \
        return VDMI_OK;                                                                                       \
    }
    VDMI_SLA_CASE(32)
    VDMI_SLA_CASE(64)
    VDMI_SLA_CASE(128)
#undef VDMI_SLA_CASE
    return VDMI_E_SHAPE;
}
