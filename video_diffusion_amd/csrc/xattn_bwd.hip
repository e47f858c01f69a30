// Backward of the cross-attention core (reference video_diffusion_pytorch/video_diffusion_pytorch.py:511-521: sim = q k^T * scale,
// softmax over the context tokens, out = attn v; torch autograd differentiates it at :1451) on the matrix cores.
//
// Conventions of the forward (xattn.hip): q rows are pre-multiplied by scale * log2 e, so P = exp2(q . k - lse) with
// the stored log2-domain log-sum-exp; the core's output (and therefore its gradient dO) has the features of a head in
// the stored order: column 16*hh + r holds feature (r&3) + 8*(r>>2) + 4*hh.  With delta = rowsum(dO * O):
//     dP = dO V^T      dS = P * (dP - delta)      dq = ln 2 * dS K      dk = ln 2 * dS^T q      dv = P^T dO
// Two kernels, each recomputing P in the orientation whose accumulator feeds its products without a transposition:
//   * dq: a wave owns 32 queries (lane = query) and walks heads and key tiles exactly like the forward; the score and
//     dP accumulators [key x query] are the B operands of dq^T += K^T dS^T.  Also writes delta for the second kernel.
//   * dk / dv: a wave owns 32 keys of one (sample, head) (lane = key) and walks a chunk of that sample's query tiles;
//     the accumulators [query x key] are the B operands of dv^T += dO^T P and dk^T += q^T dS, whose A operands (the
//     query tile transposed) come from LDS through the transposing read ds_read_b64_tr_b16.  Chunks write partial
//     sums; a third launch adds them in a fixed order (no atomics).
#include "vdmi_common.h"

namespace {

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f2bf(a[8 * s + j]);
    return f;
}

constexpr float LN2 = 0.6931471805599453f;

struct XBwdArgs {
    const __bf16* q;          // [b*n][ldq], scaled by scale * log2 e
    const __bf16* o;          // [b*n][heads*32], stored feature order
    const __bf16* dout;       // same layout: gradient of o
    __bf16* dq;               // [b*n][heads*32], natural feature order
    const __bf16* kv;         // training pack [b][4][heads][nt][2][64][8]
    const __bf16* kvrows;     // [b*l][ldkv]: K columns [0, hid), V columns [hid, 2 hid)
    const float* lse;         // [b][heads][stride]
    float* delta;             // [b][heads][stride]
    float* part;              // dk / dv partial sums [nsplit][b][heads][nt*32][64]
    long long n, stride;
    int ldq, ldkv, heads, l, nt, groups_per_sample, blocks_per_sample;
    int nsplit, tiles_per_split;
};

constexpr int XB_WAVES = 8;
constexpr int XB_CH = 4;                                  // key tiles staged per step: 4 * 6 KB, double buffered
constexpr int XB_PP = XB_CH * 384 / (XB_WAVES * 64);      // 16-byte pieces per thread per step (3)

__global__ __launch_bounds__(XB_WAVES * 64) void xattn_bwd_dq_kernel(const XBwdArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 kvbuf[2][XB_CH * 3072];     // per tile: K, V (stored order), K^T: 2 fragments each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / a.blocks_per_sample;
    const int grp = (blockIdx.x - b * a.blocks_per_sample) * XB_WAVES + wave;
    const long long row = (long long)grp * 32 + l31;
    const bool valid = grp < a.groups_per_sample && row < a.n;
    const long long grow = (long long)b * a.n + (valid ? row : 0);
    const int hid = a.heads * 32;
    const int tail = a.l - (a.nt - 1) * 32;
    unsigned tailmask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (acc_row(r, half) < tail) tailmask |= 1u << r;

    const int nchunks = (a.nt + XB_CH - 1) / XB_CH, nsteps = a.heads * nchunks;
    bf16x8 pre[XB_PP];
    auto prefetch = [&](int step) {
        const int head = step / nchunks, t0 = (step - head * nchunks) * XB_CH;
        const int cnt = min(XB_CH, a.nt - t0);
#pragma unroll
        for (int p = 0; p < XB_PP; ++p) {
            const int idx = tid + p * (XB_WAVES * 64);
            const int tt = idx / 384, within = idx - tt * 384;
            const int sel = within >> 7, kind = sel == 0 ? 0 : (sel == 1 ? 2 : 3);
            if (tt < cnt)
                pre[p] = *reinterpret_cast<const bf16x8*>(a.kv + (((long long)(b * 4 + kind) * a.heads + head) * a.nt + t0 + tt) * 1024 +
                                                          (within & 127) * 8);
        }
    };
    auto commit = [&](int step) {
        const int head = step / nchunks, t0 = (step - head * nchunks) * XB_CH;
        const int cnt = min(XB_CH, a.nt - t0);
        __bf16* dst = kvbuf[step & 1];
#pragma unroll
        for (int p = 0; p < XB_PP; ++p) {
            const int idx = tid + p * (XB_WAVES * 64);
            if (idx / 384 < cnt) *reinterpret_cast<bf16x8*>(dst + idx * 8) = pre[p];
        }
    };

    prefetch(0);
    int step = 0;
    for (int head = 0; head < a.heads; ++head) {
        bf16x8 q0 = {0, 0, 0, 0, 0, 0, 0, 0}, q1 = q0, g0 = q0, g1 = q0;
        float dl = 0.f, ls = 0.f;
        if (valid) {
            const __bf16* qp = a.q + grow * a.ldq + head * 32 + 8 * half;
            q0 = *reinterpret_cast<const bf16x8*>(qp);
            q1 = *reinterpret_cast<const bf16x8*>(qp + 16);
            const __bf16* gp = a.dout + grow * hid + head * 32 + 8 * half;
            const __bf16* op = a.o + grow * hid + head * 32 + 8 * half;
            g0 = *reinterpret_cast<const bf16x8*>(gp);
            g1 = *reinterpret_cast<const bf16x8*>(gp + 16);
            const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(op), o1 = *reinterpret_cast<const bf16x8*>(op + 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)g0[j] * (float)o0[j] + (float)g1[j] * (float)o1[j];
            ls = a.lse[((long long)b * a.heads + head) * a.stride + row];
        }
        dl = xhalf_add(dl);
        if (valid && half == 0) a.delta[((long long)b * a.heads + head) * a.stride + row] = dl;
        f32x16 negl, negd, dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) { negl[r] = -ls; negd[r] = -dl; dq[r] = 0.f; }
        for (int c = 0; c < nchunks; ++c, ++step) {
            commit(step);
            __syncthreads();
            if (step + 1 < nsteps) prefetch(step + 1);
            const int t0 = c * XB_CH, cnt = min(XB_CH, a.nt - t0);
            const __bf16* buf = kvbuf[step & 1] + lane * 8;
            for (int tt = 0; tt < cnt; ++tt) {
                const __bf16* tb = buf + tt * 3072;
                const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(tb), k1 = *reinterpret_cast<const bf16x8*>(tb + 512);
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(tb + 1024), v1 = *reinterpret_cast<const bf16x8*>(tb + 1536);
                const bf16x8 t0f = *reinterpret_cast<const bf16x8*>(tb + 2048), t1f = *reinterpret_cast<const bf16x8*>(tb + 2560);
                f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, negl, 0, 0, 0);
                f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, g0, negd, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, g1, dp, 0, 0, 0);
                const bool last = (t0 + tt == a.nt - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(st[r]);
                    if (last && !((tailmask >> r) & 1u)) p = 0.f;          // padding keys
                    st[r] = p * dp[r];                                     // dS^T = P^T * (dP^T - delta)
                }
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t0f, acc_frag(st, 0), dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t1f, acc_frag(st, 1), dq, 0, 0, 0);
            }
        }
        if (valid) {
            __bf16* dp_ = a.dq + grow * hid + head * 32 + 4 * half;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = f2bf(dq[4 * rg + e] * LN2);
                *reinterpret_cast<bf16x4*>(dp_ + 8 * rg) = o4;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// dk / dv.  Workgroup = 8 waves = 8 key tiles (256 keys) of one (sample, head); blockIdx.z = chunk of the sample's
// query tiles.  Per query tile the workgroup stages the 32 x 32 q and dO tiles in LDS with the rows permuted
// (LDS row p holds query p with bits 2 and 3 swapped: the order in which an accumulator's registers become a B
// operand), so the transposing read hands every wave q^T and dO^T as A operands whose contraction index matches.
// ------------------------------------------------------------------------------------------
typedef short s4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_fragment(const char* base) {
    typedef __attribute__((address_space(3))) s4v* lds_s4;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base + 4 * 64));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(XB_WAVES * 64) void xattn_bwd_dkv_kernel(const XBwdArgs a) {
    __shared__ __attribute__((aligned(16))) char tiles[2][2][32 * 64];        // [buffer][q | dO][row p][64 B]
    __shared__ __attribute__((aligned(16))) float stats[2][2][32];              // [buffer][lse | delta][query]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int head = blockIdx.y % a.heads, b = blockIdx.y / a.heads;
    const int kt = blockIdx.x * XB_WAVES + wave;                // this wave's key tile
    const int key = kt * 32 + l31;
    const bool key_ok = kt < a.nt && key < a.l;
    const int hid = a.heads * 32;
    const int qt_begin = blockIdx.z * a.tiles_per_split;
    const int qt_end = min(qt_begin + a.tiles_per_split, a.groups_per_sample);

    // B operands that never change: K^T [feature x key] and V^T [stored column x key] of this wave's keys
    bf16x8 kb[2], vb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        kb[s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        vb[s] = kb[s];
        if (key_ok) {
            const __bf16* kr = a.kvrows + ((long long)b * a.l + key) * a.ldkv + head * 32;
            kb[s] = *reinterpret_cast<const bf16x8*>(kr + s * 16 + 8 * half);
            const __bf16* vr = kr + hid + 16 * half + 4 * s;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr), hi = *reinterpret_cast<const bf16x4*>(vr + 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { vb[s][j] = lo[j]; vb[s][4 + j] = hi[j]; }
        }
    }
    const float keymask = key_ok ? 1.f : 0.f;

    // staging: threads 0..127 one 16-byte piece of the q tile each, 128..255 of the dO tile, 256..287 lse, 288..319 delta
    const int sp_row = (tid & 127) >> 2, sp_c = tid & 3;                       // query within the tile, 16-byte piece
    const int sp_p = (sp_row & 0x13) | ((sp_row & 4) << 1) | ((sp_row & 8) >> 1);   // LDS row: bits 2 and 3 swapped
    bf16x8 pre = {0, 0, 0, 0, 0, 0, 0, 0};
    float pre_s = 0.f;
    auto prefetch = [&](int qt) {
        const long long r0 = (long long)qt * 32;
        if (tid < 256) {
            const long long r = r0 + sp_row;
            pre = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (r < a.n) {
                const long long g = (long long)b * a.n + r;
                pre = tid < 128 ? *reinterpret_cast<const bf16x8*>(a.q + g * a.ldq + head * 32 + sp_c * 8)
                                : *reinterpret_cast<const bf16x8*>(a.dout + g * hid + head * 32 + sp_c * 8);
            }
        } else if (tid < 320) {
            const long long r = r0 + (tid & 31);
            const float* src = tid < 288 ? a.lse : a.delta;
            pre_s = r < a.n ? src[((long long)b * a.heads + head) * a.stride + r] : 0.f;
        }
    };
    auto commit = [&](int buf) {
        if (tid < 256) *reinterpret_cast<bf16x8*>(&tiles[buf][tid >> 7][sp_p * 64 + sp_c * 16]) = pre;
        else if (tid < 320) stats[buf][(tid - 256) >> 5][tid & 31] = pre_s;
    };
    // transposing-read addresses: lane 4q + p of 16-lane group g supplies row 8*(g>>1) + q, columns 16*(g&1) + 4p .. +3
    const int g16 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_off = (8 * (g16 >> 1) + tq) * 64 + (16 * (g16 & 1) + 4 * tp) * 2;

    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.f;
    if (qt_begin < qt_end) {
        prefetch(qt_begin);
        int cur = 0;
        for (int qt = qt_begin; qt < qt_end; ++qt, cur ^= 1) {
            commit(cur);
            __syncthreads();        // the tile is visible; every wave is done with the buffer the next commit overwrites
            if (qt + 1 < qt_end) prefetch(qt + 1);
            // A operands with the queries as rows: straight from memory (the rows were just read for the staging)
            const long long r = (long long)qt * 32 + l31;
            bf16x8 qa0 = {0, 0, 0, 0, 0, 0, 0, 0}, qa1 = qa0, ga0 = qa0, ga1 = qa0;
            if (r < a.n) {
                const long long g = (long long)b * a.n + r;
                const __bf16* qp = a.q + g * a.ldq + head * 32 + 8 * half;
                const __bf16* gp = a.dout + g * hid + head * 32 + 8 * half;
                qa0 = *reinterpret_cast<const bf16x8*>(qp);
                qa1 = *reinterpret_cast<const bf16x8*>(qp + 16);
                ga0 = *reinterpret_cast<const bf16x8*>(gp);
                ga1 = *reinterpret_cast<const bf16x8*>(gp + 16);
            }
            f32x16 negl, negd;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&stats[cur][0][8 * rg + 4 * half]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&stats[cur][1][8 * rg + 4 * half]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { negl[4 * rg + e] = -l4[e]; negd[4 * rg + e] = -d4[e]; }
            }
            f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa0, kb[0], negl, 0, 0, 0);
            f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga0, vb[0], negd, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa1, kb[1], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga1, vb[1], dp, 0, 0, 0);
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const float p = __builtin_amdgcn_exp2f(st[r2]) * keymask;
                st[r2] = p;
                dp[r2] = p * dp[r2];
            }
            const char* tq_ = &tiles[cur][0][tr_off];
            const char* tg_ = &tiles[cur][1][tr_off];
            const bf16x8 gt0 = tr_fragment(tg_), gt1 = tr_fragment(tg_ + 16 * 64);
            const bf16x8 qt0 = tr_fragment(tq_), qt1 = tr_fragment(tq_ + 16 * 64);
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt0, acc_frag(st, 0), dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt0, acc_frag(dp, 0), dk, 0, 0, 0);
            dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt1, acc_frag(st, 1), dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt1, acc_frag(dp, 1), dk, 0, 0, 0);
        }
    }
    if (kt < a.nt) {
        // lane = key; dk registers are natural features acc_row(r, half); dv registers are stored columns acc_row(r, half),
        // i.e. natural feature (r&3) + 16*((r>>2)&1) + 8*half + 4*(r>>3)
        float* dst = a.part + ((((long long)blockIdx.z * gridDim.y + blockIdx.y) * a.nt * 32) + key) * 64;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 k4, v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { k4[e] = dk[4 * rg + e]; v4[e] = dv[4 * rg + e]; }
            *reinterpret_cast<f32x4*>(dst + 8 * rg + 4 * half) = k4;
            *reinterpret_cast<f32x4*>(dst + 32 + 16 * (rg & 1) + 8 * half + 4 * (rg >> 1)) = v4;
        }
    }
}

// dkv rows bf16 [b*l][2*hid] = sum over the splits (fixed order); the key gradient takes the ln 2 of the log2-domain scores
__global__ __launch_bounds__(256) void xattn_bwd_reduce_kernel(const float* __restrict__ part, __bf16* __restrict__ dkv,
                                                               int nsplit, int b, int heads, int l, int nt, int ld,
                                                               long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per 4 floats
    if (i >= total) return;
    const int c4 = (int)(i & 15);
    long long r = i >> 4;
    const int key = (int)(r % l); r /= l;
    const int head = (int)(r % heads);
    const int bi = (int)(r / heads);
    const long long split_stride = (long long)b * heads * nt * 32 * 64;
    const float* src = part + (((long long)bi * heads + head) * nt * 32 + key) * 64 + c4 * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < nsplit; ++sp) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + sp * split_stride);
        s += v;
    }
    const bool is_v = c4 >= 8;
    const float f = is_v ? 1.f : LN2;
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(s[e] * f);
    const int hid = heads * 32;
    *reinterpret_cast<bf16x4*>(dkv + ((long long)bi * l + key) * ld + (is_v ? hid : 0) + head * 32 + (c4 & 7) * 4) = o;
}

int xbwd_splits(int b, int heads, int nt, long long groups) {
    const long long base = (long long)b * heads * ((nt + XB_WAVES - 1) / XB_WAVES);
    long long ns = (1024 + base - 1) / base;                  // aim at ~1024 workgroups
    if (ns > groups) ns = groups;
    if (ns > 256) ns = 256;
    if (ns < 1) ns = 1;
    return (int)ns;
}

}   // namespace

extern "C" size_t vdmi_xattn_bwd_workspace_bytes(int b, int64_t rows_per_sample, int l, int heads) {
    if (b <= 0 || rows_per_sample <= 0 || l <= 0 || heads <= 0) return 0;
    const int nt = (l + 31) / 32;
    const long long groups = (rows_per_sample + 31) / 32;
    return (size_t)xbwd_splits(b, heads, nt, groups) * b * heads * nt * 32 * 64 * sizeof(float);
}

// q, o, dout, lse as vdmi_xattn_core_train_fwd took / wrote them; kvrows: the projected context rows the pack was made
// from.  Writes dq bf16 [b*rows_per_sample][heads*32] (gradient of the scaled q rows), dkv bf16 [b*l][lddkv] (key
// columns, then value columns), and delta float [b][heads][lse_stride] (scratch the caller provides, like `workspace`).
extern "C" int vdmi_xattn_core_train_bwd(const void* q, int ldq, const void* o, const void* dout, const void* kvpack_train,
                                         const void* kvrows, int ldkv, const float* lse, float* delta, int64_t lse_stride,
                                         void* dq, void* dkv, int lddkv, int b, int64_t rows_per_sample, int l, int heads,
                                         void* workspace, vdmi_stream stream_) {
    if (!q || !o || !dout || !kvpack_train || !kvrows || !lse || !delta || !dq || !dkv || !workspace) return VDMI_E_ARG;
    const int hid = heads * 32;
    if (b <= 0 || rows_per_sample <= 0 || l <= 0 || heads <= 0 || ldq < hid || ldq % 8 || ldkv < 2 * hid || ldkv % 8 ||
        lddkv < 2 * hid || lddkv % 4 || lse_stride < rows_per_sample || b * heads > 65535)
        return VDMI_E_SHAPE;
    if (!aligned16(q) || !aligned16(o) || !aligned16(dout) || !aligned16(kvpack_train) || !aligned16(kvrows) || !aligned16(dq) ||
        !aligned16(dkv) || !aligned16(workspace))
        return VDMI_E_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    XBwdArgs a;
    a.q = static_cast<const __bf16*>(q);
    a.o = static_cast<const __bf16*>(o);
    a.dout = static_cast<const __bf16*>(dout);
    a.dq = static_cast<__bf16*>(dq);
    a.kv = static_cast<const __bf16*>(kvpack_train);
    a.kvrows = static_cast<const __bf16*>(kvrows);
    a.lse = lse;
    a.delta = delta;
    a.part = static_cast<float*>(workspace);
    a.n = rows_per_sample;
    a.stride = lse_stride;
    a.ldq = ldq;
    a.ldkv = ldkv;
    a.heads = heads;
    a.l = l;
    a.nt = (l + 31) / 32;
    const long long gps = (rows_per_sample + 31) / 32;
    const long long bps = (gps + XB_WAVES - 1) / XB_WAVES;
    if (gps > 0x7fffffffLL || bps * b > 0x7fffffffLL) return VDMI_E_SHAPE;
    a.groups_per_sample = (int)gps;
    a.blocks_per_sample = (int)bps;
    a.nsplit = xbwd_splits(b, heads, a.nt, gps);
    a.tiles_per_split = (int)((gps + a.nsplit - 1) / a.nsplit);
    hipLaunchKernelGGL(xattn_bwd_dq_kernel, dim3((unsigned)(bps * b)), dim3(XB_WAVES * 64), 0, stream, a);
    VDMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(xattn_bwd_dkv_kernel, dim3((unsigned)((a.nt + XB_WAVES - 1) / XB_WAVES), (unsigned)(b * heads), (unsigned)a.nsplit),
                       dim3(XB_WAVES * 64), 0, stream, a);
    VDMI_LAUNCH_CHECK();
    const long long total = (long long)b * heads * l * 16;
    hipLaunchKernelGGL(xattn_bwd_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a.part,
                       static_cast<__bf16*>(dkv), a.nsplit, b, heads, l, a.nt, lddkv, total);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
