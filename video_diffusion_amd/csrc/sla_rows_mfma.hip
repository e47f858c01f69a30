// Matrix-core form of the row kernels of the linear attention's training path (reference
// video_diffusion_pytorch/video_diffusion_pytorch.py:454-479: q softmax over the 32 features of a head times scale,
// out = Qt ctx; the backward is what torch autograd derives for it at :1451; the sums over a frame's positions - ctx,
// dctx, S - come from vdmi_sla_train_reduce as one 1088-float record per (frame, head)).
//
// The vector kernels give a lane one (row, head) and run its three 32 x 32 matrix-vector products out of LDS broadcast
// reads, whose return path bounds them.  Here a wave takes tiles of 32 rows of ONE head:
//   * rows arrive as natural fragments (lane = row, 8 + 8 features), which are B operands of every product that
//     contracts over features:  out^T = ctx^T Qt^T,  dQt^T = ctx g^T,  dkt^T = dctx v^T,  dv^T = dctx^T kt^T;
//   * the 32 x 32 context of the (frame, head) is the A operand, read from the wave's copy of the record in LDS and
//     split into a high and a low 16-bit part (two MFMAs per k-step), so it enters with ~16 significant bits;
//   * results come out with lane = row, registers = 16 features, where the per-feature factors (softmax weight of k,
//     S, Qt) are applied in float; they leave through LDS as 16-byte pieces, four lanes on a row's 64 bytes;
//   * a wave walks a contiguous run of tiles, so the record is copied again only when the frame changes.  No barriers.
#include "vdmi_common.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

constexpr int SR_WAVES = 8;
constexpr int SR_REC = 32 * 32 + 64;      // floats of a combined record: [32][32], then two vectors of 32

struct SrArgs {
    const __bf16* qkv;        // [frames*n][3*heads*32]
    const __bf16* qt;         // bwd: saved Qt rows [rows][heads*32]
    const __bf16* dout;       // bwd
    const float* rec_ctx;     // [frames][heads][SR_REC]: ctx[d][e], max[d], 1 / weight sum[d]
    const float* rec_dctx;    // bwd: dctx[d][e], S[d]
    __bf16* out;              // fwd: out rows [rows][heads*32]; bwd: dqkv rows
    __bf16* qt_out;           // fwd
    int n, heads, frames, tpf;            // positions per frame, tiles per frame
    long long tiles, tiles_per_wave;      // frames * tpf; run of tiles of one wave
    float scale;
};

// x = hi + lo with both parts in the storage type
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = f2bf(x[j]);
        lo[j] = f2bf(x[j] - (float)hi[j]);
    }
}

template <bool BWD>
__global__ __launch_bounds__(SR_WAVES * 64) void sla_rows_mfma_kernel(const SrArgs a) {
    __shared__ __attribute__((aligned(16))) float rec[SR_WAVES][BWD ? 2 : 1][SR_REC];
    __shared__ __attribute__((aligned(16))) char oimg[SR_WAVES][32 * 80];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.heads, ld = 3 * H * 32, ldo = H * 32;
    const long long wave_gid = (long long)blockIdx.x * SR_WAVES + wave;
    const int head = (int)(wave_gid % H);
    const long long run = wave_gid / H;                       // this wave's run of tiles among the waves of its head
    const long long t_begin = run * a.tiles_per_wave;
    const long long t_end = t_begin + a.tiles_per_wave < a.tiles ? t_begin + a.tiles_per_wave : a.tiles;
    float* rc = rec[wave][0];
    [[maybe_unused]] float* rd = rec[wave][BWD ? 1 : 0];
    char* im = oimg[wave];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // x (lane = row of the tile, register r = feature acc_row(r, half)) -> 64-byte row segments, through LDS
    auto store_acc = [&](const f32x16& x, long long row0, int nvalid, __bf16* dst, int ldd) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = f2bf(x[4 * rg + e]);
            *reinterpret_cast<bf16x4*>(im + l31 * 80 + 16 * rg + 8 * half) = o4;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rr = 16 * j + (lane >> 2);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(im + rr * 80 + 16 * (lane & 3));
            if (rr < nvalid) *reinterpret_cast<bf16x8*>(dst + (row0 + rr) * ldd + 8 * (lane & 3)) = v;
        }
    };
    // the same for natural fragments (lane = row, features 8 half .. and 16 + 8 half ..)
    auto store_nat = [&](const bf16x8& x0, const bf16x8& x1, long long row0, int nvalid, __bf16* dst, int ldd) {
        *reinterpret_cast<bf16x8*>(im + l31 * 80 + 16 * half) = x0;
        *reinterpret_cast<bf16x8*>(im + l31 * 80 + 32 + 16 * half) = x1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rr = 16 * j + (lane >> 2);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(im + rr * 80 + 16 * (lane & 3));
            if (rr < nvalid) *reinterpret_cast<bf16x8*>(dst + (row0 + rr) * ldd + 8 * (lane & 3)) = v;
        }
    };
    // A operand [row l31 of m][k = 16 s + 8 half + j] of a 32 x 32 float matrix in LDS, as high and low parts
    auto a_rows = [&](const float* m, int s, bf16x8& hi, bf16x8& lo) {
        float x[8];
        const f32x4 u = *reinterpret_cast<const f32x4*>(m + l31 * 32 + 16 * s + 8 * half);
        const f32x4 w = *reinterpret_cast<const f32x4*>(m + l31 * 32 + 16 * s + 8 * half + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = u[j]; x[4 + j] = w[j]; }
        split8(x, hi, lo);
    };
    // the same of its transpose: element [l31][k] = m[k][l31]
    auto a_cols = [&](const float* m, int s, bf16x8& hi, bf16x8& lo) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = m[(16 * s + 8 * half + j) * 32 + l31];
        split8(x, hi, lo);
    };

    int cur_frame = -1;
    for (long long tile = t_begin; tile < t_end; ++tile) {
        const int frame = (int)(tile / a.tpf);
        const int pos0 = (int)(tile - (long long)frame * a.tpf) * 32;
        const int nvalid = a.n - pos0 < 32 ? a.n - pos0 : 32;
        const bool ok = l31 < nvalid;
        const long long row0 = (long long)frame * a.n + pos0, row = row0 + (ok ? l31 : 0);
        if (frame != cur_frame) {
            cur_frame = frame;
            const float* src = a.rec_ctx + ((long long)frame * H + head) * SR_REC;
            for (int i = lane; i < SR_REC / 4; i += 64)
                *reinterpret_cast<f32x4*>(rc + 4 * i) = *reinterpret_cast<const f32x4*>(src + 4 * i);
            if constexpr (BWD) {
                const float* sd = a.rec_dctx + ((long long)frame * H + head) * SR_REC;
                for (int i = lane; i < SR_REC / 4; i += 64)
                    *reinterpret_cast<f32x4*>(rd + 4 * i) = *reinterpret_cast<const f32x4*>(sd + 4 * i);
            }
        }
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;

        if constexpr (!BWD) {
            bf16x8 q0 = zero8, q1 = zero8;
            if (ok) {
                const __bf16* p = a.qkv + row * ld + head * 32 + 8 * half;
                q0 = *reinterpret_cast<const bf16x8*>(p);
                q1 = *reinterpret_cast<const bf16x8*>(p + 16);
            }
            float qf[16], m = -1e30f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { qf[j] = (float)q0[j]; qf[8 + j] = (float)q1[j]; }
#pragma unroll
            for (int j = 0; j < 16; ++j) m = fmaxf(m, qf[j]);
            m = xhalf_max(m);
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { qf[j] = __expf(qf[j] - m); l += qf[j]; }
            l = xhalf_add(l);
            const float inv = a.scale / l;
            bf16x8 t0, t1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { t0[j] = f2bf(qf[j] * inv); t1[j] = f2bf(qf[8 + j] * inv); }
            // out^T[e][row] = sum_d ctx[d][e] Qt[row][d]
            f32x16 o = zero;
            {
                bf16x8 hi, lo;
                a_cols(rc, 0, hi, lo);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, t0, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, t0, o, 0, 0, 0);
                a_cols(rc, 1, hi, lo);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, t1, o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, t1, o, 0, 0, 0);
            }
            store_acc(o, row0, nvalid, a.out + head * 32, ldo);
            store_nat(t0, t1, row0, nvalid, a.qt_out + head * 32, ldo);
        } else {
            bf16x8 g0 = zero8, g1 = zero8, k0 = zero8, k1 = zero8, v0 = zero8, v1 = zero8;
            float tf[16], kf[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) tf[r] = kf[r] = 0.f;
            if (ok) {
                const __bf16* gp = a.dout + row * ldo + head * 32;
                const __bf16* tp = a.qt + row * ldo + head * 32 + 4 * half;
                const __bf16* kp = a.qkv + row * ld + (H + head) * 32;
                const __bf16* vp = a.qkv + row * ld + (2 * H + head) * 32 + 8 * half;
                g0 = *reinterpret_cast<const bf16x8*>(gp + 8 * half);
                g1 = *reinterpret_cast<const bf16x8*>(gp + 16 + 8 * half);
                k0 = *reinterpret_cast<const bf16x8*>(kp + 8 * half);
                k1 = *reinterpret_cast<const bf16x8*>(kp + 16 + 8 * half);
                v0 = *reinterpret_cast<const bf16x8*>(vp);
                v1 = *reinterpret_cast<const bf16x8*>(vp + 16);
                // Qt and k once more in the layout of the results (register r = feature acc_row(r, half))
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(tp + 8 * rg);
                    const bf16x4 k4 = *reinterpret_cast<const bf16x4*>(kp + 4 * half + 8 * rg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { tf[4 * rg + e] = (float)t4[e]; kf[4 * rg + e] = (float)k4[e]; }
                }
            }
            // ---- dq = Qt * (dQt - (Qt . dQt) / scale),  dQt^T[d][row] = sum_e ctx[d][e] g[row][e]
            f32x16 dq = zero;
            {
                bf16x8 hi, lo;
                a_rows(rc, 0, hi, lo);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, g0, dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, g0, dq, 0, 0, 0);
                a_rows(rc, 1, hi, lo);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, g1, dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, g1, dq, 0, 0, 0);
            }
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += tf[r] * dq[r];
            dot = xhalf_add(dot) * (1.0f / a.scale);
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[r] = tf[r] * (dq[r] - dot);
            store_acc(dq, row0, nvalid, a.out + head * 32, ld);
            // ---- kt = softmax weight of k over the frame's positions: as an operand (natural) and in the result layout
            bf16x8 kt0, kt1;
            {
                const float* mx = rc + 1024 + 8 * half;
                const float* iv = rc + 1056 + 8 * half;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    kt0[j] = f2bf(__expf((float)k0[j] - mx[j]) * iv[j]);
                    kt1[j] = f2bf(__expf((float)k1[j] - mx[16 + j]) * iv[16 + j]);
                }
            }
            // ---- dk = kt * (dkt - S),  dkt^T[d][row] = sum_e dctx[d][e] v[row][e];   dv^T[e][row] = sum_d dctx[d][e] kt[row][d]
            f32x16 dk = zero, dv = zero;
            {
                bf16x8 hi, lo;
                a_rows(rd, 0, hi, lo);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, v0, dk, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, v0, dk, 0, 0, 0);
                a_cols(rd, 0, hi, lo);
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, kt0, dv, 0, 0, 0);
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, kt0, dv, 0, 0, 0);
                a_rows(rd, 1, hi, lo);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, v1, dk, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, v1, dk, 0, 0, 0);
                a_cols(rd, 1, hi, lo);
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, kt1, dv, 0, 0, 0);
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, kt1, dv, 0, 0, 0);
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 mx = *reinterpret_cast<const f32x4*>(rc + 1024 + 8 * rg + 4 * half);
                const f32x4 iv = *reinterpret_cast<const f32x4*>(rc + 1056 + 8 * rg + 4 * half);
                const f32x4 sv = *reinterpret_cast<const f32x4*>(rd + 1024 + 8 * rg + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rg + e;
                    dk[r] = __expf(kf[r] - mx[e]) * iv[e] * (dk[r] - sv[e]);
                }
            }
            store_acc(dk, row0, nvalid, a.out + (H + head) * 32, ld);
            store_acc(dv, row0, nvalid, a.out + (2 * H + head) * 32, ld);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The sums over a frame's positions on the matrix cores:  part[d][e] = sum_n w[n][d] b[n][e]  over one split of 1024
// positions, w = exp(a - running maximum per feature d) (SOFTMAX: a = k, b = v, with the maximum and the weight sum
// kept beside the sums, as the vector kernel leaves them for the combine launch) or w = a (a = Qt, b = gradient of out).
// The contraction index is the ROW, so both operands are the transposed tiles that ds_read_b64_tr_b16 returns from
// [row][feature] LDS images (lane = feature, 8 rows): per-feature maxima and weight sums are then lane-local.  One wave
// per (frame, head, split), the eight waves of a workgroup on eight heads of the same rows; the next tile's rows are
// requested before the current tile's products.
// ------------------------------------------------------------------------------------------------------------------
typedef short s4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_fragment(const char* base) {
    typedef __attribute__((address_space(3))) s4v* lds_s4;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base + 4 * 64));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct SredArgs {
    const __bf16* a; int lda;
    const __bf16* b; int ldb;
    float* part;              // [frame][head][split][SR_REC]
    int n, heads, nsplit, split_len;
};

template <bool SOFTMAX>
__global__ __launch_bounds__(SR_WAVES * 64) void sla_reduce_mfma_kernel(const SredArgs a) {
    __shared__ __attribute__((aligned(16))) char img[SR_WAVES][2][32 * 64];
    __shared__ __attribute__((aligned(16))) float corr_l[SR_WAVES][32];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, head = blockIdx.y * SR_WAVES + wave, frame = blockIdx.z;
    if (head >= a.heads) return;
    const int p_begin = split * a.split_len, p_end = min(p_begin + a.split_len, a.n);
    const int p_row = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int wr_off = p_row * 64 + 16 * half;
    const int g16 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_off = (8 * (g16 >> 1) + tq) * 64 + (16 * (g16 & 1) + 4 * tp) * 2;
    char* ia = img[wave][0];
    char* ib = img[wave][1];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float M = -1e30f, z = 0.f;                      // of feature d = l31 (the same in both halves of the wave)

    bf16x8 na0 = zero8, na1 = zero8, nb0 = zero8, nb1 = zero8;
    auto fetch = [&](int p0) {
        const int pos = p0 + l31;
        na0 = na1 = nb0 = nb1 = zero8;
        if (p0 < p_end && pos < p_end) {
            const long long row = (long long)frame * a.n + pos;
            const __bf16* pa = a.a + row * a.lda + head * 32 + 8 * half;
            const __bf16* pb = a.b + row * a.ldb + head * 32 + 8 * half;
            na0 = *reinterpret_cast<const bf16x8*>(pa);
            na1 = *reinterpret_cast<const bf16x8*>(pa + 16);
            nb0 = *reinterpret_cast<const bf16x8*>(pb);
            nb1 = *reinterpret_cast<const bf16x8*>(pb + 16);
        }
    };
    fetch(p_begin);
    for (int p0 = p_begin; p0 < p_end; p0 += 32) {
        const bf16x8 a0 = na0, a1 = na1, b0 = nb0, b1 = nb1;
        const int nvalid = p_end - p0;              // rows of this tile that exist (>= 32: all)
        fetch(p0 + 32);
        *reinterpret_cast<bf16x8*>(ia + wr_off) = a0;
        *reinterpret_cast<bf16x8*>(ia + wr_off + 32) = a1;
        *reinterpret_cast<bf16x8*>(ib + wr_off) = b0;
        *reinterpret_cast<bf16x8*>(ib + wr_off + 32) = b1;
        // lane = feature; element j of k-step s is the row acc_row(8 s + j, half) of the tile
        const bf16x8 at0 = tr_fragment(ia + tr_off), at1 = tr_fragment(ia + tr_off + 16 * 64);
        const bf16x8 bt0 = tr_fragment(ib + tr_off), bt1 = tr_fragment(ib + tr_off + 16 * 64);
        if constexpr (SOFTMAX) {
            float kv[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) { kv[j] = (float)at0[j]; kv[8 + j] = (float)at1[j]; }
            float mt = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (acc_row(r, half) >= nvalid) kv[r] = -1e30f;         // rows past the end weigh nothing
                mt = fmaxf(mt, kv[r]);
            }
            mt = xhalf_max(mt);
            const float Mn = fmaxf(M, mt);
            const float corr = __expf(M - Mn);
            M = Mn;
            // the sums of feature d live in the registers of row d: the factor goes from lanes to registers through LDS
            if (half == 0) corr_l[wave][l31] = corr;
            float zs = 0.f, w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { w[r] = __expf(kv[r] - Mn); zs += w[r]; }
            z = z * corr + xhalf_add(zs);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(&corr_l[wave][8 * rg + 4 * half]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * rg + e] *= c4[e];
            }
            bf16x8 h0, l0, h1, l1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                h0[j] = f2bf(w[j]);
                l0[j] = f2bf(w[j] - (float)h0[j]);
                h1[j] = f2bf(w[8 + j]);
                l1[j] = f2bf(w[8 + j] - (float)h1[j]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h0, bt0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l0, bt0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, bt1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l1, bt1, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at0, bt0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at1, bt1, acc, 0, 0, 0);
        }
    }
    float* out = a.part + (((long long)frame * a.heads + head) * a.nsplit + split) * SR_REC;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[acc_row(r, half) * 32 + l31] = acc[r];
    if (half == 0) { out[1024 + l31] = M; out[1056 + l31] = z; }
}

int g_sr_enabled = -1;

}   // namespace

// a / b: first and second operand rows (already offset to their columns), as vdmi_sla_train_reduce takes them; the
// partials have the layout of the vector kernel (sums [32][32], maximum [32], weight sum [32] per split of split_len)
int vdmi_sla_reduce_mfma_launch(bool softmax, const void* a, int lda, const void* b, int ldb, float* part, int frames, int n,
                                int heads, int nsplit, int split_len, hipStream_t stream) {
    SredArgs r{static_cast<const __bf16*>(a), lda, static_cast<const __bf16*>(b), ldb, part, n, heads, nsplit, split_len};
    const dim3 grid((unsigned)nsplit, (unsigned)((heads + SR_WAVES - 1) / SR_WAVES), (unsigned)frames);
    if (softmax) hipLaunchKernelGGL(sla_reduce_mfma_kernel<true>, grid, dim3(SR_WAVES * 64), 0, stream, r);
    else hipLaunchKernelGGL(sla_reduce_mfma_kernel<false>, grid, dim3(SR_WAVES * 64), 0, stream, r);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" void vdmi_sla_rows_mfma_enable(int on) { g_sr_enabled = on ? 1 : 0; }

bool vdmi_sla_rows_mfma_covers(int frames, int n, int heads) {
    if (g_sr_enabled < 0) {
        const char* e = std::getenv("VDMI_SLA_ROWS_MFMA");
        g_sr_enabled = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_sr_enabled) return false;
    if (frames <= 0 || n <= 0 || heads <= 0 || heads > 64) return false;
    return (long long)frames * n <= 0x7fffffffLL / (3LL * heads * 32);
}

int vdmi_sla_rows_mfma_launch(bool bwd, const void* qkv, const void* qt, const void* dout, const void* rec_ctx,
                              const void* rec_dctx, void* out, void* qt_out, int frames, int n, int heads, float scale,
                              hipStream_t stream) {
    SrArgs a{};
    a.qkv = static_cast<const __bf16*>(qkv);
    a.qt = static_cast<const __bf16*>(qt);
    a.dout = static_cast<const __bf16*>(dout);
    a.rec_ctx = static_cast<const float*>(rec_ctx);
    a.rec_dctx = static_cast<const float*>(rec_dctx);
    a.out = static_cast<__bf16*>(out);
    a.qt_out = static_cast<__bf16*>(qt_out);
    a.n = n; a.heads = heads; a.frames = frames; a.scale = scale;
    a.tpf = (n + 31) / 32;
    a.tiles = (long long)frames * a.tpf;
    // waves of one head: enough to fill the chip twice over, each with a contiguous run of tiles
    long long wph = (512LL * SR_WAVES + heads - 1) / heads;
    if (wph > a.tiles) wph = a.tiles;
    a.tiles_per_wave = (a.tiles + wph - 1) / wph;
    wph = (a.tiles + a.tiles_per_wave - 1) / a.tiles_per_wave;
    const long long waves = wph * heads;
    const long long grid = (waves + SR_WAVES - 1) / SR_WAVES;     // waves past the last run find t_begin >= tiles
    if (bwd) hipLaunchKernelGGL(sla_rows_mfma_kernel<true>, dim3((unsigned)grid), dim3(SR_WAVES * 64), 0, stream, a);
    else hipLaunchKernelGGL(sla_rows_mfma_kernel<false>, dim3((unsigned)grid), dim3(SR_WAVES * 64), 0, stream, a);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
