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

int g_sr_enabled = -1;

}   // namespace

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
