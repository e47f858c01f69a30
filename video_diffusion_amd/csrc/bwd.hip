// Backward pieces of ResnetBlock for gfx950 (reference: the gradients torch autograd derives for
// video_diffusion_pytorch/video_diffusion_pytorch.py:391-452 when the Trainer calls loss.backward(), :1451).
//
//   dgrad of a (1,k,k) convolution is the forward tap-GEMM over the output gradient with the weight transposed and
//   its taps reversed: vdmi_pack_taps_device writes that weight set (and the forward one) in the tap-GEMM's fragment
//   order on the device, so a training step never packs on the host.
//
//   wgrad contracts over pixels: dW[co][ci][tap] = sum_p dy[p][co] * x[p + off(tap)][ci].  With channels-last rows
//   the contraction index is the slow one, and an MFMA operand wants 8 consecutive contraction values per lane.  The
//   rows are therefore transposed once into [channel][pixel][frame] (vdmi_rows_transpose): the FRAME is the fast
//   contraction index.  A tap moves (y, x) and never the frame, so the fragment of a shifted pixel is again 16
//   aligned bytes - no halo staging, no unaligned reads, no LDS at all: each wave streams its fragments from L1/L2
//   and keeps one 32 x 32 accumulator per tap.
//
//   The RMSNorm / scale-shift / SiLU backward is a row kernel (two row reductions, per-channel sums by atomics).
#include "vdmi_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// weight packing on the device: the layout of vdmi_pack_taps (host_utils.cpp), one 16-byte piece per thread
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_taps_kernel(const float* __restrict__ w, int n, int npad, int cin0, int cin1,
                                                        int ntaps, int bkc, int transposed, __bf16* __restrict__ out,
                                                        long long pieces) {
    const long long o8 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o8 >= pieces) return;
    const int ks_per = bkc / 16, nct = npad / 32, nch0 = (cin0 + bkc - 1) / bkc;
    const int lane = (int)(o8 & 63);
    long long r = o8 >> 6;
    const int ks = (int)(r % ks_per); r /= ks_per;
    const int ct = (int)(r % nct); r /= nct;
    const int t = (int)(r % ntaps); r /= ntaps;
    const int c = (int)r;
    const bool first = c < nch0;
    const int cbase = first ? c * bkc : (c - nch0) * bkc;
    const int clim = first ? cin0 : cin1, coff = first ? 0 : cin0, cin = cin0 + cin1;
    const int co = ct * 32 + (lane & 31);
    const int ci0 = cbase + ks * 16 + 8 * (lane >> 5);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ci0 + j;
        float f = 0.f;
        if (co < n && ci < clim) {
            // transposed: `w` is the forward convolution's weight [cin][n][ntaps] seen from its output side, taps reversed
            f = transposed ? w[((size_t)(coff + ci) * n + co) * ntaps + (ntaps - 1 - t)]
                           : w[((size_t)co * cin + coff + ci) * ntaps + t];
        }
        v[j] = f2bf(f);
    }
    *reinterpret_cast<bf16x8*>(out + o8 * 8) = v;
}

// ------------------------------------------------------------------------------------------------------------------
// rows [nf*hw][ld] (channel window c_off .. c_off + c) -> T[c][hw][nfpad], frames beyond nf are zero
// ------------------------------------------------------------------------------------------------------------------
constexpr int TP = 8;   // pixels per tile
__global__ __launch_bounds__(256) void rows_transpose_kernel(const __bf16* __restrict__ rows, int ld, int c_off, int c,
                                                             int nf, int hw, int nfpad, __bf16* __restrict__ out) {
    __shared__ __bf16 tile[16][TP][72];
    const int p0 = blockIdx.x * TP, c0 = blockIdx.y * 64, f0 = blockIdx.z * 16, tid = threadIdx.x;
    // load: 16 frames x TP pixels x 8 pieces of 8 channels
    for (int i = tid; i < 16 * TP * 8; i += 256) {
        const int pc = i & 7, p = (i >> 3) % TP, f = i / (8 * TP);
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const int ch = c0 + pc * 8;
        if (f0 + f < nf && p0 + p < hw && ch < c)
            v = *reinterpret_cast<const bf16x8*>(rows + ((long long)(f0 + f) * hw + p0 + p) * ld + c_off + ch);
        *reinterpret_cast<bf16x8*>(&tile[f][p][pc * 8]) = v;
    }
    __syncthreads();
    // store: 64 channels x TP pixels x 2 pieces of 8 frames
    for (int i = tid; i < 64 * TP * 2; i += 256) {
        const int fh = i & 1, p = (i >> 1) % TP, ch = i / (2 * TP);
        if (c0 + ch >= c || p0 + p >= hw) continue;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[fh * 8 + j][p][ch];
        *reinterpret_cast<bf16x8*>(out + ((long long)(c0 + ch) * hw + p0 + p) * nfpad + f0 + fh * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad: dW[co][cin_off + ci][tap] += sum over (pixel, frame) of dyT[co][p][f] * xT[ci][p + off(tap)][f]
// A workgroup = 4 waves = a 64 x 64 (co, ci) tile; blockIdx.z = a share of the (frame block, pixel) units.
// ------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const __bf16* xT;
    const __bf16* dyT;
    float* dw;
    int cin, cout, cin_total, cin_off;
    int h, w, nfpad;
    long long units, per;        // (frame blocks of 16) * h * w, units per blockIdx.z
    int dy[9], dx[9];
};

template <int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int co = (blockIdx.x * 2 + (wave & 1)) * 32 + l31;      // this lane's A row
    const int ci = (blockIdx.y * 2 + (wave >> 1)) * 32 + l31;     // this lane's B column
    const int hw = a.h * a.w;
    const bool co_ok = co < a.cout, ci_ok = ci < a.cin;
    const __bf16* ap = a.dyT + (long long)(co_ok ? co : 0) * hw * a.nfpad + 8 * half;
    const __bf16* bp = a.xT + (long long)(ci_ok ? ci : 0) * hw * a.nfpad + 8 * half;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    long long u = (long long)blockIdx.z * a.per;
    const long long u_end = u + a.per < a.units ? u + a.per : a.units;
    if (u < u_end) {
        int fb = (int)(u / hw), p = (int)(u - (long long)fb * hw);
        int y = p / a.w, x = p - y * a.w;
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        auto load_set = [&](bf16x8& A, bf16x8 (&B)[NT], int fb_, int p_, int y_, int x_) {
            A = co_ok ? *reinterpret_cast<const bf16x8*>(ap + (long long)p_ * a.nfpad + fb_ * 16) : zero8;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int yy = y_ + a.dy[t], xx = x_ + a.dx[t];
                const bool ok = ci_ok && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
                B[t] = ok ? *reinterpret_cast<const bf16x8*>(bp + (long long)(yy * a.w + xx) * a.nfpad + fb_ * 16) : zero8;
            }
        };
        bf16x8 A, B[NT];
        load_set(A, B, fb, p, y, x);
        for (; u < u_end; ++u) {
            // next unit's fragments are requested before this unit's products
            int nx = x + 1, ny = y, np = p + 1, nfb = fb;
            if (nx == a.w) { nx = 0; ++ny; }
            if (np == hw) { np = 0; ny = 0; ++nfb; }
            bf16x8 An = zero8, Bn[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) Bn[t] = zero8;
            if (u + 1 < u_end) load_set(An, Bn, nfb, np, ny, nx);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[t], acc[t], 0, 0, 0);
            A = An;
#pragma unroll
            for (int t = 0; t < NT; ++t) B[t] = Bn[t];
            x = nx; y = ny; p = np; fb = nfb;
        }
    }
    // accumulator register r of this lane: row co_base + (r&3) + 8*(r>>2) + 4*half, column ci
    const int co_base = (blockIdx.x * 2 + (wave & 1)) * 32;
    if (ci_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = co_base + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < a.cout) {
                float* dst = a.dw + ((size_t)row * a.cin_total + a.cin_off + ci) * NT;
#pragma unroll
                for (int t = 0; t < NT; ++t) unsafeAtomicAdd(dst + t, acc[t][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of  h = SiLU( normalize(y) * sqrt(C) * gamma * (scale + 1) + shift )   (vdp.py:391-398, 421-427)
// ------------------------------------------------------------------------------------------------------------------
struct NormBwdArgs {
    const __bf16* y;  int ldy;
    const __bf16* dh; int ldh;
    __bf16* dy;       int lddy;
    const float* gamma;
    const float* scale;          // [b][sstride] or null
    const float* shift;
    long long sstride;
    float* dgamma;               // [c]
    float* dbias;                // [c] or null
    float* dscale;               // [b][c] or null
    float* dshift;
    long long rows_per_batch;
    int c;
    float norm_scale;
    long long rows_per_chunk;
};

__global__ __launch_bounds__(256) void rmsnorm_silu_bwd_kernel(const NormBwdArgs a) {
    __shared__ float colred[4 * 2048];
    __shared__ float red1[4], red2[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LP = a.c / 8, R = 256 / LP;            // threads per row, rows per pass
    const int piece = tid % LP, rslot = tid / LP;
    const int b = blockIdx.y;
    const int ch = piece * 8;
    for (int i = tid; i < 4 * a.c; i += 256) colred[i] = 0.f;

    float g[8], sc1[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        g[e] = a.gamma[ch + e];
        sc1[e] = a.scale ? a.scale[(long long)b * a.sstride + ch + e] + 1.f : 1.f;
        sh[e] = a.shift ? a.shift[(long long)b * a.sstride + ch + e] : 0.f;
    }
    float a_dg[8], a_db[8], a_ds[8], a_dsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a_dg[e] = a_db[e] = a_ds[e] = a_dsh[e] = 0.f;

    // sum over the LP consecutive threads of a row (LP a power of two; beyond a wave through LDS)
    const int wpr = LP > 64 ? LP / 64 : 1;           // waves per row
    auto row_sum = [&](float v, float* red) -> float {
        const int lim = LP < 64 ? LP : 64;
        for (int m = 1; m < lim; m <<= 1) v += __shfl_xor(v, m);
        if (LP > 64) {
            if (lane == 0) red[wave] = v;
            __syncthreads();
            float s = 0.f;
            const int w0 = (wave / wpr) * wpr;
            for (int k = 0; k < wpr; ++k) s += red[w0 + k];
            v = s;
        }
        return v;
    };

    const long long r_begin = (long long)blockIdx.x * a.rows_per_chunk;
    const long long r_end = r_begin + a.rows_per_chunk < a.rows_per_batch ? r_begin + a.rows_per_chunk : a.rows_per_batch;
    const long long passes = (a.rows_per_chunk + R - 1) / R;
    for (long long it = 0; it < passes; ++it) {
        const long long rl = r_begin + it * R + rslot;
        const bool valid = rl < r_end;
        const long long row = (long long)b * a.rows_per_batch + (valid ? rl : r_begin);
        if (LP > 64) __syncthreads();                // red1 / red2 of the previous pass have been read
        float yv[8], dhv[8];
        {
            const bf16x8 y8 = *reinterpret_cast<const bf16x8*>(a.y + row * a.ldy + ch);
            const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(a.dh + row * a.ldh + ch);
#pragma unroll
            for (int e = 0; e < 8; ++e) { yv[e] = (float)y8[e]; dhv[e] = valid ? (float)d8[e] : 0.f; }
        }
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += yv[e] * yv[e];
        ss = row_sum(ss, red1);
        const float inv_r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        float du[8], uu[8], dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float un = yv[e] * inv_r;
            const float n = un * a.norm_scale * g[e];
            const float z = n * sc1[e] + sh[e];
            const float sig = 1.0f / (1.0f + __expf(-z));
            const float dz = dhv[e] * sig * (1.0f + z * (1.0f - sig));
            a_dsh[e] += dz;
            a_ds[e] += dz * n;
            const float dn = dz * sc1[e];
            a_dg[e] += dn * un * a.norm_scale;
            du[e] = dn * a.norm_scale * g[e];
            uu[e] = un;
            dot += un * du[e];
        }
        dot = row_sum(dot, red2);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (du[e] - uu[e] * dot) * inv_r;
            a_db[e] += d;
            o[e] = f2bf(d);
        }
        if (valid) *reinterpret_cast<bf16x8*>(a.dy + row * a.lddy + ch) = o;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(&colred[0 * a.c + ch + e], a_dg[e]);
        atomicAdd(&colred[1 * a.c + ch + e], a_db[e]);
        atomicAdd(&colred[2 * a.c + ch + e], a_ds[e]);
        atomicAdd(&colred[3 * a.c + ch + e], a_dsh[e]);
    }
    __syncthreads();
    for (int i = tid; i < a.c; i += 256) {
        unsafeAtomicAdd(a.dgamma + i, colred[i]);
        if (a.dbias) unsafeAtomicAdd(a.dbias + i, colred[a.c + i]);
        if (a.dscale) unsafeAtomicAdd(a.dscale + (long long)b * a.c + i, colred[2 * a.c + i]);
        if (a.dshift) unsafeAtomicAdd(a.dshift + (long long)b * a.c + i, colred[3 * a.c + i]);
    }
}

// out[c] += sum over rows of x[row][c]   (bias gradient of a convolution whose output gradient is x)
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16* __restrict__ x, int ld, long long rows, int c,
                                                     long long rows_per_chunk, float* __restrict__ out) {
    __shared__ float colred[2048];
    const int tid = threadIdx.x, LP = c / 8, R = 256 / LP, piece = tid % LP, rslot = tid / LP;
    for (int i = tid; i < c; i += 256) colred[i] = 0.f;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long long r0 = (long long)blockIdx.x * rows_per_chunk;
    const long long r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    if (rslot < R)
        for (long long r = r0 + rslot; r < r1; r += R) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ld + piece * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    __syncthreads();
    if (rslot < R)
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&colred[piece * 8 + e], s[e]);
    __syncthreads();
    for (int i = tid; i < c; i += 256) unsafeAtomicAdd(out + i, colred[i]);
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}   // namespace

extern "C" int vdmi_pack_taps_device(const float* w, int n, int npad, int cin0, int cin1, int ntaps, int bkc,
                                     int transposed, void* out, vdmi_stream stream) {
    if (!w || !out) return VDMI_E_ARG;
    if (n <= 0 || npad < n || npad % 32 || cin0 <= 0 || cin1 < 0 || ntaps <= 0 || (bkc != 32 && bkc != 64))
        return VDMI_E_SHAPE;
    if (!aligned16(out)) return VDMI_E_ALIGN;
    const long long pieces = (long long)vdmi_pack_taps_elems(n, npad, cin0, cin1, ntaps, bkc) / 8;
    const long long blocks = (pieces + 255) / 256;
    if (blocks > 0x7fffffffLL) return VDMI_E_SHAPE;
    hipLaunchKernelGGL(pack_taps_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), w, n,
                       npad, cin0, cin1, ntaps, bkc, transposed ? 1 : 0, static_cast<__bf16*>(out), pieces);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" size_t vdmi_rows_transpose_elems(int c, int nf, int64_t hw) {
    if (c <= 0 || nf <= 0 || hw <= 0) return 0;
    return (size_t)c * (size_t)hw * (size_t)((nf + 15) / 16 * 16);
}

extern "C" int vdmi_rows_transpose(const void* rows, int ld, int c_off, int c, int nf, int64_t hw, void* out,
                                   vdmi_stream stream) {
    if (!rows || !out) return VDMI_E_ARG;
    if (c <= 0 || c % 8 || c_off < 0 || c_off % 8 || ld % 8 || c_off + c > ld || nf <= 0 || hw <= 0) return VDMI_E_SHAPE;
    if (!aligned16(rows) || !aligned16(out)) return VDMI_E_ALIGN;
    const int nfpad = (nf + 15) / 16 * 16;
    const long long px = (hw + TP - 1) / TP;
    if (px > 0x7fffffffLL || (c + 63) / 64 > 65535 || nfpad / 16 > 65535 || (long long)nf * hw > 0x7fffffffLL)
        return VDMI_E_SHAPE;
    hipLaunchKernelGGL(rows_transpose_kernel, dim3((unsigned)px, (unsigned)((c + 63) / 64), (unsigned)(nfpad / 16)),
                       dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const __bf16*>(rows), ld, c_off, c, nf,
                       (int)hw, nfpad, static_cast<__bf16*>(out));
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_conv_wgrad(const void* xT, const void* dyT, float* dw, int cin, int cout, int cin_total, int cin_off,
                               int nf, int h, int w, int ntaps, const int8_t* dy, const int8_t* dx, vdmi_stream stream) {
    if (!xT || !dyT || !dw || !dy || !dx) return VDMI_E_ARG;
    if (cin <= 0 || cout <= 0 || cin_off < 0 || cin_off + cin > cin_total || nf <= 0 || h <= 0 || w <= 0)
        return VDMI_E_SHAPE;
    if (ntaps != 1 && ntaps != 9) return VDMI_E_SHAPE;
    if (!aligned16(xT) || !aligned16(dyT)) return VDMI_E_ALIGN;
    if ((long long)h * w > 0x3fffffffLL) return VDMI_E_SHAPE;
    WgradArgs a;
    a.xT = static_cast<const __bf16*>(xT);
    a.dyT = static_cast<const __bf16*>(dyT);
    a.dw = dw;
    a.cin = cin; a.cout = cout; a.cin_total = cin_total; a.cin_off = cin_off;
    a.h = h; a.w = w; a.nfpad = (nf + 15) / 16 * 16;
    for (int t = 0; t < 9; ++t) { a.dy[t] = t < ntaps ? dy[t] : 0; a.dx[t] = t < ntaps ? dx[t] : 0; }
    a.units = (long long)(a.nfpad / 16) * h * w;
    const int tx = (cout + 63) / 64, ty = (cin + 63) / 64;
    // enough workgroups to fill the device a few times over, each with at least 16 contraction steps
    long long split = (1024 + (long long)tx * ty - 1) / ((long long)tx * ty);
    if (split > a.units / 16) split = a.units / 16;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    a.per = (a.units + split - 1) / split;
    split = (a.units + a.per - 1) / a.per;
    if (tx > 65535 || ty > 65535) return VDMI_E_SHAPE;
    const dim3 grid((unsigned)tx, (unsigned)ty, (unsigned)split);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ntaps == 9) hipLaunchKernelGGL(wgrad_kernel<9>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(256), 0, s, a);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_rmsnorm_silu_bwd(const void* y, int ldy, const void* dh, int ldh, void* dy, int lddy, int batch,
                                     int64_t rows_per_batch, int c, const float* gamma, const float* scale,
                                     const float* shift, int64_t sstride, float norm_scale, float* dgamma, float* dbias,
                                     float* dscale, float* dshift, vdmi_stream stream) {
    if (!y || !dh || !dy || !gamma || !dgamma) return VDMI_E_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return VDMI_E_ARG;
    if ((dscale || dshift) && !scale) return VDMI_E_ARG;
    if (batch <= 0 || rows_per_batch <= 0 || c < 64 || c > 2048 || !pow2(c) || ldy % 8 || ldh % 8 || lddy % 8 ||
        ldy < c || ldh < c || lddy < c || batch > 65535)
        return VDMI_E_SHAPE;
    if (!aligned16(y) || !aligned16(dh) || !aligned16(dy)) return VDMI_E_ALIGN;
    NormBwdArgs a;
    a.y = static_cast<const __bf16*>(y); a.ldy = ldy;
    a.dh = static_cast<const __bf16*>(dh); a.ldh = ldh;
    a.dy = static_cast<__bf16*>(dy); a.lddy = lddy;
    a.gamma = gamma; a.scale = scale; a.shift = shift; a.sstride = sstride;
    a.dgamma = dgamma; a.dbias = dbias; a.dscale = dscale; a.dshift = dshift;
    a.rows_per_batch = rows_per_batch; a.c = c; a.norm_scale = norm_scale;
    const int R = 256 / (c / 8);
    // chunks of ~64 passes: long enough to amortise the per-channel atomics, short enough to fill the device
    long long per = (long long)R * 64;
    long long chunks = (rows_per_batch + per - 1) / per;
    if (chunks > 0x7fffffffLL) return VDMI_E_SHAPE;
    a.rows_per_chunk = per;
    hipLaunchKernelGGL(rmsnorm_silu_bwd_kernel, dim3((unsigned)chunks, (unsigned)batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_colsum(const void* x, int ld, int64_t rows, int c, float* out, vdmi_stream stream) {
    if (!x || !out) return VDMI_E_ARG;
    if (rows <= 0 || c < 8 || c > 2048 || c % 8 || ld % 8 || ld < c) return VDMI_E_SHAPE;
    if (!aligned16(x)) return VDMI_E_ALIGN;
    const int R = 256 / (c / 8);
    if (R < 1) return VDMI_E_SHAPE;
    const long long per = (long long)R * 64;
    const long long chunks = (rows + per - 1) / per;
    if (chunks > 0x7fffffffLL) return VDMI_E_SHAPE;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(x), ld, (long long)rows, c, per, out);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
