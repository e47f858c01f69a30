// Small kernels of the denoising path for gfx950: init conv, final 1x1 conv, time/condition
// embedding, per-block scale/shift tables and the fused DDPM posterior step.
#include "vdmi_common.h"
#include <cstdlib>
#include <mutex>

namespace {

// ------------------------------------------------------------------------------------------
// init conv: float32 channels-first input -> bf16 channels-last, direct convolution.
// One workgroup per 16x16 pixel tile of one frame; weights [cin*k*k][cout] and the haloed input
// patch live in LDS; one lane per pixel, 16 output channels at a time.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, __bf16* __restrict__ out,
                                                        int cin, int nf, int h, int wd, int cout, int ks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pdim = 16 + ks - 1, pad = ks / 2, kk = ks * ks;
    float* wl = reinterpret_cast<float*>(smem);                 // [cin*kk][cout]
    float* pl = wl + (size_t)cin * kk * cout;                   // [cin][pdim][pdim]
    const int tid = threadIdx.x;
    const int tiles_x = (wd + 15) / 16;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int frame = blockIdx.y;                               // b*nf + f
    const int b = frame / nf, f = frame % nf;

    for (int idx = tid; idx < cin * kk * cout; idx += 256) {
        const int co = idx % cout, rest = idx / cout;           // rest = ci*kk + tap
        wl[idx] = w[(size_t)co * cin * kk + rest];
    }
    const int y0 = ty * 16 - pad, x0 = tx * 16 - pad;
    for (int idx = tid; idx < cin * pdim * pdim; idx += 256) {
        const int ci = idx / (pdim * pdim), rem = idx % (pdim * pdim);
        const int iy = y0 + rem / pdim, ix = x0 + rem % pdim;
        float v = 0.f;
        if (iy >= 0 && iy < h && ix >= 0 && ix < wd)
            v = x[(((size_t)b * cin + ci) * nf + f) * h * wd + (size_t)iy * wd + ix];
        pl[idx] = v;
    }
    __syncthreads();

    const int py = tid >> 4, px = tid & 15;
    const int oy = ty * 16 + py, ox = tx * 16 + px;
    const bool valid = oy < h && ox < wd;
    const size_t orow = ((size_t)frame * h + oy) * wd + ox;
    for (int c0 = 0; c0 < cout; c0 += 16) {
        float acc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = bias ? bias[c0 + e] : 0.f;
        for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx) {
                    const float v = pl[(ci * pdim + py + ky) * pdim + px + kx];
                    const float* wr = wl + (size_t)((ci * ks + ky) * ks + kx) * cout + c0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * q + e] += v * w4[e];
                    }
                }
        if (valid) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[8 * q + e]);
                *reinterpret_cast<bf16x8*>(out + orow * cout + c0 + 8 * q) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// init conv on the matrix cores (cout a multiple of 32): the k*k*cin taps are the contraction axis,
// padded to a multiple of 16.  Both operands are split into bf16 high and low parts
// (x = xh + xl, w = wh + wl) and y = wh*xh + wh*xl + wl*xh is accumulated in fp32, which keeps the
// result within ~2^-16 of the fp32 convolution before the single rounding to bf16.
// One workgroup per 16x16 pixel tile of one frame, a wave per 64 pixels; the weight fragments
// (rows permuted so a lane ends up with 8 consecutive channels per register pair), the haloed fp32
// patch and the tap -> patch offset table live in LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int ic_out_chan(int ct, int arow) {
    const int u = arow >> 3, hh = (arow >> 2) & 1, vv = arow & 3;
    return 32 * ct + 16 * (u >> 1) + 8 * hh + 4 * (u & 1) + vv;
}

template <int CT>
__global__ __launch_bounds__(256) void init_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, __bf16* __restrict__ out,
                                                             int cin, int nf, int h, int wd, int ks, int nks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cout = CT * 32, kk = ks * ks, ktot = cin * kk, pdim = 16 + ks - 1, pad = ks / 2;
    __bf16* wh = reinterpret_cast<__bf16*>(smem);                    // [CT][nks][64][8]
    __bf16* wl = wh + CT * nks * 512;
    int* toff = reinterpret_cast<int*>(wl + CT * nks * 512);         // [nks*16] patch offset of tap k
    float* pl = reinterpret_cast<float*>(toff + nks * 16);           // [cin][pdim][pdim]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int tiles_x = (wd + 15) / 16, ntiles = tiles_x * ((h + 15) / 16);
    const int frame = blockIdx.y, b = frame / nf, f = frame % nf;

    for (int idx = tid; idx < CT * nks * 512; idx += 256) {
        const int j = idx & 7, ln = (idx >> 3) & 63, s = (idx >> 9) % nks, ct = idx / (512 * nks);
        const int k = s * 16 + 8 * (ln >> 5) + j, co = ic_out_chan(ct, ln & 31);
        const float v = k < ktot ? w[(size_t)co * ktot + k] : 0.f;
        const __bf16 hi = f2bf(v);
        wh[idx] = hi;
        wl[idx] = f2bf(v - (float)hi);
    }
    for (int k = tid; k < nks * 16; k += 256) {
        int o = 0;                                   // padded taps read element 0 (their weights are zero)
        if (k < ktot) {
            const int ci = k / kk, r = k - ci * kk, ky = r / ks, kx = r - ky * ks;
            o = (ci * pdim + ky) * pdim + kx;
        }
        toff[k] = o;
    }
    // a workgroup converts the weights once and then walks gridDim.x-strided tiles of its frame
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ty = tile / tiles_x, tx = tile % tiles_x;
    __syncthreads();      // the previous tile's patch has been consumed
    const int y0 = ty * 16 - pad, x0 = tx * 16 - pad;
    for (int idx = tid; idx < cin * pdim * pdim; idx += 256) {
        const int ci = idx / (pdim * pdim), rem = idx % (pdim * pdim);
        const int iy = y0 + rem / pdim, ix = x0 + rem % pdim;
        float v = 0.f;
        if (iy >= 0 && iy < h && ix >= 0 && ix < wd)
            v = x[(((size_t)b * cin + ci) * nf + f) * h * wd + (size_t)iy * wd + ix];
        pl[idx] = v;
    }
    __syncthreads();

    f32x16 acc[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][j][r] = 0.f;
    int pbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave * 64 + j * 32 + l31;       // pixel of the tile: row p >> 4, column p & 15
        pbase[j] = (p >> 4) * pdim + (p & 15);
    }
    for (int s = 0; s < nks; ++s) {
        int to[8];
        {
            const int* tp = toff + s * 16 + 8 * half;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(tp), t1 = *reinterpret_cast<const f32x4*>(tp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { to[e] = __float_as_int(t0[e]); to[4 + e] = __float_as_int(t1[e]); }
        }
        bf16x8 xh[2], xl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = pl[pbase[j] + to[e]];
                const __bf16 hi = f2bf(v);
                xh[j][e] = hi;
                xl[j][e] = f2bf(v - (float)hi);
            }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wh + ((ct * nks + s) * 64 + lane) * 8);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(wl + ((ct * nks + s) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[ct][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[j], acc[ct][j], 0, 0, 0);
                acc[ct][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[j], acc[ct][j], 0, 0, 0);
                acc[ct][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[j], acc[ct][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wave * 64 + j * 32 + l31;
        const int oy = ty * 16 + (p >> 4), ox = tx * 16 + (p & 15);
        if (oy < h && ox < wd) {
            const size_t orow = ((size_t)frame * h + oy) * wd + ox;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int kl = 0; kl < 2; ++kl) {
                    const int ch = (2 * ct + kl) * 16 + 8 * half;
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[ct][j][8 * kl + e] + (bias ? bias[ch + e] : 0.f));
                    *reinterpret_cast<bf16x8*>(out + orow * cout + ch) = o;
                }
        }
    }
    }
}

// final 1x1 conv: bf16 rows -> float32 channels-first.  A row's cin/8 16-byte pieces sit on adjacent
// lanes (whole rows per wave instruction); each lane's partial dot products are combined by a butterfly.
template <int LPR>
__global__ __launch_bounds__(256) void final_conv_kernel(const __bf16* __restrict__ in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         long long rows, int cin, int nf, long long hw, int cout) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const long long m = (long long)blockIdx.x * RPB + threadIdx.x / LPR;
    const long long mr = m < rows ? m : rows - 1;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    for (int c8 = sub; c8 < cin / 8; c8 += LPR) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(in + mr * cin + c8 * 8);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < cout) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + (size_t)o * cin + c8 * 8);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + (size_t)o * cin + c8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[o] += v[e] * w0[e] + v[4 + e] * w1[e];
            }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
        if (o < cout) {
#pragma unroll
            for (int s = 1; s < LPR; s <<= 1) acc[o] += __shfl_xor(acc[o], s);
        }
    if (sub == 0 && m < rows) {
        const long long frame = m / hw, sp = m - frame * hw;
        const long long b = frame / nf, f = frame - b * nf;
#pragma unroll
        for (int o = 0; o < 8; ++o)
            if (o < cout) out[((b * cout + o) * nf + f) * hw + sp] = acc[o] + (bias ? bias[o] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// time embedding: one workgroup per batch element.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

__global__ __launch_bounds__(256) void time_embed_kernel(const int64_t* __restrict__ t, int dim,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         const float* __restrict__ cond, const float* __restrict__ null_cond,
                                                         const uint8_t* __restrict__ null_mask, int cond_dim,
                                                         float* __restrict__ emb_silu, float* __restrict__ emb_raw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* e0 = reinterpret_cast<float*>(smem);   // [dim]
    float* h1 = e0 + dim;                         // [8*dim]
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    // gridDim.y workgroups share one batch element: each recomputes the (small) first layer and
    // produces a slice of the second layer's rows
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int td = 8 * dim, half = dim / 2, ed = td + cond_dim;
    const float tv = (float)t[b];
    const float step = logf(10000.0f) / (float)(half - 1);
    for (int k = tid; k < half; k += 256) {
        const float ang = tv * expf((float)k * -step);
        e0[k] = sinf(ang);
        e0[half + k] = cosf(ang);
    }
    __syncthreads();
    // layer 1: one thread per output row, the row read as independent 16-byte loads (dim % 4 == 0)
    for (int j = tid; j < td; j += 256) {
        const float* wr = w1 + (size_t)j * dim;
        float s = 0.f;
        for (int k = 0; k < dim; k += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
            s += wv[0] * e0[k] + wv[1] * e0[k + 1] + wv[2] * e0[k + 2] + wv[3] * e0[k + 3];
        }
        const float v = s + b1[j];
        h1[j] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // exact GELU
    }
    __syncthreads();
    // layer 2: this workgroup's rows, 8 lanes per row (each lane every 8th 16-byte piece), butterfly
    const int rows_per = (td + nsplit - 1) / nsplit;
    const int j0 = split * rows_per, j1 = min(td, j0 + rows_per);
    const int sub = tid & 7;
    for (int jb = j0; jb < j1; jb += 32) {
        const int j = jb + (tid >> 3);
        const bool live = j < j1;
        const float* wr = w2 + (size_t)(live ? j : j0) * td;
        float s = 0.f;
        for (int k = sub * 4; k < td; k += 32) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(h1 + k);
            s += wv[0] * hv[0] + wv[1] * hv[1] + wv[2] * hv[2] + wv[3] * hv[3];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (live && sub == 0) {
            const float v = s + b2[j];
            emb_raw[(size_t)b * ed + j] = v;
            emb_silu[(size_t)b * ed + j] = silu_f(v);
        }
    }
    for (int k = tid; split == 0 && k < cond_dim; k += 256) {
        const bool nul = null_mask && null_mask[b];
        const float v = nul ? null_cond[k] : cond[(size_t)b * cond_dim + k];
        emb_raw[(size_t)b * ed + td + k] = v;
        emb_silu[(size_t)b * ed + td + k] = silu_f(v);
    }
}

// every ResnetBlock.mlp Linear at once: one wave per (block, channel) computes the scale and the shift row
// for 16 batch elements at a time -- lane = (batch element, quarter of the embedding), two shuffles
__global__ __launch_bounds__(256) void block_scale_shift_kernel(const float* __restrict__ emb_silu, int nb, int emb_dim,
                                                                const float* __restrict__ w_all, const float* __restrict__ b_all,
                                                                const float* __restrict__ gamma_all,
                                                                const int32_t* __restrict__ block_offsets,
                                                                const int32_t* __restrict__ block_channels, int nblocks,
                                                                int half_total, float* __restrict__ gain,
                                                                float* __restrict__ shift) {
    const int lane = threadIdx.x & 63;
    const int hr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (hr >= half_total) return;
    // block of this row: the last one whose offset is <= hr (one load per lane + a ballot, nblocks <= 64)
    const int my_off = lane < nblocks ? block_offsets[lane] : 0x7fffffff;
    const int k = __popcll(__ballot(my_off <= hr)) - 1;
    const int off = block_offsets[k], ch = block_channels[k], c = hr - off;
    const float* wsc = w_all + (size_t)(2 * off + c) * emb_dim;
    const float* wsh = w_all + (size_t)(2 * off + ch + c) * emb_dim;
    const float bsc = b_all[2 * off + c], bsh = b_all[2 * off + ch + c];
    const float g = gamma_all[hr];
    const int qp = lane & 3;
    for (int b0 = 0; b0 < nb; b0 += 16) {
        const int b = b0 + (lane >> 2);
        const float* e = emb_silu + (size_t)(b < nb ? b : 0) * emb_dim;
        float s0 = 0.f, s1 = 0.f;
        for (int q = qp; q < emb_dim; q += 4) {      // lanes of one batch element interleave: whole lines per group
            const float ev = e[q];
            s0 += wsc[q] * ev;
            s1 += wsh[q] * ev;
        }
        s0 += __shfl_xor(s0, 1);
        s1 += __shfl_xor(s1, 1);
        s0 += __shfl_xor(s0, 2);
        s1 += __shfl_xor(s1, 2);
        if (qp == 0 && b < nb) {
            gain[(size_t)b * half_total + hr] = g * (s0 + bsc + 1.0f);
            shift[(size_t)b * half_total + hr] = s1 + bsh;
        }
    }
}

// ------------------------------------------------------------------------------------------
// DDPM step.  Philox4x32-10 (counter = element group + offset, key = seed) + Box-Muller.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ void normal4(uint64_t group, uint64_t seed, float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)group, (uint32_t)(group >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float k = 2.3283064365386963e-10f;   // 2^-32
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = ((float)r[2 * p] + 1.0f) * k;          // (0, 1]
        const float u2 = (float)r[2 * p + 1] * k;
        const float rad = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        z[2 * p] = rad * cs;
        z[2 * p + 1] = rad * sn;
    }
}

// predicted x_0 of one element (vdp.py:939-943); the threshold kernel and the step kernel must agree on it bit for bit
__device__ __forceinline__ float x0_of(float a, float x, float bb, float e) { return a * x - bb * e; }

// ------------------------------------------------------------------------------------------
// Dynamic thresholding (vdp.py:957-968, 987-997): s = max(1, quantile(|x_0|, q)) per sample, with torch.quantile's
// linear interpolation between the two order statistics around rank q * (n - 1).  One workgroup per sample: the
// order statistic is found exactly by a radix select over the bit patterns of |x_0| (non-negative floats order like
// unsigned integers), four passes of a 256-bin histogram from the top byte down, and a fifth pass finds the next
// larger value when the interpolation needs it.  x_0 is recomputed from x and eps on every pass (n is small: one
// sample's elements; the passes hit L2).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dynamic_threshold_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                                 const int64_t* __restrict__ t, vdmi_ddpm_coefs cf,
                                                                 float* __restrict__ thres, long long n, float q) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sel[4];      // chosen bin, remaining rank, count <= value, smallest larger pattern
    const int tid = threadIdx.x, b = blockIdx.x;
    const int64_t tb = t[b];
    const float a = cf.sqrt_recip_alphas_cumprod[tb], bb = cf.sqrt_recipm1_alphas_cumprod[tb];
    const float* xs = x + (long long)b * n;
    const float* es = eps + (long long)b * n;
    // ranks as torch.quantile computes them for float32 input
    const float pos = q * (float)(n - 1);
    const float lo_f = floorf(pos), wgt = pos - lo_f;
    const bool two = ceilf(pos) != lo_f;
    unsigned k = (unsigned)lo_f, prefix = 0, mask = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (long long i = tid; i < n; i += 1024) {
            const unsigned u = __float_as_uint(fabsf(x0_of(a, xs[i], bb, es[i])));
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins 4l .. 4l+3; inclusive scan of the lane sums, then the lane holding rank k picks its bin
            unsigned c[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[j] = hist[4 * tid + j]; sum += c[j]; }
            unsigned incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned up = __shfl_up(incl, d);
                if (tid >= d) incl += up;
            }
            unsigned below = incl - sum;
            if (k >= below && k < incl) {
                int j = 0;
                while (j < 3 && k >= below + c[j]) { below += c[j]; ++j; }
                sel[0] = 4u * tid + j;
                sel[1] = k - below;
                sel[2] = c[j];
            }
        }
        __syncthreads();
        prefix |= sel[0] << shift;
        mask |= 255u << shift;
        k = sel[1];
        __syncthreads();
    }
    // prefix is the pattern of the order statistic of rank floor(pos); k its rank among the sel[2] equal values
    const float v_lo = __uint_as_float(prefix);
    float v_hi = v_lo;
    if (two && k + 1 >= sel[2]) {      // the next rank is the smallest value above
        if (tid == 0) sel[3] = 0xffffffffu;
        __syncthreads();
        unsigned m = 0xffffffffu;
        for (long long i = tid; i < n; i += 1024) {
            const unsigned u = __float_as_uint(fabsf(x0_of(a, xs[i], bb, es[i])));
            if (u > prefix && u < m) m = u;
        }
        atomicMin(&sel[3], m);
        __syncthreads();
        if (sel[3] != 0xffffffffu) v_hi = __uint_as_float(sel[3]);
    }
    if (tid == 0) {
        // torch's lerp
        const float d = v_hi - v_lo;
        const float s = wgt < 0.5f ? v_lo + wgt * d : v_hi - d * (1.0f - wgt);
        thres[b] = fmaxf(s, 1.0f);
    }
}

__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                        const float* __restrict__ z, const int64_t* __restrict__ t,
                                                        vdmi_ddpm_coefs cf, float* __restrict__ out, long long per_sample,
                                                        long long groups_per_sample, int clip, uint64_t seed,
                                                        uint64_t offset, const uint64_t* __restrict__ rng,
                                                        const float* __restrict__ thres, long long* t_next,
                                                        unsigned* done, int t_wrap, int b_total, int b_off) {
    // rng != null: key and counter offset live in device memory (a captured graph then draws what the buffer says)
    if (rng) { seed = rng[0]; offset = rng[1]; }
    const int b = blockIdx.y;
    const int64_t tb = t[b];
    const float a = cf.sqrt_recip_alphas_cumprod[tb], bb = cf.sqrt_recipm1_alphas_cumprod[tb];
    const float c1 = cf.posterior_mean_coef1[tb], c2 = cf.posterior_mean_coef2[tb];
    const float sigma = tb != 0 ? expf(0.5f * cf.posterior_log_variance_clipped[tb]) : 0.f;
    const long long base = (long long)b * per_sample;
    const float thr = thres ? thres[b] : 1.0f;      // dynamic thresholding: clamp to +-thres[b], then divide by it
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups_per_sample; g += (long long)gridDim.x * 256) {
        const long long i0 = g * 4;
        const int cnt = (int)((per_sample - i0) < 4 ? (per_sample - i0) : 4);
        float xv[4], ev[4], zv[4];
        if (cnt == 4 && (per_sample & 3) == 0) {
            const f32x4 xx = *reinterpret_cast<const f32x4*>(x + base + i0);
            const f32x4 ee = *reinterpret_cast<const f32x4*>(eps + base + i0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[e] = xx[e]; ev[e] = ee[e]; }
            if (z) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(z + base + i0);
#pragma unroll
                for (int e = 0; e < 4; ++e) zv[e] = zz[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = e < cnt;
                xv[e] = ok ? x[base + i0 + e] : 0.f;
                ev[e] = ok ? eps[base + i0 + e] : 0.f;
                zv[e] = (ok && z) ? z[base + i0 + e] : 0.f;
            }
        }
        // the counter includes t, which lives on the device, so a captured graph draws fresh noise each replay
        if (!z)
            normal4(((uint64_t)tb * (uint64_t)b_total + (uint64_t)(b_off + b)) * (uint64_t)groups_per_sample + (uint64_t)g + offset,
                    seed, zv);
        float ov[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x0 = x0_of(a, xv[e], bb, ev[e]);
            if (thres) x0 = fminf(fmaxf(x0, -thr), thr) / thr;
            else if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            ov[e] = c1 * x0 + c2 * xv[e] + sigma * zv[e];
        }
        if (cnt == 4 && (per_sample & 3) == 0) {
            f32x4 oo;
#pragma unroll
            for (int e = 0; e < 4; ++e) oo[e] = ov[e];
            *reinterpret_cast<f32x4*>(out + base + i0) = oo;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < cnt) out[base + i0 + e] = ov[e];
        }
    }
    // Loop form: the workgroup that finishes last moves every sample's timestep on to t - 1 (wrapping to t_wrap - 1
    // below zero), so a captured step can be replayed without the host touching t.  Every workgroup has read its t
    // before it counts itself done, and the counter is back at zero for the next launch.
    if (t_next) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned total = gridDim.x * gridDim.y;
            if (atomicAdd(done, 1u) == total - 1) {
                *done = 0;
                for (unsigned i = 0; i < gridDim.y; ++i) {
                    const long long v = t_next[i];
                    t_next[i] = v > 0 ? v - 1 : t_wrap - 1;
                }
            }
        }
    }
}

}   // namespace

extern "C" int vdmi_init_conv(const float* x, const float* w, const float* bias, void* out, int b, int cin, int f,
                              int h, int wd, int cout, int ksize, vdmi_stream stream) {
    if (!x || !w || !out) return VDMI_E_ARG;
    if (b <= 0 || cin <= 0 || f <= 0 || h <= 0 || wd <= 0 || cout <= 0 || cout % 16 || ksize < 1 || !(ksize & 1) ||
        ksize > 15)
        return VDMI_E_SHAPE;
    if (!aligned16(out)) return VDMI_E_ALIGN;
    const int pdim = 16 + ksize - 1;
    const dim3 tile_grid((unsigned)(((h + 15) / 16) * ((wd + 15) / 16)), (unsigned)(b * f));
    if (cout == 32 || cout == 64 || cout == 128) {
        const int nks = (cin * ksize * ksize + 15) / 16, ct = cout / 32;
        const size_t mlds = (size_t)2 * ct * nks * 1024 + (size_t)nks * 64 + (size_t)cin * pdim * pdim * 4;
        if (mlds <= 160 * 1024) {
            // tiles a workgroup walks (it converts the weights once, then stages patch after patch, serially)
            static const int IC_TPW = [] { const char* e = std::getenv("VDMI_IC_TPW"); const int v = e ? std::atoi(e) : 2; return v < 1 ? 1 : v; }();      // 36 us at 4, 27 at 2, 29 at 1 (8 videos of 10 x 64 x 64)
#define VDMI_IC_CASE(CT_)                                                                                    \
    if (ct == CT_) {                                                                                         \
        static std::atomic<uint64_t> attr_mask{0};                                                           \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(init_conv_mfma_kernel<CT_>), attr_mask, 160 * 1024)) \
            return VDMI_E_LAUNCH;                                                                            \
        hipLaunchKernelGGL(init_conv_mfma_kernel<CT_>, dim3((tile_grid.x + IC_TPW - 1) / IC_TPW, tile_grid.y), dim3(256), mlds, \
                           static_cast<hipStream_t>(stream), x, w, bias, static_cast<__bf16*>(out), cin, f,  \
                           h, wd, ksize, nks);                                                               \
        VDMI_LAUNCH_CHECK();                                                                                 \
        return VDMI_OK;                                                                                      \
    }
            VDMI_IC_CASE(1)
            VDMI_IC_CASE(2)
            VDMI_IC_CASE(4)
#undef VDMI_IC_CASE
        }
    }
    const size_t lds = ((size_t)cin * ksize * ksize * cout + (size_t)cin * pdim * pdim) * 4;
    if (lds > 160 * 1024) return VDMI_E_LDS;
    static std::atomic<uint64_t> attr_mask{0};
    if (vdmi_set_max_lds(reinterpret_cast<const void*>(init_conv_kernel), attr_mask, 160 * 1024)) return VDMI_E_LAUNCH;
    const dim3 grid((unsigned)(((h + 15) / 16) * ((wd + 15) / 16)), (unsigned)(b * f));
    hipLaunchKernelGGL(init_conv_kernel, grid, dim3(256), lds, static_cast<hipStream_t>(stream), x, w, bias,
                       static_cast<__bf16*>(out), cin, f, h, wd, cout, ksize);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_final_conv(const void* in, const float* w, const float* bias, float* out, int b, int cin, int f,
                               int h, int wd, int cout, vdmi_stream stream) {
    if (!in || !w || !out) return VDMI_E_ARG;
    if (b <= 0 || cin <= 0 || cin % 8 || f <= 0 || h <= 0 || wd <= 0 || cout <= 0 || cout > 8) return VDMI_E_SHAPE;
    if (!aligned16(in) || !aligned16(w)) return VDMI_E_ALIGN;
    const long long rows = (long long)b * f * h * wd;
    const int pieces = cin / 8;
#define VDMI_FC_CASE(LPR_)                                                                                  \
    {                                                                                                       \
        const long long blocks = (rows + (256 / LPR_) - 1) / (256 / LPR_);                                  \
        if (blocks > 0x7fffffffLL) return VDMI_E_SHAPE;                                                     \
        hipLaunchKernelGGL(final_conv_kernel<LPR_>, dim3((unsigned)blocks), dim3(256), 0,                   \
                           static_cast<hipStream_t>(stream), static_cast<const __bf16*>(in), w, bias, out,  \
                           rows, cin, f, (long long)h * wd, cout);                                          \
    }
    if (pieces >= 8) VDMI_FC_CASE(8)
    else if (pieces >= 4) VDMI_FC_CASE(4)
    else VDMI_FC_CASE(1)
#undef VDMI_FC_CASE
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_time_embed(const int64_t* t, int b, int dim, const float* w1, const float* b1, const float* w2,
                               const float* b2, const float* cond, const float* null_cond, const uint8_t* null_mask,
                               int cond_dim, float* emb_silu, float* emb_raw, vdmi_stream stream) {
    if (!t || !w1 || !b1 || !w2 || !b2 || !emb_silu || !emb_raw) return VDMI_E_ARG;
    if (b <= 0 || dim < 4 || dim % 4 || cond_dim < 0) return VDMI_E_SHAPE;
    if (cond_dim > 0 && (!cond || (null_mask && !null_cond))) return VDMI_E_ARG;
    if (!aligned16(w1) || !aligned16(w2)) return VDMI_E_ALIGN;
    const size_t lds = (size_t)9 * dim * 4;
    if (lds > 64 * 1024) return VDMI_E_LDS;
    hipLaunchKernelGGL(time_embed_kernel, dim3((unsigned)b, 16u), dim3(256), lds, static_cast<hipStream_t>(stream), t, dim,
                       w1, b1, w2, b2, cond, null_cond, null_mask, cond_dim, emb_silu, emb_raw);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_block_scale_shift(const float* emb_silu, int b, int emb_dim, const float* w_all,
                                      const float* b_all, const float* gamma_all, const int32_t* block_offsets,
                                      const int32_t* block_channels, int nblocks, int half_total, float* gain,
                                      float* shift, vdmi_stream stream) {
    if (!emb_silu || !w_all || !b_all || !gamma_all || !block_offsets || !block_channels || !gain || !shift)
        return VDMI_E_ARG;
    if (b <= 0 || emb_dim <= 0 || nblocks <= 0 || half_total <= 0) return VDMI_E_SHAPE;
    hipLaunchKernelGGL(block_scale_shift_kernel, dim3((unsigned)((half_total + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), emb_silu, b, emb_dim, w_all, b_all, gamma_all, block_offsets,
                       block_channels, nblocks, half_total, gain, shift);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

static int ddpm_step_launch(const float* x, const float* eps, const float* z, const int64_t* t,
                            const vdmi_ddpm_coefs* coefs, float* out, int b, int64_t per_sample, int clip,
                            uint64_t seed, uint64_t offset, const uint64_t* rng, vdmi_stream stream,
                            const float* thres = nullptr, long long* t_next = nullptr, unsigned* done = nullptr,
                            int t_wrap = 0, int b_total = 0, int b_off = 0) {
    if (!x || !eps || !t || !coefs || !out) return VDMI_E_ARG;
    if (!coefs->sqrt_recip_alphas_cumprod || !coefs->sqrt_recipm1_alphas_cumprod || !coefs->posterior_mean_coef1 ||
        !coefs->posterior_mean_coef2 || !coefs->posterior_log_variance_clipped)
        return VDMI_E_ARG;
    if (b <= 0 || b > 65535 || per_sample <= 0) return VDMI_E_SHAPE;
    // the noise counter runs over the samples of the whole batch: a launch over samples [b_off, b_off + b) of b_total draws
    // what one launch over all of them would
    if (b_total == 0) b_total = b;
    if (b_off < 0 || b_off + b > b_total) return VDMI_E_SHAPE;
    if ((per_sample & 3) == 0 && (!aligned16(x) || !aligned16(eps) || !aligned16(out) || (z && !aligned16(z))))
        return VDMI_E_ALIGN;
    const long long groups = (per_sample + 3) / 4;
    long long bx = (groups + 255) / 256;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)bx, (unsigned)b), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, eps, z, t, *coefs, out, (long long)per_sample, groups, clip,
                       seed, offset, rng, thres, t_next, done, t_wrap, b_total, b_off);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_ddpm_step(const float* x, const float* eps, const float* z, const int64_t* t,
                              const vdmi_ddpm_coefs* coefs, float* out, int b, int64_t per_sample, int clip,
                              uint64_t seed, uint64_t offset, vdmi_stream stream) {
    return ddpm_step_launch(x, eps, z, t, coefs, out, b, per_sample, clip, seed, offset, nullptr, stream);
}

extern "C" int vdmi_ddpm_step_rng(const float* x, const float* eps, const int64_t* t, const vdmi_ddpm_coefs* coefs,
                                  float* out, int b, int64_t per_sample, int clip, const uint64_t* rng_state,
                                  vdmi_stream stream) {
    if (!rng_state || (reinterpret_cast<uintptr_t>(rng_state) & 7u)) return VDMI_E_ARG;
    return ddpm_step_launch(x, eps, nullptr, t, coefs, out, b, per_sample, clip, 0, 0, rng_state, stream);
}

extern "C" int vdmi_ddpm_step_loop(const float* x, const float* eps, int64_t* t, const vdmi_ddpm_coefs* coefs, float* out,
                                   int b, int64_t per_sample, int clip, const uint64_t* rng_state, uint32_t* done_counter,
                                   int timesteps, int batch_total, int batch_offset, vdmi_stream stream) {
    if (!rng_state || (reinterpret_cast<uintptr_t>(rng_state) & 7u) || !done_counter || timesteps <= 0) return VDMI_E_ARG;
    return ddpm_step_launch(x, eps, nullptr, t, coefs, out, b, per_sample, clip, 0, 0, rng_state, stream, nullptr,
                            reinterpret_cast<long long*>(t), done_counter, timesteps, batch_total, batch_offset);
}

extern "C" int vdmi_dynamic_threshold(const float* x, const float* eps, const int64_t* t, const vdmi_ddpm_coefs* coefs,
                                      float* thres, int b, int64_t per_sample, float percentile, vdmi_stream stream) {
    if (!x || !eps || !t || !coefs || !thres) return VDMI_E_ARG;
    if (!coefs->sqrt_recip_alphas_cumprod || !coefs->sqrt_recipm1_alphas_cumprod) return VDMI_E_ARG;
    if (b <= 0 || b > 65535 || per_sample <= 0 || per_sample >= (1LL << 24)) return VDMI_E_SHAPE;   // ranks are float32, as in torch
    if (!(percentile >= 0.f && percentile <= 1.f)) return VDMI_E_ARG;
    hipLaunchKernelGGL(dynamic_threshold_kernel, dim3((unsigned)b), dim3(1024), 0, static_cast<hipStream_t>(stream), x, eps,
                       t, *coefs, thres, (long long)per_sample, percentile);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_ddpm_step_thres(const float* x, const float* eps, const float* z, const int64_t* t,
                                    const vdmi_ddpm_coefs* coefs, const float* thres, float* out, int b,
                                    int64_t per_sample, const uint64_t* rng_state, vdmi_stream stream) {
    if (!thres || (z == nullptr) == (rng_state == nullptr)) return VDMI_E_ARG;      // exactly one noise source
    if (rng_state && (reinterpret_cast<uintptr_t>(rng_state) & 7u)) return VDMI_E_ARG;
    return ddpm_step_launch(x, eps, z, t, coefs, out, b, per_sample, 1, 0, 0, rng_state, stream, thres);
}

// ------------------------------------------------------------------------------------------
// Per-step rows of the precomputed scale/shift tables of a sampling loop (see vdmi_gather_scale_shift).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_scale_shift_kernel(const float* __restrict__ gt, const float* __restrict__ st,
                                                                 int T, const long long* __restrict__ t, int cols,
                                                                 float* __restrict__ gain, float* __restrict__ shift) {
    const int b = blockIdx.y;
    const long long row = t[b];
    if (row < 0 || row >= T) return;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= cols) return;
    *reinterpret_cast<f32x4*>(gain + (long long)b * cols + c) = *reinterpret_cast<const f32x4*>(gt + row * cols + c);
    *reinterpret_cast<f32x4*>(shift + (long long)b * cols + c) = *reinterpret_cast<const f32x4*>(st + row * cols + c);
}

// ------------------------------------------------------------------------------------------
// RMSNorm over channels + scale/shift + SiLU (+ identity residual) of channels-last rows: the tail of
// Block.forward (reference vdp.py:391-427) and the ResnetBlock residual add (vdp.py:452) for wide
// layers whose convolution is split over output-channel tiles.
// LPR lanes share a row, each holding NV groups of 8 channels; the row's sum of squares is an
// in-wave butterfly.
// ------------------------------------------------------------------------------------------
template <int LPR, int NV>
__global__ __launch_bounds__(256) void rmsnorm_silu_kernel(const __bf16* __restrict__ x, int ldx,
                                                           __bf16* __restrict__ out, int ldo, long long rows,
                                                           const float* __restrict__ gain, long long gain_bstride,
                                                           const float* __restrict__ shift, long long shift_bstride,
                                                           long long rows_per_batch, float norm_scale,
                                                           const __bf16* __restrict__ res, int ld_res, int nparts,
                                                           long long part_stride) {
    constexpr int RPB = 256 / LPR;
    const int sub = threadIdx.x % LPR;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / LPR;
    const bool live = row < rows;
    const long long r = live ? row : rows - 1;
    float v[NV][8];      // the row's values (the fp32 sum of the partial inputs when the producer split its contraction)
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const __bf16* src = x + r * ldx + (k * LPR + sub) * 8;
        const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[k][e] = (float)p0[e];
        for (int p = 1; p < nparts; ++p) {
            const bf16x8 pp = *reinterpret_cast<const bf16x8*>(src + p * part_stride);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] += (float)pp[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += v[k][e] * v[k][e];
    }
    // the butterfly over the row's LPR lanes: within 16 lanes by DPP (same pairs, same bits as the shuffles it replaces)
    ss = vdmi_row_sum<(LPR < 16 ? LPR : 16)>(ss);
#pragma unroll
    for (int m = 16; m < LPR; m <<= 1) ss += __shfl_xor(ss, m);
    const float rinv = norm_scale / fmaxf(sqrtf(ss), 1e-12f);
    const long long b = r / rows_per_batch;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int ch = (k * LPR + sub) * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gain + b * gain_bstride + ch);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gain + b * gain_bstride + ch + 4);
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        if (shift) {
            s0 = *reinterpret_cast<const f32x4*>(shift + b * shift_bstride + ch);
            s1 = *reinterpret_cast<const f32x4*>(shift + b * shift_bstride + ch + 4);
        }
        bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (res) rv = *reinterpret_cast<const bf16x8*>(res + r * ld_res + ch);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = e < 4 ? g0[e & 3] : g1[e & 3], sh = e < 4 ? s0[e & 3] : s1[e & 3];
            o[e] = f2bf(silu_f(fmaf(v[k][e] * rinv, g, sh)) + (float)rv[e]);
        }
        if (live) *reinterpret_cast<bf16x8*>(out + r * ldo + ch) = o;
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm + SiLU over channels-last rows (the VAE's nn.GroupNorm(G, C) + nn.SiLU pairs,
// vdp.py:93-108, 130-143): statistics per (sample, group) over all frames, pixels and the group's
// channels.  Pass 1 writes per-(sample, split, 16-byte piece) partial sums, pass 2 combines them per
// group in its prologue and normalises.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const __bf16* __restrict__ x, long long rows_per_sample,
                                                                int c, float* __restrict__ part) {
    __shared__ float sm[2][256];
    const int ppr = c / 8, rpi = 256 / ppr;            // pieces per row, rows per iteration
    const int tid = threadIdx.x, piece = tid % ppr, r0 = tid / ppr;
    const int nsplit = gridDim.x, split = blockIdx.x, b = blockIdx.y;
    const long long per = (rows_per_sample + nsplit - 1) / nsplit;
    const long long lo = split * per, hi = lo + per < rows_per_sample ? lo + per : rows_per_sample;
    const __bf16* xb = x + (long long)b * rows_per_sample * c;
    float s1 = 0.f, s2 = 0.f;
    for (long long r = lo + r0; r < hi; r += rpi) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xb + r * c + piece * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1 += f; s2 += f * f; }
    }
    sm[0][tid] = s1;
    sm[1][tid] = s2;
    __syncthreads();
    if (tid < ppr) {
        float a1 = 0.f, a2 = 0.f;
        for (int j = tid; j < 256; j += ppr) { a1 += sm[0][j]; a2 += sm[1][j]; }
        float* dst = part + (((long long)b * nsplit + split) * ppr + tid) * 2;
        dst[0] = a1;
        dst[1] = a2;
    }
}

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ out,
                                                              long long rows_per_sample, int c, int groups,
                                                              const float* __restrict__ part, int nsplit,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, int act,
                                                              const float* __restrict__ post_gain,
                                                              const float* __restrict__ post_shift, long long post_bstride,
                                                              const __bf16* __restrict__ res, int ld_res) {
    __shared__ float mean_s[256], rstd_s[256];
    const int ppr = c / 8, rpi = 256 / ppr, ppg = ppr / groups;      // pieces per group
    const int tid = threadIdx.x, piece = tid % ppr, r0 = tid / ppr;
    const int b = blockIdx.y;
    if (tid < groups) {
        float a1 = 0.f, a2 = 0.f;
        for (int sp = 0; sp < nsplit; ++sp)
            for (int q = 0; q < ppg; ++q) {
                const float* src = part + (((long long)b * nsplit + sp) * ppr + tid * ppg + q) * 2;
                a1 += src[0];
                a2 += src[1];
            }
        const float cnt = (float)rows_per_sample * (float)(c / groups);
        const float mean = a1 / cnt;
        mean_s[tid] = mean;
        rstd_s[tid] = rsqrtf(fmaxf(a2 / cnt - mean * mean, 0.f) + eps);
    }
    __syncthreads();
    const int g = piece / ppg;
    const float mean = mean_s[g], rstd = rstd_s[g];
    float ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ga[e] = gamma[piece * 8 + e] * rstd;
        be[e] = beta[piece * 8 + e] - mean * ga[e];
    }
    const long long per = (rows_per_sample + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per, hi = lo + per < rows_per_sample ? lo + per : rows_per_sample;
    const long long base = (long long)b * rows_per_sample * c;
    // optional per-(sample, channel) affine after the activation (time-embedding scale / shift), optional residual rows
    float pg[8], ps[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        pg[e] = post_gain ? post_gain[b * post_bstride + piece * 8 + e] : 1.f;
        ps[e] = post_shift ? post_shift[b * post_bstride + piece * 8 + e] : 0.f;
    }
    const long long row0 = (long long)b * rows_per_sample;
    for (long long r = lo + r0; r < hi; r += rpi) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + base + r * c + piece * 8);
        bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (res) rv = *reinterpret_cast<const bf16x8*>(res + (row0 + r) * ld_res + piece * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = fmaf((float)v[e], ga[e], be[e]);
            if (act) y = silu_f(y);
            o[e] = f2bf(fmaf(y, pg[e], ps[e]) + (float)rv[e]);
        }
        *reinterpret_cast<bf16x8*>(out + base + r * c + piece * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------
// Depthwise temporal convolution (Conv3d(C, C, (k,1,1), padding (k/2,0,0), groups=C) of
// video_diffusion_guidance.py:324-325) over channels-last rows [b][f][hw][c]: a k-tap filter along the frame
// axis per channel.  One thread per 8 channels of an output row; neighbouring frames are hw rows away.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temporal_dwconv_kernel(const __bf16* __restrict__ x, int ldx, __bf16* __restrict__ out,
                                                              int ldo, long long rows, int f, long long hw, int c, int k,
                                                              const float* __restrict__ w, const float* __restrict__ bias) {
    const int ppr = c / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = idx / ppr;
    const int piece = (int)(idx - r * ppr);
    if (r >= rows) return;
    const int fr = (int)((r / hw) % f);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[piece * 8 + e] : 0.f;
    for (int j = 0; j < k; ++j) {
        const int ff = fr + j - k / 2;
        if (ff < 0 || ff >= f) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (r + (long long)(j - k / 2) * hw) * ldx + piece * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf((float)v[e], w[(piece * 8 + e) * k + j], acc[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
    *reinterpret_cast<bf16x8*>(out + r * ldo + piece * 8) = o;
}

static int groupnorm_splits(int64_t rows_per_sample) {
    int64_t n = rows_per_sample / 512;
    return (int)(n < 1 ? 1 : (n > 64 ? 64 : n));
}

extern "C" size_t vdmi_groupnorm_workspace_bytes(int nsamples, int64_t rows_per_sample, int c) {
    if (nsamples <= 0 || rows_per_sample <= 0 || c <= 0) return 0;
    return (size_t)nsamples * groupnorm_splits(rows_per_sample) * (c / 8) * 2 * sizeof(float);
}

extern "C" int vdmi_groupnorm(const void* x, void* out, int nsamples, int64_t rows_per_sample, int c, int groups,
                              const float* gamma, const float* beta, float eps, int act, const float* post_gain,
                              const float* post_shift, int64_t post_bstride, const void* residual, int ld_res,
                              void* workspace, vdmi_stream stream_) {
    if (!x || !out || !gamma || !beta || !workspace) return VDMI_E_ARG;
    if (act < 0 || act > 1 || (post_shift && !post_gain) || (residual && (ld_res < c || ld_res % 8 || !aligned16(residual))))
        return VDMI_E_ARG;
    if (nsamples <= 0 || nsamples > 65535 || rows_per_sample <= 0 || c < 64 || c > 2048 || 256 % (c / 8) || c % 8 ||
        groups <= 0 || groups > 256 || c % groups || (c / groups) % 8)
        return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(out)) return VDMI_E_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int nsplit = groupnorm_splits(rows_per_sample);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(groupnorm_partial_kernel, dim3((unsigned)nsplit, (unsigned)nsamples), dim3(256), 0, stream,
                       static_cast<const __bf16*>(x), (long long)rows_per_sample, c, part);
    VDMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3((unsigned)nsplit, (unsigned)nsamples), dim3(256), 0, stream,
                       static_cast<const __bf16*>(x), static_cast<__bf16*>(out), (long long)rows_per_sample, c, groups,
                       part, nsplit, gamma, beta, eps, act, post_gain, post_shift, (long long)post_bstride,
                       static_cast<const __bf16*>(residual), ld_res);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_groupnorm_silu(const void* x, void* out, int nsamples, int64_t rows_per_sample, int c, int groups,
                                   const float* gamma, const float* beta, float eps, void* workspace,
                                   vdmi_stream stream) {
    return vdmi_groupnorm(x, out, nsamples, rows_per_sample, c, groups, gamma, beta, eps, 1, nullptr, nullptr, 0, nullptr, 0,
                          workspace, stream);
}

// ------------------------------------------------------------------------------------------
// The same GroupNorm + SiLU for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses
// that forbid hipMalloc): the partial sums go to a buffer that is part of the library's code object.  Every GPU the
// library runs on holds its own copy of the array at its own address; the array's name is only a host-side handle,
// so the device address is looked up once per device (with that device current) and kept.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) float vdmi_gn_scratch[VDMI_GROUPNORM_NOALLOC_SCRATCH_BYTES / sizeof(float)];

static float* gn_scratch_device_ptr(hipStream_t stream, int* status) {
    static std::mutex mu;
    static float* ptr_of_device[64] = {nullptr};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *status = VDMI_E_LAUNCH; return nullptr; }
    // the array that is handed out is the current device's: a stream of another GPU would write to the wrong copy
    if (stream) {
        hipDevice_t sdev;
        if (hipStreamGetDevice(stream, &sdev) == hipSuccess && (int)sdev != dev) { *status = VDMI_E_ARG; return nullptr; }
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr_of_device[dev]) {
        // first call on this device: a runtime call that is not a stream operation (it may load the code object), so
        // this one call must not happen while the caller's stream is being captured
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(vdmi_gn_scratch)) != hipSuccess || !p) { *status = VDMI_E_LAUNCH; return nullptr; }
        ptr_of_device[dev] = static_cast<float*>(p);
    }
    *status = VDMI_OK;
    return ptr_of_device[dev];
}

extern "C" int vdmi_groupnorm_silu_noalloc(const void* x, void* out, int nsamples, int64_t rows_per_sample, int c, int groups,
                                           const float* gamma, const float* beta, float eps, vdmi_stream stream) {
    const size_t need = vdmi_groupnorm_workspace_bytes(nsamples, rows_per_sample, c);
    if (need == 0) return VDMI_E_SHAPE;
    if (need > (size_t)VDMI_GROUPNORM_NOALLOC_SCRATCH_BYTES) return VDMI_E_SCRATCH;
    int st = VDMI_OK;
    float* scratch = gn_scratch_device_ptr(static_cast<hipStream_t>(stream), &st);
    if (!scratch) return st;
    // the first pass overwrites every partial sum the second pass reads: nothing has to start at zero
    return vdmi_groupnorm(x, out, nsamples, rows_per_sample, c, groups, gamma, beta, eps, 1, nullptr, nullptr, 0, nullptr, 0,
                          scratch, stream);
}

extern "C" int vdmi_temporal_dwconv(const void* x, int ldx, void* out, int ldo, int b, int f, int64_t hw, int c, int k,
                                    const float* w, const float* bias, vdmi_stream stream) {
    if (!x || !out || !w) return VDMI_E_ARG;
    if (b <= 0 || f <= 0 || hw <= 0 || c <= 0 || c % 8 || k < 1 || !(k & 1) || ldx < c || ldo < c || ldx % 8 || ldo % 8)
        return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(out)) return VDMI_E_ALIGN;
    const long long rows = (long long)b * f * hw, threads = rows * (c / 8), blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return VDMI_E_SHAPE;
    hipLaunchKernelGGL(temporal_dwconv_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __bf16*>(x), ldx, static_cast<__bf16*>(out), ldo, rows, f, (long long)hw, c, k, w, bias);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

// ------------------------------------------------------------------------------------------
// Backward of vdmi_groupnorm (the GroupNorm -> SiLU -> time scale / shift chain of video_diffusion_guidance.py:334-344,
// 378-382, differentiated by the guidance sampler at :725-761).  With xh = (x - mean) * rstd, y = gamma * xh + beta,
// z = act(y), out = z * post_gain + post_shift (+ residual):
//     dy  = dout * post_gain * act'(y)                      dxh = dy * gamma
//     dx  = rstd * (dxh - mean_g(dxh) - xh * mean_g(dxh * xh))          (means over the sample's group)
// The group means follow from per-channel sums: sum_g(dxh) = sum_c gamma_c * A_c, sum_g(dxh * xh) = sum_c gamma_c * B_c
// with A_c = sum dy (the sample's share of dbeta) and B_c = sum dy * xh (of dgamma).  Four launches, no atomics (every
// sum has a fixed order): statistics of x (the forward's first pass), per-(sample, split, channel) partial sums
// {A, B, P = sum dout * z, Q = sum dout}, their reduction per sample (+ the group terms), the element pass.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float dsilu_f(float y) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
    return s * fmaf(y, 1.0f - s, 1.0f);
}

__device__ __forceinline__ void groupnorm_stats_of_sample(const float* __restrict__ part, int nsplit, int ppr, int ppg,
                                                          int groups, int b, float cnt, float eps, float* mean_s,
                                                          float* rstd_s) {
    const int tid = threadIdx.x;
    if (tid < groups) {
        float a1 = 0.f, a2 = 0.f;
        for (int sp = 0; sp < nsplit; ++sp)
            for (int q = 0; q < ppg; ++q) {
                const float* src = part + (((long long)b * nsplit + sp) * ppr + tid * ppg + q) * 2;
                a1 += src[0];
                a2 += src[1];
            }
        const float mean = a1 / cnt;
        mean_s[tid] = mean;
        rstd_s[tid] = rsqrtf(fmaxf(a2 / cnt - mean * mean, 0.f) + eps);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void groupnorm_bwd_partial_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ dout, int ld_dout, long long rows_per_sample, int c, int groups,
    const float* __restrict__ part, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
    const float* __restrict__ post_gain, long long post_bstride, float* __restrict__ part2) {
    __shared__ float mean_s[256], rstd_s[256];
    __shared__ float sm[8][256];
    const int ppr = c / 8, rpi = 256 / ppr, ppg = ppr / groups;
    const int tid = threadIdx.x, piece = tid % ppr, r0 = tid / ppr;
    const int nsplit = gridDim.x, split = blockIdx.x, b = blockIdx.y;
    groupnorm_stats_of_sample(part, nsplit, ppr, ppg, groups, b, (float)rows_per_sample * (float)(c / groups), eps, mean_s,
                              rstd_s);
    const float mean = mean_s[piece / ppg], rstd = rstd_s[piece / ppg];
    float ga[8], be[8], pg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ga[e] = gamma[piece * 8 + e];
        be[e] = beta[piece * 8 + e];
        pg[e] = post_gain ? post_gain[b * post_bstride + piece * 8 + e] : 1.f;
    }
    const long long per = (rows_per_sample + nsplit - 1) / nsplit;
    const long long lo = split * per, hi = lo + per < rows_per_sample ? lo + per : rows_per_sample;
    const long long row0 = (long long)b * rows_per_sample;
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
    for (long long r = lo + r0; r < hi; r += rpi) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (row0 + r) * c + piece * 8);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + (row0 + r) * ld_dout + piece * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)v[e] - mean) * rstd;
            const float y = fmaf(ga[e], xh, be[e]);
            const float dd = (float)d[e];
            const float dy = dd * pg[e] * (act ? dsilu_f(y) : 1.f);
            acc[0][e] += dy;
            acc[1][e] = fmaf(dy, xh, acc[1][e]);
            acc[2][e] = fmaf(dd, act ? silu_f(y) : y, acc[2][e]);
            acc[3][e] += dd;
        }
    }
    float* dst = part2 + ((long long)b * nsplit + split) * c * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[e][tid] = acc[q][e];
        __syncthreads();
        for (int ch = tid; ch < c; ch += 256) {
            float s = 0.f;
            for (int j = ch >> 3; j < 256; j += ppr) s += sm[ch & 7][j];
            dst[ch * 4 + q] = s;
        }
    }
}

// per sample: chan[b][c][4] = the partial sums over the splits; gsum[b][groups][4] = {mean, rstd, mean_g(dxh), mean_g(dxh*xh)}
__global__ __launch_bounds__(256) void groupnorm_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part2,
                                                                   int nsplit, long long rows_per_sample, int c, int groups,
                                                                   const float* __restrict__ gamma, float eps,
                                                                   float* __restrict__ chan, float* __restrict__ gsum) {
    __shared__ float mean_s[256], rstd_s[256];
    __shared__ float ga_s[2048], gb_s[2048];
    const int tid = threadIdx.x, b = blockIdx.x, ppr = c / 8, ppg = ppr / groups, cpg = c / groups;
    const float cnt = (float)rows_per_sample * (float)cpg;
    groupnorm_stats_of_sample(part, nsplit, ppr, ppg, groups, b, cnt, eps, mean_s, rstd_s);
    for (int ch = tid; ch < c; ch += 256) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < nsplit; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(part2 + (((long long)b * nsplit + sp) * c + ch) * 4);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
        *reinterpret_cast<float4*>(chan + ((long long)b * c + ch) * 4) = make_float4(s[0], s[1], s[2], s[3]);
        ga_s[ch] = gamma[ch] * s[0];
        gb_s[ch] = gamma[ch] * s[1];
    }
    __syncthreads();
    if (tid < groups) {
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < cpg; ++j) { s1 += ga_s[tid * cpg + j]; s2 += gb_s[tid * cpg + j]; }
        *reinterpret_cast<float4*>(gsum + ((long long)b * groups + tid) * 4) =
            make_float4(mean_s[tid], rstd_s[tid], s1 / cnt, s2 / cnt);
    }
}

__global__ __launch_bounds__(256) void groupnorm_bwd_apply_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ dout, int ld_dout, __bf16* __restrict__ dx,
    long long rows_per_sample, int c, int groups, const float* __restrict__ gsum, const float* __restrict__ gamma,
    const float* __restrict__ beta, int act, const float* __restrict__ post_gain, long long post_bstride) {
    const int ppr = c / 8, rpi = 256 / ppr, ppg = ppr / groups;
    const int tid = threadIdx.x, piece = tid % ppr, r0 = tid / ppr;
    const int b = blockIdx.y;
    const float4 gs = *reinterpret_cast<const float4*>(gsum + ((long long)b * groups + piece / ppg) * 4);
    const float mean = gs.x, rstd = gs.y, m1 = gs.z, m2 = gs.w;
    float ga[8], be[8], pg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ga[e] = gamma[piece * 8 + e];
        be[e] = beta[piece * 8 + e];
        pg[e] = post_gain ? post_gain[b * post_bstride + piece * 8 + e] : 1.f;
    }
    const long long per = (rows_per_sample + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per, hi = lo + per < rows_per_sample ? lo + per : rows_per_sample;
    const long long row0 = (long long)b * rows_per_sample;
    for (long long r = lo + r0; r < hi; r += rpi) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (row0 + r) * c + piece * 8);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dout + (row0 + r) * ld_dout + piece * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)v[e] - mean) * rstd;
            const float y = fmaf(ga[e], xh, be[e]);
            const float dxh = (float)d[e] * pg[e] * (act ? dsilu_f(y) : 1.f) * ga[e];
            o[e] = f2bf(rstd * (dxh - m1 - xh * m2));
        }
        *reinterpret_cast<bf16x8*>(dx + (row0 + r) * c + piece * 8) = o;
    }
}

// Weight and bias gradient of the depthwise temporal convolution: part[split][c][K + 1] = per-split sums of
// du[r][ch] * x[r + (j - K/2) * hw][ch] over the split's rows (frames outside the clip contribute nothing), j < K, and
// of du[r][ch] (the bias) at j = K.
template <int K>
__global__ __launch_bounds__(256) void temporal_dwconv_wgrad_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ du,
                                                                    long long rows, int f, long long hw, int c,
                                                                    float* __restrict__ part) {
    __shared__ float sm[8][256];
    const int ppr = c / 8, rpi = 256 / ppr;
    const int tid = threadIdx.x, piece = tid % ppr, r0 = tid / ppr;
    const long long per = (rows + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    float acc[K + 1][8];
#pragma unroll
    for (int j = 0; j <= K; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    for (long long r = lo + r0; r < hi; r += rpi) {
        const int fr = (int)((r / hw) % f);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(du + r * c + piece * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[K][e] += (float)d[e];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int ff = fr + j - K / 2;
            if (ff < 0 || ff >= f) continue;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (r + (long long)(j - K / 2) * hw) * c + piece * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = fmaf((float)d[e], (float)v[e], acc[j][e]);
        }
    }
    float* dst = part + (long long)blockIdx.x * c * (K + 1);
#pragma unroll
    for (int j = 0; j <= K; ++j) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[e][tid] = acc[j][e];
        __syncthreads();
        for (int ch = tid; ch < c; ch += 256) {
            float s = 0.f;
            for (int q = ch >> 3; q < 256; q += ppr) s += sm[ch & 7][q];
            dst[ch * (K + 1) + j] = s;
        }
    }
}

static bool groupnorm_shape_ok(int nsamples, int64_t rows_per_sample, int c, int groups) {
    return !(nsamples <= 0 || nsamples > 65535 || rows_per_sample <= 0 || c < 64 || c > 2048 || 256 % (c / 8) || c % 8 ||
             groups <= 0 || groups > 256 || c % groups || (c / groups) % 8);
}

extern "C" size_t vdmi_groupnorm_bwd_workspace_bytes(int nsamples, int64_t rows_per_sample, int c, int groups) {
    if (!groupnorm_shape_ok(nsamples, rows_per_sample, c, groups)) return 0;
    const size_t nsplit = groupnorm_splits(rows_per_sample);
    return ((size_t)nsamples * nsplit * (c / 8) * 2 + (size_t)nsamples * nsplit * c * 4 + (size_t)nsamples * groups * 4) *
           sizeof(float);
}

extern "C" int vdmi_groupnorm_bwd(const void* x, const void* dout, int ld_dout, void* dx, int nsamples, int64_t rows_per_sample,
                                  int c, int groups, const float* gamma, const float* beta, float eps, int act,
                                  const float* post_gain, int64_t post_bstride, float* chan_sums, void* workspace,
                                  vdmi_stream stream_) {
    if (!x || !dout || !dx || !gamma || !beta || !chan_sums || !workspace) return VDMI_E_ARG;
    if (act < 0 || act > 1) return VDMI_E_ARG;
    if (!groupnorm_shape_ok(nsamples, rows_per_sample, c, groups) || ld_dout < c || ld_dout % 8) return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(dout) || !aligned16(dx) || !aligned16(chan_sums) || !aligned16(workspace))
        return VDMI_E_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int nsplit = groupnorm_splits(rows_per_sample);
    float* part = static_cast<float*>(workspace);
    float* part2 = part + (size_t)nsamples * nsplit * (c / 8) * 2;
    float* gsum = part2 + (size_t)nsamples * nsplit * c * 4;
    const dim3 grid((unsigned)nsplit, (unsigned)nsamples);
    const __bf16* xb = static_cast<const __bf16*>(x);
    const __bf16* db = static_cast<const __bf16*>(dout);
    hipLaunchKernelGGL(groupnorm_partial_kernel, grid, dim3(256), 0, stream, xb, (long long)rows_per_sample, c, part);
    VDMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(groupnorm_bwd_partial_kernel, grid, dim3(256), 0, stream, xb, db, ld_dout, (long long)rows_per_sample, c,
                       groups, part, gamma, beta, eps, act, post_gain, (long long)post_bstride, part2);
    VDMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(groupnorm_bwd_reduce_kernel, dim3((unsigned)nsamples), dim3(256), 0, stream, part, part2, nsplit,
                       (long long)rows_per_sample, c, groups, gamma, eps, chan_sums, gsum);
    VDMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(groupnorm_bwd_apply_kernel, grid, dim3(256), 0, stream, xb, db, ld_dout, static_cast<__bf16*>(dx),
                       (long long)rows_per_sample, c, groups, gsum, gamma, beta, act, post_gain, (long long)post_bstride);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

static int dwconv_wgrad_splits(long long rows) {
    long long n = rows / 256;
    return (int)(n < 1 ? 1 : (n > 512 ? 512 : n));
}

extern "C" size_t vdmi_temporal_dwconv_wgrad_elems(int b, int f, int64_t hw, int c, int k) {
    if (b <= 0 || f <= 0 || hw <= 0 || c <= 0 || k < 1) return 0;
    return (size_t)dwconv_wgrad_splits((long long)b * f * hw) * c * (k + 1);
}

extern "C" int vdmi_temporal_dwconv_wgrad(const void* x, const void* du, float* part, int* nsplit_out, int b, int f,
                                          int64_t hw, int c, int k, vdmi_stream stream_) {
    if (!x || !du || !part || !nsplit_out) return VDMI_E_ARG;
    if (b <= 0 || f <= 0 || hw <= 0 || c < 8 || c > 2048 || c % 8 || 256 % (c / 8)) return VDMI_E_SHAPE;
    if (k != 1 && k != 3 && k != 5 && k != 7) return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(du)) return VDMI_E_ALIGN;
    const long long rows = (long long)b * f * hw;
    const int nsplit = dwconv_wgrad_splits(rows);
    *nsplit_out = nsplit;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const __bf16* xb = static_cast<const __bf16*>(x);
    const __bf16* db = static_cast<const __bf16*>(du);
#define VDMI_DWW(K)                                                                                                        \
    hipLaunchKernelGGL(temporal_dwconv_wgrad_kernel<K>, dim3((unsigned)nsplit), dim3(256), 0, stream, xb, db, rows, f,     \
                       (long long)hw, c, part)
    if (k == 1) VDMI_DWW(1); else if (k == 3) VDMI_DWW(3); else if (k == 5) VDMI_DWW(5); else VDMI_DWW(7);
#undef VDMI_DWW
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_rmsnorm_silu_sum(const void* x, int nparts, int64_t part_stride, int ldx, void* out, int ldo, int64_t rows,
                                     int c, const float* gain, int64_t gain_bstride, const float* shift,
                                     int64_t shift_bstride, int64_t rows_per_batch, float norm_scale, const void* res,
                                     int ld_res, vdmi_stream stream_) {
    if (!x || !out || !gain || rows <= 0 || rows_per_batch <= 0 || nparts < 1 || nparts > 4) return VDMI_E_ARG;
    if (nparts > 1 && (part_stride <= 0 || part_stride % 8)) return VDMI_E_ARG;
    if (c < 64 || c % 64 || ldx % 8 || ldo % 8 || (res && ld_res % 8) || gain_bstride % 4 || shift_bstride % 4)
        return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(out) || !aligned16(gain) || (shift && !aligned16(shift)) || (res && !aligned16(res)))
        return VDMI_E_ALIGN;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const __bf16* xp = static_cast<const __bf16*>(x);
    __bf16* op = static_cast<__bf16*>(out);
    const __bf16* rp = static_cast<const __bf16*>(res);
#define VDMI_RMS_CASE(LPR_, NV_)                                                                            \
    if (c == LPR_ * NV_ * 8) {                                                                              \
        const long long blocks = (rows + (256 / LPR_) - 1) / (256 / LPR_);                                  \
        if (blocks > 0x7fffffffLL) return VDMI_E_SHAPE;                                                     \
        hipLaunchKernelGGL((rmsnorm_silu_kernel<LPR_, NV_>), dim3((unsigned)blocks), dim3(256), 0, stream,  \
                           xp, ldx, op, ldo, (long long)rows, gain, (long long)gain_bstride, shift,         \
                           (long long)shift_bstride, (long long)rows_per_batch, norm_scale, rp, ld_res,     \
                           nparts, (long long)part_stride);                                                 \
        VDMI_LAUNCH_CHECK();                                                                                \
        return VDMI_OK;                                                                                     \
    }
    VDMI_RMS_CASE(8, 1)     // 64
    VDMI_RMS_CASE(16, 1)    // 128
    VDMI_RMS_CASE(32, 1)    // 256
    VDMI_RMS_CASE(64, 1)    // 512
    VDMI_RMS_CASE(64, 2)    // 1024
    VDMI_RMS_CASE(64, 4)    // 2048
    VDMI_RMS_CASE(8, 3)     // 192
    VDMI_RMS_CASE(16, 3)    // 384
    VDMI_RMS_CASE(32, 3)    // 768
    VDMI_RMS_CASE(64, 3)    // 1536
#undef VDMI_RMS_CASE
    return VDMI_E_SHAPE;
}

extern "C" int vdmi_rmsnorm_silu(const void* x, int ldx, void* out, int ldo, int64_t rows, int c, const float* gain,
                                 int64_t gain_bstride, const float* shift, int64_t shift_bstride,
                                 int64_t rows_per_batch, float norm_scale, const void* res, int ld_res,
                                 vdmi_stream stream) {
    return vdmi_rmsnorm_silu_sum(x, 1, 0, ldx, out, ldo, rows, c, gain, gain_bstride, shift, shift_bstride, rows_per_batch,
                                 norm_scale, res, ld_res, stream);
}

extern "C" int vdmi_gather_scale_shift(const float* gain_table, const float* shift_table, int T, const int64_t* t, int b,
                                       int cols, float* gain, float* shift, vdmi_stream stream) {
    if (!gain_table || !shift_table || !t || !gain || !shift) return VDMI_E_ARG;
    if (T <= 0 || b <= 0 || b > 65535 || cols <= 0 || cols % 4) return VDMI_E_SHAPE;
    if (!aligned16(gain_table) || !aligned16(shift_table) || !aligned16(gain) || !aligned16(shift)) return VDMI_E_ALIGN;
    hipLaunchKernelGGL(gather_scale_shift_kernel, dim3((unsigned)((cols / 4 + 255) / 256), (unsigned)b), dim3(256), 0,
                       static_cast<hipStream_t>(stream), gain_table, shift_table, T,
                       reinterpret_cast<const long long*>(t), cols, gain, shift);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
