// Training kernels of the temporal attention layer (Residual(PreNorm(Attention over frames)), reference
// video_diffusion_pytorch/video_diffusion_pytorch.py:380-389, 531-612, 636; the gradients are those torch autograd
// derives when the Trainer calls loss.backward(), :1451).
//
//   vdmi_layernorm_rows / _bwd      channel LayerNorm without gamma (gamma is folded into the projection that follows)
//   vdmi_tattn_core_fwd / _bwd      softmax attention over the f frames of a site, heads of 32 features, on the raw q, k, v
//                                   rows of the to_qkv projection: q scale, half-split rotary, relative-position bias,
//                                   softmax, .v - and the backward down to the gradient of those raw rows and of the bias
//
// The projections themselves (to_qkv, to_out) and their gradients are row GEMMs: vdmi_tapgemm, vdmi_pack_taps_device,
// vdmi_conv_wgrad.  The cores are vector kernels: a site's score matrix is f x f (f <= 32) per head, far below a
// matrix-core tile, and one lane per (row, head) with the site's keys / values in LDS keeps every product in registers.
#include "vdmi_common.h"

namespace {

inline bool pow2i(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm over channels, no affine:  xh = (x - mean) / sqrt(var + eps)   (biased variance)
// backward:  dx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh))  (+ add)
// One row = c / 8 consecutive threads (c a power of two, 64..2048), 16-byte pieces.
// ------------------------------------------------------------------------------------------------------------------
struct LnArgs {
    const __bf16* x;      // fwd: input rows; bwd: the ORIGINAL input rows (statistics are recomputed)
    const __bf16* dxh;    // bwd only
    const __bf16* add;    // bwd only, optional rows added to the result
    __bf16* out;
    long long rows;
    int c;
    float eps;
};

template <bool BWD>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const LnArgs a) {
    __shared__ float red[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LP = a.c / 8, R = 256 / LP, piece = tid % LP, rslot = tid / LP;
    const int wpr = LP > 64 ? LP / 64 : 1;
    auto row_sum = [&](float v, int slot) -> float {
        // LP is a power of two from 8: the first 16 lanes by DPP, then across DPP rows
        v = LP >= 16 ? vdmi_row_sum<16>(v) : vdmi_row_sum<8>(v);
        if (LP >= 32) v += __shfl_xor(v, 16);
        if (LP >= 64) v = xhalf_add(v);
        if (LP > 64) {
            if (lane == 0) red[slot][wave] = v;
            __syncthreads();
            float s = 0.f;
            const int w0 = (wave / wpr) * wpr;
            for (int k = 0; k < wpr; ++k) s += red[slot][w0 + k];
            v = s;
        }
        return v;
    };
    const long long passes = (a.rows + (long long)gridDim.x * R - 1) / ((long long)gridDim.x * R);
    for (long long it = 0; it < passes; ++it) {
        const long long r = (it * gridDim.x + blockIdx.x) * R + rslot;
        const bool valid = r < a.rows;
        const long long row = valid ? r : 0;
        if (LP > 64) __syncthreads();
        float xv[8];
        {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.x + row * a.c + piece * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = (float)v[e];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += xv[e]; s2 += xv[e] * xv[e]; }
        s1 = row_sum(s1, 0);
        s2 = row_sum(s2, 1);
        const float mean = s1 / a.c;
        const float rstd = rsqrtf(fmaxf(s2 / a.c - mean * mean, 0.f) + a.eps);
        bf16x8 o;
        if constexpr (!BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf((xv[e] - mean) * rstd);
        } else {
            float g[8], m1 = 0.f, m2 = 0.f;
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(a.dxh + row * a.c + piece * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                g[e] = (float)d[e];
                xv[e] = (xv[e] - mean) * rstd;
                m1 += g[e];
                m2 += g[e] * xv[e];
            }
            if (LP > 64) __syncthreads();      // red[0] / red[1] are reused below
            m1 = row_sum(m1, 2) / a.c;
            if (LP > 64) __syncthreads();
            m2 = row_sum(m2, 0) / a.c;
            bf16x8 ad = {0, 0, 0, 0, 0, 0, 0, 0};
            if (a.add) ad = *reinterpret_cast<const bf16x8*>(a.add + row * a.c + piece * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(rstd * (g[e] - m1 - xv[e] * m2) + (float)ad[e]);
        }
        if (valid) *reinterpret_cast<bf16x8*>(a.out + row * a.c + piece * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Attention core over frames (or, with hw = 1, over the f consecutive rows of a group: the mid-block spatial attention over
// the positions of a frame).  qkv rows [b*f*hw][3*heads*32] (q | k | v, column = head*32 + feature); a site's f rows are
// hw rows apart.  A workgroup takes `spw` sites and `hpw` heads at a time; thread = (site, head, frame).
// ------------------------------------------------------------------------------------------------------------------
struct CoreArgs {
    const __bf16* qkv;
    const __bf16* o;        // bwd
    const __bf16* dout;     // bwd: gradient of o
    const float* lse;       // bwd: [rows][heads]
    __bf16* out;            // fwd: o rows [rows][heads*32]; bwd: dqkv rows [rows][3*heads*32]
    float* lse_out;         // fwd
    float* dbias;           // bwd: [heads][f][f], accumulated
    const float* bias;      // [heads][f][f] or null
    const float* rot_cos;   // [f][16] or null
    const float* rot_sin;
    const unsigned char* focus;   // [b] or null: samples whose frames attend to themselves only (vdp.py:590-600)
    int b, f, hw, heads;
    int spw;                // sites per workgroup pass
    int hpw, hgroups;       // heads per workgroup pass, heads / hpw (long token axes leave room for few heads at a time)
    long long groups;       // ceil(b*hw / spw) * hgroups
    float scale;
};

__device__ __forceinline__ void load32(const __bf16* p, float (&v)[32]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(p + 8 * k);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[8 * k + e] = (float)t[e];
    }
}
__device__ __forceinline__ void store32(__bf16* p, const float (&v)[32]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = f2bf(v[8 * k + e]);
        *reinterpret_cast<bf16x8*>(p + 8 * k) = t;
    }
}
// half-split rotary of a 32-feature head (feature d pairs with d + 16); inverse = the transposed rotation
__device__ __forceinline__ void rotate32(float (&v)[32], const float* c, const float* s, bool inverse) {
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        const float cs = c[d], sn = inverse ? -s[d] : s[d];
        const float lo = v[d], hi = v[d + 16];
        v[d] = lo * cs - hi * sn;
        v[d + 16] = hi * cs + lo * sn;
    }
}
__device__ __forceinline__ void lds_store32(float* p, const float (&v)[32]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4*>(p + 4 * k) = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
}
__device__ __forceinline__ float lds_dot32(const float* p, const float (&v)[32]) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * k), b = *reinterpret_cast<const f32x4*>(p + 4 * k + 4);
        s0 += a[0] * v[4 * k] + a[1] * v[4 * k + 1] + a[2] * v[4 * k + 2] + a[3] * v[4 * k + 3];
        s1 += b[0] * v[4 * k + 4] + b[1] * v[4 * k + 5] + b[2] * v[4 * k + 6] + b[3] * v[4 * k + 7];
    }
    return s0 + s1;
}
// a row read once into registers for a dot product AND an update (the LDS atomics of the bias gradient sit between the two
// uses and keep the compiler from merging two reads of the same row)
__device__ __forceinline__ void lds_load32(const float* p, float (&r)[32]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * k);
        r[4 * k] = a[0]; r[4 * k + 1] = a[1]; r[4 * k + 2] = a[2]; r[4 * k + 3] = a[3];
    }
}
__device__ __forceinline__ float dot32(const float (&r)[32], const float (&v)[32]) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 32; k += 2) { s0 += r[k] * v[k]; s1 += r[k + 1] * v[k + 1]; }
    return s0 + s1;
}
__device__ __forceinline__ void axpy32(const float (&r)[32], float w, float (&acc)[32]) {
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] += w * r[k];
}
__device__ __forceinline__ void lds_axpy32(const float* p, float w, float (&acc)[32]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * k + e] += w * a[e];
    }
}

// LDS tile of one tensor: [site][frame][head][32] floats, rows padded by 4 floats against bank conflicts between heads
#define CORE_ROWF 36

template <bool BWD>
__global__ __launch_bounds__(256) void tattn_core_kernel(const CoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int F = a.f, H = a.heads, HP = a.hpw, per_site = F * HP, ld = 3 * H * 32, ldo = H * 32;
    const int tile = a.spw * per_site * CORE_ROWF;                // floats per tensor
    float* Kl = reinterpret_cast<float*>(smem);
    float* Vl = Kl + tile;
    float* Ql = Vl + tile;                                        // bwd only
    float* Gl = Ql + tile;                                        // bwd only: gradient of o
    float* Sl = Gl + tile;                                        // bwd only: [slot][2] = (lse, D)
    float* Bl = Sl + 2 * a.spw * per_site;                        // bwd with a bias only: dbias [heads][f][f]
    const bool active = tid < a.spw * per_site;
    const int s = tid / per_site, rem = tid - s * per_site, hl = rem / F, i = rem - hl * F;
    const int slot = (s * F + i) * HP + hl;                       // this thread's row slot in the tiles
    if constexpr (BWD) {
        if (a.dbias)
            for (int k = tid; k < H * F * F; k += 256) Bl[k] = 0.f;
    }
    float rc[16], rs[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        rc[d] = (a.rot_cos && active) ? a.rot_cos[i * 16 + d] : 1.f;
        rs[d] = (a.rot_sin && active) ? a.rot_sin[i * 16 + d] : 0.f;
    }
    const long long nsites = (long long)a.b * a.hw;
    for (long long grp = blockIdx.x; grp < a.groups; grp += gridDim.x) {
        const long long site = (grp / a.hgroups) * a.spw + s;
        const int h = (int)(grp % a.hgroups) * HP + hl;           // this pass's heads
        const bool valid = active && site < nsites;
        const long long bb = valid ? site / a.hw : 0, sp = valid ? site - bb * a.hw : 0;
        const long long row = (bb * F + i) * a.hw + sp;
        float q[32], k[32], v[32];
        __syncthreads();                                          // the previous pass is done with the tiles
        if (valid) {
            load32(a.qkv + row * ld + h * 32, q);
            load32(a.qkv + row * ld + (H + h) * 32, k);
            load32(a.qkv + row * ld + (2 * H + h) * 32, v);
            rotate32(q, rc, rs, false);
            rotate32(k, rc, rs, false);
#pragma unroll
            for (int d = 0; d < 32; ++d) q[d] *= a.scale;
            lds_store32(Kl + slot * CORE_ROWF, k);
            lds_store32(Vl + slot * CORE_ROWF, v);
        }
        const float* bias_i = a.bias ? a.bias + ((long long)h * F + i) * F : nullptr;
        // focus-present samples: every score off the diagonal is masked (its probability is an exact zero, so are its
        // gradients; the diagonal's own gradient cancels: a softmax over one element is constant)
        const bool foc = a.focus && valid && a.focus[bb] != 0;
        if constexpr (!BWD) {
            __syncthreads();
            if (valid) {
                float m = -1e30f, l = 0.f, o[32];
#pragma unroll
                for (int d = 0; d < 32; ++d) o[d] = 0.f;
                for (int j = 0; j < F; ++j) {
                    const int sj = ((s * F + j) * HP + hl) * CORE_ROWF;
                    float sc = lds_dot32(Kl + sj, q) + (bias_i ? bias_i[j] : 0.f);
                    if (foc && j != i) sc = -1e30f;
                    const float mn = fmaxf(m, sc);
                    const float corr = __expf(m - mn), p = __expf(sc - mn);
                    l = l * corr + p;
#pragma unroll
                    for (int d = 0; d < 32; ++d) o[d] *= corr;
                    lds_axpy32(Vl + sj, p, o);
                    m = mn;
                }
                const float inv = 1.0f / l;
#pragma unroll
                for (int d = 0; d < 32; ++d) o[d] *= inv;
                store32(a.out + row * ldo + h * 32, o);
                a.lse_out[row * H + h] = m + __logf(l);
            }
        } else {
            float g[32], lse = 0.f, D = 0.f;
            if (valid) {
                float o[32];
                load32(a.dout + row * ldo + h * 32, g);
                load32(a.o + row * ldo + h * 32, o);
#pragma unroll
                for (int d = 0; d < 32; ++d) D += g[d] * o[d];
                lse = a.lse[row * H + h];
                lds_store32(Ql + slot * CORE_ROWF, q);
                lds_store32(Gl + slot * CORE_ROWF, g);
                Sl[2 * slot] = lse;
                Sl[2 * slot + 1] = D;
            }
            __syncthreads();
            if (valid) {
                // this row as a query: dq = sum_j dS_ij k_j,  dS_ij = p_ij (dO_i . v_j - D_i);  the bias gradient is dS
                float dq[32];
#pragma unroll
                for (int d = 0; d < 32; ++d) dq[d] = 0.f;
                for (int j = 0; j < F; ++j) {
                    const int sj = ((s * F + j) * HP + hl) * CORE_ROWF;
                    float kr[32];
                    lds_load32(Kl + sj, kr);
                    const float sc = dot32(kr, q) + (bias_i ? bias_i[j] : 0.f);
                    const float p = (foc && j != i) ? 0.f : __expf(sc - lse);
                    const float ds = foc ? 0.f : p * (lds_dot32(Vl + sj, g) - D);      // one unmasked key: the softmax is constant
                    axpy32(kr, ds, dq);
                    if (a.dbias) atomicAdd(&Bl[(h * F + i) * F + j], ds);
                }
#pragma unroll
                for (int d = 0; d < 32; ++d) dq[d] *= a.scale;
                rotate32(dq, rc, rs, true);
                store32(a.out + row * ld + h * 32, dq);
                // this row as a key / value: dk = sum_i dS_ij q_i,  dv = sum_i p_ij dO_i   (q_i already scaled)
                float dk[32], dv[32];
#pragma unroll
                for (int d = 0; d < 32; ++d) dk[d] = dv[d] = 0.f;
                for (int qi = 0; qi < F; ++qi) {
                    const int slq = (s * F + qi) * HP + hl, sq = slq * CORE_ROWF;
                    const float bq = a.bias ? a.bias[((long long)h * F + qi) * F + i] : 0.f;
                    float qr[32], gr[32];
                    lds_load32(Ql + sq, qr);
                    lds_load32(Gl + sq, gr);
                    const float sc = dot32(qr, k) + bq;
                    const float p = (foc && qi != i) ? 0.f : __expf(sc - Sl[2 * slq]);
                    const float ds = foc ? 0.f : p * (dot32(gr, v) - Sl[2 * slq + 1]);
                    axpy32(qr, ds, dk);
                    axpy32(gr, p, dv);
                }
                rotate32(dk, rc, rs, true);
                store32(a.out + row * ld + (H + h) * 32, dk);
                store32(a.out + row * ld + (2 * H + h) * 32, dv);
            }
        }
    }
    if constexpr (BWD) {
        __syncthreads();
        if (a.dbias)
            for (int k = tid; k < H * F * F; k += 256) unsafeAtomicAdd(a.dbias + k, Bl[k]);
    }
}

size_t core_lds_bytes(bool bwd, int f, int hpw, int spw, int bias_heads) {
    const size_t tile = (size_t)spw * f * hpw * CORE_ROWF * 4;
    return bwd ? 4 * tile + (size_t)2 * spw * f * hpw * 4 + (size_t)bias_heads * f * f * 4 : 2 * tile;
}

// The most heads per pass (a divisor of heads) and sites per pass that 256 threads and the LDS hold; false when no
// split fits (long token axes: the tiles and, in the backward, the bias-gradient table [heads][f][f] live in LDS).
bool core_split(bool bwd, int f, int heads, bool with_dbias, int* hpw_out, int* spw_out, size_t* lds_out) {
    if (f <= 0 || f > 256 || heads <= 0) return false;
    const int bias_heads = (bwd && with_dbias) ? heads : 0;
    for (int hp = heads; hp >= 1; --hp) {
        if (heads % hp || f * hp > 256) continue;
        for (int sp = 256 / (f * hp); sp >= 1; --sp)
            if (core_lds_bytes(bwd, f, hp, sp, bias_heads) <= 160 * 1024) {
                if (hpw_out) *hpw_out = hp;
                if (spw_out) *spw_out = sp;
                if (lds_out) *lds_out = core_lds_bytes(bwd, f, hp, sp, bias_heads);
                return true;
            }
    }
    return false;
}

int core_launch(bool bwd, CoreArgs& a, hipStream_t stream) {
    if (a.b <= 0 || a.f <= 0 || a.f > 256 || a.hw <= 0 || a.heads <= 0) return VDMI_E_SHAPE;
    if ((a.rot_cos == nullptr) != (a.rot_sin == nullptr)) return VDMI_E_ARG;
    int hpw = 0, spw = 0;
    size_t lds = 0;
    if (!core_split(bwd, a.f, a.heads, a.dbias != nullptr, &hpw, &spw, &lds)) return VDMI_E_LDS;
    a.spw = spw;
    a.hpw = hpw;
    a.hgroups = a.heads / hpw;
    const long long nsites = (long long)a.b * a.hw;
    a.groups = ((nsites + spw - 1) / spw) * a.hgroups;
    if ((long long)a.b * a.f * a.hw > 0x7fffffffLL) return VDMI_E_SHAPE;
    // persistent workgroups (the backward flushes its bias-gradient table once per workgroup)
    const long long grid = a.groups < 1024 ? a.groups : 1024;
    static std::atomic<uint64_t> mask_f{0}, mask_b{0};
    if (bwd) {
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(tattn_core_kernel<true>), mask_b, 160 * 1024)) return VDMI_E_LAUNCH;
        hipLaunchKernelGGL(tattn_core_kernel<true>, dim3((unsigned)grid), dim3(256), lds, stream, a);
    } else {
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(tattn_core_kernel<false>), mask_f, 160 * 1024)) return VDMI_E_LAUNCH;
        hipLaunchKernelGGL(tattn_core_kernel<false>, dim3((unsigned)grid), dim3(256), lds, stream, a);
    }
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

}   // namespace

// tattn_core_mfma.hip: the matrix-core form for frame axes of up to 32
bool vdmi_tattn_core_mfma_covers(int b, int f, int hw, int heads);
size_t vdmi_tattn_core_mfma_workspace(int b, int f, int hw, int heads);
int vdmi_tattn_core_mfma_launch(bool bwd, const void* qkv, const void* o, const void* dout, const float* lse, void* out,
                                float* lse_out, float* dbias, void* workspace, const float* bias, const float* rot_cos,
                                const float* rot_sin, const unsigned char* focus, int b, int f, int hw, int heads,
                                float scale, hipStream_t stream);

namespace {

int ln_check(const void* x, const void* out, int64_t rows, int c) {
    if (!x || !out) return VDMI_E_ARG;
    if (rows <= 0 || c < 64 || c > 2048 || !pow2i(c)) return VDMI_E_SHAPE;
    if (!aligned16(x) || !aligned16(out)) return VDMI_E_ALIGN;
    return VDMI_OK;
}

}   // namespace

extern "C" int vdmi_tattn_core_fits(int bwd, int f, int heads, int with_dbias) {
    return core_split(bwd != 0, f, heads, with_dbias != 0, nullptr, nullptr, nullptr) ? 1 : 0;
}

extern "C" size_t vdmi_tattn_core_bwd_workspace_bytes(int b, int f, int hw, int heads) {
    return vdmi_tattn_core_mfma_covers(b, f, hw, heads) ? vdmi_tattn_core_mfma_workspace(b, f, hw, heads) : 0;
}

extern "C" int vdmi_layernorm_rows(const void* x, void* xh, int64_t rows, int c, float eps, vdmi_stream stream) {
    const int st = ln_check(x, xh, rows, c);
    if (st != VDMI_OK) return st;
    LnArgs a{static_cast<const __bf16*>(x), nullptr, nullptr, static_cast<__bf16*>(xh), (long long)rows, c, eps};
    const int R = 256 / (c / 8);
    long long blocks = (rows + R - 1) / R;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(layernorm_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_layernorm_rows_bwd(const void* x, const void* dxh, const void* add, void* dx, int64_t rows, int c,
                                       float eps, vdmi_stream stream) {
    const int st = ln_check(x, dx, rows, c);
    if (st != VDMI_OK) return st;
    if (!dxh) return VDMI_E_ARG;
    if (!aligned16(dxh) || (add && !aligned16(add))) return VDMI_E_ALIGN;
    LnArgs a{static_cast<const __bf16*>(x), static_cast<const __bf16*>(dxh), static_cast<const __bf16*>(add),
             static_cast<__bf16*>(dx), (long long)rows, c, eps};
    const int R = 256 / (c / 8);
    long long blocks = (rows + R - 1) / R;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(layernorm_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_tattn_core_fwd(const void* qkv, void* o, float* lse, const float* bias, const float* rot_cos,
                                   const float* rot_sin, const uint8_t* focus, int b, int f, int hw, int heads, float scale,
                                   vdmi_stream stream) {
    if (!qkv || !o || !lse) return VDMI_E_ARG;
    if (!aligned16(qkv) || !aligned16(o)) return VDMI_E_ALIGN;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return VDMI_E_ARG;
    if (vdmi_tattn_core_mfma_covers(b, f, hw, heads))
        return vdmi_tattn_core_mfma_launch(false, qkv, nullptr, nullptr, nullptr, o, lse, nullptr, nullptr, bias, rot_cos,
                                           rot_sin, focus, b, f, hw, heads, scale, static_cast<hipStream_t>(stream));
    CoreArgs a{};
    a.qkv = static_cast<const __bf16*>(qkv);
    a.out = static_cast<__bf16*>(o);
    a.lse_out = lse;
    a.bias = bias; a.rot_cos = rot_cos; a.rot_sin = rot_sin; a.focus = focus;
    a.b = b; a.f = f; a.hw = hw; a.heads = heads; a.scale = scale;
    return core_launch(false, a, static_cast<hipStream_t>(stream));
}

extern "C" int vdmi_tattn_core_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv,
                                   float* dbias, const float* bias, const float* rot_cos, const float* rot_sin,
                                   const uint8_t* focus, int b, int f, int hw, int heads, float scale, void* workspace,
                                   vdmi_stream stream) {
    if (!qkv || !o || !dout || !lse || !dqkv) return VDMI_E_ARG;
    if (dbias && !bias) return VDMI_E_ARG;
    if (!aligned16(qkv) || !aligned16(o) || !aligned16(dout) || !aligned16(dqkv)) return VDMI_E_ALIGN;
    if ((rot_cos == nullptr) != (rot_sin == nullptr)) return VDMI_E_ARG;
    // the matrix-core form keeps its bias-gradient partial sums in the workspace; without one the vector form (which
    // accumulates with atomics) serves a call that wants the bias gradient
    if (vdmi_tattn_core_mfma_covers(b, f, hw, heads) && (!dbias || (workspace && aligned16(workspace))))
        return vdmi_tattn_core_mfma_launch(true, qkv, o, dout, lse, dqkv, nullptr, dbias, workspace, bias, rot_cos, rot_sin,
                                           focus, b, f, hw, heads, scale, static_cast<hipStream_t>(stream));
    CoreArgs a{};
    a.qkv = static_cast<const __bf16*>(qkv);
    a.o = static_cast<const __bf16*>(o);
    a.dout = static_cast<const __bf16*>(dout);
    a.lse = lse;
    a.out = static_cast<__bf16*>(dqkv);
    a.dbias = dbias;
    a.bias = bias; a.rot_cos = rot_cos; a.rot_sin = rot_sin; a.focus = focus;
    a.b = b; a.f = f; a.hw = hw; a.heads = heads; a.scale = scale;
    return core_launch(true, a, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------------------------
// SpatialLinearAttention (vdp.py:454-479), training form.  Per (frame, head), n = h*w positions, 32 features:
//   Kt = softmax over the positions of k,   Qt = softmax over the features of q, times scale
//   ctx[d][e] = sum_n Kt[n][d] v[n][e],     out[n][e] = sum_d Qt[n][d] ctx[d][e]
// Backward (with dctx[d][e] = sum_n Qt[n][d] dout[n][e] and S[d] = sum_e dctx[d][e] ctx[d][e], which equals
// sum_n Kt[n][d] dKt[n][d]):
//   dq = scale * sigma * (dQt - sigma . dQt),  dQt[d] = ctx[d][:] . dout      (sigma = Qt / scale)
//   dk[n][d] = Kt[n][d] * (dctx[d][:] . v[n] - S[d]),    dv[n][e] = sum_d Kt[n][d] dctx[d][e]
// Two kinds of kernel: a reduction over positions that leaves per-split partial 32 x 32 sums (with an online softmax
// of the first operand for ctx), and row kernels (one lane per (row, head)) that first combine the partials of their
// frame in LDS.  Rows: row = frame * n + position, column = head * 32 + feature.
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int SLA_SPLIT = 1024;               // positions per reduction workgroup
constexpr int SLA_PART = 32 * 32 + 64;        // floats per partial: sums [32][32], then max [32], then weight sum [32]

struct SlaRedArgs {
    const __bf16* a; int lda;                 // first operand rows (k, or the saved Qt), already offset to its columns
    const __bf16* b; int ldb;                 // second operand rows (v, or dout)
    float* part;                              // [frame][head][split][SLA_PART]
    int n, heads, nsplit;
};

template <bool SOFTMAX>
__global__ __launch_bounds__(256) void sla_reduce_kernel(const SlaRedArgs a) {
    __shared__ float al[64][33];
    __shared__ __attribute__((aligned(16))) float bl[64][32];
    const int tid = threadIdx.x, d = tid >> 3, e0 = (tid & 7) * 4;
    const int split = blockIdx.x, head = blockIdx.y, frame = blockIdx.z;
    const int p_begin = split * SLA_SPLIT, p_end = min(p_begin + SLA_SPLIT, a.n);
    float m = -1e30f, z = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = p_begin; p0 < p_end; p0 += 64) {
        __syncthreads();
        {   // one 16-byte piece of each operand per thread: position tid / 4, features (tid % 4) * 8 ..
            const int pp = tid >> 2, f8 = (tid & 3) * 8, pos = p0 + pp;
            bf16x8 va = {0, 0, 0, 0, 0, 0, 0, 0}, vb = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pos < p_end) {
                const long long row = (long long)frame * a.n + pos;
                va = *reinterpret_cast<const bf16x8*>(a.a + row * a.lda + head * 32 + f8);
                vb = *reinterpret_cast<const bf16x8*>(a.b + row * a.ldb + head * 32 + f8);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { al[pp][f8 + j] = (float)va[j]; bl[pp][f8 + j] = (float)vb[j]; }
        }
        __syncthreads();
        const int cnt = min(64, p_end - p0);
        for (int pp = 0; pp < cnt; ++pp) {
            float w = al[pp][d];
            if constexpr (SOFTMAX) {
                if (w > m) {
                    const float corr = __expf(m - w);
                    z *= corr;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] *= corr;
                    m = w;
                }
                w = __expf(w - m);
                z += w;
            }
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&bl[pp][e0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += w * bv[j];
        }
    }
    float* out = a.part + (((long long)frame * a.heads + head) * a.nsplit + split) * SLA_PART;
    *reinterpret_cast<f32x4*>(out + d * 32 + e0) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    if ((tid & 7) == 0) { out[1024 + d] = m; out[1056 + d] = z; }
}

constexpr int SLA_COMB = 32 * 32 + 64;        // floats per combined (frame, head): mode 0: ctx [32][32], max [32], 1 / weight sum [32]
                                              //                                      mode 1: dctx [32][32], S [32], unused [32]

// one workgroup per (head, frame): the partials of all splits -> one combined record
struct SlaCombArgs {
    const float* part;
    const float* comb_ctx;    // mode 1: the mode-0 record (S = sum_e dctx * ctx)
    float* comb;
    int heads, nsplit, mode;
};

__global__ __launch_bounds__(256) void sla_combine_kernel(const SlaCombArgs a) {
    __shared__ float tile[1024];
    const int head = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
    const long long fh = (long long)frame * a.heads + head;
    const float* p = a.part + fh * a.nsplit * SLA_PART;
    float* out = a.comb + fh * SLA_COMB;
    for (int i = tid; i < 1024; i += 256) {
        const int dd = i >> 5;
        float v;
        if (a.mode == 0) {
            float M = -1e30f;
            for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, p[s * SLA_PART + 1024 + dd]);
            float num = 0.f, den = 0.f;
            for (int s = 0; s < a.nsplit; ++s) {
                const float w = __expf(p[s * SLA_PART + 1024 + dd] - M);
                num += p[s * SLA_PART + i] * w;
                den += p[s * SLA_PART + 1056 + dd] * w;
            }
            v = num / den;
            if ((i & 31) == 0) { out[1024 + dd] = M; out[1056 + dd] = 1.0f / den; }
        } else {
            v = 0.f;
            for (int s = 0; s < a.nsplit; ++s) v += p[s * SLA_PART + i];
            tile[i] = v * a.comb_ctx[fh * SLA_COMB + i];
        }
        out[i] = v;
    }
    if (a.mode == 1) {
        __syncthreads();
        if (tid < 32) {
            float sacc = 0.f;
            for (int e = 0; e < 32; ++e) sacc += tile[tid * 32 + e];
            out[1024 + tid] = sacc;
            out[1056 + tid] = 0.f;
        }
    }
}

struct SlaRowsArgs {
    const __bf16* qkv;        // [rows][3*heads*32]
    const __bf16* qt;         // bwd: saved Qt rows [rows][heads*32]
    const __bf16* dout;       // bwd: gradient of out [rows][heads*32]
    const float* comb_ctx;    // combined records of the softmax reduction (k, v)
    const float* comb_dctx;   // bwd: combined records of the plain reduction (Qt, dout)
    __bf16* out;              // fwd: out rows [rows][heads*32]; bwd: dqkv rows [rows][3*heads*32]
    __bf16* qt_out;           // fwd: Qt rows
    int n, heads, frames, blocks_per_wg;
    float scale;
};

template <bool BWD>
__global__ __launch_bounds__(256) void sla_rows_kernel(const SlaRowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = a.heads, tid = threadIdx.x;
    float* c0 = reinterpret_cast<float*>(smem);           // [H][SLA_COMB]: ctx, max, 1 / weight sum
    float* c1 = c0 + H * SLA_COMB;                        // bwd: [H][SLA_COMB]: dctx, S
    const int frame = blockIdx.y;
    for (int i = tid; i < H * SLA_COMB; i += 256) {
        c0[i] = a.comb_ctx[(long long)frame * H * SLA_COMB + i];
        if constexpr (BWD) c1[i] = a.comb_dctx[(long long)frame * H * SLA_COMB + i];
    }
    __syncthreads();
    // rows: thread = (head, row of the block); a wave's lanes share the head
    const int rb = 256 / H, h = tid / rb, r = tid - h * rb;
    if (h >= H) return;
    const int ld = 3 * H * 32, ldo = H * 32;
    const float* cx = c0 + h * SLA_COMB;
    for (int blk = 0; blk < a.blocks_per_wg; ++blk) {
        const int pos = (blockIdx.x * a.blocks_per_wg + blk) * rb + r;
        if (pos >= a.n) break;
        const long long row = (long long)frame * a.n + pos;
        if constexpr (!BWD) {
            float q[32];
            load32(a.qkv + row * ld + h * 32, q);
            float m = q[0];
#pragma unroll
            for (int dd = 1; dd < 32; ++dd) m = fmaxf(m, q[dd]);
            float l = 0.f;
#pragma unroll
            for (int dd = 0; dd < 32; ++dd) { q[dd] = __expf(q[dd] - m); l += q[dd]; }
            const float inv = a.scale / l;
            float o[32];
#pragma unroll
            for (int e = 0; e < 32; ++e) o[e] = 0.f;
#pragma unroll
            for (int dd = 0; dd < 32; ++dd) {
                q[dd] *= inv;
                lds_axpy32(cx + dd * 32, q[dd], o);
            }
            store32(a.out + row * ldo + h * 32, o);
            store32(a.qt_out + row * ldo + h * 32, q);
        } else {
            const float* dcx = c1 + h * SLA_COMB;
            float g[32], t[32], dq[32];
            load32(a.dout + row * ldo + h * 32, g);
            load32(a.qt + row * ldo + h * 32, t);               // Qt = scale * sigma
            float dot = 0.f;
#pragma unroll
            for (int dd = 0; dd < 32; ++dd) { dq[dd] = lds_dot32(cx + dd * 32, g); dot += t[dd] * dq[dd]; }
            const float inv_scale = 1.0f / a.scale;
#pragma unroll
            for (int dd = 0; dd < 32; ++dd) dq[dd] = t[dd] * (dq[dd] - dot * inv_scale);
            store32(a.out + row * ld + h * 32, dq);
            float k[32], v[32], dv[32];
            load32(a.qkv + row * ld + (H + h) * 32, k);
            load32(a.qkv + row * ld + (2 * H + h) * 32, v);
#pragma unroll
            for (int e = 0; e < 32; ++e) dv[e] = 0.f;
#pragma unroll
            for (int dd = 0; dd < 32; ++dd) {
                const float kt = __expf(k[dd] - cx[1024 + dd]) * cx[1056 + dd];
                const float dkt = lds_dot32(dcx + dd * 32, v);
                lds_axpy32(dcx + dd * 32, kt, dv);
                k[dd] = kt * (dkt - dcx[1024 + dd]);
            }
            store32(a.out + row * ld + (H + h) * 32, k);
            store32(a.out + row * ld + (2 * H + h) * 32, dv);
        }
    }
}

int sla_shape_check(int frames, int n, int heads) {
    if (frames <= 0 || n <= 0 || heads <= 0 || heads > 8 || 256 % heads) return VDMI_E_SHAPE;
    if ((long long)frames * n > 0x7fffffffLL || frames > 65535) return VDMI_E_SHAPE;
    return VDMI_OK;
}

}   // namespace

// sla_rows_mfma.hip: the matrix-core forms of the row kernels and of the sums over positions
bool vdmi_sla_rows_mfma_covers(int frames, int n, int heads);
int vdmi_sla_rows_mfma_launch(bool bwd, const void* qkv, const void* qt, const void* dout, const void* rec_ctx,
                              const void* rec_dctx, void* out, void* qt_out, int frames, int n, int heads, float scale,
                              hipStream_t stream);
int vdmi_sla_reduce_mfma_launch(bool softmax, const void* a, int lda, const void* b, int ldb, float* part, int frames, int n,
                                int heads, int nsplit, int split_len, hipStream_t stream);

extern "C" size_t vdmi_sla_train_partial_bytes(int frames, int n, int heads) {
    if (sla_shape_check(frames, n, heads) != VDMI_OK) return 0;
    const size_t nsplit = (size_t)(n + SLA_SPLIT - 1) / SLA_SPLIT;
    return (size_t)frames * heads * nsplit * SLA_PART * sizeof(float);
}

extern "C" size_t vdmi_sla_train_combined_bytes(int frames, int heads) {
    if (frames <= 0 || heads <= 0) return 0;
    return (size_t)frames * heads * SLA_COMB * sizeof(float);
}

// mode 0: softmax_n(a)-weighted sums of b (a = k columns, b = v columns) -> combined = ctx, the softmax's maximum and
//         1 / weight sum per feature;
// mode 1: plain sums of a[n][d] * b[n][e] (a = Qt rows, b = gradient of out) -> combined = dctx and S[d] = sum_e dctx*ctx
//         (combined_ctx = the mode-0 record).  `partials` is scratch.
extern "C" int vdmi_sla_train_reduce(const void* a, int lda, const void* b, int ldb, void* partials, const void* combined_ctx,
                                     void* combined, int frames, int n, int heads, int mode, vdmi_stream stream) {
    if (!a || !b || !partials || !combined || (mode == 1 && !combined_ctx)) return VDMI_E_ARG;
    const int st = sla_shape_check(frames, n, heads);
    if (st != VDMI_OK) return st;
    if (lda % 8 || ldb % 8 || lda < heads * 32 || ldb < heads * 32 || (mode != 0 && mode != 1)) return VDMI_E_SHAPE;
    if (!aligned16(a) || !aligned16(b) || !aligned16(partials)) return VDMI_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    SlaRedArgs r{static_cast<const __bf16*>(a), lda, static_cast<const __bf16*>(b), ldb, static_cast<float*>(partials), n, heads,
                 (n + SLA_SPLIT - 1) / SLA_SPLIT};
    if (vdmi_sla_rows_mfma_covers(frames, n, heads)) {
        const int st2 = vdmi_sla_reduce_mfma_launch(mode == 0, a, lda, b, ldb, r.part, frames, n, heads, r.nsplit, SLA_SPLIT, s);
        if (st2 != VDMI_OK) return st2;
    } else {
        const dim3 grid((unsigned)r.nsplit, (unsigned)heads, (unsigned)frames);
        if (mode == 0) hipLaunchKernelGGL(sla_reduce_kernel<true>, grid, dim3(256), 0, s, r);
        else hipLaunchKernelGGL(sla_reduce_kernel<false>, grid, dim3(256), 0, s, r);
        VDMI_LAUNCH_CHECK();
    }
    SlaCombArgs c{static_cast<const float*>(partials), static_cast<const float*>(combined_ctx), static_cast<float*>(combined),
                  heads, r.nsplit, mode};
    hipLaunchKernelGGL(sla_combine_kernel, dim3((unsigned)heads, (unsigned)frames), dim3(256), 0, s, c);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

static int sla_rows_launch(bool bwd, SlaRowsArgs& r, hipStream_t stream) {
    const int st = sla_shape_check(r.frames, r.n, r.heads);
    if (st != VDMI_OK) return st;
    if (vdmi_sla_rows_mfma_covers(r.frames, r.n, r.heads))
        return vdmi_sla_rows_mfma_launch(bwd, r.qkv, r.qt, r.dout, r.comb_ctx, r.comb_dctx, r.out, r.qt_out, r.frames, r.n,
                                         r.heads, r.scale, stream);
    const int rb = 256 / r.heads;
    const int blocks = (r.n + rb - 1) / rb;
    // each workgroup first copies its frame's combined records (4-9 KB per head) into LDS: several row blocks per workgroup
    // once the frame has enough of them to fill the device anyway
    r.blocks_per_wg = blocks * r.frames >= 4096 ? 8 : (blocks * r.frames >= 1024 ? 2 : 1);
    const size_t lds = (size_t)r.heads * SLA_COMB * 4 * (bwd ? 2 : 1);
    const dim3 grid((unsigned)((blocks + r.blocks_per_wg - 1) / r.blocks_per_wg), (unsigned)r.frames);
    static std::atomic<uint64_t> mask_f{0}, mask_b{0};
    if (bwd) {
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(sla_rows_kernel<true>), mask_b, 96 * 1024)) return VDMI_E_LAUNCH;
        hipLaunchKernelGGL(sla_rows_kernel<true>, grid, dim3(256), lds, stream, r);
    } else {
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(sla_rows_kernel<false>), mask_f, 96 * 1024)) return VDMI_E_LAUNCH;
        hipLaunchKernelGGL(sla_rows_kernel<false>, grid, dim3(256), lds, stream, r);
    }
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" int vdmi_sla_train_rows_fwd(const void* qkv, const void* combined_ctx, void* out, void* qt, int frames, int n,
                                       int heads, float scale, vdmi_stream stream) {
    if (!qkv || !combined_ctx || !out || !qt) return VDMI_E_ARG;
    if (!aligned16(qkv) || !aligned16(out) || !aligned16(qt)) return VDMI_E_ALIGN;
    SlaRowsArgs r{};
    r.qkv = static_cast<const __bf16*>(qkv);
    r.comb_ctx = static_cast<const float*>(combined_ctx);
    r.out = static_cast<__bf16*>(out);
    r.qt_out = static_cast<__bf16*>(qt);
    r.n = n; r.heads = heads; r.frames = frames; r.scale = scale;
    return sla_rows_launch(false, r, static_cast<hipStream_t>(stream));
}

extern "C" int vdmi_sla_train_rows_bwd(const void* qkv, const void* qt, const void* dout, const void* combined_ctx,
                                       const void* combined_dctx, void* dqkv, int frames, int n, int heads, float scale,
                                       vdmi_stream stream) {
    if (!qkv || !qt || !dout || !combined_ctx || !combined_dctx || !dqkv) return VDMI_E_ARG;
    if (!aligned16(qkv) || !aligned16(qt) || !aligned16(dout) || !aligned16(dqkv)) return VDMI_E_ALIGN;
    SlaRowsArgs r{};
    r.qkv = static_cast<const __bf16*>(qkv);
    r.qt = static_cast<const __bf16*>(qt);
    r.dout = static_cast<const __bf16*>(dout);
    r.comb_ctx = static_cast<const float*>(combined_ctx);
    r.comb_dctx = static_cast<const float*>(combined_dctx);
    r.out = static_cast<__bf16*>(dqkv);
    r.n = n; r.heads = heads; r.frames = frames; r.scale = scale;
    return sla_rows_launch(true, r, static_cast<hipStream_t>(stream));
}
