// Matrix-core form of the temporal attention core of the training path (reference
// video_diffusion_pytorch/video_diffusion_pytorch.py:546-612: q * scale, rotary, sim = q k^T + relative-position bias,
// optional focus-present mask, softmax over the frames of a site, attn v; the backward is what torch autograd derives
// for it at :1451).  Serves frame axes of up to 32: longer ones stay on the vector kernels of attn_bwd.hip.
//
// A tile is 32 tokens = g = 32 / f whole sites x f frames (token = slot * f + frame; rows of a site are hw rows apart in
// the qkv tensor); scores between tokens of different sites are masked.  A wave takes (tile, head) units one after the
// other, always of the same head, with everything in registers and a private piece of LDS - no barriers:
//
//   * q, k, v, dO, O rows arrive as "natural" fragments (lane = token, 8 + 8 features: a rotary pair sits in one lane),
//     q and k are rotated in registers, q also takes scale * log2 e (scores live in the log2 domain);
//   * products that contract over FEATURES take natural fragments on both sides:  S^T = K Q^T and dP^T = V dO^T with
//     lane = query (softmax statistics, delta and the log-sum-exp are lane-local), S = Q K^T and dP = dO V^T with
//     lane = key;
//   * products that contract over TOKENS take a [key x query] or [query x key] accumulator directly as their B operand
//     (its registers are 8 contraction values in accumulator-row order) and, as A operand, the transposed tile
//     (lane = feature) that the transposing LDS read ds_read_b64_tr_b16 returns from a [token][feature] image whose
//     rows are stored in that same order:  O^T = V^T P^T,  dQ^T = K^T dS^T,  dV^T = dO^T P,  dK^T = Q^T dS;
//   * results come out with lane = token, registers = 16 features (a rotary pair in one lane again): inverse rotation,
//     then through LDS to 16-byte stores with four lanes on a row's 64 bytes (stores of 8 bytes per row and instruction
//     cost the memory system a partial-line write each: as much time as all the loads of the kernel);
//   * the bias gradient dS is summed in registers over all units of the wave and written once as a [32 x 32] partial;
//     a second launch adds the partials of a head (and the g diagonal blocks) in a fixed order - no atomics.
#include "vdmi_common.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f2bf(a[8 * s + j]);
    return f;
}

typedef short s4v __attribute__((ext_vector_type(4)));

// A operand [feature l31][8 tokens in accumulator-row order] of a [token][32 features] LDS image (64-byte rows, row p
// holds token p with bits 2 and 3 swapped); `base` already carries the lane's offset, + 16 rows for the second k-step
__device__ __forceinline__ bf16x8 tr_fragment(const char* base) {
    typedef __attribute__((address_space(3))) s4v* lds_s4;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base + 4 * 64));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr int MC_WAVES = 8;

struct McArgs {
    const __bf16* qkv;      // [b*f*hw][3*heads*32]: q | k | v, column = head * 32 + feature
    const __bf16* o;        // bwd: [rows][heads*32]
    const __bf16* dout;     // bwd: gradient of o
    const float* lse;       // bwd: [rows][heads], natural log
    __bf16* out;            // fwd: o rows; bwd: dqkv rows
    float* lse_out;         // fwd
    float* dbias_part;      // bwd: [waves][32 queries][32 keys] or null
    const float* bias;      // [heads][f][f] or null
    const float* rot_cos;   // [f][16] or null
    const float* rot_sin;
    const unsigned char* focus;   // [b] or null
    int b, f, hw, heads, g;
    long long nsites, units;      // b * hw; tiles * heads
    float scale;
};

template <bool BWD>
__global__ __launch_bounds__(MC_WAVES * 64) void tattn_core_mfma_kernel(const McArgs a) {
    // per wave: transposition images (fwd: V; bwd: K, Q, dO) and, bwd, log-sum-exp and delta by token
    __shared__ __attribute__((aligned(16))) char img[MC_WAVES][BWD ? 3 : 1][32 * 64];
    // results on their way out: [token][80 B] (64 B of a head's features + padding), written as the accumulators hold them
    // (lane = token, 4 features per piece) and read back as 16-byte pieces, four lanes to a row
    __shared__ __attribute__((aligned(16))) char oimg[MC_WAVES][32 * 80];
    __shared__ __attribute__((aligned(16))) float stats[BWD ? MC_WAVES : 1][2][32];
    // bwd: the bias in both orientations, [orientation][register / 4][lane] float4 (registers go to the next unit's rows)
    __shared__ __attribute__((aligned(16))) f32x4 btab[BWD ? MC_WAVES : 1][2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = a.f, H = a.heads, ld = 3 * H * 32, ldo = H * 32;
    const int slot = l31 / F, fi = l31 - slot * F;
    const bool in_tile = slot < a.g;

    unsigned mask_same = 0, mask_self = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, half);
        if (in_tile && t / F == slot) mask_same |= 1u << r;
        if (in_tile && t == l31) mask_self |= 1u << r;
    }
    // rotary factors of this lane's frame: natural layout (feature 8 half + e pairs with + 16) and accumulator layout
    // (register r < 8 = feature (r&3) + 8 (r>>2) + 4 half, pairs with register r + 8)
    float rc[8], rs[8], rc2[8], rs2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bool on = a.rot_cos && in_tile;
        const int d1 = 8 * half + e, d2 = (e & 3) + 8 * (e >> 2) + 4 * half;
        rc[e] = on ? a.rot_cos[fi * 16 + d1] : 1.f;
        rs[e] = on ? a.rot_sin[fi * 16 + d1] : 0.f;
        rc2[e] = on ? a.rot_cos[fi * 16 + d2] : 1.f;
        rs2[e] = on ? a.rot_sin[fi * 16 + d2] : 0.f;
    }
    const long long wave_gid = (long long)blockIdx.x * MC_WAVES + wave, stride = (long long)gridDim.x * MC_WAVES;
    const int head = (int)(wave_gid % H);               // the same for every unit of this wave: stride % heads == 0
    // bias in the log2 domain: [key register][query lane] and, bwd, [query register][key lane]
    f32x16 bT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int fj = acc_row(r, half) % F;
        const bool on = a.bias && in_tile;
        bT[r] = on ? a.bias[((long long)head * F + fi) * F + fj] * LOG2E : 0.f;
        if constexpr (BWD) {
            const float bs = on ? a.bias[((long long)head * F + fj) * F + fi] * LOG2E : 0.f;
            reinterpret_cast<float*>(&btab[wave][0][r >> 2][lane])[r & 3] = bT[r];
            reinterpret_cast<float*>(&btab[wave][1][r >> 2][lane])[r & 3] = bs;
        }
    }
    f32x16 dbacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dbacc[r] = 0.f;

    const int p_row = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int wr_off = p_row * 64 + 16 * half;
    const int g16 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const int tr_off = (8 * (g16 >> 1) + tq) * 64 + (16 * (g16 & 1) + 4 * tp) * 2;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // the tokens whose rows this lane writes: 16 j + lane / 4, piece lane % 4
    int oslot[2], ofi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tok = 16 * j + (lane >> 2);
        oslot[j] = tok / F;
        ofi[j] = tok - oslot[j] * F;
    }

    auto rotate = [&](bf16x8& x0, bf16x8& x1, float mul) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float lo = (float)x0[e], hi = (float)x1[e];
            x0[e] = f2bf((lo * rc[e] - hi * rs[e]) * mul);
            x1[e] = f2bf((hi * rc[e] + lo * rs[e]) * mul);
        }
    };
    auto to_image = [&](char* im, const bf16x8& x0, const bf16x8& x1) {
        *reinterpret_cast<bf16x8*>(im + wr_off) = x0;
        *reinterpret_cast<bf16x8*>(im + wr_off + 32) = x1;
    };
    // gradient of a rotated row in accumulator layout -> gradient of the raw row, times mul
    auto unrotate = [&](f32x16& x, float mul) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float lo = x[r], hi = x[r + 8];
            x[r] = (lo * rc2[r] + hi * rs2[r]) * mul;
            x[r + 8] = (hi * rc2[r] - lo * rs2[r]) * mul;
        }
    };
    // x (lane = token, register r = feature acc_row(r, half)) -> 64-byte row segments at dst + row * ldd, through LDS
    auto store_rows = [&](const f32x16& x, long long tile, __bf16* dst, int ldd) {
        char* im = oimg[wave];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = f2bf(x[4 * rg + e]);
            *reinterpret_cast<bf16x4*>(im + l31 * 80 + 16 * rg + 8 * half) = o4;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(im + (16 * j + (lane >> 2)) * 80 + 16 * (lane & 3));
            const long long site = tile * a.g + oslot[j];
            if (oslot[j] < a.g && site < a.nsites) {
                const long long bb = site / a.hw, sp = site - bb * a.hw;
                *reinterpret_cast<bf16x8*>(dst + ((bb * F + ofi[j]) * a.hw + sp) * ldd + 8 * (lane & 3)) = v;
            }
        }
    };

    // One unit's rows, requested one unit ahead of their use (the loads of unit u + 1 are in flight during the products
    // of unit u; with only the wave's own loads outstanding a unit would cost a full memory round trip).
    struct Unit {
        bf16x8 q0, q1, k0, k1, v0, v1, g0, g1, o0, o1;
        float lse;
        long long row;
        bool ok, foc;
    };
    auto fetch = [&](long long u, Unit& n) {
        const long long tile = u / H;
        const long long site = tile * a.g + slot;
        n.ok = in_tile && u < a.units && site < a.nsites;
        const long long bb = n.ok ? site / a.hw : 0, sp = n.ok ? site - bb * a.hw : 0;
        n.row = (bb * F + fi) * a.hw + sp;
        n.foc = a.focus && n.ok && a.focus[bb] != 0;
        n.q0 = n.q1 = n.k0 = n.k1 = n.v0 = n.v1 = zero8;
        if constexpr (BWD) { n.g0 = n.g1 = n.o0 = n.o1 = zero8; n.lse = 0.f; }
#ifdef VDMI_MC_ABL_NOLOAD       // timing-only ablation (wrong results): no row is read
        if (n.ok && a.scale < -1.f) {
#else
        if (n.ok) {
#endif
#ifdef VDMI_MC_ABL_COALESCED    // timing-only ablation (wrong results): four lanes read the 64 contiguous bytes of a row
            long long rj[2];
            for (int j = 0; j < 2; ++j) {
                const int tok = 16 * j + (lane >> 2), sl = tok / F, fr = tok - sl * F;
                long long st2 = tile * a.g + sl;
                if (st2 >= a.nsites) st2 = a.nsites - 1;
                const long long b2 = st2 / a.hw, s2 = st2 - b2 * a.hw;
                rj[j] = (b2 * F + fr) * a.hw + s2;
            }
            const __bf16* p0 = a.qkv + rj[0] * ld + head * 32 + 8 * (lane & 3);
            const __bf16* p1 = a.qkv + rj[1] * ld + head * 32 + 8 * (lane & 3);
            n.q0 = *reinterpret_cast<const bf16x8*>(p0);
            n.q1 = *reinterpret_cast<const bf16x8*>(p1);
            n.k0 = *reinterpret_cast<const bf16x8*>(p0 + ldo);
            n.k1 = *reinterpret_cast<const bf16x8*>(p1 + ldo);
            n.v0 = *reinterpret_cast<const bf16x8*>(p0 + 2 * ldo);
            n.v1 = *reinterpret_cast<const bf16x8*>(p1 + 2 * ldo);
#else
            const __bf16* p = a.qkv + n.row * ld + head * 32 + 8 * half;
            n.q0 = *reinterpret_cast<const bf16x8*>(p);
            n.q1 = *reinterpret_cast<const bf16x8*>(p + 16);
            n.k0 = *reinterpret_cast<const bf16x8*>(p + ldo);
            n.k1 = *reinterpret_cast<const bf16x8*>(p + ldo + 16);
            n.v0 = *reinterpret_cast<const bf16x8*>(p + 2 * ldo);
            n.v1 = *reinterpret_cast<const bf16x8*>(p + 2 * ldo + 16);
#endif
            if constexpr (BWD) {
                const __bf16* gp = a.dout + n.row * ldo + head * 32 + 8 * half;
                const __bf16* op = a.o + n.row * ldo + head * 32 + 8 * half;
                n.g0 = *reinterpret_cast<const bf16x8*>(gp);
                n.g1 = *reinterpret_cast<const bf16x8*>(gp + 16);
                n.o0 = *reinterpret_cast<const bf16x8*>(op);
                n.o1 = *reinterpret_cast<const bf16x8*>(op + 16);
                n.lse = a.lse[n.row * H + head];
            }
        }
    };

    Unit nx;
    fetch(wave_gid, nx);
    for (long long u = wave_gid; u < a.units; u += stride) {
        const Unit c = nx;
        fetch(u + stride, nx);
#ifdef VDMI_MC_ABL_NOSTORE      // timing-only ablation: nothing is written
        const bool ok = c.ok && a.scale < -1.f, foc = c.foc;
#else
        const bool ok = c.ok, foc = c.foc;
#endif
        const long long row = c.row;
        const unsigned mask = ok ? (foc ? mask_self : mask_same) : 0u;
        bf16x8 q0 = c.q0, q1 = c.q1, k0 = c.k0, k1 = c.k1;
        const bf16x8 v0 = c.v0, v1 = c.v1;
        rotate(q0, q1, a.scale * LOG2E);
        rotate(k0, k1, 1.f);
        if constexpr (!BWD) {
            to_image(img[wave][0], v0, v1);
            f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, bT, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, st, 0, 0, 0);
            float m = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = ((mask >> r) & 1u) ? st[r] : -1e30f;
                m = fmaxf(m, st[r]);
            }
            m = xhalf_max(m);
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - m);
                l += st[r];
            }
            l = xhalf_add(l);
            const char* iv = img[wave][0] + tr_off;
            const bf16x8 vt0 = tr_fragment(iv), vt1 = tr_fragment(iv + 16 * 64);
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt0, acc_frag(st, 0), o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt1, acc_frag(st, 1), o, 0, 0, 0);
            const float inv = 1.0f / l;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= inv;
            store_rows(o, u / H, a.out + head * 32, ldo);
            if (ok && half == 0) a.lse_out[row * H + head] = (m + __builtin_amdgcn_logf(l)) * LN2;
        } else {
            const bf16x8 g0 = c.g0, g1 = c.g1;
            float delta = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += (float)g0[e] * (float)c.o0[e] + (float)g1[e] * (float)c.o1[e];
            delta = xhalf_add(delta);
            const float lse2 = c.lse * LOG2E;
            to_image(img[wave][0], k0, k1);
            to_image(img[wave][1], q0, q1);
            to_image(img[wave][2], g0, g1);
            if (half == 0) {
                stats[wave][0][l31] = lse2;
                stats[wave][1][l31] = delta;
            }
            // ---- lane = query:  P^T, dS^T = P^T (dP^T - delta), dq^T = K^T dS^T
            {
                f32x16 seed, negd;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 b4 = btab[wave][0][rg][lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { seed[4 * rg + e] = b4[e] - lse2; negd[4 * rg + e] = -delta; }
                }
                f32x16 st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, seed, 0, 0, 0);
                f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, g0, negd, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, g1, dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // a focused sample's only unmasked key makes the softmax constant: its score gradient is an exact zero
                    const float p = (((mask >> r) & 1u) && !foc) ? __builtin_amdgcn_exp2f(st[r]) : 0.f;
                    st[r] = p * dp[r];
                    dbacc[r] += st[r];
                }
                const char* ik = img[wave][0] + tr_off;
                const bf16x8 kt0 = tr_fragment(ik), kt1 = tr_fragment(ik + 16 * 64);
                f32x16 dq;
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[r] = 0.f;
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt0, acc_frag(st, 0), dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt1, acc_frag(st, 1), dq, 0, 0, 0);
                unrotate(dq, a.scale);
                store_rows(dq, u / H, a.out + head * 32, ld);
            }
            // ---- lane = key:  P, dS = P (dP - delta), dv^T = dO^T P, dk^T = Q^T dS
            {
                f32x16 seed, negd;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 b4 = btab[wave][1][rg][lane];
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(&stats[wave][0][8 * rg + 4 * half]);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(&stats[wave][1][8 * rg + 4 * half]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { seed[4 * rg + e] = b4[e] - l4[e]; negd[4 * rg + e] = -d4[e]; }
                }
                f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0, k0, seed, 0, 0, 0);
                f32x16 dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, v0, negd, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1, k1, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, v1, dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = ((mask >> r) & 1u) ? __builtin_amdgcn_exp2f(s[r]) : 0.f;
                    s[r] = p;
                    dp[r] = foc ? 0.f : p * dp[r];
                }
                const char* iq = img[wave][1] + tr_off;
                const char* ig = img[wave][2] + tr_off;
                const bf16x8 gt0 = tr_fragment(ig), gt1 = tr_fragment(ig + 16 * 64);
                const bf16x8 qt0 = tr_fragment(iq), qt1 = tr_fragment(iq + 16 * 64);
                f32x16 dk, dv;
#pragma unroll
                for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.f;
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt0, acc_frag(s, 0), dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt0, acc_frag(dp, 0), dk, 0, 0, 0);
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt1, acc_frag(s, 1), dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt1, acc_frag(dp, 1), dk, 0, 0, 0);
                unrotate(dk, LN2);
                store_rows(dk, u / H, a.out + (H + head) * 32, ld);
                store_rows(dv, u / H, a.out + (2 * H + head) * 32, ld);
            }
        }
    }
    if constexpr (BWD) {
        if (a.dbias_part) {
            float* dst = a.dbias_part + wave_gid * 1024 + l31 * 32 + 4 * half;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *reinterpret_cast<f32x4*>(dst + 8 * rg) = f32x4{dbacc[4 * rg], dbacc[4 * rg + 1], dbacc[4 * rg + 2], dbacc[4 * rg + 3]};
        }
    }
}

// dbias[h][i][j] += sum over the waves of head h (wave % heads == h) and the g diagonal blocks of their partials
// [query token][key token]; one workgroup per (h, i), eight interleaved chains of waves added in a fixed order
__global__ __launch_bounds__(256) void tattn_core_dbias_kernel(const float* __restrict__ part, float* __restrict__ dbias,
                                                               int f, int heads, int g, long long nwaves) {
    __shared__ float red[8][32];
    const int h = blockIdx.x / f, i = blockIdx.x - h * f;
    const int j = threadIdx.x & 31, chain = threadIdx.x >> 5;
    float s = 0.f;
    if (j < f) {
        const long long step = (long long)heads * 8;
#pragma unroll 8
        for (long long w = h + (long long)heads * chain; w < nwaves; w += step) {
            const float* p = part + w * 1024;
            for (int sl = 0; sl < g; ++sl) s += p[(sl * f + i) * 32 + sl * f + j];
        }
    }
    red[chain][j] = s;
    __syncthreads();
    if (chain == 0 && j < f) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) t += red[c][j];
        dbias[((long long)h * f + i) * f + j] += t;
    }
}

int g_mc_enabled = -1;

// workgroups of a launch: one wave per (tile, head) unit up to what the chip holds at once, a multiple of `heads` so
// that a wave's units all have one head
long long mc_grid(bool bwd, int b, int f, int hw, int heads) {
    const int g = 32 / f;
    const long long nsites = (long long)b * hw, tiles = (nsites + g - 1) / g, units = tiles * heads;
    long long grid = (units + MC_WAVES - 1) / MC_WAVES;
    const long long cap = bwd ? 256 : 512;              // workgroups the chip holds at once (registers: one / two per CU)
    if (grid > cap) grid = cap;
    grid = (grid + heads - 1) / heads * heads;
    return grid;
}

}   // namespace

extern "C" void vdmi_tattn_core_mfma_enable(int on) { g_mc_enabled = on ? 1 : 0; }

// Whether the matrix-core form serves this shape (frame axes of up to 32, rows addressable with 32 bits of elements).
bool vdmi_tattn_core_mfma_covers(int b, int f, int hw, int heads) {
    if (g_mc_enabled < 0) {
        const char* e = std::getenv("VDMI_TATTN_CORE_MFMA");
        g_mc_enabled = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_mc_enabled) return false;
    if (b <= 0 || f <= 0 || f > 32 || hw <= 0 || heads <= 0 || heads > 64) return false;
    return (long long)b * f * hw <= 0x7fffffffLL / (3LL * heads * 32);
}

size_t vdmi_tattn_core_mfma_workspace(int b, int f, int hw, int heads) {
    return (size_t)mc_grid(true, b, f, hw, heads) * MC_WAVES * 1024 * sizeof(float);
}

// fwd: o, lse_out set; bwd: o, dout, lse, dqkv set, dbias with its workspace or neither
int vdmi_tattn_core_mfma_launch(bool bwd, const void* qkv, const void* o, const void* dout, const float* lse, void* out,
                                float* lse_out, float* dbias, void* workspace, const float* bias, const float* rot_cos,
                                const float* rot_sin, const unsigned char* focus, int b, int f, int hw, int heads,
                                float scale, hipStream_t stream) {
    McArgs a{};
    a.qkv = static_cast<const __bf16*>(qkv);
    a.o = static_cast<const __bf16*>(o);
    a.dout = static_cast<const __bf16*>(dout);
    a.lse = lse;
    a.out = static_cast<__bf16*>(out);
    a.lse_out = lse_out;
    a.dbias_part = dbias ? static_cast<float*>(workspace) : nullptr;
    a.bias = bias; a.rot_cos = rot_cos; a.rot_sin = rot_sin; a.focus = focus;
    a.b = b; a.f = f; a.hw = hw; a.heads = heads; a.g = 32 / f;
    a.nsites = (long long)b * hw;
    a.units = (a.nsites + a.g - 1) / a.g * heads;
    a.scale = scale;
    const long long grid = mc_grid(bwd, b, f, hw, heads);
    if (bwd) {
        hipLaunchKernelGGL(tattn_core_mfma_kernel<true>, dim3((unsigned)grid), dim3(MC_WAVES * 64), 0, stream, a);
        if (dbias)
            hipLaunchKernelGGL(tattn_core_dbias_kernel, dim3((unsigned)(heads * f)), dim3(256), 0, stream, a.dbias_part, dbias,
                               f, heads, a.g, grid * MC_WAVES);
    } else {
        hipLaunchKernelGGL(tattn_core_mfma_kernel<false>, dim3((unsigned)grid), dim3(MC_WAVES * 64), 0, stream, a);
    }
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
