// Implicit-GEMM "tap" convolution for gfx950 (MI355X) with fused epilogues.
//
// Orientation: the MFMA A operand is the weight matrix (rows = output channels), the B operand
// is the pixel tile (columns = pixels).  The 32x32 accumulator of v_mfma_f32_32x32x16_bf16 then
// holds ONE pixel per lane (col = lane & 31) and 16 output channels in its registers
// (row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)), so every per-pixel reduction over channels
// (RMSNorm, softmax over a head, rotary pairs d / d+16) is in-lane plus one cross-half shuffle.
//
// Data movement per workgroup:
//   * the input tile with its halo ("patch", all taps read it) is staged ONCE per channel chunk
//     in LDS, rows padded by 16 B so the per-lane 16-B fragment reads spread over the banks;
//   * weights are pre-packed on the host in MFMA fragment order, so a (chunk, tap) weight tile is
//     one contiguous block: it is prefetched global->registers during the previous step's MFMAs
//     and written linearly to a double-buffered LDS tile (conflict-free ds_read_b128 by
//     construction: lane l reads bytes [16 l, 16 l + 16) of a 1-KiB fragment).
//   * one __syncthreads() per (chunk, tap) step.
#include "vdmi_common.h"
#include <cstring>

// Diagnostic builds only (-DVDMI_DIAG_STAMPS, never shipped): thread 0 of every workgroup records
// s_memtime at a few points into a buffer whose address comes from the VDMI_DIAG_PTR environment
// variable; nothing else reads that buffer.
#ifdef VDMI_DIAG_STAMPS
#include <cstdlib>
#define VDMI_STAMP(ptr, slot)                                                                         \
    do {                                                                                              \
        if ((ptr) && threadIdx.x == 0)                                                                \
            (ptr)[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = \
                __builtin_readcyclecounter();                                                         \
    } while (0)
#define VDMI_STAMP_T() __builtin_readcyclecounter()
#define VDMI_STAMP_ADD(ptr, slot, val)                                                                \
    do {                                                                                              \
        if ((ptr) && threadIdx.x == 0)                                                                \
            (ptr)[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] += (val); \
    } while (0)
#else
#define VDMI_STAMP(ptr, slot) do { } while (0)
#define VDMI_STAMP_T() 0ull
#define VDMI_STAMP_ADD(ptr, slot, val) do { } while (0)
#endif

// steps ahead a weight tile is requested in the deep-prefetch kernels (register sets = this + 1; 1: one set,
// refilled in place): plain epilogue / RMSNorm epilogue
#ifndef VDMI_WPD
#define VDMI_WPD 3
#endif
#ifndef VDMI_WPD_NORM
#define VDMI_WPD_NORM 1
#endif
// the flat row GEMMs (1-tap passes, QKV projections): weight tiles AND row chunks this many steps ahead
#ifndef VDMI_WPD_ROWS
#define VDMI_WPD_ROWS 2
#endif

namespace {

struct PassDev {
    const __bf16* src0;
    const __bf16* src1;
    int c0, c1;
    int nch0, nch;
    int in_h, in_w, stride;
    int ntaps;
    int min_dy, min_dx;
    int ph, pw;
    int toff[VDMI_MAX_TAPS];   // LDS offset of each tap inside the patch, in elements (line skew included)
    int lskew;                 // extra elements between consecutive patch lines (a multiple of 8; see pick_lskew)
    const __bf16* w;
    long long w_group_stride;
    int w_group_frames;
    long long wtile_all;
    // frame stacking: a tile covers `fstack` whole frames of fs_rows tile rows each (images smaller than
    // the pixel tile); every frame has its own haloed band of `phf` patch rows.  fstack == 1: off.
    int fstack, fs_rows, phf, nframes;
    float inv_pw, inv_phf;   // 1/pw, 1/phf: row -> (y, x) of a patch row without integer division
    // phases (main pass of a PLAIN launch): the taps are `phases` equal groups, each an independent
    // convolution with its own weight set and output parity; blockIdx.z enumerates (frame group, phase)
    int phases;
    long long w_phase_stride;
    // space-to-depth view of src0 (see vdmi_tap_pass): in_h / in_w are the view's extent, sh / sw the source
    // image, csub the source's channels; chunk c reads sub-pixel c / (csub / chunk)
    int s2d, sh, sw, csub;
};

struct KArgs {
    PassDev main, res;
    int has_res;
    const __bf16* res_identity;
    int ld_res;
    const float* res_bias;
    int frames_per_batch;
    int grid_h, grid_w, tiles_x, ltw;
    int out_mul, out_off_y, out_off_x, out_h, out_w;
    __bf16* out;
    int ldo, n;
    const float* bias;
    const float* gain;
    const float* shift;
    long long gain_bstride, shift_bstride;
    float norm_scale;
    const float* colsum;
    float ln_eps, q_scale;
    const float* rot_cos;
    const float* rot_sin;
    int rot_frames, rot_hw;
    __bf16* out2;
    int sla_hw;
    int heads32;
    int act;
    int ksplit;
    long long out_split_stride;
    unsigned long long* diag;
};

// pieces (16 B) of the haloed patch one thread may hold in registers while they are in flight
constexpr int patch_lb(int bm, int nt) {
    return bm >= 256 ? (13 * 256 + nt - 1) / nt : (bm >= 64 ? (12 * 256 + nt - 1) / nt : 2);
}

// ROWS: the pass is a flat row GEMM (one tap, tiles never leave a row range): the loader takes whole rows of a
// channel chunk, and because every step is a new chunk it keeps a ring of them in flight (the same depth as the
// weight ring), not just the next one.  STATS (the LayerNorm-folding QKV epilogues) implies ROWS.
template <int BM, int BN, int WM, int WN, int BKC, bool STATS, bool ROWS, int WPD, class Pro>
__device__ __forceinline__ void run_pass(const PassDev& g, f32x16 (&acc)[WN / 32][WM / 32], __bf16* wbuf,
                                         __bf16* patch, float* stats, int frame, int ty, int tx, int ltw,
                                         int nt, int tid, float ln_eps, Pro&& after_first_loads,
                                         unsigned long long* diag = nullptr, int phase = 0, int c_begin = 0,
                                         int c_end = -1) {
    constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
    constexpr int MI = WN / 32, NJ = WM / 32, KS = BKC / 16, PROW = BKC + 8, PPR = BKC / 8;
    constexpr int WP = BN * BKC / 8 / NT;
    static_assert((BN * BKC / 8) % NT == 0, "weight tile must split evenly over the workgroup");
    constexpr int KMAX = (BM * PPR + NT - 1) / NT;   // patch pieces per thread in 1-tap mode

    const int lane = tid & 63, wave = tid >> 6, wm = wave % NWM, wn = wave / NWM;
    const int TW = 1 << ltw, TH = BM >> ltw;

    int rowoff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = (wm * NJ + j) * 32 + (lane & 31);
        const int py = p >> ltw, px = p & (TW - 1);
        int prow = py * g.stride;
        if (g.fstack > 1) {
            const int fi = min(py / g.fs_rows, g.fstack - 1);
            prow = fi * g.phf + (py - fi * g.fs_rows) * g.stride;
        }
        rowoff[j] = (prow * g.pw + px * g.stride) * PROW + prow * g.lskew + (lane >> 5) * 8;
    }
    const int iy0 = ty * TH * g.stride + g.min_dy, ix0 = tx * TW * g.stride + g.min_dx;
    const __bf16* wg = g.w + (long long)nt * BN * BKC;
    if (g.w_group_stride) wg += (long long)(frame / g.w_group_frames) * g.w_group_stride;
    const int ntp = g.phases > 1 ? g.ntaps / g.phases : g.ntaps, tap0 = phase * ntp;   // this phase's taps
    wg += phase * g.w_phase_stride;

    // weight tiles travel global -> registers -> LDS.  WPD > 1: WPD + 1 register sets, so the tile of step s + WPD
    // is requested while step s computes (one step of MFMAs is shorter than an L2 round trip when few workgroups
    // share a CU); WPD == 1: one set, refilled for step s + 1 as soon as step s's tile is in LDS.
    constexpr bool DEEP = WPD > 1;
    constexpr int PD = WPD, RING = DEEP ? PD + 1 : 1;
    bf16x8 wreg[RING][WP];
    auto load_w = [&](bf16x8 (&dst)[WP], int s) {
        const __bf16* p = wg + (long long)s * g.wtile_all;
#pragma unroll
        for (int k = 0; k < WP; ++k) dst[k] = *reinterpret_cast<const bf16x8*>(p + (tid + k * NT) * 8);
    };

    float s1[KMAX], s2[KMAX];
    if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s1[k] = s2[k] = 0.f;
    }

    // this workgroup's share of the contraction: channel chunks [c_begin, c_end) (all of them unless K is split)
    if (c_end < 0) c_end = g.nch;
    wg += (long long)c_begin * ntp * g.wtile_all;
    const int nsteps = (c_end - c_begin) * ntp;
    const int total = g.ph * g.pw * PPR;
    load_w(wreg[0], 0);
    if constexpr (DEEP) {
#pragma unroll
        for (int u = 1; u < PD; ++u)
            if (u < nsteps) load_w(wreg[u], u);
    }

    // ---- patch staging, software pipelined: the loads of chunk c+1 are in flight during chunk c's MFMAs
    constexpr int LB = patch_lb(BM, NT);
    constexpr int RSTEP = NT / PPR;
    static_assert(NT % PPR == 0, "row step must be whole");
    const int piece = tid % PPR, r0 = tid / PPR, nrows = g.ph * g.pw;
    static_assert(ROWS || !STATS, "the statistics ride on the row loader");
    bf16x8 pv[ROWS ? 1 : LB];
    // source pixel of each patch piece this thread stages (-1: padding) -- the same for every chunk
    int ppix[ROWS ? 1 : LB];
    // ROWS: ring of row-chunk register sets, slot = step % RING (the step loop is unrolled by RING)
    bf16x8 sv[ROWS ? RING : 1][KMAX];
    auto rows_load = [&](int c, bf16x8 (&dst)[KMAX]) {
        const __bf16* src;
        int cs, cbase;
        if (c < g.nch0) { src = g.src0; cs = g.c0; cbase = c * BKC; }
        else            { src = g.src1; cs = g.c1; cbase = (c - g.nch0) * BKC; }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int idx = tid + k * NT;
            const int r = idx / PPR, pc = idx % PPR;
            const int ix = ix0 + r, ch = cbase + pc * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (idx < total && ix < g.in_w && ch < cs)
                v = *reinterpret_cast<const bf16x8*>(src + ((long long)frame * g.in_h * g.in_w + ix) * cs + ch);
            dst[k] = v;
        }
    };
    if constexpr (!ROWS) {
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            // patch row -> (ry, rx): (r + 0.5) / pw is never within 0.5 / pw of an integer, far above fp32 rounding here
            const int r = r0 + u * RSTEP;
            const int ry = (int)(((float)r + 0.5f) * g.inv_pw), rx = r - ry * g.pw;
            int iy = iy0 + ry, fr = frame;
            const int ix = ix0 + rx;
            bool ok = r < nrows;
            if (g.fstack > 1) {
                const int fi = (int)(((float)ry + 0.5f) * g.inv_phf);
                iy = g.min_dy + ry - fi * g.phf;
                fr = frame + fi;
                ok = ok && fr < g.nframes;
            }
            ok = ok && iy >= 0 && iy < g.in_h && ix >= 0 && ix < g.in_w;
            if (g.s2d) {
                // source pixel (2iy, 2ix) in the low 27 bits; bits 27..30: rows 2iy-1, 2iy and columns 2ix-1, 2ix exist
                const int sy = 2 * iy, sx = 2 * ix;
                const int m = (sy - 1 >= 0 && sy - 1 < g.sh ? 1 : 0) | (sy < g.sh ? 2 : 0) |
                              (sx - 1 >= 0 && sx - 1 < g.sw ? 4 : 0) | (sx < g.sw ? 8 : 0);
                ppix[u] = ok ? (((fr * g.sh + sy) * g.sw + sx) | (m << 27)) : 0;
            } else {
                ppix[u] = ok ? (fr * g.in_h + iy) * g.in_w + ix : -1;
            }
        }
    }
    auto patch_prefetch = [&](int c) {
        if constexpr (!ROWS) {
            const __bf16* src;
            int cs, cbase;
            if (c < g.nch0) { src = g.src0; cs = g.c0; cbase = c * BKC; }
            else            { src = g.src1; cs = g.c1; cbase = (c - g.nch0) * BKC; }
            const int ch = cbase + piece * 8;
            const bool chok = ch < cs;
            if (g.s2d) {
                const int cps = g.csub / BKC, sub = c / cps, sa = sub >> 1, sb = sub & 1;
                const int chs = (c - sub * cps) * BKC + piece * 8;
                const int need = ((1 << sa) | (4 << sb)) << 27, back = (1 - sa) * g.sw + (1 - sb);
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    pv[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    if ((ppix[u] & need) == need)
                        pv[u] = *reinterpret_cast<const bf16x8*>(
                            g.src0 + (long long)((ppix[u] & 0x7ffffff) - back) * g.csub + chs);
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                pv[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (chok && ppix[u] >= 0)
                    pv[u] = *reinterpret_cast<const bf16x8*>(src + (long long)ppix[u] * cs + ch);
            }
        }
    };
    auto patch_commit = [&]() {
        if constexpr (!ROWS) {
            if (g.lskew) {
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int r = r0 + u * RSTEP;
                    const int ry = (int)(((float)r + 0.5f) * g.inv_pw);
                    if (r < nrows) *reinterpret_cast<bf16x8*>(patch + r * PROW + ry * g.lskew + piece * 8) = pv[u];
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < LB; ++u)
                if (r0 + u * RSTEP < nrows) *reinterpret_cast<bf16x8*>(patch + (r0 + u * RSTEP) * PROW + piece * 8) = pv[u];
        }
    };
    if constexpr (ROWS) {
#pragma unroll
        for (int u = 0; u < RING; ++u) rows_load(min(c_begin + u, c_end - 1), sv[u]);
    } else {
        patch_prefetch(c_begin);
    }
    after_first_loads();   // per-channel parameters, accumulator init: overlaps the loads above

    int c = c_begin, t = 0;   // chunk and tap of the current step
    auto step = [&](int s, bf16x8 (&cur)[WP], bf16x8 (&nxt)[WP], bf16x8 (&rows)[KMAX]) {
        if (t == 0) {
            __syncthreads();   // every wave is done with the previous chunk's patch
            if constexpr (ROWS) {
                // this chunk's rows: registers -> LDS (+ LayerNorm sums), then the set is refilled RING chunks ahead
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int idx = tid + k * NT;
                    if (idx < total) {
                        const int r = idx / PPR, pc = idx % PPR;
                        const bf16x8 v = rows[k];
                        *reinterpret_cast<bf16x8*>(patch + r * PROW + pc * 8) = v;
                        if constexpr (STATS) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1[k] += f; s2[k] += f * f; }
                        }
                    }
                }
                rows_load(min(c + RING, c_end - 1), rows);
            } else {
                patch_commit();
                if (c + 1 < c_end) patch_prefetch(c + 1);
            }
        }
        __bf16* wb = wbuf + (s & 1) * BN * BKC;
        [[maybe_unused]] const unsigned long long ts0 = VDMI_STAMP_T();
#pragma unroll
        for (int k = 0; k < WP; ++k) *reinterpret_cast<bf16x8*>(wb + (tid + k * NT) * 8) = cur[k];
        [[maybe_unused]] const unsigned long long ts1 = VDMI_STAMP_T();
#if !defined(VDMI_ABL) || VDMI_ABL != 4
        __syncthreads();   // patch and this step's weight tile are visible
#else
        if (s == 0) __syncthreads();
#endif
        [[maybe_unused]] const unsigned long long ts2 = VDMI_STAMP_T();
        VDMI_STAMP_ADD(diag, 5, ts1 - ts0);
        VDMI_STAMP_ADD(diag, 6, ts2 - ts1);
        if (s == 0) VDMI_STAMP(diag, 1);
#if !defined(VDMI_ABL) || VDMI_ABL != 3
        if (s + PD < nsteps) load_w(nxt, s + PD);
#endif
        const int to = g.toff[tap0 + t];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[MI], bfr[NJ];
#if !defined(VDMI_ABL) || VDMI_ABL != 2
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(wb + (((wn * MI + i) * KS + ks) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(patch + rowoff[j] + to + ks * 16);
#else
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = cur[0];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = cur[0];
#endif
#if !defined(VDMI_ABL) || VDMI_ABL != 1
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
#else
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(bfr[j]));
#endif
        }
        VDMI_STAMP_ADD(diag, 7, VDMI_STAMP_T() - ts2);
        if (++t == ntp) { t = 0; ++c; }
    };
    // the ring of register sets has fixed names, so the step loop is unrolled by its depth
    if constexpr (DEEP) {
        for (int s = 0; s < nsteps; s += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u)
                if (s + u < nsteps) step(s + u, wreg[u], wreg[(u + PD) % RING], sv[ROWS ? u : 0]);
        }
    } else {
        for (int s = 0; s < nsteps; ++s) step(s, wreg[0], wreg[0], sv[0]);
    }

    if constexpr (STATS) {
        // LayerNorm statistics of each row (biased variance): the PPR threads that loaded one row
        // are consecutive lanes.
        const int ctot = g.c0 + g.c1;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            float a = s1[k], b = s2[k];
            a = vdmi_group_sum<PPR>(a);
            b = vdmi_group_sum<PPR>(b);
            const int idx = tid + k * NT;
            if (idx < total && (idx % PPR) == 0) {
                const float mean = a / ctot;
                const float var = fmaxf(b / ctot - mean * mean, 0.f);
                stats[2 * (idx / PPR)] = mean;
                stats[2 * (idx / PPR) + 1] = rsqrtf(var + ln_eps);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BKC, int EPI, bool FLAT = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64)
__attribute__((amdgpu_waves_per_eu(((BM / WM) * (BN / WN) >= 8 && BM >= 128) ? 4 : 2))) void tapgemm_kernel(const KArgs a) {
    constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
    constexpr int MI = WN / 32, NJ = WM / 32;
    // deeper weight prefetch where the registers allow it: one 16-byte weight piece per thread and step
    constexpr bool STATS = (EPI == VDMI_EPI_QKV_ATTN || EPI == VDMI_EPI_QKV_SLA);
    constexpr bool ROWS = STATS || FLAT;
    constexpr int WPD_MAIN = ROWS ? VDMI_WPD_ROWS : BN * BKC / 8 != NT ? 1 : EPI == VDMI_EPI_PLAIN ? VDMI_WPD : EPI == VDMI_EPI_NORM_SILU ? VDMI_WPD_NORM : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* wbuf = reinterpret_cast<__bf16*>(smem);
    float* par = reinterpret_cast<float*>(smem + 2 * BN * BKC * 2);
    float* red = par + 3 * BN;
    float* stats = red + NWN * BM;
    __bf16* patch = reinterpret_cast<__bf16*>(stats + 2 * BM);

    // The kernel arguments live in the kernarg segment and reach SGPRs by scalar loads.  Left alone the compiler
    // loads each field where it is first needed, one dword at a time, and the tile set-up below becomes a chain of
    // ~25 dependent load-wait pairs (several thousand cycles before the first global load is even issued).  Asking
    // for the fields of the set-up in one place makes it issue them together (merged into wide loads) behind one wait:
    // -2 900 cycles of 8 800 before the first barrier of a row GEMM.  Only there: in the convolution kernels the
    // set-up is a small share of a workgroup's life and the extra live scalars cost more than they save.
    if constexpr (STATS || FLAT) {
    asm volatile("" ::"s"(a.tiles_x), "s"(a.ltw), "s"(a.grid_h), "s"(a.grid_w), "s"(a.ksplit), "s"(a.n), "s"(a.ldo),
                 "s"(a.out_mul), "s"(a.out_h), "s"(a.out_w), "s"(a.frames_per_batch), "s"(a.has_res),
                 "s"(a.main.phases), "s"(a.main.fstack), "s"(a.main.nframes), "s"(a.main.nch), "s"(a.main.nch0),
                 "s"(a.main.c0), "s"(a.main.c1), "s"(a.main.in_h), "s"(a.main.in_w), "s"(a.main.stride),
                 "s"(a.main.ntaps), "s"(a.main.min_dy), "s"(a.main.min_dx), "s"(a.main.ph), "s"(a.main.pw),
                 "s"(a.main.lskew), "s"(a.main.s2d));
    asm volatile("" ::"s"(a.main.src0), "s"(a.main.src1), "s"(a.main.w), "s"(a.main.wtile_all), "s"(a.main.w_group_stride),
                 "s"(a.main.w_phase_stride), "s"(a.out), "s"(a.bias), "s"(a.out_split_stride), "s"(a.main.inv_pw),
                 "s"(a.main.inv_phf), "s"(a.main.fs_rows), "s"(a.main.phf));
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave % NWM, wn = wave / NWM;
    const int h = lane >> 5;
    const int tile = blockIdx.x, ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int nph = a.main.phases > 1 ? a.main.phases : 1;
    // split K (PLAIN only): blockIdx.z also enumerates the `ksplit` shares of the channel chunks; share k writes its
    // partial sums to out + k * out_split_stride (the consumer adds them), share 0 carries the bias
    const int nks = a.ksplit > 1 ? a.ksplit : 1;
    const int ksp = blockIdx.z % nks, zz = blockIdx.z / nks;
    const int phase = zz % nph;
    const int nt = blockIdx.y, frame = (zz / nph) * a.main.fstack, n0 = nt * BN;
    const float* const bias_in = ksp == 0 ? a.bias : nullptr;
    const int ooy = a.out_off_y + phase / a.out_mul, oox = a.out_off_x + phase % a.out_mul;
    const int TW = 1 << a.ltw, TH = BM >> a.ltw;
    // output row of tile pixel p (-1: outside the image / past the last frame)
    auto pixel_orow = [&](int p) -> long long {
        int gy = ty * TH + (p >> a.ltw), fr = frame;
        const int gx = tx * TW + (p & (TW - 1));
        if (a.main.fstack > 1) {
            const int fi = gy / a.main.fs_rows;
            gy -= fi * a.main.fs_rows;
            fr += fi;
            if (fi >= a.main.fstack || fr >= a.main.nframes) return -1;
        }
        return (gy < a.grid_h && gx < a.grid_w)
                   ? ((long long)fr * a.out_h + gy * a.out_mul + ooy) * a.out_w + gx * a.out_mul + oox
                   : -1;
    };

    f32x16 acc[MI][NJ];
    // per-channel epilogue parameters -> LDS (visible after the barriers of the main loop) and
    // accumulator init; run_pass calls this right after issuing its first loads
    auto prologue = [&]() {
        for (int c = tid; c < BN; c += NT) {
            const int gc = n0 + c;
            const bool ok = gc < a.n;
            if constexpr (EPI == VDMI_EPI_NORM_SILU) {
                const int b = frame / a.frames_per_batch;
                par[c] = (ok && bias_in) ? bias_in[gc] : 0.f;
                par[BN + c] = ok ? a.gain[b * a.gain_bstride + gc] : 0.f;
                par[2 * BN + c] = (ok && a.shift) ? a.shift[b * a.shift_bstride + gc] : 0.f;
            } else if constexpr (EPI == VDMI_EPI_PLAIN) {
                par[c] = (ok && bias_in) ? bias_in[gc] : 0.f;
            } else {
                par[c] = ok ? a.colsum[gc] : 0.f;
            }
        }

        // accumulators start at the bias (NORM / PLAIN), so the epilogue has no separate bias pass
    #pragma unroll
        for (int i = 0; i < MI; ++i)
    #pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == VDMI_EPI_NORM_SILU || EPI == VDMI_EPI_PLAIN) {
                    const int gc = n0 + (wn * MI + i) * 32 + 8 * rg + 4 * h;
                    if (bias_in && gc < a.n) bv = *reinterpret_cast<const f32x4*>(bias_in + gc);
                }
    #pragma unroll
                for (int j = 0; j < NJ; ++j)
    #pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * rg + e] = bv[e];
            }


    };

    VDMI_STAMP(a.diag, 0);
    run_pass<BM, BN, WM, WN, BKC, STATS, ROWS, WPD_MAIN>(a.main, acc, wbuf, patch, stats, frame, ty, tx, a.ltw, nt, tid, a.ln_eps, prologue, a.diag, phase,
                                                   ksp * a.main.nch / nks, (ksp + 1) * a.main.nch / nks);
    VDMI_STAMP(a.diag, 2);
    if constexpr (STATS) __syncthreads();   // row statistics written by the loader threads

    // output row of each of this lane's pixels (-1: outside the image)
    long long orow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        orow[j] = pixel_orow((wm * NJ + j) * 32 + (lane & 31));
    }

    if constexpr (EPI == VDMI_EPI_NORM_SILU) {
        float ss[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ss[j] = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ss[j] += acc[i][j][r] * acc[i][j][r];
#pragma unroll
        for (int j = 0; j < NJ; ++j) ss[j] = xhalf_add(ss[j]);
        if constexpr (NWN > 1) {
            __syncthreads();
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) red[wn * BM + (wm * NJ + j) * 32 + lane] = ss[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NWN; ++w) t += red[w * BM + (wm * NJ + j) * 32 + (lane & 31)];
                ss[j] = t;
            }
        }
        float rinv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) rinv[j] = a.norm_scale / fmaxf(sqrtf(ss[j]), 1e-12f);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch = (wn * MI + i) * 32 + 8 * rg + 4 * h;
                const f32x4 gv = *reinterpret_cast<const f32x4*>(par + BN + ch);
                const f32x4 sv = *reinterpret_cast<const f32x4*>(par + 2 * BN + ch);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = fmaf(acc[i][j][4 * rg + e] * rinv[j], gv[e], sv[e]);
                        acc[i][j][4 * rg + e] = silu_f(y);
                    }
            }
        if (a.has_res == 1) {
            run_pass<BM, BN, WM, WN, BKC, false, false, 1>(a.res, acc, wbuf, patch, stats, frame, ty, tx, a.ltw, nt, tid, 0.f, [] {});
            if (a.res_bias) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int gc = n0 + (wn * MI + i) * 32 + 8 * rg + 4 * h;
                        if (gc < a.n) {
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.res_bias + gc);
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][j][4 * rg + e] += bv[e];
                        }
                    }
            }
        }
    } else if constexpr (EPI == VDMI_EPI_PLAIN) {
        // bias already in the accumulators
        if (a.act == 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = silu_f(acc[i][j][r]);
        }
    } else {
        // folded LayerNorm: y = rstd * (acc - mean * colsum)
        float mean[NJ], rstd[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = (wm * NJ + j) * 32 + (lane & 31);
            mean[j] = stats[2 * p];
            rstd[j] = stats[2 * p + 1];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int gc0 = n0 + (wn * MI + i) * 32;
            const int which = gc0 / a.heads32;   // 0: q, 1: k, 2: v   (uniform per 32-channel tile)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch = (wn * MI + i) * 32 + 8 * rg + 4 * h;
                const f32x4 cs = *reinterpret_cast<const f32x4*>(par + ch);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j][4 * rg + e] = rstd[j] * (acc[i][j][4 * rg + e] - mean[j] * cs[e]);
            }
            if constexpr (EPI == VDMI_EPI_QKV_ATTN) {
                if (which == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= a.q_scale;
                }
                if (which < 2 && a.rot_cos) {
                    // half-split rotary: feature d pairs with d + 16, i.e. register r with r + 8
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const long long row = orow[j] < 0 ? 0 : orow[j];
                        const int pos = (int)((row / a.rot_hw) % a.rot_frames);
#pragma unroll
                        for (int rg = 0; rg < 2; ++rg) {
                            const f32x4 cv = *reinterpret_cast<const f32x4*>(a.rot_cos + pos * 16 + 8 * rg + 4 * h);
                            const f32x4 sv = *reinterpret_cast<const f32x4*>(a.rot_sin + pos * 16 + 8 * rg + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x1 = acc[i][j][4 * rg + e], x2 = acc[i][j][4 * rg + e + 8];
                                acc[i][j][4 * rg + e] = x1 * cv[e] - x2 * sv[e];
                                acc[i][j][4 * rg + e + 8] = x2 * cv[e] + x1 * sv[e];
                            }
                        }
                    }
                }
            } else {   // VDMI_EPI_QKV_SLA
                if (which == 0) {
                    // q: softmax over the 32 features of the head (16 in this lane, 16 in lane ^ 32)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float m = acc[i][j][0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
                        m = xhalf_max(m);
                        float sum = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { const float e = __expf(acc[i][j][r] - m); acc[i][j][r] = e; sum += e; }
                        sum = xhalf_add(sum);
                        const float inv = a.q_scale / sum;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
                    }
                } else {
                    // k, v: transposed store [frame][channel][position]
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (orow[j] >= 0) {
                            const long long fr = orow[j] / a.sla_hw;
                            const int pos = (int)(orow[j] - fr * a.sla_hw);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int gc = gc0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                                if (gc < a.n)
                                    a.out2[(fr * (2 * a.heads32) + (gc - a.heads32)) * a.sla_hw + pos] = f2bf(acc[i][j][r]);
                            }
                        }
                    }
                }
            }
        }
    }

    VDMI_STAMP(a.diag, 3);
    // ---- store through LDS: each wave transposes its WM x WN tile so that 8 lanes write one pixel's
    // 128 contiguous bytes (16 B per lane, whole cache lines) instead of 32 lines x 16 B per instruction.
    constexpr int SROW = WN + 8;                 // staging row (elements), padded against bank conflicts
    constexpr int CPR = WN / 8;                  // 16-byte chunks per staged row
    constexpr int RPI = 64 / CPR;                // rows covered by one wave instruction
    static_assert(64 % CPR == 0 && WM % RPI == 0, "staging geometry");
    __syncthreads();                             // every wave is done with the patch / weight tiles
    __bf16* stg = reinterpret_cast<__bf16*>(smem) + wave * WM * SROW;
    const int srow0 = lane / CPR, schunk = lane % CPR;
    const int gcw = n0 + wn * WN + schunk * 8;   // first of this lane's 8 channels in the coalesced phases
    auto staged_orow = [&](int row) -> long long { return pixel_orow(wm * WM + row); };
    bool add_res = false;
    if constexpr (EPI == VDMI_EPI_PLAIN || EPI == VDMI_EPI_NORM_SILU) {
        add_res = a.has_res == 2;
        if (add_res) {
            // identity residual: fetched with the same coalesced pattern, then added in fp32 below
#pragma unroll
            for (int k = 0; k < WM / RPI; ++k) {
                const int row = k * RPI + srow0;
                const long long orw = staged_orow(row);
                bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (orw >= 0 && gcw < a.n) rv = *reinterpret_cast<const bf16x8*>(a.res_identity + orw * a.ld_res + gcw);
                *reinterpret_cast<bf16x8*>(stg + row * SROW + schunk * 8) = rv;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                __bf16* sp = stg + (j * 32 + (lane & 31)) * SROW + i * 32 + 8 * rg + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rg + e];
                if (add_res) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(sp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
                *reinterpret_cast<bf16x4*>(sp) = o;
            }
    bool chan_ok = gcw < a.n;
    if constexpr (EPI == VDMI_EPI_QKV_SLA) chan_ok = chan_ok && (gcw / a.heads32 == 0);   // k, v went to out2
#pragma unroll
    for (int k = 0; k < WM / RPI; ++k) {
        const int row = k * RPI + srow0;
        const long long orw = staged_orow(row);
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(stg + row * SROW + schunk * 8);
        if (orw >= 0 && chan_ok) *reinterpret_cast<bf16x8*>(a.out + ksp * a.out_split_stride + orw * a.ldo + gcw) = ov;
    }
    VDMI_STAMP(a.diag, 4);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct TileCfg { int bm, bn, wm, wn, bkc; };

int build_pass(const vdmi_tap_pass& p, int bkc, int npad, PassDev* d) {
    if (!p.src0 || !p.w) return VDMI_E_ARG;
    if (p.c0 <= 0 || p.c0 % 8 || p.c1 < 0 || p.c1 % 8 || (p.c1 && !p.src1)) return VDMI_E_SHAPE;
    if (p.ntaps < 1 || p.ntaps > VDMI_MAX_TAPS || p.stride < 1 || p.stride > 2) return VDMI_E_SHAPE;
    if (!aligned16(p.src0) || (p.src1 && !aligned16(p.src1)) || !aligned16(p.w)) return VDMI_E_ALIGN;
    d->src0 = static_cast<const __bf16*>(p.src0);
    d->src1 = static_cast<const __bf16*>(p.src1);
    d->c0 = p.c0;
    d->c1 = p.c1;
    d->nch0 = (p.c0 + bkc - 1) / bkc;
    d->nch = d->nch0 + (p.c1 + bkc - 1) / bkc;
    d->in_h = p.in_h;
    d->in_w = p.in_w;
    d->stride = p.stride;
    d->ntaps = p.ntaps;
    int mny = 127, mxy = -128, mnx = 127, mxx = -128;
    for (int t = 0; t < p.ntaps; ++t) {
        mny = p.dy[t] < mny ? p.dy[t] : mny;
        mxy = p.dy[t] > mxy ? p.dy[t] : mxy;
        mnx = p.dx[t] < mnx ? p.dx[t] : mnx;
        mxx = p.dx[t] > mxx ? p.dx[t] : mxx;
    }
    d->min_dy = mny;
    d->min_dx = mnx;
    d->ph = mxy - mny;   // completed with the tile extent by the caller
    d->pw = mxx - mnx;
    d->w = static_cast<const __bf16*>(p.w);
    d->w_group_stride = p.w_group_stride;
    d->w_group_frames = p.w_group_frames > 0 ? p.w_group_frames : 1;
    d->wtile_all = (long long)npad * bkc;
    d->s2d = 0;
    if (p.space_to_depth) {
        if (p.stride != 1 || p.c1 || p.c0 % 4 || (p.c0 / 4) % bkc || p.in_h < 1 || p.in_w < 1) return VDMI_E_SHAPE;
        d->s2d = 1;
        d->sh = p.in_h;
        d->sw = p.in_w;
        d->csub = p.c0 / 4;
        d->in_h = p.in_h / 2 + 1;
        d->in_w = p.in_w / 2 + 1;
    }
    d->phases = p.phases > 1 ? p.phases : 1;
    d->w_phase_stride = d->phases > 1 ? p.w_phase_stride : 0;
    if (d->phases > 1 && (p.ntaps % d->phases || p.w_phase_stride % 8)) return VDMI_E_SHAPE;
    return VDMI_OK;
}

void finish_pass(const vdmi_tap_pass& p, PassDev* d, int th, int tw, int bkc, int lskew) {
    const int span_y = d->ph, span_x = d->pw;
    d->ph = (th - 1) * p.stride + span_y + 1;
    d->pw = (tw - 1) * p.stride + span_x + 1;
    d->lskew = lskew;
    for (int t = 0; t < p.ntaps; ++t) {
        const int ly = p.dy[t] - d->min_dy;
        d->toff[t] = (ly * d->pw + (p.dx[t] - d->min_dx)) * (bkc + 8) + ly * lskew;
    }
}

// A wave reads its pixel fragments from the patch with ds_read_b128: lane l (pixel l & 31 of the wave's 32) takes 16
// bytes at row(l) * (bkc + 8) * 2 + 16 * (l >> 5).  The LDS serves such a read in four groups of 16 lanes, and the 16
// addresses of a group are conflict-free only when their 16-byte slots differ mod 16.  Consecutive patch rows do
// (the row pitch is an odd number of slots), but a tile narrower than 32 pixels puts several patch lines into one
// group, and with the halo the line pitch is not a multiple of 16 slots: every fragment read then takes 2-3x its
// cycles.  Skewing each patch line by a few 16-byte slots restores distinct slots; the best skew is found by
// enumerating the groups, among those whose extra bytes (`lines` patch lines) keep the workgroup within the LDS it
// would otherwise use per CU slot (two workgroups per CU when the unskewed layout allows that).  Returns the skew in
// elements per line.
int pick_lskew(int ltw, int pw, int stride, int bkc, int lines, size_t lds_unskewed) {
    if (bkc != 64 || (1 << ltw) >= 32) return 0;
    const size_t cap = lds_unskewed <= 80 * 1024 ? 80 * 1024 : 160 * 1024;
    if (lds_unskewed > cap) return 0;
    const size_t budget = cap - lds_unskewed;
    static const int group_of[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0,
                                     1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
    const int slots_per_row = (bkc + 8) / 8, tw = 1 << ltw;
    int best = 0, best_extra = 1 << 30;
    for (int skew = 0; skew < 16; ++skew) {
        if ((size_t)lines * skew * 16 > budget) break;
        int extra = 0;
        for (int g = 0; g < 2; ++g) {
            int cnt[16] = {0}, mx = 0;
            for (int l = 0; l < 32; ++l) {
                if (group_of[l] != g) continue;
                const int line = (l >> ltw) * stride, px = (l & (tw - 1)) * stride;
                const int slot = ((line * pw + px) * slots_per_row + line * skew) & 15;
                mx = ++cnt[slot] > mx ? cnt[slot] : mx;
            }
            extra += mx - 1;
        }
        if (extra < best_extra) { best_extra = extra; best = skew; }
    }
    return best * 8;
}

size_t lds_bytes(const TileCfg& c, int patch_rows, int skew_elems = 0) {
    const int nwn = c.bn / c.wn, nwaves = (c.bm / c.wm) * nwn;
    const size_t main_loop = (size_t)2 * c.bn * c.bkc * 2 + (size_t)3 * c.bn * 4 + (size_t)nwn * c.bm * 4 +
                             (size_t)2 * c.bm * 4 + (size_t)patch_rows * (c.bkc + 8) * 2 + (size_t)skew_elems * 2;
    const size_t staging = (size_t)nwaves * c.wm * (c.wn + 8) * 2;   // output transpose, reuses the same LDS
    return main_loop > staging ? main_loop : staging;
}

constexpr size_t kLdsMax = 160 * 1024;

// patch rows the loader can hold in registers while a chunk is in flight (patch_lb pieces per thread)
int patch_row_cap(const TileCfg& c) {
    const int nthreads = (c.bm / c.wm) * (c.bn / c.wn) * 64;
    return patch_lb(c.bm, nthreads) * nthreads / (c.bkc / 8);
}

template <int BM, int BN, int WM, int WN, int BKC, int EPI, bool FLAT = false>
int launch(const KArgs& k, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_mask{0};
    auto fn = tapgemm_kernel<BM, BN, WM, WN, BKC, EPI, FLAT>;
    if (vdmi_set_max_lds(reinterpret_cast<const void*>(fn), attr_mask, (int)kLdsMax)) return VDMI_E_LAUNCH;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    hipLaunchKernelGGL(fn, grid, dim3(NT), lds, stream, k);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

int pick_tiles(int epi, int n, int stride, TileCfg* cands, int* ncand, int tile_bn = 0) {
    int k = 0;
    if (epi == VDMI_EPI_NORM_SILU) {
        if (n <= 64) cands[k++] = {256, 64, 32, 64, 64};
        else if (n <= 128) cands[k++] = {128, 128, 32, 64, 64};
        else if (n <= 256) cands[k++] = {64, 256, 64, 64, 64};
        else if (n <= 512) cands[k++] = {32, 512, 32, 128, 32};
        else if (n <= 1024) cands[k++] = {32, 1024, 32, 128, 32};
        else return VDMI_E_SHAPE;
    } else if (epi == VDMI_EPI_PLAIN) {
        if (n <= 64 || tile_bn == 64) { cands[k++] = {256, 64, 32, 64, 64}; cands[k++] = {64, 64, 32, 32, 64}; }
        else {
            cands[k++] = {128, 128, 32, 64, 64};
            cands[k++] = {64, 128, 32, 64, 64};
        }
    } else if (epi == VDMI_EPI_QKV_ATTN || epi == VDMI_EPI_QKV_SLA) {
        cands[k++] = {128, 128, 32, 64, 64};
    } else {
        return VDMI_E_ARG;
    }
    (void)stride;
    *ncand = k;
    return VDMI_OK;
}

}   // namespace

extern "C" int vdmi_tapgemm_tile(int epilogue, int n, int* bn, int* bkc) {
    TileCfg c[4];
    int nc = 0;
    const int st = pick_tiles(epilogue, n, 1, c, &nc);
    if (st != VDMI_OK) return st;
    if (bn) *bn = c[0].bn;
    if (bkc) *bkc = c[0].bkc;
    return VDMI_OK;
}

int vdmi_wsconv_try(const vdmi_tapgemm_args& A, hipStream_t stream);   // wsconv.hip
int vdmi_dconv_try(const vdmi_tapgemm_args& A, hipStream_t stream);    // dconv.hip

extern "C" int vdmi_tapgemm(const vdmi_tapgemm_args* args, vdmi_stream stream_) {
    if (!args || !args->out) return VDMI_E_ARG;
    const vdmi_tapgemm_args& A = *args;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int epi = A.epilogue;
    if (A.n <= 0 || A.n % 32 || A.ldo % 8 || A.frames <= 0 || A.grid_h <= 0 || A.grid_w <= 0) return VDMI_E_SHAPE;
    if (!aligned16(A.out)) return VDMI_E_ALIGN;
    if (epi == VDMI_EPI_NORM_SILU && !A.gain) return VDMI_E_ARG;
    if ((epi == VDMI_EPI_QKV_ATTN || epi == VDMI_EPI_QKV_SLA) && !A.colsum) return VDMI_E_ARG;
    if (epi == VDMI_EPI_QKV_SLA && (!A.out2 || A.sla_hw <= 0)) return VDMI_E_ARG;
    if (A.act < 0 || A.act > 1 || (A.act && epi != VDMI_EPI_PLAIN)) return VDMI_E_ARG;
    if (A.q_cols < 0 || A.q_cols % 32 || (A.q_cols && epi != VDMI_EPI_QKV_ATTN)) return VDMI_E_ARG;
    if (A.has_res == 2 && (!A.res_identity || A.ld_res % 8 || !aligned16(A.res_identity))) return VDMI_E_ARG;
    if (A.has_res == 1 && epi != VDMI_EPI_NORM_SILU) return VDMI_E_ARG;
    if (A.main.phases > 1 && (epi != VDMI_EPI_PLAIN || A.main.phases != A.out_mul * A.out_mul)) return VDMI_E_ARG;
    // the loader indexes source pixels with 32-bit integers
    if ((long long)A.frames * A.main.in_h * A.main.in_w > (A.main.space_to_depth ? 0x7ffffffLL : 0x7fffffffLL))
        return VDMI_E_SHAPE;
    if (A.main.space_to_depth && (epi == VDMI_EPI_QKV_ATTN || epi == VDMI_EPI_QKV_SLA)) return VDMI_E_ARG;
    if (A.k_split < 0 || A.k_split > 4) return VDMI_E_ARG;
    if (A.k_split > 1 && (epi != VDMI_EPI_PLAIN || A.act || A.has_res || A.out_split_stride <= 0 || A.out_split_stride % 8))
        return VDMI_E_ARG;

    // 64 -> 64 channel (1,3,3) convolutions of wide images: the weight-stationary kernel; 128 output channels on
    // 8- to 32-pixel-wide images: the streaming kernel with the normalisation epilogue
    if (epi == VDMI_EPI_NORM_SILU) {
        const int ws = vdmi_wsconv_try(A, stream);
        if (ws <= 0) return ws;
        const int dc = vdmi_dconv_try(A, stream);
        if (dc <= 0) return dc;
    }

    // (1,3,3) convolutions of the wide, small-image layers: the streaming kernel
    if (epi == VDMI_EPI_PLAIN && A.tile_bn == 64) {
        const int dc = vdmi_dconv_try(A, stream);
        if (dc <= 0) return dc;
    }

    TileCfg cands[4];
    int ncand = 0;
    int st = pick_tiles(epi, A.n, A.main.stride, cands, &ncand, A.tile_bn);
    if (st != VDMI_OK) return st;

    const bool is_qkv = (epi == VDMI_EPI_QKV_ATTN || epi == VDMI_EPI_QKV_SLA);
    // a 1-tap, stride-1 pass with shared weights and no per-batch tables is a plain row GEMM:
    // run it over the flattened row index so tiles never straddle image edges.
    const bool flat = (A.main.ntaps == 1 && A.main.stride == 1 && A.main.dy[0] == 0 && A.main.dx[0] == 0 &&
                       A.main.w_group_stride == 0 && epi != VDMI_EPI_NORM_SILU && A.out_mul == 1 &&
                       A.out_off_y == 0 && A.out_off_x == 0 && A.grid_h == A.out_h && A.grid_w == A.out_w &&
                       A.main.in_h == A.out_h && A.main.in_w == A.out_w);
    if (is_qkv && !flat) return VDMI_E_SHAPE;

    vdmi_tap_pass mainp = A.main;
    int frames = A.frames, grid_h = A.grid_h, grid_w = A.grid_w, out_h = A.out_h, out_w = A.out_w;
    if (flat) {
        const long long rows = (long long)A.frames * A.out_h * A.out_w;
        if (rows > 0x7fffffffLL) return VDMI_E_SHAPE;
        mainp.in_h = 1;
        mainp.in_w = (int)rows;
        frames = 1;
        grid_h = out_h = 1;
        grid_w = out_w = (int)rows;
    }

    for (int ci = 0; ci < ncand; ++ci) {
        const TileCfg& c = cands[ci];
        const int npad = ((A.n + c.bn - 1) / c.bn) * c.bn;
        KArgs k;
        memset(&k, 0, sizeof(k));
        st = build_pass(mainp, c.bkc, npad, &k.main);
        if (st != VDMI_OK) return st;
        PassDev res_base;
        memset(&res_base, 0, sizeof(res_base));
        if (A.has_res == 1) {
            if (A.res.ntaps != 1 || A.res.stride != 1 || A.res.dy[0] || A.res.dx[0]) return VDMI_E_SHAPE;
            st = build_pass(A.res, c.bkc, npad, &res_base);
            if (st != VDMI_OK) return st;
        }
        // choose the tile shape (TH x TW = bm, TW a power of two) with the fewest tiles that fits LDS
        int best_ltw = -1;
        long long best_tiles = 0;
        int best_rows = 0, best_skew = 0, best_ph = 0;
        for (int ltw = 0; (1 << ltw) <= c.bm; ++ltw) {
            const int tw = 1 << ltw, th = c.bm >> ltw;
            if (flat && th != 1) continue;
            const int ph = (th - 1) * mainp.stride + k.main.ph + 1, pw = (tw - 1) * mainp.stride + k.main.pw + 1;
            int rows = ph * pw;
            if (A.has_res == 1 && c.bm > rows) rows = c.bm;
            const int skew = (flat || is_qkv) ? 0 : pick_lskew(ltw, pw, mainp.stride, c.bkc, ph, lds_bytes(c, rows));
            if (lds_bytes(c, rows, ph * skew) > kLdsMax) continue;
            {   // the loader keeps a whole chunk of the patch in registers: rows it can cover
                if (!is_qkv && rows > patch_row_cap(c)) continue;
            }
            const long long tiles = (long long)((grid_h + th - 1) / th) * ((grid_w + tw - 1) / tw);
            if (best_ltw < 0 || tiles < best_tiles || (tiles == best_tiles && rows < best_rows)) {
                best_ltw = ltw;
                best_tiles = tiles;
                best_rows = rows;
                best_skew = skew;
                best_ph = ph;
            }
        }
        // images smaller than the pixel tile: stack whole frames into one tile (shared weights only)
        int fstack = 1;
        if (epi == VDMI_EPI_PLAIN && !flat && frames > 1 && mainp.w_group_stride == 0) {
            long long best_blocks = best_ltw < 0 ? (1LL << 62) : best_tiles * frames;
            for (int ltw = 0; (1 << ltw) <= c.bm; ++ltw) {
                const int tw = 1 << ltw, th = c.bm >> ltw;
                if (tw < grid_w || th < 2 * grid_h) continue;
                const int fs = th / grid_h < frames ? th / grid_h : frames;
                const int phf = (grid_h - 1) * mainp.stride + k.main.ph + 1, pw = (tw - 1) * mainp.stride + k.main.pw + 1;
                const int rows = fs * phf * pw;
                const int skew = pick_lskew(ltw, pw, mainp.stride, c.bkc, fs * phf, lds_bytes(c, rows));
                if (lds_bytes(c, rows, fs * phf * skew) > kLdsMax || rows > patch_row_cap(c)) continue;
                const long long blocks = (frames + fs - 1) / fs;
                if (blocks < best_blocks) {
                    best_blocks = blocks;
                    best_ltw = ltw;
                    best_tiles = 1;
                    best_rows = rows;
                    best_skew = skew;
                    best_ph = fs * phf;
                    fstack = fs;
                }
            }
        }
        if (best_ltw < 0) continue;   // try a smaller pixel tile
        const int tw = 1 << best_ltw, th = c.bm >> best_ltw;
        finish_pass(mainp, &k.main, th, tw, c.bkc, best_skew);
        k.main.fstack = fstack;
        k.main.nframes = frames;
        if (fstack > 1) {
            k.main.fs_rows = grid_h;
            k.main.phf = (grid_h - 1) * mainp.stride + (k.main.ph - ((th - 1) * mainp.stride + 1)) + 1;
            k.main.ph = fstack * k.main.phf;
            frames = (frames + fstack - 1) / fstack;   // grid.z: frame groups
        }
        k.main.inv_pw = 1.0f / (float)k.main.pw;
        k.main.inv_phf = k.main.phf > 0 ? 1.0f / (float)k.main.phf : 0.f;
        if (A.has_res == 1) {
            k.res = res_base;
            finish_pass(A.res, &k.res, th, tw, c.bkc, 0);
            k.res.inv_pw = 1.0f / (float)k.res.pw;
            k.res.inv_phf = 0.f;
        }
        k.has_res = A.has_res;
        k.res_identity = static_cast<const __bf16*>(A.res_identity);
        k.ld_res = A.ld_res;
        k.res_bias = A.res_bias;
        k.frames_per_batch = A.frames_per_batch > 0 ? A.frames_per_batch : 1;
        k.grid_h = grid_h;
        k.grid_w = grid_w;
        k.tiles_x = (grid_w + tw - 1) / tw;
        k.ltw = best_ltw;
        k.out_mul = A.out_mul > 0 ? A.out_mul : 1;
        k.out_off_y = A.out_off_y;
        k.out_off_x = A.out_off_x;
        k.out_h = out_h;
        k.out_w = out_w;
        k.out = static_cast<__bf16*>(A.out);
        k.ldo = A.ldo;
        k.n = A.n;
        k.bias = A.bias;
        k.gain = A.gain;
        k.shift = A.shift;
        k.gain_bstride = A.gain_bstride;
        k.shift_bstride = A.shift_bstride;
        k.norm_scale = A.norm_scale;
        k.colsum = A.colsum;
        k.ln_eps = A.ln_eps;
        k.q_scale = A.q_scale;
        k.act = A.act;
        k.ksplit = A.k_split > 1 ? A.k_split : 1;
        k.out_split_stride = A.out_split_stride;
        if (k.ksplit > k.main.nch) return VDMI_E_SHAPE;      // every share needs at least one channel chunk
        k.rot_cos = A.rot_cos;
        k.rot_sin = A.rot_sin;
        k.rot_frames = A.rot_frames > 0 ? A.rot_frames : 1;
        k.rot_hw = A.rot_hw > 0 ? A.rot_hw : 1;
        k.out2 = static_cast<__bf16*>(A.out2);
        k.sla_hw = A.sla_hw > 0 ? A.sla_hw : 1;
        k.heads32 = (epi == VDMI_EPI_QKV_ATTN && A.q_cols > 0) ? A.q_cols : A.n / 3;
#ifdef VDMI_DIAG_STAMPS
        if (const char* dp = std::getenv("VDMI_DIAG_PTR")) k.diag = reinterpret_cast<unsigned long long*>(std::strtoull(dp, nullptr, 0));
#endif
        if (is_qkv && (k.heads32 % 32 || A.n % k.heads32)) return VDMI_E_SHAPE;

        frames *= k.main.phases * k.ksplit;    // grid.z: (frame group, phase, K share)
        if (best_tiles > 0x7fffffffLL || frames > 65535) return VDMI_E_SHAPE;
        const dim3 grid((unsigned)best_tiles, (unsigned)(npad / c.bn), (unsigned)frames);
        if (grid.y > 65535) return VDMI_E_SHAPE;
        const size_t lds = lds_bytes(c, best_rows, best_ph * best_skew);

        // plain row GEMMs (1 tap over flattened rows): the loader with the row-chunk ring
        if (flat && epi == VDMI_EPI_PLAIN && k.main.fstack <= 1 && !k.main.s2d && k.main.phases <= 1) {
            if (c.bm == 128 && c.bn == 128 && c.wm == 32 && c.wn == 64 && c.bkc == 64)
                return launch<128, 128, 32, 64, 64, VDMI_EPI_PLAIN, true>(k, grid, lds, stream);
            if (c.bm == 64 && c.bn == 128 && c.wm == 32 && c.wn == 64 && c.bkc == 64)
                return launch<64, 128, 32, 64, 64, VDMI_EPI_PLAIN, true>(k, grid, lds, stream);
        }
#define VDMI_CASE(BM_, BN_, WM_, WN_, BKC_, EPI_)                                                   \
    if (c.bm == BM_ && c.bn == BN_ && c.wm == WM_ && c.wn == WN_ && c.bkc == BKC_ && epi == EPI_)  \
        return launch<BM_, BN_, WM_, WN_, BKC_, EPI_>(k, grid, lds, stream);
        VDMI_CASE(256, 64, 64, 64, 64, VDMI_EPI_NORM_SILU)
        VDMI_CASE(256, 64, 32, 64, 64, VDMI_EPI_NORM_SILU)
        VDMI_CASE(128, 128, 64, 64, 64, VDMI_EPI_NORM_SILU)
        VDMI_CASE(128, 128, 32, 64, 64, VDMI_EPI_NORM_SILU)
        VDMI_CASE(64, 256, 64, 64, 64, VDMI_EPI_NORM_SILU)
        VDMI_CASE(32, 512, 32, 128, 32, VDMI_EPI_NORM_SILU)
        VDMI_CASE(32, 1024, 32, 128, 32, VDMI_EPI_NORM_SILU)
        VDMI_CASE(256, 64, 64, 64, 64, VDMI_EPI_PLAIN)
        VDMI_CASE(256, 64, 32, 64, 64, VDMI_EPI_PLAIN)
        VDMI_CASE(64, 64, 32, 32, 64, VDMI_EPI_PLAIN)
        VDMI_CASE(128, 128, 64, 64, 64, VDMI_EPI_PLAIN)
        VDMI_CASE(128, 128, 32, 64, 64, VDMI_EPI_PLAIN)
        VDMI_CASE(64, 128, 32, 64, 64, VDMI_EPI_PLAIN)
        VDMI_CASE(128, 128, 64, 64, 64, VDMI_EPI_QKV_ATTN)
        VDMI_CASE(128, 128, 64, 64, 64, VDMI_EPI_QKV_SLA)
        VDMI_CASE(128, 128, 32, 64, 64, VDMI_EPI_QKV_ATTN)
        VDMI_CASE(128, 128, 32, 64, 64, VDMI_EPI_QKV_SLA)
#undef VDMI_CASE
        return VDMI_E_ARG;
    }
    return VDMI_E_LDS;
}
