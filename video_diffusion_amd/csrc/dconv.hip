// Streaming (1,3,3) convolution for the wide, small-image layers (the 16x16 and 8x8 levels of the denoiser: 128..1024
// input channels, 128..512 output channels; reference Block.proj, video_diffusion_pytorch/video_diffusion_pytorch.py:412-420
// of the reference).  Plain epilogue: bias, bf16 rows out, optional split of the contraction over workgroups whose partial
// sums the normalisation kernel adds (vdmi_rmsnorm_silu_sum).
//
// Why a third conv kernel.  These launches are small (5k-20k pixels), so the tap-GEMM runs them as 160-320 short-lived
// workgroups of 256 pixels x 64 channels that stage patch and weights through registers and meet at a barrier every 8
// MFMAs; a workgroup spends 40 % of its life outside the matrix loop.  Here a workgroup owns the whole LDS of a CU:
//
//   * tile = 256 pixels x 128 output channels; 8 waves = 4 pixel groups (64 pixels) x 2 channel groups (64 channels);
//     a wave's 64 x 64 block takes 4 ds_read_b128 per 4 MFMAs (the 32 x 64 blocks of the tap-GEMM take 6).
//   * nothing is staged through registers.  The haloed input patch of a 64-channel chunk (all nine taps read it) and the
//     weight tile of every (chunk, tap) step arrive by LDS-DMA (buffer_load ... lds): the patch of chunk c + 1 lands
//     in a second buffer during the nine steps of chunk c, weight tiles cycle through a ring of three and are requested
//     two steps ahead.  A wave waits with a counted vmcnt for ITS pieces of the step's tile only; one raw s_barrier
//     per step (16 MFMAs per wave) then publishes everybody's pieces.  Every step issues the same number of DMA
//     instructions (past the end of the contraction they read out of range: zeros into a slot nobody reads), so the
//     counts are compile-time constants.
//   * LDS image of a patch: [patch pixel][128 B], the eight 16-byte pieces of a pixel XOR-swizzled by
//     ((px >> 1) + S * row) & 7 (S = 4 for 8-pixel-wide images, else 0): a DMA piece is 8 whole source rows (coalesced)
//     and a wave's fragment read of 32 pixels - one to four image rows - spreads over all banks.
//   * images smaller than the tile are stacked: four 8x8 frames, each with its own halo, make one tile.
#include "vdmi_common.h"
#include <cstdlib>

namespace {

struct DcArgs {
    const __bf16* src0;
    const __bf16* src1;
    const __bf16* w;
    const float* bias;
    __bf16* out;
    long long out_split_stride;     // elements between the partial outputs
    int c0, c1, nch0, nch;          // channels of the sources; 64-channel chunks of the first / of both
    int ldo, n, npad32;
    int H, frames, bands;           // image height, frames, bands (tiles) per frame
    int npx, nnt, ksplit;           // pixel tiles, 128-channel tiles, shares of the contraction
    unsigned src0_bytes, src1_bytes, w_bytes;
    long long rows;                 // frames * H * W
    // RMSNorm epilogue (n == 128: the tile holds every output channel of its pixels)
    const float* gain;              // [B][gain_bs]: gamma * (scale + 1)
    const float* shift;             // [B][shift_bs] or null
    long long gain_bs, shift_bs;
    const __bf16* res;              // identity residual rows [rows][ld_res] or null
    int ld_res, fpb;                // frames per batch element
    float norm_scale;
    unsigned long long* diag;       // diagnostic builds only
};

typedef __attribute__((address_space(3))) void* lds_ptr;

// Diagnostic builds only (-DVDMI_DIAG_STAMPS, never shipped): lane 0 of wave 0 of every workgroup records the cycle
// counter at a few points into a buffer whose address comes from the VDMI_DIAG_PTR environment variable.
#ifdef VDMI_DIAG_STAMPS
#define DC_STAMP(slot)                                                                                 \
    do {                                                                                               \
        if (a.diag && threadIdx.x == 0) a.diag[(size_t)blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define DC_STAMP(slot) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 8, "vmcnt");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// W: image width; R: image rows per sub-block of the tile (a whole 8x8 / 16x16 frame, or a band of a larger one);
// the tile stacks NF = 256 / (R * W) sub-blocks.
// NORM: bias, RMSNorm over the 128 output channels, * gain + shift, SiLU, + identity residual (Block.forward and the
// ResnetBlock sum, reference vdp.py:391-427, 452) instead of the plain epilogue.
// N64: the layer has 64 output channels (the 128 -> 64 Blocks of the full-resolution levels).  The tile is 256 pixels x
// 64 channels and the two wave groups split the CONTRACTION instead of the channels: wave group wk takes k-steps 2 wk and
// 2 wk + 1 of every (chunk, tap) step over all 64 channels (the same 64 x 64 block per wave, 8 MFMAs per wave and step),
// weight tiles are 8 KiB, and the two partial blocks of a pixel group are added through LDS before the epilogue, which
// each wave then runs for 32 of the 64 pixels with all of their channels in its own registers.  NORM only.
template <int W, int R, bool NORM, bool N64 = false, bool PRES = false>
__global__ __launch_bounds__(512) void dconv_kernel(const DcArgs a) {
    static_assert(!N64 || NORM, "the 64-channel form exists with the normalisation epilogue only");
    static_assert(!PRES || !NORM, "PRES: the plain epilogue + an identity residual (the normalising one takes a.res as it is)");
    constexpr int NF = 256 / (R * W), PW = W + 2, PR = R + 2, NPP = NF * PR * PW;
    constexpr int NPIECE = (NPP + 7) / 8, PBUF = NPIECE * 1024, NU = (NPIECE + 7) / 8;
    constexpr int S = (W == 8) ? 4 : 0;
    constexpr int WOFF = 2 * PBUF, WSLOT = (N64 ? 64 : 128) * 64 * 2, RING = 3;
    static_assert(NF * R * W == 256 && NU <= 7, "tile geometry");
    static_assert(WOFF + RING * WSLOT <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    DC_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;

    // ---- which tile: workgroups of one XCD (blockIdx.x % 8 equal) take a run of consecutive work items, whose
    // pixel tiles differ and whose weight slice is the same
    const int nwg = gridDim.x;
    int wid = blockIdx.x;
    if ((nwg & 7) == 0) wid = (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
    const int tile = wid % a.npx;
    const int rest = wid / a.npx;
    const int ksp = rest % a.ksplit, nt = rest / a.ksplit;
    const int c_begin = ksp * a.nch / a.ksplit, c_end = (ksp + 1) * a.nch / a.ksplit;
    const int nsteps = (c_end - c_begin) * 9;

    int f0, y0;
    if constexpr (NF > 1) { f0 = tile * NF; y0 = 0; }
    else { f0 = tile / a.bands; y0 = (tile - f0 * a.bands) * R; }

    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, a.w_bytes, 0x00020000);

    // ---- what this lane fetches for each of its patch pieces: source row * 8 + 16-byte piece, or -1 (halo outside the
    // image, padding of the last piece, frame past the end).  Waves whose piece index runs past the last piece fetch the
    // last piece once more (same bytes to the same place), so that every wave issues the same number of requests.
    int psrc[NU], pdst[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = min(wave + 8 * u, NPIECE - 1);
        pdst[u] = j * 1024;
        const int pp = j * 8 + (lane >> 3);
        const int ry = pp / PW, px = pp - ry * PW;
        const int fi = ry / PR, ryl = ry - fi * PR;
        const int g = ((px >> 1) + S * ry) & 7;
        const int c16 = (lane & 7) ^ g;
        const int fr = f0 + fi, y = y0 + ryl - 1, x = px - 1;
        const bool ok = pp < NPP && fr < a.frames && y >= 0 && y < a.H && x >= 0 && x < W;
        psrc[u] = ok ? (((fr * a.H + y) * W + x) << 3 | c16) : -1;
    }
    // the chunk whose patch is being requested: tensor it lies in, row pitch and byte offset of its channels
    // (pvalid = 0: past this share's last chunk, every lane fetches out of range)
    const __bf16* pbase = a.src0;
    unsigned pbytes = 0;
    int ppitch = 0, pcb = 0, pvalid = 0;
    auto patch_chunk = [&](int c) {
        const bool first = c < a.nch0;
        ppitch = (first ? a.c0 : a.c1) * 2;
        pcb = (first ? c : c - a.nch0) * 128;
        pbase = first ? a.src0 : a.src1;
        pbytes = first ? a.src0_bytes : a.src1_bytes;
        pvalid = c < c_end;
    };
    auto issue_patch_piece = [&](int buf, int u) {
        const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pbase), 0, pbytes, 0x00020000);
        const int ps = psrc[u];
        const unsigned off = (ps < 0 || !pvalid) ? 0x80000000u : (unsigned)((ps >> 3) * ppitch + (ps & 7) * 16 + pcb);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lds_ptr)(smem + buf * PBUF + pdst[u]), 16, off, 0, 0, 0);
    };
    // weight tile of step s = (chunk, tap): 16 KiB contiguous in the packed order; this wave's two 1-KiB pieces
    const unsigned wlane = (unsigned)(lane * 16 + wave * 1024);
    const unsigned wbase0 = (unsigned)((c_begin * 9 * a.npad32 + nt * (N64 ? 2 : 4)) * 4096), wstep = (unsigned)(a.npad32 * 4096);
    auto issue_weights = [&](int s, int slot) {
        const unsigned off = s < nsteps ? wbase0 + (unsigned)s * wstep + wlane : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(smem + WOFF + slot * WSLOT + wave * 1024), 16, off, 0, 0, 0);
        if constexpr (!N64)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(smem + WOFF + slot * WSLOT + 8192 + wave * 1024), 16,
                                                     s < nsteps ? off + 8192u : 0x80000000u, 0, 0, 0);
    };

    // ---- patch addresses of this lane's pixel in each of the wave's two 32-pixel blocks
    int rowb[2], lowx[2][3], xvx[2][3], rflip[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wm * 64 + j * 32 + l31;
        const int fi = p / (R * W), q = p - fi * (R * W);
        const int y = q / W, x = q - y * W;
        const int ryc = fi * PR + y + 1, pxc = x + 1;
        rowb[j] = ryc * PW * 128;
        rflip[j] = S ? (ryc & 1) << 6 : 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int pxp = pxc + d - 1;
            const int gx = (pxp >> 1) & 7;
            lowx[j][d] = pxp * 128 + ((h ^ (gx & 1)) << 4);
            xvx[j][d] = (gx >> 1) << 5;
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragments of one k-step (16 input channels): two weight fragments (32 output channels each) and two pixel
    // fragments (32 pixels each).  Two register sets alternate, so the reads of k-step k + 1 are in flight during the
    // products of k-step k; the order is pinned (left alone the compiler reads, waits and multiplies per k-step).
    bf16x8 fa[2][2], fb[2][2];
    const char* wlds = smem + WOFF + (N64 ? 0 : wn * 8192) + lane * 16;
    [[maybe_unused]] const int ka = 2 * wn, kb = 2 * wn + 1;        // N64: this wave group's two k-steps of every step
    auto tap_addr = [&](int t, int (&ab)[2], int (&xv)[2]) {
        const int dy = t / 3 - 1, dxi = t % 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ab[j] = rowb[j] + dy * PW * 128 + lowx[j][dxi];
            xv[j] = xvx[j][dxi] ^ (S ? (rflip[j] ^ ((dy & 1) << 6)) : 0);
        }
    };
    // k-step ks of the step whose weights sit in ring slot `slot` and whose pixels are read at (pb, ab, xv): the four
    // products of register set `set`, each preceded by one fragment read of the NEXT k-step (nslot, npb, nab, nxv, nks)
    auto kstep = [&](int set, const char* nwb, const char* npb, const int (&nab)[2], const int (&nxv)[2], int nks) {
        const int o = set ^ 1;
        fa[o][0] = *reinterpret_cast<const bf16x8*>(nwb + nks * 1024);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][0], fb[set][0], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fb[o][0] = *reinterpret_cast<const bf16x8*>(npb + nab[0] + ((nks << 5) ^ nxv[0]));
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][0], fb[set][1], acc[0][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fa[o][1] = *reinterpret_cast<const bf16x8*>(nwb + (4 + nks) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][1], fb[set][0], acc[1][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fb[o][1] = *reinterpret_cast<const bf16x8*>(npb + nab[1] + ((nks << 5) ^ nxv[1]));
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][1], fb[set][1], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the first chunk's patch and the weight tiles of steps 0 and 1; then the fragments of k-step 0
    patch_chunk(c_begin);
#pragma unroll
    for (int u = 0; u < NU; ++u) issue_patch_piece(0, u);
    issue_weights(0, 0);
    issue_weights(1, 1);
    DC_STAMP(1);
    wait_vm<(N64 ? 1 : 2)>();           // patch and tile 0 (tile 1 may still be on its way)
    __builtin_amdgcn_s_barrier();
    DC_STAMP(2);
    {
        int ab[2], xv[2];
        tap_addr(0, ab, xv);
        const int k0 = N64 ? ka : 0;
        fa[0][0] = *reinterpret_cast<const bf16x8*>(wlds + k0 * 1024);
        fa[0][1] = *reinterpret_cast<const bf16x8*>(wlds + (4 + k0) * 1024);
        fb[0][0] = *reinterpret_cast<const bf16x8*>(smem + ab[0] + ((k0 << 5) ^ xv[0]));
        fb[0][1] = *reinterpret_cast<const bf16x8*>(smem + ab[1] + ((k0 << 5) ^ xv[1]));
    }

    // ---- main loop.  Step s = (chunk c, tap t):
    //   k-steps 0, 1;  wait for this wave's pieces of tile s + 1 (and, with them, of every older request);  barrier:
    //   tile s + 1 is complete, and every wave has left step s - 1, whose ring slot and patch buffer are free;
    //   request tile s + 2 into that slot (+ one patch piece of chunk c + 1 in steps t < NU);  k-steps 2, 3, the last
    //   one reading the first fragments of step s + 1.
    int s = 0, slot = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const char* pb = smem + buf * PBUF;
        patch_chunk(c + 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int ab[2], xv[2], nab[2], nxv[2];
            tap_addr(t, ab, xv);
            tap_addr(t == 8 ? 0 : t + 1, nab, nxv);
            const char* wb = wlds + slot * WSLOT;
            int slot1 = slot + 1;
            if (slot1 == RING) slot1 = 0;
            int slot2 = slot1 + 1;
            if (slot2 == RING) slot2 = 0;
            if constexpr (N64) {
                kstep(0, wb, pb, ab, xv, kb);
            } else {
                kstep(0, wb, pb, ab, xv, 1);
                kstep(1, wb, pb, ab, xv, 2);
            }
            // the only request younger than tile s + 1 is the patch piece of the step before (steps 1 .. NU)
#ifndef VDMI_DC_ABL_NOWAIT      // timing-only ablation build (wrong results): how much of a step is waiting for the weight tile
            if (t >= 1 && t <= NU) wait_vm<1>();
            else wait_vm<0>();
#endif
#ifndef VDMI_DC_ABL_NOBARRIER   // second timing-only ablation: the cost of the mid-step barrier
            __builtin_amdgcn_s_barrier();
#endif
            issue_weights(s + 2, slot2);
            if (t < NU) issue_patch_piece(buf ^ 1, t);
            if constexpr (N64) {
                kstep(1, wlds + slot1 * WSLOT, t == 8 ? smem + (buf ^ 1) * PBUF : pb, nab, nxv, ka);
            } else {
                kstep(0, wb, pb, ab, xv, 3);
                kstep(1, wlds + slot1 * WSLOT, t == 8 ? smem + (buf ^ 1) * PBUF : pb, nab, nxv, 0);
            }
            ++s;
            slot = slot1;
        }
    }
    DC_STAMP(3);
    wait_vm<0>();                       // the out-of-range requests of the last steps still write (zeros) into LDS
    __builtin_amdgcn_s_barrier();
    DC_STAMP(4);

    if constexpr (N64) {
        // ---- 64-channel form: add the two partial blocks of each pixel group, then every wave finishes 32 pixels.
        // Wave (wm, wk) keeps pixel block wk and hands block 1 - wk to its partner wave ^ 4: 32 registers per lane.
        constexpr int SROW = 64 + 8;
        const int wk = wn;
        float keep[2][16];
        {
            float* xch = reinterpret_cast<float*>(smem);                  // [wave][32 registers][64 lanes]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    xch[(wave * 32 + i * 16 + r) * 64 + lane] = wk ? acc[i][0][r] : acc[i][1][r];
                    keep[i][r] = wk ? acc[i][1][r] : acc[i][0][r];
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep[i][r] += xch[((wave ^ 4) * 32 + i * 16 + r) * 64 + lane];
            __syncthreads();                                              // the staging below reuses this memory
        }
        const int n0 = nt * 64;
        if (a.bias) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n0 + i * 32 + 8 * rg + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) keep[i][4 * rg + e] += bv[e];
                }
        }
        float ssq = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ssq = fmaf(keep[i][r], keep[i][r], ssq);
        ssq = xhalf_add(ssq);
        const float rinv = a.norm_scale / fmaxf(sqrtf(ssq), 1e-12f);
        __bf16* stg = reinterpret_cast<__bf16*>(smem) + wave * 32 * SROW;
        const int srow0 = lane >> 3, schunk = lane & 7;
        const long long row0 = (long long)tile * 256 + wm * 64 + wk * 32;
        if (a.res) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = k * 8 + srow0;
                bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (row0 + row < a.rows) rv = *reinterpret_cast<const bf16x8*>(a.res + (row0 + row) * a.ld_res + n0 + schunk * 8);
                *reinterpret_cast<bf16x8*>(stg + row * SROW + schunk * 8) = rv;
            }
        }
        const int frame = NF > 1 ? f0 + wm : f0;
        const int b = frame / a.fpb;
        const bool bok = frame < a.frames;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch = n0 + i * 32 + 8 * rg + 4 * h;
                f32x4 gv = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
                if (bok) {
                    gv = *reinterpret_cast<const f32x4*>(a.gain + b * a.gain_bs + ch);
                    if (a.shift) sv = *reinterpret_cast<const f32x4*>(a.shift + b * a.shift_bs + ch);
                }
                __bf16* sp = stg + l31 * SROW + i * 32 + 8 * rg + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(fmaf(keep[i][4 * rg + e] * rinv, gv[e], sv[e]));
                if (a.res) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(sp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
                *reinterpret_cast<bf16x4*>(sp) = o;
            }
        // (a wave reads back only what it wrote itself: no barrier)
        __bf16* op = a.out + n0 + schunk * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = k * 8 + srow0;
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(stg + row * SROW + schunk * 8);
            if (row0 + row < a.rows) *reinterpret_cast<bf16x8*>(op + (row0 + row) * a.ldo) = ov;
        }
        DC_STAMP(5);
        return;
    }

    // ---- epilogue: + bias (share 0) [NORM: RMSNorm, gain / shift, SiLU, residual], bf16, transposed through LDS so that
    // 8 lanes store one pixel's 128 contiguous bytes
    const int n0 = nt * 128 + wn * 64;
    if (a.bias && ksp == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n0 + i * 32 + 8 * rg + 4 * h);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * rg + e] += bv[e];
            }
    }
    constexpr int SROW = 64 + 8;
    constexpr int RED_OFF = 8 * 64 * SROW * 2;          // behind the staging regions: [2 channel groups][256 pixels] floats
    __bf16* stg = reinterpret_cast<__bf16*>(smem) + wave * 64 * SROW;
    const int srow0 = lane >> 3, schunk = lane & 7;
    const long long row0 = (long long)tile * 256 + wm * 64;
    if constexpr (NORM) {
        // sum of squares over this wave's 64 channels; the wave with the other 64 channels of the same pixels is wave ^ 4
        float ss[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) v = fmaf(acc[i][j][r], acc[i][j][r], v);
            ss[j] = xhalf_add(v);
        }
        float* red = reinterpret_cast<float*>(smem + RED_OFF);
        if (lane < 32) {
            red[wn * 256 + wm * 64 + l31] = ss[0];
            red[wn * 256 + wm * 64 + 32 + l31] = ss[1];
        }
        // identity residual: fetched with the coalesced pattern of the stores below, added in fp32 further down
        if (a.res) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = k * 8 + srow0;
                bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (row0 + row < a.rows) rv = *reinterpret_cast<const bf16x8*>(a.res + (row0 + row) * a.ld_res + n0 + schunk * 8);
                *reinterpret_cast<bf16x8*>(stg + row * SROW + schunk * 8) = rv;
            }
        }
        __syncthreads();
        const int frame = NF > 1 ? f0 + wm : f0;
        const int b = frame / a.fpb;
        float rinv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float tot = ss[j] + red[(wn ^ 1) * 256 + wm * 64 + j * 32 + l31];
            rinv[j] = a.norm_scale / fmaxf(sqrtf(tot), 1e-12f);
        }
        const bool bok = frame < a.frames;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch = n0 + i * 32 + 8 * rg + 4 * h;
                f32x4 gv = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
                if (bok) {
                    gv = *reinterpret_cast<const f32x4*>(a.gain + b * a.gain_bs + ch);
                    if (a.shift) sv = *reinterpret_cast<const f32x4*>(a.shift + b * a.shift_bs + ch);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j][4 * rg + e] = silu_f(fmaf(acc[i][j][4 * rg + e] * rinv[j], gv[e], sv[e]));
            }
    }
    if constexpr (PRES) {
        // identity residual of the plain epilogue (the input gradient of a block's first convolution + the gradient of
        // its residual branch): through the wave's own staging rows, as above; a wave reads only what it wrote
        if (a.res) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = k * 8 + srow0;
                bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
                if (row0 + row < a.rows) rv = *reinterpret_cast<const bf16x8*>(a.res + (row0 + row) * a.ld_res + n0 + schunk * 8);
                *reinterpret_cast<bf16x8*>(stg + row * SROW + schunk * 8) = rv;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                __bf16* sp = stg + (j * 32 + l31) * SROW + i * 32 + 8 * rg + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rg + e];
                if constexpr (NORM || PRES) {
                    if (a.res) {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(sp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
                *reinterpret_cast<bf16x4*>(sp) = o;
            }
    // (a wave reads back only what it wrote itself: no barrier)
    __bf16* op = a.out + (long long)ksp * a.out_split_stride + n0 + schunk * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int row = k * 8 + srow0;
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(stg + row * SROW + schunk * 8);
        if (row0 + row < a.rows) *reinterpret_cast<bf16x8*>(op + (row0 + row) * a.ldo) = ov;
    }
    DC_STAMP(5);
}

int g_dc_enabled = -1;
int g_dc_norm_all = 0;      // tests: the normalisation epilogue on every image width the kernel is built for
int g_dc_n64 = -1;          // the 64-output-channel form: off unless VDMI_DCONV_N64=1 (it lost to the tap-GEMM, see below)

bool dc_covers(const vdmi_tapgemm_args& A) {
    if (g_dc_enabled < 0) {
        const char* e = std::getenv("VDMI_DCONV");
        g_dc_enabled = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_dc_enabled) return false;
    if (g_dc_n64 < 0) {
        // measured per launch of 8 samples on 64x64 images (128 -> 64 channels): 76-78 us against the tap-GEMM's 67-68,
        // and 3.86 against 3.65 ms per denoise step: with 8 MFMAs per wave between barriers and one workgroup per CU the
        // per-step synchronisation and the partial-sum exchange cost more than the 64 x 64 wave tiles save
        const char* e = std::getenv("VDMI_DCONV_N64");
        g_dc_n64 = (e && e[0] == '1') ? 1 : 0;
    }
    const vdmi_tap_pass& m = A.main;
    const bool norm = A.epilogue == VDMI_EPI_NORM_SILU;
    if (norm) {
        // the tile must hold every output channel of its pixels; residual: none or an identity tensor
        if ((A.n != 128 && A.n != 64) || A.has_res == 1 || !A.gain || A.frames_per_batch <= 0 || A.k_split > 1) return false;
        // 64 output channels: the form that splits the contraction over the wave groups; more than one input chunk
        // (the 64 -> 64 layers of these images are the weight-stationary kernel's)
        if (A.n == 64 && (g_dc_n64 == 0 || A.main.c0 + A.main.c1 < 128)) return false;
        // measured per launch of 8 samples: 32-pixel-wide images 42-45 -> 40-44 us, smaller images (80-tile launches)
        // 17-18 -> 18-19 us against the tap-GEMM's fused epilogue: only the former take this kernel.  (The choice
        // depends on the layer's shape alone, never on the batch.)
        if (A.main.in_w != 32 && A.main.in_w != 64 && !g_dc_norm_all) return false;
        if (A.has_res == 2 && (!A.res_identity || A.ld_res % 8 || !aligned16(A.res_identity))) return false;
        if (A.gain_bstride % 4 || A.shift_bstride % 4 || !aligned16(A.gain) || (A.shift && !aligned16(A.shift))) return false;
        if (A.n == 64 && A.main.in_w != 32 && A.main.in_w != 64 && !g_dc_norm_all) return false;
    } else {
        if (A.epilogue != VDMI_EPI_PLAIN || A.act || A.has_res == 1 || A.n % 128 || A.tile_bn != 64) return false;
        // plain + identity residual: one share only (the residual is added where the sum is complete)
        if (A.has_res == 2 && (A.k_split > 1 || !A.res_identity || A.ld_res % 8 || A.ld_res < A.n || !aligned16(A.res_identity)))
            return false;
    }
    if (m.ntaps != 9 || m.stride != 1 || m.c0 <= 0 || m.c0 % 64 || m.c1 % 64) return false;
    for (int t = 0; t < 9; ++t)
        if (m.dy[t] != t / 3 - 1 || m.dx[t] != t % 3 - 1) return false;
    if (m.w_group_stride || m.phases > 1 || m.space_to_depth) return false;
    if (A.out_mul > 1 || A.out_off_y || A.out_off_x) return false;
    const int H = m.in_h, W = m.in_w;
    if (A.grid_h != H || A.grid_w != W || A.out_h != H || A.out_w != W) return false;
    if (!((W == 8 && H == 8) || (W == 16 && H % 16 == 0) || (W == 32 && H % 8 == 0) || (W == 64 && H % 4 == 0))) return false;
    const long long rows = (long long)A.frames * H * W;
    const int cmax = m.c0 > m.c1 ? m.c0 : m.c1;
    if (rows * cmax * 2 >= 0x7fffffffLL || rows * 8 >= 0x7fffffffLL) return false;
    if (!aligned16(m.src0) || (m.c1 && !aligned16(m.src1)) || !aligned16(m.w) || !aligned16(A.out) || A.ldo % 8) return false;
    const int ks = A.k_split > 1 ? A.k_split : 1;
    const int nch = m.c0 / 64 + m.c1 / 64;
    if (ks > nch) return false;
    const long long wbytes = (long long)nch * 9 * A.n * 64 * 2;
    if (wbytes >= 0x7fffffffLL) return false;
    return true;
}

}   // namespace

extern "C" void vdmi_dconv_enable(int on) {
    g_dc_enabled = on ? 1 : 0;
    g_dc_norm_all = on > 1;
    if (on > 1) g_dc_n64 = 1;           // tests: every form the kernel is built for
    else if (g_dc_n64 > 0) g_dc_n64 = -1;   // back to what the environment says
}
extern "C" int vdmi_dconv_covers(const vdmi_tapgemm_args* a) { return a && dc_covers(*a) ? 1 : 0; }

// Runs `A` on the streaming kernel when the launch has its shape.  Returns VDMI_OK when it was launched, 1 when the
// shape is not covered (the caller then takes the tap-GEMM), a negative status on a launch error.
int vdmi_dconv_try(const vdmi_tapgemm_args& A, hipStream_t stream) {
    if (!dc_covers(A)) return 1;
    const vdmi_tap_pass& m = A.main;
    const int H = m.in_h, W = m.in_w;
    DcArgs k;
    k.src0 = static_cast<const __bf16*>(m.src0);
    k.src1 = m.c1 ? static_cast<const __bf16*>(m.src1) : k.src0;
    k.w = static_cast<const __bf16*>(m.w);
    k.bias = A.bias;
    k.out = static_cast<__bf16*>(A.out);
    k.out_split_stride = A.k_split > 1 ? A.out_split_stride : 0;
    k.c0 = m.c0;
    k.c1 = m.c1;
    k.nch0 = m.c0 / 64;
    k.nch = k.nch0 + m.c1 / 64;
    k.ldo = A.ldo;
    k.n = A.n;
    k.npad32 = A.n / 32;            // n is a multiple of 64: packed without padding for 64-channel tiles
    k.H = H;
    k.frames = A.frames;
    k.rows = (long long)A.frames * H * W;
    const int R = W == 8 ? 8 : (W == 16 ? 16 : (W == 32 ? 8 : 4));
    k.bands = H / R;
    k.npx = W == 8 ? (A.frames + 3) / 4 : A.frames * k.bands;
    const bool n64 = A.n == 64;
    k.nnt = n64 ? 1 : A.n / 128;
    k.ksplit = A.k_split > 1 ? A.k_split : 1;
    k.src0_bytes = (unsigned)(k.rows * m.c0 * 2);
    k.src1_bytes = m.c1 ? (unsigned)(k.rows * m.c1 * 2) : 0u;
    k.w_bytes = (unsigned)((long long)k.nch * 9 * A.n * 64 * 2);
    const bool norm = A.epilogue == VDMI_EPI_NORM_SILU;
    k.gain = A.gain;
    k.shift = A.shift;
    k.gain_bs = A.gain_bstride;
    k.shift_bs = A.shift_bstride;
    k.res = A.has_res == 2 ? static_cast<const __bf16*>(A.res_identity) : nullptr;
    k.ld_res = A.ld_res;
    k.fpb = A.frames_per_batch > 0 ? A.frames_per_batch : 1;
    k.norm_scale = A.norm_scale;
    k.diag = nullptr;
#ifdef VDMI_DIAG_STAMPS
    if (const char* dp = std::getenv("VDMI_DIAG_PTR")) k.diag = reinterpret_cast<unsigned long long*>(std::strtoull(dp, nullptr, 0));
#endif
    const long long nwg = (long long)k.npx * k.nnt * k.ksplit;
    if (nwg > 0x7fffffffLL) return VDMI_E_SHAPE;

    static std::atomic<uint64_t> attr_mask[16];
    const int PW = W + 2, PR = R + 2, NF = 256 / (R * W);
    const int npiece = (NF * PR * PW + 7) / 8;
    const size_t lds = (size_t)2 * npiece * 1024 + 3 * (n64 ? 8192 : 16384);
    const size_t stg = n64 ? (size_t)8 * 32 * 64 * 4 : (size_t)8 * 64 * 72 * 2 + 2 * 256 * 4;   // exchange / staging after the loop
    const size_t need = lds > stg ? lds : stg;
#define VDMI_DC_CASE(W_, R_, NORM_, IDX_)                                                                          \
    {                                                                                                              \
        auto fn = dconv_kernel<W_, R_, NORM_>;                                                                     \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(fn), attr_mask[IDX_], 160 * 1024)) return VDMI_E_LAUNCH; \
        hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(512), need, stream, k);                                   \
    }
#define VDMI_DC_CASE64(W_, R_, IDX_)                                                                               \
    {                                                                                                              \
        auto fn = dconv_kernel<W_, R_, true, true>;                                                                \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(fn), attr_mask[IDX_], 160 * 1024)) return VDMI_E_LAUNCH; \
        hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(512), need, stream, k);                                   \
    }
#define VDMI_DC_CASE_RES(W_, R_, IDX_)                                                                             \
    {                                                                                                              \
        auto fn = dconv_kernel<W_, R_, false, false, true>;                                                        \
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(fn), attr_mask[IDX_], 160 * 1024)) return VDMI_E_LAUNCH; \
        hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(512), need, stream, k);                                   \
    }
    if (!norm && k.res) {
        if (W == 8) VDMI_DC_CASE_RES(8, 8, 12) else if (W == 16) VDMI_DC_CASE_RES(16, 16, 13)
        else if (W == 32) VDMI_DC_CASE_RES(32, 8, 14) else VDMI_DC_CASE_RES(64, 4, 15)
    } else
    if (n64) {
        if (W == 8) VDMI_DC_CASE64(8, 8, 8) else if (W == 16) VDMI_DC_CASE64(16, 16, 9)
        else if (W == 32) VDMI_DC_CASE64(32, 8, 10) else VDMI_DC_CASE64(64, 4, 11)
    } else
    if (W == 8) { if (norm) VDMI_DC_CASE(8, 8, true, 0) else VDMI_DC_CASE(8, 8, false, 1) }
    else if (W == 16) { if (norm) VDMI_DC_CASE(16, 16, true, 2) else VDMI_DC_CASE(16, 16, false, 3) }
    else if (W == 32) { if (norm) VDMI_DC_CASE(32, 8, true, 4) else VDMI_DC_CASE(32, 8, false, 5) }
    else { if (norm) VDMI_DC_CASE(64, 4, true, 6) else VDMI_DC_CASE(64, 4, false, 7) }
#undef VDMI_DC_CASE
#undef VDMI_DC_CASE64
#undef VDMI_DC_CASE_RES
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
