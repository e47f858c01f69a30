// Weight-stationary (1,3,3) convolution + RMSNorm + scale/shift + SiLU (+ identity residual) for the
// 64 -> 64 channel layers of the full-resolution levels (reference Block / ResnetBlock,
// video_diffusion_pytorch/video_diffusion_pytorch.py:412-452 of the reference).
//
// Why a second conv kernel: at 64 input and 64 output channels the whole weight set is 72 KiB.  The tap-GEMM
// streams it through LDS once per 256-pixel tile; here every wave keeps its share in REGISTERS for the lifetime
// of a persistent workgroup, so the only LDS traffic of the main loop is the pixel operand.
//
//   * workgroup = 8 waves = 4 pixel groups (64 pixels each) x 2 output-channel slices (32 channels each);
//     a wave holds W[32 channels][9 taps * 64 inputs] as 36 MFMA A-fragments (144 VGPRs).
//   * tile = 256 pixels = a band of TH = 256 / W full image rows (W = 64 or 32).  The band plus its one-pixel halo
//     is loaded straight into LDS by buffer_load ... lds (no register staging; pixels outside the image come
//     back as zeros from the buffer bounds check) into one of two buffers: tile i+1 lands while tile i computes.
//     LDS image of a buffer: [patch pixel][128 B], the eight 16-byte chunks of a pixel stored at slot
//     chunk ^ ((pixel >> 1) & 7): every 1-KiB DMA piece is 8 whole rows of the source tensor (fully coalesced) and a
//     wave's ds_read_b128 of one chunk of 32 consecutive pixels is conflict-free.
//   * matrix phase of a tile: 72 steps of {request the pixel fragment needed 8 steps later, MFMA}, pinned in that
//     order (the scheduler, left alone, requests each fragment right before its use and the wave then waits out
//     the LDS latency 72 times per tile).
//   * one workgroup barrier per tile.  RMSNorm needs the sum of squares over all 64 channels: the two waves
//     that share a pixel group exchange their halves through LDS with a sequence flag (no barrier).
//   * output: each wave transposes its 64 x 32 result through a private 4-KiB LDS region (the identity
//     residual is DMA'd into the same region beforehand) and stores 16 B per lane, 64 B per pixel.
//
// Measured on the device while building this (tools/probes/coissue_probe.hip, mix_probe.hip): a wave that issues
// MFMAs back to back leaves the other wave of its SIMD about one vector instruction per MFMA (a VALU-bound partner
// ran ~13x slower), so the epilogue of one wave does NOT hide behind the matrix phase of its SIMD partner: letting
// waves w and w+4 run half a tile apart bought nothing, and every wave here simply runs matrix phase, then epilogue.
// Vector work hides only inside the issuing wave's own MFMA gaps (about 6 instructions per gap at one wave per
// SIMD); a variant built that way (4 waves of 512 registers, epilogue of the previous tile woven into the MFMA
// stream) was correct but issue-bound at ~60 cycles per MFMA and slower than this kernel.
#include "vdmi_common.h"
#include <cstdlib>

namespace {

struct WsArgs {
    const __bf16* in;          // [frames*H*W][64]
    const __bf16* w;           // vdmi_pack_taps order, one 64-channel chunk: [tap 9][n/32 = 2][k/16 = 4][lane 64][8]
    const float* bias;         // [64] or null
    const float* gain;         // [B][gain_bs]
    const float* shift;        // [B][shift_bs] or null
    const __bf16* res;         // identity residual rows [frames*H*W][ld_res] or null
    __bf16* out;               // [frames*H*W][ldo]
    long long gain_bs, shift_bs;
    int ld_res, ldo;
    int H, W, F, bands, ntiles;
    unsigned in_bytes, res_bytes, out_bytes, gain_bytes, shift_bytes;
    float norm_scale;
    unsigned long long* diag;  // diagnostic builds only: per-wave phase cycle sums
};

// Diagnostic builds (-DVDMI_DIAG_STAMPS, never shipped): every wave sums the cycles it spends in each phase of
// the tile loop and writes the eight sums to diag[(block * 8 + wave) * 8 ..] when it ends.
#ifdef VDMI_DIAG_STAMPS
#define WS_T() __builtin_readcyclecounter()
#define WS_ACC(slot, t0) do { const unsigned long long t1__ = WS_T(); dsum[slot] += t1__ - (t0); (t0) = t1__; } while (0)
#else
#define WS_T() 0ull
#define WS_ACC(slot, t0) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef volatile __attribute__((address_space(3))) float* lds_vfloat;      // LDS words another wave writes / reads
typedef volatile __attribute__((address_space(3))) int* lds_vint;

constexpr bool magic_ok(int pw, int n) {
    const int m = (65536 + pw - 1) / pw;
    for (int p = 0; p < n; ++p)
        if (((p * m) >> 16) != p / pw) return false;
    return true;
}

template <int TW, bool RES>
__global__ __launch_bounds__(512) void wsconv64_kernel(const WsArgs a) {
    constexpr int TH = 256 / TW, PW = TW + 2, PH = TH + 2, NPP = PH * PW;
    constexpr int NINSTR = (NPP + 7) / 8, PBUF = NINSTR * 1024;      // 1-KiB DMA pieces: 8 patch pixels x 128 B
    constexpr int MAGIC = (65536 + PW - 1) / PW;            // pp / PW == (pp * MAGIC) >> 16 for every pp < NINSTR * 8
    static_assert(magic_ok(PW, NINSTR * 8), "reciprocal multiply must be exact over the patch");
    constexpr int OFF_STG = 2 * PBUF, OFF_XCH = OFF_STG + 8 * 4096, OFF_PAR = OFF_XCH + 4096;
    constexpr int OFF_BIAS = OFF_PAR + 3 * 512, OFF_FLAG = OFF_BIAS + 256;
    constexpr int RPG = 64 / TW;                            // image rows per pixel group
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = wave & 1, g = wave >> 1;         // output-channel slice, pixel group

    // ---- this wave's weights: 32 output channels x (9 taps x 64 inputs), resident in registers
    bf16x8 wr[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            wr[t][ks] = *reinterpret_cast<const bf16x8*>(a.w + ((size_t)((t * 2 + s) * 4 + ks) * 64 + lane) * 8);

    // the loop must see these loads as complete (its waits would otherwise also cover the LDS DMAs in flight)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(wr[t][ks]));

    if (tid < 64) reinterpret_cast<float*>(smem + OFF_BIAS)[tid] = a.bias ? a.bias[tid] : 0.f;
    if (tid < 3 * 128) reinterpret_cast<float*>(smem + OFF_PAR)[tid] = 0.f;     // shift stays zero when there is none
    if (tid < 16) reinterpret_cast<int*>(smem + OFF_FLAG)[tid] = 0;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in), 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.res), 0, a.res_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_gain = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gain), 0, a.gain_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_shift = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.shift), 0, a.shift_bytes, 0x00020000);

    // ---- patch DMA.  LDS image of a patch buffer: [patch pixel][128 B], the eight 16-byte chunks of a pixel stored at
    // slot chunk ^ ((pixel >> 1) & 7): a wave's ds_read_b128 of one chunk of 32 consecutive pixels then spreads over all
    // banks.  Piece j (of NINSTR per tile, 1 KiB = 8 whole pixel rows, fully coalesced in memory) is issued by wave
    // j % 8.  What a lane fetches for piece j never changes relative to the tile's first haloed pixel, so it is kept:
    // prel = byte offset | flags (bit 0: pixel of the top halo row, bit 1: of the bottom halo row, bit 2: never
    // inside the image -- left / right halo column or padding of the last piece).
    constexpr int DW = 8, NJ = (NINSTR + DW - 1) / DW;
    int prel[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
        const int j = wave + DW * u;
        const int pp = j * 8 + (lane >> 3);
        const int py = (pp * MAGIC) >> 16, px = pp - py * PW;
        const int c = (lane & 7) ^ ((pp >> 1) & 7);
        const int flags = (py == 0 ? 1 : 0) | (py == PH - 1 ? 2 : 0) | ((pp >= NPP || px == 0 || px == PW - 1) ? 4 : 0);
        prel[u] = ((py * TW + px) * 128 + c * 16) | flags;
    }
    auto issue_patch = [&](int t, int buf, int pk) {
        const int frame = t / a.bands, y0 = (t - frame * a.bands) * TH;
        const int base = ((frame * a.H + y0 - 1) * TW - 1) * 128;     // byte offset of patch pixel (0, 0); may be negative
        const int m = 4 | (y0 == 0 ? 1 : 0) | (y0 + TH == a.H ? 2 : 0);
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int j = wave + DW * u;
            if (j < NINSTR) {
                const unsigned off = (prel[u] & m) ? 0x80000000u : (unsigned)((prel[u] & ~15) + base);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr)(smem + buf * PBUF + j * 1024), 16, off, 0, 0, 0);
            }
        }
        const int b = frame / a.F;
        if (wave == 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_gain, (lds_ptr)(smem + OFF_PAR + pk * 512), 4,
                                                     (unsigned)(lane * 4), (unsigned)(b * a.gain_bs * 4), 0, 0);
        if (wave == 6 && a.shift)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_shift, (lds_ptr)(smem + OFF_PAR + pk * 512 + 256), 4,
                                                     (unsigned)(lane * 4), (unsigned)(b * a.shift_bs * 4), 0, 0);
    };
    // first global pixel row of this wave's 64 pixels of tile t
    auto wave_row0 = [&](int t) {
        const int frame = t / a.bands, y0 = (t - frame * a.bands) * TH;
        return (frame * a.H + y0 + g * RPG) * TW;
    };
    // Coalesced 16-byte view of this wave's 64 pixel x 32 channel block: instruction `it` covers rows 16*it .. 16*it+15,
    // lane -> row 16*it + lane/4, LDS slot lane%4 of the row's four (the slot holds channel piece slot ^ ((row>>1)&3),
    // so the accumulator-layout accesses below spread over the banks).  The piece does not depend on `it`.
    const int qrow = lane >> 2, qpiece = (lane & 3) ^ ((qrow >> 1) & 3);
    const unsigned res_voff = (unsigned)(qrow * a.ld_res * 2 + s * 64 + qpiece * 16);
    const unsigned out_voff = (unsigned)(qrow * a.ldo * 2 + s * 64 + qpiece * 16);
    // identity residual of the block -> the wave's staging region
    auto issue_res = [&](int t) {
        const int prow0 = wave_row0(t);
#pragma unroll
        for (int it = 0; it < 4; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (lds_ptr)(smem + OFF_STG + wave * 4096 + it * 1024), 16, res_voff,
                                                     (unsigned)((prow0 + it * 16) * a.ld_res * 2), 0, 0);
    };

    // patch pixel of this lane's pixel in each of the wave's two 32-pixel sub-tiles (centre tap)
    int pq[2];                   // 128 * patch pixel
    const int hb = h << 4;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
        pq[nb] = 128 * (TW == 64 ? (g + 1) * PW + 32 * nb + l31 + 1 : (2 * g + nb + 1) * PW + l31 + 1);

    f32x16 acc[2];
    float ssown[2];
    [[maybe_unused]] unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] unsigned long long ts = 0;

    auto init_acc = [&]() {
        const float* bl = reinterpret_cast<const float*>(smem + OFF_BIAS) + s * 32 + 4 * h;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + 8 * rg);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nb][4 * rg + e] = bv[e];
        }
    };

    // Matrix phase: 72 steps, step i = (tap i / 8, k-step (i % 8) / 2, sub-tile i % 2).  The pixel fragment of step
    // i + DIST is requested before the MFMA of step i, in exactly this order (the scheduler left to itself requests
    // each fragment right before its use and the wave then waits out the LDS latency 72 times per tile).
    auto mfma_phase = [&](int buf) {
        constexpr int DIST = 8, NBUF = DIST + 1;
        const char* pb = smem + buf * PBUF;
        // keep the per-tap addresses inside the loop (hoisted, they cost ~40 registers)
        asm volatile("" : "+v"(pq[0]), "+v"(pq[1]));
        bf16x8 fr[NBUF];
        int a1[2], y[2];
        auto request = [&](int i) {
            const int t = i >> 3, ks = (i & 7) >> 1, nb = i & 1;
            if (ks == 0) {
                a1[nb] = pq[nb] + ((t / 3 - 1) * PW + (t % 3 - 1)) * 128;      // 128 * patch pixel of this tap
                y[nb] = hb ^ ((a1[nb] >> 4) & 0x70);                           // 16 * (h ^ swizzle of that pixel)
            }
            fr[i % NBUF] = *reinterpret_cast<const bf16x8*>(pb + a1[nb] + (y[nb] ^ (ks << 5)));
        };
#pragma unroll
        for (int i = 0; i < DIST; ++i) request(i);
#pragma unroll
        for (int i = 0; i < 72; ++i) {
            if (i + DIST < 72) request(i + DIST);
            __builtin_amdgcn_sched_barrier(0);
            acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[i >> 3][(i & 7) >> 1], fr[i % NBUF], acc[i & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // sum of squares over this wave's 32 channels, published for the wave that holds the other 32
    auto publish_ss = [&](int seq) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaf(acc[nb][r], acc[nb][r], v);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            ssown[nb] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // lane l: v[l % 32] + v[l % 32 + 32]
        }
        lds_vfloat mx = (lds_vfloat)(smem + OFF_XCH) + (wave * 2 + (seq & 1)) * 64;
        if (lane < 32) { mx[l31] = ssown[0]; mx[32 + l31] = ssown[1]; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) ((lds_vint)(smem + OFF_FLAG))[wave] = seq + 1;
    };

    auto epilogue = [&](int t, int seq) {
        char* stg = smem + OFF_STG + wave * 4096;
        // staging address of this lane's 4 channels 8*rg + 4*h .. of pixel row nb*32 + l31: row*64 + ((rg ^ swz) << 4) + 8*h
        const int srow[2] = {l31 * 64 + 8 * h, (32 + l31) * 64 + 8 * h};
        const int swz = ((l31 >> 1) & 3) << 4;
        bf16x4 rv[2][4];
        if constexpr (RES) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) rv[nb][rg] = *reinterpret_cast<const bf16x4*>(stg + srow[nb] + ((rg << 4) ^ swz));
        }
        const float* par = reinterpret_cast<const float*>(smem + OFF_PAR) + (seq % 3) * 128 + s * 32 + 4 * h;
        lds_vint pf = (lds_vint)(smem + OFF_FLAG) + (wave ^ 1);
        while (__builtin_amdgcn_readfirstlane(*pf) < seq + 1) __builtin_amdgcn_s_sleep(1);
        lds_vfloat ox = (lds_vfloat)(smem + OFF_XCH) + ((wave ^ 1) * 2 + (seq & 1)) * 64;
        float rinv[2];
        // x / max(|x|, 1e-12) * sqrt(C)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) rinv[nb] = a.norm_scale * __builtin_amdgcn_rsqf(fmaxf(ssown[nb] + ox[nb * 32 + l31], 1e-24f));
        WS_ACC(1, ts);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(par + 8 * rg);
            const f32x4 sv = *reinterpret_cast<const f32x4*>(par + 64 + 8 * rg);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(fmaf(acc[nb][4 * rg + e] * rinv[nb], gv[e], sv[e]));
                if constexpr (RES) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rv[nb][rg][e];
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
                *reinterpret_cast<bf16x4*>(stg + srow[nb] + ((rg << 4) ^ swz)) = o;
            }
        }
        WS_ACC(5, ts);
        const int prow0 = wave_row0(t);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(stg + it * 1024 + lane * 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), rs_out, out_voff,
                                                       (unsigned)((prow0 + it * 16) * a.ldo * 2), 0);
        }
        WS_ACC(7, ts);
    };

    // Tile of this workgroup in round r of the persistent loop.  Workgroups are dealt to the 8 XCDs round-robin, so
    // within a full round the workgroups of one XCD (blockIdx.x % 8 equal) take a run of consecutive tiles: the bands
    // of a frame, whose halo rows they share, then meet in that XCD's L2 instead of being fetched by two XCDs.
    const int stride = gridDim.x;
    auto tile_of = [&](int r) -> int {
        const int base = r * stride;
        if ((stride & 7) == 0 && base + stride <= a.ntiles)
            return base + (int)(blockIdx.x & 7) * (stride >> 3) + (int)(blockIdx.x >> 3);
        return base + (int)blockIdx.x;
    };
    int tile = tile_of(0);
    __syncthreads();                       // bias / parameter / flag initialisation
    if (tile < a.ntiles) issue_patch(tile, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    ts = WS_T();
    int k = 0;
    for (int nxt; tile < a.ntiles; tile = nxt, ++k) {
        nxt = tile_of(k + 1);
        if (nxt < a.ntiles) issue_patch(nxt, (k + 1) & 1, (k + 1) % 3);
        WS_ACC(0, ts);
        if constexpr (RES) issue_res(tile);
        init_acc();
        WS_ACC(2, ts);
        mfma_phase(k & 1);
        WS_ACC(3, ts);
        publish_ss(k);
        WS_ACC(4, ts);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // patch of the next tile, residual of this one
        WS_ACC(4, ts);
        epilogue(tile, k);
        __builtin_amdgcn_s_barrier();
        WS_ACC(6, ts);
    }
#ifdef VDMI_DIAG_STAMPS
    if (a.diag && lane == 0)
        for (int i = 0; i < 8; ++i) a.diag[((size_t)blockIdx.x * 8 + wave) * 8 + i] = dsum[i];
#endif
}


int g_ws_enabled = -1;

}   // namespace

// Test / tool switch: 0 = never use the weight-stationary kernel, 1 = use it where it applies (default; the
// environment variable VDMI_WSCONV=0 turns it off for a whole process).
extern "C" void vdmi_wsconv_enable(int on) { g_ws_enabled = on ? 1 : 0; }

// Whether a tap-GEMM launch has the shape the weight-stationary kernel covers (and the kernel is switched on).
static bool ws_covers(const vdmi_tapgemm_args& A) {
    if (g_ws_enabled < 0) {
        const char* e = std::getenv("VDMI_WSCONV");
        g_ws_enabled = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_ws_enabled) return false;
    const vdmi_tap_pass& m = A.main;
    if (A.epilogue != VDMI_EPI_NORM_SILU || A.n != 64 || m.c0 != 64 || m.c1 != 0 || m.ntaps != 9 || m.stride != 1) return false;
    for (int t = 0; t < 9; ++t)
        if (m.dy[t] != t / 3 - 1 || m.dx[t] != t % 3 - 1) return false;
    if (m.w_group_stride || m.phases > 1 || m.space_to_depth) return false;
    if (A.has_res == 1) return false;
    if (A.out_mul > 1 || A.out_off_y || A.out_off_x) return false;
    const int H = m.in_h, W = m.in_w;
    if ((W != 64 && W != 32) || A.grid_h != H || A.grid_w != W || A.out_h != H || A.out_w != W) return false;
    const int TH = 256 / W;
    if (H % TH) return false;
    if (!A.gain || A.frames_per_batch <= 0 || A.frames % A.frames_per_batch) return false;
    const long long rows = (long long)A.frames * H * W;
    if (rows * 128 >= 0x7fffffffLL || rows * A.ldo * 2 >= 0x7fffffffLL) return false;
    if (A.has_res == 2 && ((long long)rows * A.ld_res * 2 >= 0x7fffffffLL || A.ld_res % 8 || !A.res_identity)) return false;
    if (!aligned16(m.src0) || !aligned16(m.w) || !aligned16(A.out) || A.ldo % 8) return false;
    const int nb = A.frames / A.frames_per_batch;
    if ((long long)nb * A.gain_bstride >= 0x1fffffffLL || (long long)nb * A.shift_bstride >= 0x1fffffffLL) return false;
    return true;
}

extern "C" int vdmi_wsconv_covers(const vdmi_tapgemm_args* a) { return a && ws_covers(*a) ? 1 : 0; }

// Runs `A` on the weight-stationary kernel when the launch has its shape.  Returns VDMI_OK when it was launched,
// 1 when the shape is not covered (the caller then takes the tap-GEMM), a negative status on a launch error.
int vdmi_wsconv_try(const vdmi_tapgemm_args& A, hipStream_t stream) {
    if (!ws_covers(A)) return 1;
    const vdmi_tap_pass& m = A.main;
    const int H = m.in_h, W = m.in_w;
    const int TH = 256 / W;
    const long long rows = (long long)A.frames * H * W;
    const int nb = A.frames / A.frames_per_batch;

    WsArgs k;
    k.in = static_cast<const __bf16*>(m.src0);
    k.w = static_cast<const __bf16*>(m.w);
    k.bias = A.bias;
    k.gain = A.gain;
    k.shift = A.shift;
    k.res = A.has_res == 2 ? static_cast<const __bf16*>(A.res_identity) : nullptr;
    k.out = static_cast<__bf16*>(A.out);
    k.gain_bs = A.gain_bstride;
    k.shift_bs = A.shift_bstride;
    k.ld_res = A.ld_res;
    k.ldo = A.ldo;
    k.H = H;
    k.W = W;
    k.F = A.frames_per_batch;
    k.bands = H / TH;
    k.ntiles = A.frames * k.bands;
    k.in_bytes = (unsigned)(rows * 128);
    k.res_bytes = k.res ? (unsigned)(rows * A.ld_res * 2) : 0u;
    k.out_bytes = (unsigned)(rows * A.ldo * 2);
    k.gain_bytes = (unsigned)(((long long)(nb - 1) * A.gain_bstride + 64) * 4);
    k.shift_bytes = A.shift ? (unsigned)(((long long)(nb - 1) * A.shift_bstride + 64) * 4) : 0u;
    k.norm_scale = A.norm_scale;
    k.diag = nullptr;
#ifdef VDMI_DIAG_STAMPS
    if (const char* dp = std::getenv("VDMI_DIAG_PTR")) k.diag = reinterpret_cast<unsigned long long*>(std::strtoull(dp, nullptr, 0));
#endif

    int dev = 0;
    static std::atomic<int> cu_count[64];             // per device, 0 until asked (zero-initialised)
    static std::atomic<uint64_t> attr_mask[4];        // per kernel variant, one bit per device
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VDMI_E_LAUNCH;
    int ncu = cu_count[dev].load(std::memory_order_relaxed);
    if (!ncu) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return VDMI_E_LAUNCH;
        cu_count[dev].store(ncu, std::memory_order_relaxed);
    }
    const int grid = k.ntiles < ncu ? k.ntiles : ncu;
    const int wi = (W == 64 ? 0 : 1) + (k.res ? 2 : 0);
    const int PW = W + 2, PH = TH + 2, ninstr = (PH * PW + 7) / 8;
    const size_t lds = (size_t)2 * ninstr * 1024 + 8 * 4096 + 4096 + 3 * 512 + 256 + 64;
    auto fn = W == 64 ? (k.res ? wsconv64_kernel<64, true> : wsconv64_kernel<64, false>)
                      : (k.res ? wsconv64_kernel<32, true> : wsconv64_kernel<32, false>);
    if (vdmi_set_max_lds(reinterpret_cast<const void*>(fn), attr_mask[wi], 160 * 1024)) return VDMI_E_LAUNCH;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, stream, k);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}
