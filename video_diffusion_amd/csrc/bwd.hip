// Backward pieces of ResnetBlock for gfx950 (reference: the gradients torch autograd derives for
// video_diffusion_pytorch/video_diffusion_pytorch.py:391-452 when the Trainer calls loss.backward(), :1451).
//
//   dgrad of a (1,k,k) convolution is the forward tap-GEMM over the output gradient with the weight transposed and
//   its taps reversed: vdmi_pack_taps_device writes that weight set (and the forward one) in the tap-GEMM's fragment
//   order on the device, so a training step never packs on the host.
//
//   wgrad contracts over pixels: dW[co][ci][tap] = sum_p dy[p][co] * x[p + off(tap)][ci].  With channels-last rows
//   the contraction index is the slow one, and an MFMA operand wants 8 consecutive contraction values per lane.  The
//   rows are therefore transposed once into [frame block][channel / 32][pixel][channel % 32][16 frames]
//   (vdmi_rows_transpose): the FRAME is the fast contraction index, and the fragment a wave needs for one pixel - 32
//   channels x 16 frames - is one contiguous KiB.  A tap moves (y, x) and never the frame, so the fragment of a shifted pixel is again 16
//   aligned bytes - no unaligned reads, and one 32 x 32 accumulator per tap.  The 1x1 form streams its fragments
//   straight from L1/L2; the 3x3 form stages rows through LDS (see wgrad3x3_lds_kernel).  Workgroups split the
//   pixels of a (co, ci) tile, write partial tiles and a second kernel adds them (floating-point atomics to the
//   same few addresses from a thousand workgroups cost a millisecond per layer).
//
//   The RMSNorm / scale-shift / SiLU backward is a row kernel (two row reductions, per-channel sums by atomics).
#include "vdmi_common.h"

#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------------------------
// weight packing on the device: the layout of vdmi_pack_taps (host_utils.cpp), one 16-byte piece per thread
// ------------------------------------------------------------------------------------------------------------------
// `in_scale` (float [cin0 + cin1] of the FORWARD weight's input channels, or null) multiplies every element by the
// factor of its input channel: a LayerNorm gamma folded into the projection that follows it
// `gather` (int32 [n][cin0 + cin1][ntaps], or null): the weight set is a re-arrangement of a parameter - element e of the
// set's source tensor is w[gather[e]] (the space-to-depth and per-phase views of the resampling convolutions' kernels)
__device__ __forceinline__ void pack_taps_piece(const float* __restrict__ w, const float* __restrict__ in_scale,
                                                const int* __restrict__ gather, int n, int npad,
                                                int cin0, int cin1, int ntaps, int bkc, int transposed,
                                                __bf16* __restrict__ out, long long o8) {
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
            // (this pass's output channel co is the forward weight's input channel)
            const size_t e = transposed ? ((size_t)(coff + ci) * n + co) * ntaps + (ntaps - 1 - t)
                                        : ((size_t)co * cin + coff + ci) * ntaps + t;
            f = w[gather ? (size_t)gather[e] : e];
            if (in_scale) f *= transposed ? in_scale[co] : in_scale[coff + ci];
        }
        v[j] = f2bf(f);
    }
    *reinterpret_cast<bf16x8*>(out + o8 * 8) = v;
}

__global__ __launch_bounds__(256) void pack_taps_kernel(const float* __restrict__ w, int n, int npad, int cin0, int cin1,
                                                        int ntaps, int bkc, int transposed, __bf16* __restrict__ out,
                                                        long long pieces) {
    const long long o8 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o8 >= pieces) return;
    pack_taps_piece(w, nullptr, nullptr, n, npad, cin0, cin1, ntaps, bkc, transposed, out, o8);
}

// many weight sets in one launch: the job table (device copy of vdmi_pack_job records) is sorted by first workgroup
struct PackJobDev {
    const float* w;
    const float* in_scale;
    const int* gather;
    __bf16* out;
    long long pieces, block_begin;
    int n, npad, cin0, cin1, ntaps, bkc, transposed, pad_;
};
static_assert(sizeof(PackJobDev) == 80, "job record");

__global__ __launch_bounds__(256) void pack_taps_batch_kernel(const PackJobDev* __restrict__ jobs, int count) {
    int lo = 0, hi = count - 1;                 // the last job whose first workgroup is <= this one
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_begin <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJobDev j = jobs[lo];
    const long long o8 = ((long long)blockIdx.x - j.block_begin) * 256 + threadIdx.x;
    if (o8 >= j.pieces) return;
    pack_taps_piece(j.w, j.in_scale, j.gather, j.n, j.npad, j.cin0, j.cin1, j.ntaps, j.bkc, j.transposed, j.out, o8);
}

// ------------------------------------------------------------------------------------------------------------------
// rows [nf*hw][ld] (channel window c_off .. c_off + c) -> T[frame block][channel block][pixel][32][16]; frames beyond nf and
// channels beyond c (up to the end of the last block) are zero
// ------------------------------------------------------------------------------------------------------------------
constexpr int TP = 8;   // pixels per tile

// element (channel c, pixel p, frame f) of a transposed tensor with `cb` channel blocks of 32
__device__ __forceinline__ long long t_off(int c, long long p, int fb, int cb, long long hw) {
    return ((((long long)fb * cb + (c >> 5)) * hw + p) * 32 + (c & 31)) * 16;
}

__global__ __launch_bounds__(256) void rows_transpose_kernel(const __bf16* __restrict__ rows, int ld, int c_off, int c,
                                                             int nf, int hw, __bf16* __restrict__ out) {
    __shared__ __bf16 tile[16][TP][72];
    const int p0 = blockIdx.x * TP, c0 = blockIdx.y * 64, fb = blockIdx.z, f0 = fb * 16, tid = threadIdx.x;
    const int cb = (c + 31) / 32;
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
    // store: 2 channel blocks x TP pixels x 32 channels x 2 pieces of 8 frames; a (block, pixel) is one contiguous KiB
    for (int i = tid; i < 64 * TP * 2; i += 256) {
        const int fh = i & 1, cl = (i >> 1) & 31, p = (i >> 6) % TP, cbl = i / (64 * TP);
        const int ch = cbl * 32 + cl;
        if (c0 + cbl * 32 >= cb * 32 || p0 + p >= hw) continue;      // channels up to the block's end are written (zeros)
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[fh * 8 + j][p][ch];
        *reinterpret_cast<bf16x8*>(out + t_off(c0 + ch, p0 + p, fb, cb, hw) + fh * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad: dW[co][cin_off + ci][tap] = sum over (pixel, frame) of dyT[co][p][f] * xT[ci][p + off(tap)][f]
// A workgroup = 4 waves = a 64 x 64 (co, ci) tile; blockIdx.z = a share of the (frame block, pixel) units.
// ------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
    const __bf16* xT;
    const __bf16* dyT;
    float* part;                 // partial tiles [gridDim.z][tiles][NT][64][64]
    int cin, cout, cin_total, cin_off;
    int h, w, nfpad;
    long long units, per;        // (frame blocks of 16) * h * w, units per blockIdx.z
    int dy[9], dx[9];
};

// One tap.  A wave's only work per unit is two fragment loads and one MFMA, so what bounds it is how many loads it has
// in flight: units are taken eight at a time, the next eight requested before the current eight multiply (one unit
// ahead: 70 TFLOP/s, the latency of a load pair per MFMA).
constexpr int W1_U = 8;
__global__ __launch_bounds__(256) void wgrad1_kernel(const WgradArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int co = (blockIdx.x * 2 + (wave & 1)) * 32 + l31;      // this lane's A row
    const int ci = (blockIdx.y * 2 + (wave >> 1)) * 32 + l31;     // this lane's B column
    const int hw = a.h * a.w;
    const bool co_ok = co < a.cout, ci_ok = ci < a.cin;
    const int cbo = (a.cout + 31) / 32, cbi = (a.cin + 31) / 32;
    const __bf16* ap = a.dyT + t_off(co_ok ? co : 0, 0, 0, cbo, hw) + 8 * half;
    const __bf16* bp = a.xT + t_off(ci_ok ? ci : 0, 0, 0, cbi, hw) + 8 * half;
    const long long fbs_a = (long long)cbo * hw * 512, fbs_b = (long long)cbi * hw * 512;    // elements per frame block
    const int oy = a.dy[0], ox = a.dx[0];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long u_begin = (long long)blockIdx.z * a.per;
    const long long u_end = u_begin + a.per < a.units ? u_begin + a.per : a.units;
    // position of the next unit to be requested
    long long un = u_begin;
    int fb = (int)(un / hw), p = (int)(un - (long long)fb * hw);
    int y = p / a.w, x = p - y * a.w;
    auto load_batch = [&](bf16x8 (&A)[W1_U], bf16x8 (&B)[W1_U]) {
#pragma unroll
        for (int j = 0; j < W1_U; ++j) {
            bf16x8 va = zero8, vb = zero8;
            if (un < u_end) {
                if (co_ok) va = *reinterpret_cast<const bf16x8*>(ap + (long long)p * 512 + fb * fbs_a);
                const int yy = y + oy, xx = x + ox;
                if (ci_ok && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w)
                    vb = *reinterpret_cast<const bf16x8*>(bp + (long long)(yy * a.w + xx) * 512 + fb * fbs_b);
            }
            A[j] = va;
            B[j] = vb;
            ++un; ++p; ++x;
            if (x == a.w) { x = 0; ++y; }
            if (p == hw) { p = 0; y = 0; ++fb; }
        }
    };
    bf16x8 A0[W1_U], B0[W1_U], A1[W1_U], B1[W1_U];
    if (u_begin < u_end) {
        load_batch(A0, B0);
        for (long long u = u_begin; u < u_end; u += 2 * W1_U) {
            load_batch(A1, B1);
#pragma unroll
            for (int j = 0; j < W1_U; j += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[j], B0[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[j + 1], B0[j + 1], acc1, 0, 0, 0);
            }
            load_batch(A0, B0);
#pragma unroll
            for (int j = 0; j < W1_U; j += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[j], B1[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[j + 1], B1[j + 1], acc1, 0, 0, 0);
            }
        }
    }
    // partial tile of this workgroup: [64 rows][64 columns]; register r of this lane is row
    // (r&3) + 8*(r>>2) + 4*half of the wave's 32, column l31
    float* part = a.part + ((size_t)blockIdx.z * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        part[((wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + (wave >> 1) * 32 + l31] = acc0[r] + acc1[r];
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1 convolution / Linear straight from the channels-last rows (no transposed copy):
//   dW[co][ci] = sum over rows r of dy[r][co] * x[r][ci]
// The contraction index (the row) is the slow one in memory.  A workgroup copies 64 rows x 64 channels of each operand
// into LDS as they lie (16-byte pieces, coalesced) and the waves read their MFMA fragments with the transposing LDS
// read of gfx950 (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns and each lane receives one column's 4
// values), two reads per operand and k-step.  LDS rows are 192 bytes apart: the 4 rows x 2 column blocks a 32-lane
// half touches then fall into 64 different banks.  Stages are double buffered with the next stage's global loads in
// flight during the current stage's MFMAs; one barrier per stage.
// ------------------------------------------------------------------------------------------------------------------
constexpr int RW_KC = 64;               // rows per stage
constexpr int RW_PITCH = 192;           // bytes per LDS row (64 channels = 128 bytes, + 64)
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

struct RowsWgradArgs {
    const __bf16* x;
    const __bf16* dy;
    float* part;                 // partial tiles [gridDim.z][tiles][64][64]
    int ldx, lddy, cin, cout;
    long long rows, per;         // rows in all, rows per blockIdx.z (a multiple of RW_KC)
};

__device__ __forceinline__ bf16x8 lds_tr_fragment(const char* base) {
    // rows base .. base + 3 and base + 4 .. base + 7 of this lane's column: the 8 consecutive contraction values
    typedef __attribute__((address_space(3))) s4v* lds_s4;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base + 4 * RW_PITCH));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256) void wgrad_rows_kernel(const RowsWgradArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][RW_KC * RW_PITCH];      // [stage buffer][dy | x]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
    const long long r_begin = (long long)blockIdx.z * a.per;
    const long long r_end = r_begin + a.per < a.rows ? r_begin + a.per : a.rows;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // staging: piece q = tid + 256 i of a tile is (row q >> 3, channels 8 (q & 7) ..)
    const int prow = tid >> 3, pc = tid & 7;
    const bool a_ok = co0 + pc * 8 < a.cout, b_ok = ci0 + pc * 8 < a.cin;
    bf16x8 va[2], vb[2];
    auto load = [&](long long r0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long r = r0 + prow + 32 * i;
            va[i] = (a_ok && r < r_end) ? *reinterpret_cast<const bf16x8*>(a.dy + r * a.lddy + co0 + pc * 8) : zero8;
            vb[i] = (b_ok && r < r_end) ? *reinterpret_cast<const bf16x8*>(a.x + r * a.ldx + ci0 + pc * 8) : zero8;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<bf16x8*>(&lds[buf][0][(prow + 32 * i) * RW_PITCH + pc * 16]) = va[i];
            *reinterpret_cast<bf16x8*>(&lds[buf][1][(prow + 32 * i) * RW_PITCH + pc * 16]) = vb[i];
        }
    };
    // fragment addresses: lane 4q + p of a 16-lane group g supplies row q, columns 4p .. 4p + 3 of the group's block;
    // groups 0 / 1 are the operand's columns 0-15 / 16-31 at contraction values 0-7, groups 2 / 3 the same at 8-15
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int frag_row = 8 * (g >> 1) + q, frag_col = 16 * (g & 1) + 4 * pp;
    const int a_off = frag_row * RW_PITCH + ((wave & 1) * 32 + frag_col) * 2;
    const int b_off = frag_row * RW_PITCH + ((wave >> 1) * 32 + frag_col) * 2;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    int cur = 0;
    if (r_begin < r_end) {
        load(r_begin);
        store(0);
        __syncthreads();
        for (long long r0 = r_begin; r0 < r_end; r0 += RW_KC) {
            const bool more = r0 + RW_KC < r_end;
            if (more) load(r0 + RW_KC);
            const char* ta = &lds[cur][0][a_off];
            const char* tb = &lds[cur][1][b_off];
#pragma unroll
            for (int ks = 0; ks < RW_KC / 16; ks += 2) {
                const bf16x8 A0 = lds_tr_fragment(ta + ks * 16 * RW_PITCH), B0 = lds_tr_fragment(tb + ks * 16 * RW_PITCH);
                const bf16x8 A1 = lds_tr_fragment(ta + (ks + 1) * 16 * RW_PITCH), B1 = lds_tr_fragment(tb + (ks + 1) * 16 * RW_PITCH);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc1, 0, 0, 0);
            }
            if (more) store(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    float* part = a.part + ((size_t)blockIdx.z * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        part[((wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + (wave >> 1) * 32 + l31] = acc0[r] + acc1[r];
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the (1,3,3) convolution straight from the channels-last rows (no transposed copies):
//   dW[co][ci][ky][kx] = sum over frames and pixels p of dy[p][co] * x[p + (ky - 1, kx - 1)][ci]
// The contraction runs over 16 consecutive pixels of an image row.  A workgroup stages a tile of TY image rows x XT
// pixels of dy and the same tile of x with a one-pixel halo into LDS as the rows lie in memory (16-byte pieces), and
// the waves read their fragments with the transposing LDS read, as wgrad_rows_kernel does: a tap is nothing but an
// offset of (ky * (XT + 2) + kx) LDS rows on the x operand's address.  A wave walks down the TY rows of a 16-pixel
// column and keeps the two upper rows of the 3 x 3 window in registers: three new x fragments and one dy fragment per
// 16 pixels feed nine MFMAs.  Pixels outside the image are zeros in LDS.  Partial tiles in the format of
// wgrad3x3_lds_kernel ([tap][64 co][64 ci] per workgroup), added by wgrad_reduce_kernel.
// ------------------------------------------------------------------------------------------------------------------
struct Wgrad3RowsArgs {
    const __bf16* x;
    const __bf16* dy;
    float* part;
    int ldx, lddy, cin, cout, h, w, strips, ytiles;
    long long units, per;        // units: (frame, strip, row tile) triples, row tile fastest; units per blockIdx.z
};

template <int XT, int TY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad3x3_rows_kernel(const Wgrad3RowsArgs a) {
    constexpr int HX = XT + 2, XPX = HX * (TY + 2), DPX = XT * TY, P = RW_PITCH;
    constexpr int XN = (XPX * 8 + 255) / 256, DN = DPX * 8 / 256;
    static_assert(DPX * 8 % 256 == 0 && XT % 16 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xl = smem;                              // [(TY + 2) x (XT + 2) pixels][P]
    char* dl = smem + XPX * P;                    // [TY x XT pixels][P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
    const long long u_begin = (long long)blockIdx.z * a.per;
    const long long u_end = u_begin + a.per < a.units ? u_begin + a.per : a.units;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // staging: piece tid + 256 k of a tile is (pixel (tid >> 3) + 32 k, channels 8 (tid & 7) ..)
    const int ppx = tid >> 3, pc = tid & 7;
    const bool a_ok = co0 + pc * 8 < a.cout, b_ok = ci0 + pc * 8 < a.cin;
    bf16x8 xv[XN], dv[DN];
    auto load = [&](long long u) {
        const int yt = (int)(u % a.ytiles);
        const long long t = u / a.ytiles;
        const int strip = (int)(t % a.strips);
        const long long frame = t / a.strips;
        const int x0 = strip * XT, y0 = yt * TY;
        const __bf16* xb = a.x + frame * a.h * a.w * a.ldx + ci0 + pc * 8;
        const __bf16* db = a.dy + frame * a.h * a.w * a.lddy + co0 + pc * 8;
#pragma unroll
        for (int k = 0; k < XN; ++k) {
            const int p = ppx + 32 * k, hy = p / HX, hx = p % HX;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = b_ok && p < XPX && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
            xv[k] = ok ? *reinterpret_cast<const bf16x8*>(xb + ((long long)gy * a.w + gx) * a.ldx) : zero8;
        }
#pragma unroll
        for (int k = 0; k < DN; ++k) {
            const int p = ppx + 32 * k, ty = p / XT, tx = p % XT;
            const int gy = y0 + ty, gx = x0 + tx;
            const bool ok = a_ok && gy < a.h && gx < a.w;
            dv[k] = ok ? *reinterpret_cast<const bf16x8*>(db + ((long long)gy * a.w + gx) * a.lddy) : zero8;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int k = 0; k < XN; ++k) {
            const int p = ppx + 32 * k;
            if (p < XPX) *reinterpret_cast<bf16x8*>(xl + p * P + pc * 16) = xv[k];
        }
#pragma unroll
        for (int k = 0; k < DN; ++k) *reinterpret_cast<bf16x8*>(dl + (ppx + 32 * k) * P + pc * 16) = dv[k];
    };
    // fragment addresses as in wgrad_rows_kernel: an LDS row is a pixel, 16 consecutive rows one contraction step
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int frag_row = 8 * (g >> 1) + q, frag_col = 16 * (g & 1) + 4 * pp;
    const char* fa = dl + frag_row * P + (wm * 32 + frag_col) * 2;
    const char* fb = xl + frag_row * P + (wn * 32 + frag_col) * 2;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (u_begin < u_end) {
        load(u_begin);
        for (long long u = u_begin; u < u_end; ++u) {
            store();
            __syncthreads();
            if (u + 1 < u_end) load(u + 1);        // in flight during this tile's products
#pragma unroll
            for (int cx = 0; cx < XT / 16; ++cx) {
                bf16x8 c[3][3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    c[0][kx] = lds_tr_fragment(fb + (0 * HX + cx * 16 + kx) * P);
                    c[1][kx] = lds_tr_fragment(fb + (1 * HX + cx * 16 + kx) * P);
                }
#pragma unroll
                for (int ty = 0; ty < TY; ++ty) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) c[2][kx] = lds_tr_fragment(fb + ((ty + 2) * HX + cx * 16 + kx) * P);
                    const bf16x8 A = lds_tr_fragment(fa + (ty * XT + cx * 16) * P);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, c[ky][kx], acc[3 * ky + kx], 0, 0, 0);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) { c[0][kx] = c[1][kx]; c[1][kx] = c[2][kx]; }
                }
            }
            __syncthreads();                       // the next tile overwrites both buffers
        }
    }
    float* part = a.part + ((size_t)blockIdx.z * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * (9 * 4096);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            part[(t * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + wn * 32 + l31] = acc[t][r];
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the (1,4,4) / stride 2 / pad 1 convolution (Downsample, vdp.py:377-378; with the roles of input and
// output gradient swapped, of the transposed convolution of Upsample, vdp.py:374-375) straight from the channels-last rows:
//   dW[cs][cb][ky][kx] = sum over frames and small-image pixels q of small[q][cs] * big[2 q + (ky - 1, kx - 1)][cb]
// Same scheme as wgrad3x3_rows_kernel: the contraction runs over 16 consecutive pixels of a small-image row, the operands
// are staged as the rows lie and read with the transposing LDS read; the big image's fragment takes every second LDS row
// (a lane's address is free), a tap is an offset of ky * (2 XT + 2) + kx rows.  Sixteen taps x a 32 x 32 block would be
// 256 accumulators, so the workgroup's tile is 64 (cs) x 32 (cb) and its waves are (cs half) x (ky parity): a wave keeps
// the taps ky in {kyp, kyp + 2}, kx in 0..3 (128 accumulators) and, walking down the rows of a 16-pixel column, the
// fragments of tap row kyp + 2 become those of tap row kyp one step later: four new big fragments and one small
// fragment per eight MFMAs.  The big tile's rows are 96 bytes apart (64 bytes of channels + 32): rows two apart then
// fall 48 banks apart and a half-wave's four rows x 64 bytes cover all 64 banks once.
// Partial tiles [16 taps][64 cs][32 cb] per workgroup, added in a fixed order by wgrad4_reduce_kernel.
// ------------------------------------------------------------------------------------------------------------------
constexpr int W4_XT = 16, W4_TY = 6, W4_PB = 96;
constexpr int W4_HXB = 2 * W4_XT + 2, W4_BPX = W4_HXB * (2 * W4_TY + 2), W4_SPX = W4_XT * W4_TY;
constexpr int W4_LDS = W4_BPX * W4_PB + W4_SPX * RW_PITCH;

struct Wgrad4Args {
    const __bf16* big;
    const __bf16* small;
    float* part;
    int ldb, lds, cb, cs, h, w, strips, ytiles;      // h x w: the big image
    long long units, per;                            // units: (frame, strip, row tile) triples, row tile fastest
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad4x4s2_rows_kernel(const Wgrad4Args a) {
    constexpr int XT = W4_XT, TY = W4_TY, HXB = W4_HXB, BPX = W4_BPX, SPX = W4_SPX, PB = W4_PB, PS = RW_PITCH;
    constexpr int BN = (BPX * 4 + 255) / 256, SN = SPX * 8 / 256;
    static_assert(SPX * 8 % 256 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* bl = smem;                              // [(2 TY + 2) x (2 XT + 2) big pixels][PB]: 32 channels
    char* sl = smem + BPX * PB;                   // [TY x XT small pixels][PS]: 64 channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 1, kyp = wave >> 1;
    const int cs0 = blockIdx.x * 64, cb0 = blockIdx.y * 32;
    const int oh = a.h >> 1, ow = a.w >> 1;
    const long long u_begin = (long long)blockIdx.z * a.per;
    const long long u_end = u_begin + a.per < a.units ? u_begin + a.per : a.units;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 bv[BN], sv[SN];
    auto load = [&](long long u) {
        const int yt = (int)(u % a.ytiles);
        const long long t = u / a.ytiles;
        const int strip = (int)(t % a.strips);
        const long long frame = t / a.strips;
        const int x0 = strip * XT, y0 = yt * TY;
        const __bf16* bb = a.big + frame * a.h * a.w * a.ldb + cb0 + (tid & 3) * 8;
        const __bf16* sb = a.small + frame * oh * ow * a.lds + cs0 + (tid & 7) * 8;
        const bool b_ok = cb0 + (tid & 3) * 8 < a.cb, s_ok = cs0 + (tid & 7) * 8 < a.cs;
#pragma unroll
        for (int k = 0; k < BN; ++k) {
            const int p = (tid >> 2) + 64 * k, hy = p / HXB, hx = p % HXB;
            const int gy = 2 * y0 - 1 + hy, gx = 2 * x0 - 1 + hx;
            const bool ok = b_ok && p < BPX && gy >= 0 && gy < a.h && gx >= 0 && gx < a.w;
            bv[k] = ok ? *reinterpret_cast<const bf16x8*>(bb + ((long long)gy * a.w + gx) * a.ldb) : zero8;
        }
#pragma unroll
        for (int k = 0; k < SN; ++k) {
            const int p = (tid >> 3) + 32 * k, ty = p / XT, tx = p % XT;
            const int gy = y0 + ty, gx = x0 + tx;
            const bool ok = s_ok && gy < oh && gx < ow;
            sv[k] = ok ? *reinterpret_cast<const bf16x8*>(sb + ((long long)gy * ow + gx) * a.lds) : zero8;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int k = 0; k < BN; ++k) {
            const int p = (tid >> 2) + 64 * k;
            if (p < BPX) *reinterpret_cast<bf16x8*>(bl + p * PB + (tid & 3) * 16) = bv[k];
        }
#pragma unroll
        for (int k = 0; k < SN; ++k) *reinterpret_cast<bf16x8*>(sl + ((tid >> 3) + 32 * k) * PS + (tid & 7) * 16) = sv[k];
    };
    // fragment addresses: an LDS row is a pixel; the small operand's contraction step is 16 consecutive rows, the big
    // operand's 16 rows two apart
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int frag_row = 8 * (g >> 1) + q, frag_col = 16 * (g & 1) + 4 * pp;
    const char* fa = sl + frag_row * PS + (wm * 32 + frag_col) * 2;
    const char* fb = bl + 2 * frag_row * PB + frag_col * 2;
    typedef __attribute__((address_space(3))) s4v* lds_s4;
    auto big_fragment = [&](const char* base) -> bf16x8 {
        const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base));
        const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base + 8 * PB));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[8];                                // [ky in {kyp, kyp + 2}][kx]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (u_begin < u_end) {
        load(u_begin);
        for (long long u = u_begin; u < u_end; ++u) {
            store();
            __syncthreads();
            if (u + 1 < u_end) load(u + 1);        // in flight during this tile's products
            bf16x8 c[2][4];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) c[0][kx] = big_fragment(fb + (kyp * HXB + kx) * PB);
#pragma unroll
            for (int ty = 0; ty < TY; ++ty) {
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) c[1][kx] = big_fragment(fb + ((2 * ty + kyp + 2) * HXB + kx) * PB);
                const bf16x8 A = lds_tr_fragment(fa + ty * XT * PS);
#pragma unroll
                for (int kl = 0; kl < 2; ++kl)
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx)
                        acc[4 * kl + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, c[kl][kx], acc[4 * kl + kx], 0, 0, 0);
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) c[0][kx] = c[1][kx];
            }
            __syncthreads();                       // the next tile overwrites both buffers
        }
    }
    float* part = a.part + ((size_t)blockIdx.z * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * (16 * 2048);
#pragma unroll
    for (int kl = 0; kl < 2; ++kl)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                part[(((kyp + 2 * kl) * 4 + kx) * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[4 * kl + kx][r];
}

// dw[cs][cb][16] = sum over the nz partial sets of part[z][tile (cb tile slow)][tap][64][32], in order
__global__ __launch_bounds__(256) void wgrad4_reduce_kernel(const float* __restrict__ part, int nz, int tx, int ty,
                                                            float* __restrict__ dw, int cs, int cb) {
    const size_t per_z = (size_t)tx * ty * 16 * 2048;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= per_z) return;
    const int c = (int)(e & 31), r = (int)((e >> 5) & 63), t = (int)((e >> 11) & 15);
    const int tile = (int)(e >> 15), txi = tile % tx, tyi = tile / tx;
    const int co = txi * 64 + r, ci = tyi * 32 + c;
    if (co >= cs || ci >= cb) return;
    float sum = 0.f;
#pragma unroll 8
    for (int z = 0; z < nz; ++z) sum += part[(size_t)z * per_z + e];
    dw[((size_t)co * cb + ci) * 16 + t] = sum;
}

// The same product with a 128 x 128 tile per workgroup (eight waves: 2 x 2 blocks of 64 x 64 output channels, times two
// halves of a stage's 64 rows).  A 64 x 64 tile makes the launch read the rows of x once per 64 output channels and the
// rows of dy once per 64 input channels through L2 - three to six times the operands' size on the layers of a training
// step, which is what bounds the small tile at ~2.5 TB/s of algorithmic traffic; here it is half of that, every wave
// has eight MFMAs between two barriers instead of four, and one fragment read per MFMA instead of two.  The two row
// halves are added through LDS before the partial tiles (the 64 x 64 format of wgrad_reduce_kernel) are written.
constexpr int RW2_PITCH = 320;          // bytes per LDS row: 128 channels = 256 bytes, + 64

__device__ __forceinline__ bf16x8 lds_tr_fragment2(const char* base) {
    typedef __attribute__((address_space(3))) s4v* lds_s4;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base + 4 * RW2_PITCH));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(512) void wgrad_rows128_kernel(const RowsWgradArgs a, int tx, int ty) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][RW_KC * RW2_PITCH];     // [stage buffer][dy | x]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 1, wn = (wave >> 1) & 1, wk = wave >> 2;
    const int co0 = blockIdx.x * 128, ci0 = blockIdx.y * 128;
    const long long r_begin = (long long)blockIdx.z * a.per;
    const long long r_end = r_begin + a.per < a.rows ? r_begin + a.per : a.rows;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // staging: piece q = tid + 512 i of a tile is (row q >> 4, channels 8 (q & 15) ..)
    const int prow = tid >> 4, pc = tid & 15;
    const bool a_ok = co0 + pc * 8 < a.cout, b_ok = ci0 + pc * 8 < a.cin;
    bf16x8 va[2], vb[2];
    auto load = [&](long long r0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long r = r0 + prow + 32 * i;
            va[i] = (a_ok && r < r_end) ? *reinterpret_cast<const bf16x8*>(a.dy + r * a.lddy + co0 + pc * 8) : zero8;
            vb[i] = (b_ok && r < r_end) ? *reinterpret_cast<const bf16x8*>(a.x + r * a.ldx + ci0 + pc * 8) : zero8;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<bf16x8*>(&lds[buf][0][(prow + 32 * i) * RW2_PITCH + pc * 16]) = va[i];
            *reinterpret_cast<bf16x8*>(&lds[buf][1][(prow + 32 * i) * RW2_PITCH + pc * 16]) = vb[i];
        }
    };
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int frag_row = 8 * (g >> 1) + q, frag_col = 16 * (g & 1) + 4 * pp;
    const int a_off = (wk * 32 + frag_row) * RW2_PITCH + (wm * 64 + frag_col) * 2;
    const int b_off = (wk * 32 + frag_row) * RW2_PITCH + (wn * 64 + frag_col) * 2;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int cur = 0;
    if (r_begin < r_end) {
        load(r_begin);
        store(0);
        __syncthreads();
        for (long long r0 = r_begin; r0 < r_end; r0 += RW_KC) {
            const bool more = r0 + RW_KC < r_end;
            if (more) load(r0 + RW_KC);
            const char* ta = &lds[cur][0][a_off];
            const char* tb = &lds[cur][1][b_off];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {              // this wave's two k-steps (16 rows each) of the stage
                const bf16x8 A0 = lds_tr_fragment2(ta + ks * 16 * RW2_PITCH), A1 = lds_tr_fragment2(ta + ks * 16 * RW2_PITCH + 64);
                const bf16x8 B0 = lds_tr_fragment2(tb + ks * 16 * RW2_PITCH), B1 = lds_tr_fragment2(tb + ks * 16 * RW2_PITCH + 64);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[1][1], 0, 0, 0);
            }
            if (more) store(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    // the second row half hands its sums to the first through LDS (64 KiB of the stage buffers), which writes the tiles
    float* xch = reinterpret_cast<float*>(&lds[0][0][0]) + (wm + 2 * wn) * 4096;
    __syncthreads();
    if (wk == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wk == 0) {
        const int t_x = blockIdx.x * 2 + wm, t_y = blockIdx.y * 2 + wn;       // this wave's 64 x 64 tile
        if (t_x < tx && t_y < ty) {
            float* part = a.part + ((size_t)blockIdx.z * ((size_t)tx * ty) + (size_t)t_y * tx + t_x) * 4096;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        part[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + j * 32 + l31] =
                            acc[i][j][r] + xch[((i * 2 + j) * 16 + r) * 64 + lane];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad of a (1,kh,kw) convolution with few input channels (init_conv: 3 -> dim, 7 x 7; torch's weight gradient,
// vdp.py:644).  A tap-per-accumulator kernel would idle 29 of 32 columns, so here a column IS a (ci, ky, kx) triple:
// n = (ci * kh + ky) * kw + kx, exactly the flattened weight index, and each lane gathers the fragment of its own
// shifted pixel.  Same partial-tile output as wgrad_kernel<1> over cin * kh * kw "channels".
// ------------------------------------------------------------------------------------------------------------------
struct WgradGatherArgs {
    const __bf16* xT;
    const __bf16* dyT;
    float* part;
    int cin, cout, kh, kw;
    int h, w, nfpad;
    long long units, per;
};

__global__ __launch_bounds__(256) void wgrad_gather_kernel(const WgradGatherArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int co = (blockIdx.x * 2 + (wave & 1)) * 32 + l31;
    const int n = (blockIdx.y * 2 + (wave >> 1)) * 32 + l31;
    const int hw = a.h * a.w, ntaps = a.kh * a.kw;
    const bool co_ok = co < a.cout, n_ok = n < a.cin * ntaps;
    const int ci = n_ok ? n / ntaps : 0, t = n_ok ? n - ci * ntaps : 0;
    const int ody = t / a.kw - a.kh / 2, odx = t % a.kw - a.kw / 2;
    const int cbo = (a.cout + 31) / 32, cbi = (a.cin + 31) / 32;
    const __bf16* ap = a.dyT + t_off(co_ok ? co : 0, 0, 0, cbo, hw) + 8 * half;
    const __bf16* bp = a.xT + t_off(ci, 0, 0, cbi, hw) + 8 * half;
    const long long fbs_a = (long long)cbo * hw * 512, fbs_b = (long long)cbi * hw * 512;
    f32x16 acc0, acc1;      // even / odd units: neighbouring MFMAs never share an accumulator
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    long long u = (long long)blockIdx.z * a.per;
    const long long u_end = u + a.per < a.units ? u + a.per : a.units;
    if (u < u_end) {
        int fb = (int)(u / hw), p = (int)(u - (long long)fb * hw);
        int y = p / a.w, x = p - y * a.w;
        auto load_set = [&](bf16x8& A, bf16x8& B, int fb_, int p_, int y_, int x_) {
            A = co_ok ? *reinterpret_cast<const bf16x8*>(ap + (long long)p_ * 512 + fb_ * fbs_a) : zero8;
            const int yy = y_ + ody, xx = x_ + odx;
            const bool ok = n_ok && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
            B = ok ? *reinterpret_cast<const bf16x8*>(bp + (long long)(yy * a.w + xx) * 512 + fb_ * fbs_b) : zero8;
        };
        auto advance = [&]() {
            ++p; ++x;
            if (x == a.w) { x = 0; ++y; }
            if (p == hw) { p = 0; y = 0; ++fb; }
        };
        for (; u + 1 < u_end; u += 2) {
            bf16x8 A0, B0, A1, B1;
            load_set(A0, B0, fb, p, y, x);
            advance();
            load_set(A1, B1, fb, p, y, x);
            advance();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc1, 0, 0, 0);
        }
        if (u < u_end) {
            bf16x8 A0, B0;
            load_set(A0, B0, fb, p, y, x);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc0, 0, 0, 0);
        }
    }
    float* part = a.part + ((size_t)blockIdx.z * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        part[((wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + (wave >> 1) * 32 + l31] = acc0[r] + acc1[r];
}

// dw[co][cin_off + ci][t] = sum over the nz partial tiles of a (co, ci) tile (one thread per element: deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nz, int tx, int ty, int nt,
                                                           float* __restrict__ dw, int cout, int cin, int cin_total,
                                                           int cin_off) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per_z = (long long)tx * ty * nt * 4096;
    if (e >= per_z) return;
    const int cl = (int)(e & 63), rl = (int)((e >> 6) & 63);
    const int t = (int)((e >> 12) % nt), tile = (int)((e >> 12) / nt);
    const int bx = tile % tx, by = tile / tx;
    const int co = bx * 64 + rl, ci = by * 64 + cl;
    if (co >= cout || ci >= cin) return;
    float sum = 0.f;
#pragma unroll 8
    for (int z = 0; z < nz; ++z) sum += part[(size_t)z * per_z + e];      // eight loads in flight, added in order
    dw[((size_t)co * cin_total + cin_off + ci) * nt + t] = sum;
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad of the (1,3,3) convolution, staged through LDS.  The streaming kernel above asks the memory pipeline for 64
// scattered 16-byte pieces per fragment (one per channel: the channel stride is a whole image), ten times per pixel,
// and spends its time there.  Here a workgroup walks down a column strip of XT pixels: per image row it copies the
// input row below (64 channels x (XT + 2) pixels x 16 frames, 32 contiguous bytes per pixel and channel) into a ring of
// three rows and the output-gradient row into a second buffer, both as coalesced 16-byte pieces, and the waves read
// their fragments from LDS (channel pitch = an odd number of 16-byte slots: conflict-free).  A wave keeps the three
// patch columns of the 3 x 3 window in registers and slides them along the row: three new fragment reads and one
// gradient fragment per pixel feed nine MFMAs.  80 KB of LDS: two workgroups per CU cover each other's barriers.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WG_XT = 8;                                  // pixels per strip
constexpr int WG_XPITCH = (WG_XT + 2) * 32 + 16;          // bytes per channel of an input row in LDS
constexpr int WG_DPITCH = WG_XT * 32 + 16;
constexpr int WG_LDS = 3 * 64 * WG_XPITCH + 64 * WG_DPITCH;

struct Wgrad3Args {
    const __bf16* xT;
    const __bf16* dyT;
    float* part;
    int cin, cout, cin_total, cin_off;
    int h, w, nfpad;
    int strips, ysegs, yseg_rows;
};

__global__ __launch_bounds__(256) void wgrad3x3_lds_kernel(const Wgrad3Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xl = smem;                              // [3 rows][64 ci][WG_XPITCH]
    char* dl = smem + 3 * 64 * WG_XPITCH;         // [64 co][WG_DPITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
    int z = blockIdx.z;
    const int yseg = z % a.ysegs; z /= a.ysegs;
    const int strip = z % a.strips;
    const int fb = z / a.strips;
    const int x0 = strip * WG_XT;
    const int y0 = yseg * a.yseg_rows, y1 = min(y0 + a.yseg_rows, a.h);
    const long long hw = (long long)a.h * a.w;
    const int cbo = (a.cout + 31) / 32, cbi = (a.cin + 31) / 32;

    // staging pieces: x row = 64 ci x (XT+2) px x 2 halves, dy row = 64 co x XT px x 2 halves (16 bytes each)
    constexpr int XP = 64 * (WG_XT + 2) * 2, DP = 64 * WG_XT * 2, XN = (XP + 255) / 256, DN = DP / 256;
    bf16x8 xv[XN], dv[DN];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_x = [&](int row) {       // input row `row` (may be outside the image: zeros)
#pragma unroll
        for (int k = 0; k < XN; ++k) {
            const int i = tid + k * 256;
            const int hf = i & 1, ch = ((i >> 1) & 31) + 32 * (i / (64 * (WG_XT + 2))), px = (i >> 6) % (WG_XT + 2);
            const int gx = x0 - 1 + px, gc = ci0 + ch;
            bf16x8 v = zero8;
            if (i < XP && row >= 0 && row < a.h && gx >= 0 && gx < a.w && gc < a.cin)
                v = *reinterpret_cast<const bf16x8*>(a.xT + t_off(gc, (long long)row * a.w + gx, fb, cbi, hw) + hf * 8);
            xv[k] = v;
        }
    };
    auto store_x = [&](int row) {
        char* dst = xl + ((row + 3) % 3) * 64 * WG_XPITCH;
#pragma unroll
        for (int k = 0; k < XN; ++k) {
            const int i = tid + k * 256;
            const int hf = i & 1, ch = ((i >> 1) & 31) + 32 * (i / (64 * (WG_XT + 2))), px = (i >> 6) % (WG_XT + 2);
            if (i < XP) *reinterpret_cast<bf16x8*>(dst + ch * WG_XPITCH + px * 32 + hf * 16) = xv[k];
        }
    };
    auto load_d = [&](int row) {
#pragma unroll
        for (int k = 0; k < DN; ++k) {
            const int i = tid + k * 256;
            const int hf = i & 1, ch = ((i >> 1) & 31) + 32 * (i / (64 * WG_XT)), px = (i >> 6) % WG_XT;
            const int gx = x0 + px, gc = co0 + ch;
            bf16x8 v = zero8;
            if (row < a.h && gx < a.w && gc < a.cout)
                v = *reinterpret_cast<const bf16x8*>(a.dyT + t_off(gc, (long long)row * a.w + gx, fb, cbo, hw) + hf * 8);
            dv[k] = v;
        }
    };
    auto store_d = [&]() {
#pragma unroll
        for (int k = 0; k < DN; ++k) {
            const int i = tid + k * 256;
            const int hf = i & 1, ch = ((i >> 1) & 31) + 32 * (i / (64 * WG_XT)), px = (i >> 6) % WG_XT;
            *reinterpret_cast<bf16x8*>(dl + ch * WG_DPITCH + px * 32 + hf * 16) = dv[k];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // rows y0 - 1 and y0 of the input, then the pipeline: iteration y stores (x row y + 1, dy row y), which were
    // requested during the previous iteration's products
    load_x(y0 - 1);
    store_x(y0 - 1);
    load_x(y0);
    store_x(y0);
    load_x(y0 + 1);
    load_d(y0);
    const char* afrag = dl + (wm * 32 + l31) * WG_DPITCH + half * 16;
    const int boff = (wn * 32 + l31) * WG_XPITCH + half * 16;
    for (int y = y0; y < y1; ++y) {
        store_x(y + 1);
        store_d();
        __syncthreads();
        if (y + 1 < y1) {
            load_x(y + 2);
            load_d(y + 1);
        }
        const char* r0 = xl + ((y - 1 + 3) % 3) * 64 * WG_XPITCH + boff;
        const char* r1 = xl + ((y + 3) % 3) * 64 * WG_XPITCH + boff;
        const char* r2 = xl + ((y + 1 + 3) % 3) * 64 * WG_XPITCH + boff;
        bf16x8 c0[3], c1[3], c2[3];
        c0[0] = *reinterpret_cast<const bf16x8*>(r0);
        c0[1] = *reinterpret_cast<const bf16x8*>(r1);
        c0[2] = *reinterpret_cast<const bf16x8*>(r2);
        c1[0] = *reinterpret_cast<const bf16x8*>(r0 + 32);
        c1[1] = *reinterpret_cast<const bf16x8*>(r1 + 32);
        c1[2] = *reinterpret_cast<const bf16x8*>(r2 + 32);
#pragma unroll
        for (int px = 0; px < WG_XT; ++px) {
            c2[0] = *reinterpret_cast<const bf16x8*>(r0 + (px + 2) * 32);
            c2[1] = *reinterpret_cast<const bf16x8*>(r1 + (px + 2) * 32);
            c2[2] = *reinterpret_cast<const bf16x8*>(r2 + (px + 2) * 32);
            const bf16x8 A = *reinterpret_cast<const bf16x8*>(afrag + px * 32);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                acc[3 * r + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, c0[r], acc[3 * r + 0], 0, 0, 0);
                acc[3 * r + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, c1[r], acc[3 * r + 1], 0, 0, 0);
                acc[3 * r + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, c2[r], acc[3 * r + 2], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; }
        }
        __syncthreads();       // the next iteration overwrites the dy row and the oldest input row
    }

    float* part = a.part + ((size_t)blockIdx.z * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * (9 * 4096);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            part[(t * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + wn * 32 + l31] = acc[t][r];
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
    float* part;                 // [b][chunk][4][c] per-workgroup sums (then nothing is added to dgamma .. dshift), or null
};

__global__ __launch_bounds__(256) void rmsnorm_silu_bwd_kernel(const NormBwdArgs a) {
    extern __shared__ float colred[];                // [4][c]: the four per-channel sums of this workgroup
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
        // LP is a power of two from 8: the first 16 lanes by DPP, then across DPP rows
        v = LP >= 16 ? vdmi_row_sum<16>(v) : vdmi_row_sum<8>(v);
        if (LP >= 32) v += __shfl_xor(v, 16);
        if (LP >= 64) v = xhalf_add(v);
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
    // the rows of pass it + 1 are requested before pass it computes: a pass is a dependent load - reduce - reduce - store
    // chain, and this keeps two rows per thread in flight
    auto row_of = [&](long long it) -> long long {
        const long long rl = r_begin + it * R + rslot;
        return (long long)b * a.rows_per_batch + (rl < r_end ? rl : r_begin);
    };
    bf16x8 y8 = *reinterpret_cast<const bf16x8*>(a.y + row_of(0) * a.ldy + ch);
    bf16x8 d8 = *reinterpret_cast<const bf16x8*>(a.dh + row_of(0) * a.ldh + ch);
    for (long long it = 0; it < passes; ++it) {
        const long long rl = r_begin + it * R + rslot;
        const bool valid = rl < r_end;
        const long long row = row_of(it);
        if (LP > 64) __syncthreads();                // red1 / red2 of the previous pass have been read
        float yv[8], dhv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { yv[e] = (float)y8[e]; dhv[e] = valid ? (float)d8[e] : 0.f; }
        if (it + 1 < passes) {
            const long long rn = row_of(it + 1);
            y8 = *reinterpret_cast<const bf16x8*>(a.y + rn * a.ldy + ch);
            d8 = *reinterpret_cast<const bf16x8*>(a.dh + rn * a.ldh + ch);
        }
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += yv[e] * yv[e];
        ss = row_sum(ss, red1);
        const float inv_r = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(ss), 1e-12f));
        float du[8], uu[8], dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float un = yv[e] * inv_r;
            const float n = un * a.norm_scale * g[e];
            const float z = n * sc1[e] + sh[e];
            const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
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
    if (a.part) {
        // no atomics: every row slot leaves its sums in LDS ([slot][4][c], 32 KiB whatever c), one thread per (sum,
        // channel) adds the slots in order and writes the workgroup's partial; a second launch adds the partials in order
        float* tile = colred;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            tile[(rslot * 4 + 0) * a.c + ch + e] = a_dg[e];
            tile[(rslot * 4 + 1) * a.c + ch + e] = a_db[e];
            tile[(rslot * 4 + 2) * a.c + ch + e] = a_ds[e];
            tile[(rslot * 4 + 3) * a.c + ch + e] = a_dsh[e];
        }
        __syncthreads();
        float* out = a.part + ((size_t)b * gridDim.x + blockIdx.x) * 4 * a.c;
        for (int i = tid; i < 4 * a.c; i += 256) {
            float sum = 0.f;
            for (int rs = 0; rs < R; ++rs) sum += tile[rs * 4 * a.c + i];
            out[i] = sum;
        }
        return;
    }
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
// part: per-workgroup sums [chunk][c] for colsum_reduce_kernel (then out is not touched), or null
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16* __restrict__ x, int ld, long long rows, int c,
                                                     long long rows_per_chunk, float* __restrict__ out,
                                                     float* __restrict__ part) {
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
    if (part) {
        // no atomics: the row slots' sums side by side in LDS ([slot][c], at most 2048 floats), added in slot order
        if (rslot < R)
#pragma unroll
            for (int e = 0; e < 8; ++e) colred[rslot * c + piece * 8 + e] = s[e];
        __syncthreads();
        for (int i = tid; i < c; i += 256) {
            float sum = 0.f;
            for (int rs = 0; rs < R; ++rs) sum += colred[rs * c + i];
            part[(size_t)blockIdx.x * c + i] = sum;
        }
        return;
    }
    if (rslot < R)
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&colred[piece * 8 + e], s[e]);
    __syncthreads();
    for (int i = tid; i < c; i += 256) unsafeAtomicAdd(out + i, colred[i]);
}

// out[i] = sum over chunks of part[chunk][i], in a fixed order: 32 channels x 32 interleaved chains per workgroup
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ part, int chunks, int c,
                                                             float* __restrict__ out) {
    __shared__ float red[32][33];
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), chain = threadIdx.x >> 5;
    float s = 0.f;
    if (i < c) {
#pragma unroll 8
        for (int k = chain; k < chunks; k += 32) s += part[(size_t)k * c + i];
    }
    red[chain][threadIdx.x & 31] = s;
    __syncthreads();
    if (chain == 0 && i < c) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
        out[i] = t;
    }
}

// dgamma[i], dbias[i] = sum over samples and chunks, dscale[b][i], dshift[b][i] = sum over chunks of the per-workgroup
// partials [b][chunk][4][c] of rmsnorm_silu_bwd_kernel, in a fixed order: a workgroup takes 32 channels of one output
// row (q = 0: dgamma, 1: dbias, 2 + b: dscale[b], 2 + batch + b: dshift[b]), 32 interleaved chains of chunks each
__global__ __launch_bounds__(1024) void norm_bwd_reduce_kernel(const float* __restrict__ part, int chunks, int batch, int c,
                                                               float* dgamma, float* dbias, float* dscale, float* dshift) {
    __shared__ float red[32][33];
    const int q = blockIdx.y, i = blockIdx.x * 32 + (threadIdx.x & 31), chain = threadIdx.x >> 5;
    const int which = q == 0 ? 0 : (q == 1 ? 1 : (q < 2 + batch ? 2 : 3));
    const int b_begin = q < 2 ? 0 : (q < 2 + batch ? q - 2 : q - 2 - batch), b_end = q < 2 ? batch : b_begin + 1;
    float s = 0.f;
    if (i < c) {
        for (int b = b_begin; b < b_end; ++b) {
            const float* p = part + ((size_t)b * chunks * 4 + which) * c + i;
#pragma unroll 8
            for (int k = chain; k < chunks; k += 32) s += p[(size_t)k * 4 * c];
        }
    }
    red[chain][threadIdx.x & 31] = s;
    __syncthreads();
    if (chain == 0 && i < c) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
        float* dst = q == 0 ? dgamma : (q == 1 ? dbias : (q < 2 + batch ? dscale + (size_t)(q - 2) * c
                                                                           : dshift + (size_t)(q - 2 - batch) * c));
        if (dst) dst[i] = t;
    }
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

extern "C" size_t vdmi_pack_taps_batch_table_bytes(int count) { return count > 0 ? (size_t)count * sizeof(PackJobDev) : 0; }

extern "C" int vdmi_pack_taps_batch(const vdmi_pack_job* jobs, int count, void* device_table, vdmi_stream stream) {
    if (!jobs || !device_table || count <= 0) return VDMI_E_ARG;
    if (count > 65536) return VDMI_E_SHAPE;
    // the host copy of the table: a ring of buffers that live on after the call, so that the bytes are still there should
    // the runtime read a pageable source after hipMemcpyAsync has returned (a buffer is reused eight calls later)
    static std::mutex ring_mutex;
    static std::vector<PackJobDev> ring[8];
    static unsigned ring_next = 0;
    std::lock_guard<std::mutex> lock(ring_mutex);
    std::vector<PackJobDev>& tab = ring[ring_next++ & 7u];
    tab.assign((size_t)count, PackJobDev{});
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        const vdmi_pack_job& j = jobs[i];
        if (!j.w || !j.out) return VDMI_E_ARG;
        if (j.n <= 0 || j.npad < j.n || j.npad % 32 || j.cin0 <= 0 || j.cin1 < 0 || j.ntaps <= 0 || (j.bkc != 32 && j.bkc != 64))
            return VDMI_E_SHAPE;
        if (!aligned16(j.out)) return VDMI_E_ALIGN;
        PackJobDev& d = tab[(size_t)i];
        d.w = j.w;
        d.in_scale = j.in_scale;
        d.gather = j.gather;
        d.out = static_cast<__bf16*>(j.out);
        d.pieces = (long long)vdmi_pack_taps_elems(j.n, j.npad, j.cin0, j.cin1, j.ntaps, j.bkc) / 8;
        d.block_begin = blocks;
        d.n = j.n; d.npad = j.npad; d.cin0 = j.cin0; d.cin1 = j.cin1; d.ntaps = j.ntaps; d.bkc = j.bkc;
        d.transposed = j.transposed ? 1 : 0;
        d.pad_ = 0;
        blocks += (d.pieces + 255) / 256;
    }
    if (blocks <= 0 || blocks > 0x7fffffffLL) return VDMI_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(device_table, tab.data(), tab.size() * sizeof(PackJobDev), hipMemcpyHostToDevice, st) != hipSuccess)
        return VDMI_E_LAUNCH;
    hipLaunchKernelGGL(pack_taps_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                       static_cast<const PackJobDev*>(device_table), count);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" size_t vdmi_rows_transpose_elems(int c, int nf, int64_t hw) {
    if (c <= 0 || nf <= 0 || hw <= 0) return 0;
    return (size_t)((c + 31) / 32 * 32) * (size_t)hw * (size_t)((nf + 15) / 16 * 16);
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
                       (int)hw, static_cast<__bf16*>(out));
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

namespace {
// how a weight-gradient launch splits the contraction (a function of the shapes only: the same every call)
struct WgradPlan {
    int tx, ty, nfb, strips, ysegs, yseg_rows;
    long long units, per, gz;
};
int wgrad_plan(int cin, int cout, int nf, int h, int w, int ntaps, WgradPlan* p) {
    if (cin <= 0 || cout <= 0 || nf <= 0 || h <= 0 || w <= 0 || (ntaps != 1 && ntaps != 9)) return VDMI_E_SHAPE;
    if ((long long)h * w > 0x3fffffffLL) return VDMI_E_SHAPE;
    p->tx = (cout + 63) / 64;
    p->ty = (cin + 63) / 64;
    p->nfb = (nf + 15) / 16;
    if (p->tx > 65535 || p->ty > 65535) return VDMI_E_SHAPE;
    const long long tiles = (long long)p->tx * p->ty;
    constexpr long long kTargetWgs = 768;       // ~3 per CU; every workgroup writes a 64 x 64 x taps partial tile
    if (ntaps == 9) {
        // the LDS kernel holds 80 KB: two workgroups per CU, 512 on the device - one round of them, not one and a half
        constexpr long long kTargetWgs3 = 512;
        p->strips = (w + WG_XT - 1) / WG_XT;
        const long long base = tiles * p->nfb * p->strips;
        long long segs = kTargetWgs3 / base;
        if (segs > (h + 3) / 4) segs = (h + 3) / 4;      // a segment re-reads two halo rows: at least 4 rows each
        if (segs < 1) segs = 1;
        p->yseg_rows = (int)((h + segs - 1) / segs);
        p->ysegs = (h + p->yseg_rows - 1) / p->yseg_rows;
        p->gz = (long long)p->nfb * p->strips * p->ysegs;
        p->units = p->per = 0;
    } else {
        p->strips = p->ysegs = p->yseg_rows = 0;
        p->units = (long long)p->nfb * h * w;
        long long split = (kTargetWgs + tiles - 1) / tiles;
        if (split > p->units / 16) split = p->units / 16;
        if (split < 1) split = 1;
        p->per = (p->units + split - 1) / split;
        p->gz = (p->units + p->per - 1) / p->per;
    }
    return p->gz > 65535 ? VDMI_E_SHAPE : VDMI_OK;
}
}   // namespace

extern "C" size_t vdmi_conv_wgrad_workspace_bytes(int cin, int cout, int nf, int h, int w, int ntaps) {
    WgradPlan p;
    if (wgrad_plan(cin, cout, nf, h, w, ntaps, &p) != VDMI_OK) return 0;
    return (size_t)p.gz * p.tx * p.ty * ntaps * 4096 * sizeof(float);
}

extern "C" int vdmi_conv_wgrad(const void* xT, const void* dyT, float* dw, int cin, int cout, int cin_total, int cin_off,
                               int nf, int h, int w, int ntaps, const int8_t* dy, const int8_t* dx, void* workspace,
                               vdmi_stream stream) {
    if (!xT || !dyT || !dw || !dy || !dx || !workspace) return VDMI_E_ARG;
    if (cin_off < 0 || cin_off + cin > cin_total) return VDMI_E_SHAPE;
    WgradPlan p;
    const int st = wgrad_plan(cin, cout, nf, h, w, ntaps, &p);
    if (st != VDMI_OK) return st;
    if (!aligned16(xT) || !aligned16(dyT) || !aligned16(workspace)) return VDMI_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    const dim3 grid((unsigned)p.tx, (unsigned)p.ty, (unsigned)p.gz);
    if (ntaps == 9) {
        // the LDS kernel slides a 3 x 3 window: the tap table must be the row-major (ky - 1, kx - 1)
        for (int t = 0; t < 9; ++t)
            if (dy[t] != t / 3 - 1 || dx[t] != t % 3 - 1) return VDMI_E_SHAPE;
        Wgrad3Args g;
        g.xT = static_cast<const __bf16*>(xT);
        g.dyT = static_cast<const __bf16*>(dyT);
        g.part = part;
        g.cin = cin; g.cout = cout; g.cin_total = cin_total; g.cin_off = cin_off;
        g.h = h; g.w = w; g.nfpad = p.nfb * 16;
        g.strips = p.strips; g.ysegs = p.ysegs; g.yseg_rows = p.yseg_rows;
        static std::atomic<uint64_t> attr_mask{0};
        if (vdmi_set_max_lds(reinterpret_cast<const void*>(wgrad3x3_lds_kernel), attr_mask, WG_LDS)) return VDMI_E_LAUNCH;
        hipLaunchKernelGGL(wgrad3x3_lds_kernel, grid, dim3(256), WG_LDS, s, g);
    } else {
        WgradArgs a;
        a.xT = static_cast<const __bf16*>(xT);
        a.dyT = static_cast<const __bf16*>(dyT);
        a.part = part;
        a.cin = cin; a.cout = cout; a.cin_total = cin_total; a.cin_off = cin_off;
        a.h = h; a.w = w; a.nfpad = p.nfb * 16;
        for (int t = 0; t < 9; ++t) { a.dy[t] = t < ntaps ? dy[t] : 0; a.dx[t] = t < ntaps ? dx[t] : 0; }
        a.units = p.units;
        a.per = p.per;
        hipLaunchKernelGGL(wgrad1_kernel, grid, dim3(256), 0, s, a);
    }
    VDMI_LAUNCH_CHECK();
    const long long elems = (long long)p.tx * p.ty * ntaps * 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, part, (int)p.gz, p.tx,
                       p.ty, ntaps, dw, cout, cin, cin_total, cin_off);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

namespace {
struct Wgrad3RowsPlan { int tx, ty, strips, ytiles; bool narrow; long long units, per, gz; };
int wgrad3_rows_plan(int cin, int cout, int nf, int h, int w, Wgrad3RowsPlan* p) {
    if (cin <= 0 || cout <= 0 || cin % 8 || cout % 8 || nf <= 0 || h <= 0 || w <= 0) return VDMI_E_SHAPE;
    if ((long long)nf * h * w > 0x7fffffffLL) return VDMI_E_SHAPE;
    p->tx = (cout + 63) / 64;
    p->ty = (cin + 63) / 64;
    if (p->tx > 65535 || p->ty > 65535) return VDMI_E_SHAPE;
    p->narrow = w <= 16;                                   // 16 x 8 pixel tiles instead of 32 x 4
    const int xt = p->narrow ? 16 : 32, tyl = p->narrow ? 8 : 4;
    p->strips = (w + xt - 1) / xt;
    p->ytiles = (h + tyl - 1) / tyl;
    p->units = (long long)nf * p->strips * p->ytiles;
    // 62 KiB of LDS: two workgroups per CU, one round of 512 on the device; every workgroup writes a partial tile
    long long split = 512 / ((long long)p->tx * p->ty);
    if (split > p->units) split = p->units;
    if (split < 1) split = 1;
    p->per = (p->units + split - 1) / split;
    p->gz = (p->units + p->per - 1) / p->per;
    return p->gz > 65535 ? VDMI_E_SHAPE : VDMI_OK;
}
}   // namespace

extern "C" size_t vdmi_conv_wgrad_rows_workspace_bytes(int cin, int cout, int nf, int h, int w) {
    Wgrad3RowsPlan p;
    if (wgrad3_rows_plan(cin, cout, nf, h, w, &p) != VDMI_OK) return 0;
    return (size_t)p.gz * p.tx * p.ty * 9 * 4096 * sizeof(float);
}

extern "C" int vdmi_conv_wgrad_rows(const void* x, int ldx, const void* dy, int lddy, float* dw, int cin, int cout,
                                    int cin_total, int cin_off, int nf, int h, int w, void* workspace, vdmi_stream stream) {
    if (!x || !dy || !dw || !workspace) return VDMI_E_ARG;
    if (cin_off < 0 || cin_off + cin > cin_total || ldx < cin || lddy < cout || ldx % 8 || lddy % 8) return VDMI_E_SHAPE;
    Wgrad3RowsPlan p;
    const int st = wgrad3_rows_plan(cin, cout, nf, h, w, &p);
    if (st != VDMI_OK) return st;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(workspace)) return VDMI_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Wgrad3RowsArgs a;
    a.x = static_cast<const __bf16*>(x);
    a.dy = static_cast<const __bf16*>(dy);
    a.part = static_cast<float*>(workspace);
    a.ldx = ldx; a.lddy = lddy; a.cin = cin; a.cout = cout; a.h = h; a.w = w;
    a.strips = p.strips; a.ytiles = p.ytiles; a.units = p.units; a.per = p.per;
    const dim3 grid((unsigned)p.tx, (unsigned)p.ty, (unsigned)p.gz);
    if (p.narrow) {
        constexpr int lds = (18 * 10 + 128) * RW_PITCH;
        hipLaunchKernelGGL((wgrad3x3_rows_kernel<16, 8>), grid, dim3(256), lds, s, a);
    } else {
        constexpr int lds = (34 * 6 + 128) * RW_PITCH;
        hipLaunchKernelGGL((wgrad3x3_rows_kernel<32, 4>), grid, dim3(256), lds, s, a);
    }
    VDMI_LAUNCH_CHECK();
    const long long elems = (long long)p.tx * p.ty * 9 * 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, a.part, (int)p.gz, p.tx,
                       p.ty, 9, dw, cout, cin, cin_total, cin_off);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

namespace {
struct Wgrad4Plan { int tx, ty, strips, ytiles; long long units, per, gz; };
int wgrad4_plan(int cb, int cs, int nf, int h, int w, Wgrad4Plan* p) {
    if (cb <= 0 || cs <= 0 || cb % 8 || cs % 8 || nf <= 0 || h <= 0 || w <= 0 || h % 2 || w % 2) return VDMI_E_SHAPE;
    if ((long long)nf * h * w > 0x7fffffffLL) return VDMI_E_SHAPE;
    p->tx = (cs + 63) / 64;
    p->ty = (cb + 31) / 32;
    if (p->tx > 65535 || p->ty > 65535) return VDMI_E_SHAPE;
    p->strips = (w / 2 + W4_XT - 1) / W4_XT;
    p->ytiles = (h / 2 + W4_TY - 1) / W4_TY;
    p->units = (long long)nf * p->strips * p->ytiles;
    long long split = 512 / ((long long)p->tx * p->ty);      // 63 KiB of LDS: two workgroups per CU, one round
    if (split > p->units) split = p->units;
    if (split < 1) split = 1;
    p->per = (p->units + split - 1) / split;
    p->gz = (p->units + p->per - 1) / p->per;
    return p->gz > 65535 ? VDMI_E_SHAPE : VDMI_OK;
}
}   // namespace

extern "C" size_t vdmi_strided_wgrad_rows_workspace_bytes(int cb, int cs, int nf, int h, int w) {
    Wgrad4Plan p;
    if (wgrad4_plan(cb, cs, nf, h, w, &p) != VDMI_OK) return 0;
    return (size_t)p.gz * p.tx * p.ty * 16 * 2048 * sizeof(float);
}

extern "C" int vdmi_strided_wgrad_rows(const void* big, int ldb, const void* small, int lds, float* dw, int cb, int cs,
                                       int nf, int h, int w, void* workspace, vdmi_stream stream) {
    if (!big || !small || !dw || !workspace) return VDMI_E_ARG;
    if (ldb < cb || lds < cs || ldb % 8 || lds % 8) return VDMI_E_SHAPE;
    Wgrad4Plan p;
    const int st = wgrad4_plan(cb, cs, nf, h, w, &p);
    if (st != VDMI_OK) return st;
    if (!aligned16(big) || !aligned16(small) || !aligned16(workspace)) return VDMI_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    Wgrad4Args a;
    a.big = static_cast<const __bf16*>(big);
    a.small = static_cast<const __bf16*>(small);
    a.part = static_cast<float*>(workspace);
    a.ldb = ldb; a.lds = lds; a.cb = cb; a.cs = cs; a.h = h; a.w = w;
    a.strips = p.strips; a.ytiles = p.ytiles; a.units = p.units; a.per = p.per;
    hipLaunchKernelGGL(wgrad4x4s2_rows_kernel, dim3((unsigned)p.tx, (unsigned)p.ty, (unsigned)p.gz), dim3(256), W4_LDS, s, a);
    VDMI_LAUNCH_CHECK();
    const long long elems = (long long)p.tx * p.ty * 16 * 2048;
    hipLaunchKernelGGL(wgrad4_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, a.part, (int)p.gz, p.tx,
                       p.ty, dw, cs, cb);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

namespace {
int g_rw128 = -1;        // the 128 x 128 tile: on unless VDMI_WGRAD_ROWS128=0
struct RowsWgradPlan { int tx, ty; long long per, gz; bool big; };
int rows_wgrad_plan(int cin, int cout, int64_t rows, RowsWgradPlan* p) {
    if (cin <= 0 || cout <= 0 || cin % 8 || cout % 8 || rows <= 0) return VDMI_E_SHAPE;
    if (g_rw128 < 0) {
        const char* e = std::getenv("VDMI_WGRAD_ROWS128");
        g_rw128 = (e && e[0] == '0') ? 0 : 1;
    }
    p->tx = (cout + 63) / 64;
    p->ty = (cin + 63) / 64;
    if (p->tx > 65535 || p->ty > 65535) return VDMI_E_SHAPE;
    p->big = g_rw128 && cin >= 128 && cout >= 128;
    const long long tiles = p->big ? (long long)((p->tx + 1) / 2) * ((p->ty + 1) / 2) : (long long)p->tx * p->ty;
    // small tile: ~3 workgroups per CU (48 KB of LDS each); big tile: one and a half rounds of one per CU (fewer, longer
    // workgroups also mean fewer partial tiles for the reduce launch)
    static int big_wgs = 0;
    if (!big_wgs) {
        const char* e = std::getenv("VDMI_WGRAD_ROWS128_WGS");
        big_wgs = e ? std::atoi(e) : 384;       // measured over the training step: 256 / 384 / 512 / 768 -> 30.15 / 29.96 / 30.39 / 30.6 ms
        if (big_wgs < 64 || big_wgs > 4096) big_wgs = 384;
    }
    long long split = ((p->big ? big_wgs : 768) + tiles - 1) / tiles;
    const long long stages = (rows + RW_KC - 1) / RW_KC;
    if (split > stages) split = stages;
    p->per = (stages + split - 1) / split * RW_KC;
    p->gz = (rows + p->per - 1) / p->per;
    return p->gz > 65535 ? VDMI_E_SHAPE : VDMI_OK;
}
}   // namespace

extern "C" size_t vdmi_rows_wgrad_workspace_bytes(int cin, int cout, int64_t rows) {
    RowsWgradPlan p;
    if (rows_wgrad_plan(cin, cout, rows, &p) != VDMI_OK) return 0;
    return (size_t)p.gz * p.tx * p.ty * 4096 * sizeof(float);
}

extern "C" int vdmi_rows_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, int cin, int cout,
                               int cin_total, int cin_off, int64_t rows, void* workspace, vdmi_stream stream) {
    if (!x || !dy || !dw || !workspace) return VDMI_E_ARG;
    if (cin_off < 0 || cin_off + cin > cin_total || ldx < cin || lddy < cout || ldx % 8 || lddy % 8) return VDMI_E_SHAPE;
    RowsWgradPlan p;
    const int st = rows_wgrad_plan(cin, cout, rows, &p);
    if (st != VDMI_OK) return st;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(workspace)) return VDMI_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    RowsWgradArgs a;
    a.x = static_cast<const __bf16*>(x); a.ldx = ldx;
    a.dy = static_cast<const __bf16*>(dy); a.lddy = lddy;
    a.part = static_cast<float*>(workspace);
    a.cin = cin; a.cout = cout; a.rows = rows; a.per = p.per;
    if (p.big)
        hipLaunchKernelGGL(wgrad_rows128_kernel, dim3((unsigned)((p.tx + 1) / 2), (unsigned)((p.ty + 1) / 2), (unsigned)p.gz),
                           dim3(512), 0, s, a, p.tx, p.ty);
    else
        hipLaunchKernelGGL(wgrad_rows_kernel, dim3((unsigned)p.tx, (unsigned)p.ty, (unsigned)p.gz), dim3(256), 0, s, a);
    VDMI_LAUNCH_CHECK();
    const long long elems = (long long)p.tx * p.ty * 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, a.part, (int)p.gz, p.tx,
                       p.ty, 1, dw, cout, cin, cin_total, cin_off);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

extern "C" size_t vdmi_conv_wgrad_gather_workspace_bytes(int cin, int cout, int nf, int h, int w, int kh, int kw) {
    if (kh <= 0 || kw <= 0 || cin <= 0 || (long long)cin * kh * kw > 65535) return 0;
    return vdmi_conv_wgrad_workspace_bytes(cin * kh * kw, cout, nf, h, w, 1);
}

extern "C" int vdmi_conv_wgrad_gather(const void* xT, const void* dyT, float* dw, int cin, int cout, int nf, int h, int w,
                                      int kh, int kw, void* workspace, vdmi_stream stream) {
    if (!xT || !dyT || !dw || !workspace) return VDMI_E_ARG;
    if (kh <= 0 || kw <= 0 || !(kh & 1) || !(kw & 1) || cin <= 0 || (long long)cin * kh * kw > 65535) return VDMI_E_SHAPE;
    const int ncol = cin * kh * kw;
    WgradPlan p;
    const int st = wgrad_plan(ncol, cout, nf, h, w, 1, &p);
    if (st != VDMI_OK) return st;
    if (!aligned16(xT) || !aligned16(dyT) || !aligned16(workspace)) return VDMI_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WgradGatherArgs a;
    a.xT = static_cast<const __bf16*>(xT);
    a.dyT = static_cast<const __bf16*>(dyT);
    a.part = static_cast<float*>(workspace);
    a.cin = cin; a.cout = cout; a.kh = kh; a.kw = kw;
    a.h = h; a.w = w; a.nfpad = p.nfb * 16;
    a.units = p.units; a.per = p.per;
    hipLaunchKernelGGL(wgrad_gather_kernel, dim3((unsigned)p.tx, (unsigned)p.ty, (unsigned)p.gz), dim3(256), 0, s, a);
    VDMI_LAUNCH_CHECK();
    const long long elems = (long long)p.tx * p.ty * 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, a.part, (int)p.gz, p.tx,
                       p.ty, 1, dw, cout, ncol, ncol, 0);
    VDMI_LAUNCH_CHECK();
    return VDMI_OK;
}

namespace {
long long norm_bwd_chunks(int64_t rows_per_batch, int c) {
    const int R = 256 / (c / 8);
    const long long per = (long long)R * 16;
    return (rows_per_batch + per - 1) / per;
}
}   // namespace

static int rmsnorm_silu_bwd_impl(const void* y, int ldy, const void* dh, int ldh, void* dy, int lddy, int batch,
                                 int64_t rows_per_batch, int c, const float* gamma, const float* scale,
                                 const float* shift, int64_t sstride, float norm_scale, float* dgamma, float* dbias,
                                 float* dscale, float* dshift, void* workspace, vdmi_stream stream) {
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
    // 16 passes per workgroup: a pass is one load - reduce - reduce - store chain, and only other workgroups on the CU
    // cover its latency (64 passes: 130 us per layer at 16 x 128 x 128, 16 passes: 65 us; 8 / 4 / 2 passes cost the
    // training step +0.4 / +1.9 / +4.8 ms when the sums went to float atomics on the same few addresses)
    const long long per = (long long)R * 16;
    const long long chunks = norm_bwd_chunks(rows_per_batch, c);
    if (chunks > 0x7fffffffLL) return VDMI_E_SHAPE;
    a.rows_per_chunk = per;
    a.part = static_cast<float*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = workspace ? (size_t)R * 4 * c * sizeof(float) : (size_t)4 * c * sizeof(float);
    hipLaunchKernelGGL(rmsnorm_silu_bwd_kernel, dim3((unsigned)chunks, (unsigned)batch), dim3(256), lds, st, a);
    VDMI_LAUNCH_CHECK();
    if (workspace) {
        const int nq = 2 + (dscale || dshift ? 2 * batch : 0);
        hipLaunchKernelGGL(norm_bwd_reduce_kernel, dim3((unsigned)((c + 31) / 32), (unsigned)nq), dim3(1024), 0, st,
                           a.part, (int)chunks, batch, c, dgamma, dbias, dscale, dshift);
        VDMI_LAUNCH_CHECK();
    }
    return VDMI_OK;
}

extern "C" int vdmi_rmsnorm_silu_bwd(const void* y, int ldy, const void* dh, int ldh, void* dy, int lddy, int batch,
                                     int64_t rows_per_batch, int c, const float* gamma, const float* scale,
                                     const float* shift, int64_t sstride, float norm_scale, float* dgamma, float* dbias,
                                     float* dscale, float* dshift, vdmi_stream stream) {
    return rmsnorm_silu_bwd_impl(y, ldy, dh, ldh, dy, lddy, batch, rows_per_batch, c, gamma, scale, shift, sstride, norm_scale,
                                 dgamma, dbias, dscale, dshift, nullptr, stream);
}

extern "C" size_t vdmi_rmsnorm_silu_bwd_workspace_bytes(int batch, int64_t rows_per_batch, int c) {
    if (batch <= 0 || rows_per_batch <= 0 || c < 64 || c > 2048 || (c & (c - 1))) return 0;
    return (size_t)batch * (size_t)norm_bwd_chunks(rows_per_batch, c) * 4 * c * sizeof(float);
}

extern "C" int vdmi_rmsnorm_silu_bwd_ws(const void* y, int ldy, const void* dh, int ldh, void* dy, int lddy, int batch,
                                        int64_t rows_per_batch, int c, const float* gamma, const float* scale,
                                        const float* shift, int64_t sstride, float norm_scale, float* dgamma, float* dbias,
                                        float* dscale, float* dshift, void* workspace, vdmi_stream stream) {
    if (!workspace || !aligned16(workspace)) return workspace ? VDMI_E_ALIGN : VDMI_E_ARG;
    return rmsnorm_silu_bwd_impl(y, ldy, dh, ldh, dy, lddy, batch, rows_per_batch, c, gamma, scale, shift, sstride, norm_scale,
                                 dgamma, dbias, dscale, dshift, workspace, stream);
}

static int colsum_impl(const void* x, int ld, int64_t rows, int c, float* out, void* workspace, vdmi_stream stream) {
    if (!x || !out) return VDMI_E_ARG;
    if (rows <= 0 || c < 8 || c > 2048 || c % 8 || ld % 8 || ld < c) return VDMI_E_SHAPE;
    if (!aligned16(x)) return VDMI_E_ALIGN;
    const int R = 256 / (c / 8);
    if (R < 1) return VDMI_E_SHAPE;
    const long long per = (long long)R * 16;
    const long long chunks = (rows + per - 1) / per;
    if (chunks > 0x7fffffffLL) return VDMI_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)chunks), dim3(256), 0, st, static_cast<const __bf16*>(x), ld,
                       (long long)rows, c, per, out, static_cast<float*>(workspace));
    VDMI_LAUNCH_CHECK();
    if (workspace) {
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((c + 31) / 32)), dim3(1024), 0, st,
                           static_cast<const float*>(workspace), (int)chunks, c, out);
        VDMI_LAUNCH_CHECK();
    }
    return VDMI_OK;
}

extern "C" int vdmi_colsum(const void* x, int ld, int64_t rows, int c, float* out, vdmi_stream stream) {
    return colsum_impl(x, ld, rows, c, out, nullptr, stream);
}

extern "C" size_t vdmi_colsum_workspace_bytes(int64_t rows, int c) {
    if (rows <= 0 || c < 8 || c > 2048 || c % 8) return 0;
    const long long per = (long long)(256 / (c / 8)) * 16;
    return (size_t)((rows + per - 1) / per) * c * sizeof(float);
}

extern "C" int vdmi_colsum_ws(const void* x, int ld, int64_t rows, int c, float* out, void* workspace, vdmi_stream stream) {
    if (!workspace) return VDMI_E_ARG;
    if (!aligned16(workspace)) return VDMI_E_ALIGN;
    return colsum_impl(x, ld, rows, c, out, workspace, stream);
}
