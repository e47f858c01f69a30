/*
 * vdmi — MI355X (gfx950) kernels for the denoising hot path of kasjfksj/video-diffusion:
 * the Unet3D forward called at every step of GaussianDiffusion.p_sample_loop, and the
 * DDPM posterior update around it.
 *
 * The reference is pure Python and has no FFI of its own; the seam it offers is the
 * `denoise_fn` callable and the nn.Module tree.  Every entry point below therefore cites
 * the reference *arithmetic* it replaces (file:line under the reference repository,
 * vdp.py = video_diffusion_pytorch/video_diffusion_pytorch.py).
 *
 * Conventions
 *  - Activations are channels-last bf16: a tensor (B, C, F, H, W) of the reference is held as
 *    rows [B*F*H*W][C], row index ((b*F + f)*H + y)*W + x.
 *  - All pointers are device pointers unless a function is marked HOST.
 *  - Every function enqueues on `stream` and returns immediately; nothing allocates, frees or
 *    synchronises, no pointer is retained after the call (graph-capture safe).
 *  - Return value: 0 on success, a negative vdmi_status otherwise; vdmi_strerror() names it.
 *    No C++ exception crosses this boundary.
 */
#ifndef VDMI_H
#define VDMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vdmi_stream;            /* hipStream_t */

enum vdmi_status {
    VDMI_OK = 0,
    VDMI_E_ARG = -1,                  /* null pointer / bad enum / inconsistent sizes */
    VDMI_E_SHAPE = -2,                /* shape outside what the kernels support       */
    VDMI_E_ALIGN = -3,                /* pointer or leading dimension not 16-byte aligned */
    VDMI_E_LAUNCH = -4,               /* hipLaunchKernel reported an error            */
    VDMI_E_LDS = -5,                  /* tile does not fit the 160 KiB LDS            */
    VDMI_E_SCRATCH = -6               /* input needs more than the library's static scratch buffer */
};

const char* vdmi_strerror(int status);            /* HOST */
int vdmi_version(void);                           /* HOST */

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM "tap" convolution with fused epilogues.
 *
 *   out[pix][n] = EPILOGUE( sum_{tap, cin} W[n][cin][tap] * in[pix*stride + (dy,dx)_tap][cin] )
 *
 * One kernel family covers, through the tap table and the epilogue:
 *   Block.proj (1,3,3) conv + RMSNorm + (scale+1)/shift + SiLU        vdp.py:412-427, 391-398
 *   ResnetBlock residual (1x1 res_conv or identity), fused            vdp.py:439, 452
 *   Downsample (1,4,4)/s2 conv, Upsample transposed conv (4 phases)   vdp.py:374-378
 *   PreNorm LayerNorm folded into to_qkv (Linear / Conv2d 1x1)        vdp.py:380-389, 400-408, 460, 560
 *   q scaling + rotary on q,k                                         vdp.py:579-583
 *   SpatialLinearAttention q softmax / k,v transposed staging         vdp.py:467-473
 *   to_out projections + residual add                                 vdp.py:331-337, 478, 612
 *   torch.cat skip connections: two input tensors read in place       vdp.py:817, 830
 * ------------------------------------------------------------------------------------------ */

#define VDMI_MAX_TAPS 16

enum vdmi_epilogue {
    VDMI_EPI_PLAIN = 0,      /* + bias, + optional identity residual                       */
    VDMI_EPI_NORM_SILU = 1,  /* + bias, RMSNorm over n, *gain[b] + shift[b], SiLU, + residual */
    VDMI_EPI_QKV_ATTN = 2,   /* folded LayerNorm, q*scale, optional rotary on q,k          */
    VDMI_EPI_QKV_SLA = 3     /* folded LayerNorm, q softmax over head dim * scale -> out;
                                k,v transposed per frame -> out2                            */
};

typedef struct vdmi_tap_pass {
    const void* src0;        /* bf16 [frames*in_h*in_w][c0]                                */
    const void* src1;        /* optional second tensor, channels appended after src0       */
    int32_t c0, c1;          /* channels of each source (multiples of 8; c1 = 0 if unused) */
    int32_t in_h, in_w;      /* input image per frame                                      */
    int32_t stride;          /* input pixel = grid pixel * stride + (dy, dx)               */
    int32_t ntaps;
    int8_t dy[VDMI_MAX_TAPS];
    int8_t dx[VDMI_MAX_TAPS];
    const void* w;           /* weights packed by vdmi_pack_taps (bf16)                    */
    int64_t w_group_stride;  /* elements between per-group weight sets (0: one shared set) */
    int32_t w_group_frames;  /* frames per weight group (used when w_group_stride != 0)    */
    int32_t phases;          /* 0 / 1: off.  P = out_mul*out_mul > 1 (main pass, PLAIN): the taps are P
                                equal groups run as independent convolutions of one launch; group p uses
                                the weight set at w + p*w_phase_stride and writes output pixel
                                grid*out_mul + out_off + (p / out_mul, p % out_mul) -- the output
                                parities of a stride-2 transposed conv (vdp.py:374-375)          */
    int64_t w_phase_stride;  /* elements between the phases' weight sets                          */
    int32_t space_to_depth;  /* 1: src0 is bf16 [frames*in_h*in_w][c0/4] and the pass sees its 2x2
                                space-to-depth view: pixel (Y, X), channel (2a+b)*(c0/4) + ci is source
                                pixel (2Y-1+a, 2X-1+b), zero outside the image.  A 4x4 / stride 2 / pad 1
                                convolution (Downsample, vdp.py:377-378) is then the stride-1 taps
                                (0,0),(0,1),(1,0),(1,1) over 4*C channels.  Needs stride 1, c1 = 0,
                                (c0/4) a multiple of the input chunk, frames*in_h*in_w < 2^27.       */
    int32_t reserved;
} vdmi_tap_pass;

typedef struct vdmi_tapgemm_args {
    vdmi_tap_pass main;
    vdmi_tap_pass res;             /* NORM_SILU only: 1-tap pass added after the activation  */
    int32_t has_res;               /* 0: none, 1: `res` pass (+res_bias), 2: identity tensor */
    const void* res_identity;      /* bf16 [out rows][ldo_res], read at column n0            */
    int32_t ld_res;
    const float* res_bias;         /* [n] or NULL                                            */

    int32_t frames;                /* B*F                                                    */
    int32_t frames_per_batch;      /* F: batch index of a frame = frame / F                  */
    int32_t grid_h, grid_w;        /* tile grid extent per frame (output positions computed) */
    int32_t out_mul, out_off_y, out_off_x;  /* output pixel = grid pixel*out_mul + off       */
    int32_t out_h, out_w;          /* output image per frame                                 */
    void* out;                     /* bf16 [frames*out_h*out_w][ldo]                         */
    int32_t ldo;
    int32_t n;                     /* output channels (multiple of 32)                       */
    int32_t epilogue;              /* enum vdmi_epilogue                                     */

    const float* bias;             /* [n] or NULL                                            */
    const float* gain;             /* NORM: gamma*(scale+1), [B][gain_bstride]               */
    const float* shift;            /* NORM: shift, [B][shift_bstride] or NULL                */
    int64_t gain_bstride, shift_bstride;   /* 0: same vector for every batch element        */
    float norm_scale;              /* sqrt(n)                                                */

    const float* colsum;           /* QKV: sum_k W'[n][k] of the packed (bf16) weights       */
    float ln_eps;
    float q_scale;
    const float* rot_cos;          /* [rot_frames][16] or NULL (no rotary)                   */
    const float* rot_sin;
    int32_t rot_frames;            /* F                                                      */
    int32_t rot_hw;                /* H*W: position of a row = (row / rot_hw) % rot_frames   */
    void* out2;                    /* QKV_SLA: bf16 [frames][512][sla_hw] (k then v)         */
    int32_t sla_hw;
    int32_t tile_bn;               /* PLAIN only: 0 = automatic, 64 = force 64-channel output tiles
                                      (more pixels per workgroup, less weight traffic)        */
    int32_t q_cols;                /* QKV_ATTN: leading output columns that are queries (scaled by
                                      q_scale); 0 = n/3 (q, k, v thirds), n = a query-only projection */
    int32_t act;                   /* PLAIN only: 0 = none, 1 = SiLU after the bias (before the residual) */
    int32_t k_split;               /* PLAIN only, no act / residual: 0 or 1 = off; k = 2..4: the channel chunks are dealt
                                      to k workgroups per tile (k times the workgroups of a small grid); share i writes
                                      its partial sums to out + i * out_split_stride and the consumer adds them
                                      (vdmi_rmsnorm_silu_sum); share 0 carries the bias */
    int64_t out_split_stride;      /* elements between the partial outputs (multiple of 8)   */
} vdmi_tapgemm_args;

/* HOST. Tile the library will use for (epilogue, n): *bn = output-channel tile, *bkc = input
 * channel chunk.  Weights must be packed with the same bkc and n padded to a multiple of bn. */
int vdmi_tapgemm_tile(int epilogue, int n, int* bn, int* bkc);

/* HOST. Pack float weights w[n][cin0+cin1][ntaps] into the fragment order the kernel streams:
 * [chunk][tap][npad/32][bkc/16][64 lanes][8] bf16, each source's channels padded to a multiple
 * of bkc.  `out` holds vdmi_pack_taps_elems() uint16 values.  If colsum != NULL it receives
 * sum over (cin, tap) of the bf16-rounded weights per output channel (float[n]). */
size_t vdmi_pack_taps_elems(int n, int npad, int cin0, int cin1, int ntaps, int bkc);
int vdmi_pack_taps(const float* w, int n, int npad, int cin0, int cin1, int ntaps, int bkc,
                   uint16_t* out, float* colsum);

int vdmi_tapgemm(const vdmi_tapgemm_args* args, vdmi_stream stream);

/* HOST. vdmi_tapgemm runs the 64 -> 64 channel (1,3,3) NORM_SILU convolutions of 32- and 64-pixel-wide images
 * (Block.proj + RMSNorm + SiLU, vdp.py:412-427) on a weight-stationary kernel; 0 keeps every launch on the
 * general tap-GEMM (tests and A/B timing; the environment variable VDMI_WSCONV=0 does the same for a process). */
void vdmi_wsconv_enable(int on);
/* 1 when vdmi_tapgemm would run this launch on the weight-stationary kernel, else 0 (for profiles and tools that
 * label launches by the kernel that runs them). */
int vdmi_wsconv_covers(const vdmi_tapgemm_args* a);

/* HOST. vdmi_tapgemm runs the plain-epilogue (1,3,3) convolutions of 8-, 16- and 32-pixel-wide images with 64-channel
 * input chunks and a multiple of 128 output channels (tile_bn = 64: Block.proj of the wide layers, vdp.py:412-420),
 * and the NORM_SILU convolutions with exactly 128 output channels on 32-pixel-wide images (Block, vdp.py:412-427),
 * on a streaming kernel (256-pixel x 128-channel tiles, patch and weights by LDS-DMA); 0 keeps them on the general
 * tap-GEMM (tests and A/B timing; the environment variable VDMI_DCONV=0 does the same for a process). */
void vdmi_dconv_enable(int on);      /* 2: also the RMSNorm-epilogue form on every width it is built for (tests) */
/* 1 when vdmi_tapgemm would run this launch on the streaming kernel, else 0. */
int vdmi_dconv_covers(const vdmi_tapgemm_args* a);

/* ------------------------------------------------------------------------------------------
 * Cross-attention to previous-frame features: CrossAttention (vdp.py:480-528) inside
 * Residual(PreNorm(.)) at vdp.py:803-805 and :824-826, heads of 32 features.
 *
 * Keys/values depend only on the context.  vdmi_xattn_context turns the fp32 context
 * [b][l][d] into bf16 rows [b*l][d], multiplied per sample by keep[b] (1 - the null-conditioning mask
 * of vdp.py:782-789; NULL = all ones).  The caller projects those rows with vdmi_tapgemm onto
 * [b*l][ldkv] (per layer: heads*32 key columns, then heads*32 value columns) and vdmi_xattn_pack_kv
 * reorders them into MFMA operand fragments, vdmi_xattn_kv_elems() bf16 elements in total, layer
 * `i` starting at element i * vdmi_xattn_kv_elems(1, b, l, heads).
 *
 * vdmi_xattn_core: q bf16 [b*rows_per_sample][ldq] (already LayerNormed, projected and multiplied by
 * dim_head^-0.5 * log2(e): vdmi_tapgemm with the QKV_ATTN epilogue and q_cols = n) -> o bf16
 * [b*rows_per_sample][heads*32] = softmax(q k^T) v per head.  Within a head's 32 output columns the
 * feature order is permuted: column 16*h + r holds feature (r&3) + 8*(r>>2) + 4*h  (h = 0,1; r = 0..15);
 * the out-projection weights must be packed with their input columns in the same order.
 * ------------------------------------------------------------------------------------------ */
size_t vdmi_xattn_kv_elems(int nlayers, int b, int l, int heads);                              /* HOST */
int vdmi_xattn_context(const float* ctx, const float* keep, void* rows, int b, int l, int d, vdmi_stream stream);
int vdmi_xattn_pack_kv(const void* kvrows, int ldkv, void* kvpack, int nlayers, int b, int l, int heads,
                       vdmi_stream stream);
int vdmi_xattn_core(const void* q, int ldq, void* o, const void* kvpack_layer, int b, int64_t rows_per_sample,
                    int l, int heads, vdmi_stream stream);

/* Training form of the same core (what loss.backward() derives for CrossAttention.forward, vdp.py:511-521, when the
 * frame-conditioning layers are trained, vdp.py:1146-1161, 1451).  vdmi_xattn_pack_kv_train packs one layer's projected
 * context rows into 2 * vdmi_xattn_kv_elems(1, b, l, heads) elements (the two operand forms of vdmi_xattn_pack_kv plus
 * V with the feature order of the core's output and K transposed).  vdmi_xattn_core_train_fwd is vdmi_xattn_core over
 * that pack and also writes lse float [b][heads][lse_stride] (lse_stride >= rows_per_sample): the log2-domain
 * log-sum-exp of every (query, head).  vdmi_xattn_core_train_bwd takes the same q, the forward's o and lse, the
 * gradient dout of o (same layout and feature order as o) and writes dq bf16 [b*rows_per_sample][heads*32] (gradient
 * of the scaled q rows, natural feature order), dkv bf16 [b*l][lddkv] (gradients of the key columns, then of the value
 * columns, of kvrows) and delta float [b][heads][lse_stride] (scratch).  `workspace`:
 * vdmi_xattn_bwd_workspace_bytes() bytes (partial key / value gradients, added in a fixed order: reproducible). */
int vdmi_xattn_pack_kv_train(const void* kvrows, int ldkv, void* kvpack, int b, int l, int heads, vdmi_stream stream);
int vdmi_xattn_core_train_fwd(const void* q, int ldq, void* o, const void* kvpack_train, float* lse, int64_t lse_stride,
                              int b, int64_t rows_per_sample, int l, int heads, vdmi_stream stream);
size_t vdmi_xattn_bwd_workspace_bytes(int b, int64_t rows_per_sample, int l, int heads);               /* HOST */
int vdmi_xattn_core_train_bwd(const void* q, int ldq, const void* o, const void* dout, const void* kvpack_train,
                              const void* kvrows, int ldkv, const float* lse, float* delta, int64_t lse_stride,
                              void* dq, void* dkv, int lddkv, int b, int64_t rows_per_sample, int l, int heads,
                              void* workspace, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm(groups, c) + SiLU of the VAE (vdp.py:93-108, 130-143) over channels-last bf16 rows
 * [nsamples * rows_per_sample][c]: statistics per (sample, group) over the sample's rows and the
 * group's c/groups channels (a multiple of 8), biased variance, eps inside the square root.
 * c: 64..2048 with 256 % (c/8) == 0.  `workspace` holds vdmi_groupnorm_workspace_bytes() bytes
 * (partial sums); x and out may alias.
 * ------------------------------------------------------------------------------------------ */
size_t vdmi_groupnorm_workspace_bytes(int nsamples, int64_t rows_per_sample, int c);      /* HOST */
int vdmi_groupnorm_silu(const void* x, void* out, int nsamples, int64_t rows_per_sample, int c, int groups,
                        const float* gamma, const float* beta, float eps, void* workspace, vdmi_stream stream);
/* vdmi_groupnorm_silu without a workspace argument, for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The
 * partial sums live in a static device array inside the library (VDMI_GROUPNORM_NOALLOC_SCRATCH_BYTES per GPU, held
 * until the process ends); the call allocates and frees nothing.  Inputs that need more than that, i.e.
 * vdmi_groupnorm_workspace_bytes(nsamples, rows_per_sample, c) > VDMI_GROUPNORM_NOALLOC_SCRATCH_BYTES, are rejected with
 * VDMI_E_SCRATCH.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race; calls on one stream do not.  The FIRST call on each GPU looks the array's address up
 * (hipGetSymbolAddress, not a stream operation): make it once, eagerly, before capturing the call into a HIP graph;
 * later calls are stream operations only.  `stream` must belong to the current device (VDMI_E_ARG otherwise). */
#define VDMI_GROUPNORM_NOALLOC_SCRATCH_BYTES (1 << 20)
int vdmi_groupnorm_silu_noalloc(const void* x, void* out, int nsamples, int64_t rows_per_sample, int c, int groups,
                                const float* gamma, const float* beta, float eps, vdmi_stream stream);
/* The general form: act 0 = none, 1 = SiLU; then an optional per-(sample, channel) affine
 * y * post_gain[b][ch] + post_shift[b][ch] (row stride post_bstride; the time-embedding scale + 1 / shift of
 * video_diffusion_guidance.py:378-382), then an optional residual add of bf16 rows [..][ld_res]. */
int vdmi_groupnorm(const void* x, void* out, int nsamples, int64_t rows_per_sample, int c, int groups,
                   const float* gamma, const float* beta, float eps, int act, const float* post_gain,
                   const float* post_shift, int64_t post_bstride, const void* residual, int ld_res, void* workspace,
                   vdmi_stream stream);

/* Backward of vdmi_groupnorm (the reference differentiates these layers through PyTorch autograd:
 * video_diffusion_guidance.py:334-344, 378-382 under the guidance sampler's torch.autograd.grad at :725-761).
 * x: the forward's input rows, dout: gradient rows [..][ld_dout] of the forward's output (the residual's gradient is dout
 * itself), dx: gradient rows [..][c] of x.  chan_sums: float [nsamples][c][4] = per sample and channel
 * {sum dy, sum dy * xhat, sum dout * act(y), sum dout}: summed over the samples the first two are dbeta and dgamma, the
 * last two are the gradients of post_gain / post_shift.  No atomics: results are the same every call.
 * `workspace`: vdmi_groupnorm_bwd_workspace_bytes() bytes. */
size_t vdmi_groupnorm_bwd_workspace_bytes(int nsamples, int64_t rows_per_sample, int c, int groups);      /* HOST */
int vdmi_groupnorm_bwd(const void* x, const void* dout, int ld_dout, void* dx, int nsamples, int64_t rows_per_sample,
                       int c, int groups, const float* gamma, const float* beta, float eps, int act,
                       const float* post_gain, int64_t post_bstride, float* chan_sums, void* workspace,
                       vdmi_stream stream);

/* Depthwise temporal convolution Conv3d(c, c, (k,1,1), padding (k/2,0,0), groups=c) of
 * video_diffusion_guidance.py:324-325 over channels-last bf16 rows [b*f*hw][ldx] -> [b*f*hw][ldo] (the first c
 * columns of each); w: float [c][k], bias: float [c] or NULL; k odd. */
int vdmi_temporal_dwconv(const void* x, int ldx, void* out, int ldo, int b, int f, int64_t hw, int c, int k,
                         const float* w, const float* bias, vdmi_stream stream);

/* Weight and bias gradient of vdmi_temporal_dwconv from its input rows x and output-gradient rows du (both [b*f*hw][c]):
 * part: float [*nsplit_out][c][k + 1] partial sums (vdmi_temporal_dwconv_wgrad_elems() floats); summed over the first
 * axis, [..][j] (j < k) is the gradient of w[ch][j] and [..][k] that of the bias.  k: 1, 3, 5 or 7.  (The gradient with
 * respect to x is vdmi_temporal_dwconv itself with the taps reversed.) */
size_t vdmi_temporal_dwconv_wgrad_elems(int b, int f, int64_t hw, int c, int k);                          /* HOST */
int vdmi_temporal_dwconv_wgrad(const void* x, const void* du, float* part, int* nsplit_out, int b, int f, int64_t hw,
                               int c, int k, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * Softmax attention core over a small token axis (temporal attention over frames with
 * relative-position bias, vdp.py:585-612; mid spatial attention, vdp.py:708-710).
 * qkv: bf16 [rows][3*heads*32] (q already scaled and rotated), out: bf16 [rows][heads*32].
 * Row m = outer*(n_tok*tok_stride) + i*tok_stride + inner; tokens of one group share
 * (outer, inner).  bias: float [heads][n_tok][n_tok] or NULL.  dim_head is 32.
 * focus: uint8 [rows / rows_per_sample] or NULL — the focus-present mask of vdp.py:590-600: in a
 * sample whose flag is set every token attends only to itself (output = its own v).
 * ------------------------------------------------------------------------------------------ */
int vdmi_attn_core(const void* qkv, void* out, int64_t rows, int heads, int n_tok, int64_t tok_stride,
                   const float* bias, const uint8_t* focus, int64_t rows_per_sample, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * Fused temporal attention: Residual(PreNorm(EinopsToAndFrom('b c f h w','b (h w) f c', Attention)))
 * of the reference in one kernel (vdp.py:331-337, 380-408, 531-612, 636): channel LayerNorm, to_qkv,
 * q scaling, rotary, q.k^T + relative-position bias, softmax over the f frames of each site, .v,
 * to_out and the residual add.  x, y: bf16 [b*f*hw][c] (y may not alias x); c in {32, 64, 128},
 * f <= 32, dim_head 32.  wpack: vdmi_pack_tattn output.  bias: float [heads][f][f] or NULL;
 * rot_cos/rot_sin: float [f][16] or NULL (no rotary).  focus: uint8 [b] or NULL, the focus-present
 * mask of vdp.py:590-600 (flagged samples: every frame attends only to itself).
 * ------------------------------------------------------------------------------------------ */
size_t vdmi_pack_tattn_elems(int c, int heads);                                           /* HOST */
int vdmi_pack_tattn(const float* wqkv_gamma_folded, const float* wout, int c, int heads, uint16_t* out);  /* HOST */
int vdmi_tattn_fused(const void* x, void* y, const void* wpack, const float* bias, const float* rot_cos,
                     const float* rot_sin, int b, int f, int hw, int c, int heads, float q_scale, float eps,
                     const uint8_t* focus, vdmi_stream stream);

/* The same layer for wide channel counts on small images (c in {256, 512}, heads = 8): one workgroup per two groups
 * of floor(32/f) sites x f frames, one wave per head, the head's weight fragments streamed from memory into
 * registers (they never pass through LDS).  Same arguments, same packed weights (vdmi_pack_tattn) and the same
 * arithmetic as vdmi_tattn_fused; which of the two is faster depends on the rows per launch (this one below
 * ~10k rows per sample batch of 8, where vdmi_tattn_fused would start too few workgroups). */
int vdmi_tattn_fused_wide(const void* x, void* y, const void* wpack, const float* bias, const float* rot_cos,
                          const float* rot_sin, int b, int f, int hw, int c, int heads, float q_scale, float eps,
                          const uint8_t* focus, vdmi_stream stream);

/* Residual(PreNorm(EinopsToAndFrom('b c f h w', 'b f (h w) c', Attention))) of the mid block (vdp.py:708-710, 531-612
 * without rotary and bias) in one launch, on the same kernel family: full softmax attention over the hw <= 64
 * positions of each frame, one workgroup per frame, one wave per head; x, y: bf16 [frames*hw][c], c in {256, 512},
 * heads = 8, wpack from vdmi_pack_tattn (to_qkv with the LayerNorm gamma folded in, to_out). */
int vdmi_sattn_fused_wide(const void* x, void* y, const void* wpack, int frames, int hw, int c, int heads,
                          float q_scale, float eps, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * Fused Residual(PreNorm(SpatialLinearAttention)) (vdp.py:454-479, 331-337, 380-408): three
 * launches (k/v + context per frame, context fold into to_out, q + output) with x read twice and
 * y written once.  x, y: bf16 [frames*hw][c]; c in {32, 64, 128}; heads <= 8; dim_head 32.
 * wpack: vdmi_pack_sla output; w_out: float [c][heads*32]; b_out: float [c];
 * workspace: vdmi_sla_workspace_bytes() bytes of device memory, 256-byte aligned.
 * ------------------------------------------------------------------------------------------ */
size_t vdmi_pack_sla_elems(int c, int heads);                                             /* HOST */
int vdmi_pack_sla(const float* wqkv_gamma_folded, int c, int heads, uint16_t* out);        /* HOST */
size_t vdmi_sla_workspace_bytes(int frames, int hw, int c, int heads);                     /* HOST */
int vdmi_sla_fused(const void* x, void* y, const void* wpack, const float* w_out, const float* b_out,
                   void* workspace, int frames, int hw, int c, int heads, float q_scale, float eps,
                   vdmi_stream stream);

/* The same layer in ONE launch for wide channel counts (c in {256, 512}, heads = 8): a workgroup per frame, a wave per
 * head, the 32 x 32 context kept in registers; frames of more than 64 positions are swept twice in pairs of 32-row
 * tiles (k, v and the context with an online softmax over the positions, then q and the output; c = 256 only).  wpack: vdmi_pack_tattn of to_qkv (with
 * the LayerNorm gamma folded in) and to_out; b_out: float [c] or NULL. */
int vdmi_sla_fused_wide(const void* x, void* y, const void* wpack, const float* b_out, int frames, int hw, int c,
                        int heads, float q_scale, float eps, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * SpatialLinearAttention pass 1 (vdp.py:471-476): per (frame, head) softmax of k over the
 * n = H*W positions, context = k·vᵀ (32x32), folded into the output projection:
 *   W'[frame][c][h*32+d] = sum_e to_out.weight[c][h*32+e] * context[h][d][e]
 * written in vdmi_pack_taps order (1 tap, cin = heads*32) so that pass 2 is vdmi_tapgemm with
 * per-frame weights over the softmaxed q.
 * kvT: bf16 [frames][2*heads*32][hw] from VDMI_EPI_QKV_SLA; w_out: float [c][heads*32].
 * ------------------------------------------------------------------------------------------ */
int vdmi_sla_context(const void* kvT, int frames, int heads, int hw, const float* w_out, int c, int cpad,
                     int bkc, void* w_folded, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * Unet3D.init_conv (vdp.py:644,772): Conv3d(cin -> cout, (1,k,k), pad k/2) on the float32
 * channels-first input x (B, cin, F, H, W); writes channels-last bf16 [B*F*H*W][cout].
 * w: float [cout][cin][k][k], bias: float [cout].
 * ------------------------------------------------------------------------------------------ */
int vdmi_init_conv(const float* x, const float* w, const float* bias, void* out,
                   int b, int cin, int f, int h, int wd, int cout, int ksize, vdmi_stream stream);

/* final_conv[1] (vdp.py:728): 1x1 conv bf16 [rows][cin] -> float32 channels-first (B, cout, F, H, W). */
int vdmi_final_conv(const void* in, const float* w, const float* bias, float* out,
                    int b, int cin, int f, int h, int wd, int cout, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * Time / condition embedding (vdp.py:339-351, 651-656, 778-781) and every ResnetBlock.mlp
 * (vdp.py:432-435, 446-448) in two launches.
 *   emb[b] = [ Linear(GELU(Linear(sinusoidal(t[b])))) , cond'[b] ]   (cond' = null_cond_emb where null_mask[b])
 *   for every block k: (scale, shift) = Linear_k(SiLU(emb[b]));  gain = gamma_k*(scale+1)
 * w_all: float [rows_total][emb_dim] = the blocks' mlp.1.weight stacked (scale rows then shift
 * rows per block), b_all likewise, gamma_all: float [rows_total/2] stacked RMSNorm gammas.
 * gain/shift out: float [B][rows_total/2].
 * ------------------------------------------------------------------------------------------ */
int vdmi_time_embed(const int64_t* t, int b, int dim, const float* w1, const float* b1, const float* w2,
                    const float* b2, const float* cond, const float* null_cond, const uint8_t* null_mask,
                    int cond_dim, float* emb_silu, float* emb_raw, vdmi_stream stream);
int vdmi_block_scale_shift(const float* emb_silu, int b, int emb_dim, const float* w_all, const float* b_all,
                           const float* gamma_all, const int32_t* block_offsets, const int32_t* block_channels,
                           int nblocks, int half_total, float* gain, float* shift, vdmi_stream stream);

/* Scale/shift tables of a sampling loop: the time embedding and every ResnetBlock.mlp (vdp.py:339-351, 432-448,
 * 651-656) depend on the timestep only when there is no condition vector, so a loop computes gain/shift for all T
 * timesteps once (vdmi_time_embed + vdmi_block_scale_shift with b = T) and each step copies its rows:
 * gain[i][:] = gain_table[t[i]][:], shift likewise.  Tables float [T][cols], t int64 [b] with 0 <= t[i] < T
 * (rows outside are left untouched), outputs float [b][cols]; cols a multiple of 4. */
int vdmi_gather_scale_shift(const float* gain_table, const float* shift_table, int T, const int64_t* t, int b, int cols,
                            float* gain, float* shift, vdmi_stream stream);

/* HOST. RelativePositionBias bucket ids (vdp.py:286-304), int32 [n][n], entry (i,j) = query i, key j. */
int vdmi_relpos_buckets(int n, int num_buckets, int max_distance, int32_t* out);

/* RMSNorm over channels, * gain, + shift, SiLU, optional + res: the tail of Block.forward (vdp.py:391-427:
 * F.normalize(x, dim=1) * gamma * sqrt(C), x * (scale + 1) + shift, SiLU) and ResnetBlock's identity
 * residual add (vdp.py:452), for layers whose convolution ran with VDMI_EPI_PLAIN over split
 * output-channel tiles.  x, out, res: bf16 [rows][ld*]; gain = gamma*(scale+1) and shift as in
 * vdmi_tapgemm_args (float, row's batch element = row / rows_per_batch, bstride 0 = shared vector);
 * c: 64 * {1,2,3,4,6,8,12,16,24,32}. */
int vdmi_rmsnorm_silu(const void* x, int ldx, void* out, int ldo, int64_t rows, int c, const float* gain,
                      int64_t gain_bstride, const float* shift, int64_t shift_bstride, int64_t rows_per_batch,
                      float norm_scale, const void* res, int ld_res, vdmi_stream stream);
/* The same over the sum of `nparts` (1..4) partial inputs x + i * part_stride (elements): the partial outputs of a
 * K-split vdmi_tapgemm are added in fp32 on the way in. */
int vdmi_rmsnorm_silu_sum(const void* x, int nparts, int64_t part_stride, int ldx, void* out, int ldo, int64_t rows, int c,
                          const float* gain, int64_t gain_bstride, const float* shift, int64_t shift_bstride,
                          int64_t rows_per_batch, float norm_scale, const void* res, int ld_res, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * One DDPM ancestral step (vdp.py:939-952, 985-1002; coefficient gather = extract, vdp.py:835-838):
 *   x0 = srac[t]*x - srmac[t]*eps;  x0 = clamp(x0,-1,1) if clip
 *   x' = c1[t]*x0 + c2[t]*x + (t != 0) * exp(0.5*logvar[t]) * z
 * x, eps, out: float32 [b][per_sample]; t: int64 [b] on the device.  z: float32 noise, or NULL to
 * draw it in-kernel (Philox4x32-10 + Box-Muller, keyed by seed and the element counter+offset).
 * ------------------------------------------------------------------------------------------ */
typedef struct vdmi_ddpm_coefs {
    const float* sqrt_recip_alphas_cumprod;
    const float* sqrt_recipm1_alphas_cumprod;
    const float* posterior_mean_coef1;
    const float* posterior_mean_coef2;
    const float* posterior_log_variance_clipped;
} vdmi_ddpm_coefs;

int vdmi_ddpm_step(const float* x, const float* eps, const float* z, const int64_t* t,
                   const vdmi_ddpm_coefs* coefs, float* out, int b, int64_t per_sample, int clip,
                   uint64_t seed, uint64_t offset, vdmi_stream stream);

/* The same step with in-kernel noise (the randn_like of vdp.py:1000) whose Philox key and counter offset are read
 * from device memory: rng_state = uint64 [2] = {seed, offset}, 8-byte aligned.  A sampling loop captured in a HIP graph
 * gets fresh noise on every call by rewriting these two words before it replays; the counter of element group g of
 * sample i at timestep t is (t * b + i) * groups_per_sample + g + offset, so within one loop no counter repeats. */
int vdmi_ddpm_step_rng(const float* x, const float* eps, const int64_t* t, const vdmi_ddpm_coefs* coefs,
                       float* out, int b, int64_t per_sample, int clip, const uint64_t* rng_state, vdmi_stream stream);

/* vdmi_ddpm_step_rng for a loop replayed from a captured HIP graph (p_sample_loop, vdp.py:1004-1014: t runs from
 * T - 1 down to 0): after the step, the workgroup that finishes last replaces every t[i] by t[i] - 1, or by
 * timesteps - 1 when t[i] was 0, so the host never writes t between replays.  done_counter: one uint32 in device
 * memory, zero before the first call (the kernel leaves it at zero).  The b samples may be samples
 * [batch_offset, batch_offset + b) of a batch of batch_total (0: b) that other calls step independently - samples never
 * interact in the loop - and they then draw the noise one call over the whole batch would draw: sample i counts as
 * (t * batch_total + batch_offset + i). */
int vdmi_ddpm_step_loop(const float* x, const float* eps, int64_t* t, const vdmi_ddpm_coefs* coefs, float* out, int b,
                        int64_t per_sample, int clip, const uint64_t* rng_state, uint32_t* done_counter, int timesteps,
                        int batch_total, int batch_offset, vdmi_stream stream);

/* Dynamic thresholding of the predicted x0 (vdp.py:957-968, 987-997; use_dynamic_thres of GaussianDiffusion):
 *   thres[i] = max(1, torch.quantile(|x0_i|, percentile))   over the per_sample elements of sample i,
 * x0 = srac[t]*x - srmac[t]*eps as in vdmi_ddpm_step, the quantile with torch's linear interpolation between the
 * order statistics of ranks floor / ceil of percentile * (per_sample - 1) (float32 rank arithmetic, per_sample < 2^24).
 * The order statistics are exact (radix select over the float bit patterns).  thres: float [b]. */
int vdmi_dynamic_threshold(const float* x, const float* eps, const int64_t* t, const vdmi_ddpm_coefs* coefs,
                           float* thres, int b, int64_t per_sample, float percentile, vdmi_stream stream);
/* vdmi_ddpm_step / vdmi_ddpm_step_rng with x0 = clamp(x0, -thres[i], thres[i]) / thres[i] in place of the clip to
 * [-1, 1] (vdp.py:966-968).  Exactly one of z (float32 noise) and rng_state (see vdmi_ddpm_step_rng) is non-NULL. */
int vdmi_ddpm_step_thres(const float* x, const float* eps, const float* z, const int64_t* t,
                         const vdmi_ddpm_coefs* coefs, const float* thres, float* out, int b, int64_t per_sample,
                         const uint64_t* rng_state, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * Backward pieces of ResnetBlock (the gradients torch autograd derives for vdp.py:391-452 when the
 * reference's Trainer calls loss.backward(), vdp.py:1451).  Activations and their gradients are
 * channels-last bf16 rows as everywhere else; parameter gradients are float32 and are ACCUMULATED
 * (the caller zeroes them), in the layout of the reference's parameters.
 * ------------------------------------------------------------------------------------------ */

/* vdmi_pack_taps on the device (same output, bit for bit, for transposed = 0): w float [n][cin0+cin1][ntaps] in
 * device memory.  transposed = 1 packs the weight set of the convolution's input gradient instead: w is then the
 * forward weight float [cin0+cin1][n][ntaps] (its OUTPUT channels are this pass's contraction axis) and the taps
 * are taken in reverse order, so that with the same (dy, dx) table
 *   dx[p][n'] = sum_{tap, k} w[k][n'][ntaps-1-tap] * dyout[p + (dy,dx)_tap][k]
 * is the gradient of a stride-1 "same" convolution whose tap table is point-symmetric (the (1,3,3) and 1x1
 * convolutions of ResnetBlock). */
int vdmi_pack_taps_device(const float* w, int n, int npad, int cin0, int cin1, int ntaps, int bkc, int transposed,
                          void* out, vdmi_stream stream);

/* Many weight sets in ONE launch (a training step packs every weight of the model again after each optimizer step: some
 * two hundred sets, most of them a few microseconds of work).  Each job is one vdmi_pack_taps_device call; in_scale
 * (device float [cin0 + cin1] over the FORWARD weight's input channels, or NULL) multiplies every element by the factor of
 * its input channel - the LayerNorm gamma that the attention layers fold into to_qkv (vdp.py:386-389, 546) - so the
 * folded weight never exists as a tensor.  gather (device int32 [n][cin0 + cin1][ntaps], or NULL): the set's source tensor
 * is a re-arrangement of w, element e of it being w[gather[e]] - the space-to-depth view of Downsample's (1,4,4) kernel
 * and the four per-parity tap sets of Upsample's (vdp.py:374-378) come straight from the parameter.  `jobs` is host memory; device_table: vdmi_pack_taps_batch_table_bytes(count)
 * bytes of device memory that the launch reads (keep it until the stream has passed the call).  Has no counterpart in
 * the reference (torch keeps one weight layout). */
typedef struct vdmi_pack_job {
    const float* w;
    const float* in_scale;
    const int32_t* gather;
    void* out;
    int n, npad, cin0, cin1, ntaps, bkc, transposed;
} vdmi_pack_job;
size_t vdmi_pack_taps_batch_table_bytes(int count);
int vdmi_pack_taps_batch(const vdmi_pack_job* jobs, int count, void* device_table, vdmi_stream stream);

/* rows bf16 [nf*hw][ld], channel window [c_off, c_off + c)  ->  out bf16 [ceil(nf/16)][ceil(c/32)][hw][32][16]
 * (frame block, channel block, pixel, channel in block, frame in block), frames beyond nf and channels beyond c zero:
 * the operand layout of vdmi_conv_wgrad (the frame is the fast contraction index, so a tap shift never breaks a
 * 16-byte fragment, and the 32 channels x 16 frames a wave needs for one pixel are one contiguous KiB).  out holds
 * vdmi_rows_transpose_elems() elements. */
size_t vdmi_rows_transpose_elems(int c, int nf, int64_t hw);                              /* HOST */
int vdmi_rows_transpose(const void* rows, int ld, int c_off, int c, int nf, int64_t hw, void* out, vdmi_stream stream);

/* Weight gradient of a stride-1 (1,k,k) convolution (torch's conv3d weight gradient, vdp.py:415, 439):
 *   dw[co][cin_off + ci][tap] = sum over frames f and pixels p of dy[f][p][co] * x[f][p + (dy,dx)_tap][ci]
 * (zero outside the image; written, not accumulated: dw needs no initialisation).  xT: vdmi_rows_transpose of the input rows (cin channels), dyT: of the output-gradient
 * rows (cout channels), both over nf frames of h x w pixels; dw float [cout][cin_total][ntaps]; ntaps 1, or 9 with
 * the row-major tap table (ky - 1, kx - 1).  workspace: vdmi_conv_wgrad_workspace_bytes() bytes (partial tiles of
 * the workgroups that share a (co, ci) tile, added by a second launch in a fixed order: results are repeatable). */
size_t vdmi_conv_wgrad_workspace_bytes(int cin, int cout, int nf, int h, int w, int ntaps);   /* HOST */
int vdmi_conv_wgrad(const void* xT, const void* dyT, float* dw, int cin, int cout, int cin_total, int cin_off,
                    int nf, int h, int w, int ntaps, const int8_t* dy, const int8_t* dx, void* workspace,
                    vdmi_stream stream);

/* Weight gradient of a 1x1 convolution / Linear layer straight from the channels-last rows (torch's weight gradient of
 * to_qkv, to_out, res_conv: vdp.py:439, 460-461, 560-561), without the transposed copies:
 *   dw[co][cin_off + ci] = sum over rows r of dy[r][co] * x[r][ci]          (written, not accumulated)
 * x bf16 [rows][ldx], dy bf16 [rows][lddy]; cin, cout multiples of 8; dw float [cout][cin_total]; workspace:
 * vdmi_rows_wgrad_workspace_bytes() bytes (partial tiles, added in a fixed order: results are repeatable). */
size_t vdmi_rows_wgrad_workspace_bytes(int cin, int cout, int64_t rows);                   /* HOST */
int vdmi_rows_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, int cin, int cout, int cin_total,
                    int cin_off, int64_t rows, void* workspace, vdmi_stream stream);

/* The (1,3,3) weight gradient straight from the channels-last rows, without transposed copies: x [nf*h*w][ldx] (the
 * convolution's input rows, cin channels from column 0), dy [nf*h*w][lddy] (gradient of its output rows), both bf16;
 * dw float [cout][cin_total][3][3], written at input channels cin_off .. cin_off + cin (not accumulated).  cin, cout,
 * ldx, lddy multiples of 8.  workspace: vdmi_conv_wgrad_rows_workspace_bytes() bytes (0: shape not taken).
 * Replaces the weight-gradient half of F.conv3d's backward for Block.project (vdp.py:387, 391-398). */
size_t vdmi_conv_wgrad_rows_workspace_bytes(int cin, int cout, int nf, int h, int w);   /* HOST */
int vdmi_conv_wgrad_rows(const void* x, int ldx, const void* dy, int lddy, float* dw, int cin, int cout, int cin_total,
                         int cin_off, int nf, int h, int w, void* workspace, vdmi_stream stream);

/* Weight gradient of the (1,4,4) / stride 2 / pad 1 convolution between a big image (h x w, even) and a small one
 * (h/2 x w/2), straight from the channels-last rows:
 *   dw[cs][cb][ky][kx] = sum over frames and small pixels q of small[q][cs] * big[2 q + (ky - 1, kx - 1)][cb]
 * big [nf*h*w][ldb] (cb channels), small [nf*(h/2)*(w/2)][lds] (cs channels), both bf16; dw float [cs][cb][4][4],
 * written.  Downsample (vdp.py:377-378): big = its input rows, small = its output gradient, dw = the weight gradient
 * (cout, cin, 1, 4, 4); Upsample's ConvTranspose3d (vdp.py:374-375): big = its output gradient, small = its input rows,
 * dw = the weight gradient (cin, cout, 1, 4, 4).  cb, cs, ldb, lds multiples of 8.
 * workspace: vdmi_strided_wgrad_rows_workspace_bytes() bytes (0: shape not taken). */
size_t vdmi_strided_wgrad_rows_workspace_bytes(int cb, int cs, int nf, int h, int w);   /* HOST */
int vdmi_strided_wgrad_rows(const void* big, int ldb, const void* small, int lds, float* dw, int cb, int cs, int nf, int h,
                            int w, void* workspace, vdmi_stream stream);

/* The same gradient for a (1,kh,kw) "same" convolution with few input channels (Unet3D.init_conv, vdp.py:644: 3 -> dim,
 * 7 x 7): a column of the contraction is a (ci, ky, kx) triple, so no lane idles on padding channels.  kh, kw odd;
 * dw float [cout][cin][kh][kw], written (not accumulated); workspace: vdmi_conv_wgrad_gather_workspace_bytes() bytes. */
size_t vdmi_conv_wgrad_gather_workspace_bytes(int cin, int cout, int nf, int h, int w, int kh, int kw);   /* HOST */
int vdmi_conv_wgrad_gather(const void* xT, const void* dyT, float* dw, int cin, int cout, int nf, int h, int w,
                           int kh, int kw, void* workspace, vdmi_stream stream);

/* Backward of the tail of Block.forward (vdp.py:391-398, 421-427):
 *   h = SiLU( F.normalize(y, dim=channels) * sqrt(c) * gamma * (scale[b] + 1) + shift[b] )
 * y: the convolution's output rows [batch*rows_per_batch][ldy] (what the forward normalised), dh: gradient of h;
 * writes dy (gradient of y) and accumulates dgamma [c], dbias [c] (= column sums of dy, the convolution's bias
 * gradient; may be NULL), dscale / dshift [batch][c] (NULL when scale / shift are NULL: a block without a time
 * embedding).  scale / shift: float [batch][sstride].  c: a power of two in 64..2048.  The accumulated outputs are
 * added across workgroups with float atomics: their last bits differ from run to run (dy is reproducible). */
int vdmi_rmsnorm_silu_bwd(const void* y, int ldy, const void* dh, int ldh, void* dy, int lddy, int batch,
                          int64_t rows_per_batch, int c, const float* gamma, const float* scale, const float* shift,
                          int64_t sstride, float norm_scale, float* dgamma, float* dbias, float* dscale, float* dshift,
                          vdmi_stream stream);

/* The same gradients without float atomics: every workgroup leaves its four per-channel sums in `workspace`
 * (vdmi_rmsnorm_silu_bwd_workspace_bytes() bytes, 16-byte aligned; 0 for a shape the kernel does not take) and a second
 * launch adds them in a fixed order, so dgamma, dbias, dscale and dshift are WRITTEN (no zero fill needed) and are the
 * same bits on every run.  All other arguments as above. */
size_t vdmi_rmsnorm_silu_bwd_workspace_bytes(int batch, int64_t rows_per_batch, int c);   /* HOST */
int vdmi_rmsnorm_silu_bwd_ws(const void* y, int ldy, const void* dh, int ldh, void* dy, int lddy, int batch,
                             int64_t rows_per_batch, int c, const float* gamma, const float* scale, const float* shift,
                             int64_t sstride, float norm_scale, float* dgamma, float* dbias, float* dscale, float* dshift,
                             void* workspace, vdmi_stream stream);

/* out[ch] += sum over rows of x[row][ch]  (bias gradient of res_conv, vdp.py:439); x bf16 [rows][ld], c <= 2048. */
int vdmi_colsum(const void* x, int ld, int64_t rows, int c, float* out, vdmi_stream stream);
/* out[ch] = the same sum, WRITTEN, without float atomics: per-workgroup partial sums in `workspace`
 * (vdmi_colsum_workspace_bytes() bytes, 16-byte aligned) added in a fixed order by a second launch - the same bits on
 * every run. */
size_t vdmi_colsum_workspace_bytes(int64_t rows, int c);   /* HOST */
int vdmi_colsum_ws(const void* x, int ld, int64_t rows, int c, float* out, void* workspace, vdmi_stream stream);

/* ------------------------------------------------------------------------------------------
 * Training kernels of the temporal attention layer, Residual(PreNorm(EinopsToAndFrom('b c f h w', 'b (h w) f c',
 * Attention))) (vdp.py:380-389, 531-612, 636), forward and backward.  The projections are vdmi_tapgemm row GEMMs
 * (weights from vdmi_pack_taps_device, weight gradients from vdmi_conv_wgrad); these are the pieces between them.
 * ------------------------------------------------------------------------------------------ */

/* Channel LayerNorm without gamma (vdp.py:386-389; gamma is folded into to_qkv): xh = (x - mean) / sqrt(var + eps),
 * biased variance; bf16 rows [rows][c], c a power of two in 64..2048.  Backward: from the layer's input rows x and the
 * gradient of xh,  dx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)) + add  (add: optional rows, e.g. the residual
 * branch's gradient). */
int vdmi_layernorm_rows(const void* x, void* xh, int64_t rows, int c, float eps, vdmi_stream stream);
int vdmi_layernorm_rows_bwd(const void* x, const void* dxh, const void* add, void* dx, int64_t rows, int c, float eps,
                            vdmi_stream stream);

/* Attention over the f frames of every site (vdp.py:560-606 without the focus mask), heads of 32 features, on the RAW
 * rows of to_qkv: qkv bf16 [b*f*hw][3*heads*32] (q | k | v, column = head*32 + feature; row = (b*f + frame)*hw + site).
 * q * scale, half-split rotary on q and k (rot_cos / rot_sin: float [f][16] or NULL), + bias (float [heads][f][f],
 * entry (query, key), or NULL), softmax over the keys, . v  ->  o bf16 [rows][heads*32], lse float [rows][heads]
 * (log-sum-exp of each row's scores, kept for the backward).  f <= 256: a workgroup takes as many heads at a time as
 * 256 threads and the LDS hold.  With hw = 1 the "frames" of a site are f consecutive rows: the mid-block spatial
 * attention over the positions of a frame (vdp.py:708-710; b = frames, f = positions, no rotary, no bias).
 * Backward: from qkv, o, lse and dout (gradient of o) the gradient of the raw qkv rows (dqkv, same layout) and of the
 * bias (dbias float [heads][f][f], ACCUMULATED; may be NULL).  focus: uint8 [b] or NULL - the focus-present mask of
 * vdp.py:590-600: in a flagged sample every frame attends to itself only (scores off the diagonal are masked).
 * Frame axes of up to 32 run on the matrix cores (tiles of 32 tokens = 32 / f whole sites, one wave per (tile, head);
 * the rotated q and k, the probabilities and the score gradients are rounded to the storage type as matrix operands);
 * that form sums the bias gradient in registers and needs `workspace` (vdmi_tattn_core_bwd_workspace_bytes() bytes of
 * device memory, 16-byte aligned, contents irrelevant) for the per-wave partial sums, which a second launch adds in a
 * fixed order.  Without a workspace, and for longer axes, the backward is the vector kernel, whose bias gradient uses
 * float atomics.  VDMI_TATTN_CORE_MFMA=0 / vdmi_tattn_core_mfma_enable(0) keep every call on the vector kernels. */
/* HOST. 1 when vdmi_tattn_core_fwd (bwd = 0) / vdmi_tattn_core_bwd (bwd = 1; with_dbias: a relative-position-bias
 * gradient is asked for) can run a token axis of f positions with `heads` heads: the site's tiles and, in the
 * backward, the bias-gradient table [heads][f][f] must fit 160 KiB of LDS (with 8 heads the backward ends near
 * f = 62), and f <= 256.  Callers fall back to torch ops for longer axes. */
int vdmi_tattn_core_fits(int bwd, int f, int heads, int with_dbias);
int vdmi_tattn_core_fwd(const void* qkv, void* o, float* lse, const float* bias, const float* rot_cos,
                        const float* rot_sin, const uint8_t* focus, int b, int f, int hw, int heads, float scale,
                        vdmi_stream stream);
int vdmi_tattn_core_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, float* dbias,
                        const float* bias, const float* rot_cos, const float* rot_sin, const uint8_t* focus, int b, int f,
                        int hw, int heads, float scale, void* workspace, vdmi_stream stream);
/* HOST. Bytes of `workspace` for that shape (0 when the shape runs on the vector kernel). */
size_t vdmi_tattn_core_bwd_workspace_bytes(int b, int f, int hw, int heads);
/* HOST. Tests / A-B runs: 0 keeps the attention cores on the vector kernels, 1 allows the matrix-core form again. */
void vdmi_tattn_core_mfma_enable(int on);

/* ------------------------------------------------------------------------------------------
 * Training kernels of Residual(PreNorm(SpatialLinearAttention)) (vdp.py:454-479), forward and backward, on the RAW rows of
 * to_qkv: qkv bf16 [frames*n][3*heads*32] (q | k | v, column = head*32 + feature, row = frame*n + position), heads of 32
 * features, heads in {1, 2, 4, 8}.  Per (frame, head): Kt = softmax of k over the n positions, Qt = softmax of q over the
 * 32 features times scale, ctx = Kt^T v (32 x 32), out = Qt ctx.
 *   vdmi_sla_train_reduce   sums over the positions (per-split partials in `partials`, scratch of
 *                           vdmi_sla_train_partial_bytes() bytes, then one combined record per (frame, head) in
 *                           `combined`, vdmi_sla_train_combined_bytes() bytes):
 *                           mode 0: a = the k columns, b = the v columns  -> ctx with the softmax's maximum / weight sum;
 *                           mode 1: a = the saved Qt rows, b = the gradient of out, combined_ctx = the mode-0 record
 *                                   -> the gradient of ctx and S[d] = sum_e dctx[d][e] ctx[d][e]
 *   vdmi_sla_train_rows_fwd out rows and Qt rows [frames*n][heads*32] from qkv and the mode-0 record
 *   vdmi_sla_train_rows_bwd gradient of the raw qkv rows from qkv, Qt, the gradient of out and both records
 * The projections around them (to_qkv, to_out + bias + residual) and their gradients are vdmi_tapgemm / vdmi_conv_wgrad /
 * vdmi_colsum, the LayerNorm vdmi_layernorm_rows.
 * ------------------------------------------------------------------------------------------ */
size_t vdmi_sla_train_partial_bytes(int frames, int n, int heads);                         /* HOST */
size_t vdmi_sla_train_combined_bytes(int frames, int heads);                               /* HOST */
int vdmi_sla_train_reduce(const void* a, int lda, const void* b, int ldb, void* partials, const void* combined_ctx,
                          void* combined, int frames, int n, int heads, int mode, vdmi_stream stream);
int vdmi_sla_train_rows_fwd(const void* qkv, const void* combined_ctx, void* out, void* qt, int frames, int n, int heads,
                            float scale, vdmi_stream stream);
int vdmi_sla_train_rows_bwd(const void* qkv, const void* qt, const void* dout, const void* combined_ctx,
                            const void* combined_dctx, void* dqkv, int frames, int n, int heads, float scale,
                            vdmi_stream stream);
/* HOST. Tests / A-B runs: 0 keeps vdmi_sla_train_rows_fwd / _bwd on the vector kernels (one lane per (row, head)), 1 allows
 * the matrix-core form again (tiles of 32 rows of one head, the 32 x 32 context as a high + low pair of 16-bit operands);
 * VDMI_SLA_ROWS_MFMA=0 in the environment does the same.  No counterpart in the reference. */
void vdmi_sla_rows_mfma_enable(int on);

#ifdef __cplusplus
}
#endif
#endif /* VDMI_H */
