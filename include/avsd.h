/*
 * avsd.h — C ABI of libavsd_hip.so: the MI355X (gfx950) kernels behind the AVSyncD
 * denoising path (per-step AudioUNet3D forward, CFG + scheduler update, VAE decode).
 *
 * The reference (lzhangbj/ASVA) has no FFI layer: its hot path is PyTorch module calls.
 * Each entry point below names the reference call site (file:line under /root/reference)
 * whose arithmetic it replaces.  The host side (the asva_amd Python package) binds these with ctypes;
 * INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns 0 on success, a negative AVSD_E* code otherwise;
 *     avsd_last_error() returns a thread-local message for the last failure.
 *   - all pointers are DEVICE pointers owned by the caller (torch tensors) unless the
 *     parameter name ends in _host.  Nothing here allocates or frees device memory, and
 *     nothing synchronises: every launch goes to `stream` (a hipStream_t), so the whole
 *     step is capturable in a hipGraph.
 *   - "bf16" buffers are raw uint16 bfloat16; statistics / biases / norm parameters are f32.
 *   - activations are channels-last: a (B, F, H, W, C) tensor is the row-major matrix
 *     [M = B*F*H*W, C]; "ld*" arguments are row strides in ELEMENTS.
 */
#ifndef AVSD_H
#define AVSD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVSD_OK 0
#define AVSD_EINVAL (-1)   /* bad argument (shape / alignment / null pointer)   */
#define AVSD_ELAUNCH (-2)  /* hipLaunch / runtime error                          */
#define AVSD_ENODEV (-3)   /* no gfx950 device / wrong architecture              */

#define AVSD_ABI_VERSION 10  /* v10: AVSD_GEMM_OUT_REST, AVSD_GEMM_X2 with out_master; v9: tile ids 67 (128 x 320 asm tile) and 70 (A-resident N-streaming tile), AVSD_GEMM_W_FRAG */

/* ---- library ------------------------------------------------------------------------ */
int avsd_abi_version(void);
/* "bf16" or "fp16": the 16-bit storage type this build of the library computes in. */
const char* avsd_precision(void);
const char* avsd_last_error(void);
/* Fills name[<=len] with the device arch string (e.g. "gfx950:sramecc+:xnack-"), and the
 * CU count.  AVSD_ENODEV when no device is visible. */
int avsd_device_info(char* name_host, int len, int* num_cu_host);

/* ---- GEMM family ---------------------------------------------------------------------
 * out[M, N] = epilogue( alpha * A'[M, K] . W[N, K]^T )
 * W is always [N][ldw] (K contiguous) — torch nn.Linear / packed conv weight layout.
 * A' is produced by the A-loader selected by `mode`:
 *
 *   AVSD_GEMM_PLAIN  A'[m, k] = k < k_split ? A[m*lda + k] : A2[m*lda2 + (k - k_split)]
 *                    (k_split == K, A2 == NULL for a single source).  Replaces nn.Linear /
 *                    1x1 nn.Conv2d: utils.py:123-131,159 (q/k/v/out), proj_in/proj_out
 *                    ff_spatio_audio_temp_transformer_3d.py:66,92, GEGLU FeedForward :276,
 *                    conv_shortcut ff_spatio_temp_resnet_3d.py:159 on a torch.cat input
 *                    (unet_3d_blocks.py:358,1038).
 *   AVSD_GEMM_TMIX   the temporal mixing linear of FFInflatedConv3d (utils.py:43-53):
 *                    K = 3*cseg; for row m = (b, f, p) (p in [0,hw)) segment s of K reads
 *                    row (b, g_s(f), p) of A with g = {0, max(f-1,0), f}.
 *   AVSD_GEMM_CONV3  implicit-GEMM 3x3 pad-1 convolution over channels-last images
 *                    (utils.py:37-38 -> nn.Conv2d): K = 9*cin (tap-major, cin-minor),
 *                    row m = (n, ho, wo); stride 1 or 2; `ups`=1 reads the input through a
 *                    nearest x2 upsample (ff_spatio_temp_resnet_3d.py:48); `pad` = top/left padding.
 *                    `ups`=2 (v10) is the SUB-PIXEL form of the same upsample + convolution: output pixel (2y + dy, 2x + dx) of the
 *                    upsampled 3x3 convolution only sees input rows y + dy - 1, y + dy and columns x + dx - 1, x + dx, so it is a 2x2
 *                    convolution on the ORIGINAL image with the kernel rows / columns that land on the same input pixel summed
 *                    (asva_amd/weights.py subpixel_conv3x3): the same function with 4/9 of the multiplies.  Descriptor: rows
 *                    m = (img, y, x) over the INPUT image (ho = hs, wo = ws), K = 4*cin (tap (i, j)-major: input pixel
 *                    (y + dy - 1 + i, x + dx - 1 + j)), N = 4*cout with column n = (2 dy + dx) * cout + co and W [4*cout][4*cin] likewise
 *                    (bias [4*cout] = the layer's bias four times); the epilogue stores element (m, n) to channel co of OUTPUT pixel
 *                    (img, 2y + dy, 2x + dx) of out [n_img*2hs*2ws][ldc] (out_master / the rest plane likewise).  stride 1, pad 1,
 *                    cin % 64 == 0, cout % 64 == 0 and cout a multiple of the tile's column width; bias only (no residual, row vector,
 *                    activation, statistics); LDS-direct tiles (4..38), AVSD_GEMM_X2 included.
 *
 * epilogue, in f32:  v = act(alpha*acc + bias[n] + rowvec[(m / rows_per_vec)*ldv + n])
 *                                   + res1[m*ldr1 + n] + res2[m*ldr2 + n]
 *   AVSD_GEMM_GELU:  act = gelu_erf (the MLP of the audio front-end's ViT blocks); identity otherwise.
 *   AVSD_GEMM_GEGLU: W rows are packed per 32-row block as [16 value rows | 16 gate rows];
 *                    out[M, N/2] = value * gelu_erf(gate)   (diffusers GEGLU)
 *   AVSD_GEMM_OUT_F32: out is f32 instead of 16-bit.
 *   AVSD_GEMM_RES1_F32 / RES2_F32: res1 / res2 are f32 [M][ldr] instead of 16-bit.  With `out_master` != NULL the epilogue
 *                    also stores the UN-ROUNDED f32 result to out_master[m*ldm + n]: together they keep a residual
 *                    stream (h = h + f(h), ff_spatio_audio_temp_transformer_3d.py:300-371; x + h, ff_spatio_temp_resnet_3d.py:189)
 *                    in f32 while the 16-bit copy in `out` feeds the next matrix multiply.
 * "16-bit" = bfloat16 in libavsd_hip.so, IEEE half in libavsd_hip_f16.so (the same sources built with -DAVSD_F16=1;
 *   avsd_precision() tells which); accumulation and the epilogue are f32 in both.
 *   AVSD_GEMM_KROT:  scheduling hint for the asm tiles (gemm4.hip): row bands start their K walk at different tiles and wrap (the weights of
 *                    the low-resolution layers stream from HBM: a lockstep walk is one chain of round trips).  Deterministic; the f32
 *                    summation order of a band then starts mid-K, so results differ from the unrotated walk in the last bits.
 *   AVSD_GEMM_XCD_N: scheduling hint, no effect on the result — tiles are dealt to the 8 XCDs in bands of N instead of
 *                    bands of M, so the weights (not the activations) are the operand each L2 fetches only once.
 *   AVSD_GEMM_W_FRAG: W is stored in MFMA-fragment order instead of [N][ldw]: [N / 32][K / 16][64][8] 16-bit values with element
 *                    (f, s, l, e) = W[32 f + (l & 31)][16 s + 8 (l >> 5) + e] (asva_amd/weights.py pack_frag) — one k-step of one 32-column
 *                    fragment is 1 KB, contiguous.  Only tile AVSD_GEMM_TILE_NSTREAM reads it (and refuses anything else); ldw is ignored.
 *   AVSD_GEMM_OUT_REST: a ONE-pass product whose 16-bit output also gets its rest plane: besides main = round16(v) at `out`, the epilogue
 *                    stores rest = round16(v - main) at `out + out_lo` elements (same strides) — the (main, rest) pair a three-pass product
 *                    (AVSD_GEMM_X2) reads as its A operand, written by the producer of the tensor instead of by a separate avsd_split_f32 pass
 *                    over its f32 master (per-layer precision plan: the ResBlock / transformer outputs that feed the three-pass shortcut and
 *                    sampler convolutions, ff_spatio_temp_resnet_3d.py:30-62,88-96,159).  16-bit output only, not with GEGLU; LayerNorm row
 *                    statistics (ROWSTATS) stay those of the main plane, which is what the one-pass consumers read.
 * blockIdx.z batches: pointers advance by batch_stride_* elements (0 = shared).
 */
enum { AVSD_GEMM_PLAIN = 0, AVSD_GEMM_TMIX = 1, AVSD_GEMM_CONV3 = 2 };
enum { AVSD_GEMM_GEGLU = 1, AVSD_GEMM_OUT_F32 = 2, AVSD_GEMM_GELU = 4, AVSD_GEMM_XCD_N = 8, AVSD_GEMM_ROWSTATS = 16,
       AVSD_GEMM_LNFUSE = 32, AVSD_GEMM_RES1_F32 = 64, AVSD_GEMM_RES2_F32 = 128, AVSD_GEMM_X2 = 256, AVSD_GEMM_KROT = 512, AVSD_GEMM_W_FRAG = 1024,
       AVSD_GEMM_OUT_REST = 2048 };
#define AVSD_GEMM_MAX_TILE 33
#define AVSD_GEMM_MAX_TILE_X2 36   /* AVSD_GEMM_X2 also has tiles 34..36 (gemm.hip dispatch_tile_x2) */
/* 256 x 160 LDS-direct tile, 8 MFMA + 4 loader waves (gemm.hip dispatch_tile; not with AVSD_GEMM_X2) */
#define AVSD_GEMM_TILE_256x160_8W 38
/* 3x3 stride-1 pad-1 convolution tiles with the input tile resident in LDS (conv3r.hip): K is walked chunk-major (64 input
 * channels at a time), the BM output rows plus one image row of halo on either side are staged once per chunk and read by all
 * nine taps; only the weight tile streams per K tile.  CONV3 descriptors with stride 1, ups 0, pad 1, one source,
 * cin % 64 == 0, image width <= 32 and a tile of whole image rows / whole images (avsd_gemm_conv3r_supported); split_k cuts
 * the channel chunks (split_k <= cin / 64).  No AVSD_GEMM_X2 / GEGLU / LNFUSE.  Other descriptors are refused with these ids. */
/* 4-wave tiles with a hand-scheduled (inline-asm) main loop, register-staged operands (gemm4.hip): 60 = 256 x 256, 61 = 256 x 128,
 * 62 = 128 x 256, 63 = 128 x 128, 64 = 128 x 64, 65 = 64 x 128, 66 = 64 x 64, 67 = 128 x 320 (PLAIN / TMIX, no AVSD_GEMM_X2).  PLAIN single-source descriptors with K % 64 == 0, no AVSD_GEMM_X2; every epilogue flag; split_k. */
#define AVSD_GEMM_TILE_ASM_FIRST 60
#define AVSD_GEMM_TILE_ASM_LAST 67
/* A-resident, N-streaming tile (nstream.hip): the workgroup keeps its 96-row band of A (all of K) in LDS and its 8 independent waves
 * stream W fragments straight from memory into MFMA operands — for the wide short-K projections (the GEGLU projection of a transformer
 * block: K = 320 / 640, N = 8 K).  PLAIN single-source descriptors with AVSD_GEMM_W_FRAG, K = 320 or 640, N % 32 == 0 (built for N % 256 == 0: 8 waves x whole fragments); every epilogue
 * flag except the position tables; no split_k, no AVSD_GEMM_X2. */
#define AVSD_GEMM_TILE_NSTREAM 70
#define AVSD_GEMM_TILE_CONV3R_FIRST 40
#define AVSD_GEMM_TILE_CONV3R_LAST 49
/* the same convolution with RECTANGULAR resident tiles (TH image rows x 32 pixels + a one-pixel halo, positions outside the image
 * zero-filled at load time) for images wider than 32 pixels: width % 32 == 0, height % TH == 0 (avsd_gemm_conv3r2d_supported);
 * otherwise as the ids above. */
#define AVSD_GEMM_TILE_CONV3R2D_FIRST 51
#define AVSD_GEMM_TILE_CONV3R2D_LAST 54
typedef struct avsd_gemm_desc {
  const void* A;        /* bf16 */
  const void* A2;       /* bf16 or NULL */
  const void* W;        /* bf16 [N][ldw] */
  void* out;            /* bf16 or f32 [M][ldc] */
  const float* bias;    /* [N] or NULL */
  const float* rowvec;  /* [ceil(M/rows_per_vec)][ldv] or NULL */
  const void* res1;     /* bf16 [M][ldr1] or NULL */
  const void* res2;     /* bf16 [M][ldr2] or NULL */
  int32_t M, N, K;
  int32_t lda, lda2, k_split, ldw, ldc, ldr1, ldr2;
  int32_t rows_per_vec, ldv;
  float alpha;
  int32_t mode, flags;
  int32_t batch;                        /* >= 1 */
  int64_t batch_stride_a, batch_stride_w, batch_stride_out;
  /* TMIX */
  int32_t hw, frames, cseg;
  /* CONV3: source image (hs, ws) with cin channels at row stride lda; output (ho, wo) */
  int32_t hs, ws, ho, wo, cin, stride, ups;
  int32_t pad;                          /* CONV3 top/left zero padding: 1 (symmetric "padding=1") or 0 (the VAE encoder's
                                           F.pad(0,1,0,1) + stride-2 conv); bottom/right reads beyond the image are zero */
  int32_t tile;                         /* 0 = library heuristic; 1..3 register-staged tiles, 4..AVSD_GEMM_MAX_TILE LDS-direct
                                           tiles (see gemm.hip dispatch_tile) */
  /* split-K (LDS-direct tiles only): K is cut into split_k slices computed by separate workgroups that
   * store f32 partial tiles to splitk_ws[split_k][M][N]; a second launch reduces them and applies the epilogue.
   * Deterministic (no atomics).  split_k <= 1 disables it.  Not combinable with GEGLU or batch > 1. */
  int32_t split_k;
  float* splitk_ws;
  /* LayerNorm folded into the GEMMs around it (ff_spatio_audio_temp_transformer_3d.py:300-371: every LayerNorm there
   * feeds linear layers only).  Producer side, AVSD_GEMM_ROWSTATS: besides `out`, the epilogue writes for every row m and
   * every 32-column block j the pair (sum, sum of squares) of the bf16-ROUNDED outputs to rowstats[(m * N/32 + j) * 2]
   * (N % 32 == 0; deterministic, no atomics).  Consumer side, AVSD_GEMM_LNFUSE: A is the un-normalised tensor, W has
   * the LayerNorm gain folded in (W' = W * gamma along K), ln_colsum[n] = sum_k W'[n, k], `bias` carries
   * sum_k beta[k] W[n, k] (+ the layer's own bias), and the epilogue computes
   *     v = rstd[m] * (alpha * acc - mean[m] * ln_colsum[n]) + bias[n] + ...
   * with mean / rstd of row m folded from the ln_nblk (= K / 32, or 1: pre-folded by avsd_ln_fold) pairs of ln_stats — exactly LayerNorm(A) . W^T + b.
   * Batched launches read the statistics of row (batch * batch_stride_a / lda + m). */
  float* rowstats;
  const float* ln_stats;
  const float* ln_colsum;
  int32_t ln_nblk;
  float ln_eps;
  float* out_master;                    /* f32 [M][ldm] or NULL: un-rounded copy of the result (not with GEGLU) */
  int32_t ldm;
  int32_t raster_g;                     /* scheduling knob, no effect on the result: rows of the tile blocks an XCD walks (tile_of_item,
                                           gemm_common.h: 0 = the library default of 8; 1..64 = probe values, tools/asm_bench.py).  Must be
                                           0..64: avsd_gemm_bf16 refuses anything else (a large value would overflow the work-item -> tile
                                           map and drop or duplicate tiles).  Was `reserved0` up to ABI v8. */
  /* AVSD_GEMM_X2 (split precision, see "split-precision storage" below): every 16-bit operand is a pair of planes; these are
   * the ELEMENT offsets from each main plane to its rest plane (same strides).  The product is accumulated as
   * W.A + Wr.A + W.Ar (three MFMA passes into one f32 accumulator); 16-bit residuals are read as main + rest and the output
   * is written as main = round16(v), rest = round16(v - main) — and, with `out_master`, the un-rounded f32 value as well (v10).
   * LDS-direct tiles 7, 11, 13, 24, 25, 34, 35, 36 and the asm tiles.  `out_lo` is also the rest-plane offset of AVSD_GEMM_OUT_REST. */
  int64_t a_lo, a2_lo, w_lo, out_lo, res1_lo, res2_lo;
  /* LayerNorm(x + pos[frame]) folded like the plain LayerNorm above (ff_spatio_audio_temp_transformer_3d.py:346-356: norm_temp of
   * h + the temporal position embedding feeds the q|k|v projection of the temporal attention).  Row m belongs to frame
   * f(m) = (m / pos_hw) % pos_frames.  Producer side, `stats_pos` f32 [pos_frames][N] with AVSD_GEMM_ROWSTATS: the row statistics are
   * those of (rounded output + stats_pos[f(m)]) — `out` itself is unchanged.  Consumer side, `ln_rowvec` f32 [pos_frames][N] =
   * pos . W'^T with AVSD_GEMM_LNFUSE: v = rstd[m] * (alpha * acc + ln_rowvec[f(m)][n] - mean[m] * ln_colsum[n]) + bias[n], i.e.
   * LayerNorm(A + pos) . W^T + b with A un-normalised and pos never added to it.  Not with AVSD_GEMM_X2. */
  const float* stats_pos;
  const float* ln_rowvec;
  int32_t pos_hw, pos_frames;
} avsd_gemm_desc;

int avsd_gemm_bf16(const avsd_gemm_desc* desc_host, void* stream);
/* rows per tile of conv3r tile id `tile` if an (hs x ws)-pixel image with cin channels can use it, else 0 */
int avsd_gemm_conv3r_supported(int tile, int hs, int ws, int cin);
int avsd_gemm_conv3r2d_supported(int tile, int hs, int ws, int cin);
/* sizeof(avsd_gemm_desc) as compiled: lets an FFI binding verify its mirror of the struct. */
int avsd_sizeof_gemm_desc(void);

/* ---- fused cross-attention block --------------------------------------------------------------------------------
 * One launch per residual-stream update of the audio / text cross-attention of BasicTransformerBlock
 * (ff_spatio_audio_temp_transformer_3d.py:315-341; diffusers Attention + AttnProcessor2_0):
 *     out = res + to_out( softmax( (LayerNorm(h) Wq^T) K^T * scale ) V ) + o_bias
 * h [M][C] is the 16-bit residual stream with its LayerNorm statistics `ln_stats` [M][C/32][2] as written by
 * AVSD_GEMM_ROWSTATS; wq / q_colsum / q_bias are the LayerNorm-folded to_q of AVSD_GEMM_LNFUSE; `res` is h again (16-bit)
 * or its f32 master (res_f32).  K / V are the step-invariant projections of the conditioning, cached by the host in the
 * layout this kernel stages: k [nkv][lk_pad][C] (rows >= lk are padding) and vt [nkv][C][lk_pad] (V transposed), where
 * rows [q*L, (q+1)*L) of h use block q / q_per_kv — for the audio branch the host gathers the keys the segment mask
 * leaves visible per frame (segmask_imagebind.py:62-78,104-114), so masked keys are simply absent.
 * Outputs like avsd_gemm_bf16: `out` 16-bit [M][ldo], optional `out_master` f32 and `rowstats` of the rounded output.
 * Built for C = 320 with 8 heads (SD1.5 level 0) and lk_pad in {32, 64, 96}: avsd_cross_attention_block_supported()
 * tells; other shapes run the three separate kernels. */
typedef struct avsd_xattn_desc {
  const void* h;        int32_t ldh;
  int32_t res_f32;
  const void* res;      int32_t ldres;
  int32_t M, C, heads, L;
  const float* ln_stats; float ln_eps;
  float scale;
  const void* wq;       int32_t ldwq;  int32_t lk;
  const float* q_colsum;
  const float* q_bias;
  const void* k;        const void* vt;
  int32_t lk_pad, q_per_kv;
  const void* wo;       int32_t ldwo;  int32_t ldo;
  const float* o_bias;
  void* out;
  float* out_master;    int32_t ldm;   int32_t reserved0;
  float* rowstats;
  const float* stats_pos;   /* as avsd_gemm_desc.stats_pos: [pos_frames][C], the row statistics are those of out + stats_pos[frame of the row] */
  int32_t pos_hw, pos_frames;
} avsd_xattn_desc;
int avsd_cross_attention_block_supported(int C, int heads, int lk_pad);
int avsd_cross_attention_block(const avsd_xattn_desc* desc_host, void* stream);
int avsd_sizeof_xattn_desc(void);

/* out[M, N] (f32) = act_out( act_in(x[M, K] f32) . W[N, K]^T + bias ), M <= 16.
 * act: 0 none, 1 SiLU.  Time-embedding MLP and the per-ResBlock time_emb_proj
 * (audio_cond_unet_3d_condition.py:673-680, ff_spatio_temp_resnet_3d.py:170), and the
 * temporal position MLP (ff_spatio_audio_temp_transformer_3d.py:348-349). */
int avsd_linear_small_m(const float* x, const void* W_bf16, const float* bias, float* out,
                        int M, int N, int K, int ldw, int act_in, int act_out, void* stream);

/* ---- normalisation -------------------------------------------------------------------
 * GroupNorm over channels-last data: stats (partials + finalize) then apply.
 * stats: for each of `nb` normalisation batches (a batch = `rows_per_batch` consecutive
 *   rows: F*H*W for the 5-D GroupNorm of ff_spatio_temp_resnet_3d.py:130,146 /
 *   audio_cond_unet_3d_condition.py:445, H*W for the per-frame GroupNorm of
 *   ff_spatio_audio_temp_transformer_3d.py:62) reduces (sum, sumsq) of each of `groups`
 *   channel groups over `nchunks` row chunks (deterministic, no atomics) into `scratch`
 *   (avsd_groupnorm_scratch_floats floats).  groups: a power of two in 4..64.  The input is the channel concat
 *   [x1 (c1 channels) | x2 (c2 channels)] (c2 may be 0) — the UNet skip concat is never
 *   materialised.
 * apply: every workgroup folds the partials of its batch (double, fixed order) into per-channel
 *   (scale, shift) = (rstd*gamma, beta - mean*rstd*gamma), biased variance, eps inside the sqrt, then streams
 *   y[m, c] = act( x * scale[c] + shift[c] ), act 0 none / 1 SiLU, bf16. */
int avsd_groupnorm_stats(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2,
                         int nb, int rows_per_batch, int groups, float* scratch, int nchunks, void* stream);
int avsd_groupnorm_apply(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2,
                         int nb, int rows_per_batch, int groups, const float* gamma, const float* beta, float eps,
                         const float* scratch, int nchunks, int act, void* y, int ldy, void* stream);
/* One-launch form for SMALL batches: a workgroup keeps whole groups (rows_per_batch rows x the channels of a few groups,
 * <= ~1900 16-byte vectors) in registers between the statistics and the apply: no scratch, no second read of the input.
 * Same arithmetic as the pair above (f32 sums per thread, folded in double in a fixed order; results differ from the pair
 * in the last bits only).  avsd_groupnorm_fused_supported(...) != 0 tells whether a geometry qualifies: the UNet's ResBlock
 * norms at 4 x 4 and its per-frame Transformer3D norms up to 16 x 16 do (5-7 us against 10-12 us for the pair); larger
 * batches are refused — too few workgroups own whole groups for the element-wise work (profiles/r3_gn_probe.txt).
 * Replaces the same reference sites as the pair. */
int avsd_groupnorm_fused_supported(int nb, int rows_per_batch, int groups, int c1, int c2, int split);
int avsd_groupnorm_fused(const void* x1, int ld1, int c1, const void* x2, int ld2, int c2, int nb, int rows_per_batch,
                         int groups, const float* gamma, const float* beta, float eps, int act, void* y, int ldy, void* stream);
/* Suggested nchunks, and the scratch size in floats for it (pure host arithmetic). */
int avsd_groupnorm_nchunks(int nb, int rows_per_batch, int channels);
int avsd_groupnorm_scratch_floats(int nb, int nchunks, int groups, int channels);

/* Pre-folds the row statistics of AVSD_GEMM_ROWSTATS: stats [M][nblk][2] -> out [M][2] (sum, sumsq over the whole row, added in ascending
 * block order).  An AVSD_GEMM_LNFUSE consumer given `out` with ln_nblk = 1 computes the same mean / rstd bit for bit without re-folding
 * nblk pairs per row in every column tile. */
int avsd_ln_fold(const float* stats, int M, int nblk, float* out, void* stream);
/* LayerNorm over the last dim (eps 1e-5 in the reference): y = LN(x + pos[f(m)]) with
 * pos == NULL for plain LN; f(m) = (m / hw) % frames.
 * (ff_spatio_audio_temp_transformer_3d.py:300,317,330,354,361) */
int avsd_layernorm(const void* x, int ldx, void* y, int ldy, int M, int C,
                   const float* gamma, const float* beta, float eps,
                   const float* pos, int hw, int frames, void* stream);

/* Row softmax: P[r, :] = softmax(S[r, :L]) ; S f32, P bf16 (VAE mid-block attention). */
int avsd_softmax_rows(const float* S, int lds, void* P, int ldp, int rows, int L, void* stream);

/* ---- attention -----------------------------------------------------------------------
 * Multi-head attention with queries [Bq][Lq] and keys/values [Bk][Lk]; query batch qb
 * attends to kv batch qb / q_per_kv (first-frame attention utils.py:133-153: q_per_kv =
 * frames, K/V computed for frame 0 only; audio / text cross-attention
 * ff_spatio_audio_temp_transformer_3d.py:319-341: K/V computed once per clip branch).
 * Head h occupies columns [h*d, (h+1)*d) of each row.  key_index (optional, int32
 * [frames][Lk]) gathers key rows for frame qb % frames — the boolean audio segment mask
 * (segmask_imagebind.py:104-114) turned into the list of unmasked keys.
 * kv_rows = rows per kv batch in the K/V buffers (229 audio tokens while Lk = 25 gathered).
 * softmax scale = scale (d^-1/2).  d in {40, 64, 80, 128, 160}. */
int avsd_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                   void* O, int ldo, int Bq, int Lq, int Lk, int kv_rows, int heads, int d,
                   int q_per_kv, const int32_t* key_index, int frames, float scale, void* stream);

/* FP8 (OCP e4m3) variant of avsd_attention — BASELINE.json configuration 5: Q, K, V and the probabilities are rounded to
 * e4m3 and both matrix products run on v_mfma_f32_32x32x16_fp8_fp8; softmax, running statistics and accumulation stay
 * f32.  Same tensors (16-bit in, 16-bit out), shapes and gather list as avsd_attention; q / k / v are multiplied by
 * q_scale / k_scale / v_scale before rounding (1.0 = raw values; |x| <= 448 is representable) and the scales are folded
 * back out.  Replaces the SDPA calls of utils.py:151-153 and ff_spatio_audio_temp_transformer_3d.py:315-341. */
int avsd_attention_fp8(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                       int Bq, int Lq, int Lk, int kv_rows, int heads, int d, int q_per_kv,
                       const int32_t* key_index, int frames, float scale, float q_scale, float k_scale,
                       float v_scale, void* stream);

/* Temporal self-attention across frames for every pixel
 * (ff_spatio_audio_temp_transformer_3d.py:352-358): qkv is [B*F*hw][ldqkv] with q|k|v at
 * column offsets 0, C, 2C; sequence (b, p) = rows {(b*F + f)*hw + p : f}. */
int avsd_temporal_attention(const void* QKV, int ldqkv, void* O, int ldo, int B, int frames,
                            int hw, int heads, int d, float scale, void* stream);

/* ---- elementwise / layout ------------------------------------------------------------ */
/* (B, C, F, H, W) f32  ->  channels-last bf16 [rep*B*F*H*W][cpad] = scale * src, channels >= C
 * zeroed, the batch repeated `rep` times (torch.cat([latents] * k), pipeline...py:331-336). */
int avsd_ncfhw_to_rows(const float* src, void* dst, int B, int C, int F, int HW, int cpad,
                       int rep, float scale, void* stream);
/* channels-last f32 [B*F*HW][ld] (first C columns) -> (B, C, F, H, W) f32. */
int avsd_rows_to_ncfhw(const float* src, int ld, float* dst, int B, int C, int F, int HW,
                       void* stream);
/* Sinusoidal embedding, diffusers Timesteps(dim, flip_sin_to_cos=True, shift=0):
 * out[i, :] = [cos(t_i * w) | sin(t_i * w)], w_j = exp(-ln(1e4) * j / (dim/2)).
 * t is a device f32 array of n values. */
int avsd_timestep_embedding(const float* t, float* out, int n, int dim, void* stream);

/* Guidance + multistep scheduler update on (B, C, F, H, W) f32 latents, frame 0 pinned
 * (pipeline_audio_cond_animation.py:349-364):
 *   n_branch 1: eps = e[0]                                         (no guidance)
 *   n_branch 2: eps = e[0] + g * (e[1] - e[0])                     (:354-361; e = noise_pred of the [null-audio,
 *               audio] batch with g = audio scale, or of [null-text, text] with g = text scale)
 *   n_branch 3: eps = e[0] + g * (e[1] - e[0]) + g2 * (e[2] - e[1]) (:349-353, dual guidance; branches
 *               [uncond, text, text+audio], g = text_guidance_scale, g2 = audio_guidance_scale)
 *   eps_hist[store_slot] = eps                (if store_slot >= 0)
 *   eps'     = w_cur * eps + sum_k w[k] * eps_hist[hist_idx[k]]      (n_hist <= 4 terms)
 *   x_out[:, :, 1:] = ca * x_in[:, :, 1:] + cb * eps'[:, :, 1:] ;  x_out[:, :, 0] = x_in[:, :, 0]
 * PNDM (PLMS, incl. its averaged second step on the saved sample) and DDIM (eta = 0) are both
 * of this form; the scalar schedule lives on the host (asva_amd/schedulers.py). */
int avsd_guided_step(const float* noise_pred, int n_branch, float g, float g2, float* eps_hist,
                     int store_slot, float w_cur, const int32_t* hist_idx_host,
                     const float* w_host, int n_hist, const float* x_in, float* x_out, float ca,
                     float cb, int B, int C, int F, int HW, void* stream);

/* VAE post-processing (pipeline_audio_cond_animation.py:212): channels-last bf16
 * [N*H*W][ld] (3 channels) -> (N, 3, H, W) f32 = clamp(x / 2 + 0.5, 0, 1). */
int avsd_vae_postprocess(const void* src, int ld, float* dst, int N, int HW, void* stream);
/* Same input -> uint8 frames (N, H, W, 3) = trunc(clamp(x / 2 + 0.5, 0, 1) * 255): the post-processing above followed
 * by generate_videos' `(video.permute(0, 2, 3, 1) * 255).byte()` (pipeline_audio_cond_animation.py:448), on device —
 * 4x fewer bytes cross PCIe per clip. */
int avsd_vae_postprocess_u8(const void* src, int ld, void* dst_u8, int N, int HW, void* stream);

/* ---- audio conditioning front-end (once per clip, not on the denoising loop) --------------------------
 * Kaldi-compatible log-mel filterbank + transpose + pad/crop + normalisation: replaces
 * `waveform_to_melspectrogram` (avgen/data/utils.py:26-55) = ImageBind `waveform2melspec` (torchaudio.compliance.
 * kaldi.fbank: htk_compat, hanning window, dither 0, snip_edges) followed by `Normalize(mean, std)`.
 *   wave [batch][wave_stride] f32 (n_samples valid), window [win] f32, mel_fb [n_mel][nfft/2+1] f32 (host-built,
 *   asva_amd/audio_features.py) -> out [batch][n_mel][t_out] f32; frames past 1 + (n_samples - win) / shift are the
 *   normalised zero padding.  nfft must be a power of two; win <= 512. */
int avsd_kaldi_fbank(const float* wave, int batch, int n_samples, int64_t wave_stride, const float* window,
                     const float* mel_fb, int win, int shift, int nfft, int n_mel, float preemph, int remove_dc,
                     float* out, int t_out, float mean, float std, void* stream);
/* im2col of non-padded strided patches: src (B, C, H, W) f32 -> dst bf16 [B*ph*pw][C*kh*kw] (c-major, then kh, kw:
 * the order of a flattened nn.Conv2d weight).  Front of ImageBind's audio stem Conv2d(1, 768, 16, stride 10). */
int avsd_patchify(const float* src, void* dst, int B, int C, int H, int W, int kh, int kw, int stride, void* stream);
/* ViT token matrix: out bf16 [B][1 + n_patches + tail_rows][C]; row 0 = cls + pos[0], row 1+p = patches[b, p] + pos[1+p],
 * tail rows = 0 (the add_bias_kv slot of the ImageBind audio trunk's attention).  cls [C], pos [1+n_patches][C] f32. */
int avsd_vit_tokens(const void* patches, const float* cls, const float* pos, void* out, int B, int n_patches, int C,
                    int tail_rows, void* stream);

/* ---- device copies: with these, everything a denoising step does on the device is an entry point of this library
 *      (and therefore part of a launch plan, below) --------------------------------------------------------------------- */
/* dst[r * bytes + i] = src[i] for r in [0, rep): a device-to-device copy (rep = 1), or the torch.cat([x] * rep) along the
 * batch of guidance branches that are still identical (pipeline_audio_cond_animation.py:331-336).  bytes % 16 == 0. */
int avsd_copy(const void* src, void* dst, int64_t bytes, int rep, void* stream);
/* Cached cross-attention K|V rows  kv [n_kv * rows][2C] 16-bit  ->  the operand layout avsd_cross_attention_block stages:
 * k_out [nb][lk_pad][C] and vt_out [nb][C][lk_pad].  Without a gather list (idx == NULL) nb = n_kv and block n holds keys
 * 0..rows-1 of clip n; with idx [n_frames][nk] int32 (the visible keys of each frame under the audio segment mask,
 * audio_attn_mask in segmask_imagebind.py:104-114) nb = n_kv * n_frames and block n * n_frames + f holds keys idx[f][:] of
 * clip n.  The padding slots lk..lk_pad-1 (lk = nk or rows) are written as zeros by the same launch.  Once per clip. */
int avsd_xattn_pack_kv(const void* kv, int n_kv, int rows, int C, const int32_t* idx, int n_frames, int nk,
                       void* k_out, void* vt_out, int lk_pad, void* stream);

/* ---- split-precision ("x2") storage: the mode that meets north_star's 1e-3 against the reference's fp32 pipeline ----------
 * (scripts/animation_gen.py:43-44 runs fp32).  A 16-bit tensor becomes a PAIR of planes with identical strides:
 *     main = round16(v),  rest = round16(v - main)          (16 significant bits in bf16, no range loss)
 * `*_lo` arguments are the ELEMENT offsets from a main plane to its rest plane.  Matrix products run as three MFMA passes
 * (main.main + rest.main + main.rest; the rest.rest term is 2^-18 of the product) into the same f32 accumulator — the
 * GEMM family through AVSD_GEMM_X2, the attentions below — and every other kernel reconstructs v = main + rest (exact in
 * f32), computes in f32 as before and writes both planes.  Same arithmetic otherwise as the entry points they mirror. */
int avsd_linear_small_m_x2(const float* x, const void* W, int64_t w_lo, const float* bias, float* out,
                           int M, int N, int K, int ldw, int act_in, int act_out, void* stream);
int avsd_groupnorm_stats_x2(const void* x1, int ld1, int c1, int64_t x1_lo, const void* x2, int ld2, int c2, int64_t x2_lo,
                            int nb, int rows_per_batch, int groups, float* scratch, int nchunks, void* stream);
int avsd_groupnorm_fused_x2(const void* x1, int ld1, int c1, int64_t x1_lo, const void* x2, int ld2, int c2, int64_t x2_lo,
                            int nb, int rows_per_batch, int groups, const float* gamma, const float* beta, float eps, int act,
                            void* y, int ldy, int64_t y_lo, void* stream);
int avsd_groupnorm_apply_x2(const void* x1, int ld1, int c1, int64_t x1_lo, const void* x2, int ld2, int c2, int64_t x2_lo,
                            int nb, int rows_per_batch, int groups, const float* gamma, const float* beta, float eps,
                            const float* scratch, int nchunks, int act, void* y, int ldy, int64_t y_lo, void* stream);
int avsd_layernorm_x2(const void* x, int ldx, int64_t x_lo, void* y, int ldy, int64_t y_lo, int M, int C,
                      const float* gamma, const float* beta, float eps, const float* pos, int hw, int frames, void* stream);
int avsd_attention_x2(const void* Q, int ldq, int64_t q_lo, const void* K, int ldk, int64_t k_lo, const void* V, int ldv,
                      int64_t v_lo, void* O, int ldo, int64_t o_lo, int Bq, int Lq, int Lk, int kv_rows, int heads, int d,
                      int q_per_kv, const int32_t* key_index, int frames, float scale, void* stream);
int avsd_temporal_attention_x2(const void* QKV, int ldqkv, int64_t qkv_lo, void* O, int ldo, int64_t o_lo, int B, int frames,
                               int hw, int heads, int d, float scale, void* stream);
int avsd_softmax_rows_x2(const float* S, int lds, void* P, int ldp, int64_t p_lo, int rows, int L, void* stream);
int avsd_ncfhw_to_rows_x2(const float* src, void* dst, int64_t dst_lo, int B, int C, int F, int HW, int cpad, int rep,
                          float scale, void* stream);
/* f32 [n] -> planes (conditioning inputs: text / audio encodings handed over in f32). */
int avsd_split_f32(const float* src, void* dst, int64_t dst_lo, int64_t n, void* stream);
int avsd_vae_postprocess_x2(const void* src, int ld, int64_t src_lo, float* dst, int N, int HW, void* stream);
int avsd_vae_postprocess_u8_x2(const void* src, int ld, int64_t src_lo, void* dst_u8, int N, int HW, void* stream);

/* ---- exact-f32 yardstick ---------------------------------------------------------------------------------------------------
 * out[M, N] = A[M, K] . W[N, K]^T + bias[n], all f32, on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: bitwise a
 * k-ordered fmaf chain, 157 TFLOP/s peak = 1/16 of the bf16 rate).  Not on the product path: it is the on-box reference that
 * separates kernel arithmetic from storage rounding when the 16-bit and split-precision GEMMs are validated, and the rate
 * bench.py quotes next to theirs.  The reference computes the same products with fp32 torch.nn.Linear / nn.Conv2d
 * (scripts/animation_gen.py:43-44).  K, lda, ldw multiples of 4; A, W 16-byte aligned. */
int avsd_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N, int K,
                  void* stream);

/* ---- launch plans (a host without Python runs the path) ----------------------------------------------------
 * A plan is the sequence of calls to the entry points above that one operation of the reference issues — the UNet forward
 * of a denoising step (audio_cond_unet_3d_condition.py:598-798), the per-clip conditioning projections, the VAE decode
 * (pipeline_audio_cond_animation.py:206-213) — for ONE geometry (network config, latent shape, guidance branches), with
 * every device pointer expressed as (buffer, byte offset).  asva_amd/plan.py records plans from the Python host (which owns
 * the network description) and writes a bundle: a buffer table shared by all its plans + the call lists.  Any host then
 *   loads the bundle, allocates (or maps) the buffers, binds them, uploads weights and inputs, and calls avsd_plan_run:
 * the same kernels with the same arguments as the recording run, so the results are bit-identical to the Python host's.
 * tools/plan_host.cpp is such a host (C++, no Python, no torch).
 * Memory model: the bundle's BUFFERS are the device allocations of the recording run (the caching allocator's segments, with
 * their sizes); a replaying host allocates each one zero-filled and binds it.  Named REGIONS (buffer, offset, bytes, kind) say
 * where in those buffers the things a host touches live: */
enum { AVSD_REGION_CONST = 1,    /* weights and tables: contents ship with the bundle (<bundle>.d/<name>.bin); upload once */
       AVSD_REGION_INPUT = 2,    /* written by the host before a run (latents, timestep, conditioning embeddings)         */
       AVSD_REGION_OUTPUT = 3 }; /* read by the host after a run (noise prediction, decoded frames)                       */
typedef struct avsd_plan_bundle avsd_plan_bundle;
int avsd_plan_bundle_load(const char* path_host, avsd_plan_bundle** out_host);
void avsd_plan_bundle_free(avsd_plan_bundle* b);
int avsd_plan_bundle_num_buffers(const avsd_plan_bundle* b);
int64_t avsd_plan_bundle_buffer_bytes(const avsd_plan_bundle* b, int i);                 /* -1 for a bad index */
int avsd_plan_bundle_bind(avsd_plan_bundle* b, int i, void* device_ptr);               /* 256-byte aligned */
int avsd_plan_bundle_num_regions(const avsd_plan_bundle* b);
/* region j: its name (lives as long as the bundle), the buffer it is in, its byte offset and size, its kind */
int avsd_plan_bundle_region(const avsd_plan_bundle* b, int j, const char** name_host, int* buffer_host, int64_t* offset_host,
                            int64_t* bytes_host, int* kind_host);
int avsd_plan_bundle_find_region(const avsd_plan_bundle* b, const char* name_host);    /* index, or -1 */
int avsd_plan_bundle_num_plans(const avsd_plan_bundle* b);
const char* avsd_plan_bundle_plan_name(const avsd_plan_bundle* b, int k);
int avsd_plan_bundle_find_plan(const avsd_plan_bundle* b, const char* name_host);       /* index, or -1 */
int avsd_plan_num_calls(const avsd_plan_bundle* b, int plan);
/* Issues every call of plan `plan` on `stream`, in recording order.  All buffers the plan touches must be bound.
 * Capturable in a hipGraph like the calls themselves.  A bundle is not thread-safe (run patches the stream into its
 * argument lists): one bundle per host thread; bundles on distinct streams are independent. */
int avsd_plan_run(avsd_plan_bundle* b, int plan, void* stream);

/* Operation-level calls on a bundle recorded with the conventional region names (tools/export_plan.py) — the surface
 * SURVEY 8(b)-3 sketched, for hosts that do not want to handle regions themselves.  All pointers are DEVICE pointers of the
 * regions' sizes (the 16-bit CFG-batched text / audio encodings, f32 latents, one f32 timestep); every call only enqueues
 * work on `stream`.  avsd_unet_set_conditioning: audio_cond_unet_3d_condition.py:598-798's step-invariant part, once per clip
 * ("text", "audio" -> plan "set_conditioning").  avsd_unet_forward: one UNet evaluation of the CFG batch ("x", "t" -> plan
 * "forward" -> "noise_pred"; noise_pred_out may be NULL: read the region instead).  avsd_vae_decode:
 * pipeline_audio_cond_animation.py:206-213 + :448 ("latents" -> plan "decode" -> uint8 "frames").  The guidance + scheduler
 * update between two forwards is avsd_guided_step.  avsd_plan_region_ptr: device address (and size) of a region in the
 * bound buffers, NULL if absent or unbound.
 * A "forward" plan bakes in the launch sequence of its recording run, including the shared guidance prefix: when the recording
 * host saw the SAME text rows in every guidance branch (audio-only guidance, text [t, t], pipeline_audio_cond_animation.py:155)
 * the layers in front of the first audio cross-attention were recorded once for all branches.  Such a bundle is valid only for
 * text inputs with that property; record with AVSD_SHARE_PREFIX=0 for per-branch text (text or dual guidance).  The region
 * "share_prefix" (4 bytes, CONST: 1 = the prefix is shared) states which kind a bundle is. */
int avsd_unet_set_conditioning(avsd_plan_bundle* b, const void* text, const void* audio, void* stream);
int avsd_unet_forward(avsd_plan_bundle* b, const float* sample, const float* timestep, float* noise_pred_out, void* stream);
int avsd_vae_decode(avsd_plan_bundle* b, const float* latents, void* frames_u8_out, void* stream);
const void* avsd_plan_region_ptr(avsd_plan_bundle* b, const char* name_host, int64_t* bytes_host);

#ifdef __cplusplus
}
#endif
#endif /* AVSD_H */
