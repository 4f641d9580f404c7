/* vdn.h — C-ABI of libvdn_hip.so: the MI355X (gfx950) kernels behind
 * DepthAnythingV2.forward / VideoDepthAnything.forward.
 *
 * The reference has no FFI for this path: callers hold a torch.nn.Module and every op below is a
 * torch.nn.functional call inside it. Each entry point therefore names the
 * reference call site(s) it replaces (paths relative to the reference root). The Python host
 * mirror (video-depth-normal-v2_amd/vdn) binds these with ctypes; INTEGRATION.md shows the stub
 * a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers + sizes only; every buffer is caller-owned device memory (hipMalloc'd by
 *    PyTorch-ROCm), the library allocates nothing and keeps no state;
 *  - every launch is asynchronous on `stream` (torch.cuda.current_stream().cuda_stream);
 *  - return 0 on success, a negative vdn_status on a rejected argument, or -(1000+hipError_t);
 *  - "half" = the 16-bit MFMA operand type chosen per call by `dt`: VDN_F16 (IEEE fp16, default:
 *    same MFMA rate as bf16 with 8x smaller rounding, and what the reference's own video driver
 *    autocasts to, video_depth_anything/video_depth.py:106) or VDN_BF16;
 *  - activations are channels-last: tokens [B, N, C] == NHWC feature maps [B, H, W, C].
 */
#ifndef VDN_H
#define VDN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vdn_stream; /* hipStream_t */

enum vdn_status { VDN_OK = 0, VDN_EINVAL = -1, VDN_EUNSUPPORTED = -2, VDN_EALIGN = -3 };
enum vdn_dtype { VDN_F16 = 0, VDN_BF16 = 1, VDN_F32 = 2, VDN_NONE = 3 };
enum vdn_act { VDN_ACT_NONE = 0, VDN_ACT_GELU = 1, VDN_ACT_RELU = 2,
               VDN_ACT_SILU = 3 /* only as the gate of VDN_ST_GEGLU: out = h * silu(gate) (SwiGLU, dinov2_layers/swiglu_ffn.py:29-33) */ };
enum vdn_amode { VDN_A_PLAIN = 0, VDN_A_CONV3X3 = 1 };
enum vdn_store {
  VDN_ST_PLAIN = 0,   /* out[row(m) * ldc + n]                                                   */
  VDN_ST_HEADS = 1,   /* n -> (split, head, e<64); per-split buffer, token- or dim-major          */
  VDN_ST_CONVT = 2,   /* ConvTranspose2d with kernel == stride: pixel-shuffle scatter to NHWC      */
  VDN_ST_GEGLU = 3    /* weight rows packed as 16-row blocks [h | gate]; out = h * gelu(gate), or h * silu(gate) with act = VDN_ACT_SILU */
};

/* Kernel-selection knobs of vdn_gemm (which tile / pipeline / split-K variant a shape gets; never what is computed beyond
 * fp32 summation order). The library holds NO mutable selection state: the defaults are read from the environment variables
 * named below once, at the first use (vdn_gemm_get_tuning returns them), and a launch that wants other values points its
 * descriptor's `tuning` at a struct of its own (tests pin a tile with force_bm; tools run A/B experiments). Thread-safe.   */
typedef struct vdn_gemm_tuning {
  int force_bm;     /* VDN_GEMM_BM       0 = cost model; 128 | 192 | 256 force the 8-wave kernels' M tile                  */
  int p8;           /* VDN_GEMM_P8       1 = ping-pong 3-product kernel at BM 256 (default), 2 = also BM 192, 0 = never   */
  int no_splitk;    /* VDN_GEMM_NOSPLITK 1 = never split K                                                                */
  int no_pipe;      /* VDN_GEMM_NOPIPE   1 = lock-step loop without the phase shift                                       */
  int splitk_p8;    /* VDN_SPLITK_P8     >= 2: K slices for deep residual linears on the ping-pong kernel                 */
  int cus;          /* VDN_GEMM_CUS      > 0 overrides the CU count tiles are sized for (else desc.cu_hint / 256)          */
  int splitk_occ;   /* VDN_SPLITK_OCC    split K when the 128-row tile grid covers <= this percent of the CUs (50)         */
  int splitk_max;   /* VDN_SPLITK_MAX    most K slices (8)                                                                */
  int min_tiles;    /* VDN_GEMM_MIN_TILES plain-A problems with fewer 128x256 tiles use the 4-wave 128x128 kernel (96)     */
  float f128, f192; /* VDN_GEMM_F128/F192 cost factors of the smaller M tiles of the 3-product kernels (1.3, 1.2)          */
  int x8;           /* VDN_GEMM_X8       1 (default): launches that carry A8 / W8 planes run on the 8-bit cross-term kernel; 0: rejected */
} vdn_gemm_tuning;
int vdn_gemm_get_tuning(vdn_gemm_tuning* out);   /* the environment-derived defaults */

/* One descriptor drives every GEMM-shaped op on the path:
 *   out = epilogue( A[M,K] x W[N,K]^T ),  half operands, fp32 accumulate on MFMA.
 * Replaces F.linear / nn.Conv2d(1x1, 3x3 s1|s2 p1) / nn.ConvTranspose2d(k==s) at:
 *   depth_anything_v2/dinov2_layers/attention.py:51,60 (qkv, proj), mlp.py:36,39 (fc1, fc2),
 *   patch_embed.py:76 (14x14 s14 conv == GEMM on patchified rows),
 *   depth_anything_v2/dpt.py:129-130 (projects, resize_layers), util/blocks.py:68-74 (RCU convs),
 *   :146 (out_conv), dpt.py:135-138 (layerN_rn), :145,148 (output_conv1/2),
 *   video_depth_anything/motion_module/motion_module.py:116,131 (proj_in/out), :273,280-281
 *   (to_q/k/v), :315 (to_out), attention.py:383-384 (GEGLU), :329 (ff out),
 *   sam2/modeling/sam/transformer.py:279-281,309 (q/k/v/out_proj), memory_attention.py:96
 *   (linear1/2), memory_encoder.py:108-110 (pwconv1/2), :172 (pix_feat_proj).                   */
typedef struct vdn_gemm_desc {
  int32_t dt;            /* vdn_dtype of A and W (VDN_F16 | VDN_BF16)                             */
  int32_t M, N, K;       /* K = logical reduction length (conv: 9*Cin), multiple of 8             */
  /* A operand */
  const void* A;         /* plain: half [M, lda]; conv: half NHWC [cB, cH, cW, cC]                */
  int32_t a_mode;        /* vdn_amode                                                             */
  int32_t lda;           /* elements                                                              */
  int32_t relu_a;        /* apply ReLU to A on load (ResidualConvUnit's activation on its input)  */
  int32_t cB, cH, cW, cC, cOH, cOW, cstride; /* conv geometry (pad 1), M == cB*cOH*cOW            */
  /* W operand: half [N, ldb], K-contiguous, zero-padded to ldb (multiple of 64, >= K)            */
  const void* W;
  int32_t ldb;
  /* epilogue, applied in fp32 in this order:
   *   v = acc + bias[n] + rowadd[m];  v = act(v);  v *= gamma[n];
   *   v += tab[(m % tab_mod + tab_off), n];  v += res1[m,n];  v += res2[m,n]                     */
  const float* bias;     /* [N] or NULL                                                           */
  const float* rowadd;   /* [M] or NULL                                                           */
  int32_t act;           /* vdn_act                                                               */
  const float* gamma;    /* [N] or NULL (LayerScale / CXBlock gamma)                              */
  const float* tab;      /* f32 [*, N] or NULL (pos_embed)                                        */
  int32_t tab_mod, tab_off;
  const void* res1;      /* [M, ldr1] or NULL                                                     */
  int32_t res1_dt, ldr1;
  const void* res2;
  int32_t res2_dt, ldr2;
  /* store */
  int32_t store;         /* vdn_store                                                             */
  void* out;             /* PLAIN/CONVT/GEGLU destination                                         */
  int32_t out_dt;        /* VDN_F16|VDN_BF16|VDN_F32                                              */
  int32_t ldc;
  int32_t row_group, row_skip; /* PLAIN: out row = m + (m / row_group + 1) * row_skip if row_group>0
                                  (patch tokens land after each image's cls row)                 */
  /* HEADS: N == nsplit * heads * 64. split s goes to dst[s]:
   *   token-major  [Bt, heads, tpad, 64]  (transposed[s] == 0)
   *   dim-major    [Bt, heads, 64, tpad]  (transposed[s] == 1, the V^T image the attention
   *                                        kernel reads with contiguous keys)
   * with m -> (bt = m / tokens, t = m % tokens + tok_off).
   * rope[s] != 0 rotates adjacent pairs (2i,2i+1) of each head by rope_cs[(t % rope_mod), i]
   * = (cos, sin) (sam2/modeling/position_encoding.py:212-239); requires the weight rows of that
   * split to be packed pair-split (VDN_PACK_ROPE of vdn_pack_weight below).                      */
  void* dst[3];
  int32_t nsplit, heads, tokens, tok_off, tpad;
  int32_t transposed[3];
  int32_t rope[3];
  const float* rope_cs;  /* f32 [rope_mod, 32, 2]                                                 */
  int32_t rope_mod;
  /* CONVT: N == ck*ck*cout, n = (ky*ck + kx)*cout + co; m = (b, y, x) on a cB x cH x cW grid;
   * out NHWC [cB, cH*ck, cW*ck, cout]                                                            */
  int32_t ck, cout;
  const void* zeros;     /* >= 16 bytes of zeros (conv padding taps / K tail read from here)      */
  /* Split-precision ("x3") planes, all optional. A 16-bit tensor t may come with a second plane
   * t_lo = half(t_exact - float(t_hi)); the kernel then accumulates
   *     A_hi*W_hi  +  A_hi*W_lo (if W_lo)  +  A_lo*W_hi (if A_lo)
   * as extra K segments of the same MFMA loop (fp32-faithful to ~2^-21 instead of 2^-11), and
   * half outputs are written as (hi = round-toward-zero, lo = remainder) when out_lo / dst_lo is
   * given. hi and lo share their sign, so relu_a acts on each plane independently. TRANSPOSED head splits (V^T) are the
   * exception: hi is rounded to nearest, and their dst_lo may be NULL on its own (the one-product P V attention reads hi only). */
  const void* A_lo;
  const void* W_lo;
  void* out_lo;
  void* dst_lo[3];
  const void* res1_lo;
  const void* res2_lo;
  /* conv K order: 0 = (tap, ci) ; 1 = (ci/64, tap, ci%64) — requires cC % 64 == 0. With order 1
   * consecutive K steps read the same pixels' neighbouring taps / the two halves of one 128-byte
   * line, so the 9x re-read of the input map is served from L2 instead of HBM.                   */
  int32_t conv_korder;
  /* Tile-shape hint: how many CUs this launch can count on (0 = the whole chip, 256). A caller that keeps two
   * independent launch streams busy passes 128: the M tile is then chosen for the share of the machine one
   * kernel really gets while the other stream's kernels co-run (measured +3.6 % end to end). Never changes results. */
  int32_t cu_hint;
  /* Optional split-K workspace (f32, caller-owned, >= 16-byte aligned): a 3x3 convolution whose tile grid covers
   * only a fraction of the chip (low-resolution maps with 9*Cin = 9216-deep reductions) is cut into up to 8 K
   * slices that write partial sums here and a second kernel adds them in a fixed order and applies the epilogue
   * (deterministic). NULL / 0 disables. `ksplit` is set by the library and must be 0 on entry.                */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
  int32_t ksplit;
  /* HEADS, split-plane mode, VDN_F16 only: optional 8-bit planes of a token-major split s (Q / K) for the attention
   * kernel's cross terms (vdn_flash_attn Q8 / K8): u8 [Bt, heads, tpad, 128], per token 64 bytes of e5m2(v) followed by
   * 64 bytes of e5m2((v - hi(v)) * 2^10), same token mapping as dst[s]. NULL = not written. Must be NULL for
   * transposed splits. A split with 8-bit planes may leave its dst_lo NULL (the attention then never reads the fp16 lo plane).                                                                                            */
  void* dst8[3];
  /* Cross-term planes of the GEMM operands themselves (plain A, VDN_F16, K % 64 == 0; all three optional). When A8 and W8 are
   * given, the kernel accumulates A_hi W_hi^T on fp16 MFMAs and the two cross terms A_hi W_lo^T + A_lo W_hi^T on the
   * block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e3m2 operands: 4.2x the fp16 rate) instead of two more fp16
   * products; A_lo / W_lo are then not read.
   *   A8 u8 [2, M, K], W8 u8 [2, N, ldb]: plane 0 describes hi(v), plane 1 the remainder v - hi(v); same row-major shape
   *   as the fp16 operand (lda == K), in "x6 rows": per row and 64 of K a 64-byte row-slab of two 32-byte HALVES (one per
   *   K half of the MFMA), each [24 B: 32 e3m2 codes, 6-bit fields, little-endian][1 B: E8M0 scale byte s, value = code
   *   2^(s - 127)][7 B unused]; s = floor(log2 max|hi| of the half) - 4 + 127 for plane 0 and 10 less for plane 1.
   *   WHICH 32 columns of the slab a half holds, and in what order, is the producer's choice (vdn_pack_x8 `order`): the
   *   hardware only needs A and W to agree, so the weight planes are packed in the order of the kernel that writes A.
   *   out8: the same planes of a half-precision output of the bias + GELU flavour (u8 [2, M, ldc], N == ldc, N % 64 == 0),
   *   written next to out / out_lo for the GEMM that consumes it (order 1). vdn_pack_x8 / vdn_layernorm / vdn_flash_attn
   *   (their out8 arguments) produce the planes of weights and of the other activations.                             */
  const void* A8;
  const void* W8;
  void* out8;
  /* K-tile-major operand planes of the 8-bit cross-term kernel (all 0 = row-major). A 16-bit plane [rows, K] is stored as
   * [K / 32][rows][32] and a byte plane as [K / 64][rows][64]: the 16 rows x 64 bytes one LDS-DMA instruction moves are then
   * one contiguous KiB (8 full cache lines) instead of 16 half-used lines — 67 against 44 GB/s per CU of L2 -> LDS feed.
   *   a_kt:   A and A8 are K-tile-major (lda is ignored; rows = M);   w_kt: W and W8 (rows = N, K extent = ldb; vdn_pack_x8);
   *   out_kt: a 16-bit PLAIN output (out, out_lo if given, out8) is WRITTEN K-tile-major with rows = M, columns = N == ldc,
   *           i.e. as the a_kt operand of the next GEMM (bias + GELU / plain half-plane flavours of the 8-bit kernel only).
   * vdn_layernorm and vdn_flash_attn produce the same layout (their `kt` arguments).                                      */
  int32_t a_kt, w_kt, out_kt;
  /* 8-bit cross-term kernel: which cross terms this launch accumulates. 0 = both (fp32-faithful, the default);
   * 1 = without A_lo W_hi^T (A enters as its 16-bit hi plane alone: 2^-12 per activation element);
   * 2 = without A_hi W_lo^T (W as its hi plane alone). A per-launch precision knob for the per-layer budget of
   * DESIGN.md §3 / profiles/r03_precision_budget.md; the engines pass 0 unless VDN_X8_TERMS says otherwise.           */
  int32_t x8_terms;
  const vdn_gemm_tuning* tuning;   /* NULL = the library defaults (vdn_gemm_get_tuning); else this launch's own knobs */
} vdn_gemm_desc;

int vdn_gemm(const vdn_gemm_desc* d, vdn_stream stream);

/* LayerNorm over the last dim of [rows, C] (fp32 statistics, two-pass in registers).
 *   y = LN(x) * w + b;  y += alpha * addvec[c];  y += addtab[(row / tab_div) % tab_mod, c]
 * writes out_h (half, optional) and out_f (f32, optional).
 * Replaces nn.LayerNorm at dinov2_layers/block.py:84,87 + dinov2.py:310 (eps 1e-6),
 * memory_attention.py:60,74,93,162 (eps 1e-5), motion_module.py:179,189 (eps 1e-5, with the
 * sinusoidal PE add of :211 fused as addtab), LayerNorm2d sam2_utils.py:148-153 on NHWC rows.
 *   out_group > 0 drops the first row of every `out_group` rows (the cls token, dinov2.py:312)
 *   and writes the remaining rows compacted.
 *   out8 (VDN_F16, C % 64 == 0, optional): u8 [2, rows, C], the x6 rows of y (vdn_gemm_desc.A8; order 0): the A8 planes of the
 *   GEMM that consumes y (out_h_lo may then be NULL). kt != 0: out_h / out_h_lo / out8 are written K-tile-major (vdn_gemm_desc.a_kt). */
int vdn_layernorm(const void* x, int x_dt, int rows, int C, const float* w, const float* b, float eps,
                  const float* addvec, float alpha, const float* addtab, int tab_div, int tab_mod,
                  int out_group, void* out_h, void* out_h_lo, int h_dt, float* out_f, void* out8, int kt,
                  vdn_stream stream);

/* Fused attention forward, head_dim 64: out[b, q, h*64+e] = softmax(scale * Q K^T) V.
 *   Q  half [BH, nq_pad, 64] (rows >= nq never read), K half [BH, nk_pad, 64],
 *   Vt half [BH, 64, nk_pad] (dim-major; columns >= nk must be finite), nk_pad % 64 == 0.
 * Scores never touch HBM. Replaces dinov2_layers/attention.py:53-59 and
 * F.scaled_dot_product_attention at sam2/modeling/sam/transformer.py:306.
 *   Q8 / K8 (both or neither; split fp16 planes only): the 8-bit planes vdn_gemm wrote through dst8
 *   (u8 [BH, n_pad, 128]); the two cross terms K_hi Q_lo^T + K_lo Q_hi^T of the scores then run on the block-scaled
 *   8-bit MFMA (e5m2, 2.3x the fp16 rate) instead of two fp16 products; logits stay within ~1e-5. NULL = 3 fp16 products.
 *   out8 (with Q8 / K8 only, optional): u8 [2, B nq, H 64], the x6 rows of the output (vdn_gemm_desc.A8; order 2): the A8 planes of
 *   the projection that consumes the output (out_lo may then be NULL); out_kt != 0: out / out_lo / out8 are written K-tile-major
 *   with rows = B nq (vdn_gemm_desc.a_kt).                                                                                 */
int vdn_flash_attn(int dt, const void* Q, const void* K, const void* Vt, void* out, const void* Q_lo,
                   const void* K_lo, const void* Vt_lo, void* out_lo, const void* Q8, const void* K8, void* out8, int out_kt,
                   int B, int H, int nq, int nq_pad, int nk, int nk_pad, float scale, int pv_products, vdn_stream stream);
/* pv_products (split-plane mode only; 0 = the default, 1): MFMA products per P V term — a PER-CALL argument, the library keeps
 * no selection state. 2: the softmax weights, born in registers, are rounded once to 16 bits and the row sum uses the same
 * rounded weights (O = sum p~ V / sum p~, V at 21 bits): each weight is off by <= 2^-11 relative, common factors cancel.
 * 3: P is split into hi / lo planes too (every output within ~1e-6 of fp64 at ~13 % more kernel time; reads the fp16 lo
 * planes of Q and K, so their producer must write them). 1 (fp16 planes with Q8 / K8 only, else treated as 2): V enters as
 * its hi plane alone, which the producer rounds to nearest (vdn_gemm writes the hi plane of TRANSPOSED head splits that
 * way; lo = remainder): every output element is a convex combination of fp16-rounded V values, i.e. within 2^-12 relative of
 * the 2-product result at worst; 2e-5..9e-5 end to end on the fixtures against the 1e-3 tolerance (2: 7e-6..2e-5), 14-17 %
 * less kernel time. A memory bank keeps the V planes its producer wrote: use one value per model. */

/* Temporal attention over <=64 frames per (pixel, head) (32 in the 32-frame windows, 64 in the v5 refiner): qkv half [(b f), D, 3c] packed
 * [q | k | v], out half [(b f), D, c]. Replaces motion_module/attention.py:182-211 (_attention)
 * with the rearranges of motion_module.py:255,320. rope_cs (pe = 'rope', motion_module.py:236-240,279-282; NULL for 'ape',
 * whose position term enters before the projections): f32 [T, c/2, 2] = (cos, sin)(t * 10000^(-2i/c)); adjacent channel
 * pairs (2i, 2i+1) of q and k are rotated by their frame's angles on load (attention.py:403-429).                       */
int vdn_temporal_attn(int dt, const void* qkv, void* out, const void* qkv_lo, void* out_lo, int Bv, int T, int D,
                      int c, int heads, float scale, const float* rope_cs, vdn_stream stream);

/* GroupNorm over NHWC half [F, HW, C] (fp32 stats per (frame, group)); `partial` is
 * f32 [F, nsplit, groups, 2] scratch. Replaces motion_module.py:112 (32 groups, eps 1e-6).       */
int vdn_groupnorm(int dt, const void* x, const void* x_lo, void* y, void* y_lo, int F, int HW, int C, int groups,
                  const float* w, const float* b, float eps, float* partial, int nsplit, vdn_stream stream);

/* Bilinear resize, align_corners=True, NHWC half (C % 8 == 0) or single-channel f32.
 * Replaces F.interpolate at util/blocks.py:144, dpt.py:147, depth_anything_v2.py:63,
 * video_depth.py:63.                                                                             */
int vdn_upsample_bilinear(int dt, const void* x, const void* x_lo, void* y, void* y_lo, int B, int IH, int IW, int OH,
                          int OW, int C, vdn_stream stream);
int vdn_upsample_bilinear_f32(const float* x, float* y, int B, int IH, int IW, int OH, int OW, int relu,
                              vdn_stream stream);

/* f32 NCHW image [B,3,H,W] -> half rows [B*ph*pw, ldk] with k = (c*14+ky)*14+kx, zero tail.
 * (the im2col-free view of PatchEmbed's 14x14 stride-14 conv, patch_embed.py:76)                 */
int vdn_patchify(int dt, const float* img, void* rows, void* rows_lo, int B, int H, int W, int ldk,
                 vdn_stream stream);

/* x[b*rows_per_b + row, :] = vec[:] (cls_token + pos_embed[0], dinov2.py:219-220)                */
int vdn_fill_row(float* x, const float* vec, int B, int rows_per_b, int row, int C, vdn_stream stream);

/* Bicubic resample (A=-0.75, align_corners=False, src = (dst+0.5)/scale - 0.5, clamped taps) of a
 * channels-last f32 [ih, iw, C] grid to [oh, ow, C]: torch's scale-factor bicubic used on the
 * pos_embed grid (dinov2.py:193-203, scale_rows = sx, scale_cols = sy there) and the same
 * kernel cv2.INTER_CUBIC applies to frames (util/transform.py:113).                               */
int vdn_bicubic(const float* src, float* dst, int ih, int iw, int oh, int ow, int C, float scale_rows,
                float scale_cols, vdn_stream stream);

/* The whole pre-processing of a batch of frames in ONE launch (depth_anything_v2.py:67-92, util/transform.py:109-157,
 * video_depth.py:73-99): u8 [n, h, w, 3] HOST-ORDER RGB (swap_rb = 0) or BGR (swap_rb = 1: cv2.cvtColor(BGR2RGB)) -> /255 ->
 * cubic resize to (H, W) with the taps of vdn_bicubic (identity when (H, W) == (h, w)) -> (v - mean[c]) / std[c] ->
 * f32 [n, 3, H, W]. mean3 / std3: 3 HOST floats each (ImageNet: .485 .456 .406 / .229 .224 .225).                    */
int vdn_preprocess(const uint8_t* frames, int n, int h, int w, int swap_rb, float* out, int H, int W, const float* mean3,
                   const float* std3, vdn_stream stream);

/* y = x + alpha * vec[c]  (memory_attention.py:141)                                               */
int vdn_add_vec(const float* x, const float* vec, float alpha, float* y, int rows, int C, vdn_stream stream);

/* depth[m] = (relu?) (bias + sum_c w[c] * feat[m, c]), feat half [M, C<=64] already ReLU'd
 * (the 1x1 conv + ReLU closing output_conv2, dpt.py:111-112)                                      */
int vdn_head_out(int dt, const void* feat, const void* feat_lo, const float* w, float bias, float* depth, int M,
                 int C, int relu, vdn_stream stream);

/* MaskDownSampler stages of memory_block.py:72-75 (sam2/modeling/memory_encoder.py:36-58):
 * stage 1: sigmoid -> conv3x3 s2 p1 (1->4) -> LayerNorm2d -> GELU -> conv1x1 (4->1)
 * stage 2:            conv7x7 s7    (1->49) -> LayerNorm2d -> GELU -> conv1x1 (49->1)
 * w: packed f32 parameter block [conv w | conv b | ln w | ln b | proj w | proj b].                */
int vdn_mask_down1(const float* depth, float* out, int B, int H, int W, int OH, int OW, const float* w,
                   vdn_stream stream);
int vdn_mask_down2(const float* in, float* out, int B, int H, int W, int OH, int OW, const float* w,
                   vdn_stream stream);

/* Depthwise 7x7 pad 3 conv on f32 NHWC [B,H,W,C], w f32 [49, C], bias [C] (CXBlock.dwconv,
 * memory_encoder.py:101).                                                                         */
int vdn_dwconv7(const float* x, float* y, int B, int H, int W, int C, const float* w, const float* bias,
                vdn_stream stream);

/* y(half planes) = x(f32) + tab[(row / tab_div) % tab_mod]: the sinusoidal frame-position add of
 * motion_module.py:211 applied to a window assembled from cached (already LayerNorm-ed) states in the
 * streaming mode (video_depth_stream.py:133-144, motion_module.py:255-266).                        */
int vdn_addtab_cast(int dt, const float* x, const float* tab, int tab_div, int tab_mod, void* y, void* y_lo,
                    size_t rows, int C, vdn_stream stream);

/* Streaming temporal attention over a PROJECTED key/value cache (video_depth_stream.py:133-158,
 * motion_module.py:255-277). pool: a caller-owned ring of frame slots, f32 [slots, slot_stride >= HW*3c];
 * a slot holds one cached frame's [HW, 3c] = (q | k | v) projections of its LayerNorm-ed hidden state WITHOUT the
 * frame-position term. slots: HOST array of the T ring-slot indices of the window, oldest first (copied into the launch
 * arguments: no pointer table in HBM, nothing allocated per step). pe_q / pe_k / pe_v: f32 [>= T, c] =
 * PositionalEncoding.pe @ W_{q,k,v}^T (added on load: W(x + pe) = Wx + W pe). Only the newest frame (slots[T-1],
 * position T-1) queries; 8 heads of c/8; out: half planes [HW, c] (out_lo may be NULL).
 * c in {64, 128, 192, 256, 384, 512, 768, 1024}, T <= 32.                                                            */
int vdn_temporal_attn_last(int dt, const float* pool, size_t slot_stride, const int32_t* slots, int T, int HW, int c,
                           const float* pe_q, const float* pe_k, const float* pe_v, float scale, void* out, void* out_lo,
                           vdn_stream stream);

/* Device-side window stitcher of VideoDepthAnything.infer_video_depth (video_depth.py:118-156):
 * vdn_stitch_fit   — closed-form least-squares scale/shift of `pred` onto `target` over n f32 values with an all-ones
 *                    mask (compute_scale_and_shift_full, utils/util.py:40-62): coef[0..1] = {scale, shift}, {1, 0} when
 *                    the normal matrix is singular. Deterministic fp64 sums; `workspace` = vdn_stitch_workspace_bytes().
 * vdn_stitch_apply — for a window [T, hw] f32: frames align_len..overlap-1 are clamped-affine-mapped and cross-faded
 *                    into out_tail[overlap-align_len, hw] with weights 0, 1/(n-1), .., 1 (get_interpolate_frames,
 *                    utils/util.py:65-73), frames overlap..T-1 are mapped into out_new, and the mapped frame
 *                    `ref_frame` is also written to ref1 (the moving second alignment target, video_depth.py:150-152). */
size_t vdn_stitch_workspace_bytes(void);
int vdn_stitch_fit(const float* pred, const float* target, size_t n, void* workspace, float* coef, vdn_stream stream);
int vdn_stitch_apply(const float* window, const float* coef, float* out_tail, float* out_new, float* ref1, size_t hw,
                     int T, int align_len, int overlap, int ref_frame, vdn_stream stream);

/* Depth-refiner wrappers v4 / v5 (models/video_depth_model_v5.py:63-87,160-192, models/video_depth_model_v4.py:117-148,
 * utils/normal_utils.py:4-51). f32 throughout, frames are [frames, n = H*W] row-major.
 * vdn_frame_median  — median[f] = torch.quantile(x[f], 0.5) (linear interpolation), exact radix select;
 *                     workspace = vdn_frame_median_workspace_bytes(frames).
 * vdn_refine_scale  — out = x / max_depth * s_f, s_f = exp(tanh(w * median[f] / max_depth + b) * max_log_scale)
 *                     (GlobalScaleHead: quantile pool -> 1x1 conv -> TanhToExp); scale_out[f] = s_f (may be NULL).
 * vdn_refine_pack   — encoder input [frames,3,H,W] = (d, nx, ny); normals != 0: Sobel/8 on a reflect-padded map,
 *                     n = (-Ix,-Iy,1)/sqrt(Ix^2+Iy^2+1+1e-8); normals == 0: d broadcast to 3 channels.
 * vdn_refine_finish — out = (scaled + (w * depth + b)) * max_depth (residual != 0) or depth * max_depth.           */
size_t vdn_frame_median_workspace_bytes(int frames);
int vdn_frame_median(const float* x, int frames, size_t n, float* median, void* workspace, vdn_stream stream);
int vdn_refine_scale(const float* x, const float* median, int frames, size_t n, float w, float b, float max_log_scale,
                     float max_depth, float* out, float* scale_out, vdn_stream stream);
int vdn_refine_pack(const float* d, float* out, int frames, int H, int W, int normals, vdn_stream stream);
int vdn_refine_finish(const float* scaled, const float* depth, float w, float b, float max_depth, int residual, float* out,
                      size_t n, vdn_stream stream);

/* Fused depth tail (depth_anything_v2/dpt.py:146-151; video_depth_anything/dpt_temporal.py:106-111 runs the same ops
 * in micro-batches): bilinear resize (align_corners=True) of the fp32 NHWC map x [B, IH, IW, C] (output_conv1's result,
 * written as ONE fp32 plane) to (OH, OW), Conv3x3(C -> 32, pad 1) + bias2 + ReLU, Conv1x1(32 -> 1) + b1 [+ ReLU when
 * relu != 0] -> depth f32 [B, OH, OW]. The up-sampled map and the 32-channel map stay on chip. w / w_lo: split planes
 * [32, ldb] with K = tap * C + ci (tap = 3 ky + kx; VDN_PACK_CONV3X3_TAPS), ldb >= 9 C; C a multiple of 32; `dt` is
 * the 16-bit type of the weight planes and of the on-chip up-sampled fragments; 3 MFMA products per term. The scale
 * (IH-1)/(OH-1) must be <= 10/17 (a 13 x 13 source patch covers a tile's halo): VDN_EUNSUPPORTED otherwise.           */
int vdn_depth_tail(int dt, const float* x, int B, int IH, int IW, int C, const void* w, const void* w_lo, int ldb,
                   const float* bias2, const float* w1, float b1, float* depth, int OH, int OW, int relu,
                   vdn_stream stream);

/* One-time weight packing ON THE DEVICE, so that a host in any language can feed the library from the reference's
 * fp32 parameter tensors as torch.nn stores them (vdn/pack.py is a thin caller of these). `w` is the contiguous fp32
 * parameter, (hi, lo) the [rows, ldb] planes vdn_gemm reads (lo may be NULL for the 1-product modes): K contiguous,
 * zero padded to ldb = vdn_pack_ldb(..) (a multiple of 64); hi = nearest(w), lo = nearest(w - hi).
 *   kind                 parameter (d0, d1, d2)                rows              K order
 *   VDN_PACK_LINEAR      [N, K] (d0 = N, d1 = K); also 1x1 convs and the 14x14 patch embedding (K = 588)
 *   VDN_PACK_CONV3X3     [Co, Ci, 3, 3] (d0 = Co, d1 = Ci)      Co                (ci/64, tap, ci%64) if Ci % 64 == 0 else (tap, ci)
 *   VDN_PACK_CONV3X3_TAPS  same                                 Co                (tap, ci)            [vdn_depth_tail]
 *   VDN_PACK_CONVT       [Ci, Co, k, k], kernel == stride (d0 = Ci, d1 = Co, d2 = k)   k*k*Co rows (ky, kx, co); K = ci
 *   VDN_PACK_GEGLU       [2 Nh, K] = [h ; gate]                 2 Nh, 16-row blocks alternating h / gate   [VDN_ST_GEGLU]
 *   VDN_PACK_ROPE        [N, K], N % 64 == 0                    per head (2i, 2i+1) -> [re 0-15 | im 0-15 | re 16-31 | im 16-31]
 * Concatenated projections (q|k|v) are packed one after the other into row ranges of one plane (hi + row0 * ldb).
 * vdn_pack_bias writes the matching bias order (row permutation / ConvTranspose repeat) as fp32 [rows].
 * Reference layouts: nn.Linear / nn.Conv2d / nn.ConvTranspose2d weights of dinov2_layers/*.py, util/blocks.py,
 * dpt.py:55-83, motion_module/attention.py:370-384 (GEGLU), sam2/modeling/sam/transformer.py:279-281 (RoPE q/k).   */
enum { VDN_PACK_LINEAR = 0, VDN_PACK_CONV3X3 = 1, VDN_PACK_CONV3X3_TAPS = 2, VDN_PACK_CONVT = 3, VDN_PACK_GEGLU = 4,
       VDN_PACK_ROPE = 5 };
int vdn_pack_rows(int kind, int d0, int d1, int d2);  /* rows of the packed planes, or a negative vdn_status */
int vdn_pack_ldb(int kind, int d0, int d1, int d2);   /* plane stride in elements */
int vdn_pack_weight(int dt, int kind, const float* w, int d0, int d1, int d2, void* hi, void* lo, int ldb, vdn_stream stream);
int vdn_pack_bias(int kind, const float* b, int d0, int d1, int d2, float* out, vdn_stream stream);
/* Operand planes of the cross-term GEMM (vdn_gemm_desc.A8 / W8) from fp16 split planes (hi, lo) [rows, ld] as vdn_pack_weight
 * wrote them (ld % 64 == 0): planes8 = u8 [2, rows, ld], the x6 rows of hi and of lo; kt != 0 stores both K-tile-major
 * ([ld/64][rows][64] each) and, if hi_kt is given, the hi plane again as [ld/32][rows][32].
 * order = which columns of a 64-wide slab half h holds at stream position p (both operands of a GEMM must use the order of
 * the kernel that writes its A planes):
 *   0  col = 32 h + p                                            A from vdn_layernorm or from this function
 *   1  col = 32 (p >> 4) + 16 ((p >> 3) & 1) + 8 h + (p & 7)     A from vdn_gemm out8 (bias + GELU)
 *   2  col = 32 (p >> 4) + 8 ((p >> 2) & 3) + 4 h + (p & 3)      A from vdn_flash_attn out8                              */
int vdn_pack_x8(const void* hi, const void* lo, int rows, int ld, void* hi_kt, void* planes8, int kt, int order,
                vdn_stream stream);
/* The same from an fp32 activation x [rows, ld] (ld % 64 == 0) whose producer writes no half planes (a residual stream that
 * feeds a GEMM without a LayerNorm in between: the memory feature of memory_encoder.py:173-181 before the key / value
 * projections of memory_attention.py): hi_kt = fp16 hi plane toward zero, K-tile-major; planes8 = u8 [2][ld/64][rows][64], order 0. */
int vdn_pack_x8_f32(const float* x, int rows, int ld, void* hi_kt, void* planes8, vdn_stream stream);

/* Workspace sizing (the library allocates nothing): bytes of split-K scratch worth passing as vdn_gemm_desc.splitk_ws
 * for this descriptor (0 = the shape never splits), and the partial-sum buffer of vdn_groupnorm.
 * See also vdn_stitch_workspace_bytes / vdn_frame_median_workspace_bytes.                                       */
size_t vdn_gemm_workspace_bytes(const vdn_gemm_desc* d);
size_t vdn_groupnorm_workspace_bytes(int frames, int groups, int nsplit);

/* misc */
int vdn_cast(const void* x, int x_dt, void* y, int y_dt, size_t n, vdn_stream stream);
size_t vdn_sizeof_gemm_desc(void);      /* layout probes for FFI bindings */
size_t vdn_offsetof_gemm_zeros(void);
size_t vdn_offsetof_gemm_res2_lo(void);
const char* vdn_version(void);
int vdn_arch_ok(void); /* 1 if device 0 is gfx950 */

#ifdef __cplusplus
}
#endif
#endif
