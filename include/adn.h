/*
 * adn.h -- C ABI of libadn.so, the MI355X (gfx950) kernel library behind the
 * audio-depth-estimation hot path.
 *
 * The reference (Kang-ChangWoo/audio-depth-estimation) has no native code and no FFI: its hot
 * path sits behind Python call signatures.  Each entry point below
 * therefore cites the *reference Python call site* whose arithmetic it replaces; the Python
 * mirror in audio-depth-estimation_amd/ keeps those signatures and calls these symbols through
 * ctypes (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - plain C symbols, POD arguments, no torch types; every pointer is a DEVICE pointer owned by
 *     the caller (PyTorch's allocator); the library never allocates, frees or retains memory.
 *   - every call only LAUNCHES work on the given hipStream_t (void* here) and returns; no host
 *     synchronisation, so a caller may capture a sequence of calls into a hipGraph.
 *   - return value: 0 = ok, negative = error; adn_last_error() gives a thread-local message.
 *   - activations are NHWC ("pixel-major, channel-contiguous") in dtype ADN_F32 or ADN_BF16;
 *     accumulation, BatchNorm statistics, loss and optimizer state are always f32 (or f64 for
 *     final reductions).
 *   - "packed weights": S2 form [N][16][C] (tap = kh*4+kw), T2 form [4][N][4][C]
 *     (phase = (oy&1)*2+(ox&1), tap = ty*2+tx), S1 form [N][ks*ks][C] with every row zero-padded to a
 *     multiple of 128 bytes; see adn_pack_weights / adn_pack_transpose_taps.
 */
#ifndef ADN_H_
#define ADN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADN_OK 0
#define ADN_ERR_ARG (-1)
#define ADN_ERR_LAUNCH (-2)
#define ADN_ERR_UNSUPPORTED (-3)

enum { ADN_F32 = 0, ADN_BF16 = 1 };

/* gather geometry of an implicit GEMM */
enum {
  ADN_GEMM_S2 = 0, /* k4 s2 p1 conv forward / k4 s2 p1 transposed-conv dgrad: out = small grid */
  ADN_GEMM_T2 = 1, /* k4 s2 p1 transposed-conv forward / conv dgrad: out = large grid, 4 phases */
  ADN_GEMM_S1 = 2  /* ks x ks, stride 1, pad ks/2 conv (ks = 1 or 3) forward and dgrad: out = in grid
                      (nn.Conv2d(k3,p1) of DoubleConv, binaural_attention_model.py:30,33; 1x1 convs :96-101,:245) */
};

/* epilogue of an implicit GEMM */
enum {
  ADN_EPI_RAW = 0,   /* out0 = v as f32 (debug / building block)                                 */
  ADN_EPI_Z_STATS,   /* z = v (+bias[n]); out0 = z (dtype), partial sums of z and z*z per channel
                        (train-mode BN)                                                          */
  ADN_EPI_ACT,       /* y = v*scale[n]+shift[n] (+bias[n]); out0 = leaky(y,slope) if out0,
                        out1 = relu(y) if out1   (layers without BN, eval-mode BN)               */
  ADN_EPI_BWD,       /* g = v * (ref>0 ? 1 : slope) (+ out0 if accumulate); out0 = g;
                        optional partial sums of g and g*xhat, xhat=(z-mean)*istd                */
  ADN_EPI_FINAL,     /* out0(f32) = final_act(v + bias[n]); final_act: 0 relu, 1 sigmoid         */
  ADN_EPI_ADD        /* out0 (dtype) = (v + bias[n]) * scale[n] (+ ref) (+ out0 if accumulate); bias/scale/ref
                        optional; final_act != 0: scale is ONE device scalar (residual gate gamma).
                        Plain input-gradient accumulation, linear 1x1 projections, x + gamma * proj(att)   */
};

/* One channel segment of the output (virtual concat: the output channels [0,N) may be split
 * in two consecutive segments living in different tensors). */
typedef struct {
  void* out0;         /* primary output, [pixels][channels] of this segment                      */
  void* out1;         /* secondary output (ADN_EPI_ACT relu copy) or NULL                        */
  const void* ref;    /* ADN_EPI_BWD: activated forward tensor (mask source)                     */
  const void* z;      /* ADN_EPI_BWD stats: raw forward conv output                              */
  const float* mean;  /* ADN_EPI_BWD stats                                                       */
  const float* istd;  /* ADN_EPI_BWD stats                                                       */
  const float* scale; /* ADN_EPI_ACT: per-channel scale or NULL (=1).  ADN_EPI_BWD with stats, optional: the scale /
                         shift the forward applied to z (ref = act(z * scale + shift)); a kernel may then take the
                         mask from z and not read ref (the ring kernel does: one operand stream less)              */
  const float* shift; /* ADN_EPI_ACT: per-channel shift or NULL (=0); ADN_EPI_BWD: see scale                      */
  const float* bias;  /* ADN_EPI_ACT / ADN_EPI_FINAL: per-channel bias or NULL                   */
  float* partials;    /* stats partial sums [P][2][channels] or NULL (no stats)                  */
  int32_t channels;   /* channels in this segment                                                */
  float slope;        /* leaky slope (ACT: out0; BWD: mask value where ref<=0)                   */
  int32_t accumulate; /* ADN_EPI_BWD: add the existing out0                                      */
  int32_t final_act;  /* ADN_EPI_FINAL                                                           */
} AdnEpiSeg;

typedef struct {
  int32_t dtype;      /* ADN_F32 / ADN_BF16: dtype of activations and packed weights             */
  int32_t geom;       /* ADN_GEMM_S2 / ADN_GEMM_T2                                               */
  int32_t B;          /* batch                                                                   */
  int32_t Hs, Ws;     /* SMALL grid (S2: output; T2: input)                                      */
  int32_t C0, C1;     /* channels of the two gathered input sources (virtual concat), C1 may be 0*/
  int32_t N;          /* output channels = seg[0].channels + seg[1].channels                     */
  const void* in0;    /* gathered input source 0, NHWC                                           */
  const void* in1;    /* gathered input source 1 or NULL                                         */
  const void* w;      /* packed weights for this geometry                                        */
  int32_t epi;        /* ADN_EPI_*                                                               */
  AdnEpiSeg seg[2];
  void* workspace;    /* split-K / generic-path scratch (f32), adn_igemm_workspace_bytes()       */
  int64_t workspace_bytes;
  int32_t ks;         /* ADN_GEMM_S1: kernel side, 1 or 3 (Hs, Ws = the common grid)              */
  int32_t reserved;
} AdnIgemmDesc;

const char* adn_last_error(void);
int adn_version(void);   /* ABI revision of this header: 3 (round 3: adn_grad_sqsum_partials, adn_debug_poison_lds,
                            adn_debug_stream_rmw) */
/* Debugging aid: fills the LDS of every CU with 0xFFFFFFFF so that a kernel reading LDS it never wrote produces NaN
 * instead of values that depend on the previous kernel (the Python binding calls it in front of every launch when
 * ADN_LDS_POISON=1).  No counterpart in the reference. */
int adn_debug_poison_lds(void* stream);
/* Measurement aid (tools/overlap_experiment.py): dst += src over `bytes` on exactly `workgroups` workgroups, `passes` times --
 * the local HBM traffic of a ring all-reduce kernel, to price the exchange beside the backward pass on one GPU. */
int adn_debug_stream_rmw(const void* src, void* dst, int64_t bytes, int32_t workgroups, int32_t passes, void* stream);

/* Number of stats partial rows P the implicit GEMM will write for this descriptor. */
int64_t adn_igemm_num_partials(const AdnIgemmDesc* d);
int64_t adn_igemm_workspace_bytes(const AdnIgemmDesc* d);

/* Implicit-GEMM convolution family.
 * Replaces: nn.Conv2d(k4,s2,p1) forward  (models/unetbaseline_model.py:187-188)   [S2]
 *           nn.ConvTranspose2d(k4,s2,p1) forward (:196-198,:209-211,:218-220)      [T2]
 *           and their input-gradient passes inside loss.backward() (train.py:674). */
int adn_igemm(const AdnIgemmDesc* d, void* stream);

/* Weight-gradient pass of both layer kinds (train.py:674 loss.backward()).
 *   dW[r][tap][c] = sum_m plain[m][r] * gather_tap(gath)[m][c]        (f32, [R][16][C])
 * conv:  plain = dZ (small grid, R = Cout), gath = layer input  (large grid, C = Cin, 2 sources)
 * convT: plain = layer input (small grid, R = Cin, 2 sources), gath = dZ (large grid, C = Cout) */
typedef struct {
  int32_t dtype;
  int32_t B, Hs, Ws;          /* small grid */
  const void* plain0; const void* plain1; int32_t R0, R1;
  const void* gath0;  const void* gath1;  int32_t C0, C1;
  float* dw;                  /* [R0+R1][16][c_valid] f32 */
  void* workspace; int64_t workspace_bytes;
  int32_t c_valid;            /* 0 = all gathered channels; else only c < c_valid are stored (the edge
                                 layers run with their 2 / 1 real channels zero-padded to one 16-byte chunk) */
  int32_t geom;               /* 0: k4 s2 p1 pair (16 taps, gathered tensor on the 2x grid);
                                 ADN_GEMM_S1: ks x ks stride-1 conv (ks*ks taps, same grid), dw [R][ks*ks][c] */
  int32_t ks;
  double* sq_partials;        /* optional (NULL = off): the kernel that writes the final dw also writes
                                 adn_wgrad_sq_count(d) partial sums of dw^2 here (one double per workgroup of that
                                 kernel), the per-layer share of clip_grad_norm_'s total norm (train.py:689) -- saves
                                 re-reading the gradient.  Summed by adn_grad_norm_ranges.  When the count is 0 (this
                                 descriptor's kernel path has no fused form) nothing is written and the caller covers dw
                                 with a range of adn_grad_norm_ranges instead. */
} AdnWgradDesc;
int64_t adn_wgrad_workspace_bytes(const AdnWgradDesc* d);
int adn_wgrad(const AdnWgradDesc* d, void* stream);
/* Up to 8 independent weight-gradient problems in ONE launch (k4 pair, bf16): the five 512-channel blocks of unet_256
 * (unetbaseline_model.py:141-148) have <= 4 x 4 images -- their weight gradients are six launches of 256-512 short
 * workgroups whose time is launch ramp and dW write latency.  adn_wgrad_batchable: 0 = not batchable, 1 / 2 = class of the
 * problem (one class per launch).  Inside a batch every problem runs unsplit: it writes the final dW and, if asked,
 * adn_wgrad_batch_sq_count(d) norm partials (not adn_wgrad_sq_count: a lone launch of the same layer may split the pixels
 * and leave its partials to the slab sum); dW equals adn_wgrad's up to the summation order of a split launch. */
int32_t adn_wgrad_batchable(const AdnWgradDesc* d);
int32_t adn_wgrad_batch_sq_count(const AdnWgradDesc* d);
int adn_wgrad_batch(const AdnWgradDesc* descs, int32_t n, void* stream);
/* 2 .. 4 PATCH-STAGED weight gradients (the levels with >= 16 x 16 images: L1-L3, D1-D3 of unet_256) in ONE launch, each
 * with 1/n of the pixel splits of a lone launch: the f32 slabs of a launch are (workgroups x 128 KB) whatever the layer,
 * so n layers sharing the workgroups write and re-read 1/n of the slab bytes each.  Slabs back to back in
 * descs[0].workspace (adn_wgrad_patch_batch_workspace_bytes; < 0: some problem is not a patch-staged layer); per problem
 * dW, and sq_partials with adn_wgrad_sq_count entries, as adn_wgrad (equal up to the summation order). */
int64_t adn_wgrad_patch_batch_workspace_bytes(const AdnWgradDesc* descs, int32_t n);
int adn_wgrad_patch_batch(const AdnWgradDesc* descs, int32_t n, void* stream);
/* Number of doubles adn_wgrad writes to d->sq_partials (0: not fused for this descriptor). */
int32_t adn_wgrad_sq_count(const AdnWgradDesc* d);

/* Cast/pack master f32 weights ([X][4][4][Y] memory order = torch channels_last of an
 * [X,Y,4,4] parameter) into the two GEMM operand forms.
 *   s2_out: [X][16][y_pad] in dtype (cast; channels Y..y_pad-1 zero)   or NULL
 *   t2_out: [4][Y][4][X] in dtype (phase split)                        or NULL */
int adn_pack_weights(const float* master, int32_t X, int32_t Y, int32_t y_pad, int32_t dtype,
                     void* s2_out, void* t2_out, void* stream);

/* Stride-1 (S1) operand packs from a master [X][taps][Y] f32 (= channels_last memory of an [X,Y,k,k] Conv2d
 * weight).  Rows of both outputs are zero padded to row_stride elements (a multiple of the 128-byte K-step).
 *   adn_pack_rows:           out [X][taps][y_pad] + zero tail     (forward operand; y_pad pads thin inputs)
 *   adn_pack_transpose_taps: out [Y][taps'][X], taps' flipped     (input-gradient operand of the same conv) */
int adn_pack_rows(const float* master, int32_t X, int32_t taps, int32_t Y, int32_t y_pad,
                  int32_t row_stride, int32_t dtype, void* out, void* stream);
int adn_pack_transpose_taps(const float* master, int32_t X, int32_t taps, int32_t Y, int32_t flip,
                            int32_t row_stride, int32_t dtype, void* out, void* stream);

/* ---- Block-scaled fp8 (OCP MX: e4m3 elements + one E8M0 scale byte per 32 channels) path of the 3 x 3 stride-1
 * convolutions (BASELINE config 5: models/rgb_depth_model.py:21-77 DoubleConv / Down / Up at 512 x 512; forward of
 * nn.Conv2d(k3, p1) and its input-gradient pass inside loss.backward()).  v_mfma_scale_f32_16x16x128_f8f6f4; outputs
 * and epilogues in bf16 exactly as adn_igemm with dtype ADN_BF16, geometry ADN_GEMM_S1, ks 3. ---- */
/* bf16 [rows][C] -> e4m3 [rows][C] + E8M0 [rows][C/32]  (C % 32 == 0, rows*C % 128 == 0). */
int adn_mx8_quantize(const void* src_bf16, int64_t rows, int32_t C, void* dst_e4m3, void* scales_e8m0, void* stream);
/* f32 master [X][9][Y] (channels_last memory of an [X,Y,3,3] Conv2d weight) -> packed operand of adn_conv3x3_mx8:
 *   transpose 0 (forward):        w8 [X][10][Y],  blocks of 32 along Y;  wsc [X][Y/64][5][4]
 *   transpose 1 (input gradient): w8 [Y][10][X],  taps flipped, blocks of 32 along X;  wsc [Y][X/64][5][4]
 * (tap 9 is an all-zero padding tap: 9 taps = 4.5 K-steps of 2 taps.) */
int adn_mx8_pack(const float* master, int32_t X, int32_t Y, int32_t transpose, void* w8, void* wsc, void* stream);
/* The producers of the fp8 path: adn_bn_act / adn_bn_bwd_apply (below) that ALSO write the MX-fp8 copy of their bf16
 * result (bit-identical to adn_mx8_quantize of that result), saving one pass over the tensor.  C % 32 == 0. */
int adn_bn_act_mx8(const void* z, int64_t pixels, int32_t C, const float* scale, const float* shift, void* out_relu,
                   void* out8, void* out_scales, void* stream);
int adn_bn_bwd_apply_mx8(void* g, const void* z, int64_t pixels, int32_t C, const float* scale, const float* mean,
                         const float* istd, const float* coef, void* out8, void* out_scales, void* stream);
int adn_maxpool2_fwd_mx8(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, void* out8,
                         void* out_scales, void* stream);      /* adn_maxpool2_fwd (bf16) + fp8 copy of dst */
int adn_upsample2x_fwd_mx8(const void* src, void* dst, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                           int32_t C, void* out8, void* out_scales, void* stream);   /* adn_upsample2x_fwd (bf16) + copy */
typedef struct {
  int32_t B, H, W;     /* common grid of input and output (H % 8 == 0, W % 16 == 0)                 */
  int32_t C0, C1, N;   /* gathered sources (virtual concat, multiples of 64; C1 may be 0), outputs  */
  const void* in0;     /* e4m3 [B][H][W][C0]                                                       */
  const void* sc0;     /* E8M0 [B][H][W][C0/32]                                                    */
  const void* in1;
  const void* sc1;
  const void* w;       /* adn_mx8_pack w8                                                          */
  const void* wsc;     /* adn_mx8_pack wsc                                                         */
  int32_t epi;         /* ADN_EPI_Z_STATS / ADN_EPI_ACT / ADN_EPI_BWD / ADN_EPI_ADD                 */
  int32_t reserved;
  AdnEpiSeg seg[2];    /* bf16 tensors                                                             */
} AdnMx8ConvDesc;
int64_t adn_conv3x3_mx8_num_partials(const AdnMx8ConvDesc* d);   /* stats partial rows = B*H*W / rows per workgroup (128 | 256) */
int adn_conv3x3_mx8(const AdnMx8ConvDesc* d, void* stream);

/* ---- DoubleConv U-Net family (DoubleConv / Down / Up: binaural_attention_model.py:22-78, identical copies
 * rgb_depth_model.py:21-77, adabins_distillation_model.py:27-82).  NHWC activations in dtype. ---- */
/* Tail-fused forms of adn_maxpool2_bwd / adn_upsample2x_bwd: when this kernel is the LAST writer of the gradient of a
 * ConvBNReLU output it also applies that output's ReLU mask (y > 0) and writes the BatchNorm-backward partial sums
 * [adn_tail_stats_blocks(work, C)][2][C] (what adn_relu_bwd_stats would do in a separate pass over the tensor).
 * work = B * ceil(H/2) * ceil(W/2) * C/8 (max-pool) or B * Hi * Wi * C/8 (upsample). */
int64_t adn_tail_stats_blocks(int64_t work, int32_t C);
int adn_maxpool2_bwd_tail(const void* gdst, const void* y, void* gsrc, int32_t B, int32_t H, int32_t W, int32_t C,
                          int32_t accumulate, int32_t dtype, const void* z, const float* mean, const float* istd,
                          float* partials, void* stream);
int adn_upsample2x_bwd_tail(const void* gdst, void* gsrc, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                            int32_t C, int32_t accumulate, int32_t dtype, const void* y, const void* z, const float* mean,
                            const float* istd, float* partials, void* stream);
/* nn.MaxPool2d(2): src [B][H][W][C] -> dst [B][H/2][W/2][C]; backward routes each window's gradient to its
 * first maximum (torch's tie rule), gsrc = (accumulate ? gsrc : 0) + routed. */
int adn_maxpool2_fwd(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C,
                     int32_t dtype, void* stream);
int adn_maxpool2_bwd(const void* gdst, const void* y, void* gsrc, int32_t B, int32_t H, int32_t W,
                     int32_t C, int32_t accumulate, int32_t dtype, void* stream);
/* nn.Upsample(scale_factor=2, bilinear, align_corners=True) + F.pad to the skip's Ho x Wo
 * (pad top/left = diff // 2).  src [B][Hi][Wi][C] -> dst [B][Ho][Wo][C]; backward is a gather. */
int adn_upsample2x_fwd(const void* src, void* dst, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho,
                       int32_t Wo, int32_t C, int32_t dtype, void* stream);
int adn_upsample2x_bwd(const void* gdst, void* gsrc, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho,
                       int32_t Wo, int32_t C, int32_t accumulate, int32_t dtype, void* stream);
/* nn.ConvTranspose2d(C_in, C, kernel_size=2, stride=2) of `Up(bilinear=False)` (binaural_attention_model.py:64-67,
 * rgb_depth_model.py:63-66) = a 1x1 implicit GEMM (adn_igemm, ADN_GEMM_S1, ks 1) to packed [B][H][W][4][C] (tap
 * t = 2i + j) + this shuffle: spatial [B][2H][2W][C] with spatial[b][2y+i][2x+j][c] = packed[b][y][x][2i+j][c].
 * inverse = 0: src packed -> dst spatial; inverse = 1: src spatial -> dst packed (gradient gather). */
int adn_pixel_shuffle2(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C,
                       int32_t inverse, int32_t dtype, void* stream);
/* ReLU + BatchNorm backward, first pass, for a gradient that did not come through a GEMM epilogue:
 * g <- g * (y > 0) in place and partial sums [P][2][C] of g and g * xhat for adn_bn_bwd_finalize
 * (P = adn_relu_bwd_stats_num_partials). */
int64_t adn_relu_bwd_stats_num_partials(int64_t pixels, int32_t C);
int adn_relu_bwd_stats(void* g, const void* y, const void* z, const float* mean, const float* istd,
                       int64_t pixels, int32_t C, int32_t dtype, float* partials, void* stream);
/* 1x1 conv to one channel + output activation (rgb_depth_model.py:195-209: outc, clamp(0, max_depth);
 * binaural_attention_model.py:330-337: sigmoid(outc) * max_depth, clamp).  act 0 clamp, 1 sigmoid,
 * 3: identity (the clamp follows the final resize, adn_clamp_range);
 * 2: tanh(z) * max_depth with no clamp (AdaBins residual head, adabins_distillation_model.py:330-335; pass
 * max_depth = 0.05 * cfg max_depth).
 * zpre/out f32 [pixels]; backward writes gx (dtype [pixels][C]), dw [C], db [1]. */
int adn_head1x1_fwd(const void* x, const float* w, const float* bias, int64_t pixels, int32_t C,
                    int32_t dtype, int32_t act, float max_depth, float* zpre, float* out, void* stream);
int64_t adn_head1x1_bwd_workspace_bytes(int64_t pixels, int32_t C);
int adn_head1x1_bwd(const float* gout, const float* zpre, const void* x, const float* w, int64_t pixels,
                    int32_t C, int32_t dtype, int32_t act, float max_depth, void* gx, float* dw,
                    float* db, void* workspace, int64_t workspace_bytes, void* stream);
/* torch.clamp(x, 0, max_depth) applied AFTER the final resize (rgb_depth_model.py:209): g == NULL: out = clamp(x);
 * g != NULL: out = g where 0 <= x <= max_depth else 0 (its backward).  The head then runs with act 3 (identity). */
int adn_clamp_range(const float* x, const float* g, int64_t n, float max_depth, float* out, void* stream);
/* DepthLoss (train_rgb_depth.py:43-87): lambda_l1 * mean|p-g| + lambda_smooth * (mean|dx p| + mean|dy p|),
 * unmasked.  stats f64[4] = [sum|p-g|, sum|dx|, sum|dy|, 0]; a data-parallel caller all-reduces stats and
 * passes replicas = world size (the means are over the global batch). */
int64_t adn_l1tv_workspace_bytes(int64_t n);
int adn_l1tv_stats(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W, double* stats,
                   void* workspace, int64_t workspace_bytes, void* stream);
int adn_l1tv_finish(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W,
                    const double* stats, int32_t replicas, float lambda_l1, float lambda_smooth,
                    float* loss_out, float* grad, void* stream);

/* ---- Binaural cross-attention (BinauralCrossAttention.forward, binaural_attention_model.py:106-153) ----
 * Streaming-softmax attention over the N = H*W tokens of NHWC tensors; the N x N scores are never stored.
 * Rows are token-major and channel-contiguous with arbitrary row strides (slices of a fused q|k|v buffer).
 * Entry b of the B2 stacked batch entries attends from its queries to the keys/values of entry
 * (b + kv_shift) % B2: [left; right] stacked along the batch gives both directions in one launch.
 *   forward : o = softmax_keys(scale * q k^T) v,  lse = logsumexp_keys(scale * q k^T)       (:120-127)
 *   backward: dq, dk, dv from dout (recomputing the probabilities from lse); workspace holds
 *             D = rowsum(dout * o), f32 [B2][N] (adn_attn_bwd_workspace_bytes). */
typedef struct {
  int32_t dtype;
  int32_t B2, N, dqk, dv, kv_shift;
  const void* q; const void* k; const void* v;       /* [B2][N][ld_*] */
  int32_t ld_q, ld_k, ld_v;
  void* o; int32_t ld_o;                             /* [B2][N][ld_o], written by fwd, read by bwd */
  float* lse;                                        /* [B2][N] f32 */
  float scale;                                       /* 1 / sqrt(C) (:121) */
  /* backward only */
  const void* dout; int32_t ld_do;
  void* dq; void* dk; void* dvp;
  int32_t ld_dq, ld_dk, ld_dv;
  void* workspace; int64_t workspace_bytes;
} AdnAttnDesc;
int adn_attn_fwd(const AdnAttnDesc* d, void* stream);
int64_t adn_attn_bwd_workspace_bytes(const AdnAttnDesc* d);
int adn_attn_bwd(const AdnAttnDesc* d, void* stream);
/* out[c] = sum over rows of x[row][c], x [rows][ld] (bias gradients of the 1x1 projections). */
int64_t adn_channel_sum_workspace_bytes(int64_t rows, int32_t C);
int adn_channel_sum(const void* x, int64_t rows, int32_t C, int32_t ld, int32_t dtype, float* out,
                    void* workspace, int64_t workspace_bytes, void* stream);
/* Backward of the gated residual x + gamma * out_proj(att) (:130-133).  t (dtype, n elements) holds the
 * input gradient of out_proj for the UNSCALED upstream gradient G; att the attention output; gsum[c] = sum G.
 * dgamma = sum(t * att) + sum_c bias[c] * gsum[c];  t <- gamma * t;  dbias = gamma * gsum;  dw *= gamma
 * (dw = wgrad(G, att) computed by the caller, nw elements).  workspace: 8 KiB. */
int adn_gate_bwd(void* t, const void* att, int64_t n, int32_t dtype, const float* gamma,
                 const float* gsum, const float* bias, int32_t C, float* dgamma, float* dbias,
                 float* dw, int64_t nw, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- AdaBins distillation model (adabins_distillation_model.py:105-207, 301-399) and DistillationLoss
 * (utils_distillation_loss.py:48-238): everything that is not a convolution. ---- */
/* Per-sample reductions over the HW rows of x [B][HW][ld] (first C columns):
 *   nq = 1: out [B][C]    = scale * sum x        (AdaptiveAvgPool2d(1) :139, spatial mean of the bin logits :118-119)
 *   nq = 3: out [B][3][C] = sum x^2, sum y^2, sum x*y   (cosine similarity of spatially normalised features :86-93) */
int64_t adn_pool_workspace_bytes(int32_t B, int32_t HW, int32_t C, int32_t nq);
int adn_pool(const void* x, const void* y, int32_t B, int32_t HW, int32_t C, int32_t ld, int32_t nq,
             int32_t dtype, float scale, float* out, void* workspace, int64_t workspace_bytes, void* stream);
/* AdaBinsBinPredictor (:127-149) on the pooled features g [B][Cb]: Linear(Cb,Hd) + ReLU + Dropout(mask, may be
 * NULL) + Linear(Hd,nb) + Softmax -> widths; cumsum edges * max_depth -> centres (midpoints).  h1 = hidden
 * activations after ReLU and dropout.  Backward takes d loss / d centres and writes PER-SAMPLE partial weight
 * gradients dW2p [B][nb][Hd], db2p [B][nb], dW1p [B][Hd][Cb], db1p [B][Hd] (sum over B with adn_channel_sum) and
 * dg [B][Cb]. */
int adn_binpred_fwd(const float* g, const float* W1, const float* b1, const float* W2, const float* b2,
                    const uint8_t* mask, float drop_p, float max_depth, int32_t B, int32_t Cb,
                    int32_t Hd, int32_t nb, float* h1, float* widths, float* centers, void* stream);
int adn_binpred_bwd(const float* dcent, const float* widths, const float* h1, const float* g,
                    const float* W1, const float* W2, int32_t has_mask, float drop_p, float max_depth,
                    int32_t B, int32_t Cb, int32_t Hd, int32_t nb, float* dW2p, float* db2p, float* dW1p,
                    float* db1p, float* dg, void* stream);
/* Bernoulli(1-p) keep mask from a counter-based hash of (seed, index): nn.Dropout(0.1) :144 (statistically
 * equivalent draw; torch's RNG stream is not reproducible from outside torch).  counter (optional, device f64[1],
 * e.g. the optimizer's step count) is mixed into the seed ON THE DEVICE so that a hipGraph replay draws a fresh mask. */
int adn_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, const double* counter, void* stream);
/* gx [B][HW][C] (+)= scale * dg [B][C]: backward of the average pool. */
int adn_bcast_add(void* gx, const float* dg, int32_t B, int32_t HW, int32_t C, float scale,
                  int32_t accumulate, int32_t dtype, void* stream);
/* Soft binning (:198-201): base[pix] = sum_k softmax(logits[pix])_k * centres[b][k].  Backward:
 * dlogits = p_k (c_k - base) dbase + dmean[b][k] / HW (dmean: gradient wrt the spatial mean of the logits, may be
 * NULL), dcent [B][nb] = sum_pix p_k dbase. */
int adn_bins_fwd(const void* logits, const float* centers, int32_t B, int32_t HW, int32_t nb,
                 int32_t dtype, float* base, void* stream);
int64_t adn_bins_bwd_workspace_bytes(int32_t B, int32_t HW, int32_t nb);
int adn_bins_bwd(const void* logits, const float* centers, const float* base, const float* dbase,
                 const float* dmean, int32_t B, int32_t HW, int32_t nb, int32_t dtype, void* dlogits,
                 float* dcent, int32_t dcent_accumulate, void* workspace, int64_t workspace_bytes,
                 void* stream);
/* Pixel terms of DistillationLoss with valid = gt > 0 (train_adabins_distillation.py:449):
 * final = clamp(base + resid, 0, max_depth) (:337); stats f64[4] = [N_valid, sum|final-gt|, sum(final-teacher)^2,
 * sum|resid|] (teacher may be NULL); gradients wrt base and resid for
 * lambda_task * L1 + lambda_response * MSE + lambda_sparse * mean|resid|.  workspace: 32 KiB. */
int adn_distill_pix_stats(const float* base, const float* resid, const float* gt, const float* teacher,
                          int64_t n, float max_depth, float* final_out, double* stats, void* workspace,
                          int64_t workspace_bytes, void* stream);
int adn_distill_pix_grad(const float* base, const float* resid, const float* gt, const float* teacher,
                         int64_t n, float max_depth, const double* stats, float lambda_task,
                         float lambda_response, float lambda_sparse, float* dbase, float* dres,
                         void* stream);
/* ga += coef * d cos(a, r) / d a per (sample, channel) over the spatial positions, stats from adn_pool(nq=3). */
int adn_featcos_grad(const void* a, const void* r, const float* stats, int32_t B, int32_t HW, int32_t C,
                     int32_t dtype, float coef, void* ga, void* stream);
/* The small terms and the total (utils_distillation_loss.py:105-143, 220-226): feature loss from the five
 * statistics, KL(batchmean) of the temperature-softened mean logits, bin-centre MSE; terms f32[8] = task, response,
 * feature, bin, bin_centers, sparse, total, N_valid; dmean / dcent [B][nb] = gradients wrt the student's mean logits /
 * the extra gradient wrt its bin centres. */
typedef struct {
  const float* mean_student; const float* mean_teacher;        /* [B][nb] */
  const float* centers_student; const float* centers_teacher;  /* [B][nb] */
  const float* feat_stats[5]; int32_t feat_channels[5];        /* [B][3][C] per level x1..x5 */
  const double* pix_stats;
  int32_t B, nb, has_teacher;
  float temperature, lambda_task, lambda_response, lambda_feature, lambda_bin, lambda_sparse;
  float* terms; float* dmean; float* dcent;
} AdnDistillSmall;
int adn_distill_small(const AdnDistillSmall* d, void* stream);

/* ---- Base + Residual model (base_residual_model.py:83-217, utils_base_residual_loss.py:28-160) ---- */
/* Structural target: avg_pool2d(gt, k, stride 1, padding k/2) (zeros counted) then bilinear resize
 * (align_corners=False) back to H x W (utils_base_residual_loss.py:91-107).  gt, out f32 [B][H][W]. */
int64_t adn_lowpass_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t k);
int adn_lowpass(const float* gt, int32_t B, int32_t H, int32_t W, int32_t k, float* out, void* workspace,
                int64_t workspace_bytes, void* stream);
/* out = clamp(a + b, 0, max_depth): final_depth (base_residual_model.py:197-200). */
int adn_clamp_add(const float* a, const float* b, int64_t n, float max_depth, float* out, void* stream);
/* With valid = gt > 0: stats f64[3] = [N_valid, sum|base - struct|, sum|resid|]; terms f32[4] = recon (as
 * delivered, already weighted), mean|base - struct|, mean|resid|, total = recon + lambda_base * .. +
 * lambda_sparse * ...  Gradients: gfinal = d (weighted recon) / d final from adn_loss_finish; dbase / dres include the
 * clamp mask and the structural / sparsity terms.  workspace: 24 KiB. */
int adn_baseres_stats(const float* base, const float* resid, const float* strct, const float* gt, int64_t n,
                      const float* recon, float lambda_recon, float lambda_base, float lambda_sparse,
                      double* stats, float* terms, void* workspace, int64_t workspace_bytes, void* stream);
int adn_baseres_grad(const float* base, const float* resid, const float* strct, const float* gt,
                     const float* gfinal, int64_t n, float max_depth, const double* stats,
                     float lambda_base, float lambda_sparse, float* dbase, float* dres, void* stream);

/* NCHW f32 <-> NHWC dtype layout conversion of the network input/output
 * (model(audio) boundary, train.py:642).  dst has c_pad >= C channels, the extra ones zero. */
int adn_nchw_to_nhwc(const float* src, void* dst, int32_t B, int32_t C, int32_t c_pad, int32_t H,
                     int32_t W, int32_t dtype, void* stream);
int adn_nhwc_to_nchw(const void* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                     int32_t dtype, void* stream);
/* Same for the channel range [c_lo, c_lo + C) of a [B][C_total][H][W] tensor (x[:, 0:1] / x[:, 1:2] split of the
 * binaural input, binaural_attention_model.py:292-293). */
int adn_nchw_slice_to_nhwc(const float* src, void* dst, int32_t B, int32_t C_total, int32_t c_lo,
                           int32_t C, int32_t c_pad, int32_t H, int32_t W, int32_t dtype, void* stream);

/* BatchNorm2d (nn.BatchNorm2d via get_norm_layer('batch'), unetbaseline_model.py:68-69,190-192).
 * Train-mode forward finalize: partial sums [P][2][C] -> mean, istd, scale=gamma*istd,
 * shift=beta-mean*scale; running stats updated with momentum (unbiased variance). */
int adn_bn_fwd_finalize(const float* partials, int64_t P, int32_t C, int64_t count,
                        const float* gamma, const float* beta, float eps, float momentum,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        float* mean, float* istd, float* scale, float* shift, void* stream);
/* Small tensors (the innermost levels): statistics finalize + apply in ONE launch, a workgroup per 32 channels and row
 * slice (each re-reduces the partial rows of its channels: they are L2 resident).  Same arithmetic as
 * adn_bn_fwd_finalize + adn_bn_act, resp. adn_bn_bwd_finalize + adn_bn_bwd_apply (g in place).  C % 32 == 0. */
int adn_bn_fwd_fused(const float* partials, int64_t P, int32_t C, int64_t count, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float* mean, float* istd,
                     float* scale, float* shift, const void* z, int64_t pixels, int32_t dtype, float slope,
                     void* out_leaky, void* out_relu, void* stream);
int adn_bn_bwd_fused(const float* partials, int64_t P, int32_t C, int64_t count, float* dgamma,
                     float* dbeta, void* g, const void* z, int64_t pixels, int32_t dtype,
                     const float* scale, const float* mean, const float* istd, void* stream);

/* Eval-mode: scale/shift from running statistics. */
int adn_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int32_t C, float* scale, float* shift,
                       void* stream);
/* y = z*scale+shift; out_leaky = leaky(y, slope) (optional), out_relu = relu(y) (optional). */
int adn_bn_act(const void* z, int64_t pixels, int32_t C, int32_t dtype, const float* scale,
               const float* shift, float slope, void* out_leaky, void* out_relu, void* stream);
/* Backward finalize: partial sums of g and g*xhat -> dgamma, dbeta and c1=sum g/count,
 * c2=sum g*xhat/count (coef[0..C) = c1, coef[C..2C) = c2). */
/* Pre-reduction of BatchNorm partial rows (forward or backward statistics) [P][2][C] -> [slices][2][C], coalesced, for layers
 * with thousands of rows; adn_bn_fwd_finalize / adn_bn_bwd_finalize then run on the `slices` rows.  C % 32 == 0. */
int adn_bn_partials_reduce(const float* partials, int64_t P, int32_t C, int32_t slices, float* out_rows, void* stream);
int adn_bn_bwd_finalize(const float* partials, int64_t P, int32_t C, int64_t count, float* dgamma,
                        float* dbeta, float* coef, void* stream);
/* In place: g <- scale * (g - c1 - xhat*c2), xhat = (z-mean)*istd. */
int adn_bn_bwd_apply(void* g, const void* z, int64_t pixels, int32_t C, int32_t dtype,
                     const float* scale, const float* mean, const float* istd, const float* coef,
                     void* stream);

/* Masked depth loss (train.py:646-669, utils_loss.py:29-49).
 * stats[0..4) (f64) = N, sum|p-g|, sum d, sum d^2 over valid pixels; mask_mode 0: gt != 0, 1: gt > 0,
 * 2: every element (inputs already gathered, utils_loss.py call sites).
 * adn_loss_stats zeroes nothing: it overwrites stats.  A data-parallel caller all-reduces stats
 * (4 doubles) between the two calls to reproduce the reference's single global-batch loss. */
int adn_loss_stats(const float* pred, const float* gt, int64_t n, float scale, int32_t mask_mode,
                   float eps, double* stats, void* workspace, int64_t workspace_bytes, void* stream);
int64_t adn_loss_workspace_bytes(int64_t n);
/* criterion: 0 L1, 1 SIlog, 2 Combined (l1_weight * L1 + silog_weight * SIlog), 4 masked MSE scaled by l1_weight (its
 * statistics are taken with mask_mode | 4: sum of squared instead of absolute errors).  loss_out (f32 scalar) and grad
 * (f32, d loss/d pred). */
int adn_loss_finish(const float* pred, const float* gt, int64_t n, float scale, int32_t mask_mode,
                    float eps, const double* stats, int32_t criterion, float l1_weight,
                    float silog_weight, float silog_lambda, float* loss_out, float* grad,
                    void* stream);
/* adn_loss_finish for a 1-channel prediction that is the OUTPUT of the generator's last activation (final_act 0 ReLU,
 * 1 Sigmoid; train.py:656-674 through unetbaseline_model.py:201-206): dz (f32, n) = d loss / d pred * act'(pred), i.e. what
 * adn_final_act_bwd makes of adn_loss_finish's grad, and bias_grad (optional, f32 scalar) = sum(dz), the gradient of the last
 * layer's bias -- both without a pass of their own.  criterion 0-2.  workspace: min(4096, ceil(n / 256)) doubles. */
int adn_loss_finish_dz(const float* pred, const float* gt, int64_t n, float scale, int32_t mask_mode, float eps,
                       const double* stats, int32_t criterion, float l1_weight, float silog_weight, float silog_lambda,
                       float* loss_out, float* dz, int32_t final_act, float* bias_grad, void* workspace,
                       int64_t workspace_bytes, void* stream);
/* Derivative of the generator's last activation (ReLU or Sigmoid, unetbaseline_model.py:201-206):
 * dz (dtype, [pixels][c_pad], channel 0 = gout * act'(out), other channels zero). */
int adn_final_act_bwd(const float* gout, const float* out, int64_t n, int32_t final_act,
                      int32_t dtype, int32_t c_pad, void* dz, void* stream);

/* Outermost ConvTranspose2d(k4,s2,p1) with ONE output channel + bias + ReLU/Sigmoid
 * (unetbaseline_model.py:196-206), as a pointwise GEMM P[m][16 taps] = in[m][:] . W[:][tap] on the
 * small grid followed by a 4-tap gather per output pixel (col2im).  w = f32 master [C0+C1][16].
 * out: f32 [B][2Hs][2Ws].  Workspace holds P (f32 [B*Hs*Ws][16]). */
int64_t adn_convt_n1_workspace_bytes(int32_t B, int32_t Hs, int32_t Ws);
int adn_convt_n1_forward(int32_t dtype, int32_t B, int32_t Hs, int32_t Ws, const void* in0, int32_t C0,
                         const void* in1, int32_t C1, const float* w, const float* bias,
                         int32_t final_act, float* out, void* workspace, int64_t workspace_bytes,
                         void* stream);
/* Thin outermost layers of unet_256 on the bf16 path (csrc/edge.hip): HBM-bound kernels, one MFMA per 16 pixels, the
 * thin operand read as planar f32 and rounded to bf16 in registers (replace the channel-padded adn_igemm / adn_wgrad
 * calls of the first Conv2d(2->64,k4,s2,p1) and the last ConvTranspose2d(128->1,k4,s2,p1),
 * unetbaseline_model.py:187-198).
 * adn_l0_forward: x f32 [B][2][2Hs][2Ws] (the network input, NCHW), w f32 [64][16][2] (parameter memory,
 *   channels_last), outputs bf16 [B][Hs][Ws][64]: leaky (slope) and / or relu copy of the conv result.  Ws % 16 == 0. */
int adn_l0_forward(const float* x, const float* w, int32_t B, int32_t Hs, int32_t Ws, int32_t cin,
                   int32_t cout, float slope, void* out_leaky, void* out_relu, void* stream);
/* adn_d0_dgrad: dz f32 [B][2Hs][2Ws] (gradient of the 1-channel output before the final activation), w f32 [128][16]
 *   ([Cin][kh][kw][1]); seg0 / seg1 = skip / up half of the 128 input channels (64 each, bf16 NHWC): out0 = dIn masked
 *   by ref > 0 (else * slope); seg1 may carry BatchNorm-backward statistics (z, mean, istd, partials: one row of
 *   [2][64] per workgroup, adn_d0_dgrad_num_partials rows).  Ws % 16 == 0. */
int64_t adn_d0_dgrad_num_partials(int32_t B, int32_t Hs, int32_t Ws);
int adn_d0_dgrad(const float* dz, const float* w, int32_t B, int32_t Hs, int32_t Ws, const AdnEpiSeg* seg0,
                 const AdnEpiSeg* seg1, void* stream);
/* adn_thin_wgrad: dw[c][tap*ct_n + ct] = sum over pixels plain[b,i,j,c] * thin[b,ct,2i-1+kh,2j-1+kw]; thin f32
 *   [B][ct_n][2Hs][2Ws], plain0 / plain1 bf16 [B][Hs][Ws][c0 | c1].  (ct_n, c0, c1) = (1, 64, 64): weight gradient of the
 *   last transposed conv (thin = dz); (2, 64, 0): of the first conv (plain = its output gradient, thin = the network
 *   input).  Deterministic (per-workgroup slabs in the workspace + fixed-order sum).  Ws % 32 == 0. */
int64_t adn_thin_wgrad_workspace_bytes(int32_t B, int32_t Hs, int32_t Ws, int32_t ct_n, int32_t c0, int32_t c1);
int adn_thin_wgrad(const float* thin, int32_t ct_n, const void* plain0, int32_t c0, const void* plain1,
                   int32_t c1, int32_t B, int32_t Hs, int32_t Ws, float* dw, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* sum over n f32/dtype elements into one f32 (bias gradient of the outermost ConvTranspose2d). */
int adn_sum_to_scalar(const void* x, int64_t n, int32_t dtype, float* out, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* Sums of squares of grads[0:n] as adn_grad_norm_workspace_bytes(n)/8 doubles (the first half of adn_grad_norm).  Used per
 * all-reduce bucket by the data-parallel step (ddp.py; replaces the post-exchange norm pass of
 * torch.nn.utils.clip_grad_norm_, train.py:267); finish with adn_grad_norm_ranges(extra = the collected slots). */
int adn_grad_sqsum_partials(const float* grads, int64_t n, double* partials, int64_t partials_bytes, void* stream);

/* Gradient clipping + optimizer (train.py:689-691: clip_grad_norm_(params, 1.0); optimizer.step()).
 * state (device, f64[8]): [0] step count, [1] bias_corr1, [2] bias_corr2, [3] total grad norm,
 * [4] clip coefficient.  All on device: no host round trip, graph-capturable. */
int adn_grad_norm(const float* grads, int64_t n, float max_norm, double* state, void* workspace,
                  int64_t workspace_bytes, void* stream);
int64_t adn_grad_norm_workspace_bytes(int64_t n);
/* The same total norm / clip coefficient from two sources: (1) `ranges` = n_ranges rows (offset, length) in elements
 * into grads (offset % 4 == 0, length <= 8192: one workgroup per row) for the gradients nobody summed yet, and
 * (2) n_extra partial sums of squares already written by the gradients' producers (AdnWgradDesc.sq_partials).
 * state[3] = total norm, state[4] = clip coefficient as adn_grad_norm.  workspace: n_ranges doubles. */
int adn_grad_norm_ranges(const float* grads, const int64_t* ranges, int32_t n_ranges, const double* extra,
                         int32_t n_extra, float max_norm, double* state, void* workspace, int64_t workspace_bytes,
                         void* stream);
/* kind: 0 AdamW (decoupled decay), 1 Adam (L2 in gradient), 2 SGD. Advances state[0..2].
 * bf16_copy (optional, n bf16): mirror of the updated parameters, i.e. next step's S2 GEMM operands. */
int adn_optimizer_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                       int64_t n, int32_t kind, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int32_t use_clip, double* state, void* bf16_copy,
                       void* stream);
/* All T2 (phase-split) weight packs of a network in ONE launch.  table (device, int64 [L][5]):
 * master offset (elements) into flat_master, X, Y, t2 offset (elements) into t2_base, first block index;
 * total_blocks = sum over layers of ceil(X/64)*ceil(Y/64)*16.  flat_master holds the parameters in
 * master_dtype: ADN_F32 (the f32 masters) or ADN_BF16 (the bf16 mirror adn_optimizer_step keeps, same
 * offsets: half the read traffic, identical values since bf16(master) is what the mirror stores). */
int adn_pack_t2_multi(const void* flat_master, int32_t master_dtype, const int64_t* table, int32_t layers,
                      int64_t total_blocks, int32_t dtype, void* t2_base, void* stream);

/* Edge-aware / smoothness loss of the binaural family (utils_binaural_attention_loss.py:15-156) on single-channel f32
 * maps [B][H][W]: lambda_recon * L1 over valid (gt > 0) + lambda_edge * |Sobel magnitude(pred) - Sobel magnitude(gt)|
 * over the 3x3-dilated valid mask + lambda_smooth * (|Sx pred| + |Sy pred|) * exp(-Sobel magnitude(gt)) over valid.
 * stats (5 doubles: n valid, sum |e|, n dilated, sum edge, sum smooth), terms (4 floats: recon, edge, smooth, total),
 * grad (optional, f32 [B][H][W]): d total / d pred. */
int64_t adn_edge_loss_workspace_bytes(int32_t B, int32_t H, int32_t W);
int adn_edge_loss(const float* pred, const float* gt, int32_t B, int32_t H, int32_t W, float lambda_recon,
                  float lambda_edge, float lambda_smooth, double* stats, float* terms, float* grad,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* Evaluation metrics (compute_errors, utils_criterion.py:6-90), one set of 7 floats per sample:
 * (abs_rel, rmse, a1, a2, a3, log_10, mae). gt/pred: [samples][pixels] f32. */
int adn_compute_errors(const float* gt, const float* pred, int32_t samples, int64_t pixels,
                       float* out7, void* workspace, int64_t workspace_bytes, void* stream);
int64_t adn_compute_errors_workspace_bytes(int32_t samples, int64_t pixels);

/* Audio front-end (BatvisionV2_Dataset.py:94-135,177-197; BatvisionV1_Dataset.py:68-95;
 * utils_dataset.py:18-20): wave [B][2][T] f32 -> input [B][2][S][S] f32.
 * mode 0: BV2 mel (hop 32, 32 mel bins, log, min-max); 1: BV2 linear (hop 16, log, min-max);
 * 2: BV1 linear (hop 16, raw magnitude).  antialias: torchvision Resize antialias flag. */
int64_t adn_frontend_workspace_bytes(int32_t B, int32_t T, int32_t mode);
int adn_frontend(const float* wave, int32_t B, int32_t T, int32_t mode, int32_t S,
                 int32_t antialias, float* out, void* workspace, int64_t workspace_bytes,
                 void* stream);
/* Depth-target preparation on the device (BatvisionV2_Dataset.py:65-78, BatvisionV1_Dataset.py:45-64): src = raw
 * depth maps in millimetres [planes][H][W] (src_type 0 f32, 1 u16, 2 i32) -> out f32 [planes][S][S] in metres: NaN and
 * +-inf -> 0, clip to max_depth (if > 0), negatives -> 0, cv2.INTER_NEAREST resize, then / norm if norm > 0
 * (depth_norm: norm = max_depth). */
int adn_depth_prepare(const void* src, int32_t src_type, int32_t planes, int32_t H, int32_t W, int32_t S,
                      float max_depth, float norm, float* out, void* stream);
/* Camera-image preparation on the device (BatvisionV2_Dataset.py:199-210 _load_image, everything after cv2.imread):
 * src u8 [B][H][W][3] BGR -> out f32 [B][3][S][S] RGB in [0,1]: BGR2RGB, cv2.resize((S,S)) with the integer arithmetic
 * of OpenCV's 8-bit INTER_LINEAR path (11-bit coefficients), / 255, HWC -> CHW. */
int adn_image_prepare(const void* src_bgr_u8, int32_t B, int32_t H, int32_t W, int32_t S, float* out, void* stream);
/* F.interpolate(..., size=(S,S), mode='nearest') of f32 maps [planes][H][W] (source index floor(dst * in / out)): the
 * AdaBins model's logits / residual when output_size != input size (adabins_distillation_model.py:196-198, 334-337, 383-386;
 * softmax expectation, tanh and clamp are per-pixel, so resizing their results is the same as resizing their inputs). */
int adn_resize_nearest(const float* src, int64_t planes, int32_t H, int32_t W, int32_t S, float* out, void* stream);
/* Its backward (the AdaBins distillation step when output_size != input size): gsrc [planes][H][W] <- sum of gout
 * [planes][S][S] over the output pixels that read each source pixel. */
int adn_resize_nearest_bwd(const float* gout, int64_t planes, int32_t H, int32_t W, int32_t S, float* gsrc, void* stream);
/* transforms.Resize((S,S)) alone (utils_dataset.py:18-20): src f32 [planes][H][W] -> out [planes][S][S]. */
int adn_resize_bilinear(const float* src, int32_t planes, int32_t H, int32_t W, int32_t S,
                        int32_t antialias, float* out, void* stream);
/* Gradient of the non-antialiased resize (the models' final F.interpolate(..., mode='bilinear', align_corners=False)
 * when output_size != input size: rgb_depth_model.py:200-206, binaural_attention_model.py:326-333):
 * gout [planes][S][S] -> gin [planes][H][W], a gather per source pixel (deterministic, no atomics). */
int adn_resize_bilinear_bwd(const float* gout, int32_t planes, int32_t H, int32_t W, int32_t S,
                            float* gin, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADN_H_ */
