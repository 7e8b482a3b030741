/* libdiagan_hip.so -- C ABI of the MI355X (gfx950) Dia-GAN hot path.
 *
 * The reference (grayhong/self-diagnosing-gan) has NO FFI on this path: its SNGAN/DCGAN train
 * step and LDR scorer are Python calling ATen/cuDNN ops and NumPy.  The only native boundary in
 * the reference is the StyleGAN2 pybind pair (diagan-pkg/diagan/models/op/fused_bias_act.cpp:4-20,
 * upfirdn2d.cpp:4-22), whose conventions this ABI keeps: launch on the caller's stream, no hidden
 * synchronisation, no global mutable state, errors surfaced to Python as RuntimeError.
 * Every entry point below names the reference Python op (file:line) it replaces.
 *
 * Conventions
 *   - plain C types only; all pointers are DEVICE pointers owned by the caller (torch allocates);
 *   - every function returns 0 or a negative DIAGAN_E* code; diagan_last_error() gives the text
 *     (thread local);  the python shim (diagan/_native/__init__.py) raises RuntimeError on != 0;
 *   - `stream` is a hipStream_t (0 = default stream); nothing synchronises;
 *   - activations are NHWC fp32 ("pixels x channels" row-major), channel counts padded to a
 *     multiple of 4; weights are consumed in the packed GEMM layouts produced by
 *     diagan_weight_prep (see DESIGN.md "Data layout in HBM").
 */
#ifndef DIAGAN_HIP_H
#define DIAGAN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DIAGAN_OK 0
#define DIAGAN_EINVAL (-1)
#define DIAGAN_EHIP (-2)
#define DIAGAN_EUNSUP (-3)

const char* diagan_last_error(void);
int diagan_abi_version(void);
const char* diagan_target_arch(void); /* "gfx950" */

/* ---- LDR scorer and logit record --------------------------------------------------------- */

/* calculate_scores, diagan-pkg/diagan/utils/plot.py:220-249 (float64, bit-exact with NumPy).
 * rec[T][row_stride] holds the window's snapshots in step order (plot.py:239).
 * ldr/ldrd/ldrv/ldrm: [N] outputs or NULL (plot.py:243-246).
 * conf[n_t][N]: clip_max_ratio(clip_min(mean + t*std, floor), ratio) for each t in t_vals
 * (plot.py:247-248; floor 1e-2, ratio 50).  workspace: >= 8*n_t bytes. */
int diagan_ldr_scores_f64(const double* rec, int T, int64_t N, int64_t row_stride, double* ldr,
                          double* ldrd, double* ldrv, double* ldrm, const double* t_vals, int n_t,
                          double* conf, double floor_val, double ratio, void* workspace,
                          void* stream);

/* Same scores in fp32 with wave-shuffle reductions over T (not bit-exact; |err| ~ 1e-6).
 * workspace: >= 4*n_t bytes. */
int diagan_ldr_scores_f32(const float* rec, int T, int64_t N, int64_t row_stride, float* ldr,
                          float* ldrd, float* ldrv, float* ldrm, const float* t_vals, int n_t,
                          float* conf, float floor_val, float ratio, void* workspace, void* stream);

/* logit_list[idx] = logit, diagan-pkg/diagan/trainer/trainer.py:154 (and the DP variant
 * stylegan2/train_ffhq.py:139-141).  rec_row: one [N] row of the record (f64 or f32).
 * Out-of-range indices are counted in *oob_counter (device int) and skipped. */
int diagan_logit_scatter(const float* logit, const int64_t* idx, int64_t n, void* rec_row,
                         int64_t N, int out_is_f64, int* oob_counter, void* stream);

/* ---- convolution (implicit GEMM on the fp32 matrix cores) --------------------------------- */

/* Gather geometry shared by the three conv entry points (DESIGN.md "Gather formula"):
 *   iy_num = oy*sy + r*dr + off; the tap is valid iff iy_num >= 0, iy_num % up == 0 and
 *   iy_num/up < Hi (same in x).  conv(stride s,pad p): (s,+1,-p,1); transposed conv and every
 *   data-gradient: (1,-1,+p,s).
 * Prologue modes (applied to gathered values): 0 none, 1 ReLU, 2 scale[c]*x+shift[c] then ReLU
 * (BatchNorm apply), 3 LeakyReLU(0.2), 4 affine only. */

/* Forward conv / transposed conv / data-gradient:
 *   y[b,oy,ox,n] = epi(out_scale * sum_{r,s,c} pro(x[b,iy,ix,c]) * w[n][(r*S+s)*Ci+c])
 *   epi: + bias[n], + residual (or max(residual,0) if res_relu & 1), then (mask_src > 0 ? v : mask_slope*v).
 *        res_relu & 2: `residual` is a HALF-resolution tensor [B,Ho/2,Wo/2,Co] and its bilinear x2 up-sampling
 *        (align_corners = false, the arithmetic of diagan_upsample2x) is added -- mimicry GBlock's up-sampled shortcut
 *        without the full-resolution tensor; Winograd kernel only (diagan_conv_gemm_pick_cfg_geom(...) == 9), no mask,
 *        no ReLU on it.
 * Replaces F.conv2d / nn.ConvTranspose2d forward and their input gradient
 * (SNGAN blocks: SURVEY §8 a2-a7; DCGAN: diagan-pkg/diagan/models/mnist.py:55-71,163-190).
 * x NHWC [B,Hi,Wi,Ci] (Ci % 4 == 0), w packed [Co][Kp], y NHWC [B,Ho,Wo,Co]. tile_cfg 0 = auto, 1 = 128x128, 3 / 7 = 64x64,
 * 5 = 256x64, 8 = 128x64, 14 = 64x64 with the K loop shared by two four-wave groups of ONE workgroup (launches of at most one
 * tile per CU; round 3), 9 / 11 / 12 / 13 = the Winograd kernels below.
 * scale0/scale1 (device scalars, optional): pixel rows m < scale_split are scaled by *scale0, the rest
 * by *scale1 instead of out_scale -- two forwards of one spectral-norm layer (different sigma) batched
 * into one GEMM on the un-normalised weight. */
int diagan_conv_gemm(const float* x, const float* w, float* y, const float* bias,
                     const float* residual, int res_relu, const float* mask_src, float mask_slope,
                     const float* pro_scale, const float* pro_shift, int pro_mode, float out_scale,
                     const float* scale0, const float* scale1, int scale_split,
                     int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                     int dr, int off, int up, int Kp, int tile_cfg, float* splitk_ws, int64_t splitk_ws_floats,
                     float* stat_partials, int pro_group_rows, void* stream);
/* pro_group_rows > 0: the rows (pixels) form M / pro_group_rows groups and group g reads its affine prologue from
 * pro_scale / pro_shift + g*Ci -- several independently batch-normalised batches (the n_dis generator forwards of one
 * global step) as ONE GEMM.  Must be a multiple of the tile's row count and divide M. */
/* stat_partials (optional, [ceil(M/BM)][2][Co] floats, BM = 128 for tile_cfg 1 else 64): the epilogue also
 * writes per-tile column sums of y and y^2 -- the BatchNorm statistics of the layer that consumes y
 * (diagan_bn_stats_fused), so the activation is not re-read; disables split-K. */
/* splitk_ws (optional scratch, splitk_ws_floats floats): lets small-output / long-K problems split the K
 * loop over several workgroups (deterministic two-stage reduction); diagan_conv_gemm_pick_ksplit tells
 * the factor that would be used (1 = none). */
int diagan_conv_gemm_pick_ksplit(int M, int Co, int Kp, int cfg);

/* 3x3 / stride 1 / pad 1 convolution (or its data-gradient) to FOUR output channels, Ci % 16 == 0, on the
 * 4x4x1 matrix instruction (N is exactly 4: no wasted columns) -- the generator's last conv (mimicry
 * SNGANGenerator32.c5 / Generator64.c6) and D's first-layer data-gradient.
 * Same prologue / bias / residual semantics as diagan_conv_gemm. */
int diagan_conv3x3_co4_supported(int Ci, int Co, int R, int S, int sy, int dr, int off, int up);
int diagan_conv3x3_co4(const float* x, const float* w, float* y, const float* bias, const float* residual,
                       const float* pro_scale, const float* pro_shift, int pro_mode, int B, int H, int W, int Ci,
                       int dr, int off, int Kp, int group_imgs, void* stream);
/* group_imgs > 0: image b reads the affine prologue of group b / group_imgs (see diagan_conv_gemm pro_group_rows). */

/* 3x3 / stride 1 / pad 1 convolution FROM four input channels (RGB + pad) to 64 or 128 channels: the first layer of the
 * discriminators (mimicry SNGANDiscriminator32 / 64 block1.c1; replaces that layer's cuDNN forward).  K = 36 leaves the
 * implicit GEMM's tile with nothing to overlap its LDS round trip and epilogue with; here a wave keeps the weights in registers,
 * streams 32 pixels at a time and writes 16-byte channel quads (no LDS, no barrier): bound by the output stream.
 * y = conv(x) * scale + bias, scale = *scale0 for pixel rows < scale_split and *scale1 behind it (both null: out_scale), as
 * diagan_conv_gemm.  DIAGAN_CONV_CI4=0 makes _supported answer 0 (A/B runs against the implicit GEMM). */
int diagan_conv3x3_ci4_supported(int Ci, int Co, int R, int S, int sy, int dr, int off, int up);
int diagan_conv3x3_ci4(const float* x, const float* w, float* y, const float* bias, float out_scale, const float* scale0,
                       const float* scale1, int scale_split, int B, int H, int W, int Co, int Kp, void* stream);

/* Weight (+ bias) gradient of the same layer, Ci in {64,128,256}, Kp == 9*Ci: the whole [4][Kp] gradient lives in
 * each wave's accumulators; writes diagan_conv3x3_co4_wgrad_splits(B, H) partial slabs
 * slab[split][slab_stride] in the packed-weight layout (bias column sums at bias_off if >= 0), to be summed by
 * diagan_wgrad_finish_batched / diagan_wgrad_reduce like the slabs of diagan_conv_wgrad. */
int diagan_conv3x3_co4_wgrad_supported(int Ci, int Co, int R, int S, int sy, int dr, int off, int up);
int diagan_conv3x3_co4_wgrad_splits(int B, int H);
int diagan_conv3x3_co4_wgrad(const float* dy, const float* x, float* slab, int64_t slab_stride, int64_t bias_off,
                             const float* pro_scale, const float* pro_shift, int pro_mode, int B, int H, int W,
                             int Ci, int Kp, void* stream);

/* tile config chosen when tile_cfg == 0 (host only); allow_split: split-K is available to the call (workspace given,
 * no stat_partials) */
int diagan_conv_gemm_pick_cfg(int M, int Co, int Kp, int allow_split);
/* Tile configurations: 1 = 128x128, 3 = 64x64, 5 = 256x64, 7 / 8 = 64x64 / 128x64 with double-buffered MFMA fragments
 * (2, 4, 6 -- other 128x64 wave layouts, 64-wide K-steps -- and 10 -- a staged-input Winograd kernel, 10 % slower than 9 -- were
 * retired: profiles/r02_wino_ablation.md); rows / columns of a
 * configuration (0 for an unknown one): */
int diagan_conv_gemm_tile_rows(int cfg);
int diagan_conv_gemm_tile_cols(int cfg);
/* tile_cfg 9: Winograd F(2x2,3x3) (csrc/conv_wino.hip) for 3x3 / stride 1 / pad 1 layers and their data-gradients --
 * same arguments, prologues and epilogues as the implicit GEMM, 16/36 of its multiply-accumulates; the weights are
 * transformed into splitk_ws (16 * roundup(Co, 64) * Ci floats) by a small kernel in front of the launch.  fp32
 * throughout; results differ from the implicit GEMM by rounding only (~1e-6 relative).  256 pixel rows x 64 columns
 * per workgroup (stat_partials / pro_group_rows granularity).  Returns 1 when the geometry qualifies. */
int diagan_conv_wino_supported(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off,
                               int up);
/* tile_cfg 11: Winograd F(2x2,3x3) FOLLOWED BY F.avg_pool2d(., 2) in one launch (csrc/conv_wino_pool.hip) -- the end of
 * mimicry's DBlock / DBlockOptimized with downsample=True (predefined_models.py:38-40,76-78).  A Winograd tile is one
 * pooling window and the window's sum needs only 9 of the 16 transform-domain products (c = (1,2,0,-1): frequency row /
 * column 2 drops out): 9/36 of the direct convolution's multiply-accumulates, and the full-resolution activation is never
 * written.  With tile_cfg 11, `y` and `residual` of diagan_conv_gemm are the POOLED tensors [B,Ho/2,Wo/2,Co]
 * (y = avg_pool(out_scale * conv + bias) + residual; Ho, Wo stay the convolution's output size); forward geometry only,
 * Co % 64 == 0, prologue 0 / 1, no mask / statistics / prologue groups; 64 pooled pixels x 128 columns per workgroup
 * (128 x 64 when Co is not a multiple of 128).
 * Never chosen by tile_cfg 0 (the output shape differs): this query says whether the launch qualifies and is worth it
 * (enough workgroups, with split-K over the slab behind the transformed weights where needed). */
int diagan_conv_wino_pool_supported(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                    int off, int up, int pro_mode, int64_t ws_floats);
/* tile_cfg 12: the data-gradient of such a layer straight from the pooled gradient, dx = conv^T(avg_pool2d_backward(g)) (the
 * backward of the same DBlocks): `x` of diagan_conv_gemm is the HALF-resolution gradient [B,Ho/2,Wo/2,Ci] (data-gradient
 * geometry: dr = -1, off = +1; Hi = Ho, Wi = Wo the full resolution), y / residual / mask_src full resolution.  The
 * up-sampled gradient is constant over each pooling window, so again only nine transform-domain products are non-zero;
 * the loader reads 9 instead of 16 pixels per tile.  Co % 64 == 0 (the layer's INPUT channels), no prologue. */
int diagan_conv_wino_unpool_supported(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                      int off, int up, int64_t ws_floats);
/* The configuration diagan_conv_gemm uses for tile_cfg == 0 on this geometry: 9 (Winograd) where the layer qualifies and
 * ws_floats holds the transformed weights, else diagan_conv_gemm_pick_cfg.  DIAGAN_WINO=0 in the environment turns
 * Winograd off. */
int diagan_conv_gemm_set_wino(int mode);   /* run-time form of DIAGAN_WINO: 0 off, 1 on, -1 environment / default (on) */
int diagan_conv_gemm_get_wino(void);
/* Winograd F(4x4,3x3) (tile_cfg 13, csrc/conv_wino4.hip; round 3): same reference ops and arguments as tile_cfg 9 for 3x3 /
 * stride 1 / pad 1 layers with H and W multiples of 4 -- 36 products per 4x4 output tile instead of 144 (F(2x2): 64), fp32,
 * error ~1e-5 of the output scale (cuDNN's non-fused Winograd for the reference's F.conv2d is the same F(4x4,3x3)).  The
 * automatic choice takes it for launches of >= 512 workgroups (32 tiles x 64 channels each).  0 = never, 1 / -1 = default. */
int diagan_conv_gemm_set_wino4(int mode);
/* Round 5 ("X3"): the F(4x4,3x3) launches (tile_cfg 13 / 15) whose K loop is a multiple of four 8-channel steps (Ci % 32 == 0)
 * run their 36 frequency GEMMs as v_mfma_f32_32x32x16_bf16 with every fp32 operand split EXACTLY into three bf16 pieces (six
 * piece products, fp32 accumulation: ~2^-23 relative per product, i.e. fp32-level; no scaling, fp32's exponent range): the same
 * reference ops, arguments and results to the tolerance of tile_cfg 13.  1 on, 0 off, -1 the environment's DIAGAN_WINO4_X3. */
int diagan_conv_gemm_set_wino4x(int mode);
int diagan_conv_gemm_get_wino4x(void);
/* Split-K launches of the Winograd kernels (few output tiles, long channel loops): the tile's LAST workgroup to deliver its partial
 * sums adds the slabs (in slab order) and runs the epilogue itself instead of a second launch (round 5).  0: always the second
 * launch; 1: in-kernel where the kernel has it (the F(2x2) kernel); -1: DIAGAN_SPLITK_FUSED, default OFF: measured neutral
 * (csrc/conv_gemm.hip, splitk_tickets). */
int diagan_conv_gemm_set_splitk_fused(int mode);
/* The ticket buffer of that in-kernel combine is CALLER-OWNED (round 6: the library allocates no device memory): `slots` zero-initialised
 * ints (one per 64 x 64 output tile of the largest split launch; every launch leaves them at zero), handed over per call through
 * diagan_conv_opts or -- diagnostics, process-wide -- registered here.  Without a buffer the second launch runs, as with the switch off. */
int diagan_conv_gemm_set_splitk_tickets(int* buf, int64_t slots);
/* Per-call selection options (round 6).  The kernel-selection switches above (diagan_conv_gemm_set_wino / _wino4 / _wino4x / _x3 / _x3b /
 * _splitk_fused, diagan_conv_gemm_tune) are PROCESS-GLOBAL and exist for diagnostics, A/B runs and the tests' like-with-like comparisons
 * only; a caller that wants another selection than the defaults passes it WITH the call: diagan_conv_gemm_next_opts stores the options
 * for the next diagan_conv_gemm call of the calling thread (thread-local; consumed by that call whatever path it takes, like the weights
 * hint and the output map), and two threads -- or two calls of one thread -- with different options never see each other's.
 * Every field: -1 = the process default (environment variable at first use, or the diagnostic setter), else the value the setter of the
 * same name takes; force_ksplit: 0 = the launch policy, > 1 = that many K splits where the configuration can split;
 * tune: ConvGemmArgs::tune bits, -1 = production default; tickets / ticket_slots: see diagan_conv_gemm_set_splitk_tickets. */
typedef struct diagan_conv_opts {
  int32_t wino, wino4, wino4x, gemm_x3, gemm_x3b, splitk_fused, force_ksplit, tune;
  int32_t* tickets;
  int64_t ticket_slots;
} diagan_conv_opts;
int diagan_conv_gemm_next_opts(const diagan_conv_opts* opts);
/* The tile configuration the last diagan_conv_gemm call of the calling thread resolved to (0: it failed before choosing). */
int diagan_conv_gemm_last_cfg(void);
/* The lone-tile implicit-GEMM launches (at most one 64 x 64 output tile per CU, long K loop: the 8x8 maps of SNGAN-32's discriminator;
 * tile_cfg 14) on the bf16 matrix pipe with every fp32 operand split EXACTLY into three bf16 pieces, six piece products accumulated
 * in fp32 (csrc/conv_gemm_x3.hip; fp32-grade results, held to float64 by tests/test_conv_gpu.py).  tile_cfg 16 asks for that kernel
 * by name (stride 1, Ci % 32 == 0, an even number of 32-channel K-steps, prologue none / ReLU); mode 1 / 0 / -1: automatic use
 * on / off / DIAGAN_GEMM_X3 (default on: 25.9 -> 23.2 us per launch at M = 8192, N = 128, K = 1152; SNGAN-32 +1.5 %). */
int diagan_conv_gemm_set_x3(int mode);
int diagan_conv_gemm_get_x3(void);
/* Round 6: the LARGE implicit-GEMM launches no Winograd kernel takes -- StyleGAN2's 3x3 / stride 2 convolutions, the 2x2 / 2x1 / 1x2 /
 * 1x1 parity classes of its stride-2 transposed convolutions and its 1x1 convolutions (reference: F.conv2d / F.conv_transpose2d in
 * diagan-pkg/diagan/models/stylegan2.py:224-265,553-614) -- on the bf16 matrix pipe with the same exact three-way operand split, 128 x 128
 * output tiles, two workgroups per CU (csrc/conv_gemm_x3b.hip; fp32-grade results, held to float64 by tests/test_conv_gpu.py).
 * tile_cfg 17 asks for that kernel by name (no up-sampling gather, Ci % 32 == 0, Kp == R*S*Ci, prologue none / ReLU / leaky ReLU,
 * epilogue out_scale + bias + residual only); mode 1 / 0 / -1: the automatic upgrade of an implicit-GEMM pick with >= 192 tiles of
 * 128 x 128, >= 4 K-steps and >= 4e9 multiply-accumulates on / off / DIAGAN_GEMM_X3B (default on). */
int diagan_conv_gemm_set_x3b(int mode);
int diagan_conv_gemm_get_x3b(void);
/* Pieces per operand of the LARGE split-operand kernels (tile_cfg 17 and the weight gradient of csrc/conv_wgrad_x3.hip): 3 (default) =
 * six piece products, fp32-grade results; 2 = OPT-IN: (a0 + a1)(b0 + b1) in two MFMAs per 8 channels, operands at ~2^-16, 1.5e-5-2e-5 of
 * the output scale per layer against float64 (the class of the F(4x4) Winograd layers) for 1.4-1.5x shorter launches; tile_cfg 17 then
 * runs its first form.  0: back to the environment's DIAGAN_X3_PIECES.  Process-level switch (diagnostics / an explicit user choice);
 * the lone-tile kernel (tile_cfg 16) always uses three. */
int diagan_conv_gemm_set_x3_pieces(int n);
int diagan_conv_gemm_get_x3_pieces(void);
/* tile_cfg 17 has two forms: 1 = 128 x 128 tiles, two workgroups per CU (small launches, short K loops); 2 = 256 x 128 tiles, four MFMA waves
 * + four loader waves in one workgroup per CU (the large launches).  Tests / diagnostics only, process-global: force one (0: automatic). */
int diagan_conv_gemm_x3b_force_form(int form);
/* Output map of the NEXT diagan_conv_gemm call of the calling thread (held for exactly one call, like the weights hint): pixel
 * (b, oy, ox) of the launch's Ho x Wo output grid is written to (b, mul*(oy-y0)+offy, mul*(ox-x0)+offx) of y[B][OH][OW][Co] (residual,
 * if given, is read there too); pixels outside [y0,y1) x [x0,x1) are dropped.  Written by tile_cfg 17 only (the call fails otherwise:
 * ask diagan_conv_gemm_final_cfg first).  Use: the four dense parity classes of a stride-2 transposed convolution interleave
 * themselves (mul = 2) instead of being copied into place.  mul = 0 clears a pending map. */
int diagan_conv_gemm_out_map(int mul, int offy, int offx, int y0, int y1, int x0, int x1, int OH, int OW);
/* The tile configuration diagan_conv_gemm(tile_cfg = 0) ends up with, including the upgrades to the split-operand kernels (16, 17)
 * that the pick_cfg queries do not model.  plain_epilogue: no backward mask, per-half scales, ReLU on the residual or half-resolution
 * residual; want_stats: BatchNorm statistics from the epilogue are requested. */
int diagan_conv_gemm_final_cfg(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off, int up,
                               int Kp, int allow_split, int64_t ws_floats, int pro_group_rows, int pro_mode, int plain_epilogue,
                               int want_stats);
/* tile_cfg 11 / 12 (convolution + 2x2 average pool, and its data gradient from the pooled gradient) run on the same F(4x4) kernel
 * in 25 products per 4x4 tile (frequency row / column 2 never reaches a pooling-window sum) where the launch has >= 192 workgroups
 * and H, W are multiples of 4; this query says whether a geometry does (kernel names, executed-FLOP accounting). */
int diagan_conv_wino4_pool_used(int B, int Ho, int Wo, int Ci, int Co, int64_t ws_floats);
/* tile_cfg 15 (round 4): y = conv3x3(F.interpolate(pro(x), scale_factor=2, mode='bilinear', align_corners=False)) + epilogue, the
 * start of mimicry's GBlock residual branch (BN -> ReLU -> up-sampling -> c1; GBlock._upsample_conv as selected at
 * diagan-pkg/diagan/models/predefined_models.py:19,57), as ONE launch of the F(4x4) kernel on the HALF-resolution input: `x` of
 * diagan_conv_gemm is [B,Hi/2,Wi/2,Ci] while Hi = Ho, Wi = Wo are the up-sampled sizes.  The interpolation is folded into the
 * Winograd input transform (16 instead of 36 pixels loaded per tile, prologue before the interpolation as in the reference);
 * forward geometry only, every prologue / epilogue of tile_cfg 13 except the backward mask and split-K.  This query says whether
 * a launch qualifies (same launch-size policy as tile_cfg 13 without channel splits; DIAGAN_WINO4_UPIN=0 turns it off). */
int diagan_conv_wino4_upin_supported(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                     int off, int up, int64_t ws_floats, int pro_group_rows);
/* Winograd weight transforms made ahead of the launches that use them (round 4).  Every Winograd launch of diagan_conv_gemm
 * (tile_cfg 9, 11, 12, 13, 15) first transforms its weights into `splitk_ws` with a small kernel of its own: 48 (SNGAN-32) to
 * 124 (SNGAN-64) launches of 5-7 us per training step.  A caller that knows which layers it is about to run can instead
 *   1. transform the weights of MANY layers in one launch into buffers it owns: diagan_wino_weights_batched takes a device
 *      table of n { const float* w; float* u; int Co, Ci, Kp, kind, flip, blk0; float scale; int pad; } records (kind 2:
 *      F(2x2,3x3) format, used by tile_cfg 9 and the F(2x2) pooled launches; 40: F(4x4,3x3), tile_cfg 13 and -- with scale
 *      1/16 -- 15; 41: the 25-frequency format of tile_cfg 11 / 12 on the F(4x4) kernel; flip = 1 for a data gradient;
 *      blk0 = prefix sum of diagan_wino_weight_blocks(Co, Ci) over the table, `blocks` its total);
 *   2. tell the next diagan_conv_gemm call of this thread that its transformed weights are at `u` in that format
 *      (diagan_conv_gemm_weights_hint); the call uses them and skips its own transform if -- and only if -- the format is the
 *      one it needs, so a wrong guess costs nothing but the unused hint.
 * diagan_conv_gemm_last_weight_format reports the format (and size in floats) the last call of this thread needed (kind 0: it ran
 * no Winograd kernel) and how many per-launch transforms this thread has issued so far: a caller learns the formats from an
 * ordinary first pass.  Reference ops replaced: none new -- this is launch-count bookkeeping around F.conv2d's Winograd form. */
int diagan_conv_gemm_weights_hint(const float* u, int kind, int flip, float scale);
int diagan_conv_gemm_last_weight_format(int* kind, int* flip, float* scale, int64_t* floats, int64_t* launches);
int64_t diagan_wino_weight_blocks(int Co, int Ci);
int diagan_wino_weights_batched(const void* jobs, int n, int blocks, void* stream);
int diagan_conv_gemm_pick_cfg_geom(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                   int off, int up, int Kp, int allow_split, int64_t ws_floats);
/* The same choice for a launch with a GROUPED prologue (pro_group_rows > 0: one affine row per group of that many GEMM
 * rows -- the stacked generator forward, BaseGenerator.prefetch_fakes): a tile must not straddle two groups, so a choice
 * whose tile height does not divide pro_group_rows falls back to diagan_conv_gemm_pick_cfg's and then to the 64-row
 * tile.  This is exactly what diagan_conv_gemm does for tile_cfg == 0; callers that size the statistics buffer or ask
 * whether a half-resolution residual will be fused use it to get the launch's own answer. */
int diagan_conv_gemm_pick_cfg_grouped(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                      int off, int up, int Kp, int allow_split, int64_t ws_floats, int pro_group_rows);
/* Diagnostics and tuning sweeps only (tools/stamp_report.py, tools/bench_conv.py; no reference counterpart, never
 * called by the product path).  While a stamp buffer is set, diagan_conv_gemm launches a diagnostic build of its
 * kernel (prologue modes 0 and 1) in which every workgroup records, at slot blockIdx.y*gridDim.x + blockIdx.x,
 * 8 x uint64: [0] the 100 MHz real-time counter at entry, [1] at exit, [2..6] shader cycles spent in loader
 * set-up / first tile / K loop / epilogue issue / store drain, [7] HW_ID | XCC_ID << 32.  buf = NULL ends it.
 * diagan_conv_gemm_tune: force_ksplit > 0 forces that split-K factor for every tile configuration; flags >= 0 overrides
 * the kernel's tuning bits (bit 0 / 1: raised wave priority during set-up / epilogue; -1 = production default);
 * lds_delta_bytes is added to the dynamic LDS request (occupancy probe; negative values: timing only). */
int diagan_conv_gemm_set_stamp_buffer(unsigned long long* buf, int64_t slots);
int diagan_conv_gemm_tune(int force_ksplit, int flags, int lds_delta_bytes);

/* Weight gradient, split over pixels: slab[s][n][k] = sum_{m in split s} dy[m][n]*pro(x gathered).
 * Replaces the weight half of conv2d / conv_transpose2d backward (errD.backward()/errG.backward()
 * in the train steps, diagan-pkg/diagan/models/topk_models.py:90, mnist.py:126).
 * dy NHWC [B,Ho,Wo,Co] (Co % 4 == 0), slab [splits][slab_stride] (slab_stride >= Co*Kp).
 * segments: the pixel range is cut into `segments` equal parts and no split straddles a part
 * (splits % segments == 0), so each batched forward's gradient can be reduced separately.
 * bias_off >= 0: the bias gradient partials sum_m dy[m][n] are written to slab[s][bias_off + n]. */
int diagan_conv_wgrad(const float* dy, const float* x, float* slab, int splits, int segments, int64_t slab_stride,
                      int64_t bias_off, const float* pro_scale, const float* pro_shift, int pro_mode, int B, int Hi, int Wi, int Ci, int Ho,
                      int Wo, int Co, int R, int S, int sy, int dr, int off, int up, int Kp,
                      void* stream);
int diagan_conv_wgrad_splits(int M, int Co, int Kp);
/* Round 5: the weight gradients of SEVERAL layers of one backward pass (layers of one diagan_conv_wgrad_batch_class, at most
 * diagan_conv_wgrad_batch_max() of them) in ONE launch: the layers share the chip's workgroups
 * in proportion to their work, so a small layer needs 4-8 slabs instead of the 64 its own chip-filling launch writes, and the
 * per-workgroup fixed cost is paid once per ~30 K-steps instead of once per 4.  Every job carries the arguments of a
 * diagan_conv_wgrad call (same slab layout: the deferred reduction does not change); `splits` is the caller's allocation for
 * that layer (a multiple of `segments`).  The caller keeps dy and x of every job alive until this call. */
typedef struct {
  const float* dy;
  const float* x;
  float* slab;
  const float* pro_scale;
  const float* pro_shift;
  int64_t slab_stride, bias_off;
  int32_t splits, segments, pro_mode, B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, pad_;
} diagan_wgrad_job;
int diagan_conv_wgrad_batched(const diagan_wgrad_job* jobs, int n, void* stream);
int diagan_conv_wgrad_batch_max(void);
/* The jobs of one diagan_conv_wgrad_batched call run the SAME kernel template: this is its identity for a geometry and prologue
 * mode (0: the layer cannot be batched and launches through diagan_conv_wgrad; Winograd layers 100 + mode, the implicit-GEMM
 * kernel 1000 for its 64 x 64 tile and 2000 + ... for the 128-column tiles with prologue none / ReLU).  Equal non-zero
 * classes may share a call. */
int diagan_conv_wgrad_batch_class(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off, int up,
                                  int Kp, int pro_mode);
/* The Winograd F(3x3,2x2) weight gradient (csrc/conv_wgrad_wino.hip: 16/36 of the multiply-accumulates, same slab layout
 * and deferred reduction) takes the 3x3 / stride 1 / pad 1 layers inside diagan_conv_wgrad; uses_wino tells whether a
 * geometry qualifies (DIAGAN_WINO / diagan_conv_gemm_set_wino / DIAGAN_WINO_WGRAD=0 turn it off), splits_geom the split
 * count a caller should allocate slabs for on this geometry. */
int diagan_conv_wgrad_uses_wino(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off,
                                int up, int Kp);
int diagan_conv_wgrad_splits_geom(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                  int off, int up, int Kp); /* host heuristic: number of splits */
/* Round 6: inside diagan_conv_wgrad the large plain launches of the 128 x 128 tile (no prologue, no bias column, whole tiles,
 * Ci % 32 == 0, >= 4e9 multiply-accumulates) run on the bf16 matrix pipe with both operands split exactly in three
 * (csrc/conv_wgrad_x3.hip; same splits, slabs and second-stage sum, fp32-grade results).  uses_x3 tells whether a launch
 * qualifies (for kernel-name bookkeeping); set_x3: 0 off, 1 on, 2 on without the work floor (tests), -1 the environment's
 * DIAGAN_WGRAD_X3 (default on) -- a process-level diagnostic switch.  diagan_conv_gemm_set_x3b(0) (the exact-fp32 mode) turns it off as well. */
int diagan_conv_wgrad_uses_x3(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off,
                              int up, int Kp, int pro_mode, int64_t bias_off);
int diagan_conv_wgrad_set_x3(int on);

/* Deferred epilogue of a whole backward pass, all layers in two launches: per layer
 * G = sum_s slab[s] (fixed order); plain layers: grad += G; spectral-norm layers:
 * grad += (G - <G,W>/sigma u^T v)/sigma on the weight part, grad += G on the bias part. */
typedef struct {
  float* slab[2];        /* per context: [splits][stride] */
  const float* u[2];     /* SN context of the forward each slab belongs to */
  const float* v[2];
  const float* state[2]; /* {sigma, 1/sigma} */
  double* partials[2];   /* ceil(n_elem/1024) doubles each (SN layers) */
  float* grad;           /* weight gradient [Co*Kp] followed by the bias gradient */
  const float* W;        /* master weight for SN layers, NULL otherwise */
  int64_t stride;
  int splits, n_elem, n_w, Kp;  /* splits per context */
  int nctx, first_block; /* nctx: 1, or 2 when the backward covered two batched forwards; first_block: index of the
                            layer's first workgroup in the finish kernels' 1-D grid (prefix sum of ceil(n_elem/1024)) */
} diagan_wgrad_layer;
int diagan_wgrad_finish_batched(const void* table_dev, int n_layers, int64_t total_blocks, int any_sn, void* stream);
/* elements of a layer that one workgroup of the finish kernels handles, for a layer of `splits` splits per context: the host
 * lays out diagan_wgrad_layer::first_block (and sizes the <G, W> partial arrays) in units of it (1024 x {8, 4, 2, 1} for at most
 * 2, 4, 8 and more splits).  diagan_wgrad_layer::stride must be below 2^25 floats (the kernels address 16 splits of a layer through
 * one 2 GiB buffer window). */
int diagan_wgrad_finish_block_elems(int splits);

/* out (+)= sum_s slab[s]; if w: dot_partials[block] = partial <sum, w> (fp64, for the SN backward);
 * ceil(n_elem/1024) partials. */
int diagan_wgrad_reduce(const float* slab, int splits, int64_t n_elem, float* out, int accumulate,
                        const float* w, double* dot_partials, void* stream);

/* ---- spectral norm (torch_mimicry SpectralNorm) ------------------------------- */

/* One power iteration + sigma on packed master weight W[Co][Kp]:
 *   v = normalize(u W), u' = normalize(W v), sigma = u' W v  (eps 1e-12).
 * u_out/v_out: [Co]/[Kp] copies kept for the backward; state[0]=sigma, state[1]=1/sigma;
 * u_buffer/sigma_buffer are the module buffers, overwritten iff update_buffers (training mode).
 * work: Kp + Co floats. */
int diagan_sn_power_iter(const float* W, float* u_buffer, float* sigma_buffer, float* u_out,
                         float* v_out, float* state, float* work, int Co, int Kp, float eps,
                         int update_buffers, void* stream);

/* All SN layers of one network in four launches.  table_dev: device array of n_layers descriptors. */
typedef struct {
  const float* W;    /* master weight [Co][Kp] */
  float* u_buf;      /* module buffer sn_u [Co] (updated iff update_buffers) */
  float* sigma_buf;  /* module buffer sn_sigma [1] */
  float* u_out;      /* [Co] u' of this forward */
  float* v_out;      /* [Kp] v of this forward */
  float* state;      /* {sigma, 1/sigma} */
  float* work;       /* 8*Kp + Co floats */
  float* Wf;         /* [Co][Kp] = W/sigma, or NULL */
  float* Wd;         /* [Ci][Kd] data-gradient operand / sigma, or NULL */
  int Co, Ci, RS, Kp, Kd, pad;
} diagan_sn_layer;
int diagan_sn_prepare_batched(const void* table_dev, int n_layers, int max_Co, int max_Ci, int max_RS,
                              int max_Kp, float eps, int update_buffers, int write_wd, void* stream);
/* write_wd: 1 pack Wf and Wd, 0 pack Wf only, -1 power iteration only.  diagan_pack_batched runs the
 * packing step alone (scale = state[1] of each descriptor). */
int diagan_pack_batched(const void* table_dev, int n_layers, int max_Co, int max_Ci, int max_RS, int write_wd,
                        void* stream);

/* GEMM operand packing: Wf = W*inv (same [Co][Kp] layout), Wd[ci][(rs)*Co+co] = W[co][(rs)*Ci+ci]*inv
 * (data-gradient operand, row length Kd).  inv_sigma: device float* or NULL (= 1). */
int diagan_pack_weights(const float* W, const float* inv_sigma, float* Wf, float* Wd, int Co, int Ci,
                        int RS, int Kp, int Kd, void* stream);
/* Round 6, StyleGAN2 weight preparation (reference: `self.weight * self.scale` in front of F.conv2d / F.conv_transpose2d,
 * diagan-pkg/diagan/models/stylegan2.py:94-129,224-265) in one launch each instead of a scalar multiply + permuted copy + pads (+ zero fill +
 * transposed pack in the backward) per layer and pass.  diagan_pack_oihw: w[Co_src][Ci_src][R][S] * scale -> Wf[Co][Kp] (k = (r S + s) Ci + c,
 * rows / channels / columns beyond the source zero) and / or the data-gradient operand Wd[Ci][Kd] (k = (r S + s) Co + n) of the same values.
 * diagan_unpack_oihw: its adjoint, gw[Co_src][Ci_src][R][S] = scale * gWp[n][(r S + s) Ci + c] (pack is linear: these two are each other's
 * backward to any order). */
int diagan_pack_oihw(const float* w, float scale, float* Wf, float* Wd, int Co_src, int Ci_src, int R, int S, int Co, int Ci, int Kp,
                     int Kd, void* stream);
int diagan_unpack_oihw(const float* gWp, float scale, float* gw, int Co_src, int Ci_src, int R, int S, int Ci, int Kp, void* stream);
/* The sub-kernels w[:, cy::2, cx::2] (taps in correlation order) of the four output-parity classes of a stride-2 transposed gather
 * (diagan/ops/diffconv.py: the generator's up-convolutions and the data gradients of the stride-2 layers run as four dense stride-1
 * convolutions), from the packed operand w[n][Kp] (k = (r S + s) C + c) into ONE buffer: class (cy, cx) = (cls >> 1, cls & 1) at float offset
 * off[cls] with rows of kp[cls] floats (zero-padded).  adjoint = 1: the reverse -- the full operand's gradient gw[n][Kp] from the classes'
 * weight gradients in `buf` (w unused).  R, S <= 4. */
int diagan_parity_weights(const float* w, float* buf, float* gw, int n, int R, int S, int C, int Kp, const int* kp, const int* off,
                          int adjoint, void* stream);


/* Backward through W/sigma: grad (+)= (G - <G,W>/sigma * u^T v) / sigma. */
int diagan_sn_grad_fix(const float* G, const double* dot_partials, int nparts, const float* u,
                       const float* v, const float* state, float* grad, int Co, int Kp,
                       int accumulate, void* stream);

/* ---- HBM-bound layers between the convolutions (NHWC fp32, C % 4 == 0) ------------------- */

/* NCHW [B,C,H,W] <-> NHWC [B,H,W,Cp] (zero padded channels): the dataloader / generate_images
 * boundary (reference tensors are NCHW: diagan-pkg/diagan/datasets/transform.py:9-10). */
int diagan_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream);
int diagan_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream);

/* torch.tanh at the generator output and its backward gx = g*(1-y^2). */
int diagan_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int diagan_tanh_bwd(const float* y, const float* g, float* gx, int64_t n, void* stream);

/* bytes of scratch for the column reductions below */
int64_t diagan_colred_workspace(int64_t M, int C);

/* nn.BatchNorm2d statistics over x[M][C] (training: batch stats + running-stat update with
 * unbiased variance; eval: running stats).  Emits mean/invstd and the affine form
 * scale = gamma*invstd, shift = beta - mean*scale consumed by the conv prologue. */
int diagan_bn_stats(const float* x, int64_t M, int C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, int training,
                    float* mean_out, float* invstd_out, float* scale_out, float* shift_out,
                    void* workspace, void* stream);
/* The same for `groups` batches stacked along the rows of x[groups * M][C] (the stacked generator forward), each with its OWN
 * batch statistics, the running statistics taking the momentum updates in group order: one reduction launch and one
 * finalisation for all groups.  Training mode only; outputs [groups][C]; workspace: groups * diagan_colred_workspace(M, C). */
int diagan_bn_stats_grouped(const float* x, int64_t M, int C, int groups, const float* gamma, const float* beta, float eps,
                            float momentum, float* running_mean, float* running_var, float* mean_out, float* invstd_out,
                            float* scale_out, float* shift_out, void* workspace, void* stream);

/* BatchNorm (training mode) from the per-tile sums written by diagan_conv_gemm(stat_partials).  groups > 1: `groups`
 * independently normalised batches of M rows / `tiles` tiles each, contiguous in `partials`; finalised in order in one
 * launch (the running statistics take `groups` momentum updates); the four outputs are [groups][C].
 * workspace (optional, workspace_doubles doubles): long tile lists are first summed by
 * diagan_bn_stats_fused_splits(tiles, C, groups) workgroups per 16 channels and group (fixed order: deterministic);
 * needs groups * splits * 2 * C doubles, otherwise the single-stage kernel runs. */
int diagan_bn_stats_fused(const float* partials, int tiles, int64_t M, int C, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, float* mean_out,
                          float* invstd_out, float* scale_out, float* shift_out, int groups, double* workspace,
                          int64_t workspace_doubles, void* stream);
int diagan_bn_stats_fused_splits(int tiles, int C, int groups);

/* Backward of [BatchNorm -> optional (Leaky)ReLU -> optional dropout]: dx (+ residual), dgamma/dbeta (+)=.
 * relu != 0: g' = g * drop * drop_scale * (y > 0 ? 1 : slope), y = scale*x+shift (slope 0 = ReLU).  coef: 2*C floats.
 * drop_scale (round 5): 1 / (1 - p) for a 0 / 1 keep-mask as torch's bernoulli_ writes it (no separate scaling pass over
 * the mask), 1 for a mask that carries the scale.
 * batch_stats = 0: eval-mode BatchNorm (running statistics): dx = scale * g'. */
int diagan_bn_bwd(const float* g, const float* x, int64_t M, int C, const float* scale, const float* shift,
                  const float* mean, const float* invstd, int batch_stats, int relu, float slope,
                  const float* drop, float drop_scale, float* dgamma, float* dbeta,
                  int accumulate_param_grads, const float* residual, float* dx, float* coef,
                  void* workspace, void* stream);

/* out[c] (+)= sum_m x[m][c]  (bias gradients). */
int diagan_colsum(const float* x, int64_t M, int C, float* out, int accumulate, void* workspace, void* stream);

/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) with the conv prologue
 * modes applied to the source pixels, and its adjoint (+ residual). x [B,H,W,C] -> [B,2H,2W,C]. */
int diagan_upsample2x(const float* x, float* out, int B, int H, int W, int C, int pro_mode,
                      const float* scale, const float* shift, int group_imgs, void* stream);
int diagan_upsample2x_bwd(const float* g, float* out, int B, int H, int W, int C, const float* residual,
                          void* stream);

/* F.avg_pool2d(x, 2) (+ residual at the pooled resolution) and its adjoint (+ residual). H, W = input size. */
/* out = avgpool2x2(relu_in ? max(x,0) : x) + residual */
int diagan_avgpool2(const float* x, float* out, int B, int H, int W, int C, const float* residual, int relu_in,
                    void* stream);
/* out[B][H+1][W+1][C] = 0.25 * (2x2 box sums of act(x), zero outside the image; act = ReLU if relu_in): the weight gradient
 * of avg_pool2d(conv3x3(act(x)), 2) -- mimicry's down-sampling DBlocks -- equals the weight gradient of a 3x3 / stride 2 /
 * pad 0 convolution over this image against the POOLED output gradient (diagan_conv_wgrad with sy = 2, off = 0): 9 instead
 * of 36 multiply-accumulates per pooled pixel, weight and tap, with no transform at all. */
int diagan_boxsum2(const float* x, float* out, int B, int H, int W, int C, int relu_in, void* stream);
int diagan_avgpool2_bwd(const float* g, float* out, int B, int H, int W, int C, const float* residual,
                        void* stream);

/* SNGAN discriminator head: pooled = sum_hw relu(x); logit = inv_sigma * pooled.w + bias
 * (torch.sum(activation(h), dim=(2,3)) -> SNLinear(C, 1)). */
int diagan_head_fwd(const float* x, const float* w, const float* inv_sigma, const float* inv_sigma1, int split_b,
                    const float* bias, float* pooled, float* logit, int B, int HW, int C, void* stream);
/* inv_sigma1 (optional): samples b >= split_b use *inv_sigma1 (second batched forward).
 * gx = dlogit*w*inv_sigma*(x>0) (if gx); G[c] = sum_b dlogit*pooled, dot = <G,w>, dbias (if G). */
int diagan_head_bwd(const float* dlogit, const float* w, const float* inv_sigma, const float* inv_sigma1, int split_b,
                    const float* x, const float* pooled, float* gx, float* G, double* dot, float* dbias,
                    int accumulate_bias, int B, int HW, int C, void* stream);

/* DCGAN discriminator activations (diagan-pkg/diagan/models/mnist.py:163-190): out = act(x*scale+shift)*drop*drop_scale,
 * act(v) = v > 0 ? v : slope*v (LeakyReLU 0.2), drop = dropout keep-mask or NULL, drop_scale as for diagan_bn_bwd; and the
 * backward of the un-normalised first layer. */
int diagan_act_fwd(const float* x, const float* scale, const float* shift, float slope, const float* drop, float drop_scale,
                   float* out, int64_t M, int C, void* stream);
int diagan_act_bwd(const float* g, const float* x, float slope, const float* drop, float drop_scale, float* out, int64_t n,
                   void* stream);
/* nn.Linear(C, 1) (mnist.py:191 out_d): logit = x.w + bias; dw += dlogit^T x, dbias += sum dlogit;
 * input gradient gx[b][j] = dlogit[b]*w[j]. */
int diagan_linear1_fwd(const float* x, const float* w, const float* bias, float* logit, int B, int C, void* stream);
int diagan_linear1_wgrad(const float* dlogit, const float* x, float* dw, float* dbias, int B, int C, void* stream);
int diagan_linear1_bwd_input(const float* dlogit, const float* w, float* gx, int B, int C, void* stream);

int diagan_add(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ---- loss heads and optimiser ---------------------------------------------------------------- */

/* loss types: 0 'gan', 1 'ns', 2 'hinge', 3 'wasserstein' (torch_mimicry.modules.losses).
 * Discriminator: out3 = {errD, D(x), D(G(z))}; d_real/d_fake = dL/dlogit.  gold: GOLD re-weighting
 * of the fake term (diagan-pkg/diagan/models/gold_reweight_models.py:10-61). */
int diagan_loss_dis(const float* out_real, int n_real, const float* out_fake, int n_fake, int loss_type,
                    int gold, float* d_real, float* d_fake, float* out3, void* stream);
/* Generator loss over the k largest logits (TopKGenerator.get_topk, topk_models.py:31-38; k = n: all). */
int diagan_loss_gen(const float* out_fake, int n, int k, int loss_type, float* d_fake, float* out1, void* stream);

/* torch.optim.Adam.step on one flat buffer (predefined_models.py:32,51,70,89,114,123).  grad_scale multiplies the
 * gradient as it is read: 1/W after a data-parallel SUM all-reduce (the mean of DistributedDataParallel,
 * stylegan2/train_ffhq.py:572-585, without a separate pass over the slab); 1 otherwise. */
int diagan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                     float beta2, float eps, float bias_correction1, float bias_correction2_sqrt, float grad_scale,
                     void* stream);
/* The same step with {lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt, grad_scale, 0} read from a DEVICE
 * row of eight floats: the launch can be captured in a hipGraph and replayed with values the host writes before each
 * replay. */
int diagan_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper8, void* stream);

/* ---- data-parallel exchange (RCCL over xGMI; csrc/comm.hip) ------------------------------------------------------
 * An explicit context per process (one process per GPU) holding the RCCL communicator; RCCL itself is dlopen-ed at
 * context creation (no link-time dependency, single-process runs never load it).
 *   diagan_comm_unique_id: 128 bytes made by rank 0 and handed to every rank by the launcher (any host channel);
 *   diagan_ctx_create: collective over all `world` ranks (ncclCommInitRank) on HIP device `device`;
 *   diagan_allreduce_grads: in-place SUM of a network's flat fp32 gradient slab on `stream` -- the gradient averaging of
 *     DistributedDataParallel (stylegan2/train_ffhq.py:572-585); the 1/W is applied by diagan_adam_step(grad_scale);
 *   diagan_allgather_logits: rank-major concatenation of equally sized per-rank rows (elem_bytes 4 / 8: fp32 logits /
 *     float64 record shards; copied, never summed) -- concat_all_gather of stylegan2/train_ffhq.py:150-161.
 * Both collectives are stream-ordered launches: no host synchronisation, capturable in a hipGraph. */
typedef struct diagan_ctx diagan_ctx;
int diagan_comm_unique_id(void* id128);
int diagan_ctx_create(diagan_ctx** out, const void* id128, int rank, int world, int device);
int diagan_ctx_destroy(diagan_ctx* ctx);
int diagan_ctx_rank(const diagan_ctx* ctx);
int diagan_ctx_world(const diagan_ctx* ctx);
int diagan_allreduce_grads(diagan_ctx* ctx, float* slab, int64_t n, void* stream);
int diagan_allgather_logits(diagan_ctx* ctx, const void* send, void* recv, int64_t n_per_rank, int elem_bytes,
                            void* stream);

/* ---- StyleGAN2 native ops (the reference's only native code) -------------- */

/* fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale), fused_bias_act.cpp:4-20:
 * out[i] = act(x[i] + bias[(i / step_b) % size_b]) * scale; act 1 linear, 3 leaky ReLU; grad 1 gates by
 * refer[i] > 0 (first derivative), grad 2 gives zeros.  bias / refer may be NULL. */
int diagan_fused_bias_act(const float* x, const float* bias, const float* refer, float* out, int64_t n,
                          int64_t step_b, int size_b, int act, int grad, float alpha, float scale, void* stream);

/* The tail of StyledConv.forward (diagan-pkg/diagan/models/stylegan2.py:323-329) in one pass over a channels-last
 * activation x[B][P][C]:  out = leaky_relu(x * demod[b][c] + strength[0] * noise[b or 0][p] + bias[c], alpha) * scale.
 * demod (the demodulation factor of :231-233 applied activation-side), noise and bias may each be NULL;
 * noise_per_image: noise is [B][P] (1) or one [P] map shared by the batch (0).  C % 4 == 0. */
int diagan_styled_bias_act(const float* x, const float* demod, const float* noise, const float* strength,
                           const float* bias, float* out, int B, int P, int C, int noise_per_image, float alpha,
                           float scale, void* stream);

/* out[b][c] = sum_p a[b][p][c] * b[b][p][c] over channels-last [B][P][C] tensors: the gradient of a per-(image, channel)
 * scale -- `style` in `weight = scale * W * style` (stylegan2.py:227-228) and `demod` (:231-233) when the modulated
 * convolution is evaluated activation-side.  C a power of two in [4, 1024]; workspace: B * diagan_rowdot_chunks(B, P)
 * * C floats.  Deterministic (fixed-order partial sums, combined in double). */
int diagan_rowdot_chunks(int B, int P);
int diagan_rowdot(const float* a, const float* b, float* out, float* workspace, int B, int P, int C, void* stream);
/* First-order backward of diagan_styled_bias_act -- and of bias + leaky ReLU (diagan_fused_bias_act act 3) when x, demod and
 * noise are NULL -- in ONE pass over the incoming gradient gy[B][P][C] (round 4; the reference's FusedLeakyReLU backward,
 * op/fused_act.py:21-60, plus the NoiseInjection / demodulation gradients of stylegan2.py:268-329):
 *   gpre = gy * scale * (y > 0 ? 1 : alpha);  gx = gpre * demod[b][c] (or gpre);
 *   work_d[blk][c] = sum_p gpre * x, work_b[blk][c] = sum_p gpre, work_s[blk] = sum_p noise[p] * sum_c gpre
 * for the B * diagan_rowdot_chunks(B, P) blocks of pixels (rows blk = b * chunks + k): the caller adds the blocks' rows -- d(demod)[b][c]
 * over a sample's chunks, d(bias)[c] over all blocks, d(strength) over all blocks.  gx may be NULL; x / work_d and noise / work_s are
 * given together or not at all.  C a power of two in [4, 1024].  Deterministic (fixed-order sums).  The higher-order
 * backward (R1, path-length penalty) keeps the differentiable composition of diagan_fused_bias_act / diagan_rowdot. */
int diagan_styled_bias_act_bwd(const float* gy, const float* y, const float* x, const float* demod, const float* noise,
                               float* gx, float* work_d, float* work_b, float* work_s, int B, int P, int C,
                               int noise_per_image, float alpha, float scale, void* stream);
/* ... and the reduction of its partial sums (work_d [B][chunks][C], work_b [B * chunks][C], work_s [B * chunks]; chunks =
 * diagan_rowdot_chunks(B, P)) to gd [B][C], gb [C], gs [1] in one launch, accumulated in double in a fixed order (round 6; any of the three
 * pairs may be NULL). */
int diagan_styled_bias_act_bwd_finish(const float* work_d, const float* work_b, const float* work_s, float* gd, float* gb, float* gs, int B,
                                      int P, int C, void* stream);

/* Round 6: the small dense pieces of the modulated convolution as single launches (csrc/stylegan_dense.hip; reference stylegan2.py:132-166
 * EqualLinear, :236-238 the demodulation), each with its first-order backward:
 *   diagan_small_linear_fwd  out[b][c] = scale * sum_k W[c][k] * x[b][k] + bias[c] * bias_mul   (x [B][K], W [C][K], K % 4 == 0; bias may be NULL)
 *   diagan_small_linear_bwd  gW [C][K], gbias [C] (may be NULL), gx [B][K] (may be NULL) from g [B][C]
 *   diagan_demod_fwd         wsq[co][ci] = sum_taps w[co][ci][tap]^2;  d[b][co] = rsqrt(scale2 * sum_ci s[b][ci]^2 * wsq[co][ci] + eps)
 *   diagan_demod_bwd         gw (shape of w) and gs [B][Ci] (may be NULL) from gd [B][Co]; B <= 64 */
int diagan_small_linear_fwd(const float* x, const float* W, const float* bias, float* out, int B, int K, int C, float scale, float bias_mul,
                            void* stream);
int diagan_small_linear_bwd(const float* g, const float* x, const float* W, float* gW, float* gbias, float* gx, int B, int K, int C,
                            float scale, float bias_mul, void* stream);
int diagan_demod_fwd(const float* s, const float* w, float* d, float* wsq, int B, int Ci, int Co, int taps, float scale2, float eps,
                     void* stream);
int diagan_demod_bwd(const float* gd, const float* d, const float* s, const float* w, const float* wsq, float* gw, float* gs, int B, int Ci,
                     int Co, int taps, float scale2, void* stream);

/* Round 6: the StyledConv tail that also leaves the NEXT layer's modulated input (out_mod = out * post[b][c]: that layer's
 * scale_rows) in the same pass, and its first-order backward: the incoming gradient is gy (may be NULL: out has no other consumer)
 * + gmod * post; everything diagan_styled_bias_act_bwd computes from it, plus work_p[blk][c] = sum_p gmod * y -> d(post) (finish:
 * diagan_styled_bias_act_bwd_finish with work_p in the place of work_d). */
int diagan_styled_bias_act_mod(const float* x, const float* demod, const float* noise, const float* strength, const float* bias,
                               const float* post, float* out, float* out_mod, int B, int P, int C, int noise_per_image, float alpha,
                               float scale, void* stream);
int diagan_styled_bias_act_mod_bwd(const float* gy, const float* gmod, const float* post, const float* y, const float* x,
                                   const float* demod, const float* noise, float* gx, float* work_d, float* work_b, float* work_s,
                                   float* work_p, int B, int P, int C, int noise_per_image, float alpha, float scale, void* stream);

/* Round 6: activation passes folded into the pass next to them (reference: diagan-pkg/diagan/models/stylegan2.py:553-614, the
 * discriminator's ConvLayer / ResBlock; :268-329 the generator's StyledConv).  Each is bit-identical to the two launches it replaces.
 *   diagan_bias_act_fir       out = FIR(leaky_relu(x + bias[c]) * scale) on x[major][in_h][in_w][minor], zero padding of the ACTIVATED
 *                             tensor (ConvLayer's FusedLeakyReLU followed by the Blur of the next, sub-sampling ConvLayer); 4-tap-wide
 *                             filters, minor % 4 == 0; out is [major][in_h + pad_y0 + pad_y1 - kh + 1][in_w + pad_x0 + pad_x1 - kw + 1][minor]
 *   diagan_bias_act_gate_bwd  its (and diagan_bias_act_add's) first-order backward gate from the PRE-activation z and the bias:
 *                             gx = gy * scale * (z + bias[c] > 0 ? 1 : alpha), work_b as diagan_styled_bias_act_bwd (finish: ..._bwd_finish)
 *   diagan_bias_act_add       out = leaky_relu(x + bias[c]) * scale + addend, channels-last, n elements (ResBlock: activation of conv2 + skip)
 *   diagan_fir_styled_act     out = [post[b][c] *] leaky_relu(FIR(x) * demod[b][c] + strength * noise[b or 0][p] + bias[c]) * scale: the
 *                             StyledConv tail (diagan_styled_bias_act) and optionally the next layer's style applied to the blurred output
 *                             of an up-sampling convolution on its way out (demod / noise / bias / post optional; used when no graph is
 *                             recorded: the backward needs the blurred tensor itself) */
/* Round 6: the generator's ToRGB (reference stylegan2.py:332-351: a modulated 1x1 convolution to 3 planes, no demodulation) in ONE read
 * of its full-resolution input instead of three passes, its first-order backward in one read + one write instead of six:
 *   diagan_torgb_fwd   out[b][p][0..2] = sum_c w[o][c] * (x[b][p][c] * s[b][c]) + bias[o], out[b][p][3] = 0;  x [B][P][C] channels-last,
 *                      s [B][C], w [3][C] (the scaled weight), bias [3] or NULL, out [B][P][4]; C a power of two in [4, 1024]
 *   diagan_torgb_bwd   gx = (sum_o gy[..][o] * w[o][c]) * s[b][c]  (gx may be NULL), gs [B][C] = d(s), gw [3][C] = d(w); work: B *
 *                      (diagan_rowdot_chunks(B, P) + 1) * 3 * C floats of scratch.  Deterministic (fixed-order sums, accumulated in double).
 * The higher-order backward keeps the composition scale_rows + diagan_conv_gemm / diagan_conv_wgrad. */
int diagan_torgb_fwd(const float* x, const float* s, const float* w, const float* bias, float* out, int B, int P, int C, void* stream);
int diagan_torgb_bwd(const float* gy, const float* x, const float* s, const float* w, float* gx, float* work, float* gs, float* gw, int B,
                     int P, int C, void* stream);
/* Round 6: the discriminator's first ConvLayer (reference stylegan2.py:553-595: EqualConv2d 3 -> C, 1x1, + FusedLeakyReLU) as ONE write
 * of its output, and its first-order backward as one read of gy and y:
 *   diagan_fromrgb_fwd   y[b][p][c] = leaky_relu(wscale * sum_{i<3} w[c][i] * x[b][p][i] + bias[c]) * scale;  x [B][P][4] (RGB + a zero
 *                        plane), w [C][3] (the raw parameter), bias [C] or NULL; C a power of two in [4, 1024]
 *   diagan_fromrgb_bwd   gz = gy * scale * (y > 0 ? 1 : alpha); work[blk][i][c] = sum_p gz * x[..][i] (i < 3) and sum_p gz (i = 3) for the
 *                        B * diagan_rowdot_chunks(B, P) blocks (finish: diagan_styled_bias_act_bwd_finish over 4 * C columns);
 *                        gx[b][p][0..2] = wscale * sum_c gz * w[c][i] when gx is not NULL; C <= 256 */
int diagan_fromrgb_fwd(const float* x, const float* w, const float* bias, float* y, int B, int P, int C, float wscale, float alpha,
                       float scale, void* stream);
int diagan_fromrgb_bwd(const float* gy, const float* y, const float* x, const float* w, float* gx, float* work, int B, int P, int C,
                       float wscale, float alpha, float scale, void* stream);
int diagan_bias_act_fir(const float* input, const float* bias, const float* kernel, float* out, int major, int in_h, int in_w, int minor,
                        int kernel_h, int kernel_w, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float alpha, float scale,
                        void* stream);
int diagan_bias_act_gate_bwd(const float* gy, const float* z, const float* bias, float* gx, float* work_b, int B, int P, int C,
                             float alpha, float scale, void* stream);
/* ... and the whole first-order backward of diagan_bias_act_fir in ONE pass: gz = FIR'(g) * scale * (ref + bias[c] > 0 ? 1 : alpha) with
 * FIR' the ADJOINT filter (the caller hands in flipped taps and the adjoint pads; output size = size of ref), work_b [major *
 * diagan_rowdot_chunks(major, out_h * out_w)][minor] the bias gradient's partial sums (finish: diagan_styled_bias_act_bwd_finish). */
int diagan_fir_gate_bwd(const float* g, const float* kernel, const float* ref, const float* bias, float* gz, float* work_b, int major,
                        int in_h, int in_w, int minor, int kernel_h, int kernel_w, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                        float alpha, float scale, void* stream);
/* out = upfirdn2d(input, kernel, up, down, pads) + addend (addend and out [major][out_h][out_w][minor], minor % 4 == 0) in one pass: the
 * gradient of a tensor that feeds BOTH a convolution and a resampling filter (ResBlock's input: conv1 and the skip branch's
 * blur + sub-sampling) -- the filter's adjoint adds the other branch's gradient on its way out instead of a separate accumulation pass. */
int diagan_upfirdn2d_add(const float* input, const float* kernel, const float* addend, float* out, int major, int in_h, int in_w, int minor,
                         int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                         int pad_y1, void* stream);
int diagan_bias_act_add(const float* x, const float* bias, const float* addend, float* out, int64_t n, int C, float alpha, float scale,
                        void* stream);
int diagan_fir_styled_act(const float* input, const float* kernel, float* out, int major, int in_h, int in_w, int minor, int kernel_h,
                          int kernel_w, int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* demod, const float* noise,
                          const float* strength, const float* bias, const float* post, int noise_per_image, float alpha, float scale,
                          void* stream);


/* upfirdn2d.upfirdn2d(input[major,H,W,minor], kernel[kh,kw], up, down, pads), upfirdn2d.cpp:4-22.
 * out == NULL: size query only (writes *out_h, *out_w). */
int diagan_upfirdn2d(const float* input, const float* kernel, float* out, int major, int in_h, int in_w, int minor,
                     int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                     int pad_y0, int pad_y1, int* out_h, int* out_w, void* stream);

/* ---- precision / recall in feature space (diagan-pkg/diagan/trainer/compute_pr.py:11-124) --------------------
 * The pairwise matrix T[r][c] = |b_c|^2 - 2 a_r.b_c is produced by diagan_conv_gemm (1x1 geometry, out_scale -2,
 * bias = squared norms of b); the consumers add the row term |a_r|^2 (row_add) on the fly.
 * row_sqnorm: out[r] = sum_c x[r][c]^2 (torch.sum(torch.square(x), dim=1), compute_pr.py:26-27).
 * kth_smallest_rows: out[r] = k-th smallest of T[r][:] (+ row_add[r]), k in [1,16] (get_kth_value, :34-50).
 * any_lt_rows / any_lt_cols: out = 1.0 where ANY element of the row / column satisfies
 *   T[r][c] + row_add[r] < thr, thr = thr_col[c] or thr_row[r] (exactly one given)   ((dist < radii).any(axis), :85-93). */
int diagan_row_sqnorm(const float* x, float* out, int N, int D, int ld, void* stream);
int diagan_kth_smallest_rows(const float* T, const float* row_add, int rows, int cols, int ld, int k, float* out,
                             void* stream);
int diagan_any_lt_rows(const float* T, const float* row_add, const float* thr_col, const float* thr_row, int rows,
                       int cols, int ld, float* out, void* stream);
int diagan_any_lt_cols(const float* T, const float* row_add, const float* thr_col, const float* thr_row, int rows,
                       int cols, int ld, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
