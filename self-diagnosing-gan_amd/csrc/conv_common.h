// Shared definitions of the implicit-GEMM convolution kernels (fwd/dgrad in conv_gemm.hip,
// wgrad in conv_wgrad.hip).
//
// Tensor layout: activations NHWC fp32 (C % 4 == 0), packed weights Wp[Co][Kp] with
// k = (r*S + s)*Ci + c, Kp = roundup(R*S*Ci, 32), zero padded.
//
// Gather geometry (one formula for conv, strided conv, transposed conv and every dgrad):
//   iy_num = oy*sy + r*dr + off ;  valid iff iy_num >= 0, iy_num % up == 0, iy_num/up < Hi
//     forward conv (stride s, pad p)         : sy = s, dr = +1, off = -p, up = 1
//     transposed conv / dgrad (stride s, p)  : sy = 1, dr = -1, off = +p, up = s
#pragma once
#include "common.h"

namespace diagan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum ProMode : int {
  PRO_NONE = 0,
  PRO_RELU = 1,          // a = max(x, 0)
  PRO_AFFINE_RELU = 2,   // a = max(x*scale[c] + shift[c], 0)   (BatchNorm apply + ReLU)
  PRO_LRELU = 3,         // a = x > 0 ? x : 0.2 x
  PRO_AFFINE = 4,        // a = x*scale[c] + shift[c]
  // weight gradient only (conv_wgrad.hip): the gathered tensor is the (H+1) x (W+1) image of 0.25 * 2x2 box sums of x / of
  // max(x, 0) (what diagan_boxsum2 writes), summed by the LOADER from the H x W tensor itself
  PRO_BOX = 5,
  PRO_BOX_RELU = 6,
};

struct ConvGeom {
  int B, Hi, Wi, Ci;   // gathered tensor (conv input for fwd / wgrad, dy for dgrad)
  int Ho, Wo, Co;      // pixel-indexed tensor (conv output for fwd, dx for dgrad, dy for wgrad)
  int R, S;
  int sy, dr, off, up; // gather formula above (same for x and y directions)
  int K;               // R*S*Ci
  int Kp;              // padded K (multiple of 32): row length of packed weights
};

__device__ __forceinline__ f32x4 apply_pro(f32x4 v, int mode, const float* __restrict__ scale,
                                           const float* __restrict__ shift, int c) {
  if (mode == PRO_NONE) return v;
  if (mode == PRO_AFFINE_RELU || mode == PRO_AFFINE) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
    v = v * sc + sh;
    if (mode == PRO_AFFINE) return v;
  }
  if (mode == PRO_LRELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
    return v;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  return v;
}

struct FastDiv {  // unsigned division by a runtime constant: q = (n * mul) >> 32 >> shift  (n < 2^31)
  unsigned mul, shift, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  unsigned l = 0;
  while ((1u << l) < d) ++l;                                   // ceil(log2 d)
  const unsigned long long m = ((1ull << (32 + l)) + d - 1) / d;  // needs 33 bits in general
  f.mul = (unsigned)(m - (1ull << 32));
  f.shift = l;
  return f;
}
// bilinear x2 (align_corners = false) of a half-resolution NHWC tensor r[B][H][W][C] at output pixel (oy, ox), channels
// c .. c+3: same taps, weights and grouping of the sums as upsample2x_kernel (elementwise.hip)
__device__ __forceinline__ f32x4 residual_up2(const float* __restrict__ r, int b, int oy, int ox, int H, int W, int C, int c) {
  const int jy = oy >> 1, jx = ox >> 1;
  const int y0 = (oy & 1) ? jy : max(jy - 1, 0), y1 = (oy & 1) ? min(jy + 1, H - 1) : jy;
  const int x0 = (ox & 1) ? jx : max(jx - 1, 0), x1 = (ox & 1) ? min(jx + 1, W - 1) : jx;
  const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
  const float wy1 = 1.f - wy0, wx1 = 1.f - wx0;
  const float* base = r + (long)b * H * W * C + c;
  const f32x4 a00 = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x0) * C);
  const f32x4 a01 = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x1) * C);
  const f32x4 a10 = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x0) * C);
  const f32x4 a11 = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x1) * C);
  const f32x4 top = wx0 * a00 + wx1 * a01;
  const f32x4 bot = wx0 * a10 + wx1 * a11;
  return wy0 * top + wy1 * bot;
}

__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  if (f.d == 1) return n;
  const unsigned t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

// Output map of a launch that writes into a LARGER tensor (conv_gemm_x3b.hip only; mul == 0: off): pixel (b, oy, ox) of the launch's
// own Ho x Wo grid goes to (b, mul * (oy - y0) + offy, mul * (ox - x0) + offx) of y[B][OH][OW][Co]; pixels outside [y0, y1) x [x0, x1)
// are computed and dropped.  The parity classes of a stride-2 transposed gather use it (mul = 2).
struct OutMap {
  int mul, offy, offx, y0, y1, x0, x1, OH, OW;
};

// Arguments of the forward / data-gradient kernels (conv_gemm.hip: implicit GEMM; conv_wino.hip: Winograd F(2x2,3x3))
struct ConvGemmArgs {
  const float* x;         // gathered tensor, NHWC [B,Hi,Wi,Ci]
  const float* w;         // packed weights [Co][Kp]
  float* y;               // output NHWC [B,Ho,Wo,Co]
  const float* bias;      // [Co] or null
  const float* residual;  // same shape as y or null: y += residual
  const float* mask_src;  // same shape as y or null: y = mask_src > 0 ? y : lrelu_slope*y
  const float* pro_scale; // [Ci] for PRO_AFFINE*
  const float* pro_shift;
  int pro_group_rows;     // > 0: rows [g*pro_group_rows, (g+1)*pro_group_rows) use pro_scale/shift + g*Ci (several
                          // independently normalised batches -- BatchNorm statistics per group -- in ONE GEMM)
  float mask_slope;       // 0 for ReLU backward, 0.2 for LeakyReLU backward
  float out_scale;        // multiplies the accumulator before bias/residual (1.0 normally)
  const float* scale0;    // optional device scalars: rows m < scale_split use *scale0, the others *scale1
  const float* scale1;    // (two forwards with different spectral-norm sigmas batched into one GEMM)
  int scale_split;
  float* stat_partials;   // optional [tiles_m][2][Co]: per-tile column sums of y and y*y (fused BatchNorm statistics)
  float* slab;            // split-K: raw partial sums go to slab[split][M][Co] (epilogue applied by a 2nd kernel ...
  int* tickets;           // ... or, where a kernel supports it and this is not null, by the tile's last-arriving workgroup:
                          // one zero-initialised counter per output tile, left at zero)
  int ksplit;             // number of K splits (gridDim.y); 1 = no split
  int res_relu;           // residual is added as max(residual, 0) (DBlock identity shortcut sees relu(x))
  int res_up;             // residual is a HALF-resolution tensor [B,Ho/2,Wo/2,Co]: 1: its bilinear x2 up-sampling is added
                          // (GBlock's up-sampled shortcut); 2: 0.25 * its value at (oy / 2, ox / 2) is added = the gradient of a
                          // 2x2 average pool (DBlock's pooled shortcut in the data gradient).  Winograd kernels and the split-K
                          // epilogue only
  int pro_mode;
  int M;                  // B*Ho*Wo
  ConvGeom g;
  FastDiv dWo, dHo;       // pixel index -> (b, oy, ox) without integer division
  unsigned long long* stamps;   // diagnostic build only (STAMP kernels): [workgroups][8] cycle / real-time stamps
  int tune;               // tuning sweeps: bit 0 = raised wave priority while the loader state is set up and the first
                          // tile staged, bit 1 = raised priority in the epilogue (a new / finishing wave otherwise gets
                          // the vector-issue slots its older MFMA-bound neighbours leave over)
};
// (OutMap travels beside ConvGemmArgs, as a kernel argument of its own: grown by its 36 bytes the argument block changed the register
//  allocation of conv_wino4_kernel<2,0> -- 55 -> 146 spilled VGPRs, 2.04 -> 2.24 ms per SNGAN-32 step; round 6)

// Arguments of the weight-gradient kernels (conv_wgrad.hip: implicit GEMM; conv_wgrad_wino.hip: Winograd F(3x3,2x2))
struct WgradArgs {
  const float* dy;        // pixel tensor [M][Co]
  const float* x;         // gathered tensor NHWC [B,Hi,Wi,Ci]
  float* slab;            // [splits][Co][Kp]
  const float* pro_scale;
  const float* pro_shift;
  int pro_mode;
  int M;
  int steps_per_split;    // K-steps (of 32 pixels) per split
  int seg_steps;          // K-steps per pixel segment (total steps when there is one segment)
  int splits_per_seg;     // splits never straddle a segment (= one of several batched forwards)
  long slab_stride;       // floats between consecutive split slabs (>= Co*Kp)
  long bias_off;          // >= 0: column sums of dy (bias gradient) go to slab[split][bias_off + n]
  ConvGeom g;
  FastDiv dWo, dHo;
  int adv_b, adv_y, adv_x;  // 32 pixels = adv_b images + adv_y rows + adv_x columns (pixel coordinates advance incrementally)
  int lgW, lgHW;            // log2(Wo), log2(Ho*Wo) when both are powers of two (P2 kernels)
  int tiles;                // output tiles (the grid is tiles * splits workgroups)
  int same;                 // stride-1, un-dilated, same-size conv (Hi == Ho, Wi == Wo): linear gather offsets
};

// Several layers' weight gradients in one launch (round 5: conv_wgrad_wino_batched_kernel, conv_wgrad_batched_kernel): passed BY VALUE
// as the kernel argument (< 4 KB); a workgroup finds its layer through the first-workgroup prefix
constexpr int WG_BATCH_MAX = 8;
struct WgradBatchArgs {
  int n;
  int blk0[WG_BATCH_MAX];      // first workgroup of layer j (a multiple of 8: the XCD remap inside a layer's range stays a permutation)
  int cnt[WG_BATCH_MAX];       // its workgroups: tiles x splits
  WgradArgs a[WG_BATCH_MAX];
};

// Loader of the Winograd weight-transform kernels (conv_wino.hip, conv_wino4.hip): the 3x3 taps of a block of 64 output x 32
// input channels of the packed weights w[Co][Kp], for thread (column = tid & 63, channel quad = tid >> 6) of a 512-thread
// workgroup.  Read with eight lanes along the input channels -- one whole 128-byte line per output channel and tap; with the
// lanes along the output channels, as the transform and its stores want them, every lane touched a line of its own for 16
// bytes -- and transposed through LDS ([tap][quad][64 + 1 columns]: the eight quads of a column fall on eight different bank
// quads).  flip: taps reversed (data gradient).  Returns false for a thread whose channel quad lies behind Ci.
constexpr int WT_LDS_F4 = 9 * 8 * 65;                  // f32x4 elements of the staging buffer (74 880 bytes)
__device__ __forceinline__ bool wino_stage_taps(const float* __restrict__ w, int co0, int c0, int Co, int Ci, int Kp, int flip,
                                                f32x4* __restrict__ sg, f32x4 (&g)[3][3]) {
  const int tid = threadIdx.x;
  {
    const int col = tid >> 3, q = tid & 7;
    const bool ok = co0 + col < Co && c0 + q * 4 < Ci;
    const float* src = w + (long)(co0 + col) * Kp + c0 + q * 4;
    f32x4 v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = ok ? *reinterpret_cast<const f32x4*>(src + (flip ? 8 - t : t) * Ci) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) sg[(t * 8 + q) * 65 + col] = v[t];
  }
  __syncthreads();
  const int col = tid & 63, q = tid >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = sg[(t * 8 + q) * 65 + col];
  return c0 + q * 4 < Ci;
}

// Transformed Winograd weights made AHEAD of the launches that use them (round 4): `diagan_wino_weights_batched` transforms the
// weights of many layers in one launch; a launch whose transformed weights already sit in a caller-owned buffer is told so
// through diagan_conv_gemm_weights_hint and skips its own per-launch transform kernel.  A format is (kind, flip, scale):
// kind 2 = F(2x2,3x3) image of wino_weight_kernel, 40 = F(4x4,3x3) MODE 0 / 3 unit order, 41 = the pooled modes' 25-frequency order;
// flip: taps reversed (data gradient); scale: 1, or 1/16 for the up-sampled-input mode.  42 / 43 (round 5): the F(4x4,3x3) formats
// split into three bf16 pieces per element, in the per-wave stream order of the X3 kernels (conv_wino4.hip).
// 50 (round 5): not a Winograd format -- the packed GEMM operand split into three bf16 planes [piece][Co][Kp] (conv_gemm_x3.hip)
enum WinoKind : int { WK_F2 = 2, WK_F4 = 40, WK_F4_POOL = 41, WK_F4X = 42, WK_F4X_POOL = 43, WK_GX3 = 50 };
struct WinoJob {               // one layer of a batched transform (device table)
  const float* w;              // packed weights [Co][Kp]
  float* u;                    // transformed weights (format's own size: wino_ws_floats / wino4_ws_floats)
  int Co, Ci, Kp, kind, flip, blk0;   // blk0: first workgroup of this job in the batched grid
  float scale;
  int pad_;
};
// (conv_gemm.hip) the transformed weights this launch may use instead of transforming into `ws`: returns the hinted buffer if
// the caller's hint matches the format, else nullptr; either way notes the format for diagan_conv_gemm_last_weight_format
const float* wino_weights_ready(int kind, int flip, float scale, long floats);

// Bijective XCD-aware remap of a linear workgroup id (cdna guide T1): consecutive logical tiles
// land on the same XCD (= same L2), so neighbouring tiles share halo rows and weight panels.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

}  // namespace diagan
