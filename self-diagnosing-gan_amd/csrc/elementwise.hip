// HBM-bound kernels of the SNGAN / DCGAN stacks: layout conversion at the NCHW boundary,
// BatchNorm statistics / backward, bilinear 2x upsample (fwd + adjoint), 2x2 average pool
// (fwd + adjoint), the discriminator head (ReLU + global sum pool + SN linear), bias gradients.
//
// They replace the ATen ops between the convolutions of torch_mimicry's GBlock / DBlock /
// DBlockOptimized and the SNGAN heads: nn.BatchNorm2d (batch statistics),
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False), F.avg_pool2d(x, 2),
// torch.sum(h, dim=(2,3)), torch.tanh, nn.ReLU, and the matching autograd backward formulas.
//
// Everything is NHWC fp32 with C % 4 == 0, processed as float4 (16 B per lane, coalesced).
// Reductions are two-stage with fp64 partials combined in a fixed order: deterministic, no atomics.
// Roofline: HBM (each kernel reads its inputs once and writes its outputs once).
#include "conv_common.h"

namespace diagan {

constexpr int EW_T = 256;
static inline int ew_blocks(long n) {
  long b = (n + EW_T - 1) / EW_T;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));  // grid-stride beyond 8192 blocks
}

// ---- layout conversion at the API boundary ---------------------------------------------------
// dst[b,h,w,0..Cp) = src[b,0..C,h,w] (zero padded channels)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, long npix, int HW, int C,
                                    int Cp) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long b = p / HW, hw = p - b * HW;
    for (int c = 0; c < Cp; ++c) dst[p * Cp + c] = c < C ? src[(b * C + c) * HW + hw] : 0.f;
  }
}
// dst[b,c,h,w] = src[b,h,w,c], c < C
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, long npix, int HW, int C,
                                    int Cp) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long b = p / HW, hw = p - b * HW;
    for (int c = 0; c < C; ++c) dst[(b * C + c) * HW + hw] = src[p * Cp + c];
  }
}

// mode 0: y = tanh(x) ; mode 1 (backward): y = g * (1 - t*t) with t = tanh output (in `x`)
__global__ void tanh_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ y, long n4,
                            int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (mode == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
    } else {
      const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
      v = gv * (1.f - v * v);
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

// ---- column reductions over [M][C] ------------------------------------------------------------
// Stage 1: thread = (row lane, float4 column); block (CW columns x RL row lanes) reduces a row range.
// MODE 0: {sum x, sum x^2}            (BatchNorm statistics)
// MODE 1: {sum g', sum g' * xhat}     (BatchNorm backward; g' = g masked by the ReLU after BN)
// MODE 2: {sum x, -}                  (bias gradient)
struct ColRedArgs {
  const float* x;      // MODE 0/2: input ; MODE 1: pre-BN activation
  const float* g;      // MODE 1: upstream gradient
  const float* scale;  // MODE 1: gamma*invstd
  const float* shift;  // MODE 1: beta - mean*scale
  const float* mean;
  const float* invstd;
  double* partials;    // [splits][2][C]
  long M;
  int C;
  int rows_per_split;
  int relu;            // MODE 1: g' = g * (y > 0 ? 1 : slope) with y = scale*x+shift (ReLU: slope 0)
  float slope;
  const float* drop;   // MODE 1: optional dropout keep-mask, multiplies g together with drop_scale (1 / (1 - p) for a 0 / 1 mask;
  float drop_scale;    //         1 for a mask that carries the scale itself)
  int splits = 0;      // MODE 0 with gridDim.z > 1: group z reduces rows [z M, (z + 1) M) into partials + z * splits * 2 * C
};

template <int MODE>
__global__ __launch_bounds__(EW_T) void colred_kernel(const ColRedArgs a) {
  __shared__ double red[2][EW_T][4];
  const int C4 = a.C >> 2;
  const int CW = C4 < 64 ? C4 : 64;          // float4 columns per block
  const int RL = EW_T / CW;                  // row lanes
  const int cc = threadIdx.x % CW, rl = threadIdx.x / CW;
  const int c4 = blockIdx.x * CW + cc;
  const bool col_ok = c4 < C4 && rl < RL;
  const long rg = (long)blockIdx.z * a.M;                                   // (stacked batches: one launch for all groups)
  const long r0 = rg + (long)blockIdx.y * a.rows_per_split;
  const long r1 = r0 + a.rows_per_split < rg + a.M ? r0 + a.rows_per_split : rg + a.M;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (col_ok) {
    f32x4 sc, sh, mu, is;
    if (MODE == 1) {
      sc = reinterpret_cast<const f32x4*>(a.scale)[c4];
      sh = reinterpret_cast<const f32x4*>(a.shift)[c4];
      mu = reinterpret_cast<const f32x4*>(a.mean)[c4];
      is = reinterpret_cast<const f32x4*>(a.invstd)[c4];
    }
    // Four rows' loads in flight per thread (one row per iteration left a thread with two loads in flight: 1.5 TB/s); the rows
    // are accumulated in the same order as before, so the sums are unchanged.  Rows behind the range re-read its last row
    // (no branch around the loads) and are skipped by the accumulation.
    constexpr int UR = 4;
    for (long rb = r0 + rl; rb < r1; rb += (long)UR * RL) {
      f32x4 xs[UR], gs[UR], ds[UR];
#pragma unroll
      for (int k = 0; k < UR; ++k) {
        const long r = rb + (long)k * RL < r1 ? rb + (long)k * RL : r1 - 1;
        xs[k] = reinterpret_cast<const f32x4*>(a.x)[r * C4 + c4];
        if (MODE == 1) gs[k] = reinterpret_cast<const f32x4*>(a.g)[r * C4 + c4];
      }
      if (MODE == 1 && a.drop) {
#pragma unroll
        for (int k = 0; k < UR; ++k) {
          const long r = rb + (long)k * RL < r1 ? rb + (long)k * RL : r1 - 1;
          ds[k] = reinterpret_cast<const f32x4*>(a.drop)[r * C4 + c4];
        }
      }
#pragma unroll
      for (int k = 0; k < UR; ++k) {
        if (rb + (long)k * RL >= r1) break;
        const f32x4 xv = xs[k];
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += xv[e]; s2[e] += (double)xv[e] * xv[e]; }
        } else if (MODE == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s1[e] += xv[e];
        } else {
          f32x4 gv = gs[k];
          const f32x4 xh = (xv - mu) * is;
          if (a.drop) gv *= ds[k] * a.drop_scale;
          if (a.relu) {
            const f32x4 y = xv * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = y[e] > 0.f ? gv[e] : gv[e] * a.slope;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { s1[e] += gv[e]; s2[e] += (double)gv[e] * xh[e]; }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  if (rl == 0 && c4 < C4) {
    double t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
    for (int k = 0; k < RL; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) { t1[e] += red[0][k * CW + cc][e]; t2[e] += red[1][k * CW + cc][e]; }
    double* p = a.partials + ((long)blockIdx.z * a.splits + blockIdx.y) * 2 * a.C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[c4 * 4 + e] = t1[e]; p[a.C + c4 * 4 + e] = t2[e]; }
  }
}

// Deterministic combine of the per-split partials: block = 64 channels x 4 split-lanes; lane l sums
// splits l, l+4, ... and the four lane sums are added in a fixed order.  True for the lanes that own a channel.
__device__ __forceinline__ bool combine_partials(const double* __restrict__ partials, int splits, int C,
                                                 double (*red)[64][2], int* c_out, double* s1, double* s2) {
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double a = 0, b = 0;
  if (c < C) {
    // eight splits' loads in flight, added in the order of the plain loop (same sums): one split per iteration was a chain of
    // splits / 4 dependent loads, 22 us for the 1024 partials of a large layer
    int k = sl;
    for (; k + 28 < splits; k += 32) {
      double pa[8], pb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        pa[u] = partials[(long)(k + 4 * u) * 2 * C + c];
        pb[u] = partials[(long)(k + 4 * u) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += pa[u]; b += pb[u]; }
    }
    for (; k < splits; k += 4) { a += partials[(long)k * 2 * C + c]; b += partials[(long)k * 2 * C + C + c]; }
  }
  red[sl][cl][0] = a;
  red[sl][cl][1] = b;
  __syncthreads();
  *c_out = c;
  if (sl != 0 || c >= C) return false;
  *s1 = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
  *s2 = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
  return true;
}

// BatchNorm forward finalize: batch statistics -> affine form + running stats (momentum 0.1, unbiased var)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partials, int splits, int C, long M,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out, int training) {
  __shared__ double red[4][64][2];
  int c;
  double s1 = 0, s2 = 0;
  if (training) {
    if (!combine_partials(partials, splits, C, red, &c, &s1, &s2)) return;
  } else {
    c = blockIdx.x * 64 + (threadIdx.x & 63);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
  }
  float mean, var;
  if (training) {
    const double m = s1 / (double)M;
    double v = s2 / (double)M - m * m;
    v = v < 0 ? 0 : v;
    mean = (float)m;
    var = (float)v;
    const float unbiased = M > 1 ? (float)(v * (double)M / (double)(M - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float invstd = 1.f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  mean_out[c] = mean;
  invstd_out[c] = invstd;
  scale_out[c] = sc;
  shift_out[c] = beta[c] - mean * sc;
}

// BatchNorm finalize from the per-tile column sums emitted by the producing conv's epilogue
// (partials[tile][2][C] floats; summed over tiles in double, fixed order)
// groups > 1: `groups` independently normalised batches whose tiles follow each other in `partials` (the stacked
// generator forward): one launch finalises them IN ORDER, so the running statistics see the same sequence of momentum
// updates as `groups` separate forwards; outputs are [groups][C].
// First stage for long tile lists (a stacked 64x64 generator block has 100 000 tiles for 64 channels, i.e. FOUR
// 16-channel blocks in the finalize kernel below: 170 us on average, 1 ms at worst): grid (C/16, S, groups), block
// (s, g) sums tiles [s*chunk, (s+1)*chunk) of group g in double and writes ws[g][s][2][C]; the finalize kernel then
// runs over the S partial rows.  Fixed split count and fixed order: deterministic.
__global__ __launch_bounds__(256) void bn_partials_reduce_kernel(const float* __restrict__ partials, int tiles, int C, int S,
                                                                 double* __restrict__ ws) {
  __shared__ double red[16][16][2];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl, s = blockIdx.y, g = blockIdx.z;
  const int chunk = (tiles + S - 1) / S, k0 = s * chunk, k1 = min(tiles, k0 + chunk);
  const float* part = partials + (long)g * tiles * 2 * C;
  double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  if (c < C) {
    int k = k0 + sl;
    for (; k + 48 < k1; k += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] += part[(long)(k + 16 * u) * 2 * C + c];
        b[u] += part[(long)(k + 16 * u) * 2 * C + C + c];
      }
    }
    for (; k < k1; k += 16) { a[0] += part[(long)k * 2 * C + c]; b[0] += part[(long)k * 2 * C + C + c]; }
  }
  red[sl][cl][0] = (a[0] + a[1]) + (a[2] + a[3]);
  red[sl][cl][1] = (b[0] + b[1]) + (b[2] + b[3]);
  __syncthreads();
  if (sl == 0 && c < C) {
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
    double* o = ws + ((long)g * S + s) * 2 * C;
    o[c] = s1;
    o[C + c] = s2;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_fused_kernel(const T* __restrict__ partials, int tiles, int C, long M,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out, int groups) {
  // block = 16 channels x 16 tile-lanes; every lane keeps 4 independent double accumulators so that
  // its loads overlap; lanes and accumulators are combined in a fixed order (deterministic)
  __shared__ double red[16][16][2];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float rm = 0.f, rv = 0.f;
  if (sl == 0 && c < C) { rm = running_mean[c]; rv = running_var[c]; }
  for (int g = 0; g < groups; ++g) {
    const T* part = partials + (long)g * tiles * 2 * C;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (c < C) {
      int k = sl;
      for (; k + 48 < tiles; k += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a[u] += part[(long)(k + 16 * u) * 2 * C + c];
          b[u] += part[(long)(k + 16 * u) * 2 * C + C + c];
        }
      }
      for (; k < tiles; k += 16) { a[0] += part[(long)k * 2 * C + c]; b[0] += part[(long)k * 2 * C + C + c]; }
    }
    if (g > 0) __syncthreads();          // the previous group's reduction has been read
    red[sl][cl][0] = (a[0] + a[1]) + (a[2] + a[3]);
    red[sl][cl][1] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    if (sl == 0 && c < C) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) { s1 += red[k][cl][0]; s2 += red[k][cl][1]; }
      const double m = s1 / (double)M;
      double v = s2 / (double)M - m * m;
      v = v < 0 ? 0 : v;
      const float mean = (float)m, var = (float)v;
      const float unbiased = M > 1 ? (float)(v * (double)M / (double)(M - 1)) : var;
      rm = (1.f - momentum) * rm + momentum * mean;
      rv = (1.f - momentum) * rv + momentum * unbiased;
      const float invstd = 1.f / sqrtf(var + eps);
      const float sc = gamma[c] * invstd;
      const long o = (long)g * C + c;
      mean_out[o] = mean;
      invstd_out[o] = invstd;
      scale_out[o] = sc;
      shift_out[o] = beta[c] - mean * sc;
    }
  }
  if (sl == 0 && c < C) { running_mean[c] = rm; running_var[c] = rv; }
}

// BatchNorm backward finalize: dgamma += s2, dbeta += s1, coef = {s1/M, s2/M}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partials, int splits, int C, long M,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef, int accumulate, int batch_stats) {
  __shared__ double red[4][64][2];
  int c;
  double s1, s2;
  if (!combine_partials(partials, splits, C, red, &c, &s1, &s2)) return;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
  // eval-mode BatchNorm (running statistics) is a fixed affine map: no mean / variance terms
  coef[c] = batch_stats ? (float)(s1 / (double)M) : 0.f;
  coef[C + c] = batch_stats ? (float)(s2 / (double)M) : 0.f;
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const double* __restrict__ partials, int splits, int C, float* __restrict__ out,
                                       int accumulate) {
  __shared__ double red[4][64][2];
  int c;
  double s1, s2;
  if (!combine_partials(partials, splits, C, red, &c, &s1, &s2)) return;
  out[c] = (accumulate ? out[c] : 0.f) + (float)s1;
}

// dx = scale * (g' - c1 - xhat*c2) (+ residual)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ coef, const float* __restrict__ residual,
                                    float* __restrict__ out, long M, int C, int relu, float slope,
                                    const float* __restrict__ drop, float drop_scale) {
  const int C4 = C >> 2;
  const long n4 = M * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4];
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[c4];
    const f32x4 k1 = reinterpret_cast<const f32x4*>(coef)[c4];
    const f32x4 k2 = reinterpret_cast<const f32x4*>(coef + C)[c4];
    if (drop) gv *= reinterpret_cast<const f32x4*>(drop)[i] * drop_scale;
    if (relu) {
      const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[c4];
      const f32x4 y = xv * sc + sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) gv[e] = y[e] > 0.f ? gv[e] : gv[e] * slope;
    }
    f32x4 o = sc * (gv - k1 - (xv - mu) * is * k2);
    if (residual) o += reinterpret_cast<const f32x4*>(residual)[i];
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// ---- bilinear 2x upsample (align_corners=False) and its adjoint ---------------------------------
// out[2j]   = 0.25*x[max(j-1,0)] + 0.75*x[j] ; out[2j+1] = 0.75*x[j] + 0.25*x[min(j+1,H-1)]
// One thread per INPUT position and 4 channels: the 3x3 neighbourhood (9 loads, prologue applied once each) gives the 2x2
// outputs (2.25 loads per output; one thread per output read 4 and ran the prologue 4 times: 2.6 TB/s).  Same blend grouping
// as before -- (w0*a + w1*b) along x inside the y blend, as PyTorch evaluates it -- so the values are unchanged.
__global__ __launch_bounds__(EW_T) void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C,
                                  int pro_mode, const float* __restrict__ scale, const float* __restrict__ shift,
                                  int group_imgs) {
  const unsigned C4 = (unsigned)C >> 2;
  const unsigned n4 = (unsigned)B * H * W * C4;            // (< 2^31: checked by the launcher)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const unsigned c4 = i % C4;
    unsigned p = i / C4;
    const int jx = (int)(p % (unsigned)W); p /= (unsigned)W;
    const int jy = (int)(p % (unsigned)H);
    const int b = (int)(p / (unsigned)H);
    const float* base = x + (long)b * H * W * C + c4 * 4;
    const int pc = (group_imgs > 0 ? (b / group_imgs) * C : 0) + (int)c4 * 4;     // per-group BatchNorm scale / shift
    const int ys[3] = {max(jy - 1, 0), jy, min(jy + 1, H - 1)};
    const int xs[3] = {max(jx - 1, 0), jx, min(jx + 1, W - 1)};
    f32x4 v[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) v[r][q] = *reinterpret_cast<const f32x4*>(base + ((long)ys[r] * W + xs[q]) * C);
    f32x4 he[3], ho[3];                                     // x blends of the three rows: even / odd output column
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int q = 0; q < 3; ++q) v[r][q] = apply_pro(v[r][q], pro_mode, scale, shift, pc);
      he[r] = 0.25f * v[r][0] + 0.75f * v[r][1];
      ho[r] = 0.75f * v[r][1] + 0.25f * v[r][2];
    }
    f32x4* o = reinterpret_cast<f32x4*>(out) + (((long)b * 2 * H + 2 * jy) * 2 * W + 2 * jx) * C4 + c4;
    const long row = 2L * W * C4;
    o[0] = 0.25f * he[0] + 0.75f * he[1];
    o[C4] = 0.25f * ho[0] + 0.75f * ho[1];
    o[row] = 0.75f * he[1] + 0.25f * he[2];
    o[row + C4] = 0.75f * ho[1] + 0.25f * ho[2];
  }
}

// adjoint: gx[j] = 0.25*g[clamp(2j-1)] + 0.75*g[2j] + 0.75*g[2j+1] + 0.25*g[clamp(2j+2)]
__global__ void upsample2x_bwd_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int H, int W, int C,
                                      const float* __restrict__ residual) {
  const int C4 = C >> 2, Ho = 2 * H, Wo = 2 * W;
  const long n4 = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int jx = (int)(p % W); p /= W;
    const int jy = (int)(p % H);
    const int b = (int)(p / H);
    const float* base = g + (long)b * Ho * Wo * C + c4 * 4;
    const float wt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const int yy = min(max(2 * jy - 1 + dy, 0), Ho - 1);
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int xx = min(max(2 * jx - 1 + dx, 0), Wo - 1);
        row += wt[dx] * *reinterpret_cast<const f32x4*>(base + ((long)yy * Wo + xx) * C);
      }
      acc += wt[dy] * row;
    }
    if (residual) acc += reinterpret_cast<const f32x4*>(residual)[i];
    reinterpret_cast<f32x4*>(out)[i] = acc;
  }
}

// ---- 2x2 average pool and its adjoint ----------------------------------------------------------
__global__ void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C,
                                const float* __restrict__ residual, int relu_in) {
  const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long n4 = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const float* base = x + (((long)b * H + 2 * oy) * W + 2 * ox) * C + c4 * 4;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(base), a1 = *reinterpret_cast<const f32x4*>(base + C);
    f32x4 a2 = *reinterpret_cast<const f32x4*>(base + (long)W * C);
    f32x4 a3 = *reinterpret_cast<const f32x4*>(base + (long)W * C + C);
    if (relu_in) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] = fmaxf(a0[e], 0.f); a1[e] = fmaxf(a1[e], 0.f); a2[e] = fmaxf(a2[e], 0.f); a3[e] = fmaxf(a3[e], 0.f);
      }
    }
    f32x4 o = ((a0 + a1) + (a2 + a3)) * 0.25f;
    if (residual) o += reinterpret_cast<const f32x4*>(residual)[i];
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// out[b,y,x,c] = 0.25*g[b,y/2,x/2,c] (+ residual)
__global__ void avgpool2_bwd_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int H, int W, int C,
                                    const float* __restrict__ residual) {
  const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long n4 = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    f32x4 o = reinterpret_cast<const f32x4*>(g)[(((long)b * Ho + (yy >> 1)) * Wo + (xx >> 1)) * C4 + c4] * 0.25f;
    if (residual) o += reinterpret_cast<const f32x4*>(residual)[i];
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// out[b, y', x', c] = 0.25 * sum over the 2x2 window of act(x) whose LOWER-RIGHT corner is pixel (y', x') (zero outside the
// image), y' = 0 .. H, x' = 0 .. W: the weight gradient of avg_pool2d(conv3x3(act(x)), 2) is the weight gradient of a 3x3 /
// stride 2 / pad 0 convolution over this (H+1) x (W+1) image against the POOLED gradient -- a quarter of the pixels
__global__ void boxsum2_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C, int relu_in) {
  const int C4 = C >> 2, Ho = H + 1, Wo = W + 1;
  const long n4 = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long p = i / C4;
    const int xx = (int)(p % Wo); p /= Wo;
    const int yy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 0; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 0; ++dx) {
        const int y = yy + dy, xq = xx + dx;
        if (y >= 0 && y < H && xq >= 0 && xq < W) {
          f32x4 v = reinterpret_cast<const f32x4*>(x)[(((long)b * H + y) * W + xq) * C4 + c4];
          if (relu_in) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          s += v;
        }
      }
    reinterpret_cast<f32x4*>(out)[i] = s * 0.25f;
  }
}

// ---- discriminator head: ReLU -> sum over H,W -> (SN) linear to one logit ----------------------
// pooled[b][c] = sum_hw relu(x[b,hw,c])           grid (C4/64, B)
__global__ __launch_bounds__(EW_T) void relu_sumpool_kernel(const float* __restrict__ x, float* __restrict__ pooled,
                                                            int HW, int C) {
  __shared__ float red[EW_T][4];
  const int C4 = C >> 2;
  const int CW = C4 < 64 ? C4 : 64, RL = EW_T / CW;
  const int cc = threadIdx.x % CW, rl = threadIdx.x / CW;
  const int c4 = blockIdx.x * CW + cc, b = blockIdx.y;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c4 < C4 && rl < RL)
    for (int r = rl; r < HW; r += RL) {
      const f32x4 v = reinterpret_cast<const f32x4*>(x)[((long)b * HW + r) * C4 + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += fmaxf(v[e], 0.f);
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if (rl == 0 && c4 < C4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < RL; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += red[k * CW + cc][e];
    reinterpret_cast<f32x4*>(pooled)[(long)b * C4 + c4] = t;
  }
}

// logit[b] = inv_sigma * sum_c pooled[b][c]*w[c] + bias        (one wave per sample)
__global__ void head_linear_kernel(const float* __restrict__ pooled, const float* __restrict__ w,
                                   const float* __restrict__ inv_sigma, const float* __restrict__ inv_sigma1,
                                   int split_b, const float* __restrict__ bias, float* __restrict__ logit, int B,
                                   int C) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s = fmaf(pooled[(long)b * C + c], w[c], s);
  s = wave_sum(s);
  const float* ip = (inv_sigma1 && b >= split_b) ? inv_sigma1 : inv_sigma;
  if (lane == 0) logit[b] = s * (ip ? ip[0] : 1.f) + (bias ? bias[0] : 0.f);
}

// gx[b,hw,c] = dlogit[b] * w[c] * inv_sigma * (x > 0)
__global__ void head_bwd_kernel(const float* __restrict__ dlogit, const float* __restrict__ w,
                                const float* __restrict__ inv_sigma, const float* __restrict__ inv_sigma1,
                                int split_b, const float* __restrict__ x, float* __restrict__ gx, long n4, int HW,
                                int C) {
  const int C4 = C >> 2;
  const float inv0 = inv_sigma ? inv_sigma[0] : 1.f, inv1 = inv_sigma1 ? inv_sigma1[0] : inv0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long b = i / ((long)C4 * HW);
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 wv = reinterpret_cast<const f32x4*>(w)[c4];
    const float d = dlogit[b] * (b >= split_b ? inv1 : inv0);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = xv[e] > 0.f ? d * wv[e] : 0.f;
    reinterpret_cast<f32x4*>(gx)[i] = o;
  }
}

// single block: G[c] = sum_b dlogit[b]*pooled[b][c]; dot = <G, w>; dbias (+)= sum_b dlogit[b]
__global__ __launch_bounds__(EW_T) void head_wgrad_kernel(const float* __restrict__ dlogit, const float* __restrict__ pooled,
                                                          const float* __restrict__ w, float* __restrict__ G,
                                                          double* __restrict__ dot, float* __restrict__ dbias, int B,
                                                          int C, int accumulate_bias) {
  __shared__ double red[4];
  double d = 0.0;
  for (int c = threadIdx.x; c < C; c += EW_T) {
    float s = 0.f;
#pragma unroll 16
    for (int b = 0; b < B; ++b) s = fmaf(dlogit[b], pooled[(long)b * C + c], s);   // unrolled: 16 loads in flight
    G[c] = s;
    d += (double)s * w[c];
  }
  d = wave_sum(d);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    dot[0] = (red[0] + red[1]) + (red[2] + red[3]);
    if (dbias) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += dlogit[b];
      dbias[0] = (accumulate_bias ? dbias[0] : 0.f) + s;
    }
  }
}

// a = act(x*scale[c]+shift[c]) * drop * drop_scale ; act: v > 0 ? v : slope*v   (slope 0 ReLU, 0.2 LeakyReLU, 1 identity)
__global__ void act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                               const float* __restrict__ shift, float slope, const float* __restrict__ drop, float drop_scale,
                               float* __restrict__ out, long n4, int C4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    if (scale) {
      const int c4 = (int)(i % C4);
      v = v * reinterpret_cast<const f32x4*>(scale)[c4] + reinterpret_cast<const f32x4*>(shift)[c4];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
    if (drop) v *= reinterpret_cast<const f32x4*>(drop)[i] * drop_scale;
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

// backward of act(x)*drop*drop_scale without BatchNorm: gx = g * drop * drop_scale * (x > 0 ? 1 : slope)
__global__ void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, float slope,
                               const float* __restrict__ drop, float drop_scale, float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    if (drop) gv *= reinterpret_cast<const f32x4*>(drop)[i] * drop_scale;
#pragma unroll
    for (int e = 0; e < 4; ++e) gv[e] = xv[e] > 0.f ? gv[e] : gv[e] * slope;
    reinterpret_cast<f32x4*>(out)[i] = gv;
  }
}

// dw[c] += sum_b dlogit[b]*x[b][c] ; dbias += sum_b dlogit[b]     (weight gradient of a Linear(C, 1))
__global__ __launch_bounds__(EW_T) void linear1_wgrad_kernel(const float* __restrict__ dlogit, const float* __restrict__ x,
                                                             float* __restrict__ dw, float* __restrict__ dbias, int B,
                                                             int C) {
  const int c = blockIdx.x * EW_T + threadIdx.x;
  if (c < C) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s = fmaf(dlogit[b], x[(long)b * C + c], s);
    dw[c] += s;
  }
  if (dbias && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogit[b];
    dbias[0] += s;
  }
}

// gx[b][j] = dlogit[b] * w[j]   (input gradient of a Linear(C, 1))
__global__ void linear1_bwd_input_kernel(const float* __restrict__ dlogit, const float* __restrict__ w,
                                         float* __restrict__ gx, long n4, int C4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    reinterpret_cast<f32x4*>(gx)[i] = reinterpret_cast<const f32x4*>(w)[i % C4] * dlogit[i / C4];
}

// out = a + b (float4)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

static int colred_geometry(long M, int C, int* splits, int* rows_per_split, dim3* grid) {
  const int C4 = C / 4;
  const int CW = C4 < 64 ? C4 : 64;
  const int gx = cdiv(C4, CW);
  // a thread walks rows_per_split / (256 / columns) rows, four loads in flight: with 256 rows per split and 1024 blocks a
  // 65536 x 256 reduction was 16 dependent iterations = 21 us; 128 / 2048: SNGAN-32 +0.8 %, SNGAN-64 +0.2 % end to end
  // (64 / 4096: +1.0 % / -0.45 %; tools/probe/colred_sweep.sh)
  static const int blocks = getenv("DIAGAN_COLRED_BLOCKS") ? atoi(getenv("DIAGAN_COLRED_BLOCKS")) : 2048;
  static const int minrows = getenv("DIAGAN_COLRED_ROWS") ? atoi(getenv("DIAGAN_COLRED_ROWS")) : 128;
  int s = cdiv(blocks, gx);                     // ~8 blocks per CU in total
  const long max_s = (M + minrows - 1) / minrows;           // at least 128 rows per split
  if (s > max_s) s = (int)max_s;
  if (s < 1) s = 1;
  *rows_per_split = (int)((M + s - 1) / s);
  *splits = (int)((M + *rows_per_split - 1) / *rows_per_split);
  *grid = dim3(gx, *splits);
  return 0;
}

}  // namespace diagan

using namespace diagan;

#define ST ((hipStream_t)stream)

DIAGAN_API int diagan_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream) {
  DG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nchw_to_nhwc: bad args");
  const long npix = (long)B * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks(npix)), dim3(EW_T), 0, ST, src, dst, npix, H * W, C, Cp);
  return check_launch("nchw_to_nhwc");
}

DIAGAN_API int diagan_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream) {
  DG_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nhwc_to_nchw: bad args");
  const long npix = (long)B * H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks(npix)), dim3(EW_T), 0, ST, src, dst, npix, H * W, C, Cp);
  return check_launch("nhwc_to_nchw");
}

DIAGAN_API int diagan_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
  DG_REQUIRE(x && y && n > 0 && (n & 3) == 0, "tanh_fwd: bad args");
  hipLaunchKernelGGL(tanh_kernel, dim3(ew_blocks(n / 4)), dim3(EW_T), 0, ST, x, nullptr, y, (long)(n / 4), 0);
  return check_launch("tanh_fwd");
}

DIAGAN_API int diagan_tanh_bwd(const float* y, const float* g, float* gx, int64_t n, void* stream) {
  DG_REQUIRE(y && g && gx && n > 0 && (n & 3) == 0, "tanh_bwd: bad args");
  hipLaunchKernelGGL(tanh_kernel, dim3(ew_blocks(n / 4)), dim3(EW_T), 0, ST, y, g, gx, (long)(n / 4), 1);
  return check_launch("tanh_bwd");
}

// workspace: diagan_colred_workspace(M, C) bytes
DIAGAN_API int64_t diagan_colred_workspace(int64_t M, int C) {
  int splits, rps;
  dim3 grid;
  colred_geometry(M, C, &splits, &rps, &grid);
  return (int64_t)splits * 2 * C * sizeof(double);
}

DIAGAN_API int diagan_bn_stats(const float* x, int64_t M, int C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, int training,
                               float* mean_out, float* invstd_out, float* scale_out, float* shift_out,
                               void* workspace, void* stream) {
  DG_REQUIRE(x && gamma && beta && running_mean && running_var && mean_out && invstd_out && scale_out && shift_out,
             "bn_stats: null pointer");
  DG_REQUIRE(M > 0 && C > 0 && (C & 3) == 0, "bn_stats: bad dims M=%ld C=%d", (long)M, C);
  int splits = 0, rps = 0;
  dim3 grid;
  if (training) {
    DG_REQUIRE(workspace, "bn_stats: workspace required in training mode");
    colred_geometry(M, C, &splits, &rps, &grid);
    ColRedArgs a{x, nullptr, nullptr, nullptr, nullptr, nullptr, (double*)workspace, (long)M, C, rps, 0, 0.f, nullptr, 1.f};
    hipLaunchKernelGGL(colred_kernel<0>, grid, dim3(EW_T), 0, ST, a);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(256), 0, ST, (const double*)workspace, splits, C,
                     (long)M, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out,
                     shift_out, training);
  return check_launch("bn_stats");
}

// training-mode statistics of `groups` batches stacked along the rows of x[groups * M][C], each normalised on its own, the
// running statistics updated in group order (as diagan_bn_stats called once per group -- which cost a 20 us launch pair per
// group for the generator's first BatchNorm): ONE column-reduction launch and ONE finalisation.  Outputs [groups][C];
// workspace: groups * diagan_colred_workspace(M, C) bytes.
DIAGAN_API int diagan_bn_stats_grouped(const float* x, int64_t M, int C, int groups, const float* gamma, const float* beta, float eps,
                                       float momentum, float* running_mean, float* running_var, float* mean_out,
                                       float* invstd_out, float* scale_out, float* shift_out, void* workspace, void* stream) {
  DG_REQUIRE(x && gamma && beta && running_mean && running_var && mean_out && invstd_out && scale_out && shift_out && workspace,
             "bn_stats_grouped: null pointer");
  DG_REQUIRE(M > 0 && C > 0 && (C & 3) == 0 && groups >= 1 && groups <= 65535, "bn_stats_grouped: bad dims M=%ld C=%d groups=%d", (long)M, C, groups);
  int splits = 0, rps = 0;
  dim3 grid;
  colred_geometry(M, C, &splits, &rps, &grid);
  ColRedArgs a{x, nullptr, nullptr, nullptr, nullptr, nullptr, (double*)workspace, (long)M, C, rps, 0, 0.f, nullptr, 1.f, splits};
  grid.z = groups;
  hipLaunchKernelGGL(colred_kernel<0>, grid, dim3(EW_T), 0, ST, a);
  hipLaunchKernelGGL(bn_finalize_fused_kernel<double>, dim3(cdiv(C, 16)), dim3(256), 0, ST, (const double*)workspace, splits, C,
                     (long)M, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out,
                     groups);
  return check_launch("bn_stats_grouped");
}

// split count of the two-stage path (1 = single stage); the workspace must hold groups * splits * 2 * C doubles
DIAGAN_API int diagan_bn_stats_fused_splits(int tiles, int C, int groups) {
  // (a stacked forward's groups are finalised IN ORDER by one workgroup per 16 channels: with six groups of 128 tiles that
  //  serial walk took 21 us; summed first by splits x groups workgroups it is a handful of partials per group)
  if ((long)tiles * groups < 256) return 1;
  long s = 1024 / ((long)cdiv(C, 16) * groups);
  const long smax = tiles / 32;          // at least two tiles per lane of a block
  if (s > smax) s = smax;
  return s < 2 ? 1 : (int)s;
}

DIAGAN_API int diagan_bn_stats_fused(const float* partials, int tiles, int64_t M, int C, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, float* mean_out, float* invstd_out, float* scale_out,
                                     float* shift_out, int groups, double* workspace, int64_t workspace_doubles,
                                     void* stream) {
  DG_REQUIRE(partials && gamma && beta && running_mean && running_var && mean_out && invstd_out && scale_out && shift_out,
             "bn_stats_fused: null pointer");
  DG_REQUIRE(tiles > 0 && M > 0 && C > 0 && groups > 0, "bn_stats_fused: bad dims");
  const int S = diagan_bn_stats_fused_splits(tiles, C, groups);
  if (S > 1 && workspace && workspace_doubles >= (int64_t)groups * S * 2 * C) {
    hipLaunchKernelGGL(bn_partials_reduce_kernel, dim3(cdiv(C, 16), S, groups), dim3(256), 0, ST, partials, tiles, C, S,
                       workspace);
    hipLaunchKernelGGL(bn_finalize_fused_kernel<double>, dim3(cdiv(C, 16)), dim3(256), 0, ST, (const double*)workspace, S,
                       C, (long)M, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out,
                       shift_out, groups);
  } else {
    hipLaunchKernelGGL(bn_finalize_fused_kernel<float>, dim3(cdiv(C, 16)), dim3(256), 0, ST, partials, tiles, C, (long)M,
                       gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out,
                       groups);
  }
  return check_launch("bn_stats_fused");
}

// dx = BN_backward(relu_backward(g)); dgamma/dbeta (+)=.  coef: 2*C floats scratch.
DIAGAN_API int diagan_bn_bwd(const float* g, const float* x, int64_t M, int C, const float* scale, const float* shift,
                             const float* mean, const float* invstd, int batch_stats, int relu, float slope,
                             const float* drop, float drop_scale, float* dgamma, float* dbeta,
                             int accumulate_param_grads, const float* residual, float* dx, float* coef,
                             void* workspace, void* stream) {
  DG_REQUIRE(g && x && scale && shift && mean && invstd && dgamma && dbeta && dx && coef && workspace,
             "bn_bwd: null pointer");
  DG_REQUIRE(M > 0 && C > 0 && (C & 3) == 0, "bn_bwd: bad dims");
  int splits, rps;
  dim3 grid;
  colred_geometry(M, C, &splits, &rps, &grid);
  ColRedArgs a{x, g, scale, shift, mean, invstd, (double*)workspace, (long)M, C, rps, relu, slope, drop, drop_scale};
  hipLaunchKernelGGL(colred_kernel<1>, grid, dim3(EW_T), 0, ST, a);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(256), 0, ST, (const double*)workspace, splits,
                     C, (long)M, dgamma, dbeta, coef, accumulate_param_grads, batch_stats);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(M * (C / 4))), dim3(EW_T), 0, ST, g, x, scale, shift, mean,
                     invstd, coef, residual, dx, (long)M, C, relu, slope, drop, drop_scale);
  return check_launch("bn_bwd");
}

DIAGAN_API int diagan_colsum(const float* x, int64_t M, int C, float* out, int accumulate, void* workspace,
                             void* stream) {
  DG_REQUIRE(x && out && workspace && M > 0 && C > 0 && (C & 3) == 0, "colsum: bad args");
  int splits, rps;
  dim3 grid;
  colred_geometry(M, C, &splits, &rps, &grid);
  ColRedArgs a{x, nullptr, nullptr, nullptr, nullptr, nullptr, (double*)workspace, (long)M, C, rps, 0, 0.f, nullptr, 1.f};
  hipLaunchKernelGGL(colred_kernel<2>, grid, dim3(EW_T), 0, ST, a);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, 64)), dim3(256), 0, ST, (const double*)workspace, splits, C,
                     out, accumulate);
  return check_launch("colsum");
}

DIAGAN_API int diagan_upsample2x(const float* x, float* out, int B, int H, int W, int C, int pro_mode,
                                 const float* scale, const float* shift, int group_imgs, void* stream) {
  DG_REQUIRE(group_imgs >= 0 && (group_imgs == 0 || B % group_imgs == 0), "upsample2x: group_imgs=%d must divide B=%d", group_imgs, B);
  DG_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "upsample2x: bad args");
  DG_REQUIRE(pro_mode >= 0 && pro_mode <= 4, "upsample2x: bad pro_mode");
  DG_REQUIRE(!(pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE) || (scale && shift), "upsample2x: affine needs scale/shift");
  DG_REQUIRE((long)B * H * W * C < (1L << 31), "upsample2x: input must have fewer than 2^31 elements");
  hipLaunchKernelGGL(upsample2x_kernel, dim3(ew_blocks((long)B * H * W * C / 4)), dim3(EW_T), 0, ST, x, out, B, H, W, C,
                     pro_mode, scale, shift, group_imgs);
  return check_launch("upsample2x");
}

DIAGAN_API int diagan_upsample2x_bwd(const float* g, float* out, int B, int H, int W, int C, const float* residual,
                                     void* stream) {
  DG_REQUIRE(g && out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "upsample2x_bwd: bad args");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_blocks((long)B * H * W * C / 4)), dim3(EW_T), 0, ST, g, out, B, H,
                     W, C, residual);
  return check_launch("upsample2x_bwd");
}

DIAGAN_API int diagan_avgpool2(const float* x, float* out, int B, int H, int W, int C, const float* residual,
                               int relu_in, void* stream) {
  DG_REQUIRE(x && out && B > 0 && H > 1 && W > 1 && (H & 1) == 0 && (W & 1) == 0 && C > 0 && (C & 3) == 0,
             "avgpool2: bad args (even H, W required)");
  hipLaunchKernelGGL(avgpool2_kernel, dim3(ew_blocks((long)B * H * W * C / 16)), dim3(EW_T), 0, ST, x, out, B, H, W, C,
                     residual, relu_in);
  return check_launch("avgpool2");
}

DIAGAN_API int diagan_avgpool2_bwd(const float* g, float* out, int B, int H, int W, int C, const float* residual,
                                   void* stream) {
  DG_REQUIRE(g && out && B > 0 && H > 1 && W > 1 && (H & 1) == 0 && (W & 1) == 0 && C > 0 && (C & 3) == 0,
             "avgpool2_bwd: bad args");
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(ew_blocks((long)B * H * W * C / 4)), dim3(EW_T), 0, ST, g, out, B, H, W,
                     C, residual);
  return check_launch("avgpool2_bwd");
}

DIAGAN_API int diagan_boxsum2(const float* x, float* out, int B, int H, int W, int C, int relu_in, void* stream) {
  DG_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "boxsum2: bad args");
  DG_REQUIRE((long)B * (H + 1) * (W + 1) * C * 4 < (1L << 31), "boxsum2: output must be smaller than 2 GiB");
  hipLaunchKernelGGL(boxsum2_kernel, dim3(ew_blocks((long)B * (H + 1) * (W + 1) * C / 4)), dim3(EW_T), 0, ST, x, out, B, H, W,
                     C, relu_in);
  return check_launch("boxsum2");
}

DIAGAN_API int diagan_head_fwd(const float* x, const float* w, const float* inv_sigma, const float* inv_sigma1,
                               int split_b, const float* bias, float* pooled, float* logit, int B, int HW, int C,
                               void* stream) {
  DG_REQUIRE(x && w && pooled && logit && B > 0 && HW > 0 && C > 0 && (C & 3) == 0, "head_fwd: bad args");
  const int C4 = C / 4, CW = C4 < 64 ? C4 : 64;
  hipLaunchKernelGGL(relu_sumpool_kernel, dim3(cdiv(C4, CW), B), dim3(EW_T), 0, ST, x, pooled, HW, C);
  hipLaunchKernelGGL(head_linear_kernel, dim3(cdiv(B, 4)), dim3(256), 0, ST, pooled, w, inv_sigma, inv_sigma1,
                     inv_sigma1 ? split_b : B, bias, logit, B, C);
  return check_launch("head_fwd");
}

DIAGAN_API int diagan_head_bwd(const float* dlogit, const float* w, const float* inv_sigma, const float* inv_sigma1,
                               int split_b, const float* x, const float* pooled, float* gx, float* G, double* dot,
                               float* dbias, int accumulate_bias, int B, int HW, int C, void* stream) {
  DG_REQUIRE(dlogit && w && x && B > 0 && HW > 0 && C > 0 && (C & 3) == 0, "head_bwd: bad args");
  if (gx) {
    const long n4 = (long)B * HW * C / 4;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(ew_blocks(n4)), dim3(EW_T), 0, ST, dlogit, w, inv_sigma, inv_sigma1,
                       inv_sigma1 ? split_b : B, x, gx, n4, HW, C);
  }
  if (G) {
    DG_REQUIRE(pooled && dot, "head_bwd: weight gradient needs pooled and dot");
    hipLaunchKernelGGL(head_wgrad_kernel, dim3(1), dim3(EW_T), 0, ST, dlogit, pooled, w, G, dot, dbias, B, C,
                       accumulate_bias);
  }
  return check_launch("head_bwd");
}

DIAGAN_API int diagan_act_fwd(const float* x, const float* scale, const float* shift, float slope, const float* drop,
                              float drop_scale, float* out, int64_t M, int C, void* stream) {
  DG_REQUIRE(x && out && M > 0 && C > 0 && (C & 3) == 0, "act_fwd: bad args");
  DG_REQUIRE(!scale == !shift, "act_fwd: scale and shift must be given together");
  const long n4 = M * (C / 4);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n4)), dim3(EW_T), 0, ST, x, scale, shift, slope, drop, drop_scale, out, n4, C / 4);
  return check_launch("act_fwd");
}

DIAGAN_API int diagan_act_bwd(const float* g, const float* x, float slope, const float* drop, float drop_scale, float* out,
                              int64_t n, void* stream) {
  DG_REQUIRE(g && x && out && n > 0 && (n & 3) == 0, "act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(EW_T), 0, ST, g, x, slope, drop, drop_scale, out, (long)(n / 4));
  return check_launch("act_bwd");
}

DIAGAN_API int diagan_linear1_fwd(const float* x, const float* w, const float* bias, float* logit, int B, int C,
                                  void* stream) {
  DG_REQUIRE(x && w && logit && B > 0 && C > 0, "linear1_fwd: bad args");
  hipLaunchKernelGGL(head_linear_kernel, dim3(cdiv(B, 4)), dim3(256), 0, ST, x, w, (const float*)nullptr,
                     (const float*)nullptr, B, bias, logit, B, C);
  return check_launch("linear1_fwd");
}

DIAGAN_API int diagan_linear1_wgrad(const float* dlogit, const float* x, float* dw, float* dbias, int B, int C,
                                    void* stream) {
  DG_REQUIRE(dlogit && x && dw && B > 0 && C > 0, "linear1_wgrad: bad args");
  hipLaunchKernelGGL(linear1_wgrad_kernel, dim3(cdiv(C, EW_T)), dim3(EW_T), 0, ST, dlogit, x, dw, dbias, B, C);
  return check_launch("linear1_wgrad");
}

DIAGAN_API int diagan_linear1_bwd_input(const float* dlogit, const float* w, float* gx, int B, int C, void* stream) {
  DG_REQUIRE(dlogit && w && gx && B > 0 && C > 0 && (C & 3) == 0, "linear1_bwd_input: bad args");
  const long n4 = (long)B * (C / 4);
  hipLaunchKernelGGL(linear1_bwd_input_kernel, dim3(ew_blocks(n4)), dim3(EW_T), 0, ST, dlogit, w, gx, n4, C / 4);
  return check_launch("linear1_bwd_input");
}

DIAGAN_API int diagan_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  DG_REQUIRE(a && b && out && n > 0 && (n & 3) == 0, "add: bad args");
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n / 4)), dim3(EW_T), 0, ST, a, b, out, (long)(n / 4));
  return check_launch("add");
}
