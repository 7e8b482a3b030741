// The two native ops of the reference's StyleGAN2 path, as gfx950
// HIP kernels behind the same operator semantics:
//
//   fused_bias_act  diagan-pkg/diagan/models/op/fused_bias_act.cpp:4-20, fused_bias_act_kernel.cu:18-49
//       y = act(x + b[(i / step_b) % size_b]) * scale ;  act*10+grad in {10,11: linear, 12: 0,
//       30: leaky ReLU, 31: its derivative gated by ref > 0, 32: 0}
//   upfirdn2d       diagan-pkg/diagan/models/op/upfirdn2d.cpp:4-22, upfirdn2d_kernel.cu:49-207
//       zero-insertion upsample (up), pad / crop, FIR filter with the flipped kernel, decimate (down)
//       on a [major, H, W, minor] tensor; CPU statement of the same op: op/upfirdn2d.py:159-200.
//
// upfirdn2d here is ONE gather formula instead of the reference's six tile specialisations:
//   out[oy][ox] = sum_{iy,ix} in[iy][ix] * k[kh-1-(iy*up_y+pad_y0-oy*down_y)][kw-1-(ix*up_x+pad_x0-ox*down_x)]
// over the (iy, ix) whose kernel index is in range; the filter taps sit in LDS, lanes run along ox
// (then minor) so global reads are coalesced.  Roofline: HBM (in read ~once through L2, out written once).
#include "common.h"
#include <stdlib.h>

namespace diagan {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// channels-last fast path: bias runs along the innermost dimension (step_b == 1), 4 channels per lane
__global__ __launch_bounds__(256) void fused_bias_act_cl4_kernel(const f32x4* __restrict__ x, const float* __restrict__ b,
                                                                 const f32x4* __restrict__ ref, f32x4* __restrict__ out,
                                                                 long n4, int size_b, int mode, float alpha, float scale,
                                                                 const f32x4* __restrict__ addend) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  // size_b % 4 == 0: a lane's channel quad advances by (stride * 4) % size_b per trip
  int c = (int)((i * 4) % size_b);
  const int dc = (int)((stride * 4) % size_b);
  for (; i < n4; i += stride) {
    f32x4 v = x[i];
    if (b) v += *reinterpret_cast<const f32x4*>(b + c);
    f32x4 y;
    if (mode == 30) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = v[e] > 0.f ? v[e] : v[e] * alpha;
    } else if (mode == 31) {
      const f32x4 r = ref[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = r[e] > 0.f ? v[e] : v[e] * alpha;
    } else if (mode == 12 || mode == 32) {
      y = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      y = v;
    }
    f32x4 o;
    {                          // (two roundings, as the activation followed by a separate add: no fused multiply-add)
#pragma clang fp contract(off)
      o = y * scale;
      if (addend) o += addend[i];
    }
    out[i] = o;
    c += dc;
    if (c >= size_b) c -= size_b;
  }
}

// StyledConv tail in one pass (stylegan2.py:323-329 after the activation-side modulated convolution):
//   out[b,p,c] = lrelu( x[b,p,c] * demod[b,c] + strength * noise[b or 0, p] + bias[c] ) * scale
// x [B, P, C] channels-last, 4 channels per lane; demod / noise / bias optional.
__global__ __launch_bounds__(256) void styled_act_cl4_kernel(const f32x4* __restrict__ x, const float* __restrict__ demod,
                                                             const float* __restrict__ noise, const float* __restrict__ strength,
                                                             const float* __restrict__ bias, f32x4* __restrict__ out,
                                                             long n4, int P, int C, int noise_per_image, float alpha,
                                                             float scale, const float* __restrict__ post, f32x4* __restrict__ out2) {
  const int q = C >> 2;
  const float w = (noise && strength) ? strength[0] : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long pix = i / q;                 // b * P + p
    const int c = (int)(i - pix * q) << 2;
    const int b = (int)(pix / P);
    f32x4 v = x[i];
    if (demod) v *= *reinterpret_cast<const f32x4*>(demod + (long)b * C + c);
    if (noise) v += w * noise[noise_per_image ? pix : pix - (long)b * P];
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * alpha) * scale;
    out[i] = v;
    // (round 6) the NEXT layer's modulated input in the same pass: out2 = out * post[b][c]
    if (out2) out2[i] = v * *reinterpret_cast<const f32x4*>(post + (long)b * C + c);
  }
}

// Per-image channel dot products over the pixels: out[b][c] = sum_p a[b][p][c] * b_[b][p][c]  (the gradient of a
// per-(image, channel) scale: style modulation and demodulation of the activation-side modulated convolution).
// Stage 1: one block per (image, pixel chunk); 256 threads = (C/4 channel quads) x (256/(C/4) pixel lanes);
// fp32 partial sums, combined across pixel lanes through LDS in a fixed order.  Stage 2 sums the chunks in double.
__global__ __launch_bounds__(256) void rowdot_partial_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b_,
                                                             float* __restrict__ partial, int P, int C, int chunks) {
  __shared__ f32x4 red[256];
  const int q = C >> 2, lanes = 256 / q;
  const int cq = threadIdx.x % q, pl = threadIdx.x / q;
  const int img = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int per = (P + chunks - 1) / chunks;
  const int p0 = ch * per, p1 = min(p0 + per, P);
  const long base = (long)img * P * q + cq;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int p = p0 + pl;
  for (; p + lanes < p1; p += 2 * lanes) {
    acc0 += a[base + (long)p * q] * b_[base + (long)p * q];
    acc1 += a[base + (long)(p + lanes) * q] * b_[base + (long)(p + lanes) * q];
  }
  if (p < p1) acc0 += a[base + (long)p * q] * b_[base + (long)p * q];
  red[threadIdx.x] = acc0 + acc1;
  __syncthreads();
  if (pl == 0) {
    f32x4 s = red[cq];
    for (int l = 1; l < lanes; ++l) s += red[l * q + cq];
    *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.x * C) + cq * 4) = s;
  }
}

// First-order backward of the StyledConv tail / of bias + leaky ReLU in ONE pass over the incoming gradient (round 4):
//   gpre = gy * scale * (y > 0 ? 1 : alpha)                      (the activation's gate: sign(y) = sign(pre-activation))
//   gx[b,p,c]   = gpre * demod[b,c]          (or gpre)            -> gradient of the convolution output
//   wd[blk][c]  = sum_p gpre * x             (x given)            -> d(demod):  summed over a block's pixels
//   wb[blk][c]  = sum_p gpre                                      -> d(bias) after the sum over blocks
//   ws[blk]     = sum_p noise[p] * sum_c gpre (noise given)       -> d(noise strength)
// instead of the gate kernel + a broadcast multiply + rowdot + (multiply, sum) + sum: five to eight passes over activation-sized
// tensors.  Block / thread layout and partial sums as rowdot_partial_kernel; the caller adds the blocks' rows (tiny).
__global__ __launch_bounds__(256) void styled_act_bwd_kernel(const f32x4* __restrict__ gy, const f32x4* __restrict__ y,
                                                             const f32x4* __restrict__ x, const float* __restrict__ demod,
                                                             const float* __restrict__ noise, f32x4* __restrict__ gx,
                                                             float* __restrict__ wd, float* __restrict__ wb, float* __restrict__ ws,
                                                             int P, int C, int chunks, int noise_per_image, float alpha, float scale,
                                                             const float* __restrict__ ref_bias, const f32x4* __restrict__ g2,
                                                             const float* __restrict__ post, float* __restrict__ wp) {
  __shared__ f32x4 red_d[256];
  __shared__ f32x4 red_b[256];
  __shared__ float red_s[256];
  const int q = C >> 2, lanes = 256 / q;
  const int cq = threadIdx.x % q, pl = threadIdx.x / q;
  const int img = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int per = (P + chunks - 1) / chunks;
  const int p0 = ch * per, p1 = min(p0 + per, P);
  const long base = (long)img * P * q + cq;
  f32x4 dm = {1.f, 1.f, 1.f, 1.f};
  if (demod) dm = *reinterpret_cast<const f32x4*>(demod + (long)img * C + cq * 4);
  f32x4 accd = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
  float accs = 0.f;
  // ref_bias (round 6): `y` is the PRE-activation without its bias (the fused bias + activation + blur / + residual ops keep only
  // their input): the gate is the sign of y + ref_bias[c], the sum the forward took the sign of
  f32x4 rb = {0.f, 0.f, 0.f, 0.f};
  if (ref_bias) rb = *reinterpret_cast<const f32x4*>(ref_bias + cq * 4);
  // g2 / post / wp (round 6): y also left the forward as y * post[b][c] (the next layer's modulated input, diagan_styled_bias_act_mod);
  // g2 is that output's gradient: the incoming gradient is gy (may be null) + g2 * post, and wp[blk][c] = sum_p g2 * y -> d(post)
  f32x4 pm = {0.f, 0.f, 0.f, 0.f}, accp = {0.f, 0.f, 0.f, 0.f};
  if (g2) pm = *reinterpret_cast<const f32x4*>(post + (long)img * C + cq * 4);
  for (int p = p0 + pl; p < p1; p += lanes) {
    const long i = base + (long)p * q;
    f32x4 g = gy ? gy[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 r = y[i];
    if (g2) {
      const f32x4 h = g2[i];
      accp += h * r;
      {                                 // (two roundings, as scale_rows' backward followed by the accumulation: no fused multiply-add)
#pragma clang fp contract(off)
        const f32x4 hm = h * pm;
        g = gy ? g + hm : hm;
      }
    }
    if (ref_bias) r += rb;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = g[e] * (r[e] > 0.f ? scale : scale * alpha);
    if (gx) gx[i] = demod ? g * dm : g;
    if (x) accd += g * x[i];
    accb += g;
    if (noise) accs += noise[noise_per_image ? (long)img * P + p : p] * ((g[0] + g[1]) + (g[2] + g[3]));
  }
  red_d[threadIdx.x] = accd;
  red_b[threadIdx.x] = accb;
  red_s[threadIdx.x] = accs;
  __syncthreads();
  if (pl == 0) {
    f32x4 sd = red_d[cq], sb = red_b[cq];
    for (int l = 1; l < lanes; ++l) { sd += red_d[l * q + cq]; sb += red_b[l * q + cq]; }
    if (wd) *reinterpret_cast<f32x4*>(wd + ((long)blockIdx.x * C) + cq * 4) = sd;
    *reinterpret_cast<f32x4*>(wb + ((long)blockIdx.x * C) + cq * 4) = sb;
  }
  if (wp) {                                         // (block-uniform)
    __syncthreads();
    red_d[threadIdx.x] = accp;
    __syncthreads();
    if (pl == 0) {
      f32x4 sp = red_d[cq];
      for (int l = 1; l < lanes; ++l) sp += red_d[l * q + cq];
      *reinterpret_cast<f32x4*>(wp + ((long)blockIdx.x * C) + cq * 4) = sp;
    }
  }
  if (ws && threadIdx.x < 64) {                    // fixed-order sum of the 256 per-thread scalars
    float t = (red_s[threadIdx.x] + red_s[threadIdx.x + 64]) + (red_s[threadIdx.x + 128] + red_s[threadIdx.x + 192]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (threadIdx.x == 0) ws[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void rowdot_finish_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            int BC, int C, int chunks) {
  const int i = blockIdx.x * 256 + threadIdx.x;       // b * C + c
  if (i >= BC) return;
  const int img = i / C, c = i - img * C;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += (double)partial[((long)img * chunks + k) * C + c];
  out[i] = (float)s;
}

__global__ __launch_bounds__(256) void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                             const float* __restrict__ ref, float* __restrict__ out,
                                                             long n, long step_b, int size_b, int mode, float alpha,
                                                             float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = x[i];
    if (b) v += b[(i / step_b) % size_b];
    float y;
    switch (mode) {
      case 12: case 32: y = 0.f; break;
      case 30: y = v > 0.f ? v : v * alpha; break;
      case 31: y = (ref ? ref[i] : 0.f) > 0.f ? v : v * alpha; break;
      default: y = v; break;   // 10, 11 and anything else: linear
    }
    out[i] = y * scale;
  }
}

struct UpFirDnArgs {
  const float* in;
  const float* k;
  float* out;
  int major, in_h, in_w, minor, kh, kw, out_h, out_w;
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  // fused forms of fir_cl4_kernel (round 6): FUSE 1 -- every input element goes through bias + leaky ReLU * scale on its way in (the
  // discriminator's conv1 -> FusedLeakyReLU -> Blur); FUSE 2 -- every output element goes through the StyledConv tail (demodulation,
  // noise, bias, leaky ReLU * scale) and optionally the NEXT layer's style on its way out (the generator's up-sampling convolution ->
  // Blur -> NoiseInjection -> FusedLeakyReLU [-> modulation] when no graph is recorded)
  const float* f_bias;     // [minor]
  const float* f_demod;    // [major][minor] or null
  const float* f_noise;    // [major or 1][out_h][out_w] or null
  const float* f_strength; // [1]
  const float* f_post;     // [major][minor] or null
  // FUSE 3 -- the adjoint blur of a gradient followed by the gate of bias + leaky ReLU (first-order backward of FUSE 1): every output
  // element is multiplied by scale * (f_ref + f_bias[c] > 0 ? 1 : alpha); the lanes' sums of what they wrote go to f_work[block][minor]
  const float* f_ref;      // [major][out_h][out_w][minor]: the pre-activation without its bias
  float* f_work;           // [gridDim.x][minor]
  int f_noise_per_image;
  float f_alpha, f_scale;
};

// DIAGAN_FIR_ROWS=0: the one-output-row-per-lane form of the blur kernels (A/B; default: four rows per lane)
static bool fir_rows() {
  static const int env = getenv("DIAGAN_FIR_ROWS") ? atoi(getenv("DIAGAN_FIR_ROWS")) : 1;
  return env != 0;
}

static __host__ __device__ __forceinline__ int floor_div_i(int a, int b) {
  int q = a / b;
  return (q * b > a) ? q - 1 : q;
}
static __host__ __device__ __forceinline__ int ceil_div_i(int a, int b) { return -floor_div_i(-a, b); }

// FIR without resampling (up = down = 1: StyleGAN2's Blur and its adjoint) on channels-last data with minor % 4 == 0:
// a lane owns 4 channels x OXT consecutive output columns and slides a (KW + OXT - 1)-wide register window down the
// kh filter rows, so that every input element is fetched (KW + OXT - 1) / OXT times instead of KW times and as 16-byte
// loads; lanes run along the channel quads, then along the column groups (coalesced 512 B+ segments).
// DOWN = 2 (round 6): the same with every second output row / column kept -- blur + stride-2 sub-sampling in one pass (the
// discriminator's skip branch: Blur, then a 1x1 convolution that reads every second pixel; reference stylegan2.py:553-614): the
// window is KW + (OXT - 1) * DOWN wide and a quarter of the blurred image is ever written.
template <int KW, int OXT, int DOWN = 1, int FUSE = 0>
__global__ __launch_bounds__(256) void fir_cl4_kernel(const UpFirDnArgs a) {
  __shared__ float taps[64];
  if (threadIdx.x < a.kh * KW) taps[threadIdx.x] = a.k[threadIdx.x];
  __syncthreads();
  const int q = a.minor >> 2, gx = (a.out_w + OXT - 1) / OXT;
  const long total = (long)a.major * a.out_h * gx * q;
  const f32x4* __restrict__ in4 = reinterpret_cast<const f32x4*>(a.in);
  f32x4* __restrict__ out4 = reinterpret_cast<f32x4*>(a.out);
  const bool slope01 = a.f_alpha > 0.f && a.f_alpha < 1.f;
  f32x4 gsum = {0.f, 0.f, 0.f, 0.f};          // FUSE 3: this lane's channel quad is the same on every trip (q divides the stride)
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
    const int c4 = (int)(o % q);
    long t = o / q;
    const int ox0 = (int)(t % gx) * OXT; t /= gx;
    const int oy = (int)(t % a.out_h);
    const int mj = (int)(t / a.out_h);
    f32x4 acc[OXT];
#pragma unroll
    for (int j = 0; j < OXT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bx = ox0 * DOWN - a.pad_x0;
    f32x4 pb = {0.f, 0.f, 0.f, 0.f};
    if (FUSE == 1 || FUSE == 3) pb = *reinterpret_cast<const f32x4*>(a.f_bias + c4 * 4);
    for (int dy = 0; dy < a.kh; ++dy) {
      const int iy = oy * DOWN - a.pad_y0 + dy;
      if (iy < 0 || iy >= a.in_h) continue;
      const f32x4* row = in4 + ((long)mj * a.in_h + iy) * a.in_w * q + c4;
      f32x4 win[KW + (OXT - 1) * DOWN];
#pragma unroll
      for (int u = 0; u < KW + (OXT - 1) * DOWN; ++u) {
        const int ix = bx + u;
        win[u] = (ix >= 0 && ix < a.in_w) ? row[(long)ix * q] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (FUSE == 1) {          // (a loop of its own, branch-free: the window's loads above stay back to back; the padding is zeros of the
                                //  ACTIVATED tensor; values of fused_bias_act_cl4_kernel)
#pragma unroll
        for (int u = 0; u < KW + (OXT - 1) * DOWN; ++u) {
          const int ix = bx + u;
          f32x4 v = win[u] + pb;
          if (slope01) {        // 0 < slope < 1: v > 0 ? v : v * slope == max(v, v * slope)
            const f32x4 vs = v * a.f_alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], vs[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.f_alpha;
          }
          v = v * a.f_scale;
          const bool in = ix >= 0 && ix < a.in_w;
#pragma unroll
          for (int e = 0; e < 4; ++e) win[u][e] = in ? v[e] : 0.f;
        }
      }
      const float* kr = taps + (a.kh - 1 - dy) * KW;
#pragma unroll
      for (int dx = 0; dx < KW; ++dx) {
        const float w = kr[KW - 1 - dx];
#pragma unroll
        for (int j = 0; j < OXT; ++j) acc[j] += win[j * DOWN + dx] * w;
      }
    }
    f32x4* dst = out4 + (((long)mj * a.out_h + oy) * a.out_w + ox0) * q + c4;
    if (FUSE == 2) {                                    // (arithmetic of styled_act_cl4_kernel, then scale_rows')
      const float w = (a.f_noise && a.f_strength) ? a.f_strength[0] : 0.f;
#pragma unroll
      for (int j = 0; j < OXT; ++j) {
        if (ox0 + j >= a.out_w) continue;
        f32x4 v = acc[j];
        if (a.f_demod) v *= *reinterpret_cast<const f32x4*>(a.f_demod + (long)mj * a.minor + c4 * 4);
        if (a.f_noise) v += w * a.f_noise[((long)(a.f_noise_per_image ? mj : 0) * a.out_h + oy) * a.out_w + ox0 + j];
        if (a.f_bias) v += *reinterpret_cast<const f32x4*>(a.f_bias + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * a.f_alpha) * a.f_scale;
        if (a.f_post) v = v * *reinterpret_cast<const f32x4*>(a.f_post + (long)mj * a.minor + c4 * 4);
        dst[(long)j * q] = v;
      }
      continue;
    }
    if (FUSE == 3) {
      const f32x4* ref = reinterpret_cast<const f32x4*>(a.f_ref) + (((long)mj * a.out_h + oy) * a.out_w + ox0) * q + c4;
#pragma unroll
      for (int j = 0; j < OXT; ++j) {
        if (ox0 + j >= a.out_w) continue;
        const f32x4 r = ref[(long)j * q] + pb;
        f32x4 v = acc[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * (r[e] > 0.f ? a.f_scale : a.f_scale * a.f_alpha);     // (styled_act_bwd_kernel's gate)
        dst[(long)j * q] = v;
        gsum += v;
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < OXT; ++j)
      if (ox0 + j < a.out_w) dst[(long)j * q] = acc[j];
  }
  if (FUSE == 3) {                              // the lanes of a channel quad, in a fixed order, to f_work[block][c]
    __shared__ f32x4 red[256];
    red[threadIdx.x] = gsum;
    __syncthreads();
    if ((int)threadIdx.x < q) {
      f32x4 sum = red[threadIdx.x];
      for (int l = threadIdx.x + q; l < 256; l += q) sum += red[l];
      *reinterpret_cast<f32x4*>(a.f_work + (long)blockIdx.x * a.minor + threadIdx.x * 4) = sum;
    }
  }
}

// The same filter (no resampling) with OYT output ROWS per lane as well: every loaded window row serves the up to min(kh, OYT) output
// rows it reaches, so an input element is fetched (KW + OXT - 1)(kh + OYT - 1) / (OXT * OYT) = 3.06 times (4 x 4 outputs, 4 x 4 taps) instead
// of 7 -- the one-row form above runs at 3.5 TB/s of its bytes where a plain elementwise pass reaches 4.7 (tools/probe/fir_fused_time.py):
// the re-reads come from L1 / L2 / the memory-side cache, but they are not free.  Same order of additions per output (tap rows ascending,
// taps ascending within a row): bit-identical results.  FUSE as above.
template <int KW, int OXT, int OYT, int FUSE>
__global__ __launch_bounds__(256) void fir_cl4_rows_kernel(const UpFirDnArgs a) {
  __shared__ float taps[64];
  if (threadIdx.x < a.kh * KW) taps[threadIdx.x] = a.k[threadIdx.x];
  __syncthreads();
  const int q = a.minor >> 2, gx = (a.out_w + OXT - 1) / OXT, gy = (a.out_h + OYT - 1) / OYT;
  const long total = (long)a.major * gy * gx * q;
  const f32x4* __restrict__ in4 = reinterpret_cast<const f32x4*>(a.in);
  f32x4* __restrict__ out4 = reinterpret_cast<f32x4*>(a.out);
  const bool slope01 = a.f_alpha > 0.f && a.f_alpha < 1.f;
  f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
    const int c4 = (int)(o % q);
    long t = o / q;
    const int ox0 = (int)(t % gx) * OXT; t /= gx;
    const int oy0 = (int)(t % gy) * OYT;
    const int mj = (int)(t / gy);
    f32x4 acc[OYT][OXT];
#pragma unroll
    for (int r = 0; r < OYT; ++r)
#pragma unroll
      for (int j = 0; j < OXT; ++j) acc[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bx = ox0 - a.pad_x0;
    f32x4 pb = {0.f, 0.f, 0.f, 0.f};
    if (FUSE == 1 || FUSE == 3) pb = *reinterpret_cast<const f32x4*>(a.f_bias + c4 * 4);
    for (int ry = 0; ry < OYT + a.kh - 1; ++ry) {
      const int iy = oy0 - a.pad_y0 + ry;
      if (iy < 0 || iy >= a.in_h) continue;
      const f32x4* row = in4 + ((long)mj * a.in_h + iy) * a.in_w * q + c4;
      f32x4 win[KW + OXT - 1];
#pragma unroll
      for (int u = 0; u < KW + OXT - 1; ++u) {
        const int ix = bx + u;
        win[u] = (ix >= 0 && ix < a.in_w) ? row[(long)ix * q] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (FUSE == 1) {
#pragma unroll
        for (int u = 0; u < KW + OXT - 1; ++u) {
          const int ix = bx + u;
          f32x4 v = win[u] + pb;
          if (slope01) {
            const f32x4 vs = v * a.f_alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], vs[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.f_alpha;
          }
          v = v * a.f_scale;
          const bool in = ix >= 0 && ix < a.in_w;
#pragma unroll
          for (int e = 0; e < 4; ++e) win[u][e] = in ? v[e] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < OYT; ++r) {
        const int dy = ry - r;                       // (block-uniform)
        if (dy < 0 || dy >= a.kh) continue;
        const float* kr = taps + (a.kh - 1 - dy) * KW;
#pragma unroll
        for (int dx = 0; dx < KW; ++dx) {
          const float w = kr[KW - 1 - dx];
#pragma unroll
          for (int j = 0; j < OXT; ++j) acc[r][j] += win[j + dx] * w;
        }
      }
    }
    const float nw = (FUSE == 2 && a.f_noise && a.f_strength) ? a.f_strength[0] : 0.f;
#pragma unroll
    for (int r = 0; r < OYT; ++r) {
      const int oy = oy0 + r;
      if (oy >= a.out_h) continue;
      const long pix0 = ((long)mj * a.out_h + oy) * a.out_w + ox0;
      f32x4* dst = out4 + pix0 * q + c4;
#pragma unroll
      for (int j = 0; j < OXT; ++j) {
        if (ox0 + j >= a.out_w) continue;
        f32x4 v = acc[r][j];
        if (FUSE == 2) {                             // (arithmetic of styled_act_cl4_kernel, then scale_rows')
          if (a.f_demod) v *= *reinterpret_cast<const f32x4*>(a.f_demod + (long)mj * a.minor + c4 * 4);
          if (a.f_noise) v += nw * a.f_noise[((long)(a.f_noise_per_image ? mj : 0) * a.out_h + oy) * a.out_w + ox0 + j];
          if (a.f_bias) v += *reinterpret_cast<const f32x4*>(a.f_bias + c4 * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * a.f_alpha) * a.f_scale;
          if (a.f_post) v = v * *reinterpret_cast<const f32x4*>(a.f_post + (long)mj * a.minor + c4 * 4);
        }
        if (FUSE == 3) {                             // (styled_act_bwd_kernel's gate)
          const f32x4 rf = reinterpret_cast<const f32x4*>(a.f_ref)[(pix0 + j) * q + c4] + pb;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * (rf[e] > 0.f ? a.f_scale : a.f_scale * a.f_alpha);
          gsum += v;
        }
        dst[(long)j * q] = v;
      }
    }
  }
  if (FUSE == 3) {                              // the lanes of a channel quad, in a fixed order, to f_work[block][c]
    __shared__ f32x4 red[256];
    red[threadIdx.x] = gsum;
    __syncthreads();
    if ((int)threadIdx.x < q) {
      f32x4 sum = red[threadIdx.x];
      for (int l = threadIdx.x + q; l < 256; l += q) sum += red[l];
      *reinterpret_cast<f32x4*>(a.f_work + (long)blockIdx.x * a.minor + threadIdx.x * 4) = sum;
    }
  }
}

// Any up / down factors on channels-last data with minor % 4 == 0 (round 6: the adjoint of the sub-sampling blur above is an
// up = 2 filter, the generator's RGB Upsample is one too): a lane owns one output pixel x 4 channels and walks the (at most
// ceil(kh / up) x ceil(kw / up)) input pixels that reach it, 16-byte loads, lanes along the channel quads.
__global__ __launch_bounds__(256) void updn_cl4_kernel(const UpFirDnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float taps[];
  for (int i = threadIdx.x; i < a.kh * a.kw; i += 256) taps[i] = a.k[i];
  __syncthreads();
  const int q = a.minor >> 2;
  const long total = (long)a.major * a.out_h * a.out_w * q;
  const f32x4* __restrict__ in4 = reinterpret_cast<const f32x4*>(a.in);
  f32x4* __restrict__ out4 = reinterpret_cast<f32x4*>(a.out);
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
    const int c4 = (int)(o % q);
    long t = o / q;
    const int ox = (int)(t % a.out_w); t /= a.out_w;
    const int oy = (int)(t % a.out_h);
    const int mj = (int)(t / a.out_h);
    const int by = oy * a.down_y - a.pad_y0, bx = ox * a.down_x - a.pad_x0;
    const int iy0 = max(ceil_div_i(by, a.up_y), 0), iy1 = min(floor_div_i(by + a.kh - 1, a.up_y), a.in_h - 1);
    const int ix0 = max(ceil_div_i(bx, a.up_x), 0), ix1 = min(floor_div_i(bx + a.kw - 1, a.up_x), a.in_w - 1);
    const f32x4* src = in4 + (long)mj * a.in_h * a.in_w * q + c4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int iy = iy0; iy <= iy1; ++iy) {
      const int ky = a.kh - 1 - (iy * a.up_y - by);
      for (int ix = ix0; ix <= ix1; ++ix) {
        const int kx = a.kw - 1 - (ix * a.up_x - bx);
        const f32x4 xin = src[((long)iy * a.in_w + ix) * q];
        const float w = taps[ky * a.kw + kx];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(xin[e], w, v[e]);
      }
    }
    if (a.f_ref) v += reinterpret_cast<const f32x4*>(a.f_ref)[o];     // (round 6: the other branch's gradient, see diagan_upfirdn2d_add)
    out4[o] = v;
  }
}

__global__ __launch_bounds__(256) void upfirdn2d_kernel(const UpFirDnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float taps[];
  for (int i = threadIdx.x; i < a.kh * a.kw; i += 256) taps[i] = a.k[i];
  __syncthreads();
  const long per_major = (long)a.out_h * a.out_w * a.minor;
  const long total = per_major * a.major;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
    const int mi = (int)(o % a.minor);
    long t = o / a.minor;
    const int ox = (int)(t % a.out_w); t /= a.out_w;
    const int oy = (int)(t % a.out_h);
    const int mj = (int)(t / a.out_h);
    // valid input rows: 0 <= iy*up + pad0 - oy*down < kh
    const int by = oy * a.down_y - a.pad_y0, bx = ox * a.down_x - a.pad_x0;
    const int iy0 = max(ceil_div_i(by, a.up_y), 0), iy1 = min(floor_div_i(by + a.kh - 1, a.up_y), a.in_h - 1);
    const int ix0 = max(ceil_div_i(bx, a.up_x), 0), ix1 = min(floor_div_i(bx + a.kw - 1, a.up_x), a.in_w - 1);
    const float* src = a.in + (long)mj * a.in_h * a.in_w * a.minor + mi;
    float v = 0.f;
    for (int iy = iy0; iy <= iy1; ++iy) {
      const int ky = a.kh - 1 - (iy * a.up_y - by);
      for (int ix = ix0; ix <= ix1; ++ix) {
        const int kx = a.kw - 1 - (ix * a.up_x - bx);
        v = fmaf(src[((long)iy * a.in_w + ix) * a.minor], taps[ky * a.kw + kx], v);
      }
    }
    a.out[o] = v;
  }
}

}  // namespace diagan

using namespace diagan;

DIAGAN_API int diagan_fused_bias_act(const float* x, const float* bias, const float* refer, float* out, int64_t n,
                                     int64_t step_b, int size_b, int act, int grad, float alpha, float scale,
                                     void* stream) {
  DG_REQUIRE(x && out && n >= 0, "fused_bias_act: null tensor");
  DG_REQUIRE(!bias || (step_b > 0 && size_b > 0), "fused_bias_act: bad bias geometry");
  if (n == 0) return DIAGAN_OK;
  if ((!bias || (step_b == 1 && (size_b & 3) == 0)) && (n & 3) == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)refer |
                                                                      (uintptr_t)bias) & 15) == 0) {
    const long n4 = n / 4;
    long blocks4 = (n4 + 255) / 256;
    if (blocks4 > 8192) blocks4 = 8192;
    hipLaunchKernelGGL(fused_bias_act_cl4_kernel, dim3((int)blocks4), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)x, bias, (const f32x4*)refer, (f32x4*)out, n4, bias ? size_b : 4, act * 10 + grad,
                       alpha, scale, (const f32x4*)nullptr);
    return check_launch("fused_bias_act");
  }
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, refer, out,
                     (long)n, (long)step_b, size_b, act * 10 + grad, alpha, scale);
  return check_launch("fused_bias_act");
}

// see include/diagan_hip.h: out = leaky_relu(x + bias[c]) * scale + addend on channels-last data in ONE pass
DIAGAN_API int diagan_bias_act_add(const float* x, const float* bias, const float* addend, float* out, int64_t n, int C, float alpha,
                                   float scale, void* stream) {
  DG_REQUIRE(x && bias && addend && out && n > 0 && C > 0 && (C & 3) == 0 && n % C == 0, "bias_act_add: bad args (C must be a multiple of 4)");
  DG_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)addend | (uintptr_t)bias) & 15) == 0, "bias_act_add: pointers must be 16-byte aligned");
  const long n4 = n / 4;
  long blocks4 = (n4 + 255) / 256;
  if (blocks4 > 8192) blocks4 = 8192;
  hipLaunchKernelGGL(fused_bias_act_cl4_kernel, dim3((int)blocks4), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, bias,
                     (const f32x4*)nullptr, (f32x4*)out, n4, C, 30, alpha, scale, (const f32x4*)addend);
  return check_launch("bias_act_add");
}

DIAGAN_API int diagan_styled_bias_act(const float* x, const float* demod, const float* noise, const float* strength,
                                      const float* bias, float* out, int B, int P, int C, int noise_per_image,
                                      float alpha, float scale, void* stream) {
  DG_REQUIRE(x && out && B > 0 && P > 0 && C > 0 && (C & 3) == 0, "styled_bias_act: bad dims (C must be a multiple of 4)");
  DG_REQUIRE(!noise || strength, "styled_bias_act: noise needs its strength");
  DG_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)demod | (uintptr_t)bias) & 15) == 0,
             "styled_bias_act: pointers must be 16-byte aligned");
  const long n4 = (long)B * P * (C / 4);
  long blocks = (n4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(styled_act_cl4_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, demod,
                     noise, strength, bias, (f32x4*)out, n4, P, C, noise_per_image, alpha, scale, (const float*)nullptr, (f32x4*)nullptr);
  return check_launch("styled_bias_act");
}

// see include/diagan_hip.h: the tail and the next layer's modulated input in one pass
DIAGAN_API int diagan_styled_bias_act_mod(const float* x, const float* demod, const float* noise, const float* strength, const float* bias,
                                          const float* post, float* out, float* out_mod, int B, int P, int C, int noise_per_image,
                                          float alpha, float scale, void* stream) {
  DG_REQUIRE(x && out && post && out_mod && B > 0 && P > 0 && C > 0 && (C & 3) == 0, "styled_bias_act_mod: bad args (C must be a multiple of 4)");
  DG_REQUIRE(!noise || strength, "styled_bias_act_mod: noise needs its strength");
  DG_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)out_mod | (uintptr_t)demod | (uintptr_t)bias | (uintptr_t)post) & 15) == 0,
             "styled_bias_act_mod: pointers must be 16-byte aligned");
  const long n4 = (long)B * P * (C / 4);
  long blocks = (n4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(styled_act_cl4_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, demod,
                     noise, strength, bias, (f32x4*)out, n4, P, C, noise_per_image, alpha, scale, post, (f32x4*)out_mod);
  return check_launch("styled_bias_act_mod");
}

DIAGAN_API int diagan_rowdot_chunks(int B, int P) {
  int chunks = 2048 / (B > 0 ? B : 1);                 // ~2048 blocks: 8 per CU
  if (chunks > P / 64) chunks = P / 64;
  return chunks < 1 ? 1 : chunks;
}

DIAGAN_API int diagan_rowdot(const float* a, const float* b, float* out, float* workspace, int B, int P, int C,
                             void* stream) {
  DG_REQUIRE(a && b && out && workspace && B > 0 && P > 0, "rowdot: bad args");
  DG_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "rowdot: C=%d must be a power of two in [4, 1024]", C);
  DG_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)workspace) & 15) == 0, "rowdot: pointers must be 16-byte aligned");
  const int chunks = diagan_rowdot_chunks(B, P);
  hipLaunchKernelGGL(rowdot_partial_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)a,
                     (const f32x4*)b, workspace, P, C, chunks);
  hipLaunchKernelGGL(rowdot_finish_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, workspace, out,
                     B * C, C, chunks);
  return check_launch("rowdot");
}

// see include/diagan_hip.h: first-order backward of diagan_styled_bias_act / of bias + leaky ReLU in one pass
DIAGAN_API int diagan_styled_bias_act_bwd(const float* gy, const float* y, const float* x, const float* demod, const float* noise,
                                          float* gx, float* work_d, float* work_b, float* work_s, int B, int P, int C,
                                          int noise_per_image, float alpha, float scale, void* stream) {
  DG_REQUIRE(gy && y && work_b && B > 0 && P > 0, "styled_bias_act_bwd: bad args");
  DG_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "styled_bias_act_bwd: C=%d must be a power of two in [4, 1024]", C);
  DG_REQUIRE(!x == !work_d && !noise == !work_s, "styled_bias_act_bwd: x / work_d and noise / work_s come in pairs");
  DG_REQUIRE((((uintptr_t)gy | (uintptr_t)y | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)demod | (uintptr_t)work_d | (uintptr_t)work_b) & 15) == 0,
             "styled_bias_act_bwd: pointers must be 16-byte aligned");
  const int chunks = diagan_rowdot_chunks(B, P);
  hipLaunchKernelGGL(styled_act_bwd_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)gy, (const f32x4*)y,
                     (const f32x4*)x, demod, noise, (f32x4*)gx, work_d, work_b, work_s, P, C, chunks, noise_per_image, alpha, scale,
                     (const float*)nullptr, (const f32x4*)nullptr, (const float*)nullptr, (float*)nullptr);
  return check_launch("styled_bias_act_bwd");
}

// see include/diagan_hip.h: first-order backward of diagan_styled_bias_act_mod in one pass
DIAGAN_API int diagan_styled_bias_act_mod_bwd(const float* gy, const float* gmod, const float* post, const float* y, const float* x,
                                              const float* demod, const float* noise, float* gx, float* work_d, float* work_b, float* work_s,
                                              float* work_p, int B, int P, int C, int noise_per_image, float alpha, float scale, void* stream) {
  DG_REQUIRE(gmod && post && y && work_b && work_p && B > 0 && P > 0, "styled_bias_act_mod_bwd: bad args");
  DG_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "styled_bias_act_mod_bwd: C=%d must be a power of two in [4, 1024]", C);
  DG_REQUIRE(!x == !work_d && !noise == !work_s, "styled_bias_act_mod_bwd: x / work_d and noise / work_s come in pairs");
  DG_REQUIRE((((uintptr_t)gy | (uintptr_t)gmod | (uintptr_t)y | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)demod | (uintptr_t)post |
               (uintptr_t)work_d | (uintptr_t)work_b | (uintptr_t)work_p) & 15) == 0, "styled_bias_act_mod_bwd: pointers must be 16-byte aligned");
  const int chunks = diagan_rowdot_chunks(B, P);
  hipLaunchKernelGGL(styled_act_bwd_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)gy, (const f32x4*)y,
                     (const f32x4*)x, demod, noise, (f32x4*)gx, work_d, work_b, work_s, P, C, chunks, noise_per_image, alpha, scale,
                     (const float*)nullptr, (const f32x4*)gmod, post, work_p);
  return check_launch("styled_bias_act_mod_bwd");
}

// see include/diagan_hip.h: the gate of bias + leaky ReLU from the PRE-activation z and the bias (sign of z + bias), bias gradient partials
DIAGAN_API int diagan_bias_act_gate_bwd(const float* gy, const float* z, const float* bias, float* gx, float* work_b, int B, int P, int C,
                                        float alpha, float scale, void* stream) {
  DG_REQUIRE(gy && z && bias && gx && work_b && B > 0 && P > 0, "bias_act_gate_bwd: bad args");
  DG_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "bias_act_gate_bwd: C=%d must be a power of two in [4, 1024]", C);
  DG_REQUIRE((((uintptr_t)gy | (uintptr_t)z | (uintptr_t)gx | (uintptr_t)bias | (uintptr_t)work_b) & 15) == 0,
             "bias_act_gate_bwd: pointers must be 16-byte aligned");
  const int chunks = diagan_rowdot_chunks(B, P);
  hipLaunchKernelGGL(styled_act_bwd_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)gy, (const f32x4*)z,
                     (const f32x4*)nullptr, (const float*)nullptr, (const float*)nullptr, (f32x4*)gx, (float*)nullptr, work_b,
                     (float*)nullptr, P, C, chunks, 0, alpha, scale, bias, (const f32x4*)nullptr, (const float*)nullptr, (float*)nullptr);
  return check_launch("bias_act_gate_bwd");
}

// The partial sums of diagan_styled_bias_act_bwd to their results in ONE launch (round 6; the host side summed them with torch: a
// float64 conversion, a reduction and a conversion back per result = nine 4-10 us launches per call, ~540 per StyleGAN2 iteration):
//   gd[b][c] = sum_chunks work_d[b][chunk][c],  gb[c] = sum_rows work_b[row][c] (rows = B * chunks),  gs = sum work_s[rows]
// accumulated in double in a fixed order (rows interleaved four ways per column, then a fixed LDS tree): deterministic.
namespace diagan {
// 1024 threads = 64 columns x 16 lanes.  Blocks [0, nd): (image b, 64-column group) -> gd, the lanes take the chunks; blocks [nd, nd + nb):
// 64-column group -> gb, the lanes take the B * chunks rows (independent loads, eight in flight); the last block -> gs.  Every sum in
// double: a lane's rows in order, then the sixteen lanes in order (deterministic).
__global__ __launch_bounds__(1024) void styled_act_bwd_finish_kernel(const float* __restrict__ work_d, const float* __restrict__ work_b,
                                                                     const float* __restrict__ work_s, float* __restrict__ gd,
                                                                     float* __restrict__ gb, float* __restrict__ gs, int B, int chunks,
                                                                     int C, int nd, int nb) {
  __shared__ double red[1024];
  const int bid = blockIdx.x, tid = threadIdx.x, cl = tid & 63, lane = tid >> 6;
  const int groups = (C + 63) >> 6;
  double t = 0.0;
  int col = -1;
  float* dst = nullptr;
  if (bid < nd) {
    const int b = bid / groups;
    col = (bid - b * groups) * 64 + cl;
    if (col < C) {
      const float* src = work_d + (long)b * chunks * C + col;
      for (int k = lane; k < chunks; k += 16) t += (double)src[(long)k * C];
      dst = gd + (long)b * C + col;
    }
  } else if (bid < nd + nb) {
    col = (bid - nd) * 64 + cl;
    const int rows = B * chunks;
    if (col < C) {
      const float* src = work_b + col;
      int r = lane;
      for (; r + 7 * 16 < rows; r += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(long)(r + 16 * u) * C];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += (double)v[u];
      }
      for (; r < rows; r += 16) t += (double)src[(long)r * C];
      dst = gb + col;
    }
  } else {
    const int rows = B * chunks;
    for (int r = tid; r < rows; r += 1024) t += (double)work_s[r];
    red[tid] = t;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) gs[0] = (float)red[0];
    return;
  }
  red[tid] = t;
  __syncthreads();
  if (lane == 0 && dst) {
    double a = 0.0;
#pragma unroll
    for (int l = 0; l < 16; ++l) a += red[l * 64 + cl];
    *dst = (float)a;
  }
}
}  // namespace diagan

DIAGAN_API int diagan_styled_bias_act_bwd_finish(const float* work_d, const float* work_b, const float* work_s, float* gd, float* gb,
                                                 float* gs, int B, int P, int C, void* stream) {
  DG_REQUIRE(B > 0 && P > 0 && C > 0, "styled_bias_act_bwd_finish: bad dims");
  DG_REQUIRE(!work_d == !gd && !work_b == !gb && !work_s == !gs && (gd || gb || gs), "styled_bias_act_bwd_finish: partials and results come in pairs");
  const int chunks = diagan_rowdot_chunks(B, P);
  const int groups = cdiv(C, 64);
  const int nd = gd ? B * groups : 0, nb = gb ? groups : 0, ns = gs ? 1 : 0;
  hipLaunchKernelGGL(diagan::styled_act_bwd_finish_kernel, dim3(nd + nb + ns), dim3(1024), 0, (hipStream_t)stream, work_d, work_b, work_s,
                     gd, gb, gs, B, chunks, C, nd, nb);
  return check_launch("styled_bias_act_bwd_finish");
}

// see include/diagan_hip.h: blur(leaky_relu(x + bias) * scale) in ONE pass over x (FUSE 1 of fir_cl4_kernel)
DIAGAN_API int diagan_bias_act_fir(const float* input, const float* bias, const float* kernel, float* out, int major, int in_h, int in_w,
                                   int minor, int kernel_h, int kernel_w, int pad_x0, int pad_x1, int pad_y0, int pad_y1, float alpha,
                                   float scale, void* stream) {
  DG_REQUIRE(input && bias && kernel && out && major > 0 && in_h > 0 && in_w > 0 && minor > 0, "bias_act_fir: bad args");
  DG_REQUIRE(kernel_w == 4 && kernel_h >= 1 && kernel_h <= 16 && (minor & 3) == 0, "bias_act_fir: 4-tap-wide filters on channel counts that are multiples of 4 (got %d x %d taps, %d channels)", kernel_h, kernel_w, minor);
  DG_REQUIRE((((uintptr_t)input | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, "bias_act_fir: pointers must be 16-byte aligned");
  const int oh = in_h + pad_y0 + pad_y1 - kernel_h + 1, ow = in_w + pad_x0 + pad_x1 - kernel_w + 1;
  DG_REQUIRE(oh > 0 && ow > 0, "bias_act_fir: empty output (%d x %d)", oh, ow);
  UpFirDnArgs a{input, kernel, out, major, in_h, in_w, minor, kernel_h, kernel_w, oh, ow, 1, 1, 1, 1, pad_x0, pad_y0};
  a.f_bias = bias; a.f_alpha = alpha; a.f_scale = scale;
  if (fir_rows()) {
    const long work4 = (long)major * ((oh + 3) / 4) * ((ow + 3) / 4) * (minor / 4);
    long fb4 = (work4 + 255) / 256;
    if (fb4 > 32768) fb4 = 32768;
    hipLaunchKernelGGL((fir_cl4_rows_kernel<4, 4, 4, 1>), dim3((int)fb4), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("bias_act_fir");
  }
  const long work = (long)major * oh * ((ow + 3) / 4) * (minor / 4);
  long fb = (work + 255) / 256;
  if (fb > 32768) fb = 32768;
  hipLaunchKernelGGL((fir_cl4_kernel<4, 4, 1, 1>), dim3((int)fb), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("bias_act_fir");
}

// see include/diagan_hip.h: out = upfirdn2d(input) + addend in one pass (channels-last, minor % 4 == 0)
DIAGAN_API int diagan_upfirdn2d_add(const float* input, const float* kernel, const float* addend, float* out, int major, int in_h, int in_w,
                                    int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                    int pad_x1, int pad_y0, int pad_y1, void* stream) {
  DG_REQUIRE(input && kernel && addend && out && major > 0 && in_h > 0 && in_w > 0 && minor > 0 && (minor & 3) == 0, "upfirdn2d_add: bad args");
  DG_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0 && kernel_h * kernel_w <= 4096, "upfirdn2d_add: bad filter geometry");
  DG_REQUIRE((((uintptr_t)input | (uintptr_t)out | (uintptr_t)addend) & 15) == 0, "upfirdn2d_add: pointers must be 16-byte aligned");
  const int oh = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) / down_y + 1;
  const int ow = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) / down_x + 1;
  DG_REQUIRE(oh > 0 && ow > 0, "upfirdn2d_add: empty output (%d x %d)", oh, ow);
  UpFirDnArgs a{input, kernel, out, major, in_h, in_w, minor, kernel_h, kernel_w, oh, ow, up_x, up_y, down_x, down_y, pad_x0, pad_y0};
  a.f_ref = addend;
  const long work = (long)major * oh * ow * (minor / 4);
  long fb = (work + 255) / 256;
  if (fb > 32768) fb = 32768;
  hipLaunchKernelGGL(updn_cl4_kernel, dim3((int)fb), dim3(256), (size_t)kernel_h * kernel_w * sizeof(float), (hipStream_t)stream, a);
  return check_launch("upfirdn2d_add");
}

// see include/diagan_hip.h: gz = gate(ref + bias) * FIR(g) and the bias gradient's partial sums in ONE pass (FUSE 3): first-order
// backward of diagan_bias_act_fir with the ADJOINT filter geometry (flipped taps, adjoint pads) handed in by the caller
DIAGAN_API int diagan_fir_gate_bwd(const float* g, const float* kernel, const float* ref, const float* bias, float* gz, float* work_b,
                                   int major, int in_h, int in_w, int minor, int kernel_h, int kernel_w, int pad_x0, int pad_x1, int pad_y0,
                                   int pad_y1, float alpha, float scale, void* stream) {
  DG_REQUIRE(g && kernel && ref && bias && gz && work_b && major > 0 && in_h > 0 && in_w > 0, "fir_gate_bwd: bad args");
  DG_REQUIRE(kernel_w == 4 && kernel_h >= 1 && kernel_h <= 16, "fir_gate_bwd: 4-tap-wide filters (got %d x %d taps)", kernel_h, kernel_w);
  DG_REQUIRE(minor >= 4 && minor <= 1024 && (minor & (minor - 1)) == 0, "fir_gate_bwd: %d channels: a power of two in [4, 1024]", minor);
  DG_REQUIRE((((uintptr_t)g | (uintptr_t)gz | (uintptr_t)ref | (uintptr_t)bias | (uintptr_t)work_b) & 15) == 0, "fir_gate_bwd: pointers must be 16-byte aligned");
  const int oh = in_h + pad_y0 + pad_y1 - kernel_h + 1, ow = in_w + pad_x0 + pad_x1 - kernel_w + 1;
  DG_REQUIRE(oh > 0 && ow > 0, "fir_gate_bwd: empty output (%d x %d)", oh, ow);
  UpFirDnArgs a{g, kernel, gz, major, in_h, in_w, minor, kernel_h, kernel_w, oh, ow, 1, 1, 1, 1, pad_x0, pad_y0};
  a.f_bias = bias; a.f_ref = ref; a.f_work = work_b; a.f_alpha = alpha; a.f_scale = scale;
  // work_b has one row per workgroup: major * diagan_rowdot_chunks(major, oh * ow) of them (diagan_styled_bias_act_bwd_finish sums those)
  const int blocks = major * diagan_rowdot_chunks(major, oh * ow);
  if (fir_rows()) hipLaunchKernelGGL((fir_cl4_rows_kernel<4, 4, 4, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((fir_cl4_kernel<4, 4, 1, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("fir_gate_bwd");
}

// see include/diagan_hip.h: the StyledConv tail (and optionally the next layer's style) applied to blur(x) in ONE pass (FUSE 2)
DIAGAN_API int diagan_fir_styled_act(const float* input, const float* kernel, float* out, int major, int in_h, int in_w, int minor,
                                     int kernel_h, int kernel_w, int pad_x0, int pad_x1, int pad_y0, int pad_y1, const float* demod,
                                     const float* noise, const float* strength, const float* bias, const float* post,
                                     int noise_per_image, float alpha, float scale, void* stream) {
  DG_REQUIRE(input && kernel && out && major > 0 && in_h > 0 && in_w > 0 && minor > 0, "fir_styled_act: bad args");
  DG_REQUIRE(kernel_w == 4 && kernel_h >= 1 && kernel_h <= 16 && (minor & 3) == 0, "fir_styled_act: 4-tap-wide filters on channel counts that are multiples of 4 (got %d x %d taps, %d channels)", kernel_h, kernel_w, minor);
  DG_REQUIRE(!noise || strength, "fir_styled_act: noise needs its strength");
  DG_REQUIRE((((uintptr_t)input | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)demod | (uintptr_t)post) & 15) == 0,
             "fir_styled_act: pointers must be 16-byte aligned");
  const int oh = in_h + pad_y0 + pad_y1 - kernel_h + 1, ow = in_w + pad_x0 + pad_x1 - kernel_w + 1;
  DG_REQUIRE(oh > 0 && ow > 0, "fir_styled_act: empty output (%d x %d)", oh, ow);
  UpFirDnArgs a{input, kernel, out, major, in_h, in_w, minor, kernel_h, kernel_w, oh, ow, 1, 1, 1, 1, pad_x0, pad_y0};
  a.f_bias = bias; a.f_demod = demod; a.f_noise = noise; a.f_strength = strength; a.f_post = post;
  a.f_noise_per_image = noise_per_image; a.f_alpha = alpha; a.f_scale = scale;
  if (fir_rows()) {
    const long work4 = (long)major * ((oh + 3) / 4) * ((ow + 3) / 4) * (minor / 4);
    long fb4 = (work4 + 255) / 256;
    if (fb4 > 32768) fb4 = 32768;
    hipLaunchKernelGGL((fir_cl4_rows_kernel<4, 4, 4, 2>), dim3((int)fb4), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("fir_styled_act");
  }
  const long work = (long)major * oh * ((ow + 3) / 4) * (minor / 4);
  long fb = (work + 255) / 256;
  if (fb > 32768) fb = 32768;
  hipLaunchKernelGGL((fir_cl4_kernel<4, 4, 1, 2>), dim3((int)fb), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("fir_styled_act");
}

// ---- ToRGB in one pass (round 6) ----------------------------------------------------------------------------------------------------
// The generator's ToRGB (reference stylegan2.py:332-351) is a modulated 1x1 convolution to 3 planes without demodulation:
//   out[b,p,o] = sum_c w[o][c] * (x[b,p,c] * s[b,c]) + bias[o]
// As scale_rows + implicit GEMM it costs three passes over the layer's full-resolution input (multiply: read + write, convolution: read)
// for 3 output planes, and its backward six more.  Here: one read of x forward; backward one read of x and one write of gx:
//   gx[b,p,c]     = (sum_o gy[b,p,o] * w[o][c]) * s[b,c]
//   work[b][k][o][c] = sum over the pixels of chunk k of gy[b,p,o] * x[b,p,c]       (-> d(s) and d(w) in torgb_finish_kernel)
// Layout: 256 threads = L lanes along the channel quads (L = min(C / 4, 64): one pixel's lanes sit in one wave) x 256 / L pixels; a
// lane owns the quads cq, cq + L, ... (NQ = C / 4 / L <= 4); blocks = (image, pixel chunk) as rowdot_partial_kernel.
// v + (v of the lane `ctrl` points at; 0 where that lane does not exist or the row is masked out): one full-rate VALU instruction
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// sum over each group of L consecutive lanes (L a power of two <= 64, groups aligned), valid in the group's LAST lane: the prefix-sum
// ladder row_shr:1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 / :31 across them -- no LDS crossbar (ds_bpermute) involved
__device__ __forceinline__ float group_sum_last(float v, int L) {
  if (L >= 2) v = dpp_add<0x111, 0xf>(v);
  if (L >= 4) v = dpp_add<0x112, 0xf>(v);
  if (L >= 8) v = dpp_add<0x114, 0xf>(v);
  if (L >= 16) v = dpp_add<0x118, 0xf>(v);
  if (L >= 32) v = dpp_add<0x142, 0xa>(v);
  if (L >= 64) v = dpp_add<0x143, 0xc>(v);
  return v;
}

template <int NQ>
__global__ __launch_bounds__(256) void torgb_fwd_kernel(const f32x4* __restrict__ x, const float* __restrict__ s, const float* __restrict__ w,
                                                        const float* __restrict__ bias, f32x4* __restrict__ out, int P, int C, int chunks) {
  constexpr int U = 4;                                     // pixels per lane and trip: U independent loads in flight
  const int q = C >> 2, L = q / NQ, ppb = 256 / L;
  const int cq = threadIdx.x % L, pl = threadIdx.x / L;
  const int img = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int per = (P + chunks - 1) / chunks;
  const int p0 = ch * per, p1 = min(p0 + per, P);
  f32x4 sq[NQ], wq[NQ][3];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = (cq + j * L) * 4;
    sq[j] = *reinterpret_cast<const f32x4*>(s + (long)img * C + c);
#pragma unroll
    for (int o = 0; o < 3; ++o) wq[j][o] = *reinterpret_cast<const f32x4*>(w + (long)o * C + c);
  }
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) { bv[0] = bias[0]; bv[1] = bias[1]; bv[2] = bias[2]; }
  const long base = (long)img * P * q + cq;
  const int iters = (p1 - p0 + U * ppb - 1) / (U * ppb);   // (block-uniform trip count: the lane exchanges below need every lane)
  for (int it = 0; it < iters; ++it) {
    f32x4 v[U][NQ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + (it * U + u) * ppb + pl;
#pragma unroll
      for (int j = 0; j < NQ; ++j) v[u][j] = p < p1 ? x[base + (long)p * q + j * L] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + (it * U + u) * ppb + pl;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const f32x4 xm = v[u][j] * sq[j];                  // (the modulated input as scale_rows rounds it)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0 = fmaf(xm[e], wq[j][0][e], a0);
          a1 = fmaf(xm[e], wq[j][1][e], a1);
          a2 = fmaf(xm[e], wq[j][2][e], a2);
        }
      }
      a0 = group_sum_last(a0, L);
      a1 = group_sum_last(a1, L);
      a2 = group_sum_last(a2, L);
      if (p < p1 && cq == L - 1) out[(long)img * P + p] = f32x4{a0 + bv[0], a1 + bv[1], a2 + bv[2], 0.f};
    }
  }
}

template <int NQ>
__global__ __launch_bounds__(256) void torgb_bwd_kernel(const f32x4* __restrict__ gy, const f32x4* __restrict__ x, const float* __restrict__ s,
                                                        const float* __restrict__ w, f32x4* __restrict__ gx, float* __restrict__ work,
                                                        int P, int C, int chunks) {
  __shared__ f32x4 red[256];
  const int q = C >> 2, L = q / NQ, ppb = 256 / L;
  const int cq = threadIdx.x % L, pl = threadIdx.x / L;
  const int img = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int per = (P + chunks - 1) / chunks;
  const int p0 = ch * per, p1 = min(p0 + per, P);
  f32x4 sq[NQ], wq[NQ][3], acc[NQ][3];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = (cq + j * L) * 4;
    sq[j] = *reinterpret_cast<const f32x4*>(s + (long)img * C + c);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      wq[j][o] = *reinterpret_cast<const f32x4*>(w + (long)o * C + c);
      acc[j][o] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const long base = (long)img * P * q + cq;
  constexpr int U = 4;                                     // pixels per lane and trip (independent loads in flight)
  for (int pp = p0 + pl; pp < p1; pp += U * ppb) {
    f32x4 g[U], xv[U][NQ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pp + u * ppb;
      const bool ok = p < p1;
      g[u] = ok ? gy[(long)img * P + p] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NQ; ++j) xv[u][j] = ok ? x[base + (long)p * q + j * L] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pp + u * ppb;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        f32x4 t = g[u][0] * wq[j][0];
        t += g[u][1] * wq[j][1];
        t += g[u][2] * wq[j][2];
        if (gx && p < p1) gx[base + (long)p * q + j * L] = t * sq[j];
        acc[j][0] += g[u][0] * xv[u][j];                   // (pixels past the chunk add zeros, in the same order every run)
        acc[j][1] += g[u][1] * xv[u][j];
        acc[j][2] += g[u][2] * xv[u][j];
      }
    }
  }
  // the pixel lanes' partial sums, in a fixed order, to work[block][o][c]
#pragma unroll
  for (int j = 0; j < NQ; ++j)
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      __syncthreads();
      red[threadIdx.x] = acc[j][o];
      __syncthreads();
      if (pl == 0) {
        f32x4 t = red[cq];
        for (int l = 1; l < ppb; ++l) t += red[l * L + cq];
        *reinterpret_cast<f32x4*>(work + ((long)blockIdx.x * 3 + o) * C + (cq + j * L) * 4) = t;
      }
    }
}

// T[b][o][c] = the chunks' partial sums added in double in a fixed order (one block per (b, o)); then
// d(s)[b][c] = sum_o w[o][c] * T[b][o][c] (blocks 0 .. B-1), d(w)[o][c] = sum_b s[b][c] * T[b][o][c] (blocks B .. B+2)
__global__ __launch_bounds__(256) void torgb_sum_kernel(const float* __restrict__ work, float* __restrict__ T, int C, int chunks) {
  const int b = blockIdx.x / 3, o = blockIdx.x - b * 3;
  for (int c = threadIdx.x; c < C; c += 256) {
    double u = 0.0;
    for (int k = 0; k < chunks; ++k) u += (double)work[(((long)b * chunks + k) * 3 + o) * C + c];
    T[(long)blockIdx.x * C + c] = (float)u;
  }
}
__global__ __launch_bounds__(256) void torgb_finish_kernel(const float* __restrict__ T, const float* __restrict__ s, const float* __restrict__ w,
                                                           float* __restrict__ gs, float* __restrict__ gw, int B, int C) {
  const int blk = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    if (blk < B) {
      double t = 0.0;
      for (int o = 0; o < 3; ++o) t += (double)w[(long)o * C + c] * (double)T[((long)blk * 3 + o) * C + c];
      gs[(long)blk * C + c] = (float)t;
    } else {
      const int o = blk - B;
      double t = 0.0;
      for (int b = 0; b < B; ++b) t += (double)s[(long)b * C + c] * (double)T[((long)b * 3 + o) * C + c];
      gw[(long)o * C + c] = (float)t;
    }
  }
}

// ---- FromRGB in one pass (round 6) ----------------------------------------------------------------------------------------------------
// The discriminator's first ConvLayer (reference stylegan2.py:553-595, :616-640: EqualConv2d 3 -> C, 1x1, then FusedLeakyReLU):
//   y[b,p,c] = leaky_relu(sum_{i<3} w[c][i] * wscale * x[b,p,i] + bias[c]) * scale
// As implicit GEMM (K = 4) + activation pass it writes the full-resolution C-channel tensor, reads it and writes it again; the backward
// reads gy and y, writes the gated gradient, and reads that again for the weight gradient.  Here: ONE write forward; backward one read of gy
// and y:  gz = gy * scale * (y > 0 ? 1 : alpha);  work[blk][i][c] = sum_p gz * x[.,i] (i < 3: d(w) / wscale), work[blk][3][c] = sum_p gz
// (d(bias));  gx[b,p,i] = wscale * sum_c gz[c] * w[c][i] (only when the images need a gradient: the generator's step).
__global__ __launch_bounds__(256) void fromrgb_fwd_kernel(const f32x4* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          f32x4* __restrict__ y, long npix, int C, float wscale, float alpha, float scale) {
  const int q = C >> 2;
  const long total = npix * q, stride = (long)gridDim.x * 256;
  long o = (long)blockIdx.x * 256 + threadIdx.x;
  const int cq = (int)(o % q);                     // (q divides the stride: the lane's channel quad never changes)
  f32x4 wr[3], bv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[i][e] = w[(cq * 4 + e) * 3 + i] * wscale;
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + cq * 4);
  for (; o < total; o += stride) {
    const f32x4 xv = x[o / q];
    f32x4 v = xv[0] * wr[0];
    v += xv[1] * wr[1];
    v += xv[2] * wr[2];
    v += bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.f ? v[e] : v[e] * alpha) * scale;
    y[o] = v;
  }
}

__global__ __launch_bounds__(256) void fromrgb_bwd_kernel(const f32x4* __restrict__ gy, const f32x4* __restrict__ y, const f32x4* __restrict__ x,
                                                          const float* __restrict__ w, f32x4* __restrict__ gx, float* __restrict__ work,
                                                          int P, int C, int chunks, float wscale, float alpha, float scale) {
  __shared__ f32x4 red[256];
  const int q = C >> 2, lanes = 256 / q;
  const int cq = threadIdx.x % q, pl = threadIdx.x / q;
  const int img = blockIdx.x / chunks, ch = blockIdx.x % chunks;
  const int per = (P + chunks - 1) / chunks;
  const int p0 = ch * per, p1 = min(p0 + per, P);
  f32x4 wr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[i][e] = w[(cq * 4 + e) * 3 + i] * wscale;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long base = (long)img * P * q + cq;
  const int iters = (p1 - p0 + lanes - 1) / lanes;       // (block-uniform trip count: the lane exchange below needs every lane of a pixel)
  for (int it = 0; it < iters; ++it) {
    const int p = p0 + it * lanes + pl;
    const bool ok = p < p1;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      const long i = base + (long)p * q;
      g = gy[i];
      const f32x4 r = y[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = g[e] * (r[e] > 0.f ? scale : scale * alpha);
      xv = x[(long)img * P + p];
    }
    acc[0] += g * xv[0];
    acc[1] += g * xv[1];
    acc[2] += g * xv[2];
    acc[3] += g;
    if (gx) {                                            // (block-uniform; q <= 64: one pixel's lanes sit in one wave)
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t0 = fmaf(g[e], wr[0][e], t0);
        t1 = fmaf(g[e], wr[1][e], t1);
        t2 = fmaf(g[e], wr[2][e], t2);
      }
      t0 = group_sum_last(t0, q);
      t1 = group_sum_last(t1, q);
      t2 = group_sum_last(t2, q);
      if (ok && cq == q - 1) gx[(long)img * P + p] = f32x4{t0, t1, t2, 0.f};
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    __syncthreads();
    red[threadIdx.x] = acc[i];
    __syncthreads();
    if (pl == 0) {
      f32x4 t = red[cq];
      for (int l = 1; l < lanes; ++l) t += red[l * q + cq];
      *reinterpret_cast<f32x4*>(work + ((long)blockIdx.x * 4 + i) * C + cq * 4) = t;
    }
  }
}

// see include/diagan_hip.h
DIAGAN_API int diagan_fromrgb_fwd(const float* x, const float* w, const float* bias, float* y, int B, int P, int C, float wscale, float alpha,
                                  float scale, void* stream) {
  DG_REQUIRE(x && w && y && B > 0 && P > 0, "fromrgb_fwd: bad args");
  DG_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "fromrgb_fwd: C=%d must be a power of two in [4, 1024]", C);
  DG_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) & 15) == 0, "fromrgb_fwd: pointers must be 16-byte aligned");
  const long npix = (long)B * P, total = npix * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(fromrgb_fwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, w, bias, (f32x4*)y, npix, C,
                     wscale, alpha, scale);
  return check_launch("fromrgb_fwd");
}

DIAGAN_API int diagan_fromrgb_bwd(const float* gy, const float* y, const float* x, const float* w, float* gx, float* work, int B, int P, int C,
                                  float wscale, float alpha, float scale, void* stream) {
  DG_REQUIRE(gy && y && x && w && work && B > 0 && P > 0, "fromrgb_bwd: bad args");
  DG_REQUIRE(C >= 4 && C <= 256 && (C & (C - 1)) == 0, "fromrgb_bwd: C=%d must be a power of two in [4, 256]", C);
  DG_REQUIRE((((uintptr_t)gy | (uintptr_t)y | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)work) & 15) == 0, "fromrgb_bwd: pointers must be 16-byte aligned");
  const int chunks = diagan_rowdot_chunks(B, P);
  hipLaunchKernelGGL(fromrgb_bwd_kernel, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)gy, (const f32x4*)y, (const f32x4*)x,
                     w, (f32x4*)gx, work, P, C, chunks, wscale, alpha, scale);
  return check_launch("fromrgb_bwd");
}

static int torgb_nq(int C) { return C <= 256 ? 1 : C / 256; }

// see include/diagan_hip.h
DIAGAN_API int diagan_torgb_fwd(const float* x, const float* s, const float* w, const float* bias, float* out, int B, int P, int C,
                                void* stream) {
  DG_REQUIRE(x && s && w && out && B > 0 && P > 0, "torgb_fwd: bad args");
  DG_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "torgb_fwd: C=%d must be a power of two in [4, 1024]", C);
  DG_REQUIRE((((uintptr_t)x | (uintptr_t)s | (uintptr_t)w | (uintptr_t)out) & 15) == 0, "torgb_fwd: pointers must be 16-byte aligned");
  const int chunks = diagan_rowdot_chunks(B, P);
  const dim3 grid(B * chunks);
  hipStream_t st = (hipStream_t)stream;
  switch (torgb_nq(C)) {
    case 1: hipLaunchKernelGGL((torgb_fwd_kernel<1>), grid, dim3(256), 0, st, (const f32x4*)x, s, w, bias, (f32x4*)out, P, C, chunks); break;
    case 2: hipLaunchKernelGGL((torgb_fwd_kernel<2>), grid, dim3(256), 0, st, (const f32x4*)x, s, w, bias, (f32x4*)out, P, C, chunks); break;
    default: hipLaunchKernelGGL((torgb_fwd_kernel<4>), grid, dim3(256), 0, st, (const f32x4*)x, s, w, bias, (f32x4*)out, P, C, chunks); break;
  }
  return check_launch("torgb_fwd");
}

DIAGAN_API int diagan_torgb_bwd(const float* gy, const float* x, const float* s, const float* w, float* gx, float* work, float* gs, float* gw,
                                int B, int P, int C, void* stream) {
  DG_REQUIRE(gy && x && s && w && work && gs && gw && B > 0 && P > 0, "torgb_bwd: bad args");
  DG_REQUIRE(C >= 4 && C <= 1024 && (C & (C - 1)) == 0, "torgb_bwd: C=%d must be a power of two in [4, 1024]", C);
  DG_REQUIRE((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)s | (uintptr_t)w | (uintptr_t)gx | (uintptr_t)work) & 15) == 0,
             "torgb_bwd: pointers must be 16-byte aligned");
  const int chunks = diagan_rowdot_chunks(B, P);
  const dim3 grid(B * chunks);
  hipStream_t st = (hipStream_t)stream;
  switch (torgb_nq(C)) {
    case 1: hipLaunchKernelGGL((torgb_bwd_kernel<1>), grid, dim3(256), 0, st, (const f32x4*)gy, (const f32x4*)x, s, w, (f32x4*)gx, work, P, C, chunks); break;
    case 2: hipLaunchKernelGGL((torgb_bwd_kernel<2>), grid, dim3(256), 0, st, (const f32x4*)gy, (const f32x4*)x, s, w, (f32x4*)gx, work, P, C, chunks); break;
    default: hipLaunchKernelGGL((torgb_bwd_kernel<4>), grid, dim3(256), 0, st, (const f32x4*)gy, (const f32x4*)x, s, w, (f32x4*)gx, work, P, C, chunks); break;
  }
  float* T = work + (long)B * chunks * 3 * C;
  hipLaunchKernelGGL(torgb_sum_kernel, dim3(B * 3), dim3(256), 0, st, work, T, C, chunks);
  hipLaunchKernelGGL(torgb_finish_kernel, dim3(B + 3), dim3(256), 0, st, T, s, w, gs, gw, B, C);
  return check_launch("torgb_bwd");
}

// out dims: ((in*up + pad0 + pad1 - k) / down) + 1 ; returns them through out_h/out_w when out == NULL
DIAGAN_API int diagan_upfirdn2d(const float* input, const float* kernel, float* out, int major, int in_h, int in_w,
                                int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                                int pad_x0, int pad_x1, int pad_y0, int pad_y1, int* out_h, int* out_w, void* stream) {
  DG_REQUIRE(major >= 0 && in_h > 0 && in_w > 0 && minor > 0 && kernel_h > 0 && kernel_w > 0, "upfirdn2d: bad dims");
  DG_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "upfirdn2d: up/down must be positive");
  DG_REQUIRE(kernel_h * kernel_w <= 4096, "upfirdn2d: filter larger than 4096 taps");
  const int oh = (in_h * up_y + pad_y0 + pad_y1 - kernel_h) / down_y + 1;
  const int ow = (in_w * up_x + pad_x0 + pad_x1 - kernel_w) / down_x + 1;
  if (out_h) *out_h = oh;
  if (out_w) *out_w = ow;
  if (!out) return DIAGAN_OK;    // size query
  DG_REQUIRE(input && kernel, "upfirdn2d: null tensor");
  DG_REQUIRE(oh > 0 && ow > 0, "upfirdn2d: empty output (%d x %d)", oh, ow);
  if (major == 0) return DIAGAN_OK;
  UpFirDnArgs a{input, kernel, out, major, in_h, in_w, minor, kernel_h, kernel_w, oh, ow,
                up_x, up_y, down_x, down_y, pad_x0, pad_y0};
  if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kernel_w == 4 && kernel_h <= 16 && (minor & 3) == 0 &&
      (((uintptr_t)input | (uintptr_t)out) & 15) == 0) {
    if (fir_rows()) {
      const long work4 = (long)major * ((oh + 3) / 4) * ((ow + 3) / 4) * (minor / 4);
      long fb4 = (work4 + 255) / 256;
      if (fb4 > 32768) fb4 = 32768;
      hipLaunchKernelGGL((fir_cl4_rows_kernel<4, 4, 4, 0>), dim3((int)fb4), dim3(256), 0, (hipStream_t)stream, a);
      return check_launch("upfirdn2d");
    }
    const long work = (long)major * oh * ((ow + 3) / 4) * (minor / 4);
    long fb = (work + 255) / 256;
    if (fb > 32768) fb = 32768;
    hipLaunchKernelGGL((fir_cl4_kernel<4, 4>), dim3((int)fb), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("upfirdn2d");
  }
  const bool cl4 = (minor & 3) == 0 && (((uintptr_t)input | (uintptr_t)out) & 15) == 0;
  if (cl4 && up_x == 1 && up_y == 1 && down_x == 2 && down_y == 2 && kernel_w == 4 && kernel_h <= 16) {
    const long work = (long)major * oh * ((ow + 1) / 2) * (minor / 4);
    long fb = (work + 255) / 256;
    if (fb > 32768) fb = 32768;
    hipLaunchKernelGGL((fir_cl4_kernel<4, 2, 2>), dim3((int)fb), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("upfirdn2d");
  }
  if (cl4) {
    const long work = (long)major * oh * ow * (minor / 4);
    long fb = (work + 255) / 256;
    if (fb > 32768) fb = 32768;
    hipLaunchKernelGGL(updn_cl4_kernel, dim3((int)fb), dim3(256), (size_t)kernel_h * kernel_w * sizeof(float), (hipStream_t)stream, a);
    return check_launch("upfirdn2d");
  }
  const long total = (long)major * oh * ow * minor;
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3((int)blocks), dim3(256), (size_t)kernel_h * kernel_w * sizeof(float),
                     (hipStream_t)stream, a);
  return check_launch("upfirdn2d");
}
