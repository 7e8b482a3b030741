// The small dense pieces of StyleGAN2's modulated convolution as single launches (round 6).
//
// Every ModulatedConv2d (reference: diagan-pkg/diagan/models/stylegan2.py:169-265) computes, per forward,
//   s[b][ci]  = EqualLinear(style):  scale_l * sum_k W_l[ci][k] * style[b][k] + bias[ci] * lr_mul               (:132-166, :217, :228)
//   d[b][co]  = rsqrt(scale^2 * sum_ci s[b][ci]^2 * sum_taps w[co][ci][tap]^2 + eps)                            (:236-238: demodulation)
// -- a few MFLOP each.  Through the general convolution path (pack, implicit GEMM, bias add; square, tap sum, pack, implicit GEMM, + eps,
// rsqrt) they were 10 launches forward and ~14 backward per layer, ~600 of an iteration's 3300 launches, each 5-14 us of kernel time plus
// the gap to the next one.  Here: one launch each forward, one each (first-order) backward.  No matrix pipe: the operands are tiny and
// the kernels are bound by launch latency and a few loads in flight.
#include "common.h"
#include <stdint.h>

namespace diagan {

typedef float sd_f32x4 __attribute__((ext_vector_type(4)));

// out[b][c] = scale * sum_k W[c][k] * x[b][k] + bias[c] * bias_mul.  These kernels are chains of dependent loads, not arithmetic: eight
// independent loads in flight per trip.  256 threads = 8 columns c x 32 rows b (rows b + 32, ... in further passes).  (Splitting K over
// four lanes of 1024-thread blocks measured slower: 30 against 20 us -- every lane then walks its own cache lines.)
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                               float* __restrict__ out, int B, int K, int C, float scale, float bias_mul) {
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5), bl = threadIdx.x & 31;
  if (c >= C) return;
  const sd_f32x4* wr = reinterpret_cast<const sd_f32x4*>(W + (long)c * K);
  for (int b = bl; b < B; b += 32) {
    const sd_f32x4* xr = reinterpret_cast<const sd_f32x4*>(x + (long)b * K);
    sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= (K >> 2); k += 8) {
      sd_f32x4 wv[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { wv[u] = wr[k + u]; xv[u] = xr[k + u]; }
#pragma unroll
      for (int u = 0; u < 8; u += 2) { acc += wv[u] * xv[u]; acc2 += wv[u + 1] * xv[u + 1]; }
    }
    for (; k < (K >> 2); ++k) acc += wr[k] * xr[k];
    acc += acc2;
    const float t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    out[(long)b * C + c] = t * scale + (bias ? bias[c] * bias_mul : 0.f);
  }
}

// blocks [0, nw): gW[c][k4] = scale * sum_b g[b][c] * x[b][k4] (one thread per (c, k quad)), and gbias[c] = bias_mul * sum_b g[b][c] by the
// threads of k quad 0;  blocks [nw, nw + nx): gx[b][k4] = scale * sum_c g[b][c] * W[c][k4] (one thread per (b, k quad)); gx may be null
__global__ __launch_bounds__(256) void small_linear_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ W,
                                                               float* __restrict__ gW, float* __restrict__ gbias, float* __restrict__ gx, int B,
                                                               int K, int C, float scale, float bias_mul, int nw) {
  const int kq = K >> 2;
  if ((int)blockIdx.x < nw) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long)C * kq) return;
    const int c = (int)(o / kq), k4 = (int)(o - (long)c * kq);
    sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float sb = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
      const float gv = g[(long)b * C + c];
      acc += gv * reinterpret_cast<const sd_f32x4*>(x + (long)b * K)[k4];
      sb += gv;
    }
    reinterpret_cast<sd_f32x4*>(gW + (long)c * K)[k4] = acc * scale;
    if (gbias && k4 == 0) gbias[c] = sb * bias_mul;
    return;
  }
  // gx: one block per (row b, group of 32 k quads); 256 threads = 32 k quads x 8 parts of the C columns, added through LDS in a fixed order
  __shared__ sd_f32x4 red[256];
  const int kgroups = (kq + 31) >> 5;
  const int gb = (blockIdx.x - nw) / kgroups, k4 = ((blockIdx.x - nw) % kgroups) * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
  const int per = (C + 7) >> 3, c0 = part * per, c1 = min(c0 + per, C);
  sd_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (k4 < kq) {
#pragma unroll 8
    for (int c = c0; c < c1; ++c) acc += g[(long)gb * C + c] * reinterpret_cast<const sd_f32x4*>(W + (long)c * K)[k4];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (part == 0 && k4 < kq) {
    sd_f32x4 t = red[threadIdx.x];
#pragma unroll
    for (int l = 1; l < 8; ++l) t += red[l * 32 + threadIdx.x];
    reinterpret_cast<sd_f32x4*>(gx + (long)gb * K)[k4] = t * scale;
  }
}

// One block per output channel co.  wsq[co][ci] = sum_taps w[co][ci][tap]^2 (kept for the backward);
// d[b][co] = rsqrt(scale2 * sum_ci s[b][ci]^2 * wsq[co][ci] + eps).   256 threads: phase 2 = 8 groups of 32 lanes, group = rows b, b + 8, ...
__global__ __launch_bounds__(256) void demod_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w, float* __restrict__ d,
                                                        float* __restrict__ wsq, int B, int Ci, int Co, int taps, float scale2, float eps) {
  extern __shared__ float sq[];                       // [Ci]
  const int co = blockIdx.x;
  for (int ci = threadIdx.x; ci < Ci; ci += 256) {
    const float* p = w + ((long)co * Ci + ci) * taps;
    float t = 0.f;
    for (int k = 0; k < taps; ++k) t = fmaf(p[k], p[k], t);
    sq[ci] = t;
    wsq[(long)co * Ci + ci] = t;
  }
  __syncthreads();
  const int grp = threadIdx.x >> 5, lane = threadIdx.x & 31;
  for (int b0 = 0; b0 < B; b0 += 8) {                 // (block-uniform trip count; rows past B compute a dummy)
    const int b = b0 + grp;
    float t = 0.f;
    if (b < B)
#pragma unroll 8
      for (int ci = lane; ci < Ci; ci += 32) {
        const float sv = s[(long)b * Ci + ci];
        t = fmaf(sv * sv, sq[ci], t);
      }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (b < B && lane == 0) d[(long)b * Co + co] = rsqrtf(t * scale2 + eps);
  }
}

// G[b][co] = gd[b][co] * (-0.5) * d[b][co]^3 * scale2 is the gradient of the sum inside the rsqrt.
// blocks [0, Co): gw[co][ci][tap] = 2 * w[co][ci][tap] * sum_b G[b][co] * s[b][ci]^2
// blocks [Co, Co + B * ceil(Ci / 32)): gs[b][ci] = 2 * s[b][ci] * sum_co G[b][co] * wsq[co][ci]
__global__ __launch_bounds__(256) void demod_bwd_kernel(const float* __restrict__ gd, const float* __restrict__ d, const float* __restrict__ s,
                                                        const float* __restrict__ w, const float* __restrict__ wsq, float* __restrict__ gw,
                                                        float* __restrict__ gs, int B, int Ci, int Co, int taps, float scale2) {
  __shared__ float G[64];
  if ((int)blockIdx.x < Co) {
    const int co = blockIdx.x;
    for (int b = threadIdx.x; b < B; b += 256) {
      const float dv = d[(long)b * Co + co];
      G[b] = gd[(long)b * Co + co] * -0.5f * dv * dv * dv * scale2;
    }
    __syncthreads();
    for (int ci = threadIdx.x; ci < Ci; ci += 256) {
      float t = 0.f;
#pragma unroll 8
      for (int b = 0; b < B; ++b) {
        const float sv = s[(long)b * Ci + ci];
        t = fmaf(G[b], sv * sv, t);
      }
      const float* p = w + ((long)co * Ci + ci) * taps;
      float* q = gw + ((long)co * Ci + ci) * taps;
      for (int k = 0; k < taps; ++k) q[k] = 2.f * p[k] * t;
    }
    return;
  }
  if (!gs) return;
  // gs: one block per (row b, group of 32 columns ci); 256 threads = 32 columns x 8 parts of the Co rows, added through LDS in a fixed order
  __shared__ float redf[256];
  const int cgroups = (Ci + 31) >> 5;
  const int b = (blockIdx.x - Co) / cgroups, ci = ((blockIdx.x - Co) % cgroups) * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
  const int per = (Co + 7) >> 3, o0 = part * per, o1 = min(o0 + per, Co);
  float t = 0.f;
  if (ci < Ci) {
#pragma unroll 8
    for (int co = o0; co < o1; ++co) {
      const float dv = d[(long)b * Co + co];
      t = fmaf(gd[(long)b * Co + co] * -0.5f * dv * dv * dv * scale2, wsq[(long)co * Ci + ci], t);
    }
  }
  redf[threadIdx.x] = t;
  __syncthreads();
  if (part == 0 && ci < Ci) {
    float u = redf[threadIdx.x];
#pragma unroll
    for (int l = 1; l < 8; ++l) u += redf[l * 32 + threadIdx.x];
    gs[(long)b * Ci + ci] = 2.f * s[(long)b * Ci + ci] * u;
  }
}

}  // namespace diagan

using namespace diagan;

// see include/diagan_hip.h
DIAGAN_API int diagan_small_linear_fwd(const float* x, const float* W, const float* bias, float* out, int B, int K, int C, float scale,
                                       float bias_mul, void* stream) {
  DG_REQUIRE(x && W && out && B > 0 && K > 0 && C > 0 && (K & 3) == 0, "small_linear_fwd: bad args (K must be a multiple of 4)");
  DG_REQUIRE((((uintptr_t)x | (uintptr_t)W) & 15) == 0, "small_linear_fwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, x, W, bias, out, B, K, C, scale, bias_mul);
  return check_launch("small_linear_fwd");
}

DIAGAN_API int diagan_small_linear_bwd(const float* g, const float* x, const float* W, float* gW, float* gbias, float* gx, int B, int K, int C,
                                       float scale, float bias_mul, void* stream) {
  DG_REQUIRE(g && x && W && gW && B > 0 && K > 0 && C > 0 && (K & 3) == 0, "small_linear_bwd: bad args (K must be a multiple of 4)");
  DG_REQUIRE((((uintptr_t)x | (uintptr_t)W | (uintptr_t)gW | (uintptr_t)gx) & 15) == 0, "small_linear_bwd: pointers must be 16-byte aligned");
  const int kq = K >> 2, nw = cdiv((long)C * kq, 256), nx = gx ? B * cdiv(kq, 32) : 0;
  hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(nw + nx), dim3(256), 0, (hipStream_t)stream, g, x, W, gW, gbias, gx, B, K, C, scale, bias_mul, nw);
  return check_launch("small_linear_bwd");
}

DIAGAN_API int diagan_demod_fwd(const float* s, const float* w, float* d, float* wsq, int B, int Ci, int Co, int taps, float scale2, float eps,
                                void* stream) {
  DG_REQUIRE(s && w && d && wsq && B > 0 && Ci > 0 && Co > 0 && taps > 0, "demod_fwd: bad args");
  DG_REQUIRE(Ci <= 8192, "demod_fwd: Ci=%d too large for the LDS row", Ci);
  hipLaunchKernelGGL(demod_fwd_kernel, dim3(Co), dim3(256), (size_t)Ci * sizeof(float), (hipStream_t)stream, s, w, d, wsq, B, Ci, Co, taps, scale2, eps);
  return check_launch("demod_fwd");
}

DIAGAN_API int diagan_demod_bwd(const float* gd, const float* d, const float* s, const float* w, const float* wsq, float* gw, float* gs, int B,
                                int Ci, int Co, int taps, float scale2, void* stream) {
  DG_REQUIRE(gd && d && s && w && wsq && gw && B > 0 && B <= 64 && Ci > 0 && Co > 0 && taps > 0, "demod_bwd: bad args (B <= 64)");
  const int ns = gs ? B * cdiv(Ci, 32) : 0;
  hipLaunchKernelGGL(demod_bwd_kernel, dim3(Co + ns), dim3(256), 0, (hipStream_t)stream, gd, d, s, w, wsq, gw, gs, B, Ci, Co, taps, scale2);
  return check_launch("demod_bwd");
}
