// Weight transforms of the Winograd kernels as device functions (one workgroup of 512 threads = a block of 64 output x 32
// input channels), shared by the per-launch transform kernels of conv_wino.hip / conv_wino4.hip and by the batched transform
// of many layers in one launch (diagan_wino_weights_batched, conv_wino4.hip).  Formats: conv_common.h (WinoKind).
#pragma once
#include "conv_common.h"

namespace diagan {

// ---- F(2x2,3x3): U[f][co][ci] = (G g G^T)[i][j], f = 4 i + j, in conv_wino_kernel's LDS image order ----
__device__ __forceinline__ void wino_weight_body(const float* __restrict__ w, float* __restrict__ ug, int Co, int Ci, int Kp, int flip,
                                                 int bx, int by, f32x4* __restrict__ sg) {
  f32x4 g[3][3];
  if (!wino_stage_taps(w, by * 64, bx * 32, Co, Ci, Kp, flip, sg, g)) return;
  const int col = threadIdx.x & 63, c = bx * 32 + (threadIdx.x >> 6) * 4;
  f32x4 t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
    t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
    t[3][s] = g[2][s];
  }
  const int ks = c >> 3, kq = (c >> 2) & 1;
  float* base = ug + ((long)by * (Ci >> 3) + ks) * (16 * 2 * 64 * 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 u[4];
    u[0] = t[i][0];
    u[1] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    u[2] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    u[3] = t[i][2];
    // row i = 3 of V is staged NEGATED (t3 - t1: every lane's column transform is then own + sc * partner); the sign
    // moves into U so that the products are unchanged
    const float sg = i == 3 ? -1.f : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* plane = base + ((i * 4 + j) * 2 + kq) * 256;
      const f32x4 v = u[j] * sg;
      *reinterpret_cast<f32x4*>(plane + col * 4) = v;
    }
  }
}


// ---- F(4x4,3x3): mode helpers (conv_wino4.hip documents the modes) and the transform ----
template <int MODE> struct W4M {
  static constexpr bool pooled = MODE == 1 || MODE == 2;
  static constexpr bool upin = MODE == 3;
  static constexpr int NS = pooled ? 7 : 9;          // slots (frequency, column half) per wave
  static constexpr int U_FLOATS = 8 * NS * 256;       // weight units of one K-step
  static constexpr int NI = (MODE == 2 || MODE == 3) ? 4 : 6;        // input loads per thread and K-step
};
__host__ __device__ __forceinline__ int w4p_start(int g) { return g == 0 ? 0 : 7 + 6 * (g - 1); }
__host__ __device__ __forceinline__ int w4p_count(int g) { return g == 0 ? 7 : 6; }
__host__ __device__ __forceinline__ int w4p_freq(int v) { return v < 2 ? v : v + 1; }        // {0, 1, 3, 4, 5}

// ---- X3 (round 5): the frequency-domain operands as THREE bf16 pieces per fp32 value ----
// v = p0 + p1 + p2 EXACTLY: each piece is the truncation of what is left to the top 16 bits of its fp32 pattern (a bf16 number:
// 8 significant bits, fp32's exponent range -- no scaling, no overflow / underflow cases), so the three pieces carry all 24
// significant bits.  The six piece products of order <= 2^-16 (p0 q0, p0 q1, p1 q0, p0 q2, p1 q1, p2 q0; each one exact in the
// fp32 accumulator of v_mfma_f32_32x32x16_bf16) reproduce the fp32 product to ~2^-23 relative.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int X3_UNIT = 1536;          // bytes of one weight unit: 3 pieces x 32 columns x 8 channels bf16
__device__ __forceinline__ void x3_split(const f32x4& v, u32x2& p0, u32x2& p1, u32x2& p2) {
  unsigned x[4], r[4], q[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = __float_as_uint(v[e]);
  p0[0] = __builtin_amdgcn_perm(x[1], x[0], 0x07060302u);        // (hi16 of x1) << 16 | hi16 of x0
  p0[1] = __builtin_amdgcn_perm(x[3], x[2], 0x07060302u);
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = __float_as_uint(v[e] - __uint_as_float(x[e] & 0xffff0000u));      // exact
  p1[0] = __builtin_amdgcn_perm(r[1], r[0], 0x07060302u);
  p1[1] = __builtin_amdgcn_perm(r[3], r[2], 0x07060302u);
#pragma unroll
  for (int e = 0; e < 4; ++e) q[e] = __float_as_uint(__uint_as_float(r[e]) - __uint_as_float(r[e] & 0xffff0000u));
  p2[0] = __builtin_amdgcn_perm(q[1], q[0], 0x07060302u);
  p2[1] = __builtin_amdgcn_perm(q[3], q[2], 0x07060302u);
}

// ---- format WK_GX3 (conv_gemm_x3.hip): the packed GEMM operand itself, split -------------------------------------------
// packed fp32 weights [Co][Kp] -> three bf16 planes [piece][Co][Kp]; quad = index of a 4-float group
__device__ __forceinline__ void gx3_split_quad(const float* __restrict__ w, unsigned short* __restrict__ wx, long quad, long plane) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(w + quad * 4);
  u32x2 p0, p1, p2;
  x3_split(v, p0, p1, p2);
  *reinterpret_cast<u32x2*>(wx + quad * 4) = p0;
  *reinterpret_cast<u32x2*>(wx + plane + quad * 4) = p1;
  *reinterpret_cast<u32x2*>(wx + 2 * plane + quad * 4) = p2;
}

// one job of wino_weights_batched_kernel (512 threads per workgroup, `nblk` workgroups for the layer)
__device__ __forceinline__ void gx3_weight_job(const float* __restrict__ w, float* __restrict__ u, int Co, int Kp, int lb, int nblk) {
  const long quads = (long)Co * Kp / 4, plane = quads * 4;
  const long per = (quads + nblk - 1) / nblk;
  const long q1 = min(quads, (long)(lb + 1) * per);
  for (long i = (long)lb * per + threadIdx.x; i < q1; i += 512) gx3_split_quad(w, reinterpret_cast<unsigned short*>(u), i, plane);
}


// U for the X3 kernels: the same (G g G^T) as wino4_weight_body, split, in the order ONE WAVE streams it:
// [64-column block][wave = group + 4 * column half][K-step][slot][piece][column (32)][8 channels bf16] -- a contiguous stream
// of NS x 1536 bytes per wave and K-step, fetched by that wave in 1 KB LDS-DMA granules through a 4-unit ring.
template <int MODE>
__device__ __forceinline__ void wino4x_weight_body(const float* __restrict__ w, float* __restrict__ ug, int Co, int Ci, int Kp, int flip,
                                                   float wscale, int bx, int by, f32x4* __restrict__ sg) {
  f32x4 g[3][3];
  const int nb = by;
  if (!wino_stage_taps(w, nb * 64, bx * 32, Co, Ci, Kp, flip, sg, g)) return;
  if (wscale != 1.f) {
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] *= wscale;
  }
  const int col = threadIdx.x & 63, c = bx * 32 + (threadIdx.x >> 6) * 4;
  constexpr float k4 = 0.25f, k6 = 1.f / 6.f, k12 = 1.f / 12.f, k24 = 1.f / 24.f;
  auto gt = [&](const f32x4& g0, const f32x4& g1, const f32x4& g2, f32x4* o) {
    o[0] = k4 * g0;
    o[1] = -k6 * (g0 + g1 + g2);
    o[2] = -k6 * (g0 - g1 + g2);
    o[3] = k24 * g0 + k12 * g1 + k6 * g2;
    o[4] = k24 * g0 - k12 * g1 + k6 * g2;
    o[5] = g2;
  };
  f32x4 t[3][6];
#pragma unroll
  for (int s = 0; s < 3; ++s) gt(g[0][s], g[1][s], g[2][s], t[s]);
  constexpr int NS = W4M<MODE>::NS;
  const int nk = Ci >> 3, ks = c >> 3, kh = (c >> 2) & 1, nh = col >> 5, n = col & 31;
  char* base = reinterpret_cast<char*>(ug) + (long)nb * 8 * nk * NS * X3_UNIT + n * 16 + kh * 8;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    f32x4 u[6];
    gt(t[0][i], t[1][i], t[2][i], u);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      int grp, slot;
      if (MODE == 0) {
        grp = (i / 3) * 2 + j / 3;
        slot = (i % 3) * 3 + j % 3;
      } else {
        if (i == 2 || j == 2) continue;
        const int l = 5 * (i < 2 ? i : i - 1) + (j < 2 ? j : j - 1);
        grp = l < 7 ? 0 : 1 + (l - 7) / 6;
        slot = l - w4p_start(grp);
      }
      u32x2 p0, p1, p2;
      x3_split(u[j], p0, p1, p2);
      char* d = base + ((long)((grp + 4 * nh) * nk + ks) * NS + slot) * X3_UNIT;
      *reinterpret_cast<u32x2*>(d) = p0;
      *reinterpret_cast<u32x2*>(d + 512) = p1;
      *reinterpret_cast<u32x2*>(d + 1024) = p2;
    }
  }
}

template <int MODE>
__device__ __forceinline__ void wino4_weight_body(const float* __restrict__ w, float* __restrict__ ug, int Co, int Ci, int Kp, int flip,
                                                  float wscale, int bx, int by, f32x4* __restrict__ sg) {
  f32x4 g[3][3];
  const int nb = by;
  if (!wino_stage_taps(w, nb * 64, bx * 32, Co, Ci, Kp, flip, sg, g)) return;
  if (wscale != 1.f) {                                  // (MODE 3: the 1/16 of the two interpolation passes, exact)
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] *= wscale;
  }
  const int col = threadIdx.x & 63, c = bx * 32 + (threadIdx.x >> 6) * 4;
  constexpr float k4 = 0.25f, k6 = 1.f / 6.f, k12 = 1.f / 12.f, k24 = 1.f / 24.f;
  auto gt = [&](const f32x4& g0, const f32x4& g1, const f32x4& g2, f32x4* o) {
    o[0] = k4 * g0;
    o[1] = -k6 * (g0 + g1 + g2);
    o[2] = -k6 * (g0 - g1 + g2);
    o[3] = k24 * g0 + k12 * g1 + k6 * g2;
    o[4] = k24 * g0 - k12 * g1 + k6 * g2;
    o[5] = g2;
  };
  f32x4 t[3][6];                                        // t[s][i] = (G g)[i][s]
#pragma unroll
  for (int s = 0; s < 3; ++s) gt(g[0][s], g[1][s], g[2][s], t[s]);
  const int nk = Ci >> 3, ks = c >> 3, kh = (c >> 2) & 1, nh = col >> 5, n = col & 31;
  float* base = ug + ((long)nb * nk + ks) * W4M<MODE>::U_FLOATS + (kh * 32 + n) * 4;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    f32x4 u[6];
    gt(t[0][i], t[1][i], t[2][i], u);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (MODE == 0) {
        // wave group g owns the 3 x 3 block of frequencies i in 3 (g >> 1) .. + 2, j in 3 (g & 1) .. + 2; slot 3 (i % 3) + j % 3
        const int unit = ((((i / 3) * 2 + j / 3) * 2 + nh) * 9) + (i % 3) * 3 + j % 3;
        *reinterpret_cast<f32x4*>(base + unit * 256) = u[j];
      } else if (i != 2 && j != 2) {
        const int l = 5 * (i < 2 ? i : i - 1) + (j < 2 ? j : j - 1);          // live frequency index
        const int gq = l < 7 ? 0 : 1 + (l - 7) / 6, sl = l - w4p_start(gq);
        *reinterpret_cast<f32x4*>(base + ((gq * 2 + nh) * 7 + sl) * 256) = u[j];
      }
    }
  }
}


}  // namespace diagan
