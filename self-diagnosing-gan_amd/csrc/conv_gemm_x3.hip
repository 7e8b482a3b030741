// Implicit GEMM of the small-map 3x3 / stride 1 / pad 1 layers on the bf16 matrix pipe with EXACTLY split operands (round 5).
//
// Replaces, for the launches that have at most one 64 x 64 output tile per CU and a long K loop (SNGAN-32's discriminator blocks 3 / 4
// at 8x8: M = 8192, N = 128, K = 1152; `tile_cfg` 14 of conv_gemm.hip otherwise), the fp32 `v_mfma_f32_32x32x2_f32` loop: those
// launches are bound by the matrix pipe itself (15.4 of their 26 us at 100 % of it), so the lever is fewer pipe cycles.
// Every fp32 operand is the exact sum of three bf16 pieces (wino_weights.h: x3_split); six piece products accumulated in fp32 by
// `v_mfma_f32_32x32x16_bf16` reproduce the fp32 product to ~2^-23: per 8 channels THREE MFMAs of 32 cycles
//     (a0|a1).(b0|b0) + (a0|a1).(b1|b1) + (a0|a2).(b2|b0)            (lanes 0-31 | 32-63 of the k = 16 operand)
// instead of four of 64.  The weights are split once per weight update (format WK_GX3 of diagan_wino_weights_batched, or by this
// file's own kernel when the caller gave no hint); the activations in the loader, on their way to LDS (2 x (and, sub) + perms per
// element).  Same gather formula, prologue (none / ReLU) and epilogue (scale, bias, residual, backward mask) as conv_gemm_kernel.
//
// Workgroup = 64 pixels x 64 channels, 512 threads = two K-groups of four waves (2 x 2 tiles of 32 x 32), each group walks half of
// the K-steps (32 channels of one tap) with its own double-buffered LDS stage; the groups' accumulators meet in LDS at the end.
// LDS per group and stage: A and B as three piece planes [64 rows][32 channels bf16], 80-byte rows (conflict-free 16-byte reads).
// Roofline: bf16 MFMA (dense 2.5 PFLOP/s / 6 products = 417 TFLOP/s fp32-equivalent); HBM traffic = operands once.
//
// Measured (M = 8192 / 4096, N = 128, K = 1152; tools/probe/gemm_x3_time.py): 23.2 / 19.9 us against the fp32 kernel's 25.9 / 24.1.
// A K-step takes ~1900 cycles for 768 cycles of MFMA per SIMD: the split's vector work, the LDS staging and the MFMAs of the two
// waves of a SIMD add up.  Variants built on this kernel and measured, none kept: staging one step ahead only (25.5 us: every step
// waits for its loads); the staging's four parts pinned between the 8-channel blocks' MFMAs (25.4); three accumulator chains
// alone (no change: the chain is not the bound); every wave all four tiles of one 8-channel block (half the fragment reads: no
// change); producer waves 4-7 / consumer waves 0-3 with two LDS stages (25.1: the producers' 24 KB of LDS writes per step land
// on the consumers' reads; ablation: consumers alone 14.6 us, + global loads 14.9, + split 17.0, + LDS writes 23.4); the same
// with the weight fragments loaded straight from L2 into registers (31.8: one cache line per lane); `if`s around the MFMA
// block for odd step counts (26.5: hence the even-step requirement).
#include "conv_common.h"
#include "wino_weights.h"
#include <stdlib.h>

namespace diagan {

constexpr int GX_ROW = 40;                         // bf16 per LDS row: 32 channels + 16 bytes of padding
constexpr int GX_PLANE = 64 * GX_ROW;              // one piece plane of a tile (bf16 elements)
constexpr int GX_STAGE = 6 * GX_PLANE;             // A (3 planes) + B (3 planes)
constexpr int GX_LDS_BYTES = 4 * GX_STAGE * 2;     // 2 groups x 2 stages (120 KB: one workgroup per CU)

typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gx_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gx3_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wx, long quads) {
  const long plane = quads * 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long)gridDim.x * 256) gx3_split_quad(w, wx, i, plane);
}

template <int PRO>
__global__ __launch_bounds__(512) void conv_gemm_x3_kernel(const ConvGemmArgs a, const unsigned short* __restrict__ wx) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, wg = wave & 3, wm = wg >> 1, wn = wg & 1;
  const int tg = tid & 255, lrow = tg >> 2, lq = tg & 3;                  // loader: row / column of the tile, 8-channel chunk
  const int tiles_n = (g.Co + 63) >> 6;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
  unsigned short* const stage0 = lds + kg * 2 * GX_STAGE;

  // K-steps: 32 channels of one tap; this group's half (an even number of steps: gemm_x3_geom_ok)
  const int cpt = g.Ci >> 5;                       // K-steps per tap
  const int nk = g.R * g.S * cpt, kh = nk >> 1;
  const int k_begin = kg * kh, k_end = k_begin + kh;

  // loader state: this thread's pixel (GEMM row) and its byte offset at tap (0, 0)
  const int m = m0 + lrow;
  int oy = 0, ox = 0, pixbase = 0;
  const bool mv = m < a.M;
  {
    const unsigned t = fdiv((unsigned)(mv ? m : 0), a.dWo);
    ox = (mv ? m : 0) - (int)t * g.Wo;
    const unsigned b = fdiv(t, a.dHo);
    oy = (int)t - (int)b * g.Ho;
    pixbase = (((int)b * g.Hi + oy + g.off) * g.Wi + ox + g.off) * g.Ci * 4 + lq * 32;
  }
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Ci * 4u), 0x00020000);
  const long wplane = (long)g.Co * g.Kp;
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(wx), 0, (int)((unsigned)(3 * wplane) * 2u), 0x00020000);
  const int n = n0 + lrow;
  const unsigned wrow = n < g.Co ? (unsigned)n * (unsigned)g.Kp * 2u + (unsigned)lq * 16u : 0x80000000u;
  const unsigned wpl = (unsigned)wplane * 2u;

  // Two register sets of staged loads (named: the indices stay compile-time): the operands of step k + 2 are requested while step k
  // computes and written to LDS at the end of step k + 1 (one step ahead, every step waited ~2 k cycles for its loads and the
  // kernel ran exactly as fast as the fp32 one: 25.5 us)
  struct Staged { f32x4 a[2]; gx_u32x4 b[3]; };
  Staged sA, sB;
  auto load_step = [&](int kk, Staged& r) __attribute__((always_inline)) {
    const int tap = kk / cpt, c0 = (kk - tap * cpt) << 5;
    const int rr = tap / g.S, ss = tap - rr * g.S;
    const int iy = oy + g.off + rr * g.dr, ix = ox + g.off + ss * g.dr;
    const bool ok = mv && kk < k_end && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
    const unsigned off = (unsigned)(pixbase + ((rr * g.Wi + ss) * g.dr * g.Ci + c0) * 4) | (ok ? 0u : 0x80000000u);
    r.a[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
    r.a[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 16, 0));
    const unsigned wo = (kk < k_end ? wrow : 0x80000000u) + (unsigned)kk * 64u;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.b[p] = __builtin_bit_cast(gx_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wo + p * wpl, 0, 0));
  };
  const int sto = lrow * GX_ROW + lq * 8;          // this thread's slot in a plane (bf16 elements)
  auto store_step = [&](int buf, const Staged& r) __attribute__((always_inline)) {
    unsigned short* st = stage0 + buf * GX_STAGE;
    f32x4 v0 = r.a[0], v1 = r.a[1];
    if (PRO == PRO_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
    }
    u32x2 a0, a1, a2, b0, b1, b2;
    x3_split(v0, a0, a1, a2);
    x3_split(v1, b0, b1, b2);
    *reinterpret_cast<gx_u32x4*>(st + sto) = gx_u32x4{a0[0], a0[1], b0[0], b0[1]};
    *reinterpret_cast<gx_u32x4*>(st + GX_PLANE + sto) = gx_u32x4{a1[0], a1[1], b1[0], b1[1]};
    *reinterpret_cast<gx_u32x4*>(st + 2 * GX_PLANE + sto) = gx_u32x4{a2[0], a2[1], b2[0], b2[1]};
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<gx_u32x4*>(st + (3 + p) * GX_PLANE + sto) = r.b[p];
  };

  f32x16 acc3[3];                                  // one chain per product kind
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc3[p][e] = 0.f;
  const int fi = lane & 31, fh = lane >> 5;
  // fragment offsets (bf16 elements) inside a stage: A planes (0 | 1), (0 | 2); B planes 0, 1, (2 | 0)
  const int fa = (wm * 32 + fi) * GX_ROW, fb = (wn * 32 + fi) * GX_ROW;
  const int oa01 = (fh ? GX_PLANE : 0) + fa, oa02 = (fh ? 2 * GX_PLANE : 0) + fa;
  const int ob00 = 3 * GX_PLANE + fb, ob11 = 4 * GX_PLANE + fb, ob20 = (fh ? 3 : 5) * GX_PLANE + fb;
  auto mfmas = [&](int buf) __attribute__((always_inline)) {
    const unsigned short* st = stage0 + buf * GX_STAGE;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const gx_bf16x8 a01 = *reinterpret_cast<const gx_bf16x8*>(st + oa01 + c * 8);
      const gx_bf16x8 a02 = *reinterpret_cast<const gx_bf16x8*>(st + oa02 + c * 8);
      const gx_bf16x8 b00 = *reinterpret_cast<const gx_bf16x8*>(st + ob00 + c * 8);
      const gx_bf16x8 b11 = *reinterpret_cast<const gx_bf16x8*>(st + ob11 + c * 8);
      const gx_bf16x8 b20 = *reinterpret_cast<const gx_bf16x8*>(st + ob20 + c * 8);
      acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b00, acc3[0], 0, 0, 0);
      acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b11, acc3[1], 0, 0, 0);
      acc3[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a02, b20, acc3[2], 0, 0, 0);
    }
  };

  // (loads past the group's range are masked out: they return zeros without touching memory and are never stored)
  load_step(k_begin, sA);
  load_step(k_begin + 1, sB);
  store_step(0, sA);
  __syncthreads();
  for (int kk = k_begin; kk < k_end; kk += 2) {
    load_step(kk + 2, sA);
    mfmas(0);
    if (kk + 1 < k_end) store_step(1, sB);
    __syncthreads();
    if (kk + 1 >= k_end) break;
    load_step(kk + 3, sB);
    mfmas(1);
    if (kk + 2 < k_end) store_step(0, sA);
    __syncthreads();
  }
  f32x16 acc = acc3[0] + acc3[1] + acc3[2];

  // the second group's sums join the first's through LDS ([16][256] floats: lane-contiguous)
  float* xch = reinterpret_cast<float*>(lds);
  if (kg == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) xch[e * 256 + tg] = acc[e];
  }
  __syncthreads();
  if (kg == 1) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] += xch[e * 256 + tg];

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ----
  const float sc0 = a.scale0 ? a.scale0[0] : a.out_scale, sc1 = a.scale1 ? a.scale1[0] : a.out_scale;
  const int split = a.scale0 ? a.scale_split : 0x7fffffff;
  const unsigned rowbytes = (unsigned)g.Co * 4u, ybytes = (unsigned)a.M * rowbytes;
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.residual ? a.residual : a.y), 0, a.residual ? (int)ybytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.mask_src ? a.mask_src : a.y), 0, a.mask_src ? (int)ybytes : 0, 0x00020000);
  const float rfloor = a.res_relu ? 0.f : -__builtin_huge_valf();
  const int nc = n0 + wn * 32 + fi;
  const bool col_ok = nc < g.Co;
  const float bv = (a.bias && col_ok) ? a.bias[nc] : 0.f;
  const int mrow = m0 + wm * 32 + 4 * fh;
  const unsigned vbase = col_ok ? ((unsigned)mrow * g.Co + nc) * 4u : 0x80000000u;
  const bool hr = a.residual != nullptr, hm = a.mask_src != nullptr;
  float rres[16], rmsk[16];
  if (hr) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      rres[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vbase, (int)(((e & 3) + 8 * (e >> 2)) * rowbytes), 0));
  }
  if (hm) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      rmsk[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(msrc, vbase, (int)(((e & 3) + 8 * (e >> 2)) * rowbytes), 0));
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = (e & 3) + 8 * (e >> 2);
    float v = fmaf(acc[e], (mrow + k) < split ? sc0 : sc1, bv);
    if (hr) v += fmaxf(rres[e], rfloor);
    if (hm) v = rmsk[e] > 0.f ? v : v * a.mask_slope;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ysrc, vbase, (int)(k * rowbytes), 0);
  }
}

// floats of workspace the split weights need
long gemm_x3_ws_floats(int Co, int Kp) { return ((long)Co * Kp * 3 + 1) / 2; }

// geometry this kernel takes: 3x3 (any R x S) / stride 1 / no up-sampling gather with Ci a multiple of 32 (a K-step lies inside one
// tap), an even number of K-steps (two K-groups), Kp == R S Ci (no K padding), prologue none / ReLU
bool gemm_x3_geom_ok(const ConvGemmArgs& a) {
  const ConvGeom& g = a.g;
  return g.sy == 1 && g.up == 1 && (g.Ci & 31) == 0 && g.Kp == g.R * g.S * g.Ci && ((g.R * g.S * (g.Ci >> 5)) & 1) == 0 &&
         (g.Co & 3) == 0 && (a.pro_mode == PRO_NONE || a.pro_mode == PRO_RELU) && !a.stat_partials && a.pro_group_rows == 0 &&
         !a.res_up && (long)g.Co * g.Kp * 6 < (1L << 31);
}

int launch_gemm_x3(const ConvGemmArgs& a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  const long fl = gemm_x3_ws_floats(g.Co, g.Kp);
  const float* ready = wino_weights_ready(WK_GX3, 0, 1.f, fl);
  const unsigned short* wx = reinterpret_cast<const unsigned short*>(ready);
  if (!ready) {
    const long quads = (long)g.Co * g.Kp / 4;
    long blocks = (quads + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gx3_weight_kernel, dim3((int)blocks), dim3(256), 0, st, a.w, reinterpret_cast<unsigned short*>(ws), quads);
    wx = reinterpret_cast<const unsigned short*>(ws);
  }
  const int tiles = cdiv(a.M, 64) * cdiv(g.Co, 64);
  static FuncAttrLatch latch_none, latch_relu;
  if (a.pro_mode == PRO_RELU) {
    DG_LDS(latch_relu, conv_gemm_x3_kernel<PRO_RELU>, GX_LDS_BYTES);
    hipLaunchKernelGGL(conv_gemm_x3_kernel<PRO_RELU>, dim3(tiles), dim3(512), GX_LDS_BYTES, st, a, wx);
  } else {
    DG_LDS(latch_none, conv_gemm_x3_kernel<PRO_NONE>, GX_LDS_BYTES);
    hipLaunchKernelGGL(conv_gemm_x3_kernel<PRO_NONE>, dim3(tiles), dim3(512), GX_LDS_BYTES, st, a, wx);
  }
  return check_launch("conv_gemm_x3");
}

}  // namespace diagan
