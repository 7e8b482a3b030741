// Large implicit GEMMs on the bf16 matrix pipe with EXACTLY split operands: 128 x 128 output tiles (round 6).
//
// Replaces the fp32 `v_mfma_f32_32x32x2_f32` implicit GEMM (conv_gemm.hip, tile_cfg 1) for the launches no Winograd kernel
// takes and that are bound by the matrix pipe itself -- in StyleGAN2 (reference: diagan-pkg/diagan/models/stylegan2.py:224-265 the
// modulated convolution with its stride-2 transposed form, :553-595 / :597-614 the discriminator's blur + stride-2 convolutions):
// the 3x3 / stride 2 convolutions, the 2x2 / 2x1 / 1x2 / 1x1 parity classes of the stride-2 transposed gathers (ops/diffconv.py)
// and the 1x1 convolutions; 84 of a 277 ms iteration at 0.67-0.87 of the fp32 MFMA peak, i.e. the lever left is fewer pipe cycles.
//
// Arithmetic (as conv_gemm_x3.hip): every fp32 operand is the exact sum of three bf16 pieces (wino_weights.h: x3_split); six piece
// products, accumulated in fp32 by `v_mfma_f32_32x32x16_bf16`, reproduce the fp32 product to ~2^-23 -- per 8 channels THREE MFMAs
// of 32 cycles whose k = 16 holds two pieces x 8 channels
//     (a0|a1).(b0|b0) + (a0|a1).(b1|b1) + (a0|a2).(b2|b0)            (lanes 0-31 | 32-63 of the operand)
// instead of four fp32 MFMAs of 64 cycles: 2.67x fewer pipe cycles.  Weights are split once per launch (gx3_weight_kernel; format
// WK_GX3 when the caller hands them over), activations in the loader on their way to LDS.
//
// What is different from conv_gemm_x3.hip (the lone-tile kernel: 64 x 64 tiles, every wave one 32 x 32 tile, 5 fragment reads per
// 3 MFMAs = the LDS array's limit of ~2 ds_read_b128 per MFMA and SIMD, MI355X_MICROARCH.md "LDS"): here a wave owns 64 x 64 of a
// 128 x 128 tile, 10 fragment reads per 12 MFMAs, and TWO workgroups (61 440 bytes of LDS each, one stage) share a CU: while one
// splits and stores its next K-step (vector + LDS-write work between two barriers) the other's four waves run their 48 MFMAs.
//
// K-step = 32 channels of one tap.  LDS: A and B as three piece planes [128 rows][32 channels bf16], 80-byte rows (conflict-free
// 16-byte fragment reads).  Gather formula of conv_common.h with up == 1 (any stride, dr = +-1).  Epilogue: out_scale, bias,
// residual; optionally the output MAP of a parity class (ConvGemmArgs::map: pixel (b, oy, ox) of this launch's output grid is
// written to (b, mul * (oy - y0) + offy, mul * (ox - x0) + offx) of a larger tensor, pixels outside [y0, y1) x [x0, x1) are dropped) --
// the four dense sub-convolutions of a stride-2 transposed gather then interleave themselves and no copy pass follows.
// Roofline: bf16 MFMA (dense 2.5 PFLOP/s / 6 products = 416.7 TFLOP/s fp32-equivalent); HBM traffic = operands once.
#include "conv_common.h"
#include "wino_weights.h"
#include <stdlib.h>

namespace diagan {

constexpr int XB_ROW = 40;                         // bf16 per LDS row: 32 channels + 16 bytes of padding
constexpr int XB_PLANE = 128 * XB_ROW;             // one piece plane of a tile (bf16 elements)
constexpr int XB_STAGE = 6 * XB_PLANE;             // A (3 planes) + B (3 planes)
constexpr int XB_LDS_BYTES = XB_STAGE * 2;         // 61 440: two workgroups per CU

typedef __bf16 xb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned xb_u32x4 __attribute__((ext_vector_type(4)));
// Diagnostic builds only (tools/build_variant.sh <name> conv_gemm_x3b.hip "-DXB_ABL=<bits>"; results are then garbage): parts of the K
// loop removed at compile time so that their cost can be read off the launch time -- 1 MFMAs, 2 fragment reads, 4 global loads,
// 8 the split's arithmetic, 16 LDS writes, 32 barriers
#ifndef XB_ABL
#define XB_ABL 0
#endif
#ifndef XB_PRIO
#define XB_PRIO 1             // wave priority while a wave runs its MFMA phase (the partner workgroup's store phase yields the issue slots)
#endif

__global__ __launch_bounds__(256) void gx3b_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wx, long quads) {
  const long plane = quads * 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long)gridDim.x * 256) gx3_split_quad(w, wx, i, plane);
}

template <int PRO, bool MAP>
__global__ __launch_bounds__(256, 2) void conv_gemm_x3b_kernel(const ConvGemmArgs a, const unsigned short* __restrict__ wx) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 2, lq = tid & 3;                                // loader: row (and row + 64) of the tile, 8-channel chunk
  const int tiles_n = (g.Co + 127) >> 7;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;

  const int cpt = g.Ci >> 5;                       // K-steps per tap
  const int nk = g.R * g.S * cpt;

  // loader state: this thread's two pixels (GEMM rows lrow, lrow + 64): gathered coordinates at tap (0, 0) and the byte offset there
  int iy0[2], ix0[2], pixbase[2];
  bool mv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int m = m0 + lrow + 64 * h;
    mv[h] = m < a.M;
    const unsigned t = fdiv((unsigned)(mv[h] ? m : 0), a.dWo);
    const int ox = (mv[h] ? m : 0) - (int)t * g.Wo;
    const unsigned b = fdiv(t, a.dHo);
    const int oy = (int)t - (int)b * g.Ho;
    iy0[h] = oy * g.sy + g.off;
    ix0[h] = ox * g.sy + g.off;
    pixbase[h] = (((int)b * g.Hi + iy0[h]) * g.Wi + ix0[h]) * g.Ci * 4 + lq * 32;
  }
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Ci * 4u), 0x00020000);
  const long wplane = (long)g.Co * g.Kp;
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(wx), 0, (int)((unsigned)(3 * wplane) * 2u), 0x00020000);
  unsigned wrow[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = n0 + lrow + 64 * h;
    wrow[h] = n < g.Co ? (unsigned)n * (unsigned)g.Kp * 2u + (unsigned)lq * 16u : 0x80000000u;
  }
  const unsigned wpl = (unsigned)wplane * 2u;

  struct Staged { f32x4 a[2][2]; xb_u32x4 b[2][3]; };
  Staged sr;
  // (tap, channel block) of the step that is loaded next, kept incrementally
  int l_r = 0, l_s = 0, l_c = 0, l_k = 0;
  auto load_step = [&]() __attribute__((always_inline)) {
    if ((XB_ABL & 4) && l_k > 0) { ++l_k; return; }
    const int dy = l_r * g.dr, dx = l_s * g.dr;
    const int toff = ((dy * g.Wi + dx) * g.Ci + (l_c << 5)) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int iy = iy0[h] + dy, ix = ix0[h] + dx;
      const bool ok = mv[h] && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
      const unsigned off = (unsigned)(pixbase[h] + toff) | (ok ? 0u : 0x80000000u);
      sr.a[h][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
      sr.a[h][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 16, 0));
      const unsigned wo = wrow[h] + (unsigned)l_k * 64u;
#pragma unroll
      for (int p = 0; p < 3; ++p) sr.b[h][p] = __builtin_bit_cast(xb_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wo + p * wpl, 0, 0));
    }
    ++l_k;
    if (++l_c == cpt) {
      l_c = 0;
      if (++l_s == g.S) {
        l_s = 0;
        ++l_r;
      }
    }
  };
  const int sto = lrow * XB_ROW + lq * 8;          // this thread's slot in a plane (bf16 elements); second row: + 64 rows
  auto store_step = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned short* st = lds + sto + h * 64 * XB_ROW;
      f32x4 v0 = sr.a[h][0], v1 = sr.a[h][1];
      if (PRO == PRO_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
      }
      if (PRO == PRO_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = v0[e] > 0.f ? v0[e] : 0.2f * v0[e]; v1[e] = v1[e] > 0.f ? v1[e] : 0.2f * v1[e]; }
      }
      u32x2 a0, a1, a2, b0, b1, b2;
      if (XB_ABL & 8) {
        a0 = u32x2{__float_as_uint(v0[0]), __float_as_uint(v0[1])}; a1 = u32x2{__float_as_uint(v0[2]), __float_as_uint(v0[3])}; a2 = a0;
        b0 = u32x2{__float_as_uint(v1[0]), __float_as_uint(v1[1])}; b1 = u32x2{__float_as_uint(v1[2]), __float_as_uint(v1[3])}; b2 = b0;
      } else {
        x3_split(v0, a0, a1, a2);
        x3_split(v1, b0, b1, b2);
      }
      if ((XB_ABL & 16) && l_k > 1) {
        if (a0[0] == 0x12345678u && b2[1] == 0x9abcdef0u && sr.b[h][2][3] == 77u) *reinterpret_cast<xb_u32x4*>(st) = xb_u32x4{a1[0], a2[1], b0[0], b1[1]};
        continue;
      }
      *reinterpret_cast<xb_u32x4*>(st) = xb_u32x4{a0[0], a0[1], b0[0], b0[1]};
      *reinterpret_cast<xb_u32x4*>(st + XB_PLANE) = xb_u32x4{a1[0], a1[1], b1[0], b1[1]};
      *reinterpret_cast<xb_u32x4*>(st + 2 * XB_PLANE) = xb_u32x4{a2[0], a2[1], b2[0], b2[1]};
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<xb_u32x4*>(st + (3 + p) * XB_PLANE) = sr.b[h][p];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int fi = lane & 31, fh = lane >> 5;
  // fragment offsets (bf16 elements): A planes (0 | 1), (0 | 2); B planes 0, 1, (2 | 0)
  const int fa = (wm * 64 + fi) * XB_ROW, fb = (wn * 64 + fi) * XB_ROW;
  const int oa01 = (fh ? XB_PLANE : 0) + fa, oa02 = (fh ? 2 * XB_PLANE : 0) + fa;
  const int ob00 = 3 * XB_PLANE + fb, ob11 = 4 * XB_PLANE + fb, ob20 = (fh ? 3 : 5) * XB_PLANE + fb;
  // fragments of one 8-channel block, double-buffered in registers: slots 0 a01[0], 1 b00[0], 2 b00[1], 3 a01[1], 4 b11[0], 5 b11[1],
  // 6 a02[0], 7 b20[0], 8 b20[1], 9 a02[1] -- the order of their first use.  Block c + 1's ten reads are issued ONE BEHIND EACH of
  // block c's first ten MFMAs (a scheduling barrier after every pair keeps that order): left to itself the compiler reads a block's
  // fragments right before their first use and waits for them with the pipe idle (five waits per block); ten reads in a row in
  // front of the MFMAs keep the wave -- in-order issue -- away from the matrix pipe for their ~150 issue cycles.
  xb_bf16x8 fr[2][10];
  auto read_slot = [&](int buf, int sl, int c) __attribute__((always_inline)) {
    const int t = (sl == 2 || sl == 3 || sl == 5 || sl == 8 || sl == 9) ? 32 * XB_ROW : 0;
    const int base = (sl == 0 || sl == 3) ? oa01 : (sl == 6 || sl == 9) ? oa02 : (sl == 1 || sl == 2) ? ob00 : (sl == 4 || sl == 5) ? ob11 : ob20;
    fr[buf][sl] = *reinterpret_cast<const xb_bf16x8*>(lds + base + t + c * 8);
  };
  bool first_step = true;
  auto mfmas = [&]() __attribute__((always_inline)) {
    if (!(XB_ABL & 2) || first_step) {
#pragma unroll
      for (int sl = 0; sl < 10; ++sl) read_slot(0, sl, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(XB_PRIO);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cur = c & 1, nxt = cur ^ 1;
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int p = q >> 2, i = (q >> 1) & 1, j = q & 1;
        const int sa = p == 2 ? (i ? 9 : 6) : (i ? 3 : 0);
        const int sb = p == 0 ? 1 + j : (p == 1 ? 4 + j : 7 + j);
        if (!(XB_ABL & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[cur][sa], fr[cur][sb], acc[i][j], 0, 0, 0);
        if (c < 3 && q < 10 && (!(XB_ABL & 2) || first_step)) read_slot(nxt, q, c + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (XB_ABL & 1) {         // (keep the fragments alive)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (__builtin_bit_cast(xb_u32x4, fr[t][0])[0] == 0x12345678u && __builtin_bit_cast(xb_u32x4, fr[t][8])[3] == 7u) acc[0][0][t] += 1.f;
    }
    first_step = false;
  };

  load_step();
  store_step();
  __syncthreads();
  for (int kk = 0; kk < nk; ++kk) {
    const bool more = kk + 1 < nk;
    if (more) load_step();
    mfmas();
    if (!(XB_ABL & 32)) __syncthreads();
    if (more) {
      store_step();
      if (!(XB_ABL & 32)) __syncthreads();
    }
  }

  // ---- epilogue ----
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).  A wave turns its 64 x 64 block, one
  // 32-column half at a time, through a private [64 rows][32 columns] LDS image (un-padded: conflict-free both ways, see below) so
  // that a lane stores 16 bytes and eight lanes a 128-byte row segment: 16 store instructions per wave instead of 64 of 4 bytes
  // (the 4-byte form was 0.3 of a 1.8 ms launch: store-issue bound).
  // banks: writes -- 32 lanes of a row, consecutive columns; reads -- the 16-lane groups of ds_read_b128 hold rows r .. r + 3 in
  // the column halves (0, 1, 1, 0) / (1, 0, 0, 1): 32-float rows put those on four disjoint quarters of the 64 banks.
  const float sc = a.out_scale;
  const bool hr = a.residual != nullptr;
  const OutMap& mp = a.map;
  const long ypix = MAP ? (long)g.B * mp.OH * mp.OW : (long)a.M;
  const unsigned ybytes = (unsigned)(ypix * g.Co * 4);
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(hr ? a.residual : a.y), 0, hr ? (int)ybytes : 0, 0x00020000);
  float* xim = reinterpret_cast<float*>(lds) + wave * 2048;             // (the K loop ended behind a barrier: the stage is free)
  const int er = lane >> 3, ec = (lane & 7) * 4;                        // read-back role: row er + 8 k, columns ec .. ec + 3
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) xim[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + fi] = acc[i][j][e];
    const int nc = n0 + wn * 64 + j * 32 + ec;
    const bool col_ok = nc < g.Co;                                      // (Co % 4 == 0: the quad is whole)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias && col_ok) bv = *reinterpret_cast<const f32x4*>(a.bias + nc);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int rl = er + 8 * k;
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(xim + rl * 32 + ec);
      const int m = m0 + wm * 64 + rl;
      unsigned voff;
      if (MAP) {
        const bool in = m < a.M;
        const unsigned t = fdiv((unsigned)(in ? m : 0), a.dWo);
        const int ox = (in ? m : 0) - (int)t * g.Wo;
        const unsigned b = fdiv(t, a.dHo);
        const int oy = (int)t - (int)b * g.Ho;
        const bool ok = in && col_ok && oy >= mp.y0 && oy < mp.y1 && ox >= mp.x0 && ox < mp.x1;
        const int py = mp.mul * (oy - mp.y0) + mp.offy, px = mp.mul * (ox - mp.x0) + mp.offx;
        voff = ok ? (unsigned)((((int)b * mp.OH + py) * mp.OW + px) * g.Co + nc) * 4u : 0x80000000u;
      } else {
        voff = (col_ok && m < a.M) ? ((unsigned)m * g.Co + nc) * 4u : 0x80000000u;
      }
      f32x4 v = v4 * sc + bv;
      if (hr) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xb_u32x4, v), ysrc, voff, 0, 0);
    }
  }
}

// floats of workspace the split weights need
long gemm_x3b_ws_floats(int Co, int Kp) { return ((long)Co * Kp * 3 + 1) / 2; }

// geometry this kernel takes: no up-sampling gather (any stride, dr = +-1), Ci a multiple of 32 (a K-step lies inside one tap),
// Kp == R S Ci (no K padding), prologue none / ReLU / leaky ReLU, plain epilogue (out_scale, bias, full-resolution residual)
bool gemm_x3b_geom_ok(const ConvGemmArgs& a) {
  const ConvGeom& g = a.g;
  return g.up == 1 && (g.Ci & 31) == 0 && g.Kp == g.R * g.S * g.Ci && (g.Co & 3) == 0 &&
         (a.pro_mode == PRO_NONE || a.pro_mode == PRO_RELU || a.pro_mode == PRO_LRELU) && !a.stat_partials && a.pro_group_rows == 0 &&
         !a.res_up && !a.res_relu && !a.mask_src && !a.scale0 && (long)g.Co * g.Kp * 6 < (1L << 31);
}

template <int PRO, bool MAP>
static void launch_x3b_one(const ConvGemmArgs& a, const unsigned short* wx, int tiles, hipStream_t st) {
  auto kern = conv_gemm_x3b_kernel<PRO, MAP>;
  static int attr_dev = -1;                          // (the attribute is per device: re-set when the current device changes)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, XB_LDS_BYTES);
    attr_dev = dev;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), XB_LDS_BYTES, st, a, wx);
}

int launch_gemm_x3b(const ConvGemmArgs& a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  const long fl = gemm_x3b_ws_floats(g.Co, g.Kp);
  const float* ready = wino_weights_ready(WK_GX3, 0, 1.f, fl);
  const unsigned short* wx = reinterpret_cast<const unsigned short*>(ready);
  if (!ready) {
    const long quads = (long)g.Co * g.Kp / 4;
    long blocks = (quads + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gx3b_weight_kernel, dim3((int)blocks), dim3(256), 0, st, a.w, reinterpret_cast<unsigned short*>(ws), quads);
    wx = reinterpret_cast<const unsigned short*>(ws);
  }
  const int tiles = cdiv(a.M, 128) * cdiv(g.Co, 128);
  const bool map = a.map.mul != 0;
  switch (a.pro_mode) {
    case PRO_RELU: map ? launch_x3b_one<PRO_RELU, true>(a, wx, tiles, st) : launch_x3b_one<PRO_RELU, false>(a, wx, tiles, st); break;
    case PRO_LRELU: map ? launch_x3b_one<PRO_LRELU, true>(a, wx, tiles, st) : launch_x3b_one<PRO_LRELU, false>(a, wx, tiles, st); break;
    default: map ? launch_x3b_one<PRO_NONE, true>(a, wx, tiles, st) : launch_x3b_one<PRO_NONE, false>(a, wx, tiles, st);
  }
  return check_launch("conv_gemm_x3b");
}

}  // namespace diagan
