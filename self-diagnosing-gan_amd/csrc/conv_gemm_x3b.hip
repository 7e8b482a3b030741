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
// 8 the split's arithmetic, 16 LDS writes, 32 barriers; 64 (first form only; results stay VALID at two-piece accuracy): the third piece
// pair dropped everywhere -- its plane loads, LDS writes, fragment reads and MFMAs -- i.e. (a0 + a1)(b0 + b1) in two MFMAs per 8 channels,
// to price that trade (profiles/r06_x3b.md)
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

// TWO (round 6, opt-in: DIAGAN_X3_PIECES=2 / diagan_conv_gemm_set_x3_pieces): the third piece pair dropped everywhere -- its plane loads, LDS
// writes, fragment reads and MFMAs -- i.e. (a0 + a1)(b0 + b1) in two MFMAs per 8 channels: ~2^-16 operands, 1.5e-5-2e-5 of the output scale
// per layer (the F(4x4) Winograd layers' class) for 1.46x shorter launches.  NOT the default: the default stays fp32-grade.
template <int PRO, bool MAP, bool TWO = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_x3b_kernel(const ConvGemmArgs a, const OutMap mp, const unsigned short* __restrict__ wx) {
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
      for (int p = 0; p < ((TWO || (XB_ABL & 64)) ? 2 : 3); ++p) sr.b[h][p] = __builtin_bit_cast(xb_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wo + p * wpl, 0, 0));
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
      if (!(TWO || (XB_ABL & 64))) *reinterpret_cast<xb_u32x4*>(st + 2 * XB_PLANE) = xb_u32x4{a2[0], a2[1], b2[0], b2[1]};
#pragma unroll
      for (int p = 0; p < ((TWO || (XB_ABL & 64)) ? 2 : 3); ++p) *reinterpret_cast<xb_u32x4*>(st + (3 + p) * XB_PLANE) = sr.b[h][p];
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
      for (int sl = 0; sl < ((TWO || (XB_ABL & 64)) ? 6 : 10); ++sl) read_slot(0, sl, 0);
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
        if ((TWO || (XB_ABL & 64)) && p == 2) continue;
        if (!(XB_ABL & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[cur][sa], fr[cur][sb], acc[i][j], 0, 0, 0);
        if (c < 3 && q < ((TWO || (XB_ABL & 64)) ? 6 : 10) && (!(XB_ABL & 2) || first_step)) read_slot(nxt, q, c + 1);
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

#ifdef XB_STAMP
  // diagnostic build (tools/build_variant.sh stamp conv_gemm_x3b.hip "-DXB_STAMP"): cycles per wave in the phases of the K loop,
  // written to a.stamps[(workgroup * 4 + wave) * 8 + phase]: 0 load issue, 1 MFMA phase, 2 first barrier, 3 wait for the loads,
  // 4 split + LDS writes, 5 second barrier, 6 whole loop, 7 epilogue
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tl = __builtin_amdgcn_s_memtime();
  const unsigned long long t_loop0 = tl;
#define XB_TICK(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tl; tl = now_; }
#else
#define XB_TICK(i)
#endif
  load_step();
  store_step();
  __syncthreads();
  XB_TICK(0);
  for (int kk = 0; kk < nk; ++kk) {
    const bool more = kk + 1 < nk;
    if (more) load_step();
    XB_TICK(0);
    mfmas();
#ifdef XB_STAMP
    asm volatile("s_nop 0" ::: "memory");
#endif
    XB_TICK(1);
    if (!(XB_ABL & 32)) __syncthreads();
    XB_TICK(2);
    if (more) {
#ifdef XB_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      XB_TICK(3);
      store_step();
#ifdef XB_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      XB_TICK(4);
      if (!(XB_ABL & 32)) __syncthreads();
      XB_TICK(5);
    }
  }
#ifdef XB_STAMP
  const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue ----
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).  A wave turns its 64 x 64 block, one
  // 32-column half at a time, through a private [64 rows][32 columns] LDS image (un-padded: conflict-free both ways, see below) so
  // that a lane stores 16 bytes and eight lanes a 128-byte row segment: 16 store instructions per wave instead of 64 of 4 bytes
  // (the 4-byte form was 0.3 of a 1.8 ms launch: store-issue bound).
  // banks: writes -- 32 lanes of a row, consecutive columns; reads -- the 16-lane groups of ds_read_b128 hold rows r .. r + 3 in
  // the column halves (0, 1, 1, 0) / (1, 0, 0, 1): 32-float rows put those on four disjoint quarters of the 64 banks.
  const float sc = a.out_scale;
  const bool hr = a.residual != nullptr;
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
#ifdef XB_STAMP
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 6; ++i) o[i] = tacc[i];
    o[6] = t_loop1 - t_loop0;
    o[7] = __builtin_amdgcn_s_memtime() - t_loop1;
  }
#endif
}

// ---- second form (large launches): producer / consumer waves inside ONE workgroup per CU, 256 x 128 tiles ------------------------------
// The stamps of the kernel above (tools/probe/x3b_stamps.py, 3x3 / stride 2, 128 -> 256 at 256 x 256) read, per K-step and wave:
// 1150 cycles to get the ten global loads accepted, 1900 in the MFMA phase (1536 of matrix pipe), 1180 for the split and the LDS writes,
// 460 at the two barriers -- 4950 with the pipe busy 0.62 of the time.  A first producer / consumer split of the same 128 x 128 tile
// (waves 0-3 MFMAs only, waves 4-7 loads + split + writes, two LDS stages) ran no faster, and ITS stamps said why: a vector-memory
// instruction of 1 KB -- load or LDS-DMA, full cache lines or not -- takes a wave ~130 cycles to issue with four waves issuing, i.e.
// the CU takes in ~31 bytes per cycle from L2, and a 128 x 128 K-step needs 48 KB (16 KB of fp32 activations + 32 KB of split
// weights): 1550 cycles of that path for 1536 of matrix pipe.  So the second form halves the weight bytes per MFMA:
//   * a workgroup owns 256 pixels x 128 columns as two 128-row HALVES that take turns ("sub-steps") and share a K-step's weights;
//   * waves 0-3 (one per SIMD) only read fragments and issue MFMAs (two accumulator sets);
//   * waves 4-7 -- their SIMD partners -- fill the other half's A buffer for the next sub-step: full-line loads (eight lanes = one
//     pixel's 128 bytes; two register sets, so a load has a whole sub-step to land), split, 8-byte LDS writes; and copy the NEXT
//     K-step's weights -- which the per-launch kernel below has already laid out in global memory AS the padded LDS image of every
//     (column tile, K-step) -- by LDS-DMA, half an image per sub-step: linear 1 KB copies, no register, no LDS-write instruction;
//   * one raw s_barrier per sub-step; the producers' waits are counted (vector-memory operations of a wave complete in order:
//     DMA pieces are issued BEFORE the loads of the same sub-step, so `vmcnt(4)` retires them and leaves those loads in flight).
// LDS: A buffers 2 x 30 720 (half 0 | half 1), B buffers 2 x 32 768 (K-step parity) = 126 976 bytes.
constexpr int XB2_AIMG = 3 * XB_PLANE * 2;         // bytes of the A planes of one half (30 720)
constexpr int XB2_BIMG = 32768;                    // bytes of a B image: 30 720 of planes + 2 KB so that every producer wave copies 8 KB
constexpr int XB2_BOFF = 2 * XB2_AIMG;             // first B buffer
constexpr int XB2_LDS_BYTES = XB2_BOFF + 2 * XB2_BIMG;       // 126 976

// packed fp32 weights [Co][Kp] -> for every (128-column tile, K-step) the LDS image [piece][128 rows][40 bf16] (rows behind Co: zeros)
__global__ __launch_bounds__(256) void gx3b2_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, int Co, int Kp,
                                                           int nk) {
  const int nt = blockIdx.y, ks = blockIdx.x, row = threadIdx.x >> 1, q0 = (threadIdx.x & 1) * 4;
  const int n = nt * 128 + row;
  unsigned short* dst = img + ((long)nt * nk + ks) * (XB2_BIMG / 2) + row * XB_ROW;
#pragma unroll
  for (int q = q0; q < q0 + 4; ++q) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < Co) v = *reinterpret_cast<const f32x4*>(w + (long)n * Kp + ks * 32 + q * 4);
    u32x2 p0, p1, p2;
    x3_split(v, p0, p1, p2);
    *reinterpret_cast<u32x2*>(dst + q * 4) = p0;
    *reinterpret_cast<u32x2*>(dst + XB_PLANE + q * 4) = p1;
    *reinterpret_cast<u32x2*>(dst + 2 * XB_PLANE + q * 4) = p2;
  }
}

#define XB2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define XB2_WAIT(n, r) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory")
#ifdef XB_STAMP
#define XB2_TICK(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tl; tl = now_; }
#else
#define XB2_TICK(i)
#endif

template <int PRO, bool MAP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_gemm_x3b2_kernel(const ConvGemmArgs a, const OutMap mp, const unsigned short* __restrict__ wimg) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  char* const lds8 = reinterpret_cast<char*>(lds);
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.Co + 127) >> 7;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = tile % tiles_n;
  const int m0 = (tile / tiles_n) * 256, n0 = nt * 128;
  const int cpt = g.Ci >> 5;                       // K-steps per tap
  const int nk = g.R * g.S * cpt, ns = 2 * nk;     // K-steps, sub-steps (K-step k, half h): s = 2 k + h

  if (wave >= 4) {
    // ================= producers =================
    const int ptid = tid - 256, pw = wave - 4;
    const int prow = ptid >> 3, pc = ptid & 7;                            // rows prow + 32 j of a half, 4-channel chunk
    int iy0[2][4], ix0[2][4], pixbase[2][4];
    bool mv[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + 128 * h + prow + 32 * j;
        mv[h][j] = m < a.M;
        const unsigned t = fdiv((unsigned)(mv[h][j] ? m : 0), a.dWo);
        const int ox = (mv[h][j] ? m : 0) - (int)t * g.Wo;
        const unsigned b = fdiv(t, a.dHo);
        const int oy = (int)t - (int)b * g.Ho;
        iy0[h][j] = oy * g.sy + g.off;
        ix0[h][j] = ox * g.sy + g.off;
        pixbase[h][j] = (((int)b * g.Hi + iy0[h][j]) * g.Wi + ix0[h][j]) * g.Ci * 4 + pc * 16;
      }
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 xsrc;
    {
      const unsigned long long xb = (unsigned long long)a.x;
      xsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
      xsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffffu));
      xsrc[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)g.B * g.Hi * g.Wi * g.Ci * 4u));
      xsrc[3] = 0x00020000;
    }
    const int soff = 0;
    f32x4 ra0[4], ra1[4];                          // activations of half 0 / half 1 on their way to LDS
    int l_r = 0, l_s = 0, l_c = 0;                 // (tap, channel block) of the K-step whose halves are requested next
    // (inline assembly: the compiler's wait-count pass would put vmcnt(0) in front of the first use of a loaded register, draining the
    //  DMA in flight too; this way the only waits are the counted ones below)
    auto issue_a = [&](f32x4 (&r)[4], int h) __attribute__((always_inline)) {
      const int dy = l_r * g.dr, dx = l_s * g.dr;
      const int toff = ((dy * g.Wi + dx) * g.Ci + (l_c << 5)) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = iy0[h][j] + dy, ix = ix0[h][j] + dx;
        const bool ok = mv[h][j] && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
        const unsigned off = (unsigned)(pixbase[h][j] + toff) | (ok ? 0u : 0x80000000u);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[j]) : "v"(off), "s"(xsrc), "s"(soff) : "memory");
      }
      if (h == 1 && ++l_c == cpt) {                // (both halves of a K-step requested: on to the next one)
        l_c = 0;
        if (++l_s == g.S) {
          l_s = 0;
          ++l_r;
        }
      }
    };
    const char* const wsrc = reinterpret_cast<const char*>(wimg) + (long)nt * nk * XB2_BIMG + pw * 8192 + lane * 16;
    // pieces [4 part, 4 part + 4) of this wave's eight 1 KB pieces of K-step ks's image
    auto issue_b = [&](int ks, int part) __attribute__((always_inline)) {
      const char* src = wsrc + (long)ks * XB2_BIMG + part * 4096;
      char* dst = lds8 + XB2_BOFF + (ks & 1) * XB2_BIMG + pw * 8192 + part * 4096;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + t * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + t * 1024), 16, 0, 0);
    };
    const int sto = prow * XB_ROW + pc * 4;        // bf16 elements; row j: + 32 j rows
    auto split_store = [&](f32x4 (&r)[4], int h) __attribute__((always_inline)) {
      unsigned short* st0 = reinterpret_cast<unsigned short*>(lds8 + h * XB2_AIMG) + sto;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned short* st = st0 + j * 32 * XB_ROW;
        f32x4 v = r[j];
        if (PRO == PRO_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (PRO == PRO_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
        }
        u32x2 p0, p1, p2;
        x3_split(v, p0, p1, p2);
        *reinterpret_cast<u32x2*>(st) = p0;
        *reinterpret_cast<u32x2*>(st + XB_PLANE) = p1;
        *reinterpret_cast<u32x2*>(st + 2 * XB_PLANE) = p2;
      }
    };
#ifdef XB_STAMP
    // diagnostic build: a.stamps[(workgroup * 8 + wave) * 8 + i]; producers: 0 DMA issue, 1 load issue, 2 wait for the loads, 3 split +
    // LDS writes (drained), 4 wait for the DMA, 5 barrier; consumers: 0 fragment reads + MFMAs, 5 barrier, 7 epilogue
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tl = __builtin_amdgcn_s_memtime();
#endif
    // sub-step s = 2 k + H of the producers, k + 1 < nk: request sub-step s + 2 (same half H) into `rl`, copy half H of K-step k + 1's
    // weight image, then write sub-step s + 1 (the other half, waiting in `rs`) into its A buffer.
    // (No branch in here: the first build chose between counted waits inside this body, the compiler merged the two paths' common
    //  split code and copied the still-in-flight registers of `rs` to the merged copy IN FRONT of the wait -- results wrong once in a
    //  few launches.  The last K-step, which requests nothing, is peeled off below instead.)
    auto fill = [&](int s, f32x4 (&rs)[4], f32x4 (&rl)[4], int H) __attribute__((always_inline)) {
      issue_b((s >> 1) + 1, H);                    // (that B buffer was read in K-step k - 1)
      XB2_TICK(0);
      issue_a(rl, H);
      XB2_TICK(1);
      XB2_WAIT(8, rs);                             // sub-step s + 1's loads: older than the 4 DMA pieces and the 4 loads just issued
      XB2_TICK(2);
      split_store(rs, H ^ 1);
#ifdef XB_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      XB2_TICK(3);
      if (H == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // K-step k + 1's image has landed; the 4 loads stay in flight
      XB2_TICK(4);
      XB2_BARRIER();
      XB2_TICK(5);
    };
    issue_b(0, 0);
    issue_b(0, 1);
    issue_a(ra0, 0);
    issue_a(ra1, 1);
    XB2_WAIT(4, ra0);                              // the image of K-step 0 and half 0 have landed
    split_store(ra0, 0);
    XB2_BARRIER();                                 // sub-step 0 is ready
#ifdef XB_STAMP
    tl = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s + 2 < ns; s += 2) {
      fill(s, ra1, ra0, 0);
      fill(s + 1, ra0, ra1, 1);
    }
    XB2_WAIT(0, ra1);                              // the last K-step: its second half is waiting in ra1, nothing is requested any more
    split_store(ra1, 1);
    XB2_BARRIER();
    XB2_BARRIER();
#ifdef XB_STAMP
    if (a.stamps && lane == 0) {
      unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int i = 0; i < 6; ++i) o[i] = tacc[i];
    }
#endif
    return;
  }

  // ================= consumers =================
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2][2];                             // [half][row tile][column tile]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.f;
  const int fi = lane & 31, fh = lane >> 5;
  const int fa = (wm * 64 + fi) * XB_ROW, fb = XB2_BOFF / 2 + (wn * 64 + fi) * XB_ROW;
  const int oa01 = (fh ? XB_PLANE : 0) + fa, oa02 = (fh ? 2 * XB_PLANE : 0) + fa;
  const int ob00 = fb, ob11 = XB_PLANE + fb, ob20 = (fh ? 0 : 2) * XB_PLANE + fb;
  xb_bf16x8 fr[2][10];
  // slots 0, 3, 6, 9 are A fragments (buffer of the half), the others B fragments (buffer of the K-step's parity)
  auto read_slot = [&](const unsigned short* sa, const unsigned short* sb, int buf, int sl, int c) __attribute__((always_inline)) {
    const int t = (sl == 2 || sl == 3 || sl == 5 || sl == 8 || sl == 9) ? 32 * XB_ROW : 0;
    const bool isa = sl == 0 || sl == 3 || sl == 6 || sl == 9;
    const int base = (sl == 0 || sl == 3) ? oa01 : (sl == 6 || sl == 9) ? oa02 : (sl == 1 || sl == 2) ? ob00 : (sl == 4 || sl == 5) ? ob11 : ob20;
    fr[buf][sl] = *reinterpret_cast<const xb_bf16x8*>((isa ? sa : sb) + base + t + c * 8);
  };
#ifdef XB_STAMP
  unsigned long long cacc[2] = {0, 0};
  unsigned long long ctl = 0;
#endif
  auto half_step = [&](int k, int H) __attribute__((always_inline)) {
    const unsigned short* sa = lds + H * (XB2_AIMG / 2);
    const unsigned short* sb = lds + (k & 1) * (XB2_BIMG / 2);
#pragma unroll
    for (int sl = 0; sl < 10; ++sl) read_slot(sa, sb, 0, sl, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cur = c & 1, nxt = cur ^ 1;
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int p = q >> 2, i = (q >> 1) & 1, j = q & 1;
        const int fa_ = p == 2 ? (i ? 9 : 6) : (i ? 3 : 0);
        const int fb_ = p == 0 ? 1 + j : (p == 1 ? 4 + j : 7 + j);
        acc[H][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[cur][fa_], fr[cur][fb_], acc[H][i][j], 0, 0, 0);
        if (c < 3 && q < 10) read_slot(sa, sb, nxt, q, c + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef XB_STAMP
    { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); cacc[0] += now_ - ctl; ctl = now_; }
#endif
    XB2_BARRIER();
#ifdef XB_STAMP
    { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); cacc[1] += now_ - ctl; ctl = now_; }
#endif
  };
  __builtin_amdgcn_s_setprio(XB_PRIO);
  XB2_BARRIER();                                   // sub-step 0 is ready
#ifdef XB_STAMP
  ctl = __builtin_amdgcn_s_memtime();
#endif
  for (int k = 0; k < nk; ++k) {
    half_step(k, 0);
    half_step(k, 1);
  }
  __builtin_amdgcn_s_setprio(0);
#ifdef XB_STAMP
  const unsigned long long t_epi0 = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue (as above; the four consumer waves, through their private images in the now idle A buffers) ----
  const float sc = a.out_scale;
  const bool hr = a.residual != nullptr;
  const long ypix = MAP ? (long)g.B * mp.OH * mp.OW : (long)a.M;
  const unsigned ybytes = (unsigned)(ypix * g.Co * 4);
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(hr ? a.residual : a.y), 0, hr ? (int)ybytes : 0, 0x00020000);
  float* xim = reinterpret_cast<float*>(lds) + wave * 2048;
  const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) xim[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + fi] = acc[h][i][j][e];
      const int nc = n0 + wn * 64 + j * 32 + ec;
      const bool col_ok = nc < g.Co;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias && col_ok) bv = *reinterpret_cast<const f32x4*>(a.bias + nc);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int rl = er + 8 * k;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(xim + rl * 32 + ec);
        const int m = m0 + 128 * h + wm * 64 + rl;
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
#ifdef XB_STAMP
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = cacc[0];
    o[5] = cacc[1];
    o[7] = __builtin_amdgcn_s_memtime() - t_epi0;
  }
#endif
}

// floats of workspace the split weights need (the larger of the two forms' formats: planes [piece][Co][Kp] / one 32 KB LDS image per
// 128-column tile and K-step)
long gemm_x3b_ws_floats(int Co, int Kp) {
  const long f1 = ((long)Co * Kp * 3 + 1) / 2, f2 = (long)cdiv(Co, 128) * (Kp / 32) * (XB2_BIMG / 4);
  return f1 > f2 ? f1 : f2;
}

// geometry this kernel takes: no up-sampling gather (any stride, dr = +-1), Ci a multiple of 32 (a K-step lies inside one tap),
// Kp == R S Ci (no K padding), prologue none / ReLU / leaky ReLU, plain epilogue (out_scale, bias, full-resolution residual)
bool gemm_x3b_geom_ok(const ConvGemmArgs& a) {
  const ConvGeom& g = a.g;
  return g.up == 1 && (g.Ci & 31) == 0 && g.Kp == g.R * g.S * g.Ci && (g.Co & 3) == 0 &&
         (a.pro_mode == PRO_NONE || a.pro_mode == PRO_RELU || a.pro_mode == PRO_LRELU) && !a.stat_partials && a.pro_group_rows == 0 &&
         !a.res_up && !a.res_relu && !a.mask_src && !a.scale0 && (long)g.Co * g.Kp * 6 < (1L << 31);
}

// which form: 2 = producer / consumer waves on 256 x 128 tiles (one workgroup per CU), 1 = 128 x 128 tiles, two workgroups per CU.
// DIAGAN_GEMM_X3B_FORM forces one; default: form 2 where its tiles fill the chip at least `min_tiles2` / 256 times (the one-workgroup
// form has nothing to run under its epilogue or beside a half-empty last round).  Form 2's launches are 5-14 % shorter one by one
// (profiles/r06_x3b.md); what that is worth to the StyleGAN2 iteration depends on the box: same-box pairs of the final build read
// form 1 everywhere +0.8 % and +1.6 % on two boxes, -1.8 % and -1.3 ... -2.4 % on two others (profiles/r06_raw/sg2_ab*.txt) -- inside the
// spread of this pool.
// pieces per operand of the LARGE split-operand kernels (this file's first form, conv_wgrad_x3.hip): 3 = fp32-grade (default), 2 = opt-in
static int g_x3_pieces = 0;                        // 0: the environment's DIAGAN_X3_PIECES (default 3)
void x3_set_pieces(int n) { g_x3_pieces = (n == 2 || n == 3) ? n : 0; }
int x3_pieces() {
  static const int env = getenv("DIAGAN_X3_PIECES") ? atoi(getenv("DIAGAN_X3_PIECES")) : 3;
  const int n = g_x3_pieces ? g_x3_pieces : env;
  return n == 2 ? 2 : 3;
}
static int g_x3b_form = 0;                          // diagnostics / tests: 1 / 2 force a form (diagan_conv_gemm_x3b_force_form), 0: automatic
void gemm_x3b_force_form(int form) { g_x3b_form = form; }
static int x3b_form(long tiles2, int nk) {
  static const int env0 = getenv("DIAGAN_GEMM_X3B_FORM") ? atoi(getenv("DIAGAN_GEMM_X3B_FORM")) : 0;
  const int env = g_x3b_form ? g_x3b_form : env0;
  static const int min_tiles2 = getenv("DIAGAN_GEMM_X3B_FORM2_TILES") ? atoi(getenv("DIAGAN_GEMM_X3B_FORM2_TILES")) : 512;
  if (x3_pieces() == 2) return 1;                    // (the two-piece mode exists in the first form only)
  if (env == 1 || env == 2) return env;
  // (K loops of fewer than 8 steps: 1x1 / stride 2 from 128 channels, 316 us against 288 -- the epilogue is a quarter of such a tile)
  return tiles2 >= min_tiles2 && nk >= 8 ? 2 : 1;
}

template <int PRO, bool MAP>
static int launch_x3b_two(const ConvGemmArgs& a, const OutMap& mp, const unsigned short* wimg, int tiles, hipStream_t st) {
  auto kern = conv_gemm_x3b2_kernel<PRO, MAP>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, XB2_LDS_BYTES);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), XB2_LDS_BYTES, st, a, mp, wimg);
  return DIAGAN_OK;
}

template <int PRO, bool MAP>
static int launch_x3b_one(const ConvGemmArgs& a, const OutMap& mp, const unsigned short* wx, int tiles, hipStream_t st) {
  if (x3_pieces() == 2) {
    auto kern2 = conv_gemm_x3b_kernel<PRO, MAP, true>;
    static FuncAttrLatch latch2;
    DG_LDS(latch2, kern2, XB_LDS_BYTES);
    hipLaunchKernelGGL(kern2, dim3(tiles), dim3(256), XB_LDS_BYTES, st, a, mp, wx);
    return DIAGAN_OK;
  }
  auto kern = conv_gemm_x3b_kernel<PRO, MAP>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, XB_LDS_BYTES);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), XB_LDS_BYTES, st, a, mp, wx);
  return DIAGAN_OK;
}

int launch_gemm_x3b(const ConvGemmArgs& a, const OutMap& mp, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  const int tiles = cdiv(a.M, 128) * cdiv(g.Co, 128);
  const bool map = mp.mul != 0;
  const int nk = g.Kp / 32;
  const long tiles2 = (long)cdiv(a.M, 256) * cdiv(g.Co, 128);
  if (x3b_form(tiles2, nk) == 2) {
    const int tiles = (int)tiles2;
    unsigned short* img = reinterpret_cast<unsigned short*>(ws);
    hipLaunchKernelGGL(gx3b2_weight_kernel, dim3(nk, cdiv(g.Co, 128)), dim3(256), 0, st, a.w, img, g.Co, g.Kp, nk);
    int rc;
    switch (a.pro_mode) {
      case PRO_RELU: rc = map ? launch_x3b_two<PRO_RELU, true>(a, mp, img, tiles, st) : launch_x3b_two<PRO_RELU, false>(a, mp, img, tiles, st); break;
      case PRO_LRELU: rc = map ? launch_x3b_two<PRO_LRELU, true>(a, mp, img, tiles, st) : launch_x3b_two<PRO_LRELU, false>(a, mp, img, tiles, st); break;
      default: rc = map ? launch_x3b_two<PRO_NONE, true>(a, mp, img, tiles, st) : launch_x3b_two<PRO_NONE, false>(a, mp, img, tiles, st);
    }
    if (rc != DIAGAN_OK) return rc;
    return check_launch("conv_gemm_x3b (producer / consumer form)");
  }
  const long fl = ((long)g.Co * g.Kp * 3 + 1) / 2;
  const float* ready = wino_weights_ready(WK_GX3, 0, 1.f, fl);
  const unsigned short* wx = reinterpret_cast<const unsigned short*>(ready);
  if (!ready) {
    const long quads = (long)g.Co * g.Kp / 4;
    long blocks = (quads + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gx3b_weight_kernel, dim3((int)blocks), dim3(256), 0, st, a.w, reinterpret_cast<unsigned short*>(ws), quads);
    wx = reinterpret_cast<const unsigned short*>(ws);
  }
  int rc;
  switch (a.pro_mode) {
    case PRO_RELU: rc = map ? launch_x3b_one<PRO_RELU, true>(a, mp, wx, tiles, st) : launch_x3b_one<PRO_RELU, false>(a, mp, wx, tiles, st); break;
    case PRO_LRELU: rc = map ? launch_x3b_one<PRO_LRELU, true>(a, mp, wx, tiles, st) : launch_x3b_one<PRO_LRELU, false>(a, mp, wx, tiles, st); break;
    default: rc = map ? launch_x3b_one<PRO_NONE, true>(a, mp, wx, tiles, st) : launch_x3b_one<PRO_NONE, false>(a, mp, wx, tiles, st);
  }
  if (rc != DIAGAN_OK) return rc;
  return check_launch("conv_gemm_x3b");
}

}  // namespace diagan
