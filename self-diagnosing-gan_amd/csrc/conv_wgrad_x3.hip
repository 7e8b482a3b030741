// Implicit-GEMM weight gradient on the bf16 matrix pipe with EXACTLY split operands (round 6).
//
// Replaces conv_wgrad_kernel<128,128,PRO_NONE,*> (conv_wgrad.hip: `v_mfma_f32_32x32x2_f32`, 0.64-0.70 of the fp32 MFMA peak, i.e. at
// the pipe's own plateau) for the large weight gradients no Winograd kernel takes -- in StyleGAN2 (reference:
// diagan-pkg/diagan/models/stylegan2.py:553-614 the discriminator's blur + stride-2 convolutions, :224-265 the stride-2 transposed
// form of the modulated convolution, whose weight gradient ops/diffconv.py takes parity class by parity class, and the 1x1 skips):
// 32 of a 245 ms iteration.
//
//   dWp[n][k] = sum_m dY[m][n] * X(m, k),   k = (r, s, c),   X = the gathered input exactly as the forward convolution reads it
//
// Arithmetic (as conv_gemm_x3b.hip): both operands are activations here, so BOTH are split in the loader -- every fp32 value into
// three bf16 pieces that sum to it exactly (wino_weights.h: x3_split) -- and six piece products per fp32 product run as three
// `v_mfma_f32_32x32x16_bf16` per 8 pixels, fp32 accumulation: (a0|a1).(b0|b0) + (a0|a1).(b1|b1) + (a0|a2).(b2|b0), lanes 0-31 | 32-63.
//
// The reduction runs over PIXELS, the slow index of both NHWC operands, while a bf16 MFMA operand wants 8 consecutive reduction
// elements per lane.  The transposition happens in the loader's registers: a thread loads the same 4 channels of 4 consecutive pixels
// (four 16-byte loads, eight lanes = one pixel's 128 bytes), splits them and writes, per channel and piece, 4 pixels x bf16 = 8 bytes
// to an LDS image [piece][128 rows = channels n / columns k][32 pixels], 80-byte rows -- the image conv_gemm_x3b.hip's first form reads,
// so the fragment / MFMA loop is that kernel's: a wave owns 64 x 64 of the 128 x 128 output tile, ten 16-byte fragment reads per 12
// MFMAs, one LDS stage (61 440 bytes), TWO workgroups per CU so that one's split + store phase runs under the other's 48 MFMAs.
// Split-K over pixels, slabs and the deterministic second-stage sum exactly as conv_wgrad.hip (same WgradArgs, same grid).
// Roofline: bf16 MFMA (dense 2.5 PFLOP/s / 6 products = 416.7 TFLOP/s fp32-equivalent).
#include "conv_common.h"
#include "wino_weights.h"
#include <stdlib.h>

namespace diagan {

constexpr int WX_ROW = 40;                         // bf16 per LDS row: 32 pixels + 16 bytes of padding
constexpr int WX_PLANE = 128 * WX_ROW;             // one piece plane of a tile (bf16 elements)
constexpr int WX_LDS_BYTES = 6 * WX_PLANE * 2;     // A (3 planes) + B (3 planes) = 61 440: two workgroups per CU

typedef __bf16 wx_bf16x8 __attribute__((ext_vector_type(8)));

// NOPAD: every gathered coordinate of every pixel is inside the image (no padding, e.g. the 3x3 / stride 2 / pad 0 convolutions behind
// the blur): the loader skips the border tests
// TWO: the opt-in two-piece mode of conv_gemm_x3b.hip (DIAGAN_X3_PIECES=2): third plane, its fragment reads and MFMAs dropped
template <bool NOPAD, bool TWO = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_k = g.Kp >> 7;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int split = logical / a.tiles, tile = logical - split * a.tiles;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int seg = split / a.splits_per_seg, sub = split - seg * a.splits_per_seg;
  const int step0 = seg * a.seg_steps + sub * a.steps_per_split;
  const int step1 = min(step0 + a.steps_per_split, (seg + 1) * a.seg_steps);

  // loader role: pixels 4 pq .. 4 pq + 3 of the K-step, channels / columns 4 lc .. 4 lc + 3 of the tile (both operands)
  const int pq = tid & 7, lc = tid >> 3;
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dy), 0, (int)((unsigned)a.M * g.Co * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Ci * 4u), 0x00020000);
  unsigned aoff = (((unsigned)(step0 * 32 + 4 * pq)) * g.Co + n0 + 4 * lc) * 4u;      // rows past M fall outside num_records
  const unsigned astep = 32u * g.Co * 4u;
  const int arow = g.Co * 4;
  const int kf = k0 + 4 * lc;
  const int tap = kf / g.Ci, kc = kf - tap * g.Ci;
  const int kr = tap / g.S, ks = tap - kr * g.S;
  const bool b_ok = kf < g.K;
  const int dyo = kr * g.dr + g.off, dxo = ks * g.dr + g.off;
  const int cb = g.Ci * 4;
  // pixel (b, oy, ox) of this thread's first row, advanced by 32 pixels per K-step without divisions
  int pb, py, px;
  {
    const int m = step0 * 32 + 4 * pq;
    const unsigned t = fdiv((unsigned)m, a.dWo);
    px = m - (int)t * g.Wo;
    const unsigned b = fdiv(t, a.dHo);
    py = (int)t - (int)b * g.Ho;
    pb = (int)b;
  }
  f32x4 ra[4], rb[4];
  auto load_step = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)      // (the row offset belongs in the vector offset: a scalar offset is outside the range check)
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ysrc, aoff + (unsigned)(j * arow), 0, 0));
    aoff += astep;
    int x = px, y = py, b = pb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = y * g.sy + dyo, ix = x * g.sy + dxo;
      bool ok = b_ok && b < g.B;
      if (!NOPAD) ok = ok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
      const unsigned off = (unsigned)(((b * g.Hi + iy) * g.Wi + ix) * cb + kc * 4) | (ok ? 0u : 0x80000000u);
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
      if (j < 3) {
        ++x;
        const int cx = x >= g.Wo ? 1 : 0;
        x = cx ? 0 : x;
        y += cx;
        const int cy = y >= g.Ho ? 1 : 0;
        y = cy ? 0 : y;
        b += cy;
      }
    }
    {
      int x2 = px + a.adv_x, y2 = py + a.adv_y, b2 = pb + a.adv_b;
      const int cx = x2 >= g.Wo ? 1 : 0;
      x2 -= cx ? g.Wo : 0;
      y2 += cx;
      const int cy = y2 >= g.Ho ? 1 : 0;
      y2 -= cy ? g.Ho : 0;
      b2 += cy;
      px = x2; py = y2; pb = b2;
    }
  };
  const int sto = (4 * lc) * WX_ROW + 4 * pq;       // this thread's slot in a plane (bf16 elements): row 4 lc (+ e), pixels 4 pq ..
  auto store_step = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      u32x2 p0, p1, p2;
      x3_split(f32x4{ra[0][e], ra[1][e], ra[2][e], ra[3][e]}, p0, p1, p2);
      unsigned short* st = lds + sto + e * WX_ROW;
      *reinterpret_cast<u32x2*>(st) = p0;
      *reinterpret_cast<u32x2*>(st + WX_PLANE) = p1;
      if (!TWO) *reinterpret_cast<u32x2*>(st + 2 * WX_PLANE) = p2;
      x3_split(f32x4{rb[0][e], rb[1][e], rb[2][e], rb[3][e]}, p0, p1, p2);
      *reinterpret_cast<u32x2*>(st + 3 * WX_PLANE) = p0;
      *reinterpret_cast<u32x2*>(st + 4 * WX_PLANE) = p1;
      if (!TWO) *reinterpret_cast<u32x2*>(st + 5 * WX_PLANE) = p2;
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
  const int fa = (wm * 64 + fi) * WX_ROW, fb = (wn * 64 + fi) * WX_ROW;
  const int oa01 = (fh ? WX_PLANE : 0) + fa, oa02 = (fh ? 2 * WX_PLANE : 0) + fa;
  const int ob00 = 3 * WX_PLANE + fb, ob11 = 4 * WX_PLANE + fb, ob20 = (fh ? 3 : 5) * WX_PLANE + fb;
  // fragments of one 8-pixel block, double-buffered in registers, in the order of their first use (conv_gemm_x3b.hip): slots
  // 0 a01[0], 1 b00[0], 2 b00[1], 3 a01[1], 4 b11[0], 5 b11[1], 6 a02[0], 7 b20[0], 8 b20[1], 9 a02[1]; block c + 1's ten reads are
  // issued one behind each of block c's first ten MFMAs
  wx_bf16x8 fr[2][10];
  auto read_slot = [&](int buf, int sl, int c) __attribute__((always_inline)) {
    const int t = (sl == 2 || sl == 3 || sl == 5 || sl == 8 || sl == 9) ? 32 * WX_ROW : 0;
    const int base = (sl == 0 || sl == 3) ? oa01 : (sl == 6 || sl == 9) ? oa02 : (sl == 1 || sl == 2) ? ob00 : (sl == 4 || sl == 5) ? ob11 : ob20;
    fr[buf][sl] = *reinterpret_cast<const wx_bf16x8*>(lds + base + t + c * 8);
  };
  auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < (TWO ? 6 : 10); ++sl) read_slot(0, sl, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cur = c & 1, nxt = cur ^ 1;
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int p = q >> 2, i = (q >> 1) & 1, j = q & 1;
        const int sa = p == 2 ? (i ? 9 : 6) : (i ? 3 : 0);
        const int sb = p == 0 ? 1 + j : (p == 1 ? 4 + j : 7 + j);
        if (TWO && p == 2) continue;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[cur][sa], fr[cur][sb], acc[i][j], 0, 0, 0);
        if (c < 3 && q < (TWO ? 6 : 10)) read_slot(nxt, q, c + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  if (step0 < step1) {
    load_step();
    store_step();
  }
  __syncthreads();
  for (int step = step0; step < step1; ++step) {
    const bool more = step + 1 < step1;
    if (more) load_step();
    mfmas();
    __syncthreads();
    if (more) {
      store_step();
      __syncthreads();
    }
  }

  // C/D map of the 32x32 MFMA: col = lane & 31 (k), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (n); raw buffer stores as conv_wgrad.hip
  float* out = a.slab + (long)split * a.slab_stride;
  const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((unsigned)g.Co * g.Kp * 4u), 0x00020000);
  const unsigned rowbytes = (unsigned)g.Kp * 4u;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wn * 64 + j * 32 + fi;
      const int nrow = n0 + wm * 64 + i * 32 + 4 * fh;
      const unsigned vbase = ((unsigned)nrow * g.Kp + k) * 4u;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][j][e]), osrc, vbase, (int)(((e & 3) + 8 * (e >> 2)) * rowbytes), 0);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static int wgrad_x3_switch = -1;     // process-level diagnostic switch (-1: the environment's DIAGAN_WGRAD_X3, default on)
void wgrad_x3_set(int on) { wgrad_x3_switch = on; }
static bool wgrad_x3_enabled() {
  static const int env = getenv("DIAGAN_WGRAD_X3") ? atoi(getenv("DIAGAN_WGRAD_X3")) : 1;
  return (wgrad_x3_switch >= 0 ? wgrad_x3_switch : env) != 0;
}

// the launches this kernel takes from conv_wgrad_kernel<128,128,PRO_NONE,*>: whole 128 x 128 tiles, whole 32-channel blocks of a tap per
// eight lanes, no prologue, no bias column, and enough work that the matrix pipe -- not the launch -- is what the time goes to
bool wgrad_x3_takes(const WgradArgs& a, bool x3_on) {
  const ConvGeom& g = a.g;
  if (!x3_on || !wgrad_x3_enabled()) return false;
  if (a.pro_mode != PRO_NONE || a.bias_off >= 0 || g.up != 1) return false;
  if ((g.Co & 127) || (g.Kp & 127) || g.K != g.Kp || (g.Ci & 31)) return false;
  if (wgrad_x3_switch == 2) return true;             // (tests: every geometry the kernel can run)
  static const double floor_mac = getenv("DIAGAN_WGRAD_X3_MIN_MAC") ? atof(getenv("DIAGAN_WGRAD_X3_MIN_MAC")) : 4e9;
  return (double)a.M * g.Co * g.K >= floor_mac;
}

static bool wgrad_x3_nopad(const ConvGeom& g) {
  const int lo = g.dr > 0 ? g.off : g.off - (g.R - 1), hi_y = (g.Ho - 1) * g.sy + (g.dr > 0 ? g.off + g.R - 1 : g.off);
  const int lo_x = g.dr > 0 ? g.off : g.off - (g.S - 1), hi_x = (g.Wo - 1) * g.sy + (g.dr > 0 ? g.off + g.S - 1 : g.off);
  return lo >= 0 && lo_x >= 0 && hi_y < g.Hi && hi_x < g.Wi;
}

int x3_pieces();                                     // conv_gemm_x3b.hip

int launch_wgrad_x3(const WgradArgs& a, int splits, hipStream_t st) {
  static FuncAttrLatch l0, l1, m0, m1;
  const dim3 grid(a.tiles * splits);
  if (x3_pieces() == 2) {
    if (wgrad_x3_nopad(a.g)) {
      DG_LDS(m1, (conv_wgrad_x3_kernel<true, true>), WX_LDS_BYTES);
      hipLaunchKernelGGL((conv_wgrad_x3_kernel<true, true>), grid, dim3(256), WX_LDS_BYTES, st, a);
    } else {
      DG_LDS(m0, (conv_wgrad_x3_kernel<false, true>), WX_LDS_BYTES);
      hipLaunchKernelGGL((conv_wgrad_x3_kernel<false, true>), grid, dim3(256), WX_LDS_BYTES, st, a);
    }
    return check_launch("conv_wgrad_x3 (two pieces)");
  }
  if (wgrad_x3_nopad(a.g)) {
    DG_LDS(l1, conv_wgrad_x3_kernel<true>, WX_LDS_BYTES);
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<true>), grid, dim3(256), WX_LDS_BYTES, st, a);
  } else {
    DG_LDS(l0, conv_wgrad_x3_kernel<false>, WX_LDS_BYTES);
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<false>), grid, dim3(256), WX_LDS_BYTES, st, a);
  }
  return check_launch("conv_wgrad_x3");
}

}  // namespace diagan
