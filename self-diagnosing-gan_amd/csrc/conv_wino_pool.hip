// Winograd F(2x2, 3x3) convolution FOLLOWED BY a 2x2 average pool, in 9 instead of 16 multiplies per tile.
//
// Replaces F.avg_pool2d(c2(relu(h)), 2) at the end of mimicry's DBlock / DBlockOptimized (downsample=True; selected at
// diagan-pkg/diagan/models/predefined_models.py:38-40,76-78).  A Winograd tile IS one pooling window: with
// Y = A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) the window's sum is  sum_ab Y[a][b] = c^T M c  with c = 1^T A^T = (1, 2, 0, -1):
// frequency row / column 2 never contributes, so only the 9 products M[i][j], i, j in {0, 1, 3}, are computed --
// 9/16 of the matrix work of conv_wino.hip (1/4 of the direct convolution's) and the full-resolution activation is
// neither written nor read back.
//
// One workgroup = 512 threads = 8 waves = 64 tiles (= 64 pooled pixels) x 128 output channels (128 tiles x 64 channels for
// layers with 64 output channels; the planes below are for the first shape); wave (th, cq) owns ALL nine
// frequencies of its 32 tiles x 32 channels (144 accumulator registers), so the pooled value is finished in registers:
// no exchange through LDS in the epilogue.  K loop, input loader / transform and the transformed weights are those of
// conv_wino.hip (same `ug` image from wino_weight_kernel; only the live planes are fetched):
//   V planes [(ri, ji, q)] = 18 KB, U planes [(column block, ri)][8 slots (j, kq), the j = 2 slots unused] = 48 KB per stage.
//
// UNPOOL = true is the adjoint situation, the data-gradient of such a layer: dx = conv^T(avg_pool2d_backward(g)).  The
// up-sampled gradient is constant over every pooling window, the 4x4 input patch of a tile reads (a, b, b, c) along each
// axis, and B^T (a, b, b, c) = (a - b, 2b, 0, b - c): again only the nine frequencies i, j in {0, 1, 3} are non-zero.  The
// loader reads the 3x3 HALF-resolution neighbourhood of the tile (9 instead of 16 pixels), the K loop is the same, and
// the epilogue applies A^T . A to the nine products in registers, turns each wave's 32 tiles x 4 pixels x 32 channels
// through 16 KB of LDS and writes full-resolution pixels (mask, residual, 1/sigma applied) in 16-byte pieces.
//
// The weight gradient of such a layer needs no kernel of its own: against the pooled gradient it is the weight gradient of
// a 3x3 / stride-2 / pad-0 convolution over the (H+1) x (W+1) image of 2x2 box sums of relu(h) (diagan_boxsum2 +
// diagan_conv_wgrad; ConvLayer.wgrad_pooled) -- also a quarter of the products.
// Roofline: MFMA fp32; executes 9/36 of the direct convolution's multiply-accumulates.  Measured: profiles/r02_winograd.md.
#include "conv_common.h"
#include <type_traits>

namespace diagan {

constexpr int PK = 8;                   // input channels per K-step
constexpr int P_UPLANE = 64 * 4;        // floats of one U plane: 64 columns x 4 channels
constexpr int P_VPL = 18;               // V planes per stage: (ri, ji, q)

// NB = 64-column blocks per workgroup: 2 -> 64 tiles x 128 columns (waves 2 x 4), 1 -> 128 tiles x 64 columns (waves 4 x 2,
// the layers with 64 output channels); PT tiles, a V plane holds PT rows x 4 channels.
template <int NB> struct PoolShape {
  static constexpr int PT = NB == 2 ? 64 : 128;
  static constexpr int PN = 64 * NB;
  static constexpr int VPLANE = PT * 4;
  static constexpr int STAGE = P_VPL * VPLANE + NB * 3 * 8 * P_UPLANE;   // floats: 66 KB (NB = 2), 60 KB (NB = 1)
  static constexpr int SUB = PT / 64;                                      // loader roles (tiles) per thread
};

template <int PRO, bool UNPOOL, int NB>
__global__ __launch_bounds__(512, 2) void conv_wino_pool_kernel(const ConvGemmArgs a, const float* __restrict__ ug) {
  using SH = PoolShape<NB>;
  constexpr int PT = SH::PT, PN = SH::PN, VPLANE = SH::VPLANE, P_STAGE = SH::STAGE, SUB = SH::SUB;
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 stages][V 18 planes | U NB x 24 planes]
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = g.Co / PN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int t0 = (tile / tiles_n) * PT, nb = tile % tiles_n, n0 = nb * PN;
  const int TW = g.Wo >> 1, TH = g.Ho >> 1;
  const int MT = g.B * TH * TW;                         // 2x2 output tiles = pooled pixels
  const int nk = g.Ci / PK;
  const int k_per = (nk + a.ksplit - 1) / a.ksplit;
  const int k_begin = blockIdx.y * k_per, k_end = min(k_begin + k_per, nk);

  // ---- loader role (as in conv_wino.hip): (tile lt [+ 64], channel quad q, patch row r) ----
  const int lr = tid & 3, lq = (tid >> 2) & 1, lt = tid >> 3;
  constexpr int NLD = UNPOOL ? 3 : 4;                    // pixels per loader row
  unsigned off[SUB][NLD], inv[SUB][NLD];
#pragma unroll
  for (int sb = 0; sb < SUB; ++sb) {
    const int gt = t0 + lt + sb * 64;
    const bool tv = gt < MT;
    const unsigned q1 = fdiv((unsigned)(tv ? gt : 0), a.dWo);          // dWo: divisor TW
    const int tx = (tv ? gt : 0) - (int)q1 * TW;
    const unsigned b = fdiv(q1, a.dHo);                                // dHo: divisor TH
    const int ty = (int)q1 - (int)b * TH;
    if (!UNPOOL) {
      const int iy = 2 * ty - 1 + lr, ix0 = 2 * tx - 1;
      const bool rv = tv && iy >= 0 && iy < g.Hi;
      const int rowbase = (((int)b * g.Hi + iy) * g.Wi + ix0) * g.Ci * 4 + lq * 16;
#pragma unroll
      for (int c = 0; c < NLD; ++c) {
        const bool ok = rv && ix0 + c >= 0 && ix0 + c < g.Wi;
        off[sb][c] = ok ? (unsigned)(rowbase + c * g.Ci * 4) : 0u;
        inv[sb][c] = ok ? 0u : 0x80000000u;             // beyond num_records: the hardware returns zeros (relu(0) = 0)
      }
    } else {
      // half-resolution gradient [B][TH][TW][Ci]: lanes r = 0, 1, 3 of the quad hold rows ty - 1, ty, ty + 1 (r = 2: nothing);
      // outside the image the up-sampled gradient is the convolution's zero padding
      const int u = lr == 3 ? 2 : lr;
      const int iy = ty - 1 + u, ix0 = tx - 1;
      const bool rv = tv && lr != 2 && iy >= 0 && iy < TH;
      const int rowbase = (((int)b * TH + iy) * TW + ix0) * g.Ci * 4 + lq * 16;
#pragma unroll
      for (int c = 0; c < NLD; ++c) {
        const bool ok = rv && ix0 + c >= 0 && ix0 + c < TW;
        off[sb][c] = ok ? (unsigned)(rowbase + c * g.Ci * 4) : 0u;
        inv[sb][c] = ok ? 0u : 0x80000000u;
      }
    }
  }
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((unsigned)g.B * (UNPOOL ? TH * TW : g.Hi * g.Wi) * g.Ci * 4u), 0x00020000);
  // column transform as ONE fmac through DPP: V[r] = t[r] + sc * t[partner].  pooled forward: partner by quad_perm
  // [2,2,1,1] (rows 0: t0 - t2, 1: t1 + t2, 3: t3 - t1 = -(B^T row 3), compensated in U); UNPOOL: every lane's partner is
  // lane 1 (rows 0: u0 - u1, 1: u1 + u1, 3: u2 - u1 = -(b - c))
  const float sc = lr == 1 ? 1.f : -1.f;
  const int lri = lr == 3 ? 2 : lr;                      // live row index (row 2 is a partner only: it stores nothing)
  const int vslot = (lt ^ (lq | (lr << 1))) * 4;         // (the second tile, lt + 64, sits 64 rows further: same swizzle)
  float* const vst0 = smem + ((lri * 3) * 2 + lq) * VPLANE + vslot;
  const bool vlive = lr != 2;

  // weight DMA role: waves 0 .. 3 NB - 1 fetch the six live planes of one (column block, frequency row)
  const int dcb = wave / 3, dri = wave - dcb * 3, di = dri == 2 ? 3 : dri;
  const float* const ublock = ug + (long)(nb * NB + dcb) * nk * (32 * P_UPLANE) + di * 8 * P_UPLANE;

  f32x4 ra[SUB][NLD];
  auto issue_loads = [&](int kk, int stage) {
    if (wave < 3 * NB) {
      float* ul = smem + stage * P_STAGE + P_VPL * VPLANE + (dcb * 3 + dri) * 8 * P_UPLANE;
      const unsigned long long ub = (unsigned long long)(ublock + (long)kk * (32 * P_UPLANE));
      const unsigned long long us64 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ub) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ub >> 32)) << 32;
      const float* up = reinterpret_cast<const float*>(us64) + (unsigned)(lane * 4);
      const float* up6 = up + 6 * P_UPLANE;              // planes (j = 3, kq): beyond the 4 KB immediate range
      float* ul6 = ul + 6 * P_UPLANE;
#define POOL_DMA(G, L, I)                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(G),               \
                                   (__attribute__((address_space(3))) void*)(L), 16, (I) * P_UPLANE * 4, 0)
      POOL_DMA(up, ul, 0);
      POOL_DMA(up, ul, 1);
      POOL_DMA(up, ul, 2);
      POOL_DMA(up, ul, 3);
      POOL_DMA(up6, ul6, 0);
      POOL_DMA(up6, ul6, 1);
#undef POOL_DMA
    }
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb)
#pragma unroll
      for (int c = 0; c < NLD; ++c)
        ra[sb][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off[sb][c] | inv[sb][c], kk * (PK * 4), 0));
  };
  float t[4][4];                                        // row-transformed patch row: [column j][channel] (j = 2 unused)
  auto transform_rows = [&](int sb) {
    f32x4 d[NLD];
#pragma unroll
    for (int c = 0; c < NLD; ++c) {
      f32x4 v = ra[sb][c];
      if (PRO == PRO_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __int_as_float(max(__float_as_int(v[e]), 0));   // one v_max_i32 on the bits
      }
      d[c] = v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!UNPOOL) {
        t[0][e] = d[0][e] - d[2][e];
        t[1][e] = d[1][e] + d[2][e];
        t[3][e] = d[1][e] - d[NLD - 1][e];
      } else {                                          // B^T (a, b, b, c) = (a - b, 2 b, 0, b - c)
        t[0][e] = d[0][e] - d[1][e];
        t[1][e] = d[1][e] + d[1][e];
        t[3][e] = d[1][e] - d[2][e];
      }
    }
  };
  auto transform_store = [&](int stage, int sb, int j, int ji) {
    float o0 = t[j][0], o1 = t[j][1], o2 = t[j][2], o3 = t[j][3];
    if (!UNPOOL)
      asm volatile(
          "s_nop 1\n\t"
          "v_fmac_f32_dpp %0, %0, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %1, %1, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %2, %2, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %3, %3, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
          : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3)
          : "v"(sc));
    else
      asm volatile(
          "s_nop 1\n\t"
          "v_fmac_f32_dpp %0, %0, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %1, %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %2, %2, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_fmac_f32_dpp %3, %3, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
          : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3)
          : "v"(sc));
    const f32x4 o = {o0, o1, o2, o3};
    if (vlive) *reinterpret_cast<f32x4*>(vst0 + stage * P_STAGE + ji * 2 * VPLANE + sb * 256) = o;
  };
  // the input transform of one loader tile in four pieces: rows, then the three live columns
  auto transform_piece = [&](int stage, int piece) {
    const int sb = piece >> 2, w = piece & 3;
    if (w == 0) transform_rows(sb);
    else transform_store(stage, sb, w == 3 ? 3 : w - 1, w - 1);
  };

  // wave (th, cq): tiles th * 32 .. + 31, channels cq * 32 .. + 31, all nine frequencies f = ri * 3 + ji
  const int th = NB == 2 ? wave >> 2 : wave >> 1, cq = NB == 2 ? wave & 3 : wave & 1;
  f32x16 acc[9];
#pragma unroll
  for (int f = 0; f < 9; ++f)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
  const int fi = lane & 31, fh = lane >> 5;
  const float* fa_base[3];
  const float* fb_base[3];
#pragma unroll
  for (int ri = 0; ri < 3; ++ri) {
    const int r = ri == 2 ? 3 : ri;
    fa_base[ri] = smem + ((ri * 3) * 2 + fh) * VPLANE + (((th * 32 + fi) ^ (fh | (r << 1))) << 2);
    fb_base[ri] = smem + P_VPL * VPLANE + (((cq >> 1) * 3 + ri) * 8 + fh) * P_UPLANE + (((cq & 1) * 32 + fi) << 2);
  }

  if (k_begin < k_end) {
    issue_loads(k_begin, 0);
#pragma unroll
    for (int piece = 0; piece < 4 * SUB; ++piece) transform_piece(0, piece);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto kstep = [&](int kk, auto has_next) {
    const int cur = (kk - k_begin) & 1;
    if (decltype(has_next)::value) issue_loads(kk + 1, cur ^ 1);
    f32x4 fa[2][3], fb[2][3];
    auto read_row = [&](int ri) {
#pragma unroll
      for (int ji = 0; ji < 3; ++ji) {
        const int j = ji == 2 ? 3 : ji;
        fa[ri & 1][ji] = *reinterpret_cast<const f32x4*>(fa_base[ri] + cur * P_STAGE + ji * 2 * VPLANE);
        fb[ri & 1][ji] = *reinterpret_cast<const f32x4*>(fb_base[ri] + cur * P_STAGE + j * 2 * P_UPLANE);
      }
    };
    read_row(0);
#pragma unroll
    for (int ri = 0; ri < 3; ++ri) {
      if (ri < 2) read_row(ri + 1);
#pragma unroll
      for (int ji = 0; ji < 3; ++ji)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int grp = (ri * 3 + ji) * 4 + e;
          // the next stage's input transform, in 4 (8) pieces from MFMA 18 (12) on, one every third (second) MFMA
          constexpr int G0 = SUB == 1 ? 18 : 12, GS = SUB == 1 ? 3 : 2;
          if (decltype(has_next)::value && grp >= G0 && (grp - G0) % GS == 0 && (grp - G0) / GS < 4 * SUB) {
            __builtin_amdgcn_sched_barrier(0);
            transform_piece(cur ^ 1, (grp - G0) / GS);
          }
          acc[ri * 3 + ji] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ri & 1][ji][e], fb[ri & 1][ji][e], acc[ri * 3 + ji], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA pieces of the next stage have landed
    __syncthreads();
  };
  for (int kk = k_begin; kk + 1 < k_end; ++kk) kstep(kk, std::true_type{});
  if (k_begin < k_end) kstep(k_end - 1, std::false_type{});

  // ---- epilogue, all in this wave's registers ----
  const bool raw = a.ksplit > 1;
  const float sc0 = a.scale0 ? a.scale0[0] : a.out_scale, sc1 = a.scale1 ? a.scale1[0] : a.out_scale;
  const int n = n0 + cq * 32 + fi;
  const float bv = (!raw && a.bias) ? a.bias[n] : 0.f;
  const float rfloor = a.res_relu ? 0.f : -__builtin_huge_valf();
  if (!UNPOOL) {
    // pooled = 1/4 c^T M c, c = (1, 2, -1) over the live rows / columns
    const int split = a.scale0 ? (a.scale_split >> 2) : 0x7fffffff;   // tile index where the second sigma starts
    float* ydst = raw ? a.slab + (long)blockIdx.y * MT * g.Co : a.y;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int gt = t0 + th * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      float s = (acc[0][e] + 2.f * acc[1][e] - acc[2][e]) + 2.f * (acc[3][e] + 2.f * acc[4][e] - acc[5][e]) -
                (acc[6][e] + 2.f * acc[7][e] - acc[8][e]);
      s *= 0.25f;
      if (gt < MT) {
        const long o = (long)gt * g.Co + n;
        if (!raw) {
          s = s * (gt < split ? sc0 : sc1) + bv;
          if (a.residual) s += fmaxf(a.residual[o], rfloor);
        }
        ydst[o] = s;
      }
    }
  } else {
    // Y = A^T M A over the live frequencies: rows / columns weigh (1, 1, 0) for pixel 0 and (0, 1, -1) for pixel 1; the 1/4
    // of the pooling's backward is applied here.  The wave's 32 tiles x 4 pixels x 32 channels are turned through its own
    // 16 KB of LDS (the stages are dead: the K loop ended on a barrier) so that mask, residual and output move as 16-byte
    // pieces -- a lane holds ONE channel of 16 tiles, which as 4-byte accesses was the slowest part of the launch.
    const int split = a.scale0 ? a.scale_split : 0x7fffffff;          // pixel-row index where the second sigma starts
    float* ydst = raw ? a.slab + (long)blockIdx.y * a.M * g.Co : a.y;
    float* wl = smem + wave * 4096;                                    // [32 tiles][4 pixels][32 channels]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int trow = (e & 3) + 8 * (e >> 2) + 4 * fh;
      const float y4[4] = {(acc[0][e] + acc[1][e]) + (acc[3][e] + acc[4][e]), (acc[1][e] - acc[2][e]) + (acc[4][e] - acc[5][e]),
                           (acc[3][e] + acc[4][e]) - (acc[6][e] + acc[7][e]), (acc[4][e] - acc[5][e]) - (acc[7][e] - acc[8][e])};
#pragma unroll
      for (int p = 0; p < 4; ++p) wl[(trow * 4 + p) * 32 + fi] = 0.25f * y4[p];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the wave's own writes (in order, one wave: no barrier)
    const int c4 = lane & 7, n4 = n0 + cq * 32 + c4 * 4;
    f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
    if (!raw && a.bias) bv4 = *reinterpret_cast<const f32x4*>(a.bias + n4);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int item = lane + 64 * k, trow = item >> 5, p = (item >> 3) & 3;
      const int gt = t0 + th * 32 + trow;
      if (gt >= MT) continue;
      f32x4 y = *reinterpret_cast<const f32x4*>(wl + item * 4);
      const unsigned q1 = fdiv((unsigned)gt, a.dWo);
      const int tx = gt - (int)q1 * TW;
      const unsigned b = fdiv(q1, a.dHo);
      const int ty = (int)q1 - (int)b * TH;
      const int prow = ((int)b * g.Ho + 2 * ty + (p >> 1)) * g.Wo + 2 * tx + (p & 1);
      const long o = (long)prow * g.Co + n4;
      if (!raw) {
        y = y * (prow < split ? sc0 : sc1) + bv4;
        if (a.residual) {
          f32x4 r = *reinterpret_cast<const f32x4*>(a.residual + o);
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = fmaxf(r[q], rfloor);
          y += r;
        }
        if (a.mask_src) {
          const f32x4 ms = *reinterpret_cast<const f32x4*>(a.mask_src + o);
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = ms[q] > 0.f ? y[q] : y[q] * a.mask_slope;
        }
      }
      *reinterpret_cast<f32x4*>(ydst + o) = y;
    }
  }
}

template <int PRO, bool UNPOOL, int NB>
static int launch_wino_pool_pro(const ConvGemmArgs& a, const float* ug, hipStream_t st) {
  using SH = PoolShape<NB>;
  const int MT = a.g.B * (a.g.Ho >> 1) * (a.g.Wo >> 1);
  const int wgs = cdiv(MT, SH::PT) * (a.g.Co / SH::PN);
  // (the data-gradient's epilogue turns every wave's 16 KB of outputs through LDS: 128 KB)
  const size_t lds = UNPOOL ? (size_t)(128 << 10) + (NB == 2 ? (4 << 10) : 0) : (size_t)2 * SH::STAGE * sizeof(float);
  auto kern = conv_wino_pool_kernel<PRO, UNPOOL, NB>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, lds);
  hipLaunchKernelGGL(kern, dim3(wgs, a.ksplit), dim3(512), lds, st, a, ug);
  return check_launch("conv_wino_pool");
}

const float* launch_wino_weights(const float* w, float* ug, int Co, int Ci, int Kp, int flip, hipStream_t st);   // conv_wino.hip

// Split-K factor of the pooled kernels: 1 when the launch has >= min_wgs workgroups, else the smallest split that reaches
// 256 workgroups with at least 8 K-steps each; 0 = neither (the caller keeps the separate convolution + pooling).
// Workgroups: 64 tiles x 128 columns (Co % 128 == 0) or 128 tiles x 64 columns.
int wino_pool_ksplit(int B, int Ho, int Wo, int Ci, int Co, long slab_tiles, int min_wgs) {
  const long MT = (long)B * (Ho >> 1) * (Wo >> 1);
  const long wgs = Co % 128 == 0 ? cdiv(MT, 64) * (Co / 128) : cdiv(MT, 128) * (Co / 64);
  if (wgs >= min_wgs) return 1;
  const int nk = Ci / PK;
  for (int ks = 2; ks <= 8; ++ks)
    if (wgs * ks >= 256 && nk / ks >= 8 && (long)ks * MT * Co <= slab_tiles) return ks;
  return 0;
}

int launch_wino_pool(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 1));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 1));
  const float* ug = launch_wino_weights(a.w, ws, g.Co, g.Ci, g.Kp, g.dr < 0 ? 1 : 0, st);
  if (g.Co % 128 == 0)
    return a.pro_mode == PRO_RELU ? launch_wino_pool_pro<PRO_RELU, false, 2>(a, ug, st) : launch_wino_pool_pro<PRO_NONE, false, 2>(a, ug, st);
  return a.pro_mode == PRO_RELU ? launch_wino_pool_pro<PRO_RELU, false, 1>(a, ug, st) : launch_wino_pool_pro<PRO_NONE, false, 1>(a, ug, st);
}

// tile_cfg 12: a.x is the HALF-resolution gradient [B][Ho/2][Wo/2][Ci]; y / mask_src / residual are full resolution
int launch_wino_unpool(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 1));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 1));
  const float* ug = launch_wino_weights(a.w, ws, g.Co, g.Ci, g.Kp, g.dr < 0 ? 1 : 0, st);
  return g.Co % 128 == 0 ? launch_wino_pool_pro<PRO_NONE, true, 2>(a, ug, st) : launch_wino_pool_pro<PRO_NONE, true, 1>(a, ug, st);
}

}  // namespace diagan
