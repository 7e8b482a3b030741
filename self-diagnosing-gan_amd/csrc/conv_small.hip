// 3x3 convolution to FOUR output channels (RGB + pad), stride 1, pad 1 -- the generator's last conv
// and the data-gradient of the discriminator's first conv (SURVEY §8 a2/a4 "c5/c6", a3/a5 block1) --
// and the weight gradient of the same layer (conv3x3_co4_wgrad, below).
//
// On the 32x32 matrix tiles these cost a full 32-wide (64 with the tile) N dimension for 4 useful columns
// (measured 5.5 TFLOP/s, 218 us for M=65536, K=2304).  gfx950 has a matrix instruction whose N is exactly 4:
// v_mfma_f32_4x4x1_16B_f32 = 16 independent 4x4 outer products per wave, D_b[i][j] += A_b[i] * B_b[j],
// lane 4b+i supplies A_b[i], lane 4b+j supplies B_b[j] and receives column j of D_b (layout probed on the
// hardware: tools/probe/mfma4x4.hip).  It runs at the same 256 FLOP/cycle/CU as the big fp32 tiles.
//
//   forward / dgrad:  i = one of 4 pixels of block b (lane = pixel), j = output channel, one k per instruction:
//                     a wave produces 64 pixels x 4 channels; A comes from an LDS tile of the (prologue-
//                     transformed) input with halo, B from the LDS copy of the packed weights.
//   weight gradient:  i = input channel 4b+i of a 64-channel chunk (lane = channel: coalesced global reads, no
//                     LDS), j = output channel, one PIXEL per instruction; the whole 4 x 9 x Ci gradient lives
//                     in the wave's accumulators (4 VGPRs per tap per chunk).
//
// Same gather formula / prologue / epilogue semantics as conv_gemm_kernel.  Roofline: LDS bandwidth for the
// forward (one 16-byte A read per lane per 4 instructions), HBM / L2 for the weight gradient.
#include "conv_common.h"
#include <stdlib.h>

namespace diagan {

struct SmallCoArgs {
  const float* x;         // gathered tensor NHWC [B,H,W,Ci]
  const float* w;         // packed [4][Kp]
  float* y;               // [B,H,W,4]
  const float* bias;      // [4] or null
  const float* residual;  // [B,H,W,4] or null
  const float* pro_scale;
  const float* pro_shift;
  int pro_mode;
  int B, H, W, Ci, Kp;
  int dr, off;            // tap r reads row oy + r*dr + off (conv pad 1: +1,-1; its data-gradient: -1,+1)
  int tiles_x, tiles_y;
  int group_imgs;         // > 0: image b reads its affine prologue from pro_scale / pro_shift + (b / group_imgs)*Ci
};

constexpr int SC_TH = 4, SC_TW = 32;        // output tile of a workgroup: 4 rows x 32 columns (2 waves of 64 pixels)
constexpr int SC_CC = 16;                   // channels per LDS stage
constexpr int SC_PS = SC_CC + 4;            // padded pixel stride in floats: conflict-free ds_read_b128 across pixels
constexpr int SC_TILE = (SC_TH + 2) * (SC_TW + 2) * SC_PS;
constexpr int SC_WT = 9 * 4 * SC_CC;
constexpr int SC_NLOAD = (SC_TH + 2) * (SC_TW + 2) * (SC_CC / 4);   // float4 loads per stage (816)
constexpr int SC_LPT = (SC_NLOAD + 255) / 256;

// 256 threads = 4 waves: wave = (pixel group pg = rows 2pg, 2pg+1 of the tile) x (channel half kh of each stage);
// the two channel halves are summed through LDS at the end.
__global__ __launch_bounds__(256) void conv3x3_co4_kernel(const SmallCoArgs a) {
  __shared__ __attribute__((aligned(16))) float xt[2][SC_TILE];
  __shared__ __attribute__((aligned(16))) float wt[2][SC_WT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pg = wave & 1, kh = wave >> 1;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int oy0 = ty * SC_TH, ox0 = tx * SC_TW;
  const bool affine = a.pro_mode == PRO_AFFINE_RELU || a.pro_mode == PRO_AFFINE;
  const int nchunk = a.Ci / SC_CC;
  const int pgo = a.group_imgs > 0 ? (b / a.group_imgs) * a.Ci : 0;

  // ---- stage loader: thread-fixed (tile pixel, channel quad) slots ----------------------------------
  int goff[SC_LPT], loff[SC_LPT];
  bool gok[SC_LPT], lok[SC_LPT];
#pragma unroll
  for (int i = 0; i < SC_LPT; ++i) {
    const int s = tid + 256 * i;
    const int q = s & 3, pix = s >> 2;
    const int px = pix % (SC_TW + 2), py = pix / (SC_TW + 2);
    const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
    lok[i] = s < SC_NLOAD;
    gok[i] = lok[i] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    goff[i] = gok[i] ? ((b * a.H + iy) * a.W + ix) * a.Ci + q * 4 : 0;
    loff[i] = pix * SC_PS + q * 4;
  }
  const int q_own = tid & 3;   // 256 % 4 == 0: every slot of a thread has the same channel quad
  // weights of a stage: wt[tap][j][16] <- w[j][tap*Ci + chunk*16 ...]: 144 float4
  const bool w_ok = tid < 9 * 4 * 4;
  const int w_q = tid & 3, w_j = (tid >> 2) & 3, w_tap = tid >> 4;
  const int w_goff = w_j * a.Kp + w_tap * a.Ci + w_q * 4;
  const int w_loff = (w_tap * 4 + w_j) * SC_CC + w_q * 4;

  f32x4 rx[SC_LPT], rw, psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  auto load_stage = [&](int chunk) {
    const int c0 = chunk * SC_CC;
#pragma unroll
    for (int i = 0; i < SC_LPT; ++i)
      rx[i] = gok[i] ? *reinterpret_cast<const f32x4*>(a.x + goff[i] + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
    rw = w_ok ? *reinterpret_cast<const f32x4*>(a.w + w_goff + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (affine) {
      psc = *reinterpret_cast<const f32x4*>(a.pro_scale + pgo + c0 + q_own * 4);
      psh = *reinterpret_cast<const f32x4*>(a.pro_shift + pgo + c0 + q_own * 4);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < SC_LPT; ++i) {
      f32x4 v = rx[i];
      if (affine) v = v * psc + psh;
      if (a.pro_mode == PRO_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
      } else if (a.pro_mode == PRO_RELU || a.pro_mode == PRO_AFFINE_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (!gok[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};     // padding is zero AFTER the transform
      if (lok[i]) *reinterpret_cast<f32x4*>(&xt[buf][loff[i]]) = v;
    }
    if (w_ok) *reinterpret_cast<f32x4*>(&wt[buf][w_loff]) = rw;
  };

  // ---- MFMA side --------------------------------------------------------------------------------------
  const int pr = 2 * pg + (lane >> 5), pc = lane & 31, j = lane & 3;
  // window offset of tap (r, s) inside the haloed tile: row pr + 1 + (r*dr + off)
  int aoff[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, s = tap % 3;
    aoff[tap] = ((pr + 1 + r * a.dr + a.off) * (SC_TW + 2) + (pc + 1 + s * a.dr + a.off)) * SC_PS;
  }
  f32x4 acc[4];   // one accumulator per k of a 16-byte operand: consecutive instructions never depend on each other
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int cur = chunk & 1;
    if (chunk + 1 < nchunk) load_stage(chunk + 1);
    const float* xc = xt[cur];
    const float* wc = wt[cur];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c4 = 2 * kh + cc;
        const f32x4 av = *reinterpret_cast<const f32x4*>(xc + aoff[tap] + c4 * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(wc + (tap * 4 + j) * SC_CC + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], bv[e], acc[e], 0, 0, 0);
      }
    }
    if (chunk + 1 < nchunk) store_stage(cur ^ 1);
    __syncthreads();
  }
  f32x4 o = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  // sum the two channel halves: kh == 1 waves park their accumulators in LDS (the tiles are done with)
  float* red = xt[0];
  if (kh == 1) *reinterpret_cast<f32x4*>(red + (pg * 64 + lane) * 4) = o;
  __syncthreads();
  if (kh == 0) {
    o += *reinterpret_cast<const f32x4*>(red + (pg * 64 + lane) * 4);
    const float bj = a.bias ? a.bias[j] : 0.f;
    // lane 4g+j holds out[pixel 4g+e][j], e = 0..3; pixels 4g..4g+3 are consecutive columns of one row
    const int p0 = lane & ~3;
    const int oy = oy0 + 2 * pg + (p0 >> 5);
    if (oy < a.H) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ox = ox0 + (p0 & 31) + e;
        if (ox >= a.W) continue;
        const long m = ((long)b * a.H + oy) * a.W + ox;
        float v = o[e] + bj;
        if (a.residual) v += a.residual[m * 4 + j];
        a.y[m * 4 + j] = v;
      }
    }
  }
}

// ---- weight gradient of the same layer --------------------------------------------------------------
struct SmallCoWgradArgs {
  const float* dy;        // [B,H,W,4]
  const float* x;         // [B,H,W,Ci]
  float* slab;            // [gridDim.x][slab_stride]: weight partials laid out as the packed weight, [4][Kp]
  const float* pro_scale;
  const float* pro_shift;
  int pro_mode;
  int B, H, W, Ci, Kp;
  long slab_stride;
  long bias_off;          // >= 0: column sums of dy go to slab[blk][bias_off + j]
  int rows_per_block;     // image rows (of B*H) per workgroup
};

// Wave w of a workgroup owns the 64-channel chunk (w % NCH) -- lane = channel, 9 taps x 4 accumulator VGPRs -- and
// the image rows (w / NCH) mod RG of the workgroup's row range (RG = 4 / NCH).  Work item = 8 consecutive
// pixels of one row: its 3 x 10 input values per lane and 8 dy values are loaded as ONE burst of branch-free
// clamped loads into the register buffer that is not being consumed (two buffers, explicitly unrolled by two, so
// the burst of item q+1 is in flight under the 72 matrix instructions of item q); padding and the prologue
// are applied when a buffer is consumed.  Per pixel: B = dy[p][lane & 3], A = pro(x[p + tap][channel]).
constexpr int SW_SEG = 8;    // 38 loads per item: the previous burst can be waited for with an encodable vmcnt

template <int NCH>
__global__ __launch_bounds__(256) void conv3x3_co4_wgrad_kernel(const SmallCoWgradArgs a) {
  constexpr int RG = 4 / NCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = wave % NCH, rg = wave / NCH;
  const int j = lane & 3;
  const bool affine = a.pro_mode == PRO_AFFINE_RELU || a.pro_mode == PRO_AFFINE;
  const float psc = affine ? a.pro_scale[chunk * 64 + lane] : 1.f;
  const float psh = affine ? a.pro_shift[chunk * 64 + lane] : 0.f;
  // branch-free prologue: v*scale+shift (1, 0 when not affine), then max(v, slope*v) with slope 1 / 0 / 0.2
  const float slope = (a.pro_mode == PRO_RELU || a.pro_mode == PRO_AFFINE_RELU) ? 0.f : (a.pro_mode == PRO_LRELU ? 0.2f : 1.f);

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const int total_rows = a.B * a.H;
  const int row0 = blockIdx.x * a.rows_per_block;
  const int row1 = min(row0 + a.rows_per_block, total_rows);
  const int nseg = (a.W + SW_SEG - 1) / SW_SEG;
  const int my_rows = (row1 - row0 - rg + RG - 1) / RG;          // rows row0 + rg + RG*i < row1
  const int nitems = my_rows > 0 ? my_rows * nseg : 0;
  const float* xc = a.x + chunk * 64 + lane;

  struct Item { float x[3][SW_SEG + 2]; float d[SW_SEG]; };
  auto issue = [&](int q, Item& it) {
    const int qq = min(q, max(nitems - 1, 0));
    const int ri = qq / nseg, seg = qq - ri * nseg;
    const int row = min(row0 + rg + RG * ri, total_rows - 1);
    const int b = row / a.H, oy = row - b * a.H;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int iy = min(max(oy + d - 1, 0), a.H - 1);
      const float* xr = xc + ((long)b * a.H + iy) * a.W * a.Ci;
#pragma unroll
      for (int k = 0; k < SW_SEG + 2; ++k) it.x[d][k] = xr[(long)min(max(seg * SW_SEG - 1 + k, 0), a.W - 1) * a.Ci];
    }
    const float* dr = a.dy + (long)row * a.W * 4 + j;
#pragma unroll
    for (int k = 0; k < SW_SEG; ++k) it.d[k] = dr[min(seg * SW_SEG + k, a.W - 1) * 4];
  };
  auto consume = [&](int q, Item& it) {
    const bool live = q < nitems;
    const int qq = min(q, max(nitems - 1, 0));
    const int ri = qq / nseg, seg = qq - ri * nseg;
    const int row = min(row0 + rg + RG * ri, total_rows - 1);
    const int oy = row % a.H;
    // validity masks are wave-uniform scalars
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool rok = (oy + d - 1) >= 0 && (oy + d - 1) < a.H;
#pragma unroll
      for (int k = 0; k < SW_SEG + 2; ++k) {
        const int ix = seg * SW_SEG - 1 + k;
        float v = fmaf(it.x[d][k], psc, psh);
        v = fmaxf(v, v * slope);
        it.x[d][k] = (rok && ix >= 0 && ix < a.W) ? v : 0.f;     // padding is zero AFTER the transform
      }
    }
#pragma unroll
    for (int k = 0; k < SW_SEG; ++k) {
      const float dv = (live && seg * SW_SEG + k < a.W) ? it.d[k] : 0.f;
      bsum += dv;
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2)
          acc[d * 3 + s2] = __builtin_amdgcn_mfma_f32_4x4x1f32(it.x[d][k + s2], dv, acc[d * 3 + s2], 0, 0, 0);
    }
  };

  Item bufA, bufB;
  if (nitems > 0) {
    issue(0, bufA);
    // sched_barrier(0): the scheduler may not move anything across -- otherwise it sinks each load next to
    // its first use to save registers and the burst is gone
    for (int q = 0; q < nitems; q += 2) {
      issue(q + 1, bufB);
      __builtin_amdgcn_sched_barrier(0);
      consume(q, bufA);
      __builtin_amdgcn_sched_barrier(0);
      issue(q + 2, bufA);
      __builtin_amdgcn_sched_barrier(0);
      consume(q + 1, bufB);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // lane 4g+jj, element e of acc[t] = dW[jj][t*Ci + chunk*64 + 4g + e]; row groups are summed through LDS in a
  // fixed order, then ONE slab per workgroup
  __shared__ __attribute__((aligned(16))) float red[RG > 1 ? RG - 1 : 1][NCH][9][64 * 4];
  __shared__ float bred[4][4];
  float* out = a.slab + (long)blockIdx.x * a.slab_stride;
  if (RG > 1 && rg > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(&red[rg - 1][chunk][t][lane * 4]) = acc[t];
  }
  if (lane < 4) bred[wave][lane] = bsum;
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      f32x4 s = acc[t];
#pragma unroll
      for (int r = 0; r < RG - 1; ++r) s += *reinterpret_cast<const f32x4*>(&red[r][chunk][t][lane * 4]);
      *reinterpret_cast<f32x4*>(out + (long)j * a.Kp + t * a.Ci + chunk * 64 + (lane & ~3)) = s;
    }
  }
  if (a.bias_off >= 0 && tid < 4) {
    // every wave of chunk 0 summed the dy columns of its rows: add the RG row groups (waves 0, NCH, 2*NCH, ...)
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) t += bred[r * NCH][tid];
    out[a.bias_off + tid] = t;
  }
}


// ---- 3x3 convolution FROM four input channels (RGB + pad), stride 1, pad 1: the discriminators' first layer (round 5) ----------
// K = 36: on the implicit GEMM's tiles the launch is one LDS round trip and one epilogue per 64 x 64 outputs with nothing to
// overlap them -- 35 / 62 us for 67 / 134 MB of output (a plain fill of that size: 11 / 20 us; tools/probe/ci4_rate.py).
// Here a workgroup takes 128 consecutive pixels x ALL output channels: the packed weights go through LDS once (coalesced), a wave
// loads the nine taps of its 32 pixels once and then walks the 32-channel tiles -- 18 MFMAs, turn the tile through LDS, four
// 1 KB stores, next tile -- without ever waiting for a store: ~16 KB per wave in flight, which is what the write stream needs
// (one small wave per tile, retired when its 4 KB are acknowledged: 2.8 TB/s; fat persistent waves with the weights in registers:
// bound by their own serial chain, 24 / 35 us).  The product is taken TRANSPOSED (D[channel][pixel] = W . X) so that an
// accumulator quad is four consecutive channels of one pixel; the LDS turn makes a store instruction eight whole 128-byte lines
// (16 bytes per lane strided by a pixel row reached a third of the store rate).  k = tap * 4 + c as in the packed weights; the
// sums run k = 0, 1, .. 35 as the implicit GEMM's do.  Roofline: the output stream (HBM).
struct FirstConvArgs {
  const float* x;          // [B,H,W,4]
  const float* w;          // packed [Co][Kp], Kp >= 36
  float* y;                // [B,H,W,Co]
  const float* bias;       // [Co] or null
  const float* scale0;     // as ConvGemmArgs: rows < scale_split use *scale0, the others *scale1; null: out_scale
  const float* scale1;
  float out_scale;
  int scale_split;
  int M, H, W, Co, Kp, tiles;       // tiles = Co / 32
  FastDiv dW, dH;
};

constexpr int FC_RS = 36;                  // floats per pixel row of a wave's exchange tile (32 channels + 16 bytes: conflict-free)

__global__ __launch_bounds__(256) void conv3x3_ci4_kernel(const FirstConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];        // [Co][36] weights | [Co] bias | [4 waves][32][FC_RS]
  float* const ws = lds;
  float* const bs = lds + a.Co * 36;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
  float* const xt = bs + a.Co + wave * (32 * FC_RS);
  // B operand first (the longest latency): the nine taps of pixel m (zeros outside the image and past M: out-of-range bit of
  // the buffer offset); a lane multiplies channels (h, 2 + h) of each tap
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((unsigned)a.M * 16u), 0x00020000);
  const int g = blockIdx.x * 4 + wave, m = g * 32 + j;
  float xv[18];
  {
    const unsigned q1 = fdiv((unsigned)m, a.dW);
    const int ox = m - (int)q1 * a.W;
    const unsigned b = fdiv(q1, a.dH);
    const int oy = (int)q1 - (int)b * a.H;
    const bool mv = m < a.M;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int yy = oy + r - 1, xx = ox + q - 1;
        const bool ok = mv && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        const unsigned off = (unsigned)((m + (r - 1) * a.W + (q - 1)) * 16 + 4 * h) | (ok ? 0u : 0x80000000u);
        xv[2 * (r * 3 + q)] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, off, 0, 0));
        xv[2 * (r * 3 + q) + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, off, 8, 0));
      }
  }
  for (int i = tid; i < a.Co * 9; i += 256) {          // weight rows of 9 x 16 bytes: coalesced, once per workgroup
    const int n = i / 9, q = i - 9 * n;
    *reinterpret_cast<f32x4*>(ws + n * 36 + 4 * q) = *reinterpret_cast<const f32x4*>(a.w + (long)n * a.Kp + 4 * q);
  }
  for (int i = tid; i < a.Co; i += 256) bs[i] = a.bias ? a.bias[i] : 0.f;
  __syncthreads();
  const float sc0 = a.scale0 ? a.scale0[0] : a.out_scale, sc1 = a.scale1 ? a.scale1[0] : a.out_scale;
  const __amdgpu_buffer_rsrc_t ysrc =
      __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((unsigned)a.M * (unsigned)a.Co * 4u), 0x00020000);
  const int prl = lane >> 3, ccl = lane & 7;           // read-back: this lane's row (of 8 per store) and channel quad
  float scv[4];
  unsigned yoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mm = g * 32 + i * 8 + prl;
    scv[i] = mm < a.scale_split ? sc0 : sc1;
    yoff[i] = (unsigned)mm * (unsigned)(a.Co * 4) + (unsigned)(ccl * 16);
  }
  for (int t = 0; t < a.tiles; ++t) {
    // A operand: W[n = 32 t + j][k = 2 i + h], i = 0 .. 17 (row stride 36 words: conflict-free)
    const float* wr = ws + (32 * t + j) * 36 + h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[4 * q], xv[2 * q], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[4 * q + 2], xv[2 * q + 1], acc, 0, 0, 0);
    }
    // this lane holds pixel j, channels 8 q + 4 h + (0 .. 3) of the tile in accumulator elements 4 q .. 4 q + 3: parked in the
    // wave's exchange tile [pixel][channel], read back by rows (8 lanes = one pixel's 128 bytes), scaled and biased there
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = acc[4 * q + e];
      *reinterpret_cast<f32x4*>(xt + j * FC_RS + 8 * q + 4 * h) = y;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (one wave: its LDS instructions execute in order)
    const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + 32 * t + 4 * ccl);
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(xt + (i * 8 + prl) * FC_RS + 4 * ccl);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 y = v[i] * scv[i] + bq;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y), ysrc, yoff[i],
                                             t * 128, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the tile is read before the next one is parked
  }
}

}  // namespace diagan

using namespace diagan;

// 1 if diagan_conv3x3_co4 supports this geometry (host-side check, no device work)
DIAGAN_API int diagan_conv3x3_co4_supported(int Ci, int Co, int R, int S, int sy, int dr, int off, int up) {
  const bool geo = R == 3 && S == 3 && sy == 1 && up == 1 && ((dr == 1 && off == -1) || (dr == -1 && off == 1));
  return geo && Co == 4 && Ci >= SC_CC && (Ci % SC_CC) == 0;
}

DIAGAN_API int diagan_conv3x3_co4(const float* x, const float* w, float* y, const float* bias, const float* residual,
                                  const float* pro_scale, const float* pro_shift, int pro_mode, int B, int H, int W,
                                  int Ci, int dr, int off, int Kp, int group_imgs, void* stream) {
  DG_REQUIRE(x && w && y, "conv3x3_co4: null tensor");
  DG_REQUIRE(group_imgs >= 0 && (group_imgs == 0 || B % group_imgs == 0), "conv3x3_co4: group_imgs=%d must divide B=%d", group_imgs, B);
  DG_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_co4: bad dims");
  DG_REQUIRE(diagan_conv3x3_co4_supported(Ci, 4, 3, 3, 1, dr, off, 1), "conv3x3_co4: unsupported geometry Ci=%d dr=%d off=%d", Ci, dr, off);
  DG_REQUIRE(Kp >= 9 * Ci && pro_mode >= 0 && pro_mode <= 4, "conv3x3_co4: bad Kp / pro_mode");
  DG_REQUIRE(!(pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE) || (pro_scale && pro_shift), "conv3x3_co4: affine prologue needs scale/shift");
  DG_REQUIRE((long)B * H * W * Ci * 4 < (1L << 31), "conv3x3_co4: tensors must be smaller than 2 GiB");
  SmallCoArgs a{x, w, y, bias, residual, pro_scale, pro_shift, pro_mode, B, H, W, Ci, Kp, dr, off,
                cdiv(W, SC_TW), cdiv(H, SC_TH), group_imgs};
  const long blocks = (long)B * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL(conv3x3_co4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("conv3x3_co4");
}


// 1 if diagan_conv3x3_ci4 takes this layer: forward 3x3 / stride 1 / pad 1 from 4 input channels, Co a multiple of 32
DIAGAN_API int diagan_conv3x3_ci4_supported(int Ci, int Co, int R, int S, int sy, int dr, int off, int up) {
  static const int env = getenv("DIAGAN_CONV_CI4") ? atoi(getenv("DIAGAN_CONV_CI4")) : 1;
  return env && R == 3 && S == 3 && sy == 1 && up == 1 && dr == 1 && off == -1 && Ci == 4 && Co >= 32 && Co <= 256 && (Co % 32) == 0;
}

// y = conv3x3(x) * scale + bias for such a layer (no prologue, residual, mask or statistics: the discriminators' first
// convolution); scale = *scale0 for pixel rows < scale_split, *scale1 behind it (both null: out_scale), as diagan_conv_gemm
DIAGAN_API int diagan_conv3x3_ci4(const float* x, const float* w, float* y, const float* bias, float out_scale, const float* scale0,
                                  const float* scale1, int scale_split, int B, int H, int W, int Co, int Kp, void* stream) {
  DG_REQUIRE(x && w && y, "conv3x3_ci4: null tensor");
  DG_REQUIRE(B > 0 && H > 0 && W > 0 && Kp >= 36 && (Kp & 3) == 0, "conv3x3_ci4: bad dims");
  DG_REQUIRE(diagan_conv3x3_ci4_supported(4, Co, 3, 3, 1, 1, -1, 1), "conv3x3_ci4: Co=%d unsupported (a multiple of 32)", Co);
  DG_REQUIRE((scale0 == nullptr) == (scale1 == nullptr), "conv3x3_ci4: scale0 and scale1 come together");
  DG_REQUIRE((long)B * H * W * Co * 4 < (1L << 31), "conv3x3_ci4: tensors must be smaller than 2 GiB");
  FirstConvArgs a;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.scale0 = scale0; a.scale1 = scale1; a.out_scale = out_scale;
  a.scale_split = scale0 ? scale_split : 0x7fffffff;
  a.M = B * H * W; a.H = H; a.W = W; a.Co = Co; a.Kp = Kp; a.tiles = Co / 32;
  a.dW = make_fastdiv((unsigned)W);
  a.dH = make_fastdiv((unsigned)H);
  const size_t lds = (size_t)(Co * 36 + Co + 4 * 32 * FC_RS) * sizeof(float);        // 37 KB at 128 channels
  DG_REQUIRE(lds <= 64 * 1024, "conv3x3_ci4: Co=%d needs %zu bytes of LDS", Co, lds);
  hipLaunchKernelGGL(conv3x3_ci4_kernel, dim3(cdiv(cdiv(a.M, 32), 4)), dim3(256), lds, (hipStream_t)stream, a);
  return check_launch("conv3x3_ci4");
}

// 1 if diagan_conv3x3_co4_wgrad supports the layer (3x3, stride 1, pad 1, 4 output channels)
DIAGAN_API int diagan_conv3x3_co4_wgrad_supported(int Ci, int Co, int R, int S, int sy, int dr, int off, int up) {
  return R == 3 && S == 3 && sy == 1 && up == 1 && dr == 1 && off == -1 && Co == 4 && (Ci == 64 || Ci == 128 || Ci == 256);
}

// number of slabs (= workgroups) diagan_conv3x3_co4_wgrad writes for this problem
DIAGAN_API int diagan_conv3x3_co4_wgrad_splits(int B, int H) {
  const int rows = B * H;
  int per = cdiv(rows, 512);          // <= 512 workgroups (2 per CU): one 36*Ci-float slab each
  if (per < 4) per = 4;
  return cdiv(rows, per);
}

DIAGAN_API int diagan_conv3x3_co4_wgrad(const float* dy, const float* x, float* slab, int64_t slab_stride,
                                        int64_t bias_off, const float* pro_scale, const float* pro_shift, int pro_mode,
                                        int B, int H, int W, int Ci, int Kp, void* stream) {
  DG_REQUIRE(dy && x && slab, "conv3x3_co4_wgrad: null tensor");
  DG_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_co4_wgrad: bad dims");
  DG_REQUIRE(Ci == 64 || Ci == 128 || Ci == 256, "conv3x3_co4_wgrad: Ci=%d unsupported (64, 128, 256)", Ci);
  DG_REQUIRE(Kp >= 9 * Ci && pro_mode >= 0 && pro_mode <= 4, "conv3x3_co4_wgrad: bad Kp / pro_mode");
  DG_REQUIRE(!(pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE) || (pro_scale && pro_shift), "conv3x3_co4_wgrad: affine prologue needs scale/shift");
  DG_REQUIRE(slab_stride >= (int64_t)4 * Kp && (bias_off < 0 || bias_off + 4 <= slab_stride), "conv3x3_co4_wgrad: slab_stride too small");
  const int splits = diagan_conv3x3_co4_wgrad_splits(B, H);
  SmallCoWgradArgs a{dy, x, slab, pro_scale, pro_shift, pro_mode, B, H, W, Ci, Kp, (long)slab_stride, (long)bias_off,
                     cdiv(B * H, splits)};
  // K padding columns of the slab (k >= 9*Ci) are never written by the kernel: the packed layout has none for these Ci
  DG_REQUIRE(Kp == 9 * Ci, "conv3x3_co4_wgrad: padded Kp=%d != 9*Ci", Kp);
  hipStream_t st = (hipStream_t)stream;
  if (Ci == 256) hipLaunchKernelGGL(conv3x3_co4_wgrad_kernel<4>, dim3(splits), dim3(256), 0, st, a);
  else if (Ci == 128) hipLaunchKernelGGL(conv3x3_co4_wgrad_kernel<2>, dim3(splits), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv3x3_co4_wgrad_kernel<1>, dim3(splits), dim3(256), 0, st, a);
  return check_launch("conv3x3_co4_wgrad");
}
