// Implicit-GEMM weight gradient on the fp32 matrix cores, split-K over pixels.
//
// Replaces the weight-gradient half of autograd's conv2d / conv_transpose2d backward in the
// SNGAN / DCGAN stacks (the errD.backward() / errG.backward() calls of the train steps,
// diagan-pkg/diagan/models/topk_models.py:90, mnist.py:126 and torch_mimicry's base train_step).
//
//   dWp[n][k] = sum_m dY[m][n] * A(m,k),  k = (r,s,c),  A = pro(gathered X) exactly as in forward
//
// GEMM view: rows n (Co), cols k (packed weight index), reduction over pixels m (65k..262k long),
// so the output is small and the reduction is split over gridDim.y workgroups that each write one
// fp32 slab; a second kernel sums the slabs in a fixed order (deterministic, no float atomics).
// LDS holds both operands pixel-major ([32 pixels][128]) straight from their NHWC rows, so the
// MFMA fragments are conflict-free ds_read_b32 (lanes = consecutive channels).
//
// Roofline: MFMA fp32 (algorithmic FLOP = 2*M*Co*K).
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>


namespace diagan {
bool wgrad_wino_supported(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off, int up,
                          int Kp);                                    // conv_wgrad_wino.hip
int wgrad_wino_splits(int B, int Ho, int Wo, int Ci, int Co);
int launch_wgrad_wino(WgradArgs a, int splits, int segments, hipStream_t st);
int launch_wgrad_wino_batched(const WgradArgs* jobs, const int* splits, const int* segments, int n, hipStream_t st);
int wgrad_wino_batch_max();
bool wgrad_x3_takes(const WgradArgs& a, bool x3_on);                  // conv_wgrad_x3.hip
int launch_wgrad_x3(const WgradArgs& a, int splits, hipStream_t st);
void wgrad_x3_set(int on);


// P2: Ho and Wo are powers of two -- pixel coordinates come from shifts and masks of the pixel index instead of
// the incrementally advanced (b, oy, ox) registers (the gather arithmetic is ~7 % of the kernel otherwise)
// bid / nblk: this workgroup's index and the workgroup count of ITS layer (the whole grid for a one-layer launch, a range
// of the grid in conv_wgrad_batched_kernel)
template <int BNn, int BNk, int PRO, bool P2>
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs& a, const int bid, const int nblk) {
  constexpr int BK = 32;                       // pixels per K-step
  constexpr int TM = BNn / 64, TN = BNk / 64;  // 2x2 waves
  constexpr int AC = BNn / 4, BC = BNk / 4;    // 16-byte chunks per pixel row
  constexpr int AJ = BK * AC / 256, BJ = BK * BC / 256;
  constexpr int APR = 256 / AC, BPR = 256 / BC;  // pixel rows covered per pass
  // BOX (round 5): the gathered image is never materialised -- g.Hi x g.Wi is the (H+1) x (W+1) grid of 2x2 box sums and every
  // gathered 16-byte piece is the sum of FOUR loads from the H x W tensor, in diagan_boxsum2's order (bit-identical operands)
  constexpr bool BOX = PRO == PRO_BOX || PRO == PRO_BOX_RELU;
  constexpr int BQ = BOX ? 4 : 1;
  __shared__ __attribute__((aligned(16))) float As[2][BK * BNn];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * BNk];

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_k = (g.Kp + BNk - 1) / BNk;
  // 1-D grid of tiles x splits, XCD-aware: consecutive logical ids -- the tiles of ONE split, which share its dy
  // rows and re-read the same x rows tap by tap -- land on the same XCD (= the same L2)
  const int logical = xcd_remap(bid, nblk);
  const int split = logical / a.tiles, tile = logical - split * a.tiles;
  const int n0 = (tile / tiles_k) * BNn, k0 = (tile % tiles_k) * BNk;
  const int seg = split / a.splits_per_seg, sub = split - seg * a.splits_per_seg;
  const int step0 = seg * a.seg_steps + sub * a.steps_per_split;
  const int step1 = min(step0 + a.steps_per_split, (seg + 1) * a.seg_steps);

  const int pro_mode = PRO >= 0 ? PRO : a.pro_mode;   // compile-time in the specialised kernels
  // A' loader (dy rows): fixed channel chunk, pixel rows ap + APR*j
  const int ac = tid % AC, ap = tid / AC;
  const int an = n0 + ac * 4;
  const unsigned a_kill = an < g.Co ? 0u : 0x80000000u;   // bit 31 -> beyond num_records -> zeros
  // B' loader (gathered x): fixed (tap, c) per thread
  const int bc = tid % BC, bp = tid / BC;
  const int kf = k0 + bc * 4;
  const int tap = kf / g.Ci, kc = kf - tap * g.Ci;
  const int kr = tap / g.S, ks = tap - kr * g.S;
  const bool b_ok = kf < g.K;
  const int dyo = kr * g.dr + g.off, dxo = ks * g.dr + g.off;
  const int upm = g.up - 1, ush = g.up >> 1;
  const int pstep = (g.Ci * 4) >> ush;                      // bytes per numerator unit along x (see conv_gemm.hip)
  const int img_bytes = BOX ? (g.Hi - 1) * (g.Wi - 1) * g.Ci * 4 : g.Hi * g.Wi * g.Ci * 4;
  const unsigned ylim = (unsigned)g.Hi << ush, xlim = (unsigned)g.Wi << ush;

  // branch-free raw buffer loads (out-of-range -> zeros); prologue applied at LDS-store time so the
  // loads stay in flight under the MFMAs (see conv_gemm.hip)
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dy), 0, (int)((unsigned)a.M * g.Co * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((unsigned)g.B * (BOX ? (g.Hi - 1) * (g.Wi - 1) : g.Hi * g.Wi) * g.Ci * 4u), 0x00020000);
  const bool affine = pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE;
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if (affine && b_ok) {
    psc = *reinterpret_cast<const f32x4*>(a.pro_scale + kc);
    psh = *reinterpret_cast<const f32x4*>(a.pro_shift + kc);
  }
  // pixel coordinates of this thread's BJ rows, advanced by 32 pixels per K-step without divisions
  int pb[BJ], py[BJ], px[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int m = step0 * BK + bp + BPR * j;
    const unsigned t = fdiv((unsigned)m, a.dWo);
    px[j] = m - (int)t * g.Wo;
    const unsigned b = fdiv(t, a.dHo);
    py[j] = (int)t - (int)b * g.Ho;
    pb[j] = (int)b;
  }
  int mstep = step0 * BK;
  unsigned xoff[BJ];         // (P2 && same) linear offsets of the gathered rows
  const unsigned xstep = (unsigned)BK * g.Ci * 4u;
#pragma unroll
  for (int j = 0; j < BJ; ++j)
    xoff[j] = (unsigned)((step0 * BK + bp + BPR * j + dyo * g.Wi + dxo) * g.Ci * 4 + kc * 4);
  unsigned aoff[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) aoff[j] = (((unsigned)(step0 * BK + ap + APR * j)) * g.Co + an) * 4u;
  const unsigned astep = (unsigned)BK * g.Co * 4u;
  f32x4 ra[AJ], rb[BJ * BQ];
  unsigned bmask = 0;
  // bias gradient = column sums of dy, taken by the k0 == 0 tile column from the staging REGISTERS of the dy
  // loader (every thread adds its own 16-byte pieces; one LDS reduction over the APR pixel-row groups at the end).
  // (Summing from the LDS tile cost those workgroups an LDS round trip per s-step -- and a launch is one round
  // of workgroups, so its slowest workgroups set the kernel time.)
  const bool bias_tile = a.bias_off >= 0 && k0 == 0;
  f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
  // one 16-byte load of the next tile (piece p < AJ: dy rows; else gathered x rows with their coordinates)
  auto load_piece = [&](int p) {
    if (p < AJ) {     // rows past M fall outside num_records
      ra[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ysrc, aoff[p] | a_kill, 0, 0));
      aoff[p] += astep;
      return;
    }
    const int j = p - AJ;
    if (j == 0) bmask = 0;
    if (!BOX && P2 && a.same) {
      // same-size stride-1 conv on a power-of-two image: the gathered offset is LINEAR in the pixel index
      // (m*Ci*4 + a per-thread tap constant); only the border test needs (oy, ox), from shifts and masks
      const int m = mstep + bp + BPR * j;
      const int yn = ((m >> a.lgW) & (g.Ho - 1)) + dyo, xn = (m & (g.Wo - 1)) + dxo;
      const bool ok = b_ok && m < a.M && (unsigned)yn < (unsigned)g.Hi && (unsigned)xn < (unsigned)g.Wi;
      const unsigned okb = ok ? 1u : 0u;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[j] | ((okb ^ 1u) << 31), 0, 0));
      xoff[j] += xstep;
      bmask |= okb << j;
      if (j == BJ - 1) mstep += BK;
      return;
    }
    int pbj, pyj, pxj;
    if (P2) {
      const int m = mstep + bp + BPR * j;
      pxj = m & (g.Wo - 1);
      pyj = (m >> a.lgW) & (g.Ho - 1);
      pbj = m >> a.lgHW;
    } else {
      pbj = pb[j]; pyj = py[j]; pxj = px[j];
    }
    const int yn = pyj * g.sy + dyo, xn = pxj * g.sy + dxo;
    if constexpr (BOX) {
      const int H = g.Hi - 1, W = g.Wi - 1;
      const bool okp = b_ok && pbj < g.B;
#pragma unroll
      for (int q = 0; q < 4; ++q) {     // the window whose lower-right corner is (yn, xn): rows first, as boxsum2_kernel adds them
        const int y = yn - 1 + (q >> 1), x = xn - 1 + (q & 1);
        const bool ok = okp && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const unsigned off = (unsigned)(pbj * img_bytes + (y * W + x) * pstep + kc * 4) | (ok ? 0u : 0x80000000u);
        rb[j * 4 + q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
      }
    } else {
    const bool ok = b_ok && pbj < g.B && (unsigned)yn < ylim && (unsigned)xn < xlim && ((yn | xn) & upm) == 0;
    const unsigned okb = ok ? 1u : 0u;
    const unsigned off = (unsigned)(pbj * img_bytes + (yn * g.Wi + xn) * pstep + kc * 4) | ((okb ^ 1u) << 31);
    rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
    bmask |= okb << j;
    }
    if (!P2) {      // advance to the same row of the next K-step
      int x = px[j] + a.adv_x, y = py[j] + a.adv_y, b = pb[j] + a.adv_b;
      const int cx = x >= g.Wo ? 1 : 0;
      x -= cx ? g.Wo : 0;
      y += cx;
      const int cy = y >= g.Ho ? 1 : 0;
      y -= cy ? g.Ho : 0;
      b += cy;
      px[j] = x; py[j] = y; pb[j] = b;
    }
    if (j == BJ - 1) mstep += BK;
  };
  auto load_tiles = [&]() {
#pragma unroll
    for (int p = 0; p < AJ + BJ; ++p) load_piece(p);
  };
  // one 16-byte piece of the next tile (prologue applied on the way from the staging registers to LDS)
  auto store_piece = [&](int buf, int p) {
    if (p < AJ) {
      *reinterpret_cast<f32x4*>(&As[buf][(ap + APR * p) * BNn + ac * 4]) = ra[p];
      if (bias_tile) bacc += ra[p];
    } else {
      const int j = p - AJ;
      f32x4 v = rb[j * BQ];
      if constexpr (BOX) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 t = rb[j * 4 + q];
          if (PRO == PRO_BOX_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], 0.f);
          }
          sum += t;
        }
        v = sum * 0.25f;
      } else if (pro_mode != PRO_NONE) {
        if (affine) v = v * psc + psh;
        if (pro_mode == PRO_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
        } else if (pro_mode != PRO_AFFINE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (affine) v *= (float)((bmask >> j) & 1u);   // padding is zero AFTER the transform
      }
      *reinterpret_cast<f32x4*>(&Bs[buf][(bp + BPR * j) * BNk + bc * 4]) = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int p = 0; p < AJ + BJ; ++p) store_piece(buf, p);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fi = lane & 31, fh = lane >> 5;
  if (step0 < step1) {
    load_tiles();
    store_tiles(0);
  }
  __syncthreads();
  // One K-step (32 pixels): next tile's loads first; its AJ+BJ staging pieces are written to the other LDS buffer
  // one per s-step during the LAST s-steps, in the shadow of the MFMAs (see conv_gemm.hip); the bias column sums
  // are spread over the s-steps the same way.
  constexpr int NS = BK / 2, NP = AJ + BJ;
  static_assert(2 * NP <= NS, "load and store pieces of a tile must not share an s-step");
  auto kstep = [&](int step, auto has_next) {
    const int cur = (step - step0) & 1;
    const float* Ac = As[cur];
    const float* Bc = Bs[cur];
    // MFMA fragments are read PF s-steps AHEAD (PF + 1 register sets): the LDS latency of step s+1 is covered by the
    // 4 x 64 cycles of step s instead of opening a bubble in the matrix pipe every s-step
    constexpr int PF = 1;          // s-steps of read-ahead (2 measured the same)
    float fa[PF + 1][TM], fb[PF + 1][TN];
    auto read_frag = [&](int s, int set) {
      const int p = 2 * s + fh;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][i] = Ac[p * BNn + wm * (TM * 32) + i * 32 + fi];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[set][j] = Bc[p * BNk + wn * (TN * 32) + j * 32 + fi];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) read_frag(s, s);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + PF < NS) read_frag(s + PF, (s + PF) % (PF + 1));
      // (the scheduler is free to spread a piece's instructions between the MFMAs that follow it; the barrier
      //  BEFORE a store piece keeps it from being hoisted to where its load has not landed yet)
      if (decltype(has_next)::value && s < NP) load_piece(s);       // next tile's loads: first NP s-steps
      if (decltype(has_next)::value && s >= NS - NP) {               // ... and its LDS stores in the last NP s-steps
        __builtin_amdgcn_sched_barrier(0);
        store_piece(cur ^ 1, s - (NS - NP));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s % (PF + 1)][i], fb[s % (PF + 1)][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int step = step0; step + 1 < step1; ++step) kstep(step, std::true_type{});
  if (step0 < step1) kstep(step1 - 1, std::false_type{});

  float* out = a.slab + (long)split * a.slab_stride;
  if (bias_tile) {     // (block-uniform) tiles are done with: reuse As for the APR partial rows
    __syncthreads();
    *reinterpret_cast<f32x4*>(&As[0][ap * BNn + ac * 4]) = bacc;
    __syncthreads();
    if (tid < BNn && n0 + tid < g.Co) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < APR; ++r) t += As[0][r * BNn + tid];
      out[a.bias_off + n0 + tid] = t;
    }
  }
  // raw buffer stores: one lane offset per accumulator tile + a scalar row offset per element; rows past Co fall
  // outside num_records, columns past Kp get the out-of-range bit (see the epilogue of conv_gemm.hip)
  const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((unsigned)g.Co * g.Kp * 4u), 0x00020000);
  const unsigned rowbytes = (unsigned)g.Kp * 4u;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int k = k0 + wn * (TN * 32) + j * 32 + fi;
      const int nrow = n0 + wm * (TM * 32) + i * 32 + 4 * fh;
      const unsigned vbase = k < g.Kp ? ((unsigned)nrow * g.Kp + k) * 4u : 0x80000000u;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float av = acc[i][j][e];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av), osrc, vbase, (int)(((e & 3) + 8 * (e >> 2)) * rowbytes), 0);
      }
    }
}

template <int BNn, int BNk, int PRO = -1, bool P2 = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  conv_wgrad_body<BNn, BNk, PRO, P2>(a, blockIdx.x, gridDim.x);
}

// Round 5: several layers of one backward pass in ONE launch (see conv_wgrad_wino.hip: conv_wgrad_wino_batched_kernel; the
// layers of a launch share the kernel's template, i.e. tile, prologue mode and power-of-two image)
template <int BNn, int BNk, int PRO = -1, bool P2 = false>
__global__ __launch_bounds__(256) void conv_wgrad_batched_kernel(const WgradBatchArgs b) {
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.blk0[j + 1]) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  const int local = (int)blockIdx.x - b.blk0[j];
  if (local >= b.cnt[j]) return;
  conv_wgrad_body<BNn, BNk, PRO, P2>(b.a[j], local, b.cnt[j]);
}


// out[i] (+)= sum_s slab[s][i]; optionally per-block partial of <G, W> for the SN backward
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int splits,
                                                           long n4, float* __restrict__ out, int accumulate,
                                                           const float* __restrict__ w,
                                                           double* __restrict__ dot_partials) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  double dot = 0.0;
  if (i < n4) {
    f32x4 s = reinterpret_cast<const f32x4*>(slab)[i];
    for (int k = 1; k < splits; ++k) s += reinterpret_cast<const f32x4*>(slab)[(long)k * n4 + i];
    if (w) {
      const f32x4 wv = reinterpret_cast<const f32x4*>(w)[i];
      dot = (double)s[0] * wv[0] + (double)s[1] * wv[1] + (double)s[2] * wv[2] + (double)s[3] * wv[3];
    }
    if (accumulate) s += reinterpret_cast<f32x4*>(out)[i];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
  if (dot_partials) {
    __shared__ double red[4];
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) dot_partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}


// ---- deferred, batched epilogue of a whole backward pass --------------------------------------
// One descriptor per parameterised layer; blockIdx.z selects the layer.  A layer whose backward
// covered two batched forwards (nctx == 2) has two slab halves, each with its own SN context; both
// contributions are added by the same thread (no two writers per gradient element).
struct WgFinish {          // mirrored by diagan_wgrad_layer in include/diagan_hip.h (120 bytes)
  float* slab[2];          // [splits][stride] per context: weight partials (+ bias partials)
  const float* u[2];       // SN: u', v, state of the forward each context belongs to
  const float* v[2];
  const float* state[2];
  double* partials[2];     // SN: one fp64 partial of <G, W> per block of phase A, per context
  float* grad;             // flat gradient region of the layer: weight [Co*Kp] then bias
  const float* W;          // master weight (SN layers) or NULL
  long stride;
  int splits, n_elem, n_w, Kp;   // splits per context; n_elem = Co*Kp (+ Co if bias), n_w = Co*Kp
  int nctx, first_block; // first_block: index of the layer's first workgroup in the 1-D grid of the finish kernels
};

// The finish kernels run on a 1-D grid with exactly ceil(n_elem / (1024 fin_u(splits))) workgroups per layer (a [blocks of the largest
// layer] x [layers] grid launched 69 000 mostly empty workgroups for SNGAN-64 and spent its time dispatching them).
__device__ __forceinline__ int find_layer(const WgFinish* __restrict__ tab, int n_layers, int bid) {
  int l = 0;
  while (l + 1 < n_layers && tab[l + 1].first_block <= bid) ++l;
  return l;
}

// The layer's descriptor, made wave-uniform field by field (v_readfirstlane): the finish kernels build buffer resources from
// its pointers, and a resource the compiler cannot prove uniform costs a waterfall loop per load.
typedef unsigned u32x4_t __attribute__((__vector_size__(16)));
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uni(T* p) {
  const unsigned long v = (unsigned long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ WgFinish load_uniform(const WgFinish* __restrict__ d) {
  WgFinish L;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    L.slab[c] = uni(d->slab[c]); L.u[c] = uni(d->u[c]); L.v[c] = uni(d->v[c]); L.state[c] = uni(d->state[c]);
    L.partials[c] = uni(d->partials[c]);
  }
  L.grad = uni(d->grad); L.W = uni(d->W);
  L.stride = (long)uni((int)d->stride);                 // (the host checks 16 * stride * 4 < 2^31)
  L.splits = uni(d->splits); L.n_elem = uni(d->n_elem); L.n_w = uni(d->n_w); L.Kp = uni(d->Kp);
  L.nctx = uni(d->nctx); L.first_block = uni(d->first_block);
  return L;
}

// Elements of a layer handled by one workgroup of the finish kernels: 256 threads x U positions x 4 floats.  Every thread keeps
// U x R = 16 independent 16-byte loads in flight (R = splits of one position per round): U = 8, 4, 2, 1 for layers of at most
// 2, 4, 8 and more splits per context (fin_u).  Every layer's slab is about one round of workgroup tiles (~17 MB per
// context) whatever its size, so n_elem x splits -- and with it the number of workgroups per layer, n_elem / (1024 U) -- is
// about the same for all layers.  The loads are raw buffer loads over the R splits of a round: positions behind the
// layer's end carry bit 31 in their offset and splits behind the last one fall behind num_records, both read as zero WITHOUT a
// branch -- the first version of this kernel guarded every position with an `if`, the compiler put a `s_waitcnt vmcnt(0)`
// at each join and two loads per thread were in flight at a time (3.1 TB/s).
__host__ __device__ __forceinline__ int fin_u(int splits) { return splits <= 2 ? 8 : splits <= 4 ? 4 : splits <= 8 ? 2 : 1; }

template <int U>
__device__ __forceinline__ void fin_a(const WgFinish& L, int lb, double* red) {
  constexpr int ELEMS = 1024 * U;
  constexpr int R = 16 / U;
  const bool sn = L.W != nullptr;
  unsigned voff[U];                                      // byte offset of the position inside a split; bit 31: behind the layer
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long pos = (long)lb * ELEMS + (u * 256 + threadIdx.x) * 4;
    voff[u] = pos < L.n_elem ? (unsigned)pos * 4u : 0x80000000u;
  }
  const unsigned sbytes = (unsigned)L.stride * 4u;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[U];                                          // plain layers: runs on over both contexts
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = z4;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c >= L.nctx) break;
    float* sl = L.slab[c];
    for (int k = 0; k < L.splits; k += R) {              // fixed assignment of splits to partial sums: deterministic
      const int nr = min(R, L.splits - k);
      const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(sl + (long)k * L.stride, 0, (int)(nr * sbytes), 0x00020000);
      f32x4 t[U][R];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int u = 0; u < U; ++u)
          t[u][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, voff[u] + (unsigned)r * sbytes, 0, 0));
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int w = 1; w < R; w <<= 1)
#pragma unroll
          for (int r = 0; r + w < R; r += 2 * w) t[u][r] += t[u][r + w];
        acc[u] += t[u][0];
      }
    }
    if (sn) {                                            // G_c back into split 0 of its slab, the block's partial of <G_c, W>
      double dot = 0.0;
      const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.W), 0, L.n_w * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(sl, 0, L.n_elem * 4, 0x00020000);
      f32x4 wv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) wv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff[u], 0, 0));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const f32x4 g = acc[u];
        if (L.splits > 1)        // (one split: G_c IS split 0 of the slab already -- a fifth of this kernel's bytes on SNGAN-64)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, g), dst, voff[u], 0, 0);
        dot += (double)g[0] * wv[u][0] + (double)g[1] * wv[u][1] + (double)g[2] * wv[u][2] + (double)g[3] * wv[u][3];
        acc[u] = z4;
      }
      dot = wave_sum(dot);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
      __syncthreads();
      if (threadIdx.x == 0) L.partials[c][lb] = (red[0] + red[1]) + (red[2] + red[3]);
    }
  }
  if (!sn) {
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(L.grad, 0, L.n_elem * 4, 0x00020000);
    f32x4 og[U];
#pragma unroll
    for (int u = 0; u < U; ++u) og[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gsrc, voff[u], 0, 0));
#pragma unroll
    for (int u = 0; u < U; ++u)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, og[u] + acc[u]), gsrc, voff[u], 0, 0);
  }
}

template <typename F>
__device__ __forceinline__ void fin_dispatch(int U, F&& f) {
  switch (U) {
    case 8: f(std::integral_constant<int, 8>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    default: f(std::integral_constant<int, 1>{}); break;
  }
}

// phase A: G_c = sum over the splits of context c (fixed order).  plain layers: grad += sum_c G_c.
// SN layers: G_c kept in slab[c][0..] and the block's partial of <G_c, W> is written.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void wgrad_finish_a_kernel(const WgFinish* __restrict__ tab, int n_layers) {
  const WgFinish L = load_uniform(tab + find_layer(tab, n_layers, blockIdx.x));
  const int lb = blockIdx.x - L.first_block;          // workgroup index inside the layer
  const int U = fin_u(L.splits);
  if ((long)lb * 1024 * U >= L.n_elem) return;
  __shared__ double red[4];
  fin_dispatch(U, [&](auto u) { fin_a<decltype(u)::value>(L, lb, red); });
}

// between A and B: <G_c, W> of every SN layer = sum of its phase-A partials, ONCE per layer (one workgroup each,
// fixed reduction pattern: deterministic); stored behind the partials, at index nparts
__global__ __launch_bounds__(256) void wgrad_finish_dot_kernel(const WgFinish* __restrict__ tab) {
  const WgFinish L = tab[blockIdx.x];
  if (L.W == nullptr) return;
  __shared__ double red[256];
  const int fe = 1024 * fin_u(L.splits), nparts = (L.n_elem + fe - 1) / fe;
  for (int c = 0; c < L.nctx; ++c) {
    double d = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) d += L.partials[c][k];
    red[threadIdx.x] = d;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) L.partials[c][nparts] = red[0];
    __syncthreads();
  }
}

// phase B (SN layers): grad += sum_c (G_c - <G_c,W>/sigma_c * u_c^T v_c) / sigma_c ; bias part: += G_c
// Branch-free like phase A: positions behind the layer, the bias part's u / v and the absent second context read zero through
// out-of-range buffer offsets; two positions (ten 16-byte loads) per thread in flight at a time.
template <int U>
__device__ __forceinline__ void fin_b(const WgFinish& L, int lb) {
  constexpr int ELEMS = 1024 * U;
  constexpr int UC = U > 1 ? 2 : 1;
  const int nparts = (L.n_elem + ELEMS - 1) / ELEMS;
  float inv[2] = {0.f, 0.f}, coef[2] = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c >= L.nctx) break;
    inv[c] = L.state[c][1];
    coef[c] = (float)(L.partials[c][nparts] * (double)inv[c]);
  }
  const int Co = L.n_w / L.Kp;
  const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(L.grad, 0, L.n_elem * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t qsrc[2], vsrc[2], usrc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bool on = c < L.nctx;
    qsrc[c] = __builtin_amdgcn_make_buffer_rsrc(L.slab[c], 0, on ? L.n_elem * 4 : 0, 0x00020000);
    vsrc[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.v[c]), 0, on ? L.Kp * 4 : 0, 0x00020000);
    usrc[c] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L.u[c]), 0, on ? Co * 4 : 0, 0x00020000);
  }
#pragma unroll
  for (int u0 = 0; u0 < U; u0 += UC) {
    unsigned po[UC];
    bool okw[UC];
    f32x4 og[UC], gq[UC][2], vv[UC][2];
    float uu[UC][2];
#pragma unroll
    for (int i = 0; i < UC; ++i) {
      const long pos = (long)lb * ELEMS + ((u0 + i) * 256 + threadIdx.x) * 4;
      po[i] = pos < L.n_elem ? (unsigned)pos * 4u : 0x80000000u;
      okw[i] = pos < L.n_w;
      const unsigned n = okw[i] ? (unsigned)pos / (unsigned)L.Kp : 0u;       // (pos < n_w < 2^31: no 64-bit division)
      const unsigned ko = okw[i] ? ((unsigned)pos - n * (unsigned)L.Kp) * 4u : 0x80000000u;
      const unsigned no = okw[i] ? n * 4u : 0x80000000u;
      og[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gsrc, po[i], 0, 0));
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        gq[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qsrc[c], po[i], 0, 0));
        vv[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vsrc[c], ko, 0, 0));
        uu[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(usrc[c], no, 0, 0));
      }
    }
#pragma unroll
    for (int i = 0; i < UC; ++i) {
      f32x4 o = og[i];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 w = (gq[i][c] - (uu[i][c] * coef[c]) * vv[i][c]) * inv[c];
        o += okw[i] ? w : gq[i][c];
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o),
                                             gsrc, po[i], 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256) void wgrad_finish_b_kernel(const WgFinish* __restrict__ tab, int n_layers) {
  const WgFinish L = load_uniform(tab + find_layer(tab, n_layers, blockIdx.x));
  const int lb = blockIdx.x - L.first_block;
  const int U = fin_u(L.splits);
  if (L.W == nullptr || (long)lb * 1024 * U >= L.n_elem) return;
  fin_dispatch(U, [&](auto u) { fin_b<decltype(u)::value>(L, lb); });
}

// elements of a layer per workgroup of the finish kernels for a layer of `splits` splits per context (the host sizes
// first_block and the partial arrays with it)
extern "C" __attribute__((visibility("default"))) int diagan_wgrad_finish_block_elems(int splits) { return 1024 * fin_u(splits); }

}  // namespace diagan

using namespace diagan;
extern "C" int diagan_conv_gemm_get_wino(void);
extern "C" int diagan_conv_gemm_get_x3b(void);
extern "C" int diagan_conv_wgrad_splits(int M, int Co, int Kp);
extern "C" int diagan_conv_wgrad_uses_wino(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                           int off, int up, int Kp);

DIAGAN_API int diagan_wgrad_finish_batched(const void* table_dev, int n_layers, int64_t total_blocks, int any_sn,
                                           void* stream) {
  DG_REQUIRE(table_dev && n_layers > 0 && total_blocks > 0 && total_blocks < (1L << 31), "wgrad_finish_batched: bad args");
  static_assert(sizeof(WgFinish) == 128, "descriptor layout");
  const WgFinish* tab = (const WgFinish*)table_dev;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad_finish_a_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, tab, n_layers);
  if (any_sn) {
    hipLaunchKernelGGL(wgrad_finish_dot_kernel, dim3(n_layers), dim3(256), 0, st, tab);
    hipLaunchKernelGGL(wgrad_finish_b_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, tab, n_layers);
  }
  return check_launch("wgrad_finish_batched");
}

// tile (rows of Co x columns of packed k) of the weight-gradient GEMM
static void wgrad_tile(int Co, int Kp, int* bn, int* bk) {
  *bn = Co <= 64 ? 64 : 128;
  *bk = Kp <= 64 ? 64 : 128;
  if (*bn == 128 && *bk == 64) *bn = 64;   // no <128,64> instantiation
}

// the implicit-GEMM kernel's own fields of WgradArgs and its template choice (tile; power-of-two image)
static void wgrad_gemm_fields(WgradArgs& a, int* bn, int* bk, bool* p2) {
  const ConvGeom& g = a.g;
  wgrad_tile(g.Co, g.Kp, bn, bk);
  a.tiles = cdiv(g.Co, *bn) * cdiv(g.Kp, *bk);
  a.same = (g.Hi == g.Ho && g.Wi == g.Wo && g.sy == 1 && g.up == 1) ? 1 : 0;
  *p2 = (g.Ho & (g.Ho - 1)) == 0 && (g.Wo & (g.Wo - 1)) == 0;
  a.lgW = a.lgHW = 0;
  while ((1 << a.lgW) < g.Wo) ++a.lgW;
  while ((1 << a.lgHW) < g.Ho * g.Wo) ++a.lgHW;
}

// argument checks and the launch-independent part of WgradArgs (shared by the one-layer and the batched entry point)
static int wgrad_fill_args(WgradArgs& a, const float* dy, const float* x, float* slab, int splits, int segments,
                           int64_t slab_stride, int64_t bias_off, const float* pro_scale, const float* pro_shift, int pro_mode, int B,
                           int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off, int up, int Kp) {
  DG_REQUIRE(dy && x && slab, "conv_wgrad: null tensor");
  DG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && R > 0 && S > 0, "conv_wgrad: bad dims");
  DG_REQUIRE(Ci > 0 && (Ci & 3) == 0 && Co > 0 && (Co & 3) == 0, "conv_wgrad: Ci=%d, Co=%d must be multiples of 4", Ci, Co);
  DG_REQUIRE(up == 1 || up == 2, "conv_wgrad: up=%d unsupported", up);
  DG_REQUIRE(dr == 1 || dr == -1, "conv_wgrad: dr must be +-1");
  DG_REQUIRE(Kp % 32 == 0 && Kp >= R * S * Ci, "conv_wgrad: bad Kp=%d", Kp);
  DG_REQUIRE(splits >= 1, "conv_wgrad: splits=%d", splits);
  DG_REQUIRE(pro_mode >= 0 && pro_mode <= PRO_BOX_RELU, "conv_wgrad: bad pro_mode %d", pro_mode);
  DG_REQUIRE(!(pro_mode == PRO_BOX || pro_mode == PRO_BOX_RELU) || (up == 1 && dr == 1 && Hi >= 2 && Wi >= 2),
             "conv_wgrad: the box-sum loader (pro_mode 5 / 6) gathers forward geometries without up-sampling");
  DG_REQUIRE(!(pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE) || (pro_scale && pro_shift),
             "conv_wgrad: affine prologue needs scale/shift");
  DG_REQUIRE(slab_stride >= (int64_t)Co * Kp && (bias_off < 0 || bias_off + Co <= slab_stride),
             "conv_wgrad: slab_stride=%ld too small for Co*Kp=%ld (+bias)", (long)slab_stride, (long)Co * Kp);
  DG_REQUIRE((long)Co * Kp * 4 < (1L << 31), "conv_wgrad: weight tensor must be smaller than 2 GiB");
  DG_REQUIRE((long)B * Ho * Wo * Co * 4 < (1L << 31) && (long)B * Hi * Wi * Ci * 4 < (1L << 31),
             "conv_wgrad: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
  a.dy = dy; a.x = x; a.slab = slab; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.pro_mode = pro_mode;
  a.M = B * Ho * Wo;
  a.g = ConvGeom{B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, R * S * Ci, Kp};
  a.dWo = make_fastdiv((unsigned)Wo);
  a.dHo = make_fastdiv((unsigned)Ho);
  a.adv_b = 32 / (Ho * Wo);
  a.adv_y = (32 % (Ho * Wo)) / Wo;
  a.adv_x = (32 % (Ho * Wo)) % Wo;
  const int total_steps = cdiv(a.M, 32);
  DG_REQUIRE(segments >= 1 && splits % segments == 0, "conv_wgrad: splits=%d not a multiple of segments=%d", splits, segments);
  DG_REQUIRE(segments == 1 || (a.M % (32 * segments)) == 0, "conv_wgrad: %d pixels do not split into %d segments of whole K-steps", a.M, segments);
  a.seg_steps = cdiv(total_steps, segments);
  a.splits_per_seg = splits / segments;
  a.steps_per_split = cdiv(a.seg_steps, a.splits_per_seg);
  a.slab_stride = slab_stride;
  a.bias_off = bias_off;
  return DIAGAN_OK;
}

DIAGAN_API int diagan_conv_wgrad(const float* dy, const float* x, float* slab, int splits, int segments,
                                 int64_t slab_stride, int64_t bias_off, const float* pro_scale, const float* pro_shift, int pro_mode, int B,
                                 int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                 int dr, int off, int up, int Kp, void* stream) {
  WgradArgs a;
  const int rc = wgrad_fill_args(a, dy, x, slab, splits, segments, slab_stride, bias_off, pro_scale, pro_shift, pro_mode, B, Hi, Wi, Ci,
                                 Ho, Wo, Co, R, S, sy, dr, off, up, Kp);
  if (rc != DIAGAN_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (diagan_conv_wgrad_uses_wino(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp))
    return launch_wgrad_wino(a, splits, segments, st);
  int bn, bk;
  bool p2;
  wgrad_gemm_fields(a, &bn, &bk, &p2);
  // the large plain launches: the same tiles, splits and slabs on the bf16 matrix pipe with exactly split operands
  if (bn == 128 && bk == 128 && wgrad_x3_takes(a, diagan_conv_gemm_get_x3b() != 0)) return launch_wgrad_x3(a, splits, st);
  const dim3 grid(a.tiles * splits);
  // one straight-line kernel per prologue mode (x power-of-two image or not) for the two production tiles
#define DG_WG(BN_, PRO_) do { if (p2) hipLaunchKernelGGL((conv_wgrad_kernel<BN_, 128, PRO_, true>), grid, dim3(256), 0, st, a); \
                              else hipLaunchKernelGGL((conv_wgrad_kernel<BN_, 128, PRO_, false>), grid, dim3(256), 0, st, a); } while (0)
#define DG_WG_ALL(BN_) switch (pro_mode) { \
      case PRO_NONE: DG_WG(BN_, PRO_NONE); break; \
      case PRO_RELU: DG_WG(BN_, PRO_RELU); break; \
      case PRO_AFFINE_RELU: DG_WG(BN_, PRO_AFFINE_RELU); break; \
      case PRO_LRELU: DG_WG(BN_, PRO_LRELU); break; \
      case PRO_BOX: DG_WG(BN_, PRO_BOX); break; \
      case PRO_BOX_RELU: DG_WG(BN_, PRO_BOX_RELU); break; \
      default: DG_WG(BN_, PRO_AFFINE); break; }
  if (bn == 64 && bk == 64) {
    DG_REQUIRE(pro_mode <= PRO_AFFINE, "conv_wgrad: no box-sum loader on the 64 x 64 tile (Kp <= 64)");
    hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, dim3(256), 0, st, a);
  } else if (bn == 64) {
    DG_WG_ALL(64)
  } else {
    DG_WG_ALL(128)
  }
#undef DG_WG_ALL
#undef DG_WG
  return check_launch("conv_wgrad");
}

// see include/diagan_hip.h: the Winograd weight gradients of several layers (one prologue mode) in one launch
struct diagan_wgrad_job {    // mirrors the typedef of the same name in include/diagan_hip.h (128 bytes)
  const float* dy;
  const float* x;
  float* slab;
  const float* pro_scale;
  const float* pro_shift;
  int64_t slab_stride, bias_off;
  int32_t splits, segments, pro_mode, B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, pad_;
};
static_assert(sizeof(diagan_wgrad_job) == 128, "diagan_wgrad_job layout");
// Which launches may share a batched launch: equal class = the same kernel template.  0: this layer launches on its own.
//   100 + pro                         the Winograd F(3x3,2x2) weight gradient (conv_wgrad_wino.hip)
//   1000                              implicit GEMM, 64 x 64 tile (prologue mode at run time)
//   2000 + 100 (bn == 128) + 2 pro + p2   implicit GEMM, bn x 128 tile, prologue none / ReLU / box sums (the modes the networks'
//                                     1x1, strided and pooled layers use; the other modes are not instantiated for batches)
DIAGAN_API int diagan_conv_wgrad_batch_class(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off,
                                             int up, int Kp, int pro_mode) {
  if (diagan_conv_wgrad_uses_wino(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp)) return 100 + pro_mode;
  int bn, bk;
  wgrad_tile(Co, Kp, &bn, &bk);
  // (the 64 x 64 tile has no box-sum loader -- the one-layer launch rejects those modes on it -- so such a layer launches alone)
  if (bn == 64 && bk == 64) return pro_mode >= PRO_BOX ? 0 : 1000;
  if (pro_mode != PRO_NONE && pro_mode != PRO_RELU && pro_mode != PRO_BOX && pro_mode != PRO_BOX_RELU) return 0;
  const bool p2 = (Ho & (Ho - 1)) == 0 && (Wo & (Wo - 1)) == 0;
  return 2000 + (bn == 128 ? 100 : 0) + 2 * pro_mode + (p2 ? 1 : 0);
}

template <int BN_, int BK_, int PRO_, bool P2_>
static void launch_wgrad_gemm_batched(const WgradBatchArgs& b, int wgs, hipStream_t st) {
  hipLaunchKernelGGL((conv_wgrad_batched_kernel<BN_, BK_, PRO_, P2_>), dim3(wgs), dim3(256), 0, st, b);
}

DIAGAN_API int diagan_conv_wgrad_batched(const diagan_wgrad_job* jobs, int n, void* stream) {
  DG_REQUIRE(jobs && n >= 1 && n <= WG_BATCH_MAX, "conv_wgrad_batched: 1 .. %d jobs", WG_BATCH_MAX);
  WgradArgs a[WG_BATCH_MAX];
  int splits[WG_BATCH_MAX], segments[WG_BATCH_MAX], cls = -1;
  for (int j = 0; j < n; ++j) {
    const diagan_wgrad_job& q = jobs[j];
    const int c = diagan_conv_wgrad_batch_class(q.Hi, q.Wi, q.Ci, q.Ho, q.Wo, q.Co, q.R, q.S, q.sy, q.dr, q.off, q.up, q.Kp, q.pro_mode);
    DG_REQUIRE(c != 0 && (cls < 0 || c == cls), "conv_wgrad_batched: job %d has batch class %d, the launch's is %d "
               "(diagan_conv_wgrad_batch_class: equal non-zero classes share a launch)", j, c, cls);
    cls = c;
    const int rc = wgrad_fill_args(a[j], q.dy, q.x, q.slab, q.splits, q.segments, q.slab_stride, q.bias_off, q.pro_scale, q.pro_shift,
                                   q.pro_mode, q.B, q.Hi, q.Wi, q.Ci, q.Ho, q.Wo, q.Co, q.R, q.S, q.sy, q.dr, q.off, q.up, q.Kp);
    if (rc != DIAGAN_OK) return rc;
    splits[j] = q.splits;
    segments[j] = q.segments;
  }
  hipStream_t st = (hipStream_t)stream;
  if (cls < 1000) return launch_wgrad_wino_batched(a, splits, segments, n, st);
  WgradBatchArgs b;
  b.n = n;
  int wgs = 0, bn = 0, bk = 0;
  bool p2 = false;
  for (int j = 0; j < n; ++j) {
    wgrad_gemm_fields(a[j], &bn, &bk, &p2);
    b.a[j] = a[j];
    b.blk0[j] = wgs;
    b.cnt[j] = a[j].tiles * splits[j];
    wgs += (b.cnt[j] + 7) & ~7;
  }
  for (int j = n; j < WG_BATCH_MAX; ++j) b.blk0[j] = wgs, b.cnt[j] = 0;
  const int pro = jobs[0].pro_mode;
  if (cls == 1000) {
    launch_wgrad_gemm_batched<64, 64, -1, false>(b, wgs, st);
    return check_launch("conv_wgrad_batched");
  }
#define DG_WGB(BN_, PRO_) do { if (p2) launch_wgrad_gemm_batched<BN_, 128, PRO_, true>(b, wgs, st); \
                               else launch_wgrad_gemm_batched<BN_, 128, PRO_, false>(b, wgs, st); } while (0)
#define DG_WGB_ALL(BN_) switch (pro) { \
      case PRO_NONE: DG_WGB(BN_, PRO_NONE); break; \
      case PRO_RELU: DG_WGB(BN_, PRO_RELU); break; \
      case PRO_BOX: DG_WGB(BN_, PRO_BOX); break; \
      default: DG_WGB(BN_, PRO_BOX_RELU); break; }
  if (bn == 128) {
    DG_WGB_ALL(128)
  } else {
    DG_WGB_ALL(64)
  }
#undef DG_WGB_ALL
#undef DG_WGB
  return check_launch("conv_wgrad_batched");
}
DIAGAN_API int diagan_conv_wgrad_batch_max(void) { return WG_BATCH_MAX; }

// The Winograd F(3x3,2x2) weight gradient (conv_wgrad_wino.hip) takes the 3x3 / stride 1 / pad 1 layers unless
// DIAGAN_WINO=0 / diagan_conv_gemm_set_wino(0) is set.
DIAGAN_API int diagan_conv_wgrad_uses_wino(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                           int off, int up, int Kp) {
  static const int wino_env = getenv("DIAGAN_WINO") ? atoi(getenv("DIAGAN_WINO")) : 1;
  static const int wg_env = getenv("DIAGAN_WINO_WGRAD") ? atoi(getenv("DIAGAN_WINO_WGRAD")) : 1;
  const int sw = diagan_conv_gemm_get_wino();
  const int on = sw >= 0 ? sw : wino_env;
  return on && wg_env &&
         wgrad_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp);
}

// 1: diagan_conv_wgrad runs this launch on the split-operand kernel (conv_wgrad_x3.hip; for kernel-name bookkeeping)
DIAGAN_API int diagan_conv_wgrad_uses_x3(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off,
                                         int up, int Kp, int pro_mode, int64_t bias_off) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || diagan_conv_wgrad_uses_wino(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp)) return 0;
  int bn, bk;
  wgrad_tile(Co, Kp, &bn, &bk);
  if (bn != 128 || bk != 128) return 0;
  WgradArgs a;
  a.pro_mode = pro_mode; a.bias_off = bias_off; a.M = B * Ho * Wo;
  a.g = ConvGeom{B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, R * S * Ci, Kp};
  return wgrad_x3_takes(a, diagan_conv_gemm_get_x3b() != 0) ? 1 : 0;
}
// process-level diagnostic switch of that kernel: 0 off, 1 on, -1 back to the environment's DIAGAN_WGRAD_X3 (default on)
DIAGAN_API int diagan_conv_wgrad_set_x3(int on) {
  wgrad_x3_set(on);
  return DIAGAN_OK;
}

// split count for a full geometry: the Winograd kernel's own policy where it applies, else diagan_conv_wgrad_splits
DIAGAN_API int diagan_conv_wgrad_splits_geom(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                             int dr, int off, int up, int Kp) {
  if (diagan_conv_wgrad_uses_wino(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp))
    return wgrad_wino_splits(B, Ho, Wo, Ci, Co);
  return diagan_conv_wgrad_splits(B * Ho * Wo, Co, Kp);
}

// how many splits conv_wgrad should use for this problem (host-side heuristic, no device work)
DIAGAN_API int diagan_conv_wgrad_splits(int M, int Co, int Kp) {
  int bn, bk;
  wgrad_tile(Co, Kp, &bn, &bk);
  const int tiles = cdiv(Co, bn) * cdiv(Kp, bk);
  const int total_steps = cdiv(M, 32);
  // 512 resident workgroups (2 per CU).  Blocks run their whole K range, so the launch takes
  //   rounds(s) * (steps / s + c)   K-step times,   rounds(s) = ceil(tiles * s / 512),
  // c = fixed cost of a block (prologue, slab write-out) in K-steps.  One round of blocks is best whenever tiles
  // divides the 512 slots well (9, 18, 36, 72 tiles: 504 blocks); tile counts that do not (144 tiles: 3 splits fill
  // only 432 slots) are better served by a few FULL rounds -- 7 splits, 1008 blocks, 2 rounds: measured
  // 5.8 -> 5.0 ms at M=131072, Co=512, K=4608.  (A second, nearly empty round is what this model prices out.)
  static const int min_steps = getenv("DIAGAN_WGRAD_MINSTEPS") ? atoi(getenv("DIAGAN_WGRAD_MINSTEPS")) : 4;
  static const double fixed = getenv("DIAGAN_WGRAD_FIXED") ? atof(getenv("DIAGAN_WGRAD_FIXED")) : 6.0;
  int smax = total_steps / min_steps;               // at least min_steps K-steps per split
  if (smax > 256) smax = 256;
  int splits = 1;
  double best = 1e30;
  const int slots = 512;                            // two resident workgroups per CU
  for (int s = 1; s <= smax; ++s) {
    const double t = (double)cdiv(tiles * s, slots) * ((double)total_steps / s + fixed);
    if (t < best * 0.995) { best = t; splits = s; }   // ties (and near-ties) go to fewer splits: less slab traffic
  }
  if (splits < 1) splits = 1;
  if (splits > 256) splits = 256;
  return splits;
}

DIAGAN_API int diagan_wgrad_reduce(const float* slab, int splits, int64_t n_elem, float* out, int accumulate,
                                   const float* w, double* dot_partials, void* stream) {
  DG_REQUIRE(slab && out && splits >= 1 && n_elem > 0 && (n_elem & 3) == 0, "wgrad_reduce: bad args");
  const long n4 = n_elem / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, slab,
                     splits, n4, out, accumulate, w, dot_partials);
  return check_launch("wgrad_reduce");
}
