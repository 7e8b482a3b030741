// Winograd F(2x2, 3x3) forward / data-gradient convolution, staged-input kernel (tile_cfg 10).
//
// Same mathematics, operand images and epilogue as conv_wino.hip (tile_cfg 9; reference ops: F.conv2d and its input
// gradient in mimicry's GBlock / DBlock, diagan-pkg/diagan/models/predefined_models.py:19-21,38-40,57-59,76-78), built
// around what the ablation of that kernel showed (profiles/r02_wino_ablation.md): with every thread fetching its own
// 4x4 patch the input is requested 4x over in 32-byte pieces (a quarter of every cache line) -- the loads, not the
// MFMAs, set the K-step -- and one 512-thread workgroup per CU has nobody to hide its barriers and epilogue behind.
//
//   * One workgroup = 256 threads = 4 waves = a rectangular block of 32 tiles (BW x BH tiles of NI images, 8 x 4 x 1
//     for images of 16x16 and up) x 64 output channels; 71-79 KB of LDS, 2 workgroups per CU.
//   * The block's input region ((2 BH + 2) x (2 BW + 2) pixels per image: every pixel ONCE) goes global -> LDS by LDS-DMA,
//     16 channels (64 contiguous bytes) per pixel and request, double buffered; padding pixels are zeroed once and never
//     requested.  No load result ever sits in a register.
//   * K loop in steps of 4 input channels: 16 frequency planes V[f][32 tiles][4] and U[f][64 cols][4] per stage (24 KB,
//     two stages; U in the lane order of its fragments), fragments by ds_read_b64 / b128, wave w owns the four frequencies of row i = w on all 32 x 64 outputs
//     (4 x 2 accumulator tiles of v_mfma_f32_32x32x2_f32).  U arrives by LDS-DMA from the image wino_weight_kernel writes.
//   * The input transform of step s + 1 (LDS raw -> prologue -> B^T d B -> V planes) is done during step s by ONE wave pair
//     (waves 0-1 for even steps, 2-3 for odd ones: 32 tiles x 4 patch rows = 128 threads), in pieces between the MFMAs.
//   * epilogue as in conv_wino.hip: j-half of A^T . A in registers, the four rows meet in LDS (64 KB), two (tile, 4
//     channels) items per thread.
//
// Roofline: MFMA fp32; 16/36 of the direct convolution's multiply-accumulates.
#include "conv_common.h"
#include <type_traits>

namespace diagan {

typedef float f32x2 __attribute__((ext_vector_type(2)));

void launch_wino_weights(const float* w, float* ug, int Co, int Ci, int Kp, int flip, int staged, hipStream_t st);   // conv_wino.hip
long wino_ws_floats(int Co, int Ci);

constexpr int ZT = 32;                       // tiles per workgroup
constexpr int ZN = 64;                       // output channels per workgroup
constexpr int ZVP = ZT * 4;                  // floats of a V plane (32 tiles x 4 channels)
constexpr int ZUP = ZN * 4;                  // floats of a U plane (64 columns x 4 channels)
constexpr int ZSTAGE = 16 * (ZVP + ZUP);     // one stage: 24 KB
constexpr int ZRAWI = 256;                   // floats one raw LDS-DMA wave-instruction delivers (64 lanes x 16 bytes)

// blk = log2 BW | log2 BH << 4; npx = pixels of the block's input region (all images); nri = raw wave-instructions per buffer
template <int PRO>
__global__ __launch_bounds__(256, 2) void conv_wino_s_kernel(const ConvGemmArgs a, const float* __restrict__ ug, int blk,
                                                             int npx, int nri) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 stages][V 16 planes | U 16 planes] | raw [2][nri * 256]
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.Co + ZN - 1) / ZN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int nb = tile % tiles_n, blkid = tile / tiles_n, n0 = nb * ZN;
  const int lbw = blk & 15, lbh = (blk >> 4) & 15, lni = 5 - lbw - lbh;
  const int TW = g.Wo >> 1, TH = g.Ho >> 1;
  const int bxn = TW >> lbw, byn = TH >> lbh;                       // blocks along x / y of one image (exact)
  const int bxi = blkid % bxn, bq = blkid / bxn, byi = bq % byn, ig = bq / byn;
  const int b0 = ig << lni, ty0 = byi << lbh, tx0 = bxi << lbw;
  const int RW = (2 << lbw) + 2, RP = RW * ((2 << lbh) + 2);        // region width, pixels per image
  const bool affine = PRO == PRO_AFFINE_RELU || PRO == PRO_AFFINE;
  float* raw = smem + 2 * ZSTAGE;
  const int rawf = nri * ZRAWI;                                     // floats per raw buffer

  // K range in steps of 4 channels; split-K (gridDim.y) in multiples of 4 steps = one raw buffer fill
  const int nk = g.Ci >> 2;
  const int k_per = (((nk + a.ksplit - 1) / a.ksplit) + 3) & ~3;
  const int k_begin = blockIdx.y * k_per, k_end = min(k_begin + k_per, nk);
  const int nsup = (k_end - k_begin) >> 2;                          // raw buffer fills ("super-steps") of this workgroup
  const int R0 = k_begin >> 2;

  // ---- raw loader role: wave-instruction I = i * 4 + wave (i < 4) delivers region slots L = I * 64 + lane;
  //      slot L = (pixel L >> 2, 16-byte part L & 3 of its 16 channels) ----
  unsigned soff[4];
  bool sok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = (i * 4 + wave) * 64 + lane, P = L >> 2;
    const unsigned img = fdiv((unsigned)P, a.dHo);                   // dHo: divisor RP
    const unsigned rem = (unsigned)P - img * RP;
    const unsigned row = fdiv(rem, a.dWo);                           // dWo: divisor RW
    const int col = (int)(rem - row * RW);
    const int iy = 2 * ty0 - 1 + (int)row, ix = 2 * tx0 - 1 + col;
    sok[i] = (i * 4 + wave) < nri && P < npx && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
    soff[i] = sok[i] ? (unsigned)((((b0 + (int)img) * g.Hi + iy) * g.Wi + ix) * g.Ci + (L & 3) * 4) : 0u;   // floats
    if ((i * 4 + wave) < nri && !sok[i]) {                          // padding: zero once, never requested
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(raw + L * 4) = z;
      *reinterpret_cast<f32x4*>(raw + rawf + L * 4) = z;
    }
  }
  // ---- transform role: thread (tile lt, patch row lr) of wave pair lp; the 4 lanes of a quad hold the 4 rows of one patch ----
  const int lr = tid & 3, lt = (tid >> 2) & 31, lp = wave >> 1;
  const int timg = lt >> (lbw + lbh), tby = (lt >> lbw) & ((1 << lbh) - 1), tbx = lt & ((1 << lbw) - 1);
  const int rpix = (timg * RP + (2 * tby + lr) * RW + 2 * tbx) * 16;   // float offset of this row's first pixel in a raw buffer
  float keep[4];
  {
    const int iy = 2 * (ty0 + tby) - 1 + lr, ix0 = 2 * (tx0 + tbx) - 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) keep[c] = (iy >= 0 && iy < g.Hi && ix0 + c >= 0 && ix0 + c < g.Wi) ? 1.f : 0.f;
  }
  const int pro_group_off = a.pro_group_rows > 0 ? ((b0 * g.Ho * g.Wo) / a.pro_group_rows) * g.Ci : 0;
  // column transform of this lane's row: V[r] = t[r] + sc * t[partner], partner by quad_perm [2,2,1,1]
  // (r = 0: t0 - t2; 1: t1 + t2; 2: t2 - t1; 3: t3 - t1 = -(B^T row 3), compensated in U)
  const float sc = lr == 1 ? 1.f : -1.f;
  const int vslot = (lt ^ (lr << 1)) * 4;                           // 8 lanes of a ds_write_b128 group: 8 distinct 16-byte slots

  const float* ublock = ug + (long)nb * (g.Ci >> 3) * (32 * ZUP);

  auto issue_u = [&](int s_abs, int stage) {                        // U of 4-channel step s_abs: planes (f, k-quad s_abs & 1)
    float* us = smem + stage * ZSTAGE + 16 * ZVP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = wave * 4 + i;
      const unsigned long long ub = (unsigned long long)(ublock + ((long)(s_abs >> 1) * 32 + f * 2 + (s_abs & 1)) * ZUP);
      const unsigned long long us64 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ub) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ub >> 32)) << 32;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(us64) + (unsigned)(lane * 4)),
          (__attribute__((address_space(3))) void*)(us + f * ZUP), 16, 0, 0);
    }
  };
  auto issue_raw = [&](int R_abs, int i) {                          // wave-instruction i of raw buffer fill R_abs
    float* dst = raw + ((R_abs - R0) & 1) * rawf + (i * 4 + wave) * ZRAWI;
    if (sok[i])
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.x + soff[i] + (long)R_abs * 16),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  f32x4 d[4];
  float t[4][4];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  int kbound[4];                                        // BatchNorm + ReLU: upper clamp of the activation, 0 on padding pixels
#pragma unroll
  for (int c = 0; c < 4; ++c) kbound[c] = keep[c] != 0.f ? 0x7fffffff : 0;
  auto tr_read = [&](int s_abs) {                                   // this row's 4 pixels x 4 channels of step s_abs
    const float* src = raw + (((s_abs >> 2) - R0) & 1) * rawf + rpix + (s_abs & 3) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) d[c] = *reinterpret_cast<const f32x4*>(src + c * 16);
    if (affine) {
      psc = *reinterpret_cast<const f32x4*>(a.pro_scale + pro_group_off + s_abs * 4);
      psh = *reinterpret_cast<const f32x4*>(a.pro_shift + pro_group_off + s_abs * 4);
    }
  };
  // (vector instructions are not hidden behind this wave's MFMAs -- profiles/r02_wino_ablation.md -- so every piece below
  //  is written for the fewest of them, as in conv_wino.hip)
  auto tr_prologue = [&]() {
    if (PRO == PRO_NONE) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 v = d[c];
      if (affine) v = v * psc + psh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float q = v[e];
        float r;
        if (PRO == PRO_LRELU) {
          const float q2 = 0.2f * q;
          asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(q), "v"(q2));
        } else if (PRO == PRO_RELU) {
          r = __int_as_float(max(__float_as_int(q), 0));
        } else if (PRO == PRO_AFFINE_RELU) {
          asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(q), "v"(kbound[c]));      // ReLU + zero padding in one clamp
        } else {
          r = q * keep[c];                               // padding is zero AFTER the transform
        }
        v[e] = r;
      }
      d[c] = v;
    }
  };
  auto tr_rows = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[0][e] = d[0][e] - d[2][e];
      t[1][e] = d[1][e] + d[2][e];
      t[2][e] = d[2][e] - d[1][e];
      t[3][e] = d[1][e] - d[3][e];
    }
  };
  float* const vst0 = smem + lr * 4 * ZVP + vslot;
  auto tr_store = [&](int stage, int j) {
    float o0 = t[j][0], o1 = t[j][1], o2 = t[j][2], o3 = t[j][3];
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
        : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3)
        : "v"(sc));
    const f32x4 o = {o0, o1, o2, o3};
    *reinterpret_cast<f32x4*>(vst0 + stage * ZSTAGE + j * ZVP) = o;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int fl = 0; fl < 4; ++fl)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fl][h][e] = 0.f;
  const int fi = lane & 31, fh = lane >> 5;
  const int sw = wave << 1;                                         // slot swizzle of this wave's V planes (row i = wave)

  // ---- first raw buffer, U of the first step, V of the first step ----
  if (nsup > 0) {
    issue_u(k_begin, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_raw(R0, i);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nsup > 0 && lp == 0) {
    tr_read(k_begin);
    tr_prologue();
    tr_rows();
#pragma unroll
    for (int j = 0; j < 4; ++j) tr_store(0, j);
  }
  __syncthreads();

  // One step (4 channels, 8 groups of two MFMAs) of super-step m.  J = step within the super-step (stage J & 1); the wave
  // pair (J + 1) & 1 builds V of the next step in pieces: raw reads at group 0, prologue 2, row transform 3, column
  // transform + store 4..7.  Every wave sends 4 planes of the next U and (J < 3) one raw instruction of the next buffer.
  auto step = [&](int m, auto jc, auto lastc) {
    constexpr int J = decltype(jc)::value;
    constexpr bool LAST = decltype(lastc)::value;
    constexpr bool NEXT = !(LAST && J == 3);
    const int s_abs = k_begin + m * 4 + J;
    const float* vs = smem + (J & 1) * ZSTAGE;
    const float* us = vs + 16 * ZVP;
    if (NEXT) issue_u(s_abs + 1, (J + 1) & 1);
    if (!LAST && J < 3) issue_raw(R0 + m + 1, J);
    if (!LAST && J == 2) issue_raw(R0 + m + 1, 3);      // regions of more than 192 pixels
    __builtin_amdgcn_sched_barrier(0);
    f32x2 fa[4];
    f32x4 fb[4];                                          // [h * 2 + e]: both column halves of this lane's two k values (U image order)
#pragma unroll
    for (int fl = 0; fl < 4; ++fl) {
      const int f = wave * 4 + fl;
      fa[fl] = *reinterpret_cast<const f32x2*>(vs + f * ZVP + ((fi ^ sw) << 2) + fh * 2);
      fb[fl] = *reinterpret_cast<const f32x4*>(us + f * ZUP + ((fh * 32 + fi) << 2));
    }
    const bool tr = NEXT && lp == ((J + 1) & 1);        // wave-uniform: this wave pair builds V of the next step
#pragma unroll
    for (int fl = 0; fl < 4; ++fl)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int grp = fl * 2 + e;
        if (NEXT && grp != 1) {
          __builtin_amdgcn_sched_barrier(0);
          if (tr) {
            if (grp == 0) tr_read(s_abs + 1);
            else if (grp == 2) tr_prologue();
            else if (grp == 3) tr_rows();
            else tr_store((J + 1) & 1, grp - 4);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
          acc[fl][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fl][e], fb[fl][h * 2 + e], acc[fl][h], 0, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA pieces (next U, next raw buffer) have landed
    __syncthreads();
  };
  auto super = [&](int m, auto lastc) {
    step(m, std::integral_constant<int, 0>{}, lastc);
    step(m, std::integral_constant<int, 1>{}, lastc);
    step(m, std::integral_constant<int, 2>{}, lastc);
    step(m, std::integral_constant<int, 3>{}, lastc);
  };
  for (int m = 0; m + 1 < nsup; ++m) super(m, std::false_type{});
  if (nsup > 0) super(nsup - 1, std::true_type{});

  // ---- epilogue ----
  // s[i][b] = sum_j A^T[b][j] M[i][j] in registers (b = 0: M0 + M1 + M2; b = 1: M1 - M2 - M3), then the four rows i meet
  // in LDS ([i][b][32 tiles][64 channels] = 64 KB) and every thread finishes two (tile, 4 channels) items:
  // Y[a][b] = sum_i A^T[a][i] s[i][b].  (Row 3 of V and of U are both staged negated: M is what it always was.)
  const float sc0 = a.scale0 ? a.scale0[0] : a.out_scale, sc1 = a.scale1 ? a.scale1[0] : a.out_scale;
  const int split = a.scale0 ? a.scale_split : 0x7fffffff;            // pixel-row index where the second sigma starts
  const bool rawout = a.ksplit > 1;                                    // split-K: un-scaled partial sums to the slab
  const bool hr = !rawout && a.residual != nullptr, hm = !rawout && a.mask_src != nullptr, hs = !rawout && a.stat_partials != nullptr;
  float* ydst = rawout ? a.slab + (long)blockIdx.y * a.M * g.Co : a.y;
  const float rfloor = a.res_relu ? 0.f : -__builtin_huge_valf();
  const int et = tid >> 4, ec = (tid & 15) * 4;                       // this thread's tile (within a half) and channel quad
  const int n = n0 + ec;
  const bool col_ok = n < g.Co;                                        // Co % 4 == 0
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (!rawout && a.bias && col_ok) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
  f32x4 cs1 = {0.f, 0.f, 0.f, 0.f}, cs2 = {0.f, 0.f, 0.f, 0.f};
  float* ss = smem;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int trow = (e & 3) + 8 * (e >> 2) + 4 * fh;
      const float m0 = acc[0][h][e], m1 = acc[1][h][e], m2 = acc[2][h][e], m3 = acc[3][h][e];
      ss[((wave * 2 + 0) * ZT + trow) * ZN + h * 32 + fi] = m0 + m1 + m2;
      ss[((wave * 2 + 1) * ZT + trow) * ZN + h * 32 + fi] = m1 - m2 - m3;
    }
  // this thread's two tiles and their 4 output pixels each; residual / mask loads are issued BEFORE the barrier
  long o4[2][4];
  int prow4[2][4];
  f32x4 rres[2][4], rmsk[2][4];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int tl = it * 16 + et;
    const int img = tl >> (lbw + lbh), by = (tl >> lbw) & ((1 << lbh) - 1), bx = tl & ((1 << lbw) - 1);
    const int b = b0 + img, ty = ty0 + by, tx = tx0 + bx;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      prow4[it][p] = (b * g.Ho + 2 * ty + (p >> 1)) * g.Wo + 2 * tx + (p & 1);       // pixel (GEMM row) index
      o4[it][p] = (long)prow4[it][p] * g.Co + n;
    }
    if (hr && col_ok) {
#pragma unroll
      for (int p = 0; p < 4; ++p) rres[it][p] = *reinterpret_cast<const f32x4*>(a.residual + o4[it][p]);
    }
    if (hm && col_ok) {
#pragma unroll
      for (int p = 0; p < 4; ++p) rmsk[it][p] = *reinterpret_cast<const f32x4*>(a.mask_src + o4[it][p]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    f32x4 y4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 sa = *reinterpret_cast<const f32x4*>(ss + ((i * 2 + 0) * ZT + it * 16 + et) * ZN + ec);
      const f32x4 sb = *reinterpret_cast<const f32x4*>(ss + ((i * 2 + 1) * ZT + it * 16 + et) * ZN + ec);
      if (i < 3) { y4[0] += sa; y4[1] += sb; }
      if (i == 1) { y4[2] += sa; y4[3] += sb; }
      if (i >= 2) { y4[2] -= sa; y4[3] -= sb; }
    }
    if (col_ok) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        f32x4 y = rawout ? y4[p] : y4[p] * (prow4[it][p] < split ? sc0 : sc1) + bv;
        if (hr) {
          f32x4 r = rres[it][p];
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], rfloor);
          y += r;
        }
        if (hm) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = rmsk[it][p][e] > 0.f ? y[e] : y[e] * a.mask_slope;
        }
        *reinterpret_cast<f32x4*>(ydst + o4[it][p]) = y;
        if (hs) {
          cs1 += y;
          cs2 += y * y;
        }
      }
    }
  }
  if (hs) {
    // column sums over the workgroup's 128 pixels: lanes with equal (tid & 15) hold the same channels
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs1[e] += __shfl_xor(cs1[e], 16, 64);
      cs2[e] += __shfl_xor(cs2[e], 16, 64);
      cs1[e] += __shfl_xor(cs1[e], 32, 64);
      cs2[e] += __shfl_xor(cs2[e], 32, 64);
    }
    float* red = smem;                                                 // [4 waves][2][64]
    if (lane < 16) {
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * 64 + ec) = cs1;
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * 64 + ec) = cs2;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, col = tid & 63;
      float tsum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) tsum += red[(w * 2 + which) * 64 + col];
      if (n0 + col < g.Co) a.stat_partials[(long)blkid * 2 * g.Co + which * g.Co + n0 + col] = tsum;
    }
  }
}

// Block shape of the staged kernel for a geometry: BW x BH tiles of NI images (BW * BH * NI = 32) that tile the batch
// exactly and whose input region fits the raw buffers; 0 when there is none (the caller falls back to conv_wino.hip).
// Returns log2 BW | log2 BH << 4 | 0x100.
int wino_s_block(int B, int Ho, int Wo, int Ci, int pro_group_rows) {
  if (Ci % 16 != 0 || (Ho & 1) || (Wo & 1)) return 0;
  const int TW = Wo >> 1, TH = Ho >> 1;
  int lbw = 0;
  while (lbw < 3 && TW % (2 << lbw) == 0) ++lbw;
  int lbh = 0;
  while (lbw + lbh < 5 && TH % (2 << lbh) == 0) ++lbh;
  const int ni = 32 >> (lbw + lbh);
  if (B % ni != 0) return 0;
  const int npx = ni * ((2 << lbw) + 2) * ((2 << lbh) + 2);
  if (npx > 240) return 0;                                // 2 x 15 KB of raw buffers: 78 KB per workgroup, two per CU
  if (pro_group_rows > 0 && pro_group_rows % (ni * Ho * Wo) != 0) return 0;
  return lbw | lbh << 4 | 0x100;
}

// Split-K factor as wino_ksplit of conv_wino.hip, for 32-tile workgroups that run two to a CU
int wino_s_ksplit(int B, int Ho, int Wo, int Ci, int Co, int allow_split, long ws_floats, int min_wgs) {
  const long wgs = (long)(B * (Ho >> 1) * (Wo >> 1) / ZT) * cdiv(Co, ZN);
  if (wgs >= min_wgs) return 1;
  if (!allow_split) return 0;
  for (int ks = 2; ks <= 4; ++ks) {
    if (Ci / 8 / ks < 16) break;
    if (wgs * ks >= min_wgs && wino_ws_floats(Co, Ci) + (long)ks * B * Ho * Wo * Co <= ws_floats) return ks;
  }
  return 0;
}

template <int PRO>
static int launch_wino_s_pro(const ConvGemmArgs& a, const float* ug, int blk, int npx, hipStream_t st) {
  const int MT = a.g.B * (a.g.Ho >> 1) * (a.g.Wo >> 1);
  const int wgs = (MT / ZT) * cdiv(a.g.Co, ZN);
  const int nri = cdiv(npx * 4, 64);
  size_t lds = (size_t)(2 * ZSTAGE + 2 * nri * ZRAWI) * sizeof(float);
  if (lds < (size_t)8 * ZT * ZN * sizeof(float)) lds = (size_t)8 * ZT * ZN * sizeof(float);      // the epilogue's exchange
  auto kern = conv_wino_s_kernel<PRO>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(wgs, a.ksplit), dim3(256), lds, st, a, ug, blk & 0xff, npx, nri);
  return check_launch("conv_wino_s");
}

// `a` as prepared by diagan_conv_gemm; ws: wino_ws_floats(Co, Ci) floats for the transformed weights
int launch_wino_s(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  const int blk = wino_s_block(g.B, g.Ho, g.Wo, g.Ci, a.pro_group_rows);
  if (!blk) return set_err(DIAGAN_EINVAL, "conv_gemm: tile_cfg 10 (staged Winograd) needs Ci %% 16 == 0 and a batch that its 32-tile blocks tile exactly");
  const int lbw = blk & 15, lbh = (blk >> 4) & 15;
  const int RW = (2 << lbw) + 2, RH = (2 << lbh) + 2, ni = 32 >> (lbw + lbh);
  a.dWo = make_fastdiv((unsigned)RW);
  a.dHo = make_fastdiv((unsigned)(RW * RH));
  launch_wino_weights(a.w, ws, g.Co, g.Ci, g.Kp, g.dr < 0 ? 1 : 0, 1, st);
  const int npx = ni * RW * RH;
  switch (a.pro_mode) {
    case PRO_NONE: return launch_wino_s_pro<PRO_NONE>(a, ws, blk, npx, st);
    case PRO_RELU: return launch_wino_s_pro<PRO_RELU>(a, ws, blk, npx, st);
    case PRO_AFFINE_RELU: return launch_wino_s_pro<PRO_AFFINE_RELU>(a, ws, blk, npx, st);
    case PRO_LRELU: return launch_wino_s_pro<PRO_LRELU>(a, ws, blk, npx, st);
    default: return launch_wino_s_pro<PRO_AFFINE>(a, ws, blk, npx, st);
  }
}

}  // namespace diagan
