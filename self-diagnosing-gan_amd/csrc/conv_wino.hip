// Winograd F(2x2, 3x3) forward / data-gradient convolution on the fp32 matrix cores.
//
// Replaces the same reference ops as conv_gemm.hip (F.conv2d and its input gradient in mimicry's GBlock / DBlock,
// selected at diagan-pkg/diagan/models/predefined_models.py:19-21,38-40,57-59,76-78) for the 3x3 / stride 1 / pad 1
// layers, which carry ~85 % of the SNGAN FLOP:  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A  per 2x2 output tile --
// 16 multiply-accumulates per tile, output and input channel instead of 36, i.e. 2.25x fewer MFMA cycles than the
// implicit GEMM for the same result (fp32 throughout; transform coefficients 0, +-1, 1/2, 1/4).
//
// One workgroup = 512 threads = 8 waves = 64 tiles (256 output pixels) x 64 output channels, one per CU (128 KB of LDS).
//   * K loop over input channels in steps of 8.  Per step the 16 "frequency" GEMMs  M_f[64 x 64] += V_f[64 x 8] U_f[64 x 8]^T
//     run as v_mfma_f32_32x32x2_f32: wave w owns the four frequencies of row i = w >> 1 on column half w & 1 (4 x 2 accumulator
//     tiles = 128 registers).
//   * U (transformed weights) is produced once per launch by wino_weight_kernel in exactly the LDS image order
//     [64-col block][K-step][f][k-quad][64 cols][4], so a K-step's 32 KB go global -> LDS by LDS-DMA, no registers.
//   * V (transformed input): thread (tile, channel quad, patch row r) loads its 4 pixels x 4 channels (prologue applied
//     here), transforms along the row, exchanges with its quad by DPP for the column transform and writes its 4
//     frequencies (r, j) to LDS.  Planes are [f][k-quad][tile][4 channels]; the tile slot is XOR-ed with (q | r << 1) so the
//     ds_write_b128 of a quad's 8 lanes hit 8 different bank groups; fragment reads stay conflict-free.
//   * epilogue: a wave applies the j half of A^T . A to its row in registers; the four rows of a (tile, channel) then meet
//     in LDS once (128 KB): every thread owns two (tile, 4 channels) items, finishes the transform and the usual epilogue
//     (per-half 1/sigma, bias, residual, ReLU-backward mask, BatchNorm statistics) and stores 16-byte pixels.
//
// Roofline: MFMA fp32; the kernel executes 16/36 of the direct convolution's multiply-accumulates.
#include "conv_common.h"
#include <type_traits>
#include "wino_weights.h"

namespace diagan {

// U[f][co][ci] = (G g G^T)[i][j], f = 4 i + j, written in the LDS image order (see above).  flip: the data-gradient of a
// stride-1 convolution is the correlation with the taps reversed.
__global__ __launch_bounds__(512) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ ug, int Co, int Ci,
                                                          int Kp, int flip) {
  __shared__ f32x4 sg[WT_LDS_F4];
  wino_weight_body(w, ug, Co, Ci, Kp, flip, blockIdx.x, blockIdx.y, sg);
}

constexpr int WT = 64;                 // tiles per workgroup
constexpr int WN = 64;                 // output channels per workgroup
constexpr int WK = 8;                  // input channels per K-step
constexpr int W_PLANE = 64 * 4;        // floats of one (f, k-quad) plane: 64 rows x 4 channels
constexpr int W_STAGE = 2 * 32 * W_PLANE;   // V planes + U planes of one stage (floats)

// Diagnostic build only (make EXTRA=-DDIAGAN_WINO_ABLATE; tools/wino_ablate.py, tools/wino_stamps.py): ConvGemmArgs::tune
// bits switch parts of the K loop off (bit 4 transform, 5 input loads, 6 weight DMA, 7 barrier, 8 MFMAs, 9 epilogue: the
// results are then garbage) so that their cost can be read off the launch time, and bit 10 makes every wave sum the
// s_memtime cycles of the phases of its K-steps into the stamp buffer.  Findings: profiles/r02_wino_ablation.md.
#ifdef DIAGAN_WINO_ABLATE
#define WINO_ON(bit) (!(a.tune & (bit)))
#else
#define WINO_ON(bit) true
#endif

template <int PRO>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const ConvGemmArgs a, const float* __restrict__ ug) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 stages][V 32 planes | U 32 planes] = 128 KB
#ifdef DIAGAN_WINO_ABLATE
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (g.Co + WN - 1) / WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int t0 = (tile / tiles_n) * WT, nb = tile % tiles_n, n0 = nb * WN;
  const int TW = g.Wo >> 1, TH = g.Ho >> 1;
  const int MT = g.B * TH * TW;                         // 2x2 output tiles in all
  const int nk = g.Ci / WK;
  // K-step range of this workgroup (split-K over gridDim.y for launches with few output tiles: raw partial outputs go
  // to a.slab and splitk_epilogue_kernel of conv_gemm.hip applies the epilogue after a fixed-order sum)
  const int k_per = (nk + a.ksplit - 1) / a.ksplit;
  const int k_begin = blockIdx.y * k_per, k_end = min(k_begin + k_per, nk);
  const bool affine = PRO == PRO_AFFINE_RELU || PRO == PRO_AFFINE;

  // ---- loader role: (tile lt, channel quad q, patch row r); the 4 lanes of a quad hold the 4 rows of one patch ----
  const int lr = tid & 3, lq = (tid >> 2) & 1, lt = tid >> 3;
  unsigned off[4], inv[4];                              // byte offsets of this row's 4 patch pixels; inv: bit 31 if outside
  float keep[4];
  {
    const int gt = t0 + lt;
    const bool tv = gt < MT;
    const unsigned q1 = fdiv((unsigned)(tv ? gt : 0), a.dWo);          // dWo: divisor TW
    const int tx = (tv ? gt : 0) - (int)q1 * TW;
    const unsigned b = fdiv(q1, a.dHo);                                // dHo: divisor TH
    const int ty = (int)q1 - (int)b * TH;
    const int iy = 2 * ty - 1 + lr, ix0 = 2 * tx - 1;
    const bool rv = tv && iy >= 0 && iy < g.Hi;
    const int rowbase = (((int)b * g.Hi + iy) * g.Wi + ix0) * g.Ci * 4 + lq * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool ok = rv && ix0 + c >= 0 && ix0 + c < g.Wi;
      off[c] = ok ? (unsigned)(rowbase + c * g.Ci * 4) : 0u;
      inv[c] = ok ? 0u : 0x80000000u;                   // beyond num_records (< 2 GiB): the hardware returns zeros
      keep[c] = ok ? 1.f : 0.f;
    }
  }
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Ci * 4u), 0x00020000);
  const int pro_group_off = a.pro_group_rows > 0 ? ((t0 * 4) / a.pro_group_rows) * g.Ci : 0;
  // column transform of this lane's row: V[r] = t[r] + sc * t[partner], partner by quad_perm [2,2,1,1]
  // (r = 0: t0 - t2; 1: t1 + t2; 2: t2 - t1; 3: t3 - t1 = -(B^T row 3), compensated in U)
  const float sc = lr == 1 ? 1.f : -1.f;
  // LDS slot of this thread's 4 output planes: plane p = ((r * 4 + j) * 2 + q), slot = tile ^ (q | r << 1)
  const int vslot = (lt ^ (lq | (lr << 1))) * 4;
  float* const vst0 = smem + (lr * 8 + lq) * W_PLANE + vslot;

  const float* ublock = ug + (long)nb * nk * (32 * W_PLANE);

  f32x4 ra[4], psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  auto issue_loads = [&](int kk, int stage) {
    // U: 32 planes of 1 KB, 4 per wave, straight into LDS (lane-linear image == the global order)
    float* us = smem + stage * W_STAGE + 32 * W_PLANE;
    if (WINO_ON(64)) {
      // ONE wave-uniform base (scalar registers) + the lane's 32-bit offset for the wave's four planes: the instruction's
      // immediate offset (applied to the global AND the LDS address) steps through them, 1 KB apart on both sides
      const unsigned long long ub = (unsigned long long)(ublock + (long)kk * (32 * W_PLANE) + wave * 4 * W_PLANE);
      const unsigned long long us64 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ub) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ub >> 32)) << 32;
      const float* up = reinterpret_cast<const float*>(us64) + (unsigned)(lane * 4);
      float* ul = us + wave * 4 * W_PLANE;
#define WINO_DMA(I)                                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)up,                \
                                   (__attribute__((address_space(3))) void*)ul, 16, (I) * W_PLANE * 4, 0)
      WINO_DMA(0);
      WINO_DMA(1);
      WINO_DMA(2);
      WINO_DMA(3);
#undef WINO_DMA
    }
    if (WINO_ON(32))
#pragma unroll
    for (int c = 0; c < 4; ++c)
      ra[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off[c] | inv[c], kk * (WK * 4), 0));   // step offset: scalar
    if (affine) {
      psc = *reinterpret_cast<const f32x4*>(a.pro_scale + pro_group_off + kk * WK + lq * 4);
      psh = *reinterpret_cast<const f32x4*>(a.pro_shift + pro_group_off + kk * WK + lq * 4);
    }
  };
  f32x4 d[4];
  float t[4][4];                                        // row-transformed patch row: [column j][channel]
  // (vector instructions are NOT hidden behind this wave's MFMAs -- tools/micro/mfma_valu.hip: every one costs its ~4 issue
  //  cycles of the SIMD -- so each prologue is written for the fewest of them)
  int kbound[4];                                        // BatchNorm + ReLU: upper clamp of the activation, 0 on padding pixels
#pragma unroll
  for (int c = 0; c < 4; ++c) kbound[c] = keep[c] != 0.f ? 0x7fffffff : 0;
  auto transform_prologue = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 v = ra[c];
      if (PRO != PRO_NONE) {
        if (affine) v = v * psc + psh;
        if (PRO == PRO_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {               // max(x, 0.2 x) without fmaxf's canonicalising extra v_max
            const float q = v[e], q2 = 0.2f * q;
            float r;
            asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(q), "v"(q2));
            v[e] = r;
          }
        } else if (PRO == PRO_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {               // max(v, 0) as ONE v_max_i32 on the bits (fmaxf costs a canonicalising
            const float q = v[e];                     // v_max first; negative floats are negative integers)
            v[e] = __int_as_float(max(__float_as_int(q), 0));
          }
        } else if (PRO == PRO_AFFINE_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {               // ReLU and "padding is zero AFTER the transform" as ONE v_med3_i32:
            const float q = v[e];                     // clamp of the bits to [0, kbound] (positive floats order like integers)
            float r;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(q), "v"(kbound[c]));
            v[e] = r;
          }
        } else {
          v *= keep[c];                                // padding is zero AFTER the transform
        }
      }
      d[c] = v;
    }
  };
  auto transform_rows = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[0][e] = d[0][e] - d[2][e];
      t[1][e] = d[1][e] + d[2][e];
      t[2][e] = d[2][e] - d[1][e];
      t[3][e] = d[1][e] - d[3][e];
    }
  };
  auto transform_store = [&](int stage, int j) {
    float* vs = vst0 + stage * W_STAGE;                // plane p = ((r * 4 + j) * 2 + q) of the stage, this thread's slot
    // V[r][j] = t[r][j] + sc * t[partner][j] as ONE v_fmac_f32 whose first factor comes through DPP (quad_perm [2,2,1,1]):
    // every lane reads its partner's OLD value before any lane writes (the builtin + fma pair costs the DPP move and the
    // fma, and vector instructions do not hide behind this wave's MFMAs)
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
    *reinterpret_cast<f32x4*>(vs + j * 2 * W_PLANE) = o;
  };

  // wave w owns the four frequencies of row i = w >> 1 (f = 4 i + j) on column block tn = w & 1: 4 x 2 accumulator
  // tiles of 32 tiles x 32 channels.  Keeping a whole row in one wave lets it apply the j-half of the output transform
  // in registers before the frequencies meet in LDS (half the exchange traffic).
  const int wi = wave >> 1, tn = wave & 1;
  f32x16 acc[4][2];
#pragma unroll
  for (int fl = 0; fl < 4; ++fl)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fl][i][e] = 0.f;

  const int fi = lane & 31, fh = lane >> 5;
  const int sw = fh | (wi << 1);                          // slot swizzle of this wave's V planes (q | r << 1)
  if (k_begin < k_end) {
    issue_loads(k_begin, 0);
    transform_prologue();
    transform_rows();
#pragma unroll
    for (int j = 0; j < 4; ++j) transform_store(0, j);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef DIAGAN_WINO_ABLATE
  const bool stamping = a.stamps && (a.tune & 1024);
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = t_entry;
  auto tick = [&](int i) {
    if (stamping) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[i] += now - tlast;
      tlast = now;
    }
  };
  tick(6);                   // set-up + first stage
  if (a.stamps && !stamping) {
#else
  if (a.stamps) {            // diagnostic (tools/wino_debug.py): stage 0 of workgroup 0 as it sits in LDS, then stop
#endif
    if (blockIdx.x == 0)
      for (int i = tid; i < W_STAGE; i += 512) reinterpret_cast<float*>(a.stamps)[i] = smem[i];
    return;
  }

  // One K-step: 16 groups of two MFMAs (frequency fl, k pair e).  The next stage's LDS-DMA pieces and global loads are
  // issued at the top; its input transform is cut into six pieces (prologue, row transform, four column-transform +
  // store pieces) that ride in the shadow of groups 8..13, one per group, as in conv_gemm.hip.
  // (fragment addresses = one base per operand + the stage: address arithmetic inside the loop is not free)
  const float* const fa_base = smem + (wi * 8 + fh) * W_PLANE + ((fi ^ sw) << 2);
  const float* const fb_base = smem + (32 + wi * 8 + fh) * W_PLANE + ((tn * 32 + fi) << 2);
  auto kstep = [&](int kk, auto has_next) {
    const int cur = (kk - k_begin) & 1;
    if (decltype(has_next)::value) issue_loads(kk + 1, cur ^ 1);
    f32x4 fa[4][2], fb[4];
#pragma unroll
    for (int fl = 0; fl < 4; ++fl) {
#pragma unroll
      for (int i = 0; i < 2; ++i)          // ((i * 32 + fi) ^ sw) == i * 32 + (fi ^ sw): sw < 8
        fa[fl][i] = *reinterpret_cast<const f32x4*>(fa_base + cur * W_STAGE + fl * 2 * W_PLANE + i * 128);
      fb[fl] = *reinterpret_cast<const f32x4*>(fb_base + cur * W_STAGE + fl * 2 * W_PLANE);
    }
#ifdef DIAGAN_WINO_ABLATE
    if (stamping) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tick(0);               // DMA / load issue + fragment reads
    }
#endif
#pragma unroll
    for (int fl = 0; fl < 4; ++fl)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int grp = fl * 4 + e;
        if (decltype(has_next)::value && grp >= 8 && grp < 14 && WINO_ON(16)) {
          __builtin_amdgcn_sched_barrier(0);             // keep the piece HERE (its loads have had >= 1000 cycles to land)
#ifdef DIAGAN_WINO_ABLATE
          if (grp == 8 && stamping) {
            tick(1);           // first 16 MFMAs issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tick(2);           // wait for the input loads
          }
#endif
          if (grp == 8) transform_prologue();
          else if (grp == 9) transform_rows();
          else transform_store(cur ^ 1, grp - 10);
        }
        if (WINO_ON(256))
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[fl][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fl][i][e], fb[fl][e], acc[fl][i], 0, 0, 0);
      }
#ifdef DIAGAN_WINO_ABLATE
    tick(3);                   // second 16 MFMAs + transform
#endif
    if (WINO_ON(128)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's LDS-DMA pieces of the next stage have landed
#ifdef DIAGAN_WINO_ABLATE
      tick(4);                 // wait for the weight DMA
#endif
      __syncthreads();
#ifdef DIAGAN_WINO_ABLATE
      tick(5);                 // barrier
#endif
    }
  };
  for (int kk = k_begin; kk + 1 < k_end; ++kk) kstep(kk, std::true_type{});
  if (k_begin < k_end) kstep(k_end - 1, std::false_type{});

#ifdef DIAGAN_WINO_ABLATE
  if (stamping && lane == 0) {
    unsigned long long* o = a.stamps + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
    for (int i = 0; i < 7; ++i) o[i] = tacc[i];
    o[7] = __builtin_amdgcn_s_memtime() - t_entry;       // up to the epilogue
  }
  if (a.tune & 512) {
    if (acc[0][0][0] == 123.456f) a.y[0] = acc[1][1][3] + acc[2][0][5] + acc[3][1][7];
    return;
  }
#endif
  // ---- epilogue ----
  // s[i][b] = sum_j A^T[b][j] M[i][j] in registers (b = 0: M0 + M1 + M2; b = 1: M1 - M2 - M3), then the four rows i meet
  // in LDS ([i][b][64 tiles][64 channels] = 128 KB) and every thread finishes two (tile, 4 channels) items:
  // Y[a][b] = sum_i A^T[a][i] s[i][b].  (Row 3 of V and of U are both staged negated: M is what it always was.)
  const float sc0 = a.scale0 ? a.scale0[0] : a.out_scale, sc1 = a.scale1 ? a.scale1[0] : a.out_scale;
  const int split = a.scale0 ? a.scale_split : 0x7fffffff;            // pixel-row index where the second sigma starts
  const bool raw = a.ksplit > 1;                                       // split-K: un-scaled partial sums to the slab
  const bool hr = !raw && a.residual != nullptr, hm = !raw && a.mask_src != nullptr, hs = !raw && a.stat_partials != nullptr;
  float* ydst = raw ? a.slab + (long)blockIdx.y * a.M * g.Co : a.y;
  const float rfloor = a.res_relu ? 0.f : -__builtin_huge_valf();
  const int et = tid >> 4, ec = (tid & 15) * 4;                       // this thread's tile (within a half) and channel quad
  const int n = n0 + ec;
  const bool col_ok = n < g.Co;                                        // Co % 4 == 0
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (!raw && a.bias && col_ok) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
  f32x4 cs1 = {0.f, 0.f, 0.f, 0.f}, cs2 = {0.f, 0.f, 0.f, 0.f};
  float* ss = smem;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int trow = tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      const float m0 = acc[0][tm][e], m1 = acc[1][tm][e], m2 = acc[2][tm][e], m3 = acc[3][tm][e];
      ss[((wi * 2 + 0) * 64 + trow) * 64 + tn * 32 + fi] = m0 + m1 + m2;
      ss[((wi * 2 + 1) * 64 + trow) * 64 + tn * 32 + fi] = m1 - m2 - m3;
    }
  // this thread's two tiles and their 4 output pixels each; residual / mask loads are issued BEFORE the barrier
  long o4[2][4];
  int prow4[2][4];
  bool ok2[2];
  f32x4 rres[2][9], rmsk[2][4];   // residual: 4 output pixels, or the 3x3 half-resolution neighbourhood (res_up)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int gt = t0 + it * 32 + et;
    ok2[it] = gt < MT && col_ok;
    const unsigned q1 = fdiv((unsigned)(ok2[it] ? gt : 0), a.dWo);
    const int tx = (ok2[it] ? gt : 0) - (int)q1 * TW;
    const unsigned b = fdiv(q1, a.dHo);
    const int ty = (int)q1 - (int)b * TH;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      prow4[it][p] = ((int)b * g.Ho + 2 * ty + (p >> 1)) * g.Wo + 2 * tx + (p & 1);       // pixel (GEMM row) index
      o4[it][p] = (long)prow4[it][p] * g.Co + n;
    }
    if (hr && ok2[it]) {
      if (a.res_up == 2) {
        // the tile's 2x2 output pixels all take a quarter of half-resolution pixel (ty, tx)
        rres[it][0] = *reinterpret_cast<const f32x4*>(a.residual + (((long)b * TH + ty) * TW + tx) * g.Co + n);
      } else if (a.res_up) {
        // the tile's 2x2 output pixels blend the 3x3 neighbourhood of half-resolution pixel (ty, tx), edges clamped
        const float* rb = a.residual + ((long)b * TH * TW) * g.Co + n;
        const int yy[3] = {max(ty - 1, 0), ty, min(ty + 1, TH - 1)}, xx[3] = {max(tx - 1, 0), tx, min(tx + 1, TW - 1)};
#pragma unroll
        for (int q = 0; q < 9; ++q)
          rres[it][q] = *reinterpret_cast<const f32x4*>(rb + ((long)yy[q / 3] * TW + xx[q % 3]) * g.Co);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) rres[it][p] = *reinterpret_cast<const f32x4*>(a.residual + o4[it][p]);
      }
    }
    if (hm && ok2[it]) {
#pragma unroll
      for (int p = 0; p < 4; ++p) rmsk[it][p] = *reinterpret_cast<const f32x4*>(a.mask_src + o4[it][p]);
    }
  }
  __syncthreads();
  // the epilogue proper of (tile it, its 4 output pixels): scale, bias, residual, mask, store, statistics
  const __amdgpu_buffer_rsrc_t psrc =
      __builtin_amdgcn_make_buffer_rsrc(ydst, 0, raw ? (int)((unsigned)a.M * (unsigned)g.Co * 4u) : 0, 0x00020000);
  auto finish = [&](int it, const f32x4* y4, const f32x4* res, const f32x4* msk, bool wr, bool wm, bool ws) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x4 y = y4[p] * (prow4[it][p] < split ? sc0 : sc1) + bv;
      if (wr) {
        f32x4 r;
        if (a.res_up == 2) {
          r = 0.25f * res[0];
        } else if (a.res_up) {
          // upsample2x_kernel's arithmetic: (w0 * a + w1 * b) along x inside the y blend
          const int ya = p >> 1, xa = p & 1;
          const float wy0 = ya ? 0.75f : 0.25f, wx0 = xa ? 0.75f : 0.25f, wy1 = 1.f - wy0, wx1 = 1.f - wx0;
          const f32x4 top = wx0 * res[ya * 3 + xa] + wx1 * res[ya * 3 + xa + 1];
          const f32x4 bot = wx0 * res[ya * 3 + 3 + xa] + wx1 * res[ya * 3 + 3 + xa + 1];
          r = wy0 * top + wy1 * bot;
        } else {
          r = res[p];
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], rfloor);
        }
        y += r;
      }
      if (wm) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = msk[p][e] > 0.f ? y[e] : y[e] * a.mask_slope;
      }
      *reinterpret_cast<f32x4*>(a.y + o4[it][p]) = y;
      if (ws) {
        cs1 += y;
        cs2 += y * y;
      }
    }
  };
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    f32x4 y4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 sa = *reinterpret_cast<const f32x4*>(ss + ((i * 2 + 0) * 64 + it * 32 + et) * 64 + ec);
      const f32x4 sb = *reinterpret_cast<const f32x4*>(ss + ((i * 2 + 1) * 64 + it * 32 + et) * 64 + ec);
      if (i < 3) { y4[0] += sa; y4[1] += sb; }
      if (i == 1) { y4[2] += sa; y4[3] += sb; }
      if (i >= 2) { y4[2] -= sa; y4[3] -= sb; }
    }
    if (ok2[it]) {
      if (raw && a.tickets) {
        // write-through (sc1) stores: the partial sums go to memory as they are written, no release fence (L2 write-back) later
#pragma unroll
        for (int p = 0; p < 4; ++p)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y4[p]), psrc,
                                                 (unsigned)o4[it][p] * 4u, 0, 16);
      } else if (raw) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(ydst + o4[it][p]) = y4[p];
      } else {
        finish(it, y4, rres[it], rmsk[it], hr, hm, hs);
      }
    }
  }
  if (raw && a.tickets) {
    // Round 5: the LAST workgroup of a tile to deliver its partial sums adds the slabs and runs the epilogue itself (no second
    // launch).  Publish = write-through stores, every wave drains them, one lane draws a ticket; the last arriver acquires
    // (invalidates this CU's L1) and every thread reads back, in slab order, the positions it wrote itself.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int tk = __hip_atomic_fetch_add(a.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int lastwg = tk == a.ksplit - 1 ? 1 : 0;
      if (lastwg) {
        __hip_atomic_store(a.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      reinterpret_cast<volatile int*>(smem)[0] = lastwg;
    }
    __syncthreads();
    if (reinterpret_cast<volatile int*>(smem)[0]) {
      const bool fr = a.residual != nullptr, fm = a.mask_src != nullptr;
      if (a.bias && col_ok) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
      const long slab_elems = (long)a.M * g.Co;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (!ok2[it]) continue;
        f32x4 y4[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) y4[p] = *reinterpret_cast<const f32x4*>(a.slab + o4[it][p]);
        for (int k = 1; k < a.ksplit; ++k) {
#pragma unroll
          for (int p = 0; p < 4; ++p) y4[p] += *reinterpret_cast<const f32x4*>(a.slab + k * slab_elems + o4[it][p]);
        }
        f32x4 res[9], msk[4];
        if (fr) {
          if (a.res_up) {
            const int gt = t0 + it * 32 + et;
            const unsigned q1 = fdiv((unsigned)gt, a.dWo);
            const int tx = gt - (int)q1 * TW;
            const unsigned b = fdiv(q1, a.dHo);
            const int ty = (int)q1 - (int)b * TH;
            if (a.res_up == 2) res[0] = *reinterpret_cast<const f32x4*>(a.residual + (((long)b * TH + ty) * TW + tx) * g.Co + n);
            const float* rb = a.residual + ((long)b * TH * TW) * g.Co + n;
            const int yy[3] = {max(ty - 1, 0), ty, min(ty + 1, TH - 1)}, xx[3] = {max(tx - 1, 0), tx, min(tx + 1, TW - 1)};
            if (a.res_up == 1) {
#pragma unroll
              for (int q = 0; q < 9; ++q) res[q] = *reinterpret_cast<const f32x4*>(rb + ((long)yy[q / 3] * TW + xx[q % 3]) * g.Co);
            }
          } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) res[p] = *reinterpret_cast<const f32x4*>(a.residual + o4[it][p]);
          }
        }
        if (fm) {
#pragma unroll
          for (int p = 0; p < 4; ++p) msk[p] = *reinterpret_cast<const f32x4*>(a.mask_src + o4[it][p]);
        }
        finish(it, y4, res, msk, fr, fm, false);
      }
    }
  }
  if (hs) {
    // column sums over the workgroup's 256 pixels: lanes with equal (tid & 15) hold the same channels
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs1[e] += __shfl_xor(cs1[e], 16, 64);
      cs2[e] += __shfl_xor(cs2[e], 16, 64);
      cs1[e] += __shfl_xor(cs1[e], 32, 64);
      cs2[e] += __shfl_xor(cs2[e], 32, 64);
    }
    float* red = smem;                                                 // [8 waves][2][64]
    if (lane < 16) {
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * 64 + ec) = cs1;
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * 64 + ec) = cs2;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, col = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[(w * 2 + which) * 64 + col];
      if (n0 + col < g.Co) a.stat_partials[(long)(tile / tiles_n) * 2 * g.Co + which * g.Co + n0 + col] = t;
    }
  }
}

template <int PRO>
static int launch_wino_pro(const ConvGemmArgs& a, const float* ug, hipStream_t st) {
  const int MT = a.g.B * (a.g.Ho >> 1) * (a.g.Wo >> 1);
  const int wgs = cdiv(MT, WT) * cdiv(a.g.Co, WN);
  const size_t lds = (size_t)2 * W_STAGE * sizeof(float);
  auto kern = conv_wino_kernel<PRO>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, lds);
  hipLaunchKernelGGL(kern, dim3(wgs, a.ksplit), dim3(512), lds, st, a, ug);
  return check_launch("conv_wino");
}

long wino_ws_floats(int Co, int Ci);
// transformed weights of this launch: the caller's ready-made buffer if it hinted the right format, else `ug` after a transform
const float* launch_wino_weights(const float* w, float* ug, int Co, int Ci, int Kp, int flip, hipStream_t st) {
  if (const float* ready = wino_weights_ready(WK_F2, flip, 1.f, wino_ws_floats(Co, Ci))) return ready;
  hipLaunchKernelGGL(wino_weight_kernel, dim3(cdiv(Ci, 32), cdiv(Co, WN)), dim3(512), 0, st, w, ug, Co, Ci, Kp, flip);
  return ug;
}


// floats of workspace the transformed weights need
long wino_ws_floats(int Co, int Ci) { return (long)cdiv(Co, WN) * WN * Ci * 16; }

// Split-K factor for launches with fewer workgroups than ~3/4 of the chip (the 4x4 / 8x8 blocks at batch 64): the smallest
// split that fills the chip (256 workgroups) with at least 8 K-steps each (64 x 8x8 x 256 -> 256: split 4 = 40.7 us
// against 53.2 with split 2 and 51.0 on the implicit GEMM), else the smallest that reaches 128 workgroups with at least 16
// K-steps each (with 8, M = 8192 / Ci = 128 runs 35.0 us against the implicit GEMM's 28.0); 0 = such a launch is better
// served by the implicit GEMM's smaller tiles.  allow_split: a slab can be used (workspace, no fused statistics).
int wino_ksplit(int B, int Ho, int Wo, int Ci, int Co, int allow_split, long ws_floats, int min_wgs) {
  const long wgs = (long)cdiv((long)B * (Ho >> 1) * (Wo >> 1), WT) * cdiv(Co, WN);
  if (wgs >= min_wgs) return 1;
  if (!allow_split) return 0;
  const int nk = Ci / WK;
  auto fits = [&](int ks) { return wino_ws_floats(Co, Ci) + (long)ks * B * Ho * Wo * Co <= ws_floats; };
  for (int ks = 2; ks <= 4; ++ks)
    if (nk / ks >= 8 && wgs * ks >= 256 && fits(ks)) return ks;
  for (int ks = 2; ks <= 4; ++ks)
    if (nk / ks >= 16 && wgs * ks >= 128 && fits(ks)) return ks;
  return 0;
}

// `a` as prepared by diagan_conv_gemm (dWo / dHo re-made here for the TILE grid); ws: wino_ws_floats(Co, Ci) floats
int launch_wino(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 1));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 1));
  const float* ug = launch_wino_weights(a.w, ws, g.Co, g.Ci, g.Kp, g.dr < 0 ? 1 : 0, st);
  switch (a.pro_mode) {
    case PRO_NONE: return launch_wino_pro<PRO_NONE>(a, ug, st);
    case PRO_RELU: return launch_wino_pro<PRO_RELU>(a, ug, st);
    case PRO_AFFINE_RELU: return launch_wino_pro<PRO_AFFINE_RELU>(a, ug, st);
    case PRO_LRELU: return launch_wino_pro<PRO_LRELU>(a, ug, st);
    default: return launch_wino_pro<PRO_AFFINE>(a, ug, st);
  }
}

}  // namespace diagan
