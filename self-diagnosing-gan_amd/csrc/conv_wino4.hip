// Winograd F(4x4, 3x3) forward / data-gradient convolution on the fp32 matrix cores (round 3).
//
// Replaces the same reference ops as conv_wino.hip (F.conv2d and its input gradient in mimicry's GBlock / DBlock,
// selected at diagan-pkg/diagan/models/predefined_models.py:19-21,38-40,57-59,76-78) for the large 3x3 / stride 1 / pad 1
// launches:  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A  per 4x4 output tile with the 6x6 transforms of Lavin & Gray
// (interpolation points 0, +-1, +-2, inf) -- 36 multiply-accumulates per tile, output and input channel instead of 144,
// i.e. 4x fewer MFMA cycles than the implicit GEMM and 1.78x fewer than F(2x2,3x3).  fp32 throughout; rounding error
// against float64 ~1e-5 of the output scale (tools/micro/wino_f4_error.py; F(2x2): 5e-7, direct fp32: 3e-7).  cuDNN --
// what the reference's F.conv2d runs -- uses the same F(4x4,3x3) for these layers in its non-fused Winograd algorithm.
//
// One workgroup = 512 threads = 8 waves = 32 tiles (512 output pixels) x 64 output channels, one per CU (146 KB of LDS).
//   * K loop over input channels in steps of 8.  Per step the 36 "frequency" GEMMs  M_f[32 x 64] += V_f[32 x 8] U_f[64 x 8]^T
//     run as v_mfma_f32_32x32x2_f32: wave w owns a 3 x 3 block of the 6 x 6 frequencies (group w & 3) on column half w >> 2
//     (9 accumulator tiles = 144 registers).
//   * U (transformed weights, written once per launch by wino4_weight_kernel in the order the waves consume it): every U
//     element is used by exactly ONE wave, so a wave fetches its own nine 1 KB units per step by LDS-DMA into wave-private
//     LDS slots (no registers held across the step, no barrier for them); the unit of slot s is re-filled for the next
//     step as soon as this step's fragment of it has been read.
//   * V (transformed input): thread (tile, channel quad, patch row r < 6) loads its 6 pixels x 4 channels (prologue
//     applied here), transforms along the row and parks the result in the V planes; thread (tile, quad, column j) of the
//     same 16-lane group then transforms its column in place (LDS operations of one wave execute in order: no barrier).
//     Planes are [quad][j][i (+ 1 dummy)][tile][4 channels] with a 132-float stride: conflict-free ds_write_b128 in both passes;
//     an MFMA fragment (lane = tile, k half = quad) is ONE conflict-free ds_read_b128 per frequency and step.
//   * epilogue: the 36 products of a (tile, channel) meet in LDS ([f][tile][32 channels] = 144 KB, one column half at a
//     time); thread (tile, channel quad, row pair) applies A^T . A and the usual epilogue (per-half 1/sigma, bias, residual
//     (also the bilinear x2 of a half-resolution one), ReLU-backward mask, BatchNorm statistics), 16-byte stores.
//
// Roofline: MFMA fp32; the kernel executes 36/144 of the direct convolution's multiply-accumulates.
#include "conv_common.h"
#include <type_traits>
#include "wino_weights.h"

namespace diagan {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4T = 32;                 // 4x4-output tiles per workgroup
constexpr int W4N = 64;                 // output channels per workgroup
constexpr int W4K = 8;                  // input channels per K-step
constexpr int W4_PS = 132;              // floats between consecutive V planes (32 tiles x 4 channels + 4 of padding)
// V plane of frequency (i, j) and channel quad q: q * 42 + 7 j + i.  Row stride 1 plane and column stride 7 planes (one
// dummy plane per column) are both odd multiples of 4 banks with the 132-float plane stride, so the six lanes of a group hit
// six different bank quads in the row pass (lanes differ in i) AND in the column pass (lanes differ in j): the 13-cycle
// ds_write_b128 stays at its conflict-free cost (a 12-plane row stride put lanes i and i + 2 on the same banks: 3-way).
constexpr int W4_VSTAGE = 84 * W4_PS;   // V planes of one stage: 2 channel quads x 42
constexpr int W4_U = 72 * 256;          // U of one K-step: 72 units (frequency, column half) of 64 lanes x 4 floats
#ifndef W4_ROW_AT
#define W4_ROW_AT 3           // slot of a K-step after which the next step's row pass / column pass is placed
#endif
#ifndef W4_COL_AT
#define W4_COL_AT 5
#endif
#ifndef W4_DMA_IMM
#define W4_DMA_IMM 1
#endif
#ifndef W4_ULOAD_POOLED
#define W4_ULOAD_POOLED 0     // ... the pooled modes too (K loops of 3 m + 1 / 3 m + 2 steps: one instantiation per remainder).  Built and
#endif                        // measured, NOT kept: <0,2> 1.25 -> 1.22 ms but <1,1> 1.37 -> 1.49 ms per SNGAN-64 step (the unrolled loop
                              // reloads the transform constants from scratch: vector-memory operations the counted waits then drain)
#ifndef W4_PRO_LDS
#define W4_PRO_LDS 1          // ... and, in those modes, the BatchNorm rows from LDS instead of two vector-memory loads per step
#endif
#ifndef W4_ULOAD
#define W4_ULOAD 1            // MODE 0 / 3, fp32 build: weight units by ordinary loads into registers instead of LDS-DMA + ds_read
#endif
// Experiments on the arbitration between the two waves of a SIMD (profiles/r04_w4_kstep_stamps.md: waves 4-7 run ~1400 cycles per
// step behind waves 0-3, which then wait at the barrier): W4_PRIO_B = n > 0 runs waves 4-7 at priority n for the whole kernel;
// W4_PRIO_ALT = n > 0 alternates priority n / 0 slot by slot, in opposite phase for the two wave groups.
#ifndef W4_PRIO_B
#define W4_PRIO_B 0
#endif
#ifndef W4_PRIO_ALT
#define W4_PRIO_ALT 0
#endif
// The same for waves 4-7 (the SIMD partners of waves 0-3: a workgroup's waves w and w + 4 share a SIMD).  With equal placements
// the two waves of a SIMD run their transform passes -- ~70 + ~45 vector / LDS instructions in a row -- at the same time and
// the matrix pipe has only the eight MFMAs already in flight to chew on; with different placements one wave's pass sits beside
// its partner's MFMA slots.  The two groups run two COPIES of the K loop, chosen once by a wave-uniform branch outside it (a
// branch inside would put a control-flow join into the loop, behind which the compiler waits for every DMA in flight).
#ifndef W4_ROW_AT2
#define W4_ROW_AT2 W4_ROW_AT
#endif
#ifndef W4_COL_AT2
#define W4_COL_AT2 W4_COL_AT
#endif
static_assert((2 * W4_VSTAGE + W4_U) * 4 <= 163840 && 36 * 32 * 32 <= 2 * W4_VSTAGE + W4_U,
              "162 432 bytes of the CU's 163 840; the epilogue's 36 x 32 x 32 floats fit inside");

// ---- X3 (round 5): the 36 frequency GEMMs on the bf16 matrix pipe by exact operand splitting ----
// Why: the fp32 MFMA issues at the vector rate on the datapath the transform passes need (profiles/r04_w4_kstep_stamps.md: a
// SIMD's time is the SUM of its MFMAs and its vector instructions, K loop at 0.60 of the pipe at best); v_mfma_f32_32x32x16_bf16
// has 16x the rate and holds the vector issue for 8 of its 32 cycles, so the transforms run UNDER the matrix work.  Every fp32
// operand is split exactly into three bf16 pieces (wino_weights.h: x3_split) and the six piece products of order <= 2^-16
// are accumulated in fp32: per slot and K-step (8 channels) three MFMAs whose k = 16 holds TWO pieces x 8 channels,
//     acc += (a0 | a1) . (b0 | b0)  +  (a0 | a1) . (b1 | b1)  +  (a0 | a2) . (b2 | b0)
// (lanes 0-31 | lanes 32-63) = 27 MFMAs of 32 cycles per wave and step instead of 36 of 64.
// LDS (161 088 bytes): V as piece planes [stage][j][i][piece][32 tiles][8 channels bf16] -- 512 bytes per piece, 16 bytes of
// padding per row i and per column j so that the row pass (lanes differ in i) and the column pass (lanes differ in j) write
// conflict-free 8-byte chunks; the row pass parks its fp32 intermediate in the chunks of pieces 0 / 1 of the plane the column
// pass then overwrites with the pieces (same thread, in-order LDS).  U: every wave streams its own units through a FOUR-unit
// ring (6 KB) in 1 KB LDS-DMA granules: stream byte b lives at ring byte b mod 6144, a granule is issued as soon as the units
// it overwrites have been read, ~3 slots before its own unit is.  The ring's phase repeats every four K-steps: the K loop is
// unrolled four-fold (the launch takes this kernel only for K loops of a multiple of four steps).
// Diagnostic builds only (tools/build_variant.sh <name> conv_wino4.hip "-DX3_ABL=<bits>"; results are then garbage): parts of the
// X3 K loop removed at COMPILE time (the loop stays one basic block) so that their cost can be read off the launch time --
// 1 MFMAs, 2 fragment reads, 4 weight granules, 8 the operand split's arithmetic, 16 row + column pass, 32 input loads, 64 barrier,
// 128 (both builds, MODE 0 / 1): input loads at the addresses of a channel-blocked layout
#ifndef X3_ABL
#define X3_ABL 0
#endif
constexpr int X3_SP = 512;                       // bytes between the pieces of a frequency
constexpr int X3_SI = 3 * X3_SP + 16;            // ... between rows i      (388 dwords = 4 mod 32)
constexpr int X3_SJ = 6 * X3_SI + 16;            // ... between columns j   (2332 dwords = 28 mod 32)
constexpr int X3_VSTAGE = 6 * X3_SJ;             // 55 968 bytes per stage
constexpr int X3_RING = 4 * X3_UNIT;             // 6 144 bytes per wave
constexpr int X3_LDS = 2 * X3_VSTAGE + 8 * X3_RING;
static_assert(X3_LDS <= 163840 && 36 * 32 * 32 * 4 <= X3_LDS, "161 088 bytes of the CU's 163 840; the epilogue's exchange image fits inside");
// granules (1 KB) of a wave's stream: the last one that may be in flight once unit u has been read, the last one unit v needs
constexpr int x3_gmax(int u) { return (3 * (u + 1)) / 2 + 5; }
constexpr int x3_gneed(int v) { return (3 * (v + 1) + 1) / 2 - 1; }
// vmcnt bookkeeping, evaluated at compile time: the number of vector-memory operations (granules, input loads) a wave has issued
// AFTER the operation it is about to wait for, in the steady state of the four-step schedule of kstep3 (ph: step mod 4; hn: a next
// step exists; NX: input loads per step).  kind 0: the last granule of the unit read at the top of the step (s = -1) or inside
// slot s (the unit of slot s + 1); kind 1: the input loads (issued at the top of slot load_at), waited for behind slot row_at's granules.
constexpr int x3_vm_younger(int NS, int ph, bool hn, int NX, int load_at, int row_at, int kind, int s) {
  int seq[256] = {};
  for (int i = 0; i < 256; ++i) seq[i] = -1000;
  int n = 0, seqx = -1000;
  const int t = 4 + ph;
  for (int tt = t - 2; tt <= t; ++tt) {
    const bool h = tt < t ? true : hn;
    const int last = x3_gneed(NS * tt + NS - 1);
    if (tt == t && kind == 0 && s == -1) return n - seq[x3_gneed(NS * tt) - x3_gmax(NS * (t - 2) - 1)] - 1;
    for (int ss = 0; ss < NS; ++ss) {
      const int u = NS * tt + ss;
      if (h && ss == load_at) {              // the input loads of the next step: at the top of slot load_at
        n += NX;
        seqx = n - 1;
      }
      if (ss + 1 < NS && tt == t && kind == 0 && s == ss) return n - seq[x3_gneed(u + 1) - x3_gmax(NS * (t - 2) - 1)] - 1;
      for (int g = x3_gmax(u - 1) + 1; g <= x3_gmax(u); ++g)
        if (h || g <= last) seq[g - x3_gmax(NS * (t - 2) - 1)] = n++;
      if (ss == row_at && h && tt == t && kind == 1) return n - seqx - 1;
    }
  }
  return -1;
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int... V>
constexpr int x3_seq_at(std::integer_sequence<int, V...>, int i) {
  const int v[] = {V...};
  return v[i];
}
template <int N>
__device__ __forceinline__ void x3_wait_vm() {
  static_assert(N >= 0 && N < 48, "vmcnt bookkeeping of the X3 schedule");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0: the convolution.  MODE 1 ("pool"): the convolution FOLLOWED BY F.avg_pool2d(., 2) -- the end of mimicry's DBlock /
// DBlockOptimized with downsample = True (predefined_models.py:38-40,76-78), y / residual are the POOLED tensors.  A 4x4 tile
// holds four pooling windows; window sums are (P A^T) M (P A^T)^T with P A^T = [[1, 2, 0, 3, -1, 0], [0, 2, 0, 12, -4, 1]]:
// frequency row / column 2 never contributes, 25 of the 36 products remain (6.25 per pooled pixel; the F(2x2) pooled kernel
// needs 9, the direct convolution 36).  MODE 2 ("unpool"): the data gradient of such a layer from the POOLED gradient --
// avg_pool2d_backward is constant over each window, the 6-pixel patch of a tile reads (a, b, b, c, c, d) along each axis and
// B^T (a, b, b, c, c, d) = (4a - 5b + c, -8b + 2c, 0, 3c - 3b, b - c, 4b - 5c + d): the same 25 frequencies, and the loader
// reads the 4x4 HALF-resolution neighbourhood (16 instead of 36 pixels per tile).
// The 25 live frequencies l = 5 i' + j' (i', j' index {0, 1, 3, 4, 5}) are dealt to the four wave groups as 7 + 6 + 6 + 6;
// every wave runs 7 slots (the seventh of groups 1-3 multiplies a dummy unit into an accumulator nobody reads).
//
// MODE 3 ("upin"): the convolution of the BILINEAR x2 UP-SAMPLING (align_corners = False) of a half-resolution input --
// the start of mimicry's GBlock residual branch, BN -> ReLU -> F.interpolate(scale_factor = 2, bilinear) -> c1
// (GBlock._upsample_conv, selected at predefined_models.py:19,57).  The six up-sampled pixels of a tile's patch along one
// axis are a fixed linear map of the four half-resolution pixels (a, b, c, d) = rows / columns 2 t - 1 .. 2 t + 2:
//   (3a + b, a + 3b, 3b + c, b + 3c, 3c + d, c + 3d) / 4,
// so the transform passes interpolate first (w4_up below; the two factors 1/4 go into the transformed weights) and the loader
// reads the 4 x 4 half-resolution neighbourhood -- 16 instead of 36 pixels per tile, the prologue (BatchNorm + ReLU acts
// BEFORE the interpolation) on 16 instead of 36, and the up-sampled tensor is never written.  All 36 frequencies are live.
// Borders: interpolation clamps its taps at the image edge (the loads clamp their coordinates), while the convolution's
// zero padding lies at the HIGH resolution: patch position 0 of a first tile / 5 of a last tile is multiplied by zero.
// U[f][co][ci] = (G g G^T)[i][j], f = 6 i + j, in the order the main kernel's waves consume it:
// [64-column block][K-step][unit = wave * 9 + slot][lane = k half * 32 + column][4 channels], wave = group * 2 + column half,
// group = (i / 3) * 2 + j / 3 (a 3 x 3 block of the 6 x 6 frequencies), slot = 3 (i % 3) + j % 3.
// flip: the data-gradient of a stride-1 convolution is the correlation with the taps reversed.
template <int MODE>
__global__ __launch_bounds__(512) void wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ ug, int Co, int Ci,
                                                           int Kp, int flip, float wscale) {
  __shared__ f32x4 sg[WT_LDS_F4];
  wino4_weight_body<MODE>(w, ug, Co, Ci, Kp, flip, wscale, blockIdx.x, blockIdx.y, sg);
}
template <int MODE>
__global__ __launch_bounds__(512) void wino4x_weight_kernel(const float* __restrict__ w, float* __restrict__ ug, int Co, int Ci,
                                                            int Kp, int flip, float wscale) {
  __shared__ f32x4 sg[WT_LDS_F4];
  wino4x_weight_body<MODE>(w, ug, Co, Ci, Kp, flip, wscale, blockIdx.x, blockIdx.y, sg);
}

// the transforms of MANY layers in one launch (diagan_wino_weights_batched): workgroup -> job through the jobs' first-block
// prefix (a handful of jobs: linear scan), then the job's own (channel block, column block) in the job's format
__global__ __launch_bounds__(512) void wino_weights_batched_kernel(const WinoJob* __restrict__ jobs, int n) {
  __shared__ f32x4 sg[WT_LDS_F4];
  int j = 0;
  while (j + 1 < n && (int)blockIdx.x >= jobs[j + 1].blk0) ++j;
  const WinoJob job = jobs[j];
  const int lb = blockIdx.x - job.blk0, nbx = (job.Ci + 31) >> 5;
  if (job.kind == WK_GX3) gx3_weight_job(job.w, job.u, job.Co, job.Kp, lb, nbx * ((job.Co + 63) >> 6));
  else if (job.kind == WK_F4X) wino4x_weight_body<0>(job.w, job.u, job.Co, job.Ci, job.Kp, job.flip, job.scale, lb % nbx, lb / nbx, sg);
  else if (job.kind == WK_F4X_POOL) wino4x_weight_body<1>(job.w, job.u, job.Co, job.Ci, job.Kp, job.flip, job.scale, lb % nbx, lb / nbx, sg);
  else if (job.kind == WK_F4) wino4_weight_body<0>(job.w, job.u, job.Co, job.Ci, job.Kp, job.flip, job.scale, lb % nbx, lb / nbx, sg);
  else if (job.kind == WK_F4_POOL) wino4_weight_body<1>(job.w, job.u, job.Co, job.Ci, job.Kp, job.flip, job.scale, lb % nbx, lb / nbx, sg);
  else wino_weight_body(job.w, job.u, job.Co, job.Ci, job.Kp, job.flip, lb % nbx, lb / nbx, sg);
}

// 1-D input transform B^T (6 -> 6), in place.  Every line is a * K + b with a SCALAR K the compiler cannot see through
// (W4Consts: values parked in scalar registers behind an empty asm), so each becomes one v_pk_fma_f32 per channel pair --
// 12 per pair, 24 per call; written with literal constants the compiler turns the +-1 / +-2 products back into
// subtractions and sign flips that it does not pack (24 v_sub_f32 + 20 v_xor_b32 per call on top of the packed ones).
struct W4Consts {
  float m1, p1, m2, p2, m4, p4, m5, p3;
};
__device__ __forceinline__ W4Consts w4_consts() {
  W4Consts k = {-1.f, 1.f, -2.f, 2.f, -4.f, 4.f, -5.f, 3.f};
  asm volatile("" : "+s"(k.m1), "+s"(k.p1), "+s"(k.m2), "+s"(k.p2), "+s"(k.m4), "+s"(k.p4), "+s"(k.m5), "+s"(k.p3));
  return k;
}
template <typename V>
__device__ __forceinline__ void w4_bt(V* d, const W4Consts& k) {
  const V t0 = d[0] * k.p4 + (d[2] * k.m5 + d[4]);
  const V t5 = d[1] * k.p4 + (d[3] * k.m5 + d[5]);
  const V a = d[2] * k.m4 + d[4], b = d[1] * k.m4 + d[3], c = d[2] * k.m1 + d[4], e = d[1] * k.m1 + d[3];
  d[0] = t0;
  d[1] = b * k.p1 + a;
  d[2] = b * k.m1 + a;
  d[3] = e * k.p2 + c;
  d[4] = e * k.m2 + c;
  d[5] = t5;
}

// B^T of the patch (a, b, b, c, c, d) of an up-sampled (2x2-replicated) tensor, from d[0..3] = (a, b, c, d):
// (4a - 5b + c, -8b + 2c, 0, 3c - 3b, b - c, 4b - 5c + d)
__device__ __forceinline__ void w4_bt_dup(const f32x4* d, f32x4* t, const W4Consts& k) {
  const f32x4 e = d[1] * k.m1 + d[2], f = d[1] * k.m4 + d[2];
  t[0] = d[0] * k.p4 + (d[1] * k.m5 + d[2]);
  t[1] = f * k.p2;
  t[2] = f32x4{0.f, 0.f, 0.f, 0.f};
  t[3] = e * k.p2 + e;
  t[4] = e * k.m1;
  t[5] = d[1] * k.p4 + (d[2] * k.m5 + d[3]);
}

// MODE 3: the six up-sampled patch values (times 4) from the four half-resolution values d[0..3] = (a, b, c, d), then B^T:
//   4 u = (3a + b, a + 3b, 3b + c, b + 3c, 3c + d, c + 3d)
// (6 + 12 multiply-adds per channel, the constant 3 beside w4_bt's).  f0 / f5: per-lane 1, or 0 on a first / last tile, where
// patch position 0 / 5 lies in the convolution's zero padding (with the loads' clamped coordinates it would read 4b / 4c).
template <typename V>
__device__ __forceinline__ void w4_up(const V* d, V* t, const W4Consts& k, float f0, float f5) {
  t[0] = (d[0] * k.p3 + d[1]) * f0;
  t[1] = d[1] * k.p3 + d[0];
  t[2] = d[1] * k.p3 + d[2];
  t[3] = d[2] * k.p3 + d[1];
  t[4] = d[2] * k.p3 + d[3];
  t[5] = (d[3] * k.p3 + d[2]) * f5;
  w4_bt(t, k);
}

// Diagnostic build only (make EXTRA=-DDIAGAN_WINO_ABLATE; tools/wino4_ablate.py): ConvGemmArgs::tune bits switch parts of the K
// loop off (16 row pass, 4096 column pass, 32 input loads, 64 weight DMA, 128 barrier, 256 MFMAs, 2048 fragment reads: the
// results are then garbage) so that their cost can be read off the launch time.
#ifdef DIAGAN_WINO_ABLATE
#define W4_ON(bit) (!(a.tune & (bit)))
#else
#define W4_ON(bit) true
#endif

// LEFT (pooled modes only): 1 / 2 = the ULOAD build for K loops of 3 m + LEFT steps (see ULOAD below); 0 = the LDS-DMA build
template <int PRO, int MODE = 0, bool X3 = false, int LEFT = 0>
__global__ __launch_bounds__(512) void conv_wino4_kernel(const ConvGemmArgs a, const float* __restrict__ ug) {
  using MD = W4M<MODE>;
  constexpr int NS = MD::NS, NI = MD::NI;
#ifdef DIAGAN_W4_STAMP
  const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#endif
  constexpr bool POOL = MODE == 1, UNPOOL = MODE == 2, UPIN = MODE == 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 V stages | U] <= 159 KB
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (W4_PRIO_B > 0 && wave >= 4) __builtin_amdgcn_s_setprio(W4_PRIO_B);
  const int tiles_n = (g.Co + W4N - 1) / W4N;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int t0 = (tile / tiles_n) * W4T, nb = tile % tiles_n, n0 = nb * W4N;
  const int TW = g.Wo >> 2, TH = g.Ho >> 2;
  const int MT = g.B * TH * TW;                         // 4x4 output tiles in all
  const int nk = g.Ci / W4K;
  const int k_per = (nk + a.ksplit - 1) / a.ksplit;
  const int k_begin = blockIdx.y * k_per, k_end = min(k_begin + k_per, nk);
  constexpr bool affine = PRO == PRO_AFFINE_RELU || PRO == PRO_AFFINE;

  // ---- loader role: (tile lt, channel quad lq, patch row / column lr < 6); a 16-lane group holds one tile ----
  // (unpool: the patch is the 4 x 4 HALF-resolution neighbourhood, rows / columns 2 t - 1 .. 2 t + 2 of the pooled gradient)
  // (upin: the row pass splits a tile's 4 rows x 2 channel quads over ALL sixteen lanes -- lane = (row lr >> 1, channel PAIR
  //  lr & 1) with 8-byte loads and LDS writes -- so no lane idles in it and its arithmetic is on two channels, not four)
  const int lr = tid & 7, lq = (tid >> 3) & 1, lt = tid >> 4;
  const int ur = lr >> 1, uc = lr & 1;
  const bool ract = UPIN || lr < (UNPOOL ? 4 : 6);                          // row pass: this lane holds patch row lr
  const bool cact = lr < 6 && !(MD::pooled && lr == 2);                     // column pass: this lane holds column j = lr
  unsigned off[NI];                                      // byte offsets of this row's patch pixels, bit 31 set if outside
  int kbound[NI];                                        // upper clamp of the activation: 0 on padding pixels
  float ex0 = 1.f, ex5 = 1.f, ey0 = 1.f, ey5 = 1.f;      // upin: border factors of the row (x) / column (y) pass, see w4_up
  {
    const int gt = t0 + lt;
    const bool tv = gt < MT && ract;
    const unsigned q1 = fdiv((unsigned)(tv ? gt : 0), a.dWo);          // dWo: divisor TW
    const int tx = (tv ? gt : 0) - (int)q1 * TW;
    const unsigned b = fdiv(q1, a.dHo);                                // dHo: divisor TH
    const int ty = (int)q1 - (int)b * TH;
    const bool half = UNPOOL || UPIN;
    const int Hx = half ? g.Hi >> 1 : g.Hi, Wx = half ? g.Wi >> 1 : g.Wi;     // the gathered tensor's own size
    if (UPIN) {
      // interpolation taps clamp at the image edge: every load is a real pixel (no "padding is zero" in the prologue)
      const int iy = min(max(2 * ty - 1 + ur, 0), Hx - 1);
      const int rowbase = ((int)b * Hx + iy) * Wx;
#pragma unroll
      for (int c = 0; c < NI; ++c) {
        const int ix = min(max(2 * tx - 1 + c, 0), Wx - 1);
        off[c] = tv ? (unsigned)((rowbase + ix) * g.Ci * 4 + lq * 16 + uc * 8) : 0x80000000u;
        kbound[c] = 0x7fffffff;
      }
      ex0 = tx == 0 ? 0.f : 1.f;
      ex5 = tx == TW - 1 ? 0.f : 1.f;
      ey0 = ty == 0 ? 0.f : 1.f;
      ey5 = ty == TH - 1 ? 0.f : 1.f;
    } else {
    const int iy = (UNPOOL ? 2 : 4) * ty - 1 + lr, ix0 = (UNPOOL ? 2 : 4) * tx - 1;
    const bool rv = tv && iy >= 0 && iy < Hx;
    const int rowbase = (((int)b * Hx + iy) * Wx + ix0) * g.Ci * 4 + lq * 16;
#pragma unroll
    for (int c = 0; c < NI; ++c) {
      const bool ok = rv && ix0 + c >= 0 && ix0 + c < Wx;
      off[c] = ok ? (unsigned)(rowbase + c * ((X3_ABL & 128) ? 32 : g.Ci * 4)) : 0x80000000u;  // beyond num_records: the hardware returns zeros
      // (X3_ABL & 128, timing probe: the six pixels of a patch row as six consecutive 32-byte chunks -- the cache lines a
      //  channel-blocked activation layout would touch, a third of NHWC's)
      kbound[c] = ok ? 0x7fffffff : 0;
    }
    }
  }
  // raw buffer descriptor {base, stride 0, num_records, flags}.  The six input loads of a step are issued through inline
  // assembly: the compiler's wait-count pass assumes that LDS-DMA units and ordinary loads return out of order and puts
  // vmcnt(0) in front of the first use of a loaded register, i.e. it would wait for every weight unit in flight; this way
  // the ONLY wait is the explicit vmcnt(n) of wait_inputs (vector memory operations of a wave do return in order).
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 xsrc;
  {
    const unsigned long long xb = (unsigned long long)a.x;
    xsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
    xsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffffu));
    xsrc[2] = __builtin_amdgcn_readfirstlane((int)((unsigned)g.B * g.Hi * g.Wi * g.Ci * ((UNPOOL || UPIN) ? 1u : 4u)));
    xsrc[3] = 0x00020000;
  }
  const int pro_group_off = a.pro_group_rows > 0 ? ((t0 * 16) / a.pro_group_rows) * g.Ci : 0;
  i32x4 scsrc = xsrc, shsrc = xsrc;
  const unsigned poff = (unsigned)(pro_group_off + lq * 4 + (UPIN ? uc * 2 : 0)) * 4u;
  if (affine) {
    const unsigned long long sb = (unsigned long long)a.pro_scale, hb = (unsigned long long)a.pro_shift;
    scsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)sb);
    scsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(sb >> 32) & 0xffffu));
    shsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)hb);
    shsrc[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(hb >> 32) & 0xffffu));
    scsrc[2] = shsrc[2] = 0x7fffffff;       // (the rows are sized by the caller: [groups][Ci])
  }
  // this thread's V slots: row pass writes planes (lr, j), column pass reads / writes planes (i, lr); plane = (6 i + j) * 2 + quad
  // (the idle lanes lr = 6, 7 of a group point far outside the LDS allocation: the hardware drops such writes and returns
  //  zeros for such reads -- no branch around the passes, which keeps the K loop one basic block: behind a join the
  //  compiler's wait-count pass loses track of the in-flight LDS-DMA units and waits for ALL of them before any ds_read)
  float* const vrow = smem + (UPIN ? (lq * 42 + ur) * W4_PS + lt * 4 + uc * 2
                                   : (ract ? (lq * 42 + lr) * W4_PS + lt * 4 : (1 << 22)));          // + j * 7 * W4_PS
  float* const vcol = smem + (cact ? (lq * 42 + 7 * lr) * W4_PS + lt * 4 : (1 << 22));      // + i * W4_PS
  // X3: 8-byte chunks (4 channels bf16, or half of an fp32 quad) of plane (i, j, piece) at i * X3_SI + j * X3_SJ + piece * X3_SP
  // (kept as 32-bit LDS offsets: the passes step ONE running offset from plane to plane behind an empty asm -- the padded strides
  //  fit neither the 8-bit offsets of ds_write2_b64 nor each other's, and twelve ready-made addresses per stage cost registers
  //  the loop does not have: the first X3 build spilled them to scratch inside the K loop)
  char* const sm8 = reinterpret_cast<char*>(smem);
  const unsigned vrow3 = UPIN ? ur * X3_SI + lt * 16 + lq * 8 + uc * X3_SP
                              : (ract ? lr * X3_SI + lt * 16 + lq * 8 : (1u << 24));               // + j * X3_SJ
  const unsigned vcol3 = cact ? lr * X3_SJ + lt * 16 + lq * 8 : (1u << 24);                       // + i * X3_SI

  float* const ulds = smem + 2 * W4_VSTAGE;
  // (weight units are stored [group][column half]: this wave's nine are unit block (wave & 3) * 2 + (wave >> 2))
  const float* ublock = ug + (long)nb * nk * MD::U_FLOATS + ((wave & 3) * 2 + (wave >> 2)) * NS * 256;
  float* const uslot = ulds + wave * NS * 256;                         // this wave's private units

  // wave-uniform unit address (kernel argument + block / wave / step indices: scalar registers) + the lane's 16 bytes
#if W4_DMA_IMM
  // ONE global base and ONE LDS base (M0) for units 0 .. 7 of a step: the unit's distance sits in the instruction's immediate
  // offset, which the hardware adds to BOTH addresses (13 bits, signed: -4096 .. 3072 around unit 4; unit 8 has its own base) --
  // instead of a 64-bit vector add, an M0 write and its wait state per unit
#define W4_DMA_CASE(u)                                                                                                   \
  case u:                                                                                                                \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(up + 4 * 256),                     \
                                     (__attribute__((address_space(3))) void*)(uslot + 4 * 256), 16, ((u) - 4) * 1024, 0); \
    break;
  f32x4 uprobe = {0.f, 0.f, 0.f, 0.f};
  auto issue_u = [&](int kk, int s) {
    const float* up = ublock + (long)kk * MD::U_FLOATS + lane * 4;
    if (X3_ABL & 512) {        // timing probe: the unit fetched by an ordinary load into a register nobody reads (results garbage)
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(uprobe) : "v"(up + s * 256) : "memory");     // ("+v": the register stays reserved)
      return;
    }
    switch (s) {
      W4_DMA_CASE(0) W4_DMA_CASE(1) W4_DMA_CASE(2) W4_DMA_CASE(3) W4_DMA_CASE(4) W4_DMA_CASE(5) W4_DMA_CASE(6) W4_DMA_CASE(7)
      default:
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(up + s * 256),
                                         (__attribute__((address_space(3))) void*)(uslot + s * 256), 16, 0, 0);
    }
  };
#undef W4_DMA_CASE
#else
  auto issue_u = [&](int kk, int s) {
    const float* up = ublock + (long)kk * MD::U_FLOATS + s * 256 + lane * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)up,
                                     (__attribute__((address_space(3))) void*)(uslot + s * 256), 16, 0, 0);
  };
#endif
  const W4Consts kc = w4_consts();
  // ULOAD modes (below) leave the kernel's U region of LDS unused: the BatchNorm rows of this workgroup's group (scale | shift,
  // 2 Ci floats) are parked there once and read back per K-step by two ds_reads instead of two vector-memory loads per step
  constexpr bool PRO_LDS = W4_ULOAD && W4_PRO_LDS && !X3 && (!MD::pooled || LEFT > 0) && affine;
  float* const pls = smem + 2 * W4_VSTAGE;
  int x_kk = 0;                                          // the K-step whose input loads are in flight / being transformed
  f32x4 ra[6], psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  f32x2 rb[4], psc2 = {1.f, 1.f}, psh2 = {0.f, 0.f};    // upin: this lane's channel PAIR of the four pixels
#pragma unroll
  for (int c = NI; c < 6; ++c) ra[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue_x = [&](int kk) __attribute__((always_inline)) {
    const int soff = __builtin_amdgcn_readfirstlane(kk * (W4K * 4));
    x_kk = kk;
    if (UPIN) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(rb[c]) : "v"(off[c]), "s"(xsrc), "s"(soff) : "memory");
      if (affine && !PRO_LDS) {
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(psc2) : "v"(poff), "s"(scsrc), "s"(soff) : "memory");
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(psh2) : "v"(poff), "s"(shsrc), "s"(soff) : "memory");
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < NI; ++c)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[c]) : "v"(off[c]), "s"(xsrc), "s"(soff) : "memory");
    if (affine && !PRO_LDS) {         // BatchNorm scale / shift of this lane's channel quad, same mechanism
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(psc) : "v"(poff), "s"(scsrc), "s"(soff) : "memory");
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(psh) : "v"(poff), "s"(shsrc), "s"(soff) : "memory");
    }
  };
#define W4_WAIT_IN(n)                                                                                            \
  case n:                                                                                                        \
    if (UPIN)                                                                                                    \
      asm volatile("s_waitcnt vmcnt(" #n ")"                                                                     \
                   : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(psc2), "+v"(psh2)::"memory");        \
    else                                                                                                         \
      asm volatile("s_waitcnt vmcnt(" #n ")"                                                                     \
                   : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(psc), "+v"(psh)::"memory"); \
    break;
  // the input loads have landed when at most n younger operations (weight units of the next step) are still in flight
  auto wait_inputs = [&](int n) {
    switch (n) {
      W4_WAIT_IN(0) W4_WAIT_IN(1) W4_WAIT_IN(2) W4_WAIT_IN(3) W4_WAIT_IN(4) W4_WAIT_IN(5) W4_WAIT_IN(6) W4_WAIT_IN(7) W4_WAIT_IN(8)
      W4_WAIT_IN(9)
      default:
        if (UPIN) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(psc2), "+v"(psh2)::"memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(psc), "+v"(psh)::"memory");
    }
  };
#undef W4_WAIT_IN
  // prologue on the loaded pixels + row transform + park in the V planes of `stage`
  auto row_pass = [&](int stage) __attribute__((always_inline)) {
    if (PRO_LDS) {
      const float* pr = pls + x_kk * W4K + lq * 4 + (UPIN ? uc * 2 : 0);
      if (UPIN) {
        psc2 = *reinterpret_cast<const f32x2*>(pr);
        psh2 = *reinterpret_cast<const f32x2*>(pr + g.Ci);
      } else {
        psc = *reinterpret_cast<const f32x4*>(pr);
        psh = *reinterpret_cast<const f32x4*>(pr + g.Ci);
      }
    }
    if (UPIN) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x2 v = rb[c];
        if (affine) v = v * psc2 + psh2;
        if (PRO == PRO_RELU || PRO == PRO_AFFINE_RELU) {
#pragma unroll
          for (int e = 0; e < 2; ++e) v[e] = __int_as_float(max(__float_as_int(v[e]), 0));
        } else if (PRO == PRO_LRELU) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float q = v[e], q2 = 0.2f * q;
            asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(q), "v"(q2));
          }
        }
        rb[c] = v;
      }
      f32x2 t[6];
      w4_up(rb, t, kc, ex0, ex5);
      if (X3) {            // this lane's channel pair is half uc of the quad: chunk `piece uc` of the intermediate
        unsigned o = vrow3 + stage * X3_VSTAGE;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          *reinterpret_cast<f32x2*>(sm8 + o) = t[j];
          o += X3_SJ;
          asm volatile("" : "+v"(o));
        }
        return;
      }
      float* vs = vrow + stage * W4_VSTAGE;
#pragma unroll
      for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2*>(vs + j * 7 * W4_PS) = t[j];
      return;
    }
#pragma unroll
    for (int c = 0; c < NI; ++c) {
      f32x4 v = ra[c];
      if (PRO != PRO_NONE) {
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
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(q), "v"(kbound[c]));     // ReLU and "padding is zero" in one
          } else {
            r = kbound[c] ? q : 0.f;                    // PRO_AFFINE: padding is zero AFTER the affine map
          }
          v[e] = r;
        }
      }
      ra[c] = v;
    }
    f32x4 t[6];
    if (UNPOOL) w4_bt_dup(ra, t, kc);
    else {
#pragma unroll
      for (int c = 0; c < 6; ++c) t[c] = ra[c];
      w4_bt(t, kc);
    }
    if (X3) {              // the fp32 intermediate of (row lr, column j) as two 8-byte halves in the chunks of pieces 0 / 1
      unsigned o = vrow3 + stage * X3_VSTAGE;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (!(MD::pooled && j == 2)) {
          *reinterpret_cast<f32x2*>(sm8 + o) = f32x2{t[j][0], t[j][1]};
          *reinterpret_cast<f32x2*>(sm8 + o + X3_SP) = f32x2{t[j][2], t[j][3]};
        }
        o += X3_SJ;
        asm volatile("" : "+v"(o));
      }
      return;
    }
    float* vs = vrow + stage * W4_VSTAGE;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (!(MD::pooled && j == 2)) *reinterpret_cast<f32x4*>(vs + j * 7 * W4_PS) = t[j];
  };
  // column transform of column lr, in place (reads what the row pass of this 16-lane group parked)
  auto col_pass = [&](int stage) __attribute__((always_inline)) {
    float* vs = vcol + stage * W4_VSTAGE;
    f32x4 d[6];
    if (X3) {
      unsigned o = vcol3 + stage * X3_VSTAGE;
#pragma unroll
      for (int i = 0; i < ((UNPOOL || UPIN) ? 4 : 6); ++i) {
        const f32x2 lo = *reinterpret_cast<const f32x2*>(sm8 + o), hi = *reinterpret_cast<const f32x2*>(sm8 + o + X3_SP);
        d[i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        o += X3_SI;
        asm volatile("" : "+v"(o));
      }
    } else {
#pragma unroll
    for (int i = 0; i < ((UNPOOL || UPIN) ? 4 : 6); ++i) d[i] = *reinterpret_cast<const f32x4*>(vs + i * W4_PS);
    }
    if (UPIN) {
      f32x4 t[6];
      w4_up(d, t, kc, ey0, ey5);
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = t[i];
    } else if (UNPOOL) {
      f32x4 t[6];
      w4_bt_dup(d, t, kc);
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = t[i];
    } else {
      w4_bt(d, kc);
    }
    if (X3) {              // split, and write the three pieces over the intermediate this thread has just read
      unsigned o = vcol3 + stage * X3_VSTAGE;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (!(MD::pooled && i == 2)) {
          u32x2 p0, p1, p2;
          if (X3_ABL & 8) {
            p0 = u32x2{__float_as_uint(d[i][0]), __float_as_uint(d[i][1])};
            p1 = u32x2{__float_as_uint(d[i][2]), __float_as_uint(d[i][3])};
            p2 = p0;
          } else
          x3_split(d[i], p0, p1, p2);
          *reinterpret_cast<u32x2*>(sm8 + o) = p0;
          *reinterpret_cast<u32x2*>(sm8 + o + X3_SP) = p1;
          *reinterpret_cast<u32x2*>(sm8 + o + 2 * X3_SP) = p2;
        }
        o += X3_SI;
        asm volatile("" : "+v"(o));
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (!(MD::pooled && i == 2)) *reinterpret_cast<f32x4*>(vs + i * W4_PS) = d[i];
  };

  // wave w owns the 3 x 3 frequency block of group w & 3 on column half w >> 2
  // (column half = wave >> 2: the two waves of a SIMD, w and w + 4, then sit in different halves, and each epilogue phase
  //  below keeps one wave per SIMD busy)
  const int grp = wave & 3, nh = wave >> 2;
  f32x16 acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
  const int fi = lane & 31, kh = lane >> 5;

  // ---- X3: this wave's weight stream (bytes; + 3072: the six granules of a ring pass are addressed around its middle, the
  // distance -3072 .. 2048 in the DMA instruction's immediate offset, which the hardware adds to the global AND the LDS address)
  char* const uring = sm8 + 2 * X3_VSTAGE + wave * X3_RING;
  const char* ugrp = reinterpret_cast<const char*>(ug) + ((long)(nb * 8 + wave) * nk + k_begin) * (NS * X3_UNIT) + lane * 16 + 3072;
#define X3_DMA_CASE(r)                                                                                                  \
  case r:                                                                                                                \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,                                  \
                                     (__attribute__((address_space(3))) void*)(uring + 3072), 16, ((r) - 3) * 1024, 0);  \
    break;
  auto x3_issue_g = [&](const char* grp_base, int g) __attribute__((always_inline)) {        // g: granule of the four-step group (compile-time after unrolling)
    const int r = g % 6;
    const char* gp = grp_base + (g - r) * 1024;
    switch (r) { X3_DMA_CASE(0) X3_DMA_CASE(1) X3_DMA_CASE(2) X3_DMA_CASE(3) X3_DMA_CASE(4) X3_DMA_CASE(5) }
  };
#undef X3_DMA_CASE
  // ---- ULOAD (round 5, MODE 0 / 3 of the fp32 build): the weight units by ordinary loads, straight into the MFMA's B registers ----
  // Every unit is read by exactly one wave, once: the detour through LDS (one LDS-DMA per unit, one ds_read_b128 to get it back)
  // buys nothing but the long look-ahead.  Three register quads per wave form a ring: the unit of slot s + 3 is requested right
  // behind slot s's MFMAs (its register is free then) and waited for in front of slot s + 3's -- two to three slots (1 700-2 500
  // cycles) of flight.  72 KB of the 159 KB of LDS are then unused.  Timing probe before the build: profiles/r05_bf16x6_wino.md.
  // (nine units per step: a unit's place in the ring of three is s mod 3.  The pooled modes' seven units per step make it
  //  (7 step + s) mod 3: the K loop is unrolled three-fold and its tail of LEFT = 1 / 2 steps is straight-line code of a kernel
  //  instantiated for that remainder -- a run-time tail spilled the accumulators across its paths (400+ registers); K loops
  //  of a multiple of three steps keep the LDS-DMA path, LEFT = 0)
  constexpr bool ULOAD = W4_ULOAD && !X3 && (!MD::pooled || LEFT > 0);
  f32x4 ub[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const unsigned ulane = lane * 16;
#define W4_UL_LOAD(u, r) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(ub[r]) : "v"(ulane), "s"(ub4), "n"(((u) - 4) * 1024) : "memory")
#define W4_UL_CASE(u)                  \
  case u:                              \
    if (r == 0) W4_UL_LOAD(u, 0);      \
    else if (r == 1) W4_UL_LOAD(u, 1); \
    else W4_UL_LOAD(u, 2);             \
    break;
  auto issue_ul = [&](int kk, int u, int r) __attribute__((always_inline)) {       // unit u of step kk -> ub[r]
    // (units 0 .. 7 around one scalar base, the distance in the instruction's 13-bit offset; unit 8 from a base of its own)
    const unsigned long long ub4 = (unsigned long long)(ublock + (long)kk * MD::U_FLOATS) + 4096;
    const unsigned long long ub8 = ub4 + 4096;
    switch (u) {
      W4_UL_CASE(0) W4_UL_CASE(1) W4_UL_CASE(2) W4_UL_CASE(3) W4_UL_CASE(4) W4_UL_CASE(5) W4_UL_CASE(6) W4_UL_CASE(7)
      default: asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ub[2]) : "v"(ulane), "s"(ub8) : "memory");      // (unit 8: nine-unit steps, ring place 2)
    }
  };
#undef W4_UL_CASE
#undef W4_UL_LOAD
  if (PRO_LDS) {
    for (int c = tid * 4; c < 2 * g.Ci; c += 2048) {
      const float* src = c < g.Ci ? a.pro_scale + pro_group_off + c : a.pro_shift + pro_group_off + (c - g.Ci);
      *reinterpret_cast<f32x4*>(pls + c) = *reinterpret_cast<const f32x4*>(src);
    }
    __syncthreads();
  }
  if (k_begin < k_end) {
    if (X3) {
#pragma unroll
      for (int g = 0; g <= x3_gmax(-1); ++g) x3_issue_g(ugrp, g);
    } else if (ULOAD) {
#pragma unroll
      for (int s = 0; s < 3; ++s) issue_ul(k_begin, s, s);
    } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) issue_u(k_begin, s);
    }
    issue_x(k_begin);
    wait_inputs(0);
    row_pass(0);
    col_pass(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fi0 = 3 * (grp >> 1), fj0 = 3 * (grp & 1);      // (MODE 0) this wave's block of frequencies
  // plane (7 j + i) of this wave's slot s, relative to plane 0 of its channel quad:
  //   MODE 0: slot s = 3 di + dj of the block at (fi0, fj0): a compile-time distance from the block's first plane;
  //   pooled : live frequency l = w4p_start(group) + s (the dummy seventh slot of groups 1-3 re-reads the group's first plane)
  auto slot_plane = [&](int s) {                              // (MODE 0: relative to the block's first plane, a compile-time constant)
    if (!MD::pooled) return 7 * (s % 3) + s / 3;
    const int l = w4p_start(grp) + (s < w4p_count(grp) ? s : 0);
    return 7 * w4p_freq(l % 5) + w4p_freq(l / 5);
  };
  const float* const fa_lane = smem + (kh * 42 + (MD::pooled ? 0 : 7 * fj0 + fi0)) * W4_PS + fi * 4;
  int fa_off[NS];                                             // pooled modes: wave-uniform (scalar registers)
#pragma unroll
  for (int s = 0; s < NS; ++s) fa_off[s] = slot_plane(s) * W4_PS;
  const float* const fb_base = uslot + lane * 4;
  // vm-counter bookkeeping: a wave's DMA unit of slot s for step kk + 1 is issued right after slot s of step kk has been
  // consumed and gets a WHOLE step to land -- the wait sits in front of the fragment read of step kk + 1, not at the end
  // of step kk.  Operations younger than D(kk, s + 1) when fragment s + 1 is about to be read: D(kk, s + 2 .. NS - 1), the NI
  // input loads of step kk + 1 (+ 2 with the BatchNorm rows) and D(kk + 1, 0 .. s - 1): NS - 2 + NI (+ 2) in every slot.
#define W4_WAIT_VM(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
  auto wait_vm = [&](int n) {          // n is a compile-time constant after unrolling: the switch folds to one s_waitcnt
    switch (n) {
      W4_WAIT_VM(0) W4_WAIT_VM(1) W4_WAIT_VM(2) W4_WAIT_VM(3) W4_WAIT_VM(4) W4_WAIT_VM(5) W4_WAIT_VM(6) W4_WAIT_VM(7)
      W4_WAIT_VM(8) W4_WAIT_VM(9) W4_WAIT_VM(10) W4_WAIT_VM(11) W4_WAIT_VM(12) W4_WAIT_VM(13) W4_WAIT_VM(14) W4_WAIT_VM(15)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
#undef W4_WAIT_VM
#ifdef DIAGAN_W4_STAMP
  // Diagnostic build only (tools/build_variant.sh stamp conv_wino4.hip "-DDIAGAN_W4_STAMP", tools/wino4_stamps.py): per-wave
  // s_memtime deltas of the phases of a K-step, summed over the K loop, written to
  // ConvGemmArgs::stamps[(workgroup * 8 + wave) * 16 + phase]; [12] the K loop, [13] its steps, [14] set-up, [15] epilogue.  No
  // branch on a run-time flag (the K loop must stay one basic block); every stamp drains the wave's LDS queue (s_memtime
  // returns through lgkmcnt), so a stamped step is a few hundred cycles longer than a production one.
  unsigned tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long t_entry = tlast;
  auto tick = [&](int i) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    tacc[i] += (unsigned)(now - tlast);
    tlast = now;
  };
#define W4_TICK(i) tick(i)
#else
#define W4_TICK(i)
#endif
  auto kstep = [&](int kk, auto has_next, auto row_at, auto col_at, auto prio_phase) {
    constexpr bool HN = decltype(has_next)::value;
    constexpr int PPH = decltype(prio_phase)::value;          // experiment W4_PRIO_ALT: -1 none; 0 / 1: high priority in even / odd slots
    constexpr int ROW_AT = decltype(row_at)::value, COL_AT = decltype(col_at)::value;
    const int cur = (kk - k_begin) & 1;
    W4_TICK(9);                                               // (behind the barrier of the previous step)
    if (HN) wait_vm(NS - 1);                                  // D(kk, 0) has landed (D(kk, 1 .. NS - 1) may still fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4_TICK(0);                                               // wait for the first weight unit
    const float* const fa_base = fa_lane + cur * W4_VSTAGE;
    f32x4 fa[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, fb[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
    if (W4_ON(2048)) {
      fa[0] = *reinterpret_cast<const f32x4*>(fa_base + fa_off[0]);
      if (!(X3_ABL & 256)) fb[0] = *reinterpret_cast<const f32x4*>(fb_base);
    }
    if (HN && W4_ON(32) && !(X3_ABL & 32)) issue_x(kk + 1);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) {
        if (HN) wait_vm(NS - 2 + NI + (affine ? 2 : 0));
        if (W4_ON(2048)) {
          fa[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(fa_base + fa_off[s + 1]);
          if (!(X3_ABL & 256)) fb[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(fb_base + (s + 1) * 256);
        }
      }
      if (PPH >= 0) {
        if ((s + PPH) & 1) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(W4_PRIO_ALT);
      }
      if (W4_ON(256) && !(X3_ABL & 1))
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][e], fb[s & 1][e], acc[s], 0, 0, 0);
      if (HN) {
        // slot s has been read into registers (the MFMAs above needed it): re-fill it for the next step
        __builtin_amdgcn_sched_barrier(0);
        if (W4_ON(64) && !(X3_ABL & 4)) issue_u(kk + 1, s);
        if (s == (ROW_AT < NS - 2 ? ROW_AT : NS - 4)) {
          W4_TICK(1);                                     // slots 0 .. ROW_AT: fragment reads, input-load issue, MFMA issue
          wait_inputs(s + 1);                             // the input loads have landed (the s + 1 younger DMAs may still fly)
          W4_TICK(2);                                     // wait for the input loads
          if (W4_ON(16) && !(X3_ABL & 16)) row_pass(cur ^ 1);
          W4_TICK(3);                                     // row pass (issue; the stamp also drains its LDS writes)
        } else if (s == (COL_AT < NS - 1 ? COL_AT : NS - 2)) {
          W4_TICK(4);                                     // slots ROW_AT + 1 .. COL_AT
          if (W4_ON(16) && W4_ON(4096) && !(X3_ABL & 16)) col_pass(cur ^ 1);
          W4_TICK(5);                                     // column pass (LDS reads, arithmetic, LDS writes drained)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // V of the next step is complete (LDS writes: lgkmcnt; U is wave-private and waited for where it is read).  NOT
    // __syncthreads(): its fence would make the compiler wait for every DMA unit in flight
    W4_TICK(6);                                               // slots COL_AT + 1 .. NS - 1
#ifdef DIAGAN_W4_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_TICK(7);                                               // LDS queue drained
#endif
    if (X3_ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (W4_ON(128)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    W4_TICK(8);                                               // barrier
  };
  // ---- ULOAD K-step: as kstep, the B operand from the register ring ub[] (see ULOAD above) ----
  // vmcnt bookkeeping (in issue order: at the top of a step the NX input loads of the next one, behind slot s's MFMAs the load
  // of unit s + 3): in front of slot s's MFMAs units s + 1 and s + 2 are younger than unit s -- and, for s < 3, the input
  // loads issued since (unit s was requested in the previous step); the input loads are waited for behind slot ROW_AT's request
  auto kstep_ul = [&](int kk, auto has_next, auto phase) __attribute__((always_inline)) {
    constexpr bool HN = decltype(has_next)::value;
    constexpr int R0 = (NS * decltype(phase)::value) % 3;     // ring place of this step's unit 0
    constexpr int ROW_AT = W4_ROW_AT, COL_AT = W4_COL_AT;
    constexpr int NXL = NI + (affine && !PRO_LDS ? 2 : 0);
    const int cur = (kk - k_begin) & 1;
    W4_TICK(9);
    const float* const fa_base = fa_lane + cur * W4_VSTAGE;
    f32x4 fa[2];
    fa[0] = *reinterpret_cast<const f32x4*>(fa_base + fa_off[0]);
    W4_TICK(0);
    if (HN) issue_x(kk + 1);
    static_for<0, NS>([&](auto slot) __attribute__((always_inline)) {
      constexpr int s = decltype(slot)::value;
      if (s + 1 < NS) fa[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(fa_base + fa_off[s + 1 < NS ? s + 1 : 0]);
      {
        constexpr int NW = (HN ? 2 + (s < 3 ? (s == 0 ? 0 : NXL) : 0) : (NS - 1 - s < 2 ? NS - 1 - s : 2));
        // (s == 0: the input loads of this step are issued BEHIND this wait only if the wait came first -- it does not: they
        //  were issued above, so they count)
        constexpr int NW0 = HN && s == 0 ? 2 + NXL : NW;
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ub[(R0 + s) % 3]) : "n"(NW0) : "memory");
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][e], ub[(R0 + s) % 3][e], acc[s], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 3 < NS) issue_ul(kk, s + 3, (R0 + s) % 3);            // (the unit three slots ahead takes the register just read)
      else if (HN) issue_ul(kk + 1, s + 3 - NS, (R0 + s) % 3);
      if (HN) {
        if (s == ROW_AT) {
          W4_TICK(1);
          constexpr int NR = ROW_AT + 1;                      // the requests behind slots 0 .. ROW_AT are younger
          if (UPIN)
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(psc2), "+v"(psh2) : "n"(NR) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(psc), "+v"(psh)
                         : "n"(NR)
                         : "memory");
          W4_TICK(2);
          row_pass(cur ^ 1);
          W4_TICK(3);
        } else if (s == COL_AT) {
          W4_TICK(4);
          col_pass(cur ^ 1);
          W4_TICK(5);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    W4_TICK(6);
#ifdef DIAGAN_W4_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_TICK(7);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    W4_TICK(8);
  };

  // ---- X3 K-step (see the notes at X3_SP above) ----
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int NX = NI + (affine ? 2 : 0);
  // placement of the next step's input loads / row pass / column pass (the slot they follow): waves 0-3 early, their SIMD partners
  // 4-7 late -- while one wave of a SIMD runs its transform passes (vector / LDS work) the other is in MFMA slots, and the two
  // groups' load bursts and LDS write bursts do not coincide (two copies of the loop, chosen by a wave-uniform branch outside it)
#ifndef X3_PLACE_A
#define X3_PLACE_A 0, 3, 5
#endif
#ifndef X3_PLACE_B
#define X3_PLACE_B 3, 6, 8
#endif
  constexpr int x3_pa[3] = {X3_PLACE_A}, x3_pb[3] = {X3_PLACE_B};
  constexpr int sh3 = 9 - NS;                      // (pooled modes: seven slots)
  using PlA = std::integer_sequence<int, x3_pa[0], x3_pa[1] - (x3_pa[1] > 2 ? sh3 : 0), x3_pa[2] - sh3>;
  using PlB = std::integer_sequence<int, x3_pb[0] - (x3_pb[0] > 1 ? sh3 : 0), x3_pb[1] - sh3, x3_pb[2] - sh3>;
  // fragment addresses: A = V pieces of (tile fi): (a0 | a1) and (a0 | a2) for lanes (0-31 | 32-63); B = this wave's ring:
  // (b0 | b0) and (b1 | b1) through one address, (b2 | b0) through the other
  const char* const aP = sm8 + (MD::pooled ? 0 : fi0 * X3_SI + fj0 * X3_SJ) + kh * X3_SP + fi * 16;
  const char* const aQ = aP + kh * X3_SP;
  const char* const b0 = uring + fi * 16;
  const char* const b3 = uring + (kh ? 0 : 2 * X3_SP) + fi * 16;
  int so3[NS];                                                 // slot s: byte offset of its frequency's piece-0 plane
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (!MD::pooled) so3[s] = (s / 3) * X3_SI + (s % 3) * X3_SJ;
    else {
      const int l = w4p_start(grp) + (s < w4p_count(grp) ? s : 0);
      so3[s] = w4p_freq(l / 5) * X3_SI + w4p_freq(l % 5) * X3_SJ;
    }
  }
  auto kstep3 = [&](int kk, auto phase, auto has_next, const char* grp_base, auto place) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase)::value;
    constexpr bool HN = decltype(has_next)::value;
    constexpr int cur = PH & 1;
    constexpr int X3_LOAD_AT = x3_seq_at(place, 0), X3_ROW_AT = x3_seq_at(place, 1), X3_COL_AT = x3_seq_at(place, 2);
    static_assert(X3_LOAD_AT >= 0 && X3_LOAD_AT < X3_ROW_AT && X3_ROW_AT < X3_COL_AT && X3_COL_AT < NS, "placement of the passes");
    W4_TICK(9);
    x3_wait_vm<x3_vm_younger(NS, PH, HN, NX, X3_LOAD_AT, X3_ROW_AT, 0, -1)>();
    W4_TICK(0);
    const char* const fap = aP + cur * X3_VSTAGE;
    const char* const faq = aQ + cur * X3_VSTAGE;
    // ONE set of fragment registers, refilled on the fly: each read of the NEXT slot's fragments is issued right behind the last MFMA
    // that uses its destination (fb[0] after the first, fa[0] / fb[1] after the second, fa[1] / fb[2] after the third) -- the
    // partner wave's MFMAs and this wave's next two cover the LDS latency; a second set would cost 20 registers the loop lacks
    u32x4 fa[2] = {{1, 1, 1, 1}, {1, 1, 1, 1}}, fb[3] = {{1, 1, 1, 1}, {1, 1, 1, 1}, {1, 1, 1, 1}};
    if (!(X3_ABL & 2)) {
      constexpr int ro = ((NS * PH) & 3) * X3_UNIT;
      fa[0] = *reinterpret_cast<const u32x4*>(fap + so3[0]);
      fa[1] = *reinterpret_cast<const u32x4*>(faq + so3[0]);
      fb[0] = *reinterpret_cast<const u32x4*>(b0 + ro);
      fb[1] = *reinterpret_cast<const u32x4*>(b0 + ro + X3_SP);
      fb[2] = *reinterpret_cast<const u32x4*>(b3 + ro);
    }
    static_for<0, NS>([&](auto slot) __attribute__((always_inline)) {
      constexpr int s = decltype(slot)::value, u = NS * PH + s;
      if (HN && s == X3_LOAD_AT && !(X3_ABL & 32)) {
        issue_x(kk + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int sn = s + 1 < NS ? s + 1 : 0, ro = ((u + 1) & 3) * X3_UNIT;          // the next slot's planes and ring position
      constexpr bool rd = s + 1 < NS && !(X3_ABL & 2), mm = !(X3_ABL & 1);
      if (mm) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[0]), __builtin_bit_cast(bf16x8, fb[0]), acc[s], 0, 0, 0);
      if (s + 1 < NS) {
        __builtin_amdgcn_sched_barrier(0);
        x3_wait_vm<x3_vm_younger(NS, PH, HN, NX, X3_LOAD_AT, X3_ROW_AT, 0, s + 1 < NS ? s : 0)>();       // the next unit's granules have landed
        if (rd) fb[0] = *reinterpret_cast<const u32x4*>(b0 + ro);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (mm) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[0]), __builtin_bit_cast(bf16x8, fb[1]), acc[s], 0, 0, 0);
      if (rd) {
        __builtin_amdgcn_sched_barrier(0);
        fa[0] = *reinterpret_cast<const u32x4*>(fap + so3[sn]);
        fb[1] = *reinterpret_cast<const u32x4*>(b0 + ro + X3_SP);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (mm) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[1]), __builtin_bit_cast(bf16x8, fb[2]), acc[s], 0, 0, 0);
      if (rd) {
        __builtin_amdgcn_sched_barrier(0);
        fa[1] = *reinterpret_cast<const u32x4*>(faq + so3[sn]);
        fb[2] = *reinterpret_cast<const u32x4*>(b3 + ro);
      }
      __builtin_amdgcn_sched_barrier(0);
      // unit u has been read into registers: the granules that overwrite nothing younger may fly
#pragma unroll
      for (int g = x3_gmax(u - 1) + 1; g <= x3_gmax(u); ++g)
        if ((HN || g <= x3_gneed(NS * PH + NS - 1)) && !(X3_ABL & 4)) x3_issue_g(grp_base, g);
      if (HN && !(X3_ABL & 16)) {
        if (s == X3_ROW_AT) {
          W4_TICK(1);
          constexpr int NW = x3_vm_younger(NS, PH, true, NX, X3_LOAD_AT, X3_ROW_AT, 1, 0);
          static_assert(NW >= 0 && NW < 48, "input-load wait of the X3 schedule");
          if (UPIN)
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(psc2), "+v"(psh2) : "n"(NW) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(psc), "+v"(psh)
                         : "n"(NW)
                         : "memory");
          W4_TICK(2);
          row_pass(cur ^ 1);
          W4_TICK(3);
        } else if (s == X3_COL_AT) {
          W4_TICK(4);
          col_pass(cur ^ 1);
          W4_TICK(5);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    W4_TICK(6);
#ifdef DIAGAN_W4_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4_TICK(7);
#endif
    if (X3_ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    W4_TICK(8);
  };
  if constexpr (X3) {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto kloop3 = [&](auto place) __attribute__((always_inline)) {
      int kk = k_begin;
      for (; kk + 4 < k_end; kk += 4) {                            // (k_end - k_begin is a multiple of four: launch condition)
        kstep3(kk, I0{}, std::true_type{}, ugrp, place);
        kstep3(kk + 1, I1{}, std::true_type{}, ugrp, place);
        kstep3(kk + 2, I2{}, std::true_type{}, ugrp, place);
        kstep3(kk + 3, I3{}, std::true_type{}, ugrp, place);
        ugrp += 4 * NS * X3_UNIT;
      }
      if (k_begin < k_end) {
        kstep3(kk, I0{}, std::true_type{}, ugrp, place);
        kstep3(kk + 1, I1{}, std::true_type{}, ugrp, place);
        kstep3(kk + 2, I2{}, std::true_type{}, ugrp, place);
        kstep3(kk + 3, I3{}, std::false_type{}, ugrp, place);
      }
    };
    if (std::is_same<PlA, PlB>::value || wave < 4) kloop3(PlA{});
    else kloop3(PlB{});
  } else if constexpr (ULOAD) {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    if (NS % 3 == 0) {
      for (int kk = k_begin; kk + 1 < k_end; ++kk) kstep_ul(kk, std::true_type{}, P0{});
      if (k_begin < k_end) kstep_ul(k_end - 1, std::false_type{}, P0{});
    } else {                                   // seven units per step: the ring's phase repeats every three steps
      int kk = k_begin;
      for (; kk + 3 < k_end; kk += 3) {          // (k_end - k_begin = 3 m + LEFT: the launch picks the instantiation)
        kstep_ul(kk, std::true_type{}, P0{});
        kstep_ul(kk + 1, std::true_type{}, P1{});
        kstep_ul(kk + 2, std::true_type{}, P2{});
      }
      if (LEFT == 1) kstep_ul(kk, std::false_type{}, P0{});
      else {
        kstep_ul(kk, std::true_type{}, P0{});
        kstep_ul(kk + 1, std::false_type{}, P1{});
      }
    }
  } else {
  using RowA = std::integral_constant<int, W4_ROW_AT>;
  using ColA = std::integral_constant<int, W4_COL_AT>;
  using RowB = std::integral_constant<int, W4_ROW_AT2>;
  using ColB = std::integral_constant<int, W4_COL_AT2>;
  using PN = std::integral_constant<int, -1>;
  using PA = std::integral_constant<int, (W4_PRIO_ALT > 0) ? 0 : -1>;
  using PB = std::integral_constant<int, (W4_PRIO_ALT > 0) ? 1 : -1>;
  if ((W4_ROW_AT2 == W4_ROW_AT && W4_COL_AT2 == W4_COL_AT && W4_PRIO_ALT == 0) || wave < 4) {
    for (int kk = k_begin; kk + 1 < k_end; ++kk) kstep(kk, std::true_type{}, RowA{}, ColA{}, PA{});
  } else {
    for (int kk = k_begin; kk + 1 < k_end; ++kk) kstep(kk, std::true_type{}, RowB{}, ColB{}, PB{});
  }
  if (W4_PRIO_ALT > 0) __builtin_amdgcn_s_setprio(0);
  if (k_begin < k_end) kstep(k_end - 1, std::false_type{}, RowA{}, ColA{}, PN{});
  }

#ifdef DIAGAN_W4_STAMP
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16;
    for (int i = 0; i < 12; ++i) o[i] = tacc[i];
    o[12] = __builtin_amdgcn_s_memtime() - t_entry;      // the K loop
    o[13] = k_end - k_begin;
    o[14] = t_entry - t_kernel;                          // set-up + first stage
  }
  const unsigned long long t_epi = __builtin_amdgcn_s_memtime();
#endif
#ifdef DIAGAN_WINO_ABLATE
  if (a.tune & 512) {
    if (acc[0][0] == 123.456f) a.y[0] = acc[1][3] + acc[5][5] + acc[8][7];
    return;
  }
#endif
  if (X3_ABL & 512) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(uprobe)::"memory");
    if (uprobe[0] == 123.456f) a.y[0] = uprobe[1];
  }
  // ---- epilogue ----
  const float sc0 = a.scale0 ? a.scale0[0] : a.out_scale, sc1 = a.scale1 ? a.scale1[0] : a.out_scale;
  // pixel-row index (of the tensor that is written: the pooled one in MODE 1) where the second sigma starts
  const int split = a.scale0 ? (POOL ? a.scale_split >> 2 : a.scale_split) : 0x7fffffff;
  const bool raw = a.ksplit > 1;                                       // split-K: un-scaled partial sums to the slab
  const bool hr = !raw && a.residual != nullptr, hm = !raw && a.mask_src != nullptr, hs = !raw && a.stat_partials != nullptr;
  const int Mout = POOL ? a.M >> 2 : a.M;                              // pixels of the written tensor
  float* ydst = raw ? a.slab + (long)blockIdx.y * Mout * g.Co : a.y;
  const float rfloor = a.res_relu ? 0.f : -__builtin_huge_valf();
  const int et = tid >> 4, eq = (tid >> 1) & 7, eh = tid & 1;         // tile, channel quad (of a 32-column half), row pair
  // this thread's tile and its output pixels (2 x 4 of the tile's 4 x 4; pool: one row of the tile's 2 x 2 pooled pixels):
  // ONE 32-bit byte offset per pass, the pixels' distances are wave-uniform (scalar offsets of raw buffer loads / stores)
  const int gt = t0 + et;
  const bool tv = gt < MT;
  const unsigned q1 = fdiv((unsigned)(tv ? gt : 0), a.dWo);
  const int tx = (tv ? gt : 0) - (int)q1 * TW;
  const unsigned eb = fdiv(q1, a.dHo);
  const int ty = (int)q1 - (int)eb * TH;
  const int Hy = POOL ? g.Ho >> 1 : g.Ho, Wy = POOL ? g.Wo >> 1 : g.Wo;        // the written tensor's size
  const int oy0 = (POOL ? 2 * ty + eh : 4 * ty + 2 * eh), ox0 = (POOL ? 2 : 4) * tx;
  const int prow0 = ((int)eb * Hy + oy0) * Wy + ox0;                   // pixel (GEMM row) index of the first of them
  const float scv = (prow0 < split ? sc0 : sc1) * (MD::pooled ? 0.25f : 1.f);   // (the halves of a paired pass are whole images)
  const int ybytes = (int)((unsigned)Mout * (unsigned)g.Co * 4u);
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(ydst, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.residual), 0, hr && !a.res_up ? ybytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask_src), 0, hm ? ybytes : 0, 0x00020000);
  const int pixb = g.Co * 4, rowb = Wy * pixb;                        // bytes to the next pixel / the next image row
#ifdef DIAGAN_W4_STAMP
  unsigned eacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long elast = t_epi;
  auto etick = [&](int i) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    eacc[i] += (unsigned)(now - elast);
    elast = now;
  };
#define W4_ETICK(i) etick(i)
#else
#define W4_ETICK(i)
#endif
  float* ss = smem;
  f32x4 cs1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, cs2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // index of frequency (i, j) in the exchange image [.][32 tiles][32 channels] (pooled modes: the 25 live ones)
  auto xidx = [&](int i, int j) { return MD::pooled ? 5 * (i < 2 ? i : i - 1) + (j < 2 ? j : j - 1) : 6 * i + j; };
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    W4_ETICK(p == 0 ? 0 : 9);                                  // (p = 0: the scalar set-up of the epilogue)
    if (nh == p) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int fx = MD::pooled ? w4p_start(grp) + s : 6 * (fi0 + s / 3) + fj0 + s % 3;
        if (!MD::pooled || s < w4p_count(grp)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = 8 * (e >> 2) + 4 * kh + (e & 3);
            ss[(fx * 32 + m) * 32 + fi] = acc[s][e];
          }
        }
      }
    }
#ifdef DIAGAN_W4_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    W4_ETICK(1 + 4 * p);                                       // products parked (the four owner waves; drained)
    __syncthreads();
    W4_ETICK(2 + 4 * p);                                       // barrier
    const int n = n0 + p * 32 + eq * 4;
    const bool ok = tv && n < g.Co;
    const unsigned voff = ok ? (unsigned)(prow0 * g.Co + n) * 4u : 0x80000000u;   // (nothing to store: beyond the descriptor)
    f32x4 bv = z4;
    if (!raw && a.bias && ok) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
    if (POOL) {
      // pooled row eh of the tile: (P A^T) M (P A^T)^T with P A^T = [[1, 2, 0, 3, -1, 0], [0, 2, 0, 12, -4, 1]]
      f32x4 sj[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (j == 2) continue;
        f32x4 m[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
          if (i != 2) m[i] = *reinterpret_cast<const f32x4*>(ss + (xidx(i, j) * 32 + et) * 32 + eq * 4);
        sj[j] = eh ? (2.f * m[1] + 12.f * m[3]) + (m[5] - 4.f * m[4]) : (m[0] + 2.f * m[1]) + (3.f * m[3] - m[4]);
      }
      f32x4 y2[2];
      y2[0] = (sj[0] + 2.f * sj[1]) + (3.f * sj[3] - sj[4]);
      y2[1] = (2.f * sj[1] + 12.f * sj[3]) + (sj[5] - 4.f * sj[4]);
#pragma unroll
      for (int bc = 0; bc < 2; ++bc) {
        const int soff = bc * pixb;
        f32x4 y = raw ? y2[bc] * 0.25f : y2[bc] * scv + bv;
        if (hr) {
          f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], rfloor);
          y += r;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y), ysrc, voff, soff, 0);
      }
    } else {
    // s[a][j] = sum_i A^T[a][i] M[i][j] for this thread's two rows a = 2 eh, 2 eh + 1:
    //   a = 0: m0 + (m1 + m2) + (m3 + m4)        a = 1: (m1 - m2) + 2 (m3 - m4)
    //   a = 2: (m1 + m2) + 4 (m3 + m4)           a = 3: (m1 - m2) + 8 (m3 - m4) + m5
    // (unpool: the products of frequency row / column 2 are zero and were neither computed nor exchanged)
    f32x4 sr[2][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (UNPOOL && j == 2) {
        sr[0][j] = sr[1][j] = z4;
        continue;
      }
      f32x4 m[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        m[i] = (UNPOOL && i == 2) ? z4 : *reinterpret_cast<const f32x4*>(ss + (xidx(i, j) * 32 + et) * 32 + eq * 4);
      const f32x4 pp = m[1] + m[2], qq = m[1] - m[2], rr = m[3] + m[4], tt = m[3] - m[4];
      if (eh == 0) {
        sr[0][j] = m[0] + pp + rr;
        sr[1][j] = qq + 2.f * tt;
      } else {
        sr[0][j] = pp + 4.f * rr;
        sr[1][j] = qq + 8.f * tt + m[5];
      }
    }
#pragma unroll
    for (int ar = 0; ar < 2; ++ar) {
      const f32x4* s6 = sr[ar];
      const f32x4 pp = s6[1] + s6[2], qq = s6[1] - s6[2], rr = s6[3] + s6[4], tt = s6[3] - s6[4];
      f32x4 y4[4];
      y4[0] = s6[0] + pp + rr;
      y4[1] = qq + 2.f * tt;
      y4[2] = pp + 4.f * rr;
      y4[3] = qq + 8.f * tt + s6[5];
      // half-resolution residual (GBlock shortcut): output row oy0 + ar blends half-resolution rows (jy - 1, jy) [ar = 0] or
      // (jy, jy + 1) [ar = 1], jy = oy0 / 2, and its four pixels blend columns 2 tx - 1 .. 2 tx + 2, edges clamped: eight
      // loads for the row instead of four per pixel
      f32x4 rup[8];
      if (hr && a.res_up == 2 && ok) {
        // un-pooled half-resolution residual (DBlock's pooled shortcut in the data gradient): both rows of the pair lie in
        // half-resolution row oy0 / 2, the four pixels in its columns ox0 / 2 and ox0 / 2 + 1
        const float* rb = a.residual + (((long)eb * (g.Ho >> 1) + (oy0 >> 1)) * (g.Wo >> 1) + (ox0 >> 1)) * g.Co + n;
        rup[0] = 0.25f * *reinterpret_cast<const f32x4*>(rb);
        rup[1] = 0.25f * *reinterpret_cast<const f32x4*>(rb + g.Co);
      } else if (hr && a.res_up && ok) {
        const int Hh = g.Ho >> 1, Wh = g.Wo >> 1, jy = oy0 >> 1, jx = ox0 >> 1;
        const int y0 = ar ? jy : max(jy - 1, 0), y1 = ar ? min(jy + 1, Hh - 1) : jy;
        const int xx[4] = {max(jx - 1, 0), jx, jx + 1, min(jx + 2, Wh - 1)};
        const float* rb = a.residual + ((long)eb * Hh * Wh) * g.Co + n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rup[q] = *reinterpret_cast<const f32x4*>(rb + ((long)y0 * Wh + xx[q]) * g.Co);
          rup[4 + q] = *reinterpret_cast<const f32x4*>(rb + ((long)y1 * Wh + xx[q]) * g.Co);
        }
      }
#pragma unroll
      for (int bc = 0; bc < 4; ++bc) {
        const int soff = ar * rowb + bc * pixb;
        f32x4 y = raw ? (UNPOOL ? y4[bc] * 0.25f : y4[bc]) : y4[bc] * scv + bv;
        if (hr) {
          f32x4 r;
          if (a.res_up == 2) {
            r = rup[bc >> 1];
          } else if (a.res_up) {
            // upsample2x_kernel's arithmetic: (w0 * a + w1 * b) along x inside the y blend
            const int c0 = bc == 0 ? 0 : (bc == 3 ? 2 : 1);
            const float wy0 = ar ? 0.75f : 0.25f, wx0 = (bc & 1) ? 0.75f : 0.25f, wy1 = 1.f - wy0, wx1 = 1.f - wx0;
            const f32x4 top = wx0 * rup[c0] + wx1 * rup[c0 + 1];
            const f32x4 bot = wx0 * rup[4 + c0] + wx1 * rup[4 + c0 + 1];
            r = wy0 * top + wy1 * bot;
          } else {
            r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], rfloor);
          }
          y += r;
        }
        if (hm) {
          const f32x4 mk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(msrc, voff, soff, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = mk[e] > 0.f ? y[e] : y[e] * a.mask_slope;
        }
#ifdef DIAGAN_WINO_ABLATE
        if (!W4_ON(1024)) {
          if (y[0] == 123.456f) ydst[0] = y[1];
        } else
#endif
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y), ysrc, voff, soff, 0);
        if (hs && ok) {
          cs1[p] += y;
          cs2[p] += y * y;
        }
      }
    }
    }
    W4_ETICK(3 + 4 * p);                                       // LDS reads, output transform, residual / mask loads, store issue
    __syncthreads();
    W4_ETICK(4 + 4 * p);                                       // barrier
  }
  if (hs) {
    // column sums over the workgroup's 512 pixels: lanes that differ in bit 0 (row pair) and bits 4, 5 (tile) hold the
    // same channels; then the eight waves meet in LDS, fixed order
    float* red = smem;                                                 // [8 waves][2 halves][2][32]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v1 = cs1[p][e], v2 = cs2[p][e];
        v1 += __shfl_xor(v1, 1, 64);
        v2 += __shfl_xor(v2, 1, 64);
        v1 += __shfl_xor(v1, 16, 64);
        v2 += __shfl_xor(v2, 16, 64);
        v1 += __shfl_xor(v1, 32, 64);
        v2 += __shfl_xor(v2, 32, 64);
        cs1[p][e] = v1;
        cs2[p][e] = v2;
      }
      if ((lane & 0x31) == 0) {
        *reinterpret_cast<f32x4*>(red + ((wave * 2 + p) * 2 + 0) * 32 + eq * 4) = cs1[p];
        *reinterpret_cast<f32x4*>(red + ((wave * 2 + p) * 2 + 1) * 32 + eq * 4) = cs2[p];
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, col = tid & 63, p = col >> 5, c32 = col & 31;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[((w * 2 + p) * 2 + which) * 32 + c32];
      if (n0 + col < g.Co) a.stat_partials[(long)(tile / tiles_n) * 2 * g.Co + which * g.Co + n0 + col] = t;
    }
  }
#ifdef DIAGAN_W4_STAMP
  if (a.stamps && lane == 0) {    // the epilogue up to the issue of its last store; its phases behind the K-loop stamps
    unsigned long long* o = a.stamps + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16;
    o[15] = __builtin_amdgcn_s_memtime() - t_epi;
    unsigned long long* o2 = a.stamps + ((size_t)gridDim.y * gridDim.x * 8 + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16;
    for (int i = 0; i < 10; ++i) o2[i] = eacc[i];
  }
#endif
}

template <int PRO, int MODE = 0, bool X3 = false, int LEFT = 0>
static int launch_wino4_pro(const ConvGemmArgs& a, const float* ug, hipStream_t st) {
  const int MT = a.g.B * (a.g.Ho >> 2) * (a.g.Wo >> 2);
  const int wgs = cdiv(MT, W4T) * cdiv(a.g.Co, W4N);
  // [2 V stages | U]; the epilogue's exchange image (36 or 25 frequencies x 32 tiles x 32 channels) fits inside
  const size_t lds = X3 ? (size_t)X3_LDS : (size_t)(2 * W4_VSTAGE + W4M<MODE>::U_FLOATS) * sizeof(float);
  auto kern = conv_wino4_kernel<PRO, MODE, X3, LEFT>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, lds);
  hipLaunchKernelGGL(kern, dim3(wgs, a.ksplit), dim3(512), lds, st, a, ug);
  return check_launch("conv_wino4");
}

// X3 (bf16 x 3 operands, see X3_SP): -1 = DIAGAN_WINO4_X3 / default, 0 / 1 = diagan_conv_gemm_set_wino4x
static int g_wino4x = -1;
void wino4_set_x3(int mode) { g_wino4x = mode; }
int call_opt_wino4x();                             // conv_gemm.hip: the current call's option (-1: none)
int wino4_get_x3() {
  static const int env = getenv("DIAGAN_WINO4_X3") ? atoi(getenv("DIAGAN_WINO4_X3")) : 0;
  const int c = call_opt_wino4x();
  return c >= 0 ? c : (g_wino4x >= 0 ? g_wino4x : env);
}
// floats of workspace the transformed weights need (with X3 enabled: room for the split format, 6 instead of 4 bytes per element)
static long x3_floats(long fp32_floats) { return wino4_get_x3() > 0 ? fp32_floats + fp32_floats / 2 : fp32_floats; }
long wino4_ws_floats(int Co, int Ci) { return x3_floats((long)cdiv(Co, W4N) * W4N * Ci * 36); }
// the X3 kernel takes a launch whose K loop is a multiple of four steps per channel split (Ci % 32 == 0)
static bool wino4_x3_ok(const ConvGemmArgs& a) {
  const int nk = a.g.Ci / W4K, ks = a.ksplit > 0 ? a.ksplit : 1;
  return wino4_get_x3() > 0 && (a.g.Ci % 32) == 0 && nk % ks == 0 && (nk / ks) % 4 == 0;
}

// transformed weights of this launch (weight-kernel mode WM: 0 = all 36 frequencies, 1 = the pooled modes' 25): the caller's
// ready-made buffer if it hinted this format (diagan_conv_gemm_weights_hint), else `ws` after the per-launch transform
template <int WM>
static const float* wino4_weights(const ConvGemmArgs& a, float* ws, int flip, float scale, long floats, hipStream_t st, bool x3 = false) {
  const int kind = x3 ? (WM ? WK_F4X_POOL : WK_F4X) : (WM ? WK_F4_POOL : WK_F4);
  if (const float* ready = wino_weights_ready(kind, flip, scale, floats)) return ready;
  const ConvGeom& g = a.g;
  if (x3)
    hipLaunchKernelGGL(wino4x_weight_kernel<WM>, dim3(cdiv(g.Ci, 32), cdiv(g.Co, W4N)), dim3(512), 0, st, a.w, ws, g.Co, g.Ci, g.Kp, flip,
                       scale);
  else
  hipLaunchKernelGGL(wino4_weight_kernel<WM>, dim3(cdiv(g.Ci, 32), cdiv(g.Co, W4N)), dim3(512), 0, st, a.w, ws, g.Co, g.Ci, g.Kp, flip,
                     scale);
  return ws;
}

int launch_wino_weights_batched(const WinoJob* jobs, int n, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(wino_weights_batched_kernel, dim3(blocks), dim3(512), 0, st, jobs, n);
  return check_launch("wino_weights_batched");
}

// geometry the F(4x4,3x3) kernel takes on top of diagan_conv_wino_supported: H and W multiples of 4
bool wino4_geom_ok(int Ho, int Wo, int Ci) { return !(Ho & 3) && !(Wo & 3) && (Ci & 7) == 0; }

// Launch-size policy (tools/wino4_policy.py on the SNGAN-32 / SNGAN-64 layer shapes, r3): channel splits for this launch, 0 =
// leave it to the F(2x2) kernel / the implicit GEMM.  One 512-thread workgroup per CU, so what counts is how well the
// workgroup count fills rounds of 256: >= 192 workgroups run 1.15-1.30x the F(2x2) kernel when both fill their last round
// equally well (measured 1.27 x the ratio of the two fill factors: 384 workgroups = 1.5 rounds against F(2x2)'s exact 3
// rounds came out at 0.96x); 96-191 workgroups with >= 32 K-steps split two ways (1.04-1.14x); below that the split-K
// F(2x2) / implicit-GEMM launches win.
static int wino4_ksplit_impl(int B, int Ho, int Wo, int Ci, int Co, int allow_split, long ws_floats, double bar);
int wino4_ksplit(int B, int Ho, int Wo, int Ci, int Co, int allow_split, long ws_floats) {
  return wino4_ksplit_impl(B, Ho, Wo, Ci, Co, allow_split, ws_floats, 1.05);
}
// bar: how far ahead of the F(2x2) kernel the model must put this kernel before it is taken (1.05 for the plain convolution)
static int wino4_ksplit_impl(int B, int Ho, int Wo, int Ci, int Co, int allow_split, long ws_floats, double bar) {
  const long wgs4 = (long)cdiv((long)B * (Ho >> 2) * (Wo >> 2), W4T) * cdiv(Co, W4N);
  const long wgs2 = (long)cdiv((long)B * (Ho >> 1) * (Wo >> 1), 64) * cdiv(Co, 64);
  auto fill = [](long w) { return (double)w / (256.0 * ((w + 255) / 256)); };
  if (wgs4 >= 192) {
    // a launch of 1 - 2 rounds whose last round is badly filled, with a long K loop (>= 32 K-steps per half): split two ways
    // when that fills the rounds better (0.93: the slab and the second-stage kernel).  SNGAN-64's stacked generator forward at
    // 8x8 (384 workgroups = 1.5 rounds, Ci = 1024): 852 us on the F(2x2) kernel's exact 3 rounds.
    const bool can2 = allow_split && wgs4 <= 512 && Ci >= 512 && wino4_ws_floats(Co, Ci) + 2L * B * Ho * Wo * Co <= ws_floats;
    const double f1 = fill(wgs4), f2 = can2 ? 0.93 * fill(2 * wgs4) : 0.0;
    const double f = f2 > f1 ? f2 : f1;
    if (1.27 * f / (wgs2 >= 192 ? fill(wgs2) : 1.0) < bar) return 0;
    return f2 > f1 ? 2 : 1;
  }
  if (allow_split && wgs4 * 2 >= 192 && Ci >= 256 && wino4_ws_floats(Co, Ci) + 2L * B * Ho * Wo * Co <= ws_floats) return 2;
  return 0;
}

// `a` as prepared by diagan_conv_gemm (dWo / dHo re-made here for the TILE grid); ws: wino4_ws_floats(Co, Ci) floats
int launch_wino4(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 2));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 2));
  const long f32 = (long)cdiv(g.Co, W4N) * W4N * g.Ci * 36;
  if (wino4_x3_ok(a)) {
    const float* ug = wino4_weights<0>(a, ws, g.dr < 0 ? 1 : 0, 1.f, f32 + f32 / 2, st, true);
    switch (a.pro_mode) {
      case PRO_NONE: return launch_wino4_pro<PRO_NONE, 0, true>(a, ug, st);
      case PRO_RELU: return launch_wino4_pro<PRO_RELU, 0, true>(a, ug, st);
      case PRO_AFFINE_RELU: return launch_wino4_pro<PRO_AFFINE_RELU, 0, true>(a, ug, st);
      case PRO_LRELU: return launch_wino4_pro<PRO_LRELU, 0, true>(a, ug, st);
      default: return launch_wino4_pro<PRO_AFFINE, 0, true>(a, ug, st);
    }
  }
  const float* ug = wino4_weights<0>(a, ws, g.dr < 0 ? 1 : 0, 1.f, f32, st);
  switch (a.pro_mode) {
    case PRO_NONE: return launch_wino4_pro<PRO_NONE>(a, ug, st);
    case PRO_RELU: return launch_wino4_pro<PRO_RELU>(a, ug, st);
    case PRO_AFFINE_RELU: return launch_wino4_pro<PRO_AFFINE_RELU>(a, ug, st);
    case PRO_LRELU: return launch_wino4_pro<PRO_LRELU>(a, ug, st);
    default: return launch_wino4_pro<PRO_AFFINE>(a, ug, st);
  }
}

// tile_cfg 11 / 12 on the F(4x4) kernel (MODE 1 / 2): convolution + 2x2 average pool in 25 products per 4x4 tile, and its data
// gradient from the pooled gradient.  Same arguments as launch_wino_pool / launch_wino_unpool (conv_wino_pool.hip).
// Taken where the launch fills the chip (>= 192 workgroups of 32 tiles x 64 channels, no channel split) and H, W are
// multiples of 4; DIAGAN_WINO4_POOL=0 keeps the F(2x2) pooled kernels; force: any launch size (diagan_conv_gemm_set_wino4(2), tests).
bool wino4_pool_ok(int B, int Ho, int Wo, int Ci, int Co, long ws_floats, bool force) {
  static const int env = getenv("DIAGAN_WINO4_POOL") ? atoi(getenv("DIAGAN_WINO4_POOL")) : 1;
  static const int w4 = getenv("DIAGAN_WINO4") ? atoi(getenv("DIAGAN_WINO4")) : 1;
  static const int min_wgs = getenv("DIAGAN_WINO4_POOL_MIN_WGS") ? atoi(getenv("DIAGAN_WINO4_POOL_MIN_WGS")) : 192;
  if (!env || !w4 || !wino4_geom_ok(Ho, Wo, Ci) || (Co & 3) || Ci < 32) return false;
  const long wgs = (long)cdiv((long)B * (Ho >> 2) * (Wo >> 2), W4T) * cdiv(Co, W4N);
  return (force || wgs >= min_wgs) && x3_floats((long)cdiv(Co, W4N) * W4N * Ci * 28) <= ws_floats;      // 56 units x 256 floats per 8 channels and column block
}

int launch_wino4_pool(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 2));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 2));
  const float* ug = wino4_weights<1>(a, ws, 0, 1.f, (long)cdiv(g.Co, W4N) * W4N * g.Ci * 28, st);
#if W4_ULOAD && W4_ULOAD_POOLED
  const int left = (g.Ci / W4K) % 3;          // (no channel split in the pooled modes: the K loop is Ci / 8 steps)
  if (left == 1) return a.pro_mode == PRO_RELU ? launch_wino4_pro<PRO_RELU, 1, false, 1>(a, ug, st) : launch_wino4_pro<PRO_NONE, 1, false, 1>(a, ug, st);
  if (left == 2) return a.pro_mode == PRO_RELU ? launch_wino4_pro<PRO_RELU, 1, false, 2>(a, ug, st) : launch_wino4_pro<PRO_NONE, 1, false, 2>(a, ug, st);
#endif
  return a.pro_mode == PRO_RELU ? launch_wino4_pro<PRO_RELU, 1>(a, ug, st) : launch_wino4_pro<PRO_NONE, 1>(a, ug, st);
}

int launch_wino4_unpool(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 2));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 2));
  const float* ug = wino4_weights<1>(a, ws, 1, 1.f, (long)cdiv(g.Co, W4N) * W4N * g.Ci * 28, st);
#if W4_ULOAD && W4_ULOAD_POOLED
  const int left = (g.Ci / W4K) % 3;
  if (left == 1) return launch_wino4_pro<PRO_NONE, 2, false, 1>(a, ug, st);
  if (left == 2) return launch_wino4_pro<PRO_NONE, 2, false, 2>(a, ug, st);
#endif
  return launch_wino4_pro<PRO_NONE, 2>(a, ug, st);
}

// tile_cfg 15 on the F(4x4) kernel (MODE 3): conv3x3(bilinear x2 (pro(x))) from the HALF-resolution input -- GBlock's
// BN -> ReLU -> up-sampling -> c1 in one launch.  a.g.Hi / Wi are the up-sampled sizes (= Ho / Wo), a.x is [B, Hi/2, Wi/2, Ci].
// Taken where the plain F(4x4) launch of the same convolution would be (wino4_ksplit == 1: >= 192 workgroups that fill their
// rounds at least as well as the F(2x2) kernel's); DIAGAN_WINO4_UPIN=0: off; force: any launch size (tests).
bool wino4_upin_ok(int B, int Ho, int Wo, int Ci, int Co, long ws_floats, bool force) {
  static const int env = getenv("DIAGAN_WINO4_UPIN") ? atoi(getenv("DIAGAN_WINO4_UPIN")) : 1;
  static const int w4 = getenv("DIAGAN_WINO4") ? atoi(getenv("DIAGAN_WINO4")) : 1;
  if (!env || !w4 || !wino4_geom_ok(Ho, Wo, Ci) || (Co & 3) || wino4_ws_floats(Co, Ci) > ws_floats) return false;
  // (the alternative is the F(2x2) kernel AFTER a separate up-sampling pass over a 4x larger tensor, and this mode's lighter
  //  loader runs 1.1x the plain F(4x4) convolution: the bar against F(2x2) is lower than for the plain convolution --
  //  SNGAN-64's stacked block2.c1, 384 workgroups = 1.5 rounds against F(2x2)'s exact 3, qualifies)
  return force || (Ci >= 64 && wino4_ksplit_impl(B, Ho, Wo, Ci, Co, 0, ws_floats, 0.93) == 1);
}

int launch_wino4_upin(ConvGemmArgs a, float* ws, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 2));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 2));
  const long f32 = (long)cdiv(g.Co, W4N) * W4N * g.Ci * 36;
  if (wino4_x3_ok(a)) {
    const float* ug = wino4_weights<0>(a, ws, 0, 0.0625f, f32 + f32 / 2, st, true);
    switch (a.pro_mode) {
      case PRO_NONE: return launch_wino4_pro<PRO_NONE, 3, true>(a, ug, st);
      case PRO_RELU: return launch_wino4_pro<PRO_RELU, 3, true>(a, ug, st);
      case PRO_AFFINE_RELU: return launch_wino4_pro<PRO_AFFINE_RELU, 3, true>(a, ug, st);
      case PRO_LRELU: return launch_wino4_pro<PRO_LRELU, 3, true>(a, ug, st);
      default: return launch_wino4_pro<PRO_AFFINE, 3, true>(a, ug, st);
    }
  }
  const float* ug = wino4_weights<0>(a, ws, 0, 0.0625f, f32, st);
  switch (a.pro_mode) {
    case PRO_NONE: return launch_wino4_pro<PRO_NONE, 3>(a, ug, st);
    case PRO_RELU: return launch_wino4_pro<PRO_RELU, 3>(a, ug, st);
    case PRO_AFFINE_RELU: return launch_wino4_pro<PRO_AFFINE_RELU, 3>(a, ug, st);
    case PRO_LRELU: return launch_wino4_pro<PRO_LRELU, 3>(a, ug, st);
    default: return launch_wino4_pro<PRO_AFFINE, 3>(a, ug, st);
  }
}

}  // namespace diagan
