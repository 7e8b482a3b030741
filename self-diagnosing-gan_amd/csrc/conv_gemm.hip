// Implicit-GEMM convolution, forward and data-gradient, on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s dense peak on MI355X).
//
// Replaces the ATen/cuDNN calls under F.conv2d / nn.ConvTranspose2d and their input-gradient
// in the SNGAN / DCGAN stacks (e.g. torch_mimicry GBlock/DBlock convs
// invoked from diagan-pkg/diagan/models/predefined_models.py:19-21,38-40,57-59,76-78, and
// diagan-pkg/diagan/models/mnist.py:55-71,163-190).
//
//   Y[m][n] = epi( sum_k A(m,k) * Wp[n][k] ),   m = (b,oy,ox) pixel, n = out channel,
//   k = (r,s,c);  A(m,k) = pro(X[b, iy(oy,r), ix(ox,s), c]) or 0 outside the image.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32; each wave owns a (TM*32) x (TN*32)
// sub-tile as TM x TN MFMA 32x32 accumulators.  A and B tiles are register-staged
// (global_load_dwordx4 -> prologue in VGPRs -> ds_write_b128) into double-buffered LDS with a
// 16-byte-chunk XOR swizzle so that the ds_read_b128 fragment reads are bank-conflict free; one
// s_barrier per K-step.  The im2col gather happens in the loader: no column matrix exists in HBM.
// Prologue (ReLU / BatchNorm-apply+ReLU / LeakyReLU) is applied to the gathered values on their way
// to LDS; epilogue adds bias, a residual tensor and/or a ReLU-backward mask before the store.
//
// Roofline: MFMA fp32 (algorithmic FLOP = 2*M*Co*K).
#include "conv_common.h"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

extern "C" int diagan_conv_gemm_tile_rows(int cfg);
extern "C" int diagan_conv_gemm_tile_cols(int cfg);
extern "C" int diagan_conv_gemm_pick_ksplit(int M, int Co, int Kp, int cfg);
extern "C" int diagan_conv_gemm_pick_cfg_geom(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                              int dr, int off, int up, int Kp, int allow_split, int64_t ws_floats);
extern "C" int diagan_conv_gemm_pick_cfg_grouped(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                                 int dr, int off, int up, int Kp, int allow_split, int64_t ws_floats,
                                                 int pro_group_rows);
extern "C" int diagan_conv_wino_supported(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                          int off, int up);

namespace diagan {
int launch_gemm_x3(const ConvGemmArgs& a, float* ws, hipStream_t st);     // conv_gemm_x3.hip: bf16 pipe, exactly split operands
bool gemm_x3_geom_ok(const ConvGemmArgs& a);
long gemm_x3_ws_floats(int Co, int Kp);
int launch_gemm_x3b(const ConvGemmArgs& a, const OutMap& map, float* ws, hipStream_t st);    // conv_gemm_x3b.hip: the same arithmetic on 128 x 128 tiles
bool gemm_x3b_geom_ok(const ConvGemmArgs& a);
long gemm_x3b_ws_floats(int Co, int Kp);
void gemm_x3b_force_form(int form);
int launch_wino(ConvGemmArgs a, float* ws, hipStream_t st);      // conv_wino.hip
long wino_ws_floats(int Co, int Ci);
int wino_ksplit(int B, int Ho, int Wo, int Ci, int Co, int allow_split, long ws_floats, int min_wgs);
int launch_wino4(ConvGemmArgs a, float* ws, hipStream_t st);     // conv_wino4.hip: F(4x4,3x3)
long wino4_ws_floats(int Co, int Ci);
bool wino4_geom_ok(int Ho, int Wo, int Ci);
int wino4_ksplit(int B, int Ho, int Wo, int Ci, int Co, int allow_split, long ws_floats);
bool wino4_pool_ok(int B, int Ho, int Wo, int Ci, int Co, long ws_floats, bool force);
int launch_wino4_pool(ConvGemmArgs a, float* ws, hipStream_t st);
int launch_wino4_unpool(ConvGemmArgs a, float* ws, hipStream_t st);
bool wino4_upin_ok(int B, int Ho, int Wo, int Ci, int Co, long ws_floats, bool force);
int launch_wino_weights_batched(const WinoJob* jobs, int n, int blocks, hipStream_t st);     // conv_wino4.hip (all formats)
int launch_wino4_upin(ConvGemmArgs a, float* ws, hipStream_t st);
void wino4_set_x3(int mode);
int wino4_get_x3();
int launch_wino_pool(ConvGemmArgs a, float* ws, hipStream_t st); // conv_wino_pool.hip
int launch_wino_unpool(ConvGemmArgs a, float* ws, hipStream_t st);
int wino_pool_ksplit(int B, int Ho, int Wo, int Ci, int Co, long slab_floats, int min_wgs);


// KG = 2 (tile_cfg 14): TWO K-groups of four waves in one workgroup.  Each group is the KG = 1 kernel on its own half of the
// K-steps with its own LDS stages; the second group's accumulators join the first's through LDS before the epilogue.  For
// launches with at most one 64x64 tile per CU (the 8x8 / 4x4 blocks): a lone four-wave workgroup leaves every SIMD with ONE
// wave and runs at 0.58 of the MFMA rate; split-K LAUNCHES pay for a slab and a second kernel (31.6 us unsplit, 36 split at
// M = 8192, N = 128, K = 1152).  Measured: 29-32 -> 26-29 us on that shape (HIP events around the launch); FOUR groups (nine K-steps
// each) take the same 27 us -- what is left is the launch's fixed cost, not its K loop -- so KG = 2 is the only instantiation.
template <int BM, int BN, int WM, int WN, int BK = 32, int PRO = -1, bool STAMP = false, bool FP = false, int KG = 1>
__global__ __launch_bounds__(256 * KG) void conv_gemm_kernel(const ConvGemmArgs a) {
  constexpr int CH = BK / 4;                           // 16-byte chunks per tile row
  constexpr int RP = 256 / CH;                         // tile rows covered by one pass of the 256 loaders
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;  // MFMA tiles per wave
  constexpr int AJ = BM / RP, BJ = BN / RP;            // 16-byte chunks per thread per tile
  // XOR swizzle of the chunk index: conflict-free ds_read_b128 fragments (bank row = 256 B)
  // (BK = 64: a tile row is a whole 256-byte bank row, so a 16-lane read group needs 16 distinct slots: q ^ (row & 15))
  auto swz = [](int row, int q) {
    return BK == 64 ? (q ^ (row & 15)) : BK == 32 ? (q ^ ((row >> 1) & 7)) : (q ^ ((row >> 2) & 3));
  };
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem_all[];
  constexpr int STAGE_FLOATS = 2 * (BM + BN) * BK;                       // both stages of A and B of one K-group
  const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  float* const smem = smem_all + kg * STAGE_FLOATS;
  float* As = smem;                  // [2][BM*32]
  float* Bs = smem + 2 * BM * BK;    // [2][BN*32]

  // Diagnostic build (STAMP, reached only through diagan_conv_gemm_set_stamp_buffer): lane 0 of wave 0 records the
  // shader clock at the phase boundaries of the workgroup plus the constant 100 MHz real-time counter at entry and
  // exit and the hardware id (XCD, CU), into a buffer of its own that nothing else reads.
  unsigned long long st_t[6] = {0, 0, 0, 0, 0, 0}, st_r0 = 0;
  auto stamp = [&](int i) {
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      st_t[i] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (STAMP) st_r0 = __builtin_amdgcn_s_memrealtime();
  if (a.tune & 1) __builtin_amdgcn_s_setprio(3);
  stamp(0);
  const ConvGeom& g = a.g;
  const int pro_mode = PRO >= 0 ? PRO : a.pro_mode;   // compile-time in the specialised kernels: straight-line store phase
  const int tid = KG > 1 ? (threadIdx.x & 255) : threadIdx.x, lane = tid & 63, wave = tid >> 6;   // (roles inside the K-group)
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (g.Co + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

  // ---- loader state -------------------------------------------------------------------------
  // The gather is made separable once per workgroup: per tile row j a byte offset base[j] of the
  // (r=0,s=0) tap and a validity mask (bit r: tap row r inside the image, bit 16+s: tap column s),
  // per K-step one tap offset shared by all rows.  For up=2 (transposed conv) the offset is kept in
  // half-pixel units: a VALID tap has an even numerator, so base + tapoff is exact there, and
  // invalid taps are redirected past num_records (the hardware returns zeros).
  const int lrow = tid / CH, lq = tid % CH;  // row within an RP-row group, 16-byte chunk in the K-step
  const int ush = g.up >> 1, upm = g.up - 1;  // up in {1,2}
  const int pstep = (g.Ci * 4) >> ush;        // bytes per numerator unit along x
  int base[AJ];
  unsigned vmask[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + lrow + RP * j;
    base[j] = 0;
    vmask[j] = 0;
    if (m < a.M) {
      const unsigned t = fdiv((unsigned)m, a.dWo);
      const int ox = m - (int)t * g.Wo;
      const unsigned b = fdiv(t, a.dHo);
      const int oy = (int)t - (int)b * g.Ho;
      const int iy0 = oy * g.sy + g.off, ix0 = ox * g.sy + g.off;
      base[j] = (int)b * g.Hi * g.Wi * g.Ci * 4 + (iy0 * g.Wi + ix0) * pstep;
      unsigned mk = 0;
      if (g.up == 1) {
        // closed form: taps r with 0 <= iy0 + r*dr < Hi form one interval [lo, hi]
        const int ylo = g.dr > 0 ? max(0, -iy0) : max(0, iy0 - g.Hi + 1);
        const int yhi = g.dr > 0 ? min(g.R - 1, g.Hi - 1 - iy0) : min(g.R - 1, iy0);
        const int xlo = g.dr > 0 ? max(0, -ix0) : max(0, ix0 - g.Wi + 1);
        const int xhi = g.dr > 0 ? min(g.S - 1, g.Wi - 1 - ix0) : min(g.S - 1, ix0);
        if (yhi >= ylo) mk |= (2u << yhi) - (1u << ylo);
        if (xhi >= xlo) mk |= ((2u << xhi) - (1u << xlo)) << 16;
      } else {
        for (int r = 0; r < g.R; ++r) {
          const int yn = iy0 + r * g.dr;
          if (yn >= 0 && (yn & upm) == 0 && (yn >> ush) < g.Hi) mk |= 1u << r;
        }
        for (int q = 0; q < g.S; ++q) {
          const int xn = ix0 + q * g.dr;
          if (xn >= 0 && (xn & upm) == 0 && (xn >> ush) < g.Wi) mk |= 0x10000u << q;
        }
      }
      vmask[j] = mk;
    }
  }
  // K-step range of this workgroup (split-K over gridDim.y for problems with few output tiles)
  // (K-group 0 gets the first and never the shorter range of its workgroup: it is the one that waits for the other)
  const int nk_all = g.Kp / BK;
  const int k_per = (nk_all + a.ksplit * KG - 1) / (a.ksplit * KG);
  const int k_begin = min((int)(blockIdx.y * KG + kg) * k_per, nk_all), k_end = min(k_begin + k_per, nk_all);
  // (tap, c) of this thread's chunk, advanced incrementally by BK channels per K-step
  int kc, kr, ks;
  {
    const int kflat = k_begin * BK + lq * 4, tap = kflat / g.Ci;
    kc = kflat - tap * g.Ci;
    kr = tap / g.S;
    ks = tap - kr * g.S;
  }
  const int rowstep = g.dr * g.Wi * pstep, colstep = g.dr * pstep;

  // Loads are branch-free raw buffer loads.  The prologue transform is applied when the registers are
  // written to LDS, i.e. AFTER the MFMAs of the current step, so the global loads stay in flight under
  // the matrix work instead of being waited for one by one.
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)((unsigned)g.B * g.Hi * g.Wi * g.Ci * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.w), 0, (int)((unsigned)g.Co * g.Kp * 4u), 0x00020000);
  const bool affine = pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE;
  const int pro_group_off = a.pro_group_rows > 0 ? (m0 / a.pro_group_rows) * g.Ci : 0;   // a tile never straddles groups
  unsigned wbase[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) wbase[j] = ((unsigned)(n0 + lrow + RP * j) * g.Kp + lq * 4) * 4u;

  f32x4 ra[AJ], rb[BJ], psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  unsigned a_ok = 0;

  // Uniform-tap path (Ci % BK == 0): a K-step then lies inside ONE tap for every lane, so the
  // tap walk (ukr, uks, ukc) is wave-uniform scalar work, a row's gather offset avo[j] (bit 31 = outside the image) changes
  // only when the tap does, and the K-step's channel / weight-column offsets ride in the loads' scalar offset.  Vector
  // instructions do not hide behind MFMAs on this hardware (profiles/r02_wino_ablation.md): the general walk below costs
  // ~45 of them per K-step, this one none between tap changes.
  const bool ut = (g.Ci % BK) == 0;
  int ukr = 0, uks = 0, ukc = 0;
  unsigned avo[AJ], tap_ok = 0;
  auto tap_setup = [&]() {
    const int toff = ukr * rowstep + uks * colstep + lq * 16;
    const int r_ = min(ukr, 15), s_ = 16 + uks;     // K-padding taps (ukr >= R) hit a zero mask bit
    tap_ok = 0;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const unsigned okb = (vmask[j] >> r_) & (vmask[j] >> s_) & 1u;
      avo[j] = (unsigned)(base[j] + toff) | ((okb ^ 1u) << 31);
      tap_ok |= okb << j;
    }
  };
  if (ut) {
    const int kflat0 = k_begin * BK, tap0 = kflat0 / g.Ci;
    ukc = kflat0 - tap0 * g.Ci;
    ukr = tap0 / g.S;
    uks = tap0 - ukr * g.S;
    tap_setup();
  }
  const float* const pro_sc_lane = a.pro_scale ? a.pro_scale + pro_group_off + lq * 4 : nullptr;
  const float* const pro_sh_lane = a.pro_shift ? a.pro_shift + pro_group_off + lq * 4 : nullptr;

  // one 16-byte global load of the next tile (p < AJ: gathered activation rows; else weight rows)
  int tapoff = 0, krs = 0, kss = 0;
  auto load_piece = [&](int kk, int p, auto utc) {
    if constexpr (decltype(utc)::value) {
      if (p == 0) {
        a_ok = tap_ok;
        if (affine) {
          psc = *reinterpret_cast<const f32x4*>(pro_sc_lane + ukc);
          psh = *reinterpret_cast<const f32x4*>(pro_sh_lane + ukc);
        }
      }
      if (p < AJ) {
        ra[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, avo[p], ukc * 4, 0));
      } else {
        rb[p - AJ] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wbase[p - AJ], kk * (BK * 4), 0));
      }
      if (p == AJ + BJ - 1) {            // advance the (uniform) tap walk to the next K-step
        ukc += BK;
        if (ukc >= g.Ci) {
          ukc = 0;
          if (++uks == g.S) {
            uks = 0;
            ++ukr;
          }
          tap_setup();
        }
      }
      return;
    }
    if (p == 0) {
      tapoff = kr * rowstep + ks * colstep + kc * 4;
      krs = min(kr, 15);               // K-padding taps (kr >= R) hit a zero mask bit
      kss = 16 + ks;
      a_ok = 0;
      if (affine) {
        const int kcs = min(kc, g.Ci - 4);       // kc < Ci always; keeps the address in range for the optimiser
        psc = *reinterpret_cast<const f32x4*>(a.pro_scale + pro_group_off + kcs);
        psh = *reinterpret_cast<const f32x4*>(a.pro_shift + pro_group_off + kcs);
      }
    }
    if (p < AJ) {
      const int j = p;
      const unsigned okb = (vmask[j] >> krs) & (vmask[j] >> kss) & 1u;
      // invalid -> bit 31 set: beyond num_records (< 2 GiB) whatever the garbage below it
      const unsigned off = (unsigned)(base[j] + tapoff) | ((okb ^ 1u) << 31);
      ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0));
      a_ok |= okb << j;
    } else {
      const int j = p - AJ;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wbase[j] + (unsigned)kk * (BK * 4u), 0, 0));
    }
    if (p == AJ + BJ - 1) {            // advance (tap, c) to the next K-step
      kc += BK;
      if (g.Ci >= BK) {                // at most one tap boundary per K-step: straight-line selects, no lane-dependent loop
        const bool w = kc >= g.Ci;
        kc -= w ? g.Ci : 0;
        ks += w ? 1 : 0;
        const bool r = ks == g.S;
        ks = r ? 0 : ks;
        kr += r ? 1 : 0;
      } else {                         // RGB-sized inputs (Ci = 4: two K-steps in all): plain divisions
        const int kflat = (kk + 1) * BK + lq * 4, tap = kflat / g.Ci;
        kc = kflat - tap * g.Ci;
        kr = tap / g.S;
        ks = tap - kr * g.S;
      }
    }
  };
  auto load_tiles = [&](int kk) {
    if (ut) {
#pragma unroll
      for (int p = 0; p < AJ + BJ; ++p) load_piece(kk, p, std::true_type{});
    } else {
#pragma unroll
      for (int p = 0; p < AJ + BJ; ++p) load_piece(kk, p, std::false_type{});
    }
  };
  // one 16-byte piece of the next tile: prologue on its way from the staging registers to LDS
  auto store_piece = [&](int buf, int p) {
    if (p < AJ) {
      const int j = p;
      const int row = lrow + RP * j;
      f32x4 v = ra[j];
      if (pro_mode != PRO_NONE) {
        if (affine) v = v * psc + psh;
        if (pro_mode == PRO_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
        } else if (pro_mode != PRO_AFFINE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (affine) {                            // padding is zero AFTER the transform
          const float keep = (float)((a_ok >> j) & 1u);
          v *= keep;
        }
      }
      *reinterpret_cast<f32x4*>(As + buf * BM * BK + row * BK + (swz(row, lq) << 2)) = v;
    } else {
      const int j = p - AJ;
      const int row = lrow + RP * j;
      *reinterpret_cast<f32x4*>(Bs + buf * BN * BK + row * BK + (swz(row, lq) << 2)) = rb[j];
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
  // uniform-tap path: LDS addresses of the fragments of sub-step u and of the staged pieces, per buffer, held in registers
  // (with the buffer a compile-time constant of the step, see kstep_u: no address arithmetic inside the loop).  The swizzle
  // of a row only depends on the row modulo 32 (16 for BK = 64), so tiles i / j and pieces p differ by constant offsets.
  int fao[BK / 8], fbo[BK / 8];
#pragma unroll
  for (int u = 0; u < BK / 8; ++u) {
    const int ra_ = wm * (TM * 32) + fi, rb_ = wn * (TN * 32) + fi;
    fao[u] = ra_ * BK + (swz(ra_, 2 * u + fh) << 2);
    fbo[u] = 2 * BM * BK + rb_ * BK + (swz(rb_, 2 * u + fh) << 2);
  }
  const int sto = lrow * BK + (swz(lrow, lq) << 2);      // this thread's slot in a tile (A and B alike)

  stamp(1);
  if (k_begin < k_end) {
    load_tiles(k_begin);
    store_tiles(0);
  }
  __syncthreads();

  // One K-step: the next tile's global loads are issued first; its AJ+BJ pieces are then written to the OTHER
  // LDS buffer one per e-step during the LAST e-steps of this tile's MFMAs: an MFMA occupies the matrix pipe for
  // 64 cycles after it issues, so the ~15 VALU / LDS-write instructions of a piece ride in its shadow instead of
  // forming a store phase after the MFMAs during which the pipe has nothing from this workgroup.
  constexpr int NE = (BK / 8) * 4, NP = AJ + BJ;
  constexpr int PPE = (2 * NP + NE - 1) / NE;      // pieces per e-step (1 for the square tiles; 2 for 256x64)
  constexpr int LE = (NP + PPE - 1) / PPE;         // e-steps that carry load pieces (first) / store pieces (last)
  static_assert(2 * LE <= NE, "load and store pieces of a tile must not share an e-step");
  // FP (one 32x32 accumulator per wave): the fragments of sub-step u + 1 are read into a second register set BEFORE
  // the MFMAs of sub-step u issue -- with a single accumulator chain the compiler otherwise re-uses the fragment
  // registers and every sub-step starts with an exposed LDS round trip (ds_read, lgkmcnt(0), 4 MFMAs, ds_read, ...).
  auto kstep = [&](int kk, auto has_next) {
    const int cur = (kk - k_begin) & 1;
    const float* Ac = As + cur * BM * BK;
    const float* Bc = Bs + cur * BN * BK;
    f32x4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int slot, int u) {
      const int q = 2 * u + fh;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 32) + i * 32 + fi;
        fa[slot][i] = *reinterpret_cast<const f32x4*>(Ac + row * BK + (swz(row, q) << 2));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + fi;
        fb[slot][j] = *reinterpret_cast<const f32x4*>(Bc + row * BK + (swz(row, q) << 2));
      }
    };
    if constexpr (FP) read_frags(0, 0);
#pragma unroll
    for (int u = 0; u < BK / 8; ++u) {
      const int fs = FP ? (u & 1) : 0;
      if constexpr (FP) {
        if (u + 1 < BK / 8) read_frags((u + 1) & 1, u + 1);
      } else {
        read_frags(0, u);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int es = u * 4 + e;
        if (decltype(has_next)::value && es < LE) {                          // next tile's global loads: first e-steps
#pragma unroll
          for (int pp = 0; pp < PPE; ++pp)
            if (es * PPE + pp < NP) load_piece(kk + 1, es * PPE + pp, std::false_type{});
        }
        if (decltype(has_next)::value && es >= NE - LE) {
          __builtin_amdgcn_sched_barrier(0);   // keep the piece HERE: hoisted to the top it would wait for its load first
#pragma unroll
          for (int pp = 0; pp < PPE; ++pp)
            if ((es - (NE - LE)) * PPE + pp < NP) store_piece(cur ^ 1, (es - (NE - LE)) * PPE + pp);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fs][i][e], fb[fs][j][e], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  };
  // The same step on the uniform-tap path: buffer index as a compile-time constant (the loop below is unrolled by two), all
  // LDS addresses = a register + an instruction offset.
  auto kstep_u = [&](int kk, auto curc, auto has_next) {
    constexpr int cur = decltype(curc)::value;
    f32x4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int slot, int u) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const f32x4*>(smem + fao[u] + cur * BM * BK + i * 32 * BK);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const f32x4*>(smem + fbo[u] + cur * BN * BK + j * 32 * BK);
    };
    auto store_piece_u = [&](int p) {          // store_piece(cur ^ 1, p) with constant offsets
      if (p < AJ) {
        f32x4 v = ra[p];
        if (pro_mode != PRO_NONE) {
          if (affine) v = v * psc + psh;
          if (pro_mode == PRO_LRELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
          } else if (pro_mode != PRO_AFFINE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (affine) {                            // padding is zero AFTER the transform
            const float keep = (float)((a_ok >> p) & 1u);
            v *= keep;
          }
        }
        *reinterpret_cast<f32x4*>(smem + sto + (cur ^ 1) * BM * BK + p * RP * BK) = v;
      } else {
        *reinterpret_cast<f32x4*>(smem + sto + 2 * BM * BK + (cur ^ 1) * BN * BK + (p - AJ) * RP * BK) = rb[p - AJ];
      }
    };
    if constexpr (FP) read_frags(0, 0);
#pragma unroll
    for (int u = 0; u < BK / 8; ++u) {
      const int fs = FP ? (u & 1) : 0;
      if constexpr (FP) {
        if (u + 1 < BK / 8) read_frags((u + 1) & 1, u + 1);
      } else {
        read_frags(0, u);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int es = u * 4 + e;
        if (decltype(has_next)::value && es < LE) {                          // next tile's global loads: first e-steps
#pragma unroll
          for (int pp = 0; pp < PPE; ++pp)
            if (es * PPE + pp < NP) load_piece(kk + 1, es * PPE + pp, std::true_type{});
        }
        if (decltype(has_next)::value && es >= NE - LE) {
          __builtin_amdgcn_sched_barrier(0);   // keep the piece HERE: hoisted to the top it would wait for its load first
#pragma unroll
          for (int pp = 0; pp < PPE; ++pp)
            if ((es - (NE - LE)) * PPE + pp < NP) store_piece_u((es - (NE - LE)) * PPE + pp);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fs][i][e], fb[fs][j][e], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  };
  if (a.tune & 1) __builtin_amdgcn_s_setprio(0);
  stamp(2);
  if (ut) {
    const std::integral_constant<int, 0> b0;
    const std::integral_constant<int, 1> b1;
    int kk = k_begin;
    for (; kk + 2 < k_end; kk += 2) {
      kstep_u(kk, b0, std::true_type{});
      kstep_u(kk + 1, b1, std::true_type{});
    }
    if (kk + 1 < k_end) {
      kstep_u(kk, b0, std::true_type{});
      kstep_u(kk + 1, b1, std::false_type{});
    } else if (kk < k_end) {
      kstep_u(kk, b0, std::false_type{});
    }
  } else {
    for (int kk = k_begin; kk + 1 < k_end; ++kk) kstep(kk, std::true_type{});
    if (k_begin < k_end) kstep(k_end - 1, std::false_type{});
  }
  stamp(3);
  if (a.tune & 2) __builtin_amdgcn_s_setprio(3);
  if constexpr (KG > 1) {
    // join: K-groups 1 .. KG - 1 park their accumulators in THEIR OWN stage regions (their K loops are over; no other group
    // touches those) and leave; group 0 adds them in group order.  The groups may run different numbers of K-step barriers:
    // a barrier completes when every live wave has arrived at one, a later group's last arrival is the one below, and
    // group 0 -- whose range is never the shorter -- reads after its own arrival here, which cannot precede theirs.
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) smem[((i * TN + j) * 16 + e) * 256 + tid] = acc[i][j][e];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int q = 1; q < KG; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += smem_all[q * STAGE_FLOATS + ((i * TN + j) * 16 + e) * 256 + tid];
    static_assert(TM * TN * 16 * 256 <= STAGE_FLOATS, "the accumulators of a K-group fit its stage region");
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
  // Straight-line per variant (residual / mask / statistics / raw split-K partials are compile-time flags of the
  // lambda below) with raw buffer accesses: one 32-bit lane offset per accumulator tile plus a SCALAR row offset
  // per element, rows past M and columns past Co dropped by the range check.  (The former per-element 64-bit
  // index arithmetic and uniform branches made the epilogue ~12 us per tile -- longer than 25 K-steps.)
  const float sc0 = a.scale0 ? a.scale0[0] : a.out_scale, sc1 = a.scale1 ? a.scale1[0] : a.out_scale;
  const int split = a.scale0 ? a.scale_split : 0x7fffffff;
  const unsigned rowbytes = (unsigned)g.Co * 4u;
  const unsigned ybytes = (unsigned)a.M * rowbytes;
  float* ydst = a.ksplit > 1 ? a.slab + (long)blockIdx.y * a.M * g.Co : a.y;
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(ydst, 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.residual ? a.residual : a.y), 0, a.residual ? (int)ybytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.mask_src ? a.mask_src : a.y), 0, a.mask_src ? (int)ybytes : 0, 0x00020000);
  const float rfloor = a.res_relu ? 0.f : -__builtin_huge_valf();   // max(r, rfloor): relu(r) or r
  float cs1[TN], cs2[TN];   // fused BatchNorm statistics: column sums of the stored values
#pragma unroll
  for (int j = 0; j < TN; ++j) cs1[j] = cs2[j] = 0.f;
  auto epilogue = [&](auto has_res, auto has_mask, auto has_stats, auto raw) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + fi;
        const bool col_ok = n < g.Co;
        const float bv = (!decltype(raw)::value && a.bias && col_ok) ? a.bias[n] : 0.f;
        const int mrow = m0 + wm * (TM * 32) + i * 32 + 4 * fh;
        const unsigned vbase = col_ok ? ((unsigned)mrow * g.Co + n) * 4u : 0x80000000u;
        // all residual / mask loads of the tile first: issued back to back, not one round trip per element
        float rres[16], rmsk[16];
        if (decltype(has_res)::value) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            rres[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rsrc, vbase, (int)(((e & 3) + 8 * (e >> 2)) * rowbytes), 0));
        }
        if (decltype(has_mask)::value) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            rmsk[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    msrc, vbase, (int)(((e & 3) + 8 * (e >> 2)) * rowbytes), 0));
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k = (e & 3) + 8 * (e >> 2);
          const int soff = (int)(k * rowbytes);
          const float av = acc[i][j][e];
          if (decltype(raw)::value) {
            // (__float_as_uint of a scalar copy: bit_cast of the vector element was miscompiled to element 0 here)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av), ysrc, vbase, soff, 0);
            continue;
          }
          const int m = mrow + k;
          float v = fmaf(av, m < split ? sc0 : sc1, bv);
          if (decltype(has_res)::value) v += fmaxf(rres[e], rfloor);
          if (decltype(has_mask)::value) v = rmsk[e] > 0.f ? v : v * a.mask_slope;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ysrc, vbase, soff, 0);
          if (decltype(has_stats)::value) {
            const float vm = (m < a.M && col_ok) ? v : 0.f;
            cs1[j] += vm;
            cs2[j] = fmaf(vm, vm, cs2[j]);
          }
        }
      }
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    const bool hr = a.residual != nullptr, hm = a.mask_src != nullptr, hs = a.stat_partials != nullptr;
    if (a.ksplit > 1) epilogue(F{}, F{}, F{}, T{});
    else if (hs) { if (hr) epilogue(T{}, F{}, T{}, F{}); else epilogue(F{}, F{}, T{}, F{}); }   // statistics: forward only (no mask)
    else if (hr && hm) epilogue(T{}, T{}, F{}, F{});
    else if (hr) epilogue(T{}, F{}, F{}, F{});
    else if (hm) epilogue(F{}, T{}, F{}, F{});
    else epilogue(F{}, F{}, F{}, F{});
  }
  if (a.stat_partials) {
    // combine the two half-waves (same column), then the WM wave rows through LDS (tiles are done with it)
    float* red = smem;  // [WM][BN][2]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      cs1[j] += __shfl_xor(cs1[j], 32, 64);
      cs2[j] += __shfl_xor(cs2[j], 32, 64);
      if (fh == 0) {
        const int col = wn * (TN * 32) + j * 32 + fi;
        red[(wm * BN + col) * 2 + 0] = cs1[j];
        red[(wm * BN + col) * 2 + 1] = cs2[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < g.Co) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      float* p = a.stat_partials + (long)(tile / tiles_n) * 2 * g.Co;
      p[n0 + tid] = t1;
      p[g.Co + n0 + tid] = t2;
    }
  }
  if constexpr (STAMP) {
    stamp(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the workgroup's stores have left the CU
    stamp(5);
    if (tid == 0 && a.stamps) {
      unsigned long long* o = a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
      o[0] = st_r0;
      o[1] = __builtin_amdgcn_s_memrealtime();
      for (int i = 0; i < 5; ++i) o[2 + i] = st_t[i + 1] - st_t[i];
      // HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20), 32 bits each
      o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
             ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
  }
}

// split-K second stage: y = epi(sum_s slab[s]) (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvGemmArgs a) {
  const int C4 = a.g.Co >> 2;
  const long n4 = (long)a.M * C4;
  const float sc0 = a.scale0 ? a.scale0[0] : 1.f, sc1 = a.scale1 ? a.scale1[0] : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int m = (int)(i / C4);
    f32x4 s = reinterpret_cast<const f32x4*>(a.slab)[i];
    for (int k = 1; k < a.ksplit; ++k) s += reinterpret_cast<const f32x4*>(a.slab)[(long)k * n4 + i];
    const float sc = a.scale0 ? (m < a.scale_split ? sc0 : sc1) : a.out_scale;
    s *= sc;
    if (a.bias) s += reinterpret_cast<const f32x4*>(a.bias)[c4];
    if (a.residual) {
      f32x4 r;
      if (a.res_up == 1) {
        const int ox = m % a.g.Wo, q = m / a.g.Wo, oy = q % a.g.Ho, b = q / a.g.Ho;
        r = residual_up2(a.residual, b, oy, ox, a.g.Ho >> 1, a.g.Wo >> 1, a.g.Co, c4 * 4);
      } else if (a.res_up == 2) {
        const int ox = m % a.g.Wo, q = m / a.g.Wo, oy = q % a.g.Ho, b = q / a.g.Ho;
        r = 0.25f * *reinterpret_cast<const f32x4*>(a.residual + (((long)b * (a.g.Ho >> 1) + (oy >> 1)) * (a.g.Wo >> 1) + (ox >> 1)) * a.g.Co + c4 * 4);
      } else {
        r = reinterpret_cast<const f32x4*>(a.residual)[i];
      }
      if (a.res_relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
      }
      s += r;
    }
    if (a.mask_src) {
      const f32x4 ms = reinterpret_cast<const f32x4*>(a.mask_src)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = ms[e] > 0.f ? s[e] : s[e] * a.mask_slope;
    }
    reinterpret_cast<f32x4*>(a.y)[i] = s;
  }
}

static unsigned long long* g_stamps = nullptr;   // diagnostic: stamp buffer of the STAMP kernels (null in production)
static long g_stamp_slots = 0;
static int g_force_ksplit = 0;                    // tuning sweeps only (diagan_conv_gemm_tune)
static int g_tune_flags = -1;                     // -1: production default (see kDefaultTune)
static long g_lds_delta = 0;
static int g_wino = -1;                           // -1: DIAGAN_WINO / default (on); 0 / 1: diagan_conv_gemm_set_wino
static int g_wino4 = -1;                          // -1: DIAGAN_WINO4 / default (on); 0 / 1: diagan_conv_gemm_set_wino4
// transformed-weights hand-over (conv_common.h): the hint set for the NEXT diagan_conv_gemm call of this thread, the one in
// force during the current call, and the format the last call needed
struct WinoFormat { const float* u; int kind, flip; float scale; long floats; };
static thread_local WinoFormat g_hint_next = {nullptr, 0, 0, 0.f, 0}, g_hint_now = {nullptr, 0, 0, 0.f, 0}, g_fmt_last = {nullptr, 0, 0, 0.f, 0};
static thread_local long g_weight_launches = 0;   // per-launch weight-transform kernels issued by this thread (tests)
constexpr int kDefaultTune = 0;

static int g_gemm_x3 = -1;                        // -1: DIAGAN_GEMM_X3 / default; 0 / 1: diagan_conv_gemm_set_x3 (process-wide, diagnostics)
// tile_cfg 17 (round 6, conv_gemm_x3b.hip): the large implicit GEMMs on the bf16 pipe with split operands.  -1: DIAGAN_GEMM_X3B /
// default; 0 / 1: diagan_conv_gemm_set_x3b.  The automatic choice upgrades an implicit-GEMM pick (never a Winograd one) where the
// launch has at least `x3b_min_tiles()` 128 x 128 tiles and four K-steps.
static int g_gemm_x3b = -1;
static int x3b_min_tiles() {
  static const int env = getenv("DIAGAN_GEMM_X3B_MIN_TILES") ? atoi(getenv("DIAGAN_GEMM_X3B_MIN_TILES")) : 192;
  return env;
}
// the output map of the NEXT diagan_conv_gemm call of this thread (diagan_conv_gemm_out_map); like the weights hint it holds for
// exactly one call
static thread_local OutMap g_map_next = {0, 0, 0, 0, 0, 0, 0, 0, 0};
// ---- per-call selection options (round 6; include/diagan_hip.h: diagan_conv_opts) --------------------------------------------------
// What a diagan_conv_gemm call selects with -- Winograd on / off, the split-operand kernels, a forced split-K factor, tune bits, the
// in-kernel split-K combine and its ticket buffer -- comes from the options the CALLER handed over for that call (thread-local, held for
// exactly one call like the weights hint and the output map); a field at -1 (0 for force_ksplit) means "the process default": the
// environment variable latched at first use or, for diagnostics and the tests' like-with-like runs, the process-wide setters below.
// Nothing on the product path writes a process-global.
struct CallOpts {
  int wino, wino4, wino4x, gemm_x3, gemm_x3b, splitk_fused, force_ksplit, tune;
  int* tickets;
  long ticket_slots;
};
static const CallOpts kNoOpts = {-1, -1, -1, -1, -1, -1, 0, -1, nullptr, 0};
static thread_local CallOpts g_opts_next = kNoOpts, g_opts_now = kNoOpts;
static thread_local int g_last_cfg = 0;           // tile configuration the last call of this thread resolved to
int call_opt_wino4x() { return g_opts_now.wino4x; }          // (conv_wino4.hip: wino4_get_x3)
static int sel_wino() {
  static const int env = getenv("DIAGAN_WINO") ? atoi(getenv("DIAGAN_WINO")) : 1;
  return g_opts_now.wino >= 0 ? g_opts_now.wino : (g_wino >= 0 ? g_wino : env);
}
static int sel_wino4() { return g_opts_now.wino4 >= 0 ? g_opts_now.wino4 : g_wino4; }      // (-1: the environment's DIAGAN_WINO4 decides, at its use)
static int sel_force_ksplit() { return g_opts_now.force_ksplit > 0 ? g_opts_now.force_ksplit : g_force_ksplit; }
static bool gemm_x3_on() {
  static const int env = getenv("DIAGAN_GEMM_X3") ? atoi(getenv("DIAGAN_GEMM_X3")) : 1;      // on: SNGAN-32 5270-5281 -> 5355 images/s
  return (g_opts_now.gemm_x3 >= 0 ? g_opts_now.gemm_x3 : (g_gemm_x3 >= 0 ? g_gemm_x3 : env)) != 0;
}
static bool gemm_x3b_on() {
  static const int env = getenv("DIAGAN_GEMM_X3B") ? atoi(getenv("DIAGAN_GEMM_X3B")) : 1;
  return (g_opts_now.gemm_x3b >= 0 ? g_opts_now.gemm_x3b : (g_gemm_x3b >= 0 ? g_gemm_x3b : env)) != 0;
}
static bool x3b_takes(const ConvGemmArgs& a, int cfg, int64_t ws_floats) {
  if (!gemm_x3b_on() || !(cfg == 1 || cfg == 3 || cfg == 5 || cfg == 7 || cfg == 8)) return false;
  const ConvGeom& g = a.g;
  // (... and enough work to carry the per-launch weight split: below ~4e9 multiply-accumulates -- the 15 us 1x1 shortcuts of the SNGAN
  //  nets -- the 5 us split launch in front costs more than the faster pipe returns)
  return gemm_x3b_geom_ok(a) && g.Kp >= 128 && g.Co >= 64 && (long)cdiv(a.M, 128) * cdiv(g.Co, 128) >= x3b_min_tiles() &&
         (double)a.M * g.Co * g.Kp >= 4e9 && gemm_x3b_ws_floats(g.Co, g.Kp) <= ws_floats;
}
// Ticket counters of the in-kernel split-K combine (ConvGemmArgs::tickets): one int per output tile, zero between launches (the
// last-arriving workgroup of a tile resets its counter).  CALLER-OWNED (the library allocates nothing): the call's options carry the
// buffer, or -- diagnostics -- diagan_conv_gemm_set_splitk_tickets registers one process-wide.  OFF by default: measured on the F(2x2)
// kernel's 36 / 10 split launches per SNGAN-64 / -32 step it buys nothing -- plain partial stores + an agent-scope release per workgroup:
// -0.5 % (the release writes the L2's dirty lines back); write-through partial stores, no release: +-0.1 % (2998-3005 vs 3003-3008
// images/s): draining the stores and the last arriver's serial read-back cost what the 7 us second launch costs.
static int g_splitk_fused = -1;                   // -1: DIAGAN_SPLITK_FUSED / default (off); 0 / 1: diagan_conv_gemm_set_splitk_fused
static int* g_tickets = nullptr;                  // diagnostics: diagan_conv_gemm_set_splitk_tickets
static long g_ticket_slots = 0;
static int* splitk_tickets(long tiles) {
  static const int env = getenv("DIAGAN_SPLITK_FUSED") ? atoi(getenv("DIAGAN_SPLITK_FUSED")) : 0;
  if (!(g_opts_now.splitk_fused >= 0 ? g_opts_now.splitk_fused : (g_splitk_fused >= 0 ? g_splitk_fused : env))) return nullptr;
  if (g_opts_now.tickets) return tiles <= g_opts_now.ticket_slots ? g_opts_now.tickets : nullptr;
  return (g_tickets && tiles <= g_ticket_slots) ? g_tickets : nullptr;        // (no buffer: the second launch, as with the switch off)
}

template <int BM, int BN, int WM, int WN, int BK, int PRO, bool STAMP = false, bool FP = false, int KG = 1>
static int launch_one(const ConvGemmArgs& a, hipStream_t st) {
  const int tiles = cdiv(a.M, BM) * cdiv(a.g.Co, BN);
  // (g_lds_delta: occupancy probe of the tuning sweeps, timing only -- a negative value leaves part of the tile outside
  //  the allocation, where LDS accesses are dropped by the hardware's range check)
  const size_t lds = (size_t)((long)((size_t)2 * KG * (BM + BN) * BK * sizeof(float)) + g_lds_delta);
  auto kern = conv_gemm_kernel<BM, BN, WM, WN, BK, PRO, STAMP, FP, KG>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, lds);
  hipLaunchKernelGGL(kern, dim3(tiles, a.ksplit), dim3(256 * KG), lds, st, a);
  if (a.ksplit > 1) {
    long blocks = ((long)a.M * (a.g.Co / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((int)blocks), dim3(256), 0, st, a);
  }
  return check_launch("conv_gemm");
}

// SPEC: one kernel per prologue mode (the production tiles); otherwise the mode is a run-time argument.
template <int BM, int BN, int WM, int WN, int BK = 32, bool SPEC = false, bool FP = false, int KG = 1>
static int launch_cfg(const ConvGemmArgs& a, hipStream_t st) {
  if (a.stamps && KG > 1) return set_err(DIAGAN_EUNSUP, "conv_gemm: no stamped build of the two-group kernel");
  if (a.stamps) {                             // diagnostic build: the two most common prologue modes only
    if (a.pro_mode == PRO_NONE) return launch_one<BM, BN, WM, WN, BK, PRO_NONE, true, FP>(a, st);
    if (a.pro_mode == PRO_RELU) return launch_one<BM, BN, WM, WN, BK, PRO_RELU, true, FP>(a, st);
    return set_err(DIAGAN_EUNSUP, "conv_gemm: the stamped diagnostic kernels exist for prologue modes 0 and 1");
  }
  if (SPEC) {
    switch (a.pro_mode) {
      case PRO_NONE: return launch_one<BM, BN, WM, WN, BK, SPEC ? PRO_NONE : -1, false, FP, KG>(a, st);
      case PRO_RELU: return launch_one<BM, BN, WM, WN, BK, SPEC ? PRO_RELU : -1, false, FP, KG>(a, st);
      case PRO_AFFINE_RELU: return launch_one<BM, BN, WM, WN, BK, SPEC ? PRO_AFFINE_RELU : -1, false, FP, KG>(a, st);
      case PRO_LRELU: return launch_one<BM, BN, WM, WN, BK, SPEC ? PRO_LRELU : -1, false, FP, KG>(a, st);
      default: return launch_one<BM, BN, WM, WN, BK, SPEC ? PRO_AFFINE : -1, false, FP, KG>(a, st);
    }
  }
  return launch_one<BM, BN, WM, WN, BK, -1, false, FP, KG>(a, st);
}

}  // namespace diagan

using namespace diagan;

static bool split128_enabled() {
  static const bool on = getenv("DIAGAN_SPLIT128") && atoi(getenv("DIAGAN_SPLIT128")) > 0;
  return on;
}

// Tile selection used when tile_cfg == 0.  Measured on MI355X (tools/tile_sweep.py, profiles/r02_tile_sweep.md):
//  * 64-column outputs (the 64x64-resolution blocks of SNGAN-64): a 256x64 tile whose waves own 64x64 sub-tiles needs
//    half the LDS fragment reads per MFMA of the 64x64 tile and holds a higher clock (2.2-2.3 vs 2.1 GHz): +3-5 %;
//  * few output tiles but a long K loop (the 4x4 / 8x8 blocks): 128x128 tiles with the K loop split over 2-4 workgroups
//    (a lone 128x128 workgroup on a CU still runs at 0.84 of the MFMA rate, a lone 64x64 one at 0.58): +4-7 % alone,
//    nothing inside a training step: opt-in;
//  * otherwise 128x128 against 64x64 by wave quantisation and column waste, as before -- with FOUR resident 64x64
//    workgroups per CU (the stamped build shows 4, not the 5 that 160 KB / 32 KB suggests).
// allow_split: the caller can take the split-K path (workspace present, no fused BatchNorm statistics).
DIAGAN_API int diagan_conv_gemm_pick_cfg(int M, int Co, int Kp, int allow_split) {
  const int small = 7;                     // 64x64 (with double-buffered fragments)
  // short K loops (first conv of D on RGB, 1x1 shortcuts) are bound by the output stream, not the MFMAs: more,
  // smaller workgroups in flight win (measured 52 -> 46 us at M=131072,N=128,K=36; 27 -> 19 us at K=128; round 2, fp32:
  // K = 256 at N = 256: 45.6 -> 33.4 us at M=20480, 118.8 -> 110.5 at M=81920, 27.2 -> 25.0 at M=16384)
  if (Kp <= 256) return small;
  const long t128 = (long)cdiv(M, 128) * cdiv(Co, 128), t64 = (long)cdiv(M, 64) * cdiv(Co, 64);
  // (in a training step this is within box-to-box noise of the 64x64 tile -- tools/layer_report.py, 32.8 vs 33.2 ms of
  //  GEMM time per SNGAN-64 step -- so it stays opt-in: DIAGAN_SPLIT128=1)
  if (split128_enabled() && allow_split && (Co & 127) == 0 && t128 <= 256 && diagan_conv_gemm_pick_ksplit(M, Co, Kp, 1) > 1) return 1;
  if (Co <= 64 && Kp >= 256) return M >= 65536 ? 5 : (M >= 32768 ? 8 : 7);
  // Blocks run a whole K loop, so a partially filled last round of blocks costs a full round
  // ("wave quantisation"): weigh each tile shape by tiles / (rounds * resident slots).
  const long s128 = 256 * 2, s64 = 256 * 4;     // resident workgroups
  const double q128 = (double)t128 / (double)(cdiv(t128, s128) * s128);
  const double q64 = (double)t64 / (double)(cdiv(t64, s64) * s64);
  const double waste128 = (double)M * Co / ((double)t128 * 128 * 128);
  const double waste64 = (double)M * Co / ((double)t64 * 64 * 64);
  const double bias = 1.08;
  if (bias * q128 * waste128 > q64 * waste64) return 1;
  // at most one 64x64 tile per CU and a K loop that is worth halving but too short for split-K launches: two K-groups in
  // one workgroup (tile_cfg 14) put two waves on every SIMD.  DIAGAN_KG2=0: off.
  static const int kg2 = getenv("DIAGAN_KG2") ? atoi(getenv("DIAGAN_KG2")) : 1;
  if (kg2 && t64 <= 256 && Kp >= 16 * 32 && diagan_conv_gemm_pick_ksplit(M, Co, Kp, small) == 1) return 14;
  return small;
}

// Split-K factor for the 64x64 tile (1 = none), re-measured with the overlapped K-step schedule
// (tools/bench_split.py):
//  * at most ~1 tile per CU: a workgroup alone on its CU runs a K-step in ~0.9 us, so splitting only pays for
//    long K loops (M=4096,N=256,K=2304: 60 -> 54 us with 3 splits; M=8192,N=128,K=1152: 31.6 us unsplit, 36 split);
//  * up to 2 tiles per CU: fill the 1280 resident slots, at least 8 K-steps per split;
//  * more tiles: no split.  (Splitting 1024-tile problems to even out the last round of workgroups makes the
//    GEMM itself 5-8 % faster -- 91 -> 85 us -- but the second-stage launches and slab traffic cost more than that
//    over a whole training step: 2350 -> 2323 images/s.)
DIAGAN_API int diagan_conv_gemm_pick_ksplit(int M, int Co, int Kp, int cfg) {
  static const int env_forced = getenv("DIAGAN_KSPLIT") ? atoi(getenv("DIAGAN_KSPLIT")) : 0;   // tuning experiments only
  const int forced = sel_force_ksplit() > 0 ? sel_force_ksplit() : env_forced;
  if (forced > 0 && !(Co & 3)) return forced < Kp / 64 ? forced : (Kp / 64 > 0 ? Kp / 64 : 1);
  if (Co & 3) return 1;
  const int nk = Kp / 32;
  if (cfg == 1) {                      // 128x128 tiles on problems with at most one tile per CU and a long K loop (opt-in)
    if (!split128_enabled()) return 1;
    const long t128 = (long)cdiv(M, 128) * cdiv(Co, 128);
    if (t128 > 256 || nk < 64) return 1;
    int s = t128 <= 128 ? 4 : 2;
    while (s > 1 && nk / s < 16) --s;
    return s;
  }
  if (cfg != 3 && cfg != 7) return 1;
  const long tiles = (long)cdiv(M, 64) * cdiv(Co, 64);
  if (nk < 16) return 1;
  if (tiles <= 320) {
    if (nk < 64) return 1;
    int s = nk / 24;
    return s > 4 ? 4 : s;
  }
  if (tiles > 512) return 1;
  long sp = 1024 / tiles;          // four resident 64x64 workgroups per CU
  if (sp > nk / 8) sp = nk / 8;       // at least 8 K-steps per split
  return sp < 2 ? 1 : (int)sp;
}

const float* diagan::wino_weights_ready(int kind, int flip, float scale, long floats) {
  g_fmt_last = WinoFormat{nullptr, kind, flip, scale, floats};
  if (g_hint_now.u && g_hint_now.kind == kind && g_hint_now.flip == flip && g_hint_now.scale == scale) return g_hint_now.u;
  ++g_weight_launches;
  return nullptr;
}

// see include/diagan_hip.h
DIAGAN_API int diagan_conv_gemm_weights_hint(const float* u, int kind, int flip, float scale) {
  g_hint_next = WinoFormat{u, kind, flip, scale, 0};
  return DIAGAN_OK;
}
DIAGAN_API int diagan_conv_gemm_last_weight_format(int* kind, int* flip, float* scale, int64_t* floats, int64_t* launches) {
  if (kind) *kind = g_fmt_last.kind;
  if (flip) *flip = g_fmt_last.flip;
  if (scale) *scale = g_fmt_last.scale;
  if (floats) *floats = g_fmt_last.floats;
  if (launches) *launches = g_weight_launches;
  return DIAGAN_OK;
}
DIAGAN_API int64_t diagan_wino_weight_blocks(int Co, int Ci) { return (int64_t)cdiv(Ci, 32) * cdiv(Co, 64); }
DIAGAN_API int diagan_wino_weights_batched(const void* jobs, int n, int blocks, void* stream) {
  DG_REQUIRE(jobs && n > 0 && blocks > 0, "wino_weights_batched: bad job table");
  return launch_wino_weights_batched((const WinoJob*)jobs, n, blocks, (hipStream_t)stream);
}

// see include/diagan_hip.h
DIAGAN_API int diagan_conv_gemm(const float* x, const float* w, float* y, const float* bias,
                                const float* residual, int res_relu, const float* mask_src, float mask_slope,
                                const float* pro_scale, const float* pro_shift, int pro_mode,
                                float out_scale, const float* scale0, const float* scale1, int scale_split,
                                int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                                int R, int S, int sy, int dr, int off, int up, int Kp, int tile_cfg,
                                float* splitk_ws, int64_t splitk_ws_floats, float* stat_partials, int pro_group_rows,
                                void* stream) {
  g_hint_now = g_hint_next;                 // a hint holds for exactly one call, whatever path that call takes
  g_hint_next = WinoFormat{nullptr, 0, 0, 0.f, 0};
  g_fmt_last = WinoFormat{nullptr, 0, 0, 0.f, 0};
  const OutMap map = g_map_next;            // ... and so does an output map
  g_map_next = OutMap{0, 0, 0, 0, 0, 0, 0, 0, 0};
  struct OptsScope {                        // ... and the selection options: in force until this call returns, whatever path it takes
    OptsScope() { g_opts_now = g_opts_next; g_opts_next = kNoOpts; }
    ~OptsScope() { g_opts_now = kNoOpts; }
  } opts_scope;
  g_last_cfg = 0;
  DG_REQUIRE(x && w && y, "conv_gemm: null tensor");
  DG_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Co > 0 && R > 0 && S > 0, "conv_gemm: bad dims");
  DG_REQUIRE(Ci > 0 && (Ci & 3) == 0, "conv_gemm: Ci=%d must be a positive multiple of 4 (pad the tensor)", Ci);
  DG_REQUIRE(up == 1 || up == 2, "conv_gemm: up=%d unsupported (1 or 2)", up);
  DG_REQUIRE(dr == 1 || dr == -1, "conv_gemm: dr must be +-1");
  DG_REQUIRE(R <= 15 && S <= 15, "conv_gemm: filter %dx%d larger than 15x15", R, S);
  DG_REQUIRE(Kp % 32 == 0 && Kp >= R * S * Ci, "conv_gemm: Kp=%d must be a multiple of 32 and >= R*S*Ci=%d", Kp, R * S * Ci);
  DG_REQUIRE(pro_mode >= 0 && pro_mode <= 4, "conv_gemm: bad pro_mode %d", pro_mode);
  DG_REQUIRE(!(pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE) || (pro_scale && pro_shift),
             "conv_gemm: affine prologue needs scale/shift");
  DG_REQUIRE((long)B * Ho * Wo * Co * 4 < (1L << 31) && (long)B * Hi * Wi * Ci * 4 < (1L << 31) &&
                 (long)Co * Kp * 4 < (1L << 31),
             "conv_gemm: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
  ConvGemmArgs a;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.residual = residual; a.mask_src = mask_src;
  a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.mask_slope = mask_slope; a.out_scale = out_scale;
  a.scale0 = scale0; a.scale1 = scale1; a.scale_split = scale_split;
  DG_REQUIRE(!scale0 || scale1, "conv_gemm: scale0 and scale1 must be given together");
  DG_REQUIRE(!(stat_partials && mask_src), "conv_gemm: stat_partials (forward statistics) and mask_src (backward mask) are exclusive");
  a.pro_mode = pro_mode; a.M = B * Ho * Wo; a.res_relu = res_relu & 1; a.res_up = (res_relu & 2) ? 1 : (res_relu & 4) ? 2 : 0;
  DG_REQUIRE((res_relu & ~7) == 0 && (res_relu & 6) != 6, "conv_gemm: res_relu=%d (bit 0: relu(residual), bit 1: half-resolution residual, "
             "up-sampled; bit 2: half-resolution residual, un-pooled)", res_relu);
  DG_REQUIRE(a.res_up != 1 || (residual && !(Ho & 1) && !(Wo & 1) && !mask_src && !a.res_relu),
             "conv_gemm: a half-resolution residual needs the tensor, even Ho / Wo, no backward mask and no ReLU on it");
  DG_REQUIRE(a.res_up != 2 || (residual && !(Ho & 1) && !(Wo & 1) && !a.res_relu && !stat_partials),
             "conv_gemm: an un-pooled half-resolution residual needs the tensor, even Ho / Wo, no ReLU on it, no statistics");
  a.g = ConvGeom{B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, R * S * Ci, Kp};
  a.dWo = make_fastdiv((unsigned)Wo);
  a.dHo = make_fastdiv((unsigned)Ho);
  a.stat_partials = stat_partials;
  a.pro_group_rows = pro_group_rows;
  hipStream_t st = (hipStream_t)stream;
  int cfg = tile_cfg != 0 ? tile_cfg
                          : diagan_conv_gemm_pick_cfg_grouped(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp,
                                                              (splitk_ws && !stat_partials) ? 1 : 0,
                                                              splitk_ws ? splitk_ws_floats : 0, pro_group_rows);
  if (tile_cfg == 0 && splitk_ws && x3b_takes(a, cfg, splitk_ws_floats)) cfg = 17;
  g_last_cfg = cfg;
  if (map.mul != 0) {
    DG_REQUIRE(map.mul > 0 && map.OH > 0 && map.OW > 0 && map.y0 >= 0 && map.x0 >= 0 && map.y1 <= Ho && map.x1 <= Wo &&
                   map.offy >= 0 && map.offx >= 0 && (map.y1 <= map.y0 || map.mul * (map.y1 - 1 - map.y0) + map.offy < map.OH) &&
                   (map.x1 <= map.x0 || map.mul * (map.x1 - 1 - map.x0) + map.offx < map.OW) &&
                   (long)B * map.OH * map.OW * Co * 4 < (1L << 31),
               "conv_gemm: output map (mul %d, off %d/%d, window [%d,%d) x [%d,%d), target %d x %d) does not fit", map.mul, map.offy,
               map.offx, map.y0, map.y1, map.x0, map.x1, map.OH, map.OW);
    DG_REQUIRE(cfg == 17, "conv_gemm: an output map is written by the split-operand kernel only (tile_cfg 17; ask "
               "diagan_conv_gemm_final_cfg first)");
  }
  const int bm = diagan_conv_gemm_tile_rows(cfg);
  DG_REQUIRE(bm > 0, "conv_gemm: unknown tile_cfg %d (0 = auto, 1 = 128x128, 3 = 64x64, 5 = 256x64, 7 = 64x64 with fragment "
             "prefetch, 8 = 128x64 with fragment prefetch, 9 = Winograd F(2x2,3x3), 11 / 12 = Winograd + average pool and its data-gradient, 13 = Winograd F(4x4,3x3), 15 = the same on the bilinear x2 of a half-resolution input; 2, 4, 6, 10 were retired)", tile_cfg);
  DG_REQUIRE(pro_group_rows >= 0 && (pro_group_rows == 0 || (pro_group_rows % bm == 0 && a.M % pro_group_rows == 0)),
             "conv_gemm: pro_group_rows=%d must be a multiple of the %d-row tile and divide M=%d", pro_group_rows, bm, a.M);
  a.slab = splitk_ws;
  a.tickets = nullptr;
  a.ksplit = 1;
  a.tune = g_opts_now.tune >= 0 ? g_opts_now.tune : (g_tune_flags >= 0 ? g_tune_flags : kDefaultTune);
  a.stamps = nullptr;
  if (g_stamps) {
    const long wgs = (long)cdiv(a.M, bm) * cdiv(Co, diagan_conv_gemm_tile_cols(cfg)) * 16;   // room for up to 16 K splits
    DG_REQUIRE(wgs <= g_stamp_slots, "conv_gemm: stamp buffer has %ld slots, this launch may need %ld", g_stamp_slots, wgs);
    a.stamps = g_stamps;
  }
  if (splitk_ws && !stat_partials) {
    const int ks = diagan_conv_gemm_pick_ksplit(a.M, Co, Kp, cfg);
    if (ks > 1 && (int64_t)ks * a.M * Co <= splitk_ws_floats) a.ksplit = ks;
  }
  if (cfg == 11) {
    // Winograd + 2x2 average pool (conv_wino_pool.hip): y and residual are [B, Ho/2, Wo/2, Co]
    DG_REQUIRE(diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up) && dr == 1 && Co % 64 == 0 &&
                   (pro_mode == PRO_NONE || pro_mode == PRO_RELU) && !mask_src && !stat_partials && !a.res_up && pro_group_rows == 0,
               "conv_gemm: tile_cfg 11 (Winograd + average pool) needs a forward 3x3 / stride 1 / pad 1 geometry with even H, W, "
               "Ci %% 8 == 0, Co %% 64 == 0, prologue none / ReLU, no mask, statistics or half-resolution residual");
    if (sel_wino4() != 0 && splitk_ws && wino4_pool_ok(B, Ho, Wo, Ci, Co, (long)splitk_ws_floats, sel_wino4() == 2)) {     // 25 products per 4x4 tile (conv_wino4.hip)
      a.ksplit = 1;
      return launch_wino4_pool(a, splitk_ws, st);
    }
    const long wfl = wino_ws_floats(Co, Ci);
    DG_REQUIRE(splitk_ws && splitk_ws_floats >= wfl, "conv_gemm: tile_cfg 11 needs %ld floats of workspace for the transformed weights", wfl);
    int ks = wino_pool_ksplit(B, Ho, Wo, Ci, Co, (long)splitk_ws_floats - wfl, 192);
    if (ks < 1) ks = 1;
    a.ksplit = ks;
    a.slab = splitk_ws + wfl;
    int rc = launch_wino_pool(a, splitk_ws, st);
    if (rc == DIAGAN_OK && ks > 1) {
      ConvGemmArgs e = a;                                  // second stage over the POOLED pixels
      e.M = B * (Ho >> 1) * (Wo >> 1);
      e.scale_split = a.scale_split >> 2;
      e.out_scale = a.out_scale;
      long blocks = ((long)e.M * (Co / 4) + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((int)blocks), dim3(256), 0, st, e);
      rc = check_launch("conv_wino_pool split-K epilogue");
    }
    return rc;
  }
  if (cfg == 12) {
    // data-gradient of (Winograd convolution + 2x2 average pool): x is the HALF-resolution gradient [B, Ho/2, Wo/2, Ci]
    DG_REQUIRE(diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up) && dr == -1 && Co % 64 == 0 &&
                   pro_mode == PRO_NONE && !stat_partials && !a.res_up && pro_group_rows == 0,
               "conv_gemm: tile_cfg 12 (data-gradient through an average pool, Winograd) needs the data-gradient geometry of a "
               "3x3 / stride 1 / pad 1 layer with even H, W, Ci %% 8 == 0, Co %% 64 == 0, no prologue, statistics or half-resolution residual");
    if (sel_wino4() != 0 && splitk_ws && wino4_pool_ok(B, Ho, Wo, Ci, Co, (long)splitk_ws_floats, sel_wino4() == 2)) {     // the same on the pooled gradient
      a.ksplit = 1;
      return launch_wino4_unpool(a, splitk_ws, st);
    }
    const long wfl = wino_ws_floats(Co, Ci);
    DG_REQUIRE(splitk_ws && splitk_ws_floats >= wfl, "conv_gemm: tile_cfg 12 needs %ld floats of workspace for the transformed weights", wfl);
    long slab_tiles = ((long)splitk_ws_floats - wfl) / 4;             // a split's partial is the FULL-resolution output: 4 pixels per tile
    int ks = wino_pool_ksplit(B, Ho, Wo, Ci, Co, slab_tiles, 192);
    if (ks < 1) ks = 1;
    a.ksplit = ks;
    a.slab = splitk_ws + wfl;
    int rc = launch_wino_unpool(a, splitk_ws, st);
    if (rc == DIAGAN_OK && ks > 1) {
      long blocks = ((long)a.M * (Co / 4) + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((int)blocks), dim3(256), 0, st, a);
      rc = check_launch("conv_wino_unpool split-K epilogue");
    }
    return rc;
  }
  if (cfg == 15) {
    // F(4x4,3x3) of the bilinear x2 up-sampling of a HALF-resolution input (conv_wino4.hip MODE 3): x is [B, Hi/2, Wi/2, Ci]
    DG_REQUIRE(diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up) && dr == 1 && wino4_geom_ok(Ho, Wo, Ci) &&
                   !mask_src && (pro_group_rows == 0 || pro_group_rows % 512 == 0),
               "conv_gemm: tile_cfg 15 (Winograd F(4x4,3x3) on an up-sampled input) needs a forward 3x3 / stride 1 / pad 1 geometry, "
               "H and W multiples of 4, Ci %% 8 == 0, no backward mask, prologue groups of whole 512-row tiles");
    const long wfl = wino4_ws_floats(Co, Ci);
    DG_REQUIRE(splitk_ws && splitk_ws_floats >= wfl, "conv_gemm: tile_cfg 15 needs %ld floats of workspace for the transformed weights", wfl);
    a.ksplit = 1;
    return launch_wino4_upin(a, splitk_ws, st);
  }
  if (cfg == 13) {
    // Winograd F(4x4,3x3) (conv_wino4.hip): 32 tiles of 4x4 outputs x 64 channels per workgroup
    DG_REQUIRE(diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up) && wino4_geom_ok(Ho, Wo, Ci),
               "conv_gemm: tile_cfg 13 (Winograd F(4x4,3x3)) needs a 3x3 / stride 1 / pad 1 geometry, H and W multiples of 4, Ci %% 8 == 0");
    const long wfl = wino4_ws_floats(Co, Ci);
    DG_REQUIRE(splitk_ws && splitk_ws_floats >= wfl, "conv_gemm: tile_cfg 13 needs %ld floats of workspace for the transformed weights", wfl);
    int ks = 1;
    if (tile_cfg == 0 && !stat_partials) {
      ks = wino4_ksplit(B, Ho, Wo, Ci, Co, 1, (long)splitk_ws_floats);
      if (ks < 1) ks = 1;
    } else if (sel_force_ksplit() > 1 && !stat_partials && wfl + (long)sel_force_ksplit() * a.M * Co <= splitk_ws_floats && Ci / 8 / sel_force_ksplit() >= 1) {
      ks = sel_force_ksplit();
    }
    a.ksplit = ks;
    a.slab = splitk_ws + wfl;
    int rc = launch_wino4(a, splitk_ws, st);
    if (rc == DIAGAN_OK && ks > 1) {
      long blocks = ((long)a.M * (Co / 4) + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((int)blocks), dim3(256), 0, st, a);
      rc = check_launch("conv_wino4 split-K epilogue");
    }
    return rc;
  }
  if (cfg == 9) {
    DG_REQUIRE(diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up),
               "conv_gemm: tile_cfg 9 (Winograd F(2x2,3x3)) needs a 3x3 / stride 1 / pad 1 geometry, even H and W, Ci %% 8 == 0");
    DG_REQUIRE(splitk_ws && splitk_ws_floats >= wino_ws_floats(Co, Ci),
               "conv_gemm: tile_cfg 9 needs %ld floats of workspace for the transformed weights", wino_ws_floats(Co, Ci));
    // transformed weights first, split-K slab (if any) behind them
    const long wfl = wino_ws_floats(Co, Ci);
    int ks = 1;
    if (tile_cfg == 0 && !stat_partials) {
      ks = wino_ksplit(B, Ho, Wo, Ci, Co, 1, (long)splitk_ws_floats, 192);
      if (ks < 1) ks = 1;
    } else if (sel_force_ksplit() > 1 && !stat_partials && wfl + (long)sel_force_ksplit() * a.M * Co <= splitk_ws_floats &&
               Ci / 16 / sel_force_ksplit() >= 1) {
      ks = sel_force_ksplit();
    }
    a.ksplit = ks;
    a.slab = splitk_ws + wfl;
    if (ks > 1) a.tickets = splitk_tickets((long)cdiv(a.M / 4, 64) * cdiv(Co, 64));
    int rc = launch_wino(a, splitk_ws, st);
    if (rc == DIAGAN_OK && ks > 1 && !a.tickets) {
      long blocks = ((long)a.M * (Co / 4) + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((int)blocks), dim3(256), 0, st, a);
      rc = check_launch("conv_wino split-K epilogue");
    }
    return rc;
  }
  DG_REQUIRE(!a.res_up, "conv_gemm: a half-resolution residual is added by the Winograd kernel only (tile_cfg 9; ask "
             "diagan_conv_gemm_pick_cfg_geom first)");
  // the lone-tile 3x3 launches (tile_cfg 14) on the bf16 pipe with exactly split operands where the geometry qualifies (round 5;
  // conv_gemm_x3.hip; the 1x1 shortcuts stay: a few K-steps, and their call sites hand over no pre-split weights); tile_cfg 16
  // asks for that kernel by name
  if (cfg == 16 || (cfg == 14 && tile_cfg == 0 && R == 3 && S == 3 && gemm_x3_on() && gemm_x3_geom_ok(a) && splitk_ws &&
                    gemm_x3_ws_floats(Co, Kp) <= splitk_ws_floats)) {
    DG_REQUIRE(gemm_x3_geom_ok(a) && splitk_ws && gemm_x3_ws_floats(Co, Kp) <= splitk_ws_floats,
               "conv_gemm: tile_cfg 16 (split-operand implicit GEMM) needs stride 1, no up-sampling, Ci %% 32 == 0, an even number of "
               "32-channel K-steps, Kp == R*S*Ci, prologue none / ReLU, no statistics, and %ld floats of workspace", gemm_x3_ws_floats(Co, Kp));
    a.ksplit = 1;
    return launch_gemm_x3(a, splitk_ws, st);
  }
  if (cfg == 17) {
    DG_REQUIRE(gemm_x3b_geom_ok(a) && splitk_ws && gemm_x3b_ws_floats(Co, Kp) <= splitk_ws_floats,
               "conv_gemm: tile_cfg 17 (split-operand implicit GEMM, 128 x 128 tiles) needs a gather without up-sampling, Ci %% 32 == 0, "
               "Kp == R*S*Ci, prologue none / ReLU / leaky ReLU, a plain epilogue (out_scale, bias, residual) and %ld floats of workspace",
               gemm_x3b_ws_floats(Co, Kp));
    a.ksplit = 1;
    return launch_gemm_x3b(a, map, splitk_ws, st);
  }
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2, 32, true>(a, st);
    case 3: return launch_cfg<64, 64, 2, 2, 32, true>(a, st);
    case 5: return launch_cfg<256, 64, 4, 1, 32, true>(a, st);
    case 7: return launch_cfg<64, 64, 2, 2, 32, true, true>(a, st);
    case 8: return launch_cfg<128, 64, 2, 2, 32, true, true>(a, st);
    case 14: return launch_cfg<64, 64, 2, 2, 32, true, true, 2>(a, st);
    default: return set_err(DIAGAN_EINVAL, "conv_gemm: unknown tile_cfg %d", tile_cfg);
  }
}

// rows / columns of a tile configuration (0 for an unknown one)
DIAGAN_API int diagan_conv_gemm_tile_rows(int cfg) {
  switch (cfg) { case 1: case 8: case 17: return 128; case 3: case 7: case 14: case 16: return 64; case 5: case 9: case 11: case 12: return 256; case 13: case 15: return 512; default: return 0; }
}
DIAGAN_API int diagan_conv_gemm_tile_cols(int cfg) {
  switch (cfg) { case 1: case 11: case 12: case 17: return 128; case 3: case 5: case 7: case 8: case 9: case 13: case 14: case 15: case 16: return 64; default: return 0; }
}

// Winograd F(2x2,3x3) (tile_cfg 9, conv_wino.hip): 3x3 taps, stride 1, pad 1 (forward: dr=+1, off=-1; data-gradient of
// such a layer: dr=-1, off=+1), same spatial size in and out, even H and W, input channels a multiple of 8.
DIAGAN_API int diagan_conv_wino_supported(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                          int off, int up) {
  return R == 3 && S == 3 && sy == 1 && up == 1 && ((dr == 1 && off == -1) || (dr == -1 && off == 1)) && Hi == Ho &&
         Wi == Wo && !(Ho & 1) && !(Wo & 1) && (Ci & 7) == 0 && (Co & 3) == 0;
}

// Winograd + 2x2 average pool (tile_cfg 11, conv_wino_pool.hip): does F.avg_pool2d(conv3x3(pro(x)), 2) of this layer run
// as ONE launch on 9/16 of the Winograd products?  Forward geometry, Co a multiple of 128, prologue none / ReLU, room for
// the transformed weights (+ a split-K slab where the launch is small), and enough workgroups to be worth it.
// DIAGAN_WINO=0 / DIAGAN_WINO_POOL=0 switch it off.
DIAGAN_API int diagan_conv_wino_pool_supported(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                               int dr, int off, int up, int pro_mode, int64_t ws_floats) {
  static const int pool_env = getenv("DIAGAN_WINO_POOL") ? atoi(getenv("DIAGAN_WINO_POOL")) : 1;
  const int wino = sel_wino();
  if (!wino || !pool_env || dr != 1 || Co % 64 != 0 || Ci < 16 || !(pro_mode == PRO_NONE || pro_mode == PRO_RELU)) return 0;
  if (!diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up)) return 0;
  const long wfl = wino_ws_floats(Co, Ci);
  if (ws_floats < wfl) return 0;
  return wino_pool_ksplit(B, Ho, Wo, Ci, Co, (long)ws_floats - wfl, 192) > 0 ? 1 : 0;
}

// tile_cfg 12: the data-gradient of such a layer from the HALF-resolution gradient (same nine products, see conv_wino_pool.hip)
DIAGAN_API int diagan_conv_wino_unpool_supported(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                                 int dr, int off, int up, int64_t ws_floats) {
  static const int pool_env = getenv("DIAGAN_WINO_POOL") ? atoi(getenv("DIAGAN_WINO_POOL")) : 1;
  const int wino = sel_wino();
  if (!wino || !pool_env || dr != -1 || Co % 64 != 0 || Ci < 16) return 0;
  if (!diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up)) return 0;
  const long wfl = wino_ws_floats(Co, Ci);
  if (ws_floats < wfl) return 0;
  return wino_pool_ksplit(B, Ho, Wo, Ci, Co, ((long)ws_floats - wfl) / 4, 192) > 0 ? 1 : 0;
}

// Tile configuration for a full geometry (what diagan_conv_gemm does when tile_cfg == 0): Winograd (9) where the layer
// qualifies, the workspace holds the transformed weights and the launch has enough workgroups; otherwise the
// implicit-GEMM choice of diagan_conv_gemm_pick_cfg.  DIAGAN_WINO=0 disables Winograd (A/B runs).
static int pick_cfg_geom_impl(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off,
                              int up, int Kp, int allow_split, int64_t ws_floats, bool allow_w4);
DIAGAN_API int diagan_conv_gemm_pick_cfg_geom(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                              int dr, int off, int up, int Kp, int allow_split, int64_t ws_floats) {
  return pick_cfg_geom_impl(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, allow_split, ws_floats, true);
}
static int pick_cfg_geom_impl(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off,
                              int up, int Kp, int allow_split, int64_t ws_floats, bool allow_w4) {
  const int wino = sel_wino();
  // one 512-thread workgroup per CU: below ~3/4 of the chip the implicit GEMM's smaller tiles win (8x8 / 4x4 blocks at
  // batch 64: 128 workgroups, 325 vs 317 us; their data-gradients 330 vs 168 us)
  static const int min_wgs = getenv("DIAGAN_WINO_MIN_WGS") ? atoi(getenv("DIAGAN_WINO_MIN_WGS")) : 192;
  if (wino && diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up) &&
      ws_floats >= wino_ws_floats(Co, Ci) && Ci >= 16) {
    static const int wsplit = getenv("DIAGAN_WINO_SPLIT") ? atoi(getenv("DIAGAN_WINO_SPLIT")) : 1;
    // F(4x4,3x3) (conv_wino4.hip, 32 tiles of 4x4 outputs x 64 channels per workgroup, ONE resident workgroup per CU): where its
    // launch-size policy (wino4_ksplit) expects it ahead of the F(2x2) kernel
    static const int w4_env = getenv("DIAGAN_WINO4") ? atoi(getenv("DIAGAN_WINO4")) : 1;
    static const int w4_min_ci = getenv("DIAGAN_WINO4_MIN_CI") ? atoi(getenv("DIAGAN_WINO4_MIN_CI")) : 64;
    if (allow_w4 && w4_env && sel_wino4() != 0 && wino4_geom_ok(Ho, Wo, Ci) && Ci >= w4_min_ci && ws_floats >= wino4_ws_floats(Co, Ci) &&
        wino4_ksplit(B, Ho, Wo, Ci, Co, wsplit ? allow_split : 0, (long)ws_floats) > 0)
      return 13;
    if (wino_ksplit(B, Ho, Wo, Ci, Co, wsplit ? allow_split : 0, (long)ws_floats, min_wgs) > 0) return 9;
  }
  return diagan_conv_gemm_pick_cfg(B * Ho * Wo, Co, Kp, allow_split);
}

// The automatic choice for a launch whose prologue has one affine row per group of `pro_group_rows` GEMM rows (the stacked
// generator forward): a tile must not straddle two groups, so a choice whose tile height does not divide the group falls
// back to the implicit GEMM's pick and then to its 64-row tile (batch 50 at 8x8: 3200 rows per group = 25 x 128, not a
// multiple of the Winograd kernel's 256).  pro_group_rows == 0: same as diagan_conv_gemm_pick_cfg_geom.  This is what
// diagan_conv_gemm itself does for tile_cfg 0, so a caller that sizes the statistics buffer (tile count) or asks whether
// the half-resolution residual will be fused gets the launch's own answer.
DIAGAN_API int diagan_conv_gemm_pick_cfg_grouped(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy,
                                                 int dr, int off, int up, int Kp, int allow_split, int64_t ws_floats,
                                                 int pro_group_rows) {
  int cfg = diagan_conv_gemm_pick_cfg_geom(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, allow_split, ws_floats);
  if (cfg == 13 && pro_group_rows > 0 && pro_group_rows % 512 != 0)       // F(4x4): 512-row tiles; next the F(2x2) kernel's 256
    cfg = pick_cfg_geom_impl(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, allow_split, ws_floats, false);
  if (pro_group_rows > 0 && pro_group_rows % diagan_conv_gemm_tile_rows(cfg) != 0) {
    cfg = diagan_conv_gemm_pick_cfg(B * Ho * Wo, Co, Kp, allow_split);
    if (pro_group_rows % diagan_conv_gemm_tile_rows(cfg) != 0) cfg = 3;
  }
  return cfg;
}

// Run-time form of DIAGAN_WINO (A/B runs and the tests that compare kernels like with like): 0 = implicit GEMM only,
// 1 = Winograd where it qualifies, -1 = back to the environment's choice.
DIAGAN_API int diagan_conv_gemm_get_wino(void) { return g_wino; }
DIAGAN_API int diagan_conv_gemm_set_wino(int mode) {
  DG_REQUIRE(mode >= -1 && mode <= 1, "set_wino: -1, 0 or 1");
  g_wino = mode;
  return DIAGAN_OK;
}

// Do the pooled launches of this geometry (tile_cfg 11 / 12) run on the F(4x4) kernel (25 products per 4x4 tile) rather than on
// conv_wino_pool.hip's F(2x2) kernels (9 per 2x2 tile)?  Host-only; for kernel names / executed-FLOP accounting.
DIAGAN_API int diagan_conv_wino4_pool_used(int B, int Ho, int Wo, int Ci, int Co, int64_t ws_floats) {
  return sel_wino4() != 0 && wino4_pool_ok(B, Ho, Wo, Ci, Co, (long)ws_floats, sel_wino4() == 2) ? 1 : 0;
}

// tile_cfg 15: does conv3x3(bilinear_x2(pro(x))) of this layer (Hi, Wi, Ho, Wo: the UP-SAMPLED size) run as one launch of the
// F(4x4) kernel on the half-resolution input?  Host-only (GBlock asks before it decides whether to write the up-sampled tensor).
DIAGAN_API int diagan_conv_wino4_upin_supported(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                                int off, int up, int64_t ws_floats, int pro_group_rows) {
  const int wino = sel_wino();
  if (!wino || sel_wino4() == 0 || dr != 1 || !diagan_conv_wino_supported(Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up)) return 0;
  if (pro_group_rows > 0 && pro_group_rows % 512 != 0) return 0;
  return wino4_upin_ok(B, Ho, Wo, Ci, Co, (long)ws_floats, sel_wino4() == 2) ? 1 : 0;
}

// Run-time form of DIAGAN_WINO4: 0 = the automatic choice never takes the F(4x4,3x3) kernel (tile_cfg 13), 1 / -1 = where it
// qualifies (like-with-like tests, A/B runs)
DIAGAN_API int diagan_conv_gemm_set_wino4(int mode) {
  DG_REQUIRE(mode >= -1 && mode <= 2, "set_wino4: -1, 0, 1 or 2 (2: also the pooled launches of any size -- tests)");
  g_wino4 = mode;
  return DIAGAN_OK;
}

// Run-time form of DIAGAN_WINO4_X3 (round 5): 1 = the F(4x4,3x3) launches whose K loop is a multiple of four steps run their 36
// frequency GEMMs on the bf16 matrix pipe with every fp32 operand split exactly into three bf16 pieces (conv_wino4.hip, X3);
// 0 = fp32 MFMA; -1 = the environment's choice.  Returns the mode in force through diagan_conv_gemm_get_wino4x.
DIAGAN_API int diagan_conv_gemm_set_wino4x(int mode) {
  DG_REQUIRE(mode >= -1 && mode <= 1, "set_wino4x: -1, 0 or 1");
  wino4_set_x3(mode);
  return DIAGAN_OK;
}
DIAGAN_API int diagan_conv_gemm_get_wino4x(void) { return wino4_get_x3(); }
DIAGAN_API int diagan_conv_gemm_set_x3(int mode) {
  DG_REQUIRE(mode >= -1 && mode <= 1, "set_x3: -1, 0 or 1");
  g_gemm_x3 = mode;
  return DIAGAN_OK;
}
DIAGAN_API int diagan_conv_gemm_get_x3(void) { return gemm_x3_on() ? 1 : 0; }
DIAGAN_API int diagan_conv_gemm_set_x3b(int mode) {
  DG_REQUIRE(mode >= -1 && mode <= 1, "set_x3b: -1, 0 or 1");
  g_gemm_x3b = mode;
  return DIAGAN_OK;
}
DIAGAN_API int diagan_conv_gemm_get_x3b(void) { return gemm_x3b_on() ? 1 : 0; }
namespace diagan { void x3_set_pieces(int n); int x3_pieces(); }
// see include/diagan_hip.h: pieces per operand of the large split-operand kernels (process-level diagnostic / opt-in switch)
DIAGAN_API int diagan_conv_gemm_set_x3_pieces(int n) {
  diagan::x3_set_pieces(n);
  return DIAGAN_OK;
}
DIAGAN_API int diagan_conv_gemm_get_x3_pieces(void) { return diagan::x3_pieces(); }
// tests / diagnostics: 1 = the 128 x 128 form (two workgroups per CU), 2 = the producer / consumer form on 256 x 128 tiles, 0 = automatic
DIAGAN_API int diagan_conv_gemm_x3b_force_form(int form) {
  DG_REQUIRE(form >= 0 && form <= 2, "x3b_force_form: 0, 1 or 2");
  gemm_x3b_force_form(form);
  return DIAGAN_OK;
}
DIAGAN_API int diagan_conv_gemm_out_map(int mul, int offy, int offx, int y0, int y1, int x0, int x1, int OH, int OW) {
  g_map_next = OutMap{mul, offy, offx, y0, y1, x0, x1, OH, OW};
  return DIAGAN_OK;
}
// The tile configuration diagan_conv_gemm ends up with for tile_cfg 0, INCLUDING the upgrades to the split-operand kernels
// (16: conv_gemm_x3.hip, 17: conv_gemm_x3b.hip) that the pick functions do not know: what a caller needs to name the kernel of a
// launch (kernel timers) or to know whether an output map will be honoured.  plain_epilogue: no mask, no per-half scales, no ReLU on
// the residual, no half-resolution residual; has_ws: a workspace of ws_floats floats is handed over.
DIAGAN_API int diagan_conv_gemm_final_cfg(int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr,
                                          int off, int up, int Kp, int allow_split, int64_t ws_floats, int pro_group_rows,
                                          int pro_mode, int plain_epilogue, int want_stats) {
  int cfg = diagan_conv_gemm_pick_cfg_grouped(B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, Kp, allow_split, ws_floats,
                                              pro_group_rows);
  ConvGemmArgs a = {};
  a.pro_mode = pro_mode;
  a.M = B * Ho * Wo;
  a.g = ConvGeom{B, Hi, Wi, Ci, Ho, Wo, Co, R, S, sy, dr, off, up, R * S * Ci, Kp};
  a.pro_group_rows = pro_group_rows;
  static float dummy;
  a.stat_partials = want_stats ? &dummy : nullptr;
  a.mask_src = plain_epilogue ? nullptr : &dummy;
  if (ws_floats > 0 && x3b_takes(a, cfg, ws_floats)) return 17;
  a.mask_src = nullptr;
  if (cfg == 14 && R == 3 && S == 3 && gemm_x3_on() && gemm_x3_geom_ok(a) && ws_floats >= gemm_x3_ws_floats(Co, Kp)) return 16;
  return cfg;
}
DIAGAN_API int diagan_conv_gemm_set_splitk_fused(int mode) {
  DG_REQUIRE(mode >= -1 && mode <= 1, "set_splitk_fused: -1, 0 or 1");
  g_splitk_fused = mode;
  return DIAGAN_OK;
}
// process-wide ticket buffer of the in-kernel split-K combine (diagnostics; the product path hands one over per call through
// diagan_conv_opts or leaves the combine to the second launch): caller-owned, `slots` zero-initialised ints; NULL unregisters
DIAGAN_API int diagan_conv_gemm_set_splitk_tickets(int* buf, int64_t slots) {
  DG_REQUIRE((buf && slots > 0) || (!buf && slots == 0), "set_splitk_tickets: a buffer with its slot count, or NULL and 0");
  g_tickets = buf;
  g_ticket_slots = (long)slots;
  return DIAGAN_OK;
}
struct diagan_conv_opts {     // mirrors the typedef of the same name in include/diagan_hip.h (48 bytes)
  int32_t wino, wino4, wino4x, gemm_x3, gemm_x3b, splitk_fused, force_ksplit, tune;
  int32_t* tickets;
  int64_t ticket_slots;
};
static_assert(sizeof(diagan_conv_opts) == 48, "diagan_conv_opts layout");
// Selection options of the NEXT diagan_conv_gemm call of the calling thread (thread-local, consumed by that call whatever path it takes;
// NULL clears a pending set).  See diagan_conv_opts in include/diagan_hip.h.
DIAGAN_API int diagan_conv_gemm_next_opts(const diagan_conv_opts* o) {
  if (!o) {
    g_opts_next = kNoOpts;
    return DIAGAN_OK;
  }
  DG_REQUIRE(o->wino >= -1 && o->wino <= 1 && o->wino4 >= -1 && o->wino4 <= 2 && o->wino4x >= -1 && o->wino4x <= 1 && o->gemm_x3 >= -1 &&
                 o->gemm_x3 <= 1 && o->gemm_x3b >= -1 && o->gemm_x3b <= 1 && o->splitk_fused >= -1 && o->splitk_fused <= 1 &&
                 o->force_ksplit >= 0 && o->tune >= -1,
             "conv_gemm_next_opts: a field out of range (-1 = process default; force_ksplit 0 = the launch policy)");
  DG_REQUIRE(!o->tickets || o->ticket_slots > 0, "conv_gemm_next_opts: a ticket buffer needs its slot count");
  g_opts_next = CallOpts{o->wino, o->wino4, o->wino4x, o->gemm_x3, o->gemm_x3b, o->splitk_fused, o->force_ksplit, o->tune, o->tickets,
                         (long)o->ticket_slots};
  return DIAGAN_OK;
}
// tile configuration the last diagan_conv_gemm call of the calling thread resolved to (0: it failed before choosing)
DIAGAN_API int diagan_conv_gemm_last_cfg(void) { return g_last_cfg; }

// Diagnostics / tuning sweeps (tools/stamp_report.py, tools/bench_conv.py); never called by the product path.
//  * stamp buffer: while set, diagan_conv_gemm launches the STAMP build of the kernel, which records per workgroup
//    [0] real-time counter (100 MHz) at entry, [1] at exit, [2..6] shader cycles of: loader set-up, first tile
//    (loads + LDS stores + barrier), K loop, epilogue issue, store drain; [7] HW_ID | XCC_ID << 32.
//  * diagan_conv_gemm_tune: force_ksplit > 0 overrides the split-K factor for every tile configuration; flags >= 0
//    overrides the kernel's tune bits (ConvGemmArgs::tune; -1 = production default); lds_delta_bytes is added to the
//    dynamic LDS request (occupancy probe; a negative value is for TIMING ONLY, results are undefined).
DIAGAN_API int diagan_conv_gemm_set_stamp_buffer(unsigned long long* buf, int64_t slots) {
  g_stamps = buf;
  g_stamp_slots = buf ? (long)slots : 0;
  return DIAGAN_OK;
}
DIAGAN_API int diagan_conv_gemm_tune(int force_ksplit, int flags, int lds_delta_bytes) {
  g_force_ksplit = force_ksplit;
  g_tune_flags = flags;
  g_lds_delta = lds_delta_bytes;
  return DIAGAN_OK;
}
