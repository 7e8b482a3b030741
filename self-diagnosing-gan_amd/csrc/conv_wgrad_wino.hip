// Winograd F(3x3, 2x2) weight gradient on the fp32 matrix cores.
//
// Replaces the weight half of conv2d backward (errD.backward() / errG.backward() in the train steps,
// diagan-pkg/diagan/models/topk_models.py:90, mnist.py:126) for the 3x3 / stride 1 / pad 1 layers, like
// conv_wgrad.hip but with 16/36 of its multiply-accumulates.  With the output cut into 2x2 tiles, a tile's
// contribution to dW (3x3) is the correlation of its 4x4 input patch d with its 2x2 gradient tile e, which is the
// transpose of the forward algorithm of conv_wino.hip:
//     dW = G^T [ sum_tiles (A e A^T) .* (B^T d B) ] G          (A, B, G: the F(2x2,3x3) matrices)
// The sum over tiles runs in the transformed domain as 16 GEMMs  N_f[co][ci] += E_f[tile][co] V_f[tile][ci]  (K = tiles);
// G^T . G is applied once per workgroup at the end and the result goes to the same split-K slab, in the same packed
// layout [Co][(r*3+s)*Ci + ci] (+ bias column sums), that conv_wgrad_kernel writes -- the deferred reduction
// (wgrad_finish_*_kernel) does not change.
//
// One workgroup = 512 threads = 8 waves = 64 output channels x 64 input channels x one split of the tiles, one per CU
// (128 KB of LDS, two stages).  K-step = 8 tiles (32 pixels).  Both operands are transformed by the loader:
//   V: thread (tile, channel quad, patch row r) as in conv_wino.hip (prologue, row transform, DPP column transform);
//   E: thread (tile, channel quad, row i of A e A^T) loads the tile's 2x2 gradient pixels and forms its row.
// LDS planes are tile-major [f][8 tiles][64 channels] (16-byte slots XOR-ed with the row index: conflict-free
// ds_write_b128), MFMA fragments are ds_read_b32 (K = tile).  Wave w owns row i = w >> 1 of the frequencies on the
// input-channel half w & 1: 4 x 2 accumulator tiles; the j half of G^T . G is applied in registers, the i half after
// one LDS exchange per 32 output channels.
#include "conv_common.h"
#include <type_traits>

namespace diagan {

constexpr int GW_T = 8;                        // tiles per K-step
constexpr int GW_PLANE = GW_T * 64;            // floats of one frequency plane: 8 tiles x 64 channels
constexpr int GW_STAGE = 2 * 16 * GW_PLANE;    // V planes + E planes

// bid / nblk: this workgroup's index and the workgroup count of ITS layer (the whole grid for a one-layer launch; a contiguous
// range of the grid in a batched launch of several layers, conv_wgrad_wino_batched_kernel below)
template <int PRO>
__device__ __forceinline__ void wgrad_wino_body(const WgradArgs& a, const int bid, const int nblk, float* __restrict__ smem) {
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_ci = (g.Ci + 63) / 64;
  const int logical = xcd_remap(bid, nblk);
  const int split = logical / a.tiles, tile = logical - split * a.tiles;
  const int n0 = (tile / tiles_ci) * 64, c0 = (tile % tiles_ci) * 64;      // first output / input channel
  const int TW = g.Wo >> 1, TH = g.Ho >> 1;
  const int MT = g.B * TH * TW;
  const bool affine = PRO == PRO_AFFINE_RELU || PRO == PRO_AFFINE;
  // K-step range of this split (a.seg_steps etc. count K-steps of 8 tiles = 32 pixels, as in conv_wgrad_kernel)
  const int seg = split / a.splits_per_seg, sidx = split - seg * a.splits_per_seg;
  const int total_steps = (MT + GW_T - 1) / GW_T;
  const int s_begin = seg * a.seg_steps + sidx * a.steps_per_split;
  const int s_end = min(min(s_begin + a.steps_per_split, (seg + 1) * a.seg_steps), total_steps);

  // ---- loader role: tile lt of the K-step, channel quad cq, row lr (patch row of V / row i of A e A^T) ----
  const int lr = tid & 3, cq = (tid >> 2) & 15, lt = wave;
  const bool ci_ok = c0 + cq * 4 < g.Ci, co_ok = n0 + cq * 4 < g.Co;
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if (affine && ci_ok) {
    psc = *reinterpret_cast<const f32x4*>(a.pro_scale + c0 + cq * 4);
    psh = *reinterpret_cast<const f32x4*>(a.pro_shift + c0 + cq * 4);
  }
  // column transform of V: own + sc * partner (quad_perm [2,2,1,1]); row 3 is staged negated, as in conv_wino.hip, and
  // the sign is undone in E's row 3 (ea1 below)
  const float sc = lr == 1 ? 1.f : -1.f;
  // row lr of A e A^T: (A e)[lr][x] = ea0 * e[0][x] + ea1 * e[1][x], A = [[1,0],[1,1],[1,-1],[0,-1]]; row 3 negated (+1).
  // Column 3 (-a1) is staged as +a1: the sign moves into the epilogue's m3 terms.
  const float ea0 = lr == 3 ? 0.f : 1.f, ea1 = lr == 0 ? 0.f : (lr == 2 ? -1.f : 1.f);
  const int slot = (cq ^ (lr << 1)) * 4;                 // 16-byte slot of this thread's writes (row planes 4 lr .. 4 lr + 3)
  const bool want_bias = a.bias_off >= 0 && c0 == 0;
  f32x4 bacc = {0.f, 0.f, 0.f, 0.f};

  // Loads.  The tile of a K-step is the wave's (lt = wave): its coordinates, the base addresses and the border conditions
  // are wave-uniform and live in scalar registers; a lane contributes only loop-invariant byte offsets relative to the
  // tile's patch origin (pixel (2 ty - 1, 2 tx - 1)) resp. gradient origin (2 ty, 2 tx), bit 31 set where its channels do
  // not exist.  Vector instructions do not hide behind the MFMAs (profiles/r02_wino_ablation.md): per step the address
  // side costs two ANDs and four ORs.
  unsigned xoff[4], eoff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) xoff[c] = ci_ok ? (unsigned)(((lr * g.Wi + c) * g.Ci + c0 + cq * 4) * 4) : 0x80000000u;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    eoff[p] = co_ok ? (unsigned)((((p >> 1) * g.Wo + (p & 1)) * g.Co + n0 + cq * 4) * 4) : 0x80000000u;
  const unsigned L0 = lr == 0 ? 0xffffffffu : 0u, L3 = lr == 3 ? 0xffffffffu : 0u;

  f32x4 rx[4], re[4];
  unsigned vo[4];                                        // bit 31: this patch pixel is padding (or not there at all)
  auto issue_loads = [&](int ks) {
    const int gt = ks * GW_T + lt;
    const bool tv = gt < MT;
    const unsigned q1 = fdiv((unsigned)(tv ? gt : 0), a.dWo);          // dWo: divisor TW, dHo: divisor TH
    const int tx = (tv ? gt : 0) - (int)q1 * TW;
    const unsigned b = fdiv(q1, a.dHo);
    const int ty = (int)q1 - (int)b * TH;
    // descriptors based AT the tile (a tile beyond the batch: zero records, every load returns zeros)
    const long xpix = (long)((int)b * g.Hi + 2 * ty - 1) * g.Wi + (2 * tx - 1);
    const long epix = (long)((int)b * g.Ho + 2 * ty) * g.Wo + 2 * tx;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + xpix * g.Ci, 0,
                                                                          tv ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + epix * g.Co, 0,
                                                                          tv ? 0x7fffffff : 0, 0x00020000);
    const unsigned B0 = ty == 0 ? 0x80000000u : 0u, B3 = ty == TH - 1 ? 0x80000000u : 0u;
    const unsigned C0 = tx == 0 ? 0x80000000u : 0u, C3 = tx == TW - 1 ? 0x80000000u : 0u;
    const unsigned rowbits = (L0 & B0) | (L3 & B3);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      vo[c] = xoff[c] | rowbits | (c == 0 ? C0 : (c == 3 ? C3 : 0u));
      rx[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, vo[c], 0, 0));
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) re[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ysrc, eoff[p], 0, 0));
  };
  float t[4][4];
  auto transform_v = [&]() {                // prologue + row transform of the 4 patch pixels of this row
    f32x4 d[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 v = rx[c];
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
          for (int e = 0; e < 4; ++e) {               // max(v, 0) as one v_max_i32 on the bits
            const float q = v[e];
            v[e] = __int_as_float(max(__float_as_int(q), 0));
          }
        } else {
          // padding is zero AFTER the transform: ReLU + that as ONE v_med3_i32 (clamp of the bits to [0, padding ? 0 : max])
          const int kb = (int)vo[c] < 0 ? 0 : 0x7fffffff;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q = v[e];
            float r;
            if (PRO == PRO_AFFINE_RELU) asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(q), "v"(kb));
            else r = kb ? q : 0.f;
            v[e] = r;
          }
        }
      }
      d[c] = v;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[0][e] = d[0][e] - d[2][e];
      t[1][e] = d[1][e] + d[2][e];
      t[2][e] = d[2][e] - d[1][e];
      t[3][e] = d[1][e] - d[3][e];
    }
  };
  float* const vst = smem + (lr * 4 * GW_T + lt) * 64 + slot;           // plane (lr, j = 0) of stage 0, this thread's slot
  auto store_v = [&](int stage, int j) {    // column transform + one frequency plane: ONE v_fmac_f32_dpp per element
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
    *reinterpret_cast<f32x4*>(vst + stage * GW_STAGE + j * GW_PLANE) = o;
  };
  auto store_e = [&](int stage) {           // row lr of A e A^T: four frequency planes (the last one with its sign flipped)
    float* es = vst + stage * GW_STAGE + 16 * GW_PLANE;
    const f32x4 a0 = ea0 * re[0] + ea1 * re[2], a1 = ea0 * re[1] + ea1 * re[3];
    *reinterpret_cast<f32x4*>(es + 0 * GW_PLANE) = a0;
    *reinterpret_cast<f32x4*>(es + 1 * GW_PLANE) = a0 + a1;
    *reinterpret_cast<f32x4*>(es + 2 * GW_PLANE) = a0 - a1;
    *reinterpret_cast<f32x4*>(es + 3 * GW_PLANE) = a1;
    if (want_bias && lr == 0) bacc += (re[0] + re[1]) + (re[2] + re[3]);
  };

  // wave w: row wi = w >> 1 of the frequencies (f = 4 wi + j), output channels 2 x 32 (tm), input channels tn = w & 1
  const int wi = wave >> 1, tn = wave & 1;
  f32x16 acc[4][2];
#pragma unroll
  for (int fl = 0; fl < 4; ++fl)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[fl][i][e] = 0.f;
  const int fi = lane & 31, fh = lane >> 5;
  // fragment columns (channel index inside the plane row, un-swizzling the writer's slot XOR with the row index wi)
  int acol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) acol[i] = ((((i * 32 + fi) >> 2) ^ (wi << 1)) << 2) | (fi & 3);
  const int bcol = ((((tn * 32 + fi) >> 2) ^ (wi << 1)) << 2) | (fi & 3);

  if (s_begin < s_end) {
    issue_loads(s_begin);
    transform_v();
#pragma unroll
    for (int j = 0; j < 4; ++j) store_v(0, j);
    store_e(0);
  }
  __syncthreads();

  // fragment addresses: one base per operand (+ the stage); everything else is an instruction offset
  const float* const fb_base = smem + (wi * 4 * GW_T + fh) * 64 + bcol;
  const float* const fa_base0 = smem + 16 * GW_PLANE + (wi * 4 * GW_T + fh) * 64 + acol[0];
  const float* const fa_base1 = smem + 16 * GW_PLANE + (wi * 4 * GW_T + fh) * 64 + acol[1];
  auto kstep = [&](int ks, auto has_next) {
    const int cur = (ks - s_begin) & 1;
    if (decltype(has_next)::value) issue_loads(ks + 1);
    // all 48 fragment words of the step are requested before the first MFMA (two groups of 24: the second while the first
    // is consumed) instead of 12 per frequency with a wait each: the LDS round trip is paid once per step, not four times
    float fa[4][2][4], fb[4][4];
    auto read_frags = [&](int fl) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = cur * GW_STAGE + (fl * GW_T + 2 * e) * 64;
        fa[fl][0][e] = fa_base0[row];
        fa[fl][1][e] = fa_base1[row];
        fb[fl][e] = fb_base[row];
      }
    };
    read_frags(0);
    read_frags(1);
#pragma unroll
    for (int fl = 0; fl < 4; ++fl) {
      if (fl == 0) {
        __builtin_amdgcn_sched_barrier(0);
        read_frags(2);
        read_frags(3);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int grp = fl * 4 + e;
        if (decltype(has_next)::value && grp >= 8 && grp < 14) {
          __builtin_amdgcn_sched_barrier(0);
          if (grp == 8) transform_v();
          else if (grp == 9) store_e(cur ^ 1);
          else store_v(cur ^ 1, grp - 10);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[fl][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fl][i][e], fb[fl][e], acc[fl][i], 0, 0, 0);
      }
    }
    __syncthreads();
  };
  for (int ks = s_begin; ks + 1 < s_end; ++ks) kstep(ks, std::true_type{});
  if (s_begin < s_end) kstep(s_end - 1, std::false_type{});

  // ---- epilogue: dW = G^T N G.  j half in registers (3 values per row), i half after an LDS exchange per co half ----
  float* out = a.slab + (long)split * a.slab_stride;
  float* ss = smem;                                      // [4 i][3 s][32 co][64 ci] = 96 KB
  const int eo = tid >> 4, ec = (tid & 15) * 4;         // this thread's output channel (within the half) and input quad
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * fh;
      const float m0 = acc[0][tm][e], m1 = acc[1][tm][e], m2 = acc[2][tm][e], m3 = acc[3][tm][e];
      const float h1 = 0.5f * m1, h2 = 0.5f * m2;
      ss[((wi * 3 + 0) * 32 + row) * 64 + tn * 32 + fi] = m0 + h1 + h2;
      ss[((wi * 3 + 1) * 32 + row) * 64 + tn * 32 + fi] = h1 - h2;
      ss[((wi * 3 + 2) * 32 + row) * 64 + tn * 32 + fi] = h1 + h2 - m3;       // (column 3 of E is staged with its sign flipped)
    }
    __syncthreads();
    const int co = n0 + tm * 32 + eo, ci = c0 + ec;
    if (co < g.Co && ci < g.Ci) {
      float* orow = out + (long)co * g.Kp + ci;
#pragma unroll
      for (int sx = 0; sx < 3; ++sx) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(ss + ((0 * 3 + sx) * 32 + eo) * 64 + ec);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(ss + ((1 * 3 + sx) * 32 + eo) * 64 + ec);
        const f32x4 q2 = *reinterpret_cast<const f32x4*>(ss + ((2 * 3 + sx) * 32 + eo) * 64 + ec);
        const f32x4 q3 = *reinterpret_cast<const f32x4*>(ss + ((3 * 3 + sx) * 32 + eo) * 64 + ec);
        const f32x4 h1 = 0.5f * q1, h2 = 0.5f * q2;
        *reinterpret_cast<f32x4*>(orow + (0 * 3 + sx) * g.Ci) = q0 + h1 + h2;
        *reinterpret_cast<f32x4*>(orow + (1 * 3 + sx) * g.Ci) = h1 - h2;
        *reinterpret_cast<f32x4*>(orow + (2 * 3 + sx) * g.Ci) = h1 + h2 + q3;
      }
    }
    __syncthreads();
  }
  // zero padding of the packed rows (Kp > 9 Ci): written once per output-channel block, by the workgroup of its last
  // input-channel block (the deferred reduction sums whole [Co][Kp] slabs)
  if (c0 + 64 >= g.Ci && g.Kp > 9 * g.Ci) {
    const int npad4 = (g.Kp - 9 * g.Ci) >> 2;
    for (int idx = tid; idx < 64 * npad4; idx += 512) {
      const int co = n0 + idx / npad4;
      if (co < g.Co) *reinterpret_cast<f32x4*>(out + (long)co * g.Kp + 9 * g.Ci + (idx % npad4) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (want_bias) {
    float* red = smem;                                   // [8 tiles-of-the-step lanes = waves][64 channels]
    if (lr == 0) *reinterpret_cast<f32x4*>(red + wave * 64 + cq * 4) = bacc;
    __syncthreads();
    if (tid < 64 && n0 + tid < g.Co) {
      float tsum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) tsum += red[w * 64 + tid];
      out[a.bias_off + n0 + tid] = tsum;
    }
  }
}

template <int PRO>
__global__ __launch_bounds__(512, 2) void conv_wgrad_wino_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [2 stages][V 16 planes | E 16 planes] = 128 KB
  wgrad_wino_body<PRO>(a, blockIdx.x, gridDim.x, smem);
}

// Round 5: the weight gradients of SEVERAL layers of one backward pass in ONE launch.  A layer's own launch fills the chip by
// splitting its pixel range 256 / tiles ways, which for the small maps means 4-16 K-steps per workgroup behind ~15 us of
// fixed cost (prologue, G^T . G, 147 KB of slab per workgroup) and 38 MB of slab per layer whatever its size; deferred to the
// end of the pass (the host keeps dy and x alive) the layers share the 256 workgroups in proportion to their work, every
// workgroup runs ~(sum of all K-steps) / 256 steps and a small layer writes 8 slabs instead of 64.  The layers of a launch
// share the prologue mode; a workgroup finds its layer through the first-workgroup prefix (multiples of 8: the XCD remap
// inside a layer's range stays a permutation of that range).
constexpr int GW_BATCH_MAX = WG_BATCH_MAX;
typedef WgradBatchArgs WgradWinoBatch;
template <int PRO>
__global__ __launch_bounds__(512, 2) void conv_wgrad_wino_batched_kernel(const WgradWinoBatch b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int j = 0;
  while (j + 1 < b.n && (int)blockIdx.x >= b.blk0[j + 1]) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  const int local = (int)blockIdx.x - b.blk0[j];
  if (local >= b.cnt[j]) return;                       // (padding of a range to a multiple of 8)
  wgrad_wino_body<PRO>(b.a[j], local, b.cnt[j], smem);
}

// the layers the Winograd weight gradient takes: 3x3 / stride 1 / pad 1 forward geometry, even H and W, channels % 4
bool wgrad_wino_supported(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int sy, int dr, int off, int up,
                          int Kp) {
  return R == 3 && S == 3 && sy == 1 && up == 1 && dr == 1 && off == -1 && Hi == Ho && Wi == Wo && !(Ho & 1) && !(Wo & 1) &&
         (Ci & 3) == 0 && (Co & 3) == 0 && Ci >= 16 && Co >= 16;
}

// split count: one round of 256 workgroups when the tile count allows it, at least 4 K-steps (32 tiles) per split
// (128 x 8x8 x 128 -> 128: 64 splits of 4 steps 22.4 us, 32 splits of 8 steps 29.6; everywhere else the 256-workgroup
// round is the optimum of a sweep over 1/4 .. 4x the split count)
int wgrad_wino_splits(int B, int Ho, int Wo, int Ci, int Co) {
  const long MT = (long)B * (Ho >> 1) * (Wo >> 1);
  const long steps = (MT + GW_T - 1) / GW_T;
  const int tiles = cdiv(Co, 64) * cdiv(Ci, 64);
  long s = tiles >= 256 ? 1 : (256 + tiles - 1) / tiles;
  if (s > steps / 4) s = steps / 4;
  if (s > 256) s = 256;
  return s < 1 ? 1 : (int)s;
}

template <int PRO>
static int launch_gw(const WgradArgs& a, int wgs, hipStream_t st) {
  const size_t lds = (size_t)2 * GW_STAGE * sizeof(float);
  auto kern = conv_wgrad_wino_kernel<PRO>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, lds);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, st, a);
  return DIAGAN_OK;
}

// `a` as prepared by diagan_conv_wgrad (M, g, slab, strides, prologue); the K-step bookkeeping is redone for 8-tile steps
int launch_wgrad_wino(WgradArgs a, int splits, int segments, hipStream_t st) {
  const ConvGeom& g = a.g;
  a.dWo = make_fastdiv((unsigned)(g.Wo >> 1));
  a.dHo = make_fastdiv((unsigned)(g.Ho >> 1));
  const int MT = g.B * (g.Ho >> 1) * (g.Wo >> 1);
  const int total_steps = cdiv(MT, GW_T);
  a.seg_steps = cdiv(total_steps, segments);
  a.splits_per_seg = splits / segments;
  a.steps_per_split = cdiv(a.seg_steps, a.splits_per_seg);
  a.tiles = cdiv(g.Co, 64) * cdiv(g.Ci, 64);
  const int wgs = a.tiles * splits;
  int rc;
  switch (a.pro_mode) {
    case PRO_NONE: rc = launch_gw<PRO_NONE>(a, wgs, st); break;
    case PRO_RELU: rc = launch_gw<PRO_RELU>(a, wgs, st); break;
    case PRO_AFFINE_RELU: rc = launch_gw<PRO_AFFINE_RELU>(a, wgs, st); break;
    case PRO_LRELU: rc = launch_gw<PRO_LRELU>(a, wgs, st); break;
    default: rc = launch_gw<PRO_AFFINE>(a, wgs, st); break;
  }
  if (rc != DIAGAN_OK) return rc;
  return check_launch("conv_wgrad_wino");
}

// several layers in one launch: a[j] as prepared by diagan_conv_wgrad for layer j (same prologue mode), splits[j] a multiple of
// segments[j]
template <int PRO>
static int launch_gw_batched(const WgradWinoBatch& b, int wgs, hipStream_t st) {
  const size_t lds = (size_t)2 * GW_STAGE * sizeof(float);
  auto kern = conv_wgrad_wino_batched_kernel<PRO>;
  static FuncAttrLatch latch;
  DG_LDS(latch, kern, lds);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, st, b);
  return DIAGAN_OK;
}
int wgrad_wino_batch_max() { return GW_BATCH_MAX; }
int launch_wgrad_wino_batched(const WgradArgs* jobs, const int* splits, const int* segments, int n, hipStream_t st) {
  if (n < 1 || n > GW_BATCH_MAX) return set_err(DIAGAN_EINVAL, "conv_wgrad_batched: 1 .. %d layers per launch, got %d", GW_BATCH_MAX, n);
  WgradWinoBatch b;
  b.n = n;
  int wgs = 0;
  for (int j = 0; j < n; ++j) {
    WgradArgs a = jobs[j];
    if (a.pro_mode != jobs[0].pro_mode) return set_err(DIAGAN_EINVAL, "conv_wgrad_batched: the layers of a launch share the prologue mode");
    const ConvGeom& g = a.g;
    a.dWo = make_fastdiv((unsigned)(g.Wo >> 1));
    a.dHo = make_fastdiv((unsigned)(g.Ho >> 1));
    const int MT = g.B * (g.Ho >> 1) * (g.Wo >> 1);
    const int total_steps = cdiv(MT, GW_T);
    a.seg_steps = cdiv(total_steps, segments[j]);
    a.splits_per_seg = splits[j] / segments[j];
    a.steps_per_split = cdiv(a.seg_steps, a.splits_per_seg);
    a.tiles = cdiv(g.Co, 64) * cdiv(g.Ci, 64);
    b.a[j] = a;
    b.blk0[j] = wgs;
    b.cnt[j] = a.tiles * splits[j];
    wgs += (b.cnt[j] + 7) & ~7;
  }
  for (int j = n; j < GW_BATCH_MAX; ++j) b.blk0[j] = wgs, b.cnt[j] = 0;
  int rc;
  switch (jobs[0].pro_mode) {
    case PRO_NONE: rc = launch_gw_batched<PRO_NONE>(b, wgs, st); break;
    case PRO_RELU: rc = launch_gw_batched<PRO_RELU>(b, wgs, st); break;
    case PRO_AFFINE_RELU: rc = launch_gw_batched<PRO_AFFINE_RELU>(b, wgs, st); break;
    case PRO_LRELU: rc = launch_gw_batched<PRO_LRELU>(b, wgs, st); break;
    default: rc = launch_gw_batched<PRO_AFFINE>(b, wgs, st); break;
  }
  if (rc != DIAGAN_OK) return rc;
  return check_launch("conv_wgrad_wino_batched");
}

}  // namespace diagan
