// 3x3 convolution to FOUR output channels (RGB + pad), stride 1, pad 1 -- the generator's last conv
// and the data-gradient of the discriminator's first conv (SURVEY §8 a2/a4 "c5/c6", a3/a5 block1).
//
// On the matrix cores these cost a full 32-wide (64 with the tile) N dimension for 4 useful columns
// (measured 5.5 TFLOP/s, 218 us for M=65536, K=2304).  Here the reduction over K is spread over
// the LANES instead: LPP = Ci/4 lanes share one pixel, each lane owns 4 input channels and keeps
// its 9 x 4 x 4 weights LDS-resident (staged once per workgroup); a 3x3 sliding window of float4 loads (3 new 16-byte loads per
// pixel) feeds 144 FMAs, and the four partial sums are combined across the LPP lanes with wave
// shuffles.  Same gather formula / prologue / epilogue semantics as conv_gemm_kernel.
//
// Roofline: HBM / L2 (each input row is read ~3x); algorithmic FLOP 2*M*4*9*Ci is negligible.
#include "conv_common.h"

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
};

constexpr int SC_RUN = 8;  // consecutive output pixels per lane group (sliding window along x)

template <int LPP>
__global__ __launch_bounds__(256) void conv3x3_co4_kernel(const SmallCoArgs a) {
  constexpr int G = 64 / LPP;  // pixel groups per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cl = lane % LPP, grp = lane / LPP;
  const int runs_per_row = (a.W + SC_RUN - 1) / SC_RUN;
  const long total_runs = (long)a.B * a.H * runs_per_row;
  const int c0 = cl * 4;

  // weights staged once per workgroup in LDS as wl[(dy+1)*3 + (dx+1)][n][Ci] (indexed by the WINDOW
  // offset they multiply, so conv and data-gradient share the inner loop); a lane reads its 16 bytes
  // per (offset, n) with conflict-free ds_read_b128.  Registers stay < 128 -> 4 waves/SIMD hide the
  // global-load latency of the sliding window.
  extern __shared__ __attribute__((aligned(16))) float wl[];
  for (int i = threadIdx.x; i < 9 * 4 * (a.Ci / 4); i += 256) {
    const int c4 = i % (a.Ci / 4), n = (i / (a.Ci / 4)) % 4, tap = i / (a.Ci);   // tap = i / (4 * Ci/4)
    const int r = tap / 3, sx = tap % 3;
    const int slot = (r * a.dr + a.off + 1) * 3 + (sx * a.dr + a.off + 1);
    *reinterpret_cast<f32x4*>(wl + ((long)slot * 4 + n) * a.Ci + c4 * 4) =
        *reinterpret_cast<const f32x4*>(a.w + (long)n * a.Kp + tap * a.Ci + c4 * 4);
  }
  __syncthreads();
  const bool affine = a.pro_mode == PRO_AFFINE_RELU || a.pro_mode == PRO_AFFINE;
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if (affine) {
    psc = *reinterpret_cast<const f32x4*>(a.pro_scale + c0);
    psh = *reinterpret_cast<const f32x4*>(a.pro_shift + c0);
  }
  const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias) : f32x4{0.f, 0.f, 0.f, 0.f};
  // persistent workgroups: the 36*Ci floats of LDS weights are staged once and reused for many runs;
  // a whole lane group leaves the loop together (shuffles stay inside a group)
  for (long run = ((long)blockIdx.x * 4 + wave) * G + grp; run < total_runs; run += (long)gridDim.x * 4 * G) {
  const int xr = (int)(run % runs_per_row);
  const long t = run / runs_per_row;
  const int oy = (int)(t % a.H), b = (int)(t / a.H);
  const int x0 = xr * SC_RUN;
  const float* img = a.x + (long)b * a.H * a.W * a.Ci + c0;
  bool rowok[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) rowok[d] = (oy + d - 1) >= 0 && (oy + d - 1) < a.H;

  // raw loads are branch-free (clamped address) and issued one column AHEAD of their use; the
  // prologue transform and the zero padding are applied when the column enters the window
  auto issue_col = [&](int ix, f32x4 (&raw)[3]) {
    const int ixc = min(max(ix, 0), a.W - 1);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int iyc = min(max(oy + d - 1, 0), a.H - 1);
      raw[d] = *reinterpret_cast<const f32x4*>(img + ((long)iyc * a.W + ixc) * a.Ci);
    }
  };
  auto finish_col = [&](int ix, const f32x4 (&raw)[3], f32x4 (&col)[3]) {
    const bool cok = ix >= 0 && ix < a.W;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      f32x4 v = raw[d];
      if (affine) v = v * psc + psh;
      if (a.pro_mode == PRO_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
      } else if (a.pro_mode == PRO_RELU || a.pro_mode == PRO_AFFINE_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (!(cok && rowok[d])) v = f32x4{0.f, 0.f, 0.f, 0.f};
      col[d] = v;
    }
  };

  f32x4 c_m[3], c_0[3], c_p[3], nxt[3];  // window columns x-1, x, x+1 and the raw column in flight
  issue_col(x0 - 1, nxt);
  finish_col(x0 - 1, nxt, c_m);
  issue_col(x0, nxt);
  finish_col(x0, nxt, c_0);
  issue_col(x0 + 1, nxt);
#pragma unroll 1
  for (int i = 0; i < SC_RUN; ++i) {
    const int ox = x0 + i;
    if (ox >= a.W) break;
    finish_col(ox + 1, nxt, c_p);
    issue_col(ox + 2, nxt);          // lands while this pixel's 144 FMAs and shuffles run
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + ((d * 3 + 0) * 4 + n) * a.Ci + c0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + ((d * 3 + 1) * 4 + n) * a.Ci + c0);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(wl + ((d * 3 + 2) * 4 + n) * a.Ci + c0);
        const f32x4 p = c_m[d] * w0 + c_0[d] * w1 + c_p[d] * w2;
        acc[n] += (p[0] + p[1]) + (p[2] + p[3]);
      }
#pragma unroll
    for (int o = LPP >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] += __shfl_xor(acc[n], o, 64);
    if (cl == 0) {
      const long m = ((long)b * a.H + oy) * a.W + ox;
      f32x4 o4 = {acc[0], acc[1], acc[2], acc[3]};
      o4 += bv;
      if (a.residual) o4 += reinterpret_cast<const f32x4*>(a.residual)[m];
      reinterpret_cast<f32x4*>(a.y)[m] = o4;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) { c_m[d] = c_0[d]; c_0[d] = c_p[d]; }
  }
  }
}

}  // namespace diagan

using namespace diagan;

// 1 if diagan_conv3x3_co4 supports this geometry (host-side check, no device work)
DIAGAN_API int diagan_conv3x3_co4_supported(int Ci, int Co, int R, int S, int sy, int dr, int off, int up) {
  const bool geo = R == 3 && S == 3 && sy == 1 && up == 1 && ((dr == 1 && off == -1) || (dr == -1 && off == 1));
  return geo && Co == 4 && (Ci == 64 || Ci == 128 || Ci == 256);
}

DIAGAN_API int diagan_conv3x3_co4(const float* x, const float* w, float* y, const float* bias, const float* residual,
                                  const float* pro_scale, const float* pro_shift, int pro_mode, int B, int H, int W,
                                  int Ci, int dr, int off, int Kp, void* stream) {
  DG_REQUIRE(x && w && y, "conv3x3_co4: null tensor");
  DG_REQUIRE(diagan_conv3x3_co4_supported(Ci, 4, 3, 3, 1, dr, off, 1), "conv3x3_co4: unsupported geometry Ci=%d dr=%d off=%d", Ci, dr, off);
  DG_REQUIRE(Kp >= 9 * Ci && pro_mode >= 0 && pro_mode <= 4, "conv3x3_co4: bad Kp / pro_mode");
  DG_REQUIRE(!(pro_mode == PRO_AFFINE_RELU || pro_mode == PRO_AFFINE) || (pro_scale && pro_shift), "conv3x3_co4: affine prologue needs scale/shift");
  SmallCoArgs a{x, w, y, bias, residual, pro_scale, pro_shift, pro_mode, B, H, W, Ci, Kp, dr, off};
  const int lpp = Ci / 4, groups = 64 / lpp;
  const long runs = (long)B * H * cdiv(W, SC_RUN);
  int blocks = cdiv(runs, 4L * groups);
  if (blocks > 1024) blocks = 1024;          // persistent: 4 workgroups per CU, each loops over runs
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)36 * Ci * sizeof(float);
  if (lpp == 64) hipLaunchKernelGGL(conv3x3_co4_kernel<64>, dim3(blocks), dim3(256), lds, st, a);
  else if (lpp == 32) hipLaunchKernelGGL(conv3x3_co4_kernel<32>, dim3(blocks), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(conv3x3_co4_kernel<16>, dim3(blocks), dim3(256), lds, st, a);
  return check_launch("conv3x3_co4");
}
