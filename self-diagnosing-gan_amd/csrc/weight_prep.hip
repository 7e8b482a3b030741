// Spectral normalisation (power iteration, sigma, scaled weight packing) and its backward.
//
// Replaces torch_mimicry's SpectralNorm.sn_weights() as used by SNConv2d / SNLinear in the SNGAN
// discriminators (reached from diagan-pkg/diagan/models/predefined_models.py:38-40,
// 76-78), i.e. per forward:
//     W = weight.view(Co, -1);  v = normalize(u W);  u' = normalize(v W^T)   (no grad, eps 1e-12)
//     sigma = u' W v^T;  conv uses W / sigma;  buffers u, sigma updated in training mode
// and the autograd backward through W / sigma (u', v constants):
//     dL/dW = (G - <G, W/sigma> u'^T v) / sigma,   G = dL/d(W/sigma)
//
// W is the master weight in packed layout [Co][Kp] (zero padded columns: they stay zero).
// Outputs of the forward: Wf = W/sigma in the same packed layout (forward GEMM operand) and
// Wd[ci][(r,s,co)] = W[co][(r,s,ci)]/sigma (data-gradient GEMM operand).  W/sigma is only ever
// materialised in these two GEMM-ready forms.
//
// Roofline: HBM (W is read 3x, written 2x per forward; <= 19 MB per layer).
#include "conv_common.h"

namespace diagan {

// v_raw[k] = sum_n u[n] W[n][k]     (grid over 256-column strips)
__global__ __launch_bounds__(256) void sn_gemv_cols_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                           float* __restrict__ v_raw, int Co, int Kp) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= Kp) return;
  float s = 0.f;
  for (int n = 0; n < Co; ++n) s = fmaf(u[n], W[(long)n * Kp + k], s);
  v_raw[k] = s;
}

// t_raw[n] = sum_k W[n][k] v_raw[k]  (one wave per row)
__global__ __launch_bounds__(256) void sn_gemv_rows_kernel(const float* __restrict__ W, const float* __restrict__ v_raw,
                                                           float* __restrict__ t_raw, int Co, int Kp) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= Co) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int k = lane * 4; k < Kp; k += 256) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)n * Kp + k);
    const f32x4 v = *reinterpret_cast<const f32x4*>(v_raw + k);
    s += w[0] * v[0] + w[1] * v[1] + w[2] * v[2] + w[3] * v[3];
  }
  s = wave_sum(s);
  if (lane == 0) t_raw[n] = s;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// single block: norms, v, u', sigma.   state[0] = sigma, state[1] = 1/sigma
__global__ __launch_bounds__(256) void sn_finalize_kernel(const float* __restrict__ v_raw, const float* __restrict__ t_raw,
                                                          float* __restrict__ v_out, float* __restrict__ u_out,
                                                          float* __restrict__ u_buffer, float* __restrict__ sigma_buffer,
                                                          float* __restrict__ state, int Co, int Kp, float eps,
                                                          int update_buffers) {
  __shared__ float red[4];
  float s = 0.f;
  for (int k = threadIdx.x; k < Kp; k += 256) s += v_raw[k] * v_raw[k];
  const float nv = fmaxf(sqrtf(block_sum_256(s, red)), eps);
  for (int k = threadIdx.x; k < Kp; k += 256) v_out[k] = v_raw[k] / nv;
  s = 0.f;
  for (int n = threadIdx.x; n < Co; n += 256) { const float t = t_raw[n] / nv; s += t * t; }
  const float nt = fmaxf(sqrtf(block_sum_256(s, red)), eps);
  s = 0.f;
  for (int n = threadIdx.x; n < Co; n += 256) {
    const float t = t_raw[n] / nv;
    const float un = t / nt;
    u_out[n] = un;
    if (update_buffers) u_buffer[n] = un;
    s += un * t;
  }
  const float sigma = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    state[0] = sigma;
    state[1] = 1.f / sigma;
    if (update_buffers) sigma_buffer[0] = sigma;
  }
}

// Wf = W * inv (same layout) and/or Wd[ci][(rs, co)] = W[co][(rs, ci)] * inv ; inv read from device
// tile: 32 (co) x 32 (ci) per tap through LDS so both sides are coalesced
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ W, const float* __restrict__ inv_ptr,
                                                           float* __restrict__ Wf, float* __restrict__ Wd,
                                                           int Co, int Ci, int RS, int Kp, int Kd) {
  __shared__ float tile[32][33];
  const float inv = inv_ptr ? inv_ptr[0] : 1.f;
  const int tap = blockIdx.z;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < Co && ci < Ci) {
      v = W[(long)co * Kp + tap * Ci + ci] * inv;
      if (Wf) Wf[(long)co * Kp + tap * Ci + ci] = v;
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  if (Wd) {
    for (int r = ty; r < 32; r += 8) {
      const int ci = ci0 + r, co = co0 + tx;
      if (ci < Ci && co < Co) Wd[(long)ci * Kd + tap * Co + co] = tile[tx][r];
    }
  }
}

// grad[n][k] += (G[n][k] - (dot/sigma) * u[n] * v[k]) / sigma ; dot = sum(partials) = <G, W>
__global__ __launch_bounds__(256) void sn_grad_fix_kernel(const float* __restrict__ G, const double* __restrict__ partials,
                                                          int nparts, const float* __restrict__ u,
                                                          const float* __restrict__ v, const float* __restrict__ state,
                                                          float* __restrict__ grad, int Co, int Kp, int accumulate) {
  __shared__ double sdot;
  if (threadIdx.x == 0) {
    double d = 0.0;
    for (int i = 0; i < nparts; ++i) d += partials[i];  // fixed order: deterministic
    sdot = d;
  }
  __syncthreads();
  const float inv = state[1];
  const float coef = (float)(sdot * (double)inv);  // <G, W/sigma>
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (long)Co * Kp) return;
  const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
  const f32x4 g = *reinterpret_cast<const f32x4*>(G + i);
  const f32x4 vv = *reinterpret_cast<const f32x4*>(v + k);
  const float un = u[n] * coef;
  f32x4 o = (g - un * vv) * inv;
  if (accumulate) o += *reinterpret_cast<const f32x4*>(grad + i);
  *reinterpret_cast<f32x4*>(grad + i) = o;
}

}  // namespace diagan

using namespace diagan;

// One SN forward for one layer.  work: >= (Kp + Co) floats of scratch.
DIAGAN_API int diagan_sn_power_iter(const float* W, float* u_buffer, float* sigma_buffer, float* u_out,
                                    float* v_out, float* state, float* work, int Co, int Kp, float eps,
                                    int update_buffers, void* stream) {
  DG_REQUIRE(W && u_buffer && sigma_buffer && u_out && v_out && state && work, "sn_power_iter: null pointer");
  DG_REQUIRE(Co > 0 && Kp > 0 && (Kp & 3) == 0, "sn_power_iter: bad dims Co=%d Kp=%d", Co, Kp);
  hipStream_t st = (hipStream_t)stream;
  float* v_raw = work;
  float* t_raw = work + Kp;
  hipLaunchKernelGGL(sn_gemv_cols_kernel, dim3(cdiv(Kp, 256)), dim3(256), 0, st, W, u_buffer, v_raw, Co, Kp);
  hipLaunchKernelGGL(sn_gemv_rows_kernel, dim3(cdiv(Co, 4)), dim3(256), 0, st, W, v_raw, t_raw, Co, Kp);
  hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(256), 0, st, v_raw, t_raw, v_out, u_out, u_buffer,
                     sigma_buffer, state, Co, Kp, eps, update_buffers);
  return check_launch("sn_power_iter");
}

// Wf (optional) = W*inv_sigma, Wd (optional) = transposed pack * inv_sigma.  inv_sigma: device ptr or NULL (=1)
DIAGAN_API int diagan_pack_weights(const float* W, const float* inv_sigma, float* Wf, float* Wd, int Co, int Ci,
                                   int RS, int Kp, int Kd, void* stream) {
  DG_REQUIRE(W && (Wf || Wd), "pack_weights: null pointer");
  DG_REQUIRE(Co > 0 && Ci > 0 && RS > 0 && Kp >= RS * Ci && (!Wd || Kd >= RS * Co), "pack_weights: bad dims");
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(Ci, 32), cdiv(Co, 32), RS), dim3(256), 0, (hipStream_t)stream, W,
                     inv_sigma, Wf, Wd, Co, Ci, RS, Kp, Kd);
  return check_launch("pack_weights");
}

DIAGAN_API int diagan_sn_grad_fix(const float* G, const double* dot_partials, int nparts, const float* u,
                                  const float* v, const float* state, float* grad, int Co, int Kp, int accumulate,
                                  void* stream) {
  DG_REQUIRE(G && dot_partials && u && v && state && grad, "sn_grad_fix: null pointer");
  DG_REQUIRE(Co > 0 && Kp > 0 && (Kp & 3) == 0 && nparts > 0, "sn_grad_fix: bad dims");
  hipLaunchKernelGGL(sn_grad_fix_kernel, dim3(cdiv((long)Co * Kp / 4, 256)), dim3(256), 0, (hipStream_t)stream, G,
                     dot_partials, nparts, u, v, state, grad, Co, Kp, accumulate);
  return check_launch("sn_grad_fix");
}

// ================================================================================================
// StyleGAN2 weight preparation in one launch each (round 6).  The autograd ops of diagan/ops/diffconv.py take the packed operand
// Wp[Co][Kp] of a reference-shaped OIHW parameter times the equalised-learning-rate scale (reference: `self.weight * self.scale`
// in front of F.conv2d, diagan-pkg/diagan/models/stylegan2.py:94-129,224-265); built from torch ops that was a scalar multiply, a
// permuted copy, up to two pads -- and in the backward a zero fill + the transposed pack for the data gradient -- per layer and pass:
// ~1100 of a StyleGAN2 iteration's ~5200 launches, 4-8 us each.
//   diagan_pack_oihw      w[Co'][Ci'][R][S] * scale -> Wp[Co][Kp] (k = (r S + s) Ci + c, zero-padded) and, optionally, the
//                         data-gradient operand Wd[Ci][Kd] (k = (r S + s) Co + n) of the same values
//   diagan_unpack_oihw    the adjoint: gw[Co'][Ci'][R][S] = scale * gWp[co][(r S + s) Ci + c]  (pack is linear: its backward, and the
//                         backward of that, are these two kernels again)
//   diagan_parity_weights the sub-kernels w[:, cy::2, cx::2] (taps in correlation order, diffconv._parity_taps) of the four parity
//                         classes of a stride-2 transposed gather, each zero-padded to its own Kp, into one buffer
//   diagan_parity_weights_adjoint   the weight gradient of the transposed convolution from the four classes' weight gradients
// ================================================================================================
namespace diagan {

__global__ __launch_bounds__(256) void pack_oihw_kernel(const float* __restrict__ w, float scale, float* __restrict__ Wf,
                                                        float* __restrict__ Wd, int Cos, int Cis, int R, int S, int Co, int Ci, int Kp,
                                                        int Kd) {
  const long total = (long)Co * Kp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
    const int tap = k / Ci, c = k - tap * Ci;
    float v = 0.f;
    if (n < Cos && tap < R * S && c < Cis) v = w[((long)n * Cis + c) * (R * S) + tap] * scale;
    if (Wf) Wf[i] = v;
    if (Wd && tap < R * S) Wd[(long)c * Kd + tap * Co + n] = v;
  }
}
// (the padding columns of Wd beyond R S Co are zeroed by the first Ci x (Kd - R S Co) threads)
__global__ __launch_bounds__(256) void pack_oihw_wd_pad_kernel(float* __restrict__ Wd, int Ci, int Kd, int used) {
  const int padw = Kd - used;
  const long total = (long)Ci * padw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) Wd[(i / padw) * Kd + used + (i % padw)] = 0.f;
}

__global__ __launch_bounds__(256) void unpack_oihw_kernel(const float* __restrict__ gWp, float scale, float* __restrict__ gw, int Cos,
                                                          int Cis, int R, int S, int Ci, int Kp) {
  const int RS = R * S;
  const long total = (long)Cos * Cis * RS;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int tap = (int)(i % RS);
    const long t = i / RS;
    const int c = (int)(t % Cis), n = (int)(t / Cis);
    gw[i] = gWp[(long)n * Kp + tap * Ci + c] * scale;
  }
}

struct ParityGeom {
  int n, R, S, C, Kp;          // source: w[n][Kp], k = (r S + s) C + c
  int off[4], kp[4];           // per class (cy, cx) = (cls >> 1, cls & 1): float offset of its matrix in the buffer, its row length
  int ny[2], nx[2];            // taps per parity: ceil((R - cy) / 2), ceil((S - cx) / 2)
};
// class matrix row n: k' = (u nx + v) C + c  <-  source tap (r, s) = (taps_y[u], taps_x[v]), taps = range(parity, size, 2) reversed
__global__ __launch_bounds__(256) void parity_weights_kernel(const float* __restrict__ w, float* __restrict__ out, const ParityGeom g,
                                                             int adjoint) {
  const int cls = blockIdx.y, cy = cls >> 1, cx = cls & 1;
  const int ny = g.ny[cy], nx = g.nx[cx];
  if (ny <= 0 || nx <= 0) return;
  const int kp = g.kp[cls];
  const long total = (long)g.n * kp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / kp), k = (int)(i - (long)n * kp);
    const int uv = k / g.C, c = k - uv * g.C;
    const bool live = uv < ny * nx;
    const int u = live ? uv / nx : 0, v = live ? uv - u * nx : 0;
    const int r = cy + 2 * (ny - 1 - u), s = cx + 2 * (nx - 1 - v);
    const long src = (long)n * g.Kp + (long)(r * g.S + s) * g.C + c;
    if (!adjoint) out[g.off[cls] + i] = live ? w[src] : 0.f;
    else if (live) out[src] = w[g.off[cls] + i];          // (`w`: the classes' buffer, `out`: the full operand's gradient)
  }
}
__global__ __launch_bounds__(256) void zero_pad_cols_kernel(float* __restrict__ m, int rows, int ld, int used) {
  const int padw = ld - used;
  const long total = (long)rows * padw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) m[(i / padw) * ld + used + (i % padw)] = 0.f;
}

}  // namespace diagan

DIAGAN_API int diagan_pack_oihw(const float* w, float scale, float* Wf, float* Wd, int Co_src, int Ci_src, int R, int S, int Co, int Ci,
                                int Kp, int Kd, void* stream) {
  DG_REQUIRE(w && (Wf || Wd), "pack_oihw: null pointer");
  DG_REQUIRE(Co_src > 0 && Ci_src > 0 && R > 0 && S > 0 && Co >= Co_src && Ci >= Ci_src && Kp >= R * S * Ci && (!Wd || Kd >= R * S * Co),
             "pack_oihw: bad dims");
  long blocks = ((long)Co * Kp + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_oihw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, w, scale, Wf, Wd, Co_src, Ci_src, R, S, Co, Ci,
                     Kp, Kd);
  if (Wd && Kd > R * S * Co) {
    long b2 = ((long)Ci * (Kd - R * S * Co) + 255) / 256;
    if (b2 > 1024) b2 = 1024;
    hipLaunchKernelGGL(pack_oihw_wd_pad_kernel, dim3((int)b2), dim3(256), 0, (hipStream_t)stream, Wd, Ci, Kd, R * S * Co);
  }
  return check_launch("pack_oihw");
}

DIAGAN_API int diagan_unpack_oihw(const float* gWp, float scale, float* gw, int Co_src, int Ci_src, int R, int S, int Ci, int Kp,
                                  void* stream) {
  DG_REQUIRE(gWp && gw, "unpack_oihw: null pointer");
  DG_REQUIRE(Co_src > 0 && Ci_src > 0 && R > 0 && S > 0 && Ci >= Ci_src && Kp >= R * S * Ci, "unpack_oihw: bad dims");
  long blocks = ((long)Co_src * Ci_src * R * S + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpack_oihw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, gWp, scale, gw, Co_src, Ci_src, R, S, Ci, Kp);
  return check_launch("unpack_oihw");
}

// kp[4] / off[4]: row length (a multiple of 32, >= ny nx C) and float offset of class (cy, cx) = (cls >> 1, cls & 1) in `buf`
// (classes without taps -- R or S == 1 -- are skipped).  adjoint = 0: buf <- w;  adjoint = 1: gw <- buf (every tap of gw belongs to
// exactly one class; its padding columns behind R S C are zeroed)
DIAGAN_API int diagan_parity_weights(const float* w, float* buf, float* gw, int n, int R, int S, int C, int Kp, const int* kp,
                                     const int* off, int adjoint, void* stream) {
  DG_REQUIRE(buf && kp && off && (adjoint ? gw != nullptr : w != nullptr), "parity_weights: null pointer");
  DG_REQUIRE(n > 0 && R > 0 && S > 0 && R <= 4 && S <= 4 && C > 0 && Kp >= R * S * C, "parity_weights: bad dims");
  ParityGeom g;
  g.n = n; g.R = R; g.S = S; g.C = C; g.Kp = Kp;
  long most = 0;
  for (int p = 0; p < 2; ++p) {
    g.ny[p] = p < R ? (R - p + 1) / 2 : 0;
    g.nx[p] = p < S ? (S - p + 1) / 2 : 0;
  }
  for (int cls = 0; cls < 4; ++cls) {
    g.off[cls] = off[cls];
    g.kp[cls] = kp[cls];
    const int taps = g.ny[cls >> 1] * g.nx[cls & 1];
    DG_REQUIRE(taps == 0 || kp[cls] >= taps * C, "parity_weights: class %d row of %d floats holds no %d taps x %d channels", cls, kp[cls], taps, C);
    if (taps && (long)n * kp[cls] > most) most = (long)n * kp[cls];
  }
  long blocks = (most + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (adjoint) {
    hipLaunchKernelGGL(parity_weights_kernel, dim3((int)blocks, 4), dim3(256), 0, (hipStream_t)stream, buf, gw, g, 1);
    if (Kp > R * S * C) {
      long b2 = ((long)n * (Kp - R * S * C) + 255) / 256;
      if (b2 > 1024) b2 = 1024;
      hipLaunchKernelGGL(zero_pad_cols_kernel, dim3((int)b2), dim3(256), 0, (hipStream_t)stream, gw, n, Kp, R * S * C);
    }
  } else {
    hipLaunchKernelGGL(parity_weights_kernel, dim3((int)blocks, 4), dim3(256), 0, (hipStream_t)stream, w, buf, g, 0);
  }
  return check_launch("parity_weights");
}

// ================================================================================================
// Batched spectral-norm preparation: ALL SN layers of a network in 4 launches (instead of 4 per
// layer).  blockIdx.z selects the layer through a device-resident descriptor table.
// ================================================================================================
namespace diagan {

struct SnLayer {          // mirrored by diagan_sn_layer in include/diagan_hip.h (96 bytes)
  const float* W;         // master weight [Co][Kp]
  float* u_buf;           // module buffer sn_u [Co]
  float* sigma_buf;       // module buffer sn_sigma [1]
  float* u_out;           // [Co]  u' of this forward (kept for the backward)
  float* v_out;           // [Kp]  v  of this forward
  float* state;           // {sigma, 1/sigma}
  float* work;            // [SN_RSPLIT*Kp + Co] scratch
  float* Wf;              // [Co][Kp] scaled forward operand or NULL
  float* Wd;              // [Ci][Kd] scaled data-gradient operand or NULL
  int Co, Ci, RS, Kp, Kd, pad;
};
constexpr int SN_RSPLIT = 8;
constexpr int SN_MAX_KP = 9216;

// vpart[r][k] = sum_{n in row chunk r} u[n] W[n][k].  A workgroup covers 64 columns x one row chunk with 4 row lanes
// (then an LDS reduction): 4x the workgroups of a one-thread-per-column mapping -- the large SNGAN-64 weights
// (1024 x 4608) had 144 workgroups for 19 MB.
__global__ __launch_bounds__(256) void sn_cols_batched_kernel(const SnLayer* __restrict__ tab) {
  __shared__ float red[4][64];
  const SnLayer L = tab[blockIdx.z];
  const int kl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kl;
  if (blockIdx.x * 64 >= L.Kp) return;
  const int per = (L.Co + SN_RSPLIT - 1) / SN_RSPLIT;
  const int n0 = blockIdx.y * per, n1 = min(n0 + per, L.Co);
  float s = 0.f;
  if (k < L.Kp) {
#pragma unroll 4
    for (int n = n0 + rl; n < n1; n += 4) s = fmaf(L.u_buf[n], L.W[(long)n * L.Kp + k], s);
  }
  red[rl][kl] = s;
  __syncthreads();
  if (rl == 0 && k < L.Kp) L.work[(long)blockIdx.y * L.Kp + k] = (red[0][kl] + red[1][kl]) + (red[2][kl] + red[3][kl]);
}

// v_raw = sum_r vpart[r] (into LDS; block 0 also stores it to v_out); t_raw[n] = W[n] . v_raw, 8 rows/block
__global__ __launch_bounds__(256) void sn_rows_batched_kernel(const SnLayer* __restrict__ tab) {
  __shared__ float vs[SN_MAX_KP];
  const SnLayer L = tab[blockIdx.z];
  const int row0 = blockIdx.x * 8;
  if (row0 >= L.Co) return;
  for (int k = threadIdx.x; k < L.Kp; k += 256) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < SN_RSPLIT; ++r) s += L.work[(long)r * L.Kp + k];
    vs[k] = s;
    if (blockIdx.x == 0) L.v_out[k] = s;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* t_raw = L.work + (long)SN_RSPLIT * L.Kp;
  for (int i = 0; i < 2; ++i) {
    const int n = row0 + wave * 2 + i;
    if (n >= L.Co) break;
    float s = 0.f;
    for (int k = lane * 4; k < L.Kp; k += 256) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(L.W + (long)n * L.Kp + k);
      s += w[0] * vs[k] + w[1] * vs[k + 1] + w[2] * vs[k + 2] + w[3] * vs[k + 3];
    }
    s = wave_sum(s);
    if (lane == 0) t_raw[n] = s;
  }
}

__global__ __launch_bounds__(256) void sn_finalize_batched_kernel(const SnLayer* __restrict__ tab, float eps,
                                                                  int update_buffers) {
  __shared__ float red[4];
  const SnLayer L = tab[blockIdx.z];
  const float* t_raw = L.work + (long)SN_RSPLIT * L.Kp;
  float s = 0.f;
  for (int k = threadIdx.x; k < L.Kp; k += 256) { const float v = L.v_out[k]; s += v * v; }
  const float nv = fmaxf(sqrtf(block_sum_256(s, red)), eps);
  for (int k = threadIdx.x; k < L.Kp; k += 256) L.v_out[k] = L.v_out[k] / nv;
  s = 0.f;
  for (int n = threadIdx.x; n < L.Co; n += 256) { const float t = t_raw[n] / nv; s += t * t; }
  const float nt = fmaxf(sqrtf(block_sum_256(s, red)), eps);
  s = 0.f;
  for (int n = threadIdx.x; n < L.Co; n += 256) {
    const float t = t_raw[n] / nv;
    const float un = t / nt;
    L.u_out[n] = un;
    if (update_buffers) L.u_buf[n] = un;
    s += un * t;
  }
  const float sigma = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    L.state[0] = sigma;
    L.state[1] = 1.f / sigma;
    if (update_buffers) L.sigma_buf[0] = sigma;
  }
}

// Work items = (layer, tap, 32 x 32 tile of [Co][Ci]) of the layers that have an operand to write.  A fixed grid of workgroups
// strides over the items (prefix of the per-layer counts in LDS, made from the table by every workgroup): a
// [max Ci / 32] x [max Co / 32] x [layers x taps] grid launched 147 000 workgroups for SNGAN-64's discriminator, most of
// them empty, and spent its 60 us dispatching them (0.19 of the HBM rate on the bytes it moved).
constexpr int PACK_MAX_LAYERS = 64;
constexpr int PACK_GRID = 2048;
__global__ __launch_bounds__(256) void pack_batched_kernel(const SnLayer* __restrict__ tab, int n_layers, int write_wd) {
  __shared__ float tile[32][33];
  __shared__ int first[PACK_MAX_LAYERS + 1];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int l = 0; l < n_layers; ++l) {
      first[l] = acc;
      const bool live = tab[l].Wf || (tab[l].Wd && write_wd);
      acc += live ? tab[l].RS * ((tab[l].Ci + 31) / 32) * ((tab[l].Co + 31) / 32) : 0;
    }
    first[n_layers] = acc;
  }
  __syncthreads();
  const int total = first[n_layers];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  int l = 0;
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    while (first[l + 1] <= item) ++l;                          // items only grow: the search resumes where it stopped
    const SnLayer L = tab[l];
    const int tc = (L.Ci + 31) / 32, tn = (L.Co + 31) / 32;
    int r = item - first[l];
    const int tap = r / (tc * tn);
    r -= tap * (tc * tn);
    const int co0 = (r / tc) * 32, ci0 = (r % tc) * 32;
    const float inv = L.state[1];
    for (int q = ty; q < 32; q += 8) {
      const int co = co0 + q, ci = ci0 + tx;
      float v = 0.f;
      if (co < L.Co && ci < L.Ci) {
        v = L.W[(long)co * L.Kp + tap * L.Ci + ci] * inv;
        if (L.Wf) L.Wf[(long)co * L.Kp + tap * L.Ci + ci] = v;
      }
      tile[q][tx] = v;
    }
    __syncthreads();
    if (L.Wd && write_wd) {
      for (int q = ty; q < 32; q += 8) {
        const int ci = ci0 + q, co = co0 + tx;
        if (ci < L.Ci && co < L.Co) L.Wd[(long)ci * L.Kd + tap * L.Co + co] = tile[tx][q];
      }
    }
    __syncthreads();
  }
}

}  // namespace diagan

DIAGAN_API int diagan_sn_prepare_batched(const void* table_dev, int n_layers, int max_Co, int max_Ci, int max_RS,
                                         int max_Kp, float eps, int update_buffers, int write_wd, void* stream) {
  DG_REQUIRE(table_dev && n_layers > 0 && max_Co > 0 && max_Ci > 0 && max_RS > 0 && max_Kp > 0,
             "sn_prepare_batched: bad args");
  DG_REQUIRE(max_Kp <= SN_MAX_KP, "sn_prepare_batched: Kp=%d exceeds the LDS-resident limit %d", max_Kp, SN_MAX_KP);
  DG_REQUIRE(n_layers <= PACK_MAX_LAYERS, "sn_prepare_batched: at most %d layers per table", PACK_MAX_LAYERS);
  static_assert(sizeof(SnLayer) == 96, "descriptor layout");
  hipStream_t st = (hipStream_t)stream;
  const SnLayer* tab = (const SnLayer*)table_dev;
  hipLaunchKernelGGL(sn_cols_batched_kernel, dim3(cdiv(max_Kp, 64), SN_RSPLIT, n_layers), dim3(256), 0, st, tab);
  hipLaunchKernelGGL(sn_rows_batched_kernel, dim3(cdiv(max_Co, 8), 1, n_layers), dim3(256), 0, st, tab);
  hipLaunchKernelGGL(sn_finalize_batched_kernel, dim3(1, 1, n_layers), dim3(256), 0, st, tab, eps, update_buffers);
  if (write_wd >= 0)      // write_wd < 0: power iteration only (operands are packed elsewhere)
    hipLaunchKernelGGL(pack_batched_kernel, dim3(PACK_GRID), dim3(256), 0, st, tab, n_layers, write_wd);
  return check_launch("sn_prepare_batched");
}

// Operand packing alone for a table of layers (scale = table state[1]; used with a unit-scale state to
// produce the un-normalised data-gradient operand shared by two batched forwards).
DIAGAN_API int diagan_pack_batched(const void* table_dev, int n_layers, int max_Co, int max_Ci, int max_RS,
                                   int write_wd, void* stream) {
  DG_REQUIRE(table_dev && n_layers > 0 && max_Co > 0 && max_Ci > 0 && max_RS > 0, "pack_batched: bad args");
  DG_REQUIRE(n_layers <= PACK_MAX_LAYERS, "pack_batched: at most %d layers per table", PACK_MAX_LAYERS);
  hipLaunchKernelGGL(pack_batched_kernel, dim3(PACK_GRID), dim3(256), 0, (hipStream_t)stream, (const SnLayer*)table_dev,
                     n_layers, write_wd);
  return check_launch("pack_batched");
}
