// Fused GAN loss heads and the Adam update.
//
// Loss heads replace torch_mimicry.modules.losses (minimax_loss_dis / ns_loss_gen / hinge_loss_dis /
// hinge_loss_gen / wasserstein_*), the in-tree GOLD re-weighted variants
// (diagan-pkg/diagan/models/gold_reweight_models.py:10-61), TopKGenerator.get_topk
// (diagan-pkg/diagan/models/topk_models.py:31-38) and BaseDiscriminator.compute_probs, plus their
// autograd backward: one launch emits the loss scalar, dL/dlogit for every sample and the logged
// sigmoid means, so the train step needs no host synchronisation.
//
// Adam replaces torch.optim.Adam.step (predefined_models.py:32,51,70,89,114,123) on one flat
// parameter buffer per network (parameters, gradients and both moments are contiguous slabs).
// Roofline: HBM (Adam: 4 reads + 3 writes per parameter); the loss heads are latency bound (B = 64).
#include "common.h"

namespace diagan {

enum LossType : int { LOSS_GAN = 0, LOSS_NS = 1, LOSS_HINGE = 2, LOSS_WASSERSTEIN = 3 };

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = errD, out[1] = D(x) = mean sigmoid(real), out[2] = D(G(z)) = mean sigmoid(fake)
__global__ __launch_bounds__(256) void d_loss_kernel(const float* __restrict__ real, int n_real,
                                                     const float* __restrict__ fake, int n_fake, int loss_type,
                                                     int gold, float* __restrict__ d_real, float* __restrict__ d_fake,
                                                     float* __restrict__ out) {
  __shared__ double red[4];
  double lr = 0, lf = 0, pr = 0, pf = 0;
  for (int i = threadIdx.x; i < n_real; i += 256) {
    const float x = real[i], s = sigmoid_f(x);
    float l, g;
    if (loss_type == LOSS_HINGE) { l = fmaxf(1.f - x, 0.f); g = x < 1.f ? -1.f : 0.f; }
    else if (loss_type == LOSS_WASSERSTEIN) { l = -x; g = -1.f; }
    else { l = softplus_f(-x); g = s - 1.f; }            // BCE with logits against label 1
    lr += l; pr += s;
    if (d_real) d_real[i] = g / (float)n_real;
  }
  for (int i = threadIdx.x; i < n_fake; i += 256) {
    const float x = fake[i], s = sigmoid_f(x);
    float l, g;
    if (loss_type == LOSS_HINGE) { l = fmaxf(1.f + x, 0.f); g = x > -1.f ? 1.f : 0.f; }
    else if (loss_type == LOSS_WASSERSTEIN) { l = x; g = 1.f; }
    else { l = softplus_f(x); g = s; }                    // BCE with logits against label 0
    if (gold) { l *= x; g *= x; }                         // GOLD weight = output_fake**1, detached
    lf += l; pf += s;
    if (d_fake) d_fake[i] = g / (float)n_fake;
  }
  lr = block_sum_d(lr, red); lf = block_sum_d(lf, red);
  pr = block_sum_d(pr, red); pf = block_sum_d(pf, red);
  if (threadIdx.x == 0) {
    out[0] = (float)(lr / n_real) + (float)(lf / n_fake);
    out[1] = (float)(pr / n_real);
    out[2] = (float)(pf / n_fake);
  }
}

// generator loss over the top-k logits (k == n: all).  out[0] = errG
__global__ __launch_bounds__(256) void g_loss_kernel(const float* __restrict__ fake, int n, int k, int loss_type,
                                                     float* __restrict__ d_fake, float* __restrict__ out) {
  __shared__ double red[4];
  double ls = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = fake[i];
    bool sel = true;
    if (k < n) {  // rank by value (descending), ties by index: torch.topk keeps the k largest
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const float y = fake[j];
        rank += (y > x) || (y == x && j < i);
      }
      sel = rank < k;
    }
    float l = 0.f, g = 0.f;
    if (sel) {
      if (loss_type == LOSS_NS) {          // ns_loss_gen: -mean(log(sigmoid(x) + 1e-8))
        const float s = sigmoid_f(x);
        l = -logf(s + 1e-8f);
        g = -s * (1.f - s) / (s + 1e-8f);
      } else if (loss_type == LOSS_GAN) {  // minimax_loss_gen: BCE with logits against label 1
        l = softplus_f(-x);
        g = sigmoid_f(x) - 1.f;
      }
      else { l = -x; g = -1.f; }                          // hinge and wasserstein generator losses
    }
    ls += l;
    if (d_fake) d_fake[i] = g / (float)k;
  }
  ls = block_sum_d(ls, red);
  if (threadIdx.x == 0) out[0] = (float)(ls / k);
}

struct AdamHyper { float lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, pad; };   // 8 floats (device row of the *_dev form)

// torch.optim.Adam (no weight decay, no amsgrad):  m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n4,
                                                   const AdamHyper h) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const float step_size = h.lr / h.bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * h.grad_scale;   // 1/W of a data-parallel SUM all-reduce (1 otherwise)
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pv = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mv[e] = mv[e] + (gv[e] - mv[e]) * (1.f - h.beta1);
      vv[e] = vv[e] * h.beta2 + (1.f - h.beta2) * gv[e] * gv[e];
      const float denom = sqrtf(vv[e]) / h.bc2_sqrt + h.eps;
      pv[e] = pv[e] - step_size * (mv[e] / denom);
    }
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}

// the same update with the hyper-parameters read from device memory: a captured hipGraph replays this launch with
// whatever {lr, beta1, beta2, eps, bias corrections} the host wrote into the row before the replay
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long n4,
                                                       const AdamHyper* __restrict__ hp) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const AdamHyper h = *hp;
  const float step_size = h.lr / h.bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * h.grad_scale;   // 1/W of a data-parallel SUM all-reduce (1 otherwise)
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pv = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mv[e] = mv[e] + (gv[e] - mv[e]) * (1.f - h.beta1);
      vv[e] = vv[e] * h.beta2 + (1.f - h.beta2) * gv[e] * gv[e];
      const float denom = sqrtf(vv[e]) / h.bc2_sqrt + h.eps;
      pv[e] = pv[e] - step_size * (mv[e] / denom);
    }
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}

}  // namespace diagan

using namespace diagan;

DIAGAN_API int diagan_loss_dis(const float* out_real, int n_real, const float* out_fake, int n_fake, int loss_type,
                               int gold, float* d_real, float* d_fake, float* out3, void* stream) {
  DG_REQUIRE(out_real && out_fake && out3 && n_real > 0 && n_fake > 0, "loss_dis: bad args");
  DG_REQUIRE(loss_type >= 0 && loss_type <= 3, "loss_dis: unknown loss type %d", loss_type);
  DG_REQUIRE(!gold || loss_type == LOSS_NS || loss_type == LOSS_HINGE || loss_type == LOSS_GAN,
             "loss_dis: GOLD re-weighting exists for 'ns' and 'hinge' only (gold_reweight_models.py:69)");
  hipLaunchKernelGGL(d_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out_real, n_real, out_fake, n_fake,
                     loss_type, gold, d_real, d_fake, out3);
  return check_launch("loss_dis");
}

DIAGAN_API int diagan_loss_gen(const float* out_fake, int n, int k, int loss_type, float* d_fake, float* out1,
                               void* stream) {
  DG_REQUIRE(out_fake && out1 && n > 0 && k > 0 && k <= n, "loss_gen: bad args n=%d k=%d", n, k);
  DG_REQUIRE(loss_type >= 0 && loss_type <= 3, "loss_gen: unknown loss type %d", loss_type);
  hipLaunchKernelGGL(g_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out_fake, n, k, loss_type, d_fake,
                     out1);
  return check_launch("loss_gen");
}

DIAGAN_API int diagan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                                float grad_scale, void* stream) {
  DG_REQUIRE(p && g && m && v && n > 0 && (n & 3) == 0, "adam_step: bad args (n must be a multiple of 4)");
  AdamHyper h{lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt, grad_scale, 0.f};
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, h);
  return check_launch("adam_step");
}

DIAGAN_API int diagan_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper8,
                                    void* stream) {
  DG_REQUIRE(p && g && m && v && hyper8 && n > 0 && (n & 3) == 0, "adam_step_dev: bad args (n must be a multiple of 4)");
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4,
                     reinterpret_cast<const AdamHyper*>(hyper8));
  return check_launch("adam_step_dev");
}
