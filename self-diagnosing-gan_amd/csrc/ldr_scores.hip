// LDR discrepancy scorer + per-index logit record.
//
// Replaces, on device:
//   * diagan-pkg/diagan/utils/plot.py:220-249  calculate_scores  (NumPy, float64)
//   * diagan-pkg/diagan/trainer/trainer.py:142-156  _get_logit's  logit_list[idx] = logit  scatter
//
// Data layout in HBM: the logit record is a resident float64 matrix rec[T_cap][row_stride]
// (one row per snapshot, one column per dataset index) -- the same [T, N] array the reference
// builds with np.array([...]) at plot.py:239, except that it never leaves the GPU.
//
// f64 path: one thread per sample walks the T snapshots IN ORDER, exactly like NumPy's axis-0
// reductions (row-by-row accumulation), with separate multiply and add roundings
// (-ffp-contract=off + explicit __d*_rn) so the scores -- and therefore the phase-2
// WeightedRandomSampler draws -- are bit-identical to the reference.
// f32 path: 4 lanes per sample split T, combined with wave shuffles (fast, ~1e-6 accurate).
//
// Roofline: HBM. Algorithmic bytes per sample: 8*T read + 8 per requested score written.
#include "common.h"

namespace {

constexpr int SC_THREADS = 256;

__device__ __forceinline__ void lds_min_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Kernel 1 (f64, exact): per-sample statistics + pre-clip confidence scores + global minima.
__global__ __launch_bounds__(SC_THREADS) void ldr_stats_f64_kernel(
    const double* __restrict__ rec, int T, long N, long stride, double* __restrict__ ldr,
    double* __restrict__ ldrd, double* __restrict__ ldrv, double* __restrict__ ldrm,
    const double* __restrict__ t_vals, int n_t, double* __restrict__ conf, double floor_val,
    unsigned long long* __restrict__ gmin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* smin = reinterpret_cast<unsigned long long*>(smem_raw);
  for (int k = threadIdx.x; k < n_t; k += SC_THREADS) smin[k] = ~0ull;
  __syncthreads();

  const long i = (long)blockIdx.x * SC_THREADS + threadIdx.x;
  const bool live = i < N;
  double mean = 0.0, sd = 0.0;
  if (live) {
    const double* col = rec + i;
    // pass 1: sum over T (sequential, NumPy order), |first difference| sum, last row
    double s = 0.0, dsum = 0.0, prev = 0.0, last = 0.0;
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
      const double x = col[(long)t * stride];
      s = __dadd_rn(s, x);
      if (t > 0) dsum = __dadd_rn(dsum, fabs(__dsub_rn(x, prev)));
      prev = x;
      last = x;
    }
    mean = s / (double)T;  // np.mean: add.reduce then true_divide by the count
    // pass 2: np.var(ddof=1): x = arr - mean; x = x*x; sum; / (T-1)
    double d = 0.0;
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
      const double x = __dsub_rn(col[(long)t * stride], mean);
      d = __dadd_rn(d, __dmul_rn(x, x));
    }
    const double var = d / (double)(T - 1);
    sd = __dsqrt_rn(var);
    if (ldr) ldr[i] = last;                        // plot.py:243
    if (ldrd) ldrd[i] = dsum / (double)(T - 1);    // plot.py:244
    if (ldrv) ldrv[i] = var;                       // plot.py:245
    if (ldrm) ldrm[i] = mean;                      // plot.py:246
  }
  // plot.py:247-248: clip_min(mean + t*std, 1e-2), then the cross-sample min for clip_max_ratio
  for (int k = 0; k < n_t; ++k) {
    double pre = __builtin_huge_val();
    if (live) {
      pre = __dadd_rn(mean, __dmul_rn(t_vals[k], sd));
      pre = (pre < floor_val) ? floor_val : pre;
      conf[(long)k * N + i] = pre;
    }
    const double wm = diagan::wave_min(pre);
    if ((threadIdx.x & 63) == 0) lds_min_u64(&smin[k], (unsigned long long)__double_as_longlong(wm));
  }
  __syncthreads();
  // positive doubles order like their bit patterns: one 64-bit atomic min per block per t
  for (int k = threadIdx.x; k < n_t; k += SC_THREADS)
    __hip_atomic_fetch_min(&gmin[k], smin[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Kernel 2: clip_max_ratio (plot.py:226-228): minimum(score, score.min() * ratio)
__global__ __launch_bounds__(SC_THREADS) void ldr_clip_ratio_f64_kernel(
    double* __restrict__ conf, long N, int n_t, const unsigned long long* __restrict__ gmin,
    double ratio) {
  const long i = (long)blockIdx.x * SC_THREADS + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= N) return;
  const double upper = __dmul_rn(__longlong_as_double((long long)gmin[k]), ratio);
  const double v = conf[(long)k * N + i];
  conf[(long)k * N + i] = (v > upper) ? upper : v;
}

__global__ void fill_u64_kernel(unsigned long long* p, int n, unsigned long long v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- f32 fast path: wave = 16 samples x 4 T-phases, shuffle-combined -------------------------
__global__ __launch_bounds__(SC_THREADS) void ldr_stats_f32_kernel(
    const float* __restrict__ rec, int T, long N, long stride, float* __restrict__ ldr,
    float* __restrict__ ldrd, float* __restrict__ ldrv, float* __restrict__ ldrm,
    const float* __restrict__ t_vals, int n_t, float* __restrict__ conf, float floor_val,
    unsigned int* __restrict__ gmin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned int* smin = reinterpret_cast<unsigned int*>(smem_raw);
  for (int k = threadIdx.x; k < n_t; k += SC_THREADS) smin[k] = 0x7f800000u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ph = lane >> 4;  // T-phase 0..3
  const long i = ((long)blockIdx.x * (SC_THREADS / 64) + wave) * 16 + (lane & 15);
  const bool live = i < N;
  const long ic = live ? i : 0;
  // phase p owns the contiguous snapshot range [t0, t1)
  const int per = (T + 3) >> 2;
  const int t0 = min(ph * per, T), t1 = min(t0 + per, T);
  float s = 0.f, dsum = 0.f, prev = 0.f, first = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float x = rec[(long)t * stride + ic];
    s += x;
    if (t > t0) dsum += fabsf(x - prev); else first = x;
    prev = x;
  }
  // stitch the |diff| across phase boundaries: phase p needs the last value of phase p-1
  const float prev_last = __shfl_up(prev, 16, 64);
  const int prev_cnt = __shfl_up(t1 - t0, 16, 64);
  if (ph > 0 && t1 > t0 && prev_cnt > 0) dsum += fabsf(first - prev_last);
  float tot = s;
  tot += __shfl_xor(tot, 16, 64);
  tot += __shfl_xor(tot, 32, 64);
  const float mean = tot / (float)T;
  float d = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float x = rec[(long)t * stride + ic] - mean;
    d = fmaf(x, x, d);
  }
  d += __shfl_xor(d, 16, 64);
  d += __shfl_xor(d, 32, 64);
  dsum += __shfl_xor(dsum, 16, 64);
  dsum += __shfl_xor(dsum, 32, 64);
  // last snapshot lives in the highest non-empty phase
  const int last_ph = (T - 1) / per;
  const float last = __shfl(prev, (lane & 15) + 16 * last_ph, 64);
  const float var = d / (float)(T - 1);
  const float sd = sqrtf(var);
  if (live && ph == 0) {
    if (ldr) ldr[i] = last;
    if (ldrd) ldrd[i] = dsum / (float)(T - 1);
    if (ldrv) ldrv[i] = var;
    if (ldrm) ldrm[i] = mean;
  }
  for (int k = ph; k < n_t; k += 4) {  // the 4 phase-lanes share the t list
    float pre = mean + t_vals[k] * sd;
    pre = (pre < floor_val) ? floor_val : pre;
    if (live) {
      conf[(long)k * N + i] = pre;
      atomicMin(&smin[k], __float_as_uint(pre));
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_t; k += SC_THREADS) atomicMin(&gmin[k], smin[k]);
}

__global__ __launch_bounds__(SC_THREADS) void ldr_clip_ratio_f32_kernel(
    float* __restrict__ conf, long N, int n_t, const unsigned int* __restrict__ gmin, float ratio) {
  const long i = (long)blockIdx.x * SC_THREADS + threadIdx.x;
  const int k = blockIdx.y;
  if (i >= N) return;
  const float upper = __uint_as_float(gmin[k]) * ratio;
  const float v = conf[(long)k * N + i];
  conf[(long)k * N + i] = (v > upper) ? upper : v;
}

__global__ void fill_u32_kernel(unsigned int* p, int n, unsigned int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---- logit record scatter: rec_row[idx[j]] = logit[j] (trainer.py:154) ------------------------
template <typename OUT>
__global__ void logit_scatter_kernel(const float* __restrict__ logit, const long* __restrict__ idx,
                                     long n, OUT* __restrict__ row, long N, int* __restrict__ oob) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long k = idx[j];
  if (k < 0 || k >= N) { atomicAdd(oob, 1); return; }
  row[k] = (OUT)logit[j];
}

}  // namespace

// see include/diagan_hip.h for the contract of each entry point
DIAGAN_API int diagan_ldr_scores_f64(const double* rec, int T, int64_t N, int64_t row_stride,
                                     double* ldr, double* ldrd, double* ldrv, double* ldrm,
                                     const double* t_vals, int n_t, double* conf, double floor_val,
                                     double ratio, void* workspace, void* stream) {
  DG_REQUIRE(rec != nullptr, "ldr_scores_f64: rec is null");
  DG_REQUIRE(T >= 2, "ldr_scores_f64: need at least 2 snapshots in the window, got T=%d", T);
  DG_REQUIRE(N >= 1 && row_stride >= N, "ldr_scores_f64: bad N=%ld stride=%ld", (long)N, (long)row_stride);
  DG_REQUIRE(n_t >= 0 && n_t <= 4096, "ldr_scores_f64: n_t=%d out of range", n_t);
  DG_REQUIRE(n_t == 0 || (t_vals && conf && workspace), "ldr_scores_f64: conf outputs need t_vals/conf/workspace");
  DG_REQUIRE(floor_val > 0.0, "ldr_scores_f64: floor must be > 0 (min via integer compare)");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* gmin = (unsigned long long*)workspace;
  const int blocks = diagan::cdiv(N, SC_THREADS);
  if (n_t > 0) {
    hipLaunchKernelGGL(fill_u64_kernel, dim3(diagan::cdiv(n_t, 256)), dim3(256), 0, st, gmin, n_t, ~0ull);
  }
  hipLaunchKernelGGL(ldr_stats_f64_kernel, dim3(blocks), dim3(SC_THREADS), (size_t)(n_t > 0 ? n_t : 1) * 8, st,
                     rec, T, (long)N, (long)row_stride, ldr, ldrd, ldrv, ldrm, t_vals, n_t, conf,
                     floor_val, gmin);
  if (n_t > 0) {
    hipLaunchKernelGGL(ldr_clip_ratio_f64_kernel, dim3(blocks, n_t), dim3(SC_THREADS), 0, st, conf,
                       (long)N, n_t, gmin, ratio);
  }
  return diagan::check_launch("ldr_scores_f64");
}

DIAGAN_API int diagan_ldr_scores_f32(const float* rec, int T, int64_t N, int64_t row_stride,
                                     float* ldr, float* ldrd, float* ldrv, float* ldrm,
                                     const float* t_vals, int n_t, float* conf, float floor_val,
                                     float ratio, void* workspace, void* stream) {
  DG_REQUIRE(rec != nullptr, "ldr_scores_f32: rec is null");
  DG_REQUIRE(T >= 2, "ldr_scores_f32: need at least 2 snapshots in the window, got T=%d", T);
  DG_REQUIRE(N >= 1 && row_stride >= N, "ldr_scores_f32: bad N=%ld stride=%ld", (long)N, (long)row_stride);
  DG_REQUIRE(n_t >= 0 && n_t <= 4096, "ldr_scores_f32: n_t=%d out of range", n_t);
  DG_REQUIRE(n_t == 0 || (t_vals && conf && workspace), "ldr_scores_f32: conf outputs need t_vals/conf/workspace");
  DG_REQUIRE(floor_val > 0.f, "ldr_scores_f32: floor must be > 0");
  hipStream_t st = (hipStream_t)stream;
  unsigned int* gmin = (unsigned int*)workspace;
  if (n_t > 0)
    hipLaunchKernelGGL(fill_u32_kernel, dim3(diagan::cdiv(n_t, 256)), dim3(256), 0, st, gmin, n_t, 0x7f800000u);
  const int per_block = (SC_THREADS / 64) * 16;
  hipLaunchKernelGGL(ldr_stats_f32_kernel, dim3(diagan::cdiv(N, per_block)), dim3(SC_THREADS),
                     (size_t)(n_t > 0 ? n_t : 1) * 4, st, rec, T, (long)N, (long)row_stride, ldr, ldrd,
                     ldrv, ldrm, t_vals, n_t, conf, floor_val, gmin);
  if (n_t > 0)
    hipLaunchKernelGGL(ldr_clip_ratio_f32_kernel, dim3(diagan::cdiv(N, SC_THREADS), n_t), dim3(SC_THREADS),
                       0, st, conf, (long)N, n_t, gmin, ratio);
  return diagan::check_launch("ldr_scores_f32");
}

DIAGAN_API int diagan_logit_scatter(const float* logit, const int64_t* idx, int64_t n, void* rec_row,
                                    int64_t N, int out_is_f64, int* oob_counter, void* stream) {
  DG_REQUIRE(logit && idx && rec_row && oob_counter, "logit_scatter: null pointer");
  DG_REQUIRE(n >= 0 && N >= 1, "logit_scatter: bad n=%ld N=%ld", (long)n, (long)N);
  if (n == 0) return DIAGAN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = diagan::cdiv(n, 256);
  if (out_is_f64)
    hipLaunchKernelGGL(logit_scatter_kernel<double>, dim3(blocks), dim3(256), 0, st, logit,
                       (const long*)idx, (long)n, (double*)rec_row, (long)N, oob_counter);
  else
    hipLaunchKernelGGL(logit_scatter_kernel<float>, dim3(blocks), dim3(256), 0, st, logit,
                       (const long*)idx, (long)n, (float*)rec_row, (long)N, oob_counter);
  return diagan::check_launch("logit_scatter");
}
