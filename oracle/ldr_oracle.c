/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Plain-C restatement of the reference's LDR scorer and logit record, used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg to check / time against.
 * Pinned against the reference itself: tests/golden/scorer_*.npz were produced by importing
 * /root/reference/diagan-pkg/diagan/utils/plot.py::calculate_scores (tools/gen_goldens.py) and
 * this file reproduces them bit-for-bit (tests/test_oracle_scorer.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC   (no FMA contraction: NumPy rounds a*b then +c)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* plot.py:239-248.  rec[T][stride]: snapshots of the window in step order.
 * NumPy reduces axis 0 of a C-contiguous [T,N] array row by row, i.e. sequentially over T for
 * every column, which is what the loops below do (N >= 2; see oracle/scorer.py for N == 1). */
int oracle_ldr_scores(const double* rec, int T, int64_t N, int64_t stride, double* ldr,
                      double* ldrd, double* ldrv, double* ldrm, const double* t_vals, int n_t,
                      double* conf, double floor_val, double ratio) {
  if (T < 2 || N < 1) return -1;
  double* mean = (double*)malloc(sizeof(double) * (size_t)N);
  double* sd = (double*)malloc(sizeof(double) * (size_t)N);
  if (!mean || !sd) return -2;
  for (int64_t i = 0; i < N; ++i) {
    double s = 0.0, dsum = 0.0;
    for (int t = 0; t < T; ++t) s += rec[(int64_t)t * stride + i];           /* logits_arr.mean(0) */
    for (int t = 1; t < T; ++t)                                              /* plot.py:244 */
      dsum += fabs(rec[(int64_t)t * stride + i] - rec[(int64_t)(t - 1) * stride + i]);
    const double m = s / (double)T;
    double d = 0.0;
    for (int t = 0; t < T; ++t) {                                            /* np.var(ddof=1), plot.py:245 */
      const double x = rec[(int64_t)t * stride + i] - m;
      d += x * x;
    }
    const double var = d / (double)(T - 1);
    mean[i] = m;
    sd[i] = sqrt(var);                                                       /* np.std(ddof=1) */
    if (ldr) ldr[i] = rec[(int64_t)(T - 1) * stride + i];                    /* plot.py:243 */
    if (ldrd) ldrd[i] = dsum / (double)(T - 1);
    if (ldrv) ldrv[i] = var;
    if (ldrm) ldrm[i] = m;                                                   /* plot.py:246 */
  }
  for (int k = 0; k < n_t; ++k) {                                            /* plot.py:247-248 */
    double* out = conf + (int64_t)k * N;
    double mn = INFINITY;
    for (int64_t i = 0; i < N; ++i) {
      double v = mean[i] + t_vals[k] * sd[i];
      v = (v < floor_val) ? floor_val : v;                                   /* clip_min, plot.py:230-231 */
      out[i] = v;
      if (v < mn) mn = v;
    }
    const double upper = mn * ratio;                                         /* clip_max_ratio, plot.py:226-228 */
    for (int64_t i = 0; i < N; ++i)
      if (out[i] > upper) out[i] = upper;
  }
  free(mean);
  free(sd);
  return 0;
}

/* trainer.py:144,154: logit_list = np.zeros(N); logit_list[idx] = logit (float32 -> float64) */
int oracle_logit_scatter(const float* logit, const int64_t* idx, int64_t n, double* row, int64_t N) {
  for (int64_t j = 0; j < n; ++j) {
    if (idx[j] < 0 || idx[j] >= N) return -1;
    row[idx[j]] = (double)logit[j];
  }
  return 0;
}

/* train_mimicry_phase2.py:23: weight_list = [eps if i < eps else i for i in weights] */
void oracle_weight_floor(const double* w, int64_t N, double eps, double* out) {
  for (int64_t i = 0; i < N; ++i) out[i] = (w[i] < eps) ? eps : w[i];
}
