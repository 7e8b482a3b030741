/* libdiagan_hip.so -- C ABI of the MI355X (gfx950) Dia-GAN hot path.
 *
 * The reference (grayhong/self-diagnosing-gan) has NO FFI on this path: its SNGAN/DCGAN train
 * step and LDR scorer are Python calling ATen/cuDNN ops and NumPy.  The only native boundary in
 * the reference is the StyleGAN2 pybind pair (diagan-pkg/diagan/models/op/fused_bias_act.cpp:4-20,
 * upfirdn2d.cpp:4-22), whose conventions this ABI keeps: launch on the caller's stream, no hidden
 * synchronisation, no global mutable state, errors surfaced to Python as RuntimeError.
 * Every entry point below names the reference Python op (file:line) it replaces.
 *
 * Conventions
 *   - plain C types only; all pointers are DEVICE pointers owned by the caller (torch allocates);
 *   - every function returns 0 or a negative DIAGAN_E* code; diagan_last_error() gives the text
 *     (thread local);  the python shim (diagan/_native/__init__.py) raises RuntimeError on != 0;
 *   - `stream` is a hipStream_t (0 = default stream); nothing synchronises;
 *   - activations are NHWC fp32 ("pixels x channels" row-major), channel counts padded to a
 *     multiple of 4; weights are consumed in the packed GEMM layouts produced by
 *     diagan_weight_prep (see DESIGN.md "Data layout in HBM").
 */
#ifndef DIAGAN_HIP_H
#define DIAGAN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DIAGAN_OK 0
#define DIAGAN_EINVAL (-1)
#define DIAGAN_EHIP (-2)
#define DIAGAN_EUNSUP (-3)

const char* diagan_last_error(void);
int diagan_abi_version(void);
const char* diagan_target_arch(void); /* "gfx950" */

/* ---- LDR scorer and logit record --------------------------------------------------------- */

/* calculate_scores, diagan-pkg/diagan/utils/plot.py:220-249 (float64, bit-exact with NumPy).
 * rec[T][row_stride] holds the window's snapshots in step order (plot.py:239).
 * ldr/ldrd/ldrv/ldrm: [N] outputs or NULL (plot.py:243-246).
 * conf[n_t][N]: clip_max_ratio(clip_min(mean + t*std, floor), ratio) for each t in t_vals
 * (plot.py:247-248; floor 1e-2, ratio 50).  workspace: >= 8*n_t bytes. */
int diagan_ldr_scores_f64(const double* rec, int T, int64_t N, int64_t row_stride, double* ldr,
                          double* ldrd, double* ldrv, double* ldrm, const double* t_vals, int n_t,
                          double* conf, double floor_val, double ratio, void* workspace,
                          void* stream);

/* Same scores in fp32 with wave-shuffle reductions over T (not bit-exact; |err| ~ 1e-6).
 * workspace: >= 4*n_t bytes. */
int diagan_ldr_scores_f32(const float* rec, int T, int64_t N, int64_t row_stride, float* ldr,
                          float* ldrd, float* ldrv, float* ldrm, const float* t_vals, int n_t,
                          float* conf, float floor_val, float ratio, void* workspace, void* stream);

/* logit_list[idx] = logit, diagan-pkg/diagan/trainer/trainer.py:154 (and the DP variant
 * stylegan2/train_ffhq.py:139-141).  rec_row: one [N] row of the record (f64 or f32).
 * Out-of-range indices are counted in *oob_counter (device int) and skipped. */
int diagan_logit_scatter(const float* logit, const int64_t* idx, int64_t n, void* rec_row,
                         int64_t N, int out_is_f64, int* oob_counter, void* stream);

#ifdef __cplusplus
}
#endif
#endif
