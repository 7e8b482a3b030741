// Precision / recall in feature space: the reductions over the pairwise-distance matrix of
// diagan-pkg/diagan/trainer/compute_pr.py:11-124.  The matrix itself comes from the conv GEMM (a 1x1 "conv":
// T[r][c] = |b_c|^2 - 2 a_r . b_c through out_scale = -2 and the bias operand); every consumer below adds the row
// term |a_r|^2 on the fly, so the reference's association (|a|^2 - 2ab) + |b|^2 is kept up to the symmetric
// swap and no second N x N pass is needed.
//
//   row_sqnorm         torch.sum(torch.square(x), dim=1)                      (compute_pr.py:26-27)
//   kth_smallest_rows  get_kth_value(distances, k): k-th smallest per row     (compute_pr.py:34-50)
//   any_lt_rows/cols   (distance < radii).any(axis)                           (compute_pr.py:85-93, 117-120)
//
// Roofline: HBM (each kernel reads the matrix once; N = 10 000: 400 MB).
#include "common.h"

namespace diagan {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PR_T = 256;
constexpr int PR_KMAX = 16;

__global__ __launch_bounds__(PR_T) void row_sqnorm_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                          int D, int ld) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < D; c += PR_T) {
    const float v = x[(long)r * ld + c];
    s = fmaf(v, v, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[r] = (red[0] + red[1]) + (red[2] + red[3]);
}

// k-th smallest (1-based, duplicates counted) of T[r][:] + row_add[r]
__global__ __launch_bounds__(PR_T) void kth_smallest_rows_kernel(const float* __restrict__ T, const float* __restrict__ row_add,
                                                                 int rows, int cols, int ld, int k, float* __restrict__ out) {
  __shared__ float heads[PR_T];
  __shared__ float wv[4];
  __shared__ int wi[4];
  __shared__ int winner;
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* row = T + (long)r * ld;
  // per-thread sorted list of its k smallest values
  float best[PR_KMAX];
#pragma unroll
  for (int i = 0; i < PR_KMAX; ++i) best[i] = __builtin_huge_valf();
  for (int c = tid; c < cols; c += PR_T) {
    float v = row[c];
    if (v < best[PR_KMAX - 1]) {
#pragma unroll
      for (int i = 0; i < PR_KMAX; ++i) {       // insertion: keep ascending order
        const float lo = fminf(best[i], v);
        v = fmaxf(best[i], v);
        best[i] = lo;
      }
    }
  }
  // k rounds of "pop the block-wide minimum of the list heads"
  float kth = __builtin_huge_valf();
  for (int t = 0; t < k; ++t) {
    float v = best[0];
    int who = tid;
    // wave argmin
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int ow = __shfl_xor(who, o, 64);
      if (ov < v || (ov == v && ow < who)) { v = ov; who = ow; }
    }
    if ((tid & 63) == 0) { wv[tid >> 6] = v; wi[tid >> 6] = who; }
    __syncthreads();
    if (tid == 0) {
      float bv = wv[0];
      int bw = wi[0];
      for (int w = 1; w < 4; ++w)
        if (wv[w] < bv || (wv[w] == bv && wi[w] < bw)) { bv = wv[w]; bw = wi[w]; }
      winner = bw;
      heads[0] = bv;
    }
    __syncthreads();
    kth = heads[0];
    if (tid == winner) {                        // pop this thread's head
#pragma unroll
      for (int i = 0; i + 1 < PR_KMAX; ++i) best[i] = best[i + 1];
      best[PR_KMAX - 1] = __builtin_huge_valf();
    }
    __syncthreads();
  }
  if (tid == 0) out[r] = kth + (row_add ? row_add[r] : 0.f);
}

// out[r] = any_c ( (T[r][c] + row_add[r]) < thr_col[c] )   (thr_row == null)
// out[r] = any_c ( (T[r][c] + row_add[r]) < thr_row[r] )   (thr_col == null)
__global__ __launch_bounds__(PR_T) void any_lt_rows_kernel(const float* __restrict__ T, const float* __restrict__ row_add,
                                                           const float* __restrict__ thr_col, const float* __restrict__ thr_row,
                                                           int rows, int cols, int ld, float* __restrict__ out) {
  __shared__ int flag;
  const int r = blockIdx.x;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  const float add = row_add ? row_add[r] : 0.f;
  const float tr = thr_row ? thr_row[r] : 0.f;
  int f = 0;
  for (int c = threadIdx.x; c < cols; c += PR_T) {
    const float d = T[(long)r * ld + c] + add;
    f |= d < (thr_col ? thr_col[c] : tr) ? 1 : 0;
  }
  if (f) flag = 1;      // benign race: every writer stores 1
  __syncthreads();
  if (threadIdx.x == 0) out[r] = flag ? 1.f : 0.f;
}

// out[c] = any_r ( (T[r][c] + row_add[r]) < thr_row[r] )   (thr_col == null)
// out[c] = any_r ( (T[r][c] + row_add[r]) < thr_col[c] )   (thr_row == null)
__global__ __launch_bounds__(PR_T) void any_lt_cols_kernel(const float* __restrict__ T, const float* __restrict__ row_add,
                                                           const float* __restrict__ thr_col, const float* __restrict__ thr_row,
                                                           int rows, int cols, int ld, float* __restrict__ out) {
  const int c = blockIdx.x * PR_T + threadIdx.x;
  if (c >= cols) return;
  const float tc = thr_col ? thr_col[c] : 0.f;
  int f = 0;
  for (int r = 0; r < rows; ++r) {
    const float d = T[(long)r * ld + c] + (row_add ? row_add[r] : 0.f);
    f |= d < (thr_row ? thr_row[r] : tc) ? 1 : 0;
  }
  out[c] = f ? 1.f : 0.f;
}

}  // namespace diagan

using namespace diagan;

DIAGAN_API int diagan_row_sqnorm(const float* x, float* out, int N, int D, int ld, void* stream) {
  DG_REQUIRE(x && out && N > 0 && D > 0 && ld >= D, "row_sqnorm: bad args");
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(N), dim3(PR_T), 0, (hipStream_t)stream, x, out, N, D, ld);
  return check_launch("row_sqnorm");
}

DIAGAN_API int diagan_kth_smallest_rows(const float* T, const float* row_add, int rows, int cols, int ld, int k, float* out,
                                        void* stream) {
  DG_REQUIRE(T && out && rows > 0 && cols > 0 && ld >= cols, "kth_smallest_rows: bad args");
  DG_REQUIRE(k >= 1 && k <= PR_KMAX && k <= cols, "kth_smallest_rows: k=%d must be in [1, min(%d, cols)]", k, PR_KMAX);
  hipLaunchKernelGGL(kth_smallest_rows_kernel, dim3(rows), dim3(PR_T), 0, (hipStream_t)stream, T, row_add, rows, cols, ld,
                     k, out);
  return check_launch("kth_smallest_rows");
}

DIAGAN_API int diagan_any_lt_rows(const float* T, const float* row_add, const float* thr_col, const float* thr_row, int rows,
                                  int cols, int ld, float* out, void* stream) {
  DG_REQUIRE(T && out && rows > 0 && cols > 0 && ld >= cols, "any_lt_rows: bad args");
  DG_REQUIRE(!thr_col != !thr_row, "any_lt_rows: exactly one of thr_col / thr_row");
  hipLaunchKernelGGL(any_lt_rows_kernel, dim3(rows), dim3(PR_T), 0, (hipStream_t)stream, T, row_add, thr_col, thr_row, rows,
                     cols, ld, out);
  return check_launch("any_lt_rows");
}

DIAGAN_API int diagan_any_lt_cols(const float* T, const float* row_add, const float* thr_col, const float* thr_row, int rows,
                                  int cols, int ld, float* out, void* stream) {
  DG_REQUIRE(T && out && rows > 0 && cols > 0 && ld >= cols, "any_lt_cols: bad args");
  DG_REQUIRE(!thr_col != !thr_row, "any_lt_cols: exactly one of thr_col / thr_row");
  hipLaunchKernelGGL(any_lt_cols_kernel, dim3(cdiv(cols, PR_T)), dim3(PR_T), 0, (hipStream_t)stream, T, row_add, thr_col,
                     thr_row, rows, cols, ld, out);
  return check_launch("any_lt_cols");
}
