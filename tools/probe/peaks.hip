// Achievable peaks on this box, to quote beside the vendor nominals:
//   fp32 matrix pipe: every wave issues independent v_mfma_f32_32x32x2_f32 back to back from registers
//   HBM: float4 grid-stride copy of 1 GiB (read + write counted)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/peaks tools/probe/peaks.hip && /tmp/peaks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;      // keep the loop alive
}

__global__ __launch_bounds__(256) void copy4(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) out[i] = in[i];
}

int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float* d; hipMalloc(&d, 1024);
  float ms;
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    mfma_loop<<<blocks, 256>>>(d, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<<<blocks, 256>>>(d, iters, 1.0009f, 0.9991f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16 * (32.0 * 32 * 2 * 2);
    printf("fp32 MFMA 32x32x2, %d waves/SIMD: %.1f TFLOP/s (%.2f ms)\n", blocks_per_cu, flop / ms / 1e9, ms);
  }
  const long bytes = 1L << 30;
  f32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    copy4<<<256 * 16, 256>>>(a, b, bytes / 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("HBM float4 copy of 1 GiB: %.2f TB/s (read + write), %.3f ms\n", 2.0 * bytes / ms / 1e9, ms);
  return 0;
}
