// Micro-benchmark (GPU box): cycles per v_mfma_f32_32x32x2_f32 as a function of waves per SIMD and of what else the
// waves do -- how many independent accumulators a lone wave needs, and what interleaved LDS reads / VALU cost.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[8192];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  f32x4 v = {a, b, a, b};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (MODE == 1) {                       // 3 VALU per MFMA
          v[0] = v[0] * 1.0001f + v[1];
          v[1] = v[1] * 0.9999f + v[2];
          v[2] = v[2] * 1.0001f + v[3];
        }
        if (MODE == 2 && (i & 1) == 0) {       // one ds_read_b128 per 2 MFMAs
          const f32x4 q = *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + (it + r * NACC + i) * 64) & 8188));
          a += q[0];
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v[0] + v[1] + v[2];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
void run(int threads, const char* what) {
  float* out;
  unsigned long long* cyc;
  const int wgs = 256, iters = 200;
  hipMalloc(&out, wgs * 512 * 4);
  hipMalloc(&cyc, wgs * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NACC, MODE>), dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < wgs; ++i) m += h[i];
  m /= wgs;
  printf("%-34s waves/SIMD %d  accumulators %d: %7.1f cycles per MFMA per wave, %6.1f per SIMD-MFMA\n", what, threads / 256, NACC,
         m / (iters * 32.0), m / (iters * 32.0) / (threads / 256));
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<8, 0>(256, "MFMA only");
  run<4, 0>(256, "MFMA only");
  run<2, 0>(256, "MFMA only");
  run<1, 0>(256, "MFMA only");
  run<8, 0>(512, "MFMA only");
  run<2, 0>(512, "MFMA only");
  run<1, 0>(512, "MFMA only");
  run<8, 1>(256, "3 VALU per MFMA");
  run<8, 1>(512, "3 VALU per MFMA");
  run<8, 2>(256, "ds_read_b128 per 2 MFMA");
  run<8, 2>(512, "ds_read_b128 per 2 MFMA");
  return 0;
}
