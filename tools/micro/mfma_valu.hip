// Micro-benchmark (GPU box): how much vector-ALU work hides behind v_mfma_f32_32x32x2_f32 (64 cycles each)
//   (a) in the same wave: NV independent v_fma per MFMA, 1 and 2 waves per SIMD;
//   (b) in the OTHER wave of the SIMD: waves 0-3 issue only MFMAs, waves 4-7 only v_fma.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int SPLIT>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a + i;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = !SPLIT || wave < 4, do_valu = !SPLIT || wave >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (do_valu) {
#pragma unroll
          for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * NV + q) & 15]) : "v"(a), "v"(b));
        }
        if (do_mfma) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NV, int SPLIT>
void run(int threads) {
  float* out;
  unsigned long long* cyc;
  const int wgs = 256, iters = 200;
  (void)hipMalloc(&out, wgs * 512 * 4);
  (void)hipMalloc(&cyc, wgs * 64);
  (void)hipMemset(cyc, 0, wgs * 64);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, SPLIT>), dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipDeviceSynchronize();
  static unsigned long long h[256 * 8];
  (void)hipMemcpy(h, cyc, wgs * 64, hipMemcpyDeviceToHost);
  double lo = 0, hi = 0;
  for (int i = 0; i < wgs; ++i) {
    double a = 0, b = 0;
    for (int w = 0; w < 4; ++w) a += h[i * 8 + w] / 4.0;
    for (int w = 4; w < 8; ++w) b += h[i * 8 + w] / 4.0;
    lo += a / wgs;
    hi += b / wgs;
  }
  const double n = iters * 32.0;
  if (SPLIT)
    printf("specialised waves, %2d v_fma per MFMA slot: MFMA waves %6.1f cycles per MFMA; VALU waves %6.1f cycles per slot (%4.1f per v_fma)\n",
           NV, lo / n, hi / n, hi / n / NV);
  else
    printf("%d wave(s) per SIMD, %2d v_fma per MFMA in the same wave: waves 0-3 %6.1f cycles per MFMA%s\n", threads / 256, NV, lo / n,
           threads > 256 ? (sprintf((char*)h, ", waves 4-7 %6.1f", hi / n), (char*)h) : "");
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<0, 0>(256); run<2, 0>(256); run<4, 0>(256); run<8, 0>(256); run<12, 0>(256); run<16, 0>(256);
  run<0, 0>(512); run<2, 0>(512); run<4, 0>(512); run<8, 0>(512); run<12, 0>(512); run<16, 0>(512);
  run<4, 1>(512); run<8, 1>(512); run<12, 1>(512); run<16, 1>(512);
  return 0;
}
