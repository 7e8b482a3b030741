// Feasibility probe for an fp32-accurate GEMM on the bf16 matrix pipe ("bf16x6"):
//   a = a0 + a1 + a2 exactly (three bf16 pieces of 8 significant bits each = the 24 of an fp32 mantissa),
//   a*b ~ a0*b0 + a0*b1 + a1*b0 + a1*b1 + a0*b2 + a2*b0   (the dropped terms are <= 2^-24 relative),
//   every bf16 product is exact in fp32 and the accumulation is fp32: six v_mfma_f32_32x32x16_bf16 in place of
//   eight v_mfma_f32_32x32x2_f32 per 32x32x16 block, i.e. 192 instead of 512 matrix-pipe cycles.
// Measures (1) the sustained bf16 MFMA rate with RANDOM operands in registers (the clock drops under bf16 load on
// random data) and the resulting fp32-equivalent rate of the 6-product scheme; (2) the error of one 32x32xK product
// computed on the device by the scheme, by three products (bf16x3) and by the fp32 MFMA, against double precision.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16x6 tools/probe/bf16x6.hip && /tmp/bf16x6
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void bf16_loop(const bf16x8* __restrict__ src, float* out, int iters) {
  // each lane gets its own random fragments
  bf16x8 a[3], b[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = src[(threadIdx.x + 64 * i) % 1024];
    b[i] = src[(threadIdx.x + 64 * i + 512) % 1024];
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}

__device__ inline void split3(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)x;
  const float r1 = x - (float)p0;
  p1 = (__bf16)r1;
  p2 = (__bf16)(r1 - (float)p1);
}

// one wave: C[32][32] = A[32][K] * B[K][32] (A row-major, B row-major) by the three schemes
__global__ __launch_bounds__(64) void accuracy(const float* A, const float* B, int K, float* c6, float* c3, float* c32) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc6, acc3, accf;
  for (int e = 0; e < 16; ++e) acc6[e] = acc3[e] = accf[e] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      __bf16 p0, p1, p2;
      split3(A[r * K + k0 + 8 * h + j], p0, p1, p2);
      a[0][j] = p0; a[1][j] = p1; a[2][j] = p2;
      split3(B[(k0 + 8 * h + j) * 32 + r], p0, p1, p2);
      b[0][j] = p0; b[1][j] = p1; b[2][j] = p2;
    }
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc3, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc3, 0, 0, 0);
    // smallest terms first within the step
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc6, 0, 0, 0);
    acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc6, 0, 0, 0);
    for (int kk = 0; kk < 16; kk += 2)       // fp32 MFMA: lane (r, h) supplies A[r][k+h], B[k+h][r]
      accf = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + kk + h], B[(k0 + kk + h) * 32 + r], accf, 0, 0, 0);
  }
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    c6[row * 32 + r] = acc6[e]; c3[row * 32 + r] = acc3[e]; c32[row * 32 + r] = accf[e];
  }
}

int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<unsigned short> rnd(1024 * 8);
  srand(1);
  for (auto& v : rnd) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8* src; hipMalloc(&src, rnd.size() * 2); hipMemcpy(src, rnd.data(), rnd.size() * 2, hipMemcpyHostToDevice);
  float* d; hipMalloc(&d, 1024);
  float ms;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int blocks = 256 * bpc, iters = 20000;
    bf16_loop<<<blocks, 256>>>(src, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bf16_loop<<<blocks, 256>>>(src, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 24;
    const double bf16_flops = mfmas * (32.0 * 32 * 16 * 2);
    printf("bf16 MFMA 32x32x16, random operands, %d wave(s)/SIMD: %.0f TFLOP/s bf16 = %.0f TFLOP/s fp32-equivalent by 6 products "
           "(%.0f by 3)  (%.2f ms)\n", bpc, bf16_flops / ms / 1e9, bf16_flops / 6 / ms / 1e9, bf16_flops / 3 / ms / 1e9, ms);
  }
  for (int K : {576, 2304, 9216}) {
    std::vector<float> A(32 * K), B(K * 32);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dA, *dB, *c6, *c3, *c32;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&c6, 4096); hipMalloc(&c3, 4096); hipMalloc(&c32, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    accuracy<<<1, 64>>>(dA, dB, K, c6, c3, c32);
    std::vector<float> h6(1024), h3(1024), h32(1024);
    hipMemcpy(h6.data(), c6, 4096, hipMemcpyDeviceToHost); hipMemcpy(h3.data(), c3, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(h32.data(), c32, 4096, hipMemcpyDeviceToHost);
    double e6 = 0, e3 = 0, e32 = 0, mag = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * B[k * 32 + j];
        e6 += fabs(h6[i * 32 + j] - ref); e3 += fabs(h3[i * 32 + j] - ref); e32 += fabs(h32[i * 32 + j] - ref); mag += fabs(ref);
      }
    printf("K=%5d  mean |error| / mean |value|:  fp32 MFMA %.2e   bf16x6 %.2e   bf16x3 %.2e\n", K, e32 / mag, e6 / mag, e3 / mag);
  }
  return 0;
}
