// Micro-benchmark (GPU box): what a vector-memory instruction costs the wave that issues it when all waves of a CU issue at once.
// Each wave issues N loads back to back (addresses that hit the vector L1 / L2), stamps the time the ISSUE took (s_memtime before
// and after, no wait for data), then waits for the data; repeated.  Kinds:
//   0 buffer_load_dwordx4, the wave's 64 x 16 B contiguous (8 lines of 128 B)
//   1 buffer_load_dwordx4, lane pairs 1 KB apart (the F(4x4) loader's pattern: 32 B of each pixel's 1 KB channel vector) -> 32 lines
//   2 buffer_load_dwordx2, same pattern (16 B of a line per lane pair)
//   3 global_load_lds_dwordx4 (LDS-DMA), contiguous
//   4 as 1 with lanes 12-15 of every 16 pointing outside the buffer (the loader's idle lanes)
//   5 as 1 with lanes 12-15 of every 16 switched off in EXEC
//   hipcc -O3 --offload-arch=gfx950 -o vmem_issue vmem_issue.hip && ./vmem_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int N>
__global__ __launch_bounds__(512) void k(const float* x, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* base = x + (size_t)blockIdx.x * (64 << 10) / 4;       // 64 KB per workgroup
  i32x4 src;
  {
    const unsigned long long xb = (unsigned long long)base;
    src[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb);
    src[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xb >> 32) & 0xffffu));
    src[2] = 64 << 10;
    src[3] = 0x00020000;
  }
  unsigned off;
  if (KIND == 0 || KIND == 3) off = wave * 1024 * N + lane * 16;                       // + i * 1024
  else off = ((lane >> 1) * 1024 + (lane & 1) * 16 + wave * 64) & 0xffff;             // + i * 32 (the next K-step's bytes)
  if (KIND == 4 && (lane & 15) >= 12) off = 0x80000000u;
  f32x4 r[N];
  f32x2 r2[N];
  float acc = 0.f;
  unsigned long long t_issue = 0, t_total = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 5) asm volatile("s_mov_b32 exec_lo, 0x0fff0fff\n\ts_mov_b32 exec_hi, 0x0fff0fff" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int so = __builtin_amdgcn_readfirstlane(KIND == 0 || KIND == 3 ? i * 1024 : ((it & 3) * N + i) * 32 % 1024);
      if (KIND == 3) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off + so) / 4),
                                         (__attribute__((address_space(3))) void*)(smem + (wave * N + i) * 256), 16, 0, 0);
      } else if (KIND == 2) {
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(r2[i]) : "v"(off), "s"(src), "s"(so) : "memory");
      } else {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[i]) : "v"(off), "s"(src), "s"(so) : "memory");
      }
    }
    if (KIND == 5) asm volatile("s_mov_b64 exec, -1" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (KIND == 2) asm volatile("" : "+v"(r2[i]));
      else if (KIND != 3) asm volatile("" : "+v"(r[i]));
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_issue += t1 - t0;
    t_total += t2 - t0;
    if (KIND == 3) acc += smem[(wave * N) * 256 + lane];
    else if (KIND == 2) acc += r2[0][0] + r2[N - 1][1];
    else acc += r[0][0] + r[N - 1][3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (lane == 0) {
    cyc[(blockIdx.x * 8 + wave) * 2] = t_issue;
    cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t_total;
  }
}

template <int KIND, int N>
void run(int threads, const float* x) {
  float* out;
  unsigned long long* cyc;
  const int wgs = 256, iters = 400;
  (void)hipMalloc(&out, wgs * 512 * 4);
  (void)hipMalloc(&cyc, wgs * 128);
  (void)hipMemset(cyc, 0, wgs * 128);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, N>), dim3(wgs), dim3(threads), 96 << 10, 0, x, out, cyc, iters);
  (void)hipDeviceSynchronize();
  static unsigned long long h[256 * 16];
  (void)hipMemcpy(h, cyc, wgs * 128, hipMemcpyDeviceToHost);
  const int waves = threads / 64;
  double is = 0, tot = 0, ismax = 0;
  for (int i = 0; i < wgs; ++i)
    for (int w = 0; w < waves; ++w) {
      const double a = (double)h[(i * 8 + w) * 2] / iters, b = (double)h[(i * 8 + w) * 2 + 1] / iters;
      is += a / (wgs * waves);
      tot += b / (wgs * waves);
      ismax = a > ismax ? a : ismax;
    }
  static const char* names[] = {"buffer_load_dwordx4 contiguous", "buffer_load_dwordx4 32 lines", "buffer_load_dwordx2 32 lines",
                                "global_load_lds_dwordx4", "dwordx4 32 lines, 12/16 in range", "dwordx4 32 lines, 12/16 in EXEC"};
  printf("%-34s N=%d, %d waves/CU: issue of the N loads %7.1f cycles/wave (max %7.1f) = %5.1f per load; until the data is there %7.1f; "
         "CU cycles per wave-instruction %5.1f\n",
         names[KIND], N, waves, is, ismax, is / N, tot, tot / (N * waves));
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  float* x;
  (void)hipMalloc(&x, (size_t)256 * (64 << 10));
  (void)hipMemset(x, 0, (size_t)256 * (64 << 10));
  const int T[3] = {64, 256, 512};
  for (int t = 0; t < 3; ++t) {
    run<0, 8>(T[t], x); run<1, 8>(T[t], x); run<2, 8>(T[t], x); run<3, 8>(T[t], x); run<4, 8>(T[t], x); run<5, 8>(T[t], x);
  }
  run<1, 1>(512, x); run<1, 2>(512, x); run<1, 4>(512, x); run<3, 1>(512, x); run<3, 2>(512, x);
  return 0;
}
