// Lane layout probe for v_mfma_f32_4x4x1_16B_f32 (gfx950): which lane supplies A[i], B[j] of block b,
// and where D[i][j] of block b lands.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -o /tmp/p mfma4x4.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) d[l * 4 + e] = acc[e];
}
int main() {
  float ha[64], hb[64], hd[256], *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  // A lane l = 1 + l (unique), B lane l = 100 * (1 + l): product identifies (lane_a, lane_b)
  for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 1000.f * (1 + l); }
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dd);
  hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      long v = (long)(hd[l * 4 + e] + 0.5f);
      // v = (1+la) * 1000 * (1+lb): find the pair
      int fa = -1, fb = -1;
      for (int la = 0; la < 64 && fa < 0; ++la)
        for (int lb = 0; lb < 64; ++lb)
          if ((long)(1 + la) * 1000 * (1 + lb) == v && la / 4 == lb / 4) { fa = la; fb = lb; break; }
      printf("  d[%d]=A(lane %2d)*B(lane %2d)", e, fa, fb);
    }
    printf("\n");
    if (l == 7) { printf("...\n"); l = 59; }
  }
  return 0;
}
