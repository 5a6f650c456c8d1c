// Probe of v_mfma_f32_4x4x1_16B_f32 lane/register layout on gfx950 (used by the batched tile Cholesky).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + l; }   // A value identifies the lane, B value too
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  // decode: each output = A[la] * B[lb] -> find (la, lb)
  for (int l = 0; l < 64; l += 1) {
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      int fa = -1, fb = -1;
      for (int la = 0; la < 64 && fa < 0; ++la) for (int lb = 0; lb < 64; ++lb) if (hd[l * 4 + r] == ha[la] * hb[lb]) { fa = la; fb = lb; break; }
      printf("  reg%d = A[lane %2d] * B[lane %2d]", r, fa, fb);
    }
    printf("\n");
    if (l == 7) l = 55;
  }
  return 0;
}
