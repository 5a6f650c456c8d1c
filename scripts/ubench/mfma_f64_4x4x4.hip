// Probe of v_mfma_f64_4x4x4f64 lane layout on gfx950 (4 blocks of D(4x4) = A(4x4) B(4x4) + C, one f64 per lane for A, B and D):
// for every lane la, A = one-hot at la and B = 100 + lane; the non-zero outputs tell which D lanes use A[la] and which B lane they pair it with.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* d) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    const double a = (l == la) ? 1.0 : 0.0, b = 100.0 + l;
    d[la * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  }
}
int main() {
  double* d; static double hd[64 * 64];
  hipMalloc(&d, sizeof hd);
  probe<<<1, 64>>>(d);
  hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d ->", la);
    for (int l = 0; l < 64; ++l) if (hd[la * 64 + l] != 0.0) printf(" D%d<-B%d", l, (int)(hd[la * 64 + l] - 100.0));
    printf("\n");
  }
  return 0;
}
