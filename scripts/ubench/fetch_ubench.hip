// Does a lone wavefront's STRAIGHT-LINE instruction stream (no loop: every instruction fetched once, as in the kernels compiled per mechanism) issue at the rate
// of its arithmetic, or at the rate its code bytes arrive?  4096 independent instructions (8 accumulators round-robin) of a 4-byte encoding (v_fmac_f32_e32), of
// 8-byte encodings (v_fma_f32, v_pk_fma_f32, v_fma_f64), and mixes, timed with s_memtime around the stream; one wavefront, one per SIMD (1024), two per SIMD (2048).
// Build: hipcc --offload-arch=gfx950 -O3 fetch_ubench.hip -o fetch_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define STREAM8(I0, I1, I2, I3, I4, I5, I6, I7) ".rept 512\n" I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n.endr\n"
#define KERNEL(NAME, TYPE, INIT, ASM, SUM)                                                                         \
  __global__ void NAME(const float* in, float* out, long long* t, float a, float b) {                             \
    TYPE x0 = INIT(0), x1 = INIT(1), x2 = INIT(2), x3 = INIT(3), x4 = INIT(4), x5 = INIT(5), x6 = INIT(6), x7 = INIT(7); \
    TYPE A = INIT(8), Bv = INIT(9);                                                                                \
    long long t0 = clock64();                                                                                      \
    asm volatile(ASM : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(A), "v"(Bv)); \
    long long t1 = clock64();                                                                                      \
    out[threadIdx.x + blockIdx.x * 64] = SUM;                                                                      \
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;                                                                 \
  }
#define INITF(k) (in[threadIdx.x] + k)
#define INITP(k) f2{in[threadIdx.x] + k, in[threadIdx.x + 64] + k}
#define INITD(k) ((double)in[threadIdx.x] + k)
#define SUMF (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
#define SUMP ((x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7).x)
#define SUMD ((float)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7))
KERNEL(k_fmac4, float, INITF, STREAM8("v_fmac_f32_e32 %0, %8, %9", "v_fmac_f32_e32 %1, %8, %9", "v_fmac_f32_e32 %2, %8, %9", "v_fmac_f32_e32 %3, %8, %9", "v_fmac_f32_e32 %4, %8, %9", "v_fmac_f32_e32 %5, %8, %9", "v_fmac_f32_e32 %6, %8, %9", "v_fmac_f32_e32 %7, %8, %9"), SUMF)
KERNEL(k_fma8, float, INITF, STREAM8("v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %2, %2, %8, %9", "v_fma_f32 %3, %3, %8, %9", "v_fma_f32 %4, %4, %8, %9", "v_fma_f32 %5, %5, %8, %9", "v_fma_f32 %6, %6, %8, %9", "v_fma_f32 %7, %7, %8, %9"), SUMF)
KERNEL(k_pk8, f2, INITP, STREAM8("v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9", "v_pk_fma_f32 %4, %4, %8, %9", "v_pk_fma_f32 %5, %5, %8, %9", "v_pk_fma_f32 %6, %6, %8, %9", "v_pk_fma_f32 %7, %7, %8, %9"), SUMP)
KERNEL(k_f64, double, INITD, STREAM8("v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %1, %1, %8, %9", "v_fma_f64 %2, %2, %8, %9", "v_fma_f64 %3, %3, %8, %9", "v_fma_f64 %4, %4, %8, %9", "v_fma_f64 %5, %5, %8, %9", "v_fma_f64 %6, %6, %8, %9", "v_fma_f64 %7, %7, %8, %9"), SUMD)
KERNEL(k_mix48, float, INITF, STREAM8("v_fmac_f32_e32 %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fmac_f32_e32 %2, %8, %9", "v_fma_f32 %3, %3, %8, %9", "v_fmac_f32_e32 %4, %8, %9", "v_fma_f32 %5, %5, %8, %9", "v_fmac_f32_e32 %6, %8, %9", "v_fma_f32 %7, %7, %8, %9"), SUMF)
// dependent pairs: each instruction reads the result of the one before it (chains of 2, 4 apart) — what a dependency costs in a straight-line stream
KERNEL(k_dep1, float, INITF, STREAM8("v_fmac_f32_e32 %0, %8, %9", "v_fmac_f32_e32 %0, %8, %9", "v_fmac_f32_e32 %1, %8, %9", "v_fmac_f32_e32 %1, %8, %9", "v_fmac_f32_e32 %2, %8, %9", "v_fmac_f32_e32 %2, %8, %9", "v_fmac_f32_e32 %3, %8, %9", "v_fmac_f32_e32 %3, %8, %9"), SUMF)
KERNEL(k_dep2, float, INITF, STREAM8("v_fmac_f32_e32 %0, %8, %9", "v_fmac_f32_e32 %1, %8, %9", "v_fmac_f32_e32 %0, %8, %9", "v_fmac_f32_e32 %1, %8, %9", "v_fmac_f32_e32 %2, %8, %9", "v_fmac_f32_e32 %3, %8, %9", "v_fmac_f32_e32 %2, %8, %9", "v_fmac_f32_e32 %3, %8, %9"), SUMF)
KERNEL(k_pkdep1, f2, INITP, STREAM8("v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9"), SUMP)
KERNEL(k_pkdep2, f2, INITP, STREAM8("v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9"), SUMP)
KERNEL(k_pkdep3, f2, INITP, STREAM8("v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9", "v_pk_fma_f32 %4, %4, %8, %9"), SUMP)
KERNEL(k_f64dep1, double, INITD, STREAM8("v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %1, %1, %8, %9", "v_fma_f64 %1, %1, %8, %9", "v_fma_f64 %2, %2, %8, %9", "v_fma_f64 %2, %2, %8, %9", "v_fma_f64 %3, %3, %8, %9", "v_fma_f64 %3, %3, %8, %9"), SUMD)
KERNEL(k_f64dep2, double, INITD, STREAM8("v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %1, %1, %8, %9", "v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %1, %1, %8, %9", "v_fma_f64 %2, %2, %8, %9", "v_fma_f64 %3, %3, %8, %9", "v_fma_f64 %2, %2, %8, %9", "v_fma_f64 %3, %3, %8, %9"), SUMD)
template <typename K> void run(const char* name, K kern, int blocks, const float* in, float* out) {
  long long* t; (void)hipMalloc(&t, sizeof(long long) * blocks);
  kern<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); kern<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); (void)hipDeviceSynchronize();
  std::vector<long long> h(blocks); (void)hipMemcpy(h.data(), t, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (int b = 0; b < blocks; ++b) avg += h[b]; avg /= blocks;
  printf("%-44s waves=%4d  cycles per instruction = %6.2f\n", name, blocks, avg / 4096.0);
  (void)hipFree(t);
}
int main() {
  float *in, *out; (void)hipMalloc(&in, 1 << 16); (void)hipMalloc(&out, 1 << 20); (void)hipMemset(in, 0, 1 << 16);
  for (int blocks : {1, 1024, 2048}) {
    run("v_fmac_f32_e32 (4 bytes) independent", k_fmac4, blocks, in, out);
    run("v_fma_f32 (8 bytes) independent", k_fma8, blocks, in, out);
    run("v_pk_fma_f32 (8 bytes) independent", k_pk8, blocks, in, out);
    run("v_fma_f64 (8 bytes) independent", k_f64, blocks, in, out);
    run("v_fmac_e32 / v_fma alternating (4 + 8 bytes)", k_mix48, blocks, in, out);
    run("v_fmac_f32_e32 dependent on the one before", k_dep1, blocks, in, out);
    run("v_fmac_f32_e32 dependent on two before", k_dep2, blocks, in, out);
    run("v_pk_fma_f32 dependent on the one before", k_pkdep1, blocks, in, out);
    run("v_pk_fma_f32 dependent on two before", k_pkdep2, blocks, in, out);
    run("v_pk_fma_f32 dependent on three before", k_pkdep3, blocks, in, out);
    run("v_fma_f64 dependent on the one before", k_f64dep1, blocks, in, out);
    run("v_fma_f64 dependent on two before", k_f64dep2, blocks, in, out);
  }
  return 0;
}
