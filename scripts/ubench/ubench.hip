// Micro-benchmarks that calibrate the cost model used in DESIGN.md: shader clock, dependent/independent fp64 & fp32
// FMA, DPP wave shift, ds_bpermute, exec-masked branch.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 2048
template <typename T> __global__ void dep_fma(T* out, long long* t, T a, T b) {
  T x = out[threadIdx.x];
  long long t0 = clock64(); long long w0 = wall_clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = x * a + b;
  long long t1 = clock64(); long long w1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
template <typename T> __global__ void indep_fma(T* out, long long* t, T a, T b) {
  T x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  long long t0 = clock64();
#pragma unroll 4
  for (int i = 0; i < N / 8; ++i) { x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b; x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b; }
  long long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = 0; }
}
__global__ void dpp_chain(float* out, long long* t) {
  float x = out[threadIdx.x];
  long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true)) + 1.0f;
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = 0; }
}
__global__ void bperm_chain(float* out, long long* t) {
  float x = out[threadIdx.x];
  int addr = ((threadIdx.x + 63) & 63) << 2;
  long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x))) + 1.0f;
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = 0; }
}
__global__ void bperm_tput(float* out, long long* t) {  // 8 independent bpermutes per step
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = out[threadIdx.x] + k;
  int addr = ((threadIdx.x + 63) & 63) << 2;
  long long t0 = clock64();
#pragma unroll 2
  for (int i = 0; i < N / 8; ++i)
    for (int k = 0; k < 8; ++k) x[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x[k])));
  long long t1 = clock64();
  float s = 0; for (int k = 0; k < 8; ++k) s += x[k];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = 0; }
}
__global__ void branch_chain(float* out, long long* t, int lim) {
  float x = out[threadIdx.x];
  long long t0 = clock64();
  for (int i = 0; i < N; ++i) { if ((int)(threadIdx.x & 31) == (i & lim)) x = x * 1.0001f + 1.0f; }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = 0; }
}
template <typename F> void run(const char* name, F launch, int blocks, int threads, double per) {
  long long* t; hipMalloc(&t, sizeof(long long) * 2 * blocks);
  launch(t); launch(t); hipDeviceSynchronize();
  std::vector<long long> h(2 * blocks); hipMemcpy(h.data(), t, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (int b = 0; b < blocks; ++b) avg += h[2 * b]; avg /= blocks;
  printf("%-34s blocks=%4d threads=%3d  cycles/op = %7.2f", name, blocks, threads, avg / per);
  if (h[1] > 0) printf("   shader clock = %.0f MHz", (double)h[0] / ((double)h[1] / 100.0));
  printf("\n"); hipFree(t);
}
int main() {
  void* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
  for (int blocks : {1, 1024}) for (int threads : {64, 256, 512}) {
    run("dep fma f64", [&](long long* t) { dep_fma<double><<<blocks, threads>>>((double*)buf, t, 1.0000001, 1e-9); }, blocks, threads, N);
    run("dep fma f32", [&](long long* t) { dep_fma<float><<<blocks, threads>>>((float*)buf, t, 1.0000001f, 1e-9f); }, blocks, threads, N);
    run("indep(8) fma f64", [&](long long* t) { indep_fma<double><<<blocks, threads>>>((double*)buf, t, 1.0000001, 1e-9); }, blocks, threads, N);
    run("indep(8) fma f32", [&](long long* t) { indep_fma<float><<<blocks, threads>>>((float*)buf, t, 1.0000001f, 1e-9f); }, blocks, threads, N);
    run("dep dpp wave_shr + add f32", [&](long long* t) { dpp_chain<<<blocks, threads>>>((float*)buf, t); }, blocks, threads, N);
    run("dep bpermute + add f32", [&](long long* t) { bperm_chain<<<blocks, threads>>>((float*)buf, t); }, blocks, threads, N);
    run("indep(8) bpermute", [&](long long* t) { bperm_tput<<<blocks, threads>>>((float*)buf, t); }, blocks, threads, N);
    run("masked branch (1/32 lanes) + fma", [&](long long* t) { branch_chain<<<blocks, threads>>>((float*)buf, t, 31); }, blocks, threads, N);
  }
  return 0;
}
