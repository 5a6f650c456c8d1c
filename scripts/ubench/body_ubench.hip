// Lone-wavefront cost of the straight-line fp64 blocks of aba_track_kernel (no LDS, no branches): cycles per instruction of real code.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rigidbodydynamics.jl_amd/csrc -I include scripts/ubench/body_ubench.hip -o scripts/ubench/body_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "rbd_track.hpp"
using namespace rbd;

template <typename T> __global__ void k_body(const T* in, T* out, long long* cyc, int N) {
  T rr[24], R[9], p[3], Tw[6], av[6], fe[6], f[24];
  for (int k = 0; k < 24; ++k) rr[k] = in[k] + T(threadIdx.x) * T(1e-3);
  for (int k = 0; k < 9; ++k) R[k] = in[24 + k];
  for (int k = 0; k < 3; ++k) p[k] = in[33 + k];
  for (int k = 0; k < 6; ++k) { Tw[k] = in[36 + k]; av[k] = in[42 + k]; fe[k] = in[48 + k]; }
  I4 w{0, TF_VALID << 16, 0, 0};
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    track_body<T, false>(w, rr, R, p, Tw, av, fe, f);
    for (int k = 0; k < 6; ++k) { av[k] += f[TS_PA + k] * T(1e-9); Tw[k] += f[TS_S + k] * T(1e-9); }
    p[0] += f[TS_W] * T(1e-9); R[0] += f[TS_UD] * T(1e-9);
  }
  long long t1 = clock64();
  T s = 0; for (int k = 0; k < 22; ++k) s += f[k];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename T> __global__ void k_joint(const T* in, T* out, long long* cyc, int N) {
  T IA[21], pA[6], S[6];
  for (int k = 0; k < 21; ++k) IA[k] = in[k] + (SI(k % 6, k % 6) == k ? T(5) : T(0)) + T(threadIdx.x) * T(1e-3);
  for (int k = 0; k < 6; ++k) { pA[k] = in[24 + k]; S[k] = in[30 + k]; }
  IA[0] += 5; IA[6] += 5; IA[11] += 5; IA[15] += 5; IA[18] += 5; IA[20] += 5;
  T tau = in[40];
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
    T U[6], Wd[6];
    sym6_mul(IA, S, U);
    const T Dinv = rcp_hd(dot6(S, U));
    const T ud = (tau - dot6(S, pA)) * Dinv;
    for (int k = 0; k < 6; ++k) Wd[k] = U[k] * Dinv;
    for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) IA[SI(a, b)] = IA[SI(a, b)] - Wd[a] * U[b] * T(1e-6);
    for (int k = 0; k < 6; ++k) pA[k] = pA[k] + U[k] * ud * T(1e-6);
  }
  long long t1 = clock64();
  T s = 0; for (int k = 0; k < 21; ++k) s += IA[k];
  for (int k = 0; k < 6; ++k) s += pA[k];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename T> __global__ void k_fma(const T* in, T* out, long long* cyc, int N) {  // 12 independent chains, distinct registers
  T a[12], b[12], c[12];
  for (int k = 0; k < 12; ++k) { a[k] = in[k]; b[k] = in[12 + k] * T(1e-3); c[k] = in[24 + k]; }
  long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int k = 0; k < 12; ++k) a[k] = a[k] * b[(k + r) % 12] + c[(k + 2 * r + 1) % 12];
  }
  long long t1 = clock64();
  T s = 0; for (int k = 0; k < 12; ++k) s += a[k];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.1 + 0.01 * i;
  float hf[64]; for (int i = 0; i < 64; ++i) hf[i] = (float)h[i];
  double *din, *dout; float *fin, *fout; long long* dc; long long c;
  hipMalloc(&din, sizeof h); hipMalloc(&dout, 8 * 1024 * 256); hipMalloc(&fin, sizeof hf); hipMalloc(&fout, 4 * 1024 * 256); hipMalloc(&dc, 8);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(fin, hf, sizeof hf, hipMemcpyHostToDevice);
  const int N = 200;
  for (int blocks : {1, 1024}) for (int threads : {64, 128, 256}) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_body<double>, blocks, threads, 0, 0, din, dout, dc, N); hipDeviceSynchronize(); }
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("body  f64 blocks=%4d threads=%3d: %8.1f cycles per call\n", blocks, threads, (double)c / N);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_joint<double>, blocks, threads, 0, 0, din, dout, dc, N); hipDeviceSynchronize(); }
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("joint f64 blocks=%4d threads=%3d: %8.1f cycles per call\n", blocks, threads, (double)c / N);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_fma<double>, blocks, threads, 0, 0, din, dout, dc, N); hipDeviceSynchronize(); }
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("fma96 f64 blocks=%4d threads=%3d: %8.1f cycles per call (%.2f per fma)\n", blocks, threads, (double)c / N, (double)c / N / 96);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_body<float>, blocks, threads, 0, 0, fin, fout, dc, N); hipDeviceSynchronize(); }
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("body  f32 blocks=%4d threads=%3d: %8.1f cycles per call\n", blocks, threads, (double)c / N);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_joint<float>, blocks, threads, 0, 0, fin, fout, dc, N); hipDeviceSynchronize(); }
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("joint f32 blocks=%4d threads=%3d: %8.1f cycles per call\n", blocks, threads, (double)c / N);
  }
  return 0;
}
