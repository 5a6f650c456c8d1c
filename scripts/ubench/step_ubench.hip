// Cost of the real step functions of aba_track_kernel in isolation (one wavefront, rows and plan records in LDS, steady-state steps:
// every lane chained, no mailbox traffic): cycles per call.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rigidbodydynamics.jl_amd/csrc
//   -I include scripts/ubench/step_ubench.hip -o scripts/ubench/step_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "rbd_track.hpp"
using namespace rbd;

template <typename T, int WHICH> __global__ __launch_bounds__(64) void k_step(TrackModel M, const T* seed, T* out, long long* cyc, int N) {
  extern __shared__ double lds_raw[];
  TrackCtx<T> c;
  c.M = M; c.B = 16; c.q = seed; c.v = seed; c.tau = seed; c.fext = nullptr; c.vdot = out; c.qdot = nullptr;
  c.Lq.sk = 1; c.Lq.sb = 40; c.Lv = c.Lq; c.Lf = c.Lq;
  track_ctx_lds(c, lds_raw);
  track_stage_plan(c, threadIdx.x, 64);
  __syncthreads();
  TrackLane<T> L;
  track_init<T, 4>(c, L, threadIdx.x, 0);
  using D = TrackDims<T>;
  for (int s = 0; s < M.ns; ++s) {  // plausible rows
    T f[D::NF];
    for (int k = 0; k < D::NF; ++k) f[k] = seed[k] + T(0.01) * T(threadIdx.x % 7);
    f[0] += 3; f[3] += 3; f[5] += 3;  // J diagonal dominant
    track_row_write<T, 0, D::NV>(track_row(c, s, threadIdx.x), f);
  }
  for (int k = 0; k < 9; ++k) L.R[k] = (k % 4 == 0) ? T(1) : T(0.01);
  __syncthreads();
  long long t0, t1;
  if (WHICH == 0) {
    track_prologue_a<T, 4, 1>(c, L);
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { track_step_a<T, 4, false, false, 1, 0>(c, L, 2); asm volatile("" ::: "memory"); track_step_a<T, 4, false, false, 1, 1>(c, L, 3); asm volatile("" ::: "memory"); }
    t1 = clock64();
  } else if (WHICH == 1) {
    track_prologue_b<T, 4, 1>(c, L);
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { track_step_b<T, 4, false, false, 0>(c, L, 3); asm volatile("" ::: "memory"); track_step_b<T, 4, false, false, 1>(c, L, 2); asm volatile("" ::: "memory"); }
    t1 = clock64();
  } else if (WHICH == 2) {
    track_prologue_c<T, 4>(c, L);
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { track_step_c<T, 4, false, false, 0>(c, L, 2); asm volatile("" ::: "memory"); track_step_c<T, 4, false, false, 1>(c, L, 3); asm volatile("" ::: "memory"); }
    t1 = clock64();
  } else {
    track_prologue_a<T, 4, 4>(c, L);
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { track_step_a<T, 4, false, false, 4, 0>(c, L, 2); asm volatile("" ::: "memory"); track_step_a<T, 4, false, false, 4, 1>(c, L, 3); asm volatile("" ::: "memory"); }
    t1 = clock64();
  }
  T sum = L.R[0] + L.cI[0] + L.ad[0] + L.av[0];
  out[64 + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int ns = 6, G = 4;
  std::vector<int32_t> ri(ns * G * 4, 0);
  std::vector<double> rr(ns * G * 24, 0.0);
  for (int s = 0; s < ns; ++s) for (int g = 0; g < G; ++g) {
    int32_t* w = &ri[(s * G + g) * 4];
    w[0] = (s * G + g) | ((s * G + g) << 16);
    w[1] = (6 * g) | ((TF_VALID | TF_CHAINED | (s + 1 < ns ? TF_CARRY : 0)) << 16);
    double* r = &rr[(s * G + g) * 24];
    r[0] = r[4] = r[8] = 1.0; r[1] = 0.01; r[9] = 0.1; r[12] = 1; r[15] = 1; r[17] = 1; r[18] = 0.1; r[21] = 2.0;
  }
  int32_t* dri; double* drr; double *seed, *out; long long* dc; long long c;
  hipMalloc(&dri, ri.size() * 4); hipMalloc(&drr, rr.size() * 8); hipMalloc(&seed, 4096 * 8); hipMalloc(&out, 4096 * 8); hipMalloc(&dc, 8);
  hipMemcpy(dri, ri.data(), ri.size() * 4, hipMemcpyHostToDevice); hipMemcpy(drr, rr.data(), rr.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> hs(4096); for (int i = 0; i < 4096; ++i) hs[i] = 0.1 + 0.001 * (i % 97);
  hipMemcpy(seed, hs.data(), 4096 * 8, hipMemcpyHostToDevice);
  TrackModel M{}; M.ns = ns; M.G = G; M.nA = 1; M.nB = 1; M.ri = dri; M.rr = drr; M.gravity[2] = -9.81;
  const size_t lds = track_lds_bytes<double>(M);
  const int N = 100;
#define RUN(W, name) hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step<double, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k_step<double, W>), 1, 64, lds, 0, M, seed, out, dc, N); hipDeviceSynchronize(); } \
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.1f cycles per step\n", name, (double)c / (2 * N));
  RUN(0, "step_a (NW=1: chain + body)");
  RUN(3, "step_a (NW=4: chain only)");
  RUN(1, "step_b");
  RUN(2, "step_c");
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
