// Issue cost of packed fp32 VALU instructions for ONE wavefront per SIMD (the situation of the one-lane-per-state kernels at 65 536 states), beside plain
// v_fma_f32, scalar moves that feed constants, accumulation-register moves and LDS reads.  Inline asm so that the instruction stream is exactly what is named.
// Build: hipcc --offload-arch=gfx950 -O3 pk_ubench.hip -o pk_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N 1024
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DECL8f float x0 = in[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#define DECL8p f2 x0 = {in[threadIdx.x], in[threadIdx.x + 64]}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
#define TIME_BEGIN long long t0 = clock64();
#define TIME_END(sum) long long t1 = clock64(); out[threadIdx.x] = sum; if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;

__global__ void k_fma(const float* in, float* out, long long* t, float a, float b) {
  DECL8f TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##k) : "v"(a), "v"(b));
    REP8(X)
#undef X
  }
  TIME_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
__global__ void k_fmamk(const float* in, float* out, long long* t, float b) {  // literal constant: VOP2 with a 32-bit literal
  DECL8f TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(x##k) : "v"(b));
    REP8(X)
#undef X
  }
  TIME_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
__global__ void k_pkfma(const float* in, float* out, long long* t, float a, float b) {
  DECL8p f2 A = {a, a * 1.5f}, Bv = {b, b * 2.0f};
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x##k) : "v"(A), "v"(Bv));
    REP8(X)
#undef X
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  TIME_END(s.x + s.y)
}
__global__ void k_pkfma_bcast(const float* in, float* out, long long* t, float a, float b) {  // one operand broadcast from the low half (op_sel_hi)
  DECL8p f2 A = {a, a * 1.5f}, Bv = {b, b * 2.0f};
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(x##k) : "v"(A), "v"(Bv));
    REP8(X)
#undef X
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  TIME_END(s.x + s.y)
}
__global__ void k_pkfma_sgpr(const float* in, float* out, long long* t, float a, float b) {  // constant pair in scalar registers
  DECL8p f2 Bv = {b, b * 2.0f};
  unsigned long long A = ((unsigned long long)__float_as_uint(a * 1.5f) << 32) | __float_as_uint(a);
  A = __builtin_amdgcn_readfirstlane((unsigned)A) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(A >> 32)) << 32);
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x##k) : "s"(A), "v"(Bv));
    REP8(X)
#undef X
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  TIME_END(s.x + s.y)
}
__global__ void k_pkfma_smov(const float* in, float* out, long long* t, float a, float b) {  // two s_mov_b32 of literals in front of every packed fma
  DECL8p f2 Bv = {b, b * 2.0f};
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("s_mov_b32 s20, 0x3f800347\n s_mov_b32 s21, 0x3f800123\n v_pk_fma_f32 %0, %0, s[20:21], %1" : "+v"(x##k) : "v"(Bv) : "s20", "s21");
    REP8(X)
#undef X
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  TIME_END(s.x + s.y)
}
__global__ void k_pkfma_smov64(const float* in, float* out, long long* t, float a, float b) {  // one s_mov_b64 of a 64-bit literal? (only 32-bit literals exist: lo half) -> s_mov_b32 + reuse
  DECL8p f2 Bv = {b, b * 2.0f};
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("s_mov_b32 s20, 0x3f800347\n v_pk_fma_f32 %0, %0, s[20:21], %1 op_sel_hi:[1,0,1]" : "+v"(x##k) : "v"(Bv) : "s20", "s21");
    REP8(X)
#undef X
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  TIME_END(s.x + s.y)
}
__global__ void k_pkmul(const float* in, float* out, long long* t, float a) {
  DECL8p f2 A = {a, a * 1.0001f};
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x##k) : "v"(A));
    REP8(X)
#undef X
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  TIME_END(s.x + s.y)
}
__global__ void k_pkadd(const float* in, float* out, long long* t, float a) {
  DECL8p f2 A = {a, a * 1.0001f};
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x##k) : "v"(A));
    REP8(X)
#undef X
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  TIME_END(s.x + s.y)
}
__global__ void k_pkfma_dep(const float* in, float* out, long long* t, float a, float b) {
  f2 x = {in[threadIdx.x], in[threadIdx.x + 64]}, A = {a, a * 1.5f}, Bv = {b, b * 2.0f};
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(A), "v"(Bv));
    REP8(X)
#undef X
  }
  TIME_END(x.x + x.y)
}
__global__ void k_fma_dep(const float* in, float* out, long long* t, float a, float b) {
  float x = in[threadIdx.x];
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    REP8(X)
#undef X
  }
  TIME_END(x)
}
__global__ void k_fma_salu(const float* in, float* out, long long* t, float a, float b) {  // a scalar instruction between every two vector ones
  DECL8f TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_mov_b32 s20, 0x3f800347" : "+v"(x##k) : "v"(a), "v"(b) : "s20");
    REP8(X)
#undef X
  }
  TIME_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
__global__ void k_fma_acc(const float* in, float* out, long long* t, float a, float b) {  // an accumulation-register round trip beside every fma
  DECL8f TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_accvgpr_write_b32 a" #k ", %0" : "+v"(x##k) : "v"(a), "v"(b) : "a" #k);
    REP8(X)
#undef X
  }
  TIME_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
__global__ void k_fma_mov(const float* in, float* out, long long* t, float a, float b) {  // a v_mov beside every fma (same issue port?)
  DECL8f float y0 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0, y6 = 0, y7 = 0; TIME_BEGIN
  for (int i = 0; i < N; ++i) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_mov_b32 %1, %0" : "+v"(x##k), "=v"(y##k) : "v"(a), "v"(b));
    REP8(X)
#undef X
  }
  TIME_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7)
}
__global__ void k_ds32(const float* in, float* out, long long* t, float a, float b) {  // one ds_read_b32 per two fmas, consumed 8 instructions later
  __shared__ float lds[65 * 64];
  for (int k = 0; k < 64; ++k) lds[k * 65 + threadIdx.x] = in[threadIdx.x] + k;
  __syncthreads();
  DECL8f float y0 = 0, y1 = 0, y2 = 0, y3 = 0; const unsigned ad = threadIdx.x * 4;
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
    asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:260\n ds_read_b32 %2, %4 offset:520\n ds_read_b32 %3, %4 offset:780" : "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3) : "v"(ad));
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##k) : "v"(a), "v"(b));
    REP8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %6, %6, %7" : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(x3), "+v"(y3));
  }
  TIME_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
__global__ void k_ds64(const float* in, float* out, long long* t, float a, float b) {  // the same four values as two ds_read2_b32 (two rows per instruction)
  __shared__ float lds[65 * 64];
  for (int k = 0; k < 64; ++k) lds[k * 65 + threadIdx.x] = in[threadIdx.x] + k;
  __syncthreads();
  DECL8f f2 y0 = {0, 0}, y1 = {0, 0}; const unsigned ad = threadIdx.x * 4;
  TIME_BEGIN
  for (int i = 0; i < N; ++i) {
    asm volatile("ds_read2_b32 %0, %2 offset1:65\n ds_read2_b32 %1, %2 offset0:130 offset1:195" : "=v"(y0), "=v"(y1) : "v"(ad));
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##k) : "v"(a), "v"(b));
    REP8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %2, %2, %3" : "+v"(y0), "+v"(y1), "+v"(y0), "+v"(y1));
    x0 += y0.x; x1 += y1.y;
  }
  TIME_END(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7)
}
template <typename F> void run(const char* name, F launch, int blocks, double ops) {
  long long* t; hipMalloc(&t, sizeof(long long) * blocks);
  launch(t); launch(t); hipDeviceSynchronize();
  std::vector<long long> h(blocks); hipMemcpy(h.data(), t, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (int b = 0; b < blocks; ++b) avg += h[b]; avg /= blocks;
  printf("%-56s waves=%4d  cycles per named group = %7.2f\n", name, blocks, avg / ops);
  hipFree(t);
}
int main() {
  float *in, *out; hipMalloc(&in, 1 << 16); hipMalloc(&out, 1 << 16); hipMemset(in, 0, 1 << 16);
  for (int blocks : {1, 1024}) {
    const double G = 8.0 * N;
    run("v_fma_f32 (8 independent)", [&](long long* t) { k_fma<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_fmamk_f32 literal (8 independent)", [&](long long* t) { k_fmamk<<<blocks, 64>>>(in, out, t, 1e-9f); }, blocks, G);
    run("v_pk_fma_f32 vgpr pairs", [&](long long* t) { k_pkfma<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_pk_fma_f32 one operand broadcast (op_sel_hi)", [&](long long* t) { k_pkfma_bcast<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_pk_fma_f32 one operand an SGPR pair", [&](long long* t) { k_pkfma_sgpr<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("2 s_mov_b32 literal + v_pk_fma_f32 sgpr pair", [&](long long* t) { k_pkfma_smov<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("1 s_mov_b32 literal + v_pk_fma_f32 sgpr broadcast", [&](long long* t) { k_pkfma_smov64<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_pk_mul_f32", [&](long long* t) { k_pkmul<<<blocks, 64>>>(in, out, t, 1.0000001f); }, blocks, G);
    run("v_pk_add_f32", [&](long long* t) { k_pkadd<<<blocks, 64>>>(in, out, t, 1e-9f); }, blocks, G);
    run("v_pk_fma_f32 dependent chain", [&](long long* t) { k_pkfma_dep<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_fma_f32 dependent chain", [&](long long* t) { k_fma_dep<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_fma_f32 + s_mov_b32 literal", [&](long long* t) { k_fma_salu<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_fma_f32 + v_accvgpr_write_b32", [&](long long* t) { k_fma_acc<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("v_fma_f32 + v_mov_b32", [&](long long* t) { k_fma_mov<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, G);
    run("per iteration: 4 ds_read_b32 + 8 v_fma + wait + 4 v_add", [&](long long* t) { k_ds32<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, (double)N);
    run("per iteration: 2 ds_read2_b32 + 8 v_fma + wait + 2 pk_add+2", [&](long long* t) { k_ds64<<<blocks, 64>>>(in, out, t, 1.0000001f, 1e-9f); }, blocks, (double)N);
  }
  return 0;
}
