// track_emu.hip — CPU emulation of aba_track_kernel (rigidbodydynamics.jl_amd/csrc/rbd_track.hpp) — TEST INFRASTRUCTURE ONLY.
// The kernel's step functions are __host__ __device__; this harness runs them lane by lane, step by step, with the LDS replaced
// by a host buffer, on the packed plan records exported by rbd_model_track_plan.  It lets the CPU test suite (-m "not gpu")
// check the arithmetic and the plan / mailbox bookkeeping of the kernel against the oracle without a GPU.  Built on demand by
// tests/test_track_emu.py with `hipcc --cuda-host-only`; never linked into librbd_hip.so.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rbd_track.hpp"

using namespace rbd;

template <typename T, int G, bool FLT, bool GEN, int NW>
static int emu_run(const TrackModel& M, long B, const T* q, const T* v, const T* tau, const T* fext, T* vdot, T* qdot, Layout Lq, Layout Lv, Layout Lf) {
  constexpr int SPW = 64 / G;
  const size_t lds_bytes = track_lds_bytes<T>(M);
  std::vector<char> lds(lds_bytes + 64);
  void* base = (void*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
  const long ngroups = (B + SPW - 1) / SPW;
  const int ns = M.ns;
  for (long group = 0; group < ngroups; ++group) {
    memset(base, 0xff, lds_bytes);  // NaN pattern: a read of a never-written field shows up
    TrackCtx<T> c;
    c.M = M; c.B = B; c.q = q; c.v = v; c.tau = tau; c.fext = fext; c.vdot = vdot; c.qdot = qdot; c.Lq = Lq; c.Lv = Lv; c.Lf = Lf;
    track_ctx_lds(c, base);
    for (int i = 0; i < ns + 4; ++i) c.flags[i] = 0;
    std::vector<TrackLane<T>> L(64);
    std::vector<TrackWho> H(64);
    for (int l = 0; l < 64; ++l) { track_init<T, G>(c, L[l], l, group); track_who<T, G>(c, H[l], l, group); }
    // the staging prologue, then (NW = 4) the roles in dependency order: per step the chain wave, then the helper's body part
    for (int t = 0; t < 64 * NW; ++t) track_stage_plan(c, t, 64 * NW);
    for (int wv = 0; wv < NW; ++wv)
      for (int l = 0; l < 64; ++l) track_stage_rows<T, G, FLT, GEN, (NW == 1 ? 6 : 3)>(c, H[l], wv, NW);
    for (int l = 0; l < 64; ++l) track_prologue_a<T, G, NW>(c, L[l]);
    for (int s = 0; s < ns; ++s) {
      for (int l = 0; l < 64; ++l) {
        if (s & 1) track_step_a<T, G, FLT, GEN, NW, 1>(c, L[l], s); else track_step_a<T, G, FLT, GEN, NW, 0>(c, L[l], s);
      }
      if (NW > 1)
        for (int l = 0; l < 64; ++l) track_helper_body<T, G, GEN>(c, H[l], s);
    }
    if (NW > 1)
      for (int h = 1; h <= 3; ++h) track_flag_set(c.flags + ns + h);
    for (int l = 0; l < 64; ++l) track_prologue_b<T, G, NW>(c, L[l]);
    for (int s = ns - 1; s >= 0; --s)
      for (int l = 0; l < 64; ++l) {
        if ((ns - 1 - s) & 1) track_step_b<T, G, FLT, GEN, 1>(c, L[l], s); else track_step_b<T, G, FLT, GEN, 0>(c, L[l], s);
      }
    for (int l = 0; l < 64; ++l) track_prologue_c<T, G>(c, L[l]);
    for (int s = 0; s < ns; ++s)
      for (int l = 0; l < 64; ++l) {
        if (s & 1) track_step_c<T, G, FLT, GEN, 1>(c, L[l], s); else track_step_c<T, G, FLT, GEN, 0>(c, L[l], s);
      }
  }
  return 0;
}

template <typename T, int G>
static int emu_fg(const TrackModel& M, int flt, int gen, int nw, long B, const T* q, const T* v, const T* tau, const T* fext, T* vdot, T* qdot, Layout Lq, Layout Lv, Layout Lf) {
#define RUN(F, GN) (nw == 4 ? emu_run<T, G, F, GN, 4>(M, B, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf) : emu_run<T, G, F, GN, 1>(M, B, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf))
  if (flt) return gen ? RUN(true, true) : RUN(true, false);
  return gen ? RUN(false, true) : RUN(false, false);
#undef RUN
}

template <typename T>
static int emu_t(const int32_t* dims, const int32_t* ri, const double* rr, const double* gravity, int nw, int aos, long B, int nq, int nv, int nb, const void* q,
                 const void* v, const void* tau, const void* fext, void* vdot, void* qdot) {
  TrackModel M;
  M.G = dims[0]; M.ns = dims[1]; M.nA = dims[2]; M.nB = dims[3];
  std::vector<T> rrt(rr, rr + (size_t)M.ns * M.G * TR_STRIDE + 0);
  // 16-byte aligned copies of the records
  std::vector<char> ribuf((size_t)M.ns * M.G * TI_STRIDE * 4 + 32), rrbuf(rrt.size() * sizeof(T) + 32);
  int32_t* ria = (int32_t*)(((uintptr_t)ribuf.data() + 15) & ~(uintptr_t)15);
  T* rra = (T*)(((uintptr_t)rrbuf.data() + 15) & ~(uintptr_t)15);
  memcpy(ria, ri, (size_t)M.ns * M.G * TI_STRIDE * 4);
  memcpy(rra, rrt.data(), rrt.size() * sizeof(T));
  M.ri = ria; M.rr = rra;
  const int32_t* sf = ri + (size_t)M.ns * M.G * TI_STRIDE;  // the per-step flags follow the packed records
  for (int k = 0; k < 5; ++k) { M.sfm[k] = 0; for (int s = 0; s < M.ns; ++s) M.sfm[k] |= (uint64_t)((sf[s] >> k) & 1) << s; }
  memcpy(M.gravity, gravity, sizeof M.gravity);
  auto lay = [&](long n) { Layout L; if (aos) { L.sk = 1; L.sb = n; } else { L.sk = B; L.sb = 1; } return L; };
  const Layout Lq = lay(nq), Lv = lay(nv), Lf = lay(6L * nb);
  switch (M.G) {
    case 1: return emu_fg<T, 1>(M, dims[4], dims[5], nw, B, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, (T*)vdot, (T*)qdot, Lq, Lv, Lf);
    case 2: return emu_fg<T, 2>(M, dims[4], dims[5], nw, B, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, (T*)vdot, (T*)qdot, Lq, Lv, Lf);
    case 4: return emu_fg<T, 4>(M, dims[4], dims[5], nw, B, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, (T*)vdot, (T*)qdot, Lq, Lv, Lf);
  }
  return 1;
}

extern "C" int track_emu_dynamics(const int32_t* dims, const int32_t* ri, const double* rr, const double* gravity, int f32, int nw, int aos, long B, int nq,
                                  int nv, int nb, const void* q, const void* v, const void* tau, const void* fext, void* vdot, void* qdot) {
  return f32 ? emu_t<float>(dims, ri, rr, gravity, nw, aos, B, nq, nv, nb, q, v, tau, fext, vdot, qdot)
             : emu_t<double>(dims, ri, rr, gravity, nw, aos, B, nq, nv, nb, q, v, tau, fext, vdot, qdot);
}
