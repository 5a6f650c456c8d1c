// pipe_emu.hip — CPU emulation of aba_pipe_kernel (rigidbodydynamics.jl_amd/csrc/rbd_pipe.hpp) — TEST INFRASTRUCTURE ONLY.
// The kernel's stage functions are __host__ __device__; this harness runs them lane by lane with the LDS replaced by a host buffer
// (pre-filled with a NaN pattern) on the plan records exported by rbd_model_track_plan.  Between two workgroup barriers of the kernel
// the four wavefronts (roles K, I, T, S) are run ONE AFTER THE OTHER — in ascending or descending order (`reverse`) — so a ring slot
// that is read in the iteration it is written in, or overwritten before its reader came, shows in one of the two orders.
// Built on demand by tests/test_pipe_emu.py with `hipcc --cuda-host-only`; never linked into librbd_hip.so.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rbd_pipe.hpp"

using namespace rbd;

template <typename T>
static int emu_run(const WalkModel& M, int reverse, long B, const T* q, const T* v, const T* tau, const T* fext, T* vdot, T* qdot, Layout Lq, Layout Lv, Layout Lf) {
  const size_t lds_bytes = pipe_lds_bytes(M.ns, M.nq, M.nv, M.nA, M.nB, M.nS, sizeof(T));
  std::vector<char> lds(lds_bytes + 64);
  void* base = (void*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
  const int ns = M.ns;
  const size_t nrec = (size_t)ns * 4;
  for (long group = 0; group * PIPE_STATES < B; ++group) {
    memset(base, 0xff, lds_bytes);
    PipeCtx<T> c;
    c.M = M;
    pipe_ctx_lds(c, base);
    if ((char*)(c.parkT + (size_t)M.nS * 12 * PIPE_STATES) != (char*)base + lds_bytes) return 3;  // the two layout functions disagree
    memcpy(const_cast<T*>(c.trr), M.rr, nrec * TR_STRIDE * sizeof(T));
    memcpy(const_cast<int32_t*>(c.rec), M.wk, nrec * WREC_STRIDE * 4);
    auto state_of = [&](int st) { const long s = group * PIPE_STATES + st; return s < B ? s : B - 1; };
    for (int st = 0; st < PIPE_STATES; ++st) {
      const long gs = state_of(st);
      for (int k = 0; k < M.nq; ++k) *pipe_row(c, c.rq + k, st) = q[k * Lq.sk + gs * Lq.sb];
      for (int k = 0; k < M.nv; ++k) *pipe_row(c, c.rv + k, st) = v ? v[k * Lv.sk + gs * Lv.sb] : T(0);
      for (int k = 0; k < M.nv; ++k) *pipe_row(c, c.rt + k, st) = tau ? tau[k * Lv.sk + gs * Lv.sb] : T(0);
    }
    std::vector<PipeK<T>> WK(64);
    std::vector<PipeT<T>> WT(64);
    std::vector<PipeS<T>> WS(64);
    std::vector<WalkStash<T>> StK(64), StS(64);  // the stash of a wavefront is its own registers
    for (int l = 0; l < 64; ++l) {
      memset(&WK[l], 0xff, sizeof WK[l]); memset(&WT[l], 0xff, sizeof WT[l]); memset(&WS[l], 0xff, sizeof WS[l]);
      memset(&StK[l], 0xff, sizeof StK[l]); memset(&StS[l], 0xff, sizeof StS[l]);
      pipe_identity(WK[l]);
      for (int k = 0; k < 6; ++k) { WT[l].Tw[k] = T(0); WT[l].av[k] = T(0); WS[l].cP[k] = T(0); WS[l].ad[k] = T(0); }
      for (int k = 0; k < 21; ++k) WS[l].cI[k] = T(0);
    }
    auto each_role = [&](auto&& f) {
      for (int i = 0; i < 4; ++i) f(reverse ? 3 - i : i);
    };
    for (int i = 0; i <= ns; ++i)
      each_role([&](int role) {
        for (int l = 0; l < 64; ++l) {
          if (role == PIPE_K) { if (i < ns) pipe_a_k(c, WK[l], StK[l], i, i & 1, pipe_rec(c, i, l & 3), l); }
          else if (role == PIPE_I) { if (i + 1 < ns) pipe_a_i(c, i + 1, pipe_rec(c, i + 1, l & 3), l); }
          else if (role == PIPE_T) { if (i >= 1) pipe_a_t(c, WT[l], i - 1, (i - 1) & 1, pipe_rec(c, i - 1, l & 3), l, qdot != nullptr); }
        }
      });
    for (int j = 0; j < ns + 3; ++j) {
      const int sk = ns - 1 - j;
      each_role([&](int role) {
        for (int l = 0; l < 64; ++l) {
          if (role == PIPE_K) { if (sk >= 0) pipe_b_k(c, WK[l], StK[l], sk, j % 3, pipe_rec(c, sk, l & 3), l); }
          else if (role == PIPE_I) { if (sk + 1 >= 0 && sk + 1 < ns) pipe_b_i(c, sk + 1, (j + 2) % 3, j & 1, pipe_rec(c, sk + 1, l & 3), l); }
          else if (role == PIPE_T) {
            const int s = sk + 2;
            if (s >= 0 && s < ns) {
              T fe[6];
              for (int k = 0; k < 6; ++k) fe[k] = T(0);
              if (fext) {
                const int o6 = c.rec[(s * 4 + (l & 3)) * WREC_STRIDE + WREC_ORIG6];
                for (int k = 0; k < 6; ++k) fe[k] = fext[(o6 + k) * Lf.sk + state_of(l >> 2) * Lf.sb];
              }
              pipe_b_t(c, WT[l], s, (j + 1) % 3, (j + 1) & 1, j & 1, pipe_rec(c, s, l & 3), l, fe);
            }
          } else {
            const int s = sk + 3;
            if (s >= 0 && s < ns) pipe_b_s(c, WS[l], StS[l], s, (j + 1) & 1, pipe_rec(c, s, l & 3), l);
          }
        }
      });
    }
    for (int s2 = 0; s2 < ns; ++s2)  // pass C: the S wavefront alone, no barriers
      for (int l = 0; l < 64; ++l) pipe_c_s(c, WS[l], StS[l], s2, pipe_rec(c, s2, l & 3), l);
    for (int st = 0; st < PIPE_STATES; ++st) {
      const long gs = group * PIPE_STATES + st;
      if (gs >= B) continue;
      for (int k = 0; k < M.nv; ++k) vdot[k * Lv.sk + gs * Lv.sb] = *pipe_row(c, c.rt + k, st);
      if (qdot)
        for (int k = 0; k < M.nq; ++k) qdot[k * Lq.sk + gs * Lq.sb] = *pipe_row(c, c.rq + k, st);
    }
  }
  return 0;
}

template <typename T>
static int emu_t(const int32_t* dims, const int32_t* ri, const double* rr, const double* gravity, int reverse, int aos, long B, int nq, int nv, int nb, const void* q,
                 const void* v, const void* tau, const void* fext, void* vdot, void* qdot, int32_t* info) {
  WalkModel M;
  memset(&M, 0, sizeof M);
  const int G = dims[0];
  M.G = 4; M.ns = dims[1]; M.nA = dims[2]; M.nB = dims[3]; M.nq = nq; M.nv = nv;
  if (dims[5]) return 4;  // prismatic / fixed / sin-cos joints: not this mapping
  const size_t nrec = (size_t)M.ns * G;
  std::vector<int32_t> riv(ri, ri + nrec * TI_STRIDE);
  const WalkPlan P = build_walk_plan(M.ns, G, riv);
  if (!P.ok) return 2;
  M.nS = P.nS;
  const std::vector<int32_t> rec = walk_unpack4(M.ns, G, riv, P.wk);
  const std::vector<T> rrt = walk_consts4<T>(M.ns, G, std::vector<double>(rr, rr + nrec * TR_STRIDE));
  M.rr = rrt.data(); M.wk = rec.data();
  memcpy(M.gravity, gravity, sizeof M.gravity);
  if (info) { info[0] = P.nS; info[1] = (int32_t)pipe_lds_bytes(M.ns, nq, nv, M.nA, M.nB, M.nS, sizeof(T)); }
  auto lay = [&](long n) { Layout L; if (aos) { L.sk = 1; L.sb = n; } else { L.sk = B; L.sb = 1; } return L; };
  return emu_run<T>(M, reverse, B, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, (T*)vdot, (T*)qdot, lay(nq), lay(nv), lay(6L * nb));
}

extern "C" int pipe_emu_dynamics(const int32_t* dims, const int32_t* ri, const double* rr, const double* gravity, int f32, int reverse, int aos, long B, int nq,
                                 int nv, int nb, const void* q, const void* v, const void* tau, const void* fext, void* vdot, void* qdot, int32_t* info) {
  return f32 ? emu_t<float>(dims, ri, rr, gravity, reverse, aos, B, nq, nv, nb, q, v, tau, fext, vdot, qdot, info)
             : emu_t<double>(dims, ri, rr, gravity, reverse, aos, B, nq, nv, nb, q, v, tau, fext, vdot, qdot, info);
}
