// walk_emu.hip — CPU emulation of aba_walk_kernel (rigidbodydynamics.jl_amd/csrc/rbd_walk.hpp) — TEST INFRASTRUCTURE ONLY.
// The kernel's step functions are __host__ __device__; this harness runs them lane by lane with the LDS replaced by a host buffer
// (pre-filled with a NaN pattern) on the plan records exported by rbd_model_track_plan.  Between two workgroup barriers of the
// kernel the wavefronts are run ONE AFTER THE OTHER to the next barrier — in ascending or descending order (`reverse`) — so a mailbox
// read that the kernel's barriers do not order after its write meets the NaN pattern in one of the two orders.
// Built on demand by tests/test_walk_emu.py with `hipcc --cuda-host-only`; never linked into librbd_hip.so.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rbd_walk.hpp"

using namespace rbd;

template <typename T> static void set_lane(T& x, int j, typename Lanes<T>::S v) { if constexpr (Lanes<T>::N == 1) x = v; else x[j] = v; }
template <typename T, bool FLT, bool GEN, bool RNEA>
static int emu_run(const WalkModel& M, int reverse, long B, const typename Lanes<T>::S* q, const typename Lanes<T>::S* v, const typename Lanes<T>::S* tau,
                   const typename Lanes<T>::S* fext, typename Lanes<T>::S* vdot, typename Lanes<T>::S* qdot, Layout Lq, Layout Lv, Layout Lf) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N, SPW = 64 * N;
  const size_t lds_bytes = walk_lds_bytes(M.ns, M.G, M.nq, M.nv, M.nA, M.nB, M.nS, sizeof(T), sizeof(S));
  std::vector<char> lds(lds_bytes + 64);
  void* base = (void*)(((uintptr_t)lds.data() + 15) & ~(uintptr_t)15);
  const int ns = M.ns, G = M.G;
  const size_t nrec = (size_t)ns * G;
  for (long group = 0; group * SPW < B; ++group) {
    memset(base, 0xff, lds_bytes);
    WalkCtx<T> c;
    c.M = M;
    walk_ctx_lds(c, base);
    if (RNEA) walk_twist_rows_rnea(c);
    memcpy(const_cast<I4*>(c.tri), M.ri, nrec * 16);
    memcpy(const_cast<S*>(c.trr), M.rr, nrec * TR_STRIDE * sizeof(S));
    memcpy(const_cast<int32_t*>(c.twk), M.wk, nrec * 4);
    for (int t = 0; t < 64; ++t) walk_stage_chain(c, t, 64);
    auto state_of = [&](int l) { const long st = group * SPW + l; return st < B ? st : B - 1; };  // l: state of the workgroup, 0 <= l < 64 N
    for (int l = 0; l < SPW; ++l) {
      const long st = state_of(l);
      for (int k = 0; k < M.nq; ++k) *walk_cell(c.rows, c.rq + k, l) = q[k * Lq.sk + st * Lq.sb];
      for (int k = 0; k < M.nv; ++k) *walk_cell(c.rows, c.rv + k, l) = v ? v[k * Lv.sk + st * Lv.sb] : S(0);
      for (int k = 0; k < M.nv; ++k) *walk_cell(c.rows, c.rt + k, l) = tau ? tau[k * Lv.sk + st * Lv.sb] : S(0);
    }
    std::vector<WalkRegs<T>> W((size_t)G * 64);
    std::vector<WalkStash<T>> St((size_t)G * 64);
    for (auto& w : W) { memset(&w, 0xff, sizeof w); walk_init(w); }
    for (auto& s : St) memset(&s, 0xff, sizeof s);
    auto each_wave = [&](auto&& f) {
      for (int i = 0; i < G; ++i) f(reverse ? G - 1 - i : i);
    };
    // pass A: segments end after a step whose SF_AW flag is set
    for (int s0 = 0; s0 < ns;) {
      int s1 = s0;
      while (s1 < ns - 1 && !((M.sfm[1] >> s1) & 1)) ++s1;
      each_wave([&](int g) {
        for (int s = s0; s <= s1; ++s)
          for (int l = 0; l < 64; ++l) { T rr[TR_STRIDE]; walk_consts<T, TR_STRIDE>(c, s, g, rr); walk_step_a<T, FLT, GEN, RNEA, FLT && !RNEA>(c, W[g * 64 + l], St[g * 64 + l], s, walk_rec(walk_raw(c, s, g)), rr, l, qdot != nullptr); }
      });
      s0 = s1 + 1;
    }
    for (auto& w : W) walk_init_b(w);
    // pass B (behind a workgroup barrier in the kernels): the B mailboxes take over the rows that held the twist halves of the A mailboxes
    // — poisoned here, nothing after pass A may read them
    memset(walk_row(c, c.rB, 0), 0xff, (walk_rows(M.nq, M.nv, M.nA, M.nB, M.nS) - c.rB) * WR_STRIDE * sizeof(T));
    for (int s0 = ns - 1; s0 >= 0;) {
      int s1 = s0;
      while (s1 > 0 && !((M.sfm[3] >> s1) & 1)) --s1;
      each_wave([&](int g) {
        for (int s = s0; s >= s1; --s)
          for (int l = 0; l < 64; ++l) {
            T fe[6];
            for (int k = 0; k < 6; ++k) fe[k] = T(0);
            if (fext) {
              const int o6 = c.tri[s * G + g].y & 0xffff;
              for (int k = 0; k < 6; ++k)
                for (int j = 0; j < N; ++j) set_lane(fe[k], j, fext[(o6 + k) * Lf.sk + state_of(l + 64 * j) * Lf.sb]);
            }
            T rr[TR_STRIDE]; walk_consts<T, TR_STRIDE>(c, s, g, rr);
            if (FLT && !RNEA) { const WalkRec rc = walk_rec(walk_raw(c, s, g)); if ((rc.rrf & BFD_FCARRY) && (rc.flags & TF_VALID)) { if (rc.park >= 0) walk_get_park(c, rc.park, l, W[g * 64 + l]); walk_fcarry_b(c, W[g * 64 + l], l, fe); } }
            if (RNEA) walk_step_rb<T, FLT, GEN>(c, W[g * 64 + l], St[g * 64 + l], s, walk_rec(walk_raw(c, s, g)), rr, l, fe);
            else walk_step_b<T, FLT, GEN, FLT>(c, W[g * 64 + l], St[g * 64 + l], s, walk_rec(walk_raw(c, s, g)), rr, l, fe);
          }
      });
      s0 = s1 - 1;
    }
    for (auto& w : W) walk_init_c(w);
    for (int s0 = 0; !RNEA && s0 < ns;) {
      int s1 = s0;
      while (s1 < ns - 1 && !((M.sfm[1] >> s1) & 1)) ++s1;
      each_wave([&](int g) {
        for (int s = s0; s <= s1; ++s)
          for (int l = 0; l < 64; ++l) { T rr[TR_STRIDE]; walk_consts<T, TR_STRIDE>(c, s, g, rr); walk_step_c<T, FLT, GEN, FLT>(c, W[g * 64 + l], St[g * 64 + l], s, walk_rec(walk_raw(c, s, g)), rr, l, qdot != nullptr); }
      });
      s0 = s1 + 1;
    }
    for (int l = 0; l < SPW; ++l) {
      const long st = group * SPW + l;
      if (st >= B) continue;
      for (int k = 0; k < M.nv; ++k) vdot[k * Lv.sk + st * Lv.sb] = *walk_cell(c.rows, c.rt + k, l);
    }
    if (qdot) {
      for (int l = 0; l < SPW; ++l) {
        const long st = group * SPW + l;
        if (st >= B) continue;
        for (int k = 0; k < M.nq; ++k) qdot[k * Lq.sk + st * Lq.sb] = *walk_cell(c.rows, c.rq + k, l);
      }
    }
  }
  return 0;
}

template <typename T>  // T: double, float, or f2 (two fp32 states per lane)
static int emu_t(const int32_t* dims, const int32_t* ri, const double* rr, const double* gravity, int rnea, int reverse, int aos, long B, int nq, int nv, int nb, const void* q,
                 const void* v, const void* tau, const void* fext, void* vdot, void* qdot, int32_t* info, const int32_t* wk_rr, const int32_t* chain_i,
                 const double* chain_r, const double* fxp) {
  WalkModel M;
  memset(&M, 0, sizeof M);
  M.G = dims[0]; M.ns = dims[1]; M.nA = dims[2]; M.nB = dims[3]; M.nq = nq; M.nv = nv;
  const size_t nrec = (size_t)M.ns * M.G;
  std::vector<int32_t> riv(ri, ri + nrec * TI_STRIDE);
  WalkPlan P = build_walk_plan(M.ns, M.G, riv);
  if (!P.ok) return 2;
  using S0 = typename Lanes<T>::S;
  std::vector<S0> crt;
  if (wk_rr) {  // the plan of the tree re-rooted at its centre (rbd_model_reroot_plan): parking words with the BFD_* flags, chain table
    P.wk.assign(wk_rr, wk_rr + nrec);
    M.reroot.nchain = dims[7]; M.reroot.fq = dims[8]; M.reroot.fv = dims[9];
    crt.assign(chain_r, chain_r + 15 * (size_t)dims[7]);
    M.reroot.chain_i = chain_i; M.reroot.chain_r = crt.data();
    memcpy(M.reroot.fXp, fxp, sizeof M.reroot.fXp);
  }
  M.nS = P.nS;
  using S = typename Lanes<T>::S;
  if (info) { info[0] = P.nS; info[1] = (int32_t)walk_lds_bytes(M.ns, M.G, nq, nv, M.nA, M.nB, M.nS, sizeof(T), sizeof(S)); }
  std::vector<S> rrt(rr, rr + nrec * TR_STRIDE);
  M.ri = riv.data(); M.rr = rrt.data(); M.wk = P.wk.data();
  const int32_t* sf = ri + nrec * TI_STRIDE;  // the per-step flags follow the packed records
  for (int k = 0; k < 5; ++k) { M.sfm[k] = 0; for (int s = 0; s < M.ns; ++s) M.sfm[k] |= (uint64_t)((sf[s] >> k) & 1) << s; }
  memcpy(M.gravity, gravity, sizeof M.gravity);
  auto lay = [&](long n) { Layout L; if (aos) { L.sk = 1; L.sb = n; } else { L.sk = B; L.sb = 1; } return L; };
  const Layout Lq = lay(nq), Lv = lay(nv), Lf = lay(6L * nb);
  const int flt = dims[4], gen = dims[5];
#define RUN(F, GN) (rnea ? emu_run<T, F, GN, true>(M, reverse, B, (const S*)q, (const S*)v, (const S*)tau, (const S*)fext, (S*)vdot, (S*)qdot, Lq, Lv, Lf) : emu_run<T, F, GN, false>(M, reverse, B, (const S*)q, (const S*)v, (const S*)tau, (const S*)fext, (S*)vdot, (S*)qdot, Lq, Lv, Lf))
  if (flt) return gen ? RUN(true, true) : RUN(true, false);
  return gen ? RUN(false, true) : RUN(false, false);
#undef RUN
}

// rnea = 0: dynamics! (x = τ in, y = v̇ out);  rnea = 1: inverse_dynamics! / dynamics_bias! (x = v̇ in or NULL, y = τ out)
extern "C" int walk_emu_dynamics(const int32_t* dims, const int32_t* ri, const double* rr, const double* gravity, int f32, int reverse, int aos, long B, int nq,
                                 int nv, int nb, const void* q, const void* v, const void* tau, const void* fext, void* vdot, void* qdot, int32_t* info, int rnea,
                                 const int32_t* wk_rr, const int32_t* chain_i, const double* chain_r, const double* fxp) {
  if (f32 == 2) return emu_t<f2>(dims, ri, rr, gravity, rnea, reverse, aos, B, nq, nv, nb, q, v, tau, fext, vdot, qdot, info, wk_rr, chain_i, chain_r, fxp);  // two fp32 states per lane
  return f32 ? emu_t<float>(dims, ri, rr, gravity, rnea, reverse, aos, B, nq, nv, nb, q, v, tau, fext, vdot, qdot, info, wk_rr, chain_i, chain_r, fxp)
             : emu_t<double>(dims, ri, rr, gravity, rnea, reverse, aos, B, nq, nv, nb, q, v, tau, fext, vdot, qdot, info, wk_rr, chain_i, chain_r, fxp);
}
