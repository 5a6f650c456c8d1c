// rbd_bank.hpp — "banked" lane-per-body kernels: every lane carries TWO bodies (translation unit: rbd_bank_kernels.hip).
//
// aba_kernel keeps one body per lane, so a state of n bodies needs next_pow2(n) lanes and each level-synchronous sweep
// step is issued for all of them although only the bodies of one level do useful work.  The sweeps are issue-bound
// (DESIGN.md §3.3), so the cost of a sweep step is per WAVEFRONT, not per body: packing twice as many states into a
// wavefront halves the instruction issue of the sweeps.  Here the bodies are split by level into two banks — bank 0 =
// levels [0, L0), bank 1 = levels [L0, nlevels) — and lane j of a state carries body j of bank 0 AND body j of bank 1, each
// with its own register set (Atlas: 15 + 16 bodies on 16 lanes, 4 states per wavefront instead of 2).  A sweep step at
// level l touches only the register set of the bank that owns l.  Within a bank the bodies are in DFS pre-order, so the
// first-child hop is a DPP wave shift.
//
// Round 3 (this file was rebuilt):
//  * what a lane TAKES from a neighbour is added under the lane's own exec mask (`if (takes) x += h`, kept a branch by
//    RBD_KEEP_BRANCH) instead of being selected against zero: the states of a wavefront stay independent (a NaN state cannot reach
//    its wave-mate, round 2) without the two v_cndmask per fp64 value the selects cost (+713 VALU per wavefront, the round-2 regression);
//  * every hop that is not "first child <-> previous lane" — second / third children, the hop between the banks — goes through an
//    LDS exchange column per lane as 16-byte pairs (ds_write_b128 / ds_read_b128: 14 + 14 instructions for the 27 values of a hand-off)
//    instead of two ds_bpermute_b32 + two selects per value; the hop between the banks of the FK sweep reads the parent's PARKED values;
//  * the bank that is not being swept parks its live values in LDS as pairs too (half the LDS instructions);
//  * SIMPLE instantiation for mechanisms whose tree joints are all revolute apart from 6-dof joints on the world (Atlas, Valkyrie,
//    most arms): no joint-type interpretation per lane, motion subspace = (R a; p x R a), sin/cos by sincos_fast;
//  * the integrator-fused form is its own instantiation (FUSED): the plain dynamics! launch keeps __restrict__ on q / v and carries
//    none of the stage code; the re-rooted variant of round 2 (no gain at any size) is gone.
// Scope: 1-dof and fixed tree joints, 6-dof joints on the world (everything else stays with aba_kernel).
#pragma once

// experiment switches (scripts/build_bank_variant.sh; defaults = the measured best)
#ifndef RBD_BANK_WAVES
#define RBD_BANK_WAVES 2     // wavefronts per SIMD the register budget allows (2: 256 VGPRs)
#endif
#ifndef RBD_BANK_FK_LDS
#define RBD_BANK_FK_LDS 0    // 1: every top-down hop through the exchange columns (no DPP shifts)
#endif
#ifndef RBD_BANK_HANDOFF_LDS
#define RBD_BANK_HANDOFF_LDS 0  // 1: every bottom-up hand-off through the exchange columns
#endif

// Experiment (scripts/build_bank_variant.sh fixed -DRBD_BANK_FIXED_NL=11 -DRBD_BANK_FIXED_L0=5 "-DRBD_BANK_FIXED_NS=0,3,1,1,3,1,1,1,1,1,1"): the level structure of ONE
// mechanism as compile-time constants and the level loops unrolled — what a per-mechanism compilation of this kernel (as rbd_jit.hip does for the large-batch
// kernels) would know: which levels hop by DPP and which through the exchange columns, how many later children a gather visits.  Measured in round 4: see DESIGN.md §3.5.
#ifdef RBD_BANK_FIXED_NL
#define RBD_BANK_UNROLL _Pragma("unroll")
namespace rbd { constexpr int kBankNs[] = {RBD_BANK_FIXED_NS}; }
#define RBD_BK_NL RBD_BANK_FIXED_NL
#define RBD_BK_L0 RBD_BANK_FIXED_L0
#define RBD_BK_NS(l) kBankNs[l]
#ifdef RBD_BANK_FIXED_PERM
#define RBD_BK_PERM(l) (((RBD_BANK_FIXED_PERM) >> (l)) & 1ull)
#else
#define RBD_BK_PERM(l) (kBankNs[l] > 1)
#endif
#else
#define RBD_BANK_UNROLL _Pragma("unroll 1")
#define RBD_BK_NL M.nlevels
#define RBD_BK_L0 M.L0
#define RBD_BK_NS(l) ns_next(nss, M.ns_desc)
#define RBD_BK_PERM(l) ((M.perm_down >> (l)) & 1)
#endif

namespace rbd {

template <typename T> struct alignas(2 * sizeof(T)) Pair2 { T a, b; };

// LDS written by some lanes of the wavefront and read by others: same-wave LDS operations execute in order; this only stops the compiler
RBD_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// dst[k] = src[k] in the lanes of `mask`, the other lanes keep dst — as ONE in-place operation.  The C++ form `if (m) dst = src` leaves a
// phi per value; with two code paths per level in the sweep loops the register allocator reconciled those with two full copies of the
// loop-carried state per level (60 v_mov_b64 in the FK loop).  Here the values are computed for every lane (the instruction issue is per
// wavefront anyway; lanes outside the mask compute on stale operands and drop the result) and committed by exec-masked moves.
// (s_nop: a DPP read of dst may follow; the hazard recognizer does not look inside inline asm.)
#define RBD_MASSIGN6(MOV)                                                                                                              \
  unsigned long long sv;                                                                                                               \
  asm volatile("s_and_saveexec_b64 %[sv], %[mk]\n\t" MOV " %[d0], %[s0]\n\t" MOV " %[d1], %[s1]\n\t" MOV " %[d2], %[s2]\n\t" MOV          \
               " %[d3], %[s3]\n\t" MOV " %[d4], %[s4]\n\t" MOV " %[d5], %[s5]\n\ts_mov_b64 exec, %[sv]\n\ts_nop 0"                         \
               : [sv] "=&s"(sv), [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [d4] "+v"(d[4]), [d5] "+v"(d[5])    \
               : [mk] "s"(mask), [s0] "v"(s[0]), [s1] "v"(s[1]), [s2] "v"(s[2]), [s3] "v"(s[3]), [s4] "v"(s[4]), [s5] "v"(s[5])          \
               : "scc")
RBD_DEV void massign6(unsigned long long mask, double* d, const double* s) { RBD_MASSIGN6("v_mov_b64"); }
RBD_DEV void massign6(unsigned long long mask, float* d, const float* s) { RBD_MASSIGN6("v_mov_b32"); }
template <typename T, int N> RBD_DEV void massign(bool m, T* d, const T* s) {  // N a multiple of 6 (or padded by the caller)
  static_assert(N % 6 == 0, "blocks of 6");
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(m);
#pragma unroll
  for (int k = 0; k < N; k += 6) massign6(mask, d + k, s + k);
}

// LDS layout, in pairs per lane (pair i of thread t at lds[i * 256 + t]):
//   parked kinematics of bank 0: R 9, p 3 (pairs 0..5: stay for the 6-dof joints), Tw 6, vJ 6 (pairs 6..11)
//   parked forward data of bank 1 (after bank 0's Tw, vJ are back in registers): U 6 (6..8), 1/D, u (9), S 6 (10..12), cb 6 (13..15)
//   exchange column: 14 pairs (a hand-off is 21 + 6 values)
enum { PK_KIN = 0, PK_FWD = 6, PARK_PAIRS = 16, PK_XCH = 16, XCH_PAIRS = 14, BANK_LDS_PAIRS = PARK_PAIRS + XCH_PAIRS };

template <typename T, int N> RBD_DEV void lds_put(Pair2<T>* col, int pair0, const T* x) {
#pragma unroll
  for (int i = 0; i < (N + 1) / 2; ++i) {
    Pair2<T> v;
    v.a = x[2 * i];
    v.b = (2 * i + 1 < N) ? x[2 * i + 1] : T(0);
    col[(pair0 + i) * 256] = v;
  }
}
template <typename T, int N> RBD_DEV void lds_get(const Pair2<T>* col, int pair0, T* x) {
#pragma unroll
  for (int i = 0; i < (N + 1) / 2; ++i) {
    const Pair2<T> v = col[(pair0 + i) * 256];
    x[2 * i] = v.a;
    if (2 * i + 1 < N) x[2 * i + 1] = v.b;
  }
}

__host__ __device__ constexpr int sym_row(int e) { return e < 6 ? 0 : e < 11 ? 1 : e < 15 ? 2 : e < 18 ? 3 : e < 20 ? 4 : 5; }
__host__ __device__ constexpr int sym_col(int e) { return e - SI(sym_row(e), sym_row(e)) + sym_row(e); }

template <typename T> struct BankRegs {
  Body<T> b;
  const T* rb;
  int pcol;  // LDS column (thread index in the block) of the lane that carries the parent body
  T K[24];  // kinematics: transform to root R 9, p 3, twist Tw 6, joint twist in the root frame vJ 6 (generic form only; SIMPLE keeps the joint velocity in K[18])
  RBD_DEV T* R() { return K; }
  RBD_DEV T* p() { return K + 9; }
  RBD_DEV T* Tw() { return K + 12; }
  RBD_DEV T* vJ() { return K + 18; }
  T S[6], cb[6], IA[21], pA[6], U[6], Dinv, u;
  T tj[6];  // joint torques (6 for the floating joint)
  T acc[6], vd[6];
};

template <typename T> RBD_DEV void load_bank_body(const BankModel& M, int k, long B, Body<T>& b) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  b.lane = threadIdx.x & 63;
  const long wave = tid >> 6;
  const int lps = M.lps;
  b.sub = b.lane & (lps - 1);
  b.base = b.lane - b.sub;
  b.state = wave * (64 / lps) + (b.lane / lps);
  b.valid = (b.sub < M.nbk[k]) && (b.state < B);
  const int4* ib4 = reinterpret_cast<const int4*>(M.ib[k] + (b.sub < M.nbk[k] ? b.sub : 0) * IB_STRIDE);  // hipMalloc'ed: 16-byte aligned
  const int4 w0 = ib4[0], w1 = ib4[1], w2 = ib4[2], w3 = ib4[3];
  static_assert(IB_PARENT == 0 && IB_JTYPE == 1 && IB_QOFF == 2 && IB_VOFF == 3 && IB_LEVEL == 4 && IB_NCHILD == 5 && IB_ORIG == 6 && IB_CHILD0 == 7 &&
                IB_MAXCHILD == 8 && IB_FLAGS == 15 && IB_STRIDE == 16, "record layout");
  b.parent = w0.x; b.jtype = w0.y; b.qoff = w0.z; b.voff = w0.w;
  b.level = b.valid ? w1.x : -1;
  b.nchild = w1.y; b.orig = w1.z;
  b.child[0] = w1.w; b.child[1] = w2.x; b.child[2] = w2.y; b.child[3] = w2.z; b.child[4] = w2.w; b.child[5] = w3.x; b.child[6] = w3.y; b.child[7] = w3.z;
  b.flags = w3.w;
  b.plane = b.parent >= 0 ? b.base + b.parent : b.lane;  // parent slot (of the parent's bank) in this state's lane group
}

// ---- forward kinematics -------------------------------------------------------------------------------------------------------
// The sweep carries K = (R 9, p 3, Tw 6) per lane [+ vJ 6 in the generic form].  SIMPLE (revolute joints below level 0): ta = (axis, joint
// velocity), the joint twist is S v with S = X (a; 0) = (R a; p x R a); S itself and the bias term are formed after the sweep, for every lane
// at once (terms()).  The new values are computed by EVERY lane and committed in the lanes at level l (massign).
template <typename T, bool SIMPLE>
RBD_DEV void bank_fk_commit(bool mine, BankRegs<T>& c, const T* k18 /* the parent's R 9, p 3, Tw 6 */, const T* XR, const T* Xp, const T* tl) {
  const T* pR = k18;
  const T* pp = k18 + 9;
  const T* pT = k18 + 12;
  T n[24];
  matmul3(pR, XR, n);
  matvec3(pR, Xp, n + 9);
#pragma unroll
  for (int k = 0; k < 3; ++k) n[9 + k] += pp[k];
  if (SIMPLE) {
    T S[6];
    matvec3(n, tl, S);
    cross3(n + 9, S, S + 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) n[12 + k] = pT[k] + S[k] * tl[3];
    massign<T, 18>(mine, c.K, n);
  } else {
    xmotion(n, n + 9, tl, n + 18);
#pragma unroll
    for (int k = 0; k < 6; ++k) n[12 + k] = pT[k] + n[18 + k];
    massign<T, 24>(mine, c.K, n);
  }
}
// level l inside a bank.  On most levels every body is the first child of its parent = the previous lane: (R, p, Tw) arrive by DPP.  On the
// levels where some body is a later child (perm_down) the parents leave theirs in their exchange column and every child reads its parent's.
template <typename T, bool SIMPLE> RBD_DEV void bank_fk_step(const BankModel& M, int l, bool perm, BankRegs<T>& c, Pair2<T>* lds, const T* XR, const T* Xp, const T* tl) {
  T k18[18];
  if (RBD_BANK_FK_LDS || perm) {  // uniform
    if (c.b.level == l - 1 && c.b.nchild >= 1) {
      RBD_KEEP_BRANCH();
      lds_put<T, 18>(lds + threadIdx.x, PK_XCH, c.K);
    }
    wave_lds_sync();
    lds_get<T, 18>(lds + c.pcol, PK_XCH, k18);  // (lanes that are not at level l read their parent's column too: whatever it holds is dropped)
    wave_lds_sync();
  } else {
#pragma unroll
    for (int k = 0; k < 18; ++k) k18[k] = from_prev_lane(c.K[k]);
  }
  bank_fk_commit<T, SIMPLE>(c.b.level == l, c, k18, XR, Xp, tl);
}
// level L0: the parents are bank-0 bodies, whose (R, p, Tw) have just been parked — the children read their parent's parked pairs
template <typename T, bool SIMPLE> RBD_DEV void bank_fk_cross(int L0, BankRegs<T>& c, const Pair2<T>* lds, const T* XR, const T* Xp, const T* tl) {
  T k18[18];
  lds_get<T, 18>(lds + c.pcol, PK_KIN, k18);
  bank_fk_commit<T, SIMPLE>(c.b.level == L0, c, k18, XR, Xp, tl);
}

// U = IA S, D = S'U, u = tau - S'pA for the lanes of `r` at level l (1-dof joints; fixed joints keep U = 0, 1/D = 0)
template <typename T, bool SIMPLE> RBD_DEV void bank_finish_joint(int l, BankRegs<T>& r) {
  if (r.b.level == l && (SIMPLE ? r.b.jtype != RBD_JOINT_QUAT_FLOATING : joint_nv(r.b.jtype) == 1)) {
    sym6_mul(r.IA, r.S, r.U);
    r.u = r.tj[0] - dot6(r.S, r.pA);
    r.Dinv = rcp_nr(dot6(r.S, r.U));
  }
}

// ---- bottom-up hand-off ---------------------------------------------------------------------------------------------------------
// Lanes of `gv` at level l give (Ia, pa) = (IA - U D^-1 U', pA + Ia cb + U D^-1 u); lanes of `tk` at level l-1 add what their children
// give.  MODE 0, inside a bank (gv and tk are the same register set): the first child is the next lane — DPP shift, added under the
// taker's exec mask; when some body of level l-1 has more children (ns > 1) every lane also leaves its hand-off in its exchange column and
// the takers read their later children's columns.  MODE 1, across the banks: every child through the exchange columns.
// The 27 values are formed, moved and consumed in three chunks (10 + 10 + 7) to bound the temporaries.
template <typename T, int MODE> RBD_DEV void bank_handoff(int l, int ns, const BankRegs<T>& gv, BankRegs<T>& tk, Pair2<T>* lds) {
  const bool takes = (tk.b.level == l - 1);
  const bool take0 = takes && tk.b.nchild >= 1;
  const bool xw = MODE == 1 || ns > 1;  // uniform
  Pair2<T>* const mycol = lds + threadIdx.x;
  T W[6];  // U D^-1 (6-dof joints sit at level 0 and never give)
#pragma unroll
  for (int k = 0; k < 6; ++k) W[k] = gv.U[k] * gv.Dinv;
  T Iac[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int c0 = 0; c0 < 20; c0 += 10) {
    T g[10], h[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) {
      const int i = sym_row(c0 + e), j = sym_col(c0 + e);
      g[e] = gv.IA[c0 + e] - W[i] * gv.U[j];
      Iac[i] += g[e] * gv.cb[j];
      if (j > i) Iac[j] += g[e] * gv.cb[i];
      if (MODE == 0) h[e] = from_next_lane(g[e]);
    }
    if (xw) lds_put<T, 10>(mycol, PK_XCH + c0 / 2, g);
    if (MODE == 0) {
      if (take0) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int e = 0; e < 10; ++e) tk.IA[c0 + e] += h[e];
      }
    }
  }
  {
    T g[7], h[7];
    g[0] = gv.IA[20] - W[5] * gv.U[5];
    Iac[5] += g[0] * gv.cb[5];
#pragma unroll
    for (int k = 0; k < 6; ++k) g[1 + k] = gv.pA[k] + Iac[k] + W[k] * gv.u;
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 7; ++e) h[e] = from_next_lane(g[e]);
    }
    if (xw) lds_put<T, 7>(mycol, PK_XCH + 10, g);
    if (MODE == 0) {
      if (take0) {
        RBD_KEEP_BRANCH();
        tk.IA[20] += h[0];
#pragma unroll
        for (int k = 0; k < 6; ++k) tk.pA[k] += h[1 + k];
      }
    }
  }
  if (xw) {
    wave_lds_sync();
#pragma unroll 1
    for (int s = (MODE == 1 ? 0 : 1); s < ns; ++s) {
      if (takes && s < tk.b.nchild) {
        RBD_KEEP_BRANCH();
        const Pair2<T>* src = lds + (threadIdx.x - tk.b.sub + child_sel(tk.b, s));
        T x[28];
        lds_get<T, 28>(src, PK_XCH, x);
#pragma unroll
        for (int e = 0; e < 21; ++e) tk.IA[e] += x[e];
#pragma unroll
        for (int k = 0; k < 6; ++k) tk.pA[k] += x[21 + k];
      }
    }
    wave_lds_sync();
  }
}

// v̇ = D^-1 (u - U'a'), a = a' + S v̇ with a' = a_parent + cb — computed by every lane, committed in the lanes of `mine`
template <typename T> RBD_DEV void bank_joint_accel(bool mine, BankRegs<T>& r, const T* a_parent) {
  T ap[6], n[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ap[k] = a_parent[k] + r.cb[k];
  const T x = r.Dinv * (r.u - dot6(r.U, ap));
#pragma unroll
  for (int k = 0; k < 6; ++k) n[k] = ap[k] + r.S[k] * x;
  massign<T, 6>(mine, r.acc, n);
  r.vd[0] = mine ? x : r.vd[0];
}
// top-down acceleration step at level l inside a bank (same hop rules as bank_fk_step)
template <typename T> RBD_DEV void bank_accel_step(const BankModel& M, int l, bool perm, BankRegs<T>& c, Pair2<T>* lds) {
  T ap[6];
  if (RBD_BANK_FK_LDS || perm) {  // uniform
    if (c.b.level == l - 1 && c.b.nchild >= 1) {
      RBD_KEEP_BRANCH();
      lds_put<T, 6>(lds + threadIdx.x, PK_XCH, c.acc);
    }
    wave_lds_sync();
    lds_get<T, 6>(lds + c.pcol, PK_XCH, ap);
    wave_lds_sync();
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) ap[k] = from_prev_lane(c.acc[k]);
  }
  bank_joint_accel(c.b.level == l, c, ap);
}

#ifdef RBD_PROFILE_PHASES
__device__ long long rbd_bank_phase_clock[16];
#define RBD_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) rbd_bank_phase_clock[i] = clock64(); } while (0)
#else
#define RBD_MARK(i)
#endif

template <typename T, bool R> struct InPtr { typedef const T* type; };
template <typename T> struct InPtr<T, true> { typedef const T* __restrict__ type; };

// FUSED: the launch is also a stage of a Munthe-Kaas RK4 step (`simulate`): q / v alias F.q_state / F.v_state, so no __restrict__ on them.
// SIMPLE: every tree joint is revolute, apart from 6-dof joints on the world (BankModel::simple, set by the host).
// (aba_bank_body: the kernel's code; aba_bank_kernel below wraps it for the build-time instantiations, rbd_jit.hip's program wraps it per mechanism with
// RBD_BANK_FIXED_* defined — the level loops unrolled against the mechanism's own level structure: 19.5 -> 16.7 us at 4096 fp64 Atlas states)
template <typename T, bool FUSED, bool SIMPLE>
RBD_DEV void aba_bank_body(const BankModel& M, long B, typename InPtr<T, !FUSED>::type q, typename InPtr<T, !FUSED>::type v, const T* __restrict__ tau,
                           const T* __restrict__ fext, T* __restrict__ vdot, T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf, const MkFuse& F,
                           Pair2<T>* const lds) {
  Pair2<T>* const col = lds + threadIdx.x;
  RBD_MARK(0);
  BankRegs<T> r0, r1;
  // ---- per-body set-up (once per bank, every lane busy): joint transform and joint twist in the joint frame ----
  // The global loads of BOTH banks are issued up front (two dependent round trips: body record, then q / v / tau), so that
  // bank 1's latency hides behind bank 0's sweep; the arithmetic of a bank runs right before its own sweep.
  auto fetch = [&](int k, BankRegs<T>& c, T* qj, T* vj) {
    load_bank_body(M, k, B, c.b);
    c.rb = reinterpret_cast<const T*>(M.rb[k]) + (c.b.sub < M.nbk[k] ? c.b.sub : 0) * RB_STRIDE;
    c.pcol = (int)threadIdx.x - c.b.sub + (c.b.parent >= 0 ? c.b.parent : c.b.sub);
    if (SIMPLE) {
      // lanes without a body (sub >= nbk, state >= B) load some real body's coordinates of the last state — legal addresses, no branches;
      // their level is -1, so nothing they compute is committed, taken or stored
      const bool fl = c.b.jtype == RBD_JOINT_QUAT_FLOATING;
      const long sc = c.b.state < B ? c.b.state : B - 1;
      const T* qp = q + ((long)c.b.qoff * Lq.sk + sc * Lq.sb);
      const long va = (long)c.b.voff * Lv.sk + sc * Lv.sb;
      const bool hv = v != nullptr, ht = tau != nullptr;  // uniform
      qj[0] = qp[0];
      vj[0] = hv ? v[va] : T(0);
      c.tj[0] = ht ? tau[va] : T(0);
#pragma unroll
      for (int k2 = 1; k2 < 7; ++k2) qj[k2] = T(0);
#pragma unroll
      for (int k2 = 1; k2 < 6; ++k2) { vj[k2] = T(0); c.tj[k2] = T(0); }
      if (fl) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int k2 = 1; k2 < 7; ++k2) qj[k2] = qp[k2 * Lq.sk];
        if (hv) {
#pragma unroll
          for (int k2 = 1; k2 < 6; ++k2) vj[k2] = v[va + k2 * Lv.sk];
        }
        if (ht) {
#pragma unroll
          for (int k2 = 1; k2 < 6; ++k2) c.tj[k2] = tau[va + k2 * Lv.sk];
        }
      }
    } else {
      load_joint_q(c.b, q, Lq, qj);
      load_joint_v(c.b, v, Lv, vj);
      load_joint_v(c.b, tau, Lv, c.tj);
    }
  };
  auto setup = [&](BankRegs<T>& c, T* qj, T* vj, T* XR, T* Xp, T* tl, T* ta /* SIMPLE: axis, joint velocity */) {
    if (SIMPLE) {
      const bool fl = c.b.jtype == RBD_JOINT_QUAT_FLOATING;
      const T ax[3] = {c.rb[RB_AXIS], c.rb[RB_AXIS + 1], c.rb[RB_AXIS + 2]};
      T Rj[9], pj[3] = {T(0), T(0), T(0)};
      T sn, cs;
      ta[0] = ax[0]; ta[1] = ax[1]; ta[2] = ax[2]; ta[3] = vj[0];
      sincos_fast(qj[0], &sn, &cs);
      rot_axis_sc(ax, sn, cs, Rj);
#pragma unroll
      for (int k = 0; k < 3; ++k) { tl[k] = ax[k] * vj[0]; tl[3 + k] = T(0); }
      if (fl) {  // level-0 lanes only (quaternion_floating.jl:81-83, :126-136, :182-188)
        RBD_KEEP_BRANCH();
        rot_quat(qj[0], qj[1], qj[2], qj[3], Rj);
        pj[0] = qj[4]; pj[1] = qj[5]; pj[2] = qj[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) tl[k] = vj[k];
        if (qdot != nullptr && c.b.valid) {
          const T w = qj[0], x = qj[1], y = qj[2], z = qj[3];
          T o[7];
          o[0] = (-x * vj[0] - y * vj[1] - z * vj[2]) * T(0.5);
          o[1] = (w * vj[0] - z * vj[1] + y * vj[2]) * T(0.5);
          o[2] = (z * vj[0] + w * vj[1] - x * vj[2]) * T(0.5);
          o[3] = (-y * vj[0] + x * vj[1] + w * vj[2]) * T(0.5);
          matvec3(Rj, vj + 3, o + 4);
#pragma unroll
          for (int k = 0; k < 7; ++k) qdot[(long)(c.b.qoff + k) * Lq.sk + c.b.state * Lq.sb] = o[k];
        }
      } else if (qdot != nullptr && c.b.valid) {
        qdot[(long)c.b.qoff * Lq.sk + c.b.state * Lq.sb] = vj[0];
      }
      T XpR[9], Xpp[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) XpR[k] = c.rb[RB_XPR + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Xpp[k] = c.rb[RB_XPP + k];
      matmul3(XpR, Rj, XR);
      matvec3(XpR, pj, Xp);
#pragma unroll
      for (int k = 0; k < 3; ++k) Xp[k] += Xpp[k];
    } else {
      store_qdot(c.b, qdot, Lq, qj, vj);
      local_transform(c.b, c.rb, qj, XR, Xp);
      local_joint_motion(c.b, c.rb, vj, tl);
    }
    // as if at level 0 (transform to root = local transform, twist = joint twist); deeper lanes overwrite at their level
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R()[i] = XR[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.p()[i] = Xp[i];
    xmotion(c.R(), c.p(), tl, c.Tw());
    if (SIMPLE) {
      c.K[18] = ta[3];  // the joint velocity, for the bias term
      c.K[19] = T(0);
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) c.vJ()[i] = c.Tw()[i];
    }
  };
  // ---- per-body terms in the root frame: motion subspace, bias acceleration, inertia, bias force ----
  auto terms = [&](BankRegs<T>& c, bool accumulate) {  // accumulate: IA, pA already hold what the children handed up
    const bool floating = c.b.jtype == RBD_JOINT_QUAT_FLOATING;
    if (SIMPLE) {  // S = X (a; 0) = (R a; p x R a); bias term [T, vJ] = [T, S] v
      const T ax[3] = {c.rb[RB_AXIS], c.rb[RB_AXIS + 1], c.rb[RB_AXIS + 2]};
      matvec3(c.R(), ax, c.S);
      cross3(c.p(), c.S, c.S + 3);
    } else {
      T e1[6] = {T(1), T(0), T(0), T(0), T(0), T(0)}, sl[6];
      local_joint_motion(c.b, c.rb, e1, sl);
      xmotion(c.R(), c.p(), sl, c.S);
    }
    if (floating || (!SIMPLE && joint_nv(c.b.jtype) == 0)) {  // (a 6-dof joint on the world: vJ = T, so [T, vJ] = 0 either way)
#pragma unroll
      for (int i = 0; i < 6; ++i) c.S[i] = T(0);
    }
    if (SIMPLE) {
      se3_comm(c.Tw(), c.S, c.cb);
#pragma unroll
      for (int i = 0; i < 6; ++i) c.cb[i] *= c.K[18];
    } else {
      se3_comm(c.Tw(), c.vJ(), c.cb);
    }
    RInertia<T> I;
    T Jb[6], mc[3], fe[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Jb[i] = c.rb[RB_J + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) mc[i] = c.rb[RB_MC + i];
    inertia_to_root(Jb, mc, c.rb[RB_M], c.R(), c.p(), I);
    T Io[21], po[6];
    sym6_from_inertia(I, Io);
    momentum_cross(I, c.Tw(), po);
    load_body_wrench(c.b, fext, Lf, fe);
    // (lanes without a body are never read by anyone: takers add only what their own children give, under their exec mask)
#pragma unroll
    for (int i = 0; i < 21; ++i) c.IA[i] = accumulate ? c.IA[i] + Io[i] : Io[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) c.pA[i] = accumulate ? c.pA[i] + (po[i] - fe[i]) : po[i] - fe[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) c.U[i] = T(0);
    c.Dinv = T(0);
    c.u = T(0);
    if (floating) xforce(c.R(), c.p(), c.tj, c.U);  // 6-dof joints (level 0): U carries S^-T tau; R, p stay parked for the top-down sweep
  };

  // ---- top-down: transforms to root and twists (update_transforms!, update_twists_wrt_world!) ----
  T qj0[7], vj0[6], qj1[7], vj1[6];
  fetch(0, r0, qj0, vj0);
  fetch(1, r1, qj1, vj1);
  if (FUSED) {
    // `simulate` fusion: this launch is stage F.stage of a Munthe-Kaas RK4 step (see aba_kernel).  Both banks' stage bookkeeping
    // runs here, back to back, so that their loads of the integrator buffers share one round trip and bank 1's hide behind the
    // SE(3) log/exp of bank 0's floating joint.
    if (F.close_prev) {
      mk_stage_lane(r0.b, 4, (T)F.dt, qj0, vj0, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
      mk_stage_lane(r1.b, 4, (T)F.dt, qj1, vj1, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
    }
    mk_stage_lane(r0.b, F.stage, (T)F.dt, qj0, vj0, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
    mk_stage_lane(r1.b, F.stage, (T)F.dt, qj1, vj1, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
    pd_control_lane(r0.b, F, Lq, qj0, vj0, r0.tj);
    pd_control_lane(r1.b, F, Lq, qj1, vj1, r1.tj);
  }
  {
    T XR[9], Xp[3], tl[6], ta[4];
    setup(r0, qj0, vj0, XR, Xp, tl, ta);
    RBD_MARK(1);
RBD_BANK_UNROLL
    for (int l = 1; l < RBD_BK_L0; ++l) bank_fk_step<T, SIMPLE>(M, l, RBD_BK_PERM(l), r0, lds, XR, Xp, SIMPLE ? ta : tl);
  }
  {
    T XR[9], Xp[3], tl[6], ta[4];
    RBD_MARK(2);
    setup(r1, qj1, vj1, XR, Xp, tl, ta);
    RBD_MARK(3);
    // bank 0 rests until bank 1 has been swept bottom-up; its parked (R, p, Tw) are what bank 1's first level reads
    lds_put<T, (SIMPLE ? 20 : 24)>(col, PK_KIN, r0.K);
    wave_lds_sync();
    bank_fk_cross<T, SIMPLE>(RBD_BK_L0, r1, lds, XR, Xp, SIMPLE ? ta : tl);
RBD_BANK_UNROLL
    for (int l = RBD_BK_L0 + 1; l < RBD_BK_NL; ++l) bank_fk_step<T, SIMPLE>(M, l, RBD_BK_PERM(l), r1, lds, XR, Xp, SIMPLE ? ta : tl);
  }
  RBD_MARK(4);
  terms(r1, false);
  RBD_MARK(5);

  // ---- bottom-up: articulated-body inertias and bias forces ----
  NsStream nss = ns_begin(M.ns_desc);
  (void)nss;
RBD_BANK_UNROLL
  for (int l = RBD_BK_NL - 1; l > RBD_BK_L0; --l) {
    bank_finish_joint<T, SIMPLE>(l, r1);
    bank_handoff<T, RBD_BANK_HANDOFF_LDS>(l, RBD_BK_NS(l), r1, r1, lds);
  }
  bank_finish_joint<T, SIMPLE>(RBD_BK_L0, r1);
  RBD_MARK(6);
  // across the banks: the children's hand-off lands in bank 0's (still empty) accumulators; bank 1 then keeps only what the
  // top-down sweep needs, parked while bank 0 is swept; only then does bank 0 wake up and add its own inertia and bias force
#pragma unroll
  for (int i = 0; i < 21; ++i) r0.IA[i] = T(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) r0.pA[i] = T(0);
  bank_handoff<T, 1>(RBD_BK_L0, RBD_BK_NS(RBD_BK_L0), r1, r0, lds);
  {
    lds_get<T, (SIMPLE ? 20 : 24)>(col, PK_KIN, r0.K);
    // (bank 0's Tw, vJ pairs are free again: bank 1's set takes them; R, p stay for the 6-dof joints)
    T f20[20];
#pragma unroll
    for (int i = 0; i < 6; ++i) { f20[i] = r1.U[i]; f20[8 + i] = r1.S[i]; f20[14 + i] = r1.cb[i]; }
    f20[6] = r1.Dinv;
    f20[7] = r1.u;
    lds_put<T, 20>(col, PK_FWD, f20);
  }
  RBD_MARK(7);
  terms(r0, true);
  RBD_MARK(8);
RBD_BANK_UNROLL
  for (int l = RBD_BK_L0 - 1; l >= 1; --l) {
    bank_finish_joint<T, SIMPLE>(l, r0);
    bank_handoff<T, RBD_BANK_HANDOFF_LDS>(l, RBD_BK_NS(l), r0, r0, lds);
  }
  bank_finish_joint<T, SIMPLE>(0, r0);
  RBD_MARK(9);

  // ---- top-down: accelerations and v̇ ----
#pragma unroll
  for (int i = 0; i < 6; ++i) { r0.acc[i] = T(0); r0.vd[i] = T(0); }
  {
    const T a0[6] = {T(0), T(0), T(0), T(-M.gravity[0]), T(-M.gravity[1]), T(-M.gravity[2])};  // a_world = -gravity
    BankRegs<T>& c = r0;
    const bool fl0 = c.b.level == 0 && c.b.jtype == RBD_JOINT_QUAT_FLOATING;
    T fa[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, fv[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    if (fl0) {
      // IA a = S^-T tau - pA;  v̇ = S^-1 (a - a_world)   ([T, vJ] = 0 on the world)
      RBD_KEEP_BRANCH();
      T rhs[6], d[6], k12[12];
#pragma unroll
      for (int i = 0; i < 6; ++i) rhs[i] = c.U[i] - c.pA[i];
      sym6_solve(c.IA, rhs, fa);
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = fa[i] - a0[i];
      lds_get<T, 12>(col, PK_KIN, k12);
      xmotion_inv(k12, k12 + 9, d, fv);
    }
    massign<T, 6>(fl0, c.acc, fa);
    massign<T, 6>(fl0, c.vd, fv);
    bank_joint_accel(c.b.level == 0 && !fl0, c, a0);
  }
RBD_BANK_UNROLL
  for (int l = 1; l < RBD_BK_L0; ++l) bank_accel_step<T>(M, l, RBD_BK_PERM(l), r0, lds);
  RBD_MARK(10);
  if (vdot) store_joint_v(r0.b, vdot, Lv, r0.vd);
  if (FUSED) store_joint_v(r0.b, (T*)F.W.vd[F.stage], Lv, r0.vd);
  {
    T f20[20];
    lds_get<T, 20>(col, PK_FWD, f20);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      r1.U[i] = f20[i];
      r1.S[i] = f20[8 + i];
      r1.cb[i] = f20[14 + i];
      r1.acc[i] = T(0);
      r1.vd[i] = T(0);
    }
    r1.Dinv = f20[6];
    r1.u = f20[7];
  }
  {  // across the banks: every bank-0 lane leaves its acceleration in its exchange column, the bodies of level L0 read their parent's
    lds_put<T, 6>(col, PK_XCH, r0.acc);
    wave_lds_sync();
    T ap[6];
    lds_get<T, 6>(lds + r1.pcol, PK_XCH, ap);
    wave_lds_sync();
    bank_joint_accel(r1.b.level == RBD_BK_L0, r1, ap);
  }
RBD_BANK_UNROLL
  for (int l = RBD_BK_L0 + 1; l < RBD_BK_NL; ++l) bank_accel_step<T>(M, l, RBD_BK_PERM(l), r1, lds);
  RBD_MARK(11);
  if (vdot) store_joint_v(r1.b, vdot, Lv, r1.vd);
  if (FUSED) store_joint_v(r1.b, (T*)F.W.vd[F.stage], Lv, r1.vd);
}
#ifndef RBD_BANK_FIXED_NL
template <typename T, bool FUSED, bool SIMPLE>
__global__ __launch_bounds__(256, RBD_BANK_WAVES) void aba_bank_kernel(BankModel M, long B, typename InPtr<T, !FUSED>::type q, typename InPtr<T, !FUSED>::type v,
                                                         const T* __restrict__ tau, const T* __restrict__ fext, T* __restrict__ vdot,
                                                         T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf, MkFuse F) {
  extern __shared__ double bank_lds_raw[];
  aba_bank_body<T, FUSED, SIMPLE>(M, B, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, F, reinterpret_cast<Pair2<T>*>(bank_lds_raw));
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Banked RNEA: inverse_dynamics! (vdot != nullptr) and dynamics_bias! (vdot == nullptr), src/mechanism_algorithms.jl:542-553,
// :484-498 — the same two-bodies-per-lane mapping as aba_bank_kernel.  RNEA is sums only, so every tree joint type is in scope;
// SIMPLE (revolute joints + 6-dof joints on the world) drops the joint-type interpretation as in aba_bank_kernel.
// The kinematics sweep carries K = (R 9, p 3, Tw 6, acc 6) per lane, committed by exec-masked moves; hops that are not "previous lane"
// (later children, the hop between the banks) go through the LDS exchange columns.
// ---------------------------------------------------------------------------------------------------------------------
enum { RNEA_LDS_PAIRS = 12 };
template <typename T> struct RneaRegs {
  Body<T> b;
  const T* rb;
  int pcol;
  T K[24], w[6];
  RBD_DEV T* R() { return K; }
  RBD_DEV T* p() { return K + 9; }
  RBD_DEV T* Tw() { return K + 12; }
  RBD_DEV T* acc() { return K + 18; }
};

// kinematics step with spatial accelerations (spatial_accelerations! :387-417): a_b = a_p + (-T_b) x T_p + X a_joint — computed by every lane
// from the parent's values k24, committed in the lanes of `mine`
// SIMPLE: tl = (axis, joint velocity, joint acceleration): joint twist and acceleration are S v, S v̇ with S = (R a; p x R a)
template <typename T, bool SIMPLE>
RBD_DEV void rnea_fk_commit(bool mine, RneaRegs<T>& c, const T* k24, const T* XR, const T* Xp, const T* tl, const T* al) {
  const T* pR = k24;
  const T* pp = k24 + 9;
  const T* pT = k24 + 12;
  const T* pa = k24 + 18;
  T n[24], vJ[6], nT[6], cr[6], aj[6];
  matmul3(pR, XR, n);
  matvec3(pR, Xp, n + 9);
#pragma unroll
  for (int k = 0; k < 3; ++k) n[9 + k] += pp[k];
  if (SIMPLE) {
    T S[6];
    matvec3(n, tl, S);
    cross3(n + 9, S, S + 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) { vJ[k] = S[k] * tl[3]; aj[k] = S[k] * tl[4]; }
  } else {
    xmotion(n, n + 9, tl, vJ);
    xmotion(n, n + 9, al, aj);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) { n[12 + k] = pT[k] + vJ[k]; nT[k] = -n[12 + k]; }
  se3_comm(nT, pT, cr);
#pragma unroll
  for (int k = 0; k < 6; ++k) n[18 + k] = pa[k] + cr[k] + aj[k];
  massign<T, 24>(mine, c.K, n);
}
template <typename T, bool SIMPLE>
RBD_DEV void rnea_fk_step(const BankModel& M, int l, bool perm, RneaRegs<T>& c, Pair2<T>* lds, const T* XR, const T* Xp, const T* tl, const T* al) {
  T k24[24];
  if (perm) {  // uniform
    if (c.b.level == l - 1 && c.b.nchild >= 1) {
      RBD_KEEP_BRANCH();
      lds_put<T, 24>(lds + threadIdx.x, 0, c.K);
    }
    wave_lds_sync();
    lds_get<T, 24>(lds + c.pcol, 0, k24);
    wave_lds_sync();
  } else {
#pragma unroll
    for (int k = 0; k < 24; ++k) k24[k] = from_prev_lane(c.K[k]);
  }
  rnea_fk_commit<T, SIMPLE>(c.b.level == l, c, k24, XR, Xp, tl, al);
}

// joint_wrenches_and_torques! (:442-459): lanes of `tk` at level l-1 add the wrenches of their children (lanes of `gv` at level l)
template <typename T, bool CROSS> RBD_DEV void rnea_gather(int l, int ns, const RneaRegs<T>& gv, RneaRegs<T>& tk) {
  const bool takes = (tk.b.level == l - 1);
  if (!CROSS) {
    T t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = from_next_lane(gv.w[k]);
    if (takes && tk.b.nchild >= 1) {
      RBD_KEEP_BRANCH();
#pragma unroll
      for (int k = 0; k < 6; ++k) tk.w[k] += t[k];
    }
  }
#pragma unroll 1
  for (int s = CROSS ? 0 : 1; s < ns; ++s) {
    const bool take = takes && (s < tk.b.nchild);
    const int src = take ? tk.b.base + child_sel(tk.b, s) : tk.b.lane;
    T t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = shfl(gv.w[k], src);
    if (take) {
      RBD_KEEP_BRANCH();
#pragma unroll
      for (int k = 0; k < 6; ++k) tk.w[k] += t[k];
    }
  }
}

// acc_out / jw_out (nullable): spatial accelerations of the bodies (spatial_accelerations! :387-417, the gravitational acceleration of
// the root included, as result.accelerations holds them) and joint wrenches (:442-459), 6 x n_bodies x B in reference body order, root frame
template <typename T, bool SIMPLE>
RBD_DEV void rnea_bank_body(const BankModel& M, long B, int ncol, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ vdot,
                            const T* __restrict__ fext, T* __restrict__ tau, T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf, T* __restrict__ acc_out,
                            T* __restrict__ jw_out, Pair2<T>* const lds) {
  RneaRegs<T> r0, r1;
  const bool out_vec = store6_vec(acc_out ? acc_out : jw_out, Lf, (int)sizeof(T)) && store6_vec(jw_out ? jw_out : acc_out, Lf, (int)sizeof(T));  // uniform
  auto fetch = [&](int k, RneaRegs<T>& c, T* qj, T* vj, T* aj) {
    load_bank_body(M, k, B, c.b);
    c.rb = reinterpret_cast<const T*>(M.rb[k]) + (c.b.sub < M.nbk[k] ? c.b.sub : 0) * RB_STRIDE;
    c.pcol = (int)threadIdx.x - c.b.sub + (c.b.parent >= 0 ? c.b.parent : c.b.sub);
    if (SIMPLE) {  // unconditional loads at clamped addresses (see aba_bank_kernel)
      const bool fl = c.b.jtype == RBD_JOINT_QUAT_FLOATING;
      const long sc = c.b.state < B ? c.b.state : B - 1;
      const T* qp = q + ((long)c.b.qoff * Lq.sk + sc * Lq.sb);
      const long va = (long)c.b.voff * Lv.sk + sc * Lv.sb;
      const bool hv = v != nullptr, ha = vdot != nullptr;  // uniform
      qj[0] = qp[0];
      vj[0] = hv ? v[va] : T(0);
      aj[0] = ha ? vdot[va] : T(0);
#pragma unroll
      for (int k2 = 1; k2 < 7; ++k2) qj[k2] = T(0);
#pragma unroll
      for (int k2 = 1; k2 < 6; ++k2) { vj[k2] = T(0); aj[k2] = T(0); }
      if (fl) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int k2 = 1; k2 < 7; ++k2) qj[k2] = qp[k2 * Lq.sk];
        if (hv) {
#pragma unroll
          for (int k2 = 1; k2 < 6; ++k2) vj[k2] = v[va + k2 * Lv.sk];
        }
        if (ha) {
#pragma unroll
          for (int k2 = 1; k2 < 6; ++k2) aj[k2] = vdot[va + k2 * Lv.sk];
        }
      }
    } else {
      load_joint_q(c.b, q, Lq, qj);
      load_joint_v(c.b, v, Lv, vj);
      load_joint_v(c.b, vdot, Lv, aj);
    }
  };
  auto setup = [&](RneaRegs<T>& c, const T* qj, const T* vj, const T* aj, T* XR, T* Xp, T* tl, T* al, T* ta) {
    if (SIMPLE) {
      const bool fl = c.b.jtype == RBD_JOINT_QUAT_FLOATING;
      const T ax[3] = {c.rb[RB_AXIS], c.rb[RB_AXIS + 1], c.rb[RB_AXIS + 2]};
      ta[0] = ax[0]; ta[1] = ax[1]; ta[2] = ax[2]; ta[3] = vj[0]; ta[4] = aj[0];
      T Rj[9], pj[3] = {T(0), T(0), T(0)}, sn, cs;
      sincos_fast(qj[0], &sn, &cs);
      rot_axis_sc(ax, sn, cs, Rj);
#pragma unroll
      for (int k = 0; k < 3; ++k) { tl[k] = ax[k] * vj[0]; tl[3 + k] = T(0); al[k] = ax[k] * aj[0]; al[3 + k] = T(0); }
      if (fl) {
        RBD_KEEP_BRANCH();
        rot_quat(qj[0], qj[1], qj[2], qj[3], Rj);
        pj[0] = qj[4]; pj[1] = qj[5]; pj[2] = qj[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { tl[k] = vj[k]; al[k] = aj[k]; }
        if (qdot != nullptr && c.b.valid) {
          const T w = qj[0], x = qj[1], y = qj[2], z = qj[3];
          T o[7];
          o[0] = (-x * vj[0] - y * vj[1] - z * vj[2]) * T(0.5);
          o[1] = (w * vj[0] - z * vj[1] + y * vj[2]) * T(0.5);
          o[2] = (z * vj[0] + w * vj[1] - x * vj[2]) * T(0.5);
          o[3] = (-y * vj[0] + x * vj[1] + w * vj[2]) * T(0.5);
          matvec3(Rj, vj + 3, o + 4);
#pragma unroll
          for (int k = 0; k < 7; ++k) qdot[(long)(c.b.qoff + k) * Lq.sk + c.b.state * Lq.sb] = o[k];
        }
      } else if (qdot != nullptr && c.b.valid) {
        qdot[(long)c.b.qoff * Lq.sk + c.b.state * Lq.sb] = vj[0];
      }
      T XpR[9], Xpp[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) XpR[k] = c.rb[RB_XPR + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Xpp[k] = c.rb[RB_XPP + k];
      matmul3(XpR, Rj, XR);
      matvec3(XpR, pj, Xp);
#pragma unroll
      for (int k = 0; k < 3; ++k) Xp[k] += Xpp[k];
    } else {
      store_qdot(c.b, qdot, Lq, qj, vj);
      local_transform(c.b, c.rb, qj, XR, Xp);
      local_joint_motion(c.b, c.rb, vj, tl);
      local_joint_motion(c.b, c.rb, aj, al);  // joint_spatial_acceleration: S_local * v̇
    }
    // as if at level 0: H = XL, T = vJ, a = -g + X a_joint; deeper lanes overwrite at their level
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R()[i] = XR[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.p()[i] = Xp[i];
    xmotion(c.R(), c.p(), tl, c.Tw());
    xmotion(c.R(), c.p(), al, c.acc());
    c.acc()[3] -= T(M.gravity[0]); c.acc()[4] -= T(M.gravity[1]); c.acc()[5] -= T(M.gravity[2]);
  };
  // newton_euler! (:428-439): w = I a + T x* I T - wext
  auto newton_euler = [&](RneaRegs<T>& c) {
    RInertia<T> I;
    T Jb[6], mc[3], Ia[6], x[6], fe[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = c.rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = c.rb[RB_MC + k];
    inertia_to_root(Jb, mc, c.rb[RB_M], c.R(), c.p(), I);
    mul_inertia(I, c.acc(), Ia);
    momentum_cross(I, c.Tw(), x);
    load_body_wrench(c.b, fext, Lf, fe);
#pragma unroll
    for (int k = 0; k < 6; ++k) c.w[k] = Ia[k] + x[k] - fe[k];
    if (acc_out != nullptr && c.b.valid) store6(acc_out, 6L * c.b.orig, Lf, c.b.state, c.acc(), out_vec);
  };
  auto project = [&](RneaRegs<T>& c) {  // tau = S' w
    T out[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    if (c.b.jtype == RBD_JOINT_QUAT_FLOATING) {
      xforce_inv(c.R(), c.p(), c.w, out);
    } else if (SIMPLE) {  // S = (R a; p x R a)
      const T ax[3] = {c.rb[RB_AXIS], c.rb[RB_AXIS + 1], c.rb[RB_AXIS + 2]};
      T S[6];
      matvec3(c.R(), ax, S);
      cross3(c.p(), S, S + 3);
      out[0] = dot6(S, c.w);
    } else {
      const T ax[3] = {c.rb[RB_AXIS], c.rb[RB_AXIS + 1], c.rb[RB_AXIS + 2]}, ay[3] = {c.rb[RB_AXIS2], c.rb[RB_AXIS2 + 1], c.rb[RB_AXIS2 + 2]};
      for (int k = 0; k < ncol; ++k) {  // ncol (uniform): 3 when the mechanism has QuaternionSpherical / Planar joints, else 1
        T sl[6], S[6];
        subspace_col(c.b.jtype, ax, ay, k, sl);
        xmotion(c.R(), c.p(), sl, S);
        const T d = dot6(S, c.w);
        if (k == 0) out[0] = d; else if (k == 1) out[1] = d; else out[2] = d;
      }
    }
    store_joint_v(c.b, tau, Lv, out);
    if (jw_out != nullptr && c.b.valid) store6(jw_out, 6L * c.b.orig, Lf, c.b.state, c.w, out_vec);
  };

  T qj0[7], vj0[6], aj0[6], qj1[7], vj1[6], aj1[6];
  fetch(0, r0, qj0, vj0, aj0);
  fetch(1, r1, qj1, vj1, aj1);
  {
    T XR[9], Xp[3], tl[6], al[6], ta[5];
    setup(r0, qj0, vj0, aj0, XR, Xp, tl, al, ta);
RBD_BANK_UNROLL
    for (int l = 1; l < RBD_BK_L0; ++l) rnea_fk_step<T, SIMPLE>(M, l, RBD_BK_PERM(l), r0, lds, XR, Xp, SIMPLE ? ta : tl, al);
  }
  {
    T XR[9], Xp[3], tl[6], al[6], ta[5];
    setup(r1, qj1, vj1, aj1, XR, Xp, tl, al, ta);
    {  // across the banks: every bank-0 lane leaves its kinematics in its exchange column, the bodies of level L0 read their parent's
      T k24[24];
      lds_put<T, 24>(lds + threadIdx.x, 0, r0.K);
      wave_lds_sync();
      lds_get<T, 24>(lds + r1.pcol, 0, k24);
      wave_lds_sync();
      rnea_fk_commit<T, SIMPLE>(r1.b.level == RBD_BK_L0, r1, k24, XR, Xp, SIMPLE ? ta : tl, al);
    }
RBD_BANK_UNROLL
    for (int l = RBD_BK_L0 + 1; l < RBD_BK_NL; ++l) rnea_fk_step<T, SIMPLE>(M, l, RBD_BK_PERM(l), r1, lds, XR, Xp, SIMPLE ? ta : tl, al);
  }
  newton_euler(r1);
  newton_euler(r0);
  NsStream nss = ns_begin(M.ns_desc);
  (void)nss;
RBD_BANK_UNROLL
  for (int l = RBD_BK_NL - 1; l > RBD_BK_L0; --l) rnea_gather<T, false>(l, RBD_BK_NS(l), r1, r1);
  rnea_gather<T, true>(RBD_BK_L0, RBD_BK_NS(RBD_BK_L0), r1, r0);
RBD_BANK_UNROLL
  for (int l = RBD_BK_L0 - 1; l >= 1; --l) rnea_gather<T, false>(l, RBD_BK_NS(l), r0, r0);
  project(r0);
  project(r1);
}
#ifndef RBD_BANK_FIXED_NL
template <typename T, bool SIMPLE>
__global__ __launch_bounds__(256, 2) void rnea_bank_kernel(BankModel M, long B, int ncol, const T* __restrict__ q, const T* __restrict__ v,
                                                          const T* __restrict__ vdot, const T* __restrict__ fext, T* __restrict__ tau,
                                                          T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf, T* __restrict__ acc_out,
                                                          T* __restrict__ jw_out) {
  extern __shared__ double bank_lds_raw[];
  rnea_bank_body<T, SIMPLE>(M, B, ncol, q, v, vdot, fext, tau, qdot, Lq, Lv, Lf, acc_out, jw_out, reinterpret_cast<Pair2<T>*>(bank_lds_raw));
}
#endif

}  // namespace rbd
