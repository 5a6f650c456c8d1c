// rbd_bank.hpp — "banked" lane-per-body ABA: every lane carries TWO bodies (included by rbd_kernels.hip after aba_kernel).
//
// aba_kernel keeps one body per lane, so a state of n bodies needs next_pow2(n) lanes and each level-synchronous sweep
// step is issued for all of them although only the bodies of one level do useful work.  The sweeps are issue-bound
// (DESIGN.md §3.3), so the cost of a sweep step is per WAVEFRONT, not per body: packing twice as many states into a
// wavefront halves the instruction issue of the sweeps.  Here the bodies are split by level into two banks — bank 0 =
// levels [0, L0), bank 1 = levels [L0, nlevels) — and lane j of a state carries body j of bank 0 AND body j of bank 1, each
// with its own register set (Atlas: 15 + 16 bodies on 16 lanes, 4 states per wavefront instead of 2).  A sweep step at
// level l touches only the register set of the bank that owns l, so the level loops are the same code as aba_kernel's,
// run once per bank range; the single hop that crosses banks (level L0-1 <-> L0) is a ds_bpermute between the two register
// sets.  Within a bank the bodies are in DFS pre-order, so the first-child hop stays a DPP wave shift.
// The per-body set-up (joint transform, inertia, bias terms) runs once per bank with all lanes busy.
// Scope: 1-dof and fixed tree joints, 6-dof joints on the world (everything else stays with aba_kernel).
#pragma once

namespace rbd {

template <typename T> struct BankRegs {
  Body<T> b;
  const T* rb;
  T R[9], p[3], Tw[6], vJ[6];
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
  const int32_t* ib = M.ib[k] + (b.sub < M.nbk[k] ? b.sub : 0) * IB_STRIDE;
  b.parent = ib[IB_PARENT];
  b.jtype = ib[IB_JTYPE];
  b.qoff = ib[IB_QOFF];
  b.voff = ib[IB_VOFF];
  b.level = b.valid ? ib[IB_LEVEL] : -1;
  b.nchild = ib[IB_NCHILD];
  b.orig = ib[IB_ORIG];
  b.flags = ib[IB_FLAGS];
#pragma unroll
  for (int c = 0; c < IB_MAXCHILD; ++c) b.child[c] = ib[IB_CHILD0 + c];
  b.plane = b.parent >= 0 ? b.base + b.parent : b.lane;  // parent slot (of the parent's bank) in this state's lane group
}

// top-down hop of N values: lanes of `cb` read their parent's copy of x (the parent's bank register set)
template <typename T, int N, bool CROSS> RBD_DEV void bank_pull(const BankModel& M, const Body<T>& cb, int l, const T* x, T* out) {
  if (CROSS || ((M.perm_down >> l) & 1)) {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = shfl(x[k], cb.plane);
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = from_prev_lane(x[k]);
  }
}

// forward-kinematics step at level l: `c` (bodies at level l) from `par` (their parents' bank)
template <typename T, bool CROSS> RBD_DEV void bank_fk_step(const BankModel& M, int l, const BankRegs<T>& par, BankRegs<T>& c, const T* XR, const T* Xp, const T* tl) {
  T pR[9], pp[3], pT[6];
  bank_pull<T, 9, CROSS>(M, c.b, l, par.R, pR);
  bank_pull<T, 3, CROSS>(M, c.b, l, par.p, pp);
  bank_pull<T, 6, CROSS>(M, c.b, l, par.Tw, pT);
  if (c.b.level == l) {
    matmul3(pR, XR, c.R);
    matvec3(pR, Xp, c.p);
#pragma unroll
    for (int k = 0; k < 3; ++k) c.p[k] += pp[k];
    xmotion(c.R, c.p, tl, c.vJ);
#pragma unroll
    for (int k = 0; k < 6; ++k) c.Tw[k] = pT[k] + c.vJ[k];
  }
}

// U = IA S, D = S'U, u = tau - S'pA for the lanes of `r` at level l (1-dof joints; fixed joints keep U = 0, 1/D = 0)
template <typename T> RBD_DEV void bank_finish_joint(int l, BankRegs<T>& r) {
  if (r.b.level == l && joint_nv(r.b.jtype) == 1) {
    sym6_mul(r.IA, r.S, r.U);
    r.u = r.tj[0] - dot6(r.S, r.pA);
    r.Dinv = rcp_nr(dot6(r.S, r.U));
  }
}

// bottom-up hand-off: lanes of `gv` at level l give (Ia, pa) = (IA - U D^-1 U', pA + Ia cb + U D^-1 u); lanes of `tk` at level
// l-1 add what their children give.  In-bank (gv and tk are the same register set): first child by DPP, the others by
// ds_bpermute; across banks every child by ds_bpermute.  Entries are formed and consumed one at a time (see aba_kernel).
template <typename T, bool CROSS> RBD_DEV void bank_handoff(int l, int ns, const BankRegs<T>& gv, BankRegs<T>& tk) {
  const bool takes = (tk.b.level == l - 1);
  T W[6];  // U D^-1 (6-dof joints sit at level 0 and never give: their 1/D stays 0)
#pragma unroll
  for (int k = 0; k < 6; ++k) W[k] = gv.U[k] * gv.Dinv;
  const T m0 = (!CROSS && takes && tk.b.nchild >= 1) ? T(1) : T(0);
  T Iac[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, gp[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = i; j < 6; ++j) {
      const T g = gv.IA[SI(i, j)] - W[i] * gv.U[j];
      Iac[i] += g * gv.cb[j];
      if (j > i) Iac[j] += g * gv.cb[i];
      if (!CROSS) tk.IA[SI(i, j)] += keep(from_next_lane(g), m0);
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    gp[k] = gv.pA[k] + Iac[k] + W[k] * gv.u;
    if (!CROSS) tk.pA[k] += keep(from_next_lane(gp[k]), m0);
  }
#pragma unroll 1
  for (int s = CROSS ? 0 : 1; s < ns; ++s) {
    const bool take = takes && (s < tk.b.nchild);
    const int src = take ? tk.b.base + child_sel(tk.b, s) : tk.b.lane;
    const T mask = take ? T(1) : T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      T tmp[6];
#pragma unroll
      for (int j = i; j < 6; ++j) tmp[j] = shfl(gv.IA[SI(i, j)] - W[i] * gv.U[j], src);
#pragma unroll
      for (int j = i; j < 6; ++j) tk.IA[SI(i, j)] += keep(tmp[j], mask);
    }
    T tp[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) tp[k] = shfl(gp[k], src);
#pragma unroll
    for (int k = 0; k < 6; ++k) tk.pA[k] += keep(tp[k], mask);
  }
}

// v̇ = D^-1 (u - U'a'), a = a' + S v̇ with a' = a_parent + cb
template <typename T> RBD_DEV void bank_joint_accel(BankRegs<T>& r, const T* a_parent) {
  T ap[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) ap[k] = a_parent[k] + r.cb[k];
  const T x = r.Dinv * (r.u - dot6(r.U, ap));
  r.vd[0] = x;
#pragma unroll
  for (int k = 0; k < 6; ++k) r.acc[k] = ap[k] + r.S[k] * x;
}

// Register budget: both banks' recursion state would need ~300 VGPRs in fp64, i.e. one wavefront per SIMD, and a lone
// wavefront issues at only ~5 cycles per instruction.  The bank that is not being swept therefore parks its live values in a
// lane-private LDS column (PARK_SLOTS values per lane, no synchronisation needed): the kernel fits 256 VGPRs and two
// wavefronts per SIMD interleave.
enum { PARK_KIN = 0 /* R 9, p 3, Tw 6, vJ 6 of bank 0 */, PARK_FWD = 12 /* U 6, 1/D, u, S 6, cb 6 of bank 1 */, PARK_SLOTS = 32,
       PARK_WF = 32 /* re-rooted tree: S^-T tau of the floating joint, left by the root lane for the old floating body */, PARK_SLOTS_RR = 38 };

#ifdef RBD_PROFILE_PHASES
__device__ long long rbd_bank_phase_clock[16];
#define RBD_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) rbd_bank_phase_clock[i] = clock64(); } while (0)
#else
#define RBD_MARK(i)
#endif

// RR = true: the tree re-rooted at its centre (rbd_reroot.hpp).  The level-0 body is then a VIRTUAL floating base — its pose and twist come
// from the old floating body's coordinates through the chain of original joints, its acceleration solves IA a = −pA — and the old floating
// body (flag BFD_FCARRY, now somewhere down the tree) takes the floating joint's force as an external wrench and gives back v̇ of the floating
// joint from its own spatial acceleration.  Everything else is the same code on different records (reversed joints are ordinary records).
template <typename T, bool RR = false>
__global__ __launch_bounds__(256, 2) void aba_bank_kernel(BankModel M, long B, const T* q, const T* v,  /* no restrict: F.q_state / F.v_state alias them when fused */
                                                         const T* __restrict__ tau, const T* __restrict__ fext, T* __restrict__ vdot,
                                                         T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf, MkFuse F) {
  extern __shared__ double park_raw[];
  T* const park = reinterpret_cast<T*>(park_raw) + threadIdx.x;  // slot i of this lane: park[i * 256]
  RBD_MARK(0);
  BankRegs<T> r0, r1;
  // RR: coordinates of the chain joints between the old floating body and the new root, requested with the first loads of the launch
  T cq[RC_MAX], cv[RC_MAX];
  if (RR) {
    const long st = ((long)blockIdx.x * blockDim.x + threadIdx.x) / M.lps;
    const long stc = st < B ? st : B - 1;
#pragma unroll
    for (int j = 0; j < RC_MAX; ++j) {
      const bool on = j < M.reroot.nchain;
      const int qo = on ? M.reroot.chain_i[4 * j + 1] : 0, vo = on ? M.reroot.chain_i[4 * j + 2] : 0;
      cq[j] = on ? q[(long)qo * Lq.sk + stc * Lq.sb] : T(0);
      cv[j] = (on && v) ? v[(long)vo * Lv.sk + stc * Lv.sb] : T(0);
    }
  }
  // ---- per-body set-up (once per bank, every lane busy): joint transform and joint twist in the joint frame ----
  // The global loads of BOTH banks are issued up front (two dependent round trips: body record, then q / v / tau), so that
  // bank 1's latency hides behind bank 0's sweep; the arithmetic of a bank runs right before its own sweep.
  auto fetch = [&](int k, BankRegs<T>& c, T* qj, T* vj) {
    load_bank_body(M, k, B, c.b);
    c.rb = reinterpret_cast<const T*>(M.rb[k]) + (c.b.sub < M.nbk[k] ? c.b.sub : 0) * RB_STRIDE;
    load_joint_q(c.b, q, Lq, qj);
    load_joint_v(c.b, v, Lv, vj);
    load_joint_v(c.b, tau, Lv, c.tj);
  };
  auto setup = [&](BankRegs<T>& c, T* qj, T* vj, T* XR, T* Xp, T* tl) {
    store_qdot(c.b, qdot, Lq, qj, vj);
    local_transform(c.b, c.rb, qj, XR, Xp);
    local_joint_motion(c.b, c.rb, vj, tl);
    // as if at level 0 (transform to root = local transform); deeper lanes overwrite at their level
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = XR[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.p[i] = Xp[i];
    xmotion(c.R, c.p, tl, c.vJ);
#pragma unroll
    for (int i = 0; i < 6; ++i) c.Tw[i] = c.vJ[i];
    if (RR) {
      if (c.b.valid && (c.b.flags & BFD_VROOT)) {  // qj, vj of this lane ARE the floating joint's coordinates (its record carries their offsets)
        {  // the floating joint's force as a wrench in the root frame, S^-T tau_f: this lane holds tau_f (c.tj) and is about to build the old
           // floating body's pose anyway; the old floating body picks the wrench up from this lane's LDS column in terms()
          T Rf[9], pf[3], wf[6];
          reroot_fb_pose(M.reroot, qj, Rf, pf);
          xforce(Rf, pf, c.tj, wf);
#pragma unroll
          for (int i = 0; i < 6; ++i) park[(PARK_WF + i) * 256] = wf[i];
        }
        reroot_root_kinematics<T, T>(M.reroot, qj, vj, cq, cv, c.R, c.p, c.Tw);
#pragma unroll
        for (int i = 0; i < 6; ++i) c.vJ[i] = c.Tw[i];  // as for a floating joint on the world: [T, vJ] = 0
      }
    }
  };
  // ---- per-body terms in the root frame: motion subspace, bias acceleration, inertia, bias force ----
  auto terms = [&](BankRegs<T>& c, bool accumulate) {  // accumulate: IA, pA already hold what the children handed up
    T e1[6] = {T(1), T(0), T(0), T(0), T(0), T(0)}, sl[6];
    local_joint_motion(c.b, c.rb, e1, sl);
    xmotion(c.R, c.p, sl, c.S);
    const bool floating = c.b.jtype == RBD_JOINT_QUAT_FLOATING;
    if (floating || joint_nv(c.b.jtype) == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c.S[i] = T(0);
    }
    se3_comm(c.Tw, c.vJ, c.cb);
    RInertia<T> I;
    T Jb[6], mc[3], fe[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Jb[i] = c.rb[RB_J + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) mc[i] = c.rb[RB_MC + i];
    inertia_to_root(Jb, mc, c.rb[RB_M], c.R, c.p, I);
    T Io[21], po[6];
    sym6_from_inertia(I, Io);
    momentum_cross(I, c.Tw, po);
    load_body_wrench(c.b, fext, Lf, fe);
    // idle lanes carry zeros (their values may be shifted into masked-off neighbours)
#pragma unroll
    for (int i = 0; i < 21; ++i) c.IA[i] = (accumulate ? c.IA[i] : T(0)) + (c.b.valid ? Io[i] : T(0));
#pragma unroll
    for (int i = 0; i < 6; ++i) c.pA[i] = (accumulate ? c.pA[i] : T(0)) + (c.b.valid ? po[i] - fe[i] : T(0));
#pragma unroll
    for (int i = 0; i < 6; ++i) c.U[i] = T(0);
    c.Dinv = T(0);
    c.u = T(0);
    if (floating && !RR) xforce(c.R, c.p, c.tj, c.U);  // 6-dof joints (level 0): U carries S^-T tau; R, p stay parked for the top-down sweep
    if (RR) {
      if (c.b.valid && (c.b.flags & BFD_FCARRY)) {  // the floating joint's force acts on the old floating body: pA -= S^-T tau_f
        const T* rootcol = park - c.b.sub;          // the root is slot 0 of bank 0 of this state: its LDS column
#pragma unroll
        for (int k = 0; k < 6; ++k) c.pA[k] -= rootcol[(PARK_WF + k) * 256];
      }
    }
  };
  // RR: v̇ of the floating joint from the old floating body's spatial acceleration, v̇_f = S^-1 (a - a_world)
  auto fb_accel_out = [&](const BankRegs<T>& c, bool bank0) {
    if (c.b.valid && (c.b.flags & BFD_FCARRY)) {
      T Rn[9], pn[3], Rf[9], pf[3], dd[6], vf[6];
      if (bank0) {  // bank 0's transforms stay parked through the sweeps
#pragma unroll
        for (int i = 0; i < 9; ++i) Rn[i] = park[(PARK_KIN + i) * 256];
#pragma unroll
        for (int i = 0; i < 3; ++i) pn[i] = park[(PARK_KIN + 9 + i) * 256];
        reroot_fb_pose_from_rebased<T, T>(M.reroot, Rn, pn, Rf, pf);
      } else {  // (a tree whose old floating body ends up in bank 1: its pose again from the coordinates)
        T q7[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) q7[k] = q[(long)(M.reroot.fq + k) * Lq.sk + c.b.state * Lq.sb];
        reroot_fb_pose(M.reroot, q7, Rf, pf);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) { dd[k] = c.acc[k]; dd[3 + k] = c.acc[3 + k] + T(M.gravity[k]); }
      xmotion_inv(Rf, pf, dd, vf);
      if (vdot) {
#pragma unroll
        for (int k = 0; k < 6; ++k) vdot[(long)(M.reroot.fv + k) * Lv.sk + c.b.state * Lv.sb] = vf[k];
      }
    }
  };

  // ---- top-down: transforms to root and twists (update_transforms!, update_twists_wrt_world!) ----
  T qj0[7], vj0[6], qj1[7], vj1[6];
  fetch(0, r0, qj0, vj0);
  fetch(1, r1, qj1, vj1);
  if (F.stage >= 0) {
    // `simulate` fusion: this launch is stage F.stage of a Munthe-Kaas RK4 step (see aba_kernel).  Both banks' stage bookkeeping
    // runs here, back to back, so that their loads of the integrator buffers share one round trip and bank 1's hide behind the
    // SE(3) log/exp of bank 0's floating joint.
    if (F.close_prev) {
      mk_stage_lane(r0.b, 4, (T)F.dt, qj0, vj0, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
      mk_stage_lane(r1.b, 4, (T)F.dt, qj1, vj1, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
    }
    mk_stage_lane(r0.b, F.stage, (T)F.dt, qj0, vj0, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
    mk_stage_lane(r1.b, F.stage, (T)F.dt, qj1, vj1, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
  }
  {
    T XR[9], Xp[3], tl[6];
    setup(r0, qj0, vj0, XR, Xp, tl);
    RBD_MARK(1);
#pragma unroll 1
    for (int l = 1; l < M.L0; ++l) bank_fk_step<T, false>(M, l, r0, r0, XR, Xp, tl);
  }
  {
    T XR[9], Xp[3], tl[6];
    RBD_MARK(2);
    setup(r1, qj1, vj1, XR, Xp, tl);
    RBD_MARK(3);
    bank_fk_step<T, true>(M, M.L0, r0, r1, XR, Xp, tl);
    // bank 0 rests until bank 1 has been swept bottom-up
#pragma unroll
    for (int i = 0; i < 9; ++i) park[(PARK_KIN + i) * 256] = r0.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) park[(PARK_KIN + 9 + i) * 256] = r0.p[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { park[(PARK_KIN + 12 + i) * 256] = r0.Tw[i]; park[(PARK_KIN + 18 + i) * 256] = r0.vJ[i]; }
#pragma unroll 1
    for (int l = M.L0 + 1; l < M.nlevels; ++l) bank_fk_step<T, false>(M, l, r1, r1, XR, Xp, tl);
  }
  RBD_MARK(4);
  terms(r1, false);
  RBD_MARK(5);

  // ---- bottom-up: articulated-body inertias and bias forces ----
#pragma unroll 1
  for (int l = M.nlevels - 1; l > M.L0; --l) {
    bank_finish_joint(l, r1);
    bank_handoff<T, false>(l, (int)M.nslots[l], r1, r1);
  }
  bank_finish_joint(M.L0, r1);
  RBD_MARK(6);
  // across the banks: the children's hand-off lands in bank 0's (still empty) accumulators; bank 1 then keeps only what the
  // top-down sweep needs, parked while bank 0 is swept; only then does bank 0 wake up and add its own inertia and bias force
#pragma unroll
  for (int i = 0; i < 21; ++i) r0.IA[i] = T(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) r0.pA[i] = T(0);
  bank_handoff<T, true>(M.L0, (int)M.nslots[M.L0], r1, r0);
#pragma unroll
  for (int i = 0; i < 9; ++i) r0.R[i] = park[(PARK_KIN + i) * 256];
#pragma unroll
  for (int i = 0; i < 3; ++i) r0.p[i] = park[(PARK_KIN + 9 + i) * 256];
#pragma unroll
  for (int i = 0; i < 6; ++i) { r0.Tw[i] = park[(PARK_KIN + 12 + i) * 256]; r0.vJ[i] = park[(PARK_KIN + 18 + i) * 256]; }
  // (bank 0's Tw, vJ slots are free again: bank 1's set takes them; R, p stay for the 6-dof joints)
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    park[(PARK_FWD + i) * 256] = r1.U[i];
    park[(PARK_FWD + 8 + i) * 256] = r1.S[i];
    park[(PARK_FWD + 14 + i) * 256] = r1.cb[i];
  }
  park[(PARK_FWD + 6) * 256] = r1.Dinv;
  park[(PARK_FWD + 7) * 256] = r1.u;
  RBD_MARK(7);
  terms(r0, true);
  RBD_MARK(8);
#pragma unroll 1
  for (int l = M.L0 - 1; l >= 1; --l) {
    bank_finish_joint(l, r0);
    bank_handoff<T, false>(l, (int)M.nslots[l], r0, r0);
  }
  bank_finish_joint(0, r0);
  RBD_MARK(9);

  // ---- top-down: accelerations and v̇ ----
#pragma unroll
  for (int i = 0; i < 6; ++i) { r0.acc[i] = T(0); r0.vd[i] = T(0); }
  {
    const T a0[6] = {T(0), T(0), T(0), T(-M.gravity[0]), T(-M.gravity[1]), T(-M.gravity[2])};  // a_world = -gravity
    BankRegs<T>& c = r0;
    if (c.b.level == 0) {
      if (c.b.jtype == RBD_JOINT_QUAT_FLOATING) {
        // IA a = S^-T tau - pA;  v̇ = S^-1 (a - a_world)   ([T, vJ] = 0 on the world)
        T rhs[6], d[6], Rs[9], ps[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) rhs[i] = c.U[i] - c.pA[i];
        sym6_solve(c.IA, rhs, c.acc);
        if (!RR) {
#pragma unroll
          for (int i = 0; i < 6; ++i) d[i] = c.acc[i] - a0[i];
#pragma unroll
          for (int i = 0; i < 9; ++i) Rs[i] = park[(PARK_KIN + i) * 256];
#pragma unroll
          for (int i = 0; i < 3; ++i) ps[i] = park[(PARK_KIN + 9 + i) * 256];
          xmotion_inv(Rs, ps, d, c.vd);
        }
      } else {
        bank_joint_accel(c, a0);
      }
    }
  }
#pragma unroll 1
  for (int l = 1; l < M.L0; ++l) {
    T ap[6];
    bank_pull<T, 6, false>(M, r0.b, l, r0.acc, ap);
    if (r0.b.level == l) bank_joint_accel(r0, ap);
  }
  RBD_MARK(10);
  if (RR) {
    if (vdot && !(r0.b.flags & BFD_VROOT)) store_joint_v(r0.b, vdot, Lv, r0.vd);  // the virtual root owns no coordinates
    fb_accel_out(r0, true);
  } else {
    if (vdot) store_joint_v(r0.b, vdot, Lv, r0.vd);
    if (F.stage >= 0) store_joint_v(r0.b, (T*)F.W.vd[F.stage], Lv, r0.vd);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    r1.U[i] = park[(PARK_FWD + i) * 256];
    r1.S[i] = park[(PARK_FWD + 8 + i) * 256];
    r1.cb[i] = park[(PARK_FWD + 14 + i) * 256];
    r1.acc[i] = T(0);
    r1.vd[i] = T(0);
  }
  r1.Dinv = park[(PARK_FWD + 6) * 256];
  r1.u = park[(PARK_FWD + 7) * 256];
  {
    T ap[6];
    bank_pull<T, 6, true>(M, r1.b, M.L0, r0.acc, ap);
    if (r1.b.level == M.L0) bank_joint_accel(r1, ap);
  }
#pragma unroll 1
  for (int l = M.L0 + 1; l < M.nlevels; ++l) {
    T ap[6];
    bank_pull<T, 6, false>(M, r1.b, l, r1.acc, ap);
    if (r1.b.level == l) bank_joint_accel(r1, ap);
  }
  RBD_MARK(11);
  if (vdot) store_joint_v(r1.b, vdot, Lv, r1.vd);
  if (RR) fb_accel_out(r1, false);
  else if (F.stage >= 0) store_joint_v(r1.b, (T*)F.W.vd[F.stage], Lv, r1.vd);
}

// ---------------------------------------------------------------------------------------------------------------------
// Banked RNEA: inverse_dynamics! (vdot != nullptr) and dynamics_bias! (vdot == nullptr), src/mechanism_algorithms.jl:542-553,
// :484-498 — the same two-bodies-per-lane mapping as aba_bank_kernel.  RNEA is sums only, so every tree joint type is in scope.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct RneaRegs {
  Body<T> b;
  const T* rb;
  T R[9], p[3], Tw[6], acc[6], w[6];
};

// kinematics step at level l with spatial accelerations (spatial_accelerations! :387-417): a_b = a_p + (-T_b) x T_p + X a_joint
template <typename T, bool CROSS>
RBD_DEV void rnea_fk_step(const BankModel& M, int l, const RneaRegs<T>& par, RneaRegs<T>& c, const T* XR, const T* Xp, const T* tl, const T* al) {
  T pR[9], pp[3], pT[6], pa[6];
  bank_pull<T, 9, CROSS>(M, c.b, l, par.R, pR);
  bank_pull<T, 3, CROSS>(M, c.b, l, par.p, pp);
  bank_pull<T, 6, CROSS>(M, c.b, l, par.Tw, pT);
  bank_pull<T, 6, CROSS>(M, c.b, l, par.acc, pa);
  if (c.b.level == l) {
    matmul3(pR, XR, c.R);
    matvec3(pR, Xp, c.p);
#pragma unroll
    for (int k = 0; k < 3; ++k) c.p[k] += pp[k];
    T vJ[6], nT[6], cr[6], aj[6];
    xmotion(c.R, c.p, tl, vJ);
#pragma unroll
    for (int k = 0; k < 6; ++k) { c.Tw[k] = pT[k] + vJ[k]; nT[k] = -c.Tw[k]; }
    se3_comm(nT, pT, cr);
    xmotion(c.R, c.p, al, aj);
#pragma unroll
    for (int k = 0; k < 6; ++k) c.acc[k] = pa[k] + cr[k] + aj[k];
  }
}

// joint_wrenches_and_torques! (:442-459): lanes of `tk` at level l-1 add the wrenches of their children (lanes of `gv` at level l)
template <typename T, bool CROSS> RBD_DEV void rnea_gather(int l, int ns, const RneaRegs<T>& gv, RneaRegs<T>& tk) {
  const bool takes = (tk.b.level == l - 1);
  if (!CROSS) {
    const T m0 = (takes && tk.b.nchild >= 1) ? T(1) : T(0);
    T t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = from_next_lane(gv.w[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) tk.w[k] += keep(t[k], m0);
  }
#pragma unroll 1
  for (int s = CROSS ? 0 : 1; s < ns; ++s) {
    const bool take = takes && (s < tk.b.nchild);
    const int src = take ? tk.b.base + child_sel(tk.b, s) : tk.b.lane;
    const T mask = take ? T(1) : T(0);
    T t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = shfl(gv.w[k], src);
#pragma unroll
    for (int k = 0; k < 6; ++k) tk.w[k] += keep(t[k], mask);
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void rnea_bank_kernel(BankModel M, long B, int ncol, const T* __restrict__ q, const T* __restrict__ v,
                                                          const T* __restrict__ vdot, const T* __restrict__ fext, T* __restrict__ tau,
                                                          T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf) {
  RneaRegs<T> r0, r1;
  auto fetch = [&](int k, RneaRegs<T>& c, T* qj, T* vj, T* aj) {
    load_bank_body(M, k, B, c.b);
    c.rb = reinterpret_cast<const T*>(M.rb[k]) + (c.b.sub < M.nbk[k] ? c.b.sub : 0) * RB_STRIDE;
    load_joint_q(c.b, q, Lq, qj);
    load_joint_v(c.b, v, Lv, vj);
    load_joint_v(c.b, vdot, Lv, aj);
  };
  auto setup = [&](RneaRegs<T>& c, const T* qj, const T* vj, const T* aj, T* XR, T* Xp, T* tl, T* al) {
    store_qdot(c.b, qdot, Lq, qj, vj);
    local_transform(c.b, c.rb, qj, XR, Xp);
    local_joint_motion(c.b, c.rb, vj, tl);
    local_joint_motion(c.b, c.rb, aj, al);  // joint_spatial_acceleration: S_local * v̇
    // as if at level 0: H = XL, T = vJ, a = -g + X a_joint; deeper lanes overwrite at their level
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = XR[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.p[i] = Xp[i];
    xmotion(c.R, c.p, tl, c.Tw);
    xmotion(c.R, c.p, al, c.acc);
    c.acc[3] -= T(M.gravity[0]); c.acc[4] -= T(M.gravity[1]); c.acc[5] -= T(M.gravity[2]);
  };
  // newton_euler! (:428-439): w = I a + T x* I T - wext
  auto newton_euler = [&](RneaRegs<T>& c) {
    RInertia<T> I;
    T Jb[6], mc[3], Ia[6], x[6], fe[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = c.rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = c.rb[RB_MC + k];
    inertia_to_root(Jb, mc, c.rb[RB_M], c.R, c.p, I);
    mul_inertia(I, c.acc, Ia);
    momentum_cross(I, c.Tw, x);
    load_body_wrench(c.b, fext, Lf, fe);
#pragma unroll
    for (int k = 0; k < 6; ++k) c.w[k] = c.b.valid ? (Ia[k] + x[k] - fe[k]) : T(0);
  };
  auto project = [&](RneaRegs<T>& c) {  // tau = S' w
    T out[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    if (c.b.jtype == RBD_JOINT_QUAT_FLOATING) {
      xforce_inv(c.R, c.p, c.w, out);
    } else {
      const T ax[3] = {c.rb[RB_AXIS], c.rb[RB_AXIS + 1], c.rb[RB_AXIS + 2]}, ay[3] = {c.rb[RB_AXIS2], c.rb[RB_AXIS2 + 1], c.rb[RB_AXIS2 + 2]};
      for (int k = 0; k < ncol; ++k) {  // ncol (uniform): 3 when the mechanism has QuaternionSpherical / Planar joints, else 1
        T sl[6], S[6];
        subspace_col(c.b.jtype, ax, ay, k, sl);
        xmotion(c.R, c.p, sl, S);
        const T d = dot6(S, c.w);
        if (k == 0) out[0] = d; else if (k == 1) out[1] = d; else out[2] = d;
      }
    }
    store_joint_v(c.b, tau, Lv, out);
  };

  T qj0[7], vj0[6], aj0[6], qj1[7], vj1[6], aj1[6];
  fetch(0, r0, qj0, vj0, aj0);
  fetch(1, r1, qj1, vj1, aj1);
  {
    T XR[9], Xp[3], tl[6], al[6];
    setup(r0, qj0, vj0, aj0, XR, Xp, tl, al);
#pragma unroll 1
    for (int l = 1; l < M.L0; ++l) rnea_fk_step<T, false>(M, l, r0, r0, XR, Xp, tl, al);
  }
  {
    T XR[9], Xp[3], tl[6], al[6];
    setup(r1, qj1, vj1, aj1, XR, Xp, tl, al);
    rnea_fk_step<T, true>(M, M.L0, r0, r1, XR, Xp, tl, al);
#pragma unroll 1
    for (int l = M.L0 + 1; l < M.nlevels; ++l) rnea_fk_step<T, false>(M, l, r1, r1, XR, Xp, tl, al);
  }
  newton_euler(r1);
  newton_euler(r0);
#pragma unroll 1
  for (int l = M.nlevels - 1; l > M.L0; --l) rnea_gather<T, false>(l, (int)M.nslots[l], r1, r1);
  rnea_gather<T, true>(M.L0, (int)M.nslots[M.L0], r1, r0);
#pragma unroll 1
  for (int l = M.L0 - 1; l >= 1; --l) rnea_gather<T, false>(l, (int)M.nslots[l], r0, r0);
  project(r0);
  project(r1);
}

}  // namespace rbd
