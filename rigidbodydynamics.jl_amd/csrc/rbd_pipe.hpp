// rbd_pipe.hpp — aba_pipe_kernel: the fused articulated-body algorithm for SMALL batches, with the work of ONE body-step spread over the
// four SIMDs of a compute unit BY ROLE.
//
// Same result as dynamics! (src/mechanism_algorithms.jl:845-864) and as the other mappings; root-frame quantities, the track plan's
// canonical frames and folded bias accelerations (rbd_track_plan.hpp), the walk kernel's un-compose trick (rbd_walk.hpp).  Why another
// mapping: at 4096 states every other one is bound by the tree's critical path — 3 passes x 11 levels = 33 dependent body-steps for
// Atlas, each a few hundred dependent-ish fp64 instructions at ~5 cycles for a wavefront that is alone on its SIMD — not by throughput.
// Here a workgroup is 4 wavefronts over the same 64 lanes = 16 states x 4 tracks (lane = 4 state + track), one wavefront per SIMD,
// and a body-step is cut into stages that run on different wavefronts one step apart, software-pipelined through lane-aligned LDS
// rings (lane l of the consumer reads what lane l of the producer wrote; two or three slots, one workgroup barrier per step):
//     pass A (root -> leaves)   I: sin/cos one step ahead   K: transform chain   T: twist and velocity-product acceleration chain
//     pass B (leaves -> root)   K: transform un-compose chain   I: inertia to the root frame   T: twist un-compose chain + bias force
//                               S: articulated-body recursion (U = IA S, D, the rank-1 update, hand-off to the parent)
//     pass C (root -> leaves)   S alone: v̇ and the a_Δ chain (the motion subspaces of pass B were kept in LDS)
// The dependent chain of a pass is then the longest STAGE (~100-130 fp64 operations in pass B instead of ~390), and 4096 states are
// 256 workgroups: every CU of the chip has one, every SIMD one wavefront.  Edges between tracks are edges between LANES OF ONE
// wavefront (mailboxes in LDS, program order suffices); what a body leaves behind between the passes sits in the accumulation
// registers of the wavefront that needs it again (K: sin q, cos q; S: U D⁻¹, D⁻¹u), selected by the wave-uniform step index.
// Trees of revolute joints with or without a 6-dof root (the fast path of the track plan); everything else takes the other mappings.
//
// The stage functions are __host__ __device__: tests/emu/pipe_emu.hip runs them lane by lane on the CPU, wavefront after wavefront
// between two barriers in both orders, with the LDS pre-filled with NaN.
#pragma once
#include "rbd_walk.hpp"

namespace rbd {

enum { PIPE_STATES = 16, PR_STRIDE = 17, PIPE_K = 0, PIPE_I = 1, PIPE_T = 2, PIPE_S = 3 };
enum { PRK_N = 12, PRI_N = 10, PRT_N = 34 };                    // values per lane of a ring slot: K (R, p) | I (J, c, m) | T (p̃A, I, S or S⁻ᵀτ, R, p)
enum { PRT_PA = 0, PRT_I = 6, PRT_S = 16, PRT_RP = 22 };

template <typename T> struct PipeCtx {
  WalkModel M;  // G = 4 (tracks padded with idle records); wk: the unpacked records (walk_unpack4), rr: constants [ns * 4][TR_STRIDE]
  const T* trr;
  const int32_t* rec;
  T* rows; int rq, rv, rt;                                   // [coordinate][PR_STRIDE]: q (q̇ on the way out) | v | τ (v̇ on the way out)
  T* ringK; T* ringI; T* ringT; T* ringQ;                    // [slot][value][64 lanes]
  T* sstore;                                                 // [step][6][64 lanes]: motion subspaces, pass B -> pass C (the S wavefront's own)
  T* boxAR; T* boxAT; T* boxB; T* boxC; T* parkK; T* parkT;  // [box][value][16 states]
  T a0[6];
};
// LDS: constants | records | rows | rings | mailboxes.  Must agree with pipe_lds_bytes (rbd_walk_plan.hpp).
template <typename T> RBD_HD void pipe_ctx_lds(PipeCtx<T>& c, void* lds) {
  const size_t nrec = (size_t)c.M.ns * 4;
  char* p = reinterpret_cast<char*>(lds);
  c.trr = reinterpret_cast<const T*>(p); p += (nrec * TR_STRIDE * sizeof(T) + 15) & ~(size_t)15;
  c.rec = reinterpret_cast<const int32_t*>(p); p += (nrec * WREC_STRIDE * 4 + 15) & ~(size_t)15;
  T* x = reinterpret_cast<T*>(p);
  c.rows = x; c.rq = 0; c.rv = c.M.nq; c.rt = c.M.nq + c.M.nv; x += (size_t)(c.M.nq + 2 * c.M.nv) * PR_STRIDE;
  c.ringK = x; x += 3 * PRK_N * 64;
  c.ringI = x; x += 2 * PRI_N * 64;
  c.ringT = x; x += 2 * PRT_N * 64;
  c.ringQ = x; x += 2 * 2 * 64;
  c.sstore = x; x += (size_t)c.M.ns * 6 * 64;
  c.boxAR = x; x += (size_t)c.M.nA * 12 * PIPE_STATES;
  c.boxAT = x; x += (size_t)c.M.nA * 12 * PIPE_STATES;
  c.boxB = x; x += (size_t)c.M.nB * 27 * PIPE_STATES;
  c.boxC = x; x += (size_t)c.M.nA * 6 * PIPE_STATES;
  c.parkK = x; x += (size_t)c.M.nS * 12 * PIPE_STATES;
  c.parkT = x;
#pragma unroll
  for (int k = 0; k < 3; ++k) { c.a0[k] = T(0); c.a0[3 + k] = T(-c.M.gravity[k]); }  // a_world = −gravity (mechanism_algorithms.jl:405)
}
template <typename T> RBD_HD T* pipe_row(const PipeCtx<T>& c, int row, int st) { return c.rows + (long)row * PR_STRIDE + st; }
template <typename T> RBD_HD T* pipe_ring(T* base, int nval, int slot, int lane) { return base + (long)slot * nval * 64 + lane; }           // value k at [k * 64]
template <typename T> RBD_HD T* pipe_box(T* base, int nval, int box, int st) { return base + (long)box * nval * PIPE_STATES + st; }         // value k at [k * 16]
template <typename T> RBD_HD WalkRec pipe_rec(const PipeCtx<T>& c, int s, int track) {
  const int32_t* p = c.rec + (s * 4 + track) * WREC_STRIDE;
  WalkRec r;
  r.flags = p[WREC_FLAGS]; r.qoff = p[WREC_QOFF]; r.voff = p[WREC_VOFF]; r.orig6 = p[WREC_ORIG6]; r.nbr = p[WREC_NBR];
  r.a_w = p[WREC_AW]; r.a_r = p[WREC_AR]; r.b_w = p[WREC_BW]; r.b_r0 = p[WREC_BR0]; r.park = p[WREC_PARK]; r.rrf = p[WREC_RRF];
  return r;
}
template <typename T, int K0, int N> RBD_HD void pipe_consts(const PipeCtx<T>& c, int s, int track, T* out) {
  const T* src = c.trr + (long)(s * 4 + track) * TR_STRIDE + K0;
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = src[k];
}
template <typename T, int N> RBD_HD void pipe_get(const T* m, int stride, T* out) {
#pragma unroll
  for (int k = 0; k < N; ++k) out[k] = m[k * stride];
}
template <typename T, int N> RBD_HD void pipe_put(T* m, int stride, const T* in) {
#pragma unroll
  for (int k = 0; k < N; ++k) m[k * stride] = in[k];
}

// chain registers of the roles
template <typename T> struct PipeK { T R[9], p[3]; };
template <typename T> struct PipeT { T Tw[6], av[6]; };
template <typename T> struct PipeS { T cI[21], cP[6], ad[6]; };

template <typename T> RBD_HD void pipe_identity(PipeK<T>& W) {
#pragma unroll
  for (int k = 0; k < 9; ++k) W.R[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) W.p[k] = T(0);
}
// the transform chain's parent for a body that is not chained to the step before it on its track: the world, or the parent's A mailbox
template <typename T> RBD_HD void pipe_parent_k(const PipeCtx<T>& c, const WalkRec& r, int st, PipeK<T>& W) {
  if (r.flags & TF_CHAINED) return;
  if (r.flags & TF_LEVEL0) { pipe_identity(W); return; }
  const T* m = pipe_box(c.boxAR, 12, r.a_r, st);
  pipe_get<T, 9>(m, PIPE_STATES, W.R);
  pipe_get<T, 3>(m + 9 * PIPE_STATES, PIPE_STATES, W.p);
}

// ---------------- pass A ----------------
// K, step s: transform to root of the body (mechanism_state.jl:687-700), sin/cos kept for the later passes; -> ring slot, A mailbox, parking
template <typename T> RBD_HD void pipe_a_k(const PipeCtx<T>& c, PipeK<T>& W, WalkStash<T>& St, int s, int slot, const WalkRec& r, int lane) {
  const int track = lane & 3, st = lane >> 2;
  if (!(r.flags & TF_VALID)) return;
  T rr[TR_J];
  pipe_consts<T, 0, TR_J>(c, s, track, rr);
  pipe_parent_k(c, r, st, W);
  T Rn[9], pn[3];
  if (r.flags & TF_FLOATING) {
    T q7[7];
    pipe_get<T, 7>(pipe_row(c, c.rq + r.qoff, st), PR_STRIDE, q7);
    walk_compose_floating(W.R, W.p, rr, q7, Rn, pn);
  } else {
    T sn, cs;
    if (s == 0) {
      sincos_fast(*pipe_row(c, c.rq + r.qoff, st), &sn, &cs);
    } else {  // the I wavefront (idle in this pass) had it ready one step ahead
      const T* mq = pipe_ring(c.ringQ, 2, s & 1, lane);
      sn = mq[0]; cs = mq[64];
    }
    walk_compose<T, false>(W.R, W.p, rr, r.flags, sn, cs, T(0), Rn, pn);
    RBD_WALK_SWITCH(s, { St.template put<SV, WS_SN>(sn); St.template put<SV, WS_CS>(cs); })
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) W.R[k] = Rn[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) W.p[k] = pn[k];
  T* m = pipe_ring(c.ringK, PRK_N, slot, lane);
  pipe_put<T, 9>(m, 64, W.R);
  pipe_put<T, 3>(m + 9 * 64, 64, W.p);
  if (r.a_w >= 0) { T* b = pipe_box(c.boxAR, 12, r.a_w, st); pipe_put<T, 9>(b, PIPE_STATES, W.R); pipe_put<T, 3>(b + 9 * PIPE_STATES, PIPE_STATES, W.p); }
  if (r.park >= 0) { T* b = pipe_box(c.parkK, 12, r.park, st); pipe_put<T, 9>(b, PIPE_STATES, W.R); pipe_put<T, 3>(b + 9 * PIPE_STATES, PIPE_STATES, W.p); }
}
// I, for step s (one step AHEAD of K): sin q, cos q of the joint -> ring (nothing in the chain depends on them being computed late)
template <typename T> RBD_HD void pipe_a_i(const PipeCtx<T>& c, int s, const WalkRec& r, int lane) {
  const int track = lane & 3, st = lane >> 2;
  if (!(r.flags & TF_VALID) || (r.flags & TF_FLOATING)) return;
  T sn, cs;
  sincos_fast(*pipe_row(c, c.rq + r.qoff, st), &sn, &cs);
  T* mq = pipe_ring(c.ringQ, 2, s & 1, lane);
  mq[0] = sn; mq[64] = cs;
}
// T, step s (one step behind K): twist and velocity-product acceleration (mechanism_state.jl:769-780, :814-830); q̇ over the q rows
template <typename T> RBD_HD void pipe_a_t(const PipeCtx<T>& c, PipeT<T>& W, int s, int slot, const WalkRec& r, int lane, bool want_qdot) {
  const int track = lane & 3, st = lane >> 2;
  if (!(r.flags & TF_VALID)) return;
  if (!(r.flags & TF_CHAINED)) {
    if (r.flags & TF_LEVEL0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { W.Tw[k] = T(0); W.av[k] = c.a0[k]; }
    } else {
      const T* m = pipe_box(c.boxAT, 12, r.a_r, st);
      pipe_get<T, 6>(m, PIPE_STATES, W.Tw);
      pipe_get<T, 6>(m + 6 * PIPE_STATES, PIPE_STATES, W.av);
    }
  }
  T R[9], p[3], vJ[6], cb[6];
  const T* m = pipe_ring(c.ringK, PRK_N, slot, lane);
  pipe_get<T, 9>(m, 64, R);
  pipe_get<T, 3>(m + 9 * 64, 64, p);
  if (r.flags & TF_FLOATING) {
    T q7[7], v6[6];
    pipe_get<T, 6>(pipe_row(c, c.rv + r.voff, st), PR_STRIDE, v6);
    xmotion(R, p, v6, vJ);  // twist of the joint: X(H) v, v the body-frame twist
    if (want_qdot) {  // velocity_to_configuration_derivative! (quaternion_floating.jl:126-136, spatial/util.jl:127-134), over the q rows
      pipe_get<T, 7>(pipe_row(c, c.rq + r.qoff, st), PR_STRIDE, q7);
      const T qw = q7[0], qx = q7[1], qy = q7[2], qz = q7[3];
      T o[7], Rq[9];
      o[0] = (-qx * v6[0] - qy * v6[1] - qz * v6[2]) / 2;
      o[1] = (qw * v6[0] - qz * v6[1] + qy * v6[2]) / 2;
      o[2] = (qz * v6[0] + qw * v6[1] - qx * v6[2]) / 2;
      o[3] = (-qy * v6[0] + qx * v6[1] + qw * v6[2]) / 2;
      rot_quat(qw, qx, qy, qz, Rq);
      matvec3(Rq, v6 + 3, o + 4);
      pipe_put<T, 7>(pipe_row(c, c.rq + r.qoff, st), PR_STRIDE, o);
    }
  } else {
    T S[6];
    walk_subspace<T, false>(R, p, r.flags, S);
    const T qd = *pipe_row(c, c.rv + r.voff, st);
    if (want_qdot) *pipe_row(c, c.rq + r.qoff, st) = qd;
#pragma unroll
    for (int k = 0; k < 6; ++k) vJ[k] = S[k] * qd;
  }
  se3_comm(W.Tw, vJ, cb);  // [T_parent, vJ]: the velocity-product acceleration increment
#pragma unroll
  for (int k = 0; k < 6; ++k) { W.av[k] += cb[k]; W.Tw[k] += vJ[k]; }
  if (r.a_w >= 0) { T* b = pipe_box(c.boxAT, 12, r.a_w, st); pipe_put<T, 6>(b, PIPE_STATES, W.Tw); pipe_put<T, 6>(b + 6 * PIPE_STATES, PIPE_STATES, W.av); }
  if (r.park >= 0) { T* b = pipe_box(c.parkT, 12, r.park, st); pipe_put<T, 6>(b, PIPE_STATES, W.Tw); pipe_put<T, 6>(b + 6 * PIPE_STATES, PIPE_STATES, W.av); }
}

// ---------------- pass B ----------------
// K, step s: publish the body's transform, then un-compose the joint towards the parent (H_parent = H X_joint⁻¹)
template <typename T> RBD_HD void pipe_b_k(const PipeCtx<T>& c, PipeK<T>& W, WalkStash<T>& St, int s, int slot, const WalkRec& r, int lane) {
  const int track = lane & 3, st = lane >> 2;
  if (!(r.flags & TF_VALID)) return;
  if (r.park >= 0) { const T* b = pipe_box(c.parkK, 12, r.park, st); pipe_get<T, 9>(b, PIPE_STATES, W.R); pipe_get<T, 3>(b + 9 * PIPE_STATES, PIPE_STATES, W.p); }
  T* m = pipe_ring(c.ringK, PRK_N, slot, lane);
  pipe_put<T, 9>(m, 64, W.R);
  pipe_put<T, 3>(m + 9 * 64, 64, W.p);
  if (r.flags & TF_CHAINED) {
    T rr[TR_J], sn = T(0), cs = T(1);
    pipe_consts<T, 0, TR_J>(c, s, track, rr);
    RBD_WALK_SWITCH(s, { sn = St.template get<SV, WS_SN>(); cs = St.template get<SV, WS_CS>(); })
    T Mx[9], u3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Mx[3 * i] = cs * W.R[3 * i] - sn * W.R[3 * i + 1];
      Mx[3 * i + 1] = sn * W.R[3 * i] + cs * W.R[3 * i + 1];
      Mx[3 * i + 2] = W.R[3 * i + 2];
    }
    const T* C = rr + TR_C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) W.R[3 * i + j] = Mx[3 * i] * C[3 * j] + Mx[3 * i + 1] * C[3 * j + 1] + Mx[3 * i + 2] * C[3 * j + 2];  // Mx C'
    matvec3(W.R, rr + TR_PP, u3);
#pragma unroll
    for (int k = 0; k < 3; ++k) W.p[k] -= u3[k];
  }
}
// I, step s (one behind K): the body's spatial inertia in the root frame (mechanism_state.jl:836-846)
template <typename T> RBD_HD void pipe_b_i(const PipeCtx<T>& c, int s, int slot_k, int slot_i, const WalkRec& r, int lane) {
  const int track = lane & 3;
  if (!(r.flags & TF_VALID)) return;
  T rr[10], R[9], p[3];
  pipe_consts<T, TR_J, 10>(c, s, track, rr);  // J (6) | m c (3) | m
  const T* m = pipe_ring(c.ringK, PRK_N, slot_k, lane);
  pipe_get<T, 9>(m, 64, R);
  pipe_get<T, 3>(m + 9 * 64, 64, p);
  RInertia<T> I;
  inertia_to_root(rr, rr + 6, rr[9], R, p, I);
  T* o = pipe_ring(c.ringI, PRI_N, slot_i, lane);
  pipe_put<T, 6>(o, 64, I.J);
  pipe_put<T, 3>(o + 6 * 64, 64, I.c);
  o[9 * 64] = I.m;
}
// T, step s (two behind K): p̃A = I a_vp + T ×* I T − w_ext (newton_euler, mechanism_state.jl:872-876), the motion subspace, then the
// twist chain un-composed towards the parent (T_parent = T − S q̇, a_parent = a − [T, S q̇]).  fe: the body's external wrench.
template <typename T> RBD_HD void pipe_b_t(const PipeCtx<T>& c, PipeT<T>& W, int s, int slot_k, int slot_i, int slot_t, const WalkRec& r, int lane, const T* fe) {
  const int track = lane & 3, st = lane >> 2;
  if (!(r.flags & TF_VALID)) return;
  if (r.park >= 0) { const T* b = pipe_box(c.parkT, 12, r.park, st); pipe_get<T, 6>(b, PIPE_STATES, W.Tw); pipe_get<T, 6>(b + 6 * PIPE_STATES, PIPE_STATES, W.av); }
  RInertia<T> I;
  const T* mi = pipe_ring(c.ringI, PRI_N, slot_i, lane);
  pipe_get<T, 6>(mi, 64, I.J);
  pipe_get<T, 3>(mi + 6 * 64, 64, I.c);
  I.m = mi[9 * 64];
  T R[9], p[3], pA[6], h[6];
  const T* mk = pipe_ring(c.ringK, PRK_N, slot_k, lane);
  pipe_get<T, 9>(mk, 64, R);
  pipe_get<T, 3>(mk + 9 * 64, 64, p);
  mul_inertia(I, W.av, pA);
  momentum_cross(I, W.Tw, h);
#pragma unroll
  for (int k = 0; k < 6; ++k) pA[k] += h[k] - fe[k];
  T* o = pipe_ring(c.ringT, PRT_N, slot_t, lane);
  pipe_put<T, 6>(o + PRT_PA * 64, 64, pA);
  pipe_put<T, 6>(o + PRT_I * 64, 64, I.J);
  pipe_put<T, 3>(o + (PRT_I + 6) * 64, 64, I.c);
  o[(PRT_I + 9) * 64] = I.m;
  if (r.flags & TF_FLOATING) {  // the joint's force seen from the root: S⁻ᵀτ (S = X(H))
    T t6[6], f6[6];
    pipe_get<T, 6>(pipe_row(c, c.rt + r.voff, st), PR_STRIDE, t6);
    xforce(R, p, t6, f6);
    pipe_put<T, 6>(o + PRT_S * 64, 64, f6);
    pipe_put<T, 9>(o + PRT_RP * 64, 64, R);
    pipe_put<T, 3>(o + (PRT_RP + 9) * 64, 64, p);
    return;
  }
  T S[6];
  walk_subspace<T, false>(R, p, r.flags, S);
  pipe_put<T, 6>(o + PRT_S * 64, 64, S);
  if (r.flags & TF_CHAINED) {
    const T qd = *pipe_row(c, c.rv + r.voff, st);
    T vJ[6], cb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) vJ[k] = S[k] * qd;
    se3_comm(W.Tw, vJ, cb);
#pragma unroll
    for (int k = 0; k < 6; ++k) { W.av[k] -= cb[k]; W.Tw[k] -= vJ[k]; }
  }
}
// S, step s (three behind K): the articulated-body step.  IA = I + Σ children's Ia; U = IA S, D = S'U, u = τ − S'p̃A;
// hand-off Ia = IA − U D⁻¹ U', p̃a = p̃A + U D⁻¹ u.  A 6-dof root: IA a_Δ = S⁻ᵀτ − p̃A, v̇ = S⁻¹ a_Δ.
template <typename T> RBD_HD void pipe_b_s(const PipeCtx<T>& c, PipeS<T>& W, WalkStash<T>& St, int s, int slot_t, const WalkRec& r, int lane) {
  const int track = lane & 3, st = lane >> 2;
  if (!(r.flags & TF_VALID)) return;
  const T* m = pipe_ring(c.ringT, PRT_N, slot_t, lane);
  T IA[21], pA[6], S[6];
  {
    RInertia<T> I;
    pipe_get<T, 6>(m + PRT_I * 64, 64, I.J);
    pipe_get<T, 3>(m + (PRT_I + 6) * 64, 64, I.c);
    I.m = m[(PRT_I + 9) * 64];
    sym6_from_inertia(I, IA);
  }
  pipe_get<T, 6>(m + PRT_PA * 64, 64, pA);
  pipe_get<T, 6>(m + PRT_S * 64, 64, S);
  if (!(r.flags & TF_FLOATING)) pipe_put<T, 6>(c.sstore + ((long)s * 6) * 64 + lane, 64, S);  // pass C needs it again
#pragma unroll
  for (int k = 0; k < 21; ++k) IA[k] += W.cI[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) pA[k] += W.cP[k];
  for (int j = 0; j < r.nbr; ++j) {  // hand-offs of the children that finished on other tracks (or earlier on this one)
    const T* b = pipe_box(c.boxB, 27, r.b_r0 + j, st);
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] += b[k * PIPE_STATES];
#pragma unroll
    for (int k = 0; k < 6; ++k) pA[k] += b[(21 + k) * PIPE_STATES];
  }
  if (r.flags & TF_FLOATING) {
    T f6[6], a[6], vd[6], R[9], p[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) f6[k] = S[k] - pA[k];  // the S slot of the ring holds S⁻ᵀτ for this body
    sym6_solve(IA, f6, a);
    pipe_get<T, 9>(m + PRT_RP * 64, 64, R);
    pipe_get<T, 3>(m + (PRT_RP + 9) * 64, 64, p);
    xmotion_inv(R, p, a, vd);
    pipe_put<T, 6>(pipe_row(c, c.rt + r.voff, st), PR_STRIDE, vd);
    RBD_WALK_SWITCH(s, {
      St.template put<SV, WS_W + 0>(a[0]); St.template put<SV, WS_W + 1>(a[1]); St.template put<SV, WS_W + 2>(a[2]);
      St.template put<SV, WS_W + 3>(a[3]); St.template put<SV, WS_W + 4>(a[4]); St.template put<SV, WS_W + 5>(a[5]);
    })
  } else {
    T U[6], Wd[6];
    sym6_mul(IA, S, U);
    const T Dinv = rcp_hd(dot6(S, U));
    const T tq = *pipe_row(c, c.rt + r.voff, st);
    const T ud = (tq - dot6(S, pA)) * Dinv;
#pragma unroll
    for (int k = 0; k < 6; ++k) Wd[k] = U[k] * Dinv;
    RBD_WALK_SWITCH(s, {
      St.template put<SV, WS_W + 0>(Wd[0]); St.template put<SV, WS_W + 1>(Wd[1]); St.template put<SV, WS_W + 2>(Wd[2]);
      St.template put<SV, WS_W + 3>(Wd[3]); St.template put<SV, WS_W + 4>(Wd[4]); St.template put<SV, WS_W + 5>(Wd[5]);
      St.template put<SV, WS_UD>(ud);
    })
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) W.cI[SI(i, j)] = IA[SI(i, j)] - Wd[i] * U[j];
#pragma unroll
    for (int k = 0; k < 6; ++k) W.cP[k] = pA[k] + U[k] * ud;
  }
  if (r.b_w >= 0 || (r.flags & TF_LEVEL0)) {  // the parent is not next on this track (or is the world): the hand-off leaves the registers
    if (r.b_w >= 0) {
      T* b = pipe_box(c.boxB, 27, r.b_w, st);
      pipe_put<T, 21>(b, PIPE_STATES, W.cI);
      pipe_put<T, 6>(b + 21 * PIPE_STATES, PIPE_STATES, W.cP);
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) W.cI[k] = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) W.cP[k] = T(0);
  }
}

// ---------------- pass C ----------------
// S alone (the motion subspaces were kept by pass B; the tracks' edges are lanes of this one wavefront: no barrier inside the pass):
// v̇ = D⁻¹u − (U D⁻¹)' a_Δ,parent; a_Δ += S v̇
template <typename T> RBD_HD void pipe_c_s(const PipeCtx<T>& c, PipeS<T>& W, WalkStash<T>& St, int s, const WalkRec& r, int lane) {
  const int track = lane & 3, st = lane >> 2;
  if (!(r.flags & TF_VALID)) return;
  if (!(r.flags & TF_CHAINED)) {
    if (r.flags & TF_LEVEL0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) W.ad[k] = T(0);
    } else {
      pipe_get<T, 6>(pipe_box(c.boxC, 6, r.a_r, st), PIPE_STATES, W.ad);
    }
  }
  if (r.flags & TF_FLOATING) {  // a_Δ was solved for in pass B (v̇ is already in its rows)
    RBD_WALK_SWITCH(s, {
      W.ad[0] = St.template get<SV, WS_W + 0>(); W.ad[1] = St.template get<SV, WS_W + 1>(); W.ad[2] = St.template get<SV, WS_W + 2>();
      W.ad[3] = St.template get<SV, WS_W + 3>(); W.ad[4] = St.template get<SV, WS_W + 4>(); W.ad[5] = St.template get<SV, WS_W + 5>();
    })
  } else {
    T S[6], Wd[6], ud = T(0);
    pipe_get<T, 6>(c.sstore + ((long)s * 6) * 64 + lane, 64, S);
    RBD_WALK_SWITCH(s, {
      ud = St.template get<SV, WS_UD>();
      Wd[0] = St.template get<SV, WS_W + 0>(); Wd[1] = St.template get<SV, WS_W + 1>(); Wd[2] = St.template get<SV, WS_W + 2>();
      Wd[3] = St.template get<SV, WS_W + 3>(); Wd[4] = St.template get<SV, WS_W + 4>(); Wd[5] = St.template get<SV, WS_W + 5>();
    })
    const T vd = ud - dot6(Wd, W.ad);
#pragma unroll
    for (int k = 0; k < 6; ++k) W.ad[k] += S[k] * vd;
    *pipe_row(c, c.rt + r.voff, st) = vd;
  }
  if (r.a_w >= 0) pipe_put<T, 6>(pipe_box(c.boxC, 6, r.a_w, st), PIPE_STATES, W.ad);
}

#if defined(__HIPCC__)
#ifdef RBD_PROFILE_PHASES
__device__ long long rbd_pipe_phase_clock[32];  // [wave][mark]
#define RBD_PMARK(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) rbd_pipe_phase_clock[(threadIdx.x >> 6) * 8 + (i)] = clock64(); } while (0)
#else
#define RBD_PMARK(i)
#endif
// Staging.  q, v, τ of the workgroup's 16 states -> rows; the plan's constants and records -> LDS.  EVERY load of the prologue is in flight
// before the first LDS write (a lone workgroup pays each global round trip in full: loop by loop this took 7.3 k cycles of the launch).
template <typename T> struct PipeStageIn {
  enum { UQ = 3, UC = 5, UR = 3 };  // 256 threads: 16 n <= 768 coordinates per array, <= 1280 constants, <= 768 record words in one batch
  T a[3][UQ], cr[UC];
  int32_t rw[UR];
  __device__ __forceinline__ static T ld(const T* __restrict__ src, Layout L, long state0, long B, int n, int e) {
    const int st = e / n, k = e - st * n;
    long gs = state0 + st;
    gs = gs < B ? gs : B - 1;
    return src[gs * L.sb + (long)k * L.sk];
  }
  __device__ __forceinline__ void load(const T* q, const T* v, const T* tau, Layout Lq, Layout Lv, long state0, long B, int nq, int nv, const T* gr, int ncr, const int32_t* gw,
                                       int nrw, int tid) {
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const int e = tid + u * 256;
      a[0][u] = e < nq * PIPE_STATES ? ld(q, Lq, state0, B, nq, e) : T(0);
      a[1][u] = (v && e < nv * PIPE_STATES) ? ld(v, Lv, state0, B, nv, e) : T(0);
      a[2][u] = (tau && e < nv * PIPE_STATES) ? ld(tau, Lv, state0, B, nv, e) : T(0);
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) cr[u] = tid + u * 256 < ncr ? gr[tid + u * 256] : T(0);
#pragma unroll
    for (int u = 0; u < UR; ++u) rw[u] = tid + u * 256 < nrw ? gw[tid + u * 256] : 0;
  }
  __device__ __forceinline__ void store(const PipeCtx<T>& c, int nq, int nv, int ncr, int nrw, int tid) const {
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const int e = tid + u * 256;
      if (e < nq * PIPE_STATES) { const int st = e / nq, k = e - st * nq; c.rows[(long)(c.rq + k) * PR_STRIDE + st] = a[0][u]; }
      if (e < nv * PIPE_STATES) { const int st = e / nv, k = e - st * nv; c.rows[(long)(c.rv + k) * PR_STRIDE + st] = a[1][u]; c.rows[(long)(c.rt + k) * PR_STRIDE + st] = a[2][u]; }
    }
#pragma unroll
    for (int u = 0; u < UC; ++u)
      if (tid + u * 256 < ncr) const_cast<T*>(c.trr)[tid + u * 256] = cr[u];
#pragma unroll
    for (int u = 0; u < UR; ++u)
      if (tid + u * 256 < nrw) const_cast<int32_t*>(c.rec)[tid + u * 256] = rw[u];
  }
};
// whatever is beyond the first batch (more than 48 coordinates, more than 13 steps' worth of constants)
template <typename T> __device__ __forceinline__ void pipe_stage_in_rest(const T* __restrict__ src, Layout L, long state0, long B, int n, T* rows, int row0, int tid, int e0) {
  for (int e = e0 + tid; e < n * PIPE_STATES; e += 256) {
    const int st = e / n, k = e - st * n;
    long gs = state0 + st;
    gs = gs < B ? gs : B - 1;
    rows[(long)(row0 + k) * PR_STRIDE + st] = src ? src[gs * L.sb + (long)k * L.sk] : T(0);
  }
}
template <typename T> __device__ __forceinline__ void pipe_stage_out(T* __restrict__ dst, Layout L, long state0, long B, int n, const T* rows, int row0, int tid, int nth) {
  if (!dst) return;
  for (int e = tid; e < n * PIPE_STATES; e += nth) {
    const int st = e / n, k = e - st * n;
    const long gs = state0 + st;
    if (gs < B) dst[gs * L.sb + (long)k * L.sk] = rows[(long)(row0 + k) * PR_STRIDE + st];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void aba_pipe_kernel(WalkModel M, long B, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ tau,
                                                      const T* __restrict__ fext, T* __restrict__ vdot, T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf) {
  extern __shared__ __align__(16) unsigned char pipe_lds_raw[];
  PipeCtx<T> c;
  c.M = M;
  pipe_ctx_lds(c, pipe_lds_raw);
  const int tid = threadIdx.x, nth = blockDim.x, lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long state0 = (long)blockIdx.x * PIPE_STATES;
  RBD_PMARK(0);
  {
    const int ncr = M.ns * 4 * TR_STRIDE, nrw = M.ns * 4 * WREC_STRIDE;
    const T* gr = reinterpret_cast<const T*>(M.rr);
    PipeStageIn<T> in;
    in.load(q, v, tau, Lq, Lv, state0, B, M.nq, M.nv, gr, ncr, M.wk, nrw, tid);
    in.store(c, M.nq, M.nv, ncr, nrw, tid);
    pipe_stage_in_rest(q, Lq, state0, B, M.nq, c.rows, c.rq, tid, PipeStageIn<T>::UQ * 256);
    pipe_stage_in_rest(v, Lv, state0, B, M.nv, c.rows, c.rv, tid, PipeStageIn<T>::UQ * 256);
    pipe_stage_in_rest(tau, Lv, state0, B, M.nv, c.rows, c.rt, tid, PipeStageIn<T>::UQ * 256);
    for (int i2 = PipeStageIn<T>::UC * 256 + tid; i2 < ncr; i2 += 256) const_cast<T*>(c.trr)[i2] = gr[i2];
    for (int i2 = PipeStageIn<T>::UR * 256 + tid; i2 < nrw; i2 += 256) const_cast<int32_t*>(c.rec)[i2] = M.wk[i2];
  }
  __syncthreads();
  RBD_PMARK(1);
  asm volatile("" ::: "a255");  // the kernel descriptor covers every accumulation register (WalkStash addresses them by number)
  const int ns = M.ns;
  const bool want_qdot = qdot != nullptr;
  // One loop nest per role: a wavefront carries only its own chain registers (with the four roles in one loop the compiler kept the
  // union of their state alive and shuffled it between VGPRs and AGPRs on every iteration: ~500 instructions per iteration per wavefront).
  // Every role passes the same number of barriers: (ns + 1) + (ns + 3) + 1.
  // Ring slots, pass B (iteration j, K at step ns − 1 − j, I one step behind, T two, S three): K writes j mod 3; I reads (j + 2) mod 3 and
  // writes j mod 2; T reads K's (j + 1) mod 3 and I's (j + 1) mod 2 and writes j mod 2; S reads (j + 1) mod 2.  Pass A: slot = step mod 2.
  const int track = lane & 3;
  // the record of a stage's NEXT step is read before the barrier that ends the current one (it depends on nothing the other stages produce)
  auto rec_at = [&](int s) { return pipe_rec(c, s < 0 ? 0 : (s >= ns ? ns - 1 : s), track); };
  if (role == PIPE_K) {
    WalkStash<T> St;
    PipeK<T> W;
    pipe_identity(W);
    WalkRec r = rec_at(0);
#pragma unroll 1
    for (int i = 0; i <= ns; ++i) {
      if (i < ns) pipe_a_k(c, W, St, i, i & 1, r, lane);
      r = rec_at(i + 1 < ns ? i + 1 : ns - 1);  // after the last step: the first record of pass B
      __syncthreads();
    }
    RBD_PMARK(2);
#pragma unroll 1
    for (int j = 0; j < ns + 3; ++j) {
      const int s = ns - 1 - j;
      if (s >= 0) pipe_b_k(c, W, St, s, j % 3, r, lane);
      r = rec_at(s - 1);
      __syncthreads();
    }
    RBD_PMARK(3);
  } else if (role == PIPE_I) {
    WalkRec r = rec_at(1);
#pragma unroll 1
    for (int i = 0; i <= ns; ++i) {
      if (i + 1 < ns) pipe_a_i(c, i + 1, r, lane);
      r = rec_at(i + 2 < ns ? i + 2 : ns - 1);
      __syncthreads();
    }
    RBD_PMARK(2);
#pragma unroll 1
    for (int j = 0; j < ns + 3; ++j) {
      const int s = ns - j;
      if (s >= 0 && s < ns) pipe_b_i(c, s, (j + 2) % 3, j & 1, r, lane);
      r = rec_at(s - 1);
      __syncthreads();
    }
    RBD_PMARK(3);
  } else if (role == PIPE_T) {
    PipeT<T> W;
#pragma unroll
    for (int k = 0; k < 6; ++k) { W.Tw[k] = T(0); W.av[k] = T(0); }
    WalkRec r = rec_at(0);
#pragma unroll 1
    for (int i = 0; i <= ns; ++i) {
      if (i >= 1) pipe_a_t(c, W, i - 1, (i - 1) & 1, r, lane, want_qdot);
      r = rec_at(i < ns ? i : ns - 1);
      __syncthreads();
    }
    RBD_PMARK(2);
    // the external wrench of a body is asked for (global memory) one step before its turn
    long gs = state0 + (lane >> 2);
    gs = gs < B ? gs : B - 1;
    const T* fel = fext ? fext + gs * Lf.sb : nullptr;
    T fe[6], fn[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { fe[k] = T(0); fn[k] = T(0); }
    if (fext) {
#pragma unroll
      for (int k = 0; k < 6; ++k) fe[k] = fel[(long)(r.orig6 + k) * Lf.sk];  // r: the record of step ns − 1
    }
#pragma unroll 1
    for (int j = 0; j < ns + 3; ++j) {
      const int s = ns + 1 - j;
      const WalkRec rn = rec_at(s >= ns ? ns - 1 : s - 1);
      if (s >= 0 && s < ns) {
        if (fext && s > 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) fn[k] = fel[(long)(rn.orig6 + k) * Lf.sk];
        }
        pipe_b_t(c, W, s, (j + 1) % 3, (j + 1) & 1, j & 1, r, lane, fe);
#pragma unroll
        for (int k = 0; k < 6; ++k) fe[k] = fn[k];
      }
      r = rn;
      __syncthreads();
    }
    RBD_PMARK(3);
  } else {
    WalkStash<T> St;
    PipeS<T> W;
#pragma unroll
    for (int k = 0; k < 6; ++k) { W.cP[k] = T(0); W.ad[k] = T(0); }
#pragma unroll
    for (int k = 0; k < 21; ++k) W.cI[k] = T(0);
#pragma unroll 1
    for (int i = 0; i <= ns; ++i) __syncthreads();
    RBD_PMARK(2);
    WalkRec r = rec_at(ns - 1);
#pragma unroll 1
    for (int j = 0; j < ns + 3; ++j) {
      const int s = ns + 2 - j;
      const WalkRec rn = rec_at(s >= ns ? ns - 1 : s - 1);
      if (s >= 0 && s < ns) pipe_b_s(c, W, St, s, (j + 1) & 1, r, lane);
      r = rn;
      __syncthreads();
    }
    RBD_PMARK(3);
    // pass C: this wavefront alone; the record and the motion subspace of the next step are in flight while the current one computes
    r = rec_at(0);
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
      const WalkRec rn = rec_at(s + 1);
      pipe_c_s(c, W, St, s, r, lane);
      r = rn;
    }
  }
  __syncthreads();
  RBD_PMARK(4);
  pipe_stage_out(vdot, Lv, state0, B, M.nv, c.rows, c.rt, tid, nth);
  pipe_stage_out(qdot, Lq, state0, B, M.nq, c.rows, c.rq, tid, nth);
  RBD_PMARK(5);
}
#endif

}  // namespace rbd
