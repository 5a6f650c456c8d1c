// rbd_mk_fuse.hpp — one stage of `simulate`'s Munthe-Kaas RK4 step (src/ode_integrators.jl:233-299 with the runge_kutta_4 tableau :48-55;
// src/simulate.jl:42-48) folded into the LARGE-BATCH dynamics! kernels that are compiled for the mechanism: those kernels stage q, v and τ of their states
// through LDS rows anyway, so the integrator's arithmetic runs on the rows — no launch of its own, no second trip of q and v through HBM.  This file: the
// stage struct both kernels take, and the ELEMENT-PARALLEL form of the arithmetic that aba_walk_spec (rbd_walk.hpp) uses — a workgroup's threads share the
// (joint, state) cells.  aba_spec (rbd_spec.hpp), where a lane is a state, has the same arithmetic as straight-line per-lane code of its own.  Round 3 left the stage bookkeeping of large batches in its own launches (≈ 5 per step, each streaming
// the eight stage buffers: a third of the 605 µs step at 65 536 fp64 states).
//
// The classical tableau has ONE non-zero per row (a21 = a32 = ½, a43 = 1), so the state of stage s + 1 needs only the base point of the step and the
// slopes of stage s; the closing combination needs Σ b_j k_j, kept as a running sum.  Stage s = 0..3 is therefore ONE launch of the dynamics kernel:
//   prologue (after the rows are staged; (q, v) = the state of stage s):
//     φ̇_s = local-coordinate rates of stage s (a 1-dof joint: its velocity; a 6-dof joint: log_with_time_derivative of inv(H0) H — joint_local_rate)
//     s = 0: (q0, v0) := (q, v);  accφ := b_0 φ̇_0            s = 1, 2: accφ += b_s φ̇_s
//     q_state := global_coordinates(q0, h a_{s+1} φ̇_s)        s = 3: q_state := global_coordinates(q0, h (accφ + b_3 φ̇_3))   — the end of the step
//     (the PD law of rbd_simulate_controlled on the stage state, into the τ rows)
//   the three passes of the articulated-body algorithm on the rows (unchanged) → v̇_s
//   epilogue:  s = 0: accv := b_0 v̇_0;  s = 1, 2: accv += b_s v̇_s;  v_state := v0 + h a_{s+1} v̇_s   (s = 3: v0 + h (accv + b_3 v̇_3))
// q_state / v_state are the kernel's own q / v inputs: a workgroup has read its states' rows before it overwrites them, and no other workgroup touches them.
// Four launches per step, nothing else.  Scope: what the compiled kernels take (revolute / prismatic / sin-cos / fixed joints, 6-dof joints on the world).
// mk1[3 i ..] = (q offset, v offset, joint type) of the N1 1-dof joints; mkf[2 f ..] = (q offset, v offset) of the NF 6-dof ones (tables in device memory).
#pragma once
#include "rbd_lane.hpp"  // (brings rbd_integrator.hpp: joint_local_rate, joint_global)

namespace rbd {

struct MkStage {
  int32_t stage;  // 0..3, or -1: plain dynamics!
  int32_t pd;     // 1: τ −= kp (q − q_des) + kd v on the revolute / prismatic joints, on the stage state
  double dt;
  void* q0; void* v0;      // base point of the step (layout of q / v)
  void* accp; void* accv;  // Σ b_j φ̇_j, Σ b_j v̇_j over the stages so far (layout of v)
  void* q_state; void* v_state;
  const void* kp; const void* kd; const void* qdes;
};

// The global loads of the stage (base point, running sums) depend on nothing the kernel has staged, so they are issued BEFORE the barrier that ends the staging
// of the rows (mk_pre_load / mk_post_load ahead of the last barrier of the passes) and consumed after it: a workgroup of the walk kernels is alone on its compute
// unit, nothing else would hide their round trip.  UB = elements per thread (the caller's compile-time ceiling: one batch).
// (only the loaded values are carried across the barrier — the indices are recomputed behind it: the walk kernels live within 248 registers)
template <typename S, int UB> struct MkPre {
  S q0a[UB], acc[UB];
  S fq0[7], facc[6];  // a 6-dof joint (one per thread: the first round of the loop below)
};
template <int N1> RBD_DEV void mk_elem(bool aos, int e, int nstates, int& i, int& st) {
  if (aos) { st = e / (N1 > 0 ? N1 : 1); i = e - st * N1; } else { i = e / nstates; st = e - i * nstates; }
}
template <typename S, int N1, int NF, int UB>
RBD_DEV void mk_pre_load(const MkStage& F, MkPre<S, UB>& P, const int32_t* mk1, const int32_t* mkf, long state0, long B, int nstates, Layout Lq, Layout Lv, int tid, int nth) {
  const int s = F.stage;
  const S* __restrict__ q0 = (const S*)F.q0; const S* __restrict__ accp = (const S*)F.accp;
  const bool aos = Lq.sk == 1;  // consecutive threads on consecutive coordinates of a state, or on consecutive states of a coordinate: coalesced either way
  const int tot = N1 * nstates;
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int e = tid + u * nth;
    int i, st;
    mk_elem<N1>(aos, e, nstates, i, st);
    P.q0a[u] = S(0); P.acc[u] = S(0);
    if (N1 > 0 && s > 0 && e < tot && state0 + st < B) {
      P.q0a[u] = q0[(long)mk1[3 * i] * Lq.sk + (state0 + st) * Lq.sb];
      P.acc[u] = accp[(long)mk1[3 * i + 1] * Lv.sk + (state0 + st) * Lv.sb];
    }
  }
  if constexpr (NF > 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) P.fq0[k] = S(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) P.facc[k] = S(0);
    if (tid < NF * nstates && s > 0) {
      const int f = tid / nstates, st = tid - f * nstates;
      if (state0 + st < B) {
        const long aq = (long)mkf[2 * f] * Lq.sk + (state0 + st) * Lq.sb, av = (long)mkf[2 * f + 1] * Lv.sk + (state0 + st) * Lv.sb;
#pragma unroll
        for (int k = 0; k < 7; ++k) P.fq0[k] = q0[aq + k * Lq.sk];
#pragma unroll
        for (int k = 0; k < 6; ++k) P.facc[k] = accp[av + k * Lv.sk];
      }
    }
  }
}

// CELL: (row, state of the group) -> pointer to the scalar in the LDS rows.  mkf: the 6-dof joints' table (device memory; the 1-dof joints' entries are in P).
template <typename S, int N1, int NF, int UB, typename CELL>
RBD_DEV void mk_prologue(const MkStage& F, const MkPre<S, UB>& P, CELL cell, const int32_t* mk1, const int32_t* mkf, int rq, int rv, int rt, long state0, long B,
                         int nstates, Layout Lq, Layout Lv, int tid, int nth) {
  const int s = F.stage;
  const S h = (S)F.dt;
  const S bs = (s == 0 || s == 3) ? S(1) / S(6) : S(1) / S(3);
  const S an = s < 2 ? S(0.5) : S(1);
  S* __restrict__ q0 = (S*)F.q0; S* __restrict__ v0 = (S*)F.v0; S* __restrict__ accp = (S*)F.accp; S* qs = (S*)F.q_state;
  const S* kp = (const S*)F.kp; const S* kd = (const S*)F.kd;
  if constexpr (N1 > 0) {
    const bool aos = Lq.sk == 1;
    const S* qdes = (const S*)F.qdes;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = tid + u * nth;
      int i, st;
      mk_elem<N1>(aos, e, nstates, i, st);
      if (!(e < N1 * nstates && state0 + st < B)) continue;
      const int qo = mk1[3 * i], vo = mk1[3 * i + 1], ty = mk1[3 * i + 2];
      const long aq = (long)qo * Lq.sk + (state0 + st) * Lq.sb, av = (long)vo * Lv.sk + (state0 + st) * Lv.sb;
      const bool sc = ty == RBD_JOINT_SINCOS_REVOLUTE;
      const S rate = *cell(rv + vo, st);  // a 1-dof joint's local-coordinate rate is its velocity (joint_types.jl:9-18, sin_cos_revolute.jl:173-196)
      const S qa = *cell(rq + qo, st);
      const S qb = sc ? *cell(rq + qo + 1, st) : S(0);
      const S q0a = s == 0 ? qa : P.q0a[u];
      const S q0b = !sc ? S(0) : s == 0 ? qb : q0[aq + Lq.sk];
      if (s == 0) { q0[aq] = qa; v0[av] = rate; if (sc) q0[aq + Lq.sk] = qb; }
      const S sum = P.acc[u] + bs * rate;
      if (s < 3) accp[av] = sum;
      const S phi = s < 3 ? h * an * rate : h * sum;
      if (sc) {
        S sd, cd;
        sincos_t(phi, &sd, &cd);
        qs[aq] = q0a * cd + q0b * sd;
        qs[aq + Lq.sk] = q0b * cd - q0a * sd;
      } else {
        qs[aq] = q0a + phi;
      }
      if (F.pd && (ty == RBD_JOINT_REVOLUTE || ty == RBD_JOINT_PRISMATIC)) *cell(rt + vo, st) -= kp[vo] * (qa - (qdes ? qdes[aq] : S(0))) + kd[vo] * rate;
    }
  }
  if constexpr (NF > 0) {  // 6-dof joints on the world: one thread per (state, joint) — every lane busy (the fused lane-per-body kernels spend one lane in 16 on it)
    for (int e = tid; e < NF * nstates; e += nth) {
      const int f = e / nstates, st = e - f * nstates;
      if (state0 + st >= B) continue;
      const int qo = mkf[2 * f], vo = mkf[2 * f + 1];
      const long aq = (long)qo * Lq.sk + (state0 + st) * Lq.sb, av = (long)vo * Lv.sk + (state0 + st) * Lv.sb;
      S qj[7], vj[6], q0j[7], rate[6], phi[6], qn[7], acc[6];
#pragma unroll
      for (int k = 0; k < 7; ++k) { qj[k] = *cell(rq + qo + k, st); q0j[k] = qj[k]; }
#pragma unroll
      for (int k = 0; k < 6; ++k) { vj[k] = *cell(rv + vo + k, st); acc[k] = S(0); }
      if (s > 0) {
        if (e == tid) {  // (the first round was requested ahead of the barrier)
#pragma unroll
          for (int k = 0; k < 7; ++k) q0j[k] = P.fq0[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[k] = P.facc[k];
        } else {
#pragma unroll
          for (int k = 0; k < 7; ++k) q0j[k] = q0[aq + k * Lq.sk];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[k] = accp[av + k * Lv.sk];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) q0[aq + k * Lq.sk] = qj[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) v0[av + k * Lv.sk] = vj[k];
      }
      joint_local_rate<S, 2>(RBD_JOINT_QUAT_FLOATING, q0j, qj, vj, rate);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const S sum = acc[k] + bs * rate[k];
        if (s < 3) accp[av + k * Lv.sk] = sum;
        phi[k] = s < 3 ? h * an * rate[k] : h * sum;
      }
      joint_global<S, 2>(RBD_JOINT_QUAT_FLOATING, q0j, phi, qn);
#pragma unroll
      for (int k = 0; k < 7; ++k) qs[aq + k * Lq.sk] = qn[k];
    }
  }
}

template <typename S, int UB> struct MkPost { S vb[UB], acc[UB]; };
template <int NV> RBD_DEV void mk_velem(bool aos, int e, int nstates, int& k, int& st) {
  if (aos) { st = e / NV; k = e - st * NV; } else { k = e / nstates; st = e - k * nstates; }
}
template <typename S, int NV, int UB>
RBD_DEV void mk_post_load(const MkStage& F, MkPost<S, UB>& P, long state0, long B, int nstates, Layout Lv, int tid, int nth) {
  const int s = F.stage;
  const S* __restrict__ v0 = (const S*)F.v0; const S* __restrict__ accv = (const S*)F.accv; const S* vs = (const S*)F.v_state;
  const bool aos = Lv.sk == 1;
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int e = tid + u * nth;
    int k, st;
    mk_velem<NV>(aos, e, nstates, k, st);
    P.vb[u] = S(0); P.acc[u] = S(0);
    if (e < NV * nstates && state0 + st < B) {
      const long av = (long)k * Lv.sk + (state0 + st) * Lv.sb;
      // (stage 0: v_state still holds the base point — this thread reads it before it writes it; v0 is written by another thread of this launch)
      P.vb[u] = s == 0 ? vs[av] : v0[av];
      if (s > 0) P.acc[u] = accv[av];
    }
  }
}
template <typename S, int NV, int UB, typename CELL>
RBD_DEV void mk_epilogue(const MkStage& F, const MkPost<S, UB>& P, CELL cell, int rt, long state0, long B, int nstates, Layout Lv, int tid, int nth) {
  const int s = F.stage;
  const S h = (S)F.dt;
  const S bs = (s == 0 || s == 3) ? S(1) / S(6) : S(1) / S(3);
  const S an = s < 2 ? S(0.5) : S(1);
  S* __restrict__ accv = (S*)F.accv; S* vs = (S*)F.v_state;
  const bool aos = Lv.sk == 1;
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int e = tid + u * nth;
    int k, st;
    mk_velem<NV>(aos, e, nstates, k, st);
    if (!(e < NV * nstates && state0 + st < B)) continue;
    const long av = (long)k * Lv.sk + (state0 + st) * Lv.sb;
    const S vd = *cell(rt + k, st);
    const S sum = P.acc[u] + bs * vd;
    if (s < 3) accv[av] = sum;
    vs[av] = P.vb[u] + h * (s < 3 ? an * vd : sum);
  }
}

}  // namespace rbd
