// rbd_mk_fuse.hpp — one stage of `simulate`'s Munthe-Kaas RK4 step (src/ode_integrators.jl:233-299 with the runge_kutta_4 tableau :48-55;
// src/simulate.jl:42-48) folded into the LARGE-BATCH dynamics! kernels that are compiled for the mechanism (aba_walk_spec of rbd_walk.hpp, aba_spec of
// rbd_spec.hpp): those kernels stage q, v and τ of their states through LDS rows anyway, so the integrator's arithmetic runs on the rows — no launch of
// its own, no second trip of q and v through HBM.  Round 3 left the stage bookkeeping of large batches in its own launches (≈ 5 per step, each streaming
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

// CELL: (row, state of the group) -> pointer to the scalar in the LDS rows
template <typename S, int N1, int NF, typename CELL>
RBD_DEV void mk_prologue(const MkStage& F, CELL cell, const int32_t* __restrict__ mk1, const int32_t* __restrict__ mkf, int rq, int rv, int rt, long state0, long B,
                         int nstates, Layout Lq, Layout Lv, int tid, int nth) {
  const int s = F.stage;
  const S h = (S)F.dt;
  const S bs = (s == 0 || s == 3) ? S(1) / S(6) : S(1) / S(3);
  const S an = s < 2 ? S(0.5) : S(1);
  S* __restrict__ q0 = (S*)F.q0; S* __restrict__ v0 = (S*)F.v0; S* __restrict__ accp = (S*)F.accp; S* qs = (S*)F.q_state;
  const S* kp = (const S*)F.kp; const S* kd = (const S*)F.kd; const S* qdes = (const S*)F.qdes;
  const bool aos = Lq.sk == 1;  // consecutive threads on consecutive coordinates of a state, or on consecutive states of a coordinate: coalesced either way
  if constexpr (N1 > 0) {
    for (int e = tid; e < N1 * nstates; e += nth) {
      int i, st;
      if (aos) { st = e / N1; i = e - st * N1; } else { i = e / nstates; st = e - i * nstates; }
      if (state0 + st >= B) continue;
      const int qo = mk1[3 * i], vo = mk1[3 * i + 1], ty = mk1[3 * i + 2];
      const long aq = (long)qo * Lq.sk + (state0 + st) * Lq.sb, av = (long)vo * Lv.sk + (state0 + st) * Lv.sb;
      const S rate = *cell(rv + vo, st);  // a 1-dof joint's local-coordinate rate is its velocity (joint_types.jl:9-18, sin_cos_revolute.jl:173-196)
      const S qa = *cell(rq + qo, st);
      const bool sc = ty == RBD_JOINT_SINCOS_REVOLUTE;
      S q0a, q0b = S(0);
      if (s == 0) {
        q0a = qa; q0[aq] = qa; v0[av] = rate;
        if (sc) { q0b = *cell(rq + qo + 1, st); q0[aq + Lq.sk] = q0b; }
      } else {
        q0a = q0[aq];
        if (sc) q0b = q0[aq + Lq.sk];
      }
      S phi;
      if (s == 0) { accp[av] = bs * rate; phi = h * an * rate; }
      else if (s < 3) { accp[av] += bs * rate; phi = h * an * rate; }
      else phi = h * (accp[av] + bs * rate);
      if (sc) {
        S sd, cd;
        sincos_t(phi, &sd, &cd);
        qs[aq] = q0a * cd + q0b * sd;
        qs[aq + Lq.sk] = q0b * cd - q0a * sd;
      } else {
        qs[aq] = q0a + phi;
      }
      if (F.pd && (ty == RBD_JOINT_REVOLUTE || ty == RBD_JOINT_PRISMATIC)) {
        const S qd = qdes ? qdes[aq] : S(0);
        *cell(rt + vo, st) -= kp[vo] * (qa - qd) + kd[vo] * rate;
      }
    }
  }
  if constexpr (NF > 0) {  // 6-dof joints on the world: one thread per (state, joint) — every lane busy (the fused lane-per-body kernels spend one lane in 16 on it)
    for (int e = tid; e < NF * nstates; e += nth) {
      const int f = e / nstates, st = e - f * nstates;
      if (state0 + st >= B) continue;
      const int qo = mkf[2 * f], vo = mkf[2 * f + 1];
      const long aq = (long)qo * Lq.sk + (state0 + st) * Lq.sb, av = (long)vo * Lv.sk + (state0 + st) * Lv.sb;
      S qj[7], vj[6], q0j[7], rate[6], phi[6], qn[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) qj[k] = *cell(rq + qo + k, st);
#pragma unroll
      for (int k = 0; k < 6; ++k) vj[k] = *cell(rv + vo + k, st);
      if (s == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) { q0j[k] = qj[k]; q0[aq + k * Lq.sk] = qj[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) v0[av + k * Lv.sk] = vj[k];
      } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) q0j[k] = q0[aq + k * Lq.sk];
      }
      joint_local_rate<S, 2>(RBD_JOINT_QUAT_FLOATING, q0j, qj, vj, rate);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const long a = av + k * Lv.sk;
        if (s == 0) { accp[a] = bs * rate[k]; phi[k] = h * an * rate[k]; }
        else if (s < 3) { accp[a] += bs * rate[k]; phi[k] = h * an * rate[k]; }
        else phi[k] = h * (accp[a] + bs * rate[k]);
      }
      joint_global<S, 2>(RBD_JOINT_QUAT_FLOATING, q0j, phi, qn);
#pragma unroll
      for (int k = 0; k < 7; ++k) qs[aq + k * Lq.sk] = qn[k];
    }
  }
}

template <typename S, int NV, typename CELL>
RBD_DEV void mk_epilogue(const MkStage& F, CELL cell, int rt, long state0, long B, int nstates, Layout Lv, int tid, int nth) {
  const int s = F.stage;
  const S h = (S)F.dt;
  const S bs = (s == 0 || s == 3) ? S(1) / S(6) : S(1) / S(3);
  const S an = s < 2 ? S(0.5) : S(1);
  const S* __restrict__ v0 = (const S*)F.v0; S* __restrict__ accv = (S*)F.accv; S* vs = (S*)F.v_state;
  const bool aos = Lv.sk == 1;
  for (int e = tid; e < NV * nstates; e += nth) {
    int k, st;
    if (aos) { st = e / NV; k = e - st * NV; } else { k = e / nstates; st = e - k * nstates; }
    if (state0 + st >= B) continue;
    const long av = (long)k * Lv.sk + (state0 + st) * Lv.sb;
    const S vd = *cell(rt + k, st);
    // (stage 0: v_state still holds the base point — this thread reads it before it writes it; v0 was written by another thread of this launch)
    const S vb = s == 0 ? vs[av] : v0[av];
    S slope;
    if (s == 0) { accv[av] = bs * vd; slope = an * vd; }
    else if (s < 3) { accv[av] += bs * vd; slope = an * vd; }
    else slope = accv[av] + bs * vd;
    vs[av] = vb + h * slope;
  }
}

}  // namespace rbd
