// rbd_mk_fuse.hpp — one stage of `simulate`'s Munthe-Kaas RK4 step (src/ode_integrators.jl:233-299 with the runge_kutta_4 tableau :48-55;
// src/simulate.jl:42-48) folded into the LARGE-BATCH dynamics! kernels that are compiled for the mechanism: those kernels stage q, v and τ of their states
// through LDS rows anyway, so the integrator's arithmetic runs on the rows — no launch of its own, no second trip of q and v through HBM.  This file: the
// stage struct both kernels take, and the form of the arithmetic that aba_walk_spec (rbd_walk.hpp) uses; aba_spec (rbd_spec.hpp) has the same arithmetic as
// straight-line per-lane code of its own.  Round 3 left the stage bookkeeping of large batches in its own launches (≈ 5 per step, each streaming the eight
// stage buffers: a third of the 605 µs step at 65 536 fp64 states); round 4 folded it in as ELEMENT-PARALLEL code in front of and behind the passes (a
// workgroup's threads sharing the (joint, state) cells through a table: no gain in fp64, 158 µs per launch against 94 without); round 5: the lane is the state.
//
// The classical tableau has ONE non-zero per row (a21 = a32 = ½, a43 = 1), so the state of stage s + 1 needs only the base point of the step and the
// slopes of stage s; the closing combination needs Σ b_j k_j, kept as a running sum.  Stage s = 0..3 is therefore ONE launch of the dynamics kernel:
//   the three passes of the articulated-body algorithm on the rows (q, v) = the state of stage s → v̇_s, then
//     φ̇_s = local-coordinate rates of stage s (a 1-dof joint: its velocity; a 6-dof joint: log_with_time_derivative of inv(H0) H — joint_local_rate)
//     s = 0: (q0, v0) := (q, v);  accφ := b_0 φ̇_0, accv := b_0 v̇_0           s = 1, 2: accφ += b_s φ̇_s, accv += b_s v̇_s
//     q_state := global_coordinates(q0, h a_{s+1} φ̇_s), v_state := v0 + h a_{s+1} v̇_s      (s = 3: … h (accφ + b_3 φ̇_3), v0 + h (accv + b_3 v̇_3): the end of the step)
//   (the PD law of rbd_simulate_controlled on the stage state, into the τ rows, in front of the passes)
// q_state / v_state are the kernel's own q / v inputs: a workgroup has read its states' rows before it overwrites them, and no other workgroup touches them.
// Four launches per step, nothing else — or ONE: aba_walk_spec's looped instantiation takes stage = 4, all four stages with the stage states staying in its LDS rows
// (rbd_walk.hpp; the torques of the next stage are staged in behind each stage: tau_stride).  Scope: what the compiled kernels take (revolute / prismatic / sin-cos / fixed joints, 6-dof joints on the world).
// mk1[3 i ..] = (q offset, v offset, joint type) of the N1 1-dof joints; mkf[2 f ..] = (q offset, v offset) of the NF 6-dof ones (tables in device memory).
#pragma once
#include "rbd_lane.hpp"  // (brings rbd_integrator.hpp: joint_local_rate, joint_global)

namespace rbd {

// A pointer that comes out of a struct (or out of a conditional) has lost its address space: the compiler then emits FLAT loads and stores, which count on
// the LDS counter too — every wait for an LDS read behind them waits for the memory round trip.  Say it in the type.
#if defined(__HIP_DEVICE_COMPILE__)
#define RBD_GLOBAL_PTR(T) __attribute__((address_space(1))) T*
#else
#define RBD_GLOBAL_PTR(T) T*
#endif
template <typename T> RBD_DEV RBD_GLOBAL_PTR(T) as_global(T* p) { return (RBD_GLOBAL_PTR(T))p; }
template <typename T> RBD_DEV RBD_GLOBAL_PTR(const T) as_global(const T* p) { return (RBD_GLOBAL_PTR(const T))p; }

struct MkStage {
  int32_t stage;  // 0..3, or -1: plain dynamics!, or 4: all four stages in this launch (aba_walk_spec only)
  int32_t pd;     // 1: τ −= kp (q − q_des) + kd v on the revolute / prismatic joints, on the stage state
  double dt;
  void* q0; void* v0;      // base point of the step       } nq B / nv B values each, batch-innermost: element (k, state) at k B + state, whatever the
  void* accp; void* accv;  // Σ b_j φ̇_j, Σ b_j v̇_j so far } caller's layout (both compiled kernels; nobody else reads them)
  void* q_state; void* v_state;
  const void* kp; const void* kd; const void* qdes;
  int64_t tau_stride;  // stage == 4 (all four stages in one launch, aba_walk_spec): elements from one stage's torques to the next's (0: the same torques)
};

// ---- round 5: the stage of the walk kernels with the LANE AS THE STATE ---------------------------------------------------------------------------------
// The element-parallel form of round 4 cost a workgroup of aba_walk_spec_f64 16 us of its own (four rounds of workgroups: 158 us per launch against 94 without
// the stage): it sat in front of AND behind the passes, each part a chain of table values -> the loads they address -> arithmetic -> stores, on a compute unit
// that has nothing else to run meanwhile.  Here the whole stage stands BEHIND the passes — which read the q and v rows but leave them alone, so the stage state
// is still there — and the lane is the state:
//   * base point and running sums live in the workspace's stage buffers BATCH-INNERMOST (element (k, state) at k B + state, as aba_spec keeps them, whatever
//     the caller's layout): one coalesced access per wavefront and value; every load of a wavefront goes out before the barrier that ends the passes;
//   * wavefront g takes the 1-dof joints g, g + G, ... — the joint is wave-uniform, its table entry a scalar load — and the 6-dof joints are dealt out from the
//     last wavefront down; a wavefront reads and writes only its own joints' rows: no barrier inside the stage;
//   * the next stage's q and v are written over the q and v rows and leave through the kernel's own way out (whole rows, the caller's layout), like v̇.
// The PD law (rbd_simulate_controlled) needs the stage state in front of the passes: mk_lane_pd.
template <typename S, int N1, int NF, int G, int N> struct MkLane {
  static constexpr int U1 = (N1 + G - 1) / G > 0 ? (N1 + G - 1) / G : 1, UF = (NF + G - 1) / G > 0 ? (NF + G - 1) / G : 1;
  S q0a[U1][N], q0b[U1][N], accp[U1][N], v0[U1][N], accv[U1][N];
  S fq0[UF][N][7], faccp[UF][N][6], fv0[UF][N][6], faccv[UF][N][6];
  S fqn[UF][N][7];  // the 6-dof joints' next configuration: formed in front of the barrier that ends the passes (mk_lane_float_q), written behind it
};
// which wavefront takes 6-dof joint f: the one with the shortest track first (fw), then round the workgroup
RBD_DEV int mk_float_wave(int f, int fw, int G) { return (fw + f) % G; }
// stages 1..3: base point and slopes / running sums of this wavefront's joints (stage 0 needs none: the base point is the stage state, the sums start at zero).
// In two parts, so that what the 6-dof joints' SE(3) arithmetic needs is all that is live while it runs: mk_lane_load_float (their base configuration and
// configuration sum) in front of mk_lane_float_q, mk_lane_load (everything else) behind it.
template <typename S, int N1, int NF, int G, int N>
RBD_DEV void mk_lane_load_float(const MkStage& F, MkLane<S, N1, NF, G, N>& P, const int32_t* mkf, long state0, long B, int g, int lane, int fw) {
  if constexpr (NF > 0) {
    if (F.stage <= 0) return;
    const auto q0 = as_global((const S*)F.q0), ap = as_global((const S*)F.accp);
#pragma unroll
    for (int u = 0; u < MkLane<S, N1, NF, G, N>::UF; ++u) {
      const int f = ((g - fw + G) % G) + u * G;  // (mk_float_wave(f) == g)
      if (f < NF) {
        const int qo = mkf[2 * f], vo = mkf[2 * f + 1];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const long gi = state0 + lane + 64 * j, si = gi < B ? gi : B - 1;
#pragma unroll
          for (int k = 0; k < 7; ++k) P.fq0[u][j][k] = q0[(long)(qo + k) * B + si];
#pragma unroll
          for (int k = 0; k < 6; ++k) P.faccp[u][j][k] = ap[(long)(vo + k) * B + si];
        }
      }
    }
  }
}
template <typename S, int N1, int NF, int G, int N>
RBD_DEV void mk_lane_load(const MkStage& F, MkLane<S, N1, NF, G, N>& P, const int32_t* mk1, const int32_t* mkf, long state0, long B, int g, int lane, int fw) {
  if (F.stage <= 0) return;
  const auto q0 = as_global((const S*)F.q0), v0 = as_global((const S*)F.v0), ap = as_global((const S*)F.accp), av = as_global((const S*)F.accv);
#pragma unroll
  for (int u = 0; u < MkLane<S, N1, NF, G, N>::U1; ++u) {
    const int i = g + u * G;  // wave-uniform
    if (i < N1) {
      const int qo = mk1[3 * i], vo = mk1[3 * i + 1], ty = mk1[3 * i + 2];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const long gi = state0 + lane + 64 * j, si = gi < B ? gi : B - 1;
        P.q0a[u][j] = q0[(long)qo * B + si];
        P.q0b[u][j] = ty == RBD_JOINT_SINCOS_REVOLUTE ? q0[(long)(qo + 1) * B + si] : S(0);
        P.v0[u][j] = v0[(long)vo * B + si];
        // the slopes of stages 0 and 1, kept in the two sum buffers' places: asked for by the closing stage only (mk_lane_stage)
        P.accp[u][j] = F.stage == 3 ? ap[(long)vo * B + si] : S(0);
        P.accv[u][j] = F.stage == 3 ? av[(long)vo * B + si] : S(0);
      }
    }
  }
  if constexpr (NF > 0) {
#pragma unroll
    for (int u = 0; u < MkLane<S, N1, NF, G, N>::UF; ++u) {
      const int f = ((g - fw + G) % G) + u * G;  // (mk_float_wave(f) == g)
      if (f < NF) {
        const int vo = mkf[2 * f + 1];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const long gi = state0 + lane + 64 * j, si = gi < B ? gi : B - 1;
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            P.fv0[u][j][k] = v0[(long)(vo + k) * B + si];
            P.faccv[u][j][k] = av[(long)(vo + k) * B + si];
          }
        }
      }
    }
  }
}
// CELL: (row, state of the workgroup) -> pointer to the scalar in the LDS rows.  rq / rv hold the stage state, rt the v̇ the passes left; on return rq / rv hold
// the next stage's state (the caller sends them out behind a barrier)
template <typename S, int N1, int NF, int G, int N, typename CELL>
RBD_DEV void mk_lane_stage(const MkStage& F, const MkLane<S, N1, NF, G, N>& P, CELL cell, const int32_t* mk1, const int32_t* mkf, int rq, int rv, int rt,
                           long state0, long B, int g, int lane, int fw) {
  const int s = F.stage;
  const S h = (S)F.dt;
  const S bs = (s == 0 || s == 3) ? S(1) / S(6) : S(1) / S(3);
  const S an = s < 2 ? S(0.5) : S(1);
  const auto q0 = as_global((S*)F.q0), v0 = as_global((S*)F.v0), ap = as_global((S*)F.accp), av = as_global((S*)F.accv);
#pragma unroll
  for (int u = 0; u < MkLane<S, N1, NF, G, N>::U1; ++u) {
    const int i = g + u * G;
    if (i < N1) {
      const int qo = mk1[3 * i], vo = mk1[3 * i + 1], ty = mk1[3 * i + 2];
      const bool sc = ty == RBD_JOINT_SINCOS_REVOLUTE;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int st = lane + 64 * j;
        const long gi = state0 + st;
        const bool live = gi < B;
        const S rate = *cell(rv + vo, st);  // a 1-dof joint's local-coordinate rate is its velocity (joint_types.jl:9-18, sin_cos_revolute.jl:173-196)
        const S qa = *cell(rq + qo, st);
        const S qb = sc ? *cell(rq + qo + 1, st) : S(0);
        const S vd = *cell(rt + vo, st);
        const S q0a = s == 0 ? qa : P.q0a[u][j], q0b = s == 0 ? qb : P.q0b[u][j], vb = s == 0 ? rate : P.v0[u][j];
        // A 1-dof joint's rate IS its stage velocity, and the stage velocities are the base velocity plus h a times the slopes before them — so the two running
        // sums need not be read and written at every stage (six trips of nv B values each per step): the slopes of stage 0 and stage 1 are written once, where
        // they arise (k0 in accv's place, k1 in accp's), and read once, by the closing stage, which gets k2 back from its own stage velocity,
        // k2 = (v_3 - v0) / h — exact to an ulp of v over h, which the closing combination multiplies by h again.  Same sums, same order as written out.
        S sump = S(0), sumv = S(0);
        if (s == 3) {
          const S k0 = P.accv[u][j], k1 = P.accp[u][j], k2 = (rate - vb) / h, hh = h * S(0.5);
          const S b0 = S(1) / S(6), b1 = S(1) / S(3);
          sump = ((b0 * vb + b1 * (vb + hh * k0)) + b1 * (vb + hh * k1)) + b0 * rate;  // sum of b_j v_j over the stage velocities
          sumv = ((b0 * k0 + b1 * k1) + b1 * k2) + b0 * vd;
        }
        if (live) {
          if (s == 0) { q0[(long)qo * B + gi] = qa; v0[(long)vo * B + gi] = rate; if (sc) q0[(long)(qo + 1) * B + gi] = qb; av[(long)vo * B + gi] = vd; }
          if (s == 1) ap[(long)vo * B + gi] = vd;
        }
        const S phi = s < 3 ? h * an * rate : h * sump;
        if (sc) {
          S sd, cd;
          sincos_t(phi, &sd, &cd);
          *cell(rq + qo, st) = q0a * cd + q0b * sd;
          *cell(rq + qo + 1, st) = q0b * cd - q0a * sd;
        } else {
          *cell(rq + qo, st) = q0a + phi;
        }
        *cell(rv + vo, st) = vb + h * (s < 3 ? an * vd : sumv);
      }
    }
  }
  if constexpr (NF > 0) {  // the 6-dof joints: their next configuration waits in P.fqn (mk_lane_float_q); here the velocity half and the rows
#pragma unroll
    for (int u = 0; u < MkLane<S, N1, NF, G, N>::UF; ++u) {
      const int f = ((g - fw + G) % G) + u * G;
      if (f < NF) {
        const int qo = mkf[2 * f], vo = mkf[2 * f + 1];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const int st = lane + 64 * j;
          const long gi = state0 + st;
          const bool live = gi < B;
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const S vd = *cell(rt + vo + k, st), vjk = *cell(rv + vo + k, st);
            const S sumv = (s == 0 ? S(0) : P.faccv[u][j][k]) + bs * vd;
            if (s == 0 && live) v0[(long)(vo + k) * B + gi] = vjk;
            if (s < 3 && live) av[(long)(vo + k) * B + gi] = sumv;
            *cell(rv + vo + k, st) = (s == 0 ? vjk : P.fv0[u][j][k]) + h * (s < 3 ? an * vd : sumv);
          }
#pragma unroll
          for (int k = 0; k < 7; ++k) *cell(rq + qo + k, st) = P.fqn[u][j][k];
        }
      }
    }
  }
}
// The 6-dof joints' configuration half of the stage — log_with_time_derivative and exp on SE(3), the only arithmetic of the stage that takes time: it needs
// the stage state and the base point, NOT v̇, so the wavefront runs it as soon as its own track is done (in front of the barrier that ends the passes, while
// the longer tracks finish; nothing is written to the rows: the passes of the others still read them).  Result in P.fqn.
template <typename S, int N1, int NF, int G, int N, typename CELL>
RBD_DEV void mk_lane_float_q(const MkStage& F, MkLane<S, N1, NF, G, N>& P, CELL cell, const int32_t* mkf, int rq, int rv, long state0, long B, int g, int lane, int fw) {
  if constexpr (NF > 0) {
    const int s = F.stage;
    const S h = (S)F.dt;
    const S bs = (s == 0 || s == 3) ? S(1) / S(6) : S(1) / S(3);
    const S an = s < 2 ? S(0.5) : S(1);
    const auto q0 = as_global((S*)F.q0), ap = as_global((S*)F.accp);
#pragma unroll
    for (int u = 0; u < MkLane<S, N1, NF, G, N>::UF; ++u) {
      const int f = ((g - fw + G) % G) + u * G;
      if (f < NF) {
        const int qo = mkf[2 * f], vo = mkf[2 * f + 1];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const int st = lane + 64 * j;
          const long gi = state0 + st;
          const bool live = gi < B;
          S qj[7], vj[6], q0j[7], rate[6], phi[6];
#pragma unroll
          for (int k = 0; k < 7; ++k) { qj[k] = *cell(rq + qo + k, st); q0j[k] = s == 0 ? qj[k] : P.fq0[u][j][k]; }
#pragma unroll
          for (int k = 0; k < 6; ++k) vj[k] = *cell(rv + vo + k, st);
          if (s == 0 && live) {
#pragma unroll
            for (int k = 0; k < 7; ++k) q0[(long)(qo + k) * B + gi] = qj[k];
          }
          joint_local_rate<S, 2>(RBD_JOINT_QUAT_FLOATING, q0j, qj, vj, rate);
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const S sump = (s == 0 ? S(0) : P.faccp[u][j][k]) + bs * rate[k];
            if (s < 3 && live) ap[(long)(vo + k) * B + gi] = sump;
            phi[k] = s < 3 ? h * an * rate[k] : h * sump;
          }
          joint_global<S, 2>(RBD_JOINT_QUAT_FLOATING, q0j, phi, P.fqn[u][j]);
        }
      }
    }
  }
}
// the PD law of rbd_simulate_controlled on the stage state, into the tau rows (in front of the passes; the caller puts a barrier behind it)
template <typename S, int N1, int G, int N, typename CELL>
RBD_DEV void mk_lane_pd(const MkStage& F, CELL cell, const int32_t* mk1, int rq, int rv, int rt, long state0, long B, Layout Lq, int g, int lane) {
  const auto kp = as_global((const S*)F.kp), kd = as_global((const S*)F.kd), qdes = as_global((const S*)F.qdes);
  for (int i = g; i < N1; i += G) {
    const int qo = mk1[3 * i], vo = mk1[3 * i + 1], ty = mk1[3 * i + 2];
    if (ty != RBD_JOINT_REVOLUTE && ty != RBD_JOINT_PRISMATIC) continue;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int st = lane + 64 * j;
      const long gi = state0 + st, si = gi < B ? gi : B - 1;
      const S qd = qdes ? qdes[(long)qo * Lq.sk + si * Lq.sb] : S(0);
      *cell(rt + vo, st) -= kp[vo] * (*cell(rq + qo, st) - qd) + kd[vo] * *cell(rv + vo, st);
    }
  }
}

}  // namespace rbd
