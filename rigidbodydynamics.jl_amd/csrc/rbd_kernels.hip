// rbd_kernels.hip — hand-written gfx950 kernels for the batched hot path of RigidBodyDynamics.jl:
//   aba_kernel   : dynamics!        (src/mechanism_algorithms.jl:845-864) as a fused world-frame
//                  articulated-body algorithm (same v̇ = M⁻¹(τ − c))
//   rnea_kernel  : inverse_dynamics! (:542-553) and dynamics_bias! (:484-498)
//   crba_kernel  : mass_matrix!      (:248-272)
//   kin_kernel   : momentum_matrix!, center_of_mass, energies, geometric_jacobian!   (by-products of the FK pass)
//   chol_*_kernel, loop_solve_kernel, mk_stage_kernel : dynamics_solve! (dense / loop-joint branch), Munthe-Kaas RK4 stage
//   (two bodies per lane: rbd_bank_kernels.hip; one lane per state: rbd_walk_kernels.hip, rbd_state_kernels.hip)
// Mapping here: one lane per (state, body); level-synchronous sweeps; parent/child exchange by DPP wave
// shifts (first child) and ds_bpermute; per-body quantities stay in VGPRs (see rbd_device.hpp).
#include <atomic>
#include "rbd_device.hpp"
#include "rbd_internal.hpp"
#include "rbd_hip.h"

#include "rbd_lane.hpp"
#include "rbd_loop_small.hpp"
namespace rbd {

// ---------------------------------------------------------------------------------------------
// (RBD_DEBUG_STOP: phase-ablation aid, compiled only with -DRBD_PROFILE_PHASES (scripts/gpu_ablate.sh); absent from the product build)
// Fused forward dynamics (ABA).  One launch: FK + twists, articulated inertias bottom-up, accelerations top-down.
// ---------------------------------------------------------------------------------------------
#ifdef RBD_PROFILE_PHASES
#define RBD_DEBUG_STOP(phase, expr)                                  \
  if (M.debug_stop == (phase)) {                                     \
    T chk_[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};                \
    chk_[0] = (expr);                                                \
    store_joint_v(b, vdot, Lv, chk_);                                \
    return;                                                          \
  }
#else
#define RBD_DEBUG_STOP(phase, expr)
#endif
#ifndef RBD_ABA_F32_WAVES
#define RBD_ABA_F32_WAVES 3
#endif
// packed symmetric NDOF x NDOF index
__host__ __device__ constexpr int DI(int k, int m, int n) { return k <= m ? (k * n - k * (k - 1) / 2 + (m - k)) : (m * n - m * (m - 1) / 2 + (k - m)); }

// NDOF = 1: every non-floating tree joint has one degree of freedom (the common case, e.g. Atlas);
// NDOF = 3: the mechanism also has QuaternionSpherical / Planar tree joints (S is 6x3, D a 3x3 SPD block).
template <typename T, bool INNER_FLOAT, int NDOF>
__global__ __launch_bounds__(256, (sizeof(T) == 8 ? (NDOF == 1 ? 2 : 1) : (NDOF == 1 ? RBD_ABA_F32_WAVES : 2)))
void aba_kernel(DevModel M, long B, const T* q, const T* v, const T* __restrict__ tau,  // q, v: the fused simulate path also writes them (F.q_state)
                const T* __restrict__ fext, T* __restrict__ vdot, T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf, MkFuse F) {
  constexpr int ND2 = NDOF * (NDOF + 1) / 2;
  Body<T> b;
  load_body(M, B, b);
  const T* rb = reinterpret_cast<const T*>(M.rb) + (b.sub < M.nb ? b.sub : 0) * RB_STRIDE;

  T qj[7], vj[6], tj[6];
  load_joint_q(b, q, Lq, qj);
  load_joint_v(b, v, Lv, vj);
  load_joint_v(b, tau, Lv, tj);
  if (F.stage >= 0) {
    // `simulate` fusion: this launch is stage F.stage of a Munthe-Kaas RK4 step.  (q, v) hold the previous stage state: close
    // that stage (its v̇ is already in W.vd[stage-1]), form this stage's state in registers and in the state buffers, and
    // evaluate the dynamics there; v̇ goes to W.vd[stage].  With close_prev (stage 0 of every step but the first) the previous
    // step is closed first — its four stages combined into the new (q, v) — instead of by a launch of its own.
    if (F.close_prev) mk_stage_lane(b, 4, (T)F.dt, qj, vj, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
    mk_stage_lane(b, F.stage, (T)F.dt, qj, vj, (const T*)nullptr, F.W, (T*)F.q_state, (T*)F.v_state, Lq, Lv);
    pd_control_lane(b, F, Lq, qj, vj, tj);
  }
  store_qdot(b, qdot, Lq, qj, vj);
  RBD_DEBUG_STOP(1, qj[0] + vj[0] + tj[0]);

  T XR[9], Xp[3], R[9], p[3], tl[6], Tw[6], vJ[6];
  local_transform(b, rb, qj, XR, Xp);
  local_joint_motion(b, rb, vj, tl);
  RBD_DEBUG_STOP(2, XR[0] + XR[4] + XR[8] + Xp[0] + tl[0]);
  sweep_kinematics<T, false>(M, b, XR, Xp, R, p, tl, Tw, vJ, nullptr, nullptr);
  RBD_DEBUG_STOP(3, R[0] + R[4] + R[8] + p[0] + Tw[0] + Tw[5] + vJ[2]);

  // per-body, all lanes in parallel: motion subspace, bias term, inertia, bias force
  const bool floating = (b.jtype == RBD_JOINT_QUAT_FLOATING);
  const int nvj = joint_nv(b.jtype);
  const int ndof = floating ? 0 : nvj;  // columns handled through (S, U, D); the 6-dof joint has its own closed form
  T S[NDOF][6];
#pragma unroll
  for (int k = 0; k < NDOF; ++k) {
    T e[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, sl[6];
    e[k] = T(1);
    local_joint_motion(b, rb, e, sl);  // column k of the local motion subspace
    xmotion(R, p, sl, S[k]);
    if (k >= ndof) {
#pragma unroll
      for (int j = 0; j < 6; ++j) S[k][j] = T(0);
    }
  }
  T cb[6];
  se3_comm(Tw, vJ, cb);  // [T_b, vJ]: bias acceleration increment (mechanism_state.jl:814-830)
  T IA[21], pA[6];
  {
    RInertia<T> I;
    T Jb[6], mc[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
    inertia_to_root(Jb, mc, rb[RB_M], R, p, I);
    sym6_from_inertia(I, IA);
    momentum_cross(I, Tw, pA);
    T fe[6];
    load_body_wrench(b, fext, Lf, fe);
#pragma unroll
    for (int k = 0; k < 6; ++k) pA[k] -= fe[k];
  }
  if (!b.valid) {
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) pA[k] = T(0);
  }

  RBD_DEBUG_STOP(4, IA[0] + IA[20] + IA[7] + pA[0] + pA[5] + cb[1] + S[0][2]);

  // bottom-up: articulated-body inertias and bias forces.  Lanes at level l finish (U, D, u); every lane then forms
  // its hand-off (Ia, pa) = (IA - U D^-1 U', pA + Ia cb + U D^-1 u) — only the hand-offs of level-l lanes are pulled.
  // (6-dof joints keep pa = S^-T tau — the wrench transform of tau to the root frame — in U[0], which they do not
  //  otherwise need, and park their transform in LDS until the top-down sweep: cold data out of the VGPR budget.)
  T U[NDOF][6], Dinv[ND2], u[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; ++k) {
    u[k] = T(0);
#pragma unroll
    for (int j = 0; j < 6; ++j) U[k][j] = T(0);
  }
#pragma unroll
  for (int k = 0; k < ND2; ++k) Dinv[k] = T(0);
  __shared__ T stash[12][256];
  if (floating) {
    xforce(R, p, tj, U[0]);
#pragma unroll
    for (int k = 0; k < 9; ++k) stash[k][threadIdx.x] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) stash[9 + k][threadIdx.x] = p[k];
  }
  T tauj[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; ++k) tauj[k] = tj[k];
  const bool inner_floating = floating && (b.parent >= 0);
  auto finish_joint = [&]() {  // U = IA S, D = S'U, u = tau - S'pA, Dinv = D^-1 on the active ndof x ndof block
    T D[ND2];
#pragma unroll
    for (int k = 0; k < NDOF; ++k) sym6_mul(IA, S[k], U[k]);
#pragma unroll
    for (int k = 0; k < NDOF; ++k) {
      u[k] = tauj[k] - dot6(S[k], pA);
#pragma unroll
      for (int m = k; m < NDOF; ++m) D[DI(k, m, NDOF)] = dot6(S[k], U[m]);
    }
    if (NDOF == 1) {
      Dinv[0] = rcp_nr(D[0]);
    } else {
      if (ndof == 1) {
        Dinv[DI(0, 0, NDOF)] = rcp_nr(D[DI(0, 0, NDOF)]);
      } else {  // 3x3 SPD inverse by cofactors
        const T d00 = D[DI(0, 0, NDOF)], d01 = D[DI(0, 1, NDOF)], d02 = D[DI(0, 2, NDOF)], d11 = D[DI(1, 1, NDOF)], d12 = D[DI(1, 2, NDOF)],
                d22 = D[DI(2, 2, NDOF)];
        const T c00 = d11 * d22 - d12 * d12, c01 = d02 * d12 - d01 * d22, c02 = d01 * d12 - d02 * d11;
        const T c11 = d00 * d22 - d02 * d02, c12 = d01 * d02 - d00 * d12, c22 = d00 * d11 - d01 * d01;
        const T idet = rcp_nr(d00 * c00 + d01 * c01 + d02 * c02);
        Dinv[DI(0, 0, NDOF)] = c00 * idet; Dinv[DI(0, 1, NDOF)] = c01 * idet; Dinv[DI(0, 2, NDOF)] = c02 * idet;
        Dinv[DI(1, 1, NDOF)] = c11 * idet; Dinv[DI(1, 2, NDOF)] = c12 * idet; Dinv[DI(2, 2, NDOF)] = c22 * idet;
      }
    }
  };
  NsStream nss = ns_begin(M.ns_desc);
#pragma unroll 1
  for (int l = M.nlevels - 1; l >= 1; --l) {
    if (b.level == l && ndof > 0) finish_joint();
    // Hand-off entries are formed one at a time and consumed at once (no 27-wide temporary): the first child is the
    // next lane (DPP shift); further children (branch points only) re-form the entries and pull them with ds_bpermute.
    // Givers sit at level l (mask 0: their own IA/pA are untouched), takers at level l-1.
    const T kI = (INNER_FLOAT && inner_floating) ? T(0) : T(1);
    const bool take0 = (b.level == l - 1) && (b.nchild >= 1);
    T W[NDOF][6];  // W = U D^-1
#pragma unroll
    for (int k = 0; k < NDOF; ++k)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        T w = T(0);
#pragma unroll
        for (int m = 0; m < NDOF; ++m) w += U[m][i] * Dinv[DI(k, m, NDOF)];
        W[k][i] = w;
      }
    auto entry = [&](int i, int j) {  // (Ia)_ij = IA_ij - sum_k W_k,i U_k,j
      T g = IA[SI(i, j)];
#pragma unroll
      for (int k = 0; k < NDOF; ++k) g -= W[k][i] * U[k][j];
      if (INNER_FLOAT) g *= kI;
      return g;
    };
    T Iac[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    // the first child is the next lane: row by row, formed, shifted, and added under the taker's exec mask
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      T h[6];
#pragma unroll
      for (int j = i; j < 6; ++j) {
        const T g = entry(i, j);
        Iac[i] += g * cb[j];
        if (j > i) Iac[j] += g * cb[i];
        h[j] = from_next_lane(g);
      }
      if (take0) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int j = i; j < 6; ++j) IA[SI(i, j)] += h[j];
      }
    }
    T gp[6];
    {
      T h[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        T x = pA[k] + Iac[k];
#pragma unroll
        for (int m = 0; m < NDOF; ++m) x += W[m][k] * u[m];
        gp[k] = x;
        if (INNER_FLOAT) gp[k] = inner_floating ? U[0][k] : gp[k];
        h[k] = from_next_lane(gp[k]);
      }
      if (take0) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int k = 0; k < 6; ++k) pA[k] += h[k];
      }
    }
    const int ns = ns_next(nss, M.ns_desc);
#pragma unroll 1
    for (int s = 1; s < ns; ++s) {
      const bool take = (b.level == l - 1) && (s < b.nchild);
      const int src = take ? b.base + child_sel(b, s) : b.lane;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        T tmp[6];
#pragma unroll
        for (int j = i; j < 6; ++j) tmp[j] = shfl(entry(i, j), src);
        if (take) {
          RBD_KEEP_BRANCH();
#pragma unroll
          for (int j = i; j < 6; ++j) IA[SI(i, j)] += tmp[j];
        }
      }
      T tp[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) tp[k] = shfl(gp[k], src);
      if (take) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int k = 0; k < 6; ++k) pA[k] += tp[k];
      }
    }
  }
  if (b.level == 0 && ndof > 0) finish_joint();

  RBD_DEBUG_STOP(5, IA[0] + IA[20] + pA[0] + U[0][0] + u[0] + Dinv[0]);

  // top-down: accelerations and v̇
  T acc[6], vd[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { acc[k] = T(0); vd[k] = T(0); }
  auto joint_accel = [&](const T* ap) {  // v̇ = D^-1 (u - U'a'), a = a' + S v̇   (ap = a' = a_parent + cb)
    T r[NDOF];
#pragma unroll
    for (int k = 0; k < NDOF; ++k) r[k] = u[k] - dot6(U[k], ap);
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = ap[k];
#pragma unroll
    for (int k = 0; k < NDOF; ++k) {
      T x = T(0);
#pragma unroll
      for (int m = 0; m < NDOF; ++m) x += Dinv[DI(k, m, NDOF)] * r[m];
      vd[k] = x;
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[j] += S[k][j] * x;
    }
  };
  auto finish_body = [&](const T* a_parent) {
    T ap[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) ap[k] = a_parent[k] + cb[k];
    if (floating) {
      // IA a_b = S^-T tau - pA ;  v̇ = S^-1 (a_b - a')
      T rhs[6], d[6], Rs[9], ps[3];
#pragma unroll
      for (int k = 0; k < 6; ++k) rhs[k] = U[0][k] - pA[k];
      sym6_solve(IA, rhs, acc);
#pragma unroll
      for (int k = 0; k < 6; ++k) d[k] = acc[k] - ap[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rs[k] = stash[k][threadIdx.x];
#pragma unroll
      for (int k = 0; k < 3; ++k) ps[k] = stash[9 + k][threadIdx.x];
      xmotion_inv(Rs, ps, d, vd);
    } else {
      joint_accel(ap);  // fixed joints: S = U = 0 => a = a'
    }
  };
  {
    // level 0: the parent is the world, a_world = -gravity (mechanism_algorithms.jl:405)
    const T a0[6] = {T(0), T(0), T(0), T(-M.gravity[0]), T(-M.gravity[1]), T(-M.gravity[2])};
    if (b.level == 0) finish_body(a0);
  }
#pragma unroll 1
  for (int l = 1; l < M.nlevels; ++l) {
    T ap[6];
    pull_parent<T, 6>(M, b, l, acc, ap);
    if (b.level == l) {
      if (INNER_FLOAT) {
        finish_body(ap);
      } else {  // every 6-dof joint is at level 0: only the cheap update remains in the loop
#pragma unroll
        for (int k = 0; k < 6; ++k) ap[k] += cb[k];
        joint_accel(ap);
      }
    }
  }
  if (vdot) store_joint_v(b, vdot, Lv, vd);
  if (F.stage >= 0) store_joint_v(b, (T*)F.W.vd[F.stage], Lv, vd);
}

// ---------------------------------------------------------------------------------------------
// RNEA: inverse_dynamics! (vdot != nullptr) and dynamics_bias! (vdot == nullptr).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rnea_kernel(DevModel M, long B, const T* __restrict__ q, const T* __restrict__ v,
                                                   const T* __restrict__ vdot, const T* __restrict__ fext,
                                                   T* __restrict__ tau, T* __restrict__ qdot, T* __restrict__ body_out,
                                                   Layout Lq, Layout Lv, Layout Lf, T* __restrict__ acc_out, T* __restrict__ jw_out) {
  Body<T> b;
  load_body(M, B, b);
  const T* rb = reinterpret_cast<const T*>(M.rb) + (b.sub < M.nb ? b.sub : 0) * RB_STRIDE;
  T qj[7], vj[6], aj[6];
  load_joint_q(b, q, Lq, qj);
  load_joint_v(b, v, Lv, vj);
  load_joint_v(b, vdot, Lv, aj);
  store_qdot(b, qdot, Lq, qj, vj);
  T XR[9], Xp[3], R[9], p[3], tl[6], al[6], Tw[6], vJ[6], acc[6];
  local_transform(b, rb, qj, XR, Xp);
  local_joint_motion(b, rb, vj, tl);
  local_joint_motion(b, rb, aj, al);  // joint_spatial_acceleration: S_local * v̇ (revolute.jl:76-81)
  sweep_kinematics<T, true>(M, b, XR, Xp, R, p, tl, Tw, vJ, al, acc);
  if (body_out != nullptr && b.valid) {
    // per-body kinematics for the loop-joint branch (constraint_jacobian!/constraint_bias!): [state][reference body][24] =
    // transform_to_root R (9), p (3); twist_wrt_world (6); acceleration incl. the -gravity root term (6)
    T* o = body_out + (b.state * M.nb + b.orig) * 24;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[9 + k] = p[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { o[12 + k] = Tw[k]; o[18 + k] = acc[k]; }
  }

  if (acc_out != nullptr && b.valid) {
    // accelerations[body] of inverse_dynamics! / the bias accelerations of dynamics_bias! (spatial_accelerations!, bias_accelerations!:
    // mechanism_algorithms.jl:377-417), root frame, (angular; linear), 6 x n_bodies per state in the reference's body order
#pragma unroll
    for (int k = 0; k < 6; ++k) acc_out[(long)(6 * b.orig + k) * Lf.sk + b.state * Lf.sb] = acc[k];
  }
  // newton_euler! (mechanism_algorithms.jl:428-439): w = I a + T x* I T - wext
  T w[6];
  {
    RInertia<T> I;
    T Jb[6], mc[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
    inertia_to_root(Jb, mc, rb[RB_M], R, p, I);
    T Ia[6], x[6], fe[6];
    mul_inertia(I, acc, Ia);
    momentum_cross(I, Tw, x);
    load_body_wrench(b, fext, Lf, fe);
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = b.valid ? (Ia[k] + x[k] - fe[k]) : T(0);
  }
  // joint_wrenches_and_torques! (:442-459): w_parent += w_child, bottom-up
  NsStream nss = ns_begin(M.ns_desc);
  for (int l = M.nlevels - 1; l >= 1; --l) {
    const int ns = ns_next(nss, M.ns_desc);
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
      T give[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) give[k] = w[k];
      gather_add<T, 6>(b, l, s, give, w);
    }
  }
  if (jw_out != nullptr && b.valid) {
    // jointwrenchesout[body] (joint_wrenches_and_torques!, :442-459): the wrench across the joint above each body, root frame, (torque; force)
#pragma unroll
    for (int k = 0; k < 6; ++k) jw_out[(long)(6 * b.orig + k) * Lf.sk + b.state * Lf.sb] = w[k];
  }
  // tau = S' w
  T out[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) out[k] = T(0);
  if (b.jtype == RBD_JOINT_QUAT_FLOATING) {
    xforce_inv(R, p, w, out);
  } else {
    const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]}, ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
    const int ncol = M.has3dof ? 3 : 1;  // uniform
    for (int k = 0; k < ncol; ++k) {
      T sl[6], S[6];
      subspace_col(b.jtype, ax, ay, k, sl);
      xmotion(R, p, sl, S);
      const T d = dot6(S, w);
      if (k == 0) out[0] = d; else if (k == 1) out[1] = d; else out[2] = d;
    }
  }
  store_joint_v(b, tau, Lv, out);
}

// ---------------------------------------------------------------------------------------------
// CRBA: mass_matrix! (src/mechanism_algorithms.jl:248-272) with update_crb_inertias!
// (src/mechanism_state.jl:852-868).  Row block of body i: M[i, j] = F_i · S_j for every dof j of an
// ancestor-or-self joint (support set), zero elsewhere; only the lower triangle is written.
// M_out: element (i, j) of state b at M[(j*nv + i)*Lm.sk + b*Lm.sb].
// ---------------------------------------------------------------------------------------------
#ifdef RBD_PROFILE_PHASES
__device__ long long rbd_crba_phase_clock[16];
#define RBD_CMARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) rbd_crba_phase_clock[i] = clock64(); } while (0)
extern "C" int rbd_debug_crba_phase_clock(long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(rbd_crba_phase_clock), sizeof(long long) * 16);
}
#else
#define RBD_CMARK(i)
#endif
// ROOTREC (1-dof mechanisms whose 6-dof joints all sit on the world): the only 12-value record a chain can meet is the one of its
// level-0 ancestor, so every lane fetches that once and the walk passes 6-value records — half the ds_bpermute traffic per round.
template <typename T, int NDOF, bool ROOTREC>
__global__ __launch_bounds__(256) void crba_kernel(DevModel M, long B, const T* __restrict__ q, T* __restrict__ Mout,
                                                   Layout Lq, Layout Lm, int zero_fill) {
  RBD_CMARK(0);
  Body<T> b;
  load_body(M, B, b);
  const T* rb = reinterpret_cast<const T*>(M.rb) + (b.sub < M.nb ? b.sub : 0) * RB_STRIDE;
  T qj[7];
  load_joint_q(b, q, Lq, qj);
  T XR[9], Xp[3], R[9], p[3], zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, Tw[6], vJ[6];
  local_transform(b, rb, qj, XR, Xp);
  RBD_CMARK(1);
  sweep_kinematics<T, false>(M, b, XR, Xp, R, p, zero6, Tw, vJ, nullptr, nullptr);
  RBD_CMARK(2);
  RInertia<T> Ic;
  {
    T Jb[6], mc[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
    inertia_to_root(Jb, mc, rb[RB_M], R, p, Ic);
    if (!b.valid) {
#pragma unroll
      for (int k = 0; k < 6; ++k) Ic.J[k] = T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) Ic.c[k] = T(0);
      Ic.m = T(0);
    }
  }
  // composite-rigid-body inertias, bottom-up (10 scalars per child; first child by DPP): update_crb_inertias!
  // In place: at step l only lanes at level l-1 change, and what they read (level l) was finished one step earlier.
  NsStream nss = ns_begin(M.ns_desc);
#pragma unroll 1
  for (int l = M.nlevels - 1; l >= 1; --l) {
    const bool takes = (b.level == l - 1);
    T t[10];  // what the first child (the next lane) hands up
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = from_next_lane(Ic.J[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) t[6 + k] = from_next_lane(Ic.c[k]);
    t[9] = from_next_lane(Ic.m);
    if (takes && b.nchild >= 1) {
      RBD_KEEP_BRANCH();
#pragma unroll
      for (int k = 0; k < 6; ++k) Ic.J[k] += t[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Ic.c[k] += t[6 + k];
      Ic.m += t[9];
    }
    const int ns = ns_next(nss, M.ns_desc);
#pragma unroll 1
    for (int s = 1; s < ns; ++s) {
      const bool take = takes && (s < b.nchild);
      const int src = take ? b.base + child_sel(b, s) : b.lane;
      T u[10];
#pragma unroll
      for (int k = 0; k < 6; ++k) u[k] = shfl(Ic.J[k], src);
#pragma unroll
      for (int k = 0; k < 3; ++k) u[6 + k] = shfl(Ic.c[k], src);
      u[9] = shfl(Ic.m, src);
      if (take) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int k = 0; k < 6; ++k) Ic.J[k] += u[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Ic.c[k] += u[6 + k];
        Ic.m += u[9];
      }
    }
  }
  RBD_CMARK(3);
  const int jt = b.jtype;
  const int nvi = joint_nv(jt);
  const long nv = M.nv;
  const bool own_floating = (jt == RBD_JOINT_QUAT_FLOATING);
  const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]}, ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
  // What a lane publishes about its own joint: a 6-dof joint publishes its transform (R, p); a joint with <= NDOF degrees of
  // freedom publishes its root-frame motion subspace columns.  Walking up the support chain (support_set_masks,
  // src/mechanism_state.jl:95-98) is then "pull the record my parent currently holds", tree-depth times.
  constexpr int RS = ROOTREC ? 6 : (NDOF * 6 > 12 ? NDOF * 6 : 12);
  T rec[RS];
  T root[12];  // ROOTREC: transform (R, p) of this body's level-0 ancestor
#pragma unroll
  for (int k = 0; k < RS; ++k) rec[k] = T(0);
  if (ROOTREC) {
    const int a0 = b.valid ? M.anc[(b.sub < M.nb ? b.sub : 0) * M.nlevels + b.level] : b.sub;
#pragma unroll
    for (int k = 0; k < 9; ++k) root[k] = shfl(R[k], b.base + a0);
#pragma unroll
    for (int k = 0; k < 3; ++k) root[9 + k] = shfl(p[k], b.base + a0);
  }
  if (own_floating) {
    if (!ROOTREC) {
#pragma unroll
      for (int k = 0; k < 9; ++k) rec[k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) rec[(9 + k) % RS] = p[k];
    }
  } else {
#pragma unroll
    for (int c = 0; c < NDOF; ++c) {
      T sl[6];
      subspace_col(jt, ax, ay, c, sl);
      if (c < nvi) xmotion(R, p, sl, rec + 6 * c);
    }
  }
  int rjt = jt, rvoff = b.voff;
  // own force columns F_c = Ic S_c (mass_matrix!: Fi = Ici * Si, src/mechanism_algorithms.jl:258-260)
  T F[NDOF][6];
#pragma unroll
  for (int c = 0; c < NDOF; ++c) {
#pragma unroll
    for (int k = 0; k < 6; ++k) F[c][k] = T(0);
    if (!own_floating && c < nvi) mul_inertia(Ic, rec + 6 * c, F[c]);
  }
  auto emit = [&](const T* Fc, long row) {  // M[row, cols of the recorded joint] = Fc · S_col, lower triangle only
    if (rjt == RBD_JOINT_QUAT_FLOATING) {
      T o6[6];
      const T* H = ROOTREC ? root : rec;
      xforce_inv(H, H + 9, Fc, o6);  // S' F for S = Xm(H): the inverse wrench transform
#pragma unroll
      for (int cj = 0; cj < 6; ++cj) {
        const long col = rvoff + cj;
        if (col <= row) Mout[(col * nv + row) * Lm.sk + b.state * Lm.sb] = o6[cj];
      }
    } else {
      const int anv = joint_nv(rjt);
#pragma unroll
      for (int cj = 0; cj < NDOF; ++cj) {
        const long col = rvoff + cj;
        if (cj < anv && col <= row) Mout[(col * nv + row) * Lm.sk + b.state * Lm.sb] = dot6(Fc, rec + 6 * cj);
      }
    }
  };
  RBD_CMARK(4);
#pragma unroll 1
  for (int k = 0; k < M.nlevels; ++k) {
    if (k > 0) {
#pragma unroll
      for (int j = 0; j < RS; ++j) rec[j] = shfl(rec[j], b.plane);
      rjt = __shfl(rjt, b.plane, 64);
      rvoff = __shfl(rvoff, b.plane, 64);
    }
    if (b.valid && k <= b.level) {
      if (own_floating) {
#pragma unroll 1
        for (int ci = 0; ci < 6; ++ci) {
          T sl[6], Si[6], Fc[6];
          subspace_col(jt, ax, ay, ci, sl);
          if (ROOTREC) xmotion(root, root + 9, sl, Si);  // a 6-dof joint on the world is its own level-0 ancestor
          else xmotion(R, p, sl, Si);
          mul_inertia(Ic, Si, Fc);
          emit(Fc, (long)b.voff + ci);
        }
      } else {
#pragma unroll
        for (int ci = 0; ci < NDOF; ++ci)
          if (ci < nvi) emit(F[ci], (long)b.voff + ci);
      }
    }
  }
  RBD_CMARK(5);
  // structural zeros of the lower triangle (the reference writes them too: mechanism_algorithms.jl:266-267)
  if (zero_fill && b.valid) {
    for (int ci = 0; ci < nvi; ++ci) {
      const long row = b.voff + ci;
      for (int wd = 0; wd <= (int)(row >> 6); ++wd) {  // columns 0..row, 64 per mask word
        const long top = row - 64L * wd;               // highest column of this word that is <= row
        unsigned long long m = ~M.row_mask[row * M.row_words + wd] & ((top >= 63) ? ~0ull : ((2ull << top) - 1));
        while (m) {
          const long col = 64L * wd + __builtin_ctzll(m);
          m &= m - 1;
          Mout[(col * nv + row) * Lm.sk + b.state * Lm.sb] = T(0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Kinematics by-products from the same FK pass: momentum_matrix! in the root frame
// (src/mechanism_algorithms.jl:313-327: column i = crb_inertia(body(i)) * S_i), center_of_mass (:28-50),
// kinetic_energy / gravitational_potential_energy (src/mechanism_state.jl:886-903).  Outputs are nullable:
//   A_out [6*nv x B] (6 x nv column-major per state), com_out [3 x B], energy_out [2 x B] = (kinetic, potential).
// ---------------------------------------------------------------------------------------------
template <typename T> RBD_DEV T group_sum(T x, int lps) {  // sum over the lanes of one state (lps = power of two)
  for (int o = 1; o < lps; o <<= 1) x += shfl(x, (int)((threadIdx.x & 63) ^ o));
  return x;
}

// momentum(state), momentum_rate_bias(state) in the root frame (src/mechanism_state.jl:878-884, :975-987): mom_out [12 x B] =
// (sum_b I_b T_b ; sum_b I_b a_bias_b + T_b x* I_b T_b) with the bias accelerations of update_bias_accelerations_wrt_world!
template <typename T>
__global__ __launch_bounds__(256) void momentum_kernel(DevModel M, long B, const T* __restrict__ q, const T* __restrict__ v, T* __restrict__ mom_out,
                                                       Layout Lq, Layout Lv, Layout L12) {
  Body<T> b;
  load_body(M, B, b);
  const T* rb = reinterpret_cast<const T*>(M.rb) + (b.sub < M.nb ? b.sub : 0) * RB_STRIDE;
  T qj[7], vj[6];
  load_joint_q(b, q, Lq, qj);
  load_joint_v(b, v, Lv, vj);
  T XR[9], Xp[3], R[9], p[3], tl[6], Tw[6], vJ[6], acc[6];
  const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  local_transform(b, rb, qj, XR, Xp);
  local_joint_motion(b, rb, vj, tl);
  sweep_kinematics<T, true>(M, b, XR, Xp, R, p, tl, Tw, vJ, zero6, acc);  // joint accelerations 0: bias accelerations (+ the root's -g)
  acc[3] += T(M.gravity[0]); acc[4] += T(M.gravity[1]); acc[5] += T(M.gravity[2]);  // every body carries the root term exactly once
  RInertia<T> I;
  T Jb[6], mc[3], h[6], Ia[6], x[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
  inertia_to_root(Jb, mc, rb[RB_M], R, p, I);
  mul_inertia(I, Tw, h);
  mul_inertia(I, acc, Ia);
  momentum_cross(I, Tw, x);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const T hs = group_sum(b.valid ? h[k] : T(0), M.lps), ws = group_sum(b.valid ? Ia[k] + x[k] : T(0), M.lps);
    if (b.valid && b.sub == 0) {
      mom_out[(long)k * L12.sk + b.state * L12.sb] = hs;
      mom_out[(long)(6 + k) * L12.sk + b.state * L12.sb] = ws;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void kin_kernel(DevModel M, long B, const T* __restrict__ q, const T* __restrict__ v, T* __restrict__ A_out,
                                                  T* __restrict__ com_out, T* __restrict__ energy_out, T* __restrict__ J_out, uint64_t jplus,
                                                  uint64_t jminus, Layout Lq, Layout Lv, Layout La, Layout L3, Layout L2) {
  Body<T> b;
  load_body(M, B, b);
  const T* rb = reinterpret_cast<const T*>(M.rb) + (b.sub < M.nb ? b.sub : 0) * RB_STRIDE;
  T qj[7], vj[6];
  load_joint_q(b, q, Lq, qj);
  load_joint_v(b, v, Lv, vj);
  T XR[9], Xp[3], R[9], p[3], tl[6], Tw[6], vJ[6];
  local_transform(b, rb, qj, XR, Xp);
  local_joint_motion(b, rb, vj, tl);
  sweep_kinematics<T, false>(M, b, XR, Xp, R, p, tl, Tw, vJ, nullptr, nullptr);
  RInertia<T> Ic;
  {
    T Jb[6], mc[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
    inertia_to_root(Jb, mc, rb[RB_M], R, p, Ic);
    if (!b.valid) {
#pragma unroll
      for (int k = 0; k < 6; ++k) Ic.J[k] = T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) Ic.c[k] = T(0);
      Ic.m = T(0);
    }
  }
  // energies and centre of mass: per-body terms summed over the state's lanes
  {
    T h[6];
    mul_inertia(Ic, Tw, h);
    const T ke = b.valid ? dot6(h, Tw) / 2 : T(0);
    const T pe = -(T(M.gravity[0]) * Ic.c[0] + T(M.gravity[1]) * Ic.c[1] + T(M.gravity[2]) * Ic.c[2]);  // -m g·com, m com = root-frame cross part
    const T has = (Ic.m > T(0)) ? T(1) : T(0);
    const T kes = group_sum(ke, M.lps), pes = group_sum(pe * has, M.lps);
    const T ms = group_sum(Ic.m * has, M.lps);
    const T cx = group_sum(Ic.c[0] * has, M.lps), cy = group_sum(Ic.c[1] * has, M.lps), cz = group_sum(Ic.c[2] * has, M.lps);
    if (b.valid && b.sub == 0) {
      if (energy_out) { energy_out[0 * L2.sk + b.state * L2.sb] = kes; energy_out[1 * L2.sk + b.state * L2.sb] = pes; }
      if (com_out) { com_out[0 * L3.sk + b.state * L3.sb] = cx / ms; com_out[1 * L3.sk + b.state * L3.sb] = cy / ms; com_out[2 * L3.sk + b.state * L3.sb] = cz / ms; }
    }
  }
  const int nvi = joint_nv(b.jtype);
  const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]}, ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
  if (J_out != nullptr) {
    // geometric_jacobian!(jac, state, path) in the root frame (src/mechanism_algorithms.jl:80-99): -S for the joints walked
    // upwards (bit of jminus), +S downwards (jplus), zero off the path — every column is written (fill!(jac, 0))
    const T sg = ((jplus >> b.sub) & 1) ? T(1) : ((jminus >> b.sub) & 1) ? T(-1) : T(0);
#pragma unroll 1
    for (int ci = 0; ci < nvi; ++ci) {
      T sl[6], Si[6];
      subspace_col(b.jtype, ax, ay, ci, sl);
      xmotion(R, p, sl, Si);
      if (b.valid) {
#pragma unroll
        for (int k = 0; k < 6; ++k) J_out[((long)(b.voff + ci) * 6 + k) * La.sk + b.state * La.sb] = sg * Si[k];
      }
    }
  }
  if (A_out == nullptr) return;  // uniform
  NsStream nss = ns_begin(M.ns_desc);
  for (int l = M.nlevels - 1; l >= 1; --l) {
    const int ns = ns_next(nss, M.ns_desc);
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
      T give[10], acc[10];
#pragma unroll
      for (int k = 0; k < 6; ++k) give[k] = acc[k] = Ic.J[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) give[6 + k] = acc[6 + k] = Ic.c[k];
      give[9] = acc[9] = Ic.m;
      gather_add<T, 10>(b, l, s, give, acc);
#pragma unroll
      for (int k = 0; k < 6; ++k) Ic.J[k] = acc[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Ic.c[k] = acc[6 + k];
      Ic.m = acc[9];
    }
  }
#pragma unroll 1
  for (int ci = 0; ci < nvi; ++ci) {
    T sl[6], Si[6], F[6];
    subspace_col(b.jtype, ax, ay, ci, sl);
    xmotion(R, p, sl, Si);
    mul_inertia(Ic, Si, F);
    if (b.valid) {
#pragma unroll
      for (int k = 0; k < 6; ++k) A_out[((long)(b.voff + ci) * 6 + k) * La.sk + b.state * La.sb] = F[k];
    }
  }
}

}  // namespace rbd
namespace rbd {

// PD controller of rbd_simulate_controlled for the un-fused integrator path (large batches: the walk kernel does not carry the stage): one
// thread per (dof, state), tau_out = tau_ff − kp (q − q_des) − kd v on Revolute / Prismatic joints, tau_ff elsewhere
template <typename T>
__global__ __launch_bounds__(256) void pd_control_kernel(DevModel M, long B, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ tau_ff,
                                                         const T* __restrict__ qdes, const T* __restrict__ kp, const T* __restrict__ kd, T* __restrict__ tau_out,
                                                         Layout Lq, Layout Lv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M.nv * B) return;
  // consecutive threads walk the fastest-varying index of the layout
  const long r = Lv.sk == 1 ? i % M.nv : i / B, st = Lv.sk == 1 ? i / M.nv : i % B;
  const long a = r * Lv.sk + st * Lv.sb;
  T t = tau_ff ? tau_ff[a] : T(0);
  const int32_t* ib = M.ib + (long)M.dof_body[r] * IB_STRIDE;
  const int jt = ib[IB_JTYPE];
  if (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_PRISMATIC) {
    const long qa = (long)ib[IB_QOFF] * Lq.sk + st * Lq.sb;
    t -= kp[r] * (q[qa] - (qdes ? qdes[qa] : T(0))) + kd[r] * v[a];
  }
  tau_out[a] = t;
}
template <typename T>
hipError_t launch_pd_control(const DevModel& M, long B, const void* q, const void* v, const void* tau_ff, const void* qdes, const void* kp, const void* kd,
                             void* tau_out, Layout Lq, Layout Lv, hipStream_t s) {
  const long n = (long)M.nv * B;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(pd_control_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, M, B, (const T*)q, (const T*)v, (const T*)tau_ff, (const T*)qdes,
                     (const T*)kp, (const T*)kd, (T*)tau_out, Lq, Lv);
  return hipGetLastError();
}
template hipError_t launch_pd_control<double>(const DevModel&, long, const void*, const void*, const void*, const void*, const void*, const void*, void*, Layout, Layout, hipStream_t);
template hipError_t launch_pd_control<float>(const DevModel&, long, const void*, const void*, const void*, const void*, const void*, const void*, void*, Layout, Layout, hipStream_t);

// ---- launchers -----------------------------------------------------------------------------
static inline dim3 grid_for(const DevModel& M, long B, int block) {
  const long spw = 64 / M.lps;
  const long waves = (B + spw - 1) / spw;
  const long wpb = block / 64;
  return dim3((unsigned)((waves + wpb - 1) / wpb));
}

template <typename T>
hipError_t launch_kin(const DevModel& M, long B, const void* q, const void* v, void* A, void* com, void* energy, void* J, uint64_t jplus,
                      uint64_t jminus, Layout Lq, Layout Lv, Layout La, Layout L3, Layout L2, hipStream_t s) {
  hipLaunchKernelGGL(kin_kernel<T>, grid_for(M, B, 256), dim3(256), 0, s, M, B, (const T*)q, (const T*)v, (T*)A, (T*)com, (T*)energy, (T*)J, jplus,
                     jminus, Lq, Lv, La, L3, L2);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_momentum(const DevModel& M, long B, const void* q, const void* v, void* mom, Layout Lq, Layout Lv, Layout L12, hipStream_t s) {
  hipLaunchKernelGGL(momentum_kernel<T>, grid_for(M, B, 256), dim3(256), 0, s, M, B, (const T*)q, (const T*)v, (T*)mom, Lq, Lv, L12);
  return hipGetLastError();
}
template hipError_t launch_momentum<double>(const DevModel&, long, const void*, const void*, void*, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_momentum<float>(const DevModel&, long, const void*, const void*, void*, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_kin<double>(const DevModel&, long, const void*, const void*, void*, void*, void*, void*, uint64_t, uint64_t, Layout, Layout, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_kin<float>(const DevModel&, long, const void*, const void*, void*, void*, void*, void*, uint64_t, uint64_t, Layout, Layout, Layout, Layout, Layout, hipStream_t);

template <typename T>
hipError_t launch_aba(const DevModel& M, long B, const void* q, const void* v, const void* tau, const void* fext, void* vdot,
                      void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s, const MkFuse* fuse) {
  MkFuse F{};
  F.stage = -1;
  if (fuse) F = *fuse;
#define RBD_LAUNCH_ABA(IF, ND)                                                                                                  \
  hipLaunchKernelGGL((aba_kernel<T, IF, ND>), grid_for(M, B, 256), dim3(256), 0, s, M, B, (const T*)q, (const T*)v, (const T*)tau, \
                     (const T*)fext, (T*)vdot, (T*)qdot, Lq, Lv, Lf, F)
  if (M.has3dof) { if (M.inner_floating) RBD_LAUNCH_ABA(true, 3); else RBD_LAUNCH_ABA(false, 3); }
  else { if (M.inner_floating) RBD_LAUNCH_ABA(true, 1); else RBD_LAUNCH_ABA(false, 1); }
#undef RBD_LAUNCH_ABA
  return hipGetLastError();
}
template <typename T>
hipError_t launch_rnea(const DevModel& M, long B, const void* q, const void* v, const void* vdot, const void* fext, void* tau,
                       void* qdot, void* body_out, Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out, void* jw_out) {
  hipLaunchKernelGGL(rnea_kernel<T>, grid_for(M, B, 256), dim3(256), 0, s, M, B, (const T*)q, (const T*)v, (const T*)vdot,
                     (const T*)fext, (T*)tau, (T*)qdot, (T*)body_out, Lq, Lv, Lf, (T*)acc_out, (T*)jw_out);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_crba(const DevModel& M, long B, const void* q, void* Mout, Layout Lq, Layout Lm, int zero_fill, hipStream_t s) {
  if (M.has3dof) hipLaunchKernelGGL((crba_kernel<T, 3, false>), grid_for(M, B, 256), dim3(256), 0, s, M, B, (const T*)q, (T*)Mout, Lq, Lm, zero_fill);
  else if (M.inner_floating) hipLaunchKernelGGL((crba_kernel<T, 1, false>), grid_for(M, B, 256), dim3(256), 0, s, M, B, (const T*)q, (T*)Mout, Lq, Lm, zero_fill);
  else hipLaunchKernelGGL((crba_kernel<T, 1, true>), grid_for(M, B, 256), dim3(256), 0, s, M, B, (const T*)q, (T*)Mout, Lq, Lm, zero_fill);
  return hipGetLastError();
}

// max |a - b| and max |b| over the first n states of two n x nk arrays of one layout (a value that is not finite counts as infinitely far): the comparison of
// a run-time compiled program's first result with the interpreting kernel's (rbd_capi.hip: first_use_check).  One workgroup; out[0] = max |a - b|, out[1] = max |b|.
template <typename T>
__global__ __launch_bounds__(256) void max_diff_kernel(long n, int nk, const T* __restrict__ a, const T* __restrict__ b, Layout L, double* __restrict__ out) {
  __shared__ double sd[256], sr[256];
  double d = 0, r = 0;
  for (long i = threadIdx.x; i < n * nk; i += 256) {
    const long s = i / nk, k = i % nk, o = layout_base(L, s) + k * L.sk;
    const double x = (double)a[o], y = (double)b[o], e = fabs(x - y);
    d = fmax(d, (e == e && e <= 1.7e308) ? e : 1.7e308);
    r = fmax(r, fabs(y) <= 1.7e308 ? fabs(y) : 0.0);
  }
  sd[threadIdx.x] = d; sr[threadIdx.x] = r;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { sd[threadIdx.x] = fmax(sd[threadIdx.x], sd[threadIdx.x + w]); sr[threadIdx.x] = fmax(sr[threadIdx.x], sr[threadIdx.x + w]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sd[0]; out[1] = sr[0]; }
}
template <typename T>
hipError_t launch_max_diff(long n, int nk, const void* a, const void* b, Layout L, double* out, hipStream_t s) {
  hipLaunchKernelGGL((max_diff_kernel<T>), dim3(1), dim3(256), 0, s, n, nk, (const T*)a, (const T*)b, L, out);
  return hipGetLastError();
}
template hipError_t launch_max_diff<double>(long, int, const void*, const void*, Layout, double*, hipStream_t);
template hipError_t launch_max_diff<float>(long, int, const void*, const void*, Layout, double*, hipStream_t);

template hipError_t launch_aba<double>(const DevModel&, long, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, const MkFuse*);
template hipError_t launch_aba<float>(const DevModel&, long, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, const MkFuse*);
template hipError_t launch_rnea<double>(const DevModel&, long, const void*, const void*, const void*, const void*, void*, void*, void*, Layout, Layout, Layout, hipStream_t, void*, void*);
template hipError_t launch_rnea<float>(const DevModel&, long, const void*, const void*, const void*, const void*, void*, void*, void*, Layout, Layout, Layout, hipStream_t, void*, void*);
template hipError_t launch_crba<double>(const DevModel&, long, const void*, void*, Layout, Layout, int, hipStream_t);
template hipError_t launch_crba<float>(const DevModel&, long, const void*, void*, Layout, Layout, int, hipStream_t);

}  // namespace rbd

// ---------------------------------------------------------------------------------------------
// Batched dense Cholesky + solve: dynamics_solve!'s potrf!('L') / potrs! branch
// (src/mechanism_algorithms.jl:762-766, :819).  One wavefront per state, one lane per matrix row, the
// factor lives in LDS (row stride 4*odd words: conflict-free ds_read_b128 of own rows, broadcast reads of the
// pivot row).  rhs_i = (tau_i or 0) - (c_i or 0).  Reads only the LOWER triangle of M (uplo == 'L').
// Left-looking: column j of L is finished from rows j.. using the already finished columns 0..j-1.
// ---------------------------------------------------------------------------------------------
namespace rbd {

template <typename T>
__global__ __launch_bounds__(256) void chol_solve_kernel(int nv, int stride, long B, const T* __restrict__ Mg, const T* __restrict__ tau,
                                                         const T* __restrict__ c, T* __restrict__ x, T* __restrict__ Lout,
                                                         Layout Lm, Layout Lv, int* __restrict__ notpd) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long state = (long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (state >= B) return;  // whole wave exits together (one state per wave)
  T* A = smem + (size_t)wave * ((size_t)nv * stride + 64);  // [nv][stride] factor, then 64 words of broadcast scratch
  T* bc = A + (size_t)nv * stride;
  const int i = lane;
  const bool row = i < nv;
  // load the lower triangle: for fixed column j the lanes read consecutive rows (coalesced for AOS)
  for (int j = 0; j < nv; ++j)
    if (row && i >= j) A[i * stride + j] = Mg[((long)j * nv + i) * Lm.sk + state * Lm.sb];
  T b = T(0);
  if (row) {
    if (tau) b = tau[(long)i * Lv.sk + state * Lv.sb];
    if (c) b -= c[(long)i * Lv.sk + state * Lv.sb];
  }
  __builtin_amdgcn_s_waitcnt(0);
  bool bad = false;
  for (int j = 0; j < nv; ++j) {
    // s = A[i][j] - sum_{k<j} L[i][k] L[j][k]
    T s = T(0);
    if (row && i >= j) {
      s = A[i * stride + j];
      const T* ri = A + i * stride;
      const T* rj = A + j * stride;
      for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
    }
    // pivot: lane j holds s = d
    const T d = shfl(s, j);
    if (!(d > T(0))) bad = true;
    const T dr = SqrtT<T>::f(d);
    if (row && i >= j) A[i * stride + j] = (i == j) ? dr : s / dr;
  }
  if (bad && lane == 0) atomicOr(notpd, 1);
  // forward substitution L y = b (column oriented): y_k final at step k
  for (int k = 0; k < nv; ++k) {
    const T lkk = A[k * stride + k];
    const T yk = shfl(b, k) / lkk;
    if (i == k) b = yk;
    if (row && i > k) b -= A[i * stride + k] * yk;
  }
  // backward substitution L' x = y: x_k final at step k (descending); lanes i < k subtract L[k][i] x_k
  for (int k = nv - 1; k >= 0; --k) {
    const T lkk = A[k * stride + k];
    const T xk = shfl(b, k) / lkk;
    if (i == k) b = xk;
    if (row && i < k) b -= A[k * stride + i] * xk;
  }
  if (row) x[(long)i * Lv.sk + state * Lv.sb] = b;
  if (Lout) {  // optional: the factor (lower), same layout as M
    for (int j = 0; j < nv; ++j)
      if (row && i >= j) Lout[((long)j * nv + i) * Lm.sk + state * Lm.sb] = A[i * stride + j];
  }
  (void)bc;
}

// Register-resident variant used for nv <= 64: lane i keeps row i of the matrix in VGPRs (statically indexed, loops fully
// unrolled over the padded size NVP = nv rounded up to a multiple of 8; the padding is an identity block so no guards are
// needed).  Right-looking: column j is scaled by 1/sqrt(pivot), then every lane updates its own row with the column entries
// L[k][j] broadcast by v_readlane (uniform lane index) — no LDS traffic and no waits inside the factorization.  The factor
// is then transposed through LDS once for the backward substitution.
RBD_DEV float bcast_lane(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
RBD_DEV double bcast_lane(double x, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}

template <typename T, int NVP>
__global__ __launch_bounds__(256) void chol_reg_kernel(int nv, long B, const T* __restrict__ Mg, const T* __restrict__ tau, const T* __restrict__ c,
                                                       T* __restrict__ x, T* __restrict__ Lout, Layout Lm, Layout Lv, int* __restrict__ notpd) {
  __shared__ T Lt[4][NVP][NVP + 1];  // per wave: the factor, for transposed reads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long state = (long)blockIdx.x * 4 + wave;
  const bool live = state < B;
  const int i = lane;
  T row[NVP];
#pragma unroll
  for (int j = 0; j < NVP; ++j) {
    T a = (i == j) ? T(1) : T(0);  // identity padding
    if (live && i < nv && j <= i && j < nv) a = Mg[((long)j * nv + i) * Lm.sk + state * Lm.sb];
    row[j] = a;
  }
  T b = T(0);
  if (live && i < nv) {
    if (tau) b = tau[(long)i * Lv.sk + state * Lv.sb];
    if (c) b -= c[(long)i * Lv.sk + state * Lv.sb];
  }
  bool bad = false;
  T dinv_own = T(1);
#pragma unroll
  for (int j = 0; j < NVP; ++j) {
    const T d = bcast_lane(row[j], j);
    bad |= !(d > T(0));
    const T dr = SqrtT<T>::f(d);
    const T di = T(1) / dr;
    const T l = (i == j) ? dr : row[j] * di;  // L[i][j], meaningful for i >= j
    row[j] = l;
    dinv_own = (i == j) ? di : dinv_own;
#pragma unroll
    for (int k = j + 1; k < NVP; ++k) row[k] -= l * bcast_lane(l, k);  // A[i][k] -= L[i][j] L[k][j], meaningful for i >= k
  }
  if (live && bad && lane == 0) atomicOr(notpd, 1);
  // forward substitution L y = b (column oriented)
#pragma unroll
  for (int k = 0; k < NVP; ++k) {
    const T yk = bcast_lane(b, k) * bcast_lane(dinv_own, k);
    b = (i == k) ? yk : ((i > k) ? b - row[k] * yk : b);
  }
  // transpose the factor through LDS, then backward substitution L' x = y
  if (i < NVP) {
#pragma unroll
    for (int j = 0; j < NVP; ++j) Lt[wave][i][j] = row[j];
  }
  __syncthreads();
#pragma unroll
  for (int k = NVP - 1; k >= 0; --k) {
    const T xk = bcast_lane(b, k) * bcast_lane(dinv_own, k);
    const T lki = (i < NVP) ? Lt[wave][k][i < NVP ? i : 0] : T(0);  // L[k][i]
    b = (i == k) ? xk : ((i < k) ? b - lki * xk : b);
  }
  if (live && i < nv) x[(long)i * Lv.sk + state * Lv.sb] = b;
  if (Lout && live && i < nv) {
#pragma unroll
    for (int j = 0; j < NVP; ++j)
      if (j <= i && j < nv) Lout[((long)j * nv + i) * Lm.sk + state * Lm.sb] = row[j];
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA tile Cholesky (fp32): 16 states per wavefront.  A state is a quad of lanes; the matrix is cut into 4x4 tiles and
// lane r of the quad keeps row r of every lower tile in VGPRs (tile (I,J), register c  <->  A[4I + r][4J + c]).  Per block
// column J: (1) the diagonal tile is factored inside the quad (DPP quad broadcasts), (2) the panel tiles below are solved
// against it row by row (X = T L_d^-T), (3) the trailing sub-matrix gets the rank-4 update  T(I,J') -= X_I X_J'^T  as four
// v_mfma_f32_4x4x1_16B_f32 per tile: one instruction updates the tile of all 16 states.  Operand/result layout of that
// instruction (probed on gfx950, scripts/ubench/mfma4x4.hip): D[i][j] lives in lane 4*block + j, register i and equals
// A[lane 4*block + i] * B[lane 4*block + j]; feeding A = -X_J'[:,k], B = X_I[:,k] therefore leaves tile (I,J') in exactly the
// "lane = row, register = column" form it is stored in — no transposes anywhere.  This is the one true dense contraction of
// the path (the O(n^3) part of dynamics_solve!, src/mechanism_algorithms.jl:764), which is why it is the one place MFMA is used.
// Padding to NT*4 is an identity block.  Forward/backward substitution reuse the tiles (quad broadcasts / quad sums).
// ---------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int K> RBD_DEV float quad_bcast(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, true));
}
RBD_DEV float quad_sum(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
  return x;
}

template <int NT>
__global__ __launch_bounds__(64, (NT <= 9 ? 2 : 1)) void chol_mfma_kernel(int nv, long B, const float* __restrict__ Mg, const float* __restrict__ tau,
                                                       const float* __restrict__ c, float* __restrict__ x, float* __restrict__ Lout,
                                                       Layout Lm, Layout Lv, int* __restrict__ notpd, float* __restrict__ Mcopy, Layout Lc) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 3;
  const long state = (long)blockIdx.x * 16 + (lane >> 2);
  const bool live = state < B;
  f32x4_t t[NT][NT];  // lower tiles only (I >= J)
#pragma unroll
  for (int I = 0; I < NT; ++I)
#pragma unroll
    for (int J = 0; J <= I; ++J)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int row = 4 * I + r, col = 4 * J + cc;
        float a = (row == col) ? 1.0f : 0.0f;
        if (live && row < nv && col <= row) a = Mg[((long)col * nv + row) * Lm.sk + layout_base(Lm, state)];
        t[I][J][cc] = a;
      }
  // M itself in the caller's layout, when the matrix was built in a staging buffer (one-lane-per-state CRBA writes batch-innermost).
  // Measured at 65 536 states: storing the lower triangle straight from the tiles (a quad = 16 contiguous bytes per (tile, column), 16 such
  // pieces 5 KB apart per instruction) took 160 of the launch's 225 us — partially written cache lines, 1.1 TB/s; the lower triangle alone in
  // 16-byte pieces of contiguous column runs was no better (133 us for 189 MB).  Instead the whole square is
  // written, block column by block column: the four columns 4J..4J+3 of a state are 16 nv contiguous bytes of the caller's column-major M;
  // they are assembled in LDS (lower tiles as they are, the part above the diagonal from the transposed tiles (J, I < J): the reference
  // leaves the strict upper triangle of a Symmetric(:L) undefined, here it receives the mirror image) and leave as 16-byte pieces of complete,
  // contiguous runs.  16 consecutive states = one contiguous region of the output, every byte of which this wavefront writes.
  constexpr int CB = 16 * NT, MST = CB + 4;  // floats of a block column per state; LDS stride per state (bank spread, 16-byte aligned)
  __shared__ __align__(16) float mst[16 * MST];
  if (Mcopy && nv == 4 * NT && Lc.sk == 1 && (Lc.sb & 3) == 0 && (reinterpret_cast<uintptr_t>(Mcopy) & 15) == 0) {
    float* mine = mst + (lane >> 2) * MST;
#pragma unroll
    for (int J = 0; J < NT; ++J) {
#pragma unroll
      for (int I = J; I < NT; ++I)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if (I > J || cc <= r) mine[cc * 4 * NT + 4 * I + r] = t[I][J][cc];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        if (cc < r) mine[r * 4 * NT + 4 * J + cc] = t[J][J][cc];  // above the diagonal inside the diagonal tile
#pragma unroll
      for (int I = 0; I < J; ++I) *reinterpret_cast<f32x4_t*>(mine + r * 4 * NT + 4 * I) = t[J][I];  // M[4I + c][4J + r] = M[4J + r][4I + c]
      __syncthreads();
#pragma unroll 3
      for (int c0 = 0; c0 < 16 * 4 * NT; c0 += 64) {
        const int ch = c0 + lane, st = ch / (4 * NT), piece = ch - st * (4 * NT);
        const long gs = (long)blockIdx.x * 16 + st;
        if (ch < 16 * 4 * NT && gs < B)
          // streamed out past the L2 (nontemporal): 269 -> 237 us per launch at 65 536 states
          __builtin_nontemporal_store(*reinterpret_cast<const f32x4_t*>(mst + st * MST + piece * 4), reinterpret_cast<f32x4_t*>(Mcopy + gs * Lc.sb + (long)J * CB + piece * 4));
      }
      __syncthreads();
    }
  } else if (Mcopy && live) {  // any other shape: the lower triangle, column by column
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int I = J; I < NT; ++I) {
          const int row = 4 * I + r, col = 4 * J + cc;
          if (row < nv && col <= row) Mcopy[((long)col * nv + row) * Lc.sk + state * Lc.sb] = t[I][J][cc];
        }
  }
  float y[NT];
#pragma unroll
  for (int I = 0; I < NT; ++I) {
    const int row = 4 * I + r;
    float b = 0.0f;
    if (live && row < nv) {
      if (tau) b = tau[(long)row * Lv.sk + state * Lv.sb];
      if (c) b -= c[(long)row * Lv.sk + state * Lv.sb];
    }
    y[I] = b;
  }
  bool bad = false;
#pragma unroll
  for (int J = 0; J < NT; ++J) {
    // (1) diagonal tile: unblocked Cholesky inside the quad
    float dinv[4];
#define RBD_DIAG_STEP(K)                                                              \
    {                                                                                 \
      const float d = quad_bcast<K>(t[J][J][K]);                                      \
      bad |= !(d > 0.0f);                                                             \
      const float rs = 1.0f / SqrtT<float>::f(d);                                     \
      dinv[K] = rs;                                                                   \
      const float lk = (r == K) ? d * rs : t[J][J][K] * rs;                           \
      t[J][J][K] = lk;                                                                \
      if (K < 1) t[J][J][1] -= lk * quad_bcast<1>(lk);                                \
      if (K < 2) t[J][J][2] -= lk * quad_bcast<2>(lk);                                \
      if (K < 3) t[J][J][3] -= lk * quad_bcast<3>(lk);                                \
    }
    RBD_DIAG_STEP(0) RBD_DIAG_STEP(1) RBD_DIAG_STEP(2) RBD_DIAG_STEP(3)
#undef RBD_DIAG_STEP
    // entries of the factored diagonal tile, uniform inside the quad: Ld[k][m] = L[4J+k][4J+m], m < k
    const float l10 = quad_bcast<1>(t[J][J][0]), l20 = quad_bcast<2>(t[J][J][0]), l30 = quad_bcast<3>(t[J][J][0]);
    const float l21 = quad_bcast<2>(t[J][J][1]), l31 = quad_bcast<3>(t[J][J][1]), l32 = quad_bcast<3>(t[J][J][2]);
    // (2) panel: X = T L_d^-T, one row per lane
#pragma unroll
    for (int I = J + 1; I < NT; ++I) {
      const float x0 = t[I][J][0] * dinv[0];
      const float x1 = (t[I][J][1] - x0 * l10) * dinv[1];
      const float x2 = (t[I][J][2] - x0 * l20 - x1 * l21) * dinv[2];
      const float x3 = (t[I][J][3] - x0 * l30 - x1 * l31 - x2 * l32) * dinv[3];
      t[I][J][0] = x0; t[I][J][1] = x1; t[I][J][2] = x2; t[I][J][3] = x3;
    }
    // (3) trailing update on the matrix cores: T(I,J') -= X_I X_J'^T for J < J' <= I
#pragma unroll
    for (int Jp = J + 1; Jp < NT; ++Jp) {
      const float n0 = -t[Jp][J][0], n1 = -t[Jp][J][1], n2 = -t[Jp][J][2], n3 = -t[Jp][J][3];
#pragma unroll
      for (int I = Jp; I < NT; ++I) {
        f32x4_t acc = t[I][Jp];
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n0, t[I][J][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n1, t[I][J][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n2, t[I][J][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n3, t[I][J][3], acc, 0, 0, 0);
        t[I][Jp] = acc;
      }
    }
    // forward substitution for this block column: y_J <- L_d^-1 y_J, then y_I -= X(I,J) y_J
    {
      float v0 = quad_bcast<0>(y[J]) * dinv[0];
      y[J] = (r == 0) ? v0 : y[J] - t[J][J][0] * v0;
      float v1 = quad_bcast<1>(y[J]) * dinv[1];
      y[J] = (r == 1) ? v1 : ((r > 1) ? y[J] - t[J][J][1] * v1 : y[J]);
      float v2 = quad_bcast<2>(y[J]) * dinv[2];
      y[J] = (r == 2) ? v2 : ((r > 2) ? y[J] - t[J][J][2] * v2 : y[J]);
      float v3 = quad_bcast<3>(y[J]) * dinv[3];
      y[J] = (r == 3) ? v3 : y[J];
#pragma unroll
      for (int I = J + 1; I < NT; ++I) y[I] -= t[I][J][0] * v0 + t[I][J][1] * v1 + t[I][J][2] * v2 + t[I][J][3] * v3;
    }
  }
  if (live && bad && r == 0) atomicOr(notpd, 1);
  // backward substitution L' x = y, block columns in reverse
#pragma unroll
  for (int J = NT - 1; J >= 0; --J) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;  // (X(I,J)' x_I)[c], partial per lane (row)
#pragma unroll
    for (int I = J + 1; I < NT; ++I) { a0 += t[I][J][0] * y[I]; a1 += t[I][J][1] * y[I]; a2 += t[I][J][2] * y[I]; a3 += t[I][J][3] * y[I]; }
    a0 = quad_sum(a0); a1 = quad_sum(a1); a2 = quad_sum(a2); a3 = quad_sum(a3);
    float yj = y[J] - ((r == 0) ? a0 : (r == 1) ? a1 : (r == 2) ? a2 : a3);
    // x_k = (y_k - sum_{m>k} L[m][k] x_m) / L[k][k] inside the diagonal tile: L[m][k] is register k of lane m
    const float d0 = quad_bcast<0>(t[J][J][0]), d1 = quad_bcast<1>(t[J][J][1]), d2 = quad_bcast<2>(t[J][J][2]), d3 = quad_bcast<3>(t[J][J][3]);
    // (cross-lane reads are issued unconditionally — a DPP source lane that a divergent branch has masked off reads as 0)
    const float b30 = quad_bcast<3>(t[J][J][0]), b31 = quad_bcast<3>(t[J][J][1]), b32 = quad_bcast<3>(t[J][J][2]);
    const float b20 = quad_bcast<2>(t[J][J][0]), b21 = quad_bcast<2>(t[J][J][1]);
    const float x3 = quad_bcast<3>(yj) / d3;
    const float l3r = (r == 0) ? b30 : (r == 1) ? b31 : b32;
    yj = (r == 3) ? x3 : yj - l3r * x3;
    const float x2 = quad_bcast<2>(yj) / d2;
    const float l2r = (r == 0) ? b20 : b21;
    yj = (r == 2) ? x2 : ((r < 2) ? yj - l2r * x2 : yj);
    const float x1 = quad_bcast<1>(yj) / d1;
    const float l1r = quad_bcast<1>(t[J][J][0]);
    yj = (r == 1) ? x1 : ((r < 1) ? yj - l1r * x1 : yj);
    const float x0 = quad_bcast<0>(yj) / d0;
    yj = (r == 0) ? x0 : yj;
    y[J] = yj;
  }
#pragma unroll
  for (int I = 0; I < NT; ++I) {
    const int row = 4 * I + r;
    if (live && row < nv) x[(long)row * Lv.sk + state * Lv.sb] = y[I];
  }
  if (Lout && live) {
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int row = 4 * I + r, col = 4 * J + cc;
          if (row < nv && col <= row) Lout[((long)col * nv + row) * Lm.sk + layout_base(Lm, state)] = t[I][J][cc];
        }
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA tile Cholesky, fp64: 4 states per wavefront on v_mfma_f64_4x4x4f64 (4 blocks of D(4x4) = A(4x4) B(4x4) + C, one f64 per lane and
// operand).  Lane layout of that instruction, probed on gfx950 (scripts/ubench/mfma_f64_4x4x4.hip): lane = 16 x + 4 block + y, and
//   D[i][j] at (x = i, y = j)      B[k][j] at (x = k, y = j)      A[i][k] at (x = k, y = i)
// i.e. with tiles kept in the D layout (element (r, c) at x = r, y = c), mfma(a, b, c) = a' b + c.  The lower tiles are therefore kept
// TRANSPOSED, Tt(I,J) = T(I,J)': with W = (L_d^-1)' the panel is Xt_I = L_d^-1 Tt(I,J) = mfma(W, Tt(I,J), 0) and the trailing update
// Tt(I,J') -= X_J' X_I' = mfma(-Xt_J', Xt_I, Tt(I,J')) — one instruction per tile of all 4 states, no transposes.  The 4x4 diagonal tile
// is gathered through LDS to the 16 lanes of its state, which factor and invert it redundantly in registers.  Vectors come in two forms:
// Y (u[c] on the lanes with y = c) and X (u[r] on the lanes with x = r); a tile-vector product consumes one and leaves the other (sum over
// y: inside a quad of lanes; sum over x: two cross-row shuffles), and forward / backward substitution are arranged so that the forms chain.
// (dynamics_solve!, src/mechanism_algorithms.jl:764-819; replaces chol_reg_kernel<double>, 72 us at 4096 states.)
// ---------------------------------------------------------------------------------------------
RBD_DEV double sel16(const double (&a)[4][4], int x, int y) {  // a[x][y] for a run-time (x, y): selects, no indexed registers
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r = (x == i && y == j) ? a[i][j] : r;
  return r;
}
RBD_DEV double sum_over_y(double v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; }
RBD_DEV double sum_over_x(double v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }

template <int NT>
__global__ __launch_bounds__(64, 2) void chol_mfma64_kernel(int nv, long B, const double* __restrict__ Mg, const double* __restrict__ tau,
                                                           const double* __restrict__ c, double* __restrict__ xout, double* __restrict__ Lout,
                                                           Layout Lm, Layout Lv, int* __restrict__ notpd) {
  __shared__ double dsh[64];
  const int lane = threadIdx.x & 63, x = lane >> 4, blk = (lane >> 2) & 3, y = lane & 3;
  const long state = (long)blockIdx.x * 4 + blk;
  const bool live = state < B;
  const long mbase = live ? layout_base(Lm, state) : 0;
  double t[NT][NT];  // Tt(I,J), I >= J: this lane's element = M[4I + y][4J + x]
  // (column by column: the tiles (I, J), (I + 1, J), ... of a block column continue each other's cache lines — rows 4I + y of column 4J + x)
#pragma unroll
  for (int J = 0; J < NT; ++J)
#pragma unroll
    for (int I = J; I < NT; ++I) {
      int row = 4 * I + y, col = 4 * J + x;
      if (I == J && col > row) { const int tmp = row; row = col; col = tmp; }  // only the lower triangle of M is there to read
      double a = (row == col) ? 1.0 : 0.0;  // identity padding beyond nv
      if (live && row < nv) a = Mg[((long)col * nv + row) * Lm.sk + mbase];
      t[I][J] = a;
    }
  // (staging the block columns through LDS so that the lanes read whole runs of a column instead of 32-byte pieces: slower — 52 -> 64 us at
  // 4096 states, 637 -> 855 at 65 536: 18 more barriers and the index arithmetic outweigh the better-formed loads; v_rsq_f64 + two Newton steps for the pivots instead of sqrt and a division: also slower, 52 -> 55 us)
  double bY[NT];  // right-hand side, Y form
#pragma unroll
  for (int I = 0; I < NT; ++I) {
    const int row = 4 * I + y;
    double v = 0.0;
    if (live && row < nv) {
      if (tau) v = tau[(long)row * Lv.sk + state * Lv.sb];
      if (c) v -= c[(long)row * Lv.sk + state * Lv.sb];
    }
    bY[I] = v;
  }
  double linv[NT];  // L_d^-1 of block column J: element [x][y]
  bool bad = false;
#pragma unroll
  for (int J = 0; J < NT; ++J) {
    // (1) the diagonal tile on every lane of its state
    dsh[lane] = t[J][J];
    __syncthreads();
    double d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc <= r; ++cc) d[r][cc] = dsh[16 * r + 4 * blk + cc];
    __syncthreads();
    double L[4][4] = {}, Mi[4][4] = {};
    {
      const double p0 = d[0][0];
      bad |= !(p0 > 0.0);
      const double i0 = 1.0 / sqrt(p0);
      L[0][0] = p0 * i0; L[1][0] = d[1][0] * i0; L[2][0] = d[2][0] * i0; L[3][0] = d[3][0] * i0;
      const double p1 = d[1][1] - L[1][0] * L[1][0];
      bad |= !(p1 > 0.0);
      const double i1 = 1.0 / sqrt(p1);
      L[1][1] = p1 * i1; L[2][1] = (d[2][1] - L[2][0] * L[1][0]) * i1; L[3][1] = (d[3][1] - L[3][0] * L[1][0]) * i1;
      const double p2 = d[2][2] - L[2][0] * L[2][0] - L[2][1] * L[2][1];
      bad |= !(p2 > 0.0);
      const double i2 = 1.0 / sqrt(p2);
      L[2][2] = p2 * i2; L[3][2] = (d[3][2] - L[3][0] * L[2][0] - L[3][1] * L[2][1]) * i2;
      const double p3 = d[3][3] - L[3][0] * L[3][0] - L[3][1] * L[3][1] - L[3][2] * L[3][2];
      bad |= !(p3 > 0.0);
      const double i3 = 1.0 / sqrt(p3);
      L[3][3] = p3 * i3;
      Mi[0][0] = i0; Mi[1][1] = i1; Mi[2][2] = i2; Mi[3][3] = i3;
      Mi[1][0] = -L[1][0] * Mi[0][0] * i1;
      Mi[2][0] = -(L[2][0] * Mi[0][0] + L[2][1] * Mi[1][0]) * i2;
      Mi[2][1] = -L[2][1] * Mi[1][1] * i2;
      Mi[3][0] = -(L[3][0] * Mi[0][0] + L[3][1] * Mi[1][0] + L[3][2] * Mi[2][0]) * i3;
      Mi[3][1] = -(L[3][1] * Mi[1][1] + L[3][2] * Mi[2][1]) * i3;
      Mi[3][2] = -L[3][2] * Mi[2][2] * i3;
    }
    t[J][J] = sel16(L, y, x);            // Tt(J,J) = L_d': element (x, y) = L_d[y][x]
    linv[J] = sel16(Mi, x, y);
    const double W = sel16(Mi, y, x);    // (L_d^-1)'
    // (2) panel, (3) trailing update
#pragma unroll
    for (int I = J + 1; I < NT; ++I) t[I][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(W, t[I][J], 0.0, 0, 0, 0);
#pragma unroll
    for (int Jp = J + 1; Jp < NT; ++Jp) {
      const double nx = -t[Jp][J];
#pragma unroll
      for (int I = Jp; I < NT; ++I) t[I][Jp] = __builtin_amdgcn_mfma_f64_4x4x4f64(nx, t[I][J], t[I][Jp], 0, 0, 0);
    }
  }
  if (live && bad && x == 0 && y == 0) atomicOr(notpd, 1);
  // forward substitution L yv = b (left-looking): yv_I = L_d^-1 (b_I - sum_{J<I} L(I,J) yv_J);  L(I,J) u = Tt(I,J)' u: u in X form, sum over x
  double vX[NT];  // yv, then x, in X form ... and their Y forms where the next product needs them
#pragma unroll
  for (int I = 0; I < NT; ++I) {
    double part = 0.0;
#pragma unroll
    for (int J = 0; J < I; ++J) part += t[I][J] * vX[J];
    const double rhs = bY[I] - sum_over_x(part);              // Y form
    vX[I] = sum_over_y(linv[I] * rhs);                          // (L_d^-1 rhs)[x]: element [x][y] times rhs[y], summed over y -> X form
  }
  // backward substitution L' xv = yv: xv_J = L_d^-T (yv_J - sum_{I>J} L(I,J)' xv_I);  L(I,J)' u = Tt(I,J) u: u in Y form, sum over y
  double xY[NT];
#pragma unroll
  for (int J = NT - 1; J >= 0; --J) {
    double part = 0.0;
#pragma unroll
    for (int I = J + 1; I < NT; ++I) part += t[I][J] * xY[I];
    const double rhs = vX[J] - sum_over_y(part);               // X form
    xY[J] = sum_over_x(linv[J] * rhs);                          // (L_d^-T rhs)[y] = sum_x L_d^-1[x][y] rhs[x] -> Y form
  }
#pragma unroll
  for (int J = 0; J < NT; ++J) {
    const int row = 4 * J + y;
    if (live && x == 0 && row < nv) xout[(long)row * Lv.sk + state * Lv.sb] = xY[J];
  }
  if (Lout && live) {
    const long lbase = layout_base(Lm, state);
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) {
        const int row = 4 * I + y, col = 4 * J + x;
        if (row < nv && col <= row) Lout[((long)col * nv + row) * Lm.sk + lbase] = t[I][J];
      }
  }
}

template <typename T> struct MfmaChol {
  static bool launch(int, long, const void*, const void*, const void*, void*, void*, Layout, Layout, int*, hipStream_t, void*, Layout) { return false; }
};
template <> struct MfmaChol<double> {
  static bool launch(int nv, long B, const void* M, const void* tau, const void* c, void* x, void* Lout, Layout Lm, Layout Lv, int* notpd,
                     hipStream_t s, void* Mcopy, Layout) {
    if (Mcopy) return false;
    const dim3 grid((unsigned)((B + 3) / 4));
#define RBD_CHOL_MFMA64(NT)                                                                                                            \
    if (nv <= 4 * NT) {                                                                                                               \
      hipLaunchKernelGGL((chol_mfma64_kernel<NT>), grid, dim3(64), 0, s, nv, B, (const double*)M, (const double*)tau, (const double*)c, \
                         (double*)x, (double*)Lout, Lm, Lv, notpd);                                                                   \
      return true;                                                                                                                    \
    }
    RBD_CHOL_MFMA64(1) RBD_CHOL_MFMA64(2) RBD_CHOL_MFMA64(4) RBD_CHOL_MFMA64(6) RBD_CHOL_MFMA64(8) RBD_CHOL_MFMA64(9) RBD_CHOL_MFMA64(10)
#undef RBD_CHOL_MFMA64
    return false;
  }
};
template <> struct MfmaChol<float> {
  static bool launch(int nv, long B, const void* M, const void* tau, const void* c, void* x, void* Lout, Layout Lm, Layout Lv, int* notpd,
                     hipStream_t s, void* Mcopy, Layout Lc) {
    const dim3 grid((unsigned)((B + 15) / 16));
#define RBD_CHOL_MFMA(NT)                                                                                                           \
    if (nv <= 4 * NT) {                                                                                                             \
      hipLaunchKernelGGL((chol_mfma_kernel<NT>), grid, dim3(64), 0, s, nv, B, (const float*)M, (const float*)tau, (const float*)c,  \
                         (float*)x, (float*)Lout, Lm, Lv, notpd, (float*)Mcopy, Lc);                                                \
      return true;                                                                                                                  \
    }
    RBD_CHOL_MFMA(1) RBD_CHOL_MFMA(2) RBD_CHOL_MFMA(4) RBD_CHOL_MFMA(6) RBD_CHOL_MFMA(8) RBD_CHOL_MFMA(9) RBD_CHOL_MFMA(10)
#undef RBD_CHOL_MFMA
    return false;
  }
};

template <typename T>
hipError_t launch_chol_solve(int nv, long B, const void* M, const void* tau, const void* c, void* x, void* Lout, Layout Lm, Layout Lv,
                             int* notpd, hipStream_t s, void* Mcopy, Layout Lc) {
  if (MfmaChol<T>::launch(nv, B, M, tau, c, x, Lout, Lm, Lv, notpd, s, Mcopy, Lc)) return hipGetLastError();  // fp32, nv <= 40: matrix cores
  if (Mcopy) return hipErrorInvalidValue;  // only the tile kernel re-emits M (chol_copies_m())
  const dim3 grid4((unsigned)((B + 3) / 4));
#define RBD_CHOL_REG(NVP)                                                                                                      \
  if (nv <= NVP) {                                                                                                            \
    hipLaunchKernelGGL((chol_reg_kernel<T, NVP>), grid4, dim3(256), 0, s, nv, B, (const T*)M, (const T*)tau, (const T*)c, (T*)x, \
                       (T*)Lout, Lm, Lv, notpd);                                                                              \
    return hipGetLastError();                                                                                                 \
  }
  RBD_CHOL_REG(8) RBD_CHOL_REG(16) RBD_CHOL_REG(24) RBD_CHOL_REG(32) RBD_CHOL_REG(40) RBD_CHOL_REG(48)
#undef RBD_CHOL_REG
  // larger systems: the LDS-resident kernel
  int stride = ((nv + 3) / 4) * 4;  // 4*odd words: 16-byte row alignment, rows spread over the banks
  if (((stride / 4) & 1) == 0) stride += 4;
  int wpb = 4;
  size_t shmem = (size_t)wpb * ((size_t)nv * stride + 64) * sizeof(T);
  while (shmem > 160u * 1024u && wpb > 1) {  // one CU's LDS: fewer states per workgroup for the largest systems
    wpb >>= 1;
    shmem = (size_t)wpb * ((size_t)nv * stride + 64) * sizeof(T);
  }
  if (shmem > 160u * 1024u) return hipErrorInvalidValue;
  if (shmem > 64u * 1024u) {  // above the default dynamic-LDS limit (nv 49..64 in fp64 already is): raise it once per size
    static std::atomic<size_t> raised{0};
    if (raised.load(std::memory_order_relaxed) < shmem) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_solve_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160u * 1024u));
      if (e != hipSuccess) return e;
      raised.store(160u * 1024u, std::memory_order_relaxed);
    }
  }
  const dim3 grid((unsigned)((B + wpb - 1) / wpb));
  hipLaunchKernelGGL(chol_solve_kernel<T>, grid, dim3(64 * wpb), shmem, s, nv, stride, B, (const T*)M, (const T*)tau, (const T*)c, (T*)x,
                     (T*)Lout, Lm, Lv, notpd);
  return hipGetLastError();
}
template hipError_t launch_chol_solve<double>(int, long, const void*, const void*, const void*, void*, void*, Layout, Layout, int*, hipStream_t, void*, Layout);
template hipError_t launch_chol_solve<float>(int, long, const void*, const void*, const void*, void*, void*, Layout, Layout, int*, hipStream_t, void*, Layout);
bool chol_copies_m(int es, int nv) { return es == 4 && nv <= 40; }

}  // namespace rbd

// ---------------------------------------------------------------------------------------------
// Loop-joint branch of dynamics! (src/mechanism_algorithms.jl:858-861 and dynamics_solve! :768-816): one thread per
// state.  K = constraint_jacobian! (:574-598), k = constraint_bias! (:630-673, Baumgarte term through the Linearized SE(3)
// PD law, src/pdcontrol.jl:109-122), then  L = chol(M), Y = K L^-T, z = L^-1(tau - c), A = Y Y', b = Y z + k,
// lambda = min-norm LS solution of A lambda = b (gelsy!, rcond 1e-10; restated as the truncated pseudo-inverse of the PSD
// matrix A via cyclic Jacobi), v̇ = M^-1 (tau - c - K' lambda).  Mechanisms with loops are small (four-bar: nv 3, nc 5), so
// this branch favours generality over speed: all matrices live in a per-state global scratch block.
// ---------------------------------------------------------------------------------------------
namespace rbd {


template <typename T>
__global__ __launch_bounds__(64) void loop_solve_kernel(LoopView<T> V, long B, int stabilize, const T* __restrict__ body, const T* __restrict__ Mg,
                                                        const T* __restrict__ cg, const T* __restrict__ tau, T* __restrict__ vdot,
                                                        T* __restrict__ lambda, T* __restrict__ Kg, T* __restrict__ kg, T* __restrict__ scratch,
                                                        long scratch_stride, Layout Lm, Layout Lv, Layout Lc, Layout Lk, double g0, double g1,
                                                        double g2, int* __restrict__ notpd) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  const int nv = V.nv, nc = V.nc, nb = V.nb;
  // Per-state work arrays.  The solve is a long chain of dependent small-matrix steps, so where the arrays live sets its speed:
  // in LDS (scratch == nullptr: one odd-strided column per thread, 100-cycle accesses) when a block's worth fits, else in the
  // HBM scratch of the workspace (one cache line per thread and access — 8x slower on the four-bar benchmark configuration).
  extern __shared__ double loop_lds_raw[];
  T* L = scratch ? scratch + st * scratch_stride : reinterpret_cast<T*>(loop_lds_raw) + (long)threadIdx.x * scratch_stride;  // nv*nv (column-major, lower)
  T* K = L + nv * nv;                    // nc*nv row-major
  T* Y = K + nc * nv;                    // nc*nv row-major
  T* A = Y + nc * nv;                    // nc*nc
  T* E = A + nc * nc;                    // nc*nc eigenvectors
  T* z = E + nc * nc;                    // nv
  T* rhs = z + nv;                       // nv
  T* bv = rhs + nv;                      // nc
  T* kk = bv + nc;                       // nc
  const T* bd = body + st * nb * 24;
  const T I3[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
  const T Z3[3] = {T(0), T(0), T(0)};
  for (int i = 0; i < nc * nv; ++i) K[i] = T(0);
  for (int l = 0; l < V.nloops; ++l) {
    const int32_t* li = V.li + 8 * l;
    const T* lr = V.lr + 64 * l;
    const int pred = li[0], succ = li[1], row0 = li[3], ncl = li[4];
    const T* HpR = pred >= 0 ? bd + pred * 24 : I3; const T* Hpp = pred >= 0 ? bd + pred * 24 + 9 : Z3;
    const T* HsR = succ >= 0 ? bd + succ * 24 : I3; const T* Hsp = succ >= 0 ? bd + succ * 24 + 9 : Z3;
    T FbR[9], Fbp[3], FaR[9], Fap[3];
    xf_compose(HpR, Hpp, lr, lr + 9, FbR, Fbp);        // before_to_root (src/mechanism_state.jl:707)
    xf_compose(HsR, Hsp, lr + 12, lr + 21, FaR, Fap);  // after_to_root  (:708, :791)
    T Tw[36];
    for (int ci = 0; ci < ncl; ++ci) xforce(FaR, Fap, lr + 28 + 6 * ci, Tw + 6 * ci);
    for (int e = li[5]; e < li[6]; ++e) {
      const int bj = V.path[2 * e], sign = V.path[2 * e + 1];
      const int t = V.jt[bj];
      const int nvj = joint_nv(t);
      const T* R = bd + bj * 24; const T* p = R + 9;
      for (int col = 0; col < nvj; ++col) {
        T sl[6], S[6];
        subspace_col(t, V.axis + 3 * bj, V.axis2 + 3 * bj, col, sl);
        xmotion(R, p, sl, S);
        const int vi = V.voff[bj] + col;
        for (int ci = 0; ci < ncl; ++ci) { const T d = dot6(Tw + 6 * ci, S); K[(row0 + ci) * nv + vi] = sign < 0 ? -d : d; }
      }
    }
    // constraint_bias!
    T Tp[6], Ts[6], Ap[6], As[6], cr[6], ba[6];
    const T grav[6] = {T(0), T(0), T(0), T(-g0), T(-g1), T(-g2)};  // the exported accelerations carry -gravity from the root
    for (int j = 0; j < 6; ++j) {
      Tp[j] = pred >= 0 ? bd[pred * 24 + 12 + j] : T(0); Ts[j] = succ >= 0 ? bd[succ * 24 + 12 + j] : T(0);
      Ap[j] = pred >= 0 ? bd[pred * 24 + 18 + j] - grav[j] : T(0); As[j] = succ >= 0 ? bd[succ * 24 + 18 + j] - grav[j] : T(0);
    }
    se3_comm(Ts, Tp, cr);
    for (int j = 0; j < 6; ++j) ba[j] = cr[j] + (As[j] - Ap[j]);
    if (stabilize) {
      // Tn = inv(Fb) * Fa ; joint twist in frame_after ; Linearized SE(3) PD ; back to the root frame
      T TnR[9], d3[3], Tnp[3], jt[6], jl[6], stab[6], sw[6], Rtp[3];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) TnR[3 * i + j] = FbR[i] * FaR[j] + FbR[3 + i] * FaR[3 + j] + FbR[6 + i] * FaR[6 + j];
      for (int k = 0; k < 3; ++k) d3[k] = Fap[k] - Fbp[k];
      matTvec3(FbR, d3, Tnp);
      for (int j = 0; j < 6; ++j) jt[j] = Ts[j] - Tp[j];
      xmotion_inv(FaR, Fap, jt, jl);
      const T psi[3] = {(TnR[7] - TnR[5]) / 2, (TnR[2] - TnR[6]) / 2, (TnR[3] - TnR[1]) / 2};  // spatial/util.jl:178-183
      matTvec3(TnR, Tnp, Rtp);
      for (int i = 0; i < 3; ++i) { stab[i] = -lr[24] * psi[i] - lr[25] * jl[i]; stab[3 + i] = -lr[26] * Rtp[i] - lr[27] * jl[3 + i]; }
      xmotion(FaR, Fap, stab, sw);
      for (int j = 0; j < 6; ++j) ba[j] -= sw[j];
    }
    for (int ci = 0; ci < ncl; ++ci) kk[row0 + ci] = dot6(Tw + 6 * ci, ba);
  }
  // L = chol(M)
  for (int j = 0; j < nv; ++j)
    for (int i = j; i < nv; ++i) L[j * nv + i] = Mg[((long)j * nv + i) * Lm.sk + st * Lm.sb];
  bool bad = false;
  for (int j = 0; j < nv; ++j) {
    T d = L[j * nv + j];
    for (int k = 0; k < j; ++k) d -= L[k * nv + j] * L[k * nv + j];
    if (!(d > T(0))) bad = true;
    d = SqrtT<T>::f(d);
    L[j * nv + j] = d;
    for (int i = j + 1; i < nv; ++i) {
      T s = L[j * nv + i];
      for (int k = 0; k < j; ++k) s -= L[k * nv + i] * L[k * nv + j];
      L[j * nv + i] = s / d;
    }
  }
  if (bad) atomicOr(notpd, 1);
  auto fwd = [&](T* x) { for (int i = 0; i < nv; ++i) { T s = x[i]; for (int k = 0; k < i; ++k) s -= L[k * nv + i] * x[k]; x[i] = s / L[i * nv + i]; } };
  auto bwd = [&](T* x) { for (int i = nv - 1; i >= 0; --i) { T s = x[i]; for (int k = i + 1; k < nv; ++k) s -= L[i * nv + k] * x[k]; x[i] = s / L[i * nv + i]; } };
  for (int i = 0; i < nv; ++i) {
    const T t = tau ? tau[(long)i * Lv.sk + st * Lv.sb] : T(0);
    z[i] = t - cg[(long)i * Lv.sk + st * Lv.sb];
    rhs[i] = z[i];
  }
  if (nc > 0) {
    fwd(z);
    for (int ci = 0; ci < nc; ++ci) { for (int k = 0; k < nv; ++k) Y[ci * nv + k] = K[ci * nv + k]; fwd(Y + ci * nv); }
    for (int i = 0; i < nc; ++i) {
      for (int j = 0; j < nc; ++j) { T s = T(0); for (int k = 0; k < nv; ++k) s += Y[i * nv + k] * Y[j * nv + k]; A[i * nc + j] = s; }
      T s = kk[i]; for (int k = 0; k < nv; ++k) s += Y[i * nv + k] * z[k]; bv[i] = s;
    }
    // λ = A⁺ bv, the minimum-norm solution LAPACK's gelsy! returns for the (often rank-deficient) PSD Schur matrix A = Y Y'.
    // The eigen-decomposition is done on the SMALLER Gram matrix: A itself (nc x nc) when nc <= nv, else G = Y'Y (nv x nv) —
    // A and G share their non-zero eigenvalues s_j, A's eigenvectors are Y v_j / sqrt(s_j), so A⁺ bv = Y V diag(1/s²) V' Y' bv.
    // (A four-bar linkage has nc = 5, nv = 3: 3 Jacobi rotations per sweep instead of 10.)
    const bool gram = nv < nc;
    const int n = gram ? nv : nc;
    if (gram) {
      for (int i = 0; i < nv; ++i)
        for (int j = 0; j < nv; ++j) { T sacc = T(0); for (int c = 0; c < nc; ++c) sacc += Y[c * nv + i] * Y[c * nv + j]; A[i * n + j] = sacc; }
    }
    // cyclic Jacobi eigen-decomposition of the PSD matrix (n x n, in A; eigenvectors in the columns of E)
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) E[i * n + j] = (i == j) ? T(1) : T(0);
    const T tiny = sizeof(T) == 8 ? T(1e-34) : T(1e-16);
    for (int sweep = 0; sweep < 60; ++sweep) {
      T off = T(0), dg = T(0);
      for (int i = 0; i < n; ++i) { dg += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
      if (off <= dg * tiny) break;
      for (int p = 0; p < n - 1; ++p)
        for (int q = p + 1; q < n; ++q) {
          const T apq = A[p * n + q];
          if (apq == T(0)) continue;
          const T theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
          const T at = theta >= T(0) ? theta : -theta;
          const T t = (theta >= T(0) ? T(1) : T(-1)) / (at + SqrtT<T>::f(theta * theta + 1));
          const T c = 1 / SqrtT<T>::f(t * t + 1), sn = t * c;
          for (int k = 0; k < n; ++k) { const T a = A[k * n + p], b2 = A[k * n + q]; A[k * n + p] = c * a - sn * b2; A[k * n + q] = sn * a + c * b2; }
          for (int k = 0; k < n; ++k) { const T a = A[p * n + k], b2 = A[q * n + k]; A[p * n + k] = c * a - sn * b2; A[q * n + k] = sn * a + c * b2; }
          for (int k = 0; k < n; ++k) { const T a = E[k * n + p], b2 = E[k * n + q]; E[k * n + p] = c * a - sn * b2; E[k * n + q] = sn * a + c * b2; }
        }
    }
    T emax = T(0);
    for (int i = 0; i < n; ++i) if (A[i * n + i] > emax) emax = A[i * n + i];
    const T rcond = sizeof(T) == 8 ? T(1e-10) : T(1e-6);
    if (gram) {
      // z = Y' bv (z itself is no longer needed), w = V diag(1/s²) V' z, λ = Y w
      for (int i = 0; i < nv; ++i) { T sacc = T(0); for (int c = 0; c < nc; ++c) sacc += Y[c * nv + i] * bv[c]; z[i] = sacc; }
      T* w = E + n * n;                           // E was sized nc*nc >= nv*nv + nv when nv < nc
      for (int i = 0; i < nv; ++i) w[i] = T(0);
      for (int e = 0; e < n; ++e) {
        const T ev = A[e * n + e];
        if (ev > rcond * emax) {
          T d = T(0);
          for (int i = 0; i < n; ++i) d += E[i * n + e] * z[i];
          d /= ev * ev;
          for (int i = 0; i < n; ++i) w[i] += E[i * n + e] * d;
        }
      }
      for (int c = 0; c < nc; ++c) { T sacc = T(0); for (int i = 0; i < nv; ++i) sacc += Y[c * nv + i] * w[i]; bv[c] = sacc; }
      for (int c = 0; c < nc; ++c) Y[c] = bv[c];  // λ where the common code below expects it
    } else {
      for (int i = 0; i < nc; ++i) Y[i] = T(0);    // reuse Y[0..nc) as lambda accumulator (Y no longer needed)
      for (int e = 0; e < nc; ++e) {
        const T ev = A[e * nc + e];
        if (ev > rcond * emax) {
          T d = T(0);
          for (int i = 0; i < nc; ++i) d += E[i * nc + e] * bv[i];
          d /= ev;
          for (int i = 0; i < nc; ++i) Y[i] += E[i * nc + e] * d;
        }
      }
    }
    for (int vi = 0; vi < nv; ++vi) { T s = T(0); for (int ci = 0; ci < nc; ++ci) s += K[ci * nv + vi] * Y[ci]; rhs[vi] -= s; }
    for (int ci = 0; ci < nc; ++ci) {
      if (lambda) lambda[(long)ci * Lc.sk + st * Lc.sb] = Y[ci];
      kg[(long)ci * Lc.sk + st * Lc.sb] = kk[ci];
      for (int vi = 0; vi < nv; ++vi) Kg[((long)vi * nc + ci) * Lk.sk + st * Lk.sb] = K[ci * nv + vi];
    }
  }
  fwd(rhs); bwd(rhs);
  for (int i = 0; i < nv; ++i) vdot[(long)i * Lv.sk + st * Lv.sb] = rhs[i];
}

template <typename T, int NV, int NC>
__global__ __launch_bounds__(64) void loop_solve_small_kernel(LoopView<T> V, long B, int stabilize, const T* __restrict__ body, const T* __restrict__ Mg,
                                                              const T* __restrict__ cg, const T* __restrict__ tau, T* __restrict__ vdot,
                                                              T* __restrict__ lambda, T* __restrict__ Kg, T* __restrict__ kg, Layout Lm, Layout Lv,
                                                              Layout Lc, Layout Lk, double g0, double g1, double g2, int* __restrict__ notpd) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  loop_solve_small_state<T, LoopView<T>, NV, NC>(V, st, stabilize, body + st * V.nb * 24, Mg + st * Lm.sb, Lm.sk, cg + st * Lv.sb, Lv.sk, tau, vdot, lambda, Kg, kg, Lv, Lc, Lk, g0, g1, g2, notpd);
}

// ---------------------------------------------------------------------------------------------
// Small loop mechanisms in ONE launch (the four-bar linkage of BASELINE configs[4]: 3 bodies, nv 3, nc 5): one thread per state runs the whole
// dynamics! of the reference — forward kinematics, bias accelerations, dynamics_bias! (RNEA), mass_matrix! (CRBA), constraint_jacobian!,
// constraint_bias!, the constrained solve (src/mechanism_algorithms.jl:845-864 with :484-498, :248-272, :574-673, :747-822).  Round 2 ran it as three
// launches (rnea_kernel, crba_kernel, loop_solve_small_kernel: 7.5 + 5.4 + 14.3 us, each a lone wavefront's latency plus a launch); here the first
// two are the prologue of the third.  Per-body kinematics, M and c go through the workspace buffers the three-launch form used (so that
// rbd_dynamics_result finds them) and are read back by the thread that wrote them.  Bodies are visited in the reference's order (parents first);
// at most NB of them, 1-dof or fixed tree joints.
// ---------------------------------------------------------------------------------------------
template <typename T, int NB, int NV, int NC>
__global__ __launch_bounds__(64) void loop_fused_small_kernel(LoopView<T> V, long B, int stabilize, const T* __restrict__ q, const T* __restrict__ v,
                                                              const T* __restrict__ tau, const T* __restrict__ fext, T* body, T* Mg, T* cg,
                                                              T* __restrict__ vdot, T* __restrict__ qdot, T* __restrict__ lambda, T* __restrict__ Kg,
                                                              T* __restrict__ kg, Layout Lq, Layout Lm, Layout Lv, Layout Lf, Layout Lc, Layout Lk, double g0,
                                                              double g1, double g2, int* __restrict__ notpd) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  loop_fused_small_state<T, LoopView<T>, NB, NV, NC>(V, st, stabilize, q, v, tau, fext, body, Mg, cg, vdot, qdot, lambda, Kg, kg, Lq, Lm, Lv, Lf, Lc, Lk, g0, g1, g2, notpd);
}

// one launch for small loop mechanisms (see loop_fused_small_kernel); false: out of its scope, the caller takes the three-launch route
template <typename T>
bool launch_loop_fused(const LoopView<T>& V, long B, int stabilize, const void* q, const void* v, const void* tau, const void* fext, void* body, void* M, void* c,
                       void* vdot, void* qdot, void* lambda, void* K, void* k, Layout Lq, Layout Lm, Layout Lv, Layout Lf, Layout Lc, Layout Lk,
                       const double* gravity, int* notpd, hipStream_t s) {
  if (!(V.nb <= 4 && V.nv <= 4 && V.nc <= 6 && V.nv < V.nc && V.xi != nullptr)) return false;
  const dim3 grid((unsigned)((B + 63) / 64)), block(64);
#define RBD_FUSED(NVV)                                                                                                                              \
  hipLaunchKernelGGL((loop_fused_small_kernel<T, 4, NVV, 6>), grid, block, 0, s, V, B, stabilize, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, \
                     (T*)body, (T*)M, (T*)c, (T*)vdot, (T*)qdot, (T*)lambda, (T*)K, (T*)k, Lq, Lm, Lv, Lf, Lc, Lk, gravity[0], gravity[1], gravity[2], notpd)
  if (V.nv <= 3) RBD_FUSED(3); else RBD_FUSED(4);
#undef RBD_FUSED
  return true;
}
template bool launch_loop_fused<double>(const LoopView<double>&, long, int, const void*, const void*, const void*, const void*, void*, void*, void*, void*, void*, void*, void*, void*, Layout, Layout, Layout, Layout, Layout, Layout, const double*, int*, hipStream_t);
template bool launch_loop_fused<float>(const LoopView<float>&, long, int, const void*, const void*, const void*, const void*, void*, void*, void*, void*, void*, void*, void*, void*, Layout, Layout, Layout, Layout, Layout, Layout, const double*, int*, hipStream_t);

template <typename T>
hipError_t launch_loop_solve(const LoopView<T>& V, long B, int stabilize, const void* body, const void* M, const void* c, const void* tau, void* vdot,
                             void* lambda, void* K, void* k, void* scratch, long scratch_stride, Layout Lm, Layout Lv, Layout Lc, Layout Lk,
                             const double* gravity, int* notpd, hipStream_t s) {
  if (V.nv <= 3 && V.nc <= 6 && V.nv < V.nc) {  // the four-bar linkage itself (nv 3, nc 5): a 3 x 3 eigenproblem, three rotations per sweep
    hipLaunchKernelGGL((loop_solve_small_kernel<T, 3, 6>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, V, B, stabilize, (const T*)body, (const T*)M,
                       (const T*)c, (const T*)tau, (T*)vdot, (T*)lambda, (T*)K, (T*)k, Lm, Lv, Lc, Lk, gravity[0], gravity[1], gravity[2], notpd);
    return hipGetLastError();
  }
  if (V.nv <= 4 && V.nc <= 6 && V.nv < V.nc) {  // small loop mechanisms: everything in registers
    hipLaunchKernelGGL((loop_solve_small_kernel<T, 4, 6>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, V, B, stabilize, (const T*)body, (const T*)M,
                       (const T*)c, (const T*)tau, (T*)vdot, (T*)lambda, (T*)K, (T*)k, Lm, Lv, Lc, Lk, gravity[0], gravity[1], gravity[2], notpd);
    return hipGetLastError();
  }
  const long lds_stride = scratch_stride | 1;  // odd: consecutive threads start in different LDS banks
  const size_t lds = (size_t)64 * lds_stride * sizeof(T);
  if (lds <= 64 * 1024)
    hipLaunchKernelGGL(loop_solve_kernel<T>, dim3((unsigned)((B + 63) / 64)), dim3(64), lds, s, V, B, stabilize, (const T*)body, (const T*)M, (const T*)c,
                       (const T*)tau, (T*)vdot, (T*)lambda, (T*)K, (T*)k, (T*)nullptr, lds_stride, Lm, Lv, Lc, Lk, gravity[0], gravity[1],
                       gravity[2], notpd);
  else
    hipLaunchKernelGGL(loop_solve_kernel<T>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, V, B, stabilize, (const T*)body, (const T*)M, (const T*)c,
                       (const T*)tau, (T*)vdot, (T*)lambda, (T*)K, (T*)k, (T*)scratch, scratch_stride, Lm, Lv, Lc, Lk, gravity[0], gravity[1],
                       gravity[2], notpd);
  return hipGetLastError();
}
template hipError_t launch_loop_solve<double>(const LoopView<double>&, long, int, const void*, const void*, const void*, const void*, void*, void*, void*, void*, void*, long, Layout, Layout, Layout, Layout, const double*, int*, hipStream_t);
template hipError_t launch_loop_solve<float>(const LoopView<float>&, long, int, const void*, const void*, const void*, const void*, void*, void*, void*, void*, void*, long, Layout, Layout, Layout, Layout, const double*, int*, hipStream_t);

}  // namespace rbd

// ---------------------------------------------------------------------------------------------
// Munthe-Kaas RK4 stage kernel — the consumer of dynamics!: `simulate` (src/simulate.jl:36-55) =
// MuntheKaasIntegrator.step (src/ode_integrators.jl:233-299) with the runge_kutta_4 tableau (:48-55).
// One lane per (state, joint), same slot mapping as the dynamics kernels.  Per stage s = 0..4:
//   s >= 1 : vd[s-1] <- vdot_prev (the dynamics result of the previous stage state), and
//            ϕd[s-1] <- time derivative of the local coordinates at the previous stage state (local_coordinates!)
//   s == 0 : (q0, v0) <- (q, v)                                  [base point of the step]
//   s <= 3 : ϕ = Δt Σ_j a[s][j] ϕd[j],  v = v0 + Δt Σ_j a[s][j] vd[j]
//   s == 4 : ϕ = Δt Σ_j b[j] ϕd[j],     v = v0 + Δt Σ_j b[j] vd[j]
//   (q, v) <- (global_coordinates!(q0, ϕ), v)
// local/global coordinates: joint_types.jl:9-18 (default), sin_cos_revolute.jl:173-196, quaternion_spherical.jl:139-154,
// quaternion_floating.jl:205-249 with log_with_time_derivative / exp of src/spatial/spatialmotion.jl:226-332 and
// rotation_vector_rate of src/spatial/util.jl:88-102.  Rotations are composed as quaternions (the reference goes through
// rotation matrices and converts back; identical up to rounding and the sign of the quaternion).
// ---------------------------------------------------------------------------------------------
namespace rbd {

// Two launches per stage.  mk_stage_kernel<T, 1>: the element-wise joint types (revolute, prismatic, sin-cos, planar), one lane per (state,
// body) as everywhere else — ~50 registers, 7-8 wavefronts per SIMD in flight.  mk_stage_heavy_kernel: the quaternion joints (SE(3) / SO(3)
// log and exp: ~2 000 instructions with their atan2 / sincos / divisions), ONE THREAD PER STATE looping over the model's bodies — a humanoid
// has one such joint, and one-lane-per-(state, body) left 2 of 64 lanes busy through all of it (measured as one kernel: 94 us per stage at
// 65 536 Atlas states, 89 of them this branch).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void mk_stage_kernel(DevModel M, long B, int stage, T dt, T* __restrict__ q, T* __restrict__ v,
                                                       const T* __restrict__ vdot_prev, MkBuffers W, Layout Lq, Layout Lv, int close_prev) {
  Body<T> b;
  load_body(M, B, b);
  const bool heavy = b.jtype == RBD_JOINT_QUAT_FLOATING || b.jtype == RBD_JOINT_QUAT_SPHERICAL;
  if (heavy != (MODE == 2)) return;
  T qj[7], vj[6];
  load_joint_q(b, q, Lq, qj);
  load_joint_v(b, v, Lv, vj);
  if (close_prev) mk_stage_lane<T, MODE>(b, 4, dt, qj, vj, vdot_prev, W, q, v, Lq, Lv);  // the step before closes in this launch (as in the fused kernels)
  mk_stage_lane<T, MODE>(b, stage, dt, qj, vj, vdot_prev, W, q, v, Lq, Lv);
}
template <typename T>
__global__ __launch_bounds__(256) void mk_stage_heavy_kernel(DevModel M, long B, int stage, T dt, T* __restrict__ q, T* __restrict__ v,
                                                             const T* __restrict__ vdot_prev, MkBuffers W, Layout Lq, Layout Lv, int close_prev) {
  const long state = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (state >= B) return;
  auto one = [&](int jt, int qoff, int voff) {
    Body<T> b{};
    b.jtype = jt; b.qoff = qoff; b.voff = voff; b.state = state; b.valid = true;
    T qj[7], vj[6];
    load_joint_q(b, q, Lq, qj);
    load_joint_v(b, v, Lv, vj);
    if (close_prev) mk_stage_lane<T, 2>(b, 4, dt, qj, vj, vdot_prev, W, q, v, Lq, Lv);
    mk_stage_lane<T, 2>(b, stage, dt, qj, vj, vdot_prev, W, q, v, Lq, Lv);
  };
  if (M.nheavy <= 4) {
    // which joints, and where their coordinates are, from the kernel arguments: scanning the body table for them was one dependent global
    // load per body (31 for Atlas) in front of the one joint that has work
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < M.nheavy) one(M.heavy[k][0], M.heavy[k][1], M.heavy[k][2]);
  } else {
    for (int s = 0; s < M.nb; ++s) {
      const int32_t* ib = M.ib + s * IB_STRIDE;
      const int jt = ib[IB_JTYPE];  // wave-uniform
      if (jt != RBD_JOINT_QUAT_FLOATING && jt != RBD_JOINT_QUAT_SPHERICAL) continue;
      one(jt, ib[IB_QOFF], ib[IB_VOFF]);
    }
  }
}

template <typename T>
hipError_t launch_mk_stage(const DevModel& M, long B, int stage, double dt, void* q, void* v, const void* vdot_prev, const MkBuffers& W,
                           Layout Lq, Layout Lv, hipStream_t s, int close_prev) {
  hipLaunchKernelGGL((mk_stage_kernel<T, 1>), grid_for(M, B, 256), dim3(256), 0, s, M, B, stage, (T)dt, (T*)q, (T*)v, (const T*)vdot_prev, W, Lq, Lv, close_prev);
  if (M.maxnvj > 3 || M.has3dof)  // some joint is (or may be) a quaternion joint
    hipLaunchKernelGGL((mk_stage_heavy_kernel<T>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, M, B, stage, (T)dt, (T*)q, (T*)v, (const T*)vdot_prev, W, Lq, Lv, close_prev);
  return hipGetLastError();
}
template hipError_t launch_mk_stage<double>(const DevModel&, long, int, double, void*, void*, const void*, const MkBuffers&, Layout, Layout, hipStream_t, int);
template hipError_t launch_mk_stage<float>(const DevModel&, long, int, double, void*, void*, const void*, const MkBuffers&, Layout, Layout, hipStream_t, int);

}  // namespace rbd
