// rbd_lane.hpp — what the lane-per-body kernels share: the per-lane body record, joint kernels (joint_transform, motion subspace,
// q̇ map), batch-buffer access and the level-synchronous kinematics sweep.  Included by rbd_kernels.hip (one body per lane)
// and rbd_bank_kernels.hip (two bodies per lane).
#pragma once
#include "rbd_device.hpp"
#include "rbd_hip.h"

namespace rbd {

template <typename T> struct Body {
  // topology
  int parent, jtype, qoff, voff, level, nchild, orig;
  int child[IB_MAXCHILD];
  int plane;  // lane of the parent body (own lane if parent is the world)
  int flags;  // BFD_* of a re-rooted tree (0 otherwise)
  // lane bookkeeping
  int lane, sub, base;
  long state;
  bool valid;
};

template <typename T> RBD_DEV int joint_nq(int t) {
  return t == RBD_JOINT_QUAT_FLOATING ? 7 : t == RBD_JOINT_QUAT_SPHERICAL ? 4 : t == RBD_JOINT_PLANAR ? 3 : t == RBD_JOINT_SINCOS_REVOLUTE ? 2
         : (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC) ? 1 : 0;
}
RBD_DEV int joint_nv(int t) {
  return t == RBD_JOINT_QUAT_FLOATING ? 6 : (t == RBD_JOINT_QUAT_SPHERICAL || t == RBD_JOINT_PLANAR) ? 3
         : (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_SINCOS_REVOLUTE) ? 1 : 0;
}


template <typename T> RBD_DEV int child_sel(const Body<T>& b, int s) {
  return (s == 0) ? b.child[0] : (s == 1) ? b.child[1] : (s == 2) ? b.child[2] : (s == 3) ? b.child[3] : (s == 4) ? b.child[4] : (s == 5) ? b.child[5] : (s == 6) ? b.child[6] : b.child[7];
}
// Parents at level l-1 pull N values from their s-th child (all children of a parent sit at level l) and add them.
// Slot 0 (first child == next lane in DFS pre-order) is a DPP wave shift; further children use ds_bpermute.
// All moves are issued before any is consumed; only the takers add (under their exec mask).
template <typename T, int N> RBD_DEV void gather_add(const Body<T>& b, int l, int s, const T* give, T* acc) {
  const bool take = (b.level == l - 1) && (s < b.nchild);
  constexpr int CH = 9;  // moves in flight per batch (bounds the temporaries)
  if (s == 0) {
    T t[N];
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = from_next_lane(give[k]);
    if (take) {
      RBD_KEEP_BRANCH();
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] += t[k];
    }
  } else {
    const int c = child_sel(b, s);
    const int src = take ? b.base + c : b.lane;
#pragma unroll
    for (int k0 = 0; k0 < N; k0 += CH) {
      T tmp[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k)
        if (k0 + k < N) tmp[k] = shfl(give[k0 + k], src);
      if (take) {
        RBD_KEEP_BRANCH();
#pragma unroll
        for (int k = 0; k < CH; ++k)
          if (k0 + k < N) acc[k0 + k] += tmp[k];
      }
    }
  }
}
// Top-down hop: every lane reads N values of its parent's lane (DPP shift when all level-l parents are the previous lane).
template <typename T, int N> RBD_DEV void pull_parent(const DevModel& M, const Body<T>& b, int l, const T* x, T* out) {
  if ((M.perm_down >> l) & 1) {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = shfl(x[k], b.plane);
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = from_prev_lane(x[k]);
  }
}

template <typename T> RBD_DEV void load_body(const DevModel& M, long B, Body<T>& b) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  b.lane = threadIdx.x & 63;
  const long wave = tid >> 6;
  const int lps = M.lps;
  b.sub = b.lane & (lps - 1);
  b.base = b.lane - b.sub;
  b.state = wave * (64 / lps) + (b.lane / lps);
  b.valid = (b.sub < M.nb) && (b.state < B);
  const int body = b.sub < M.nb ? b.sub : 0;
  const int32_t* ib = M.ib + body * IB_STRIDE;
  b.parent = ib[IB_PARENT];
  b.jtype = ib[IB_JTYPE];
  b.qoff = ib[IB_QOFF];
  b.voff = ib[IB_VOFF];
  b.level = b.valid ? ib[IB_LEVEL] : -1;  // idle lanes never commit
  b.nchild = ib[IB_NCHILD];
  b.orig = ib[IB_ORIG];
  b.flags = ib[IB_FLAGS];
#pragma unroll
  for (int k = 0; k < IB_MAXCHILD; ++k) b.child[k] = ib[IB_CHILD0 + k];
  b.plane = b.parent >= 0 ? b.base + b.parent : b.lane;
}

// joint_transform(q) composed with joint_to_predecessor: XL = Xpred * Tj(q)
// (src/mechanism_state.jl:699; revolute.jl:59-62, prismatic.jl:69-73, quaternion_floating.jl:81-83,
//  sin_cos_revolute.jl:69-96, fixed.jl)
template <typename T> RBD_DEV void local_transform(const Body<T>& b, const T* rb, const T* qj, T* XR, T* Xp) {
  T Rj[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
  T pj[3] = {T(0), T(0), T(0)};
  const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]};
  if (b.jtype == RBD_JOINT_REVOLUTE) {
    T s, c;
    sincos_fast(qj[0], &s, &c);  // ~35 instructions in fp64 against ~130 of the library's (rbd_device.hpp; 1.6 ulp)
    rot_axis_sc(ax, s, c, Rj);
  } else if (b.jtype == RBD_JOINT_SINCOS_REVOLUTE) {
    rot_axis_sc(ax, qj[0], qj[1], Rj);
  } else if (b.jtype == RBD_JOINT_PRISMATIC) {
#pragma unroll
    for (int k = 0; k < 3; ++k) pj[k] = qj[0] * ax[k];
  } else if (b.jtype == RBD_JOINT_QUAT_FLOATING) {
    rot_quat(qj[0], qj[1], qj[2], qj[3], Rj);
    pj[0] = qj[4]; pj[1] = qj[5]; pj[2] = qj[6];
  } else if (b.jtype == RBD_JOINT_QUAT_SPHERICAL) {  // quaternion_spherical.jl:39-42
    rot_quat(qj[0], qj[1], qj[2], qj[3], Rj);
  } else if (b.jtype == RBD_JOINT_PLANAR) {  // planar.jl:65-70: rotate about x × y after translating in the x-y plane
    const T ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
    T az[3], s, c;
    cross3(ax, ay, az);
    sincos_t(qj[2], &s, &c);
    rot_axis_sc(az, s, c, Rj);
#pragma unroll
    for (int k = 0; k < 3; ++k) pj[k] = ax[k] * qj[0] + ay[k] * qj[1];
  }
  T XpR[9], Xpp[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) XpR[k] = rb[RB_XPR + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) Xpp[k] = rb[RB_XPP + k];
  matmul3(XpR, Rj, XR);
  matvec3(XpR, pj, Xp);
#pragma unroll
  for (int k = 0; k < 3; ++k) Xp[k] += Xpp[k];
}

// local joint twist S_local * v in frame_after (revolute.jl:64-68, prismatic.jl:75-79, quaternion_floating.jl:182-188)
template <typename T> RBD_DEV void local_joint_motion(const Body<T>& b, const T* rb, const T* vj, T* o) {
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = T(0);
  if (b.jtype == RBD_JOINT_REVOLUTE || b.jtype == RBD_JOINT_SINCOS_REVOLUTE) {
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = rb[RB_AXIS + k] * vj[0];
  } else if (b.jtype == RBD_JOINT_PRISMATIC) {
#pragma unroll
    for (int k = 0; k < 3; ++k) o[3 + k] = rb[RB_AXIS + k] * vj[0];
  } else if (b.jtype == RBD_JOINT_QUAT_FLOATING) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = vj[k];
  } else if (b.jtype == RBD_JOINT_QUAT_SPHERICAL) {  // quaternion_spherical.jl:98-104
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = vj[k];
  } else if (b.jtype == RBD_JOINT_PLANAR) {  // planar.jl:72-77: linear = x v1 + y v2, angular = (x × y) v3
    const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]}, ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
    T az[3];
    cross3(ax, ay, az);
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = az[k] * vj[2]; o[3 + k] = ax[k] * vj[0] + ay[k] * vj[1]; }
  }
}

// column k of the local motion subspace of a joint of type t (axis, axis2 in frame_before == frame_after axes):
// revolute.jl:83-89, prismatic.jl:93-99, quaternion_floating.jl:85-91, quaternion_spherical.jl:43-49, planar.jl:79-86
template <typename T> RBD_DEV void subspace_col(int t, const T* ax, const T* ay, int k, T* o) {
#pragma unroll
  for (int j = 0; j < 6; ++j) o[j] = T(0);
  if (t == RBD_JOINT_QUAT_FLOATING) {
#pragma unroll
    for (int j = 0; j < 6; ++j) o[j] = (j == k) ? T(1) : T(0);
  } else if (t == RBD_JOINT_QUAT_SPHERICAL) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = (j == k) ? T(1) : T(0);
  } else if (t == RBD_JOINT_PRISMATIC) {
    o[3] = ax[0]; o[4] = ax[1]; o[5] = ax[2];
  } else if (t == RBD_JOINT_PLANAR) {
    if (k == 2) { T az[3]; cross3(ax, ay, az); o[0] = az[0]; o[1] = az[1]; o[2] = az[2]; }
    else { const T* a = (k == 0) ? ax : ay; o[3] = a[0]; o[4] = a[1]; o[5] = a[2]; }
  } else if (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_SINCOS_REVOLUTE) {
    o[0] = ax[0]; o[1] = ax[1]; o[2] = ax[2];
  }
}

template <typename T> RBD_DEV void load_joint_q(const Body<T>& b, const T* __restrict__ q, Layout L, T* qj) {
  const int n = joint_nq<T>(b.jtype);
#pragma unroll
  for (int k = 0; k < 7; ++k) qj[k] = (b.valid && k < n) ? q[(long)(b.qoff + k) * L.sk + b.state * L.sb] : T(0);
}
template <typename T> RBD_DEV void load_joint_v(const Body<T>& b, const T* __restrict__ v, Layout L, T* vj) {
  const int n = joint_nv(b.jtype);
#pragma unroll
  for (int k = 0; k < 6; ++k) vj[k] = (v != nullptr && b.valid && k < n) ? v[(long)(b.voff + k) * L.sk + b.state * L.sb] : T(0);
}
template <typename T> RBD_DEV void load_body_wrench(const Body<T>& b, const T* __restrict__ f, Layout L, T* w) {
#pragma unroll
  for (int k = 0; k < 6; ++k) w[k] = (f != nullptr && b.valid) ? f[(long)(6 * b.orig + k) * L.sk + b.state * L.sb] : T(0);
}
template <typename T> RBD_DEV void store_joint_v(const Body<T>& b, T* __restrict__ out, Layout L, const T* x) {
  const int n = joint_nv(b.jtype);
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (b.valid && k < n) out[(long)(b.voff + k) * L.sk + b.state * L.sb] = x[k];
}

// velocity_to_configuration_derivative! (quaternion_floating.jl:126-136, spatial/util.jl:127-134, sin_cos_revolute.jl; default q̇ = v)
template <typename T> RBD_DEV void store_qdot(const Body<T>& b, T* __restrict__ qdot, Layout L, const T* qj, const T* vj) {
  if (qdot == nullptr || !b.valid) return;
  T o[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) o[k] = T(0);
  if (b.jtype == RBD_JOINT_REVOLUTE || b.jtype == RBD_JOINT_PRISMATIC) {
    o[0] = vj[0];
  } else if (b.jtype == RBD_JOINT_SINCOS_REVOLUTE) {
    o[0] = qj[1] * vj[0];
    o[1] = -qj[0] * vj[0];
  } else if (b.jtype == RBD_JOINT_QUAT_FLOATING) {
    const T w = qj[0], x = qj[1], y = qj[2], z = qj[3];
    o[0] = (-x * vj[0] - y * vj[1] - z * vj[2]) / 2;
    o[1] = (w * vj[0] - z * vj[1] + y * vj[2]) / 2;
    o[2] = (z * vj[0] + w * vj[1] - x * vj[2]) / 2;
    o[3] = (-y * vj[0] + x * vj[1] + w * vj[2]) / 2;
    T R[9];
    rot_quat(w, x, y, z, R);
    matvec3(R, vj + 3, o + 4);
  } else if (b.jtype == RBD_JOINT_QUAT_SPHERICAL) {  // quaternion_spherical.jl:74-78
    const T w = qj[0], x = qj[1], y = qj[2], z = qj[3];
    o[0] = (-x * vj[0] - y * vj[1] - z * vj[2]) / 2;
    o[1] = (w * vj[0] - z * vj[1] + y * vj[2]) / 2;
    o[2] = (z * vj[0] + w * vj[1] - x * vj[2]) / 2;
    o[3] = (-y * vj[0] + x * vj[1] + w * vj[2]) / 2;
  } else if (b.jtype == RBD_JOINT_PLANAR) {  // planar.jl velocity_to_configuration_derivative!: q̇_lin = Rot2(θ) v_lin
    T sn, cs;
    sincos_t(qj[2], &sn, &cs);
    o[0] = cs * vj[0] - sn * vj[1];
    o[1] = sn * vj[0] + cs * vj[1];
    o[2] = vj[2];
  }
  const int n = joint_nq<T>(b.jtype);
#pragma unroll
  for (int k = 0; k < 7; ++k)
    if (k < n) qdot[(long)(b.qoff + k) * L.sk + b.state * L.sb] = o[k];
}

// Top-down sweep 1: transforms to root and twists (update_transforms! src/mechanism_state.jl:687-700,
// update_twists_wrt_world! :769-780).  XR/Xp: local transform in, transform-to-root out.  vJ: joint twist in the
// root frame (T_b - T_parent).  Optionally also carries spatial accelerations (spatial_accelerations!
// src/mechanism_algorithms.jl:387-417): acc_local = S_local*v̇ in, acc = a_b out, with a_root = -gravity.
template <typename T, bool WITH_ACC>
RBD_DEV void sweep_kinematics(const DevModel& M, const Body<T>& b, const T* XR, const T* Xp, T* R, T* p, const T* tj_local, T* Tw,
                              T* vJ, const T* aj_local, T* acc) {
  // level 0 (the root transform is the identity): H = XL, T = vJ, a = -g + X a_joint.  Written for every lane; deeper
  // lanes overwrite at their level.  Inside the level loop the new values never read the registers they replace, so
  // the exec-masked block writes them in place (no copies).
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = XR[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = Xp[k];
  xmotion(R, p, tj_local, vJ);
#pragma unroll
  for (int k = 0; k < 6; ++k) Tw[k] = vJ[k];
  if (WITH_ACC) {
    xmotion(R, p, aj_local, acc);
    acc[3] -= T(M.gravity[0]); acc[4] -= T(M.gravity[1]); acc[5] -= T(M.gravity[2]);
  }
  for (int l = 1; l < M.nlevels; ++l) {
    T pR[9], pp[3], pT[6], pa[6];
    pull_parent<T, 9>(M, b, l, R, pR);
    pull_parent<T, 3>(M, b, l, p, pp);
    pull_parent<T, 6>(M, b, l, Tw, pT);
    if (WITH_ACC) pull_parent<T, 6>(M, b, l, acc, pa);
    if (b.level == l) {
      matmul3(pR, XR, R);
      matvec3(pR, Xp, p);
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] += pp[k];
      xmotion(R, p, tj_local, vJ);
#pragma unroll
      for (int k = 0; k < 6; ++k) Tw[k] = pT[k] + vJ[k];
      if (WITH_ACC) {
        // a_b = a_p + (-T_b) x T_p + X a_joint    (mechanism_algorithms.jl:414)
        T nT[6], cr[6], aj[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) nT[k] = -Tw[k];
        se3_comm(nT, pT, cr);
        xmotion(R, p, aj_local, aj);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = pa[k] + cr[k] + aj[k];
      }
    }
  }
}

}  // namespace rbd
namespace rbd {
// PD law of rbd_simulate_controlled on one lane's joint (stage state qj, vj): tau -= kp (q − q_des) + kd v for Revolute / Prismatic joints
template <typename T> RBD_DEV void pd_control_lane(const Body<T>& b, const MkFuse& F, Layout Lq, const T* qj, const T* vj, T* tj) {
  if (F.pd_kp == nullptr) return;  // uniform
  if (b.valid && (b.jtype == RBD_JOINT_REVOLUTE || b.jtype == RBD_JOINT_PRISMATIC)) {
    const T kp = ((const T*)F.pd_kp)[b.voff], kd = ((const T*)F.pd_kd)[b.voff];
    const T qd = F.pd_qdes ? ((const T*)F.pd_qdes)[(long)b.qoff * Lq.sk + b.state * Lq.sb] : T(0);
    tj[0] -= kp * (qj[0] - qd) + kd * vj[0];
  }
}
}  // namespace rbd
#include "rbd_integrator.hpp"
