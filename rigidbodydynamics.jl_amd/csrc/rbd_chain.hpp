// rbd_chain.hpp — chain-scheduled fused articulated-body algorithm (included by rbd_kernels.hip after the joint helpers).
//
// Same mathematics and outputs as aba_kernel (dynamics!, src/mechanism_algorithms.jl:845-864, in the root frame), different
// mapping.  aba_kernel gives every body its own lane and runs level-synchronous sweeps, so at each level ~90 % of the
// lanes execute masked-off work: right for small batches (it exposes 32 lanes of parallelism per state), wasteful for
// large ones.  Here a state owns only G lanes ("tracks", G = 4 for a humanoid).  The host cuts the tree into chains
// (longest path first) and packs them on the tracks (rbd_chain_plan.hpp); every track walks its chains one body per
// step, carrying the recursion state of the three passes in registers:
//
//   pass A (root -> leaves)  transform to root and twist of each body                (update_transforms!, update_twists_wrt_world!)
//   pass B (leaves -> root)  articulated inertia / bias force hand-off; U, 1/D, u of each joint go to an LDS "stash" row
//   pass C (root -> leaves)  spatial accelerations and v̇
//
// Per-body data is NOT kept between passes (there is no lane to keep it in): pass B walks the chain upwards by
// un-composing the forward kinematics (H_parent = H_body * X_joint^-1, T_parent = T_body - S q̇ — exact up to rounding),
// and pass C recomputes it downwards; only the joint's sin/cos and (U, 1/D, u) — 10 values per body — are stashed.
// Edges between bodies on different tracks (branch points, 4 of 30 for Atlas) go through LDS mailboxes.
// A wavefront is a whole workgroup, so the passes need no cross-wave barrier.
#pragma once

namespace rbd {

// joint-angle functions kept between the passes: revolute (sin q, cos q); sin-cos revolute (q1, q2); prismatic (q, -)
template <typename T> RBD_DEV void joint_sc(int jt, const T* qj, T* sc) {
  sc[0] = T(0); sc[1] = T(1);
  if (jt == RBD_JOINT_REVOLUTE) sincos_t(qj[0], &sc[0], &sc[1]);
  else if (jt == RBD_JOINT_SINCOS_REVOLUTE) { sc[0] = qj[0]; sc[1] = qj[1]; }
  else if (jt == RBD_JOINT_PRISMATIC) sc[0] = qj[0];
}
// local_transform (above) from the kept sin/cos; qj is only read for the 6-dof joint
template <typename T> RBD_DEV void local_transform_sc(int jt, const T* rb, const T* sc, const T* qj, T* XR, T* Xp) {
  T XpR[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) XpR[k] = rb[RB_XPR + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) XR[k] = XpR[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) Xp[k] = rb[RB_XPP + k];
  const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]};
  if (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_SINCOS_REVOLUTE) {
    T Rj[9];
    rot_axis_sc(ax, sc[0], sc[1], Rj);
    matmul3(XpR, Rj, XR);
  } else if (jt == RBD_JOINT_PRISMATIC) {
    T pj[3] = {sc[0] * ax[0], sc[0] * ax[1], sc[0] * ax[2]}, t[3];
    matvec3(XpR, pj, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) Xp[k] += t[k];
  } else if (jt == RBD_JOINT_QUAT_FLOATING) {
    T Rj[9], t[3];
    rot_quat(qj[0], qj[1], qj[2], qj[3], Rj);
    matmul3(XpR, Rj, XR);
    matvec3(XpR, qj + 4, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) Xp[k] += t[k];
  }
}
// C = A * B'
template <typename T> RBD_DEV void matmul3_nt(const T* A, const T* B, T* C) {
  T t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
#pragma unroll
  for (int k = 0; k < 9; ++k) C[k] = t[k];
}

template <typename T, int G>
__global__ __launch_bounds__(64) void aba_chain_kernel(ChainModel C, long B, const T* __restrict__ q, const T* __restrict__ v,
                                                      const T* __restrict__ tau, const T* __restrict__ fext, T* __restrict__ vdot,
                                                      T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf) {
  constexpr int SPW = 64 / G;  // states per wavefront
  extern __shared__ double lds_raw[];
  T* const lds = reinterpret_cast<T*>(lds_raw);
  const int lane = threadIdx.x;
  const int g = lane % G, sl = lane / G;
  const long state = (long)blockIdx.x * SPW + sl;
  const bool vstate = state < B;
  T* const stash = lds + sl;                                        // field f of body row e: stash[(e * CS_FIELDS + f) * SPW]
  T* const mb = lds + (long)C.nb * CS_FIELDS * SPW + sl;            // mailbox field f: mb[f * SPW]
  const T a0[6] = {T(0), T(0), T(0), T(-C.gravity[0]), T(-C.gravity[1]), T(-C.gravity[2])};  // a_world = -gravity (mechanism_algorithms.jl:405)

  T R[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}, p[3] = {T(0), T(0), T(0)};
  T Tw[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};

  // ---------------- pass A: transforms to root and twists, root -> leaves ----------------
#pragma unroll 1
  for (int s = 0; s < C.ns; ++s) {
    const int e = C.tab[s * G + g];
    if (e >= 0 && vstate) {
      const int32_t* cb = C.cb + e * CB_STRIDE;
      const T* rb = reinterpret_cast<const T*>(C.rb) + e * RB_STRIDE;
      const int flags = cb[CB_FLAGS];
      Body<T> b;
      b.jtype = cb[CB_JTYPE]; b.qoff = cb[CB_QOFF]; b.voff = cb[CB_VOFF]; b.orig = cb[CB_ORIG]; b.state = state; b.valid = true;
      T qj[7], vj[6], sc[2], XR[9], Xp[3];
      load_joint_q(b, q, Lq, qj);
      load_joint_v(b, v, Lv, vj);
      store_qdot(b, qdot, Lq, qj, vj);
      joint_sc(b.jtype, qj, sc);
      T* row = stash + (long)e * CS_FIELDS * SPW;
      row[CS_SC * SPW] = sc[0];
      row[(CS_SC + 1) * SPW] = sc[1];
      local_transform_sc(b.jtype, rb, sc, qj, XR, Xp);
      T pR[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}, pp[3] = {T(0), T(0), T(0)};
      T pT[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
      if (flags & CF_CHAINED) {
#pragma unroll
        for (int k = 0; k < 9; ++k) pR[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pp[k] = p[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) pT[k] = Tw[k];
      } else if (!(flags & CF_LEVEL0)) {
        const T* m = mb + (long)cb[CB_MBA_R] * SPW;
#pragma unroll
        for (int k = 0; k < 9; ++k) pR[k] = m[k * SPW];
#pragma unroll
        for (int k = 0; k < 3; ++k) pp[k] = m[(9 + k) * SPW];
#pragma unroll
        for (int k = 0; k < 6; ++k) pT[k] = m[(12 + k) * SPW];
      }
      matmul3(pR, XR, R);
      matvec3(pR, Xp, p);
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] += pp[k];
      T tl[6], vJ[6];
      local_joint_motion(b, rb, vj, tl);
      xmotion(R, p, tl, vJ);
#pragma unroll
      for (int k = 0; k < 6; ++k) Tw[k] = pT[k] + vJ[k];
      const int w = cb[CB_MBA_W];
      if (w >= 0) {
        T* m = mb + (long)w * SPW;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k * SPW] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) m[(9 + k) * SPW] = p[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) m[(12 + k) * SPW] = Tw[k];
      }
    }
    __syncthreads();  // one wavefront per workgroup: orders the LDS mailbox traffic between lanes
  }

  // ---------------- pass B: articulated-body inertias and bias forces, leaves -> root ----------------
  for (int i = lane; i < C.nfs * SPW; i += 64) lds[(long)(C.nb * CS_FIELDS + C.nfl) * SPW + i] = T(0);  // hand-off accumulators
  __syncthreads();
  T cI[21], cP[6];  // hand-off (Ia, pa) of the body just finished on this track
#pragma unroll
  for (int k = 0; k < 21; ++k) cI[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) cP[k] = T(0);
#pragma unroll 1
  for (int s = C.ns - 1; s >= 0; --s) {
    const int e = C.tab[s * G + g];
    const bool act = e >= 0 && vstate;
    int accw = -1, round = 0;
    if (act) {
      const int32_t* cb = C.cb + e * CB_STRIDE;
      const T* rb = reinterpret_cast<const T*>(C.rb) + e * RB_STRIDE;
      const int flags = cb[CB_FLAGS];
      Body<T> b;
      b.jtype = cb[CB_JTYPE]; b.qoff = cb[CB_QOFF]; b.voff = cb[CB_VOFF]; b.orig = cb[CB_ORIG]; b.state = state; b.valid = true;
      T* row = stash + (long)e * CS_FIELDS * SPW;
      if (flags & CF_RESTART) {  // the registers hold another chain: take this body's kinematics from its pass-A mailbox
        const T* m = mb + (long)cb[CB_MBA_W] * SPW;
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = m[k * SPW];
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = m[(9 + k) * SPW];
#pragma unroll
        for (int k = 0; k < 6; ++k) Tw[k] = m[(12 + k) * SPW];
      }
      T vj[6], tj[6], fe[6], tl[6], vJ[6], cbias[6];
      load_joint_v(b, v, Lv, vj);
      load_joint_v(b, tau, Lv, tj);
      load_body_wrench(b, fext, Lf, fe);
      local_joint_motion(b, rb, vj, tl);
      xmotion(R, p, tl, vJ);
      se3_comm(Tw, vJ, cbias);  // [T_b, vJ] (mechanism_state.jl:814-830)
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
#pragma unroll
        for (int k = 0; k < 6; ++k) pA[k] -= fe[k];
      }
      if (flags & CF_CARRY) {
#pragma unroll
        for (int k = 0; k < 21; ++k) IA[k] += cI[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) pA[k] += cP[k];
      }
      const int ar = cb[CB_ACC_R];
      if (ar >= 0) {
        const T* m = mb + (long)ar * SPW;
#pragma unroll
        for (int k = 0; k < 21; ++k) IA[k] += m[k * SPW];
#pragma unroll
        for (int k = 0; k < 6; ++k) pA[k] += m[(21 + k) * SPW];
      }
      if (b.jtype == RBD_JOINT_QUAT_FLOATING) {
        // 6-dof joint on the world (the plan admits no other): IA a = S^-T tau - pA, v̇ = S^-1 (a - a_world); [T, vJ] = 0 here
        T f6[6], rhs[6], a[6], d[6], vd[6];
        xforce(R, p, tj, f6);
#pragma unroll
        for (int k = 0; k < 6; ++k) rhs[k] = f6[k] - pA[k];
        sym6_solve(IA, rhs, a);
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = a[k] - a0[k];
        xmotion_inv(R, p, d, vd);
        if (vdot) store_joint_v(b, vdot, Lv, vd);
#pragma unroll
        for (int k = 0; k < 6; ++k) row[(CS_U + k) * SPW] = a[k];  // pass C hands it to the children
      } else {
        T e1[6] = {T(1), T(0), T(0), T(0), T(0), T(0)}, sl6[6], S[6], U[6], W[6];
        local_joint_motion(b, rb, e1, sl6);
        xmotion(R, p, sl6, S);
        sym6_mul(IA, S, U);
        T Dinv = T(0), u = T(0);
        if (joint_nv(b.jtype) > 0) {
          Dinv = rcp_nr(dot6(S, U));
          u = tj[0] - dot6(S, pA);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) row[(CS_U + k) * SPW] = U[k];
        row[CS_DINV * SPW] = Dinv;
        row[CS_u * SPW] = u;
        // hand-off: Ia = IA - U D^-1 U', pa = pA + Ia [T, vJ] + U D^-1 u
#pragma unroll
        for (int k = 0; k < 6; ++k) W[k] = U[k] * Dinv;
        T Iac[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) {
            const T x = IA[SI(i, j)] - W[i] * U[j];
            cI[SI(i, j)] = x;
            Iac[i] += x * cbias[j];
            if (j > i) Iac[j] += x * cbias[i];
          }
#pragma unroll
        for (int k = 0; k < 6; ++k) cP[k] = pA[k] + Iac[k] + W[k] * u;
        accw = cb[CB_ACC_W];
        round = cb[CB_ROUND];
      }
      if (flags & CF_CHAINED) {  // step the kinematics up to the parent (next on this track)
        T sc[2] = {row[CS_SC * SPW], row[(CS_SC + 1) * SPW]}, XR[9], Xp[3], t[3];
        local_transform_sc(b.jtype, rb, sc, a0 /* unused: 6-dof joints are never chained */, XR, Xp);
        matmul3_nt(R, XR, R);
        matvec3(R, Xp, t);
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] -= t[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) Tw[k] -= vJ[k];
      }
    }
    const int nr = C.nrounds[s];
#pragma unroll 1
    for (int r = 0; r < nr; ++r) {  // siblings finishing at the same step add to their parent's accumulator one after the other
      if (act && accw >= 0 && round == r) {
        T* m = mb + (long)accw * SPW;
#pragma unroll
        for (int k = 0; k < 21; ++k) m[k * SPW] += cI[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) m[(21 + k) * SPW] += cP[k];
      }
      __syncthreads();
    }
    __syncthreads();
  }

  // ---------------- pass C: accelerations and v̇, root -> leaves ----------------
  T acc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
#pragma unroll 1
  for (int s = 0; s < C.ns; ++s) {
    const int e = C.tab[s * G + g];
    if (e >= 0 && vstate) {
      const int32_t* cb = C.cb + e * CB_STRIDE;
      const T* rb = reinterpret_cast<const T*>(C.rb) + e * RB_STRIDE;
      const int flags = cb[CB_FLAGS];
      Body<T> b;
      b.jtype = cb[CB_JTYPE]; b.qoff = cb[CB_QOFF]; b.voff = cb[CB_VOFF]; b.orig = cb[CB_ORIG]; b.state = state; b.valid = true;
      const T* row = stash + (long)e * CS_FIELDS * SPW;
      T vj[6], qj[7] = {T(1), T(0), T(0), T(0), T(0), T(0), T(0)};
      load_joint_v(b, v, Lv, vj);
      if (b.jtype == RBD_JOINT_QUAT_FLOATING) load_joint_q(b, q, Lq, qj);
      T pR[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}, pp[3] = {T(0), T(0), T(0)};
      T pT[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, pa[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) pa[k] = a0[k];
      if (flags & CF_CHAINED) {
#pragma unroll
        for (int k = 0; k < 9; ++k) pR[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) pp[k] = p[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { pT[k] = Tw[k]; pa[k] = acc[k]; }
      } else if (!(flags & CF_LEVEL0)) {
        const T* m = mb + (long)cb[CB_MBC_R] * SPW;
#pragma unroll
        for (int k = 0; k < 9; ++k) pR[k] = m[k * SPW];
#pragma unroll
        for (int k = 0; k < 3; ++k) pp[k] = m[(9 + k) * SPW];
#pragma unroll
        for (int k = 0; k < 6; ++k) { pT[k] = m[(12 + k) * SPW]; pa[k] = m[(18 + k) * SPW]; }
      }
      T sc[2] = {row[CS_SC * SPW], row[(CS_SC + 1) * SPW]}, XR[9], Xp[3], tl[6], vJ[6];
      local_transform_sc(b.jtype, rb, sc, qj, XR, Xp);
      matmul3(pR, XR, R);
      matvec3(pR, Xp, p);
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] += pp[k];
      local_joint_motion(b, rb, vj, tl);
      xmotion(R, p, tl, vJ);
#pragma unroll
      for (int k = 0; k < 6; ++k) Tw[k] = pT[k] + vJ[k];
      if (b.jtype == RBD_JOINT_QUAT_FLOATING) {
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = row[(CS_U + k) * SPW];  // solved at the turn of pass B
      } else {
        T cbias[6], e1[6] = {T(1), T(0), T(0), T(0), T(0), T(0)}, sl6[6], S[6], U[6], ap[6];
        se3_comm(Tw, vJ, cbias);
        local_joint_motion(b, rb, e1, sl6);
        xmotion(R, p, sl6, S);
#pragma unroll
        for (int k = 0; k < 6; ++k) { U[k] = row[(CS_U + k) * SPW]; ap[k] = pa[k] + cbias[k]; }
        const T vd = row[CS_DINV * SPW] * (row[CS_u * SPW] - dot6(U, ap));  // v̇ = D^-1 (u - U'a'), a = a' + S v̇
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = ap[k] + S[k] * vd;
        if (joint_nv(b.jtype) > 0 && vdot) vdot[(long)b.voff * Lv.sk + state * Lv.sb] = vd;
      }
      const int w = cb[CB_MBC_W];
      if (w >= 0) {
        T* m = mb + (long)w * SPW;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k * SPW] = R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) m[(9 + k) * SPW] = p[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { m[(12 + k) * SPW] = Tw[k]; m[(18 + k) * SPW] = acc[k]; }
      }
    }
    __syncthreads();
  }
}

}  // namespace rbd
