// rbd_big_kernels.hip — the fallback for trees the wavefront-shaped kernels do not take: more than 64 moving bodies (a state of the lane-per-body
// kernels lives in one wavefront; the walk / state kernels are bounded by their register stashes).  The reference has no such limit — its own
// `rand_chain_mechanism` (src/mechanism_modification.jl:402) is used with 100 joints — so these kernels restate the reference's route
//   dynamics_bias! / inverse_dynamics! (src/mechanism_algorithms.jl:484-498, :542-553), mass_matrix! (:248-272), dynamics_solve! (:764, :819)
// with ONE THREAD PER STATE and the per-body quantities in an HBM scratch laid out [field][body][state] (coalesced across the wavefront).
// Correct at any size, no speed claim: every per-body value makes a round trip through memory.  Bodies are visited in the reference's order
// (parents before children).  Tree mechanisms, every tree joint type.
#include "rbd_lane.hpp"
#include "rbd_internal.hpp"

namespace rbd {

// scratch fields per body: K = R 9, p 3, T 6, a 6 (24), w 6, composite inertia J 6, c 3, m 1 (10)
enum { BIG_K = 0, BIG_W = 24, BIG_IC = 30, BIG_FIELDS = 40 };

template <typename T> struct BigCtx {
  const BigModel& M;
  long B, st;
  T* sc;  // scratch base
  RBD_DEV T& at(int field, int body) const { return sc[((long)field * M.nb + body) * B + st]; }
};

template <typename T> RBD_DEV Body<T> big_body(const BigModel& M, int i, long st) {
  Body<T> b{};
  b.parent = M.tbl[4 * i]; b.jtype = M.tbl[4 * i + 1]; b.qoff = M.tbl[4 * i + 2]; b.voff = M.tbl[4 * i + 3];
  b.state = st; b.valid = true; b.orig = i;
  return b;
}

// forward kinematics of body i from its parent's scratch entry (the world: identity, at rest, a = -g): K = (R, p, T, a); WITH_ACC adds the
// joint acceleration X S_local v̇ (spatial_accelerations! :387-417)
template <typename T> RBD_DEV void big_fk(const BigCtx<T>& C, const Body<T>& b, const T* rb, const T* qj, const T* vj, const T* aj, T* K) {
  T XR[9], Xp[3], tl[6], al[6], pk[24];
  local_transform(b, rb, qj, XR, Xp);
  local_joint_motion(b, rb, vj, tl);
  local_joint_motion(b, rb, aj, al);
  if (b.parent >= 0) {
#pragma unroll
    for (int k = 0; k < 24; ++k) pk[k] = C.at(BIG_K + k, b.parent);
  } else {
#pragma unroll
    for (int k = 0; k < 24; ++k) pk[k] = (k < 9 && k % 4 == 0) ? T(1) : T(0);
    pk[21] = T(-C.M.gravity[0]); pk[22] = T(-C.M.gravity[1]); pk[23] = T(-C.M.gravity[2]);
  }
  matmul3(pk, XR, K);
  matvec3(pk, Xp, K + 9);
#pragma unroll
  for (int k = 0; k < 3; ++k) K[9 + k] += pk[9 + k];
  T vJ[6], nT[6], cr[6], ajw[6];
  xmotion(K, K + 9, tl, vJ);
#pragma unroll
  for (int k = 0; k < 6; ++k) { K[12 + k] = pk[12 + k] + vJ[k]; nT[k] = -K[12 + k]; }
  se3_comm(nT, pk + 12, cr);
  xmotion(K, K + 9, al, ajw);
#pragma unroll
  for (int k = 0; k < 6; ++k) K[18 + k] = pk[18 + k] + cr[k] + ajw[k];
}

// inverse_dynamics! (vdot != nullptr) / dynamics_bias! (vdot == nullptr); acc_out / jw_out as in rnea_kernel
template <typename T>
__global__ __launch_bounds__(64) void big_rnea_kernel(BigModel M, long B, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ vdot,
                                                      const T* __restrict__ fext, T* __restrict__ tau, T* __restrict__ qdot, T* __restrict__ scratch,
                                                      T* __restrict__ acc_out, T* __restrict__ jw_out, Layout Lq, Layout Lv, Layout Lf) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  const BigCtx<T> C{M, B, st, scratch};
  const T* rbase = reinterpret_cast<const T*>(M.rb);
  for (int i = 0; i < M.nb; ++i) {
    const Body<T> b = big_body<T>(M, i, st);
    const T* rb = rbase + (long)i * RB_STRIDE;
    T qj[7], vj[6], aj[6], K[24];
    load_joint_q(b, q, Lq, qj);
    load_joint_v(b, v, Lv, vj);
    load_joint_v(b, vdot, Lv, aj);
    store_qdot(b, qdot, Lq, qj, vj);
    big_fk(C, b, rb, qj, vj, aj, K);
#pragma unroll
    for (int k = 0; k < 24; ++k) C.at(BIG_K + k, i) = K[k];
    RInertia<T> I;
    T Jb[6], mc[3], Ia[6], x[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
    inertia_to_root(Jb, mc, rb[RB_M], K, K + 9, I);
    mul_inertia(I, K + 18, Ia);
    momentum_cross(I, K + 12, x);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const T fe = fext ? fext[(long)(6 * i + k) * Lf.sk + st * Lf.sb] : T(0);
      C.at(BIG_W + k, i) = Ia[k] + x[k] - fe;
      if (acc_out) acc_out[(long)(6 * i + k) * Lf.sk + st * Lf.sb] = K[18 + k];
    }
  }
  for (int i = M.nb - 1; i >= 0; --i) {  // joint_wrenches_and_torques! (:442-459)
    const Body<T> b = big_body<T>(M, i, st);
    const T* rb = rbase + (long)i * RB_STRIDE;
    T w[6], K[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = C.at(BIG_W + k, i);
#pragma unroll
    for (int k = 0; k < 12; ++k) K[k] = C.at(BIG_K + k, i);
    if (jw_out) {
#pragma unroll
      for (int k = 0; k < 6; ++k) jw_out[(long)(6 * i + k) * Lf.sk + st * Lf.sb] = w[k];
    }
    T out[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    if (b.jtype == RBD_JOINT_QUAT_FLOATING) {
      xforce_inv(K, K + 9, w, out);
    } else {
      const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]}, ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
      for (int k = 0; k < joint_nv(b.jtype); ++k) {
        T sl[6], S[6];
        subspace_col(b.jtype, ax, ay, k, sl);
        xmotion(K, K + 9, sl, S);
        const T d = dot6(S, w);
        if (k == 0) out[0] = d; else if (k == 1) out[1] = d; else out[2] = d;
      }
    }
    store_joint_v(b, tau, Lv, out);
    if (b.parent >= 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) C.at(BIG_W + k, b.parent) += w[k];
    }
  }
}

// mass_matrix! (:248-272): lower triangle, structural zeros written
template <typename T>
__global__ __launch_bounds__(64) void big_crba_kernel(BigModel M, long B, const T* __restrict__ q, T* __restrict__ Mout, T* __restrict__ scratch, Layout Lq,
                                                      Layout Lm) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  const BigCtx<T> C{M, B, st, scratch};
  const T* rbase = reinterpret_cast<const T*>(M.rb);
  const int nv = M.nv;
  for (int c2 = 0; c2 < nv; ++c2)
    for (int r = c2; r < nv; ++r) Mout[((long)c2 * nv + r) * Lm.sk + st * Lm.sb] = T(0);
  const T z6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  for (int i = 0; i < M.nb; ++i) {
    const Body<T> b = big_body<T>(M, i, st);
    const T* rb = rbase + (long)i * RB_STRIDE;
    T qj[7], K[24];
    load_joint_q(b, q, Lq, qj);
    big_fk(C, b, rb, qj, z6, z6, K);
#pragma unroll
    for (int k = 0; k < 12; ++k) C.at(BIG_K + k, i) = K[k];
#pragma unroll
    for (int k = 12; k < 24; ++k) C.at(BIG_K + k, i) = T(0);  // (twists are not needed; keep the parent reads of big_fk defined)
    RInertia<T> I;
    T Jb[6], mc[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
    inertia_to_root(Jb, mc, rb[RB_M], K, K + 9, I);
#pragma unroll
    for (int k = 0; k < 6; ++k) C.at(BIG_IC + k, i) = I.J[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) C.at(BIG_IC + 6 + k, i) = I.c[k];
    C.at(BIG_IC + 9, i) = I.m;
  }
  auto subspace = [&](int a, int col, T* S) {  // column `col` of the motion subspace of body a's joint, root frame
    const T* rb = rbase + (long)a * RB_STRIDE;
    const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]}, ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
    T sl[6], K[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) K[k] = C.at(BIG_K + k, a);
    subspace_col(M.tbl[4 * a + 1], ax, ay, col, sl);
    xmotion(K, K + 9, sl, S);
  };
  for (int i = M.nb - 1; i >= 0; --i) {  // composite inertias bottom-up (update_crb_inertias!), then row block of body i
    RInertia<T> Ic;
#pragma unroll
    for (int k = 0; k < 6; ++k) Ic.J[k] = C.at(BIG_IC + k, i);
#pragma unroll
    for (int k = 0; k < 3; ++k) Ic.c[k] = C.at(BIG_IC + 6 + k, i);
    Ic.m = C.at(BIG_IC + 9, i);
    const int p = M.tbl[4 * i];
    if (p >= 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) C.at(BIG_IC + k, p) += Ic.J[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) C.at(BIG_IC + 6 + k, p) += Ic.c[k];
      C.at(BIG_IC + 9, p) += Ic.m;
    }
    const int nvi = joint_nv(M.tbl[4 * i + 1]), vi = M.tbl[4 * i + 3];
    for (int ci = 0; ci < nvi; ++ci) {
      T Si[6], F[6];
      subspace(i, ci, Si);
      mul_inertia(Ic, Si, F);
      for (int a = i; a >= 0; a = M.tbl[4 * a]) {  // the joints that support body i (support_set_masks, mechanism_state.jl:95-98)
        const int nva = joint_nv(M.tbl[4 * a + 1]), va = M.tbl[4 * a + 3];
        for (int ca = 0; ca < nva; ++ca) {
          const int row = vi + ci, col = va + ca;
          if (col <= row) {
            T Sa[6];
            subspace(a, ca, Sa);
            Mout[((long)col * nv + row) * Lm.sk + st * Lm.sb] = dot6(F, Sa);
          }
        }
      }
    }
  }
}

// dynamics_solve! without loop joints (:764, :819): L = chol(M), x = M^-1 (rhs - c).  Like the reference, which copies M into result.L before potrf!
// (:763-764) and leaves result.massmatrix intact, the factor goes to a scratch of its own (Lg, batch-innermost: entry (row, col) of state st at
// (col nv + row) B + st) — Mg is the caller's M_out / the bound result.massmatrix / the workspace's M and is only read.
template <typename T>
__global__ __launch_bounds__(64) void big_chol_solve_kernel(int nv, long B, const T* __restrict__ Mg, T* __restrict__ Lg, const T* __restrict__ rhs,
                                                            const T* __restrict__ c, T* __restrict__ x, Layout Lm, Layout Lv, int* __restrict__ notpd) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  auto M = [&](int row, int col) -> T { return Mg[((long)col * nv + row) * Lm.sk + st * Lm.sb]; };
  auto L = [&](int row, int col) -> T& { return Lg[((long)col * nv + row) * B + st]; };
  bool bad = false;
  for (int j = 0; j < nv; ++j) {
    T d = M(j, j);
    for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > T(0))) bad = true;
    d = SqrtT<T>::f(d);
    L(j, j) = d;
    const T id = T(1) / d;
    for (int i = j + 1; i < nv; ++i) {
      T s = M(i, j);
      for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
      L(i, j) = s * id;
    }
  }
  if (bad) atomicOr(notpd, 1);
  auto X = [&](int i) -> T& { return x[(long)i * Lv.sk + st * Lv.sb]; };
  for (int i = 0; i < nv; ++i) {
    const long a = (long)i * Lv.sk + st * Lv.sb;
    T s = (rhs ? rhs[a] : T(0)) - (c ? c[a] : T(0));
    for (int k = 0; k < i; ++k) s -= L(i, k) * X(k);
    X(i) = s / L(i, i);
  }
  for (int i = nv - 1; i >= 0; --i) {
    T s = X(i);
    for (int k = i + 1; k < nv; ++k) s -= L(k, i) * X(k);
    X(i) = s / L(i, i);
  }
}

// The kinematics by-products on trees of any size (round 6; rbd_kinematics / rbd_geometric_jacobian / rbd_momentum of trees the lane-per-body kin_kernel does
// not take): momentum_matrix! (src/mechanism_algorithms.jl:313-327), center_of_mass (:28-50), kinetic_energy / gravitational_potential_energy
// (src/mechanism_state.jl:886-903), geometric_jacobian! of path(base -> target) (:80-99), momentum / momentum_rate_bias (src/mechanism_state.jl:975-987).
// One thread per state like everything in this file; every output nullable; base / target: reference body indices, -1 = the root body.
template <typename T>
__global__ __launch_bounds__(64) void big_kin_kernel(BigModel M, long B, const T* __restrict__ q, const T* __restrict__ v, T* __restrict__ A_out, T* __restrict__ com_out,
                                                     T* __restrict__ energy_out, T* __restrict__ J_out, int base, int target, T* __restrict__ mom_out,
                                                     T* __restrict__ scratch, Layout Lq, Layout Lv, Layout La, Layout L3, Layout L2, Layout L12) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  const BigCtx<T> C{M, B, st, scratch};
  const T* rbase = reinterpret_cast<const T*>(M.rb);
  const T z6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  T ke = T(0), pe = T(0), ms = T(0), cs[3] = {T(0), T(0), T(0)}, hs[6], ws[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) hs[k] = ws[k] = T(0);
  for (int i = 0; i < M.nb; ++i) {  // parents first: transforms, twists, bias accelerations (the world's -g inside), inertias in the root frame
    const Body<T> b = big_body<T>(M, i, st);
    const T* rb = rbase + (long)i * RB_STRIDE;
    T qj[7], vj[6], K[24];
    load_joint_q(b, q, Lq, qj);
    load_joint_v(b, v, Lv, vj);
    big_fk(C, b, rb, qj, vj, z6, K);
#pragma unroll
    for (int k = 0; k < 24; ++k) C.at(BIG_K + k, i) = K[k];
    RInertia<T> I;
    T Jb[6], mc[3], h[6], acc[6], Ia[6], x[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
    inertia_to_root(Jb, mc, rb[RB_M], K, K + 9, I);
#pragma unroll
    for (int k = 0; k < 6; ++k) C.at(BIG_IC + k, i) = I.J[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) C.at(BIG_IC + 6 + k, i) = I.c[k];
    C.at(BIG_IC + 9, i) = I.m;
    mul_inertia(I, K + 12, h);
    ke += dot6(h, K + 12) / 2;
    if (I.m > T(0)) {  // (bodies without mass have no centre of mass: center_of_mass / gravitational_potential_energy skip them, mechanism_algorithms.jl:36)
      pe -= T(M.gravity[0]) * I.c[0] + T(M.gravity[1]) * I.c[1] + T(M.gravity[2]) * I.c[2];
      ms += I.m;
#pragma unroll
      for (int k = 0; k < 3; ++k) cs[k] += I.c[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = K[18 + k];
    acc[3] += T(M.gravity[0]); acc[4] += T(M.gravity[1]); acc[5] += T(M.gravity[2]);  // bias accelerations proper: without the root's -g
    mul_inertia(I, acc, Ia);
    momentum_cross(I, K + 12, x);
#pragma unroll
    for (int k = 0; k < 6; ++k) { hs[k] += h[k]; ws[k] += Ia[k] + x[k]; }
  }
  if (energy_out) { energy_out[0 * L2.sk + st * L2.sb] = ke; energy_out[1 * L2.sk + st * L2.sb] = pe; }
  if (com_out) {
#pragma unroll
    for (int k = 0; k < 3; ++k) com_out[(long)k * L3.sk + st * L3.sb] = cs[k] / ms;
  }
  if (mom_out) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { mom_out[(long)k * L12.sk + st * L12.sb] = hs[k]; mom_out[(long)(6 + k) * L12.sk + st * L12.sb] = ws[k]; }
  }
  auto subspace = [&](int a, int col, T* S) {  // column `col` of the motion subspace of body a's joint, root frame
    const T* rb = rbase + (long)a * RB_STRIDE;
    const T ax[3] = {rb[RB_AXIS], rb[RB_AXIS + 1], rb[RB_AXIS + 2]}, ay[3] = {rb[RB_AXIS2], rb[RB_AXIS2 + 1], rb[RB_AXIS2 + 2]};
    T sl[6], K[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) K[k] = C.at(BIG_K + k, a);
    subspace_col(M.tbl[4 * a + 1], ax, ay, col, sl);
    xmotion(K, K + 9, sl, S);
  };
  if (J_out) {
    // fill!(jac, 0), then the joints of path(base, target) (src/graphs/tree_path.jl:41-63: both ends walked up to their lowest common ancestor; a body's
    // parent has the smaller index): -S for those walked upwards from the base, +S downwards to the target
    for (long e = 0; e < 6L * M.nv; ++e) J_out[e * La.sk + st * La.sb] = T(0);
    int a = base, bb = target;
    while (a != bb) {
      const int i = a > bb ? a : bb;
      const T sg = a > bb ? T(-1) : T(1);
      const int nvi = joint_nv(M.tbl[4 * i + 1]), vi = M.tbl[4 * i + 3];
      for (int ci = 0; ci < nvi; ++ci) {
        T Si[6];
        subspace(i, ci, Si);
#pragma unroll
        for (int k = 0; k < 6; ++k) J_out[((long)(vi + ci) * 6 + k) * La.sk + st * La.sb] = sg * Si[k];
      }
      if (a > bb) a = M.tbl[4 * a]; else bb = M.tbl[4 * bb];
    }
  }
  if (A_out) {
    for (int i = M.nb - 1; i >= 0; --i) {  // composite inertias bottom-up (update_crb_inertias!), column i = crb_inertia(body i) S_i
      RInertia<T> Ic;
#pragma unroll
      for (int k = 0; k < 6; ++k) Ic.J[k] = C.at(BIG_IC + k, i);
#pragma unroll
      for (int k = 0; k < 3; ++k) Ic.c[k] = C.at(BIG_IC + 6 + k, i);
      Ic.m = C.at(BIG_IC + 9, i);
      const int p = M.tbl[4 * i];
      if (p >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) C.at(BIG_IC + k, p) += Ic.J[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) C.at(BIG_IC + 6 + k, p) += Ic.c[k];
        C.at(BIG_IC + 9, p) += Ic.m;
      }
      const int nvi = joint_nv(M.tbl[4 * i + 1]), vi = M.tbl[4 * i + 3];
      for (int ci = 0; ci < nvi; ++ci) {
        T Si[6], F[6];
        subspace(i, ci, Si);
        mul_inertia(Ic, Si, F);
#pragma unroll
        for (int k = 0; k < 6; ++k) A_out[((long)(vi + ci) * 6 + k) * La.sk + st * La.sb] = F[k];
      }
    }
  }
}
template <typename T>
hipError_t launch_big_kin(const BigModel& M, long B, const void* q, const void* v, void* A, void* com, void* energy, void* J, int base, int target, void* mom,
                          void* scratch, Layout Lq, Layout Lv, Layout La, Layout L3, Layout L2, Layout L12, hipStream_t s) {
  hipLaunchKernelGGL(big_kin_kernel<T>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, M, B, (const T*)q, (const T*)v, (T*)A, (T*)com, (T*)energy, (T*)J, base, target,
                     (T*)mom, (T*)scratch, Lq, Lv, La, L3, L2, L12);
  return hipGetLastError();
}
template hipError_t launch_big_kin<double>(const BigModel&, long, const void*, const void*, void*, void*, void*, void*, int, int, void*, void*, Layout, Layout, Layout, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_big_kin<float>(const BigModel&, long, const void*, const void*, void*, void*, void*, void*, int, int, void*, void*, Layout, Layout, Layout, Layout, Layout, Layout, hipStream_t);

// The PD law of rbd_simulate_controlled (RBD_CONTROL_PD) over the any-size tables: one thread per (state, body); tau_out = tau_ff − kp (q − q_des) − kd v on
// Revolute / Prismatic joints, tau_ff on every other joint's coordinates (pd_control_kernel of rbd_kernels.hip reads the lane-per-body tables)
template <typename T>
__global__ __launch_bounds__(256) void big_pd_control_kernel(BigModel M, long B, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ tau_ff,
                                                             const T* __restrict__ qdes, const T* __restrict__ kp, const T* __restrict__ kd, T* __restrict__ tau_out,
                                                             Layout Lq, Layout Lv) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)M.nb * B) return;
  // consecutive threads walk the fastest-varying index of the layout
  const long st = Lv.sk == 1 ? e / M.nb : e % B;
  const int i = (int)(Lv.sk == 1 ? e % M.nb : e / B);
  const int jt = M.tbl[4 * i + 1], qo = M.tbl[4 * i + 2], vo = M.tbl[4 * i + 3], nvi = joint_nv(jt);
  for (int k = 0; k < nvi; ++k) {
    const long a = (long)(vo + k) * Lv.sk + st * Lv.sb;
    T t = tau_ff ? tau_ff[a] : T(0);
    if (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_PRISMATIC) {
      const long qa = (long)qo * Lq.sk + st * Lq.sb;
      t -= kp[vo] * (q[qa] - (qdes ? qdes[qa] : T(0))) + kd[vo] * v[a];
    }
    tau_out[a] = t;
  }
}
template <typename T>
hipError_t launch_big_pd_control(const BigModel& M, long B, const void* q, const void* v, const void* tau_ff, const void* qdes, const void* kp, const void* kd,
                                 void* tau_out, Layout Lq, Layout Lv, hipStream_t s) {
  const long n = (long)M.nb * B;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(big_pd_control_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, M, B, (const T*)q, (const T*)v, (const T*)tau_ff, (const T*)qdes,
                     (const T*)kp, (const T*)kd, (T*)tau_out, Lq, Lv);
  return hipGetLastError();
}
template hipError_t launch_big_pd_control<double>(const BigModel&, long, const void*, const void*, const void*, const void*, const void*, const void*, void*, Layout, Layout, hipStream_t);
template hipError_t launch_big_pd_control<float>(const BigModel&, long, const void*, const void*, const void*, const void*, const void*, const void*, void*, Layout, Layout, hipStream_t);

// the per-body kinematics the loop branch reads (24 per body and state: R, p, twist, bias acceleration with the world's -g, as rnea_kernel exports them) out of
// the scratch big_rnea_kernel (vdot == nullptr) has just filled: body[st * 24 nb + 24 i + k]
template <typename T>
__global__ __launch_bounds__(256) void big_export_body_kernel(BigModel M, long B, const T* __restrict__ scratch, T* __restrict__ body) {
  const long st = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (st >= B) return;
  for (int i = 0; i < M.nb; ++i)
#pragma unroll 8
    for (int k = 0; k < 24; ++k) body[(st * M.nb + i) * 24 + k] = scratch[((long)(BIG_K + k) * M.nb + i) * B + st];
}
template <typename T>
hipError_t launch_big_export_body(const BigModel& M, long B, const void* scratch, void* body, hipStream_t s) {
  hipLaunchKernelGGL(big_export_body_kernel<T>, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, M, B, (const T*)scratch, (T*)body);
  return hipGetLastError();
}
template hipError_t launch_big_export_body<double>(const BigModel&, long, const void*, void*, hipStream_t);
template hipError_t launch_big_export_body<float>(const BigModel&, long, const void*, void*, hipStream_t);

// dst[b * n + k] = src[k * ld + b]: a batch-innermost scratch (what a one-lane-per-state kernel stores in 256-byte runs) to the caller's state-major
// buffer, 1 << lts states (32; 16 when the tile would pass 64 KB) per block through LDS: 128-byte (fp32) row pieces in, ONE contiguous block out.
// blockIdx.y picks one of two buffer pairs (accelerations, joint wrenches).  Pure traffic: 2 n B elements per pair; sixteen independent loads per
// thread in flight.  Measured (rocprofv3): 32 us for the two 6 x 31 x 65 536 fp32 outputs of Atlas, 195 MB moved, 6 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void rows_to_state_major_kernel(int n, long B, long ld, int lts, const T* __restrict__ src0, T* __restrict__ dst0, const T* __restrict__ src1,
                                                                  T* __restrict__ dst1) {
  extern __shared__ __align__(16) unsigned char tile_raw[];
  T* tile = reinterpret_cast<T*>(tile_raw);
  const T* __restrict__ src = blockIdx.y ? src1 : src0;
  T* __restrict__ dst = blockIdx.y ? dst1 : dst0;
  const int np = n | 1, ts = 1 << lts;  // odd row length: the states of a row piece land in different banks
  const long s0 = (long)blockIdx.x << lts;
  const int ns = (int)((B - s0) < ts ? (B - s0) : ts), total = n << lts;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 16) {
    T r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + 256 * u, k = i >> lts, st = i & (ts - 1);
      r[u] = (i < total && st < ns) ? src[(long)k * ld + s0 + st] : T(0);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + 256 * u, k = i >> lts, st = i & (ts - 1);
      if (i < total) tile[st * np + k] = r[u];
    }
  }
  __syncthreads();
  T* out = dst + s0 * n;
  const int nout = ns * n, dk = 256 % n, ds = 256 / n;
  int st = threadIdx.x / n, k = threadIdx.x % n;
#pragma unroll 8
  for (int i = threadIdx.x; i < nout; i += 256) {
    out[i] = tile[st * np + k];
    k += dk; st += ds;
    if (k >= n) { k -= n; ++st; }
  }
}
template <typename T>
hipError_t launch_rows_to_state_major(int n, long B, long ld, const void* src0, void* dst0, const void* src1, void* dst1, hipStream_t s) {
  const int lts = (size_t)32 * (n | 1) * sizeof(T) <= 65536 ? 5 : 4;
  hipLaunchKernelGGL(rows_to_state_major_kernel<T>, dim3((unsigned)((B + (1 << lts) - 1) >> lts), src1 ? 2 : 1), dim3(256), ((size_t)(n | 1) << lts) * sizeof(T), s, n, B,
                     ld, lts, (const T*)src0, (T*)dst0, (const T*)src1, (T*)dst1);
  return hipGetLastError();
}
template hipError_t launch_rows_to_state_major<double>(int, long, long, const void*, void*, const void*, void*, hipStream_t);
template hipError_t launch_rows_to_state_major<float>(int, long, long, const void*, void*, const void*, void*, hipStream_t);

size_t big_scratch_elems(const BigModel& M, long B) { return (size_t)BIG_FIELDS * (size_t)M.nb * (size_t)B; }

template <typename T>
hipError_t launch_big_rnea(const BigModel& M, long B, const void* q, const void* v, const void* vdot, const void* fext, void* tau, void* qdot, void* scratch,
                           void* acc_out, void* jw_out, Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  hipLaunchKernelGGL(big_rnea_kernel<T>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, M, B, (const T*)q, (const T*)v, (const T*)vdot, (const T*)fext, (T*)tau,
                     (T*)qdot, (T*)scratch, (T*)acc_out, (T*)jw_out, Lq, Lv, Lf);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_big_crba(const BigModel& M, long B, const void* q, void* Mout, void* scratch, Layout Lq, Layout Lm, hipStream_t s) {
  hipLaunchKernelGGL(big_crba_kernel<T>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, M, B, (const T*)q, (T*)Mout, (T*)scratch, Lq, Lm);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_big_chol_solve(int nv, long B, const void* Mg, void* Lg, const void* rhs, const void* c, void* x, Layout Lm, Layout Lv, int* notpd, hipStream_t s) {
  hipLaunchKernelGGL(big_chol_solve_kernel<T>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, nv, B, (const T*)Mg, (T*)Lg, (const T*)rhs, (const T*)c, (T*)x, Lm, Lv, notpd);
  return hipGetLastError();
}
#define RBD_BIG_INST(T)                                                                                                                                     \
  template hipError_t launch_big_rnea<T>(const BigModel&, long, const void*, const void*, const void*, const void*, void*, void*, void*, void*, void*, Layout, \
                                         Layout, Layout, hipStream_t);                                                                                       \
  template hipError_t launch_big_crba<T>(const BigModel&, long, const void*, void*, void*, Layout, Layout, hipStream_t);                                    \
  template hipError_t launch_big_chol_solve<T>(int, long, const void*, void*, const void*, const void*, void*, Layout, Layout, int*, hipStream_t);
RBD_BIG_INST(double)
RBD_BIG_INST(float)

}  // namespace rbd

// ---- M as LAPACK's packed lower triangle (rbd_mass_matrix_solve_packed, the routes that form the square): element (i, j), i >= j, of state b from the square at
// (j nv + i) to the triangle at i + j (2 nv - j - 1) / 2.  One thread per packed value, consecutive threads along the caller's contiguous direction.
namespace rbd {
template <typename T>
__global__ __launch_bounds__(256) void pack_lower_kernel(int nv, long B, const T* __restrict__ M, T* __restrict__ P, Layout Lm, Layout Lp) {
  const long np = (long)nv * (nv + 1) / 2;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= np * B) return;
  long b, k;
  if (Lp.sk == 1) { b = e / np; k = e - b * np; } else { k = e / B; b = e - k * B; }
  // column of packed index k: the largest j with j (2 nv - j + 1) / 2 <= k
  int j = (int)(((2.0f * nv + 1.0f) - sqrtf((2.0f * nv + 1.0f) * (2.0f * nv + 1.0f) - 8.0f * (float)k)) * 0.5f);
  j = j < 0 ? 0 : (j > nv - 1 ? nv - 1 : j);
  while (j > 0 && (long)j * (2 * nv - j + 1) / 2 > k) --j;
  while (j + 1 < nv && (long)(j + 1) * (2 * nv - j) / 2 <= k) ++j;
  const int i = j + (int)(k - (long)j * (2 * nv - j + 1) / 2);
  P[k * Lp.sk + layout_base(Lp, b)] = M[((long)j * nv + i) * Lm.sk + layout_base(Lm, b)];
}
template <typename T> hipError_t launch_pack_lower(int nv, long B, const void* M, void* P, Layout Lm, Layout Lp, hipStream_t s) {
  const long total = (long)nv * (nv + 1) / 2 * B;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_lower_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, nv, B, (const T*)M, (T*)P, Lm, Lp);
  return hipGetLastError();
}
template hipError_t launch_pack_lower<float>(int, long, const void*, void*, Layout, Layout, hipStream_t);
template hipError_t launch_pack_lower<double>(int, long, const void*, void*, Layout, Layout, hipStream_t);
}  // namespace rbd

// ---- `simulate` on trees of more than 64 bodies: one stage of the Munthe-Kaas RK4 step (mk_stage_kernel of rbd_kernels.hip: same per-joint arithmetic,
// rbd_integrator.hpp) with one thread per (state, body) over the any-size tables.  Correctness, no speed claim — like every kernel of this file.
namespace rbd {
template <typename T>
__global__ __launch_bounds__(256) void big_mk_stage_kernel(BigModel M, long B, int stage, T dt, T* __restrict__ q, T* __restrict__ v, const T* __restrict__ vdot_prev,
                                                           MkBuffers W, Layout Lq, Layout Lv, int close_prev) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)M.nb * B) return;
  const long state = e / M.nb;
  const int i = (int)(e - state * M.nb);
  Body<T> b{};
  b.jtype = M.tbl[4 * i + 1]; b.qoff = M.tbl[4 * i + 2]; b.voff = M.tbl[4 * i + 3]; b.state = state; b.valid = true;
  if (b.jtype == RBD_JOINT_FIXED) return;
  T qj[7], vj[6];
  load_joint_q(b, q, Lq, qj);
  load_joint_v(b, v, Lv, vj);
  if (close_prev) mk_stage_lane<T, 0>(b, 4, dt, qj, vj, vdot_prev, W, q, v, Lq, Lv);  // the step before closes in this launch
  mk_stage_lane<T, 0>(b, stage, dt, qj, vj, vdot_prev, W, q, v, Lq, Lv);
}
template <typename T>
hipError_t launch_big_mk_stage(const BigModel& M, long B, int stage, double dt, void* q, void* v, const void* vdot_prev, const MkBuffers& W, Layout Lq, Layout Lv,
                               hipStream_t s, int close_prev) {
  const long total = (long)M.nb * B;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(big_mk_stage_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, M, B, stage, (T)dt, (T*)q, (T*)v, (const T*)vdot_prev, W, Lq, Lv, close_prev);
  return hipGetLastError();
}
template hipError_t launch_big_mk_stage<float>(const BigModel&, long, int, double, void*, void*, const void*, const MkBuffers&, Layout, Layout, hipStream_t, int);
template hipError_t launch_big_mk_stage<double>(const BigModel&, long, int, double, void*, void*, const void*, const MkBuffers&, Layout, Layout, hipStream_t, int);
}  // namespace rbd
