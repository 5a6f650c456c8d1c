// rbd_loop_small.hpp — SMALL loop mechanisms (four-bar linkage: BASELINE configs[4]) in one thread per state: kinematics, dynamics bias, mass matrix,
// constraint Jacobian and bias, and the constrained solve of `dynamics_solve!`'s loop branch (src/mechanism_algorithms.jl:574-673, :768-816).
// The code is templated on the VIEW of the mechanism's loop tables: LoopView<T> (rbd_device.hpp: pointers into device memory, the kernels of
// rbd_kernels.hip) or a type whose members are compile-time constant tables — the program rbd_jit.hip generates for ONE mechanism, where every
// table read, joint-type test and loop bound folds away.
#pragma once
#include "rbd_lane.hpp"

namespace rbd {

// C = A ∘ B for transforms (R row-major, p)
template <typename T> RBD_DEV void xf_compose(const T* AR, const T* Ap, const T* BR, const T* Bp, T* CR, T* Cp) {
  matmul3(AR, BR, CR);
  matvec3(AR, Bp, Cp);
#pragma unroll
  for (int k = 0; k < 3; ++k) Cp[k] += Ap[k];
}

// The same solve for SMALL loop mechanisms (nv <= NV, nc <= NC; the four-bar linkage of BASELINE configs[4] has nv 3, nc 5): every matrix
// is a fixed-size local array and every loop has a compile-time bound with an `i < nv` predicate, so the whole chain of dependent
// small-matrix steps runs out of VGPRs instead of LDS/HBM work arrays.  Indices that are only known at run time (constraint row,
// velocity column) are resolved by predicated writes over the compile-time range.
// the per-state work of loop_solve_small_kernel; also the second half of loop_fused_small_kernel, where body / Mg / cg are what the same thread
// has just written (hence no __restrict__ on them here)
template <typename T, typename VIEW, int NV, int NC>
RBD_DEV void loop_solve_small_state(const VIEW& V, long st, int stabilize, const T* bd, const T* Mp, long msk, const T* cp, long csk, const T* __restrict__ tau,
                                    T* __restrict__ vdot, T* __restrict__ lambda, T* __restrict__ Kg, T* __restrict__ kg, Layout Lv, Layout Lc,
                                    Layout Lk, double g0, double g1, double g2, int* __restrict__ notpd) {
  // bd: this state's per-body kinematics (24 per body: R, p, twist, bias acceleration); M(i, j) = Mp[(j nv + i) msk] (lower triangle), c(i) = cp[i csk] —
  // the caller's global buffers, or the registers of the thread that has just computed them
  const int nv = V.nv, nc = V.nc;
  T L[NV][NV], K[NC][NV], Y[NC][NV], kk[NC], bv[NC], z[NV], rhs[NV], lam[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    kk[c] = T(0); bv[c] = T(0); lam[c] = T(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) { K[c][i] = T(0); Y[c][i] = T(0); }
  }
  const T I3[9] = {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)};
  const T Z3[3] = {T(0), T(0), T(0)};
  for (int l = 0; l < V.nloops; ++l) {
    const int32_t* li = V.li + 8 * l;
    const T* lr = V.lr + 64 * l;
    const int pred = li[0], succ = li[1], row0 = li[3], ncl = li[4];
    const T* HpR = pred >= 0 ? bd + pred * 24 : I3; const T* Hpp = pred >= 0 ? bd + pred * 24 + 9 : Z3;
    const T* HsR = succ >= 0 ? bd + succ * 24 : I3; const T* Hsp = succ >= 0 ? bd + succ * 24 + 9 : Z3;
    T FbR[9], Fbp[3], FaR[9], Fap[3];
    xf_compose(HpR, Hpp, lr, lr + 9, FbR, Fbp);
    xf_compose(HsR, Hsp, lr + 12, lr + 21, FaR, Fap);
    T Tw[6][6];
#pragma unroll
    for (int ci = 0; ci < 6; ++ci) {
      const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
      xforce(FaR, Fap, ci < ncl ? lr + 28 + 6 * ci : zero6, Tw[ci]);
    }
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
#pragma unroll
        for (int ci = 0; ci < 6; ++ci) {
          const T d = dot6(Tw[ci], S);
          const T val = sign < 0 ? -d : d;
#pragma unroll
          for (int r = 0; r < NC; ++r)
#pragma unroll
            for (int cc = 0; cc < NV; ++cc)
              if (ci < ncl && r == row0 + ci && cc == vi) K[r][cc] = val;
        }
      }
    }
    T Tp[6], Ts[6], Ap[6], As[6], cr[6], ba[6];
    const T grav[6] = {T(0), T(0), T(0), T(-g0), T(-g1), T(-g2)};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      Tp[j] = pred >= 0 ? bd[pred * 24 + 12 + j] : T(0); Ts[j] = succ >= 0 ? bd[succ * 24 + 12 + j] : T(0);
      Ap[j] = pred >= 0 ? bd[pred * 24 + 18 + j] - grav[j] : T(0); As[j] = succ >= 0 ? bd[succ * 24 + 18 + j] - grav[j] : T(0);
    }
    se3_comm(Ts, Tp, cr);
#pragma unroll
    for (int j = 0; j < 6; ++j) ba[j] = cr[j] + (As[j] - Ap[j]);
    if (stabilize) {
      T TnR[9], d3[3], Tnp[3], jt[6], jl[6], stab[6], sw[6], Rtp[3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) TnR[3 * i + j] = FbR[i] * FaR[j] + FbR[3 + i] * FaR[3 + j] + FbR[6 + i] * FaR[6 + j];
#pragma unroll
      for (int k = 0; k < 3; ++k) d3[k] = Fap[k] - Fbp[k];
      matTvec3(FbR, d3, Tnp);
#pragma unroll
      for (int j = 0; j < 6; ++j) jt[j] = Ts[j] - Tp[j];
      xmotion_inv(FaR, Fap, jt, jl);
      const T psi[3] = {(TnR[7] - TnR[5]) / 2, (TnR[2] - TnR[6]) / 2, (TnR[3] - TnR[1]) / 2};
      matTvec3(TnR, Tnp, Rtp);
      const T* gl = V.gains ? V.gains + 64 * l : lr + 24;  // stabilization_gains of this call, else the model's (constants in the compiled form)
#pragma unroll
      for (int i = 0; i < 3; ++i) { stab[i] = -gl[0] * psi[i] - gl[1] * jl[i]; stab[3 + i] = -gl[2] * Rtp[i] - gl[3] * jl[3 + i]; }
      xmotion(FaR, Fap, stab, sw);
#pragma unroll
      for (int j = 0; j < 6; ++j) ba[j] -= sw[j];
    }
#pragma unroll
    for (int ci = 0; ci < 6; ++ci) {
      const T d = dot6(Tw[ci], ba);
#pragma unroll
      for (int r = 0; r < NC; ++r)
        if (ci < ncl && r == row0 + ci) kk[r] = d;
    }
  }
  // L = chol(M), identity-padded beyond nv so that the padded rows / columns are inert
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int i = 0; i < NV; ++i) L[j][i] = (i >= j && i < nv && j < nv) ? Mp[((long)j * nv + i) * msk] : ((i == j) ? T(1) : T(0));
  bool bad = false;
  T invd[NV];  // 1 / L[j][j]: the substitutions multiply (seven of them per state: a division each time was ~30 instructions in fp64)
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    T d = L[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[k][j] * L[k][j];
    if (!(d > T(0))) bad = true;
    const T id = bad ? T(0) : rsqrt_nr(d);
    L[j][j] = d * id;
    invd[j] = id;
#pragma unroll
    for (int i = j + 1; i < NV; ++i) {
      T s2 = L[j][i];
#pragma unroll
      for (int k = 0; k < j; ++k) s2 -= L[k][i] * L[k][j];
      L[j][i] = s2 * id;
    }
  }
  if (bad) atomicOr(notpd, 1);
  auto fwd = [&](T* x) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { T s2 = x[i];
#pragma unroll
      for (int k = 0; k < i; ++k) s2 -= L[k][i] * x[k];
      x[i] = s2 * invd[i]; }
  };
  auto bwd = [&](T* x) {
#pragma unroll
    for (int i = NV - 1; i >= 0; --i) { T s2 = x[i];
#pragma unroll
      for (int k = i + 1; k < NV; ++k) s2 -= L[i][k] * x[k];
      x[i] = s2 * invd[i]; }
  };
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const T t = (tau && i < nv) ? tau[(long)i * Lv.sk + st * Lv.sb] : T(0);
    z[i] = i < nv ? t - cp[(long)i * csk] : T(0);
    rhs[i] = z[i];
  }
  if (nc > 0) {
    fwd(z);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int k = 0; k < NV; ++k) Y[c][k] = K[c][k];
      fwd(Y[c]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      T s2 = kk[c];
#pragma unroll
      for (int k = 0; k < NV; ++k) s2 += Y[c][k] * z[k];
      bv[c] = s2;
    }
    // minimum-norm lambda = (Y Y')⁺ bv = Y (Y'Y)⁺² Y' bv through a RANK-REVEALING CHOLESKY of the Gram matrix G = Y'Y (NV x NV, positive semi-definite;
    // zero rows / columns beyond nv): G = R'R with R upper triangular and a ZERO ROW for every column whose pivot falls below rcond x the largest
    // diagonal entry — a dependent constraint direction (the four-bar's loop joint has five rows of rank two), the reference's `gelsy!` drops the same
    // directions by its own rcond (src/mechanism_algorithms.jl:810).  With the non-zero rows R_r (full row rank): G⁺ = R_r' (R_r R_r')⁻² R_r, and
    // C = R R' + (ones on the skipped diagonal entries) keeps the sizes fixed.  About 250 instructions where the cyclic Jacobi eigen-solve of rounds
    // 1-2 took ~1 500 on this thread.
    T G[NV][NV], R[NV][NV], C[NV][NV], cinv[NV], t2[NV], w[NV];
    T dmax = T(0);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        T s2 = T(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) s2 += Y[c][i] * Y[c][j];
        G[i][j] = s2;
        if (i == j && s2 > dmax) dmax = s2;
      }
    const T tol = (sizeof(T) == 8 ? T(1e-10) : T(1e-6)) * dmax;
    bool keep[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      T d = G[k][k];
#pragma unroll
      for (int m = 0; m < k; ++m) d -= R[m][k] * R[m][k];
      keep[k] = d > tol;
      const T ri = keep[k] ? rsqrt_nr(d) : T(0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (j < k) R[k][j] = T(0);
        else if (j == k) R[k][j] = keep[k] ? d * ri : T(0);
        else {
          T s2 = G[k][j];
#pragma unroll
          for (int m = 0; m < k; ++m) s2 -= R[m][k] * R[m][j];
          R[k][j] = s2 * ri;
        }
      }
    }
    // C = R R' (lower), Cholesky in place: C[i][j], i >= j, becomes the factor; cinv = 1 / diagonal
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        T s2 = (i == j && !keep[i]) ? T(1) : T(0);
#pragma unroll
        for (int k = i; k < NV; ++k) s2 += R[i][k] * R[j][k];
        C[i][j] = s2;
      }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      T d = C[j][j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= C[j][k] * C[j][k];
      const T id = rsqrt_nr(d);
      C[j][j] = d * id;
      cinv[j] = id;
#pragma unroll
      for (int i = j + 1; i < NV; ++i) {
        T s2 = C[i][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s2 -= C[i][k] * C[j][k];
        C[i][j] = s2 * id;
      }
    }
    auto csolve = [&](T* x) {  // x <- C⁻¹ x
#pragma unroll
      for (int i = 0; i < NV; ++i) { T s2 = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s2 -= C[i][k] * x[k];
        x[i] = s2 * cinv[i]; }
#pragma unroll
      for (int i = NV - 1; i >= 0; --i) { T s2 = x[i];
#pragma unroll
        for (int k = i + 1; k < NV; ++k) s2 -= C[k][i] * x[k];
        x[i] = s2 * cinv[i]; }
    };
    auto gplus = [&](const T* x, T* o) {  // o = G⁺ x = R' C⁻¹ C⁻¹ R x
      T y[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) { T s2 = T(0);
#pragma unroll
        for (int k = i; k < NV; ++k) s2 += R[i][k] * x[k];
        y[i] = s2; }
      csolve(y);
      csolve(y);
#pragma unroll
      for (int k = 0; k < NV; ++k) { T s2 = T(0);
#pragma unroll
        for (int i = 0; i <= k; ++i) s2 += R[i][k] * y[i];
        o[k] = s2; }
    };
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      T s2 = T(0);
#pragma unroll
      for (int c = 0; c < NC; ++c) s2 += Y[c][i] * bv[c];
      t2[i] = s2;
    }
    {
      T g1[NV];
      gplus(t2, g1);
      gplus(g1, w);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      T s2 = T(0);
#pragma unroll
      for (int i = 0; i < NV; ++i) s2 += Y[c][i] * w[i];
      lam[c] = s2;
    }
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) {
      T s2 = T(0);
#pragma unroll
      for (int c = 0; c < NC; ++c) s2 += K[c][vi] * lam[c];
      rhs[vi] -= s2;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c < nc) {
        if (lambda) lambda[(long)c * Lc.sk + st * Lc.sb] = lam[c];
        kg[(long)c * Lc.sk + st * Lc.sb] = kk[c];
#pragma unroll
        for (int vi = 0; vi < NV; ++vi)
          if (vi < nv) Kg[((long)vi * nc + c) * Lk.sk + st * Lk.sb] = K[c][vi];
      }
    }
  }
  fwd(rhs); bwd(rhs);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (i < nv) vdot[(long)i * Lv.sk + st * Lv.sb] = rhs[i];
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
// LOCAL (a VIEW of compile-time tables): M, c and the per-body kinematics also stay in the thread's registers for the solve — with run-time body / column
// indices those arrays would live in scratch memory, so the pointer VIEW reads back what the thread has just stored instead.
template <typename T, typename VIEW, int NB, int NV, int NC, bool LOCAL = false>
RBD_DEV void loop_fused_small_state(const VIEW& V, long st, int stabilize, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ tau,
                                    const T* __restrict__ fext, T* body, T* Mg, T* cg, T* __restrict__ vdot, T* __restrict__ qdot, T* __restrict__ lambda,
                                    T* __restrict__ Kg, T* __restrict__ kg, Layout Lq, Layout Lm, Layout Lv, Layout Lf, Layout Lc, Layout Lk, double g0,
                                    double g1, double g2, int* __restrict__ notpd) {
  const int nb = V.nb, nv = V.nv;
  T* bd = body + st * nb * 24;
  T Ml[NV * NV], cl[NV];  // (LOCAL)
#pragma unroll
  for (int k = 0; k < NV * NV; ++k) Ml[k] = T(0);
#pragma unroll
  for (int k = 0; k < NV; ++k) cl[k] = T(0);
  T S[NB][6], w[NB][6], Kb[NB][24];  // Kb: (R, p, T, a) of every body — a child takes its parent's from here (exported to `body` as well, for the solve)
  RInertia<T> Ic[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { S[i][k] = T(0); w[i][k] = T(0); Ic[i].J[k] = T(0); }
#pragma unroll
    for (int k = 0; k < 3; ++k) Ic[i].c[k] = T(0);
    Ic[i].m = T(0);
  }
  // ---- top-down: transforms, twists, bias accelerations (mechanism_state.jl:687-700, :769-780, :814-830), Newton-Euler wrench of every body
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    if (i < nb) {  // uniform
      Body<T> b{};
      b.jtype = V.jt[i]; b.qoff = V.xi[3 * i + 1]; b.voff = V.voff[i]; b.state = st; b.valid = true;
      const int p = V.xi[3 * i];
      const T* rb = V.rb + (long)V.xi[3 * i + 2] * RB_STRIDE;
      T qj[7], vj[6];
      load_joint_q(b, q, Lq, qj);
      load_joint_v(b, v, Lv, vj);
      store_qdot(b, qdot, Lq, qj, vj);
      T XR[9], Xp[3], tl[6], K[24], pk[24];
      local_transform(b, rb, qj, XR, Xp);
      local_joint_motion(b, rb, vj, tl);
#pragma unroll
      for (int k = 0; k < 24; ++k) pk[k] = (k < 9 && k % 4 == 0) ? T(1) : T(0);  // the world: identity, at rest, accelerating against gravity
      pk[21] = T(-g0); pk[22] = T(-g1); pk[23] = T(-g2);
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (r < i && r == p) {  // uniform: the parent's values, from registers
#pragma unroll
          for (int k = 0; k < 24; ++k) pk[k] = Kb[r][k];
        }
      }
      matmul3(pk, XR, K);
      matvec3(pk, Xp, K + 9);
#pragma unroll
      for (int k = 0; k < 3; ++k) K[9 + k] += pk[9 + k];
      T vJ[6], nT[6], cr[6];
      xmotion(K, K + 9, tl, vJ);
#pragma unroll
      for (int k = 0; k < 6; ++k) { K[12 + k] = pk[12 + k] + vJ[k]; nT[k] = -K[12 + k]; }
      se3_comm(nT, pk + 12, cr);  // a_b = a_p + (-T_b) x T_p  (v̇ = 0: spatial_accelerations! with zero joint accelerations)
#pragma unroll
      for (int k = 0; k < 6; ++k) K[18 + k] = pk[18 + k] + cr[k];
#pragma unroll
      for (int k = 0; k < 24; ++k) { if (!LOCAL) bd[i * 24 + k] = K[k]; Kb[i][k] = K[k]; }
      T Jb[6], mc[3], Ia[6], x[6], fe[6], e1[6] = {T(1), T(0), T(0), T(0), T(0), T(0)}, sl[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) Jb[k] = rb[RB_J + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) mc[k] = rb[RB_MC + k];
      inertia_to_root(Jb, mc, rb[RB_M], K, K + 9, Ic[i]);
      mul_inertia(Ic[i], K + 18, Ia);
      momentum_cross(Ic[i], K + 12, x);
#pragma unroll
      for (int k = 0; k < 6; ++k) fe[k] = fext ? fext[(long)(6 * i + k) * Lf.sk + st * Lf.sb] : T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) w[i][k] = Ia[k] + x[k] - fe[k];
      local_joint_motion(b, rb, e1, sl);
      xmotion(K, K + 9, sl, S[i]);
      if (joint_nv(b.jtype) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) S[i][k] = T(0);
      }
    }
  }
  // ---- structural zeros of the lower triangle first (the reference writes them: mechanism_algorithms.jl:266-267), then bottom-up: joint wrenches
  //      and composite inertias to the parents, c = S'w (dynamics_bias!), M[i, ancestors] = (Ic_i S_i)'S_a (mass_matrix!)
  for (int c2 = 0; c2 < nv; ++c2)
    for (int r = c2; r < nv; ++r) Mg[((long)c2 * nv + r) * Lm.sk + st * Lm.sb] = T(0);
#pragma unroll
  for (int i = NB - 1; i >= 0; --i) {
    if (i < nb) {
      const int p = V.xi[3 * i], vo = V.voff[i];
      const bool dof = joint_nv(V.jt[i]) == 1;
      if (dof) { const T ci = dot6(S[i], w[i]); cg[(long)vo * Lv.sk + st * Lv.sb] = ci; if (LOCAL) cl[vo] = ci; }
      if (dof) {
        T F[6];
        mul_inertia(Ic[i], S[i], F);
        int a = i;
#pragma unroll
        for (int r = NB - 1; r >= 0; --r) {
          if (r <= i && r == a) {  // r walks down the indices, a up the ancestors (a parent has a smaller index than its child)
            if (joint_nv(V.jt[r]) == 1) { const T mij = dot6(F, S[r]); Mg[((long)V.voff[r] * nv + vo) * Lm.sk + st * Lm.sb] = mij; if (LOCAL) Ml[V.voff[r] * nv + vo] = mij; }
            a = V.xi[3 * r];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (r < i && r == p) {
#pragma unroll
          for (int k = 0; k < 6; ++k) { w[r][k] += w[i][k]; Ic[r].J[k] += Ic[i].J[k]; }
#pragma unroll
          for (int k = 0; k < 3; ++k) Ic[r].c[k] += Ic[i].c[k];
          Ic[r].m += Ic[i].m;
        }
      }
    }
  }
  if (LOCAL) loop_solve_small_state<T, VIEW, NV, NC>(V, st, stabilize, &Kb[0][0], Ml, 1, cl, 1, tau, vdot, lambda, Kg, kg, Lv, Lc, Lk, g0, g1, g2, notpd);
  else loop_solve_small_state<T, VIEW, NV, NC>(V, st, stabilize, bd, Mg + st * Lm.sb, Lm.sk, cg + st * Lv.sb, Lv.sk, tau, vdot, lambda, Kg, kg, Lv, Lc, Lk, g0, g1, g2, notpd);
}

}  // namespace rbd
