// rbd_integrator.hpp — device functions of the Munthe-Kaas RK4 integrator (`simulate`, src/simulate.jl:36-55;
// MuntheKaasIntegrator.step, src/ode_integrators.jl:233-299; runge_kutta_4, :48-55).  Included by rbd_kernels.hip after the
// per-lane load/store helpers; used by mk_stage_kernel and, fused, by aba_kernel.
// local/global coordinates: joint_types.jl:9-18 (default), sin_cos_revolute.jl:173-196, quaternion_spherical.jl:139-154,
// quaternion_floating.jl:205-249 with log_with_time_derivative / exp of src/spatial/spatialmotion.jl:226-332 and
// rotation_vector_rate of src/spatial/util.jl:88-102.  Rotations are composed as quaternions (the reference goes through
// rotation matrices and converts back; identical up to rounding and the sign of the quaternion).
#pragma once
namespace rbd {

template <typename T> RBD_DEV void quat_mul(const T* a, const T* b, T* o) {
  const T w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const T x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const T y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const T z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
template <typename T> RBD_DEV T eps_t();
template <> RBD_DEV double eps_t<double>() { return 2.220446049250313e-16; }
template <> RBD_DEV float eps_t<float>() { return 1.1920929e-7f; }
RBD_DEV double atan2_t(double y, double x) { return atan2(y, x); }
RBD_DEV float atan2_t(float y, float x) { return atan2f(y, x); }
template <typename T> RBD_DEV T norm3(const T* a) { return SqrtT<T>::f(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// QuatRotation(RotationVec(r))
template <typename T> RBD_DEV void quat_from_rotvec(const T* r, T* q) {
  const T th = norm3(r);
  T s, c;
  sincos_t(th / 2, &s, &c);
  const T k = th < eps_t<T>() ? T(0.5) : s / th;
  q[0] = c; q[1] = k * r[0]; q[2] = k * r[1]; q[3] = k * r[2];
}
// RotationVec(quat)
template <typename T> RBD_DEV void rotvec_from_quat(const T* q, T* r) {
  const T s = norm3(q + 1);
  const T th = 2 * atan2_t(s, q[0]);
  const T k = s < eps_t<T>() ? T(2) : th / s;
  r[0] = k * q[1]; r[1] = k * q[2]; r[2] = k * q[3];
}
template <typename T> RBD_DEV void quat_rotate(const T* q, const T* x, T* o) {  // R(q) x
  T R[9];
  rot_quat(q[0], q[1], q[2], q[3], R);
  matvec3(R, x, o);
}

// ϕ̇ of local_coordinates! for one joint
// MODE 0: every joint type (the fused launches); 1: only the element-wise types (revolute, prismatic, sin-cos, planar); 2: only the
// quaternion types — mk_stage_kernel runs as two launches so that the light one keeps few registers and many wavefronts in flight
template <typename T, int MODE = 0> RBD_DEV void joint_local_rate(int t, const T* q0, const T* q, const T* v, T* o) {
#pragma unroll
  for (int k = 0; k < 6; ++k) o[k] = T(0);
  if (MODE != 2 && (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_SINCOS_REVOLUTE)) {
    o[0] = v[0];
  } else if (MODE != 2 && t == RBD_JOINT_PLANAR) {
    T s, c;
    sincos_t(q[2], &s, &c);
    o[0] = c * v[0] - s * v[1]; o[1] = s * v[0] + c * v[1]; o[2] = v[2];
  } else if (MODE != 1 && t == RBD_JOINT_QUAT_SPHERICAL) {
    const T q0c[4] = {q0[0], -q0[1], -q0[2], -q0[3]};
    T dq[4], phi[3], c1[3], c2[3];
    quat_mul(q0c, q, dq);
    rotvec_from_quat(dq, phi);
    cross3(phi, v, c1);
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = v[k] + c1[k] / 2;
    const T th = norm3(phi);
    {  // Bortz equation, spatial/util.jl:88-102: f = (1 - θ sin θ / (2 (1 - cos θ))) / θ².  Evaluated as written it divides by 1 - cos θ, which is exactly zero in
       // fp32 for θ < 3e-4 (and θ is rounding noise, not zero, whenever q = q0: the first stage of every step) — inf · 0 further down; and it loses every digit to
       // cancellation long before.  (θ/2) cot(θ/2) = 1 - θ²/12 - θ⁴/720 - ..., so f = 1/12 + θ²/720 + θ⁴/30240 + θ⁶/1209600 + θ⁸/47900160 for small θ.
      const T t2 = th * th;
      T f;
      if (th < (sizeof(T) == 4 ? T(0.5) : T(1e-2))) {
        f = T(1) / T(12) + t2 * (T(1) / T(720) + t2 * (T(1) / T(30240) + t2 * (T(1) / T(1209600) + t2 * (T(1) / T(47900160)))));
      } else {
        T s, c;
        sincos_t(th, &s, &c);
        f = (1 - (th * s) / (2 * (1 - c))) / t2;
      }
      cross3(phi, c1, c2);
#pragma unroll
      for (int k = 0; k < 3; ++k) o[k] += f * c2[k];
    }
  } else if (MODE != 1 && t == RBD_JOINT_QUAT_FLOATING) {
    // relative transform inv(T0) T, then log_with_time_derivative with the body twist (ω, v).  This branch sits on the critical path
    // of the fused `simulate` launches (one lane per state runs it while the wavefront waits), so it spends one atan2, two
    // square roots and three reciprocals: sin/cos of θ/2 are the relative quaternion's own (normalised) parts, θ is not
    // re-derived from ψ, and every division by θ, θ², θ⁴, sin(θ/2) is a multiplication by a reciprocal formed once.
    const T q0c[4] = {q0[0], -q0[1], -q0[2], -q0[3]};
    T dq[4], d[3], dp[3], psi[3];
    quat_mul(q0c, q, dq);
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = q[4 + k] - q0[4 + k];
    quat_rotate(q0c, d, dp);
    const T sv = norm3(dq + 1);                      // |vector part| = |dq| sin(θ/2)
    const T th = 2 * atan2_t(sv, dq[0]);             // rotation angle (RotationVec(quat))
    const T kq = sv < eps_t<T>() ? T(2) : th * rcp_nr(sv);
#pragma unroll
    for (int k = 0; k < 3; ++k) psi[k] = kq * dq[1 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = v[k];
    if (th > eps_t<T>()) {
      const T inorm = rcp_nr(SqrtT<T>::f(dq[0] * dq[0] + sv * sv));
      const T s2 = sv * inorm, c2 = dq[0] * inorm;   // sin(θ/2), cos(θ/2)
      const T is2 = rcp_nr(s2), ith2 = rcp_nr(th * th), h = th / 2;
      const T alpha = h * c2 * is2;
      T x1[3], x2[3], qv[3];
      cross3(psi, dp, x1);
      cross3(psi, x1, x2);
      const T ca = (1 - alpha) * ith2;
#pragma unroll
      for (int k = 0; k < 3; ++k) qv[k] = dp[k] - x1[k] / 2 + ca * x2[k];
      const T beta = h * h * is2 * is2;
      const T A = (2 * (1 - alpha) + (alpha - beta) / 2) * ith2;
      const T Bc = ((1 - alpha) + (alpha - beta) / 2) * ith2 * ith2;
      T X[6], a1[6], a2[6], a3[6], a4[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) { X[k] = psi[k]; X[3 + k] = qv[k]; }
      se3_comm(X, v, a1);
      se3_comm(X, a1, a2);
      se3_comm(X, a2, a3);
      se3_comm(X, a3, a4);
#pragma unroll
      for (int k = 0; k < 6; ++k) o[k] = v[k] + a1[k] / 2 + A * a2[k] + Bc * a4[k];
    }
  }
}

// global_coordinates! for one joint
template <typename T, int MODE = 0> RBD_DEV void joint_global(int t, const T* q0, const T* phi, T* q) {
#pragma unroll
  for (int k = 0; k < 7; ++k) q[k] = T(0);
  if (MODE != 2 && (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC)) {
    q[0] = q0[0] + phi[0];
  } else if (MODE != 2 && t == RBD_JOINT_PLANAR) {
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = q0[k] + phi[k];
  } else if (MODE != 2 && t == RBD_JOINT_SINCOS_REVOLUTE) {
    T sd, cd;
    sincos_t(phi[0], &sd, &cd);
    q[0] = q0[0] * cd + q0[1] * sd;
    q[1] = q0[1] * cd - q0[0] * sd;
  } else if (MODE != 1 && t == RBD_JOINT_QUAT_SPHERICAL) {
    T dq[4];
    quat_from_rotvec(phi, dq);
    quat_mul(q0, dq, q);
  } else if (MODE != 1 && t == RBD_JOINT_QUAT_FLOATING) {
    T dq[4], tr[3], w[3];
    const T th = norm3(phi);
    if (th < eps_t<T>()) {
      dq[0] = T(1); dq[1] = phi[0] / 2; dq[2] = phi[1] / 2; dq[3] = phi[2] / 2;  // quat_from_rotvec in the small-angle branch
#pragma unroll
      for (int k = 0; k < 3; ++k) tr[k] = phi[3 + k];
    } else {  // exp(::Twist), spatialmotion.jl:311-332 (2.36); one reciprocal of θ serves the quaternion, ω/θ and v/θ
      const T ith = rcp_nr(th);
      T sh, ch;
      sincos_t(th / 2, &sh, &ch);
      const T kq = sh * ith;
      dq[0] = ch; dq[1] = kq * phi[0]; dq[2] = kq * phi[1]; dq[3] = kq * phi[2];
      T om[3], vv[3], c[3], Rc[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { om[k] = phi[k] * ith; vv[k] = phi[3 + k] * ith; }
      cross3(om, vv, c);
      quat_rotate(dq, c, Rc);
      const T d = (om[0] * vv[0] + om[1] * vv[1] + om[2] * vv[2]) * th;
#pragma unroll
      for (int k = 0; k < 3; ++k) tr[k] = c[k] - Rc[k] + om[k] * d;
    }
    quat_mul(q0, dq, q);
    quat_rotate(q0, tr, w);
#pragma unroll
    for (int k = 0; k < 3; ++k) q[4 + k] = q0[4 + k] + w[k];
  }
}



// One lane's share of stage `stage` (0..4) of a step: see mk_stage_kernel.  qj / vj: the lane's joint state as loaded from the
// state buffers (previous stage state) in, the new stage state out; the new state is also written to q_state / v_state.
// vdot_prev == nullptr means W.vd[stage-1] already holds the previous stage's v̇ (fused launches).
// (Round 6, measured and taken back: the stage split into a half that issues EVERY load — both banks' in the two-bodies-per-lane kernel — and a half that computes
//  and stores, so that no load waits behind a store in the wavefront's one in-order memory counter.  aba_bank_fused_spec_f64 at 4096 Atlas states: 27.98 us against
//  27.13 — the general form's zeroed arrays and selects cost 440 vector instructions per wavefront and the waits did not move: profiles/r06_experiments.txt.)
template <typename T, int MODE = 0>
RBD_DEV void mk_stage_lane(const Body<T>& b, int stage, T dt, T* qj, T* vj, const T* __restrict__ vdot_prev, const MkBuffers& W,
                           T* __restrict__ q_state, T* __restrict__ v_state, Layout Lq, Layout Lv) {
  const int t = b.jtype;
  const int nq = joint_nq<T>(t), nv = joint_nv(t);
  T q0j[7], v0j[6];
  T rate[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};  // local-coordinate rates of the stage just evaluated (kept in registers: the
                                                     // tableau below needs them again, and reading them back from phid[stage-1]
                                                     // would be a store -> load round trip through L2 on the critical path)
  T vdp[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};   // v̇ of the stage just evaluated, when the caller hands it over (un-fused launches)
  T* q0 = (T*)W.q0; T* v0 = (T*)W.v0;
  if (stage == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) { q0j[k] = qj[k]; if (b.valid && k < nq) q0[(long)(b.qoff + k) * Lq.sk + b.state * Lq.sb] = qj[k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { v0j[k] = vj[k]; if (b.valid && k < nv) v0[(long)(b.voff + k) * Lv.sk + b.state * Lv.sb] = vj[k]; }
  } else {
    load_joint_q(b, q0, Lq, q0j);
    load_joint_v(b, v0, Lv, v0j);
    // rates of the stage that has just been evaluated
    joint_local_rate<T, MODE>(t, q0j, qj, vj, rate);
    T* pd = (T*)W.phid[stage - 1];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (b.valid && k < nv) pd[(long)(b.voff + k) * Lv.sk + b.state * Lv.sb] = rate[k];
    if (vdot_prev != nullptr) {
      load_joint_v(b, vdot_prev, Lv, vdp);
      T* vs = (T*)W.vd[stage - 1];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (b.valid && k < nv) vs[(long)(b.voff + k) * Lv.sk + b.state * Lv.sb] = vdp[k];
    }
  }
  // Butcher tableau of runge_kutta_4 (ode_integrators.jl:48-55)
  T w[4] = {T(0), T(0), T(0), T(0)};
  if (stage == 1) w[0] = T(0.5);
  else if (stage == 2) w[1] = T(0.5);
  else if (stage == 3) w[2] = T(1);
  else if (stage == 4) { w[0] = T(1) / 6; w[1] = T(1) / 3; w[2] = T(1) / 3; w[3] = T(1) / 6; }
  T phi[6], vn[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { phi[k] = T(0); vn[k] = v0j[k]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (w[j] != T(0)) {  // uniform
      const T wj = dt * w[j];
      const T* pd = (const T*)W.phid[j]; const T* vs = (const T*)W.vd[j];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (b.valid && k < nv) {
          const long a = (long)(b.voff + k) * Lv.sk + b.state * Lv.sb;
          phi[k] += wj * ((j == stage - 1) ? rate[k] : pd[a]);
          vn[k] += wj * ((j == stage - 1 && vdot_prev != nullptr) ? vdp[k] : vs[a]);
        }
    }
  }
  T qn[7];
  joint_global<T, MODE>(t, q0j, phi, qn);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    qj[k] = qn[k];
    if (b.valid && k < nq) q_state[(long)(b.qoff + k) * Lq.sk + b.state * Lq.sb] = qn[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    vj[k] = vn[k];
    if (b.valid && k < nv) v_state[(long)(b.voff + k) * Lv.sk + b.state * Lv.sb] = vn[k];
  }
}

}  // namespace rbd
