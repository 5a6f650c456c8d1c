// rbd_track_plan.hpp — host-side plan of the track schedule the walk kernels run (rbd_walk.hpp; rbd_walk_plan.hpp adds their parking slots).
//
// The track mapping gives a state G lanes ("tracks").  The tree is cut into chains along its longest paths and the chains are
// packed on the tracks by list scheduling (same scheduler as rbd_chain_plan.hpp); at step s track g works on one body.
// What is new here, and what makes the step code short:
//   * canonical body frames: every body frame is re-based on the host by a constant rotation P_b so that the axis of its
//     1-dof joint is +z.  All quantities of the passes live in the ROOT frame (as in the reference,
//     src/mechanism_state.jl:744-748, :776, :842), so the re-basing changes no result: it only turns "rotate about an arbitrary
//     axis" into "mix two columns", and the joint's motion subspace into a column of the body's rotation matrix.
//     Constants per body: C = P_parent' * R(joint_to_predecessor) * P_b,  pp = P_parent' * p(joint_to_predecessor),
//     inertia J' = P_b' J P_b, c' = P_b' c.
//   * edges between bodies on the same track at consecutive steps stay in registers ("chained"); every other edge goes through
//     an LDS mailbox: A (parent kinematics, pass A), B (articulated hand-off of one cross child, pass B), C (parent
//     acceleration, pass C).
// Index bookkeeping and constant folding only.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rbd_chain_plan.hpp"
#include "rbd_device.hpp"
#include "rbd_hip.h"

namespace rbd {

struct TrackPlan {
  bool ok = false;
  int G = 0, ns = 0, nA = 0, nB = 0, has_floating = 0, general = 0;  // general: some joint is prismatic / fixed / sin-cos revolute (the fast path is revolute only)
  std::vector<int32_t> ri;  // [ns * G * TI_STRIDE] packed words
  std::vector<double> rr;   // [ns * G * TR_STRIDE]
  std::vector<int32_t> tab; // [ns * G] body slot or -1 (introspection / tests)
  std::vector<int32_t> sf;  // [ns] wave-uniform step flags: 1 some body not chained to the previous step of its track, 2 some A/C mailbox
                            // written, 4 some B mailbox read, 8 some hand-off leaves its track, 16 a 6-dof root (SF_* in rbd_walk.hpp)
};

namespace trackplan {
inline void frame_with_z(const double* a, double* P /*row-major 3x3, columns x y a*/) {
  int k = 0;
  for (int i = 1; i < 3; ++i) if (std::fabs(a[i]) < std::fabs(a[k])) k = i;
  double e[3] = {0, 0, 0}; e[k] = 1.0;
  const double d = a[0] * e[0] + a[1] * e[1] + a[2] * e[2];
  double x[3] = {e[0] - a[0] * d, e[1] - a[1] * d, e[2] - a[2] * d};
  const double n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  for (int i = 0; i < 3; ++i) x[i] /= n;
  const double y[3] = {a[1] * x[2] - a[2] * x[1], a[2] * x[0] - a[0] * x[2], a[0] * x[1] - a[1] * x[0]};
  for (int i = 0; i < 3; ++i) { P[3 * i] = x[i]; P[3 * i + 1] = y[i]; P[3 * i + 2] = a[i]; }
}
inline void mm(const double* A, const double* B, double* C) {  // C = A B
  double t[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
inline void mtm(const double* A, const double* B, double* C) {  // C = A' B
  double t[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
  memcpy(C, t, sizeof t);
}
}  // namespace trackplan

// ib / rb: the slot records of rbd_model (IB_*, RB_*), nb bodies in DFS pre-order slots
inline TrackPlan build_track_plan(int nb, const std::vector<int32_t>& ib, const std::vector<double>& rb, int G) {
  using namespace trackplan;
  TrackPlan P;
  P.G = G;
  auto I = [&](int s, int f) { return ib[(size_t)s * IB_STRIDE + f]; };
  for (int s = 0; s < nb; ++s) {
    const int t = I(s, IB_JTYPE);
    if (t == RBD_JOINT_REVOLUTE) continue;
    if (t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_FIXED || t == RBD_JOINT_SINCOS_REVOLUTE) { P.general = 1; continue; }
    if (t == RBD_JOINT_QUAT_FLOATING && I(s, IB_PARENT) < 0) { P.has_floating = 1; continue; }
    return P;  // 3-dof joints, inner 6-dof joints: the lane-per-body kernels
  }
  const ChainPlan cp = build_chain_plan(nb, ib, G);  // scheduling only: which body on which track at which step
  if (!cp.ok) return P;
  P.ns = cp.ns;
  P.tab = cp.tab;
  std::vector<int> st(nb, -1), tr(nb, -1);
  for (int s = 0; s < P.ns; ++s)
    for (int g = 0; g < G; ++g) {
      const int e = cp.tab[(size_t)s * G + g];
      if (e >= 0) { st[e] = s; tr[e] = g; }
    }
  auto chained = [&](int s) { const int p = I(s, IB_PARENT); return p >= 0 && tr[p] == tr[s] && st[p] == st[s] - 1; };
  // mailboxes: A/C slot per body with a cross (non-chained) child; one B slot per cross child, the children of one parent contiguous
  std::vector<int> aidx(nb, -1), bw(nb, -1), br0(nb, -1), nbr(nb, 0);
  for (int s = 0; s < nb; ++s) {
    int n = 0;
    for (int k = 0; k < I(s, IB_NCHILD); ++k) {
      const int c = I(s, IB_CHILD0 + k);
      if (!chained(c)) { if (n == 0) br0[s] = P.nB; bw[c] = P.nB++; ++n; }
    }
    nbr[s] = n;
    if (n > 0 || I(s, IB_JTYPE) == RBD_JOINT_QUAT_FLOATING) aidx[s] = P.nA++;  // a 6-dof root keeps its transform and S⁻ᵀτ there for pass B
  }
  if (P.nA > 4095 || P.nB > 4095 || nb > 255) return P;
  // canonical frames
  std::vector<double> Pb((size_t)nb * 9, 0.0);
  for (int s = 0; s < nb; ++s) {
    double* Ps = &Pb[(size_t)s * 9];
    const int t = I(s, IB_JTYPE);
    if (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_SINCOS_REVOLUTE) frame_with_z(&rb[(size_t)s * RB_STRIDE + RB_AXIS], Ps);
    else { Ps[0] = Ps[4] = Ps[8] = 1.0; }
  }
  const double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  P.ri.assign((size_t)P.ns * G * TI_STRIDE, 0);
  P.rr.assign((size_t)P.ns * G * TR_STRIDE, 0.0);
  P.sf.assign((size_t)P.ns, 0);
  for (int s = 0; s < P.ns; ++s)
    for (int g = 0; g < G; ++g) {
      int32_t* wi = &P.ri[((size_t)s * G + g) * TI_STRIDE];
      double* wr = &P.rr[((size_t)s * G + g) * TR_STRIDE];
      wr[TR_C] = wr[TR_C + 4] = wr[TR_C + 8] = 1.0;  // idle rows: harmless identity
      const int e = cp.tab[(size_t)s * G + g];
      if (e < 0) continue;
      const int p = I(e, IB_PARENT), t = I(e, IB_JTYPE);
      const double* r = &rb[(size_t)e * RB_STRIDE];
      const double* Pp = p < 0 ? Id : &Pb[(size_t)p * 9];
      const double* Pe = &Pb[(size_t)e * 9];
      double XR[9], tmp[9];
      for (int k = 0; k < 9; ++k) XR[k] = r[RB_XPR + k];
      mm(XR, Pe, tmp);
      mtm(Pp, tmp, &wr[TR_C]);
      for (int i = 0; i < 3; ++i) wr[TR_PP + i] = Pp[i] * r[RB_XPP] + Pp[3 + i] * r[RB_XPP + 1] + Pp[6 + i] * r[RB_XPP + 2];
      const double J[9] = {r[RB_J], r[RB_J + 1], r[RB_J + 2], r[RB_J + 1], r[RB_J + 3], r[RB_J + 4], r[RB_J + 2], r[RB_J + 4], r[RB_J + 5]};
      double JP[9], PJP[9];
      mm(J, Pe, JP);
      mtm(Pe, JP, PJP);
      wr[TR_J] = PJP[0]; wr[TR_J + 1] = PJP[1]; wr[TR_J + 2] = PJP[2]; wr[TR_J + 3] = PJP[4]; wr[TR_J + 4] = PJP[5]; wr[TR_J + 5] = PJP[8];
      for (int i = 0; i < 3; ++i) wr[TR_MC + i] = Pe[i] * r[RB_MC] + Pe[3 + i] * r[RB_MC + 1] + Pe[6 + i] * r[RB_MC + 2];
      wr[TR_M] = r[RB_M];
      bool carry = false;
      for (int k = 0; k < I(e, IB_NCHILD); ++k) carry |= chained(I(e, IB_CHILD0 + k));
      int flags = TF_VALID | (p < 0 ? TF_LEVEL0 : 0) | (chained(e) ? TF_CHAINED : 0) | (carry ? TF_CARRY : 0);
      if (t == RBD_JOINT_QUAT_FLOATING) flags |= TF_FLOATING;
      if (t == RBD_JOINT_PRISMATIC) flags |= TF_PRISMATIC;
      if (t == RBD_JOINT_FIXED) flags |= TF_FIXED;
      if (t == RBD_JOINT_SINCOS_REVOLUTE) flags |= TF_SINCOS;
      const int a_w = aidx[e], a_r = (p >= 0 && !chained(e)) ? aidx[p] : -1;
      wi[TI_W0] = I(e, IB_QOFF) | (I(e, IB_VOFF) << 16);
      wi[TI_W1] = (6 * I(e, IB_ORIG)) | (flags << 16) | (nbr[e] << 24);
      wi[TI_W2] = (a_w + 1) | ((a_r + 1) << 16);
      wi[TI_W3] = (bw[e] + 1) | ((br0[e] + 1) << 16);
      P.sf[s] |= (chained(e) ? 0 : 1) | (a_w >= 0 ? 2 : 0) | (nbr[e] > 0 ? 4 : 0) | ((bw[e] >= 0 || p < 0) ? 8 : 0) | (t == RBD_JOINT_QUAT_FLOATING ? 16 : 0);
    }
  P.ok = true;
  return P;
}

}  // namespace rbd
