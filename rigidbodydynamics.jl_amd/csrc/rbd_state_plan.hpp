// rbd_state_plan.hpp — host-side plan of the one-lane-per-state kernels (rbd_state.hpp).
//
// The kernels walk the tree depth-first: ENTER(body) on the way down, EXIT(body) once its subtree is finished.  The plan is that walk
// written out as a flat op list (the slots of rbd_model are already in DFS pre-order), the table of ancestor velocity columns every EXIT
// needs, and the body constants re-based to canonical frames (joint axis = +z; same construction as rbd_track_plan.hpp).
// Index bookkeeping and constant folding only.
#pragma once
#include <cstdint>
#include <vector>

#include "rbd_device.hpp"
#include "rbd_hip.h"
#include "rbd_track_plan.hpp"

namespace rbd {

struct StatePlan {
  bool ok = false;
  bool wide = false;  // 3-dof joints or 6-dof joints below the world on the walk: the kernels compiled for the mechanism take them (rbd_spec.hpp), the interpreting ones do not
  int n3 = 0;         // 3-dof joints
  int nops = 0, nlevels = 0;
  std::vector<int32_t> ops;   // [nops * SO_STRIDE]
  std::vector<int32_t> cols;  // [nops * SC_STRIDE] (by op): velocity column of the ancestor at level k (k < level), -1 for a fixed joint, | SC_FLOATING for a 6-dof root
  std::vector<double> sr;     // [nops * TR_STRIDE] (by op)
};

inline StatePlan build_state_plan(int nb, const std::vector<int32_t>& ib, const std::vector<double>& rb, bool wide = false) {
  using namespace trackplan;
  StatePlan P;
  auto I = [&](int s, int f) { return ib[(size_t)s * IB_STRIDE + f]; };
  for (int s = 0; s < nb; ++s) {
    const int t = I(s, IB_JTYPE);
    if (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_FIXED || t == RBD_JOINT_SINCOS_REVOLUTE) continue;
    if (t == RBD_JOINT_QUAT_FLOATING && I(s, IB_PARENT) < 0) continue;
    if (!wide) return P;  // 3-dof joints, inner 6-dof joints: the lane-per-body kernels, or the kernels compiled for the mechanism (wide plan)
    if (t != RBD_JOINT_QUAT_FLOATING && t != RBD_JOINT_QUAT_SPHERICAL && t != RBD_JOINT_PLANAR) return P;
    P.wide = true;
    if (t != RBD_JOINT_QUAT_FLOATING) ++P.n3;
  }
  for (int s = 0; s < nb; ++s) {
    if (I(s, IB_LEVEL) + 1 > P.nlevels) P.nlevels = I(s, IB_LEVEL) + 1;
    if (s > 0 && I(s, IB_PARENT) >= s) return P;  // not pre-order (never happens for rbd_model slots)
  }
  if (nb == 0 || P.nlevels > SC_STRIDE) return P;
  auto op = [&](int kind, int s) {
    int32_t w[SO_STRIDE] = {0};
    const int t = I(s, IB_JTYPE);
    w[SO_W0] = kind | (I(s, IB_LEVEL) << 8) | (t << 16);
    w[SO_SLOT] = s; w[SO_QOFF] = I(s, IB_QOFF); w[SO_VOFF] = I(s, IB_VOFF); w[SO_ORIG6] = 6 * I(s, IB_ORIG);
    P.ops.insert(P.ops.end(), w, w + SO_STRIDE);
    ++P.nops;
  };
  std::vector<int> path;  // open bodies, by level
  for (int s = 0; s < nb; ++s) {
    const int l = I(s, IB_LEVEL);
    while ((int)path.size() > l) { op(SK_EXIT, path.back()); path.pop_back(); }
    if ((int)path.size() != l || (l > 0 && path.back() != I(s, IB_PARENT))) return StatePlan();
    op(SK_ENTER, s);
    path.push_back(s);
  }
  while (!path.empty()) { op(SK_EXIT, path.back()); path.pop_back(); }
  P.cols.assign((size_t)nb * SC_STRIDE, -1);
  for (int s = 0; s < nb; ++s) {
    int a = I(s, IB_PARENT);
    while (a >= 0) {
      const int t = I(a, IB_JTYPE);
      int32_t c = -1;
      if (t == RBD_JOINT_QUAT_FLOATING) c = I(a, IB_VOFF) | SC_FLOATING;
      else if (t == RBD_JOINT_QUAT_SPHERICAL) c = I(a, IB_VOFF) | SC_SPHERICAL;
      else if (t == RBD_JOINT_PLANAR) c = I(a, IB_VOFF) | SC_PLANAR;
      else if (t != RBD_JOINT_FIXED) c = I(a, IB_VOFF);
      P.cols[(size_t)s * SC_STRIDE + I(a, IB_LEVEL)] = c;
      a = I(a, IB_PARENT);
    }
  }
  // canonical frames (rbd_track_plan.hpp): C = P_parent' R(joint_to_predecessor) P_b, pp = P_parent' p, J' = P_b' J P_b, c' = P_b' c
  std::vector<double> Pb((size_t)nb * 9, 0.0);
  for (int s = 0; s < nb; ++s) {
    double* Ps = &Pb[(size_t)s * 9];
    const int t = I(s, IB_JTYPE);
    if (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_SINCOS_REVOLUTE) frame_with_z(&rb[(size_t)s * RB_STRIDE + RB_AXIS], Ps);
    else if (t == RBD_JOINT_PLANAR) {  // columns x, y, x × y (planar.jl:65-86): the joint translates along +x, +y and turns about +z
      const double* ax = &rb[(size_t)s * RB_STRIDE + RB_AXIS];
      const double* ay = &rb[(size_t)s * RB_STRIDE + RB_AXIS2];
      const double az[3] = {ax[1] * ay[2] - ax[2] * ay[1], ax[2] * ay[0] - ax[0] * ay[2], ax[0] * ay[1] - ax[1] * ay[0]};
      for (int i = 0; i < 3; ++i) { Ps[3 * i] = ax[i]; Ps[3 * i + 1] = ay[i]; Ps[3 * i + 2] = az[i]; }
    } else { Ps[0] = Ps[4] = Ps[8] = 1.0; }
  }
  const double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  P.sr.assign((size_t)nb * TR_STRIDE, 0.0);
  for (int e = 0; e < nb; ++e) {
    double* wr = &P.sr[(size_t)e * TR_STRIDE];
    const int p = I(e, IB_PARENT);
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
  }
  // the kernels read everything of an op through one index (no slot indirection on the scalar path): expand the tables per op
  {
    std::vector<int32_t> oc((size_t)P.nops * SC_STRIDE);
    std::vector<double> orr((size_t)P.nops * TR_STRIDE);
    for (int o = 0; o < P.nops; ++o) {
      const int s = P.ops[(size_t)o * SO_STRIDE + SO_SLOT];
      for (int k = 0; k < SC_STRIDE; ++k) oc[(size_t)o * SC_STRIDE + k] = P.cols[(size_t)s * SC_STRIDE + k];
      for (int k = 0; k < TR_STRIDE; ++k) orr[(size_t)o * TR_STRIDE + k] = P.sr[(size_t)s * TR_STRIDE + k];
    }
    P.cols.swap(oc);
    P.sr.swap(orr);
  }
  P.ok = true;
  return P;
}

}  // namespace rbd
