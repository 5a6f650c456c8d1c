// rbd_chain_plan.hpp — host-side plan for aba_chain_kernel (rbd_chain.hpp): cut the tree into chains and pack them on G tracks.
//
//   chains   : from every head (a root, or a child that is not its parent's "chain child") follow the child with the tallest
//              subtree — longest paths first, so the critical path of the tree is one chain;
//   packing  : heads in order of (level, longest first); a chain of length L whose parent finished at step t may start at any
//              step > t on any track with L free steps — take the earliest (greedy list scheduling);
//   edges    : child on the same track one step after its parent = "chained" (registers); all other edges use LDS mailboxes.
// Everything here is index bookkeeping; no floating-point data is touched.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "rbd_device.hpp"
#include "rbd_hip.h"

namespace rbd {

struct ChainPlan {
  bool ok = false;
  int G = 0, ns = 0, nfl = 0, nfs = 0;
  std::vector<int32_t> tab, cb;
  std::vector<uint8_t> nrounds;
  size_t lds_fields(int nb) const { return (size_t)nb * CS_FIELDS + nfl + nfs; }  // x (64 / G) x sizeof(T) bytes
};

// ib: the slot records of rbd_model (IB_*), nb bodies in DFS pre-order slots
inline ChainPlan build_chain_plan(int nb, const std::vector<int32_t>& ib, int G) {
  ChainPlan P;
  P.G = G;
  auto I = [&](int s, int f) { return ib[(size_t)s * IB_STRIDE + f]; };
  for (int s = 0; s < nb; ++s) {
    const int t = I(s, IB_JTYPE);
    const bool simple = t == RBD_JOINT_FIXED || t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_SINCOS_REVOLUTE;
    if (!simple && !(t == RBD_JOINT_QUAT_FLOATING && I(s, IB_PARENT) < 0)) return P;  // 3-dof joints, inner 6-dof joints: aba_kernel
  }
  std::vector<int> height(nb, 1), cc(nb, -1), st(nb, -1), tr(nb, -1);
  for (int s = nb - 1; s >= 0; --s) {  // children have larger slots
    for (int k = 0; k < I(s, IB_NCHILD); ++k) {
      const int c = I(s, IB_CHILD0 + k);
      if (height[c] + 1 > height[s]) { height[s] = height[c] + 1; cc[s] = c; }
    }
  }
  struct Head { int h, level, len; };
  std::vector<Head> heads;
  for (int s = 0; s < nb; ++s) {
    const int p = I(s, IB_PARENT);
    if (p < 0 || cc[p] != s) heads.push_back({s, I(s, IB_LEVEL), height[s]});
  }
  std::stable_sort(heads.begin(), heads.end(), [](const Head& a, const Head& b) { return a.level != b.level ? a.level < b.level : a.len > b.len; });
  std::vector<std::vector<char>> busy(G, std::vector<char>(2 * nb + 2, 0));
  for (const Head& hd : heads) {
    const int p = I(hd.h, IB_PARENT);
    const int est = p < 0 ? 0 : st[p] + 1;
    int best_t = 1 << 30, best_g = 0;
    for (int g = 0; g < G; ++g) {
      int t = est;
      for (;;) {
        int k = 0;
        while (k < hd.len && !busy[g][t + k]) ++k;
        if (k == hd.len) break;
        t += k + 1;
      }
      if (t < best_t) { best_t = t; best_g = g; }
    }
    int s = hd.h;
    for (int k = 0; k < hd.len; ++k, s = cc[s]) { st[s] = best_t + k; tr[s] = best_g; busy[best_g][best_t + k] = 1; }
  }
  P.ns = 0;
  for (int s = 0; s < nb; ++s) P.ns = std::max(P.ns, st[s] + 1);
  if (P.ns > MAX_LEVELS) return P;
  P.tab.assign((size_t)P.ns * G, -1);
  for (int s = 0; s < nb; ++s) P.tab[(size_t)st[s] * G + tr[s]] = s;
  auto chained = [&](int s) { const int p = I(s, IB_PARENT); return p >= 0 && tr[p] == tr[s] && st[p] == st[s] - 1; };
  // restart: going up in pass B, the registers of the track do not hold this body's kinematics
  std::vector<char> restart(nb, 0), isP(nb, 0);
  for (int s = 0; s < nb; ++s) {
    int next = -1;  // next body on the same track
    for (int t = st[s] + 1; t < P.ns && next < 0; ++t) next = P.tab[(size_t)t * G + tr[s]];
    if (next >= 0 && !(chained(next) && I(next, IB_PARENT) == s)) restart[s] = 1;
    const int p = I(s, IB_PARENT);
    if (p >= 0 && !chained(s)) isP[p] = 1;
  }
  std::vector<int> idxL(nb, -1), idxP(nb, -1);
  int nL = 0, nP = 0;
  for (int s = 0; s < nb; ++s) {
    if (restart[s]) idxL[s] = nL++;
    if (isP[s]) idxP[s] = nP++;
  }
  P.nfl = nL * MB_A_FIELDS;
  P.nfs = nP * MB_B_FIELDS;
  P.cb.assign((size_t)nb * CB_STRIDE, -1);
  P.nrounds.assign(MAX_LEVELS, 0);
  auto mba_w = [&](int s) { return restart[s] ? idxL[s] * MB_A_FIELDS : (isP[s] ? P.nfl + idxP[s] * MB_A_FIELDS : -1); };
  for (int s = 0; s < nb; ++s) {
    int32_t* c = &P.cb[(size_t)s * CB_STRIDE];
    const int p = I(s, IB_PARENT);
    const bool ch = chained(s), cross = p >= 0 && !ch;
    bool carry = false;
    for (int k = 0; k < I(s, IB_NCHILD); ++k) carry |= chained(I(s, IB_CHILD0 + k));
    c[CB_JTYPE] = I(s, IB_JTYPE); c[CB_QOFF] = I(s, IB_QOFF); c[CB_VOFF] = I(s, IB_VOFF); c[CB_ORIG] = I(s, IB_ORIG);
    c[CB_FLAGS] = (p < 0 ? CF_LEVEL0 : 0) | (ch ? CF_CHAINED : 0) | (restart[s] ? CF_RESTART : 0) | (carry ? CF_CARRY : 0);
    c[CB_MBA_W] = mba_w(s);
    c[CB_MBA_R] = cross ? mba_w(p) : -1;
    c[CB_ACC_W] = cross ? P.nfl + idxP[p] * MB_B_FIELDS : -1;
    c[CB_ACC_R] = isP[s] ? P.nfl + idxP[s] * MB_B_FIELDS : -1;
    c[CB_MBC_W] = isP[s] ? P.nfl + idxP[s] * MB_C_FIELDS : -1;
    c[CB_MBC_R] = cross ? P.nfl + idxP[p] * MB_C_FIELDS : -1;
    int round = 0;  // position among the cross-track siblings that finish at the same step
    if (cross)
      for (int k = 0; k < I(p, IB_NCHILD); ++k) {
        const int o = I(p, IB_CHILD0 + k);
        if (o < s && !chained(o) && st[o] == st[s]) ++round;
      }
    c[CB_ROUND] = round;
    if (cross && round + 1 > P.nrounds[st[s]]) P.nrounds[st[s]] = (uint8_t)(round + 1);
  }
  P.ok = true;
  return P;
}

}  // namespace rbd
