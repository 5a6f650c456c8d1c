// rbd_chain_plan.hpp — the scheduler under the track / walk plans (rbd_track_plan.hpp): cut the tree into chains and pack them on G tracks.
// (Round 1 ran a kernel directly on this plan — aba_chain_kernel; it lost at every batch size and was removed in round 3.)
//
//   chains   : from every head (a root, or a child that is not its parent's "chain child") follow the child with the tallest
//              subtree — longest paths first, so the critical path of the tree is one chain;
//   packing  : heads in order of (level, longest first); a chain of length L whose parent finished at step t may start at any
//              step > t on any track with L free steps — take the earliest (greedy list scheduling);
//   edges    : child on the same track one step after its parent = "chained" (registers); all other edges use LDS mailboxes (rbd_track_plan.hpp).
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
  int G = 0, ns = 0;
  std::vector<int32_t> tab;  // [ns * G] body slot on track g at step s, or -1
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
  P.ok = true;
  return P;
}

}  // namespace rbd
