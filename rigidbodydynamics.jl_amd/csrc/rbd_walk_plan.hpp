// rbd_walk_plan.hpp — host-side additions to the track plan (rbd_track_plan.hpp) for aba_walk_kernel (rbd_walk.hpp).
//
// The walk kernel runs the SAME schedule as the track mapping (track g works on body tab[s][g] at step s, same packed records, same
// mailbox numbering) but gives every track a whole wavefront whose 64 lanes are 64 states.  It keeps no per-body results of pass A:
// pass B re-derives a body's kinematics from its chained child's ("un-composing" the joint), so only a body WITHOUT a chained child
// needs its kinematics parked between the passes — except the body a track visits last in pass A, which pass B meets first with the
// registers still holding it.  This file numbers those parking slots and sizes the workgroup's LDS.
// Index bookkeeping only.
#pragma once
#ifndef RBD_JIT_COMPILE  // (the kernel compiled per mechanism at run time includes this file for the constants below only)
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>
#endif

#include "rbd_device.hpp"

namespace rbd {

// rows of 64 states (padded to WR_STRIDE scalars): q | v | tau (v̇ written over it) | A mailboxes, transform halves | parking slots | B mailboxes
// (pass A has the twist halves of its A mailboxes there, pass C its 6-value mailboxes); then the plan records.  Must agree with walk_ctx_lds()
// of rbd_walk.hpp.
enum { WR_STRIDE = 65, WMB_A = 12, WMB_AT = 12, WMB_B = 27, WMB_C = 6, WMB_S = 24, WALK_MAX_STEPS = 11 };
// values a step of a track leaves in the numbered accumulation registers between the passes (WalkStash, rbd_walk.hpp: WS_N); rbd_jit.hip sizes the stash with it
enum { WALK_STASH_N = 9 };

#ifndef RBD_JIT_COMPILE
struct WalkPlan {
  bool ok = false;
  int nS = 0;                // parking slots
  std::vector<int32_t> wk;   // [ns * G]: parking slot + 1 of the body of (step, track), 0 = none
};

inline size_t walk_rows(int nq, int nv, int nA, int nB, int nS) {
  const size_t bc = std::max((size_t)nB * WMB_B, (size_t)nA * WMB_AT);
  return (size_t)nq + 2 * (size_t)nv + (size_t)nA * WMB_A + (size_t)nS * WMB_S + bc;
}
// es: bytes of a row value (one per lane: 8 for fp64 and for the packed pair of fp32 states), ss: bytes of a plan constant
inline size_t walk_lds_bytes(int ns, int G, int nq, int nv, int nA, int nB, int nS, size_t es, size_t ss) {
  const size_t nrec = (size_t)ns * G;
  // plan records | constants | parking words | chain table of a re-rooted tree (RC_MAX joints: 4 ints + 15 constants each) | rows
  return nrec * 16 + nrec * TR_STRIDE * ss + ((nrec * 4 + 15) & ~(size_t)15) + 64 + ((4 * 15 * ss + 15) & ~(size_t)15) + walk_rows(nq, nv, nA, nB, nS) * WR_STRIDE * es;
}

// ri: the packed records of the track plan ([ns * G * TI_STRIDE])
inline WalkPlan build_walk_plan(int ns, int G, const std::vector<int32_t>& ri) {
  WalkPlan P;
  if (ns > WALK_MAX_STEPS || G < 1 || G > 4) return P;
  P.wk.assign((size_t)ns * G, 0);
  for (int g = 0; g < G; ++g) {
    int last = -1;
    for (int s = 0; s < ns; ++s)
      if ((ri[((size_t)s * G + g) * TI_STRIDE + TI_W1] >> 16) & TF_VALID) last = s;
    for (int s = 0; s < ns; ++s) {
      const int flags = (ri[((size_t)s * G + g) * TI_STRIDE + TI_W1] >> 16) & 0xff;
      if (!(flags & TF_VALID) || (flags & TF_CARRY) || s == last) continue;
      P.wk[(size_t)s * G + g] = ++P.nS;
    }
  }
  if (P.nS > 255) return P;
  P.ok = true;
  return P;
}

#endif  // RBD_JIT_COMPILE

}  // namespace rbd
