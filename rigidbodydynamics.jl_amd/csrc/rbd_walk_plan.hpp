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

// ---- the role-pipelined mapping (rbd_pipe.hpp): the same plan on exactly 4 tracks (idle records for the missing ones), one record of
// WREC_STRIDE ints per (step, track) with every field in a word of its own (each lane reads its own track's record from LDS)
enum { WREC_FLAGS = 0, WREC_QOFF, WREC_VOFF, WREC_ORIG6, WREC_NBR, WREC_AW, WREC_AR, WREC_BW, WREC_BR0, WREC_PARK, WREC_RRF, WREC_STRIDE = 12 };
inline std::vector<int32_t> walk_unpack4(int ns, int G, const std::vector<int32_t>& ri, const std::vector<int32_t>& wk) {
  std::vector<int32_t> out((size_t)ns * 4 * WREC_STRIDE, 0);
  for (int s = 0; s < ns; ++s)
    for (int g = 0; g < G && g < 4; ++g) {
      const size_t i = (size_t)s * G + g;
      const int32_t* w = &ri[i * TI_STRIDE];
      int32_t* o = &out[((size_t)s * 4 + g) * WREC_STRIDE];
      const int x = w[0], y = w[1], z = w[2], ww = w[3], kk = wk[i];
      o[WREC_FLAGS] = (y >> 16) & 0xff; o[WREC_QOFF] = x & 0xffff; o[WREC_VOFF] = (x >> 16) & 0xffff; o[WREC_ORIG6] = y & 0xffff; o[WREC_NBR] = (y >> 24) & 0x7f;
      o[WREC_AW] = (z & 0xffff) - 1; o[WREC_AR] = ((z >> 16) & 0xffff) - 1; o[WREC_BW] = (ww & 0xffff) - 1; o[WREC_BR0] = ((ww >> 16) & 0xffff) - 1;
      o[WREC_PARK] = (kk & 0xff) - 1; o[WREC_RRF] = (kk >> 8) & 3;
    }
  return out;
}
// the constants [ns * G][TR_STRIDE] spread to 4 tracks
template <typename S> inline std::vector<S> walk_consts4(int ns, int G, const std::vector<double>& rr) {
  std::vector<S> out((size_t)ns * 4 * TR_STRIDE, S(0));
  for (int s = 0; s < ns; ++s)
    for (int g = 0; g < G && g < 4; ++g)
      for (int k = 0; k < TR_STRIDE; ++k) out[((size_t)s * 4 + g) * TR_STRIDE + k] = (S)rr[((size_t)s * G + g) * TR_STRIDE + k];
  return out;
}
// LDS of a workgroup of 16 states: constants | records | q, v, τ rows | rings (K 3 x 12, I 2 x 10, T 2 x 34, sin/cos 2 x 2 values x 64 lanes) | the motion subspaces of the steps (6 x 64 each) | mailboxes.
// Must agree with pipe_ctx_lds() of rbd_pipe.hpp.
inline size_t pipe_lds_bytes(int ns, int nq, int nv, int nA, int nB, int nS, size_t es) {
  const size_t nrec = (size_t)ns * 4;
  const size_t cells = (size_t)(nq + 2 * nv) * 17 + (3 * 12 + 2 * 10 + 2 * 34 + 2 * 2 + (size_t)ns * 6) * 64 + ((size_t)nA * (12 + 12 + 6) + (size_t)nB * 27 + (size_t)nS * 24) * 16;
  return ((nrec * TR_STRIDE * es + 15) & ~(size_t)15) + ((nrec * WREC_STRIDE * 4 + 15) & ~(size_t)15) + cells * es;
}
#endif  // RBD_JIT_COMPILE

}  // namespace rbd
