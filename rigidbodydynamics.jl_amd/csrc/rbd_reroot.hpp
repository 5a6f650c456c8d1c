// rbd_reroot.hpp — host side: RE-ROOT a floating-base tree at its centre.
//
// The sweeps of the articulated-body kernels cost per tree LEVEL (level-synchronous mappings) or per body of the longest chain (track
// mappings).  A mechanism whose only connection to the world is a 6-dof floating joint can be swept from ANY of its bodies: which body
// "carries" the floating joint is a choice of coordinates, not of physics.  Atlas is 11 levels deep seen from the pelvis (pelvis, three
// torso links, seven arm links) but 9 from its middle torso link — two of the ten steps of every sweep disappear, and the chains of the
// track mappings come out better balanced (arms 7 + 2, legs 6 + 2).
//
// What changes, all of it on the host except the handling of the floating base itself:
//   * the edges between the old floating body c_0 and the new root c_d are REVERSED: c_{k-1} becomes the child of c_k through c_k's own
//     joint J_k (same coordinate q, same velocity, same torque).  The body frame of c_{k-1} is re-based to frame_before(J_k), which is
//     fixed in it and has its origin on the joint axis; then
//         H'_{k-1} = H'_k · E_k⁻¹ · Tj_k(q)⁻¹,   E_k = joint_to_predecessor(J_{k+1}) (identity for k = d),   Tj(q)⁻¹ = Tj(−q)
//     i.e. an ORDINARY joint record with joint_to_predecessor' = E_k⁻¹ and axis' = −axis: the relative twist of c_{k-1} with respect
//     to c_k is −axis·q̇ in its own frame, exactly what the generic kernels compute from (axis', q̇); v̇ and τ keep their meaning.
//     Inertias and the joint_to_predecessor of c_{k-1}'s other children are re-expressed in the new body frame.
//   * the new root keeps its body frame.  Its pose and twist are not coordinates any more: the kernels compute them at the root by walking
//     the chain c_0 → c_d with the ORIGINAL constants (chain table below).  Its acceleration solves IA a = −pA (no joint force).
//   * the old floating body c_0 still owns q_f, v_f, τ_f: τ_f enters as an external wrench S⁻ᵀτ_f on c_0, and v̇_f = S⁻¹(a_{c_0} − a_world)
//     is read off c_0's spatial acceleration after the top-down pass (S = X(H_{c_0}) in its original frame, recomputed from q_f).
// Results are unchanged (tested against the oracle at the reference's 1e-10 like every other mapping); only revolute / prismatic / fixed
// joints may lie on the reversed chain.  Index bookkeeping and constant folding only — no arithmetic of the dynamics happens here.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rbd_device.hpp"
#include "rbd_hip.h"

namespace rbd {

enum { BF_VROOT = BFD_VROOT, BF_FCARRY = BFD_FCARRY };  // per-body flags of a re-rooted tree: the new root (pose via the chain) / the old floating body
enum { RC_I_STRIDE = 4 /* joint type, q offset, v offset, pad */, RC_R_STRIDE = 15 /* axis 3, joint_to_predecessor R 9, p 3 */ };

struct Reroot {
  bool ok = false;
  int nb = 0, root = -1, fb = -1, depth_before = 0, depth_after = 0;
  // the re-rooted tree in REFERENCE body indices (bodies keep their identity; parents are NOT "first" any more)
  std::vector<int32_t> parent, jtype, qoff, voff, flags;
  std::vector<double> axis, axis2, XpR, Xpp, J6, mc, mass;
  // the chain from the old floating body to the new root, original constants: joints J_1 .. J_d in walking order
  std::vector<int32_t> chain_i;
  std::vector<double> chain_r;
  int32_t fq = 0, fv = 0;  // offsets of the floating joint's coordinates
  double fXp[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};  // the floating joint's joint_to_predecessor (R row-major, p)
};

namespace reroot_detail {
inline void mm3(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
inline void mv3(const double* A, const double* x, double* y) {
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
  memcpy(y, t, sizeof t);
}
struct Xf { double R[9], p[3]; };
inline Xf ident() { Xf x{{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; return x; }
inline Xf mul(const Xf& a, const Xf& b) {  // a ∘ b
  Xf c;
  mm3(a.R, b.R, c.R);
  mv3(a.R, b.p, c.p);
  for (int k = 0; k < 3; ++k) c.p[k] += a.p[k];
  return c;
}
inline Xf inv(const Xf& a) {
  Xf c;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * j + i];
  mv3(c.R, a.p, c.p);
  for (int k = 0; k < 3; ++k) c.p[k] = -c.p[k];
  return c;
}
// spatial inertia (J about the origin as xx xy xz yy yz zz, c = m·com, m) under the change of frame t (old frame -> new frame):
// transform(inertia, t), src/spatial/motion_force_interaction.jl:160-176
inline void inertia_xf(const Xf& t, double* J6, double* mc, double m) {
  const double J[9] = {J6[0], J6[1], J6[2], J6[1], J6[3], J6[4], J6[2], J6[4], J6[5]};
  double Rmc[3], mp[3], RJ[9], Rt[9], A[9], Y[9];
  mv3(t.R, mc, Rmc);
  for (int k = 0; k < 3; ++k) mp[k] = m * t.p[k];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Y[3 * i + j] = Rmc[i] * t.p[j] + Rmc[j] * t.p[i] + mp[i] * t.p[j];
  const double trY = Y[0] + Y[4] + Y[8];
  mm3(t.R, J, RJ);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3 * i + j] = t.R[3 * j + i];
  mm3(RJ, Rt, A);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[3 * i + j] += -Y[3 * i + j] + (i == j ? trY : 0.0);
  J6[0] = A[0]; J6[1] = A[1]; J6[2] = A[2]; J6[3] = A[4]; J6[4] = A[5]; J6[5] = A[8];
  for (int k = 0; k < 3; ++k) mc[k] = Rmc[k] + mp[k];
}
}  // namespace reroot_detail

inline Reroot build_reroot(const rbd_flat_model_t* d) {
  using namespace reroot_detail;
  Reroot R;
  const int nb = d->n_bodies;
  R.nb = nb;
  if (d->n_loops > 0 || nb < 3) return R;
  int fb = -1;
  for (int i = 0; i < nb; ++i)
    if (d->parent[i] < 0) { if (fb >= 0) return R; fb = i; }  // exactly one body on the world ...
  if (fb < 0 || d->joint_type[fb] != RBD_JOINT_QUAT_FLOATING) return R;  // ... through a 6-dof floating joint
  // eccentricity of every body in the undirected tree
  std::vector<std::vector<int>> adj(nb);
  for (int i = 0; i < nb; ++i)
    if (d->parent[i] >= 0) { adj[i].push_back(d->parent[i]); adj[d->parent[i]].push_back(i); }
  auto bfs = [&](int s, std::vector<int>& dist) {
    dist.assign(nb, -1);
    std::vector<int> q{s};
    dist[s] = 0;
    for (size_t h = 0; h < q.size(); ++h)
      for (int y : adj[q[h]])
        if (dist[y] < 0) { dist[y] = dist[q[h]] + 1; q.push_back(y); }
    int e = 0;
    for (int x : dist) e = x > e ? x : e;
    return e;
  };
  std::vector<int> dist, dfb;
  R.depth_before = bfs(fb, dfb) + 1;
  auto chain_ok = [&](int r) {  // every joint between fb and r must be reversible
    for (int b = r; b != fb; b = d->parent[b]) {
      const int t = d->joint_type[b];
      if (t != RBD_JOINT_REVOLUTE && t != RBD_JOINT_PRISMATIC && t != RBD_JOINT_FIXED) return false;
    }
    return true;
  };
  int best = fb, best_e = R.depth_before - 1;
  for (int r = 0; r < nb; ++r) {
    if (r == fb || !chain_ok(r)) continue;
    const int e = bfs(r, dist);
    if (e < best_e || (e == best_e && best != fb && dfb[r] < dfb[best])) { best = r; best_e = e; }
  }
  if (best == fb) return R;  // already as shallow as it gets
  R.root = best; R.fb = fb; R.depth_after = best_e + 1;
  std::vector<int> chain;  // c_0 = fb, ..., c_d = root
  for (int b = best; b != fb; b = d->parent[b]) chain.insert(chain.begin(), b);
  chain.insert(chain.begin(), fb);
  const int dlen = (int)chain.size() - 1;
  if (dlen > RC_MAX) return R;  // the kernels keep the chain's coordinates in registers
  // start from the original tree
  R.parent.assign(d->parent, d->parent + nb);
  R.jtype.assign(d->joint_type, d->joint_type + nb);
  R.qoff.assign(d->q_offset, d->q_offset + nb);
  R.voff.assign(d->v_offset, d->v_offset + nb);
  R.flags.assign(nb, 0);
  R.axis.assign(d->joint_axis, d->joint_axis + 3 * nb);
  if (d->joint_axis2) R.axis2.assign(d->joint_axis2, d->joint_axis2 + 3 * nb); else R.axis2.assign(3 * nb, 0.0);
  R.XpR.assign(d->pred_rot, d->pred_rot + 9 * nb);
  R.Xpp.assign(d->pred_trans, d->pred_trans + 3 * nb);
  R.J6.assign(6 * (size_t)nb, 0.0);
  for (int i = 0; i < nb; ++i) {
    const double* J = d->inertia_moment + 9 * i;
    const double j6[6] = {J[0], J[1], J[2], J[4], J[5], J[8]};
    memcpy(&R.J6[6 * i], j6, sizeof j6);
  }
  R.mc.assign(d->inertia_cross, d->inertia_cross + 3 * nb);
  R.mass.assign(d->inertia_mass, d->inertia_mass + nb);
  auto Xp_of = [&](int i) { Xf x; memcpy(x.R, d->pred_rot + 9 * i, sizeof x.R); memcpy(x.p, d->pred_trans + 3 * i, sizeof x.p); return x; };
  // the chain table and the floating joint's own constants (original)
  R.fq = d->q_offset[fb]; R.fv = d->v_offset[fb];
  memcpy(R.fXp, d->pred_rot + 9 * fb, sizeof(double) * 9);
  memcpy(R.fXp + 9, d->pred_trans + 3 * fb, sizeof(double) * 3);
  for (int k = 1; k <= dlen; ++k) {
    const int c = chain[k];
    const int32_t rec[RC_I_STRIDE] = {d->joint_type[c], d->q_offset[c], d->v_offset[c], 0};
    R.chain_i.insert(R.chain_i.end(), rec, rec + RC_I_STRIDE);
    R.chain_r.insert(R.chain_r.end(), d->joint_axis + 3 * c, d->joint_axis + 3 * c + 3);
    R.chain_r.insert(R.chain_r.end(), d->pred_rot + 9 * c, d->pred_rot + 9 * c + 9);
    R.chain_r.insert(R.chain_r.end(), d->pred_trans + 3 * c, d->pred_trans + 3 * c + 3);
  }
  // E[k]: new body frame of c_k -> its old body frame (identity for the new root)
  std::vector<Xf> E(dlen + 1, ident());
  for (int k = 0; k < dlen; ++k) E[k] = Xp_of(chain[k + 1]);
  // reversed edges: c_{k-1} becomes the child of c_k through joint J_k
  for (int k = 1; k <= dlen; ++k) {
    const int child = chain[k - 1], par = chain[k];
    R.parent[child] = par;
    R.jtype[child] = d->joint_type[par];
    R.qoff[child] = d->q_offset[par];
    R.voff[child] = d->v_offset[par];
    for (int j = 0; j < 3; ++j) R.axis[3 * child + j] = -d->joint_axis[3 * par + j];
    const Xf xp = inv(E[k]);  // joint_to_predecessor' : frame_after(J_k) = old frame of c_k -> new frame of c_k
    memcpy(&R.XpR[9 * child], xp.R, sizeof xp.R);
    memcpy(&R.Xpp[3 * child], xp.p, sizeof xp.p);
    inertia_xf(inv(E[k - 1]), &R.J6[6 * child], &R.mc[3 * child], R.mass[child]);
  }
  // the other children of the re-based bodies: their joint_to_predecessor now lands in the new frame of their parent
  for (int i = 0; i < nb; ++i) {
    const int p = d->parent[i];
    if (p < 0) continue;
    for (int k = 0; k < dlen; ++k)
      if (p == chain[k] && i != chain[k + 1]) {
        const Xf xp = mul(inv(E[k]), Xp_of(i));
        memcpy(&R.XpR[9 * i], xp.R, sizeof xp.R);
        memcpy(&R.Xpp[3 * i], xp.p, sizeof xp.p);
      }
  }
  // the new root carries the floating joint's coordinates in name only (its pose comes through the chain)
  R.parent[best] = -1;
  R.jtype[best] = RBD_JOINT_QUAT_FLOATING;
  R.qoff[best] = R.fq; R.voff[best] = R.fv;
  R.flags[best] = BF_VROOT;
  R.flags[fb] |= BF_FCARRY;
  const double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(&R.XpR[9 * best], Id, sizeof Id);
  for (int j = 0; j < 3; ++j) { R.Xpp[3 * best + j] = 0.0; R.axis[3 * best + j] = 0.0; }
  R.ok = true;
  return R;
}

}  // namespace rbd
