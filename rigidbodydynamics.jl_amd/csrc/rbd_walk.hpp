// rbd_walk.hpp — aba_walk_kernel: the fused articulated-body algorithm with ONE WAVEFRONT PER TRACK and ONE LANE PER STATE.
//
// Same result as the reference's dynamics! (src/mechanism_algorithms.jl:845-864: v̇ = M⁻¹(τ − c), q̇) and as the other mappings; all
// quantities in the ROOT frame like the reference (src/mechanism_state.jl:744-748, :776, :842).  Same schedule and records as the
// track mapping (rbd_track_plan.hpp: the tree cut into chains, the chains packed on G <= 4 tracks, canonical body frames with every
// 1-dof joint axis = +z, bias accelerations folded into the bias force) — but a workgroup is G wavefronts over the same 64 states:
//   * wavefront g walks track g; lane l is state l of the workgroup.  Which body, which joint type, which mailboxes: all of it is
//     wave-uniform, so the control flow runs on the scalar unit and every vector instruction does 64 states' worth of useful work
//     (the lane-per-body sweeps keep ~15 % of their lanes busy, the lane-per-track mapping ~70 % of a quarter-filled chip);
//   * NOTHING IS READ FROM OR WRITTEN TO GLOBAL MEMORY INSIDE THE PASSES (measured on the one-lane-per-state RNEA of rbd_state.hpp: a
//     wavefront that owns its SIMD pays every dependent global round trip in full).  q, v, τ of the 64 states are transposed into
//     LDS rows by the whole workgroup with coalesced loads; v̇ is written over the τ rows, q̇ over the q rows, and both leave the same way;
//   * no per-body results of pass A are kept.  Pass A only runs the kinematic chain (transform, twist, velocity-product acceleration);
//     pass B starts at the leaf of a chain with those in registers, computes the body's root-frame inertia and bias force on the spot
//     and then UN-COMPOSES the joint (H_parent = H X_joint⁻¹, T_parent = T − S q̇, a_parent = a − [T, S q̇]) to arrive at the parent;
//     pass C composes the transforms again on the way down to get the motion subspace.  What a body leaves behind is 9 scalars
//     (sin q, cos q, U D⁻¹, D⁻¹u) in ACCUMULATION REGISTERS (a wavefront that owns its SIMD has 256 of them besides the 256 VGPRs;
//     the step index selects them through a scalar `switch` — registers cannot be indexed at run time);
//   * edges between tracks go through LDS mailboxes as in the track mapping, fenced by workgroup barriers at exactly the steps whose
//     wave-uniform flags say a mailbox is written (3–4 per pass for a humanoid).
// LDS per workgroup (fp64 Atlas: 158 KB of the CU's 160 KB; fp32: half): rbd_walk_plan.hpp.
//
// The step functions are __host__ __device__: tests/emu/walk_emu.hip runs the same code lane by lane on the CPU, wavefront by
// wavefront between barriers (in both orders, so that a missing barrier reads a never-written mailbox and fails).
#pragma once
#include "rbd_device.hpp"
#include "rbd_walk_plan.hpp"
#include "rbd_mk_fuse.hpp"

namespace rbd {

struct alignas(16) I4 { int32_t x, y, z, w; };  // one packed plan record (TI_*, rbd_device.hpp)

// `switch (s)` with the step index as a compile-time constant SV inside every case
#define RBD_WALK_CASE(N, ...) case N: { constexpr int SV = N; __VA_ARGS__ } break;
#define RBD_WALK_SWITCH(s, ...)                                                                                                   \
  switch (s) {                                                                                                                    \
    RBD_WALK_CASE(0, __VA_ARGS__) RBD_WALK_CASE(1, __VA_ARGS__) RBD_WALK_CASE(2, __VA_ARGS__) RBD_WALK_CASE(3, __VA_ARGS__)        \
    RBD_WALK_CASE(4, __VA_ARGS__) RBD_WALK_CASE(5, __VA_ARGS__) RBD_WALK_CASE(6, __VA_ARGS__) RBD_WALK_CASE(7, __VA_ARGS__)        \
    RBD_WALK_CASE(8, __VA_ARGS__) RBD_WALK_CASE(9, __VA_ARGS__) RBD_WALK_CASE(10, __VA_ARGS__)                                    \
    default: break;                                                                                                               \
  }

// What a body leaves behind between the passes: 9 scalars per step of the track.
//   0 sin q · 1 cos q · [2, 8) U D⁻¹ (a 6-dof root: its a_Δ) · 8 D⁻¹u
// Device: accumulation registers addressed by immediates, counted DOWN from a255 — the register allocator hands out accumulation registers from
// a0 upwards when it runs out of VGPRs (it parks a few loop-invariant values there in the fp64 kernels), and build.sh checks on the generated
// assembly (scripts/check_walk_agprs.py) that what it takes stays below the WALK_MAX_STEPS steps' worth reserved here.  Host (emulation): an array.
enum { WS_SN = 0, WS_CS = 1, WS_W = 2, WS_UD = 8, WS_N = 9 };
static_assert(WS_N == WALK_STASH_N, "rbd_walk_plan.hpp: WALK_STASH_N");
// (RBD_WALK_STASH_ARRAY: the array on the device too.  Measured on the kernel compiled per mechanism, where every index is a constant: the allocator
// spilled 529 values around the 162 long-lived ones; with the numbered registers it needs 221 VGPRs and no scratch.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RBD_WALK_STASH_ARRAY)
#define RBD_WALK_STASH_AGPR 1
#endif
template <typename T> struct WalkStash {
#ifdef RBD_WALK_STASH_AGPR
  template <int S, int K> RBD_DEV void put(T x);
  template <int S, int K> RBD_DEV T get() const;
#else
  T v[WALK_MAX_STEPS][WS_N];
  template <int S, int K> RBD_HD void put(T x) { v[S][K] = x; }
  template <int S, int K> RBD_HD T get() const { return v[S][K]; }
#endif
};
#ifdef RBD_WALK_STASH_AGPR
template <> template <int S, int K> RBD_DEV void WalkStash<float>::put(float x) { asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(x), "n"(255 - (S * WS_N + K))); }
template <> template <int S, int K> RBD_DEV float WalkStash<float>::get() const {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(255 - (S * WS_N + K)));
  return x;
}
template <> template <int S, int K> RBD_DEV void WalkStash<double>::put(double x) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(__double2loint(x)), "n"(254 - 2 * (S * WS_N + K)));
  asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(__double2hiint(x)), "n"(255 - 2 * (S * WS_N + K)));
}
template <> template <int S, int K> RBD_DEV void WalkStash<f2>::put(f2 x) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(x.x), "n"(254 - 2 * (S * WS_N + K)));
  asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(x.y), "n"(255 - 2 * (S * WS_N + K)));
}
template <> template <int S, int K> RBD_DEV f2 WalkStash<f2>::get() const {
  f2 x;
  float lo, hi;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(lo) : "n"(254 - 2 * (S * WS_N + K)));
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(hi) : "n"(255 - 2 * (S * WS_N + K)));
  x.x = lo; x.y = hi;
  return x;
}
template <> template <int S, int K> RBD_DEV double WalkStash<double>::get() const {
  int lo, hi;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(lo) : "n"(254 - 2 * (S * WS_N + K)));
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(hi) : "n"(255 - 2 * (S * WS_N + K)));
  return __hiloint2double(hi, lo);
}
#endif

RBD_HD int walk_uniform(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(x);
#else
  return x;
#endif
}

template <typename T> struct WalkCtx {
  WalkModel M;
  // LDS (rows are [field][WR_STRIDE], a lane's value of a field at row[lane])
  const I4* tri; const int32_t* twk; const typename Lanes<T>::S* trr;  // the constants are scalars: the states of a lane share them
  T* rows;
  int rq, rv, rt, rA, rS, rB;  // first row of q | v | τ / v̇ | A mailboxes | parking slots | B mailboxes (pass C: its mailboxes)
  int rT, sT;                  // twist halves of the A mailboxes: first row, rows from one to the next (walk_twist_rows)
  RerootView rrv;              // the re-rooted tree's floating base (M.reroot with the chain table where this code can read it fast)
  T a0[6];
};

template <typename T> RBD_HD void walk_ctx_lds(WalkCtx<T>& c, void* lds) {
  const size_t nrec = (size_t)c.M.ns * c.M.G;
  c.tri = reinterpret_cast<const I4*>(lds);
  using S = typename Lanes<T>::S;
  c.trr = reinterpret_cast<const S*>(c.tri + nrec);
  c.twk = reinterpret_cast<const int32_t*>(c.trr + nrec * TR_STRIDE);
  char* after = reinterpret_cast<char*>(lds) + nrec * 16 + nrec * TR_STRIDE * sizeof(S) + ((nrec * 4 + 15) & ~(size_t)15);
  c.rrv = c.M.reroot;  // the chain table of a re-rooted tree is copied behind the parking words by the prologue (walk_stage_chain)
  c.rrv.chain_i = reinterpret_cast<const int32_t*>(after);
  c.rrv.chain_r = after + 64;
  c.rows = reinterpret_cast<T*>(after + 64 + ((RC_MAX * 15 * sizeof(S) + 15) & ~(size_t)15));
  c.rq = 0; c.rv = c.M.nq; c.rt = c.rv + c.M.nv; c.rA = c.rt + c.M.nv; c.rS = c.rA + c.M.nA * WMB_A; c.rB = c.rS + c.M.nS * WMB_S;
  c.rT = c.rB; c.sT = WMB_AT;
#pragma unroll
  for (int k = 0; k < 3; ++k) { c.a0[k] = T(0); c.a0[3 + k] = T(-c.M.gravity[k]); }  // a_world = −gravity (mechanism_algorithms.jl:405)
}
template <typename T> RBD_HD void walk_stage_chain(const WalkCtx<T>& c, int tid, int nth) {
  using S = typename Lanes<T>::S;
  const int n = c.M.reroot.nchain;
  int32_t* di = const_cast<int32_t*>(c.rrv.chain_i);
  S* dr = reinterpret_cast<S*>(const_cast<void*>(c.rrv.chain_r));
  const S* sr = reinterpret_cast<const S*>(c.M.reroot.chain_r);
  for (int i = tid; i < n * 4; i += nth) di[i] = c.M.reroot.chain_i[i];
  for (int i = tid; i < n * 15; i += nth) dr[i] = sr[i];
}
template <typename T> RBD_HD T* walk_row(const WalkCtx<T>& c, int row, int lane) { return c.rows + (long)row * WR_STRIDE + lane; }
// the scalar of state st (0 <= st < 64 N) of the workgroup in a row: lane st mod 64, component st div 64
template <typename T> RBD_HD typename Lanes<T>::S* walk_cell(T* rows, int row, int st) {
  return reinterpret_cast<typename Lanes<T>::S*>(rows + (long)row * WR_STRIDE + (st & 63)) + (st >> 6);
}

// per-lane recursion state of a track
template <typename T> struct WalkRegs {
  T R[9], p[3], Tw[6], av[6];  // kinematics of the body last visited: transform to root, twist, velocity-product acceleration
  T cI[21], cP[6];             // pass B: hand-off of the chained child (zero at the leaf of a chain)
  T ad[6];                     // pass C: a_Δ of the body last visited
};
template <typename T> RBD_HD void walk_init(WalkRegs<T>& W) {
#pragma unroll
  for (int k = 0; k < 9; ++k) W.R[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) W.p[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) { W.Tw[k] = T(0); W.av[k] = T(0); }
}
// the hand-off registers start their life with pass B, a_Δ with pass C (kept out of the earlier passes' register pressure)
template <typename T> RBD_HD void walk_init_b(WalkRegs<T>& W) {
#pragma unroll
  for (int k = 0; k < 6; ++k) W.cP[k] = T(0);
#pragma unroll
  for (int k = 0; k < 21; ++k) W.cI[k] = T(0);
}
template <typename T> RBD_HD void walk_init_c(WalkRegs<T>& W) {
#pragma unroll
  for (int k = 0; k < 6; ++k) W.ad[k] = T(0);
}

struct WalkRec { int flags, qoff, voff, orig6, nbr, a_w, a_r, b_w, b_r0, park, rrf; };  // rrf: BFD_* of a re-rooted tree
// the raw words of a record: read from LDS a step ahead (walk_raw), made wave-uniform scalars when the step starts (walk_rec)
struct WalkRaw { I4 w; int32_t k; };
template <typename T> RBD_HD WalkRaw walk_raw(const WalkCtx<T>& c, int s, int g) {
  WalkRaw x;
  x.w = c.tri[s * c.M.G + g];
  x.k = c.twk[s * c.M.G + g];
  return x;
}
RBD_HD WalkRec walk_rec(const WalkRaw& raw) {
  WalkRec r;
  const int x = walk_uniform(raw.w.x), y = walk_uniform(raw.w.y), z = walk_uniform(raw.w.z), ww = walk_uniform(raw.w.w);
  r.flags = (y >> 16) & 0xff; r.qoff = x & 0xffff; r.voff = (x >> 16) & 0xffff; r.orig6 = y & 0xffff; r.nbr = (y >> 24) & 0x7f;
  r.a_w = (z & 0xffff) - 1; r.a_r = ((z >> 16) & 0xffff) - 1; r.b_w = (ww & 0xffff) - 1; r.b_r0 = ((ww >> 16) & 0xffff) - 1;
  const int kk = walk_uniform(raw.k);
  r.park = (kk & 0xff) - 1;
  r.rrf = (kk >> 8) & 3;
  if (r.flags & TF_FIXED) { r.qoff = 0; r.voff = 0; }  // a fixed joint has no coordinates: its offsets may be one past the end
  return r;
}
// the constants of a record, LDS -> registers
template <typename T, int N> RBD_HD void walk_consts(const WalkCtx<T>& c, int s, int g, T* rr) {  // the first N: (C, pp) = 12 is all passes A and C need
  const typename Lanes<T>::S* src = c.trr + (long)(s * c.M.G + g) * TR_STRIDE;
#pragma unroll
  for (int k = 0; k < N; ++k) rr[k] = T(src[k]);
}

// m: transform (9 + 3 rows), mt: twist and velocity-product acceleration (6 + 6 rows)
template <typename T> RBD_HD void walk_put_kin(T* m, T* mt, const WalkRegs<T>& W) {
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k * WR_STRIDE] = W.R[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) m[(9 + k) * WR_STRIDE] = W.p[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) { mt[k * WR_STRIDE] = W.Tw[k]; mt[(6 + k) * WR_STRIDE] = W.av[k]; }
}
template <typename T> RBD_HD void walk_get_kin(const T* m, const T* mt, WalkRegs<T>& W) {
#pragma unroll
  for (int k = 0; k < 9; ++k) W.R[k] = m[k * WR_STRIDE];
#pragma unroll
  for (int k = 0; k < 3; ++k) W.p[k] = m[(9 + k) * WR_STRIDE];
#pragma unroll
  for (int k = 0; k < 6; ++k) { W.Tw[k] = mt[k * WR_STRIDE]; W.av[k] = mt[(6 + k) * WR_STRIDE]; }
}
// An A mailbox: the transform half lives until pass C has read it; the twist half is dead when pass A ends and sits in the rows the B
// mailboxes take over (aba_walk_kernel puts a workgroup barrier between the two passes).  A parking slot is the two halves back to back.
// The inverse-dynamics kernel hands 6 values up per B mailbox, not 27: there the twist half of A mailbox a lives in the unused rows of B
// mailbox a (walk_twist_rows) and no barrier is needed — unless there are more A than B mailboxes (a 6-dof root without cross children).
template <typename T> RBD_HD bool walk_twist_rows_rnea(WalkCtx<T>& c) {
  if (c.M.nA > c.M.nB) return false;  // keep the shared rows and the barrier
  c.rT = c.rB + 6; c.sT = WMB_B;
  return true;
}
template <typename T> RBD_HD void walk_put_box(const WalkCtx<T>& c, int a, int lane, const WalkRegs<T>& W) {
  walk_put_kin(walk_row(c, c.rA + a * WMB_A, lane), walk_row(c, c.rT + a * c.sT, lane), W);
}
template <typename T> RBD_HD void walk_get_box(const WalkCtx<T>& c, int a, int lane, WalkRegs<T>& W) {
  walk_get_kin(walk_row(c, c.rA + a * WMB_A, lane), walk_row(c, c.rT + a * c.sT, lane), W);
}
template <typename T> RBD_HD void walk_put_park(const WalkCtx<T>& c, int k, int lane, const WalkRegs<T>& W) {
  T* m = walk_row(c, c.rS + k * WMB_S, lane);
  walk_put_kin(m, m + 12 * WR_STRIDE, W);
}
template <typename T> RBD_HD void walk_get_park(const WalkCtx<T>& c, int k, int lane, WalkRegs<T>& W) {
  const T* m = walk_row(c, c.rS + k * WMB_S, lane);
  walk_get_kin(m, m + 12 * WR_STRIDE, W);
}

// 1-dof joint in the canonical frame: (Rn, pn) = (R, p) * (C, pp) * joint(q), joint = rotation by (sn, cs) about z, or translation d along z
template <typename T, bool GEN>
RBD_HD void walk_compose(const T* R, const T* p, const T* rr, int flags, T sn, T cs, T d, T* Rn, T* pn) {
  T Mx[9], u3[3];
  matmul3(R, rr + TR_C, Mx);
  matvec3(R, rr + TR_PP, u3);
#pragma unroll
  for (int k = 0; k < 3; ++k) pn[k] = p[k] + u3[k];
  if (GEN) {
    if (flags & TF_PRISMATIC) {
#pragma unroll
      for (int k = 0; k < 3; ++k) pn[k] += Mx[3 * k + 2] * d;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {  // revolute.jl:59-62 in the canonical frame: columns 0 and 1 are mixed
    Rn[3 * i] = cs * Mx[3 * i] + sn * Mx[3 * i + 1];
    Rn[3 * i + 1] = cs * Mx[3 * i + 1] - sn * Mx[3 * i];
    Rn[3 * i + 2] = Mx[3 * i + 2];
  }
}
// root-frame motion subspace of a 1-dof joint from its body's transform (axis = +z column of R)
template <typename T, bool GEN> RBD_HD void walk_subspace(const T* R, const T* p, int flags, T* S) {
  const T z[3] = {R[2], R[5], R[8]};
  if (GEN && (flags & TF_PRISMATIC)) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { S[k] = T(0); S[3 + k] = z[k]; }
  } else if (GEN && (flags & TF_FIXED)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) S[k] = T(0);
  } else {
    cross3(p, z, S + 3);
#pragma unroll
    for (int k = 0; k < 3; ++k) S[k] = z[k];
  }
}
// 6-dof joint on a parent (quaternion_floating.jl:81-83): H = H_parent * (C, pp) * (R(quat), trans)
template <typename T> RBD_HD void walk_compose_floating(const T* R, const T* p, const T* rr, const T* q7, T* Rn, T* pn) {
  T Rq[9], CR[9], t3[3], u3[3];
  rot_quat(q7[0], q7[1], q7[2], q7[3], Rq);
  matmul3(rr + TR_C, Rq, CR);
  matmul3(R, CR, Rn);
  matvec3(rr + TR_C, q7 + 4, t3);
#pragma unroll
  for (int k = 0; k < 3; ++k) t3[k] += rr[TR_PP + k];
  matvec3(R, t3, u3);
#pragma unroll
  for (int k = 0; k < 3; ++k) pn[k] = p[k] + u3[k];
}

// the parent's kinematics: nothing to do when chained (the registers hold them); the world; or the parent's A mailbox
template <typename T> RBD_HD void walk_parent_kin(const WalkCtx<T>& c, const WalkRec& r, int lane, WalkRegs<T>& W) {
  if (r.flags & TF_CHAINED) return;
  if (r.flags & TF_LEVEL0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) W.R[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) W.p[k] = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) { W.Tw[k] = T(0); W.av[k] = c.a0[k]; }
  } else {
    walk_get_box(c, r.a_r, lane, W);
  }
}

// ---------------- pass A (root -> leaves): the kinematic chain ----------------
// RNEA = true (rnea_walk_kernel): the τ rows hold v̇ on the way in, and the acceleration carried down is the full spatial acceleration
// a_b = a_parent + [T_parent, S q̇] + S v̇ (spatial_accelerations!, mechanism_algorithms.jl:387-417) instead of its velocity-product part
template <typename T, bool FLT, bool GEN, bool RNEA = false, bool RR = false>
RBD_HD void walk_step_a(const WalkCtx<T>& c, WalkRegs<T>& W, WalkStash<T>& St, int s, const WalkRec& r, const T* rr, int lane, bool want_qdot) {
  if (!(r.flags & TF_VALID)) return;
  walk_parent_kin(c, r, lane, W);
  T Rn[9], pn[3], vJ[6], cb[6], aJ[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) aJ[k] = T(0);
  bool done = false;
  if (FLT) {
    if (RR && (r.flags & TF_FLOATING) && (r.rrf & BFD_VROOT)) {
      // the root of a tree re-rooted at its centre (rbd_reroot.hpp): pose and twist through the chain of original joints from the old
      // floating body's coordinates; a_vp = a_world.  q̇ of the floating joint is written by the old floating body in pass C.
      using SS = typename Lanes<T>::S;
      T q7[7], v6[6], cq[RC_MAX], cv[RC_MAX];
#pragma unroll
      for (int k = 0; k < 7; ++k) q7[k] = *walk_row(c, c.rq + c.rrv.fq + k, lane);
#pragma unroll
      for (int k = 0; k < 6; ++k) v6[k] = *walk_row(c, c.rv + c.rrv.fv + k, lane);
#pragma unroll
      for (int j = 0; j < RC_MAX; ++j) {
        const bool on = j < c.rrv.nchain;
        cq[j] = on ? *walk_row(c, c.rq + c.rrv.chain_i[4 * j + 1], lane) : T(0);
        cv[j] = on ? *walk_row(c, c.rv + c.rrv.chain_i[4 * j + 2], lane) : T(0);
      }
      reroot_root_kinematics<T, SS>(c.rrv, q7, v6, cq, cv, W.R, W.p, W.Tw);
#pragma unroll
      for (int k = 0; k < 6; ++k) W.av[k] = c.a0[k];
      if (r.a_w >= 0) walk_put_box(c, r.a_w, lane, W);
      if (r.park >= 0) walk_put_park(c, r.park, lane, W);
      return;
    }
    if (r.flags & TF_FLOATING) {
      T q7[7], v6[6];
#pragma unroll
      for (int k = 0; k < 7; ++k) q7[k] = *walk_row(c, c.rq + r.qoff + k, lane);
#pragma unroll
      for (int k = 0; k < 6; ++k) v6[k] = *walk_row(c, c.rv + r.voff + k, lane);
      walk_compose_floating(W.R, W.p, rr, q7, Rn, pn);
      xmotion(Rn, pn, v6, vJ);  // twist of the joint: X(H) v, v the body-frame twist
      if (RNEA) {
        T a6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) a6[k] = *walk_row(c, c.rt + r.voff + k, lane);
        xmotion(Rn, pn, a6, aJ);
      }
      if (want_qdot) {  // velocity_to_configuration_derivative! (quaternion_floating.jl:126-136, spatial/util.jl:127-134), over the q rows
        const T qw = q7[0], qx = q7[1], qy = q7[2], qz = q7[3];
        T o[7], Rq[9];
        o[0] = (-qx * v6[0] - qy * v6[1] - qz * v6[2]) / 2;
        o[1] = (qw * v6[0] - qz * v6[1] + qy * v6[2]) / 2;
        o[2] = (qz * v6[0] + qw * v6[1] - qx * v6[2]) / 2;
        o[3] = (-qy * v6[0] + qx * v6[1] + qw * v6[2]) / 2;
        rot_quat(qw, qx, qy, qz, Rq);
        matvec3(Rq, v6 + 3, o + 4);
#pragma unroll
        for (int k = 0; k < 7; ++k) *walk_row(c, c.rq + r.qoff + k, lane) = o[k];
      }
      done = true;
    }
  }
  if (!done) {
    const T q0 = *walk_row(c, c.rq + r.qoff, lane);
    const T qd = (GEN && (r.flags & TF_FIXED)) ? T(0) : *walk_row(c, c.rv + r.voff, lane);
    T sn, cs;  // what the later passes need of q: (sin, cos) of a rotation — or, in the sin slot, the displacement of a prismatic joint
    if (GEN && (r.flags & TF_SINCOS)) {  // sin_cos_revolute.jl:69-96: q = (sin θ, cos θ); d/dt (sin θ, cos θ) = (cos θ, −sin θ) θ̇
      sn = q0; cs = *walk_row(c, c.rq + r.qoff + 1, lane);
      if (want_qdot) { *walk_row(c, c.rq + r.qoff, lane) = cs * qd; *walk_row(c, c.rq + r.qoff + 1, lane) = -sn * qd; }
    } else {
      if (GEN && (r.flags & (TF_PRISMATIC | TF_FIXED))) { sn = T(0); cs = T(1); }
      else sincos_fast(q0, &sn, &cs);
      if (want_qdot && !(GEN && (r.flags & TF_FIXED))) *walk_row(c, c.rq + r.qoff, lane) = qd;
    }
    walk_compose<T, GEN>(W.R, W.p, rr, r.flags, sn, cs, q0, Rn, pn);
    T S[6];
    walk_subspace<T, GEN>(Rn, pn, r.flags, S);
#pragma unroll
    for (int k = 0; k < 6; ++k) vJ[k] = S[k] * qd;
    if (RNEA) {
      const T vd = (GEN && (r.flags & TF_FIXED)) ? T(0) : *walk_row(c, c.rt + r.voff, lane);
#pragma unroll
      for (int k = 0; k < 6; ++k) aJ[k] = S[k] * vd;
    }
    if (GEN && (r.flags & TF_PRISMATIC)) sn = q0;
    RBD_WALK_SWITCH(s, { St.template put<SV, WS_SN>(sn); St.template put<SV, WS_CS>(cs); })
  }
  se3_comm(W.Tw, vJ, cb);  // [T_parent, vJ] = [T_b, vJ]: bias acceleration increment (mechanism_state.jl:814-830)
#pragma unroll
  for (int k = 0; k < 6; ++k) { W.av[k] += cb[k] + aJ[k]; W.Tw[k] += vJ[k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) W.R[k] = Rn[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) W.p[k] = pn[k];
  if (r.a_w >= 0) walk_put_box(c, r.a_w, lane, W);  // some child is not next on this track
  if (r.park >= 0) walk_put_park(c, r.park, lane, W);  // pass B will not arrive here from a chained child
}

// Pass B, re-rooted tree, the OLD floating body (before its ordinary step, and before that step's constants are fetched — few values are
// live here, and the register allocator must stay clear of WalkStash's accumulation registers): the floating joint's force is one more
// external wrench S⁻ᵀτ_f on the body (S = X(H) of its original frame, from q_f); the rows of τ_f then take the part of v̇_f this pass knows,
// a_vp − a_world in the root frame (pass C adds a_Δ).  fx: the body's external wrench, updated in place.
template <typename T> RBD_HD void walk_fcarry_b(const WalkCtx<T>& c, const WalkRegs<T>& W, int lane, T* fx) {
  T q7[7], t6[6], Rf[9], pf[3], wf[6];
#pragma unroll
  for (int k = 0; k < 7; ++k) q7[k] = *walk_row(c, c.rq + c.rrv.fq + k, lane);
#pragma unroll
  for (int k = 0; k < 6; ++k) t6[k] = *walk_row(c, c.rt + c.rrv.fv + k, lane);
  reroot_fb_pose(c.rrv, q7, Rf, pf);
  xforce(Rf, pf, t6, wf);
#pragma unroll
  for (int k = 0; k < 6; ++k) { fx[k] += wf[k]; *walk_row(c, c.rt + c.rrv.fv + k, lane) = W.av[k] - c.a0[k]; }
}

// ---------------- pass B (leaves -> root): articulated-body inertias and bias forces ----------------
// fe: this body's external wrench (zero without); the caller has it in registers before the step (prefetched a step ahead)
template <typename T, bool FLT, bool GEN, bool RR = false>
RBD_HD void walk_step_b(const WalkCtx<T>& c, WalkRegs<T>& W, WalkStash<T>& St, int s, const WalkRec& r, const T* rr, int lane, const T* fe) {
  if (!(r.flags & TF_VALID)) return;
  if (r.park >= 0) walk_get_park(c, r.park, lane, W);
  // spatial inertia in the root frame (mechanism_state.jl:836-846), p̃A = I a_vp + T ×* I T − w_ext (newton_euler, :872-876)
  T IA[21], pA[6];
  {
    RInertia<T> I;
    inertia_to_root(rr + TR_J, rr + TR_MC, rr[TR_M], W.R, W.p, I);
    T h[6];
    mul_inertia(I, W.av, pA);
    momentum_cross(I, W.Tw, h);
#pragma unroll
    for (int k = 0; k < 6; ++k) pA[k] += h[k] - fe[k] + W.cP[k];
    sym6_from_inertia(I, IA);
  }
#pragma unroll
  for (int k = 0; k < 21; ++k) IA[k] += W.cI[k];
  for (int j = 0; j < r.nbr; ++j) {  // hand-offs of the children that finished on other tracks (or earlier on this one)
    const T* m = walk_row(c, c.rB + (r.b_r0 + j) * WMB_B, lane);
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] += m[k * WR_STRIDE];
#pragma unroll
    for (int k = 0; k < 6; ++k) pA[k] += m[(21 + k) * WR_STRIDE];
  }
  bool done = false;
  if (FLT) {
    if (r.flags & TF_FLOATING) {
      // 6-dof joint on the world: IA a_Δ = S⁻ᵀτ − p̃A, v̇ = S⁻¹ a_Δ  (S = X(H): the body-frame twist basis seen from the root)
      // (the virtual root of a re-rooted tree: no joint force, and v̇ of the floating joint is read off the old floating body instead)
      T t6[6], f6[6], a[6], vd[6];
      const bool vroot = RR && (r.rrf & BFD_VROOT);
#pragma unroll
      for (int k = 0; k < 6; ++k) { t6[k] = vroot ? T(0) : *walk_row(c, c.rt + r.voff + k, lane); f6[k] = T(0); }
      if (!vroot) xforce(W.R, W.p, t6, f6);
#pragma unroll
      for (int k = 0; k < 6; ++k) f6[k] -= pA[k];
      sym6_solve(IA, f6, a);
      xmotion_inv(W.R, W.p, a, vd);
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (!vroot) *walk_row(c, c.rt + r.voff + k, lane) = vd[k];
      RBD_WALK_SWITCH(s, {
        St.template put<SV, WS_W + 0>(a[0]); St.template put<SV, WS_W + 1>(a[1]); St.template put<SV, WS_W + 2>(a[2]);
        St.template put<SV, WS_W + 3>(a[3]); St.template put<SV, WS_W + 4>(a[4]); St.template put<SV, WS_W + 5>(a[5]);
      })
      done = true;
    }
  }
  T S[6], qd = T(0);
  if (!done) {
    walk_subspace<T, GEN>(W.R, W.p, r.flags, S);
    T U[6], Wd[6];
    sym6_mul(IA, S, U);
    T Dinv = rcp_hd(dot6(S, U));
    T tq = *walk_row(c, c.rt + r.voff, lane);
    qd = *walk_row(c, c.rv + r.voff, lane);
    if (GEN) {
      if (r.flags & TF_FIXED) { Dinv = T(0); tq = T(0); qd = T(0); }  // a fixed joint has S = 0: the body hands its whole inertia up
    }
    const T ud = (tq - dot6(S, pA)) * Dinv;
#pragma unroll
    for (int k = 0; k < 6; ++k) Wd[k] = U[k] * Dinv;
    RBD_WALK_SWITCH(s, {
      St.template put<SV, WS_W + 0>(Wd[0]); St.template put<SV, WS_W + 1>(Wd[1]); St.template put<SV, WS_W + 2>(Wd[2]);
      St.template put<SV, WS_W + 3>(Wd[3]); St.template put<SV, WS_W + 4>(Wd[4]); St.template put<SV, WS_W + 5>(Wd[5]);
      St.template put<SV, WS_UD>(ud);
    })
    // hand-off: Ia = IA − U D⁻¹ U', p̃a = p̃A + U D⁻¹ u
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) W.cI[SI(i, j)] = IA[SI(i, j)] - Wd[i] * U[j];
#pragma unroll
    for (int k = 0; k < 6; ++k) W.cP[k] = pA[k] + U[k] * ud;
  }
  if (r.b_w >= 0 || (r.flags & TF_LEVEL0)) {  // the parent is not next on this track (or is the world): the hand-off leaves the registers
    if (r.b_w >= 0) {
      T* m = walk_row(c, c.rB + r.b_w * WMB_B, lane);
#pragma unroll
      for (int k = 0; k < 21; ++k) m[k * WR_STRIDE] = W.cI[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) m[(21 + k) * WR_STRIDE] = W.cP[k];
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) W.cI[k] = T(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) W.cP[k] = T(0);
  }
  if (r.flags & TF_CHAINED) {
    // un-compose the joint: the next step of this track is the parent.  T_parent = T − S q̇, a_parent = a − [T, S q̇],
    // R_parent = (R rot_z(q)⁻¹) C', p_parent = p − d z − R_parent pp
    T vJ[6], cb[6], sn = T(0), cs = T(1);
#pragma unroll
    for (int k = 0; k < 6; ++k) vJ[k] = S[k] * qd;
    se3_comm(W.Tw, vJ, cb);
#pragma unroll
    for (int k = 0; k < 6; ++k) { W.av[k] -= cb[k]; W.Tw[k] -= vJ[k]; }
    RBD_WALK_SWITCH(s, { sn = St.template get<SV, WS_SN>(); cs = St.template get<SV, WS_CS>(); })
    if (GEN) {
      if (r.flags & TF_PRISMATIC) {  // the sin slot holds the displacement
        const T d = sn;
        sn = T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) W.p[k] -= W.R[3 * k + 2] * d;
      }
    }
    T Mx[9], u3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Mx[3 * i] = cs * W.R[3 * i] - sn * W.R[3 * i + 1];
      Mx[3 * i + 1] = sn * W.R[3 * i] + cs * W.R[3 * i + 1];
      Mx[3 * i + 2] = W.R[3 * i + 2];
    }
    const T* C = rr + TR_C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) W.R[3 * i + j] = Mx[3 * i] * C[3 * j] + Mx[3 * i + 1] * C[3 * j + 1] + Mx[3 * i + 2] * C[3 * j + 2];  // Mx C'
    matvec3(W.R, rr + TR_PP, u3);
#pragma unroll
    for (int k = 0; k < 3; ++k) W.p[k] -= u3[k];
  }
}

// ---------------- inverse dynamics, pass B (leaves -> root): net wrenches and joint torques (rnea_walk_kernel) ----------------
// f_b = I a + T ×* I T − w_ext (newton_euler!, mechanism_algorithms.jl:428-439) + Σ_children f; τ = S' f (joint_wrenches_and_torques! :442-459).
// The hand-off is the 6-value wrench (W.cP; mailboxes use the first 6 rows of a B slot); the τ rows hold v̇ until this body overwrites its own.
template <typename T, bool FLT, bool GEN>
RBD_HD void walk_step_rb(const WalkCtx<T>& c, WalkRegs<T>& W, WalkStash<T>& St, int s, const WalkRec& r, const T* rr, int lane, const T* fe,
                         T* acc_o = nullptr, T* jw_o = nullptr /* per-body outputs, 6 each: spatial acceleration, joint wrench (root frame) */) {
  if (!(r.flags & TF_VALID)) return;
  if (r.park >= 0) walk_get_park(c, r.park, lane, W);
  if (acc_o) {
#pragma unroll
    for (int k = 0; k < 6; ++k) acc_o[k] = W.av[k];
  }
  T f[6];
  {
    RInertia<T> I;
    inertia_to_root(rr + TR_J, rr + TR_MC, rr[TR_M], W.R, W.p, I);
    T h[6];
    mul_inertia(I, W.av, f);
    momentum_cross(I, W.Tw, h);
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] += h[k] - fe[k] + W.cP[k];
  }
  for (int j = 0; j < r.nbr; ++j) {
    const T* m = walk_row(c, c.rB + (r.b_r0 + j) * WMB_B, lane);
#pragma unroll
    for (int k = 0; k < 6; ++k) f[k] += m[k * WR_STRIDE];
  }
  T S[6], qd = T(0), vd = T(0);
  if (FLT && (r.flags & TF_FLOATING)) {  // τ = S' f with S = X(H): the wrench seen from the body frame
    T o6[6];
    xforce_inv(W.R, W.p, f, o6);
#pragma unroll
    for (int k = 0; k < 6; ++k) *walk_row(c, c.rt + r.voff + k, lane) = o6[k];
  } else {
    walk_subspace<T, GEN>(W.R, W.p, r.flags, S);
    if (!(GEN && (r.flags & TF_FIXED))) {
      qd = *walk_row(c, c.rv + r.voff, lane);
      vd = *walk_row(c, c.rt + r.voff, lane);
      *walk_row(c, c.rt + r.voff, lane) = dot6(S, f);
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) W.cP[k] = f[k];
  if (jw_o) {
#pragma unroll
    for (int k = 0; k < 6; ++k) jw_o[k] = f[k];
  }
  if (r.b_w >= 0 || (r.flags & TF_LEVEL0)) {
    if (r.b_w >= 0) {
      T* m = walk_row(c, c.rB + r.b_w * WMB_B, lane);
#pragma unroll
      for (int k = 0; k < 6; ++k) m[k * WR_STRIDE] = f[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) W.cP[k] = T(0);
  }
  if (r.flags & TF_CHAINED) {  // un-compose the joint (see walk_step_b); the acceleration also gives back S v̇
    T vJ[6], cb[6], sn = T(0), cs = T(1);
#pragma unroll
    for (int k = 0; k < 6; ++k) vJ[k] = S[k] * qd;
    se3_comm(W.Tw, vJ, cb);
#pragma unroll
    for (int k = 0; k < 6; ++k) { W.av[k] -= cb[k] + S[k] * vd; W.Tw[k] -= vJ[k]; }
    RBD_WALK_SWITCH(s, { sn = St.template get<SV, WS_SN>(); cs = St.template get<SV, WS_CS>(); })
    if (GEN) {
      if (r.flags & TF_PRISMATIC) {
        const T d = sn;
        sn = T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) W.p[k] -= W.R[3 * k + 2] * d;
      }
    }
    T Mx[9], u3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Mx[3 * i] = cs * W.R[3 * i] - sn * W.R[3 * i + 1];
      Mx[3 * i + 1] = sn * W.R[3 * i] + cs * W.R[3 * i + 1];
      Mx[3 * i + 2] = W.R[3 * i + 2];
    }
    const T* C = rr + TR_C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) W.R[3 * i + j] = Mx[3 * i] * C[3 * j] + Mx[3 * i + 1] * C[3 * j + 1] + Mx[3 * i + 2] * C[3 * j + 2];
    matvec3(W.R, rr + TR_PP, u3);
#pragma unroll
    for (int k = 0; k < 3; ++k) W.p[k] -= u3[k];
  }
}

// ---------------- pass C (root -> leaves): v̇ and a_Δ ----------------
template <typename T, bool FLT, bool GEN, bool RR = false>
RBD_HD void walk_step_c(const WalkCtx<T>& c, WalkRegs<T>& W, WalkStash<T>& St, int s, const WalkRec& r, const T* rr, int lane, bool want_qdot = false) {
  if (!(r.flags & TF_VALID)) return;
  if (!(r.flags & TF_CHAINED)) {
    if (r.flags & TF_LEVEL0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) W.R[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) W.p[k] = T(0);
#pragma unroll
      for (int k = 0; k < 6; ++k) W.ad[k] = T(0);
    } else {
      const T* m = walk_row(c, c.rA + r.a_r * WMB_A, lane);  // the parent's transform is still in its A mailbox
#pragma unroll
      for (int k = 0; k < 9; ++k) W.R[k] = m[k * WR_STRIDE];
#pragma unroll
      for (int k = 0; k < 3; ++k) W.p[k] = m[(9 + k) * WR_STRIDE];
      const T* mc = walk_row(c, c.rB + r.a_r * WMB_C, lane);
#pragma unroll
      for (int k = 0; k < 6; ++k) W.ad[k] = mc[k * WR_STRIDE];
    }
  }
  bool done = false;
  if (FLT) {
    if (r.flags & TF_FLOATING) {  // a_Δ was solved for in pass B (v̇ is already in its rows); the body's transform is in its own A mailbox
      RBD_WALK_SWITCH(s, {
        W.ad[0] = St.template get<SV, WS_W + 0>(); W.ad[1] = St.template get<SV, WS_W + 1>(); W.ad[2] = St.template get<SV, WS_W + 2>();
        W.ad[3] = St.template get<SV, WS_W + 3>(); W.ad[4] = St.template get<SV, WS_W + 4>(); W.ad[5] = St.template get<SV, WS_W + 5>();
      })
      const T* m = walk_row(c, c.rA + r.a_w * WMB_A, lane);
#pragma unroll
      for (int k = 0; k < 9; ++k) W.R[k] = m[k * WR_STRIDE];
#pragma unroll
      for (int k = 0; k < 3; ++k) W.p[k] = m[(9 + k) * WR_STRIDE];
      done = true;
    }
  }
  if (!done) {
    T sn = T(0), cs = T(1), Wd[6], ud = T(0);
    RBD_WALK_SWITCH(s, {
      sn = St.template get<SV, WS_SN>(); cs = St.template get<SV, WS_CS>(); ud = St.template get<SV, WS_UD>();
      Wd[0] = St.template get<SV, WS_W + 0>(); Wd[1] = St.template get<SV, WS_W + 1>(); Wd[2] = St.template get<SV, WS_W + 2>();
      Wd[3] = St.template get<SV, WS_W + 3>(); Wd[4] = St.template get<SV, WS_W + 4>(); Wd[5] = St.template get<SV, WS_W + 5>();
    })
    T d = T(0);
    if (GEN) {
      if (r.flags & TF_PRISMATIC) { d = sn; sn = T(0); }  // the sin slot holds the displacement
    }
    T Rn[9], pn[3], S[6];
    walk_compose<T, GEN>(W.R, W.p, rr, r.flags, sn, cs, d, Rn, pn);
#pragma unroll
    for (int k = 0; k < 9; ++k) W.R[k] = Rn[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) W.p[k] = pn[k];
    if (!(GEN && (r.flags & TF_FIXED))) {
      walk_subspace<T, GEN>(W.R, W.p, r.flags, S);
      const T vd = ud - dot6(Wd, W.ad);  // v̇ = D⁻¹u − (U D⁻¹)' a_Δ,parent
#pragma unroll
      for (int k = 0; k < 6; ++k) W.ad[k] += S[k] * vd;
      *walk_row(c, c.rt + r.voff, lane) = vd;
    }
  }
  if (r.a_w >= 0) {
    T* m = walk_row(c, c.rB + r.a_w * WMB_C, lane);
#pragma unroll
    for (int k = 0; k < 6; ++k) m[k * WR_STRIDE] = W.ad[k];
  }
  if (FLT && RR) {
    if (r.rrf & BFD_FCARRY) {  // v̇_f = S⁻¹ (a − a_world) = S⁻¹ ((a_vp − a_world) + a_Δ); then q̇ of the floating joint over its q rows
      T q7[7], v6[6], Rf[9], pf[3], d6[6], vf[6];
#pragma unroll
      for (int k = 0; k < 7; ++k) q7[k] = *walk_row(c, c.rq + c.rrv.fq + k, lane);
#pragma unroll
      for (int k = 0; k < 6; ++k) d6[k] = *walk_row(c, c.rt + c.rrv.fv + k, lane) + W.ad[k];
      reroot_fb_pose(c.rrv, q7, Rf, pf);
      xmotion_inv(Rf, pf, d6, vf);
#pragma unroll
      for (int k = 0; k < 6; ++k) *walk_row(c, c.rt + c.rrv.fv + k, lane) = vf[k];
      if (want_qdot) {  // velocity_to_configuration_derivative! (quaternion_floating.jl:126-136)
#pragma unroll
        for (int k = 0; k < 6; ++k) v6[k] = *walk_row(c, c.rv + c.rrv.fv + k, lane);
        const T qw = q7[0], qx = q7[1], qy = q7[2], qz = q7[3];
        T o[7], Rq[9];
        o[0] = (-qx * v6[0] - qy * v6[1] - qz * v6[2]) / 2;
        o[1] = (qw * v6[0] - qz * v6[1] + qy * v6[2]) / 2;
        o[2] = (qz * v6[0] + qw * v6[1] - qx * v6[2]) / 2;
        o[3] = (-qy * v6[0] + qx * v6[1] + qw * v6[2]) / 2;
        rot_quat(qw, qx, qy, qz, Rq);
        matvec3(Rq, v6 + 3, o + 4);
#pragma unroll
        for (int k = 0; k < 7; ++k) *walk_row(c, c.rq + c.rrv.fq + k, lane) = o[k];
      }
    }
  }
}

#if defined(__HIPCC__)
#ifdef RBD_PROFILE_PHASES
__device__ long long rbd_walk_phase_clock[32];  // [wave][mark]
#define RBD_WMARK(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) rbd_walk_phase_clock[(threadIdx.x >> 6) * 8 + (i)] = clock64(); } while (0)
#else
#define RBD_WMARK(i)
#endif

// Staging.  Rows [row0, row0 + n) hold the n x (64 N) block of a batch buffer that belongs to this workgroup's states (N states per lane).
// A state-major buffer (sk == 1) is ONE contiguous run of 64 N n scalars: thread t takes elements t, t + nthreads, ... (coalesced) and element e
// belongs to row (e mod n), state (e div n) — consecutive lanes hit consecutive rows, WR_STRIDE = 65 values apart: no bank conflicts.  A
// batch-innermost buffer is taken row by row.  States past the end of the batch read the last state's values (finite, never stored).
// All global loads of the prologue are in flight before its first LDS write (a wavefront pays every dependent global round trip in full).
struct WalkSlot { int k, st; long off; bool ok; };
template <int N> RBD_DEV WalkSlot walk_slot(Layout L, long state0, long B, int n, int e, unsigned inv) {
  WalkSlot x;
  x.ok = e < n * 64 * N;
  if (L.sk == 1) {
    x.st = (int)(((unsigned)e * inv) >> 22);  // e / n for e < 2^14, n < 2^8
    x.k = e - x.st * n;
    const long lim = (B - state0) * n;        // elements of this block that exist
    x.off = state0 * n + (e < lim ? e : lim - n + x.k);
  } else {
    x.k = e / (64 * N); x.st = e % (64 * N);
    const long sc = state0 + x.st < B ? state0 + x.st : B - 1;
    x.off = (long)x.k * L.sk + sc * L.sb;
  }
  return x;
}
template <typename T, int UB> struct WalkStageIn {
  using S = typename Lanes<T>::S;
  S a[UB], b[UB], d[UB];
  WalkSlot sa[UB], sb[UB];
  __device__ __forceinline__ void load(const S* __restrict__ q, const S* __restrict__ v, const S* __restrict__ tau, Layout Lq, Layout Lv, long state0, long B,
                                       int nq, int nv, int e0, int tid, int nth) {
    const unsigned invq = (1u << 22) / (unsigned)nq + 1, invv = (1u << 22) / (unsigned)(nv > 0 ? nv : 1) + 1;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * nth + tid;
      sa[u] = walk_slot<Lanes<T>::N>(Lq, state0, B, nq, e, invq);
      sb[u] = walk_slot<Lanes<T>::N>(Lv, state0, B, nv, e, invv);
      a[u] = sa[u].ok ? q[sa[u].off] : S(0);
      b[u] = (v && sb[u].ok) ? v[sb[u].off] : S(0);
      d[u] = (tau && sb[u].ok) ? tau[sb[u].off] : S(0);
    }
  }
  __device__ __forceinline__ void store(T* rows, int rq, int rv, int rt) const {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (sa[u].ok) *walk_cell(rows, rq + sa[u].k, sa[u].st) = a[u];
      if (sb[u].ok) { *walk_cell(rows, rv + sb[u].k, sb[u].st) = b[u]; *walk_cell(rows, rt + sb[u].k, sb[u].st) = d[u]; }
    }
  }
};
template <typename T, int UB>
__device__ __forceinline__ void walk_stage_out(typename Lanes<T>::S* __restrict__ dst, Layout L, long state0, long B, int n, T* rows, int row0, int tid, int nth) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N;
  if (!dst) return;
  const unsigned inv = (1u << 22) / (unsigned)(n > 0 ? n : 1) + 1;
  for (int e0 = 0; e0 < n * 64 * N; e0 += UB * nth) {
    S a[UB];
    WalkSlot sl[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      sl[u] = walk_slot<N>(L, state0, B, n, e0 + u * nth + tid, inv);
      a[u] = sl[u].ok ? *walk_cell(rows, row0 + sl[u].k, sl[u].st) : S(0);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (sl[u].ok && state0 + sl[u].st < B) dst[sl[u].off] = a[u];
  }
}

// The common case — state-major buffers, every state of the workgroup inside the batch — without per-element divisions or 64-bit offsets:
// thread t walks elements t, t + nthreads, ... of the contiguous block; (row, state) of an element advance by (nthreads mod n, nthreads div n)
// with a carry.  cell: index of a scalar in the rows, ((row) WR_STRIDE + state mod 64) N + state div 64.
template <int N> struct WalkCursor {
  int k, st, dk, ds, n;
  __device__ __forceinline__ void init(int n_, int tid, int nth) { n = n_ > 0 ? n_ : 1; k = tid % n; st = tid / n; dk = nth % n; ds = nth / n; }
  __device__ __forceinline__ int cell(int row0) const { return ((row0 + k) * WR_STRIDE + (st & 63)) * N + (st >> 6); }
  __device__ __forceinline__ void next() {
    k += dk; st += ds;
    const bool c = k >= n;
    k -= c ? n : 0; st += c ? 1 : 0;
  }
};
template <typename T, int UB>
__device__ __forceinline__ void walk_stage_in_fast(const typename Lanes<T>::S* __restrict__ q, const typename Lanes<T>::S* __restrict__ v,
                                                   const typename Lanes<T>::S* __restrict__ tau, long state0, int nq, int nv, T* rows, int rq, int rv, int rt,
                                                   int tid, int nth) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N;
  const S* bq = q + state0 * nq;
  const auto bv = as_global(v ? v + state0 * nv : nullptr);    // (behind the conditional the pointer has lost its address space: the rows' loads would be
  const auto bt = as_global(tau ? tau + state0 * nv : nullptr);  //  FLAT instructions)
  S* cells = reinterpret_cast<S*>(rows);
  const int totq = nq * 64 * N, totv = nv * 64 * N, tot = totq > totv ? totq : totv;
  WalkCursor<N> cq, cv;
  cq.init(nq, tid, nth); cv.init(nv, tid, nth);
  for (int e0 = tid; e0 < tot + tid; e0 += UB * nth) {
    S a[UB], b[UB], d[UB];
    int iq[UB], iv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * nth;
      const bool okq = e < totq, okv = e < totv;
      a[u] = okq ? bq[e] : S(0);
      b[u] = (bv && okv) ? bv[e] : S(0);
      d[u] = (bt && okv) ? bt[e] : S(0);
      iq[u] = okq ? cq.cell(rq) : -1;
      iv[u] = okv ? cv.cell(0) : -1;
      cq.next(); cv.next();
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (iq[u] >= 0) cells[iq[u]] = a[u];
      if (iv[u] >= 0) { cells[iv[u] + rv * WR_STRIDE * N] = b[u]; cells[iv[u] + rt * WR_STRIDE * N] = d[u]; }
    }
  }
}
template <typename T, int UB>
__device__ __forceinline__ void walk_stage_out_fast(typename Lanes<T>::S* __restrict__ dst, long state0, int n, const T* rows, int row0, int tid, int nth) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N;
  if (!dst) return;
  S* base = dst + state0 * n;
  const S* cells = reinterpret_cast<const S*>(rows);
  const int tot = n * 64 * N;
  WalkCursor<N> cu;
  cu.init(n, tid, nth);
  for (int e0 = tid; e0 < tot + tid; e0 += UB * nth) {
    S a[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      a[u] = e0 + u * nth < tot ? cells[cu.cell(row0)] : S(0);
      cu.next();
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (e0 + u * nth < tot) base[e0 + u * nth] = a[u];
  }
}

// ... and the batch-innermost layout (sk = B, sb = 1), every state inside the batch: row k of the workgroup's block is 64 N contiguous scalars
template <typename T, int UB>
__device__ __forceinline__ void walk_stage_in_rows(const typename Lanes<T>::S* __restrict__ q, const typename Lanes<T>::S* __restrict__ v,
                                                   const typename Lanes<T>::S* __restrict__ tau, long B, long state0, int nq, int nv, T* rows, int rq, int rv,
                                                   int rt, int tid, int nth) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N, W = 64 * N;
  S* cells = reinterpret_cast<S*>(rows);
  const int totq = nq * W, totv = nv * W, tot = totq > totv ? totq : totv;
  for (int e0 = tid; e0 < tot + tid; e0 += UB * nth) {
    S a[UB], b[UB], d[UB];
    int ic[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * nth, k = e / W, st = e % W;
      const long off = (long)k * B + state0 + st;
      a[u] = e < totq ? q[off] : S(0);
      b[u] = (v && e < totv) ? v[off] : S(0);
      d[u] = (tau && e < totv) ? tau[off] : S(0);
      ic[u] = (k * WR_STRIDE + (st & 63)) * N + (st >> 6);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * nth;
      if (e < totq) cells[ic[u] + rq * WR_STRIDE * N] = a[u];
      if (e < totv) { cells[ic[u] + rv * WR_STRIDE * N] = b[u]; cells[ic[u] + rt * WR_STRIDE * N] = d[u]; }
    }
  }
}
template <typename T, int UB>
__device__ __forceinline__ void walk_stage_out_rows(typename Lanes<T>::S* __restrict__ dst, long B, long state0, int n, const T* rows, int row0, int tid, int nth) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N, W = 64 * N;
  if (!dst) return;
  const S* cells = reinterpret_cast<const S*>(rows);
  const int tot = n * W;
  for (int e0 = tid; e0 < tot + tid; e0 += UB * nth) {
    S a[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * nth, k = e / W, st = e % W;
      a[u] = e < tot ? cells[((row0 + k) * WR_STRIDE + (st & 63)) * N + (st >> 6)] : S(0);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * nth, k = e / W, st = e % W;
      if (e < tot) dst[(long)k * B + state0 + st] = a[u];
    }
  }
}

// RR: the plan is that of a floating-base tree re-rooted at its centre (rbd_reroot.hpp)
template <typename T, bool FLT, bool GEN, bool RR = false>
__global__ __launch_bounds__(256) void aba_walk_kernel(WalkModel M, long B, const typename Lanes<T>::S* __restrict__ q, const typename Lanes<T>::S* __restrict__ v,
                                                      const typename Lanes<T>::S* __restrict__ tau, const typename Lanes<T>::S* __restrict__ fext,
                                                      typename Lanes<T>::S* __restrict__ vdot, typename Lanes<T>::S* __restrict__ qdot, Layout Lq, Layout Lv,
                                                      Layout Lf) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N;
  extern __shared__ __align__(16) unsigned char walk_lds_raw[];
  WalkCtx<T> c;
  c.M = M;
  walk_ctx_lds(c, walk_lds_raw);
  const int lane = threadIdx.x & 63, tid = threadIdx.x, nth = blockDim.x;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long state0 = (long)blockIdx.x * (64 * N);
  RBD_WMARK(0);
  {  // the plan records -> LDS; q, v, τ of this workgroup's states -> rows.  Loads first, all of them; then the LDS writes.
    constexpr int UB = 10 * N, TB = 6;
    const int nrec = M.ns * M.G, ntr = nrec * TR_STRIDE;  // nrec <= 13 G <= nth
    const I4* gi = reinterpret_cast<const I4*>(M.ri);
    const S* gr = reinterpret_cast<const S*>(M.rr);
    I4 ti = gi[tid < nrec ? tid : 0];
    const int32_t tw = M.wk[tid < nrec ? tid : 0];
    S tr[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) tr[u] = gr[tid + u * nth < ntr ? tid + u * nth : 0];
    const bool inside = state0 + 64 * N <= B;  // wave-uniform
    const bool fast = Lq.sk == 1 && Lv.sk == 1 && inside;
    const bool fast_rows = Lq.sb == 1 && Lv.sb == 1 && Lq.sk == B && Lv.sk == B && inside && !fast;
    WalkStageIn<T, UB> in;
    if (!fast && !fast_rows) in.load(q, v, tau, Lq, Lv, state0, B, M.nq, M.nv, 0, tid, nth);
    if (tid < nrec) { const_cast<I4*>(c.tri)[tid] = ti; const_cast<int32_t*>(c.twk)[tid] = tw; }
#pragma unroll
    for (int u = 0; u < TB; ++u)
      if (tid + u * nth < ntr) const_cast<S*>(c.trr)[tid + u * nth] = tr[u];
    for (int i = tid + TB * nth; i < ntr; i += nth) const_cast<S*>(c.trr)[i] = gr[i];
    if (RR) walk_stage_chain(c, tid, nth);
    if (fast) {
      walk_stage_in_fast<T, UB>(q, v, tau, state0, M.nq, M.nv, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else if (fast_rows) {
      walk_stage_in_rows<T, UB>(q, v, tau, B, state0, M.nq, M.nv, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else {
      in.store(c.rows, c.rq, c.rv, c.rt);
      const int nmax = (M.nq > M.nv ? M.nq : M.nv) * 64 * N;
      for (int e0 = UB * nth; e0 < nmax; e0 += UB * nth) {
        in.load(q, v, tau, Lq, Lv, state0, B, M.nq, M.nv, e0, tid, nth);
        in.store(c.rows, c.rq, c.rv, c.rt);
      }
    }
  }
  __syncthreads();
  RBD_WMARK(1);
  asm volatile("" ::: "a255");  // the kernel descriptor covers every accumulation register (WalkStash addresses them by number)
  WalkRegs<T> W;
  WalkStash<T> St;
  walk_init(W);
  const int ns = M.ns;
  const bool want_qdot = qdot != nullptr;
  // external wrenches: requested from global memory a step ahead (one load per state of the lane)
  const long fsk = Lf.sk;
  RBD_GLOBAL_PTR(const S) fel[N];  // (global, said in the type: behind the conditional below the compiler would emit FLAT loads)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const long st = state0 + 64 * j + lane;
    fel[j] = as_global(fext ? fext + (st < B ? st : B - 1) * Lf.sb : nullptr);
  }
  auto wrench = [&](int o6, T* f) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      S x[N];
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = fel[j][(long)(o6 + k) * fsk];
      if constexpr (N == 1) f[k] = x[0];
      else { f[k].x = x[0]; f[k].y = x[1]; }
    }
  };
  // every pass reads the record (and passes A and C the constants) of its next step while the current one computes
  {
    WalkRaw raw = walk_raw(c, 0, g);
    T rr[TR_J];
    walk_consts<T, TR_J>(c, 0, g, rr);
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
      const int s1 = s + 1 < ns ? s + 1 : s;
      const WalkRec r = walk_rec(raw);
      raw = walk_raw(c, s1, g);
      T rn[TR_J];
      walk_consts<T, TR_J>(c, s1, g, rn);
      walk_step_a<T, FLT, GEN, false, RR>(c, W, St, s, r, rr, lane, want_qdot);
#pragma unroll
      for (int k = 0; k < TR_J; ++k) rr[k] = rn[k];
      if ((M.sfm[1] >> s) & 1) __syncthreads();  // SF_AW: an A mailbox was written at this step
    }
  }
  __syncthreads();  // the B mailboxes take over the rows of the A mailboxes' twist halves
  RBD_WMARK(2);
  {
    // pass B: the external wrench of the body of step s − 1 is requested while step s computes
    walk_init_b(W);
    T fe[6], fn[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { fe[k] = T(0); fn[k] = T(0); }
    WalkRaw raw = walk_raw(c, ns - 1, g);
    if (fext) wrench(walk_uniform(raw.w.y) & 0xffff, fe);
#pragma unroll 1
    for (int s = ns - 1; s >= 0; --s) {
      const int s1 = s > 0 ? s - 1 : 0;
      const WalkRec r = walk_rec(raw);
      if (FLT && RR) {
        if ((r.rrf & BFD_FCARRY) && (r.flags & TF_VALID)) {
          if (r.park >= 0) walk_get_park(c, r.park, lane, W);
          walk_fcarry_b(c, W, lane, fe);
        }
      }
      T rr[TR_STRIDE];
      walk_consts<T, TR_STRIDE>(c, s, g, rr);
      raw = walk_raw(c, s1, g);
      if (fext) wrench(walk_uniform(c.tri[s1 * M.G + g].y) & 0xffff, fn);
      walk_step_b<T, FLT, GEN, RR>(c, W, St, s, r, rr, lane, fe);
#pragma unroll
      for (int k = 0; k < 6; ++k) fe[k] = fn[k];
      if ((M.sfm[3] >> s) & 1) __syncthreads();  // SF_BW: a hand-off left its track at this step
    }
  }
  __syncthreads();  // pass C re-uses the B mailboxes
  RBD_WMARK(3);
  {
    walk_init_c(W);
    WalkRaw raw = walk_raw(c, 0, g);
    T rr[TR_J];
    walk_consts<T, TR_J>(c, 0, g, rr);
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
      const int s1 = s + 1 < ns ? s + 1 : s;
      const WalkRec r = walk_rec(raw);
      raw = walk_raw(c, s1, g);
      T rn[TR_J];
      walk_consts<T, TR_J>(c, s1, g, rn);
      walk_step_c<T, FLT, GEN, RR>(c, W, St, s, r, rr, lane, want_qdot);
#pragma unroll
      for (int k = 0; k < TR_J; ++k) rr[k] = rn[k];
      if ((M.sfm[1] >> s) & 1) __syncthreads();
    }
  }
  __syncthreads();
  RBD_WMARK(4);
  if (Lq.sk == 1 && Lv.sk == 1 && state0 + 64 * N <= B) {
    walk_stage_out_fast<T, 10 * N>(vdot, state0, M.nv, c.rows, c.rt, tid, nth);
    walk_stage_out_fast<T, 10 * N>(qdot, state0, M.nq, c.rows, c.rq, tid, nth);
  } else if (Lq.sb == 1 && Lv.sb == 1 && Lq.sk == B && Lv.sk == B && state0 + 64 * N <= B) {
    walk_stage_out_rows<T, 10 * N>(vdot, B, state0, M.nv, c.rows, c.rt, tid, nth);
    walk_stage_out_rows<T, 10 * N>(qdot, B, state0, M.nq, c.rows, c.rq, tid, nth);
  } else {
    walk_stage_out<T, 10>(vdot, Lv, state0, B, M.nv, c.rows, c.rt, tid, nth);
    walk_stage_out<T, 10>(qdot, Lq, state0, B, M.nq, c.rows, c.rq, tid, nth);
  }
  RBD_WMARK(5);
}
// ---- aba_walk_kernel compiled for ONE mechanism at run time (rbd_jit.hip prints PLAN: the plan's records and constants as __device__ const
// tables, its counts as constants).  Same step functions, same rows, same mailboxes and barriers; what changes is that every
// (step, track) is known when the kernel is compiled: the step loops are unrolled per track, the records' flags fold the step functions down to the
// code of that body's joint, the constants are literals and the plan takes no LDS and no staging.
template <int I> struct WalkIx { static constexpr int value = I; };
template <int I, int N, typename F> RBD_DEV void walk_sfor(F&& f) {
  if constexpr (I < N) {
    f(WalkIx<I>{});
    walk_sfor<I + 1, N>(f);
  }
}
template <typename T, typename PLAN> RBD_DEV void walk_ctx_spec(WalkCtx<T>& c, void* lds) {
  using S = typename Lanes<T>::S;
  c.M = PLAN::model();
  c.tri = reinterpret_cast<const I4*>(c.M.ri);
  c.trr = reinterpret_cast<const S*>(c.M.rr);
  c.twk = c.M.wk;
  c.rrv = c.M.reroot;
  c.rows = reinterpret_cast<T*>(lds);
  c.rq = 0; c.rv = c.M.nq; c.rt = c.rv + c.M.nv; c.rA = c.rt + c.M.nv; c.rS = c.rA + c.M.nA * WMB_A; c.rB = c.rS + c.M.nS * WMB_S;
  c.rT = c.rB; c.sT = WMB_AT;
}
// (the instruction scheduler works on whole basic blocks: without a fence between the unrolled steps it pulls the next steps' LDS reads and constants
// forward until the 512 registers of a lone wavefront are full and the allocator spills hundreds of values)
#ifndef RBD_WALK_STEP_FENCE
#define RBD_WALK_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <typename T, bool FLT, bool GEN, bool RR, typename PLAN, int GI>
RBD_DEV void aba_walk_spec_track(const WalkCtx<T>& c, long B, const typename Lanes<T>::S* __restrict__ fext, bool want_qdot, Layout Lf, long state0, int lane) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N, NS = PLAN::NS;
  WalkRegs<T> W;
  WalkStash<T> St;
  walk_init(W);
  const long fsk = Lf.sk;
  RBD_GLOBAL_PTR(const S) fel[N];  // (global, said in the type: behind the conditional below the compiler would emit FLAT loads)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const long st = state0 + 64 * j + lane;
    fel[j] = as_global(fext ? fext + (st < B ? st : B - 1) * Lf.sb : nullptr);
  }
  auto wrench = [&](int o6, T* f) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      S x[N];
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = fel[j][(long)(o6 + k) * fsk];
      if constexpr (N == 1) f[k] = x[0];
      else { f[k].x = x[0]; f[k].y = x[1]; }
    }
  };
  walk_sfor<0, NS>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    const WalkRec r = walk_rec(walk_raw(c, s, GI));
    T rr[TR_J];
    walk_consts<T, TR_J>(c, s, GI, rr);
    walk_step_a<T, FLT, GEN, false, RR>(c, W, St, s, r, rr, lane, want_qdot);
    if constexpr ((PLAN::SFM_AW >> s) & 1) __syncthreads();
    RBD_WALK_STEP_FENCE();
  });
  __syncthreads();  // the B mailboxes take over the rows of the A mailboxes' twist halves
  {
    walk_init_b(W);
    T fe[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) fe[k] = T(0);
    if (fext) wrench(walk_rec(walk_raw(c, NS - 1, GI)).orig6, fe);
    walk_sfor<0, NS>([&](auto si) __attribute__((always_inline)) {
      constexpr int s = NS - 1 - decltype(si)::value, s1 = s > 0 ? s - 1 : 0;
      const WalkRec r = walk_rec(walk_raw(c, s, GI));
      if constexpr (FLT && RR) {
        if ((r.rrf & BFD_FCARRY) && (r.flags & TF_VALID)) {
          if (r.park >= 0) walk_get_park(c, r.park, lane, W);
          walk_fcarry_b(c, W, lane, fe);
        }
      }
      T rr[TR_STRIDE], fn[6];
      walk_consts<T, TR_STRIDE>(c, s, GI, rr);
#pragma unroll
      for (int k = 0; k < 6; ++k) fn[k] = T(0);
      if (fext) wrench(walk_rec(walk_raw(c, s1, GI)).orig6, fn);  // the external wrench of the body of step s − 1 is requested while step s computes
      walk_step_b<T, FLT, GEN, RR>(c, W, St, s, r, rr, lane, fe);
#pragma unroll
      for (int k = 0; k < 6; ++k) fe[k] = fn[k];
      if constexpr ((PLAN::SFM_BW >> s) & 1) __syncthreads();
      RBD_WALK_STEP_FENCE();
    });
  }
  __syncthreads();  // pass C re-uses the B mailboxes
  walk_init_c(W);
  walk_sfor<0, NS>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    const WalkRec r = walk_rec(walk_raw(c, s, GI));
    T rr[TR_J];
    walk_consts<T, TR_J>(c, s, GI, rr);
    walk_step_c<T, FLT, GEN, RR>(c, W, St, s, r, rr, lane, want_qdot);
    if constexpr ((PLAN::SFM_AW >> s) & 1) __syncthreads();
    RBD_WALK_STEP_FENCE();
  });
}
// LOOP: the instantiation that takes all four stages of a `simulate` step in one launch (MkStage::stage = 4) — its passes stand inside a loop, which changes
// what the register allocator does with the whole kernel (it takes accumulation registers of its own: rbd_jit.hip accepts that for this instantiation only,
// and the library checks it against the single-stage instantiation on the first states of the first batch before it is used: rbd_capi.hip).  The other
// instantiation is straight-line code as before and serves everything else.
template <typename T, bool FLT, bool GEN, bool RR, typename PLAN, bool LOOP = false>
RBD_DEV void aba_walk_spec(long B, const typename Lanes<T>::S* __restrict__ q, const typename Lanes<T>::S* __restrict__ v, const typename Lanes<T>::S* __restrict__ tau,
                           const typename Lanes<T>::S* __restrict__ fext, typename Lanes<T>::S* __restrict__ vdot, typename Lanes<T>::S* __restrict__ qdot, Layout Lq,
                           Layout Lv, Layout Lf, double gx, double gy, double gz, unsigned char* lds, const MkStage& F = MkStage{-1, 0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
                           const int32_t* mk1 = nullptr, const int32_t* mkf = nullptr) {
  constexpr int N = Lanes<T>::N, NQ = PLAN::NQ, NV = PLAN::NV;
  using S = typename Lanes<T>::S;
  WalkCtx<T> c;
  walk_ctx_spec<T, PLAN>(c, lds);
  c.a0[0] = T(0); c.a0[1] = T(0); c.a0[2] = T(0);
  c.a0[3] = T(-gx); c.a0[4] = T(-gy); c.a0[5] = T(-gz);  // a_world = −gravity (mechanism_algorithms.jl:405)
  const int lane = threadIdx.x & 63, tid = threadIdx.x, nth = blockDim.x;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long state0 = (long)blockIdx.x * (64 * N);
  const bool inside = state0 + 64 * N <= B;  // wave-uniform
  const bool fast = Lq.sk == 1 && Lv.sk == 1 && inside;
  const bool fast_rows = Lq.sb == 1 && Lv.sb == 1 && Lq.sk == B && Lv.sk == B && inside && !fast;
  {
    constexpr int UB = 10 * N;
    if (fast) {
      walk_stage_in_fast<T, UB>(q, v, tau, state0, NQ, NV, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else if (fast_rows) {
      walk_stage_in_rows<T, UB>(q, v, tau, B, state0, NQ, NV, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else {
      WalkStageIn<T, UB> in;
      const int nmax = (NQ > NV ? NQ : NV) * 64 * N;
      for (int e0 = 0; e0 < nmax; e0 += UB * nth) {
        in.load(q, v, tau, Lq, Lv, state0, B, NQ, NV, e0, tid, nth);
        in.store(c.rows, c.rq, c.rv, c.rt);
      }
    }
  }
  __syncthreads();
  // `simulate`: this launch is stage F.stage of a Munthe-Kaas RK4 step — or, F.stage = 4, ALL FOUR of them — in the lane-is-state form of rbd_mk_fuse.hpp: the
  // whole stage stands behind the passes (they read the q and v rows and leave them alone when no q̇ is asked for); only the PD law needs the stage state in
  // front of them.  With the four stages in one launch the stage states never leave the rows: per step one trip of q, v in and out, four of τ in, and the
  // stage buffers (base point, two slopes) — a third of the bytes of four launches, and one launch's tail instead of four.
  // (the rows are LDS: said in the pointer's type, or the stage's stores — behind selects and uniform branches — come out as FLAT instructions)
  typedef __attribute__((address_space(3))) S* LdsS;
  auto cell = [&](int row, int st) __attribute__((always_inline)) { return (LdsS)(reinterpret_cast<S*>(c.rows + (long)row * WR_STRIDE + (st & 63)) + (st >> 6)); };
  // rows [row0, row0 + n) out to dst in the caller's layout (dst nullable)
  auto send = [&](S* dst, Layout L, int n, int row0) __attribute__((always_inline)) {
    if (fast) walk_stage_out_fast<T, 10 * N>(dst, state0, n, c.rows, row0, tid, nth);
    else if (fast_rows) walk_stage_out_rows<T, 10 * N>(dst, B, state0, n, c.rows, row0, tid, nth);
    else walk_stage_out<T, 10>(dst, L, state0, B, n, c.rows, row0, tid, nth);
  };
  // WalkStash addresses accumulation registers by number, which the compiler does not see: rbd_jit.hip checks in the code object's metadata that the register
  // allocator took NONE of its own, then rewrites the kernel descriptor to cover all 256 (jit_kd_cover_agprs) — no `a255` clobber here, the compiler must not
  // know (with it the metadata could no longer tell the allocator's registers from the stash's; round 3 compiled every program twice for that).
  const bool want_qdot = qdot != nullptr;
  const int s_first = (LOOP && F.stage == 4) ? 0 : F.stage, s_last = (LOOP && F.stage == 4) ? 3 : F.stage;  // uniform (-1: plain dynamics!)
  // one stage: the PD law, the passes, the stage behind them.  st0 / Bo / Lfo / lane_o: state0, B, Lf, lane — as they are, or made opaque per turn of the loop
  auto one_stage = [&](int sg, long st0, long Bo, Layout Lfo, int lane_o) __attribute__((always_inline)) {
    MkStage Fs = F;
    Fs.stage = sg;
    const int32_t* const mk1o = mk1; const int32_t* const mkfo = mkf;
    if (sg >= 0 && F.pd) {  // uniform
      mk_lane_pd<S, PLAN::MK_N1, PLAN::G, N>(Fs, cell, mk1o, c.rq, c.rv, c.rt, st0, Bo, Lq, g, lane_o);
      __syncthreads();
    }
    walk_sfor<0, PLAN::G>([&](auto gi) __attribute__((always_inline)) {
      constexpr int GI = decltype(gi)::value;
      if (g == GI) aba_walk_spec_track<T, FLT, GEN, RR, PLAN, GI>(c, Bo, fext, want_qdot, Lfo, st0, lane_o);
    });
#ifndef RBD_WALK_NO_STAGE
    MkLane<S, PLAN::MK_N1, PLAN::MK_NF, PLAN::G, N> mkl;
    if (sg >= 0) {  // (asked for — and the 6-dof joints' SE(3) arithmetic done — while the other tracks finish)
      mk_lane_load_float<S, PLAN::MK_N1, PLAN::MK_NF, PLAN::G, N>(Fs, mkl, mkfo, st0, Bo, g, lane_o, PLAN::MK_FW);
      mk_lane_float_q<S, PLAN::MK_N1, PLAN::MK_NF, PLAN::G, N>(Fs, mkl, cell, mkfo, c.rq, c.rv, st0, Bo, g, lane_o, PLAN::MK_FW);
      mk_lane_load<S, PLAN::MK_N1, PLAN::MK_NF, PLAN::G, N>(Fs, mkl, mk1o, mkfo, st0, Bo, g, lane_o, PLAN::MK_FW);
    }
    __syncthreads();
    if (sg >= 0) {  // uniform
      mk_lane_stage<S, PLAN::MK_N1, PLAN::MK_NF, PLAN::G, N>(Fs, mkl, cell, mk1o, mkfo, c.rq, c.rv, c.rt, st0, Bo, g, lane_o, PLAN::MK_FW);
      if (sg < s_last) {  // the next stage's torques into the τ rows (they held v̇): the table's next entry (F.tau_stride elements on), or the same ones again
        __syncthreads();  // (every wavefront has read its joints' v̇)
        const auto tn = as_global(tau ? tau + (long)(sg + 1) * F.tau_stride : nullptr);
        for (int e = tid; e < NV * 64 * N; e += nth) {
          int k, st;
          if (Lv.sk == 1) { st = e / NV; k = e - st * NV; } else { k = e / (64 * N); st = e - k * (64 * N); }  // consecutive threads on consecutive addresses either way
          const long gi = st0 + st;
          *cell(c.rt + k, st) = (tn && gi < Bo) ? tn[(long)k * Lv.sk + gi * Lv.sb] : S(0);
        }
      }
      __syncthreads();
    }
#else
    __syncthreads();
#endif
  };
  if constexpr (LOOP) {
#pragma unroll 1
    for (int sg = s_first; sg <= s_last; ++sg) {
      // every address of the loop body hangs on the workgroup's first state, the batch size, the wrenches' strides, the lane: made opaque here, or the compiler
      // lifts ~140 64-bit address computations (the external wrenches', the stage buffers'), their scalar offset products and every LDS row address out of the
      // loop and keeps them across the passes — 1.7 KB of scratch per lane, 512 registers.  (The materialisation of the passes' constants is lifted the same
      // way by the machine-level loop-invariant code motion: switched off for this program, rbd_jit.hip.)
      long st0 = state0, Bo = B;
      Layout Lfo = Lf;
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      asm volatile("" : "+s"(st0), "+s"(Bo), "+s"(Lfo.sk), "+s"(Lfo.sb));
      one_stage(sg, st0, Bo, Lfo, lane_o);
    }
  } else {
    one_stage(F.stage, state0, B, Lf, lane);
  }
  if (F.stage >= 0) {
    // F.q_state / F.v_state ARE the kernel's own q / v inputs (the arguments' address space is known to the compiler, that of the struct's pointers is not —
    // and neither is that of a pointer that was null on another path: the stores would come out as FLAT instructions): this workgroup has read its states'
    // rows, nobody else touches them
    send(const_cast<S*>(q), Lq, NQ, c.rq);
    send(const_cast<S*>(v), Lv, NV, c.rv);
  }
  send(vdot, Lv, NV, c.rt);
  send(qdot, Lq, NQ, c.rq);
}
// ... and rnea_walk_kernel (below) the same way: inverse_dynamics! / dynamics_bias! with the optional per-body outputs
template <typename T, bool FLT, bool GEN, typename PLAN, int GI>
RBD_DEV void rnea_walk_spec_track(const WalkCtx<T>& c, long B, const typename Lanes<T>::S* __restrict__ fext, bool want_qdot, Layout Lf, long state0, int lane,
                                  bool shared_rows, typename Lanes<T>::S* __restrict__ acc_out, typename Lanes<T>::S* __restrict__ jw_out) {
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N, NS = PLAN::NS;
  WalkRegs<T> W;
  WalkStash<T> St;
  walk_init(W);
  const long fsk = Lf.sk;
  RBD_GLOBAL_PTR(const S) fel[N];  // (global, said in the type: behind the conditional below the compiler would emit FLAT loads)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const long st = state0 + 64 * j + lane;
    fel[j] = as_global(fext ? fext + (st < B ? st : B - 1) * Lf.sb : nullptr);
  }
  auto wrench = [&](int o6, T* f) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      S x[N];
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = fel[j][(long)(o6 + k) * fsk];
      if constexpr (N == 1) f[k] = x[0];
      else { f[k].x = x[0]; f[k].y = x[1]; }
    }
  };
  walk_sfor<0, NS>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = decltype(si)::value;
    const WalkRec r = walk_rec(walk_raw(c, s, GI));
    T rr[TR_J];
    walk_consts<T, TR_J>(c, s, GI, rr);
    walk_step_a<T, FLT, GEN, true>(c, W, St, s, r, rr, lane, want_qdot);
    if constexpr ((PLAN::SFM_AW >> s) & 1) __syncthreads();
    RBD_WALK_STEP_FENCE();
  });
  if (shared_rows) __syncthreads();  // only then do the B mailboxes take over the rows of the A mailboxes' twist halves
#pragma unroll
  for (int k = 0; k < 6; ++k) W.cP[k] = T(0);
  T fe[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) fe[k] = T(0);
  if (fext) wrench(walk_rec(walk_raw(c, NS - 1, GI)).orig6, fe);
  const bool outs = acc_out != nullptr || jw_out != nullptr;  // uniform
  const bool out_vec = outs && store6_vec(acc_out ? acc_out : jw_out, Lf, (int)sizeof(S)) && store6_vec(jw_out ? jw_out : acc_out, Lf, (int)sizeof(S));
  walk_sfor<0, NS>([&](auto si) __attribute__((always_inline)) {
    constexpr int s = NS - 1 - decltype(si)::value, s1 = s > 0 ? s - 1 : 0;
    const WalkRec r = walk_rec(walk_raw(c, s, GI));
    T rr[TR_STRIDE], fn[6];
    walk_consts<T, TR_STRIDE>(c, s, GI, rr);
#pragma unroll
    for (int k = 0; k < 6; ++k) fn[k] = T(0);
    if (fext) wrench(walk_rec(walk_raw(c, s1, GI)).orig6, fn);
    if (outs) {
      T ao[6], jo[6];
      walk_step_rb<T, FLT, GEN>(c, W, St, s, r, rr, lane, fe, ao, jo);
      if (r.flags & TF_VALID) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const long st = state0 + 64 * j + lane;
          if (st < B) {
            S av[6], jv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              if constexpr (N == 1) { av[k] = ao[k]; jv[k] = jo[k]; }
              else { av[k] = j == 0 ? ao[k].x : ao[k].y; jv[k] = j == 0 ? jo[k].x : jo[k].y; }
            }
#ifdef RBD_WALK_ABLATE_OUT_LOCAL
            const long so = st & 63;
#else
            const long so = st;
#endif
            if (acc_out) store6(acc_out, (long)r.orig6, Lf, so, av, out_vec);
            if (jw_out) store6(jw_out, (long)r.orig6, Lf, so, jv, out_vec);
          }
        }
      }
    } else {
      walk_step_rb<T, FLT, GEN>(c, W, St, s, r, rr, lane, fe);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) fe[k] = fn[k];
    if constexpr ((PLAN::SFM_BW >> s) & 1) __syncthreads();
    RBD_WALK_STEP_FENCE();
  });
}
template <typename T, bool FLT, bool GEN, typename PLAN>
RBD_DEV void rnea_walk_spec(long B, const typename Lanes<T>::S* __restrict__ q, const typename Lanes<T>::S* __restrict__ v, const typename Lanes<T>::S* __restrict__ vdot,
                            const typename Lanes<T>::S* __restrict__ fext, typename Lanes<T>::S* __restrict__ tau, typename Lanes<T>::S* __restrict__ qdot, Layout Lq,
                            Layout Lv, Layout Lf, typename Lanes<T>::S* __restrict__ acc_out, typename Lanes<T>::S* __restrict__ jw_out, double gx, double gy, double gz,
                            unsigned char* lds) {
  constexpr int N = Lanes<T>::N, NQ = PLAN::NQ, NV = PLAN::NV;
  WalkCtx<T> c;
  walk_ctx_spec<T, PLAN>(c, lds);
  c.a0[0] = T(0); c.a0[1] = T(0); c.a0[2] = T(0);
  c.a0[3] = T(-gx); c.a0[4] = T(-gy); c.a0[5] = T(-gz);
  const bool shared_rows = !walk_twist_rows_rnea(c);
  const int lane = threadIdx.x & 63, tid = threadIdx.x, nth = blockDim.x;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long state0 = (long)blockIdx.x * (64 * N);
  const bool inside = state0 + 64 * N <= B;  // wave-uniform
  const bool fast = Lq.sk == 1 && Lv.sk == 1 && inside;
  const bool fast_rows = Lq.sb == 1 && Lv.sb == 1 && Lq.sk == B && Lv.sk == B && inside && !fast;
  {
    constexpr int UB = 10 * N;
    if (fast) {
      walk_stage_in_fast<T, UB>(q, v, vdot, state0, NQ, NV, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else if (fast_rows) {
      walk_stage_in_rows<T, UB>(q, v, vdot, B, state0, NQ, NV, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else {
      WalkStageIn<T, UB> in;
      const int nmax = (NQ > NV ? NQ : NV) * 64 * N;
      for (int e0 = 0; e0 < nmax; e0 += UB * nth) {
        in.load(q, v, vdot, Lq, Lv, state0, B, NQ, NV, e0, tid, nth);
        in.store(c.rows, c.rq, c.rv, c.rt);
      }
    }
  }
  __syncthreads();
  // (accumulation registers: see aba_walk_spec)
  const bool want_qdot = qdot != nullptr;
  walk_sfor<0, PLAN::G>([&](auto gi) __attribute__((always_inline)) {
    constexpr int GI = decltype(gi)::value;
    if (g == GI) rnea_walk_spec_track<T, FLT, GEN, PLAN, GI>(c, B, fext, want_qdot, Lf, state0, lane, shared_rows, acc_out, jw_out);
  });
  __syncthreads();
  if (fast) {
    walk_stage_out_fast<T, 10 * N>(tau, state0, NV, c.rows, c.rt, tid, nth);
    walk_stage_out_fast<T, 10 * N>(qdot, state0, NQ, c.rows, c.rq, tid, nth);
  } else if (fast_rows) {
    walk_stage_out_rows<T, 10 * N>(tau, B, state0, NV, c.rows, c.rt, tid, nth);
    walk_stage_out_rows<T, 10 * N>(qdot, B, state0, NQ, c.rows, c.rq, tid, nth);
  } else {
    walk_stage_out<T, 10>(tau, Lv, state0, B, NV, c.rows, c.rt, tid, nth);
    walk_stage_out<T, 10>(qdot, Lq, state0, B, NQ, c.rows, c.rq, tid, nth);
  }
}
// inverse_dynamics! (vdot given) / dynamics_bias! (vdot == nullptr) through the same schedule: pass A with the full accelerations, then the
// wrench pass (walk_step_rb).  src/mechanism_algorithms.jl:542-553, :484-498.
template <typename T, bool FLT, bool GEN>
__global__ __launch_bounds__(256) void rnea_walk_kernel(WalkModel M, long B, const typename Lanes<T>::S* __restrict__ q, const typename Lanes<T>::S* __restrict__ v,
                                                      const typename Lanes<T>::S* __restrict__ vdot, const typename Lanes<T>::S* __restrict__ fext,
                                                      typename Lanes<T>::S* __restrict__ tau, typename Lanes<T>::S* __restrict__ qdot, Layout Lq, Layout Lv,
                                                      Layout Lf, typename Lanes<T>::S* __restrict__ acc_out, typename Lanes<T>::S* __restrict__ jw_out) {
  // acc_out / jw_out (nullable): accelerations[body] and jointwrenches[body] of inverse_dynamics! / dynamics_bias! (spatial_accelerations!
  // src/mechanism_algorithms.jl:387-417, joint_wrenches_and_torques! :442-459), 6 x n_bodies per state in the layout of fext, root frame.  Both are
  // in registers when pass B reaches the body (its acceleration restored, its wrench summed): written from there, nothing is recomputed.
  using S = typename Lanes<T>::S;
  constexpr int N = Lanes<T>::N;
  extern __shared__ __align__(16) unsigned char walk_lds_raw[];
  WalkCtx<T> c;
  c.M = M;
  walk_ctx_lds(c, walk_lds_raw);
  const bool shared_rows = !walk_twist_rows_rnea(c);  // wave-uniform
  const int lane = threadIdx.x & 63, tid = threadIdx.x, nth = blockDim.x;
  const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long state0 = (long)blockIdx.x * (64 * N);
  RBD_WMARK(0);
  {  // the plan records -> LDS; q, v, τ of this workgroup's states -> rows.  Loads first, all of them; then the LDS writes.
    constexpr int UB = 10 * N, TB = 6;
    const int nrec = M.ns * M.G, ntr = nrec * TR_STRIDE;  // nrec <= 13 G <= nth
    const I4* gi = reinterpret_cast<const I4*>(M.ri);
    const S* gr = reinterpret_cast<const S*>(M.rr);
    I4 ti = gi[tid < nrec ? tid : 0];
    const int32_t tw = M.wk[tid < nrec ? tid : 0];
    S tr[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) tr[u] = gr[tid + u * nth < ntr ? tid + u * nth : 0];
    const bool inside = state0 + 64 * N <= B;  // wave-uniform
    const bool fast = Lq.sk == 1 && Lv.sk == 1 && inside;
    const bool fast_rows = Lq.sb == 1 && Lv.sb == 1 && Lq.sk == B && Lv.sk == B && inside && !fast;
    WalkStageIn<T, UB> in;
    if (!fast && !fast_rows) in.load(q, v, vdot, Lq, Lv, state0, B, M.nq, M.nv, 0, tid, nth);
    if (tid < nrec) { const_cast<I4*>(c.tri)[tid] = ti; const_cast<int32_t*>(c.twk)[tid] = tw; }
#pragma unroll
    for (int u = 0; u < TB; ++u)
      if (tid + u * nth < ntr) const_cast<S*>(c.trr)[tid + u * nth] = tr[u];
    for (int i = tid + TB * nth; i < ntr; i += nth) const_cast<S*>(c.trr)[i] = gr[i];
    walk_stage_chain(c, tid, nth);
    if (fast) {
      walk_stage_in_fast<T, UB>(q, v, vdot, state0, M.nq, M.nv, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else if (fast_rows) {
      walk_stage_in_rows<T, UB>(q, v, vdot, B, state0, M.nq, M.nv, c.rows, c.rq, c.rv, c.rt, tid, nth);
    } else {
      in.store(c.rows, c.rq, c.rv, c.rt);
      const int nmax = (M.nq > M.nv ? M.nq : M.nv) * 64 * N;
      for (int e0 = UB * nth; e0 < nmax; e0 += UB * nth) {
        in.load(q, v, vdot, Lq, Lv, state0, B, M.nq, M.nv, e0, tid, nth);
        in.store(c.rows, c.rq, c.rv, c.rt);
      }
    }
  }
  __syncthreads();
  RBD_WMARK(1);
  asm volatile("" ::: "a255");  // the kernel descriptor covers every accumulation register (WalkStash addresses them by number)
  WalkRegs<T> W;
  WalkStash<T> St;
  walk_init(W);
  const int ns = M.ns;
  const bool want_qdot = qdot != nullptr;
  // external wrenches: requested from global memory a step ahead (one load per state of the lane)
  const long fsk = Lf.sk;
  RBD_GLOBAL_PTR(const S) fel[N];  // (global, said in the type: behind the conditional below the compiler would emit FLAT loads)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const long st = state0 + 64 * j + lane;
    fel[j] = as_global(fext ? fext + (st < B ? st : B - 1) * Lf.sb : nullptr);
  }
  auto wrench = [&](int o6, T* f) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      S x[N];
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = fel[j][(long)(o6 + k) * fsk];
      if constexpr (N == 1) f[k] = x[0];
      else { f[k].x = x[0]; f[k].y = x[1]; }
    }
  };
  // every pass reads the record (and passes A and C the constants) of its next step while the current one computes
  {
    WalkRaw raw = walk_raw(c, 0, g);
    T rr[TR_J];
    walk_consts<T, TR_J>(c, 0, g, rr);
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
      const int s1 = s + 1 < ns ? s + 1 : s;
      const WalkRec r = walk_rec(raw);
      raw = walk_raw(c, s1, g);
      T rn[TR_J];
      walk_consts<T, TR_J>(c, s1, g, rn);
      walk_step_a<T, FLT, GEN, true>(c, W, St, s, r, rr, lane, want_qdot);
#pragma unroll
      for (int k = 0; k < TR_J; ++k) rr[k] = rn[k];
      if ((M.sfm[1] >> s) & 1) __syncthreads();  // SF_AW: an A mailbox was written at this step
    }
  }
  if (shared_rows) __syncthreads();  // only then do the B mailboxes take over the rows of the A mailboxes' twist halves
  RBD_WMARK(2);
  {
    // pass B: the external wrench of the body of step s − 1 is requested while step s computes
#pragma unroll
    for (int k = 0; k < 6; ++k) W.cP[k] = T(0);
    T fe[6], fn[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { fe[k] = T(0); fn[k] = T(0); }
    WalkRaw raw = walk_raw(c, ns - 1, g);
    if (fext) wrench(walk_uniform(raw.w.y) & 0xffff, fe);
#pragma unroll 1
    for (int s = ns - 1; s >= 0; --s) {
      const int s1 = s > 0 ? s - 1 : 0;
      const WalkRec r = walk_rec(raw);
      T rr[TR_STRIDE];
      walk_consts<T, TR_STRIDE>(c, s, g, rr);
      raw = walk_raw(c, s1, g);
      if (fext) wrench(walk_uniform(c.tri[s1 * M.G + g].y) & 0xffff, fn);
      if (acc_out != nullptr || jw_out != nullptr) {  // uniform
        const bool out_vec = store6_vec(acc_out ? acc_out : jw_out, Lf, (int)sizeof(S)) && store6_vec(jw_out ? jw_out : acc_out, Lf, (int)sizeof(S));
        T ao[6], jo[6];
        walk_step_rb<T, FLT, GEN>(c, W, St, s, r, rr, lane, fe, ao, jo);
        if (r.flags & TF_VALID) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const long st = state0 + 64 * j + lane;
            if (st < B) {
              S av[6], jv[6];
#pragma unroll
              for (int k = 0; k < 6; ++k) {
                if constexpr (N == 1) { av[k] = ao[k]; jv[k] = jo[k]; }
                else { av[k] = j == 0 ? ao[k].x : ao[k].y; jv[k] = j == 0 ? jo[k].x : jo[k].y; }
              }
              if (acc_out) store6(acc_out, (long)r.orig6, Lf, st, av, out_vec);
              if (jw_out) store6(jw_out, (long)r.orig6, Lf, st, jv, out_vec);
            }
          }
        }
      } else {
        walk_step_rb<T, FLT, GEN>(c, W, St, s, r, rr, lane, fe);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) fe[k] = fn[k];
      if ((M.sfm[3] >> s) & 1) __syncthreads();  // SF_BW: a hand-off left its track at this step
    }
  }
  __syncthreads();
  RBD_WMARK(4);
  if (Lq.sk == 1 && Lv.sk == 1 && state0 + 64 * N <= B) {
    walk_stage_out_fast<T, 10 * N>(tau, state0, M.nv, c.rows, c.rt, tid, nth);
    walk_stage_out_fast<T, 10 * N>(qdot, state0, M.nq, c.rows, c.rq, tid, nth);
  } else if (Lq.sb == 1 && Lv.sb == 1 && Lq.sk == B && Lv.sk == B && state0 + 64 * N <= B) {
    walk_stage_out_rows<T, 10 * N>(tau, B, state0, M.nv, c.rows, c.rt, tid, nth);
    walk_stage_out_rows<T, 10 * N>(qdot, B, state0, M.nq, c.rows, c.rq, tid, nth);
  } else {
    walk_stage_out<T, 10>(tau, Lv, state0, B, M.nv, c.rows, c.rt, tid, nth);
    walk_stage_out<T, 10>(qdot, Lq, state0, B, M.nq, c.rows, c.rq, tid, nth);
  }
  RBD_WMARK(5);
}
#endif

}  // namespace rbd
