// rbd_device.hpp — device-side model view and spatial-algebra primitives for the gfx950 kernels.
//
// Execution model (kernels in rbd_kernels.hip): ONE LANE PER (state, body).  A state of a
// mechanism with n_bodies moving bodies occupies LPS = next_pow2(n_bodies) adjacent lanes of a
// 64-wide wavefront (Atlas: 31 bodies -> 32 lanes -> 2 states per wave).  Every per-body quantity
// of the Featherstone passes lives in that lane's VGPRs for the whole kernel; the tree sweeps are
// level-synchronous; the parent<->first-child hop is a DPP wave shift (bodies are in DFS pre-order),
// other children go through ds_bpermute — so the only HBM traffic is the algorithmic q/v/tau in,
// vdot out.  All quantities are expressed in the ROOT frame, as in the reference
// (src/mechanism_state.jl:744-748, :776, :842), so the backward sweeps are plain sums.
// A further mapping of the same sweeps: two bodies per lane (rbd_bank.hpp: BankModel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rbd_hip.h"  // RBD_JOINT_*

namespace rbd {

// per-body integer record
enum { IB_PARENT = 0, IB_JTYPE = 1, IB_QOFF = 2, IB_VOFF = 3, IB_LEVEL = 4, IB_NCHILD = 5, IB_ORIG = 6, IB_CHILD0 = 7, IB_MAXCHILD = 8, IB_FLAGS = IB_CHILD0 + IB_MAXCHILD /* BFD_* */, IB_STRIDE = 16 /* 64 bytes: a record is four 16-byte loads */ };
// A floating-base tree re-rooted at its centre (rbd_reroot.hpp): what the kernels need beyond the ordinary per-body records.
//   chain_i[k * 4] = joint type, q offset, v offset of the k-th joint on the way from the old floating body to the new root;
//   chain_r[k * 15] = its axis (3) and joint_to_predecessor R (9), p (3) — ORIGINAL constants, kernel scalar type
enum { BFD_VROOT = 1, BFD_FCARRY = 2, BFD_NOTFIRST = 4 /* banked records: the body is not the first child of its parent (its parent is not the previous lane) */ };
struct RerootView {
  int32_t nchain, fq, fv, _pad;
  const int32_t* chain_i;
  const void* chain_r;
  double fXp[12];  // the floating joint's joint_to_predecessor (R row-major, p)
};
// per-body real record: axis(3) axis2(3) XpR(9) Xpp(3) J(6: xx xy xz yy yz zz) mc(3) m(1)
enum { RB_AXIS = 0, RB_AXIS2 = 3, RB_XPR = 6, RB_XPP = 15, RB_J = 18, RB_MC = 24, RB_M = 27, RB_STRIDE = 28 };
enum { MAX_LEVELS = 64 };

// Bodies are renumbered on the host into DFS PRE-ORDER "slots": the first child of slot s is slot s+1, so the
// parent->first-child hop of every sweep is a DPP wave shift (v_mov_b32_dpp wave_shr:1 / wave_shl:1, VALU rate)
// instead of an LDS-crossbar ds_bpermute; only branch points (2nd, 3rd ... child) use ds_bpermute.
// Per-level child-slot counts (nslots of level l = max #children of the level l-1 bodies) travel in the kernel arguments as a STREAM of
// 4-bit fields in the order the bottom-up sweeps consume them — level nlevels-1 first — and are shifted out of a scalar register.
// (As a byte array indexed by the level, the compiler fetched nslots[l] from the kernel-argument segment inside every level of the sweep,
// with a full s_waitcnt behind it: a memory round trip per level on the critical path of every lane-per-body kernel.  Found in round 3.)
struct NsStream {
  uint64_t w;
  int j;
};
__host__ __device__ inline NsStream ns_begin(const uint64_t (&desc)[4]) { return NsStream{desc[0], 0}; }
__host__ __device__ inline int ns_next(NsStream& s, const uint64_t (&desc)[4]) {  // nslots of the next level down the stream
  if (s.j != 0 && (s.j & 15) == 0) s.w = desc[(s.j >> 4) & 3];  // every 16 levels (never for a humanoid)
  const int ns = (int)(s.w & 15);
  s.w >>= 4;
  ++s.j;
  return ns;
}
inline void nslots_pack_desc(uint64_t (&desc)[4], const int32_t* per_level, int nlevels) {
  desc[0] = desc[1] = desc[2] = desc[3] = 0;
  for (int l = nlevels - 1, j = 0; l >= 1 && j < 64; --l, ++j) desc[j >> 4] |= (uint64_t)(per_level[l] & 15) << ((j & 15) * 4);
}
struct DevModel {
  int32_t nb, nq, nv;
  int32_t lps;        // lanes per state (power of two, <= 64)
  int32_t nlevels;    // tree depth
  int32_t maxchild;   // max children of any body (<= IB_MAXCHILD)
  int32_t maxnvj;     // max velocity dimension of any tree joint
  int32_t has3dof;     // some tree joint is QuaternionSpherical / Planar (selects the NDOF = 3 aba_kernel instantiation)
  int32_t inner_floating;  // some 6-dof joint is not attached to the world (selects the general aba_kernel instantiation)
  int32_t debug_stop; // profiling aid (env RBD_ABA_STOP_AFTER): aba_kernel exits after phase 1..5 with a checksum store; 0 = full
  int32_t nheavy;     // quaternion joints (QuaternionFloating / QuaternionSpherical) of the tree, and for the first HEAVY_MAX of them:
  int32_t heavy[4][3];  // joint type, q offset, v offset — the integrator's one-thread-per-state stage reads them from the kernel arguments
  const int32_t* ib;  // [nb * IB_STRIDE], indexed by slot; parent/children are slots, IB_ORIG the reference body index
  const void* rb;     // [nb * RB_STRIDE] of the kernel's scalar type, indexed by slot
  const int32_t* dof_body;  // [nv] slot of velocity index
  const uint64_t* row_mask; // [nv * row_words] bit c % 64 of word c / 64 of row r: M[r, c] is structurally non-zero (joint of dof c supports the body of dof r), c <= r
  int32_t row_words;        // 64-bit words per row = ceil(nv / 64)
  const int32_t* anc;       // [nb * nlevels] anc[s*nlevels + k] = k-th ancestor slot of s (k=0: s itself), -1 past the root
  uint64_t perm_down;       // bit l set: some body at level l has parent slot != s-1 (top-down hop needs ds_bpermute at level l)
  uint64_t ns_desc[4];  // child slots to gather when the bottom-up sweep processes level l (max #children of level l-1 parents): NsStream
  double gravity[3];
};

// ---- banked lane-per-body ABA (aba_bank_kernel, rbd_bank.hpp): lane j of a state carries body j of bank 0 (levels < L0) and
// body j of bank 1 (levels >= L0).  ib[k] / rb[k] are the IB_* / RB_* records of bank k by bank slot (DFS pre-order within
// the bank); IB_PARENT is the parent's slot in the PARENT's bank, IB_CHILD* the children's slots in the CHILDREN's bank.
struct BankModel {
  int32_t lps, nlevels, L0;
  int32_t nbk[2];
  int32_t simple;              // every tree joint is revolute, apart from 6-dof joints on the world: selects the SIMPLE instantiation of aba_bank_kernel
  const int32_t* ib[2];
  const void* rb[2];
  uint64_t perm_down;          // bit l: the in-bank top-down hop at level l needs ds_bpermute (some parent is not the previous lane)
  uint64_t ns_desc[4];         // child slots to gather at level l: NsStream
  double gravity[3];
};

// ---- the track schedule with canonical body frames (the walk kernels, rbd_walk.hpp; plan: rbd_track_plan.hpp) ----------
// One packed int4 and one 24-scalar record per (step, track):
//   w0 = qoff | voff << 16;  w1 = 6*orig | flags << 16 | n_cross_children << 24;
//   w2 = (A/C mailbox to write + 1) | (A/C mailbox to read + 1) << 16;  w3 = (B mailbox to write + 1) | (first B mailbox to read + 1) << 16
//   reals: C (9, row-major) pp (3) J (6: xx xy xz yy yz zz) mc (3) m (1), all in the canonical (joint axis = +z) body frames
enum { TI_W0 = 0, TI_W1, TI_W2, TI_W3, TI_STRIDE = 4 };
enum { TR_C = 0, TR_PP = 9, TR_J = 12, TR_MC = 18, TR_M = 21, TR_STRIDE = 24 };
enum { TF_VALID = 1, TF_LEVEL0 = 2, TF_CHAINED = 4, TF_CARRY = 8, TF_FLOATING = 16, TF_PRISMATIC = 32, TF_FIXED = 64, TF_SINCOS = 128 };
enum { TMB_A = 30, TMB_B = 27, TMB_C = 6 };  // mailbox fields: (R, p, T, a_vp; a 6-dof root adds S⁻ᵀτ), (Ia, pa), (a)
struct TrackModel {
  int32_t ns, G, nA, nB;
  const int32_t* ri;  // [ns * G * TI_STRIDE]
  const void* rr;     // [ns * G * TR_STRIDE] of the kernel's scalar type
  uint64_t sfm[5];    // wave-uniform flags of the steps (SF_* bit k of step s = bit s of sfm[k], rbd_walk.hpp): which rare blocks any lane needs
  double gravity[3];
};

// ---- the track schedule with one wavefront per track and one lane per state (aba_walk_kernel, rbd_walk.hpp; rbd_walk_plan.hpp) ----
struct WalkModel {
  int32_t ns, G, nA, nB, nS, nq, nv;
  RerootView reroot;  // nchain > 0: the plan is that of the re-rooted tree (wk carries BFD_* in bits 8..9)
  const int32_t* ri;  // [ns * G * TI_STRIDE]  track-plan records
  const void* rr;     // [ns * G * TR_STRIDE]  ... constants, kernel scalar type
  const int32_t* wk;  // [ns * G]              parking slots (rbd_walk_plan.hpp)
  uint64_t sfm[5];    // wave-uniform step flags (SF_* of rbd_walk.hpp)
  double gravity[3];
};

// ---- one lane per state (crba_state_kernel / rnea_state_kernel, rbd_state.hpp; plan: rbd_state_plan.hpp) ------------------------
// The depth-first walk of the tree as a flat op list: ENTER(body) on the way down, EXIT(body) when its subtree is done.
//   op words: w0 = kind | level << 8 | joint type << 16, slot, q offset, v offset, 6 * reference body index
//   cols[op * SC_STRIDE + k]: velocity column of the op's body's ancestor at level k (-1: fixed joint; | SC_FLOATING / SC_SPHERICAL / SC_PLANAR: the first column of a 6- / 3-dof joint)
//   sr: the canonical-frame body constants of the track mapping (TR_* below), by op
enum { SO_W0 = 0, SO_SLOT, SO_QOFF, SO_VOFF, SO_ORIG6, SO_STRIDE = 8 };
enum { SK_ENTER = 0, SK_EXIT = 1 };
enum { SC_STRIDE = 16, SC_FLOATING = 0x10000, SC_SPHERICAL = 0x20000, SC_PLANAR = 0x40000, SC_MULTI = 0x70000 };  // (the last two: the plans of the compiled kernels only)
struct StateModel {
  int32_t nb, nq, nv, nops, nlevels;
  const int32_t* ops;
  const int32_t* cols;
  const void* sr;            // [nops * TR_STRIDE] of the kernel's scalar type
  const uint64_t* row_mask;  // DevModel::row_mask
  double gravity[3];
};

// ---- trees of more than 64 bodies (rbd_big_kernels.hip): one thread per state, per-body quantities in an HBM scratch ----------------------
//   tbl[4 * i] = parent, joint type, q offset, v offset of body i in the reference's order (parents first); rb: RB_* records in the same order
struct BigModel {
  int32_t nb, nq, nv, _pad;
  const int32_t* tbl;
  const void* rb;
  double gravity[3];
};

// ---- soft contact (contact_kernel, rbd_contact_kernels.hip): contact points grouped by body, half-spaces of the environment ----------
//   cp[point * CP_STRIDE]: location (3, body frame), Hunt–Crossley k, λ, n, viscoelastic Coulomb μ, k, b;  hs[halfspace * 6]: point, unit outward normal
enum { CP_LOC = 0, CP_HCK = 3, CP_HCL = 4, CP_HCN = 5, CP_MU = 6, CP_K = 7, CP_B = 8, CP_STRIDE = 9 };
struct ContactModel {
  int32_t nb, np, nh;
  const int32_t* cbody;  // [np] reference body index of each point
  const void* cp;        // [np * CP_STRIDE] of the kernel's scalar type
  const void* hs;        // [nh * 6]
};

// element (k, b) of an n x B batch buffer
struct Layout {
  long sk, sb;  // SOA: sk = B, sb = 1;  AOS: sk = 1, sb = n;  internal staging buffers: sk = 16, sb = -n (grouped by 16 states, below)
};
// offset of state b's column.  A staging buffer the library owns may be GROUPED BY 16 STATES — element (k, b) at (b / 16)(16 n) + 16 k + b % 16,
// encoded as Layout{16, -n}: a wavefront of the tile Cholesky (16 states) then reads ONE contiguous block, whole cache lines, and the
// one-lane-per-state CRBA that fills it still writes 64-byte runs (measured with the batch-innermost staging: 4x the bytes fetched, half of
// every line belonging to the neighbouring wavefront on another XCD's L2).
__host__ __device__ inline long layout_base(Layout L, long b) { return L.sb >= 0 ? b * L.sb : (b >> 4) * (16 * -L.sb) + (b & 15); }

// loop (non-tree) joint tables for loop_solve_kernel; indices are REFERENCE body indices (not slots)
template <typename T> struct LoopView {
  int nloops, nc, nv, nb;
  const int32_t* li;   // [nloops*8]: pred, succ, type, row0, ncl, path_begin, path_end, pad   (reference body indices)
  const T* lr;         // [nloops*64]: Xpred R(9) p(3), Xsucc R(9) p(3), gains(4), local wrench basis Tl (6 x up to 6 = 36)
  const int32_t* path; // [*][2]: reference body index, sign
  const int32_t* jt;   // [nb] joint type by reference body index
  const int32_t* voff; // [nb]
  const T* axis;       // [nb*3]
  const T* axis2;      // [nb*3] (Planar y axis)
  // loop_fused_small_kernel only: parent, q offset and DFS slot by reference body index, and the slot-ordered body constants (RB_*)
  const int32_t* xi;   // [nb*3]
  const T* rb;
  // per-call Baumgarte gains (rbd_workspace_set_loop_gains): record l's four gains at gains[64 l + 0..3]; nullptr: the ones in lr (the model's).
  // Only the kernels compiled against constant tables need it — the workspace's copy of lr is patched in place for the others.
  const T* gains = nullptr;
};

// integrator scratch of mk_stage_kernel (same layout as the caller's q / v)
// fused-integrator descriptor handed to aba_kernel (stage < 0: plain dynamics!)
struct MkBuffers;
struct MkBuffers {
  void* q0; void* v0;   // base point of the step, AOS-agnostic: same layout as q / v
  void* phid[4]; void* vd[4];
};

struct MkFuse {
  int32_t stage;  // 0..3, or -1
  int32_t close_prev;  // with stage 0: first close the PREVIOUS step (stage 4: combine its four stages into the new q, v) — saves a launch per step
  double dt;
  MkBuffers W;
  void* q_state; void* v_state;
  // device-side PD controller (rbd_simulate_controlled, RBD_CONTROL_PD): tau -= kp (q − q_des) + kd v on the 1-dof joints, on the stage state
  const void* pd_kp; const void* pd_kd; const void* pd_qdes;  // [nv], [nv], nq x B (nullable); pd_kp == nullptr: off
};

#define RBD_DEV __device__ __forceinline__
// pure arithmetic helpers are host+device: the CPU emulation harness of the track kernel (tests/emu) runs the same step code
#define RBD_HD __host__ __device__ __forceinline__

template <typename T> RBD_DEV T shfl(T x, int src) { return __shfl(x, src, 64); }

// two fp32 states per lane (aba_walk_kernel); reciprocals of well-scaled positive numbers (defined below)
typedef float f2 __attribute__((ext_vector_type(2)));
RBD_HD double rcp_hd(double x);
RBD_HD float rcp_hd(float x);
RBD_HD f2 rcp_hd(f2 x);

// DPP wave shifts (gfx9 DPP_WF_SR1 = 0x138: lane i <- lane i-1; DPP_WF_SL1 = 0x130: lane i <- lane i+1; lanes shifted in read 0)
template <int CTRL> RBD_DEV float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int CTRL> RBD_DEV double dpp_mov(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <typename T> RBD_DEV T from_prev_lane(T x) { return dpp_mov<0x138>(x); }
template <typename T> RBD_DEV T from_next_lane(T x) { return dpp_mov<0x130>(x); }

// x where the mask is set, exactly 0 elsewhere.  The sweeps used to multiply a neighbour lane's value by a 0/1 mask; with several
// states in one wavefront the neighbour can belong to ANOTHER state, and NaN * 0 = NaN let one diverged state poison the state packed
// next to it.  A select keeps the states of a batch independent, as they are in the reference (one evaluation per call).
template <typename T> RBD_HD T keep(T x, T mask) { return mask != T(0) ? x : T(0); }
// An `if` whose body must stay a branch (exec mask) and not be turned into selects by if-conversion: what a lane TAKES from another lane is added
// under the lane's own exec mask — `if (takes) { RBD_KEEP_BRANCH(); acc += moved; }` — which keeps the states of a wavefront independent (no
// arithmetic ever touches a value the lane does not own) without the two v_cndmask per fp64 value that keep() costs (round 3).
#if defined(__HIP_DEVICE_COMPILE__)
#define RBD_KEEP_BRANCH() asm volatile("" ::)
#else
#define RBD_KEEP_BRANCH()
#endif

template <typename T> RBD_HD void cross3(const T* a, const T* b, T* o) {
  T x = a[1] * b[2] - a[2] * b[1];
  T y = a[2] * b[0] - a[0] * b[2];
  T z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
template <typename T> RBD_HD void matvec3(const T* R, const T* x, T* o) {
  T a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  T b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  T c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
template <typename T> RBD_HD void matTvec3(const T* R, const T* x, T* o) {
  T a = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  T b = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  T c = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
template <typename T> RBD_HD void matmul3(const T* A, const T* B, T* C) {
  T t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int k = 0; k < 9; ++k) C[k] = t[k];
}
// transform_spatial_motion (src/spatial/util.jl:104-108): (Rw, Rv + p x Rw)
template <typename T> RBD_HD void xmotion(const T* R, const T* p, const T* m, T* o) {
  T a[3], l[3], c[3];
  matvec3(R, m, a);
  matvec3(R, m + 3, l);
  cross3(p, a, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a[k]; o[3 + k] = l[k] + c[k]; }
}
// inverse of xmotion: (R'w, R'(v - p x w))
template <typename T> RBD_HD void xmotion_inv(const T* R, const T* p, const T* m, T* o) {
  T c[3], d[3];
  cross3(p, m, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = m[3 + k] - c[k];
  matTvec3(R, m, o);
  matTvec3(R, d, o + 3);
}
// wrench transform (src/spatial/spatialforce.jl:152-158): (Rt + p x Rf, Rf)
template <typename T> RBD_HD void xforce(const T* R, const T* p, const T* w, T* o) {
  T a[3], l[3], c[3];
  matvec3(R, w, a);
  matvec3(R, w + 3, l);
  cross3(p, l, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a[k] + c[k]; o[3 + k] = l[k]; }
}
// inverse wrench transform = S' w for a floating joint: (R'(t - p x f), R'f)
template <typename T> RBD_HD void xforce_inv(const T* R, const T* p, const T* w, T* o) {
  T c[3], d[3];
  cross3(p, w + 3, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = w[k] - c[k];
  matTvec3(R, d, o);
  matTvec3(R, w + 3, o + 3);
}
// se3_commutator (src/spatial/util.jl:117-121)
template <typename T> RBD_HD void se3_comm(const T* x, const T* y, T* o) {
  T a[3], b[3], c[3];
  cross3(x, y, a);
  cross3(x, y + 3, b);
  cross3(x + 3, y, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a[k]; o[3 + k] = b[k] + c[k]; }
}
template <typename T> RBD_HD T dot6(const T* a, const T* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

// rigid-body inertia in the root frame: J (6 unique: xx xy xz yy yz zz), c = m*com, m
template <typename T> struct RInertia { T J[6]; T c[3]; T m; };

// mul_inertia (src/spatial/util.jl:110-114): (J w + c x v, m v - c x w)
template <typename T> RBD_HD void mul_inertia(const RInertia<T>& I, const T* t, T* o) {
  const T* J = I.J;
  T b[3], d[3];
  cross3(I.c, t + 3, b);
  cross3(I.c, t, d);
  o[0] = J[0] * t[0] + J[1] * t[1] + J[2] * t[2] + b[0];
  o[1] = J[1] * t[0] + J[3] * t[1] + J[4] * t[2] + b[1];
  o[2] = J[2] * t[0] + J[4] * t[1] + J[5] * t[2] + b[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) o[3 + k] = I.m * t[3 + k] - d[k];
}
// transform(inertia, H) (src/spatial/motion_force_interaction.jl:160-176)
template <typename T> RBD_HD void inertia_to_root(const T* Jb /*6*/, const T* mcb, T m, const T* R, const T* p, RInertia<T>& O) {
  T Rmc[3], mp[3];
  matvec3(R, mcb, Rmc);
#pragma unroll
  for (int k = 0; k < 3; ++k) mp[k] = m * p[k];
  // Y = X + X' + mp p',  X = Rmc p'  (symmetric)
  T Y[6];
  Y[0] = 2 * Rmc[0] * p[0] + mp[0] * p[0];
  Y[1] = Rmc[0] * p[1] + Rmc[1] * p[0] + mp[0] * p[1];
  Y[2] = Rmc[0] * p[2] + Rmc[2] * p[0] + mp[0] * p[2];
  Y[3] = 2 * Rmc[1] * p[1] + mp[1] * p[1];
  Y[4] = Rmc[1] * p[2] + Rmc[2] * p[1] + mp[1] * p[2];
  Y[5] = 2 * Rmc[2] * p[2] + mp[2] * p[2];
  T trY = Y[0] + Y[3] + Y[5];
  // R J R'
  T Jf[9] = {Jb[0], Jb[1], Jb[2], Jb[1], Jb[3], Jb[4], Jb[2], Jb[4], Jb[5]};
  T RJ[9];
  matmul3(R, Jf, RJ);
  // (RJ) R' upper triangle: sum_k RJ[i][k] R[j][k]
  T A[6];
  A[0] = RJ[0] * R[0] + RJ[1] * R[1] + RJ[2] * R[2];
  A[1] = RJ[0] * R[3] + RJ[1] * R[4] + RJ[2] * R[5];
  A[2] = RJ[0] * R[6] + RJ[1] * R[7] + RJ[2] * R[8];
  A[3] = RJ[3] * R[3] + RJ[4] * R[4] + RJ[5] * R[5];
  A[4] = RJ[3] * R[6] + RJ[4] * R[7] + RJ[5] * R[8];
  A[5] = RJ[6] * R[6] + RJ[7] * R[7] + RJ[8] * R[8];
  O.J[0] = A[0] - Y[0] + trY; O.J[1] = A[1] - Y[1]; O.J[2] = A[2] - Y[2];
  O.J[3] = A[3] - Y[3] + trY; O.J[4] = A[4] - Y[4]; O.J[5] = A[5] - Y[5] + trY;
#pragma unroll
  for (int k = 0; k < 3; ++k) O.c[k] = Rmc[k] + mp[k];
  O.m = m;
}
// newton_euler cross term: T x* (I T) = (w x k + v x l ; w x l), (k, l) = I T
template <typename T> RBD_HD void momentum_cross(const RInertia<T>& I, const T* t, T* o) {
  T h[6], a[3], b[3], c[3];
  mul_inertia(I, t, h);
  cross3(t, h, a);
  cross3(t + 3, h + 3, b);
  cross3(t, h + 3, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o[k] = a[k] + b[k]; o[3 + k] = c[k]; }
}

// ---- packed symmetric 6x6 (upper, row-major): index of (i, j), i <= j ----
__host__ __device__ constexpr int SI(int i, int j) { return i <= j ? (i * 6 - i * (i - 1) / 2 + (j - i)) : (j * 6 - j * (j - 1) / 2 + (i - j)); }

template <typename T> RBD_HD void sym6_from_inertia(const RInertia<T>& I, T* A /*21*/) {
  const T z = T(0);
  const T* c = I.c;
  A[SI(0, 0)] = I.J[0]; A[SI(0, 1)] = I.J[1]; A[SI(0, 2)] = I.J[2]; A[SI(0, 3)] = z;     A[SI(0, 4)] = -c[2]; A[SI(0, 5)] = c[1];
  A[SI(1, 1)] = I.J[3]; A[SI(1, 2)] = I.J[4]; A[SI(1, 3)] = c[2];   A[SI(1, 4)] = z;     A[SI(1, 5)] = -c[0];
  A[SI(2, 2)] = I.J[5]; A[SI(2, 3)] = -c[1];  A[SI(2, 4)] = c[0];   A[SI(2, 5)] = z;
  A[SI(3, 3)] = I.m;    A[SI(3, 4)] = z;      A[SI(3, 5)] = z;
  A[SI(4, 4)] = I.m;    A[SI(4, 5)] = z;
  A[SI(5, 5)] = I.m;
}
template <typename T> RBD_HD void sym6_mul(const T* A, const T* x, T* o) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    T s = T(0);
#pragma unroll
    for (int j = 0; j < 6; ++j) s += A[SI(i, j)] * x[j];
    o[i] = s;
  }
}
// in-place LDL' of a packed SPD 6x6 (A = L D L', unit lower L stored in the strict part as A[SI(j,i)], D on the diagonal)
// and solve A x = b.  No square roots; the pivots are 1/d.
template <typename T> RBD_HD void sym6_solve(T* A, const T* b, T* x) {
  T dinv[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    T d = A[SI(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[SI(k, j)] * A[SI(k, j)] * A[SI(k, k)];
    A[SI(j, j)] = d;
    dinv[j] = rcp_hd(d);  // hardware estimate + Newton steps: the IEEE division expansion is ~5x the instructions, and the pivots of a
                          // physical articulated inertia are well scaled
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      T s = A[SI(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[SI(k, i)] * A[SI(k, j)] * A[SI(k, k)];
      A[SI(j, i)] = s * dinv[j];  // L[i][j]
    }
  }
  T y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    T s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= A[SI(k, i)] * y[k];
    y[i] = s;
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    T s = y[i] * dinv[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= A[SI(i, k)] * x[k];
    x[i] = s;
  }
}

// AngleAxis -> R given sin, cos (formula of src/joint_types/sin_cos_revolute.jl:69-96)
template <typename T> RBD_HD void rot_axis_sc(const T* ax, T s, T c, T* R) {
  T c1 = T(1) - c;
  T c1x2 = c1 * ax[0] * ax[0], c1y2 = c1 * ax[1] * ax[1], c1z2 = c1 * ax[2] * ax[2];
  T c1xy = c1 * ax[0] * ax[1], c1xz = c1 * ax[0] * ax[2], c1yz = c1 * ax[1] * ax[2];
  T sx = s * ax[0], sy = s * ax[1], sz = s * ax[2];
  R[0] = T(1) - c1y2 - c1z2; R[3] = c1xy + sz;          R[6] = c1xz - sy;
  R[1] = c1xy - sz;          R[4] = T(1) - c1x2 - c1z2; R[7] = c1yz + sx;
  R[2] = c1xz + sy;          R[5] = c1yz - sx;          R[8] = T(1) - c1x2 - c1y2;
}
// unit quaternion (w,x,y,z) -> R (Rotations.jl QuatRotation, normalize=false)
template <typename T> RBD_HD void rot_quat(T w, T x, T y, T z, T* R) {
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// reciprocal of a well-scaled positive number (joint-space inertia D): hardware estimate + 2 Newton steps (the IEEE
// division expansion costs ~3x as many instructions; D is never denormal/huge for a physical mechanism)
RBD_DEV double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return r;
}
RBD_DEV float rcp_nr(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
}
// 1/sqrt of a positive, well-scaled number: hardware estimate + Newton steps (the IEEE sqrt + division pair is ~70 instructions in fp64)
RBD_DEV double rsqrt_nr(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = r * __builtin_fma(-h * r, r, 1.5);
  r = r * __builtin_fma(-h * r, r, 1.5);
  return r;
}
RBD_DEV float rsqrt_nr(float x) {
  float r = __builtin_amdgcn_rsqf(x);
  return r * __builtin_fmaf(-0.5f * x * r, r, 1.5f);
}
template <typename T> struct SqrtT;
template <> struct SqrtT<double> { static __device__ __forceinline__ double f(double x) { return sqrt(x); } };
template <> struct SqrtT<float> { static __device__ __forceinline__ float f(float x) { return sqrtf(x); } };

RBD_DEV void sincos_t(double x, double* s, double* c) { sincos(x, s, c); }
RBD_DEV void sincos_t(float x, float* s, float* c) { sincosf(x, s, c); }

// host+device forms for code that also runs in the CPU emulation harness (the host branch is never part of the product path)
RBD_HD double rcp_hd(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rcp_nr(x);
#else
  return 1.0 / x;
#endif
}
RBD_HD float rcp_hd(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rcp_nr(x);
#else
  return 1.0f / x;
#endif
}
RBD_HD void sincos_hd(double x, double* s, double* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(x, s, c);
#else
  *s = __builtin_sin(x); *c = __builtin_cos(x);
#endif
}
// sin and cos of a joint angle in ~35 fp64 instructions (the library sincos is ~130 with its large-argument machinery inlined):
// Cody-Waite reduction by pi/2 in three FMA steps (exact products), the fdlibm kernel polynomials on [-pi/4, pi/4], quadrant fix-up.
// |x| <= 2^15 (k < 2^15: the three-term reduction keeps the reduced argument good to ~1e-17 absolute); larger or non-finite
// arguments take the library path.  Max error observed against libm on 10^7 points in [-2^15, 2^15]: 1.6 ulp.
RBD_HD void sincos_fast(double x, double* sp, double* cp) {
  if (!(__builtin_fabs(x) <= 32768.0)) { sincos_hd(x, sp, cp); return; }
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(k, -1.57079632679489655800e+00, x);
  r = __builtin_fma(k, -6.12323399573676603587e-17, r);
  r = __builtin_fma(k, 1.49738490485916983294e-33, r);
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double sn = __builtin_fma(r * z, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const int n = (int)k;
  const double a = (n & 1) ? cs : sn, b = (n & 1) ? sn : cs;
  *sp = (n & 2) ? -a : a;
  *cp = ((n + 1) & 2) ? -b : b;
}
RBD_HD void sincos_fast(float x, float* sp, float* cp);

RBD_HD void sincos_hd(float x, float* s, float* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincosf(x, s, c);
#else
  *s = __builtin_sinf(x); *c = __builtin_cosf(x);
#endif
}
// fp32: the same scheme in single precision (~22 instructions against ~50-60 of sincosf with its large-argument path inlined): two-term Cody-Waite
// reduction by pi/2 with FMAs, the cephes minimax polynomials on [-pi/4, pi/4] (< 1 ulp there), quadrant fix-up.  |x| <= 8192 (k < 2^13: the residual
// of the two-term constant, ~1e-15 k, stays below 1e-11); beyond that and for non-finite arguments the library.  Max error observed against libm
// (fp64 reference) on 10^7 points in [-8192, 8192]: 9.3e-8 absolute.
RBD_HD void sincos_fast(float x, float* sp, float* cp) {
#ifndef RBD_JIT_COMPILE  // (the run-time compiled kernels are straight-line code with a sin/cos per joint and pass: beyond 8192 rad, where one fp32 ulp
                         //  of the ANGLE is 1e-3 rad, they take the two-term reduction's error instead of a copy of the library path at every site)
  if (!(__builtin_fabsf(x) <= 8192.0f)) { sincos_hd(x, sp, cp); return; }
#endif
  const float k = __builtin_rintf(x * 6.36619772e-01f);
  float r = __builtin_fmaf(k, -1.57079637e+00f, x);
  r = __builtin_fmaf(k, 4.37113883e-08f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
  const float sn = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
  const float cs = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
  const int n = (int)k;
  const float a = (n & 1) ? cs : sn, b = (n & 1) ? sn : cs;
  *sp = (n & 2) ? -a : a;
  *cp = ((n + 1) & 2) ? -b : b;
}

// ---- two fp32 states per lane: every arithmetic instruction becomes a packed v_pk_{fma,mul,add}_f32 (aba_walk_kernel, rbd_walk.hpp).
// A clang extended vector: + - * / act element-wise, a scalar operand is splat, T(x) converts and splats.
RBD_HD f2 rcp_hd(f2 x) { f2 r; r.x = rcp_hd(x.x); r.y = rcp_hd(x.y); return r; }
// two states per lane: the polynomial part as packed arithmetic (v_pk_fma_f32), the quadrant logic per component
RBD_HD void sincos_fast(f2 x, f2* s, f2* c) {
#ifndef RBD_JIT_COMPILE  // (as in the one-state form above: the kernels compiled at run time carry no copy of the library path)
  if (!(__builtin_fabsf(x.x) <= 8192.0f && __builtin_fabsf(x.y) <= 8192.0f)) {
    float s0, c0, s1, c1;
    sincos_hd(x.x, &s0, &c0);
    sincos_hd(x.y, &s1, &c1);
    s->x = s0; s->y = s1; c->x = c0; c->y = c1;
    return;
  }
#endif
  f2 k;
  k.x = __builtin_rintf(x.x * 6.36619772e-01f); k.y = __builtin_rintf(x.y * 6.36619772e-01f);
  f2 r = k * f2(-1.57079637e+00f) + x;
  r = k * f2(4.37113883e-08f) + r;
  const f2 z = r * r;
  f2 ps = z * f2(-1.9515295891e-4f) + f2(8.3321608736e-3f);
  ps = z * ps + f2(-1.6666654611e-1f);
  const f2 sn = (r * z) * ps + r;
  f2 pc = z * f2(2.443315711809948e-5f) + f2(-1.388731625493765e-3f);
  pc = z * pc + f2(4.166664568298827e-2f);
  const f2 cs = (z * z) * pc + (z * f2(-0.5f) + f2(1.0f));
  const int n0 = (int)k.x, n1 = (int)k.y;
  const float a0 = (n0 & 1) ? cs.x : sn.x, b0 = (n0 & 1) ? sn.x : cs.x, a1 = (n1 & 1) ? cs.y : sn.y, b1 = (n1 & 1) ? sn.y : cs.y;
  s->x = (n0 & 2) ? -a0 : a0; s->y = (n1 & 2) ? -a1 : a1;
  c->x = ((n0 + 1) & 2) ? -b0 : b0; c->y = ((n1 + 1) & 2) ? -b1 : b1;
}
// Six consecutive values of a per-body output (accelerations, joint wrenches: 6 n_bodies x B buffers) of one state.  A state-major caller (sk == 1) keeps
// them contiguous — 48 bytes in fp64, 24 in fp32, starting at a multiple of 16 / 8 bytes when the buffer and the state stride are so aligned (`vec`,
// wave-uniform, store6_vec below): three 16- or 8-byte stores instead of six scalar ones (the per-body outputs are store-bound: DESIGN.md §8).
RBD_HD bool store6_vec(const void* out, Layout L, int esize) {
  return out != nullptr && L.sk == 1 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ((L.sb * esize) & 15) == 0;
}
RBD_DEV void store6(double* __restrict__ out, long i0, Layout L, long state, const double* x, bool vec) {
  if (vec) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2* p = reinterpret_cast<d2*>(out + i0 + state * L.sb);
    p[0] = d2{x[0], x[1]}; p[1] = d2{x[2], x[3]}; p[2] = d2{x[4], x[5]};
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) out[(i0 + k) * L.sk + state * L.sb] = x[k];
  }
}
RBD_DEV void store6(float* __restrict__ out, long i0, Layout L, long state, const float* x, bool vec) {
  if (vec) {
    f2* p = reinterpret_cast<f2*>(out + i0 + state * L.sb);  // (i0 = 6 body: a multiple of 8 bytes)
    p[0] = f2{x[0], x[1]}; p[1] = f2{x[2], x[3]}; p[2] = f2{x[4], x[5]};
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) out[(i0 + k) * L.sk + state * L.sb] = x[k];
  }
}

// scalar type and states per lane of a kernel value type
template <typename T> struct Lanes { using S = T; enum { N = 1 }; };
template <> struct Lanes<f2> { using S = float; enum { N = 2 }; };

// ---- the floating base of a re-rooted tree (rbd_reroot.hpp) ------------------------------------------------------------------------
// pose of the OLD floating body in the world from its coordinates: H = joint_to_predecessor ∘ (R(quat), trans)  (quaternion_floating.jl:81-83)
template <typename T> RBD_HD void reroot_fb_pose(const RerootView& V, const T* q7, T* R, T* p) {
  T Rq[9], X[9], t3[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) X[k] = T(V.fXp[k]);
  rot_quat(q7[0], q7[1], q7[2], q7[3], Rq);
  matmul3(X, Rq, R);
  matvec3(X, q7 + 4, t3);
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = t3[k] + T(V.fXp[9 + k]);
}
// pose and twist of the NEW root: walk the chain of original joints from the old floating body.  q7 / v6: the floating joint's coordinates;
// cq[j] / cv[j]: coordinate and velocity of the j-th chain joint (the caller fetched them with its first batch of loads — a dependent
// global round trip here would sit on the critical path of the launch).  S: the scalar type of the chain table.
enum { RC_MAX = 4 };  // longest chain the kernels take (the host does not re-root beyond it)
template <typename T, typename S>
RBD_HD void reroot_root_kinematics(const RerootView& V, const T* q7, const T* v6, const T* cq, const T* cv, T* R, T* p, T* Tw) {
  reroot_fb_pose(V, q7, R, p);
  xmotion(R, p, v6, Tw);
  const S* cr = reinterpret_cast<const S*>(V.chain_r);
#pragma unroll
  for (int j = 0; j < RC_MAX; ++j) {
    if (j < V.nchain) {
      const int jt = V.chain_i[4 * j];
      const S* c = cr + 15 * j;
      const T ax[3] = {T(c[0]), T(c[1]), T(c[2])};
      T XR[9], Xp[3], Rj[9], pj[3] = {T(0), T(0), T(0)}, tl[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int k = 0; k < 9; ++k) { XR[k] = T(c[3 + k]); Rj[k] = (k % 4 == 0) ? T(1) : T(0); }
#pragma unroll
      for (int k = 0; k < 3; ++k) Xp[k] = T(c[12 + k]);
      if (jt == RBD_JOINT_REVOLUTE) {
        T sn, cs;
        sincos_fast(cq[j], &sn, &cs);
        rot_axis_sc(ax, sn, cs, Rj);
#pragma unroll
        for (int k = 0; k < 3; ++k) tl[k] = ax[k] * cv[j];
      } else if (jt == RBD_JOINT_PRISMATIC) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { pj[k] = ax[k] * cq[j]; tl[3 + k] = ax[k] * cv[j]; }
      }
      // H <- H ∘ (XR, Xp) ∘ (Rj, pj)
      T A[9], a3[3], b3[3], vJ[6];
      matmul3(R, XR, A);
      matvec3(R, Xp, a3);
      matvec3(A, pj, b3);
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] += a3[k] + b3[k];
      matmul3(A, Rj, R);
      xmotion(R, p, tl, vJ);  // the joint's twist is given in its frame_after = the body frame just reached
#pragma unroll
      for (int k = 0; k < 6; ++k) Tw[k] += vJ[k];
    }
  }
}
// pose of the old floating body in its ORIGINAL frame from its pose in the re-based frame: H = H' ∘ E⁻¹, E = joint_to_predecessor of the
// first chain joint (chain_r[3..14])
template <typename T, typename S> RBD_HD void reroot_fb_pose_from_rebased(const RerootView& V, const T* Rn, const T* pn, T* R, T* p) {
  const S* c = reinterpret_cast<const S*>(V.chain_r);
  T E[9], Ep[3], t3[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) E[k] = T(c[3 + k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) Ep[k] = T(c[12 + k]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[3 * i + j] = Rn[3 * i] * E[3 * j] + Rn[3 * i + 1] * E[3 * j + 1] + Rn[3 * i + 2] * E[3 * j + 2];  // Rn E'
  matvec3(R, Ep, t3);
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = pn[k] - t3[k];
}

}  // namespace rbd
