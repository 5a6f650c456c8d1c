// rbd_track.hpp — track-scheduled fused articulated-body algorithm with canonical body frames (aba_track_kernel).
//
// Same result as the reference's dynamics! (src/mechanism_algorithms.jl:845-864: v̇ = M⁻¹(τ − c), q̇) and as the other
// mappings (aba_kernel, aba_bank_kernel, aba_chain_kernel); every quantity in the ROOT frame like the reference
// (src/mechanism_state.jl:744-748, :776, :842).  What differs is where the work runs and how much of it there is:
//
//   * a state owns G lanes ("tracks", 4 for a humanoid → 16 states per wavefront); the host cuts the tree into chains and
//     packs them on the tracks (rbd_track_plan.hpp); every track walks one body per step with the recursion state of the
//     three passes in registers — ~70 % of the lanes do useful work at every step, against ~15 % for the level-synchronous
//     lane-per-body sweeps;
//   * canonical body frames (axis of every 1-dof joint = +z, folded into the constants on the host): the joint rotation mixes
//     two columns, the motion subspace is (z-column of R; p × z-column) — no axis arithmetic, no joint-type interpreter;
//   * the velocity-product acceleration a_vp (a_vp,b = a_vp,parent + [T_parent, vJ], a_vp,world = −g) is carried down with the
//     kinematics and folded into the bias force, p̃A = I a_vp + T ×* I T − w_ext, so the articulated recursion has no bias
//     term: Ia = IA − U D⁻¹ U', p̃a = p̃A + U D⁻¹ u, and the last pass is v̇ = D⁻¹u − (U D⁻¹)' a_Δ,parent, a_Δ = a_Δ,parent + S v̇;
//   * EVERYTHING ON CHIP AFTER A STAGING PROLOGUE.  The plan records go to LDS once per workgroup; the inputs of every (step, lane)
//     — sin/cos of the joint angle, joint velocity, torque, external wrench — are loaded, converted and written to that lane's
//     LDS row before the passes start.  The three passes then touch global memory only to store v̇ / q̇.  (Measured: with the
//     inputs prefetched from global memory inside the step loops the compiler's s_waitcnt placement collapsed to vmcnt(0) at
//     every step — conditional loads make its in-order counter model conservative — and each of the 3·ns steps paid a full
//     memory round trip: 2 300 cycles per step for ~450 cycles of arithmetic.)
//   * per-body results of pass A (root-frame inertia, S, p̃A) and pass B (U D⁻¹, D⁻¹u) live in the same lane-private LDS row
//     (16-byte ds accesses, conflict-free); edges between tracks go through small LDS mailboxes; the next step's row and records
//     are read while the current step computes (rare mailbox reads first: the LDS counter is in order);
//   * rare work (mailboxes, level-0 initialisation, the 6-dof root) sits behind wave-uniform per-step flags held in scalar
//     registers: a wavefront alone on its SIMD pays ~35 cycles for every exec-masked block;
//   * NW = 4 ("latency form", small batches): a workgroup is four wavefronts over the SAME 16 states.  Wave 0 runs only what is
//     sequential along the tree — the kinematic chain of pass A, then passes B and C; waves 1–3 take the per-body work that is not
//     on that critical path (root-frame inertia, motion subspace, bias force), row by row, as soon as wave 0 publishes a body's
//     kinematics in its LDS row.  The waves meet through LDS flags.  NW = 1 does everything in one wavefront.
//
// The step functions are __host__ __device__: tests/emu runs the very same code lane by lane on the CPU.
#pragma once
#include "rbd_device.hpp"

namespace rbd {

template <typename T> struct alignas(16) Vec16 { T e[16 / sizeof(T)]; };
struct alignas(16) I4 { int32_t x, y, z, w; };

template <typename T> struct TrackDims {
  static constexpr int VW = 16 / (int)sizeof(T);           // scalars per 16-byte vector
  static constexpr int NF = sizeof(T) == 8 ? 22 : 24;      // fields per LDS row (whole vectors)
  static constexpr int NV = NF / VW;                       // vectors per row
  static constexpr int NVR = TR_STRIDE / VW;               // vectors per constant record
};
// LDS row of (step, lane), by phase:
//   input (staging -> pass A):   1-dof joint: 0 sin q · 1 cos q · 2 q̇ · 3 q (prismatic displacement) · [4,10) w_ext · 21 τ
//                                6-dof joint: [0,7) q · [7,13) v · [13,19) τ
//   kinematics (NW = 4: wave 0 -> helper): [0,6) columns 0 and 1 of R (column 2 = their cross product) · [6,9) p · [9,15) T · [15,21) a_vp · 21 τ
//   body (pass A -> pass B):     [0,6) J · [6,9) c · [9,15) S · [15,21) p̃A · 21 τ        (the mass comes from the plan record)
//   joint (pass B -> pass C):    [0,6) U D⁻¹ (or a_Δ of a 6-dof root) · 6 D⁻¹u · [9,15) S
enum { TS_W = 0, TS_UD = 6, TS_S = 9, TS_PA = 15, TS_TAU = 21 };
enum { TN_SN = 0, TN_CS = 1, TN_QD = 2, TN_Q = 3, TN_FE = 4, TN_FQ = 0, TN_FV = 7, TN_FT = 13 };
enum { TK_R01 = 0, TK_P = 6, TK_T = 9, TK_AV = 15 };
// wave-uniform per-step flags (bit k of step s = bit s of TrackModel::sfm[k])
enum { SF_NOCHAIN = 1, SF_AW = 2, SF_BR = 4, SF_BW = 8, SF_FLOAT = 16 };

template <typename T> struct TrackCtx {
  TrackModel M;
  long B;
  const T* q; const T* v; const T* tau; const T* fext;
  T* vdot; T* qdot;
  Layout Lq, Lv, Lf;
  uint32_t qsk, vsk, fsk;  // byte strides between coordinates (the launcher checks that every buffer is below 4 GB: 32-bit offsets from a scalar base)
  Vec16<T>* rows;          // [ns][NV][64]
  I4* tri;                 // [ns * G]        the plan's packed records, copied to LDS by the prologue
  Vec16<T>* trr;           // [ns * G][NVR]   ... and its constants
  T* mbA; T* mbB; T* mbC;  // [slot][field][SPW]
  int* flags;              // NW = 4: [0, ns) kinematics of row ready · [ns, ns + 4) helper finished
  int SPW;
  T a0[6];
};

template <typename T> RBD_HD size_t track_lds_bytes(const TrackModel& M) {
  const size_t spw = 64 / M.G, nrec = (size_t)M.ns * M.G;
  return (size_t)M.ns * TrackDims<T>::NV * 64 * 16 + nrec * 16 + nrec * TR_STRIDE * sizeof(T) +
         ((size_t)M.nA * (TMB_A + TMB_C) + (size_t)M.nB * TMB_B) * spw * sizeof(T) + ((size_t)M.ns + 4) * sizeof(int);
}

template <typename T> RBD_HD void track_ctx_lds(TrackCtx<T>& c, void* lds) {
  const int spw = 64 / c.M.G;
  const long nrec = (long)c.M.ns * c.M.G;
  c.SPW = spw;
  c.rows = reinterpret_cast<Vec16<T>*>(lds);
  c.tri = reinterpret_cast<I4*>(c.rows + (long)c.M.ns * TrackDims<T>::NV * 64);
  c.trr = reinterpret_cast<Vec16<T>*>(c.tri + nrec);
  c.mbA = reinterpret_cast<T*>(c.trr + nrec * TrackDims<T>::NVR);
  c.mbB = c.mbA + (long)c.M.nA * TMB_A * spw;
  c.mbC = c.mbB + (long)c.M.nB * TMB_B * spw;
  c.flags = reinterpret_cast<int*>(c.mbC + (long)c.M.nA * TMB_C * spw);
  c.qsk = (uint32_t)(c.Lq.sk * (long)sizeof(T)); c.vsk = (uint32_t)(c.Lv.sk * (long)sizeof(T)); c.fsk = (uint32_t)(c.Lf.sk * (long)sizeof(T));
#pragma unroll
  for (int k = 0; k < 3; ++k) { c.a0[k] = T(0); c.a0[3 + k] = T(-c.M.gravity[k]); }  // a_world = −gravity (mechanism_algorithms.jl:405)
}

// what every lane of every wave knows about itself
struct TrackWho {
  int lane, g, sl;
  bool vstate;
  uint32_t qb, vb, fb;  // byte offsets of this lane's state in the q / v (tau, vdot) / fext buffers
};
template <typename T, int G> RBD_HD void track_who(const TrackCtx<T>& c, TrackWho& W, int lane, long group) {
  constexpr int SPW = 64 / G;
  W.lane = lane; W.g = lane / SPW; W.sl = lane % SPW;
  const long state = group * SPW + W.sl;
  W.vstate = state < c.B;
  const long st = W.vstate ? state : 0;  // lanes past the batch load state 0 and store nothing
  W.qb = (uint32_t)(st * c.Lq.sb * (long)sizeof(T)); W.vb = (uint32_t)(st * c.Lv.sb * (long)sizeof(T)); W.fb = (uint32_t)(st * c.Lf.sb * (long)sizeof(T));
}

template <typename T> struct TrackLane {
  TrackWho who;
  T R[9], p[3], Tw[6], av[6];  // pass A: transform to root, twist, velocity-product acceleration of the body last visited
  T cI[21], cP[6];             // pass B: hand-off of the body last visited (zero at the leaf of a chain)
  T ad[6];                     // pass C: a_Δ of the body last visited
  // software pipeline: the record, constants and LDS row of the next step are read into the other buffer while this step computes
  struct Buf { I4 ri; T rr[TR_STRIDE]; T row[TrackDims<T>::NF]; } buf[2];
};

// wave-uniform flags of step s, from the five 64-bit step masks in the kernel arguments (scalar ALU only)
template <typename T> RBD_HD int track_sf(const TrackCtx<T>& c, int s) {
  int f = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) f |= (int)((c.M.sfm[k] >> s) & 1) << k;
  return f;
}
template <typename T> RBD_HD I4 track_ri(const TrackCtx<T>& c, int s, int g) { return c.tri[(long)s * c.M.G + g]; }
template <typename T, int V0, int V1> RBD_HD void track_rr(const TrackCtx<T>& c, int s, int g, T* rr) {  // vectors [V0, V1) of the record
  using D = TrackDims<T>;
  const Vec16<T>* src = c.trr + ((long)s * c.M.G + g) * D::NVR;
#pragma unroll
  for (int k = V0; k < V1; ++k) {
    const Vec16<T> x = src[k];
#pragma unroll
    for (int j = 0; j < D::VW; ++j) rr[k * D::VW + j] = x.e[j];
  }
}
template <typename T> RBD_HD T track_ld(const T* x, uint32_t sk, int k, uint32_t b) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(x) + ((uint32_t)k * sk + b));
}
template <typename T> RBD_HD void track_st(T* x, uint32_t sk, int k, uint32_t b, T val) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(x) + ((uint32_t)k * sk + b)) = val;
}
template <typename T> RBD_HD Vec16<T>* track_row(const TrackCtx<T>& c, int s, int lane) {
  return c.rows + (long)s * TrackDims<T>::NV * 64 + lane;
}
template <typename T, int V0, int V1> RBD_HD void track_row_read(const Vec16<T>* row, T* f /*NF*/) {
  using D = TrackDims<T>;
#pragma unroll
  for (int k = V0; k < V1; ++k) {
    const Vec16<T> x = row[k * 64];
#pragma unroll
    for (int j = 0; j < D::VW; ++j) f[k * D::VW + j] = x.e[j];
  }
}
template <typename T, int V0, int V1> RBD_HD void track_row_write(Vec16<T>* row, const T* f /*NF*/) {
  using D = TrackDims<T>;
#pragma unroll
  for (int k = V0; k < V1; ++k) {
    Vec16<T> x;
#pragma unroll
    for (int j = 0; j < D::VW; ++j) x.e[j] = f[k * D::VW + j];
    row[k * 64] = x;
  }
}

// ---- LDS flags between the waves of a workgroup (NW = 4).  A wavefront's DS operations execute in program order, so data written
// before the flag is visible to whoever has seen the flag; the asm statements only pin the compiler's ordering.
RBD_HD void track_flag_set(int* f) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
  *reinterpret_cast<volatile int*>(f) = 1;
  asm volatile("" ::: "memory");
#else
  *f = 1;
#endif
}
RBD_HD void track_flag_wait(int* f) {
#if defined(__HIP_DEVICE_COMPILE__)
  while (*reinterpret_cast<volatile int*>(f) == 0) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
#else
  if (*f == 0) __builtin_trap();  // the emulation runs the roles in dependency order: an unset flag is a bug in that order
#endif
}

// ---------------- staging prologue ----------------
// the plan records -> LDS (tid / nthreads: the caller's position in the workgroup)
template <typename T> RBD_HD void track_stage_plan(const TrackCtx<T>& c, int tid, int nthreads) {
  using D = TrackDims<T>;
  const long nrec = (long)c.M.ns * c.M.G;
  const I4* gi = reinterpret_cast<const I4*>(c.M.ri);
  const Vec16<T>* gr = reinterpret_cast<const Vec16<T>*>(c.M.rr);
  for (long i = tid; i < nrec; i += nthreads) c.tri[i] = gi[i];
  for (long i = tid; i < nrec * D::NVR; i += nthreads) c.trr[i] = gr[i];
}
// inputs of (step s, this lane) -> input phase of the row; also q̇ of 1-dof joints (velocity_to_configuration_derivative!, default q̇ = v).
// Two halves so that a wave can have the loads of several rows in flight before it converts the first (track_stage_rows).
template <typename T> struct TrackStageIn { I4 w; T q0, q1, v0, t0, fe[6]; };
template <typename T, bool GEN>
RBD_HD void track_stage_load(const TrackCtx<T>& c, const TrackWho& W, int s, TrackStageIn<T>& in) {
  in.w = track_ri(c, s, W.g);
  const int flags = (in.w.y >> 16) & 0xff, voff = (in.w.x >> 16) & 0xffff, orig6 = in.w.y & 0xffff;
  int qoff = in.w.x & 0xffff;
  if (GEN) qoff = (flags & TF_FIXED) ? 0 : qoff;  // a fixed joint has no position: its offset may be one past the end
  const int vo = GEN ? ((flags & TF_FIXED) ? 0 : voff) : voff;
  in.q0 = track_ld(c.q, c.qsk, qoff, W.qb);
  in.v0 = c.v ? track_ld(c.v, c.vsk, vo, W.vb) : T(0);
  in.t0 = c.tau ? track_ld(c.tau, c.vsk, vo, W.vb) : T(0);
  in.q1 = T(0);
  if (GEN) in.q1 = track_ld(c.q, c.qsk, (flags & TF_SINCOS) ? qoff + 1 : qoff, W.qb);
#pragma unroll
  for (int k = 0; k < 6; ++k) in.fe[k] = T(0);
  if (c.fext) {
#pragma unroll
    for (int k = 0; k < 6; ++k) in.fe[k] = track_ld(c.fext, c.fsk, orig6 + k, W.fb);
  }
}
template <typename T, int G, bool FLT, bool GEN>
RBD_HD void track_stage_finish(const TrackCtx<T>& c, const TrackWho& W, int s, const TrackStageIn<T>& in) {
  using D = TrackDims<T>;
  const I4 w = in.w;
  const int flags = (w.y >> 16) & 0xff, voff = (w.x >> 16) & 0xffff;
  int qoff = w.x & 0xffff;
  if (GEN) qoff = (flags & TF_FIXED) ? 0 : qoff;
  const int sf = track_sf(c, s);
  T f[D::NF];
#pragma unroll
  for (int k = 0; k < D::NF; ++k) f[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) f[TN_FE + k] = in.fe[k];
  const T q0 = in.q0, q1 = in.q1, v0 = in.v0;
  T sn, cs;
  sincos_fast(q0, &sn, &cs);
  T qd = v0, tq = in.t0;
  if (GEN) {
    if (flags & TF_SINCOS) { sn = q0; cs = q1; }  // sin_cos_revolute.jl:69-96: q = (sin θ, cos θ)
    const bool norot = flags & (TF_PRISMATIC | TF_FIXED);
    sn = norot ? T(0) : sn; cs = norot ? T(1) : cs;
    qd = (flags & TF_FIXED) ? T(0) : qd; tq = (flags & TF_FIXED) ? T(0) : tq;
  }
  f[TN_SN] = sn; f[TN_CS] = cs; f[TN_QD] = qd; f[TN_Q] = q0; f[TS_TAU] = tq;
  const bool active = (flags & TF_VALID) && W.vstate;
  bool floating = false;
  if (FLT) {
    if (sf & SF_FLOAT) {
      if (flags & TF_FLOATING) {  // the 6-dof root keeps its raw coordinates (its external wrench is loaded where it is used)
        floating = true;
#pragma unroll
        for (int k = 0; k < D::NF; ++k) f[k] = T(0);
#pragma unroll
        for (int k = 0; k < 7; ++k) f[TN_FQ + k] = track_ld(c.q, c.qsk, qoff + k, W.qb);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          f[TN_FV + k] = c.v ? track_ld(c.v, c.vsk, voff + k, W.vb) : T(0);
          f[TN_FT + k] = c.tau ? track_ld(c.tau, c.vsk, voff + k, W.vb) : T(0);
        }
      }
    }
  }
  track_row_write<T, 0, D::NV>(track_row(c, s, W.lane), f);
  if (c.qdot && active && !floating && !(GEN && (flags & TF_FIXED))) {
    if (GEN && (flags & TF_SINCOS)) {  // d/dt (sin θ, cos θ) = (cos θ, −sin θ) θ̇
      track_st(c.qdot, c.qsk, qoff, W.qb, q1 * v0);
      track_st(c.qdot, c.qsk, qoff + 1, W.qb, -q0 * v0);
    } else {
      track_st(c.qdot, c.qsk, qoff, W.qb, v0);
    }
  }
}
// rows first, first + stride, ... of this wave, NB at a time: all their loads, then all their conversions
template <typename T, int G, bool FLT, bool GEN, int NB>
RBD_HD void track_stage_rows(const TrackCtx<T>& c, const TrackWho& W, int first, int stride) {
  const int ns = c.M.ns;
  for (int s0 = first; s0 < ns; s0 += NB * stride) {
    TrackStageIn<T> in[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int s = s0 + b * stride;
      track_stage_load<T, GEN>(c, W, s < ns ? s : ns - 1, in[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int s = s0 + b * stride;
      if (s < ns) track_stage_finish<T, G, FLT, GEN>(c, W, s, in[b]);
    }
  }
}

template <typename T, int G>
RBD_HD void track_init(const TrackCtx<T>& c, TrackLane<T>& L, int lane, long group) {
  track_who<T, G>(c, L.who, lane, group);
#pragma unroll
  for (int k = 0; k < 9; ++k) L.R[k] = T(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) L.p[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) { L.Tw[k] = T(0); L.av[k] = T(0); L.cP[k] = T(0); L.ad[k] = T(0); }
#pragma unroll
  for (int k = 0; k < 21; ++k) L.cI[k] = T(0);
}

// Per-body part of pass A from the body's kinematics (R, p, T, a_vp): spatial inertia in the root frame (mechanism_state.jl:836-846),
// motion subspace (:744-763), p̃A = I a_vp + T ×* I T − w_ext (newton_euler, :872-876) -> body phase of the row (fields [0, 21))
template <typename T, bool GEN>
RBD_HD void track_body(const I4& w, const T* rr, const T* R, const T* p, const T* Tw, const T* av, const T* fe, T* f) {
  const int flags = (w.y >> 16) & 0xff;
  RInertia<T> I;
  inertia_to_root(rr + TR_J, rr + TR_MC, rr[TR_M], R, p, I);
  T h[6];
  mul_inertia(I, av, f + TS_PA);
  momentum_cross(I, Tw, h);
  const T z[3] = {R[2], R[5], R[8]};
  T S[6];
  cross3(p, z, S + 3);
#pragma unroll
  for (int k = 0; k < 3; ++k) S[k] = z[k];
  if (GEN) {  // prismatic: S = (0; z); fixed and 6-dof joints: no S
    const T rev = (flags & (TF_PRISMATIC | TF_FIXED | TF_FLOATING)) ? T(0) : T(1), pri = (flags & TF_PRISMATIC) ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < 3; ++k) { S[3 + k] = S[3 + k] * rev + z[k] * pri; S[k] = S[k] * rev; }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) { f[TS_PA + k] += h[k] - fe[k]; f[TS_W + k] = I.J[k]; f[TS_S + k] = S[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) f[TS_UD + k] = I.c[k];
}

// ---------------- pass A (root -> leaves): transforms, twists, velocity-product accelerations ----------------
template <typename T, int G, int NW>
RBD_HD void track_prologue_a(const TrackCtx<T>& c, TrackLane<T>& L) {
  using D = TrackDims<T>;
  L.buf[0].ri = track_ri(c, 0, L.who.g);
  track_rr<T, 0, (NW == 1 ? D::NVR : TR_J / D::VW)>(c, 0, L.who.g, L.buf[0].rr);
  track_row_read<T, 0, D::NV>(track_row(c, 0, L.who.lane), L.buf[0].row);
}

template <typename T, int G, bool FLT, bool GEN, int NW, int P>
RBD_HD void track_step_a(const TrackCtx<T>& c, TrackLane<T>& L, int s) {
  using D = TrackDims<T>;
  constexpr int SPW = 64 / G;
  const int ns = c.M.ns;
  const TrackWho& W = L.who;
  const int s1 = s + 1 < ns ? s + 1 : ns - 1;
  const int sf = track_sf(c, s);
  const typename TrackLane<T>::Buf& in = L.buf[P];
  const I4 w = in.ri;
  const int flags = (w.y >> 16) & 0xff;
  const bool active = (flags & TF_VALID) && W.vstate;
  // LDS order within a step: rare mailbox reads first, then the next step's record and row (consumed one step later), then this
  // step's writes.  The LDS counter is in order: a wait for the mailbox data placed after the prefetch would also wait for the prefetch.
  if (sf & SF_NOCHAIN) {
    if (active && !(flags & TF_CHAINED)) {  // the registers do not hold the parent: the world, or a parent on another track / an earlier step
      if (flags & TF_LEVEL0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) L.R[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) L.p[k] = T(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) { L.Tw[k] = T(0); L.av[k] = c.a0[k]; }
      } else {
        const T* m = c.mbA + (long)(((w.z >> 16) & 0xffff) - 1) * TMB_A * SPW + W.sl;
#pragma unroll
        for (int k = 0; k < 9; ++k) L.R[k] = m[k * SPW];
#pragma unroll
        for (int k = 0; k < 3; ++k) L.p[k] = m[(9 + k) * SPW];
#pragma unroll
        for (int k = 0; k < 6; ++k) { L.Tw[k] = m[(12 + k) * SPW]; L.av[k] = m[(18 + k) * SPW]; }
      }
    }
  }
  L.buf[1 - P].ri = track_ri(c, s1, W.g);
  track_rr<T, 0, (NW == 1 ? D::NVR : TR_J / D::VW)>(c, s1, W.g, L.buf[1 - P].rr);
  track_row_read<T, 0, D::NV>(track_row(c, s1, W.lane), L.buf[1 - P].row);  // a lane's rows are its own: no hazard with this step's write
  Vec16<T>* row = track_row(c, s, W.lane);

  if (active) {
    const T* rr = in.rr;
    const T* x = in.row;
    T Rn[9], pn[3], u3[3], fe[6];
    bool done = false;
    if (FLT) {
      if (sf & SF_FLOAT) {
        if (flags & TF_FLOATING) {
          // 6-dof joint (quaternion_floating.jl:81-83): H = H_parent * Xpred * (R(quat), trans); twist = X(H) v (v: body-frame twist)
          T Rq[9], CR[9], t3[3], vJ[6], cb[6], f6[6];
          rot_quat(x[TN_FQ], x[TN_FQ + 1], x[TN_FQ + 2], x[TN_FQ + 3], Rq);
          matmul3(rr + TR_C, Rq, CR);
          matmul3(L.R, CR, Rn);
          matvec3(rr + TR_C, x + TN_FQ + 4, t3);
#pragma unroll
          for (int k = 0; k < 3; ++k) t3[k] += rr[TR_PP + k];
          matvec3(L.R, t3, u3);
#pragma unroll
          for (int k = 0; k < 3; ++k) pn[k] = L.p[k] + u3[k];
          const T* v6 = x + TN_FV;
          xmotion(Rn, pn, v6, vJ);
          se3_comm(L.Tw, vJ, cb);
#pragma unroll
          for (int k = 0; k < 6; ++k) { L.av[k] += cb[k]; L.Tw[k] += vJ[k]; }
          if (c.qdot) {  // velocity_to_configuration_derivative! (quaternion_floating.jl:126-136, spatial/util.jl:127-134)
            const T qw = x[TN_FQ], qx = x[TN_FQ + 1], qy = x[TN_FQ + 2], qz = x[TN_FQ + 3];
            const int qoff = w.x & 0xffff;
            T o[7];
            o[0] = (-qx * v6[0] - qy * v6[1] - qz * v6[2]) / 2;
            o[1] = (qw * v6[0] - qz * v6[1] + qy * v6[2]) / 2;
            o[2] = (qz * v6[0] + qw * v6[1] - qx * v6[2]) / 2;
            o[3] = (-qy * v6[0] + qx * v6[1] + qw * v6[2]) / 2;
            matvec3(Rq, v6 + 3, o + 4);
#pragma unroll
            for (int k = 0; k < 7; ++k) track_st(c.qdot, c.qsk, qoff + k, W.qb, o[k]);
          }
          // pass B solves IA a_Δ = S⁻ᵀτ − p̃A at this body: leave S⁻ᵀτ (the wrench transform of τ to the root frame) in its mailbox
          xforce(Rn, pn, x + TN_FT, f6);
          T* m = c.mbA + (long)((w.z & 0xffff) - 1) * TMB_A * SPW + W.sl;
#pragma unroll
          for (int k = 0; k < 6; ++k) m[(24 + k) * SPW] = f6[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) fe[k] = T(0);
          if (NW == 1 && c.fext) {
#pragma unroll
            for (int k = 0; k < 6; ++k) fe[k] = track_ld(c.fext, c.fsk, (w.y & 0xffff) + k, W.fb);
          }
          done = true;
        }
      }
    }
    if (!done) {
      T Mx[9], S[6];
      matmul3(L.R, rr + TR_C, Mx);
      matvec3(L.R, rr + TR_PP, u3);
#pragma unroll
      for (int k = 0; k < 3; ++k) pn[k] = L.p[k] + u3[k];
      const T z[3] = {Mx[2], Mx[5], Mx[8]};
      const T sn = x[TN_SN], cs = x[TN_CS];
      T rev = T(1), pri = T(0);
      if (GEN) {  // prismatic / fixed joints do not rotate (staged as sin = 0, cos = 1); prismatic: translate along z, S = (0; z)
        rev = (flags & (TF_PRISMATIC | TF_FIXED)) ? T(0) : T(1); pri = (flags & TF_PRISMATIC) ? T(1) : T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) pn[k] += z[k] * (x[TN_Q] * pri);
      }
      // revolute about the body's z axis (revolute.jl:59-62 in the canonical frame): columns 0 and 1 of Mx are mixed
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        Rn[3 * i] = cs * Mx[3 * i] + sn * Mx[3 * i + 1];
        Rn[3 * i + 1] = cs * Mx[3 * i + 1] - sn * Mx[3 * i];
        Rn[3 * i + 2] = Mx[3 * i + 2];
      }
      cross3(pn, z, S + 3);
#pragma unroll
      for (int k = 0; k < 3; ++k) S[k] = z[k];
      if (GEN) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { S[3 + k] = S[3 + k] * rev + z[k] * pri; S[k] = S[k] * rev; }
      }
      T vJ[6], cb[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) vJ[k] = S[k] * x[TN_QD];
      se3_comm(L.Tw, vJ, cb);  // [T_parent, vJ] = [T_b, vJ]: bias acceleration increment (mechanism_state.jl:814-830)
#pragma unroll
      for (int k = 0; k < 6; ++k) { L.av[k] += cb[k]; L.Tw[k] += vJ[k]; fe[k] = x[TN_FE + k]; }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) L.R[k] = Rn[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) L.p[k] = pn[k];
    T f[D::NF];
#pragma unroll
    for (int k = 0; k < D::NF; ++k) f[k] = T(0);
    f[TS_TAU] = x[TS_TAU];
    if (NW == 1) {
      track_body<T, GEN>(w, rr, L.R, L.p, L.Tw, L.av, fe, f);
    } else {  // publish the kinematics: a helper wave turns them into the body phase of the row
#pragma unroll
      for (int i = 0; i < 3; ++i) { f[TK_R01 + i] = L.R[3 * i]; f[TK_R01 + 3 + i] = L.R[3 * i + 1]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) f[TK_P + k] = L.p[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) { f[TK_T + k] = L.Tw[k]; f[TK_AV + k] = L.av[k]; }
    }
    track_row_write<T, 0, (TS_TAU + 1 + D::VW - 1) / D::VW>(row, f);
    if (sf & SF_AW) {
      const int aw = (w.z & 0xffff) - 1;
      if (aw >= 0) {  // some child is not next on this track (or this is a 6-dof root): leave the kinematics in the mailbox
        T* m = c.mbA + (long)aw * TMB_A * SPW + W.sl;
#pragma unroll
        for (int k = 0; k < 9; ++k) m[k * SPW] = L.R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) m[(9 + k) * SPW] = L.p[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { m[(12 + k) * SPW] = L.Tw[k]; m[(18 + k) * SPW] = L.av[k]; }
      }
    }
  }
  if (NW > 1) track_flag_set(c.flags + s);
}

// ---------------- helper waves (NW = 4): helper h = 1..3 owns the rows s ≡ h − 1 (mod 3) ----------------
// as wave 0 publishes each row's kinematics, the helper turns them into the body phase of the row (the per-body part of pass A)
template <typename T, int G, bool GEN>
RBD_HD void track_helper_body(const TrackCtx<T>& c, const TrackWho& W, int s) {
  using D = TrackDims<T>;
  const I4 w = track_ri(c, s, W.g);
  T rr[TR_STRIDE], fe[6], k[D::NF], f[D::NF], R[9];
  track_rr<T, TR_J / D::VW, D::NVR>(c, s, W.g, rr);
#pragma unroll
  for (int j = 0; j < 6; ++j) fe[j] = T(0);
  if (c.fext) {
#pragma unroll
    for (int j = 0; j < 6; ++j) fe[j] = track_ld(c.fext, c.fsk, (w.y & 0xffff) + j, W.fb);
  }
  Vec16<T>* row = track_row(c, s, W.lane);
  track_flag_wait(c.flags + s);
  track_row_read<T, 0, (TS_TAU + 1 + D::VW - 1) / D::VW>(row, k);
#pragma unroll
  for (int i = 0; i < 3; ++i) { R[3 * i] = k[TK_R01 + i]; R[3 * i + 1] = k[TK_R01 + 3 + i]; }
  R[2] = R[3] * R[7] - R[6] * R[4]; R[5] = R[6] * R[1] - R[0] * R[7]; R[8] = R[0] * R[4] - R[3] * R[1];  // column 2 = column 0 × column 1
#pragma unroll
  for (int j = 0; j < D::NF; ++j) f[j] = T(0);
  f[TS_TAU] = k[TS_TAU];
  track_body<T, GEN>(w, rr, R, k + TK_P, k + TK_T, k + TK_AV, fe, f);
  track_row_write<T, 0, (TS_TAU + 1 + D::VW - 1) / D::VW>(row, f);
}
template <typename T, int G, bool GEN>
RBD_HD void track_helper(const TrackCtx<T>& c, const TrackWho& W, int h) {
  const int ns = c.M.ns;
  for (int s = h - 1; s < ns; s += 3) track_helper_body<T, G, GEN>(c, W, s);
  track_flag_set(c.flags + ns + h);
}

// ---------------- pass B (leaves -> root): articulated-body inertias and bias forces; U D⁻¹ and D⁻¹u of each joint -> row
template <typename T, int G, int NW>
RBD_HD void track_prologue_b(const TrackCtx<T>& c, TrackLane<T>& L) {
  using D = TrackDims<T>;
  const int ns = c.M.ns;
  if (NW > 1) {  // every row must be in its body phase
    track_flag_wait(c.flags + ns + 1);
    track_flag_wait(c.flags + ns + 2);
    track_flag_wait(c.flags + ns + 3);
  }
  L.buf[0].ri = track_ri(c, ns - 1, L.who.g);
  track_rr<T, TR_M / D::VW, TR_M / D::VW + 1>(c, ns - 1, L.who.g, L.buf[0].rr);
  track_row_read<T, 0, D::NV>(track_row(c, ns - 1, L.who.lane), L.buf[0].row);
}

template <typename T, int G, bool FLT, bool GEN, int P>
RBD_HD void track_step_b(const TrackCtx<T>& c, TrackLane<T>& L, int s) {
  using D = TrackDims<T>;
  constexpr int SPW = 64 / G;
  const TrackWho& W = L.who;
  const int s1 = s - 1 > 0 ? s - 1 : 0;
  const int sf = track_sf(c, s);
  const typename TrackLane<T>::Buf& in = L.buf[P];
  const I4 w = in.ri;
  const int flags = (w.y >> 16) & 0xff;
  const bool active = (flags & TF_VALID) && W.vstate;
  T mI[21], mP[6];  // hand-offs of the children that finished on other tracks (or earlier on this one); rare LDS reads first (see pass A)
#pragma unroll
  for (int k = 0; k < 21; ++k) mI[k] = T(0);
#pragma unroll
  for (int k = 0; k < 6; ++k) mP[k] = T(0);
  T Rn[9], pn[3], f6[6];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rn[k] = T(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) { pn[k] = T(0); f6[k] = T(0); f6[3 + k] = T(0); }
  if (sf & SF_BR) {
    if (active) {
      const int nbr = (w.y >> 24) & 0x7f, b0 = ((w.w >> 16) & 0xffff) - 1;
      for (int j = 0; j < nbr; ++j) {  // all 27 reads of a hand-off in flight, then the sums (read-wait-add per value costs an LDS round trip each)
        const T* m = c.mbB + (long)(b0 + j) * TMB_B * SPW + W.sl;
        T t[TMB_B];
#pragma unroll
        for (int k = 0; k < TMB_B; ++k) t[k] = m[k * SPW];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 21; ++k) mI[k] += t[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) mP[k] += t[21 + k];
      }
    }
  }
  if (FLT) {
    if (sf & SF_FLOAT) {
      if (active && (flags & TF_FLOATING)) {  // transform to root and S⁻ᵀτ of the 6-dof root, left by pass A
        const T* m = c.mbA + (long)((w.z & 0xffff) - 1) * TMB_A * SPW + W.sl;
#pragma unroll
        for (int k = 0; k < 9; ++k) Rn[k] = m[k * SPW];
#pragma unroll
        for (int k = 0; k < 3; ++k) pn[k] = m[(9 + k) * SPW];
#pragma unroll
        for (int k = 0; k < 6; ++k) f6[k] = m[(24 + k) * SPW];
      }
    }
  }
  L.buf[1 - P].ri = track_ri(c, s1, W.g);
  track_rr<T, TR_M / D::VW, TR_M / D::VW + 1>(c, s1, W.g, L.buf[1 - P].rr);
  track_row_read<T, 0, D::NV>(track_row(c, s1, W.lane), L.buf[1 - P].row);
  if (active) {
    Vec16<T>* row = track_row(c, s, W.lane);
    const T* f = in.row;
    // (L.cI, L.cP) hold the hand-off of the chained child, or zero at the leaf of a chain: they are cleared whenever a hand-off
    // leaves the track (below), so no per-step select is needed here
    T IA[21], pA[6];
    {
      RInertia<T> I;
#pragma unroll
      for (int k = 0; k < 6; ++k) I.J[k] = f[TS_W + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) I.c[k] = f[TS_UD + k];
      I.m = in.rr[TR_M];
      sym6_from_inertia(I, IA);
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) IA[k] += L.cI[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) pA[k] = f[TS_PA + k] + L.cP[k];
    if (sf & SF_BR) {
#pragma unroll
      for (int k = 0; k < 21; ++k) IA[k] += mI[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) pA[k] += mP[k];
    }
    const int voff = (w.x >> 16) & 0xffff;
    bool done = false;
    if (FLT) {
      if (sf & SF_FLOAT) {
        if (flags & TF_FLOATING) {
          // 6-dof joint on the world: IA a_Δ = S⁻ᵀτ − p̃A, v̇ = S⁻¹ a_Δ  (S = X(H): the body-frame twist basis seen from the root)
          T rhs[6], a[6], vd[6], fw[D::NF];
#pragma unroll
          for (int k = 0; k < 6; ++k) rhs[k] = f6[k] - pA[k];
          sym6_solve(IA, rhs, a);
          xmotion_inv(Rn, pn, a, vd);
          if (c.vdot) {
#pragma unroll
            for (int k = 0; k < 6; ++k) track_st(c.vdot, c.vsk, voff + k, W.vb, vd[k]);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) fw[k] = k < 6 ? a[k] : T(0);
          track_row_write<T, 0, (6 + D::VW - 1) / D::VW>(row, fw);
          done = true;
        }
      }
    }
    if (!done) {
      const T* S = f + TS_S;
      T U[6], Wd[6], fw[D::NF];
      sym6_mul(IA, S, U);
      T Dinv = rcp_hd(dot6(S, U));
      if (GEN) Dinv = (flags & TF_FIXED) ? T(0) : Dinv;  // a fixed joint has S = 0: the body hands its whole inertia up
      const T ud = (f[TS_TAU] - dot6(S, pA)) * Dinv;
#pragma unroll
      for (int k = 0; k < 6; ++k) { Wd[k] = U[k] * Dinv; fw[TS_W + k] = Wd[k]; }
      fw[TS_UD] = ud; fw[TS_UD + 1] = ud;
      track_row_write<T, 0, (7 + D::VW - 1) / D::VW>(row, fw);
      // hand-off: Ia = IA − U D⁻¹ U', p̃a = p̃A + U D⁻¹ u
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) L.cI[SI(i, j)] = IA[SI(i, j)] - Wd[i] * U[j];
#pragma unroll
      for (int k = 0; k < 6; ++k) L.cP[k] = pA[k] + U[k] * ud;
    }
    if (sf & SF_BW) {
      const int bw = (w.w & 0xffff) - 1;
      if (bw >= 0 || (flags & TF_LEVEL0)) {  // the parent is not next on this track (or is the world): the hand-off leaves the registers
        if (bw >= 0) {
          T* m = c.mbB + (long)bw * TMB_B * SPW + W.sl;
#pragma unroll
          for (int k = 0; k < 21; ++k) m[k * SPW] = L.cI[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) m[(21 + k) * SPW] = L.cP[k];
        }
#pragma unroll
        for (int k = 0; k < 21; ++k) L.cI[k] = T(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) L.cP[k] = T(0);
      }
    }
  }
}

// ---------------- pass C (root -> leaves): v̇ and a_Δ ----------------
template <typename T, int G>
RBD_HD void track_prologue_c(const TrackCtx<T>& c, TrackLane<T>& L) {
  using D = TrackDims<T>;
  L.buf[0].ri = track_ri(c, 0, L.who.g);
  track_row_read<T, 0, (TS_S + 6 + D::VW - 1) / D::VW>(track_row(c, 0, L.who.lane), L.buf[0].row);
}

template <typename T, int G, bool FLT, bool GEN, int P>
RBD_HD void track_step_c(const TrackCtx<T>& c, TrackLane<T>& L, int s) {
  using D = TrackDims<T>;
  constexpr int SPW = 64 / G;
  const int ns = c.M.ns;
  const TrackWho& W = L.who;
  const int s1 = s + 1 < ns ? s + 1 : ns - 1;
  const int sf = track_sf(c, s);
  constexpr int NVC = (TS_S + 6 + D::VW - 1) / D::VW;  // vectors covering U D⁻¹, D⁻¹u, S
  const T* f = L.buf[P].row;
  const I4 w = L.buf[P].ri;
  const int flags = (w.y >> 16) & 0xff;
  const bool active = (flags & TF_VALID) && W.vstate;
  if (sf & SF_NOCHAIN) {  // (LDS order as in pass A: the rare mailbox read before the next row's prefetch)
    if (active && !(flags & TF_CHAINED)) {
      if (flags & TF_LEVEL0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) L.ad[k] = T(0);
      } else {
        const T* m = c.mbC + (long)(((w.z >> 16) & 0xffff) - 1) * TMB_C * SPW + W.sl;
#pragma unroll
        for (int k = 0; k < 6; ++k) L.ad[k] = m[k * SPW];
      }
    }
  }
  L.buf[1 - P].ri = track_ri(c, s1, W.g);
  track_row_read<T, 0, NVC>(track_row(c, s1, W.lane), L.buf[1 - P].row);
  if (active) {
    bool done = false;
    if (FLT) {
      if (sf & SF_FLOAT) {
        if (flags & TF_FLOATING) {
#pragma unroll
          for (int k = 0; k < 6; ++k) L.ad[k] = f[TS_W + k];
          done = true;
        }
      }
    }
    if (!done && !(GEN && (flags & TF_FIXED))) {
      const T vd = f[TS_UD] - dot6(f + TS_W, L.ad);  // v̇ = D⁻¹u − (U D⁻¹)' a_Δ,parent
#pragma unroll
      for (int k = 0; k < 6; ++k) L.ad[k] += f[TS_S + k] * vd;
      if (c.vdot) track_st(c.vdot, c.vsk, (w.x >> 16) & 0xffff, W.vb, vd);
    }
    if (sf & SF_AW) {
      const int aw = (w.z & 0xffff) - 1;
      if (aw >= 0) {
        T* m = c.mbC + (long)aw * TMB_C * SPW + W.sl;
#pragma unroll
        for (int k = 0; k < 6; ++k) m[k * SPW] = L.ad[k];
      }
    }
  }
}

#if defined(__HIPCC__)
#ifdef RBD_PROFILE_PHASES
__device__ long long rbd_track_phase_clock[16];
#define RBD_TMARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) rbd_track_phase_clock[i] = clock64(); } while (0)
#else
#define RBD_TMARK(i)
#endif
// Wave 0's three passes.  Its LDS operations execute in program order, so the mailbox traffic between its lanes needs no barrier;
// the asm statements only stop the compiler from moving memory operations across a step boundary.  The step loops are unrolled by
// two by hand: the two copies use the two pipeline buffers in turn (no register copies).
template <typename T, int G, bool FLT, bool GEN, int NW>
__device__ __forceinline__ void track_main_wave(const TrackCtx<T>& c, int lane, long group) {
  const int ns = c.M.ns;
  TrackLane<T> L;
  track_init<T, G>(c, L, lane, group);
  track_prologue_a<T, G, NW>(c, L);
  int s = 0;
#pragma unroll 1
  for (; s + 1 < ns; s += 2) {
    track_step_a<T, G, FLT, GEN, NW, 0>(c, L, s);
    asm volatile("" ::: "memory");
    track_step_a<T, G, FLT, GEN, NW, 1>(c, L, s + 1);
    asm volatile("" ::: "memory");
  }
  if (s < ns) track_step_a<T, G, FLT, GEN, NW, 0>(c, L, s);
  asm volatile("" ::: "memory");
  RBD_TMARK(2);
  track_prologue_b<T, G, NW>(c, L);
  s = ns - 1;
#pragma unroll 1
  for (; s >= 1; s -= 2) {
    track_step_b<T, G, FLT, GEN, 0>(c, L, s);
    asm volatile("" ::: "memory");
    track_step_b<T, G, FLT, GEN, 1>(c, L, s - 1);
    asm volatile("" ::: "memory");
  }
  if (s == 0) track_step_b<T, G, FLT, GEN, 0>(c, L, 0);
  asm volatile("" ::: "memory");
  RBD_TMARK(3);
  track_prologue_c<T, G>(c, L);
  s = 0;
#pragma unroll 1
  for (; s + 1 < ns; s += 2) {
    track_step_c<T, G, FLT, GEN, 0>(c, L, s);
    asm volatile("" ::: "memory");
    track_step_c<T, G, FLT, GEN, 1>(c, L, s + 1);
    asm volatile("" ::: "memory");
  }
  if (s < ns) track_step_c<T, G, FLT, GEN, 0>(c, L, s);
  RBD_TMARK(4);
#ifdef RBD_PROFILE_PHASES
  // instruction-cache probe: pass C again (it only reads the rows), now with its code resident
  asm volatile("" ::: "memory");
  track_prologue_c<T, G>(c, L);
  s = 0;
#pragma unroll 1
  for (; s + 1 < ns; s += 2) {
    track_step_c<T, G, FLT, GEN, 0>(c, L, s);
    asm volatile("" ::: "memory");
    track_step_c<T, G, FLT, GEN, 1>(c, L, s + 1);
    asm volatile("" ::: "memory");
  }
  if (s < ns) track_step_c<T, G, FLT, GEN, 0>(c, L, s);
  RBD_TMARK(5);
#endif
}

template <typename T, int G, bool FLT, bool GEN, int NW>
__global__ __launch_bounds__(64 * NW) void aba_track_kernel(TrackModel M, long B, const T* __restrict__ q, const T* __restrict__ v,
                                                           const T* __restrict__ tau, const T* __restrict__ fext, T* __restrict__ vdot,
                                                           T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf) {
  extern __shared__ double lds_raw[];
  TrackCtx<T> c;
  c.M = M; c.B = B; c.q = q; c.v = v; c.tau = tau; c.fext = fext; c.vdot = vdot; c.qdot = qdot; c.Lq = Lq; c.Lv = Lv; c.Lf = Lf;
  track_ctx_lds(c, lds_raw);
  const int lane = threadIdx.x & 63;
  const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  RBD_TMARK(0);
  // staging prologue: plan -> LDS, then the inputs of every (step, lane) -> that lane's row (the waves share the rows out)
  track_stage_plan(c, threadIdx.x, 64 * NW);
  for (int i = threadIdx.x; i < M.ns + 4; i += 64 * NW) c.flags[i] = 0;
  __syncthreads();
  TrackWho W;
  track_who<T, G>(c, W, lane, blockIdx.x);
  track_stage_rows<T, G, FLT, GEN, (NW == 1 ? 6 : 3)>(c, W, wave, NW);
  __syncthreads();
  RBD_TMARK(1);
  if (wave == 0) {
    track_main_wave<T, G, FLT, GEN, NW>(c, lane, blockIdx.x);
  } else {
    track_helper<T, G, GEN>(c, W, wave);
  }
}
#endif

}  // namespace rbd
