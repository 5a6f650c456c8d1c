// rbd_spec.hpp — MODEL-SPECIALISED kernels: device code that is compiled at run time (hiprtc, rbd_jit.hip) for ONE mechanism.
//
// The one-lane-per-state kernels of rbd_state.hpp interpret the tree: every op of the depth-first walk costs the scalar unit a decode
// (kind, level, joint type, offsets, ancestor columns), a `switch (level)` around the registers of the path, and LDS reads of the body's
// constants — for Atlas 8.7 k scalar + 1.3 k LDS instructions beside 14.5 k vector ones, on a wavefront that is alone on its SIMD and
// issues one instruction at a time (89 us for mass_matrix! at 65 536 fp32 states, however few wavefronts run).  Here the walk is a
// compile-time constant: rbd_jit.hip writes the plan of the mechanism (rbd_state_plan.hpp) out as `constexpr` tables in namespace
// rbd_plan, includes this header and compiles the result for gfx950.  Every op becomes straight-line code with its level, joint type,
// offsets and body constants folded in; the path of transforms / inertias / motion subspaces is a set of statically indexed arrays, i.e.
// registers the allocator places; nothing is decoded at run time.  (The reference gets the same effect from Julia's JIT, which
// specialises mass_matrix! on the mechanism's joint-type tuple: src/mechanism_algorithms.jl:248-272, TypeSortedCollections.)
// In this file: crba_spec (mass_matrix!), chol_spec (the dense step over the non-zero tiles of a no-fill-in ordering) and emit_spec (M in the caller's
// layout), aba_spec (dynamics!, fp32), rnea_spec (inverse_dynamics! / dynamics_bias!; fp64 with v, v̇ read by the lane).  One hiprtc program per kernel
// family and scalar type (rbd_jit.hip); the interpreting kernels stay the fallback and the reference point of the parity tests.  DESIGN.md §3.7.
//
// Expected before inclusion:  namespace rbd_plan { constexpr int NB, NQ, NV, NOPS, NLEVELS; constexpr int OPW[NOPS][4] (word 0, q offset,
// v offset, 6 * reference body index); constexpr int COLS[NOPS][16]; constexpr double TR[NOPS][24]; constexpr unsigned long long
// ROWMASK[NV]; constexpr double GRAVITY[3]; }
// and, when NV is a multiple of 4, NV <= 40 (RBD_SPEC_EMIT; with RBD_SPEC_CHOL — fp32 — the dense step is specialised too; fp64: PERM is the identity and the
// dense kernel of rbd_kernels.hip reads the staging buffer): constexpr int NT = NV / 4, PERM[NV] (the
// position of a velocity coordinate in the factorisation order), INV[NV] (its inverse), constexpr unsigned char TMASK[NT][NT] (tile (I, J) of the
// permuted lower triangle holds a non-zero), EMIT_KMAX, EMIT_K[NT], and __constant__ unsigned EMIT[NT][4 * EMIT_KMAX] (staged entry << 16 | slot in the tile).
#pragma once
#include "rbd_device.hpp"
#ifdef RBD_SPEC_ABA
#include "rbd_mk_fuse.hpp"
#endif

namespace rbd {
namespace spec {

template <int I> struct Ix { static constexpr int value = I; };
template <int... Is> struct Seq {};
template <int N, int... Is> struct MakeSeq : MakeSeq<N - 1, N - 1, Is...> {};
template <int... Is> struct MakeSeq<0, Is...> { using type = Seq<Is...>; };
template <typename F, int... Is> RBD_DEV void sfor_impl(F&& f, Seq<Is...>) { (f(Ix<Is>{}), ...); }
template <int N, typename F> RBD_DEV void sfor(F&& f) { sfor_impl(f, typename MakeSeq<N>::type{}); }

namespace P = rbd_plan;

// LDS written by some lanes of the wavefront, read by others: order the accesses for the compiler (the hardware executes a wavefront's LDS
// instructions in order)
RBD_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// joints with more than one velocity coordinate (quaternion_floating.jl, quaternion_spherical.jl, planar.jl): their motion subspace is constant in the body's
// (canonical) frame and made of unit twists — coordinate k drives component comp_of(jt, k) of the body-frame twist (angular 0..2, linear 3..5; the planar
// joint: linear x, linear y, angular z).  Seen from the root, column k is X(R, p) e_comp; S' f picks the same components of X^-T f.
constexpr int nvj_of(int jt) { return jt == RBD_JOINT_QUAT_FLOATING ? 6 : (jt == RBD_JOINT_QUAT_SPHERICAL || jt == RBD_JOINT_PLANAR) ? 3 : jt == RBD_JOINT_FIXED ? 0 : 1; }
constexpr int comp_of(int jt, int k) { return jt == RBD_JOINT_PLANAR ? (k == 0 ? 3 : k == 1 ? 4 : 2) : k; }
constexpr int jt_of_col(int col) { return (col & SC_FLOATING) ? RBD_JOINT_QUAT_FLOATING : (col & SC_SPHERICAL) ? RBD_JOINT_QUAT_SPHERICAL : RBD_JOINT_PLANAR; }
// body-frame twist of the joint's n coordinates at c[0], c[stride], ...
template <typename T, int jt> RBD_DEV void body_twist(const T* c, int stride, T* v6) {
#pragma unroll
  for (int k = 0; k < 6; ++k) v6[k] = T(0);
#pragma unroll
  for (int k = 0; k < nvj_of(jt); ++k) v6[comp_of(jt, k)] = c[k * stride];
}
// inverse of a symmetric 3 x 3 matrix {a00, a01, a02, a11, a12, a22} (same packing out)
template <typename T> RBD_DEV void sym3_inv(const T* a, T* o) {
  const T c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
  const T id = rcp_hd(a[0] * c00 + a[1] * c01 + a[2] * c02);
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id; o[4] = (a[1] * a[2] - a[0] * a[4]) * id; o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}

// (Rl, pl) = joint_to_predecessor * joint_transform(q) of op O in canonical frames (joint axis +z; planar: x, y, x × y = +x, +y, +z); qs: this lane's column of the staged q
template <typename T, int O, int QS = 64> RBD_DEV void local_transform(const T* qs, T* Rl, T* pl) {
  constexpr int jt = P::OPW[O][0] >> 16, qoff = P::OPW[O][1];
  T C[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) C[k] = T(P::TR[O][TR_C + k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) pl[k] = T(P::TR[O][TR_PP + k]);
  if constexpr (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_SINCOS_REVOLUTE) {
    T s, c;
    if constexpr (jt == RBD_JOINT_REVOLUTE) sincos_fast(qs[qoff * QS], &s, &c);
    else { s = qs[qoff * QS]; c = qs[(qoff + 1) * QS]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Rl[3 * i] = c * C[3 * i] + s * C[3 * i + 1];
      Rl[3 * i + 1] = c * C[3 * i + 1] - s * C[3 * i];
      Rl[3 * i + 2] = C[3 * i + 2];
    }
  } else if constexpr (jt == RBD_JOINT_QUAT_FLOATING) {
    T Rq[9], pq[3], t[3];
    rot_quat(qs[qoff * QS], qs[(qoff + 1) * QS], qs[(qoff + 2) * QS], qs[(qoff + 3) * QS], Rq);
    pq[0] = qs[(qoff + 4) * QS]; pq[1] = qs[(qoff + 5) * QS]; pq[2] = qs[(qoff + 6) * QS];
    matmul3(C, Rq, Rl);
    matvec3(C, pq, t);
#pragma unroll
    for (int k = 0; k < 3; ++k) pl[k] += t[k];
  } else if constexpr (jt == RBD_JOINT_QUAT_SPHERICAL) {  // quaternion_spherical.jl:39-42
    T Rq[9];
    rot_quat(qs[qoff * QS], qs[(qoff + 1) * QS], qs[(qoff + 2) * QS], qs[(qoff + 3) * QS], Rq);
    matmul3(C, Rq, Rl);
  } else if constexpr (jt == RBD_JOINT_PLANAR) {  // planar.jl:65-70: translate in the x-y plane, turn about z
    T s, c;
    sincos_fast(qs[(qoff + 2) * QS], &s, &c);
    const T x = qs[qoff * QS], y = qs[(qoff + 1) * QS];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Rl[3 * i] = c * C[3 * i] + s * C[3 * i + 1];
      Rl[3 * i + 1] = c * C[3 * i + 1] - s * C[3 * i];
      Rl[3 * i + 2] = C[3 * i + 2];
      pl[i] += C[3 * i] * x + C[3 * i + 1] * y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rl[k] = C[k];
    if constexpr (jt == RBD_JOINT_PRISMATIC) {
      const T d = qs[qoff * QS];
#pragma unroll
      for (int k = 0; k < 3; ++k) pl[k] += d * C[3 * k + 2];
    }
  }
}

constexpr int RS = 65;  // LDS row stride in values

// rows [0, n) <- the n x 64 block of a batch buffer that belongs to this wavefront's states (states past the end read the last one).  The loads of up to
// 32 rows are all in flight before the first LDS write: a lone wavefront pays every global round trip in full (four at a time was 27 round trips
// for Atlas's q, v and tau — a fifth of the launch).
// State-major buffers: the block is one contiguous run of 64 n scalars, element e = 64 c + lane = (state e / n, row e % n).  With lane = n st0 + k0 and c a
// constant of the unrolled loop that is ONE wrap test per element (place), and a block that exists entirely needs no clamping: 15 -> 6 vector instructions per
// element (PMC, Atlas fp32: 19 913 VALU per wavefront state-major against 17 692 batch-innermost, 145 elements in and out).
template <int n> RBD_DEV void place(int c, int st0, int k0, int& st, int& k) {
  const int A = (64 * c) / n, R = (64 * c) % n;
  const int t = k0 + R;  // < 2 n
  const bool w = t >= n;
  k = w ? t - n : t;
  st = st0 + A + (w ? 1 : 0);
}
template <typename T, int n> RBD_DEV void rows_in(const T* __restrict__ src, Layout L, long state0, long B, T* rows) {
  const int lane = threadIdx.x & 63;
  constexpr int CH = 32;
  if (L.sk == 1 && L.sb == n) {
    const T* base = src + state0 * n;
    const int st0 = lane / n, k0 = lane - st0 * n;
    if (B - state0 >= 64) {  // (uniform) every state of the block exists
#pragma unroll
      for (int c0 = 0; c0 < n; c0 += CH) {
        T tmp[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j)
          if (c0 + j < n) tmp[j] = base[(c0 + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (c0 + j < n) {
            int st, k;
            place<n>(c0 + j, st0, k0, st, k);
            rows[k * RS + st] = tmp[j];
          }
        }
      }
    } else {
      const int lim = (int)(B - state0) * n;  // elements of the block that exist
#pragma unroll
      for (int c0 = 0; c0 < n; c0 += CH) {
        T tmp[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (c0 + j < n) {
            int st, k;
            place<n>(c0 + j, st0, k0, st, k);
            const int e = (c0 + j) * 64 + lane;
            tmp[j] = base[e < lim ? e : lim - n + k];
          }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if (c0 + j < n) {
            int st, k;
            place<n>(c0 + j, st0, k0, st, k);
            rows[k * RS + st] = tmp[j];
          }
        }
      }
    }
  } else {
    const long sc = state0 + lane < B ? state0 + lane : B - 1;
    const T* base = src + sc * L.sb;
#pragma unroll
    for (int c0 = 0; c0 < n; c0 += CH) {
      T tmp[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if (c0 + j < n) tmp[j] = base[(long)(c0 + j) * L.sk];
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if (c0 + j < n) rows[(c0 + j) * RS + lane] = tmp[j];
    }
  }
}
template <typename T, int n> RBD_DEV void rows_out(const T* rows, T* __restrict__ dst, Layout L, long state0, long B) {
  const int lane = threadIdx.x & 63;
  if (L.sk == 1 && L.sb == n) {
    T* base = dst + state0 * n;
    const int st0 = lane / n, k0 = lane - st0 * n;
    const int lim = B - state0 >= 64 ? 64 * n : (int)(B - state0) * n;
    if (lim == 64 * n) {
#pragma unroll
      for (int c = 0; c < n; ++c) {
        int st, k;
        place<n>(c, st0, k0, st, k);
        base[c * 64 + lane] = rows[k * RS + st];
      }
    } else {
#pragma unroll 4
      for (int c = 0; c < n; ++c) {
        const int e = c * 64 + lane, st = e / n, k = e - st * n;
        if (e < lim) base[e] = rows[k * RS + st];
      }
    }
  } else if (state0 + lane < B) {
#pragma unroll 8
    for (int k = 0; k < n; ++k) dst[(long)k * L.sk + (state0 + lane) * L.sb] = rows[k * RS + lane];
  }
}


template <typename T> struct Kin { T R[9], p[3], Tw[6], av[6]; };
template <typename T> struct Hand { T I[21], p[6]; };

// ---- limbs in lockstep -----------------------------------------------------------------------------------------------------------
// An op of the walk may stand for TWO bodies (rbd_plan::PAIR, rbd_jit.hip: merge_limbs): the same place in two sibling subtrees of the same shape, e.g. the left
// and the right knee.  Such an op computes in V = f2 — the first body in the low halves, its partner in the high halves of register pairs — and every arithmetic
// instruction of it is a packed one (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32: the issue slot of the scalar form, two bodies' worth of work; operands broadcast
// from either half for free; scripts/ubench/pk_ubench.hip).  A lone wavefront per SIMD is bound by the instructions it issues, so a humanoid's 26 limb bodies cost
// 13.  The code of an op is written once, generic in V; what differs is how a value is fetched (two LDS rows / two registers for a pair) and its constants.
template <typename V> struct NLanes { enum { N = 1 }; };
template <> struct NLanes<f2> { enum { N = 2 }; };
template <typename T, bool PR> struct PairOf { using type = T; };
template <> struct PairOf<float, true> { using type = f2; };
template <typename V, typename T> RBD_DEV V widen(T x) {  // both limbs start from their common parent's value
  if constexpr (NLanes<V>::N == 2) return V{x, x}; else return x;
}
template <typename V, typename T> RBD_DEV V mk2(T a, T b) {  // (first body, partner)
  if constexpr (NLanes<V>::N == 2) return V{a, b}; else return a;
}
// a value of either kind as V (a pair seen as one body: its first — only where a branch that is never taken must still compile)
template <typename V, typename U> RBD_DEV V conv(U x) {
  if constexpr (NLanes<V>::N == NLanes<U>::N) return x;
  else if constexpr (NLanes<V>::N == 2) return V{x, x};
  else return x.x;
}
RBD_DEV float hsum(f2 x) { return x.x + x.y; }  // what the two limbs hand their common parent
RBD_DEV float hsum(float x) { return x; }
RBD_DEV double hsum(double x) { return x; }
// LDS rows / registers of the op's body (and of its partner's)
template <typename V, typename S> RBD_DEV V rd2(const S* col, int ra, int rb) {
  if constexpr (NLanes<V>::N == 2) return V{col[ra * RS], col[rb * RS]}; else return col[ra * RS];
}
template <typename V, typename S> RBD_DEV void wr2(S* col, int ra, int rb, V x) {
  if constexpr (NLanes<V>::N == 2) { col[ra * RS] = x.x; col[rb * RS] = x.y; } else col[ra * RS] = x;
}
// constant K of op O's body record (canonical frames: TR_*), and what kind of number it is: 0, 1, -1 or any other (2).  A pair's constant is the pair of the
// two bodies' constants.
template <typename V, int O, int K> RBD_DEV V cst() {
  // (a pair of literals reaches a packed instruction through a scalar register pair: two s_mov.  The plan's constants as a table in the constant address space,
  // read with s_load several pairs at a time, measured 1 us SLOWER on Atlas: the loads' latency is exposed to a lone wavefront.)
  if constexpr (NLanes<V>::N == 2) return V{(float)P::TR[O][K], (float)P::TR2[O][K]}; else return V(P::TR[O][K]);
}
template <typename V, int O, int K> constexpr int ccls() {
  constexpr double a = P::TR[O][K], b = NLanes<V>::N == 2 ? P::TR2[O][K] : P::TR[O][K];
  return (a == 0 && b == 0) ? 0 : (a == 1 && b == 1) ? 1 : (a == -1 && b == -1) ? -1 : 2;
}
// acc + (constant K) x with the arithmetic the constant needs: none for 0, an addition for +-1 (the frames of a mechanism built from axis-aligned joints are
// signed permutations: x * 0 and x * 1 are not the compiler's to drop under IEEE rules, they are dropped here).  Sums start from -0.0: -0.0 + x IS x.
template <typename V, int O, int K> RBD_DEV void cterm(V& acc, V x) {
  constexpr int c = ccls<V, O, K>();
  if constexpr (c == 1) acc += x;
  else if constexpr (c == -1) acc -= x;
  else if constexpr (c == 2) acc += cst<V, O, K>() * x;
}
template <typename V, int O, int K0, int K1, int K2> RBD_DEV V lin3(V x0, V x1, V x2) {  // (constant K0) x0 + (constant K1) x1 + (constant K2) x2
  V acc = V(-0.0f);
  cterm<V, O, K0>(acc, x0);
  cterm<V, O, K1>(acc, x1);
  cterm<V, O, K2>(acc, x2);
  return acc;
}
template <typename V, int O, int K0, int K1, int K2> constexpr bool lin3_zero() { return ccls<V, O, K0>() == 0 && ccls<V, O, K1>() == 0 && ccls<V, O, K2>() == 0; }

// the coordinates of op O's 1-dof joint as values: c0 = the angle / the displacement / the sine of a sin-cos joint, c1 = its cosine
template <typename V> struct JointQ { V c0, c1; };
template <typename V, int O, typename S> RBD_DEV JointQ<V> joint_q(const S* qs) {
  constexpr int jt = P::OPW[O][0] >> 16, qa = P::OPW[O][1], qb = P::OPW2[O][1];
  JointQ<V> J;
  J.c0 = J.c1 = V(0.0f);
  if constexpr (jt != RBD_JOINT_FIXED) J.c0 = rd2<V>(qs, qa, qb);
  if constexpr (jt == RBD_JOINT_SINCOS_REVOLUTE) J.c1 = rd2<V>(qs, qa + 1, qb + 1);
  return J;
}
template <typename V, int jt> RBD_DEV void joint_sincos(const JointQ<V>& J, V& s, V& c) {
  if constexpr (jt == RBD_JOINT_REVOLUTE) sincos_fast(J.c0, &s, &c);
  else { s = J.c0; c = J.c1; }
}
// (Rn, pn) = (R, p) o joint_to_predecessor o joint_transform(q) for a 1-dof or fixed joint in canonical frames (joint axis +z), in the order that lets the
// constants fold: A = R C, then A Rz(q) (twelve multiplications) — C q-independent, in most mechanisms a signed permutation
template <typename V, int O> RBD_DEV void compose_1dof(const JointQ<V>& J, const V* R, const V* p, V* Rn, V* pn) {
  constexpr int jt = P::OPW[O][0] >> 16;
  V A[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    A[3 * i] = lin3<V, O, TR_C + 0, TR_C + 3, TR_C + 6>(R[3 * i], R[3 * i + 1], R[3 * i + 2]);
    A[3 * i + 1] = lin3<V, O, TR_C + 1, TR_C + 4, TR_C + 7>(R[3 * i], R[3 * i + 1], R[3 * i + 2]);
    A[3 * i + 2] = lin3<V, O, TR_C + 2, TR_C + 5, TR_C + 8>(R[3 * i], R[3 * i + 1], R[3 * i + 2]);
    pn[i] = p[i] + lin3<V, O, TR_PP, TR_PP + 1, TR_PP + 2>(R[3 * i], R[3 * i + 1], R[3 * i + 2]);
  }
  if constexpr (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_SINCOS_REVOLUTE) {
    V s, c;
    joint_sincos<V, jt>(J, s, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Rn[3 * i] = c * A[3 * i] + s * A[3 * i + 1];
      Rn[3 * i + 1] = c * A[3 * i + 1] - s * A[3 * i];
      Rn[3 * i + 2] = A[3 * i + 2];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rn[k] = A[k];
    if constexpr (jt == RBD_JOINT_PRISMATIC) {
#pragma unroll
      for (int i = 0; i < 3; ++i) pn[i] += J.c0 * A[3 * i + 2];
    }
  }
}
// the inverse: from the body's (R, p) back to its parent's.  Rp = R Rl' = (R Rz') C', pp = p - Rp pl
template <typename V, int O> RBD_DEV void uncompose_1dof(const JointQ<V>& J, V* R, V* p) {
  constexpr int jt = P::OPW[O][0] >> 16;
  V Bm[9], Rp[9];
  if constexpr (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_SINCOS_REVOLUTE) {
    V s, c;
    joint_sincos<V, jt>(J, s, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Bm[3 * i] = c * R[3 * i] - s * R[3 * i + 1];
      Bm[3 * i + 1] = s * R[3 * i] + c * R[3 * i + 1];
      Bm[3 * i + 2] = R[3 * i + 2];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) Bm[k] = R[k];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Rp[3 * i] = lin3<V, O, TR_C + 0, TR_C + 1, TR_C + 2>(Bm[3 * i], Bm[3 * i + 1], Bm[3 * i + 2]);
    Rp[3 * i + 1] = lin3<V, O, TR_C + 3, TR_C + 4, TR_C + 5>(Bm[3 * i], Bm[3 * i + 1], Bm[3 * i + 2]);
    Rp[3 * i + 2] = lin3<V, O, TR_C + 6, TR_C + 7, TR_C + 8>(Bm[3 * i], Bm[3 * i + 1], Bm[3 * i + 2]);
  }
  if constexpr (jt == RBD_JOINT_PRISMATIC) {  // pl = pp + d C[:, 2]:  Rp pl = Rp pp + d Bm[:, 2]  (Rp C[:, 2] = Bm C' C e_z = Bm[:, 2])
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] -= J.c0 * Bm[3 * i + 2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    p[i] -= lin3<V, O, TR_PP, TR_PP + 1, TR_PP + 2>(Rp[3 * i], Rp[3 * i + 1], Rp[3 * i + 2]);
    R[3 * i] = Rp[3 * i]; R[3 * i + 1] = Rp[3 * i + 1]; R[3 * i + 2] = Rp[3 * i + 2];
  }
}
// transform(inertia, H) (src/spatial/motion_force_interaction.jl:160-176) with op O's body constants folded (inertia_to_root of rbd_device.hpp, same formulas)
template <typename V, int O> RBD_DEV void inertia_to_root_c(const V* R, const V* p, RInertia<V>& Out) {
  const V m = cst<V, O, TR_M>();
  V Rmc[3], mp[3], Y[6];
  constexpr bool mcz = lin3_zero<V, O, TR_MC, TR_MC + 1, TR_MC + 2>();  // centre of mass at the frame's origin
#pragma unroll
  for (int k = 0; k < 3; ++k) { Rmc[k] = lin3<V, O, TR_MC, TR_MC + 1, TR_MC + 2>(R[3 * k], R[3 * k + 1], R[3 * k + 2]); mp[k] = m * p[k]; }
  if constexpr (mcz) {
    Y[0] = mp[0] * p[0]; Y[1] = mp[0] * p[1]; Y[2] = mp[0] * p[2]; Y[3] = mp[1] * p[1]; Y[4] = mp[1] * p[2]; Y[5] = mp[2] * p[2];
  } else {
    Y[0] = 2 * Rmc[0] * p[0] + mp[0] * p[0];
    Y[1] = Rmc[0] * p[1] + Rmc[1] * p[0] + mp[0] * p[1];
    Y[2] = Rmc[0] * p[2] + Rmc[2] * p[0] + mp[0] * p[2];
    Y[3] = 2 * Rmc[1] * p[1] + mp[1] * p[1];
    Y[4] = Rmc[1] * p[2] + Rmc[2] * p[1] + mp[1] * p[2];
    Y[5] = 2 * Rmc[2] * p[2] + mp[2] * p[2];
  }
  const V trY = Y[0] + Y[3] + Y[5];
  V RJ[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    RJ[3 * i] = lin3<V, O, TR_J + 0, TR_J + 1, TR_J + 2>(R[3 * i], R[3 * i + 1], R[3 * i + 2]);
    RJ[3 * i + 1] = lin3<V, O, TR_J + 1, TR_J + 3, TR_J + 4>(R[3 * i], R[3 * i + 1], R[3 * i + 2]);
    RJ[3 * i + 2] = lin3<V, O, TR_J + 2, TR_J + 4, TR_J + 5>(R[3 * i], R[3 * i + 1], R[3 * i + 2]);
  }
  V A[6];
  A[0] = RJ[0] * R[0] + RJ[1] * R[1] + RJ[2] * R[2];
  A[1] = RJ[0] * R[3] + RJ[1] * R[4] + RJ[2] * R[5];
  A[2] = RJ[0] * R[6] + RJ[1] * R[7] + RJ[2] * R[8];
  A[3] = RJ[3] * R[3] + RJ[4] * R[4] + RJ[5] * R[5];
  A[4] = RJ[3] * R[6] + RJ[4] * R[7] + RJ[5] * R[8];
  A[5] = RJ[6] * R[6] + RJ[7] * R[7] + RJ[8] * R[8];
  Out.J[0] = A[0] - Y[0] + trY; Out.J[1] = A[1] - Y[1]; Out.J[2] = A[2] - Y[2];
  Out.J[3] = A[3] - Y[3] + trY; Out.J[4] = A[4] - Y[4]; Out.J[5] = A[5] - Y[5] + trY;
#pragma unroll
  for (int k = 0; k < 3; ++k) Out.c[k] = mcz ? mp[k] : Rmc[k] + mp[k];
  Out.m = m;
}
// motion subspace column of a revolute / prismatic joint from the body's transform (axis +z of the canonical frame)
template <typename V, int jt> RBD_DEV void subspace_1dof(const V* R, const V* p, V* S) {
  if constexpr (jt == RBD_JOINT_PRISMATIC) { S[0] = S[1] = S[2] = V(0.0f); S[3] = R[2]; S[4] = R[5]; S[5] = R[8]; }
  else { S[0] = R[2]; S[1] = R[5]; S[2] = R[8]; cross3(p, S, S + 3); }
}
constexpr int next_enter(int o) {  // the first ENTER op after op o, or -1
  for (int k = o + 1; k < P::NOPS; ++k)
    if ((P::OPW[k][0] & 0xff) == SK_ENTER) return k;
  return -1;
}
constexpr bool jt_1dof(int jt) { return jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_PRISMATIC || jt == RBD_JOINT_SINCOS_REVOLUTE; }


// mass_matrix! (src/mechanism_algorithms.jl:248-272) of rbd_plan's mechanism, one lane per state.  Mout: any Layout (the caller's SOA
// buffer, or the staging buffer grouped by 16 states the tile Cholesky reads); zero_fill: also write the structural zeros of the lower triangle.
// PERMUTED (the staging buffer of chol_spec below): entry (row, col) goes to (max, min) of (PERM[row], PERM[col]).
template <typename T, bool PERMUTED = false>
RBD_DEV void crba_spec(long B, const T* __restrict__ q, T* __restrict__ Mout, Layout Lq, Layout Lm, int zero_fill, T* lds) {
  constexpr int ML = P::NLEVELS, NQ = P::NQ, NV = P::NV;
  using T2 = typename PairOf<T, true>::type;  // limbs in lockstep (fp32): the value type of an op that stands for two bodies
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long state0 = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 64;
  if (state0 >= B) return;  // (no workgroup barrier below: the wavefronts of a block share nothing but the launch)
  const long state_raw = state0 + lane;
  const bool live = state_raw < B;
  const long state = live ? state_raw : B - 1;
  T* qrows = lds + (size_t)wave * NQ * RS;
  rows_in<T, NQ>(q, Lq, state0, B, qrows);  // (a state-major q arrives in whole runs, not in 64 pieces of one value per load)
  const T* qs = qrows + lane;
  // byte offset of this lane's column; an entry adds a wave-uniform (row, col) term.  32-bit offsets (the host keeps buffers of 4 GB and more
  // away from this kernel): one scalar multiply and one vector add per store, scalar base address
  const unsigned lane_off = (unsigned)(layout_base(Lm, state) * (long)sizeof(T));
  const unsigned mskb = (unsigned)(Lm.sk * (long)sizeof(T));
  auto put = [&](int row, int col, T x) __attribute__((always_inline)) {
#ifdef RBD_SPEC_EMIT
    if constexpr (PERMUTED) {
      const int pr = P::PERM[row], pc = P::PERM[col];
      row = pr > pc ? pr : pc;
      col = pr > pc ? pc : pr;
    }
#endif
    if (live) *reinterpret_cast<T*>(reinterpret_cast<char*>(Mout) + (unsigned long)(lane_off + (unsigned)(col * NV + row) * mskb)) = x;
  };
  if (zero_fill) {
    sfor<NV>([&](auto rc) __attribute__((always_inline)) {
      constexpr int row = rc.value;
      sfor<row + 1>([&](auto cc) __attribute__((always_inline)) {
        constexpr int col = cc.value;
        if constexpr (!((P::ROWMASK[row] >> col) & 1ull)) put(row, col, T(0));
      });
    });
  }
  wave_sync();  // the staged rows are this wavefront's own
  // the path from the root to the body the walk is at, level by level: transforms to root (R row-major, p), inertias being accumulated (J 6, c 3, m), motion
  // subspace columns (1-dof joints) — once for the levels whose op stands for one body, once (f2: packed arithmetic, aba_spec above) for those inside a pair of limbs
  T X1[ML][12], IC1[ML][10], S1[ML][6];
  T2 X2[ML][12], IC2[ML][10], S2[ML][6];
  sfor<P::NOPS>([&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value, w0 = P::OPW[O][0], kind = w0 & 0xff, lvl = (w0 >> 8) & 0xff, jt = w0 >> 16, voff = P::OPW[O][2], voff2 = P::OPW2[O][2];
    constexpr bool PR = P::PAIR[O] != 0, ROOT = P::PROOT[O] != 0;
    using V = typename PairOf<T, PR>::type;
    auto& X = [&]() -> auto& { if constexpr (PR) return X2; else return X1; }();
    auto& IC = [&]() -> auto& { if constexpr (PR) return IC2; else return IC1; }();
    auto& S = [&]() -> auto& { if constexpr (PR) return S2; else return S1; }();
    // entry (this op's coordinate, column col) — for a pair the partner's entry (its coordinate, its column col2) rides in the high half
    auto put2 = [&](int row, int row2, int col, int col2, V x) __attribute__((always_inline)) {
      if constexpr (PR) { put(row, col, x.x); put(row2, col2, x.y); } else put(row, col, x);
    };
    if constexpr (kind == SK_ENTER) {
      V* R = X[lvl];
      V* p = X[lvl] + 9;
      if constexpr (jt_1dof(jt) || jt == RBD_JOINT_FIXED) {
        V Rp[9], pp[3];  // the parent's transform: the level above (in both halves when the limbs start here), or the world's
        if constexpr (lvl == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) Rp[k] = (k % 4 == 0) ? V(1) : V(0);
#pragma unroll
          for (int k = 0; k < 3; ++k) pp[k] = V(0);
        } else if constexpr (ROOT) {
#pragma unroll
          for (int k = 0; k < 9; ++k) Rp[k] = widen<V>(X1[lvl - 1][k]);
#pragma unroll
          for (int k = 0; k < 3; ++k) pp[k] = widen<V>(X1[lvl - 1][9 + k]);
        } else {
#pragma unroll
          for (int k = 0; k < 9; ++k) Rp[k] = X[lvl - 1][k];
#pragma unroll
          for (int k = 0; k < 3; ++k) pp[k] = X[lvl - 1][9 + k];
        }
        compose_1dof<V, O>(joint_q<V, O>(qs), Rp, pp, R, p);
      } else {
        T Rl[9], pl[3];
        local_transform<T, O, RS>(qs, Rl, pl);
        if constexpr (lvl == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) R[k] = Rl[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) p[k] = pl[k];
        } else {
          T t[3];
          matmul3(X[lvl - 1], Rl, R);
          matvec3(X[lvl - 1], pl, t);
#pragma unroll
          for (int k = 0; k < 3; ++k) p[k] = X[lvl - 1][9 + k] + t[k];
        }
      }
      if constexpr (jt_1dof(jt)) subspace_1dof<V, jt>(R, p, S[lvl]);
      RInertia<V> Ib;
      inertia_to_root_c<V, O>(R, p, Ib);
#pragma unroll
      for (int k = 0; k < 6; ++k) IC[lvl][k] = Ib.J[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) IC[lvl][6 + k] = Ib.c[k];
      IC[lvl][9] = Ib.m;
    } else {
      RInertia<V> Ic;
#pragma unroll
      for (int k = 0; k < 6; ++k) Ic.J[k] = IC[lvl][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Ic.c[k] = IC[lvl][6 + k];
      Ic.m = IC[lvl][9];
      if constexpr (lvl > 0) {
        if constexpr (ROOT) {  // the two limbs' composite inertias reach their common parent as their sum
#pragma unroll
          for (int k = 0; k < 10; ++k) IC1[lvl - 1][k] += hsum(IC[lvl][k]);
        } else {
#pragma unroll
          for (int k = 0; k < 10; ++k) IC[lvl - 1][k] += IC[lvl][k];
        }
      }
      // this body's columns against an ancestor's: one entry for a 1-dof joint, the joint's components of X_anc^-T F for one with several.  For a pair: the
      // ancestors inside the limbs are pairs themselves (their own columns each), those above the limbs are common to both
      auto ancestors = [&](auto rowc, auto row2c, const V* F) __attribute__((always_inline)) {
        constexpr int row = rowc.value, row2 = row2c.value;
        sfor<lvl>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = kc.value, col = P::COLS[O][k], col2 = PR ? P::COLS2[O][k] : col;
          constexpr bool inside = PR && col2 != col;  // (an ancestor of both limbs has ONE column)
          if constexpr (col >= 0) {
            if constexpr ((col & SC_MULTI) != 0) {  // (never inside a limb: limbs hold 1-dof and fixed joints only)
              constexpr int ajt = jt_of_col(col);
              V Xa[12], o6[6];
#pragma unroll
              for (int i = 0; i < 12; ++i) Xa[i] = widen<V>(X1[k][i]);
              xforce_inv(Xa, Xa + 9, F, o6);
#pragma unroll
              for (int cj = 0; cj < nvj_of(ajt); ++cj) put2(row, row2, (col & ~SC_MULTI) + cj, (col & ~SC_MULTI) + cj, o6[comp_of(ajt, cj)]);
            } else {
              V Sa[6];
#pragma unroll
              for (int i = 0; i < 6; ++i) {
                if constexpr (inside) Sa[i] = conv<V>(S2[k][i]); else Sa[i] = conv<V>(S1[k][i]);
              }
              put2(row, row2, col, inside ? col2 : col, dot6(F, Sa));
            }
          }
        });
      };
      if constexpr (nvj_of(jt) > 1) {  // the n x n block S' Ic S with S = X(H) E, then the ancestors, column by column
        sfor<nvj_of(jt)>([&](auto cic) __attribute__((always_inline)) {
          constexpr int ci = cic.value;
          T e[6], Si[6], Fc[6], o6[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) e[k] = (k == comp_of(jt, ci)) ? T(1) : T(0);
          xmotion(X[lvl], X[lvl] + 9, e, Si);
          mul_inertia(Ic, Si, Fc);
          xforce_inv(X[lvl], X[lvl] + 9, Fc, o6);
#pragma unroll
          for (int cj = 0; cj <= ci; ++cj) put(voff + ci, voff + cj, o6[comp_of(jt, cj)]);
          ancestors(Ix<voff + ci>{}, Ix<voff + ci>{}, Fc);
        });
      } else if constexpr (jt != RBD_JOINT_FIXED) {
        V F[6];
        mul_inertia(Ic, S[lvl], F);
        put2(voff, voff2, voff, voff2, dot6(F, S[lvl]));
        ancestors(Ix<voff>{}, Ix<voff2>{}, F);
      }
    }
  });
}


#ifdef RBD_SPEC_KIN
// ---------------------------------------------------------------------------------------------------------------------------------
// The kinematics by-products of the forward-kinematics pass (round 6; SURVEY.md §8 f3 — what the reference benchmarks beside the dynamics,
// perf/runbenchmarks.jl:69-110), one lane per state, compiled for rbd_plan's mechanism — the same depth-first walk as crba_spec above, every tree joint type:
//   WHAT = 0  momentum_matrix!(A, state) (src/mechanism_algorithms.jl:313-327: column i = crb_inertia(body(i)) S_i) and, from the same inertias,
//             center_of_mass (:28-50)                                                                                 -> A_out, com_out (nullable)
//   WHAT = 3  kinetic_energy / gravitational_potential_energy (src/mechanism_state.jl:886-903) and center_of_mass     -> energy_out, com_out (nullable)
//   WHAT = 4  center_of_mass alone (transforms and the bodies' first mass moments: the lightest walk)                    -> com_out
//   WHAT = 1  geometric_jacobian!(J, state, path) (:80-99): +S on the joints walked down to the target, -S on those walked up from the base (jplus / jminus: one
//             bit per body in depth-first order — the order of the ENTER ops), zero elsewhere                          -> J_out
//   WHAT = 2  momentum(state), momentum_rate_bias(state) (src/mechanism_state.jl:975-987): Σ I_b T_b and Σ I_b A_b + T_b x* I_b T_b with the bias accelerations
//             of update_bias_accelerations_wrt_world! (:814-830, no gravity term)                                       -> mom_out (12 per state)
// Why a kernel of its own: the lane-per-body kin_kernel (rbd_kernels.hip) keeps a state on 32 lanes of which a level sweep uses a handful — at 65 536 fp64 Atlas
// states rbd_kinematics took 207 us, rbd_geometric_jacobian 149, rbd_momentum 164 for 154 + 120 + 44 MB of compulsory traffic (profiles/r06_kernel_stats_kin.csv, first
// measurement of this row).  Here every vector instruction works on 64 states, q and v arrive through LDS rows in whole runs, and a column of A / J leaves as the
// lane's own 48 contiguous bytes (three 16-byte stores in fp64).  What is wanted is a TEMPLATE parameter, never a test of a pointer inside the walk: with run-time
// flags there (`v ? row : 0`, `if (A_out)`) the compiler turned every LDS read of the v rows into a speculated load at the top of the kernel and spilled what it
// had read — 564 spilled registers for Atlas in fp64 (2.2 KB of scratch per lane); as four instantiations the same walks take 130 - 400 registers and none.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kin_slot_of(int O) {  // depth-first ordinal of op O's body = the number of ENTER ops before it
  int n = 0;
  for (int o = 0; o < O; ++o) n += ((P::OPW[o][0] & 0xff) == SK_ENTER) ? 1 : 0;
  return n;
}
template <typename T, int WHAT>
RBD_DEV void kin_spec(long B, const T* __restrict__ q, const T* __restrict__ v, T* __restrict__ A_out, T* __restrict__ com_out, T* __restrict__ energy_out,
                      T* __restrict__ J_out, unsigned long long jplus, unsigned long long jminus, T* __restrict__ mom_out, Layout Lq, Layout Lv, Layout La, Layout L3,
                      Layout L2, Layout L12, T gx, T gy, T gz, T* lds) {
  constexpr int ML = P::NLEVELS, NQ = P::NQ, NV = P::NV;
  constexpr bool TWISTS = WHAT == 2 || WHAT == 3, INERTIAS = WHAT != 1, CRB = WHAT == 0;  // what the walk carries: twists (v needed), bodies' inertias, their composites
  const int lane = threadIdx.x & 63;
  const long state0 = (long)blockIdx.x * 64;
  if (state0 >= B) return;
  const long state_raw = state0 + lane;
  const bool live = state_raw < B;
  const long state = live ? state_raw : B - 1;
  T* qrows = lds;
  T* vrows = lds + (size_t)NQ * RS;
  rows_in<T, NQ>(q, Lq, state0, B, qrows);
  if constexpr (TWISTS) rows_in<T, NV>(v, Lv, state0, B, vrows);
  wave_sync();
  const T* qs = qrows + lane;
  // a velocity is read where the walk uses it: the rows are written once, so nothing but this `volatile` keeps the compiler from reading all of them at the top
  // of the kernel and carrying — or spilling — them from there (seen: 765 spilled registers in the fp32 momentum walk, none in the fp64 one, and the other way
  // round with a run-time test in front of the read)
  const T* vs = vrows + lane;
  auto vrow = [&](int k) __attribute__((always_inline)) { return *reinterpret_cast<const volatile T*>(vs + k * RS); };
  // a column (6 values) of A or J of this lane's state: 48 contiguous bytes when a state's values are (the caller's n x B column-major matrix: three 16-byte stores
  // in fp64), else a strided run.  (Round 6, measured and taken back: the 64 x 6 values of a column crossing through six LDS rows and leaving as the wavefront's
  // six stores of 64 consecutive values — a sixth of the cache lines per instruction: momentum matrix 58.2 against 56.1 us, Jacobian 44.1 against 46.2 at 65 536
  // fp64 Atlas states; what the walks wait for is not the number of lines their stores touch.)
  auto put_col = [&](T* out, int col, const T* x) __attribute__((always_inline)) {
    if (!live) return;
    T* dst = out + (long)(6 * col) * La.sk + state * La.sb;
    if (La.sk == 1) {
      if constexpr (sizeof(T) == 8) {
        typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < 3; ++k) { d2 w = {(double)x[2 * k], (double)x[2 * k + 1]}; *reinterpret_cast<d2*>(dst + 2 * k) = w; }  // (A / J of a state start on 16 bytes: 6 nv values of 8)
      } else {
        typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < 3; ++k) { f2v w = {(float)x[2 * k], (float)x[2 * k + 1]}; *reinterpret_cast<f2v*>(dst + 2 * k) = w; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[(long)k * La.sk] = x[k];
    }
  };
  // the path from the root to the body the walk is at: transforms to root, twists, bias accelerations, inertias being accumulated, motion subspace columns
  T X[ML][12], TW[ML][6], AB[ML][6], IC[ML][10], S[ML][6];
  T ke = T(0), pe = T(0), ms = T(0), cs[3] = {T(0), T(0), T(0)}, hs[6], ws[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) hs[k] = ws[k] = T(0);
  sfor<P::NOPS>([&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value, w0 = P::OPW[O][0], kind = w0 & 0xff, lvl = (w0 >> 8) & 0xff, jt = w0 >> 16, voff = P::OPW[O][2];
    if constexpr (kind == SK_ENTER) {
      T* R = X[lvl];
      T* p = X[lvl] + 9;
      if constexpr (jt_1dof(jt) || jt == RBD_JOINT_FIXED) {
        T Rp[9], pp[3];
        if constexpr (lvl == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) Rp[k] = (k % 4 == 0) ? T(1) : T(0);
#pragma unroll
          for (int k = 0; k < 3; ++k) pp[k] = T(0);
        } else {
#pragma unroll
          for (int k = 0; k < 9; ++k) Rp[k] = X[lvl - 1][k];
#pragma unroll
          for (int k = 0; k < 3; ++k) pp[k] = X[lvl - 1][9 + k];
        }
        compose_1dof<T, O>(joint_q<T, O>(qs), Rp, pp, R, p);
      } else {
        T Rl[9], pl[3];
        local_transform<T, O, RS>(qs, Rl, pl);
        if constexpr (lvl == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) R[k] = Rl[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) p[k] = pl[k];
        } else {
          T t[3];
          matmul3(X[lvl - 1], Rl, R);
          matvec3(X[lvl - 1], pl, t);
#pragma unroll
          for (int k = 0; k < 3; ++k) p[k] = X[lvl - 1][9 + k] + t[k];
        }
      }
      if constexpr (jt_1dof(jt)) subspace_1dof<T, jt>(R, p, S[lvl]);
      if constexpr (WHAT == 1) {
        // every column is written (fill!(jac, 0) first in the reference): ±S on the path, zeros off it
        constexpr int slot = kin_slot_of(O);
        const T sg = ((jplus >> slot) & 1ull) ? T(1) : ((jminus >> slot) & 1ull) ? T(-1) : T(0);  // (uniform)
        if constexpr (jt_1dof(jt)) {
          T c[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) c[k] = sg * S[lvl][k];
          put_col(J_out, voff, c);
        } else if constexpr (nvj_of(jt) > 1) {
          sfor<nvj_of(jt)>([&](auto cic) __attribute__((always_inline)) {
            constexpr int ci = cic.value;
            T e[6], Si[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) e[k] = (k == comp_of(jt, ci)) ? sg : T(0);
            xmotion(R, p, e, Si);
            put_col(J_out, voff + ci, Si);
          });
        }
      }
      if constexpr (TWISTS) {
        // twist_wrt_world (update_twists_wrt_world!, mechanism_state.jl:769-780) and, for the momentum rate, the bias acceleration A_b = A_p + [T_b, T_b - T_p]
        T vJ[6];
        if constexpr (jt_1dof(jt)) {
          const T qd = vrow(voff);
#pragma unroll
          for (int k = 0; k < 6; ++k) vJ[k] = S[lvl][k] * qd;
        } else if constexpr (nvj_of(jt) > 1) {
          T v6[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) v6[k] = T(0);
#pragma unroll
          for (int k = 0; k < nvj_of(jt); ++k) v6[comp_of(jt, k)] = vrow(voff + k);  // (body_twist)
          xmotion(R, p, v6, vJ);
        } else {
#pragma unroll
          for (int k = 0; k < 6; ++k) vJ[k] = T(0);
        }
        if constexpr (lvl == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) { TW[0][k] = vJ[k]; AB[0][k] = T(0); }
        } else {
#pragma unroll
          for (int k = 0; k < 6; ++k) TW[lvl][k] = TW[lvl - 1][k] + vJ[k];
          if constexpr (WHAT == 2) {
            T cr[6];
            se3_comm(TW[lvl], vJ, cr);
#pragma unroll
            for (int k = 0; k < 6; ++k) AB[lvl][k] = AB[lvl - 1][k] + cr[k];
          }
        }
      }
      if constexpr (INERTIAS) {
        RInertia<T> Ib;
        inertia_to_root_c<T, O>(R, p, Ib);
        if constexpr (CRB) {
#pragma unroll
          for (int k = 0; k < 6; ++k) IC[lvl][k] = Ib.J[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) IC[lvl][6 + k] = Ib.c[k];
          IC[lvl][9] = Ib.m;
        }
        if constexpr (WHAT == 3) {
          T h[6];
          mul_inertia(Ib, TW[lvl], h);
          ke += dot6(h, TW[lvl]) / 2;
        }
        if constexpr (WHAT != 2) {
          if constexpr (P::TR[O][TR_M] > 0.0) {  // (a body without mass has no centre of mass: center_of_mass skips it, mechanism_algorithms.jl:36)
            pe -= gx * Ib.c[0] + gy * Ib.c[1] + gz * Ib.c[2];
            ms += Ib.m;
#pragma unroll
            for (int k = 0; k < 3; ++k) cs[k] += Ib.c[k];
          }
        } else {
          T h[6], Ia[6], x[6];
          mul_inertia(Ib, TW[lvl], h);
          mul_inertia(Ib, AB[lvl], Ia);
          momentum_cross(Ib, TW[lvl], x);
#pragma unroll
          for (int k = 0; k < 6; ++k) { hs[k] += h[k]; ws[k] += Ia[k] + x[k]; }
        }
      }
    } else if constexpr (WHAT == 0) {
      // EXIT: the subtree below is finished — IC[lvl] is the composite inertia (update_crb_inertias!, mechanism_state.jl:852-868); column(s) = Ic S
      RInertia<T> Ic;
#pragma unroll
      for (int k = 0; k < 6; ++k) Ic.J[k] = IC[lvl][k];
#pragma unroll
      for (int k = 0; k < 3; ++k) Ic.c[k] = IC[lvl][6 + k];
      Ic.m = IC[lvl][9];
      if constexpr (lvl > 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) IC[lvl - 1][k] += IC[lvl][k];
      }
      {
        if constexpr (nvj_of(jt) > 1) {
          sfor<nvj_of(jt)>([&](auto cic) __attribute__((always_inline)) {
            constexpr int ci = cic.value;
            T e[6], Si[6], F[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) e[k] = (k == comp_of(jt, ci)) ? T(1) : T(0);
            xmotion(X[lvl], X[lvl] + 9, e, Si);
            mul_inertia(Ic, Si, F);
            put_col(A_out, voff + ci, F);
          });
        } else if constexpr (jt != RBD_JOINT_FIXED) {
          T F[6];
          mul_inertia(Ic, S[lvl], F);
          put_col(A_out, voff, F);
        }
      }
    }
  });
  if (!live) return;
  if constexpr (WHAT == 3) { energy_out[0 * L2.sk + state * L2.sb] = ke; energy_out[1 * L2.sk + state * L2.sb] = pe; }
  if constexpr (WHAT == 0 || WHAT == 3 || WHAT == 4) {
    if (com_out) {
#pragma unroll
      for (int k = 0; k < 3; ++k) com_out[(long)k * L3.sk + state * L3.sb] = cs[k] / ms;
    }
  }
  if constexpr (WHAT == 2) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { mom_out[(long)k * L12.sk + state * L12.sb] = hs[k]; mom_out[(long)(6 + k) * L12.sk + state * L12.sb] = ws[k]; }
  }
}
#endif  // RBD_SPEC_KIN

#ifdef RBD_SPEC_ABA
// ---------------------------------------------------------------------------------------------------------------------------------
// dynamics! (the articulated-body algorithm, src/mechanism_algorithms.jl:845-864 through the world-frame recursion of the other ABA kernels),
// one lane per state, compiled for rbd_plan's mechanism.  The depth-first walk is straight-line code, so everything the walk kernel
// (rbd_walk.hpp) keeps in LDS rows, mailboxes and switch-addressed accumulation registers is here a plain local the allocator places:
//   * ONE kinematic state (transform to root, twist, velocity-product acceleration a_vp with the world's -g folded in) walks down the tree
//     and back up: leaving a body towards its parent the joint is UN-COMPOSED (H_parent = H X_joint^-1, T_parent = T - S q', a_parent =
//     a - [T, S q']) — the walk kernel's device — so that no body's kinematics are kept, not even a branch point's;
//   * bottom-up, a chain body takes the hand-off (Ia = IA - U D^-1 U', pa = pA + U D^-1 u; the bias acceleration is folded into pA = I a_vp +
//     T x* I T - w_ext, so there is no Ia c term) straight from the registers its child left it in; a branch point sums its children's in its slot;
//   * what a body leaves behind for the top-down pass is U D^-1 (6) and D^-1 u (1): three in LDS rows the walk has no more use for, four in
//     registers named by the body; that pass composes the
//     transforms again (S is read off the body's transform), starting later children of a branch point from its saved (transform, a_delta).
// Extra plan tables: BODY[NOPS] (depth-first ordinal of the op's body), NCH[NOPS] (its children), BS[NOPS] (its branch slot, or -1), PBS[NOPS]
// (its parent's branch slot, or -1), CIDX[NOPS] (its rank among its siblings), NBS (slots), NEXT_EXIT[NOPS] (the next EXIT op after this one, or -1).
// q, v, tau of the wavefront's 64 states are staged through LDS rows (stride 65: conflict-free both ways); v̇ and q̇ leave the same way, q̇ first.
// ---------------------------------------------------------------------------------------------------------------------------------
// Template parameters beside the scalar type (both 0 for fp32):
// UDL (0..4): how many of a body's four "register" values of U D^-1 live in LDS rows of their own instead.
// GST: the spare row per body and the ten rows per 3-dof joint — what the bottom-up pass leaves for the top-down pass beside the joint's own v / tau rows — live in
// a batch-innermost HBM stash (value s of state b at stash[s B + b]: one 512-byte run per wavefront and value, NB + 10 N3 values per state) instead of LDS, and
// q̇ / the integrator's next q leave from the lanes themselves instead of through spare rows.
// The fp64 program (round 6: the mechanisms no walk kernel takes) is bound by LDS: with every row in LDS the reference's randmech() needs 236 - 311 rows of 520
// bytes — ONE wavefront per CU, a 50 us chain per round of 16 384 states.  With q, v, tau alone in LDS (122 rows, 63 KB) two wavefronts share a CU, but the chain
// is 85 us (the stash stores and the scratch reloads share one in-order counter).  rbd_jit.hip generates both (aba_spec_f64, aba_spec_gst_f64); rbd_capi.hip
// launches whichever needs less time for the batch (run_aba).
template <typename T, bool FEXT = true, int UDL = 0, int GST = 0>
RBD_DEV void aba_spec(long B, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ tau, const T* __restrict__ fext,
                      T* __restrict__ vdot, T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf, T gx, T gy, T gz, T* lds, const MkStage& F, T* __restrict__ stash = nullptr) {
  constexpr int ABA_UDL = UDL;
  constexpr bool ABA_GST = GST != 0;
  constexpr int ABA_ROWS = P::NQ + 2 * P::NV + (ABA_GST ? 0 : (P::NQ > P::NB ? P::NQ : P::NB) + 10 * P::N3) + ABA_UDL * P::NB;  // q, v, tau, [spare, ten of the 18 values of U D^-1 of every 3-dof joint,] ABA_UDL per body
  constexpr int NQ = P::NQ, NV = P::NV, NB = P::NB, NBS = P::NBS > 0 ? P::NBS : 1, NPR = P::NPAIR > 0 ? P::NPAIR : 1;
  using T2 = typename PairOf<T, true>::type;  // the value type of an op that stands for two bodies
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* rq = lds + (size_t)wave * ABA_ROWS * RS;
  T* rv = rq + NQ * RS;
  T* rt = rv + NV * RS;
  T* rx = rt + NV * RS;  // max(NQ, NB) rows: q̇ on its way out first, then one value per body for the top-down pass (ABA_GST: none — the HBM stash)
  T* r3 = rx + (ABA_GST ? 0 : (NQ > NB ? NQ : NB)) * RS;  // 10 rows per 3-dof joint: the part of its U D^-1 (18 values) that its v rows (3), its spare row and its four registers do not take (ABA_GST: none)
  T* ru = r3 + (ABA_GST ? 0 : 10 * P::N3) * RS;           // ABA_UDL rows per body (fp64: see RBD_SPEC_ABA_UDL)
  const long state0 = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 64;
  if (state0 >= B) return;
#ifdef RBD_SPEC_ABLATE_NO_LOADS  // (timing experiments, RBD_TUNE spec_variant: the passes on made-up rows, nothing read)
  for (int k = 0; k < NQ; ++k) rq[k * RS + lane] = T(0.01f) * T(k + 1) + T(0.001f) * T(lane);
  for (int k = 0; k < NV; ++k) { rv[k * RS + lane] = T(0.02f) * T(k + 1); rt[k * RS + lane] = T(0.5f); }
  rq[0 * RS + lane] = T(1); rq[1 * RS + lane] = rq[2 * RS + lane] = rq[3 * RS + lane] = T(0);
#else
  rows_in<T, NQ>(q, Lq, state0, B, rq);
  if (v) rows_in<T, NV>(v, Lv, state0, B, rv);
  else {  // (the M^-1 rhs pass: v = 0)
#pragma unroll 4
    for (int k = 0; k < NV; ++k) rv[k * RS + lane] = T(0);
  }
  if (tau) rows_in<T, NV>(tau, Lv, state0, B, rt);
  else {
#pragma unroll 4
    for (int k = 0; k < NV; ++k) rt[k * RS + lane] = T(0);
  }
#endif
  wave_sync();
#ifdef RBD_SPEC_ABLATE_STAGING_ONLY  // (timing experiments: the staging of the rows and the way out, no passes)
  if (vdot) rows_out<T, NV>(rt, vdot, Lv, state0, B);
  if (qdot) rows_out<T, NQ>(rq, qdot, Lq, state0, B);
  return;
#endif
  // `simulate`: this launch is stage F.stage of a Munthe-Kaas RK4 step (rbd_mk_fuse.hpp): the next stage's q from the staged rows before the passes, its v
  // from the v̇ rows behind them — the wavefront's own 64 states, no launch of its own
  const T* qs = rq + lane;
  T* vs = rv + lane;
  T* ts = rt + lane;
  T* xs = rx + lane;
  T* x3 = r3 + lane;
  T* us = ru + lane;
  // Here the lane IS the state and the joints are compile-time constants, so the stage is straight-line code like the passes: the base point and the running
  // sums live in the workspace's stage buffers in a layout of this kernel's own — batch-innermost, element (k, state) at k B + state, whatever the caller's
  // layout (stage 0 writes them, stages 1-3 of the same step read them: nobody else sees them) — one coalesced access per lane and value; the next q is formed in
  // the spare rows and leaves through rows_out.  ALL loads first, then the arithmetic and the stores: vmcnt counts loads and stores in order, a load issued
  // behind a store waits for that store's acknowledgement (the element-parallel form of rbd_mk_fuse.hpp, a table lookup in front of every element's loads,
  // cost the launch 32 of its 88 us at 65 536 states: 19 us waiting, 3 200 extra vector instructions per wavefront).
  const long mk_gi = state0 + lane;
  const bool mk_live = mk_gi < B;
  const long mk_si = mk_live ? mk_gi : B - 1;
  const int mk_s = F.stage;
  // what the bottom-up pass leaves per body / per 3-dof joint for the top-down pass: an LDS row of the lane's column, or (ABA_GST) the HBM stash
  const auto gst = as_global(stash);
  auto sx_put = [&](int slot, T x) __attribute__((always_inline)) { if constexpr (ABA_GST) { if (mk_live) gst[(long)slot * B + mk_gi] = x; } else xs[slot * RS] = x; };
  auto sx_get = [&](int slot) __attribute__((always_inline)) -> T { if constexpr (ABA_GST) return gst[(long)slot * B + mk_si]; else return xs[slot * RS]; };
  auto s3_put = [&](int slot, T x) __attribute__((always_inline)) { if constexpr (ABA_GST) { if (mk_live) gst[(long)(NB + slot) * B + mk_gi] = x; } else x3[slot * RS] = x; };
  auto s3_get = [&](int slot) __attribute__((always_inline)) -> T { if constexpr (ABA_GST) return gst[(long)(NB + slot) * B + mk_si]; else return x3[slot * RS]; };
  const T mk_h = (T)F.dt, mk_bs = (mk_s == 0 || mk_s == 3) ? T(1) / T(6) : T(1) / T(3), mk_an = mk_s < 2 ? T(0.5) : T(1);
  if (mk_s >= 0) {  // uniform
    const auto q0b = as_global((T*)F.q0), v0b = as_global((T*)F.v0), apb = as_global((T*)F.accp);  // (global, said in the type: pointers out of a struct come out as FLAT accesses otherwise)
    const auto kp = as_global((const T*)F.kp), kd = as_global((const T*)F.kd), qdes = as_global((const T*)F.qdes);
    T Q0[NQ > 0 ? NQ : 1], AC[NV > 0 ? NV : 1], QD[NQ > 0 ? NQ : 1];
    if (mk_s > 0) {
#pragma unroll
      for (int k = 0; k < NQ; ++k) Q0[k] = q0b[(long)k * B + mk_si];
#pragma unroll
      for (int k = 0; k < NV; ++k) AC[k] = apb[(long)k * B + mk_si];
    }
    if (F.pd) {
#pragma unroll
      for (int k = 0; k < NQ; ++k) QD[k] = qdes ? qdes[(long)k * Lq.sk + mk_si * Lq.sb] : T(0);
    }
    // the next q: through the spare rows and out in whole runs — or (ABA_GST: no spare rows) from the lane itself, over its own state's q
    auto qput = [&](int k, T x) __attribute__((always_inline)) { if constexpr (ABA_GST) { if (mk_live) const_cast<T*>(q)[(long)k * Lq.sk + mk_gi * Lq.sb] = x; } else xs[k * RS] = x; };
    sfor<NB>([&](auto jc) __attribute__((always_inline)) {  // joint by joint (the limbs one after the other: element-wise work, nothing to pair up)
      constexpr int J = jc.value, jt = P::JOINTS[J][0], qoff = P::JOINTS[J][1], voff = P::JOINTS[J][2];
      constexpr int nqj = jt == RBD_JOINT_QUAT_FLOATING ? 7 : jt == RBD_JOINT_QUAT_SPHERICAL ? 4 : jt == RBD_JOINT_PLANAR ? 3 : jt == RBD_JOINT_SINCOS_REVOLUTE ? 2 : jt == RBD_JOINT_FIXED ? 0 : 1;
      constexpr int nvj = nvj_of(jt);
      if constexpr (nvj > 0) {
        T qj[7], vj[6], q0j[7], rate[6], phi[6], qn[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) { qj[k] = k < nqj ? qs[(qoff + (k < nqj ? k : 0)) * RS] : T(0); q0j[k] = qj[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) vj[k] = k < nvj ? vs[(voff + (k < nvj ? k : 0)) * RS] : T(0);
        if (mk_s > 0) {
#pragma unroll
          for (int k = 0; k < nqj; ++k) q0j[k] = Q0[qoff + k];
        } else if (mk_live) {
#pragma unroll
          for (int k = 0; k < nqj; ++k) q0b[(long)(qoff + k) * B + mk_gi] = qj[k];
#pragma unroll
          for (int k = 0; k < nvj; ++k) v0b[(long)(voff + k) * B + mk_gi] = vj[k];
        }
        joint_local_rate<T, 0>(jt, q0j, qj, vj, rate);
#pragma unroll
        for (int k = 0; k < 6; ++k) phi[k] = T(0);
#pragma unroll
        for (int k = 0; k < nvj; ++k) {
          const T sum = (mk_s > 0 ? AC[voff + k] : T(0)) + mk_bs * rate[k];
          if (mk_s < 3 && mk_live) apb[(long)(voff + k) * B + mk_gi] = sum;
          phi[k] = mk_s < 3 ? mk_h * mk_an * rate[k] : mk_h * sum;
        }
        joint_global<T, 0>(jt, q0j, phi, qn);
#pragma unroll
        for (int k = 0; k < nqj; ++k) qput((qoff + k), qn[k]);
        if constexpr (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_PRISMATIC) {
          if (F.pd) ts[voff * RS] -= kp[voff] * (qj[0] - QD[qoff]) + kd[voff] * vj[0];
        }
      }
    });
    wave_sync();
    if constexpr (!ABA_GST) rows_out<T, NQ>(rx, const_cast<T*>(q), Lq, state0, B);  // F.q_state IS the kernel's own q input (its address space is known): this wavefront has read its block, nobody else touches it
    wave_sync();  // (the spare rows are used again below; the PD law wrote into the τ rows)
  }
  // q̇ (configuration_derivative!, src/mechanism_state.jl:905-910) depends on q and v alone: assembled in the spare rows and sent off before the
  // passes start (its stores drain while they run; the rows are free again long before pass 2 writes them)
  if (qdot) {
    auto qdput = [&](int k, T x) __attribute__((always_inline)) { if constexpr (ABA_GST) { if (mk_live) qdot[(long)k * Lq.sk + mk_gi * Lq.sb] = x; } else xs[k * RS] = x; };
    sfor<NB>([&](auto jc) __attribute__((always_inline)) {
      constexpr int J = jc.value, jt = P::JOINTS[J][0], qoff = P::JOINTS[J][1], voff = P::JOINTS[J][2];
      if constexpr (jt == RBD_JOINT_QUAT_FLOATING) {  // velocity_to_configuration_derivative! (quaternion_floating.jl:126-136)
        T v6[6], Rq[9], lin[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) v6[k] = vs[(voff + k) * RS];
        const T qw = qs[qoff * RS], qx = qs[(qoff + 1) * RS], qy = qs[(qoff + 2) * RS], qz = qs[(qoff + 3) * RS];
        qdput(qoff, (-qx * v6[0] - qy * v6[1] - qz * v6[2]) / 2);
        qdput((qoff + 1), (qw * v6[0] - qz * v6[1] + qy * v6[2]) / 2);
        qdput((qoff + 2), (qz * v6[0] + qw * v6[1] - qx * v6[2]) / 2);
        qdput((qoff + 3), (-qy * v6[0] + qx * v6[1] + qw * v6[2]) / 2);
        rot_quat(qw, qx, qy, qz, Rq);
        matvec3(Rq, v6 + 3, lin);
#pragma unroll
        for (int k = 0; k < 3; ++k) qdput((qoff + 4 + k), lin[k]);
      } else if constexpr (jt == RBD_JOINT_SINCOS_REVOLUTE) {  // d/dt (sin, cos) = (cos, -sin) q'
        const T qd = vs[voff * RS];
        qdput(qoff, qs[(qoff + 1) * RS] * qd);
        qdput((qoff + 1), -qs[qoff * RS] * qd);
      } else if constexpr (jt == RBD_JOINT_QUAT_SPHERICAL) {  // quaternion_spherical.jl velocity_to_configuration_derivative!
        const T w0 = vs[voff * RS], w1 = vs[(voff + 1) * RS], w2 = vs[(voff + 2) * RS];
        const T qw = qs[qoff * RS], qx = qs[(qoff + 1) * RS], qy = qs[(qoff + 2) * RS], qz = qs[(qoff + 3) * RS];
        qdput(qoff, (-qx * w0 - qy * w1 - qz * w2) / 2);
        qdput((qoff + 1), (qw * w0 - qz * w1 + qy * w2) / 2);
        qdput((qoff + 2), (qz * w0 + qw * w1 - qx * w2) / 2);
        qdput((qoff + 3), (-qy * w0 + qx * w1 + qw * w2) / 2);
      } else if constexpr (jt == RBD_JOINT_PLANAR) {  // planar.jl velocity_to_configuration_derivative!: q̇_lin = Rot2(θ) v_lin
        T sn, cs;
        sincos_fast(qs[(qoff + 2) * RS], &sn, &cs);
        const T vx = vs[voff * RS], vy = vs[(voff + 1) * RS];
        qdput(qoff, cs * vx - sn * vy);
        qdput((qoff + 1), sn * vx + cs * vy);
        qdput((qoff + 2), vs[(voff + 2) * RS]);
      } else if constexpr (jt != RBD_JOINT_FIXED) {
        qdput(qoff, vs[voff * RS]);
      }
    });
    wave_sync();
    if constexpr (!ABA_GST) rows_out<T, NQ>(rx, qdot, Lq, state0, B);
    wave_sync();
  }
  const long sc = state0 + lane < B ? state0 + lane : B - 1;
  const auto fel = as_global(fext ? fext + sc * Lf.sb : nullptr);  // (behind the conditional the pointer has lost its address space: FLAT loads, counted on the LDS counter too)
  const long fsk = Lf.sk;
  // the world's acceleration: -g (mechanism_algorithms.jl:396); a kernel argument, not the plan's constant: M^-1 rhs is this pass with g = 0 (rbd_mass_matrix_solve)
  const T a0[6] = {T(0), T(0), T(0), -gx, -gy, -gz};

  // the walk's state, once for the ops that stand for one body and once for those that stand for two (the allocator keeps what is live)
  Kin<T> K1;  Kin<T2> K2;       // the body the walk is at
  Hand<T> C1; Hand<T2> C2;      // hand-off of the child just finished, on its way to a chain parent
  Hand<T> SH1[NBS]; Hand<T2> SH2[NBS];  // branch points: the sum of their children's hand-offs
  // per body, for the top-down pass: D^-1 u and U D^-1.  The first goes to the body's tau row (read for the last time when u is formed, written again
  // only by that pass), two of the others to its spare row and its v row (free once the joint is un-composed), four stay in registers
  T Ud1[NB][4 - ABA_UDL > 0 ? 4 - ABA_UDL : 1]; T2 Ud2[NPR][4];  // (the first ABA_UDL of a body's four in LDS rows: ud1_put / ud1_get)
  auto ud1_put = [&](int body_, int k, T x) __attribute__((always_inline)) { if (k < ABA_UDL) us[(body_ * ABA_UDL + k) * RS] = x; else Ud1[body_][k - ABA_UDL > 0 ? k - ABA_UDL : 0] = x; };
  auto ud1_get = [&](int body_, int k) __attribute__((always_inline)) -> T { return k < ABA_UDL ? us[(body_ * ABA_UDL + k) * RS] : Ud1[body_][k - ABA_UDL > 0 ? k - ABA_UDL : 0]; };
  // external wrench of the body the next EXIT finishes (asked for one EXIT ahead).  FEXT = false: the instantiation for callers without external wrenches —
  // no registers held for them, no branch per body (the twelve registers decide whether Atlas's limbs fit the file without scratch; a kernel with ANY scratch
  // runs 3.4 times slower here: 144 us against 42, the dispatcher admits fewer wavefronts at a time)
  T fe1[6]; T2 fe2[6];
  auto load_fe = [&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value;
    if constexpr (FEXT && O >= 0) {
      if constexpr (P::PAIR[O] != 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) fe2[k] = fel ? mk2<T2>(fel[(long)(P::OPW[O][3] + k) * fsk], fel[(long)(P::OPW2[O][3] + k) * fsk]) : T2(0.0f);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) fe1[k] = fel ? fel[(long)(P::OPW[O][3] + k) * fsk] : T(0);
      }
    }
  };
  load_fe(Ix<P::FIRST_EXIT>{});

  // ---- passes 1 + 2: down with the kinematics, up with the articulated inertias ----
  sfor<P::NOPS>([&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value, w0 = P::OPW[O][0], kind = w0 & 0xff, lvl = (w0 >> 8) & 0xff, jt = w0 >> 16, voff = P::OPW[O][2], voff2 = P::OPW2[O][2];
    constexpr int body = P::BODY[O], body2 = P::BODY2[O], nch = P::NCH[O], bs = P::BS[O], pbs = P::PBS[O], cidx = P::CIDX[O];
    constexpr bool PR = P::PAIR[O] != 0, ROOT = P::PROOT[O] != 0;
    using V = typename PairOf<T, PR>::type;
    // (asking for the NEXT op's row values while this op computes — with a scheduling barrier to keep the request in place — was built and measured: no gain,
    //  42 us either way; a lone wavefront's time goes to issue slots, 4 cycles per VALU instruction and 5 per packed one, not to the ~380 LDS round trips)
    const JointQ<V> JQ = joint_q<V, O>(qs);
    auto& K = [&]() -> auto& { if constexpr (PR) return K2; else return K1; }();
    auto& C = [&]() -> auto& { if constexpr (PR) return C2; else return C1; }();
    auto& SH = [&]() -> auto& { if constexpr (PR) return SH2; else return SH1; }();
    auto& fe = [&]() -> auto& { if constexpr (PR) return fe2; else return fe1; }();
    auto ud_set = [&](int k, V x) __attribute__((always_inline)) { if constexpr (PR) Ud2[P::PIDX[O]][k] = x; else ud1_put(body, k, x); };
    if constexpr (kind == SK_ENTER) {
      if constexpr (lvl == 0) {  // the parent is the world
#pragma unroll
        for (int k = 0; k < 9; ++k) K.R[k] = (k % 4 == 0) ? V(1) : V(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) K.p[k] = V(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) { K.Tw[k] = V(0); K.av[k] = widen<V>(a0[k]); }
      } else if constexpr (ROOT) {  // two limbs leave their common parent: its kinematic state in both halves (K1 itself stays the parent's until the limbs are done)
#pragma unroll
        for (int k = 0; k < 9; ++k) K2.R[k] = widen<T2>(K1.R[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) K2.p[k] = widen<T2>(K1.p[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) { K2.Tw[k] = widen<T2>(K1.Tw[k]); K2.av[k] = widen<T2>(K1.av[k]); }
      }  // (otherwise K is the parent's: it was just entered, or the sibling finished before this body un-composed its joint)
      V Rn[9], pn[3], vJ[6], cb[6];
      if constexpr (jt_1dof(jt) || jt == RBD_JOINT_FIXED) {
        compose_1dof<V, O>(JQ, K.R, K.p, Rn, pn);
        if constexpr (jt == RBD_JOINT_FIXED) {
#pragma unroll
          for (int k = 0; k < 6; ++k) vJ[k] = V(0);
        } else {
          const V qd = rd2<V>(vs, voff, voff2);
          V S[6];
          subspace_1dof<V, jt>(Rn, pn, S);
#pragma unroll
          for (int k = 0; k < 6; ++k) vJ[k] = S[k] * qd;
        }
      } else {  // joints with several coordinates: one body per op
        T Rl[9], pl[3], t3[3], v6[6];
        local_transform<T, O, RS>(qs, Rl, pl);
        matmul3(K.R, Rl, Rn);
        matvec3(K.R, pl, t3);
#pragma unroll
        for (int k = 0; k < 3; ++k) pn[k] = K.p[k] + t3[k];
        body_twist<T, jt>(vs + voff * RS, RS, v6);
        xmotion(Rn, pn, v6, vJ);  // twist of the joint: X(H) v, v the body-frame twist
      }
      if constexpr (jt != RBD_JOINT_FIXED) {
        se3_comm(K.Tw, vJ, cb);  // [T_parent, vJ]: the bias acceleration increment (mechanism_state.jl:814-830)
#pragma unroll
        for (int k = 0; k < 6; ++k) { K.av[k] += cb[k]; K.Tw[k] += vJ[k]; }
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) K.R[k] = Rn[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) K.p[k] = pn[k];
    } else {
      // the walk is back at this body: K is its kinematic state (un-composed from its only child, restored from its slot, or — a leaf — just entered)
      RInertia<V> I;
      inertia_to_root_c<V, O>(K.R, K.p, I);
      V IA[21], pA[6], h[6];
      mul_inertia(I, K.av, pA);
      momentum_cross(I, K.Tw, h);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if constexpr (FEXT) pA[k] += h[k] - fe[k]; else pA[k] += h[k];
      }
      load_fe(Ix<P::NEXT_EXIT[O]>{});
      sym6_from_inertia(I, IA);
      if constexpr (nch == 1) {
#pragma unroll
        for (int k = 0; k < 21; ++k) IA[k] += C.I[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) pA[k] += C.p[k];
      } else if constexpr (nch >= 2) {
#pragma unroll
        for (int k = 0; k < 21; ++k) IA[k] += SH[bs].I[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) pA[k] += SH[bs].p[k];
      }
      Hand<V> H;
      V S[6], qd = V(0);
      V vJm[6];  // joints with several coordinates: their twist, kept for the un-composition (their v rows are given another use first)
      if constexpr (jt == RBD_JOINT_QUAT_FLOATING) {
        // 6-dof joint: IA a = S^-T tau - pA for the body's own a_delta whatever its parent's is, v̇ = S^-1 (a - a_parent) (S = X(H): the body-frame twist basis
        // seen from the root).  On the world a_parent = 0 and v̇ is final here; below it the top-down pass finishes it.  The parent sees no inertia
        // (IA - U D^-1 U' = 0) and the wrench S^-T tau
        T t6[6], f6[6], ad0[6], vd[6], v6[6];
        body_twist<T, jt>(vs + voff * RS, RS, v6);
        xmotion(K.R, K.p, v6, vJm);
#pragma unroll
        for (int k = 0; k < 6; ++k) t6[k] = ts[(voff + k) * RS];
        xforce(K.R, K.p, t6, f6);
#pragma unroll
        for (int k = 0; k < 21; ++k) H.I[k] = T(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) { H.p[k] = f6[k]; f6[k] -= pA[k]; }
        sym6_solve(IA, f6, ad0);
        if constexpr (lvl == 0) {
          xmotion_inv(K.R, K.p, ad0, vd);
#pragma unroll
          for (int k = 0; k < 6; ++k) ts[(voff + k) * RS] = vd[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) vs[(voff + k) * RS] = ad0[k];  // a_delta waits in the joint's v rows
      } else if constexpr (nvj_of(jt) == 3) {
        // 3-dof joint: U = IA S (6 x 3), D = S'U, u = tau - S'pA; the top-down pass needs D^-1 u (tau rows) and W = U D^-1 (18 rows of the joint's own)
        T v6[6], U[3][6], D[6], Di[6], u[3], du[3], W[3][6];
        body_twist<T, jt>(vs + voff * RS, RS, v6);
        xmotion(K.R, K.p, v6, vJm);
        T Sm[3][6];
        sfor<3>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = kc.value;
          T e[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) e[i] = (i == comp_of(jt, k)) ? T(1) : T(0);
          xmotion(K.R, K.p, e, Sm[k]);
          sym6_mul(IA, Sm[k], U[k]);
          u[k] = ts[(voff + k) * RS] - dot6(Sm[k], pA);
        });
        D[0] = dot6(Sm[0], U[0]); D[1] = dot6(Sm[0], U[1]); D[2] = dot6(Sm[0], U[2]);
        D[3] = dot6(Sm[1], U[1]); D[4] = dot6(Sm[1], U[2]); D[5] = dot6(Sm[2], U[2]);
        sym3_inv(D, Di);
        du[0] = Di[0] * u[0] + Di[1] * u[1] + Di[2] * u[2];
        du[1] = Di[1] * u[0] + Di[3] * u[1] + Di[4] * u[2];
        du[2] = Di[2] * u[0] + Di[4] * u[1] + Di[5] * u[2];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          W[0][i] = U[0][i] * Di[0] + U[1][i] * Di[1] + U[2][i] * Di[2];
          W[1][i] = U[0][i] * Di[1] + U[1][i] * Di[3] + U[2][i] * Di[4];
          W[2][i] = U[0][i] * Di[2] + U[1][i] * Di[4] + U[2][i] * Di[5];
        }
        constexpr int xr = 10 * P::X3[O];
#pragma unroll
        for (int k = 0; k < 3; ++k) ts[(voff + k) * RS] = du[k];
        sfor<18>([&](auto jc) __attribute__((always_inline)) {  // W[j / 6][j % 6]: v rows, spare row, registers, the joint's own rows
          constexpr int j = jc.value;
          const T x = W[j / 6][j % 6];
          if constexpr (j < 3) vs[(voff + j) * RS] = x;
          else if constexpr (j == 3) sx_put(body, x);
          else if constexpr (j < 8) ud1_put(body, j - 4, x);
          else s3_put(xr + j - 8, x);
        });
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) H.I[SI(i, j)] = IA[SI(i, j)] - (W[0][i] * U[0][j] + W[1][i] * U[1][j] + W[2][i] * U[2][j]);
#pragma unroll
        for (int i = 0; i < 6; ++i) H.p[i] = pA[i] + U[0][i] * du[0] + U[1][i] * du[1] + U[2][i] * du[2];
      } else if constexpr (jt == RBD_JOINT_FIXED) {  // S = 0: the body hands its whole inertia up
#pragma unroll
        for (int k = 0; k < 21; ++k) H.I[k] = IA[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) H.p[k] = pA[k];
      } else {
        subspace_1dof<V, jt>(K.R, K.p, S);
        V U[6], W[6];
        sym6_mul(IA, S, U);
        const V Dinv = rcp_hd(dot6(S, U));
        qd = rd2<V>(vs, voff, voff2);
        const V u = (rd2<V>(ts, voff, voff2) - dot6(S, pA)) * Dinv;
#pragma unroll
        for (int k = 0; k < 6; ++k) W[k] = U[k] * Dinv;
        wr2<V>(ts, voff, voff2, u);
        if constexpr (ABA_GST) sx_put(body, hsum(W[0])); else wr2<V>(xs, body, body2, W[0]);  // (ABA_GST: no pairs — hsum of a scalar is itself)
        wr2<V>(vs, voff, voff2, W[1]);
#pragma unroll
        for (int k = 0; k < 4; ++k) ud_set(k, W[2 + k]);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) H.I[SI(i, j)] = IA[SI(i, j)] - W[i] * U[j];
#pragma unroll
        for (int k = 0; k < 6; ++k) H.p[k] = pA[k] + U[k] * u;
      }
      if constexpr (lvl > 0) {
        if constexpr (ROOT) {
          // the two limbs' hand-offs go to their common parent as their sum; K1 is still that parent's kinematic state: nothing to un-compose
          if constexpr (pbs >= 0) {
            if constexpr (cidx == 0) {
#pragma unroll
              for (int k = 0; k < 21; ++k) SH1[pbs].I[k] = hsum(H.I[k]);
#pragma unroll
              for (int k = 0; k < 6; ++k) SH1[pbs].p[k] = hsum(H.p[k]);
            } else {
#pragma unroll
              for (int k = 0; k < 21; ++k) SH1[pbs].I[k] += hsum(H.I[k]);
#pragma unroll
              for (int k = 0; k < 6; ++k) SH1[pbs].p[k] += hsum(H.p[k]);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 21; ++k) C1.I[k] = hsum(H.I[k]);
#pragma unroll
            for (int k = 0; k < 6; ++k) C1.p[k] = hsum(H.p[k]);
          }
        }
        if constexpr (!ROOT) {
          if constexpr (pbs >= 0) {  // the parent is a branch point: its slot sums its children's hand-offs
            if constexpr (cidx == 0) SH[pbs] = H;
            else {
#pragma unroll
              for (int k = 0; k < 21; ++k) SH[pbs].I[k] += H.I[k];
#pragma unroll
              for (int k = 0; k < 6; ++k) SH[pbs].p[k] += H.p[k];
            }
          } else {  // a chain parent: the hand-off stays in registers
            C = H;
          }
        }
        // (where two limbs started, K1 is still their common parent's state and nothing needs un-composing — unless the registers are needed: with external
        //  wrenches in flight K1 is given up while the limbs are walked and comes back from the first halves of K2)
        constexpr bool UNC = !ROOT || FEXT;
        if constexpr (UNC) {
          // back to the parent: the joint is un-composed (a copy of every branch point's kinematics would cost more registers than the file has)
          if constexpr (jt != RBD_JOINT_FIXED) {
            V vJ[6], cb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              if constexpr (nvj_of(jt) > 1) vJ[k] = vJm[k];
              else vJ[k] = S[k] * qd;
            }
            se3_comm(K.Tw, vJ, cb);
#pragma unroll
            for (int k = 0; k < 6; ++k) { K.av[k] -= cb[k]; K.Tw[k] -= vJ[k]; }
          }
          if constexpr (jt_1dof(jt) || jt == RBD_JOINT_FIXED) {
            uncompose_1dof<V, O>(JQ, K.R, K.p);
          } else {
            T Rl[9], pl[3], Rp[9], t3[3];
            local_transform<T, O, RS>(qs, Rl, pl);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j) Rp[3 * i + j] = K.R[3 * i] * Rl[3 * j] + K.R[3 * i + 1] * Rl[3 * j + 1] + K.R[3 * i + 2] * Rl[3 * j + 2];  // R Rl'
            matvec3(Rp, pl, t3);
#pragma unroll
            for (int k = 0; k < 9; ++k) K.R[k] = Rp[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) K.p[k] -= t3[k];
          }
          if constexpr (ROOT) {  // both halves are the common parent's state again
#pragma unroll
            for (int k = 0; k < 9; ++k) K1.R[k] = conv<T>(K.R[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) K1.p[k] = conv<T>(K.p[k]);
#pragma unroll
            for (int k = 0; k < 6; ++k) { K1.Tw[k] = conv<T>(K.Tw[k]); K1.av[k] = conv<T>(K.av[k]); }
          }
        }
      }
    }
  });

  // ---- pass 3: down again with the accelerations ----
  T ad1[6]; T2 ad2[6];
  T SR1[NBS][12], SA1[NBS][6];  // branch points: transform, a_delta
  T2 SR2[NBS][12], SA2[NBS][6];
  sfor<P::NOPS>([&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value, w0 = P::OPW[O][0], kind = w0 & 0xff, lvl = (w0 >> 8) & 0xff, jt = w0 >> 16, voff = P::OPW[O][2], voff2 = P::OPW2[O][2];
    constexpr int body = P::BODY[O], body2 = P::BODY2[O], nch = P::NCH[O], bs = P::BS[O], pbs = P::PBS[O], cidx = P::CIDX[O];
    constexpr bool PR = P::PAIR[O] != 0, ROOT = P::PROOT[O] != 0;
    using V = typename PairOf<T, PR>::type;
    auto& K = [&]() -> auto& { if constexpr (PR) return K2; else return K1; }();
    auto& ad = [&]() -> auto& { if constexpr (PR) return ad2; else return ad1; }();
    auto& SR = [&]() -> auto& { if constexpr (PR) return SR2; else return SR1; }();
    auto& SA = [&]() -> auto& { if constexpr (PR) return SA2; else return SA1; }();
    auto ud_get = [&](int k) __attribute__((always_inline)) -> V { if constexpr (PR) return Ud2[P::PIDX[O]][k]; else return ud1_get(body, k); };
    if constexpr (kind == SK_ENTER) {
      const JointQ<V> JQ = joint_q<V, O>(qs);
      if constexpr (lvl == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) K.R[k] = (k % 4 == 0) ? V(1) : V(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) K.p[k] = V(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) ad[k] = V(0);
      } else if constexpr (ROOT) {  // both limbs from their common parent's transform and a_delta (its slot's, or the live ones when the limbs are its first children)
#pragma unroll
        for (int k = 0; k < 9; ++k) K2.R[k] = widen<T2>(cidx > 0 ? SR1[pbs >= 0 ? pbs : 0][k] : K1.R[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) K2.p[k] = widen<T2>(cidx > 0 ? SR1[pbs >= 0 ? pbs : 0][9 + k] : K1.p[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) ad2[k] = widen<T2>(cidx > 0 ? SA1[pbs >= 0 ? pbs : 0][k] : ad1[k]);
      } else if constexpr (cidx > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) K.R[k] = SR[pbs][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) K.p[k] = SR[pbs][9 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) ad[k] = SA[pbs][k];
      }
      if constexpr (nch > 0 || (jt != RBD_JOINT_FIXED && !(jt == RBD_JOINT_QUAT_FLOATING && lvl == 0))) {  // (a leaf on a fixed joint, or on a 6-dof joint on the world, has nothing left to do)
        if constexpr (jt_1dof(jt) || jt == RBD_JOINT_FIXED) {
          V Rn[9], pn[3];
          compose_1dof<V, O>(JQ, K.R, K.p, Rn, pn);
#pragma unroll
          for (int k = 0; k < 9; ++k) K.R[k] = Rn[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) K.p[k] = pn[k];
        } else {
          T Rl[9], pl[3], Rn[9], t3[3];
          local_transform<T, O, RS>(qs, Rl, pl);
          matmul3(K.R, Rl, Rn);
          matvec3(K.R, pl, t3);
#pragma unroll
          for (int k = 0; k < 3; ++k) K.p[k] += t3[k];
#pragma unroll
          for (int k = 0; k < 9; ++k) K.R[k] = Rn[k];
        }
        if constexpr (jt == RBD_JOINT_QUAT_FLOATING) {
          T a0b[6];
#pragma unroll
          for (int k = 0; k < 6; ++k) a0b[k] = vs[(voff + k) * RS];  // solved for on the way up
          if constexpr (lvl > 0) {  // v̇ = S^-1 (a - a_parent); on the world it is already in its rows
            T d6[6], vd6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) d6[k] = a0b[k] - ad[k];
            xmotion_inv(K.R, K.p, d6, vd6);
#pragma unroll
            for (int k = 0; k < 6; ++k) ts[(voff + k) * RS] = vd6[k];
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) ad[k] = a0b[k];
        } else if constexpr (nvj_of(jt) == 3) {
          constexpr int xr = 10 * P::X3[O];
          T Wf[18], vd3[3];
          sfor<18>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = jc.value;
            if constexpr (j < 3) Wf[j] = vs[(voff + j) * RS];
            else if constexpr (j == 3) Wf[j] = sx_get(body);
            else if constexpr (j < 8) Wf[j] = ud1_get(body, j - 4);
            else Wf[j] = s3_get(xr + j - 8);
          });
#pragma unroll
          for (int k = 0; k < 3; ++k) vd3[k] = ts[(voff + k) * RS] - dot6(Wf + 6 * k, ad);  // v̇ = D^-1 u - (U D^-1)' a_delta,parent
          T v6[6], aj[6];
          body_twist<T, jt>(vd3, 1, v6);
          xmotion(K.R, K.p, v6, aj);
#pragma unroll
          for (int k = 0; k < 6; ++k) ad[k] += aj[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) ts[(voff + k) * RS] = vd3[k];
        } else if constexpr (jt != RBD_JOINT_FIXED) {
          V S[6];
          subspace_1dof<V, jt>(K.R, K.p, S);
          const V W[6] = {[&]() -> V { if constexpr (ABA_GST) return conv<V>(sx_get(body)); else return rd2<V>(xs, body, body2); }(), rd2<V>(vs, voff, voff2), ud_get(0), ud_get(1), ud_get(2), ud_get(3)};
          const V vd = rd2<V>(ts, voff, voff2) - dot6(W, ad);  // v̇ = D^-1 u - (U D^-1)' a_delta,parent
#pragma unroll
          for (int k = 0; k < 6; ++k) ad[k] += S[k] * vd;
          wr2<V>(ts, voff, voff2, vd);
        }
        if constexpr (nch >= 2) {
#pragma unroll
          for (int k = 0; k < 9; ++k) SR[bs][k] = K.R[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) SR[bs][9 + k] = K.p[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) SA[bs][k] = ad[k];
        }
      }
    }
  });
  wave_sync();
  if (mk_s >= 0) {  // the next v from the v̇ rows, through the v rows (free by now)
    const auto v0b = as_global((const T*)F.v0); const auto avb = as_global((T*)F.accv);
    T V0[NV > 0 ? NV : 1], AV[NV > 0 ? NV : 1];
    wave_sync();  // (every lane is done with the v rows of the top-down pass)
    if (mk_s > 0) {
#pragma unroll
      for (int k = 0; k < NV; ++k) V0[k] = v0b[(long)k * B + mk_si];
#pragma unroll
      for (int k = 0; k < NV; ++k) AV[k] = avb[(long)k * B + mk_si];
    } else {  // stage 0: the base point is still in the kernel's own v input (the passes have used its rows for other things)
      rows_in<T, NV>(v, Lv, state0, B, rv);  // (F.v_state is v)
      wave_sync();
#pragma unroll
      for (int k = 0; k < NV; ++k) V0[k] = vs[k * RS];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const T vd = ts[k * RS];
      const T sum = (mk_s > 0 ? AV[k] : T(0)) + mk_bs * vd;
      if (mk_s < 3 && mk_live) avb[(long)k * B + mk_gi] = sum;
      vs[k * RS] = V0[k] + mk_h * (mk_s < 3 ? mk_an * vd : sum);
    }
    wave_sync();
    rows_out<T, NV>(rv, const_cast<T*>(v), Lv, state0, B);
  }
  if (vdot) rows_out<T, NV>(rt, vdot, Lv, state0, B);
}

// inverse_dynamics! / dynamics_bias! (src/mechanism_algorithms.jl:542-553, :484-498; spatial_accelerations! :387-417, newton_euler! :428-439,
// joint_wrenches_and_torques! :442-459), one lane per state, compiled for rbd_plan's mechanism: the same walk as aba_spec with less to carry — the
// kinematic state holds the FULL spatial acceleration (a_parent + [T_parent, S q'] + S v̇, the world's is -g), a body's net wrench
// f = I a + T x* I T - w_ext + (its children's) goes up as six values, tau = S'f on the way; every joint is un-composed back to its parent.
// Nothing is kept per body, so the kernel fits the register file in fp64 as well (rows of q, v, v̇ in LDS: four wavefronts per CU in fp32, two in fp64).
// vdot == nullptr: dynamics_bias! (v̇ = 0).  tau rows: v̇ on the way in, tau on the way out.
constexpr int RNEA_ROWS = P::NQ + 2 * P::NV;
// DIRECT (fp64): only q goes through LDS — rows of q, v and v̇ in fp64 are 57 KB per wavefront, two wavefronts per CU and two rounds at 65 536 states
// (measured: 129 us, slower than the walk kernel).  v and v̇ of a joint are read from global memory one ENTER ahead of their use and wait on a stack
// along the path until the joint is un-composed; tau is stored by the lane.  19 KB of LDS per wavefront: four wavefronts per CU again.
template <typename T, bool DIRECT = false>
RBD_DEV void rnea_spec(long B, const T* __restrict__ q, const T* __restrict__ v, const T* __restrict__ vdot, const T* __restrict__ fext,
                       T* __restrict__ tau, Layout Lq, Layout Lv, Layout Lf, T* __restrict__ acc_out, T* __restrict__ jw_out, Layout Lo, T* lds) {
  // acc_out / jw_out (nullable; 6 n_bodies x B in layout Lo — the caller's, or batch-innermost scratch the host transposes afterwards: a lane
  // that stores its own state-major row writes 24-byte pieces 6 n_bodies elements apart, 77 us for the two outputs at 65 536 fp32 states
  // against 12 us batch-innermost): accelerations[body] and jointwrenches[body] of the reference's inverse_dynamics!
  // signature (spatial_accelerations! :387-417 — the world's -g included, as the other kernels export it; joint_wrenches_and_torques! :442-459), stored by the lane
  const bool out_vec = store6_vec(acc_out ? acc_out : jw_out, Lo, (int)sizeof(T)) && store6_vec(jw_out ? jw_out : acc_out, Lo, (int)sizeof(T));  // uniform
  constexpr int NQ = P::NQ, NV = P::NV, NBS = P::NBS > 0 ? P::NBS : 1, ML = P::NLEVELS;
  using T2 = typename PairOf<T, true>::type;  // limbs in lockstep (aba_spec above): the value type of an op that stands for two bodies (fp32, rows in LDS)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* rq = lds + (size_t)wave * (DIRECT ? NQ : RNEA_ROWS) * RS;
  T* rv = rq + NQ * RS;
  T* rt = rv + NV * RS;
  const long state0 = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 64;
  if (state0 >= B) return;
  rows_in<T, NQ>(q, Lq, state0, B, rq);
  if constexpr (!DIRECT) {
    rows_in<T, NV>(v, Lv, state0, B, rv);
    if (vdot) rows_in<T, NV>(vdot, Lv, state0, B, rt);
    else {
#pragma unroll 4
      for (int k = 0; k < NV; ++k) rt[k * RS + lane] = T(0);
    }
  }
  wave_sync();
  const T* qs = rq + lane;
  const T* vs = rv + lane;
  T* ts = rt + lane;
  const bool live = state0 + lane < B;
  const long sc = live ? state0 + lane : B - 1;
  const auto fel = as_global(fext ? fext + sc * Lf.sb : nullptr);  // (behind the conditional the pointer has lost its address space: FLAT loads, counted on the LDS counter too)
  const long fsk = Lf.sk;
  // DIRECT: this lane's columns of v, v̇, tau
  const T* vg = v + sc * Lv.sb;
  const auto ag = as_global(vdot ? vdot + sc * Lv.sb : nullptr);
  T* tg = tau + sc * Lv.sb;
  const long vsk = Lv.sk;
  T nv6[6], na6[6];          // DIRECT: velocity / acceleration coordinates of the next body to be entered (more than one: 3- / 6-dof joints)
  T PV[ML], PA[ML];          // DIRECT: those of the 1-dof joints on the path
  T MV[ML][6], MA[ML][6];    // DIRECT: those of the 3- / 6-dof joints on the path (only the levels that have one exist)
  auto prefetch = [&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value;
    if constexpr (DIRECT && O >= 0) {
      constexpr int jt = P::OPW[O][0] >> 16, voff = P::OPW[O][2], n = nvj_of(jt);
#pragma unroll
      for (int k = 0; k < n; ++k) { nv6[k] = vg[(long)(voff + k) * vsk]; na6[k] = ag ? ag[(long)(voff + k) * vsk] : T(0); }
    }
  };
  struct KinA { T R[9], p[3], Tw[6], a[6]; };    // the body the walk is at: transform to root, twist, spatial acceleration
  struct KinB { T2 R[9], p[3], Tw[6], a[6]; };
  KinA K1; KinB K2;
  T C1[6]; T2 C2[6];                           // net wrench of the child just finished, on its way to a chain parent
  T SF1[NBS][6]; T2 SF2[NBS][6];               // branch points: the sum of their children's
  T fe1[6]; T2 fe2[6];
  auto load_fe = [&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value;
    if constexpr (O >= 0) {
      if constexpr (P::PAIR[O] != 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) fe2[k] = fel ? mk2<T2>(fel[(long)(P::OPW[O][3] + k) * fsk], fel[(long)(P::OPW2[O][3] + k) * fsk]) : T2(0.0f);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) fe1[k] = fel ? fel[(long)(P::OPW[O][3] + k) * fsk] : T(0);
      }
    }
  };
  load_fe(Ix<P::FIRST_EXIT>{});
  prefetch(Ix<next_enter(-1)>{});
  sfor<P::NOPS>([&](auto oc) __attribute__((always_inline)) {
    constexpr int O = oc.value, w0 = P::OPW[O][0], kind = w0 & 0xff, lvl = (w0 >> 8) & 0xff, jt = w0 >> 16, voff = P::OPW[O][2], voff2 = P::OPW2[O][2];
    constexpr int nch = P::NCH[O], bs = P::BS[O], pbs = P::PBS[O], cidx = P::CIDX[O];
    constexpr bool PR = P::PAIR[O] != 0, ROOT = P::PROOT[O] != 0;
    using V = typename PairOf<T, PR>::type;
    auto& K = [&]() -> auto& { if constexpr (PR) return K2; else return K1; }();
    auto& C = [&]() -> auto& { if constexpr (PR) return C2; else return C1; }();
    auto& SF = [&]() -> auto& { if constexpr (PR) return SF2; else return SF1; }();
    auto& fe = [&]() -> auto& { if constexpr (PR) return fe2; else return fe1; }();
    // a per-body output of this op's body (and of its partner's)
    auto store_body = [&](T* out, const V* x) __attribute__((always_inline)) {
      if constexpr (PR) {
        T xa[6], xb[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { xa[k] = x[k].x; xb[k] = x[k].y; }
        store6(out, (long)P::OPW[O][3], Lo, sc, xa, out_vec);
        store6(out, (long)P::OPW2[O][3], Lo, sc, xb, out_vec);
      } else store6(out, (long)P::OPW[O][3], Lo, sc, x, out_vec);
    };
    // the joint's twist and acceleration in the root frame, from the body's transform (used entering the body and un-composing it)
    auto joint_motion = [&](const V* R, const V* p, V* S, V* vJ, V* aJ) __attribute__((always_inline)) {
      if constexpr (nvj_of(jt) > 1) {
        T v6[6], a6[6];
        if constexpr (DIRECT) { body_twist<T, jt>(MV[lvl], 1, v6); body_twist<T, jt>(MA[lvl], 1, a6); }
        else { body_twist<T, jt>(vs + voff * RS, RS, v6); body_twist<T, jt>(ts + voff * RS, RS, a6); }
        xmotion(R, p, v6, vJ);
        xmotion(R, p, a6, aJ);
      } else if constexpr (jt == RBD_JOINT_FIXED) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { vJ[k] = V(0); aJ[k] = V(0); }
      } else {
        subspace_1dof<V, jt>(R, p, S);
        V qd, vd;
        if constexpr (DIRECT) { qd = PV[lvl]; vd = PA[lvl]; }
        else { qd = rd2<V>(vs, voff, voff2); vd = rd2<V>(ts, voff, voff2); }
#pragma unroll
        for (int k = 0; k < 6; ++k) { vJ[k] = S[k] * qd; aJ[k] = S[k] * vd; }
      }
    };
    if constexpr (kind == SK_ENTER) {
      if constexpr (DIRECT) {  // the coordinates asked for at the ENTER before this one; then ask for the next body's
        if constexpr (nvj_of(jt) > 1) {
#pragma unroll
          for (int k = 0; k < nvj_of(jt); ++k) { MV[lvl][k] = nv6[k]; MA[lvl][k] = na6[k]; }
        } else if constexpr (jt != RBD_JOINT_FIXED) {
          PV[lvl] = nv6[0]; PA[lvl] = na6[0];
        }
        prefetch(Ix<next_enter(O)>{});
      }
      if constexpr (lvl == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) K.R[k] = (k % 4 == 0) ? V(1) : V(0);
#pragma unroll
        for (int k = 0; k < 3; ++k) K.p[k] = V(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) { K.Tw[k] = V(0); K.a[k] = V(0); }
        K.a[3] = V(-P::GRAVITY[0]); K.a[4] = V(-P::GRAVITY[1]); K.a[5] = V(-P::GRAVITY[2]);  // mechanism_algorithms.jl:396
      } else if constexpr (ROOT) {  // two limbs leave their common parent (K1 itself stays the parent's until they are done)
#pragma unroll
        for (int k = 0; k < 9; ++k) K2.R[k] = widen<T2>(K1.R[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) K2.p[k] = widen<T2>(K1.p[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) { K2.Tw[k] = widen<T2>(K1.Tw[k]); K2.a[k] = widen<T2>(K1.a[k]); }
      }
      V Rn[9], pn[3], S[6], vJ[6], aJ[6], cb[6];
      if constexpr (jt_1dof(jt) || jt == RBD_JOINT_FIXED) {
        compose_1dof<V, O>(joint_q<V, O>(qs), K.R, K.p, Rn, pn);
      } else {
        T Rl[9], pl[3], t3[3];
        local_transform<T, O, RS>(qs, Rl, pl);
        matmul3(K.R, Rl, Rn);
        matvec3(K.R, pl, t3);
#pragma unroll
        for (int k = 0; k < 3; ++k) pn[k] = K.p[k] + t3[k];
      }
      joint_motion(Rn, pn, S, vJ, aJ);
      if constexpr (jt != RBD_JOINT_FIXED) {
        se3_comm(K.Tw, vJ, cb);  // a_b = a_parent + [T_parent, vJ] + S v̇  (mechanism_algorithms.jl:414)
#pragma unroll
        for (int k = 0; k < 6; ++k) { K.a[k] += cb[k] + aJ[k]; K.Tw[k] += vJ[k]; }
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) K.R[k] = Rn[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) K.p[k] = pn[k];
      if (acc_out && live) store_body(acc_out, K.a);
    } else {
      RInertia<V> I;
      inertia_to_root_c<V, O>(K.R, K.p, I);
      V f[6], h[6];
      mul_inertia(I, K.a, f);
      momentum_cross(I, K.Tw, h);
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] += h[k] - fe[k];
      load_fe(Ix<P::NEXT_EXIT[O]>{});
      if constexpr (nch == 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) f[k] += C[k];
      } else if constexpr (nch >= 2) {
#pragma unroll
        for (int k = 0; k < 6; ++k) f[k] += SF[bs][k];
      }
      if (jw_out && live) store_body(jw_out, f);
      V S[6], vJ[6], aJ[6];
      joint_motion(K.R, K.p, S, vJ, aJ);  // (reads v̇ from the tau rows before tau overwrites it)
      if constexpr (nvj_of(jt) > 1) {
        T o6[6];
        xforce_inv(K.R, K.p, f, o6);
#pragma unroll
        for (int k = 0; k < nvj_of(jt); ++k) {
          if constexpr (DIRECT) { if (live) tg[(long)(voff + k) * vsk] = o6[comp_of(jt, k)]; }
          else ts[(voff + k) * RS] = o6[comp_of(jt, k)];
        }
      } else if constexpr (jt != RBD_JOINT_FIXED) {
        const V t = dot6(S, f);
        if constexpr (DIRECT) { if (live) tg[(long)voff * vsk] = t; }
        else wr2<V>(ts, voff, voff2, t);
      }
      if constexpr (lvl > 0) {
        if constexpr (ROOT) {  // the two limbs' wrenches reach their common parent as their sum; K1 is still that parent's state
          if constexpr (pbs >= 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { if constexpr (cidx == 0) SF1[pbs][k] = hsum(f[k]); else SF1[pbs][k] += hsum(f[k]); }
          } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) C1[k] = hsum(f[k]);
          }
        } else {
          if constexpr (pbs >= 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { if constexpr (cidx == 0) SF[pbs][k] = f[k]; else SF[pbs][k] += f[k]; }
          } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) C[k] = f[k];
          }
          if constexpr (jt != RBD_JOINT_FIXED) {
            V cb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) K.Tw[k] -= vJ[k];  // T_parent
            se3_comm(K.Tw, vJ, cb);
#pragma unroll
            for (int k = 0; k < 6; ++k) K.a[k] -= cb[k] + aJ[k];
          }
          if constexpr (jt_1dof(jt) || jt == RBD_JOINT_FIXED) {
            uncompose_1dof<V, O>(joint_q<V, O>(qs), K.R, K.p);
          } else {
            T Rl[9], pl[3], Rp[9], t3[3];
            local_transform<T, O, RS>(qs, Rl, pl);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
              for (int j = 0; j < 3; ++j) Rp[3 * i + j] = K.R[3 * i] * Rl[3 * j] + K.R[3 * i + 1] * Rl[3 * j + 1] + K.R[3 * i + 2] * Rl[3 * j + 2];  // R Rl'
            matvec3(Rp, pl, t3);
#pragma unroll
            for (int k = 0; k < 9; ++k) K.R[k] = Rp[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) K.p[k] -= t3[k];
          }
        }
      }
    }
  });
  if constexpr (!DIRECT) {
    wave_sync();
    rows_out<T, NV>(rt, tau, Lv, state0, B);
  }
}
#endif  // RBD_SPEC_ABA

#ifndef RBD_SPEC_CHOL_WAVES
#define RBD_SPEC_CHOL_WAVES 2  // wavefronts per SIMD the chol_spec kernels are compiled for (their tiles: 230 registers)
#endif
#if defined(RBD_SPEC_CHOL) && !defined(RBD_SPEC_EMU)  // (the host emulation of tests/emu has no matrix cores)
// ---------------------------------------------------------------------------------------------------------------------------------
// The dense step of `dynamics_solve!` (potrf! / potrs!, src/mechanism_algorithms.jl:764, :819) specialised on the SPARSITY of the mechanism's
// mass matrix.  M[i][j] is non-zero only when one of the two coordinates is an ancestor of the other (mass_matrix!'s support sets,
// src/mechanism_state.jl:95-98); ordered children-before-parents (PERM: reverse depth-first pre-order) the Cholesky factor has exactly the same
// pattern — no fill-in (Featherstone's branch-induced sparsity) — so every 4 x 4 tile of the permuted lower triangle that is structurally zero
// (TMASK) stays zero and is skipped at compile time: in the loads, the panel solves, the matrix-core updates and both substitutions.  Atlas:
// 45 lower tiles -> TMASK keeps about half.  x = M^-1 (tau - c) is the same vector whatever the elimination order; it is stored un-permuted.
// The tile algorithm itself is chol_mfma_kernel's (rbd_kernels.hip): 16 states per wavefront, a quad of lanes per state, lane r of the quad
// keeps row r of every tile, rank-4 updates as v_mfma_f32_4x4x1 — one instruction for the tile of all 16 states.
// Mg: the staging buffer crba_spec<PERMUTED> fills (permuted lower triangle, grouped by 16 states, structural zeros never written: the host
// zeroes the buffer once).
// ---------------------------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K> RBD_DEV float qbcast(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), K | (K << 2) | (K << 4) | (K << 6), 0xf, 0xf, true));
}
RBD_DEV float qsum(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
  return x;
}
RBD_DEV int sel4(int r, int a, int b, int c, int d) { return r == 0 ? a : r == 1 ? b : r == 2 ? c : d; }


// entry (r, CC) of a symmetric 4 x 4 tile whose quad of lanes holds the lower half (lane r: row r, the columns above the diagonal zero): the lane's own
// value on and below the diagonal, lane CC's column r above it
template <int CC> RBD_DEV float diag_entry(const f32x4& tt, int r) {
  if constexpr (CC == 0) return tt[0];
  else {
    // (every DPP read on its own line, outside the selects: inside a conditional expression it would run under the condition's exec mask and read a
    //  lane that is switched off — zero)
    const float b0 = qbcast<CC>(tt[0]);
    const float b1 = CC >= 2 ? qbcast<CC>(tt[1]) : 0.0f;
    const float b2 = CC >= 3 ? qbcast<CC>(tt[2]) : 0.0f;
    const float m = r == 0 ? b0 : r == 1 ? b1 : b2;
    return r >= CC ? tt[CC] : m;
  }
}
constexpr int cmin4(int a, int b, int c, int d) { return (a < b ? a : b) < (c < d ? c : d) ? (a < b ? a : b) : (c < d ? c : d); }
constexpr int cmax4(int a, int b, int c, int d) { return (a > b ? a : b) > (c > d ? c : d) ? (a > b ? a : b) : (c > d ? c : d); }
// LDS floats of chol_spec's emission: the tiles of 16 states and, behind them, a row per state for the writes that do not belong to the block in hand
constexpr int chol_emit_lds() { return 16 * (4 * P::NV + 4) + 16 * P::NV; }

// EMIT is a template parameter, not a test of Mc: with both paths in one kernel they meet again in front of the factorisation, where the compiler must then
// wait for the tile loads of the path WITHOUT emission — s_waitcnt vmcnt(0), which on the path with emission also waits for every store just issued (vmcnt
// counts loads and stores alike): the wavefront sat out the drain of its 83 KB before it factored, 93 us for the launch.  As two kernels the factorisation
// runs while the stores drain.
// Mc + mst (LDS, chol_emit_lds() floats): the caller's M — the WHOLE square per state in the ORIGINAL coordinate order, as emit_spec below
// writes it — sent on its way from the tiles this wavefront has just loaded for the factorisation, before it factors them.  Round 3 ran emit_spec in front of
// this function: a second gather of the staged triangle, whose loads sat behind the block's stores in the wavefront's one in-order memory counter (vmcnt counts
// loads AND stores on gfx9: every block of four columns waited for the previous block's stores to be acknowledged — 9 store round trips per wavefront,
// 72 us of the launch's 106).  Here nothing is loaded after the first store: tile -> LDS -> whole 16-byte pieces of complete runs, nine blocks back to back,
// and the stores drain while the wavefront factors.
// (Every other workgroup taking the two steps in the opposite order — factor first, the tiles read a second time, then out — so that half the chip emits while
//  half factors was built and measured: 95 us against 93; every other wavefront of the first round starting 7 / 14 / 20 us late: +2.5 / +8 / +13 us.  The launch
//  is not waiting on wavefronts that march in step: a wavefront's own chain — load, through LDS and out, factor, solve: 31 us of which it issues for 7 — is the
//  time, two rounds of them with two per SIMD; the square with its stores sent to one place instead of 357 MB takes 62 us against 86, the packed triangle with
//  half the bytes 84.)
template <bool PACKED, bool EMIT>
RBD_DEV void chol_spec(long B, long group, const float* __restrict__ Mg, const float* __restrict__ tau, const float* __restrict__ c,
                       float* __restrict__ x, Layout Lv, int* __restrict__ notpd, float* __restrict__ Mc, Layout Lc, float* mst) {
  constexpr int NT = P::NT, NV = P::NV;
  const int lane = threadIdx.x & 63, r = lane & 3;
  const long state_raw = group * 16 + (lane >> 2);
  const bool live = state_raw < B;
  const long state = live ? state_raw : B - 1;
  const float* Mb = Mg + (state >> 4) * (16L * NV * NV) + (state & 15) + r * 16;  // row 4I + r of column col: Mb[(col * NV + 4I) * 16]
  f32x4 t[NT][NT];  // tiles of the mask only (the others are never touched: the optimiser drops them)
  sfor<NT>([&](auto Ic) __attribute__((always_inline)) {
    constexpr int I = Ic.value;
    sfor<I + 1>([&](auto Jc) __attribute__((always_inline)) {
      constexpr int J = Jc.value;
      if constexpr (P::TMASK[I][J] != 0) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          float a = Mb[((4 * J + cc) * NV + 4 * I) * 16];
          if (I == J && cc > r) a = 0.0f;  // above the diagonal: not part of the staged triangle
          t[I][J][cc] = a;
        }
      }
    });
  });
  float y[NT];
  int od[NT];  // the velocity coordinate this lane's row of block I stands for
  sfor<NT>([&](auto Ic) __attribute__((always_inline)) {
    constexpr int I = Ic.value;
    od[I] = sel4(r, P::INV[4 * I], P::INV[4 * I + 1], P::INV[4 * I + 2], P::INV[4 * I + 3]);
    y[I] = 0.0f;
  });
  // the right-hand side: ALL of a wavefront's loads of tau behind one (uniform) test, then all of c's — with the two tests inside the loop over the blocks
  // every block's pair of loads sat in a branch of its own with a wait behind it: nine dependent round trips in front of the factorisation
  if (tau) {
    sfor<NT>([&](auto Ic) __attribute__((always_inline)) { y[Ic.value] = tau[(long)od[Ic.value] * Lv.sk + state * Lv.sb]; });
  }
  if (c) {
    float yc[NT];
    sfor<NT>([&](auto Ic) __attribute__((always_inline)) { yc[Ic.value] = c[(long)od[Ic.value] * Lv.sk + state * Lv.sb]; });
    sfor<NT>([&](auto Ic) __attribute__((always_inline)) { y[Ic.value] -= yc[Ic.value]; });
  }
#ifdef RBD_SPEC_EMIT
  if constexpr (EMIT) {
    // every load of this wavefront — tiles, right-hand side — has arrived before its first store is issued: vmcnt counts loads and stores in order, so a value
    // first used behind a store would be waited for together with that store's acknowledgement (the compiler waits for each tile where it is first used: the
    // scatter of block 1 waited for the stores of block 0, and so on down the nine blocks)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  }
  if constexpr (PACKED) {
    // The caller's M as LAPACK's packed lower triangle (element (i, j), i >= j, at i + j (2 nv - j - 1) / 2: nv (nv + 1) / 2 values per state, columns back to
    // back) — the part of M the reference defines (Symmetric, uplo 'L': src/dynamics_result.jl:42) and half the bytes of the square.  Same scheme as the square
    // below: per block of four columns the tile goes through LDS — an entry (a, b) of the original matrix belongs to column min(a, b) — and leaves in 16-byte
    // pieces.  The pieces matter (4-byte pieces: 115 us for the launch, 8-byte: 99, 16-byte: 94 — with the round-4 address arithmetic; 84 now), and a block's
    // run starts wherever the columns before it end.  So a state's triangle is treated as one stream: a block's values are
    // laid behind the 0..3 values the block before could not complete a piece with (c), whole pieces leave, the rest is carried; a state whose run starts in the
    // middle of a piece (nv (nv + 1) / 2 = 2 mod 4: every other state) sends its first two values, the one before it its last two, as 8-byte pieces.
    if constexpr (EMIT) {
      constexpr int MST = 4 * NV + 4, NP = NV * (NV + 1) / 2, A1 = NP % 4;  // A1: where an odd state's run starts within a 16-byte piece (0 or 2)
      static_assert(A1 == 0 || A1 == 2, "nv a multiple of 4");
      float* const mine = mst + (lane >> 2) * MST;
      const int odd = (lane >> 2) & 1;
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
      const long left = B - group * 16;
      const unsigned nlive = left < 16 ? (unsigned)left : 16u;  // states of this group inside the batch
      float* const Mw = Mc + group * 16 * (long)NP;
      sfor<NT>([&](auto Joc) __attribute__((always_inline)) {
        constexpr int Jo = Joc.value, j0 = 4 * Jo;
        constexpr int LB = 4 * NV - 16 * Jo - 6;              // values of the block per state: columns j0 .. j0 + 3 from their diagonals down
        constexpr int P0 = j0 * NV - j0 * (j0 - 1) / 2;        // where column j0 starts in the packed triangle
        constexpr int C0 = P0 % 4, C1 = (A1 + P0) % 4;         // values carried in from the block before (even / odd state)
        constexpr int N0 = (C0 + LB) / 4, N1 = (C1 + LB) / 4;  // whole pieces now
        constexpr int NMAX = N0 > N1 ? N0 : N1;
        constexpr int LBP = 4 * NV - 16 * (Jo - 1) - 6, P0P = P0 - LBP;                       // the block before
        constexpr int NP0 = Jo > 0 ? ((P0P % 4) + LBP) / 4 : 0, NP1 = Jo > 0 ? (((A1 + P0P) % 4) + LBP) / 4 : 0;
        const int cm = odd ? C1 : C0;
        wave_sync();  // the tile of the block before has been read out
        float carry = 0.0f;
        if constexpr (Jo > 0) carry = mine[4 * (odd ? NP1 : NP0) + r];  // (every lane reads; what it keeps is decided where it is written)
        wave_sync();
        for (int i = lane * 4; i < 16 * MST; i += 256) *reinterpret_cast<f32x4*>(mst + i) = zero;
        wave_sync();
        if constexpr (Jo > 0) (r < cm ? mine : mst + 16 * MST + (lane >> 2) * NV)[r] = carry;
        // an entry (a, co) of the original matrix goes to column co when it lies on or below the diagonal (a >= co), to column a when above — and only when
        // that column is one of this block's four.  Which of the two is a test per lane (a = od[I] is the lane's row): the address is a select between the
        // lane's base and the state's dump word, the rest of it a constant in the instruction's offset field (the dump's base is taken back by that constant).
        // A diagonal tile is made symmetric within its quad first (diag_entry): column images only.
        float* const cbase = mine + cm;
        float* const dump0 = mst + 16 * MST + (lane >> 2) * NV;  // a row of nv words per state behind the tiles for what does not belong to the block
        sfor<NT>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int I = Ic.value;
          constexpr bool rows_in = (P::INV[4 * I] >> 2) == Jo || (P::INV[4 * I + 1] >> 2) == Jo || (P::INV[4 * I + 2] >> 2) == Jo || (P::INV[4 * I + 3] >> 2) == Jo;
          const int a = od[I];                      // this lane's row coordinate in tile row I
          const int ca = a - j0;                    // ... as a column of this block (when 0 <= ca < 4)
          const bool mir = (unsigned)ca < 4u;
          float* const colp = cbase + ca;           // column co, row a: + (start of column co in the block) - (co - j0)
          float* const mirp = cbase + (ca * (NV - j0) - (ca * (ca - 1)) / 2 - ca);  // column a, row co: + (co - j0)
          float* const mirq = mir ? mirp : dump0;  // (offsets below nv: inside the dump row)
          constexpr int AMIN = cmin4(P::INV[4 * I], P::INV[4 * I + 1], P::INV[4 * I + 2], P::INV[4 * I + 3]);
          constexpr int AMAX = cmax4(P::INV[4 * I], P::INV[4 * I + 1], P::INV[4 * I + 2], P::INV[4 * I + 3]);
          sfor<I + 1>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int J = Jc.value;
            if constexpr (P::TMASK[I][J] != 0) {
              sfor<4>([&](auto ccc) __attribute__((always_inline)) {
                constexpr int cc = ccc.value, co = P::INV[4 * J + cc];  // the entry (a, co) of the original matrix
                // where the four rows of the tile row all lie on one side of co the test is settled here, not per lane (the elimination order keeps most of
                // the coordinates' order: most tiles)
                constexpr bool all_ge = AMIN >= co, all_lt = AMAX < co;
                constexpr bool col_here = (co >> 2) == Jo && !all_lt, mir_here = I != J && rows_in && co > j0 && !all_ge;
                if constexpr (col_here || mir_here) {
                  float val;
                  if constexpr (I == J) val = diag_entry<cc>(t[I][I], r); else val = t[I][J][cc];
                  const bool ge = a >= co;
                  if constexpr (col_here) {
                    constexpr int cb = co - j0, imm = cb * (NV - j0) - (cb * (cb - 1)) / 2 - cb;
                    if constexpr (all_ge) colp[imm] = val;
                    else (ge ? colp : dump0 - imm)[imm] = val;
                  }
                  if constexpr (mir_here) {
                    constexpr int imm2 = co - j0;
                    if constexpr (all_lt) mirq[imm2] = val;
                    else ((mir && !ge) ? mirp : dump0 - imm2)[imm2] = val;
                  }
                }
              });
            }
          });
        });
        wave_sync();
#pragma unroll
        for (int c0 = 0; c0 < 16 * NMAX; c0 += 64) {
          // The way out: the block's pieces dealt out along the lanes, 64 consecutive ones per store — whole runs of a state, 1 KB per instruction.  What the
          // memory is given per request decides here, not what the wavefront issues: a quad of lanes per state (each store 16 separate 64-byte segments that
          // start anywhere in their cache lines) needs a fifth of the address arithmetic and took the launch from 84 to 133 us; the lanes without a piece
          // sending their neighbour's again instead of being switched off: 89.  So: the LDS read for every lane, the store under its predicate; byte offsets in
          // 32 bits from the wavefront's own base.
          const unsigned ch0 = (unsigned)(c0 + lane), ch = ch0 < 16u * NMAX ? ch0 : 16u * NMAX - 1u;
          const unsigned st = ch / (unsigned)NMAX, pc = ch - st * (unsigned)NMAX, so = st & 1u;
          const unsigned first = (Jo == 0 && so && A1 != 0) ? 1u : 0u;  // (an odd state's first piece starts with the state before's last two values: sent below)
          const f32x4 piece = *reinterpret_cast<const f32x4*>(mst + st * (unsigned)MST + 4u * pc);
          const unsigned gb = 4u * (st * (unsigned)NP + (unsigned)P0 - (so ? (unsigned)C1 : (unsigned)C0) + 4u * pc);
          if (ch0 < 16u * NMAX && pc >= first && pc < (so ? (unsigned)N1 : (unsigned)N0) && st < nlive)
            __builtin_nontemporal_store(piece, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(Mw) + gb));
        }
        const long gm = group * 16 + (lane >> 2);
        if constexpr (Jo == 0 && A1 != 0) {  // the first two values of an odd state
          if (odd && r == 0 && gm < B) __builtin_nontemporal_store(*reinterpret_cast<const f2*>(mine + 2), reinterpret_cast<f2*>(Mc + gm * (long)NP));
        }
        if constexpr (Jo == NT - 1) {        // what is left behind the last whole piece: two values, or none
          constexpr int R0 = (C0 + LB) % 4, R1 = (C1 + LB) % 4;
          static_assert((R0 == 0 || R0 == 2) && (R1 == 0 || R1 == 2), "a state's run ends on an 8-byte boundary");
          const int rm = odd ? R1 : R0, nm = odd ? N1 : N0;
          if (rm == 2 && r == 0 && gm < B)
            __builtin_nontemporal_store(*reinterpret_cast<const f2*>(mine + 4 * nm), reinterpret_cast<f2*>(Mc + gm * (long)NP + (P0 - cm) + 4 * nm));
        }
      });
      static_assert(NP == 4 * NV * NT - 16 * (NT * (NT - 1) / 2) - 6 * NT, "blocks cover the triangle");
    }
  } else {
    if constexpr (EMIT) {
      // Per block of four columns (16 nv contiguous bytes per state): the tile of the 16 states is cleared in LDS (the structural zeros), the entries of the
      // block are written into it — an entry (a, b) below the diagonal lands in column b, and as its mirror image in column a — and the tile leaves as whole
      // 16-byte pieces of complete runs.  What this costs is instructions, not bytes (RBD_TUNE spec_variant=1024, every wavefront's M onto the same 64 states:
      // 81 us against 97 — the memory behind the stores is worth 15 us of the 62 the emission took), so every address is a per-lane base plus a constant:
      //  * the column image of row a = od[I]: base mine + a, the column and the block in the instruction's offset field — no arithmetic per entry;
      //  * the mirror image: base = the lane's column of this block, or a dump row behind the tiles when its row belongs to another block — one select per
      //    (tile row, block), not one per entry;
      //  * a diagonal tile is made symmetric within its quad of lanes first (its upper half comes from the lanes below by DPP): column images only;
      //  * the way out: which piece of which state a lane sends in each of its stores is the same for every block — worked out once (a division by nv), the
      //    block is a constant added to the wavefront's base; states past the end of the batch send their last live neighbour's pieces again (no branch).
      using V = f32x4;
      constexpr int MST = 4 * NV + 4, CB = 4 * NV, PCS = CB / 4, NK = (16 * PCS + 63) / 64;  // values per state of the LDS tile; values / 16-byte pieces per state and block
      float* const mine = mst + (lane >> 2) * MST;
      float* const dump = mst + 16 * MST + (lane >> 2) * NV;
      const V zero = {0.0f, 0.0f, 0.0f, 0.0f};
      const long left = B - group * 16;
      const int nlive = left < 16 ? (int)left : 16;
      unsigned lo[NK], go[NK];
#ifdef RBD_SPEC_EMIT_SKIP_UPPER
      // The reference defines the LOWER triangle of M (Symmetric, uplo 'L'); the mirror image above it is written only because whole cache lines leave faster.
      // A 64-byte segment of a state's square that lies strictly above the diagonal altogether (Atlas: 15 of a state's 81) need not leave at all: the pieces of
      // such segments are switched off, every segment that does leave is still complete.  keep[k]: bit Jo = the piece this lane sends in store k of block Jo
      // belongs to a segment with an entry on or below the diagonal.
      struct KeepTab { unsigned m[PCS]; };
      static constexpr KeepTab KT = [] {
        KeepTab t{};
        for (int p = 0; p < PCS; ++p) {
          unsigned m = 0;
          for (int Jo = 0; Jo < NT; ++Jo) {
            const int g0 = (Jo * CB + 4 * p) & ~15;
            bool any = false;
            for (int e = g0; e < g0 + 16 && e < NV * NV; ++e) any = any || (e % NV >= e / NV);
            if (any) m |= 1u << Jo;
          }
          t.m[p] = m;
        }
        return t;
      }();
      unsigned keep[NK];
#endif
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        int ch = 64 * k + lane;
        ch = ch < 16 * PCS ? ch : 16 * PCS - 1;
        int st = ch / PCS;
        const int piece = ch - st * PCS;
        st = st < nlive ? st : nlive - 1;
        lo[k] = (unsigned)(st * MST + piece * 4);
        go[k] = 4u * ((unsigned)st * (unsigned)Lc.sb + (unsigned)(piece * 4));  // in bytes: the store then takes it as its 32-bit offset beside a scalar base
#ifdef RBD_SPEC_EMIT_SKIP_UPPER
        keep[k] = KT.m[piece];
#endif
      }
#ifdef RBD_SPEC_ABLATE_EMIT_LOCAL  // (timing experiments, spec_variant 1024: every wavefront's M lands on the same 64 states — the stores without the memory behind them)
      float* const Mw = Mc + ((group * 16) & 63) * Lc.sb;
#else
      float* const Mw = Mc + group * 16 * Lc.sb;
#endif
      sfor<NT>([&](auto Joc) __attribute__((always_inline)) {
        constexpr int Jo = Joc.value;
        wave_sync();  // the tile of the block before has been read out
        for (int i = lane * 4; i < 16 * MST; i += 256) *reinterpret_cast<V*>(mst + i) = zero;
        wave_sync();
        sfor<NT>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int I = Ic.value;
          // does some row of tile row I stand for a coordinate of block Jo (its entries then also land there as the mirror image)?
          constexpr bool rows_in = (P::INV[4 * I] >> 2) == Jo || (P::INV[4 * I + 1] >> 2) == Jo || (P::INV[4 * I + 2] >> 2) == Jo || (P::INV[4 * I + 3] >> 2) == Jo;
          float* const colp = mine + od[I];
          float* mp = dump;
          if constexpr (rows_in) mp = ((od[I] >> 2) == Jo) ? mine + (od[I] & 3) * NV : dump;
          sfor<I + 1>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int J = Jc.value;
            if constexpr (P::TMASK[I][J] != 0) {
              sfor<4>([&](auto ccc) __attribute__((always_inline)) {
                constexpr int cc = ccc.value, co = P::INV[4 * J + cc];  // the entry (od[I], co) of the original matrix
                if constexpr ((co >> 2) == Jo) {
                  if constexpr (I == J) colp[(co - 4 * Jo) * NV] = diag_entry<cc>(t[I][I], r);
                  else colp[(co - 4 * Jo) * NV] = t[I][J][cc];
                }
                if constexpr (I != J && rows_in) mp[co] = t[I][J][cc];
              });
            }
          });
        });
        wave_sync();
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#ifdef RBD_SPEC_EMIT_SKIP_UPPER
          const V piece = *reinterpret_cast<const V*>(mst + lo[k]);  // (the LDS read for every lane, the store under its predicate)
          if (keep[k] & (1u << Jo)) __builtin_nontemporal_store(piece, reinterpret_cast<V*>(reinterpret_cast<char*>(Mw + Jo * CB) + go[k]));
#else
          __builtin_nontemporal_store(*reinterpret_cast<const V*>(mst + lo[k]), reinterpret_cast<V*>(reinterpret_cast<char*>(Mw + Jo * CB) + go[k]));
#endif
        }
      });
    }
  }
#endif
  bool bad = false;
  sfor<NT>([&](auto Jc) __attribute__((always_inline)) {
    constexpr int J = Jc.value;
    float dinv[4];
#define RBD_DIAG_STEP(K)                                                              \
    {                                                                                 \
      const float d = qbcast<K>(t[J][J][K]);                                          \
      bad |= !(d > 0.0f);                                                             \
      const float rs = rsqrt_nr(d); /* v_rsq + one Newton step; a pivot that is not positive is reported */ \
      dinv[K] = rs;                                                                   \
      const float lk = (r == K) ? d * rs : t[J][J][K] * rs;                           \
      t[J][J][K] = lk;                                                                \
      if (K < 1) t[J][J][1] -= lk * qbcast<1>(lk);                                    \
      if (K < 2) t[J][J][2] -= lk * qbcast<2>(lk);                                    \
      if (K < 3) t[J][J][3] -= lk * qbcast<3>(lk);                                    \
    }
    RBD_DIAG_STEP(0) RBD_DIAG_STEP(1) RBD_DIAG_STEP(2) RBD_DIAG_STEP(3)
#undef RBD_DIAG_STEP
    const float l10 = qbcast<1>(t[J][J][0]), l20 = qbcast<2>(t[J][J][0]), l30 = qbcast<3>(t[J][J][0]);
    const float l21 = qbcast<2>(t[J][J][1]), l31 = qbcast<3>(t[J][J][1]), l32 = qbcast<3>(t[J][J][2]);
    // panel: X = T L_d^-T, one row per lane
    sfor<NT - J - 1>([&](auto Ic) __attribute__((always_inline)) {
      constexpr int I = J + 1 + Ic.value;
      if constexpr (P::TMASK[I][J] != 0) {
        const float x0 = t[I][J][0] * dinv[0];
        const float x1 = (t[I][J][1] - x0 * l10) * dinv[1];
        const float x2 = (t[I][J][2] - x0 * l20 - x1 * l21) * dinv[2];
        const float x3 = (t[I][J][3] - x0 * l30 - x1 * l31 - x2 * l32) * dinv[3];
        t[I][J][0] = x0; t[I][J][1] = x1; t[I][J][2] = x2; t[I][J][3] = x3;
      }
    });
    // trailing update on the matrix cores: T(I,J') -= X_I X_J'^T for J < J' <= I, where both factors (and then the target) are in the mask
    sfor<NT - J - 1>([&](auto Jpc) __attribute__((always_inline)) {
      constexpr int Jp = J + 1 + Jpc.value;
      if constexpr (P::TMASK[Jp][J] != 0) {
        const float n0 = -t[Jp][J][0], n1 = -t[Jp][J][1], n2 = -t[Jp][J][2], n3 = -t[Jp][J][3];
        sfor<NT - Jp>([&](auto Ic) __attribute__((always_inline)) {
          constexpr int I = Jp + Ic.value;
          if constexpr (P::TMASK[I][J] != 0 && P::TMASK[I][Jp] != 0) {
            f32x4 acc = t[I][Jp];
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n0, t[I][J][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n1, t[I][J][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n2, t[I][J][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(n3, t[I][J][3], acc, 0, 0, 0);
            t[I][Jp] = acc;
          }
        });
      }
    });
    // forward substitution for this block column
    {
      float v0 = qbcast<0>(y[J]) * dinv[0];
      y[J] = (r == 0) ? v0 : y[J] - t[J][J][0] * v0;
      float v1 = qbcast<1>(y[J]) * dinv[1];
      y[J] = (r == 1) ? v1 : ((r > 1) ? y[J] - t[J][J][1] * v1 : y[J]);
      float v2 = qbcast<2>(y[J]) * dinv[2];
      y[J] = (r == 2) ? v2 : ((r > 2) ? y[J] - t[J][J][2] * v2 : y[J]);
      float v3 = qbcast<3>(y[J]) * dinv[3];
      y[J] = (r == 3) ? v3 : y[J];
      sfor<NT - J - 1>([&](auto Ic) __attribute__((always_inline)) {
        constexpr int I = J + 1 + Ic.value;
        if constexpr (P::TMASK[I][J] != 0) y[I] -= t[I][J][0] * v0 + t[I][J][1] * v1 + t[I][J][2] * v2 + t[I][J][3] * v3;
      });
    }
  });
  if (live && bad && r == 0) atomicOr(notpd, 1);
  // backward substitution L' x = y, block columns in reverse
  sfor<NT>([&](auto Jr) __attribute__((always_inline)) {
    constexpr int J = NT - 1 - Jr.value;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    bool any = false;
    sfor<NT - J - 1>([&](auto Ic) __attribute__((always_inline)) {
      constexpr int I = J + 1 + Ic.value;
      if constexpr (P::TMASK[I][J] != 0) { a0 += t[I][J][0] * y[I]; a1 += t[I][J][1] * y[I]; a2 += t[I][J][2] * y[I]; a3 += t[I][J][3] * y[I]; any = true; }
    });
    if (any) { a0 = qsum(a0); a1 = qsum(a1); a2 = qsum(a2); a3 = qsum(a3); }
    float yj = y[J] - ((r == 0) ? a0 : (r == 1) ? a1 : (r == 2) ? a2 : a3);
    const float d0 = qbcast<0>(t[J][J][0]), d1 = qbcast<1>(t[J][J][1]), d2 = qbcast<2>(t[J][J][2]), d3 = qbcast<3>(t[J][J][3]);
    const float b30 = qbcast<3>(t[J][J][0]), b31 = qbcast<3>(t[J][J][1]), b32 = qbcast<3>(t[J][J][2]);
    const float b20 = qbcast<2>(t[J][J][0]), b21 = qbcast<2>(t[J][J][1]);
    const float x3 = qbcast<3>(yj) * rcp_nr(d3);
    const float l3r = (r == 0) ? b30 : (r == 1) ? b31 : b32;
    yj = (r == 3) ? x3 : yj - l3r * x3;
    const float x2 = qbcast<2>(yj) * rcp_nr(d2);
    const float l2r = (r == 0) ? b20 : b21;
    yj = (r == 2) ? x2 : ((r < 2) ? yj - l2r * x2 : yj);
    const float x1 = qbcast<1>(yj) * rcp_nr(d1);
    const float l1r = qbcast<1>(t[J][J][0]);
    yj = (r == 1) ? x1 : ((r < 1) ? yj - l1r * x1 : yj);
    const float x0 = qbcast<0>(yj) * rcp_nr(d0);
    yj = (r == 0) ? x0 : yj;
    y[J] = yj;
  });
  if (live) {
    sfor<NT>([&](auto Ic) __attribute__((always_inline)) { x[(long)od[Ic.value] * Lv.sk + state * Lv.sb] = y[Ic.value]; });
  }
}

#endif  // RBD_SPEC_CHOL

#if defined(RBD_SPEC_EMIT) && !defined(RBD_SPEC_EMU)
// The caller's M from the staging buffer: the WHOLE nv x nv square per state, column-major, in the ORIGINAL coordinate order (the reference leaves
// the strict upper triangle of its Symmetric(:L) undefined; here it holds the mirror image — chol_mfma_kernel's choice, kept: complete cache lines
// written with nontemporal 16-byte stores are more than twice as fast as the lower triangle's partial lines).  16 states per wavefront; per block
// of four columns (16 nv contiguous bytes per state) the non-zeros are gathered from the staging buffer into an LDS tile (EMIT: entry -> slot,
// four entries x 16 states per load instruction), the zeros come from clearing the tile, and the tile leaves as whole runs.
template <typename T> struct EmitVec;  // a 16-byte piece
template <> struct EmitVec<float> { typedef float type __attribute__((ext_vector_type(4))); enum { N = 4 }; };
template <> struct EmitVec<double> { typedef double type __attribute__((ext_vector_type(2))); enum { N = 2 }; };
template <typename T> constexpr int emit_mst() { return 4 * P::NV + EmitVec<T>::N; }  // values per state of the LDS tile (a spare piece for the list's padding)
template <typename T>
RBD_DEV void emit_spec(long B, long group, const T* __restrict__ Mg, T* __restrict__ Mc, Layout Lc, T* mst) {
  using V = typename EmitVec<T>::type;
  constexpr int NT = P::NT, NV = P::NV, MST = emit_mst<T>(), CB = 4 * NV, VN = EmitVec<T>::N, PCS = CB / VN;  // PCS: 16-byte pieces per state and block
  const int lane = threadIdx.x & 63, qd = lane >> 4, s = lane & 15;
  const long gs = group * 16 + s;
  const long gsl = gs < B ? gs : B - 1;
  const T* src = Mg + (gsl >> 4) * (16L * NV * NV) + (gsl & 15);
  T* mine = mst + s * MST;
  const unsigned* tab = &P::EMIT[0][0] + qd;
  // the gathers of block Jo + 1 are in flight while block Jo goes through the tile and out
  T v[2][P::EMIT_KMAX];
  int dst[2][P::EMIT_KMAX];
  auto gather = [&](auto Jc) __attribute__((always_inline)) {
    constexpr int Jo = Jc.value;
#pragma unroll
    for (int k = 0; k < P::EMIT_K[Jo]; ++k) {
      const unsigned e = tab[(Jo * P::EMIT_KMAX + k) * 4];  // staged entry << 16 | slot in the tile
      v[Jo & 1][k] = src[(long)(e >> 16) * 16];
      dst[Jo & 1][k] = (int)(e & 0xffffu);
    }
  };
  gather(Ix<0>{});
  V zero;
#pragma unroll
  for (int k = 0; k < VN; ++k) zero[k] = T(0);
  sfor<NT>([&](auto Jc) __attribute__((always_inline)) {
    constexpr int Jo = Jc.value, K = P::EMIT_K[Jo];
    if constexpr (Jo + 1 < NT) gather(Ix<Jo + 1>{});
    wave_sync();  // the tile of the block before has been read out
    for (int i = lane * VN; i < 16 * MST; i += 64 * VN) *reinterpret_cast<V*>(mst + i) = zero;
    wave_sync();
#pragma unroll
    for (int k = 0; k < K; ++k) mine[dst[Jo & 1][k]] = v[Jo & 1][k];
    wave_sync();
#pragma unroll 3
    for (int c0 = 0; c0 < 16 * PCS; c0 += 64) {
      const int ch = c0 + lane, st = ch / PCS, piece = ch - st * PCS;
      const long g2 = group * 16 + st;
      if (ch < 16 * PCS && g2 < B)
        __builtin_nontemporal_store(*reinterpret_cast<const V*>(mst + st * MST + piece * VN), reinterpret_cast<V*>(Mc + g2 * Lc.sb + (long)Jo * CB + piece * VN));
    }
  });
}
#endif  // RBD_SPEC_EMIT

}  // namespace spec
}  // namespace rbd
