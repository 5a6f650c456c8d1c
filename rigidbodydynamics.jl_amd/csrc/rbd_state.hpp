// rbd_state.hpp — ONE LANE PER STATE kernels for large batches (>= 64 states per SIMD of the chip): mass_matrix! (CRBA) and
// inverse_dynamics! / dynamics_bias! (RNEA) of src/mechanism_algorithms.jl:248-272, :387-459, :542-553.
//
// The lane-per-body kernels (rbd_kernels.hip) spend most of their instructions on lanes that idle in a level sweep and on
// interpreting the tree per lane.  Here a lane owns a whole state and the tree is walked depth-first, so
//   * the walk is WAVE-UNIFORM: the op list, joint types, offsets and body constants come through the scalar unit (s_load, scalar
//     branches) and cost no vector instructions — the model is "compiled" onto the SALU;
//   * no cross-lane traffic, no masks, every vector instruction does 64 states' worth of useful work;
//   * the only per-state storage is the root-to-body PATH (one entry per tree level).  The transforms and the accumulators of the
//     path live in VGPRs (one wavefront per SIMD owns the whole 512-register file); registers cannot be indexed by a run-time
//     level, so the level-dependent part of an op is a `switch (level)` around a handful of register moves / adds, and all the
//     arithmetic is level-agnostic code that exists once (the whole kernel stays inside the instruction cache).  The motion
//     subspace columns of the path, which CRBA reads back once per (body, ancestor) pair, live in LDS;
//   * q (and v) are staged to LDS once, so that the walk never waits on HBM.
// Body frames are the canonical ones of rbd_state_plan.hpp (joint axis = +z): a revolute joint mixes two columns of a constant
// matrix and its motion subspace is read off the body's rotation matrix; all recursion quantities live in the ROOT frame as in the
// reference (src/mechanism_state.jl:744-748), so nothing observable depends on the re-basing.
// Scope: revolute / prismatic / fixed / sin-cos joints below optional 6-dof joints on the world, nlevels <= ML.
#pragma once
#include "rbd_device.hpp"
#include "rbd_hip.h"
#include <utility>

namespace rbd {

// `switch (lvl)` over the ML levels with a compile-time level inside each case.  The empty volatile asm keeps the optimiser from
// turning the cases into selects over every level's registers.
#define RBD_LEVEL_CASE(L, ...) case L: if constexpr (L < ML) { constexpr int LV = L; asm volatile(""); __VA_ARGS__ } break;
#define RBD_LEVEL_SWITCH(lvl, ...)                                                                                               \
  switch (lvl) {                                                                                                                 \
    RBD_LEVEL_CASE(0, __VA_ARGS__) RBD_LEVEL_CASE(1, __VA_ARGS__) RBD_LEVEL_CASE(2, __VA_ARGS__) RBD_LEVEL_CASE(3, __VA_ARGS__) RBD_LEVEL_CASE(4, __VA_ARGS__)      \
    RBD_LEVEL_CASE(5, __VA_ARGS__) RBD_LEVEL_CASE(6, __VA_ARGS__) RBD_LEVEL_CASE(7, __VA_ARGS__) RBD_LEVEL_CASE(8, __VA_ARGS__) RBD_LEVEL_CASE(9, __VA_ARGS__)      \
    RBD_LEVEL_CASE(10, __VA_ARGS__) RBD_LEVEL_CASE(11, __VA_ARGS__) RBD_LEVEL_CASE(12, __VA_ARGS__) RBD_LEVEL_CASE(13, __VA_ARGS__)                          \
    RBD_LEVEL_CASE(14, __VA_ARGS__) RBD_LEVEL_CASE(15, __VA_ARGS__)                                                                            \
    default: break;                                                                                                              \
  }

// The path: ML levels x N values.  fp32: every value has a FIXED accumulation register (AGPR BASE + level * N + k), read and written by
// one v_accvgpr move through inline asm with the register number as an immediate.  A wavefront that owns its SIMD has 256 AGPRs next to
// the 256 architectural VGPRs; the working set of the walk stays below 256 VGPRs, so the compiler has no use for AGPRs of its own
// (state_claim_agprs() makes the kernel descriptor cover all of them; build.sh checks that no compiler-generated AGPR traffic exists).
// Left to the register allocator, the same values are shuffled between the two files wholesale around every `switch (level)` and at
// the loop back-edge (measured: 2.8x the instructions).  fp64 (two registers per value): plain variables, allocator-managed —
// correct but slower; shallow trees only by default.
template <int I> using IC = std::integral_constant<int, I>;
template <int N, typename F, int... Is> RBD_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, typename F> RBD_DEV void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <typename T, int ML, int N, int BASE> struct Path {
  T v[ML][N];
  template <int L, int K> RBD_DEV void put(T x) { v[L][K] = x; }
  template <int L, int K> RBD_DEV T get() const { return v[L][K]; }
};
template <int ML, int N, int BASE> struct Path<float, ML, N, BASE> {
  static_assert(BASE + ML * N <= 256, "the path must fit the 256 accumulation registers");
  template <int L, int K> RBD_DEV void put(float x) { asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(x), "n"(BASE + L * N + K)); }
  template <int L, int K> RBD_DEV float get() const {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(BASE + L * N + K));
    return x;
  }
};
template <typename T> RBD_DEV void state_claim_agprs() {}
template <> RBD_DEV void state_claim_agprs<float>() { asm volatile("" ::: "a255"); }

// a rotation matrix is kept as its first and third columns (the second is their cross product)
template <int L, typename P, typename T> RBD_DEV void path_put_transform(P& X, const T* R, const T* p) {
  X.template put<L, 0>(R[0]); X.template put<L, 1>(R[3]); X.template put<L, 2>(R[6]);
  X.template put<L, 3>(R[2]); X.template put<L, 4>(R[5]); X.template put<L, 5>(R[8]);
  X.template put<L, 6>(p[0]); X.template put<L, 7>(p[1]); X.template put<L, 8>(p[2]);
}
template <int L, typename P, typename T> RBD_DEV void path_get_transform(const P& X, T* R, T* p) {
  const T c0[3] = {X.template get<L, 0>(), X.template get<L, 1>(), X.template get<L, 2>()};
  const T c2[3] = {X.template get<L, 3>(), X.template get<L, 4>(), X.template get<L, 5>()};
  T c1[3];
  cross3(c2, c0, c1);
  R[0] = c0[0]; R[3] = c0[1]; R[6] = c0[2];
  R[1] = c1[0]; R[4] = c1[1]; R[7] = c1[2];
  R[2] = c2[0]; R[5] = c2[1]; R[8] = c2[2];
  p[0] = X.template get<L, 6>(); p[1] = X.template get<L, 7>(); p[2] = X.template get<L, 8>();
}

// (Rl, pl) = joint_to_predecessor * joint_transform(q) in canonical frames; qs: this lane's column of the staged q
template <typename T> RBD_DEV void state_local_transform(int jt, const T* r, const T* qs, int qoff, T* Rl, T* pl) {
  const T* C = r + TR_C;
    _Pragma("unroll")
  for (int k = 0; k < 3; ++k) pl[k] = r[TR_PP + k];
  if (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_SINCOS_REVOLUTE) {
    T s, c;
    if (jt == RBD_JOINT_REVOLUTE) sincos_fast(qs[qoff * 64], &s, &c);
    else { s = qs[qoff * 64]; c = qs[(qoff + 1) * 64]; }
    _Pragma("unroll")
    for (int i = 0; i < 3; ++i) {
      Rl[3 * i] = c * C[3 * i] + s * C[3 * i + 1];
      Rl[3 * i + 1] = c * C[3 * i + 1] - s * C[3 * i];
      Rl[3 * i + 2] = C[3 * i + 2];
    }
  } else if (jt == RBD_JOINT_QUAT_FLOATING) {
    T Rq[9], pq[3], t[3];
    rot_quat(qs[qoff * 64], qs[(qoff + 1) * 64], qs[(qoff + 2) * 64], qs[(qoff + 3) * 64], Rq);
    pq[0] = qs[(qoff + 4) * 64]; pq[1] = qs[(qoff + 5) * 64]; pq[2] = qs[(qoff + 6) * 64];
    matmul3(C, Rq, Rl);
    matvec3(C, pq, t);
    _Pragma("unroll")
    for (int k = 0; k < 3; ++k) pl[k] += t[k];
  } else {
    _Pragma("unroll")
    for (int k = 0; k < 9; ++k) Rl[k] = C[k];
    if (jt == RBD_JOINT_PRISMATIC) {
      const T d = qs[qoff * 64];
    _Pragma("unroll")
      for (int k = 0; k < 3; ++k) pl[k] += d * C[3 * k + 2];
    }
  }
}
// root-frame motion subspace column of a 1-dof joint from the body's transform (canonical axis +z)
template <typename T> RBD_DEV void state_subspace(int jt, const T* R, const T* p, T* S) {
  const T z[3] = {R[2], R[5], R[8]};
  if (jt == RBD_JOINT_PRISMATIC) {
    S[0] = S[1] = S[2] = T(0); S[3] = z[0]; S[4] = z[1]; S[5] = z[2];
  } else {  // revolute / sin-cos (a fixed joint's column is never used)
    S[0] = z[0]; S[1] = z[1]; S[2] = z[2];
    cross3(p, z, S + 3);
  }
}
// transform to root of the body entered at level lvl from its parent's on the path; kept on the path and returned in (R, p)
template <typename T, int ML, typename P> RBD_DEV void state_compose(P& PX, int lvl, const T* Rl, const T* pl, T* R, T* p) {
  RBD_LEVEL_SWITCH(lvl, {
    if constexpr (LV == 0) {
    _Pragma("unroll")
      for (int k = 0; k < 9; ++k) R[k] = Rl[k];
    _Pragma("unroll")
      for (int k = 0; k < 3; ++k) p[k] = pl[k];
    } else {
      T Rp[9], pp[3], t[3];
      path_get_transform<LV - 1>(PX, Rp, pp);
      matmul3(Rp, Rl, R);
      matvec3(Rp, pl, t);
    _Pragma("unroll")
      for (int k = 0; k < 3; ++k) p[k] = pp[k] + t[k];
    }
    path_put_transform<LV>(PX, R, p);
  })
}

// The op tables (ints: 4 op words + SC_STRIDE ancestor columns per op; reals: TR_STRIDE body constants per op) are copied to LDS once per
// workgroup: a scalar-cache miss per op would serialise ~2 HBM round trips per op on a cold cache (measured: 100 us of a 120 us lone wavefront).
// LDS layout: reals [nops * TR_STRIDE] | ints [nops * SI_STRIDE] | per-wave areas.
enum { SI_STRIDE = 4 + SC_STRIDE };
template <typename T> RBD_DEV T* state_stage_tables(const StateModel& M, unsigned char* lds, const T** tr, const int32_t** ti) {
  T* R = reinterpret_cast<T*>(lds);
  int32_t* I = reinterpret_cast<int32_t*>(R + (size_t)M.nops * TR_STRIDE);
  const T* sr = reinterpret_cast<const T*>(M.sr);
  for (int i = threadIdx.x; i < M.nops * TR_STRIDE; i += blockDim.x) R[i] = sr[i];
  for (int i = threadIdx.x; i < M.nops * SI_STRIDE; i += blockDim.x) {
    const int o = i / SI_STRIDE, k = i - o * SI_STRIDE;
    I[i] = k < 4 ? M.ops[o * SO_STRIDE + (k == 0 ? SO_W0 : k == 1 ? SO_QOFF : k == 2 ? SO_VOFF : SO_ORIG6)] : M.cols[o * SC_STRIDE + (k - 4)];
  }
  *tr = R; *ti = I;
  size_t bytes = (size_t)M.nops * TR_STRIDE * sizeof(T) + (size_t)M.nops * SI_STRIDE * sizeof(int32_t);
  bytes = (bytes + 15) & ~(size_t)15;
  return reinterpret_cast<T*>(lds + bytes);
}
RBD_DEV int state_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// stage n rows of a batch buffer into LDS as [k][lane]
template <typename T> RBD_DEV void state_stage(const T* __restrict__ src, Layout L, long state, int n, T* dst) {
  const int lane = threadIdx.x & 63;
  const T* base = src + state * L.sb;
#pragma unroll 8
  for (int k = 0; k < n; ++k) dst[k * 64 + lane] = base[(long)k * L.sk];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// CRBA.  ENTER(body): transform, motion subspace, own inertia in the root frame.  EXIT(body): the subtree below is finished, so
// Ic is the composite inertia (update_crb_inertias!, src/mechanism_state.jl:852-868): F = Ic S, M[row, col_j] = F . S_j for the
// joints j on the path (mass_matrix!: src/mechanism_algorithms.jl:258-267), then Ic is added to the parent's.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T, int ML>
__global__ __launch_bounds__(256) void crba_state_kernel(StateModel M, long B, const T* __restrict__ q, T* __restrict__ Mout, Layout Lq, Layout Lm,
                                                       int zero_fill) {
  extern __shared__ __align__(16) unsigned char state_lds_raw[];
  const T* tr;
  const int32_t* ti;
  T* qs = state_stage_tables<T>(M, state_lds_raw, &tr, &ti);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  qs += (size_t)wave * (M.nq + 6 * M.nlevels) * 64;
  T* Sl = qs + (size_t)M.nq * 64 + lane;  // motion subspace columns of the path: Sl[(6 * level + k) * 64]
  qs += lane;
  const long state_raw = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 64 + lane;
  const bool live = state_raw < B;
  const long state = live ? state_raw : B - 1;
  state_stage(q, Lq, state, M.nq, qs - lane);
  T* Mlane = Mout + layout_base(Lm, state);
  const long nv = M.nv, msk = Lm.sk;
  auto put = [&](long row, long col, T x) {
    if (live) Mlane[(col * nv + row) * msk] = x;
  };
  // structural zeros of the lower triangle (the reference writes them too: mechanism_algorithms.jl:266-267)
  if (zero_fill && live) {
    for (int row = 0; row < M.nv; ++row) {
      unsigned long long m = ~M.row_mask[row] & ((row >= 63) ? ~0ull : ((2ull << row) - 1));
      while (m) {
        const long col = __builtin_ctzll(m);
        m &= m - 1;
        Mlane[(col * nv + row) * msk] = T(0);
      }
    }
  }
  __syncthreads();                // the op tables (every wave of the workgroup reads them); own q column
  state_claim_agprs<T>();
  Path<T, ML, 9, 0> PX;           // path: transforms to root
  Path<T, ML, 10, 9 * ML> PI;     // path: inertias being accumulated (J 6, c 3, m)
  T R0[9], p0[3];                 // transform of the level-0 body (a 6-dof root's columns are S' F with S = Ad(H0))
  // An op's words and constants are asked for while the op before it computes (registers, two sets taken in turn): read at the top of
  // their own op, every LDS -> scalar-register hop (kind, level, offsets, each ancestor column) was a round trip the lone wavefront sat out.
  auto fetch = [&](int o, int32_t* w, T* r) {
    const int oo = o < M.nops ? o : M.nops - 1;
    _Pragma("unroll")
    for (int k = 0; k < 4 + ML; ++k) w[k] = ti[oo * SI_STRIDE + k];
    _Pragma("unroll")
    for (int k = 0; k < TR_STRIDE; ++k) r[k] = tr[oo * TR_STRIDE + k];
  };
  auto step = [&](const int32_t* op, const T* r) {
    const int w0 = state_uniform(op[0]), lvl = (w0 >> 8) & 0xff, jt = w0 >> 16, voff = state_uniform(op[2]);
    if ((w0 & 0xff) == SK_ENTER) {
      T Rl[9], pl[3], R[9], p[3], S[6];
      state_local_transform(jt, r, qs, state_uniform(op[1]), Rl, pl);
      state_compose<T, ML>(PX, lvl, Rl, pl, R, p);
      if (lvl == 0) {
    _Pragma("unroll")
        for (int k = 0; k < 9; ++k) R0[k] = R[k];
    _Pragma("unroll")
        for (int k = 0; k < 3; ++k) p0[k] = p[k];
      }
      state_subspace(jt, R, p, S);
    _Pragma("unroll")
      for (int k = 0; k < 6; ++k) Sl[(6 * lvl + k) * 64] = S[k];
      RInertia<T> Ib;
      inertia_to_root(r + TR_J, r + TR_MC, r[TR_M], R, p, Ib);
      RBD_LEVEL_SWITCH(lvl, {
        static_for<6>([&](auto k) { PI.template put<LV, k.value>(Ib.J[k.value]); });
        static_for<3>([&](auto k) { PI.template put<LV, 6 + k.value>(Ib.c[k.value]); });
        PI.template put<LV, 9>(Ib.m);
      })
    } else {
      RInertia<T> Ic;
      RBD_LEVEL_SWITCH(lvl, {
        static_for<6>([&](auto k) { Ic.J[k.value] = PI.template get<LV, k.value>(); });
        static_for<3>([&](auto k) { Ic.c[k.value] = PI.template get<LV, 6 + k.value>(); });
        Ic.m = PI.template get<LV, 9>();
        if constexpr (LV > 0) {
          static_for<6>([&](auto k) { PI.template put<LV - 1, k.value>(PI.template get<LV - 1, k.value>() + Ic.J[k.value]); });
          static_for<3>([&](auto k) { PI.template put<LV - 1, 6 + k.value>(PI.template get<LV - 1, 6 + k.value>() + Ic.c[k.value]); });
          PI.template put<LV - 1, 9>(PI.template get<LV - 1, 9>() + Ic.m);
        }
      })
      if (jt == RBD_JOINT_QUAT_FLOATING) {  // level 0: the 6 x 6 block S' Ic S with S = Ad(H)
#pragma unroll 1
        for (int ci = 0; ci < 6; ++ci) {
          T e[6], Si[6], Fc[6], o6[6];
    _Pragma("unroll")
          for (int k = 0; k < 6; ++k) e[k] = (k == ci) ? T(1) : T(0);
          xmotion(R0, p0, e, Si);
          mul_inertia(Ic, Si, Fc);
          xforce_inv(R0, p0, Fc, o6);
    _Pragma("unroll")
          for (int cj = 0; cj < 6; ++cj)
            if (cj <= ci) put(voff + ci, voff + cj, o6[cj]);
        }
      } else if (jt != RBD_JOINT_FIXED) {
        T S[6], F[6];
    _Pragma("unroll")
        for (int k = 0; k < 6; ++k) S[k] = Sl[(6 * lvl + k) * 64];
        mul_inertia(Ic, S, F);
        const long row = voff;
        put(row, row, dot6(F, S));
        // the ancestors of the path, level by level (unrolled: the columns of several levels are in flight from LDS together)
    _Pragma("unroll")
        for (int k = 0; k < ML; ++k) {
          if (k < lvl) {
            const int col = state_uniform(op[4 + k]);
            if (col >= 0) {
              if (col & SC_FLOATING) {
                T o6[6];
                xforce_inv(R0, p0, F, o6);
    _Pragma("unroll")
                for (int cj = 0; cj < 6; ++cj) put(row, (col & ~SC_FLOATING) + cj, o6[cj]);
              } else {
                T Sk[6];
    _Pragma("unroll")
                for (int j = 0; j < 6; ++j) Sk[j] = Sl[(6 * k + j) * 64];
                put(row, col, dot6(F, Sk));
              }
            }
          }
        }
      }
    }
  };
  int32_t wa[4 + ML];
  T ra[TR_STRIDE];
  if constexpr (sizeof(T) == 4) {
    int32_t wb[4 + ML];
    T rb[TR_STRIDE];
    fetch(0, wa, ra);
#pragma unroll 1
    for (int o = 0; o < M.nops; o += 2) {
      fetch(o + 1, wb, rb);
      step(wa, ra);
      fetch(o + 2, wa, ra);
      if (o + 1 < M.nops) step(wb, rb);
    }
  } else {  // fp64 keeps the path in VGPRs (no accumulation-register form): no room for a second set
#pragma unroll 1
    for (int o = 0; o < M.nops; ++o) {
      fetch(o, wa, ra);
      step(wa, ra);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// RNEA (inverse_dynamics! :542-553 / dynamics_bias! with v̇ = nullptr).  ENTER(body): transform, twist, spatial acceleration
// (spatial_accelerations! :387-417, root acceleration = -gravity), net wrench f = I a + T x* (I T) - w_ext (newton_euler! :431-440).
// EXIT(body): tau = S' f (joint_wrenches_and_torques! :442-459), f is added to the parent's.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T, int ML>
__global__ __launch_bounds__(256) void rnea_state_kernel(StateModel M, long B, const T* __restrict__ q, const T* __restrict__ v,
                                                       const T* __restrict__ vdot, const T* __restrict__ fext, T* __restrict__ tau,
                                                       T* __restrict__ qdot, Layout Lq, Layout Lv, Layout Lf) {
  extern __shared__ __align__(16) unsigned char state_lds_raw[];
  const T* tr;
  const int32_t* ti;
  T* qs = state_stage_tables<T>(M, state_lds_raw, &tr, &ti);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  qs += (size_t)wave * (M.nq + M.nv + 6 * M.nlevels) * 64;
  T* vs = qs + (size_t)M.nq * 64 + lane;
  T* Fl = vs + (size_t)M.nv * 64;  // net wrenches of the path, accumulated bottom-up: Fl[(6 * level + k) * 64]
  qs += lane;
  const long state_raw = ((long)blockIdx.x * (blockDim.x >> 6) + wave) * 64 + lane;
  const bool live = state_raw < B;
  const long state = live ? state_raw : B - 1;
  state_stage(q, Lq, state, M.nq, qs - lane);
  state_stage(v, Lv, state, M.nv, vs - lane);
  const T* vdl = vdot ? vdot + state * Lv.sb : nullptr;
  const T* fel = fext ? fext + state * Lf.sb : nullptr;
  T* taul = tau + state * Lv.sb;
  T* qdl = (qdot && live) ? qdot + state * Lq.sb : nullptr;
  const long vsk = Lv.sk, qsk = Lq.sk, fsk = Lf.sk;
  __syncthreads();
  state_claim_agprs<T>();
  Path<T, ML, 9, 0> PX;                    // path: transforms to root
  Path<T, ML, 12, 9 * ML> PK;              // path: twists (6), accelerations (6)
#pragma unroll 1
  for (int o = 0; o < M.nops; ++o) {
    const int32_t* op = ti + o * SI_STRIDE;
    const int w0 = state_uniform(op[0]), lvl = (w0 >> 8) & 0xff, jt = w0 >> 16, voff = state_uniform(op[2]);
    if ((w0 & 0xff) == SK_ENTER) {
      T r[TR_STRIDE];
    _Pragma("unroll")
      for (int k = 0; k < TR_STRIDE; ++k) r[k] = tr[o * TR_STRIDE + k];
      const int qoff = state_uniform(op[1]), orig6 = state_uniform(op[3]);
      T we[6];  // external wrench of this body: asked for first, used last
    _Pragma("unroll")
      for (int k = 0; k < 6; ++k) we[k] = fel ? fel[(long)(orig6 + k) * fsk] : T(0);
      T Rl[9], pl[3], R[9], p[3];
      state_local_transform(jt, r, qs, qoff, Rl, pl);
      state_compose<T, ML>(PX, lvl, Rl, pl, R, p);
      T vJ[6], aJ[6];
      if (jt == RBD_JOINT_QUAT_FLOATING) {
        T vj[6], aj[6];
    _Pragma("unroll")
        for (int k = 0; k < 6; ++k) { vj[k] = vs[(voff + k) * 64]; aj[k] = vdl ? vdl[(long)(voff + k) * vsk] : T(0); }
        xmotion(R, p, vj, vJ);
        xmotion(R, p, aj, aJ);
        if (qdl) {  // velocity_to_configuration_derivative! (quaternion_floating.jl:126-136)
          const T w = qs[qoff * 64], x = qs[(qoff + 1) * 64], y = qs[(qoff + 2) * 64], z = qs[(qoff + 3) * 64];
          T Rq[9], lin[3];
          rot_quat(w, x, y, z, Rq);
          matvec3(Rq, vj + 3, lin);
          qdl[(long)qoff * qsk] = (-x * vj[0] - y * vj[1] - z * vj[2]) / 2;
          qdl[(long)(qoff + 1) * qsk] = (w * vj[0] - z * vj[1] + y * vj[2]) / 2;
          qdl[(long)(qoff + 2) * qsk] = (z * vj[0] + w * vj[1] - x * vj[2]) / 2;
          qdl[(long)(qoff + 3) * qsk] = (-y * vj[0] + x * vj[1] + w * vj[2]) / 2;
    _Pragma("unroll")
          for (int k = 0; k < 3; ++k) qdl[(long)(qoff + 4 + k) * qsk] = lin[k];
        }
      } else if (jt == RBD_JOINT_FIXED) {
    _Pragma("unroll")
        for (int k = 0; k < 6; ++k) { vJ[k] = T(0); aJ[k] = T(0); }
      } else {
        T S[6];
        state_subspace(jt, R, p, S);
        const T qd = vs[voff * 64];
        const T vd = vdl ? vdl[(long)voff * vsk] : T(0);
    _Pragma("unroll")
        for (int k = 0; k < 6; ++k) { vJ[k] = S[k] * qd; aJ[k] = S[k] * vd; }
        if (qdl) {
          if (jt == RBD_JOINT_SINCOS_REVOLUTE) {  // sin_cos_revolute.jl: d/dt (sin, cos) = (cos, -sin) q̇
            qdl[(long)qoff * qsk] = qs[(qoff + 1) * 64] * qd;
            qdl[(long)(qoff + 1) * qsk] = -qs[qoff * 64] * qd;
          } else {
            qdl[(long)qoff * qsk] = qd;
          }
        }
      }
      T Tw[6], a[6];
      RBD_LEVEL_SWITCH(lvl, {
        if constexpr (LV == 0) {
    _Pragma("unroll")
          for (int k = 0; k < 6; ++k) { Tw[k] = vJ[k]; a[k] = aJ[k]; }
          a[3] -= T(M.gravity[0]); a[4] -= T(M.gravity[1]); a[5] -= T(M.gravity[2]);
        } else {
          T nT[6], cr[6], Tp[6];
          static_for<6>([&](auto k) { Tp[k.value] = PK.template get<LV - 1, k.value>(); });
    _Pragma("unroll")
          for (int k = 0; k < 6; ++k) { Tw[k] = Tp[k] + vJ[k]; nT[k] = -Tw[k]; }
          se3_comm(nT, Tp, cr);  // a_b = a_p + (-T_b) x T_p + S v̇   (mechanism_algorithms.jl:414)
          static_for<6>([&](auto k) { a[k.value] = PK.template get<LV - 1, 6 + k.value>() + cr[k.value] + aJ[k.value]; });
        }
        static_for<6>([&](auto k) { PK.template put<LV, k.value>(Tw[k.value]); PK.template put<LV, 6 + k.value>(a[k.value]); });
      })
      RInertia<T> Ib;
      inertia_to_root(r + TR_J, r + TR_MC, r[TR_M], R, p, Ib);
      T Ia[6], cx[6], f[6];
      mul_inertia(Ib, a, Ia);
      momentum_cross(Ib, Tw, cx);
    _Pragma("unroll")
      for (int k = 0; k < 6; ++k) f[k] = Ia[k] + cx[k] - we[k];
    _Pragma("unroll")
      for (int k = 0; k < 6; ++k) Fl[(6 * lvl + k) * 64] = f[k];
    } else {
      T f[6], R[9], p[3];
    _Pragma("unroll")
      for (int k = 0; k < 6; ++k) f[k] = Fl[(6 * lvl + k) * 64];
      if (lvl > 0) {
    _Pragma("unroll")
        for (int k = 0; k < 6; ++k) Fl[(6 * (lvl - 1) + k) * 64] += f[k];
      }
      RBD_LEVEL_SWITCH(lvl, { path_get_transform<LV>(PX, R, p); })
      if (jt == RBD_JOINT_QUAT_FLOATING) {
        T o6[6];
        xforce_inv(R, p, f, o6);
        if (live) {
    _Pragma("unroll")
          for (int k = 0; k < 6; ++k) taul[(long)(voff + k) * vsk] = o6[k];
        }
      } else if (jt != RBD_JOINT_FIXED) {
        T S[6];
        state_subspace(jt, R, p, S);
        const T t = dot6(S, f);
        if (live) taul[(long)voff * vsk] = t;
      }
    }
  }
}

}  // namespace rbd
