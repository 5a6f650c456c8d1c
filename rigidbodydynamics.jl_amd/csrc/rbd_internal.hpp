// rbd_internal.hpp — host-side declarations shared by rbd_kernels.hip and rbd_capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "rbd_device.hpp"

namespace rbd {
template <typename T>
hipError_t launch_aba(const DevModel& M, long B, const void* q, const void* v, const void* tau, const void* fext, void* vdot,
                      void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s, const MkFuse* fuse = nullptr);
template <typename T>
hipError_t launch_rnea(const DevModel& M, long B, const void* q, const void* v, const void* vdot, const void* fext, void* tau,
                       void* qdot, void* body_out, Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out = nullptr, void* jw_out = nullptr);
template <typename T> hipError_t launch_max_diff(long n, int nk, const void* a, const void* b, Layout L, double* out, hipStream_t s);  // out[0] = max |a - b|, out[1] = max |b| over the first n states
template <typename T>
hipError_t launch_crba(const DevModel& M, long B, const void* q, void* Mout, Layout Lq, Layout Lm, int zero_fill, hipStream_t s);
}  // namespace rbd
namespace rbd {
template <typename T>
hipError_t launch_chol_solve(int nv, long B, const void* M, const void* tau, const void* c, void* x, void* Lout, Layout Lm, Layout Lv,
                             int* notpd, hipStream_t s, void* Mcopy = nullptr, Layout Lc = Layout{0, 0});
bool chol_copies_m(int element_size, int nv);  // the kernel launch_chol_solve picks can also write M in a second layout (Mcopy)
// the lower triangle of B nv x nv matrices (column-major per state, layout Lm) as LAPACK's packed 'L' storage (layout Lp over nv (nv + 1) / 2 values per state)
template <typename T> hipError_t launch_pack_lower(int nv, long B, const void* M, void* P, Layout Lm, Layout Lp, hipStream_t s);
}
namespace rbd {
template <typename T>
hipError_t launch_loop_solve(const LoopView<T>& V, long B, int stabilize, const void* body, const void* M, const void* c, const void* tau, void* vdot,
                             void* lambda, void* K, void* k, void* scratch, long scratch_stride, Layout Lm, Layout Lv, Layout Lc, Layout Lk,
                             const double* gravity, int* notpd, hipStream_t s);
}
namespace rbd {
template <typename T>
bool launch_loop_fused(const LoopView<T>& V, long B, int stabilize, const void* q, const void* v, const void* tau, const void* fext, void* body, void* M, void* c,
                       void* vdot, void* qdot, void* lambda, void* K, void* k, Layout Lq, Layout Lm, Layout Lv, Layout Lf, Layout Lc, Layout Lk,
                       const double* gravity, int* notpd, hipStream_t s);
}
namespace rbd {
template <typename T>
hipError_t launch_mk_stage(const DevModel& M, long B, int stage, double dt, void* q, void* v, const void* vdot_prev, const MkBuffers& W,
                           Layout Lq, Layout Lv, hipStream_t s, int close_prev = 0);
}
namespace rbd {
template <typename T>
hipError_t launch_pd_control(const DevModel& M, long B, const void* q, const void* v, const void* tau_ff, const void* qdes, const void* kp, const void* kd,
                             void* tau_out, Layout Lq, Layout Lv, hipStream_t s);
}
namespace rbd {
template <typename T>
hipError_t launch_kin(const DevModel& M, long B, const void* q, const void* v, void* A, void* com, void* energy, void* J, uint64_t jplus,
                      uint64_t jminus, Layout Lq, Layout Lv, Layout La, Layout L3, Layout L2, hipStream_t s);
}
namespace rbd {
template <typename T>
hipError_t launch_aba_bank(const BankModel& M, long B, const void* q, const void* v, const void* tau, const void* fext, void* vdot, void* qdot,
                           Layout Lq, Layout Lv, Layout Lf, hipStream_t s, const MkFuse* fuse = nullptr);
}
namespace rbd {
template <typename T>
hipError_t launch_rnea_bank(const BankModel& M, long B, int ncol, const void* q, const void* v, const void* vdot, const void* fext, void* tau, void* qdot,
                            Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out = nullptr, void* jw_out = nullptr);
}
namespace rbd {
template <typename T> hipError_t configure_bank_kernels();  // rbd_bank_kernels.hip
}
namespace rbd {
template <typename T>
hipError_t launch_momentum(const DevModel& M, long B, const void* q, const void* v, void* mom, Layout Lq, Layout Lv, Layout L12, hipStream_t s);
}
namespace rbd {
// rbd_contact_kernels.hip: soft contact
template <typename T>
hipError_t launch_contact(const ContactModel& M, long B, const void* body, void* s, void* sdot, const void* fext, void* contactwrenches, void* totalwrenches,
                          Layout Ls, Layout Lf, hipStream_t st);
template <typename T> hipError_t launch_contact_stage(long n, int stage, double dt, void* s, const void* sdot, void* s0, void* acc, hipStream_t st);
// rbd_walk_kernels.hip: one wavefront per track, one lane per state
template <typename T>
hipError_t launch_aba_walk(const WalkModel& M, int flt, int gen, int pair, long B, size_t lds_bytes, const void* q, const void* v, const void* tau, const void* fext,
                           void* vdot, void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s);
template <typename T>
hipError_t launch_rnea_walk(const WalkModel& M, int flt, int gen, int pair, long B, size_t lds_bytes, const void* q, const void* v, const void* vdot, const void* fext,
                            void* tau, void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out = nullptr, void* jw_out = nullptr);
template <typename T> hipError_t configure_walk_kernel(int flt, int gen, size_t lds_bytes, size_t lds_bytes_pair);
// rbd_state_kernels.hip: one lane per state
int state_max_levels(int element_size);
template <typename T>
hipError_t launch_crba_state(const StateModel& M, long B, const void* q, void* Mout, Layout Lq, Layout Lm, int zero_fill, hipStream_t s);
template <typename T>
hipError_t launch_rnea_state(const StateModel& M, long B, const void* q, const void* v, const void* vdot, const void* fext, void* tau, void* qdot,
                             Layout Lq, Layout Lv, Layout Lf, hipStream_t s);
}
namespace rbd {
// rbd_big_kernels.hip: the any-size fallback (one thread per state, HBM scratch of big_scratch_elems() scalars)
size_t big_scratch_elems(const BigModel& M, long B);
template <typename T>
hipError_t launch_big_rnea(const BigModel& M, long B, const void* q, const void* v, const void* vdot, const void* fext, void* tau, void* qdot, void* scratch,
                           void* acc_out, void* jw_out, Layout Lq, Layout Lv, Layout Lf, hipStream_t s);
template <typename T>
hipError_t launch_big_crba(const BigModel& M, long B, const void* q, void* Mout, void* scratch, Layout Lq, Layout Lm, hipStream_t s);
template <typename T> hipError_t launch_big_export_body(const BigModel& M, long B, const void* scratch, void* body, hipStream_t s);
template <typename T>
hipError_t launch_big_mk_stage(const BigModel& M, long B, int stage, double dt, void* q, void* v, const void* vdot_prev, const MkBuffers& W, Layout Lq, Layout Lv,
                               hipStream_t s, int close_prev = 0);
// dst[b * n + k] = src[k * ld + b] for one or two (src1 != nullptr) buffer pairs (n <= 384: per-body outputs of a one-lane-per-state kernel, stored
// batch-innermost, for a state-major caller)
template <typename T> hipError_t launch_rows_to_state_major(int n, long B, long ld, const void* src0, void* dst0, const void* src1, void* dst1, hipStream_t s);
// the kinematics by-products and the PD law of rbd_simulate_controlled on trees of any size (round 6)
template <typename T>
hipError_t launch_big_kin(const BigModel& M, long B, const void* q, const void* v, void* A, void* com, void* energy, void* J, int base, int target, void* mom,
                          void* scratch, Layout Lq, Layout Lv, Layout La, Layout L3, Layout L2, Layout L12, hipStream_t s);
template <typename T>
hipError_t launch_big_pd_control(const BigModel& M, long B, const void* q, const void* v, const void* tau_ff, const void* qdes, const void* kp, const void* kd,
                                 void* tau_out, Layout Lq, Layout Lv, hipStream_t s);
template <typename T>
hipError_t launch_big_chol_solve(int nv, long B, const void* Mg, void* Lg, const void* rhs, const void* c, void* x, Layout Lm, Layout Lv, int* notpd, hipStream_t s);
}
