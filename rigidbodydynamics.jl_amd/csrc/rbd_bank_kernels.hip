// rbd_bank_kernels.hip — translation unit of the two-bodies-per-lane kernels (rbd_bank.hpp): aba_bank_kernel (`dynamics!`,
// src/mechanism_algorithms.jl:845-864, the bench workload at 4096 Atlas states) and rnea_bank_kernel (`inverse_dynamics!` :542-553,
// `dynamics_bias!` :484-498), with their launchers.
#include "rbd_lane.hpp"
#include "rbd_internal.hpp"
#include "rbd_bank.hpp"

namespace rbd {

static_assert(BANK_LDS_PAIRS == 30, "rbd_capi.hip sizes the resident batch of the banked kernel with BANK_LDS_PAIRS_HOST = 30");
template <typename T> static size_t bank_lds_bytes() { return (size_t)BANK_LDS_PAIRS * 256 * sizeof(Pair2<T>); }

template <typename T>
hipError_t launch_aba_bank(const BankModel& M, long B, const void* q, const void* v, const void* tau, const void* fext, void* vdot, void* qdot,
                           Layout Lq, Layout Lv, Layout Lf, hipStream_t s, const MkFuse* fuse) {
  MkFuse F{};
  F.stage = -1;
  if (fuse) F = *fuse;
  const long spw = 64 / M.lps, waves = (B + spw - 1) / spw;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  const size_t lds = bank_lds_bytes<T>();
#define RBD_LAUNCH_BANK(FU, SI) \
  hipLaunchKernelGGL((aba_bank_kernel<T, FU, SI>), grid, block, lds, s, M, B, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, (T*)vdot, (T*)qdot, Lq, Lv, Lf, F)
  if (F.stage >= 0) { if (M.simple) RBD_LAUNCH_BANK(true, true); else RBD_LAUNCH_BANK(true, false); }
  else if (M.simple) RBD_LAUNCH_BANK(false, true);
  else RBD_LAUNCH_BANK(false, false);
#undef RBD_LAUNCH_BANK
  return hipGetLastError();
}
// the dynamic-LDS limit is a function attribute of the CURRENT device: rbd_workspace_create calls this once per workspace (after hipSetDevice)
template <typename T> hipError_t configure_bank_kernels() {
  const int lds = (int)bank_lds_bytes<T>();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_bank_kernel<T, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_bank_kernel<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_bank_kernel<T, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_bank_kernel<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  return e;
}
template hipError_t configure_bank_kernels<double>();
template hipError_t configure_bank_kernels<float>();

#ifdef RBD_PROFILE_PHASES
extern "C" int rbd_debug_bank_phase_clock(long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(rbd_bank_phase_clock), sizeof(long long) * 16);
}
#endif
template hipError_t launch_aba_bank<double>(const BankModel&, long, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, const MkFuse*);
template hipError_t launch_aba_bank<float>(const BankModel&, long, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, const MkFuse*);

template <typename T>
hipError_t launch_rnea_bank(const BankModel& M, long B, int ncol, const void* q, const void* v, const void* vdot, const void* fext, void* tau, void* qdot,
                            Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out, void* jw_out) {
  const long spw = 64 / M.lps, waves = (B + spw - 1) / spw;
  const size_t lds = (size_t)RNEA_LDS_PAIRS * 256 * sizeof(Pair2<T>);  // exchange columns: 48 KB in fp64 (the default dynamic-LDS limit is 64 KB)
  if (M.simple)
    hipLaunchKernelGGL((rnea_bank_kernel<T, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, s, M, B, ncol, (const T*)q, (const T*)v, (const T*)vdot,
                       (const T*)fext, (T*)tau, (T*)qdot, Lq, Lv, Lf, (T*)acc_out, (T*)jw_out);
  else
    hipLaunchKernelGGL((rnea_bank_kernel<T, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, s, M, B, ncol, (const T*)q, (const T*)v, (const T*)vdot,
                       (const T*)fext, (T*)tau, (T*)qdot, Lq, Lv, Lf, (T*)acc_out, (T*)jw_out);
  return hipGetLastError();
}
template hipError_t launch_rnea_bank<double>(const BankModel&, long, int, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, void*, void*);
template hipError_t launch_rnea_bank<float>(const BankModel&, long, int, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, void*, void*);

}  // namespace rbd
