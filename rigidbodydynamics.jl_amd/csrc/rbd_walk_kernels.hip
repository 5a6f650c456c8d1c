// rbd_walk_kernels.hip — instantiations and launcher of aba_walk_kernel (rbd_walk.hpp): dynamics! (src/mechanism_algorithms.jl:845-864)
// with one wavefront per track and one lane per state.  A translation unit of its own so that it builds in parallel with the others.
#include "rbd_walk.hpp"
#include "rbd_internal.hpp"

namespace rbd {

// T: the kernel's value type — double, float, or f2 = two fp32 states per lane (S = float buffers, 128 states per workgroup)
// T: the kernel's value type — double, float, or f2 = two fp32 states per lane (S = float buffers, 128 states per workgroup).
// RNEA: rnea_walk_kernel (x = v̇ in, y = τ out) instead of aba_walk_kernel (x = τ in, y = v̇ out)
template <typename T, bool FLT, bool GEN, bool RNEA>
static hipError_t launch_walk_fg(const WalkModel& M, long B, size_t lds, const void* q, const void* v, const void* x, const void* fext, void* y, void* qdot,
                                 Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out = nullptr, void* jw_out = nullptr) {
  using S = typename Lanes<T>::S;
  constexpr int SPW = 64 * Lanes<T>::N;
  const unsigned grid = (unsigned)((B + SPW - 1) / SPW);
  if (RNEA) rnea_walk_kernel<T, FLT, GEN><<<grid, 64 * M.G, lds, s>>>(M, B, (const S*)q, (const S*)v, (const S*)x, (const S*)fext, (S*)y, (S*)qdot, Lq, Lv, Lf, (S*)acc_out, (S*)jw_out);
  else if (FLT && M.reroot.nchain > 0) aba_walk_kernel<T, FLT, GEN, FLT><<<grid, 64 * M.G, lds, s>>>(M, B, (const S*)q, (const S*)v, (const S*)x, (const S*)fext, (S*)y, (S*)qdot, Lq, Lv, Lf);
  else aba_walk_kernel<T, FLT, GEN><<<grid, 64 * M.G, lds, s>>>(M, B, (const S*)q, (const S*)v, (const S*)x, (const S*)fext, (S*)y, (S*)qdot, Lq, Lv, Lf);
  return hipGetLastError();
}
template <typename T, bool RNEA>
static hipError_t launch_walk_t(const WalkModel& M, int flt, int gen, long B, size_t lds, const void* q, const void* v, const void* x, const void* fext, void* y,
                                void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out = nullptr, void* jw_out = nullptr) {
  if (flt) return gen ? launch_walk_fg<T, true, true, RNEA>(M, B, lds, q, v, x, fext, y, qdot, Lq, Lv, Lf, s, acc_out, jw_out)
                      : launch_walk_fg<T, true, false, RNEA>(M, B, lds, q, v, x, fext, y, qdot, Lq, Lv, Lf, s, acc_out, jw_out);
  return gen ? launch_walk_fg<T, false, true, RNEA>(M, B, lds, q, v, x, fext, y, qdot, Lq, Lv, Lf, s, acc_out, jw_out)
             : launch_walk_fg<T, false, false, RNEA>(M, B, lds, q, v, x, fext, y, qdot, Lq, Lv, Lf, s, acc_out, jw_out);
}
// pair: (fp32 only) two states per lane
template <typename T>
hipError_t launch_aba_walk(const WalkModel& M, int flt, int gen, int pair, long B, size_t lds, const void* q, const void* v, const void* tau, const void* fext, void* vdot,
                           void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  if constexpr (sizeof(T) == 4) {
    if (pair) return launch_walk_t<f2, false>(M, flt, gen, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
  }
  return launch_walk_t<T, false>(M, flt, gen, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
}
template <typename T>
hipError_t launch_rnea_walk(const WalkModel& M, int flt, int gen, int pair, long B, size_t lds, const void* q, const void* v, const void* vdot, const void* fext, void* tau,
                            void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s, void* acc_out, void* jw_out) {
  if constexpr (sizeof(T) == 4) {
    if (pair) return launch_walk_t<f2, true>(M, flt, gen, B, lds, q, v, vdot, fext, tau, qdot, Lq, Lv, Lf, s, acc_out, jw_out);
  }
  return launch_walk_t<T, true>(M, flt, gen, B, lds, q, v, vdot, fext, tau, qdot, Lq, Lv, Lf, s, acc_out, jw_out);
}
template hipError_t launch_aba_walk<double>(const WalkModel&, int, int, int, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_aba_walk<float>(const WalkModel&, int, int, int, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_rnea_walk<double>(const WalkModel&, int, int, int, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, void*, void*);
template hipError_t launch_rnea_walk<float>(const WalkModel&, int, int, int, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t, void*, void*);

// dynamic LDS above the 64 KB default needs the per-function limit raised (per device; done at workspace creation)
template <typename T, bool FLT, bool GEN> static hipError_t set_walk_lds(size_t lds) {
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_walk_kernel<T, FLT, GEN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (FLT) {  // the form for a tree re-rooted at its centre
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_walk_kernel<T, FLT, GEN, FLT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e2 != hipSuccess) return e2;
  }
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&rnea_walk_kernel<T, FLT, GEN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
template <typename T> static hipError_t configure_walk_t(int flt, int gen, size_t lds) {
  if (flt) return gen ? set_walk_lds<T, true, true>(lds) : set_walk_lds<T, true, false>(lds);
  return gen ? set_walk_lds<T, false, true>(lds) : set_walk_lds<T, false, false>(lds);
}
template <typename T> hipError_t configure_walk_kernel(int flt, int gen, size_t lds, size_t lds_pair) {
  if (lds > 0) { const hipError_t e = configure_walk_t<T>(flt, gen, lds); if (e != hipSuccess) return e; }
  if constexpr (sizeof(T) == 4) {
    if (lds_pair > 0) return configure_walk_t<f2>(flt, gen, lds_pair);
  }
  return hipSuccess;
}
template hipError_t configure_walk_kernel<double>(int, int, size_t, size_t);
template hipError_t configure_walk_kernel<float>(int, int, size_t, size_t);
#ifdef RBD_PROFILE_PHASES
extern "C" int rbd_debug_walk_phase_clock(long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(rbd_walk_phase_clock), sizeof(long long) * 32);
}
#endif

}  // namespace rbd
