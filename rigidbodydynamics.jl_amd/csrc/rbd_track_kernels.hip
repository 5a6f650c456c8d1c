// rbd_track_kernels.hip — instantiations and launcher of aba_track_kernel (rbd_track.hpp): dynamics! (src/mechanism_algorithms.jl:845-864)
// through the track mapping with canonical body frames.  A translation unit of its own so that it builds in parallel with rbd_kernels.hip.
#include "rbd_track.hpp"
#include "rbd_internal.hpp"

namespace rbd {

template <typename T, int G, bool FLT, bool GEN, int NW>
static hipError_t launch_track_g(const TrackModel& M, long B, size_t lds, const void* q, const void* v, const void* tau, const void* fext, void* vdot,
                                 void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  constexpr int SPW = 64 / G;
  const unsigned grid = (unsigned)((B + SPW - 1) / SPW);
  aba_track_kernel<T, G, FLT, GEN, NW><<<grid, 64 * NW, lds, s>>>(M, B, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, (T*)vdot, (T*)qdot, Lq, Lv, Lf);
  return hipGetLastError();
}
template <typename T, int G, int NW>
static hipError_t launch_track_fg(const TrackModel& M, int flt, int gen, long B, size_t lds, const void* q, const void* v, const void* tau, const void* fext,
                                  void* vdot, void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  if (flt) return gen ? launch_track_g<T, G, true, true, NW>(M, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s)
                      : launch_track_g<T, G, true, false, NW>(M, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
  return gen ? launch_track_g<T, G, false, true, NW>(M, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s)
             : launch_track_g<T, G, false, false, NW>(M, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
}
template <typename T, int NW>
static hipError_t launch_track_nw(const TrackModel& M, int flt, int gen, long B, size_t lds, const void* q, const void* v, const void* tau, const void* fext,
                                  void* vdot, void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  switch (M.G) {
    case 1: return launch_track_fg<T, 1, NW>(M, flt, gen, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
    case 2: return launch_track_fg<T, 2, NW>(M, flt, gen, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
    case 4: return launch_track_fg<T, 4, NW>(M, flt, gen, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
    default: return hipErrorInvalidValue;
  }
}
// nw = 4: four wavefronts per group of 64 / G states (wave 0 walks the tree, three helpers do the per-body work): small batches;
// nw = 1: one wavefront per group
template <typename T>
hipError_t launch_aba_track(const TrackModel& M, int flt, int gen, int nw, long B, size_t lds, const void* q, const void* v, const void* tau, const void* fext,
                            void* vdot, void* qdot, Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  return nw == 4 ? launch_track_nw<T, 4>(M, flt, gen, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s)
                 : launch_track_nw<T, 1>(M, flt, gen, B, lds, q, v, tau, fext, vdot, qdot, Lq, Lv, Lf, s);
}
template hipError_t launch_aba_track<double>(const TrackModel&, int, int, int, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_aba_track<float>(const TrackModel&, int, int, int, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);

// dynamic LDS above the 64 KB default needs the per-function limit raised (per device; done at workspace creation)
template <typename T, int G, bool FLT, bool GEN> static hipError_t set_lds(size_t lds) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_track_kernel<T, G, FLT, GEN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_track_kernel<T, G, FLT, GEN, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
template <typename T, int G> static hipError_t set_lds_fg(int flt, int gen, size_t lds) {
  if (flt) return gen ? set_lds<T, G, true, true>(lds) : set_lds<T, G, true, false>(lds);
  return gen ? set_lds<T, G, false, true>(lds) : set_lds<T, G, false, false>(lds);
}
template <typename T> hipError_t configure_track_kernel(int G, int flt, int gen, size_t lds) {
  switch (G) {
    case 1: return set_lds_fg<T, 1>(flt, gen, lds);
    case 2: return set_lds_fg<T, 2>(flt, gen, lds);
    case 4: return set_lds_fg<T, 4>(flt, gen, lds);
    default: return hipErrorInvalidValue;
  }
}
#ifdef RBD_PROFILE_PHASES
extern "C" int rbd_debug_track_phase_clock(long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(rbd_track_phase_clock), sizeof(long long) * 16);
}
#endif
template hipError_t configure_track_kernel<double>(int, int, int, size_t);
template hipError_t configure_track_kernel<float>(int, int, int, size_t);

}  // namespace rbd
