// rbd_pipe_kernels.hip — instantiations and launcher of aba_pipe_kernel (rbd_pipe.hpp): dynamics! (src/mechanism_algorithms.jl:845-864)
// for small batches, a body-step cut into stages on the four SIMDs of a compute unit.  A translation unit of its own (parallel build).
#include "rbd_pipe.hpp"
#include "rbd_internal.hpp"

namespace rbd {

template <typename T>
hipError_t launch_aba_pipe(const WalkModel& M, long B, size_t lds, const void* q, const void* v, const void* tau, const void* fext, void* vdot, void* qdot,
                           Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  const unsigned grid = (unsigned)((B + PIPE_STATES - 1) / PIPE_STATES);
  aba_pipe_kernel<T><<<grid, 256, lds, s>>>(M, B, (const T*)q, (const T*)v, (const T*)tau, (const T*)fext, (T*)vdot, (T*)qdot, Lq, Lv, Lf);
  return hipGetLastError();
}
template hipError_t launch_aba_pipe<double>(const WalkModel&, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_aba_pipe<float>(const WalkModel&, long, size_t, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);

// dynamic LDS above the 64 KB default needs the per-function limit raised (per device; done at workspace creation)
template <typename T> hipError_t configure_pipe_kernel(size_t lds) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&aba_pipe_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
template hipError_t configure_pipe_kernel<double>(size_t);
template hipError_t configure_pipe_kernel<float>(size_t);

#ifdef RBD_PROFILE_PHASES
extern "C" int rbd_debug_pipe_phase_clock(long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(rbd_pipe_phase_clock), sizeof(long long) * 32);
}
#endif

}  // namespace rbd
