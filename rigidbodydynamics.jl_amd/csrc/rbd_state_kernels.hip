// rbd_state_kernels.hip — launchers of the one-lane-per-state kernels (rbd_state.hpp): own translation unit (the per-level
// instantiations are large).
#include "rbd_state.hpp"
#include "rbd_internal.hpp"

namespace rbd {

template <typename T> struct StateLevels;
template <> struct StateLevels<float> { static constexpr int SMALL = 6, LARGE = 12; };
template <> struct StateLevels<double> { static constexpr int SMALL = 6, LARGE = 12; };

// workgroup shape: as many of a CU's four wavefronts per workgroup as fit its 160 KB of LDS next to one copy of the op tables
static bool state_block(const StateModel& M, size_t es, size_t per_wave, int* wpb, size_t* lds) {
  size_t tables = (size_t)M.nops * TR_STRIDE * es + (size_t)M.nops * (4 + SC_STRIDE) * sizeof(int32_t);
  tables = (tables + 15) & ~(size_t)15;
  for (int w = 4; w >= 1; w >>= 1) {
    if (tables + (size_t)w * per_wave <= 160u * 1024u) { *wpb = w; *lds = tables + (size_t)w * per_wave; return true; }
  }
  return false;
}
// above the default 64 KB of dynamic LDS the limit of the kernel has to be raised (once per instantiation)
#define STATE_LDS_ATTR(KERNEL)                                                                                                   \
  do {                                                                                                                           \
    static bool raised = false;                                                                                                  \
    if (lds > 64u * 1024u && !raised) {                                                                                          \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      if (e != hipSuccess) return e;                                                                                             \
      raised = true;                                                                                                             \
    }                                                                                                                            \
  } while (0)

int state_max_levels(int es) { return es == 4 ? StateLevels<float>::LARGE : StateLevels<double>::LARGE; }

template <typename T>
hipError_t launch_crba_state(const StateModel& M, long B, const void* q, void* Mout, Layout Lq, Layout Lm, int zero_fill, hipStream_t s) {
  const size_t per_wave = (size_t)(M.nq + 6 * M.nlevels) * 64 * sizeof(T);  // staged q, motion subspace columns of the path
  int wpb;
  size_t lds;
  if (!state_block(M, sizeof(T), per_wave, &wpb, &lds)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((B + 64L * wpb - 1) / (64L * wpb)));
  if (M.nlevels <= StateLevels<T>::SMALL) {
    STATE_LDS_ATTR((crba_state_kernel<T, StateLevels<T>::SMALL>));
    hipLaunchKernelGGL((crba_state_kernel<T, StateLevels<T>::SMALL>), grid, dim3(64 * wpb), lds, s, M, B, (const T*)q, (T*)Mout, Lq, Lm, zero_fill);
  } else if (M.nlevels <= StateLevels<T>::LARGE) {
    STATE_LDS_ATTR((crba_state_kernel<T, StateLevels<T>::LARGE>));
    hipLaunchKernelGGL((crba_state_kernel<T, StateLevels<T>::LARGE>), grid, dim3(64 * wpb), lds, s, M, B, (const T*)q, (T*)Mout, Lq, Lm, zero_fill);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <typename T>
hipError_t launch_rnea_state(const StateModel& M, long B, const void* q, const void* v, const void* vdot, const void* fext, void* tau, void* qdot,
                             Layout Lq, Layout Lv, Layout Lf, hipStream_t s) {
  const size_t per_wave = (size_t)(M.nq + M.nv + 6 * M.nlevels) * 64 * sizeof(T);  // staged q, v; net wrenches of the path
  int wpb;
  size_t lds;
  if (!state_block(M, sizeof(T), per_wave, &wpb, &lds)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((B + 64L * wpb - 1) / (64L * wpb)));
  if (M.nlevels <= StateLevels<T>::SMALL) {
    STATE_LDS_ATTR((rnea_state_kernel<T, StateLevels<T>::SMALL>));
    hipLaunchKernelGGL((rnea_state_kernel<T, StateLevels<T>::SMALL>), grid, dim3(64 * wpb), lds, s, M, B, (const T*)q, (const T*)v, (const T*)vdot,
                       (const T*)fext, (T*)tau, (T*)qdot, Lq, Lv, Lf);
  } else if (M.nlevels <= StateLevels<T>::LARGE) {
    STATE_LDS_ATTR((rnea_state_kernel<T, StateLevels<T>::LARGE>));
    hipLaunchKernelGGL((rnea_state_kernel<T, StateLevels<T>::LARGE>), grid, dim3(64 * wpb), lds, s, M, B, (const T*)q, (const T*)v, (const T*)vdot,
                       (const T*)fext, (T*)tau, (T*)qdot, Lq, Lv, Lf);
  } else return hipErrorInvalidValue;
  return hipGetLastError();
}

template hipError_t launch_crba_state<double>(const StateModel&, long, const void*, void*, Layout, Layout, int, hipStream_t);
template hipError_t launch_crba_state<float>(const StateModel&, long, const void*, void*, Layout, Layout, int, hipStream_t);
template hipError_t launch_rnea_state<double>(const StateModel&, long, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);
template hipError_t launch_rnea_state<float>(const StateModel&, long, const void*, const void*, const void*, const void*, void*, void*, Layout, Layout, Layout, hipStream_t);

}  // namespace rbd
