// rbd_contact_kernels.hip — soft contact on the device: contact_dynamics! (src/mechanism_algorithms.jl:680-723) with the reference's
// default models (src/contact.jl: HuntCrossleyModel :98-119, ViscoelasticCoulombModel :122-178, HalfSpace3D :202-228), and the
// Runge–Kutta bookkeeping of the additional state for `simulate` (src/ode_integrators.jl:233-299).
//
// contact_kernel: one thread per state.  Body transforms and twists come from the per-body kinematics rnea_kernel exports
// ([state][body][24]: R, p, twist, acceleration); the thread walks the contact points in the order of the additional state, tests each
// against every half-space and adds its wrench to its body's; it writes contactwrenches and totalwrenches = wext + contactwrenches
// (mechanism_algorithms.jl:851-856) for EVERY body, so the forward-dynamics launch that follows reads one wrench buffer.
// No contact configuration is on a BASELINE hot path: clarity over speed here.
#include "rbd_device.hpp"
#include "rbd_internal.hpp"

namespace rbd {

template <typename T> RBD_DEV T pow_t(T x, T n);
template <> RBD_DEV double pow_t<double>(double x, double n) { return pow(x, n); }
template <> RBD_DEV float pow_t<float>(float x, float n) { return powf(x, n); }

template <typename T>
__global__ __launch_bounds__(256) void contact_kernel(ContactModel M, long B, const T* __restrict__ body, T* __restrict__ s, T* __restrict__ sdot,
                                                     const T* __restrict__ fext, T* __restrict__ cw, T* __restrict__ tw, Layout Ls, Layout Lf) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const T* cp = reinterpret_cast<const T*>(M.cp);
  const T* hs = reinterpret_cast<const T*>(M.hs);
  // contactwrenches = 0, totalwrenches = wext for every body; the points then add their wrenches (a thread owns its state's columns)
  for (int j = 0; j < 6 * M.nb; ++j) {
    const long o = (long)j * Lf.sk + b * Lf.sb;
    if (cw) cw[o] = T(0);
    if (tw) tw[o] = fext ? fext[o] : T(0);
  }
  T* sx = s + b * Ls.sb;
  T* sd = sdot ? sdot + b * Ls.sb : nullptr;
  for (int ip = 0; ip < M.np; ++ip) {
    const int body_i = M.cbody[ip];
    const T* k = body + (b * M.nb + body_i) * 24;
    const T* c = cp + (long)ip * CP_STRIDE;
    T w[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    // point = body_to_root * location; velocity = point_velocity(twist, point) = ω × point + v   (:697-699)
    T pt[3], vel[3], t3[3];
    matvec3(k, c + CP_LOC, pt);
#pragma unroll
    for (int j = 0; j < 3; ++j) pt[j] += k[9 + j];
    cross3(k + 12, pt, t3);
#pragma unroll
    for (int j = 0; j < 3; ++j) vel[j] = t3[j] + k[15 + j];
    for (int h = 0; h < M.nh; ++h) {
      const T* H = hs + (long)h * 6;
      const long so = (long)(ip * M.nh + h) * 3;
      const T* n = H + 3;
      const T sep = (pt[0] - H[0]) * n[0] + (pt[1] - H[1]) * n[1] + (pt[2] - H[2]) * n[2];  // separation (contact.jl:224)
      if (sep <= T(0)) {  // point_inside (:225)
        // contact_dynamics! (contact.jl:79-93)
        const T z = -sep;
        const T zd = -(vel[0] * n[0] + vel[1] * n[1] + vel[2] * n[2]);
        const T zn = pow_t(z, c[CP_HCN]);
        T fn = c[CP_HCL] * zn * zd + c[CP_HCK] * zn;  // HuntCrossley normal_force (:115-118)
        fn = fn > T(0) ? fn : T(0);
        T vt[3], x[3], fs[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { vt[j] = vel[j] + zd * n[j]; x[j] = sx[(so + j) * Ls.sk]; }
        // friction_force (:150-169): stick force −k x − b v, clipped to the friction cone
#pragma unroll
        for (int j = 0; j < 3; ++j) fs[j] = -c[CP_K] * x[j] - c[CP_B] * vt[j];
        const T n2 = fs[0] * fs[0] + fs[1] * fs[1] + fs[2] * fs[2], m2 = (c[CP_MU] * fn) * (c[CP_MU] * fn);
        if (n2 > m2) {
          const T sc = SqrtT<T>::f(m2 / n2);
#pragma unroll
          for (int j = 0; j < 3; ++j) fs[j] *= sc;
        }
        if (sd) {  // dynamics! of the friction state (:171-178): ẋ = (−k x − f_tangential) / b
#pragma unroll
          for (int j = 0; j < 3; ++j) sd[(so + j) * Ls.sk] = (-c[CP_K] * x[j] - fs[j]) / c[CP_B];
        }
        T f[3], tq[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) f[j] = fn * n[j] + fs[j];
        cross3(pt, f, tq);  // Wrench(point, force) (:712)
#pragma unroll
        for (int j = 0; j < 3; ++j) { w[j] += tq[j]; w[3 + j] += f[j]; }
      } else {  // reset! the state, zero! the derivative (:714-715)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          sx[(so + j) * Ls.sk] = T(0);
          if (sd) sd[(so + j) * Ls.sk] = T(0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const long o = (long)(6 * body_i + j) * Lf.sk + b * Lf.sb;
      if (cw) cw[o] += w[j];
      if (tw) tw[o] += w[j];
    }
  }
}

// the additional state through a Runge–Kutta step, stage by stage like mk_stage_kernel does (q, v):
//   stage 0: s0 = s;  stage k = 1..3: acc += b_k ṡ_k, s = s0 + a_{k+1,k} dt ṡ_k;  stage 4: s = s0 + dt (acc + b_4 ṡ_4)
// with the runge_kutta_4 tableau (ode_integrators.jl:48-55): a = 1/2, 1/2, 1; b = 1/6, 1/3, 1/3, 1/6.
template <typename T>
__global__ __launch_bounds__(256) void contact_stage_kernel(long n, int stage, T dt, T* __restrict__ s, const T* __restrict__ sdot, T* __restrict__ s0,
                                                           T* __restrict__ acc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (stage == 0) { s0[i] = s[i]; acc[i] = T(0); return; }
  const T bk = (stage == 1 || stage == 4) ? T(1) / T(6) : T(1) / T(3);
  const T a = acc[i] + bk * sdot[i];
  if (stage == 4) { s[i] = s0[i] + dt * a; return; }
  acc[i] = a;
  s[i] = s0[i] + (stage == 3 ? T(1) : T(0.5)) * dt * sdot[i];
}

template <typename T>
hipError_t launch_contact(const ContactModel& M, long B, const void* body, void* s, void* sdot, const void* fext, void* cw, void* tw, Layout Ls, Layout Lf,
                          hipStream_t st) {
  contact_kernel<T><<<(unsigned)((B + 255) / 256), 256, 0, st>>>(M, B, (const T*)body, (T*)s, (T*)sdot, (const T*)fext, (T*)cw, (T*)tw, Ls, Lf);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_contact_stage(long n, int stage, double dt, void* s, const void* sdot, void* s0, void* acc, hipStream_t st) {
  contact_stage_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(n, stage, (T)dt, (T*)s, (const T*)sdot, (T*)s0, (T*)acc);
  return hipGetLastError();
}
template hipError_t launch_contact<double>(const ContactModel&, long, const void*, void*, void*, const void*, void*, void*, Layout, Layout, hipStream_t);
template hipError_t launch_contact<float>(const ContactModel&, long, const void*, void*, void*, const void*, void*, void*, Layout, Layout, hipStream_t);
template hipError_t launch_contact_stage<double>(long, int, double, void*, const void*, void*, void*, hipStream_t);
template hipError_t launch_contact_stage<float>(long, int, double, void*, const void*, void*, void*, hipStream_t);

}  // namespace rbd
