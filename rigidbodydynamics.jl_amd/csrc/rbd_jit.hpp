// rbd_jit.hpp — run-time specialisation (rbd_jit.hip): source generator, hiprtc compile with an on-disk cache.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rbd_state_plan.hpp"

namespace rbd {
bool jit_available();  // libhiprtc found and RBD_JIT != 0
// One program per (family, scalar type): SPEC_MASS = mass_matrix! (+ the dense step and the emitter of M: fp32, nv a multiple of 4, nv <= 40), SPEC_ABA = dynamics!,
// SPEC_RNEA = inverse_dynamics! / dynamics_bias!.  Empty string: no such program for this mechanism (spec_has).
enum { SPEC_MASS = 0, SPEC_ABA = 1, SPEC_RNEA = 2, SPEC_FAMILIES = 3 };
// ... and (round 6) SPEC_KIN = the kinematics by-products (kin_spec / jac_spec / mom_spec of rbd_spec.hpp: momentum_matrix!, center_of_mass, energies, geometric_jacobian!,
// momentum, momentum_rate_bias).  Its public family number is 11 (3..10 were taken by the loop, walk, banked and `simulate` programs when it was added); a workspace
// keeps its module in slot spec_slot(SPEC_KIN) = 3 of the per-family arrays.
enum { SPEC_KIN = 11, SPEC_SLOTS = 4 };
inline int spec_slot(int family) { return family == SPEC_KIN ? 3 : family; }
bool spec_has(int family, int dtype, int nb, int nq, int nv, int n3 = 0);  // n3: 3-dof joints (10 more LDS rows each in dynamics!)
bool spec_has_chol(int dtype, int nv);
std::string spec_source(const StatePlan& P, int nb, int nq, int nv, const uint64_t* row_mask, const double* gravity, int dtype, int family);
// The program of ONE small loop mechanism (rbd_loop_small.hpp with the loop tables as compile-time constants): loop_spec_f32 / loop_spec_f64.
struct LoopTables {
  int nb, nq, nv, nc, nloops;
  const std::vector<int32_t>*li, *path, *jt, *voff, *xi;
  const std::vector<double>*lr, *axis, *axis2, *rb;
  const double* gravity;
};
std::string spec_loop_source(const LoopTables& L, int dtype);
// The program of the two-bodies-per-lane kernels (rbd_bank.hpp) for ONE mechanism: aba_bank_spec_<sfx>, aba_bank_fused_spec_<sfx>, rnea_bank_spec_<sfx> — the level loops
// unrolled against the mechanism's level structure (levels, first level of bank 1, later-children counts and hop kind per level).
std::string spec_bank_source(int nlevels, int L0, const int32_t* nslots, unsigned long long perm_down, int simple, int dtype);
// The program of the one-wavefront-per-track dynamics! kernel for ONE mechanism (aba_walk_spec of rbd_walk.hpp with the plan as constants): aba_walk_spec_f64.
// fp64, fp32, and fp32 with two states per lane (pair).
struct WalkTables {
  int ns, G, nA, nB, nS, nq, nv, flt, gen, rr;
  const std::vector<int32_t>*ri, *wk;
  const std::vector<double>* rrc;  // the records' constants [ns * G][TR_STRIDE]
  uint64_t sfm[5];
  int nchain, fq, fv;              // rr: the chain of the re-rooted tree's floating base
  const std::vector<int32_t>* chain_i;
  const std::vector<double>* chain_r;
  const double* fXp;
  const std::vector<int32_t>*mk1, *mkf;  // the joints as rbd_mk_fuse.hpp sees them: (q offset, v offset, type) of the 1-dof ones, (q offset, v offset) of the 6-dof ones
};
bool walk_spec_has(int dtype, int ns, int G, size_t lds_rows_bytes);
size_t walk_spec_lds_bytes(const WalkTables& W, int dtype, int pair = 0);
const char* walk_spec_suffix(int dtype, int pair);  // f64 | f32 | f32x2 (two fp32 states per lane)
std::string walk_spec_source(const WalkTables& W, int dtype, int kind = 0, int pair = 0);  // kind 0: dynamics! (aba_walk_spec_<suffix>), 1: inverse_dynamics! / dynamics_bias! (rnea_walk_spec_<suffix>)
// Code objects.  *_get: JIT_READY with the object (cache hit, or a compilation that has finished), JIT_PENDING when `wait` is false and the compilation has been
// started on a background thread — ask again later —, JIT_FAILED with the compiler's log.  The plain forms wait.  jit_async(): RBD_JIT_ASYNC != 0 (default).
enum { JIT_READY = 0, JIT_PENDING = 1, JIT_FAILED = 2 };
bool jit_async();
void jit_wait_idle();  // joins every background compilation of this process
int jit_code_object_get(const std::string& source, bool wait, std::vector<char>* code, std::string* log);
// the walk kernels' object: checked for accumulation registers of the allocator's own, then its kernel descriptor made to cover all 256 (see rbd_jit.hip)
int jit_walk_code_object_get(const std::string& source, bool wait, std::vector<char>* code, std::string* log);
int jit_kd_cover_agprs(std::vector<char>* code);
// the admission of a walk program's code object (registers against the numbered stash, no scratch, the descriptor rewritten): what jit_walk_code_object_get applies
bool jit_walk_admit(const std::string& source, std::vector<char>* code, std::string* log);
std::vector<char> jit_walk_code_object(const std::string& source, std::string* log);
std::vector<char> jit_code_object(const std::string& source, std::string* log);
void jit_cache_discard(const std::string& source);
}  // namespace rbd
