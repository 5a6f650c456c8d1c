// rbd_capi.hip — the C ABI of librbd_hip.so (include/rbd_hip.h): model flattening for the device,
// workspace/stream management, argument checking, kernel dispatch.  No torch types, no CPU fallback:
// every hot-path entry point launches HIP kernels or fails with a status code.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rbd_hip.h"
#include "rbd_internal.hpp"
#include "rbd_chain_plan.hpp"
#include "rbd_track_plan.hpp"
#include "rbd_walk_plan.hpp"
#include "rbd_reroot.hpp"
#include "rbd_state_plan.hpp"
#include "rbd_jit.hpp"
#include "rbd_mk_fuse.hpp"
enum { BANK_LDS_PAIRS_HOST = 30 };  // = BANK_LDS_PAIRS of rbd_bank.hpp (16 parked + 14 exchange pairs per lane; checked in rbd_bank_kernels.hip)

using namespace rbd;

static thread_local std::string g_last_hip_error;

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      g_last_hip_error = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
      return (e_ == hipErrorOutOfMemory) ? RBD_ERR_OUT_OF_MEMORY : RBD_ERR_HIP;                    \
    }                                                                                              \
  } while (0)

static int joint_nq_host(int t) {
  switch (t) {
    case RBD_JOINT_FIXED: return 0;
    case RBD_JOINT_REVOLUTE: case RBD_JOINT_PRISMATIC: return 1;
    case RBD_JOINT_QUAT_FLOATING: return 7;
    case RBD_JOINT_PLANAR: return 3;
    case RBD_JOINT_QUAT_SPHERICAL: return 4;
    case RBD_JOINT_SINCOS_REVOLUTE: return 2;
    default: return -1;
  }
}
static int joint_nv_host(int t) {
  switch (t) {
    case RBD_JOINT_FIXED: return 0;
    case RBD_JOINT_REVOLUTE: case RBD_JOINT_PRISMATIC: case RBD_JOINT_SINCOS_REVOLUTE: return 1;
    case RBD_JOINT_QUAT_FLOATING: return 6;
    case RBD_JOINT_PLANAR: case RBD_JOINT_QUAT_SPHERICAL: return 3;
    default: return -1;
  }
}

struct rbd_model {
  int32_t nb = 0, nq = 0, nv = 0, nc = 0, nloops = 0;
  int32_t lps = 1, nlevels = 0, maxchild = 0, maxnvj = 0;
  double gravity[3] = {0, 0, 0};
  std::vector<int32_t> ib;      // nb * IB_STRIDE
  std::vector<double> rb;       // nb * RB_STRIDE
  std::vector<int32_t> nslots;  // nlevels
  uint64_t perm_down = 0;
  int32_t inner_floating = 0, has3dof = 0;
  std::vector<int32_t> slot_of, order;  // reference body index <-> DFS pre-order slot
  std::vector<int32_t> dof_body;
  std::vector<int32_t> anc;     // nb * nlevels
  std::vector<uint64_t> row_mask;  // nv x row_words
  int32_t row_words = 1;
  std::vector<rbd_loop_joint_t> loops;
  std::vector<int32_t> loop_i, loop_path, jt_ref, voff_ref, parent_ref, qoff_ref;  // loop tables (reference body indices)
  std::vector<int32_t> mk1, mkf;  // the joints as the integrator stage folded into the compiled dynamics! kernels sees them (rbd_mk_fuse.hpp)
  bool loop_fused_ok = false;
  bool big = false;  // more than 64 bodies: only the any-size kernels of rbd_big_kernels.hip apply (reference-order tables below)
  std::vector<int32_t> big_tbl;
  std::vector<double> big_rb;  // small enough, and only 1-dof / fixed tree joints with parents before children: loop_fused_small_kernel
  std::vector<double> loop_r, axis_ref, axis2_ref;
  // banked lane-per-body mapping (aba_bank_kernel): two bodies per lane, split at level bank_L0; bank_lps == 0: not applicable
  int32_t bank_lps = 0, bank_L0 = 0, bank_nb[2] = {0, 0}, bank_aba_ok = 0;
  std::vector<int32_t> bank_ib[2];
  std::vector<double> bank_rb[2];
  uint64_t bank_perm_down = 0;
  ChainPlan chain;  // the chains of the tree packed on G tracks by list scheduling: what the track / walk plans are built on (rbd_model_chain_plan exposes it)
  TrackPlan track;  // the track schedule of the walk kernels (track.ok == false: mechanism outside their scope)
  // the tree re-rooted at its centre (rbd_reroot.hpp): its own slots, banks and track / walk plans; used by the ABA kernels that support it
  Reroot rr;
  struct RrSlots {
    bool ok = false;
    int32_t nlevels = 0;
    std::vector<int32_t> ib, nslots;
    std::vector<double> rb;
    TrackPlan track;
    WalkPlan walk;
  } rrs;
  // soft contact (src/contact.jl): points in the order of the additional state, half-spaces with unit normals
  int32_t ncp = 0, nhs = 0;
  std::vector<int32_t> cp_body;
  std::vector<double> cp_r, hs_r;  // ncp * CP_STRIDE, nhs * 6
  WalkPlan walk;    // parking slots of aba_walk_kernel on top of the track plan (walk.ok == false: track plan missing or too many steps)
  StatePlan state;  // plan of the one-lane-per-state kernels (state.ok == false: mechanism outside their scope)
  StatePlan state_wide;  // ... of the ones compiled for the mechanism when it has 3-dof joints / 6-dof joints below the world (state.ok == false, state_wide.ok)
  const StatePlan& spec_plan() const { return state.ok ? state : state_wide; }
};

struct rbd_ws {
  const rbd_model* model = nullptr;
  int32_t device = 0, dtype = RBD_F64, max_batch = 0;
  hipStream_t stream = nullptr;
  DevModel dm{};
  BankModel bm{}; void* d_bank_ib[2] = {nullptr, nullptr}; void* d_bank_rb[2] = {nullptr, nullptr};
  // the re-rooted tree (rbd_reroot.hpp): banked records, chain table, walk plan
  void* d_rr_chain_i = nullptr; void* d_rr_chain_r = nullptr;
  WalkModel wm_rr{}; bool walk_rr = false; void* d_rrtrack_ri = nullptr; void* d_rrtrack_rr = nullptr; void* d_rrwalk_wk = nullptr; size_t walk_rr_lds_bytes = 0, walk_rr_lds_bytes_pair = 0;
  TrackModel tm{}; void* d_track_ri = nullptr; void* d_track_rr = nullptr;  // the track plan's records: what the walk kernels read
  ContactModel ctm{}; void* d_cp_body = nullptr; void* d_cp_r = nullptr; void* d_hs_r = nullptr;  // soft contact tables
  void* d_tw = nullptr; void* d_cw = nullptr; void* d_s0 = nullptr; void* d_sacc = nullptr; void* d_sdot = nullptr; void* d_rows = nullptr; size_t d_rows_bytes = 0, d_tw_bytes = 0, d_cw_bytes = 0, d_s0_bytes = 0, d_sacc_bytes = 0, d_sdot_bytes = 0;
  WalkModel wm{}; void* d_walk_wk = nullptr; size_t walk_lds_bytes = 0, walk_lds_bytes_pair = 0; long walk_min_batch = 0, walk_pair_min_batch = 0, sim_walk_min_batch = 1;
  // run-time specialised kernels (rbd_jit.hip), built on the first use of a route that has them; null: not available
  bool spec_tried[SPEC_SLOTS] = {false, false, false, false}; hipModule_t spec_mod[SPEC_SLOTS] = {nullptr, nullptr, nullptr, nullptr};  // (by spec_slot(family))
  hipFunction_t spec_kin = nullptr, spec_jac = nullptr, spec_mom = nullptr, spec_energy = nullptr, spec_com = nullptr; long spec_kin_min_batch = (long)1 << 62;  // the kinematics by-products compiled for the mechanism (SPEC_KIN, round 6)
  hipFunction_t spec_crba = nullptr, spec_crba_perm = nullptr, spec_chol = nullptr, spec_chol_nom = nullptr, spec_chol_packed = nullptr, spec_emit = nullptr, spec_aba = nullptr, spec_aba_nofext = nullptr, spec_aba_gst = nullptr, spec_aba_gst_nofext = nullptr, spec_rnea = nullptr, spec_loop = nullptr;
  int spec_f64_max_scratch = 0;  // (RBD_TUNE spec_f64_max_scratch; set from the measurement in workspace_create)
  // fp64 dynamics! of those mechanisms: the program with its spare rows in the HBM stash (two wavefronts per CU, a longer chain) against the one with every row in
  // LDS (one per CU): RBD_TUNE spec_f64_stash = 1 always / 0 never / -1 whichever needs fewer chain-times for the batch; the chains' ratio in percent
  int spec_f64_stash = -1, spec_f64_stash_ratio = 170, spec_f64_stash_ratio_fext = 120, spec_ncu = 256;
  // first use of a run-time compiled dynamics! program by this workspace: its result on the first states of the call against the interpreting kernel's
  // (first_use_check; RBD_TUNE first_use_check=0 for timing experiments with programs that are wrong by construction).  [stash program][no wrenches]
  bool spec_first_use_check = true, spec_first_use_inject = false, spec_aba_checked[4] = {false, false, false, false}, spec_walk_checked[12] = {}, spec_bank_checked = false, spec_rnea_checked = false, spec_bank_rnea_checked = false, spec_mass_checked[3] = {false, false, false};  // (mass: [M emitted | M_out = NULL | packed triangle])
  double spec_check_err = 0;  // (what the last check measured: max |difference| / max(1, max |reference|))
  int spec_aba_scratch = 0, spec_aba_nofext_scratch = 0, spec_rnea_scratch = 0;  // bytes per lane spilled by those kernels: only a kernel without any is picked on its own (it runs 3.4 times slower with: the dispatcher admits fewer wavefronts)
  bool spec_loop_tried = false; hipModule_t spec_loop_mod = nullptr;
  bool spec_bank_tried = false; hipModule_t spec_bank_mod = nullptr; hipFunction_t spec_bank_aba = nullptr, spec_bank_fused = nullptr, spec_bank_rnea = nullptr; std::string spec_bank_src;  // the banked kernels compiled for the mechanism
  std::string spec_src[SPEC_SLOTS], spec_loop_src, spec_walk_src[12];  // the programs' sources while their compilation is pending (generated once)
  bool spec_walk_tried[12] = {}; hipModule_t spec_walk_mod[12] = {}; hipFunction_t spec_walk[12] = {};  // [dynamics! | inverse dynamics | dynamics!, four `simulate` stages per launch][re-rooted tree][two fp32 states per lane]
  bool no_reroot = false, loop_no_fused = false; int spec_max_scratch = 512;  // RBD_TUNE: walk_no_reroot, loop_no_fused (tests: the original tree / the three-launch loop route), spec_max_scratch (spilled bytes per lane above which a compiled kernel steps aside)
  bool spec_walk_f32 = true;  // fp32 batches through the compiled walk kernels too (RBD_SPEC_WALK_F32=0: not)
  std::vector<double> loop_gains; bool custom_gains = false;  // rbd_workspace_set_loop_gains: this workspace's Baumgarte gains (4 per loop joint), and whether they differ from the model's
  void* bound_M = nullptr; void* bound_c = nullptr;  // rbd_workspace_bind_result: the caller's own M / c buffers for the CRBA route of rbd_dynamics
  long spec_aba_min_batch = 0, spec_rnea_min_batch = 0, spec_walk_min_batch = 0, walk_one_round_batch = 0, rnea_walk_min_batch = 0;
  StateModel sm{}; void* d_state_ops = nullptr; void* d_state_cols = nullptr; void* d_state_sr = nullptr; long state_min_batch = 0; long mass_min_batch = (long)1 << 62, mass_solve_min_batch = (long)1 << 62; long spec_aba_fused_min_batch = (long)1 << 62; long sim_walk_max_batch = 0; bool state_aot = false;  // state_aot: the interpreting one-lane-per-state kernels take the mechanism
  void* d_Msoa = nullptr; size_t d_Msoa_bytes = 0; long Msoa_B = -1; int Msoa_perm = -1;  // batch-innermost staging of M for the one-lane-per-state CRBA when the caller's layout is AOS
  long bank_min_batch = 0, rnea_bank_min_batch = 0, bank_resident_states = 0;
  void* d_ib = nullptr; void* d_rb = nullptr; void* d_nslots = nullptr; void* d_dof_body = nullptr; void* d_anc = nullptr; void* d_row_mask = nullptr;
  // staging for RBD_MEM_HOST (lazy)
  void* stage[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // internal device scratch (mass matrix / bias for the CRBA route), lazy
  void* d_M = nullptr; void* d_c = nullptr; void* d_K = nullptr; void* d_k = nullptr;
  size_t d_M_bytes = 0, d_c_bytes = 0, d_K_bytes = 0, d_k_bytes = 0;
  void* d_body = nullptr; void* d_scratch = nullptr; size_t d_body_bytes = 0, d_scratch_bytes = 0;
  BigModel big{}; void* d_big_tbl = nullptr; void* d_big_rb = nullptr; void* d_big_scratch = nullptr; size_t d_big_scratch_bytes = 0; void* d_big_L = nullptr; size_t d_big_L_bytes = 0;  // rbd_big_kernels.hip (d_big_L: the Cholesky factor, result.L)
  void* d_fused_i = nullptr;  // loop_fused_small_kernel: parent, q offset, slot by reference body index
  void* d_loop_i = nullptr; void* d_loop_r = nullptr; void* d_loop_path = nullptr; void* d_jt_ref = nullptr; void* d_voff_ref = nullptr; void* d_axis_ref = nullptr; void* d_axis2_ref = nullptr;
  MkBuffers mk{}; void* d_vdwork = nullptr; size_t mk_elems = 0;  // Munthe-Kaas integrator scratch (lazy)
  void* d_tauwork = nullptr; size_t d_tauwork_bytes = 0;  // torques of the device-side PD controller (un-fused integrator path)
  int* d_notpd = nullptr;  // device flag: some state's mass matrix was not positive definite (checked by rbd_sync)
  int32_t result_layout = RBD_LAYOUT_SOA; int32_t result_B = 0;
  // timing
  int32_t timing = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_pending = false;
  const char* last_kernel = "";  // dominant kernel of the last rbd_dynamics / rbd_simulate / rbd_mass_matrix_solve call
};

// RBD_TUNE="key=value,key=value,...": the developer knobs of the tests and sweep scripts in ONE environment variable (batch thresholds between the lane
// mappings, routes forced off).  Read when a model / workspace is created; `has`: the key was given.  Users need none of them.
static long tune(const char* key, long dflt, bool* has = nullptr) {
  if (has) *has = false;
  const char* e = getenv("RBD_TUNE");
  if (!e) return dflt;
  const size_t n = strlen(key);
  for (const char* p = e; *p;) {
    const char* end = strchr(p, ',');
    const size_t len = end ? (size_t)(end - p) : strlen(p);
    if (len > n && memcmp(p, key, n) == 0 && p[n] == '=') { if (has) *has = true; return atol(p + n + 1); }
    if (len == n && memcmp(p, key, n) == 0) { if (has) *has = true; return 1; }
    if (!end) break;
    p = end + 1;
  }
  return dflt;
}

static std::string loop_program_source(const rbd_model* m, int dtype, std::vector<int32_t>* xi_store);  // (below)
static int bank_simple(const rbd_model* m);
static std::string bank_program_source(const rbd_model* m, int dtype, int simple);
static std::string walk_program_source(const rbd_model* m, int dtype, bool rerooted, int kind = 0, int pair = 0);
static bool walk_program_rerooted(const rbd_model* m, int dtype, int pair = 0);

extern "C" {

int rbd_version(void) { return RBD_HIP_H_VERSION; }
// run-time specialisation (rbd_jit.hip): the generated source of a model's kernels, and its compilation into the on-disk cache.  Neither needs a device.
// the source of one program of a model (families as in include/rbd_hip.h); empty: no such program for this mechanism
static std::string program_source(const rbd_model* m, int32_t dtype, int32_t family) {
  if (!m || (dtype != RBD_F32 && dtype != RBD_F64) || family < 0 || family > SPEC_KIN) return std::string();
  if (family == SPEC_KIN) return m->spec_plan().ok ? spec_source(m->spec_plan(), m->nb, m->nq, m->nv, m->row_mask.data(), m->gravity, dtype, SPEC_KIN) : std::string();  // (family 11: the kinematics by-products)
  if (family == SPEC_FAMILIES + 6) return walk_program_source(m, dtype, walk_program_rerooted(m, dtype), 2);  // (family 9: family 4 with the four stages of a `simulate` step in one launch)
  if (family == SPEC_FAMILIES + 7) return dtype == RBD_F32 ? walk_program_source(m, dtype, walk_program_rerooted(m, dtype, 1), 2, 1) : std::string();  // (family 10: family 6 likewise)
  std::vector<int32_t> xi;
  if (family < SPEC_FAMILIES && !m->spec_plan().ok) return std::string();
  if (family == SPEC_FAMILIES + 5) return bank_program_source(m, dtype, bank_simple(m));  // (family 8: the two-bodies-per-lane kernels)
  return family == SPEC_FAMILIES + 4 ? (dtype == RBD_F32 ? walk_program_source(m, dtype, false, 1, 1) : std::string())  // (families 6, 7: 4 and 5 with two fp32 states per lane)
         : family == SPEC_FAMILIES + 3 ? (dtype == RBD_F32 ? walk_program_source(m, dtype, walk_program_rerooted(m, dtype, 1), 0, 1) : std::string())
         : family == SPEC_FAMILIES + 2 ? walk_program_source(m, dtype, false, 1)  // (family 5: ... and its inverse_dynamics! kernel, on the original tree)
         : family == SPEC_FAMILIES + 1 ? walk_program_source(m, dtype, walk_program_rerooted(m, dtype))  // (family 4: the one-wavefront-per-track dynamics! kernel)
         : family == SPEC_FAMILIES ? loop_program_source(m, dtype, &xi)  // (family 3: the program of a small loop mechanism)
                                   : spec_source(m->spec_plan(), m->nb, m->nq, m->nv, m->row_mask.data(), m->gravity, dtype, family);
}
static bool family_is_walk(int family) { return (family > SPEC_FAMILIES && family <= SPEC_FAMILIES + 4) || family == SPEC_FAMILIES + 6 || family == SPEC_FAMILIES + 7; }
int64_t rbd_jit_source(const rbd_model_t* m, int32_t dtype, int32_t family, char* buf, int64_t cap) {
  const std::string s = program_source(m, dtype, family);
  if (s.empty()) return -1;
  if (buf && cap > 0) { const int64_t n = std::min<int64_t>(cap - 1, (int64_t)s.size()); memcpy(buf, s.data(), (size_t)n); buf[n] = 0; }
  return (int64_t)s.size();
}
// 1: the program's code object is ready (in the cache, or compiled by this process); 0: being compiled on a background thread (started by this call if nobody
// had); -1: no such program for this mechanism, no hiprtc, or the compilation failed.  Never waits.
void rbd_jit_wait_idle(void) { jit_wait_idle(); }
int rbd_jit_check_walk_object(const char* source, const void* code, int64_t size, char* log, int64_t cap) {
  if (log && cap > 0) log[0] = 0;
  if (!source || !code || size <= 0) return RBD_ERR_INVALID_ARGUMENT;
  std::vector<char> c((const char*)code, (const char*)code + size);
  std::string l;
  const bool ok = jit_walk_admit(source, &c, &l);
  if (log && cap > 0) { const int64_t n = std::min<int64_t>(cap - 1, (int64_t)l.size()); memcpy(log, l.data(), (size_t)n); log[n] = 0; }
  return ok ? RBD_OK : RBD_ERR_UNSUPPORTED;
}
int rbd_jit_status(const rbd_model_t* m, int32_t dtype, int32_t family) {
  const std::string src = program_source(m, dtype, family);
  if (src.empty() || !jit_available()) return -1;
  std::vector<char> code;
  std::string log;
  const int js = family_is_walk(family) ? jit_walk_code_object_get(src, false, &code, &log) : jit_code_object_get(src, false, &code, &log);
  return js == JIT_READY ? 1 : js == JIT_PENDING ? 0 : -1;
}
int rbd_jit_precompile(const rbd_model_t* m, int32_t dtype, char* log, int64_t cap) {
  if (log && cap > 0) log[0] = 0;
  if (!m || (dtype != RBD_F32 && dtype != RBD_F64)) return RBD_ERR_INVALID_ARGUMENT;
  if ((!m->spec_plan().ok && !m->loop_fused_ok && !(m->track.ok && m->walk.ok) && m->bank_lps <= 0) || !jit_available()) return RBD_ERR_UNSUPPORTED;
  // the model's programs of this scalar type, the longest compilations first; every one of them on its own background thread (rbd_jit.hip), then wait for all
  struct Job { std::string src; bool walk; int family; int state; double seconds; std::string log; };
  std::vector<Job> jobs;
  for (int family : {4, 9, 5, 6, 10, 7, 8, 0, 1, 2, 11, 3})
    jobs.push_back({program_source(m, dtype, family), family_is_walk(family), family, JIT_PENDING, 0.0, std::string()});
  // RBD_JIT_PRECOMPILE_PART = "k/n": only every n-th program, starting with the k-th — n processes share a model's compilations
  int part = 0, parts = 1;
  if (const char* e = getenv("RBD_JIT_PRECOMPILE_PART")) { if (sscanf(e, "%d/%d", &part, &parts) != 2 || parts < 1 || part < 0 || part >= parts) { part = 0; parts = 1; } }  // (build tool only)
  int idx = 0;
  for (Job& j : jobs) {
    if (j.src.empty()) { j.state = -1; continue; }
    if (idx++ % parts != part) j.state = -1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  int st = RBD_OK;
  for (bool pending = true; pending;) {
    pending = false;
    for (Job& j : jobs) {
      if (j.state != JIT_PENDING) continue;
      std::vector<char> code;
      j.state = j.walk ? jit_walk_code_object_get(j.src, false, &code, &j.log) : jit_code_object_get(j.src, false, &code, &j.log);
      if (j.state == JIT_PENDING) pending = true;
      else j.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (pending) usleep(20000);
  }
  std::string all;
  for (const Job& j : jobs) {
    if (j.state < 0) continue;
    char b[160];
    // (a walk program whose registers do not work out is not an error: the interpreting kernel stays)
    snprintf(b, sizeof b, "[rbd_jit] family %d (%s): %s after %.1f s\n", j.family, dtype == RBD_F64 ? "f64" : "f32", j.state == JIT_READY ? "ready" : j.walk ? "not used" : "FAILED", j.seconds);
    all += b;
    all += j.log;
    if (j.state == JIT_FAILED && !j.walk) st = RBD_ERR_HIP;
  }
  if (log && cap > 0) { const int64_t n = std::min<int64_t>(cap - 1, (int64_t)all.size()); memcpy(log, all.data(), (size_t)n); log[n] = 0; }
  return st;
}

const char* rbd_status_string(int s) {
  switch (s) {
    case RBD_OK: return "ok";
    case RBD_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RBD_ERR_DIMENSION_MISMATCH: return "dimension mismatch";
    case RBD_ERR_UNSUPPORTED: return "unsupported joint type or feature";
    case RBD_ERR_NO_DEVICE: return "no HIP device available (librbd_hip has no CPU fallback)";
    case RBD_ERR_HIP: return "HIP runtime error";
    case RBD_ERR_OUT_OF_MEMORY: return "out of memory";
    case RBD_ERR_HAS_LOOPS: return "This method can currently only handle tree Mechanisms.";
    case RBD_ERR_NOT_POSITIVE_DEFINITE: return "mass matrix not positive definite";
    default: return "unknown status";
  }
}
const char* rbd_last_hip_error(void) { return g_last_hip_error.c_str(); }

// the loop (non-tree) joints' tables, in reference body indices: records, local constraint wrench bases, tree paths (constraint_jacobian_structure,
// src/mechanism_state.jl:51, :99-100).  Independent of the lane mappings: also what trees of more than 64 bodies use.
static int build_loop_tables(rbd_model* m, const rbd_flat_model_t* d) {
  const int nb = m->nb;
  for (int l = 0; l < d->n_loops; ++l) {
    const rbd_loop_joint_t& lj = d->loops[l];
    const int nvl = joint_nv_host(lj.joint_type);
    if (nvl < 0) return RBD_ERR_INVALID_ARGUMENT;
    if (lj.predecessor >= nb || lj.successor >= nb || lj.predecessor < -1 || lj.successor < -1) return RBD_ERR_INVALID_ARGUMENT;
    const int ncl = 6 - nvl;  // num_constraints: src/joint.jl:12
    // local constraint wrench basis in frame_after(joint): revolute.jl:91-98, prismatic.jl:101-108, fixed.jl
    double Tl[36] = {0};
    const double* R = lj.rotation_from_z_aligned;
    if (lj.joint_type == RBD_JOINT_REVOLUTE || lj.joint_type == RBD_JOINT_SINCOS_REVOLUTE) {
      for (int r = 0; r < 3; ++r) { Tl[0 + r] = R[3 * r]; Tl[6 + r] = R[3 * r + 1]; Tl[12 + 3 + r] = R[3 * r]; Tl[18 + 3 + r] = R[3 * r + 1]; Tl[24 + 3 + r] = R[3 * r + 2]; }
    } else if (lj.joint_type == RBD_JOINT_PRISMATIC) {
      for (int r = 0; r < 3; ++r) { Tl[0 + r] = R[3 * r]; Tl[6 + r] = R[3 * r + 1]; Tl[12 + r] = R[3 * r + 2]; Tl[18 + 3 + r] = R[3 * r]; Tl[24 + 3 + r] = R[3 * r + 1]; }
    } else if (lj.joint_type == RBD_JOINT_FIXED) {
      for (int c = 0; c < 6; ++c) Tl[6 * c + c] = 1;
    } else if (lj.joint_type == RBD_JOINT_QUAT_SPHERICAL) {  // quaternion_spherical.jl:50-55: angular 0, linear identity
      for (int c = 0; c < 3; ++c) Tl[6 * c + 3 + c] = 1;
    } else if (lj.joint_type == RBD_JOINT_PLANAR) {  // planar.jl:96-101: (0; rot_axis), (x_axis; 0), (y_axis; 0); R columns = (x, y, x × y)
      for (int r = 0; r < 3; ++r) { Tl[0 + 3 + r] = R[3 * r + 2]; Tl[6 + r] = R[3 * r]; Tl[12 + r] = R[3 * r + 1]; }
    } else if (lj.joint_type != RBD_JOINT_QUAT_FLOATING) {
      return RBD_ERR_UNSUPPORTED;
    }
    const int path_begin = (int)m->loop_path.size() / 2;
    int a = lj.predecessor, b = lj.successor;  // TreePath(pred, succ): src/graphs/tree_path.jl:41-63
    while (a != b) {
      if (a > b) { m->loop_path.push_back(a); m->loop_path.push_back(-1); a = d->parent[a]; }
      else { m->loop_path.push_back(b); m->loop_path.push_back(1); b = d->parent[b]; }
    }
    const int path_end = (int)m->loop_path.size() / 2;
    const int32_t rec[8] = {lj.predecessor, lj.successor, lj.joint_type, m->nc, ncl, path_begin, path_end, 0};
    m->loop_i.insert(m->loop_i.end(), rec, rec + 8);
    double r64[64] = {0};
    for (int k = 0; k < 9; ++k) { r64[k] = lj.pred_rot[k]; r64[12 + k] = lj.succ_rot[k]; }
    for (int k = 0; k < 3; ++k) { r64[9 + k] = lj.pred_trans[k]; r64[21 + k] = lj.succ_trans[k]; }
    for (int k = 0; k < 4; ++k) r64[24 + k] = lj.gains[k];
    for (int k = 0; k < 36; ++k) r64[28 + k] = Tl[k];
    m->loop_r.insert(m->loop_r.end(), r64, r64 + 64);
    m->nc += ncl;
    m->loops.push_back(lj);
  }
  return RBD_OK;
}

int rbd_model_create(const rbd_flat_model_t* d, rbd_model_t** out) {
  if (!d || !out) return RBD_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (d->n_bodies < 1 || !d->parent || !d->joint_type || !d->q_offset || !d->v_offset || !d->joint_axis || !d->pred_rot ||
      !d->pred_trans || !d->inertia_moment || !d->inertia_cross || !d->inertia_mass)
    return RBD_ERR_INVALID_ARGUMENT;
  if (d->n_loops < 0 || (d->n_loops > 0 && !d->loops)) return RBD_ERR_INVALID_ARGUMENT;
  rbd_model* m = new (std::nothrow) rbd_model();
  if (!m) return RBD_ERR_OUT_OF_MEMORY;
  const int nb = d->n_bodies;
  m->nb = nb; m->nq = d->nq; m->nv = d->nv; m->nloops = d->n_loops;
  memcpy(m->gravity, d->gravity, sizeof m->gravity);
  if (d->n_contact_points < 0 || d->n_halfspaces < 0 || (d->n_contact_points > 0 && !d->contact_points) || (d->n_halfspaces > 0 && !d->halfspaces)) {
    delete m;
    return RBD_ERR_INVALID_ARGUMENT;
  }
  m->ncp = d->n_contact_points; m->nhs = d->n_halfspaces;
  for (int i = 0; i < m->ncp; ++i) {
    const rbd_contact_point_t& c = d->contact_points[i];
    if (c.body < 0 || c.body >= nb || !(c.b != 0.0)) { delete m; return RBD_ERR_INVALID_ARGUMENT; }
    m->cp_body.push_back(c.body);
    const double r[CP_STRIDE] = {c.location[0], c.location[1], c.location[2], c.hc_k, c.hc_lambda, c.hc_n, c.mu, c.k, c.b};
    m->cp_r.insert(m->cp_r.end(), r, r + CP_STRIDE);
  }
  for (int i = 0; i < m->nhs; ++i) {  // HalfSpace3D's constructor normalizes the outward normal (src/contact.jl:208-211)
    const rbd_halfspace_t& h = d->halfspaces[i];
    const double nn = std::sqrt(h.outward_normal[0] * h.outward_normal[0] + h.outward_normal[1] * h.outward_normal[1] + h.outward_normal[2] * h.outward_normal[2]);
    if (!(nn > 0.0)) { delete m; return RBD_ERR_INVALID_ARGUMENT; }
    const double r[6] = {h.point[0], h.point[1], h.point[2], h.outward_normal[0] / nn, h.outward_normal[1] / nn, h.outward_normal[2] / nn};
    m->hs_r.insert(m->hs_r.end(), r, r + 6);
  }
  // ... or a body with more children than a lane-per-body record lists (IB_MAXCHILD = 8; round 6: such a tree takes the any-size route instead of being refused —
  // the reference has no limit: rand_tree_mechanism attaches any number of joints to one body)
  bool many_children = false;
  {
    std::vector<int> nkids(nb, 0);
    for (int i = 0; i < nb; ++i) {
      const int p = d->parent[i];
      if (p >= 0 && p < i && ++nkids[p] > IB_MAXCHILD) many_children = true;
    }
  }
  if (nb > 64 || many_children) {
    // More bodies than a wavefront has lanes: the any-size fallback (rbd_big_kernels.hip) — tree mechanisms without contact points; dynamics!,
    // inverse_dynamics!, dynamics_bias!, mass_matrix!, mass_matrix_solve.  Everything else returns RBD_ERR_UNSUPPORTED for such a model.
    // (round 4: loop joints and contact points too — their tables are in reference body indices, and the loop branch's and the contact kernels read the
    // per-body kinematics from memory)
    int qs = 0, vs = 0;
    m->big_tbl.resize(4 * (size_t)nb);
    m->big_rb.assign((size_t)nb * RB_STRIDE, 0.0);
    for (int i = 0; i < nb; ++i) {
      const int t = d->joint_type[i], nqi = joint_nq_host(t), nvi = joint_nv_host(t);
      if (nqi < 0 || (t == RBD_JOINT_PLANAR && !d->joint_axis2)) { delete m; return RBD_ERR_INVALID_ARGUMENT; }
      if (d->q_offset[i] != qs || d->v_offset[i] != vs) { delete m; return RBD_ERR_DIMENSION_MISMATCH; }
      if (d->parent[i] >= i || d->parent[i] < -1) { delete m; return RBD_ERR_INVALID_ARGUMENT; }
      qs += nqi; vs += nvi;
      int32_t* tb = &m->big_tbl[4 * (size_t)i];
      tb[0] = d->parent[i]; tb[1] = t; tb[2] = d->q_offset[i]; tb[3] = d->v_offset[i];
      double* rb = &m->big_rb[(size_t)i * RB_STRIDE];
      for (int k = 0; k < 3; ++k) rb[RB_AXIS + k] = d->joint_axis[3 * i + k];
      if (d->joint_axis2) for (int k = 0; k < 3; ++k) rb[RB_AXIS2 + k] = d->joint_axis2[3 * i + k];
      for (int k = 0; k < 9; ++k) rb[RB_XPR + k] = d->pred_rot[9 * i + k];
      for (int k = 0; k < 3; ++k) rb[RB_XPP + k] = d->pred_trans[3 * i + k];
      const double* J = d->inertia_moment + 9 * i;
      rb[RB_J + 0] = J[0]; rb[RB_J + 1] = J[1]; rb[RB_J + 2] = J[2]; rb[RB_J + 3] = J[4]; rb[RB_J + 4] = J[5]; rb[RB_J + 5] = J[8];
      for (int k = 0; k < 3; ++k) rb[RB_MC + k] = d->inertia_cross[3 * i + k];
      rb[RB_M] = d->inertia_mass[i];
    }
    if (qs != d->nq || vs != d->nv) { delete m; return RBD_ERR_DIMENSION_MISMATCH; }
    m->big = true;
    m->nc = 0;
    m->jt_ref.assign(d->joint_type, d->joint_type + nb);
    m->parent_ref.assign(d->parent, d->parent + nb);
    m->voff_ref.assign(d->v_offset, d->v_offset + nb);
    m->qoff_ref.assign(d->q_offset, d->q_offset + nb);
    m->axis_ref.assign(d->joint_axis, d->joint_axis + 3 * nb);
    if (d->joint_axis2) m->axis2_ref.assign(d->joint_axis2, d->joint_axis2 + 3 * nb); else m->axis2_ref.assign(3 * nb, 0.0);
    if (int lst = build_loop_tables(m, d)) { delete m; return lst; }
    *out = m;
    return RBD_OK;
  }
  int lps = 1;
  while (lps < nb) lps <<= 1;
  m->lps = lps;
  // consistency of the q / v ranges with the joint types (qranges/vranges: src/mechanism_state.jl:47-48)
  int qsum = 0, vsum = 0;
  std::vector<std::vector<int>> kids(nb);
  std::vector<int> roots;
  for (int i = 0; i < nb; ++i) {
    const int t = d->joint_type[i];
    const int nqi = joint_nq_host(t), nvi = joint_nv_host(t);
    if (nqi < 0) { delete m; return RBD_ERR_INVALID_ARGUMENT; }
    if (t == RBD_JOINT_PLANAR || t == RBD_JOINT_QUAT_SPHERICAL) m->has3dof = 1;
    if (t == RBD_JOINT_PLANAR && !d->joint_axis2) { delete m; return RBD_ERR_INVALID_ARGUMENT; }
    if (d->q_offset[i] != qsum || d->v_offset[i] != vsum) { delete m; return RBD_ERR_DIMENSION_MISMATCH; }
    qsum += nqi; vsum += nvi;
    if (nvi > m->maxnvj) m->maxnvj = nvi;
    const int p = d->parent[i];
    if (p >= i || p < -1) { delete m; return RBD_ERR_INVALID_ARGUMENT; }  // parents first (src/mechanism_state.jl:44)
    if (p >= 0) kids[p].push_back(i); else roots.push_back(i);
  }
  if (qsum != d->nq || vsum != d->nv) { delete m; return RBD_ERR_DIMENSION_MISMATCH; }
  // DFS pre-order slots: first child of slot s is slot s+1
  m->order.clear(); m->slot_of.assign(nb, -1);
  {
    std::vector<int> stack(roots.rbegin(), roots.rend());
    while (!stack.empty()) {
      const int i = stack.back(); stack.pop_back();
      m->slot_of[i] = (int)m->order.size();
      m->order.push_back(i);
      for (auto it = kids[i].rbegin(); it != kids[i].rend(); ++it) stack.push_back(*it);
    }
  }
  std::vector<int> level(nb, 0);  // by slot
  m->ib.assign((size_t)nb * IB_STRIDE, -1);
  for (int s = 0; s < nb; ++s) {
    const int i = m->order[s];
    const int p = d->parent[i];
    const int ps = p < 0 ? -1 : m->slot_of[p];
    level[s] = ps < 0 ? 0 : level[ps] + 1;
    if (level[s] + 1 > m->nlevels) m->nlevels = level[s] + 1;
    int32_t* ib = &m->ib[(size_t)s * IB_STRIDE];
    ib[IB_PARENT] = ps; ib[IB_JTYPE] = d->joint_type[i]; ib[IB_QOFF] = d->q_offset[i]; ib[IB_VOFF] = d->v_offset[i];
    ib[IB_LEVEL] = level[s]; ib[IB_ORIG] = i; ib[IB_FLAGS] = 0;
    if (d->joint_type[i] == RBD_JOINT_QUAT_FLOATING && ps >= 0) m->inner_floating = 1;
    const int nc = (int)kids[i].size();
    if (nc > IB_MAXCHILD) { delete m; return RBD_ERR_UNSUPPORTED; }
    ib[IB_NCHILD] = nc;
    for (int k = 0; k < nc; ++k) ib[IB_CHILD0 + k] = m->slot_of[kids[i][k]];  // slot_of children assigned below for later slots
    if (nc > m->maxchild) m->maxchild = nc;
  }
  // children slots (now that every slot is known)
  for (int s = 0; s < nb; ++s) {
    const int i = m->order[s];
    for (int k = 0; k < (int)kids[i].size(); ++k) m->ib[(size_t)s * IB_STRIDE + IB_CHILD0 + k] = m->slot_of[kids[i][k]];
  }
  if (m->nlevels > MAX_LEVELS) { delete m; return RBD_ERR_UNSUPPORTED; }
  m->nslots.assign(MAX_LEVELS, 0);
  m->perm_down = 0;
  for (int s = 0; s < nb; ++s) {
    const int nc = m->ib[(size_t)s * IB_STRIDE + IB_NCHILD];
    if (nc > 0 && nc > m->nslots[level[s] + 1]) m->nslots[level[s] + 1] = nc;
    const int ps = m->ib[(size_t)s * IB_STRIDE + IB_PARENT];
    if (ps >= 0 && ps != s - 1) m->perm_down |= (uint64_t)1 << level[s];
    if (ps >= 0 && m->ib[(size_t)ps * IB_STRIDE + IB_CHILD0] != ps + 1) { delete m; return RBD_ERR_INVALID_ARGUMENT; }  // pre-order invariant
  }
  m->rb.assign((size_t)nb * RB_STRIDE, 0.0);
  for (int s = 0; s < nb; ++s) {
    const int i = m->order[s];
    double* rb = &m->rb[(size_t)s * RB_STRIDE];
    for (int k = 0; k < 3; ++k) rb[RB_AXIS + k] = d->joint_axis[3 * i + k];
    if (d->joint_axis2) for (int k = 0; k < 3; ++k) rb[RB_AXIS2 + k] = d->joint_axis2[3 * i + k];
    for (int k = 0; k < 9; ++k) rb[RB_XPR + k] = d->pred_rot[9 * i + k];
    for (int k = 0; k < 3; ++k) rb[RB_XPP + k] = d->pred_trans[3 * i + k];
    const double* J = d->inertia_moment + 9 * i;
    rb[RB_J + 0] = J[0]; rb[RB_J + 1] = J[1]; rb[RB_J + 2] = J[2]; rb[RB_J + 3] = J[4]; rb[RB_J + 4] = J[5]; rb[RB_J + 5] = J[8];
    for (int k = 0; k < 3; ++k) rb[RB_MC + k] = d->inertia_cross[3 * i + k];
    rb[RB_M] = d->inertia_mass[i];
  }
  m->dof_body.assign(m->nv > 0 ? m->nv : 1, 0);
  for (int i = 0; i < nb; ++i)
    for (int k = 0; k < joint_nv_host(d->joint_type[i]); ++k) m->dof_body[d->v_offset[i] + k] = m->slot_of[i];
  m->anc.assign((size_t)nb * m->nlevels, -1);
  for (int s = 0; s < nb; ++s) {
    int a = s;
    for (int k = 0; k < m->nlevels && a >= 0; ++k) { m->anc[(size_t)s * m->nlevels + k] = a; a = m->ib[(size_t)a * IB_STRIDE + IB_PARENT]; }
  }
  // support structure per dof row (support_set_masks, src/mechanism_state.jl:95-98): row r is row_words 64-bit words, bit c of word c / 64.
  // (64 bodies of 6-dof joints — a mechanism in maximal coordinates, test/test_mechanism_modification.jl:274-318 — make nv up to 384.)
  m->row_words = std::max(1, (m->nv + 63) / 64);
  m->row_mask.assign((size_t)(m->nv > 0 ? m->nv : 1) * m->row_words, 0);
  for (int r = 0; r < m->nv; ++r) {
    const int sr = m->dof_body[r];
    for (int c = 0; c <= r; ++c) {
      const int sc = m->dof_body[c];
      bool sup = false;
      for (int k = 0; k < m->nlevels; ++k) sup |= (m->anc[(size_t)sr * m->nlevels + k] == sc);
      if (sup) m->row_mask[(size_t)r * m->row_words + c / 64] |= (uint64_t)1 << (c % 64);
    }
  }
  m->nc = 0;
  m->jt_ref.assign(d->joint_type, d->joint_type + nb);
  m->parent_ref.assign(d->parent, d->parent + nb);
  m->voff_ref.assign(d->v_offset, d->v_offset + nb);
  m->qoff_ref.assign(d->q_offset, d->q_offset + nb);
  for (int i = 0; i < nb; ++i) {
    const int jt = d->joint_type[i];
    if (jt == RBD_JOINT_QUAT_FLOATING) { m->mkf.push_back(d->q_offset[i]); m->mkf.push_back(d->v_offset[i]); }
    else if (jt == RBD_JOINT_REVOLUTE || jt == RBD_JOINT_PRISMATIC || jt == RBD_JOINT_SINCOS_REVOLUTE) { m->mk1.push_back(d->q_offset[i]); m->mk1.push_back(d->v_offset[i]); m->mk1.push_back(jt); }
  }
  m->loop_fused_ok = d->n_loops > 0 && nb <= 4 && m->nv <= 4;
  for (int i = 0; i < nb && m->loop_fused_ok; ++i) {
    const int t = d->joint_type[i];
    m->loop_fused_ok = d->parent[i] < i && (t == RBD_JOINT_REVOLUTE || t == RBD_JOINT_PRISMATIC || t == RBD_JOINT_SINCOS_REVOLUTE || t == RBD_JOINT_FIXED);
  }
  m->axis_ref.assign(d->joint_axis, d->joint_axis + 3 * nb);
  if (d->joint_axis2) m->axis2_ref.assign(d->joint_axis2, d->joint_axis2 + 3 * nb); else m->axis2_ref.assign(3 * nb, 0.0);
  if (int lst = build_loop_tables(m, d)) { delete m; return lst; }
  m->bank_aba_ok = (m->nloops == 0 && !m->has3dof && !m->inner_floating);  // the banked ABA handles 1-dof / fixed joints and 6-dof joints on the world
  if (m->nlevels >= 2) {
    // banked mapping (the RNEA variant takes every joint type): split the levels so that both banks fit the fewest lanes
    std::vector<int> per_level(m->nlevels, 0);
    for (int s = 0; s < nb; ++s) per_level[level[s]]++;
    int best_L0 = 0, best_lanes = 1 << 30, best_diff = 1 << 30;
    for (int L0 = 1; L0 < m->nlevels; ++L0) {
      int cA = 0, cB = 0;
      for (int l = 0; l < m->nlevels; ++l) (l < L0 ? cA : cB) += per_level[l];
      int lanes = 1;
      while (lanes < std::max(cA, cB)) lanes <<= 1;
      const int diff = std::abs(cA - cB);
      if (lanes < best_lanes || (lanes == best_lanes && diff < best_diff)) { best_lanes = lanes; best_L0 = L0; best_diff = diff; }
    }
    if (best_lanes < m->lps) {  // only when it packs more states into a wavefront than one body per lane does
      m->bank_lps = best_lanes; m->bank_L0 = best_L0;
      std::vector<int> bslot(nb, -1);  // slot within the body's bank; the global DFS pre-order filtered by level keeps first child = next slot
      int cnt[2] = {0, 0};
      for (int s = 0; s < nb; ++s) bslot[s] = cnt[level[s] >= best_L0]++;
      m->bank_nb[0] = cnt[0]; m->bank_nb[1] = cnt[1];
      for (int k = 0; k < 2; ++k) { m->bank_ib[k].assign((size_t)std::max(cnt[k], 1) * IB_STRIDE, -1); m->bank_rb[k].assign((size_t)std::max(cnt[k], 1) * RB_STRIDE, 0.0); }
      m->bank_perm_down = 0;
      for (int s = 0; s < nb; ++s) {
        const int k = level[s] >= best_L0, j = bslot[s];
        const int32_t* src = &m->ib[(size_t)s * IB_STRIDE];
        int32_t* dst = &m->bank_ib[k][(size_t)j * IB_STRIDE];
        memcpy(dst, src, sizeof(int32_t) * IB_STRIDE);
        const int ps = src[IB_PARENT];
        dst[IB_PARENT] = ps < 0 ? -1 : bslot[ps];
        for (int c2 = 0; c2 < src[IB_NCHILD]; ++c2) dst[IB_CHILD0 + c2] = bslot[src[IB_CHILD0 + c2]];
        if (ps >= 0 && level[s] != best_L0 && bslot[ps] != j - 1) { m->bank_perm_down |= (uint64_t)1 << level[s]; dst[IB_FLAGS] |= BFD_NOTFIRST; }
        memcpy(&m->bank_rb[k][(size_t)j * RB_STRIDE], &m->rb[(size_t)s * RB_STRIDE], sizeof(double) * RB_STRIDE);
      }
    }
  }
  {
    // tracks per state of the chain-scheduled ABA: enough for the chains that overlap in time, at most 4 (one wavefront per
    // SIMD of a CU at the LDS-bound residency); 
    int nheads = 0;
    for (int s = 0; s < nb; ++s) {
      const int ps = m->ib[(size_t)s * IB_STRIDE + IB_PARENT];
      if (ps < 0 || m->ib[(size_t)ps * IB_STRIDE + IB_CHILD0] != s) ++nheads;
    }
    int G = 1;
    while (G < nheads && G < 4) G <<= 1;
    if (m->nloops == 0) m->chain = build_chain_plan(nb, m->ib, G);
    if (m->nloops == 0 && G <= 4) m->track = build_track_plan(nb, m->ib, m->rb, G);
    if (m->track.ok) m->walk = build_walk_plan(m->track.ns, m->track.G, m->track.ri);
    if (m->nloops == 0 && m->nv <= 64) m->state = build_state_plan(nb, m->ib, m->rb);  // (the lane-per-state kernels keep one mask word per row)
    if (m->nloops == 0 && m->nv <= 64 && !m->state.ok) m->state_wide = build_state_plan(nb, m->ib, m->rb, true);
  }
  // ---- the same tree re-rooted at its centre, for the ABA kernels that take it ----
  if (m->bank_aba_ok) {
    m->rr = build_reroot(d);
    if (m->rr.ok) {
      const Reroot& R = m->rr;
      rbd_model::RrSlots& S = m->rrs;
      std::vector<std::vector<int>> kd(nb);
      for (int i = 0; i < nb; ++i) if (R.parent[i] >= 0) kd[R.parent[i]].push_back(i);
      std::vector<int> ord, slot(nb, -1), lev(nb, 0);
      {
        std::vector<int> stack{R.root};
        while (!stack.empty()) {
          const int i = stack.back(); stack.pop_back();
          slot[i] = (int)ord.size();
          ord.push_back(i);
          for (auto it = kd[i].rbegin(); it != kd[i].rend(); ++it) stack.push_back(*it);
        }
      }
      bool fits = (int)ord.size() == nb;
      for (int i = 0; i < nb && fits; ++i) fits = (int)kd[i].size() <= IB_MAXCHILD;
      if (fits) {
        S.ib.assign((size_t)nb * IB_STRIDE, -1);
        S.rb.assign((size_t)nb * RB_STRIDE, 0.0);
        S.nslots.assign(MAX_LEVELS, 0);
        for (int s2 = 0; s2 < nb; ++s2) {
          const int i = ord[s2], p = R.parent[i], ps = p < 0 ? -1 : slot[p];
          lev[s2] = ps < 0 ? 0 : lev[ps] + 1;
          if (lev[s2] + 1 > S.nlevels) S.nlevels = lev[s2] + 1;
          int32_t* ib = &S.ib[(size_t)s2 * IB_STRIDE];
          ib[IB_PARENT] = ps; ib[IB_JTYPE] = R.jtype[i]; ib[IB_QOFF] = R.qoff[i]; ib[IB_VOFF] = R.voff[i]; ib[IB_LEVEL] = lev[s2]; ib[IB_ORIG] = i;
          ib[IB_FLAGS] = R.flags[i];
          ib[IB_NCHILD] = (int)kd[i].size();
          for (int k = 0; k < (int)kd[i].size(); ++k) ib[IB_CHILD0 + k] = slot[kd[i][k]];
          double* rb = &S.rb[(size_t)s2 * RB_STRIDE];
          for (int k = 0; k < 3; ++k) { rb[RB_AXIS + k] = R.axis[3 * i + k]; rb[RB_AXIS2 + k] = R.axis2[3 * i + k]; rb[RB_XPP + k] = R.Xpp[3 * i + k]; rb[RB_MC + k] = R.mc[3 * i + k]; }
          for (int k = 0; k < 9; ++k) rb[RB_XPR + k] = R.XpR[9 * i + k];
          for (int k = 0; k < 6; ++k) rb[RB_J + k] = R.J6[6 * i + k];
          rb[RB_M] = R.mass[i];
        }
        for (int s2 = 0; s2 < nb; ++s2) {
          const int nc = S.ib[(size_t)s2 * IB_STRIDE + IB_NCHILD];
          if (nc > 0 && nc > S.nslots[lev[s2] + 1]) S.nslots[lev[s2] + 1] = nc;
        }
        // track / walk plans of the re-rooted tree (the walk kernels)
        int nheads = 0;
        for (int s2 = 0; s2 < nb; ++s2) {
          const int ps = S.ib[(size_t)s2 * IB_STRIDE + IB_PARENT];
          if (ps < 0 || S.ib[(size_t)ps * IB_STRIDE + IB_CHILD0] != s2) ++nheads;
        }
        int G = 1;
        while (G < nheads && G < 4) G <<= 1;
        S.track = build_track_plan(nb, S.ib, S.rb, G);
        if (S.track.ok) {
          S.walk = build_walk_plan(S.track.ns, S.track.G, S.track.ri);
          if (S.walk.ok)
            for (size_t k = 0; k < S.walk.wk.size(); ++k) {
              const int e = S.track.tab[k];
              if (e >= 0) S.walk.wk[k] |= S.ib[(size_t)e * IB_STRIDE + IB_FLAGS] << 8;
            }
        }
        S.ok = true;
      }
    }
  }
  *out = m;
  return RBD_OK;
}

int rbd_model_destroy(rbd_model_t* m) {
  delete m;
  return RBD_OK;
}

int rbd_model_bank_plan(const rbd_model_t* m, int32_t* lanes, int32_t* first_level_of_bank1, int32_t* bodies_bank0, int32_t* bodies_bank1,
                        int32_t* aba_in_scope) {
  if (!m) return RBD_ERR_INVALID_ARGUMENT;
  if (m->bank_lps <= 0) return RBD_ERR_UNSUPPORTED;  // two bodies per lane would not pack more states into a wavefront
  if (lanes) *lanes = m->bank_lps;
  if (first_level_of_bank1) *first_level_of_bank1 = m->bank_L0;
  if (bodies_bank0) *bodies_bank0 = m->bank_nb[0];
  if (bodies_bank1) *bodies_bank1 = m->bank_nb[1];
  if (aba_in_scope) *aba_in_scope = m->bank_aba_ok;
  return RBD_OK;
}

int rbd_model_chain_plan(const rbd_model_t* m, int32_t* tracks, int32_t* steps, int32_t* lds_fields, int32_t* table, int32_t capacity) {
  if (!m) return RBD_ERR_INVALID_ARGUMENT;
  if (!m->chain.ok) return RBD_ERR_UNSUPPORTED;
  if (tracks) *tracks = m->chain.G;
  if (steps) *steps = m->chain.ns;
  if (lds_fields) *lds_fields = 0;  // (LDS footprint of the removed chain kernel; kept for ABI stability)
  if (table) {
    if (capacity < (int32_t)m->chain.tab.size()) return RBD_ERR_DIMENSION_MISMATCH;
    for (size_t i = 0; i < m->chain.tab.size(); ++i) table[i] = m->chain.tab[i] < 0 ? -1 : m->order[m->chain.tab[i]];  // reference body indices
  }
  return RBD_OK;
}

int rbd_model_track_plan(const rbd_model_t* m, int32_t* dims, int32_t* table, int32_t table_cap, int32_t* ri, int32_t ri_cap, double* rr, int32_t rr_cap) {
  if (!m) return RBD_ERR_INVALID_ARGUMENT;
  if (!m->track.ok) return RBD_ERR_UNSUPPORTED;
  const TrackPlan& P = m->track;
  if (dims) { dims[0] = P.G; dims[1] = P.ns; dims[2] = P.nA; dims[3] = P.nB; dims[4] = P.has_floating; dims[5] = P.general; }
  if (table) {
    if (table_cap < (int32_t)P.tab.size()) return RBD_ERR_DIMENSION_MISMATCH;
    for (size_t i = 0; i < P.tab.size(); ++i) table[i] = P.tab[i] < 0 ? -1 : m->order[P.tab[i]];  // reference body indices
  }
  if (ri) {  // the packed records, then the ns per-step flags
    if (ri_cap < (int32_t)(P.ri.size() + P.sf.size())) return RBD_ERR_DIMENSION_MISMATCH;
    memcpy(ri, P.ri.data(), P.ri.size() * sizeof(int32_t));
    memcpy(ri + P.ri.size(), P.sf.data(), P.sf.size() * sizeof(int32_t));
  }
  if (rr) { if (rr_cap < (int32_t)P.rr.size()) return RBD_ERR_DIMENSION_MISMATCH; memcpy(rr, P.rr.data(), P.rr.size() * sizeof(double)); }
  return RBD_OK;
}

int rbd_model_reroot_plan(const rbd_model_t* m, int32_t* dims, int32_t* ri, int32_t ri_cap, double* rr, int32_t rr_cap, int32_t* wk, int32_t wk_cap,
                          int32_t* chain_i, double* chain_r, double* fxp) {
  if (!m) return RBD_ERR_INVALID_ARGUMENT;
  if (!m->rrs.ok || !m->rrs.track.ok || !m->rrs.walk.ok) return RBD_ERR_UNSUPPORTED;
  const TrackPlan& P = m->rrs.track;
  const Reroot& R = m->rr;
  const int nch = (int)(R.chain_i.size() / RC_I_STRIDE);
  if (dims) {
    const int32_t d[12] = {P.G, P.ns, P.nA, P.nB, P.has_floating, P.general, m->rrs.walk.nS, nch, R.fq, R.fv, R.root, R.fb};
    memcpy(dims, d, sizeof d);
  }
  if (ri) {
    if (ri_cap < (int32_t)(P.ri.size() + P.sf.size())) return RBD_ERR_DIMENSION_MISMATCH;
    memcpy(ri, P.ri.data(), P.ri.size() * sizeof(int32_t));
    memcpy(ri + P.ri.size(), P.sf.data(), P.sf.size() * sizeof(int32_t));
  }
  if (rr) { if (rr_cap < (int32_t)P.rr.size()) return RBD_ERR_DIMENSION_MISMATCH; memcpy(rr, P.rr.data(), P.rr.size() * sizeof(double)); }
  if (wk) { if (wk_cap < (int32_t)m->rrs.walk.wk.size()) return RBD_ERR_DIMENSION_MISMATCH; memcpy(wk, m->rrs.walk.wk.data(), m->rrs.walk.wk.size() * sizeof(int32_t)); }
  if (chain_i) memcpy(chain_i, R.chain_i.data(), R.chain_i.size() * sizeof(int32_t));
  if (chain_r) memcpy(chain_r, R.chain_r.data(), R.chain_r.size() * sizeof(double));
  if (fxp) memcpy(fxp, R.fXp, sizeof R.fXp);
  return RBD_OK;
}

int rbd_model_dims(const rbd_model_t* m, int32_t* nb, int32_t* nq, int32_t* nv, int32_t* nc) {
  if (!m) return RBD_ERR_INVALID_ARGUMENT;
  if (nb) *nb = m->nb;
  if (nq) *nq = m->nq;
  if (nv) *nv = m->nv;
  if (nc) *nc = m->nc;
  return RBD_OK;
}

static int upload(void** dst, const void* src, size_t bytes) {
  HIP_TRY(hipMalloc(dst, bytes ? bytes : 16));
  if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return RBD_OK;
}

// the loop joints' tables on the device (workspace copies: rbd_workspace_set_loop_gains rewrites the gains in place)
static int upload_loop_tables(rbd_ws* w, const rbd_model* m, int dtype) {
  int st = RBD_OK;
  if (st == RBD_OK && m->nloops > 0) {
    st = upload(&w->d_loop_i, m->loop_i.data(), m->loop_i.size() * sizeof(int32_t));
    if (st == RBD_OK) st = upload(&w->d_loop_path, m->loop_path.data(), m->loop_path.size() * sizeof(int32_t));
    if (st == RBD_OK) st = upload(&w->d_jt_ref, m->jt_ref.data(), m->jt_ref.size() * sizeof(int32_t));
    if (st == RBD_OK) st = upload(&w->d_voff_ref, m->voff_ref.data(), m->voff_ref.size() * sizeof(int32_t));
    if (st == RBD_OK && m->loop_fused_ok) {
      std::vector<int32_t> xi(3 * (size_t)m->nb);
      for (int i = 0; i < m->nb; ++i) { xi[3 * i] = m->parent_ref[i]; xi[3 * i + 1] = m->qoff_ref[i]; xi[3 * i + 2] = m->slot_of[i]; }
      st = upload(&w->d_fused_i, xi.data(), xi.size() * sizeof(int32_t));
    }
    if (st == RBD_OK) {
      if (dtype == RBD_F64) {
        st = upload(&w->d_loop_r, m->loop_r.data(), m->loop_r.size() * sizeof(double));
        if (st == RBD_OK) st = upload(&w->d_axis_ref, m->axis_ref.data(), m->axis_ref.size() * sizeof(double));
        if (st == RBD_OK) st = upload(&w->d_axis2_ref, m->axis2_ref.data(), m->axis2_ref.size() * sizeof(double));
      } else {
        std::vector<float> a(m->loop_r.begin(), m->loop_r.end()), b2(m->axis_ref.begin(), m->axis_ref.end()), b3(m->axis2_ref.begin(), m->axis2_ref.end());
        st = upload(&w->d_loop_r, a.data(), a.size() * sizeof(float));
        if (st == RBD_OK) st = upload(&w->d_axis_ref, b2.data(), b2.size() * sizeof(float));
        if (st == RBD_OK) st = upload(&w->d_axis2_ref, b3.data(), b3.size() * sizeof(float));
      }
    }
  }
  return st;
}

// soft contact: the points' and half-spaces' tables on the device (contact_kernel)
static int upload_contact_tables(rbd_ws* w, const rbd_model* m, int dtype) {
  if (m->ncp <= 0) return RBD_OK;
  int st = upload(&w->d_cp_body, m->cp_body.data(), m->cp_body.size() * sizeof(int32_t));
  auto up = [&](void** dst, const std::vector<double>& src) {
    if (dtype == RBD_F64) return upload(dst, src.data(), src.size() * sizeof(double));
    std::vector<float> f(src.begin(), src.end());
    return upload(dst, f.data(), f.size() * sizeof(float));
  };
  if (st == RBD_OK) st = up(&w->d_cp_r, m->cp_r);
  if (st == RBD_OK) st = up(&w->d_hs_r, m->hs_r);
  if (st != RBD_OK) return st;
  w->ctm.nb = m->nb; w->ctm.np = m->ncp; w->ctm.nh = m->nhs;
  w->ctm.cbody = (const int32_t*)w->d_cp_body; w->ctm.cp = w->d_cp_r; w->ctm.hs = w->d_hs_r;
  return RBD_OK;
}

int rbd_workspace_create(const rbd_model_t* m, int32_t max_batch, int32_t device, int32_t dtype, void* stream, rbd_ws_t** out) {
  if (!m || !out || max_batch < 1 || (dtype != RBD_F64 && dtype != RBD_F32)) return RBD_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_last_hip_error = "hipGetDeviceCount: no device";
    return RBD_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return RBD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(device));
  rbd_ws* w = new (std::nothrow) rbd_ws();
  if (!w) return RBD_ERR_OUT_OF_MEMORY;
  w->model = m; w->device = device; w->dtype = dtype; w->max_batch = max_batch; w->stream = (hipStream_t)stream;
  if (m->big) {  // the any-size fallback needs its two tables only
    int st = upload(&w->d_big_tbl, m->big_tbl.data(), m->big_tbl.size() * sizeof(int32_t));
    if (st == RBD_OK) {
      if (dtype == RBD_F64) st = upload(&w->d_big_rb, m->big_rb.data(), m->big_rb.size() * sizeof(double));
      else { std::vector<float> f(m->big_rb.begin(), m->big_rb.end()); st = upload(&w->d_big_rb, f.data(), f.size() * sizeof(float)); }
    }
    if (st == RBD_OK) { int zero = 0; st = upload((void**)&w->d_notpd, &zero, sizeof(int)); }
    if (st == RBD_OK) st = upload_loop_tables(w, m, dtype);
    if (st == RBD_OK) st = upload_contact_tables(w, m, dtype);
    if (st != RBD_OK) { rbd_workspace_destroy(w); return st; }
    w->big.nb = m->nb; w->big.nq = m->nq; w->big.nv = m->nv; w->big.tbl = (const int32_t*)w->d_big_tbl; w->big.rb = w->d_big_rb;
    memcpy(w->big.gravity, m->gravity, sizeof w->big.gravity);
    w->last_kernel = "big_* kernels (one thread per state, HBM scratch)";
    *out = w;
    return RBD_OK;
  }
  int st = upload(&w->d_ib, m->ib.data(), m->ib.size() * sizeof(int32_t));
  if (st == RBD_OK) {
    if (dtype == RBD_F64) {
      st = upload(&w->d_rb, m->rb.data(), m->rb.size() * sizeof(double));
    } else {
      std::vector<float> rbf(m->rb.begin(), m->rb.end());
      st = upload(&w->d_rb, rbf.data(), rbf.size() * sizeof(float));
    }
  }
  if (st == RBD_OK) st = upload(&w->d_dof_body, m->dof_body.data(), m->dof_body.size() * sizeof(int32_t));
  if (st == RBD_OK) st = upload(&w->d_anc, m->anc.data(), m->anc.size() * sizeof(int32_t));
  if (st == RBD_OK) st = upload(&w->d_row_mask, m->row_mask.data(), m->row_mask.size() * sizeof(uint64_t));
  if (st == RBD_OK) { int zero = 0; st = upload((void**)&w->d_notpd, &zero, sizeof(int)); }
  if (st == RBD_OK) st = upload_loop_tables(w, m, dtype);
  if (st != RBD_OK) { rbd_workspace_destroy(w); return st; }
  DevModel& dm = w->dm;
  dm.nb = m->nb; dm.nq = m->nq; dm.nv = m->nv; dm.lps = m->lps; dm.nlevels = m->nlevels; dm.maxchild = m->maxchild; dm.maxnvj = m->maxnvj;
  dm.ib = (const int32_t*)w->d_ib; dm.rb = w->d_rb;
  dm.perm_down = m->perm_down;
  dm.inner_floating = m->inner_floating;
  dm.has3dof = m->has3dof;
  dm.nheavy = 0;
  for (int s2 = 0; s2 < m->nb; ++s2) {
    const int jt = m->ib[(size_t)s2 * IB_STRIDE + IB_JTYPE];
    if (jt != RBD_JOINT_QUAT_FLOATING && jt != RBD_JOINT_QUAT_SPHERICAL) continue;
    if (dm.nheavy < 4) { dm.heavy[dm.nheavy][0] = jt; dm.heavy[dm.nheavy][1] = m->ib[(size_t)s2 * IB_STRIDE + IB_QOFF]; dm.heavy[dm.nheavy][2] = m->ib[(size_t)s2 * IB_STRIDE + IB_VOFF]; }
    ++dm.nheavy;
  }
  nslots_pack_desc(dm.ns_desc, m->nslots.data(), m->nlevels);
  dm.dof_body = (const int32_t*)w->d_dof_body; dm.anc = (const int32_t*)w->d_anc; dm.row_mask = (const uint64_t*)w->d_row_mask; dm.row_words = m->row_words;
  memcpy(dm.gravity, m->gravity, sizeof dm.gravity);
  if (m->bank_lps > 0) {
    BankModel& bm = w->bm;
    for (int k = 0; k < 2 && st == RBD_OK; ++k) {
      st = upload(&w->d_bank_ib[k], m->bank_ib[k].data(), m->bank_ib[k].size() * sizeof(int32_t));
      if (st != RBD_OK) break;
      if (dtype == RBD_F64) st = upload(&w->d_bank_rb[k], m->bank_rb[k].data(), m->bank_rb[k].size() * sizeof(double));
      else { std::vector<float> f(m->bank_rb[k].begin(), m->bank_rb[k].end()); st = upload(&w->d_bank_rb[k], f.data(), f.size() * sizeof(float)); }
      bm.ib[k] = (const int32_t*)w->d_bank_ib[k]; bm.rb[k] = w->d_bank_rb[k]; bm.nbk[k] = m->bank_nb[k];
    }
    if (st != RBD_OK) { rbd_workspace_destroy(w); return st; }
    bm.lps = m->bank_lps; bm.nlevels = m->nlevels; bm.L0 = m->bank_L0; bm.perm_down = m->bank_perm_down;
    bm.simple = bank_simple(m);  // every tree joint revolute, apart from 6-dof joints on the world
    if (tune("bank_generic", 0)) bm.simple = 0;  // tests: the generic instantiation on a mechanism the SIMPLE one would take
    nslots_pack_desc(bm.ns_desc, m->nslots.data(), m->nlevels);
    memcpy(bm.gravity, m->gravity, sizeof bm.gravity);
    // Two bodies per lane wherever the mechanism is in its scope: since round 3 it is ahead of one body per lane at every batch size
    // (profiles/r03_mapping_sweep.txt: 18.8 vs 21.0 us at 512 Atlas states, 19.1 vs 30.1 at 4096).  RBD_BANK_MIN_BATCH: tests.
    w->bank_min_batch = 0;
    // ... except inverse dynamics of mechanisms with 3-dof joints (three columns per body in the banked kernel): one body per lane is ahead up to ≈ 7000 states —
    // fp64, 1024 / 4096 states: randmech() seeds 1-3 12.3-12.6 / 15.6-16.3 us against 16.8-18.8 / 17.6-19.8, mixed20 11.3 / 11.2 against 14.8 / 13.7; at 8192 the
    // banked kernel leads again (22.7-26.1 against 26.5-27.2) (round 6, scripts/sweep_routes.py)
    w->rnea_bank_min_batch = 0;
    {
      int ncu = 256;
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
      if (m->has3dof) w->rnea_bank_min_batch = (long)ncu * 28 + 1;  // (6000 states: 19.9 / 12.3 against 23.7 / 13.9)
    }
    { bool has; const long t = tune("bank_min_batch", 0, &has); if (has) w->bank_min_batch = w->rnea_bank_min_batch = t; }
    // ... up to the batch whose workgroups (256 lanes) are all resident at once: per compute unit as many as the LDS columns allow
    // (park + exchange pairs: 120 KB in fp64 -> one, 60 KB in fp32 -> two), at most the two the register budget allows
    {
      int ncu = 256;
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
      const size_t lds = (size_t)BANK_LDS_PAIRS_HOST * 256 * 2 * (dtype == RBD_F64 ? 8 : 4);
      const long per_cu = std::max<long>(1, std::min<long>(2, (long)(160 * 1024 / lds)));
      w->bank_resident_states = (long)ncu * per_cu * (256 / m->bank_lps);
    }
  }
  if (m->rrs.ok) {
    const Reroot& R = m->rr;
    const rbd_model::RrSlots& S = m->rrs;
    auto up_real = [&](void** dst, const std::vector<double>& src) {
      if (dtype == RBD_F64) return upload(dst, src.data(), src.size() * sizeof(double));
      std::vector<float> f(src.begin(), src.end());
      return upload(dst, f.data(), f.size() * sizeof(float));
    };
    st = upload(&w->d_rr_chain_i, R.chain_i.data(), R.chain_i.size() * sizeof(int32_t));
    if (st == RBD_OK) st = up_real(&w->d_rr_chain_r, R.chain_r);
    RerootView V{};
    V.nchain = (int32_t)(R.chain_i.size() / RC_I_STRIDE); V.fq = R.fq; V.fv = R.fv;
    V.chain_i = (const int32_t*)w->d_rr_chain_i; V.chain_r = w->d_rr_chain_r;
    memcpy(V.fXp, R.fXp, sizeof V.fXp);
    if (st == RBD_OK && S.track.ok && S.walk.ok) {
      const TrackPlan& P = S.track;
      st = upload(&w->d_rrtrack_ri, P.ri.data(), P.ri.size() * sizeof(int32_t));
      if (st == RBD_OK) st = up_real(&w->d_rrtrack_rr, P.rr);
      if (st == RBD_OK) st = upload(&w->d_rrwalk_wk, S.walk.wk.data(), S.walk.wk.size() * sizeof(int32_t));
      WalkModel& wm = w->wm_rr;
      wm.ns = P.ns; wm.G = P.G; wm.nA = P.nA; wm.nB = P.nB; wm.nS = S.walk.nS; wm.nq = m->nq; wm.nv = m->nv; wm.reroot = V;
      wm.ri = (const int32_t*)w->d_rrtrack_ri; wm.rr = w->d_rrtrack_rr; wm.wk = (const int32_t*)w->d_rrwalk_wk;
      for (int k = 0; k < 5; ++k) { wm.sfm[k] = 0; for (int s2 = 0; s2 < P.ns; ++s2) wm.sfm[k] |= (uint64_t)((P.sf[s2] >> k) & 1) << s2; }
      memcpy(wm.gravity, m->gravity, sizeof wm.gravity);
      const size_t es2 = dtype == RBD_F64 ? 8 : 4;
      w->walk_rr_lds_bytes = walk_lds_bytes(P.ns, P.G, m->nq, m->nv, P.nA, P.nB, S.walk.nS, es2, es2);
      if (w->walk_rr_lds_bytes > 160 * 1024) w->walk_rr_lds_bytes = 0;
      w->walk_rr_lds_bytes_pair = dtype == RBD_F32 ? walk_lds_bytes(P.ns, P.G, m->nq, m->nv, P.nA, P.nB, S.walk.nS, 8, 4) : 0;
      if (w->walk_rr_lds_bytes_pair > 160 * 1024) w->walk_rr_lds_bytes_pair = 0;
      w->walk_rr = st == RBD_OK && (w->walk_rr_lds_bytes > 0 || w->walk_rr_lds_bytes_pair > 0);
    }
    if (st != RBD_OK) { rbd_workspace_destroy(w); return st; }
  }
  if (m->track.ok) {
    const TrackPlan& P = m->track;
    st = upload(&w->d_track_ri, P.ri.data(), P.ri.size() * sizeof(int32_t));
    if (st == RBD_OK) {
      if (dtype == RBD_F64) st = upload(&w->d_track_rr, P.rr.data(), P.rr.size() * sizeof(double));
      else { std::vector<float> f(P.rr.begin(), P.rr.end()); st = upload(&w->d_track_rr, f.data(), f.size() * sizeof(float)); }
    }
    if (st != RBD_OK) { rbd_workspace_destroy(w); return st; }
    TrackModel& tm = w->tm;
    tm.ns = P.ns; tm.G = P.G; tm.nA = P.nA; tm.nB = P.nB; tm.ri = (const int32_t*)w->d_track_ri; tm.rr = w->d_track_rr;
    for (int k = 0; k < 5; ++k) { tm.sfm[k] = 0; for (int s2 = 0; s2 < P.ns; ++s2) tm.sfm[k] |= (uint64_t)((P.sf[s2] >> k) & 1) << s2; }
    memcpy(tm.gravity, m->gravity, sizeof tm.gravity);
  }
  if ((st = upload_contact_tables(w, m, dtype)) != RBD_OK) { rbd_workspace_destroy(w); return st; }
  if (m->track.ok && m->walk.ok) {
    const TrackPlan& P = m->track;
    st = upload(&w->d_walk_wk, m->walk.wk.data(), m->walk.wk.size() * sizeof(int32_t));
    if (st != RBD_OK) { rbd_workspace_destroy(w); return st; }
    // the walk kernel compiled for the mechanism (aba_walk_spec): from the batch size at which RBD_ALGO_ABA picks the walk kernel by itself on a humanoid — smaller
    // batches that force the mapping (tests, sweeps) keep the interpreting kernel and its zero start-up cost (the compile takes about a minute per mechanism, once)
    w->spec_walk_min_batch = 8192;
    { bool has; const long t = tune("spec_walk_min_batch", 0, &has); if (has) w->spec_walk_min_batch = t; }
    w->spec_walk_f32 = tune("spec_walk_f32", 1) != 0;
    WalkModel& wm = w->wm;
    wm.ns = P.ns; wm.G = P.G; wm.nA = P.nA; wm.nB = P.nB; wm.nS = m->walk.nS; wm.nq = m->nq; wm.nv = m->nv;
    wm.ri = (const int32_t*)w->d_track_ri; wm.rr = w->d_track_rr; wm.wk = (const int32_t*)w->d_walk_wk;
    for (int k = 0; k < 5; ++k) wm.sfm[k] = w->tm.sfm[k];
    memcpy(wm.gravity, m->gravity, sizeof wm.gravity);
    const size_t es = dtype == RBD_F64 ? 8 : 4;
    w->walk_lds_bytes = walk_lds_bytes(P.ns, P.G, m->nq, m->nv, P.nA, P.nB, m->walk.nS, es, es);
    if (w->walk_lds_bytes > 160 * 1024) w->walk_lds_bytes = 0;  // the rows of 64 states do not fit one CU's LDS: other mappings
    // fp32: two states per lane (packed arithmetic), 128 states per workgroup
    w->walk_lds_bytes_pair = dtype == RBD_F32 ? walk_lds_bytes(P.ns, P.G, m->nq, m->nv, P.nA, P.nB, m->walk.nS, 8, 4) : 0;
    if (w->walk_lds_bytes_pair > 160 * 1024) w->walk_lds_bytes_pair = 0;
    {
      const size_t l1 = std::max(w->walk_lds_bytes, w->walk_rr ? w->walk_rr_lds_bytes : (size_t)0), l2 = std::max(w->walk_lds_bytes_pair, w->walk_rr ? w->walk_rr_lds_bytes_pair : (size_t)0);
      const hipError_t e = dtype == RBD_F64 ? configure_walk_kernel<double>(P.has_floating, P.general, l1, 0)
                                            : configure_walk_kernel<float>(P.has_floating, P.general, l1, l2);
      if (e != hipSuccess) { g_last_hip_error = std::string("configure_walk_kernel: ") + hipGetErrorString(e); rbd_workspace_destroy(w); return RBD_ERR_HIP; }
    }
    // the packed form from the batch size at which the 64-state workgroups no longer fit the chip in one round (RBD_WALK_PAIR_MIN_BATCH overrides)
    {
      int ncu = 256;
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
      w->walk_pair_min_batch = (long)ncu * 64 + 1;
      { bool has; const long t = tune("walk_pair_min_batch", 0, &has); if (has) w->walk_pair_min_batch = t; }
    }
    // RBD_ALGO_ABA picks it from this batch size up (RBD_WALK_MIN_BATCH overrides): one state past what the lane-per-body kernels run in ONE
    // round of resident wavefronts — the banked kernel's resident workgroups when the mechanism has banks, else two one-body-per-lane
    // wavefronts per SIMD.  (profiles/r03_mapping_sweep.txt, Atlas: a walk launch takes the same ~37 us from 64 to 16 384 states; banked fp64
    // 19 us to 4096 states and 36 us from there to 8192; banked fp32 18 us to 4096, 26 to 8192, 36 at 12 288.)
    {
      int ncu = 256;
      (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
      w->walk_min_batch = (m->bank_lps > 0 && m->bank_aba_ok ? w->bank_resident_states : (long)ncu * 4 * 2 * (64 / m->lps)) + 1;
      // Second half of round 3 (profiles/r03_mid_batches.txt, Atlas): compiled for the mechanism the walk launch takes 23 us (fp64) / 19 us (fp32) up to 16 384 states,
      // so it is ahead of the banked kernel as soon as that one needs a second workgroup per CU (fp64 38 us, fp32 26 us from 4097 states): dynamics! switches there
      // when the compiled kernel is available.  Inverse dynamics' banked kernel keeps two workgroups per CU in both precisions and stays ahead up to their 8192 states
      // (fp64 17.7-19.0 vs 19.8 us, fp32 17.0-18.0 vs 17.8-18.1).
      const bool banked = m->bank_lps > 0 && m->bank_aba_ok;
      w->walk_one_round_batch = banked ? (long)ncu * (256 / m->bank_lps) + 1 : w->walk_min_batch;
      w->rnea_walk_min_batch = banked ? std::max<long>(w->walk_min_batch, (long)ncu * 2 * (256 / m->bank_lps) + 1) : w->walk_min_batch;
      if (bool has = false; (void)tune("spec_walk_min_batch", 0, &has), !has) w->spec_walk_min_batch = std::min<long>(w->spec_walk_min_batch, w->walk_one_round_batch);
    }
    { bool has; const long t = tune("walk_min_batch", 0, &has); if (has) w->walk_min_batch = t; }
  }
  if (m->state.ok && m->state.nlevels <= state_max_levels(dtype == RBD_F64 ? 8 : 4)) {
    const StatePlan& P = m->state;
    st = upload(&w->d_state_ops, P.ops.data(), P.ops.size() * sizeof(int32_t));
    if (st == RBD_OK) st = upload(&w->d_state_cols, P.cols.data(), P.cols.size() * sizeof(int32_t));
    if (st == RBD_OK) {
      if (dtype == RBD_F64) st = upload(&w->d_state_sr, P.sr.data(), P.sr.size() * sizeof(double));
      else { std::vector<float> f(P.sr.begin(), P.sr.end()); st = upload(&w->d_state_sr, f.data(), f.size() * sizeof(float)); }
    }
    if (st != RBD_OK) { rbd_workspace_destroy(w); return st; }
    StateModel& sm = w->sm;
    sm.nb = m->nb; sm.nq = m->nq; sm.nv = m->nv; sm.nops = P.nops; sm.nlevels = P.nlevels;
    sm.ops = (const int32_t*)w->d_state_ops; sm.cols = (const int32_t*)w->d_state_cols; sm.sr = w->d_state_sr; sm.row_mask = (const uint64_t*)w->d_row_mask;
    memcpy(sm.gravity, m->gravity, sizeof sm.gravity);
    // a lane per state pays once every SIMD of the chip has a wavefront of them (64 states x 4 SIMDs x CUs); below that the
    // lane-per-body kernels spread a small batch over more of the chip
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    w->state_min_batch = (long)ncu * 4 * 64 / 2;
    { bool has; const long t = tune("state_min_batch", 0, &has); if (has) w->state_min_batch = t; }
    // fp32 mass_matrix! + Cholesky solve on the two kernels compiled for the mechanism (crba_spec_perm + chol_spec): ahead of the one-body-per-lane kernels at
    // EVERY batch measured — Atlas: 256 states 28.1 against 40.4 us, 4096: 31.1 against 44.8, 16 384: 46.1 against 117, 24 576: 58.2 against 161; without M
    // 22 - 33 us, the packed triangle 31 - 51 against 44 - 171 (round 6, scripts/exp_mass_small.sh; until then they waited for state_min_batch = 32 768).
    // mass_matrix! alone (crba_spec_perm + emit_spec): 1024 states 31.2 against 19.0 us, 4096: 33.7 against 21.5, 16 384: 42.4 against 53.1 — from 10 241.
    // fp64 (crba_spec + emit_spec + the dense Cholesky kernel): 4096 states 94 against 54 us, 16 384: 179 against 183 — stays at state_min_batch.
    w->mass_min_batch = w->mass_solve_min_batch = w->state_min_batch;
    if (dtype == RBD_F32) { w->mass_min_batch = std::min<long>(w->state_min_batch, (long)ncu * 40 + 1); w->mass_solve_min_batch = std::min<long>(w->state_min_batch, 256); }
    { bool has; const long t = tune("mass_min_batch", 0, &has); if (has) w->mass_min_batch = w->mass_solve_min_batch = t; }
    // the kernels compiled for the mechanism (spec_load): a wavefront of 64 states per SIMD — one round of them takes the same time from one wavefront to a
    // chip-full, and beats the walk kernel's rounds of half as many states from the second of those on.  Known here, before anything is compiled, so that a
    // small batch never starts (or waits for) a compilation it would not use
    w->spec_aba_min_batch = w->spec_rnea_min_batch = (long)ncu * 4 * 64 / 2 + 1;
    // (fp64 inverse_dynamics!: the walk kernel's rounds of 16 384 states stay ahead of rnea_spec_f64 until the fourth — Atlas, 49 152 states 58.5 against 66.1 us,
    //  65 536: 77.1 against 75.7; scripts/sweep_routes.py)
    if (dtype == RBD_F64 && m->track.ok && m->walk.ok && w->walk_lds_bytes > 0) w->spec_rnea_min_batch = (long)ncu * 240 + 1;
    // (no walk kernel for the mechanism — its rows do not fit a CU's LDS —: behind rnea_spec is the banked kernel, whose time grows with the batch; limbs_humanoid
    //  fp64, 16 384 states: 85.3 us against 54.1 compiled -> the thresholds of the mechanisms outside the walk kernels' scope)
    else if (!(m->track.ok && m->walk.ok && w->walk_lds_bytes > 0)) w->spec_rnea_min_batch = (long)ncu * (dtype == RBD_F64 ? 40 : 52) + 1;
    w->spec_aba_fused_min_batch = (long)ncu * 80;  // (`simulate`: run_aba)
    // `simulate` in fp32: as long as the batch is ONE round of the walk kernel with two states per lane (128 states per workgroup, one workgroup per CU), the four
    // stages of a step in one launch of it beat four launches of the lane-per-state kernel (Atlas, 32 768 states: 128 against 150 us per step; 40 960 — a second
    // round — 237 against 161)
    w->sim_walk_max_batch = tune("sim_walk_max_batch", (long)ncu * 128);
    w->sim_walk_min_batch = tune("sim_walk_min_batch", 1);  // (`simulate` on the looped walk program from this batch on when it is compiled: simulate_core)
    { bool has; const long t = tune("spec_aba_min_batch", 0, &has); if (has) w->spec_aba_min_batch = w->spec_aba_fused_min_batch = t; }
    { bool has; const long t = tune("spec_rnea_min_batch", 0, &has); if (has) w->spec_rnea_min_batch = t; }
    // the kinematics by-products one lane per state (kin_spec / jac_spec / mom_spec): from an eighth of a chip-full of wavefronts on — a round of them takes the
    // same 30 us per kernel from one wavefront to a chip-full, the one-body-per-lane kernels' time grows with the batch (Atlas fp64, the set of three calls:
    // 4096 states 57.5 against 89.3 us compiled, 8192: 96.6 against 91.8, 16 384: 160 against 117, 32 768: 284 against 133; scripts/exp_kin_small.sh)
    w->spec_kin_min_batch = tune("spec_kin_min_batch", (long)ncu * 32 + 1);
    w->state_aot = true;
  } else if (m->state_wide.ok) {  // the compiled kernels alone (no interpreting form of them for these joint types): the same thresholds, the lane-per-body kernels behind them
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    // behind them here: one body per lane (no walk kernel), whose time grows with the batch while a lane per state takes one round up to a chip-full — randmech
    // (25 bodies, nv 39), fp32, 16 384 states: dynamics! 90 us against aba_spec's 43, mass_matrix! + Cholesky 173 against 137 for TWICE the batch; inverse
    // dynamics 33 (two bodies per lane) against 31.5, fp64 44 against 32
    w->state_min_batch = (long)ncu * 32;
    { bool has; const long t = tune("state_min_batch", 0, &has); if (has) w->state_min_batch = t; }
    w->mass_min_batch = w->mass_solve_min_batch = tune("mass_min_batch", w->state_min_batch);
    // (round 6, scripts/sweep_routes.py, randmech(): fp64 4096 states 47.6 one body per lane against 47.8 compiled, 8192: 91.7 against 48.0 — from 4097; fp32 4096:
    //  26.8 against 34.1, 8192: 50.1 against 34.3 — from 5377.  Until then both waited for 8193)
    w->spec_aba_min_batch = (long)ncu * (dtype == RBD_F64 ? 16 : 21) + 1;
    w->spec_rnea_min_batch = (long)ncu * (dtype == RBD_F64 ? 40 : 52);  // (randmech(), fp64: 8192 states 26.6 lane-per-body against 31.2 compiled, 16 384: 45.6 against 32.1; fp32 18.5 / 26.8 and 32.7 / 27.5)
    { bool has; const long t = tune("spec_aba_min_batch", 0, &has); if (has) w->spec_aba_min_batch = t; }
    { bool has; const long t = tune("spec_rnea_min_batch", 0, &has); if (has) w->spec_rnea_min_batch = t; }
    w->spec_kin_min_batch = tune("spec_kin_min_batch", (long)ncu * 32 + 1);  // (randmech(), the set of three calls: fp64 8192 states 79.6 compiled against 86.6, 12 288: 88.7 against 110; fp32 63.3 / 67.2 and 72.1 / 91.6)
    w->spec_f64_max_scratch = (int)tune("spec_f64_max_scratch", 2048);
    // randmech() (25 bodies, nv 39), one round: 50 us with every row in LDS, 85 us with the stash (with a wrench on every body: 98 and 112) — see run_aba
    w->spec_f64_stash = (int)tune("spec_f64_stash", -1);
    w->spec_f64_stash_ratio = (int)tune("spec_f64_stash_ratio", 170);
    w->spec_f64_stash_ratio_fext = (int)tune("spec_f64_stash_ratio_fext", 120);
    w->spec_ncu = ncu;
  } else {
    w->state_min_batch = (long)1 << 62;
  }
  w->spec_first_use_check = tune("first_use_check", 1) != 0;
  w->spec_first_use_inject = tune("first_use_inject", 0) != 0;  // (tests: every check finds a difference — the drop-and-recompute path of each route without a wrong program)
  {
    const hipError_t e = dtype == RBD_F64 ? configure_bank_kernels<double>() : configure_bank_kernels<float>();
    if (e != hipSuccess) { g_last_hip_error = std::string("configure_bank_kernels: ") + hipGetErrorString(e); rbd_workspace_destroy(w); return RBD_ERR_HIP; }
  }
  dm.debug_stop = 0;
  w->no_reroot = tune("walk_no_reroot", 0) != 0; w->loop_no_fused = tune("loop_no_fused", 0) != 0; w->spec_max_scratch = (int)tune("spec_max_scratch", 512);
  *out = w;
  return RBD_OK;
}

int rbd_workspace_destroy(rbd_ws_t* w) {
  if (!w) return RBD_OK;
  (void)hipSetDevice(w->device);
  void* ptrs[] = {w->d_big_L, w->d_big_tbl, w->d_big_rb, w->d_big_scratch, w->d_fused_i, w->d_tauwork, w->d_rr_chain_i, w->d_rr_chain_r, w->d_rrtrack_ri, w->d_rrtrack_rr, w->d_rrwalk_wk, w->d_cp_body, w->d_cp_r, w->d_hs_r, w->d_tw, w->d_cw, w->d_s0, w->d_sacc, w->d_sdot, w->d_rows, w->d_walk_wk, w->d_state_ops, w->d_state_cols, w->d_state_sr, w->d_Msoa, w->d_track_ri, w->d_track_rr, w->d_bank_ib[0], w->d_bank_ib[1], w->d_bank_rb[0], w->d_bank_rb[1], w->d_ib, w->d_rb, w->d_nslots, w->d_dof_body, w->d_anc, w->d_row_mask, w->d_M, w->d_c, w->d_K, w->d_k, (void*)w->d_notpd, w->d_body, w->d_scratch, w->d_loop_i, w->d_loop_r, w->d_loop_path, w->d_jt_ref, w->d_voff_ref, w->d_axis_ref, w->d_axis2_ref};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (void* p : w->stage) if (p) (void)hipFree(p);
  {
    void* mkp[] = {w->mk.q0, w->mk.v0, w->mk.phid[0], w->mk.phid[1], w->mk.phid[2], w->mk.phid[3], w->mk.vd[0], w->mk.vd[1], w->mk.vd[2], w->mk.vd[3], w->d_vdwork};
    for (void* p : mkp) if (p) (void)hipFree(p);
  }
  for (hipModule_t mod : w->spec_mod) if (mod) (void)hipModuleUnload(mod);
  if (w->spec_loop_mod) (void)hipModuleUnload(w->spec_loop_mod);
  if (w->spec_bank_mod) (void)hipModuleUnload(w->spec_bank_mod);
  for (int k = 0; k < 12; ++k) if (w->spec_walk_mod[k]) (void)hipModuleUnload(w->spec_walk_mod[k]);
  if (w->ev0) (void)hipEventDestroy(w->ev0);
  if (w->ev1) (void)hipEventDestroy(w->ev1);
  delete w;
  return RBD_OK;
}

int rbd_workspace_set_stream(rbd_ws_t* w, void* stream) {
  if (!w) return RBD_ERR_INVALID_ARGUMENT;
  w->stream = (hipStream_t)stream;
  return RBD_OK;
}

// stabilization_gains of the calls that follow (src/mechanism_algorithms.jl:614-632, :848): the four gains of every loop joint's record in THIS workspace's copy
// of the loop tables are rewritten (the kernels that interpret the tables read them there; the kernel compiled against constant tables is handed the
// pointer).  A call with the gains already in force returns without touching the device, so a host mirror may call it before every dynamics!.
int rbd_workspace_set_loop_gains(rbd_ws_t* w, const double* gains) {
  if (!w) return RBD_ERR_INVALID_ARGUMENT;
  const rbd_model* m = w->model;
  if (m->nloops == 0) return RBD_OK;  // nothing to stabilize (the reference ignores the keyword for tree mechanisms)
  std::vector<double> want(4 * (size_t)m->nloops);
  bool custom = false;
  for (int l = 0; l < m->nloops; ++l)
    for (int k = 0; k < 4; ++k) {
      const double d = m->loop_r[64 * (size_t)l + 24 + k];
      const double g = gains ? gains[4 * l + k] : d;
      if (!(g == g) || g - g != 0.0) return RBD_ERR_INVALID_ARGUMENT;  // NaN / Inf gains
      want[4 * (size_t)l + k] = g;
      custom = custom || g != d;
    }
  if (w->loop_gains.empty() && !custom) return RBD_OK;  // still the model's
  if (want == w->loop_gains) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipStreamSynchronize(w->stream));  // launches in flight read the records
  const size_t es = w->dtype == RBD_F64 ? sizeof(double) : sizeof(float);
  for (int l = 0; l < m->nloops; ++l) {
    double g64[4]; float g32[4];
    for (int k = 0; k < 4; ++k) { g64[k] = want[4 * (size_t)l + k]; g32[k] = (float)g64[k]; }
    HIP_TRY(hipMemcpy((char*)w->d_loop_r + es * (64 * (size_t)l + 24), w->dtype == RBD_F64 ? (const void*)g64 : (const void*)g32, 4 * es, hipMemcpyHostToDevice));
  }
  w->loop_gains = want;
  w->custom_gains = custom;
  return RBD_OK;
}

int rbd_workspace_bind_result(rbd_ws_t* w, void* M, void* c) {
  if (!w) return RBD_ERR_INVALID_ARGUMENT;
  // the kernels write through these pointers: they must be device memory (a host pointer here used to reach the kernels unchecked)
  for (void* p : {M, c}) {
    if (!p) continue;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return RBD_ERR_INVALID_ARGUMENT; }
    if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeManaged && a.type != hipMemoryTypeUnified) return RBD_ERR_INVALID_ARGUMENT;
  }
  w->bound_M = M;
  w->bound_c = c;
  return RBD_OK;
}

int rbd_sync(rbd_ws_t* w) {
  if (!w) return RBD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipStreamSynchronize(w->stream));
  int flag = 0;
  HIP_TRY(hipMemcpy(&flag, w->d_notpd, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) {
    HIP_TRY(hipMemset(w->d_notpd, 0, sizeof(int)));
    return RBD_ERR_NOT_POSITIVE_DEFINITE;  // LAPACK.potrf! would have thrown PosDefException
  }
  return RBD_OK;
}

int rbd_workspace_enable_timing(rbd_ws_t* w, int32_t enable) {
  if (!w) return RBD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(w->device));
  if (enable && !w->ev0) {
    HIP_TRY(hipEventCreate(&w->ev0));
    HIP_TRY(hipEventCreate(&w->ev1));
  }
  w->timing = enable ? 1 : 0;
  return RBD_OK;
}

const char* rbd_workspace_last_kernel(const rbd_ws_t* w) { return w ? w->last_kernel : ""; }

int rbd_workspace_last_kernel_ms(rbd_ws_t* w, float* ms) {
  if (!w || !ms || !w->ev_pending) return RBD_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipEventSynchronize(w->ev1));
  HIP_TRY(hipEventElapsedTime(ms, w->ev0, w->ev1));
  return RBD_OK;
}

}  // extern "C"

// ---- dispatch helpers -------------------------------------------------------------------------
namespace {

size_t esize(const rbd_ws* w) { return w->dtype == RBD_F64 ? 8 : 4; }

Layout layout_of(int layout, long n, long B) {
  Layout L;
  if (layout == RBD_LAYOUT_AOS) { L.sk = 1; L.sb = n; } else { L.sk = B; L.sb = 1; }
  return L;
}

// a buffer with n scalars per state may only be missing when n == 0 (a mechanism whose tree joints are all Fixed has nq = nv = 0)
inline bool missing(const void* p, long n) { return p == nullptr && n > 0; }

struct Opts { int layout, memory, algorithm, stabilization; };
Opts read_opts(const rbd_opts_t* o) {
  Opts r{RBD_LAYOUT_SOA, RBD_MEM_DEVICE, RBD_ALGO_ABA, 1};
  if (o) { r.layout = o->layout; r.memory = o->memory; r.algorithm = o->algorithm; r.stabilization = o->stabilization; }
  return r;
}

// a model of more than 64 bodies is taken by the entry points that declare a BigOk (the four hot-path functions and the solve); all others refuse it
thread_local int g_big_ok = 0;
struct BigOk { BigOk() { ++g_big_ok; } ~BigOk() { --g_big_ok; } };
int check_common(rbd_ws* w, int32_t B, const Opts& o) {
  if (!w) return RBD_ERR_INVALID_ARGUMENT;
  if (w->model->big && g_big_ok == 0) return RBD_ERR_UNSUPPORTED;
  if (B < 0 || B > w->max_batch) return RBD_ERR_DIMENSION_MISMATCH;
  if (o.layout != RBD_LAYOUT_SOA && o.layout != RBD_LAYOUT_AOS) return RBD_ERR_INVALID_ARGUMENT;
  if (o.memory != RBD_MEM_DEVICE && o.memory != RBD_MEM_HOST) return RBD_ERR_INVALID_ARGUMENT;
  return RBD_OK;
}

// host-memory mode: copy `src` (host) into staging slot `slot`; returns device pointer via *dev
int stage_in(rbd_ws* w, int slot, const void* src, size_t bytes, const void** dev) {
  if (!src) { *dev = nullptr; return RBD_OK; }
  if (w->stage_bytes[slot] < bytes) {
    if (w->stage[slot]) HIP_TRY(hipFree(w->stage[slot]));
    w->stage[slot] = nullptr; w->stage_bytes[slot] = 0;
    HIP_TRY(hipMalloc(&w->stage[slot], bytes));
    w->stage_bytes[slot] = bytes;
  }
  HIP_TRY(hipMemcpyAsync(w->stage[slot], src, bytes, hipMemcpyHostToDevice, w->stream));
  *dev = w->stage[slot];
  return RBD_OK;
}
int stage_out_alloc(rbd_ws* w, int slot, void* dst, size_t bytes, void** dev) {
  if (!dst) { *dev = nullptr; return RBD_OK; }
  if (w->stage_bytes[slot] < bytes) {
    if (w->stage[slot]) HIP_TRY(hipFree(w->stage[slot]));
    w->stage[slot] = nullptr; w->stage_bytes[slot] = 0;
    HIP_TRY(hipMalloc(&w->stage[slot], bytes));
    w->stage_bytes[slot] = bytes;
  }
  *dev = w->stage[slot];
  return RBD_OK;
}
int stage_out_copy(rbd_ws* w, void* dst, const void* dev, size_t bytes) {
  if (!dst) return RBD_OK;
  HIP_TRY(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, w->stream));
  return RBD_OK;
}

struct Timed {
  rbd_ws* w;
  explicit Timed(rbd_ws* w_) : w(w_) { if (w->timing) (void)hipEventRecord(w->ev0, w->stream); }
  ~Timed() { if (w->timing) { (void)hipEventRecord(w->ev1, w->stream); w->ev_pending = true; } }
};

int ensure(void** p, size_t* have, size_t need) {
  if (*have >= need) return RBD_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr; *have = 0;
  HIP_TRY(hipMalloc(p, need));
  *have = need;
  return RBD_OK;
}

}  // namespace

// the two-bodies-per-lane kernels: every tree joint revolute, apart from 6-dof joints on the world -> their SIMPLE instantiation
static int bank_simple(const rbd_model* m) {
  for (int i = 0; i < m->nb; ++i)
    if (m->jt_ref[i] != RBD_JOINT_REVOLUTE && !(m->jt_ref[i] == RBD_JOINT_QUAT_FLOATING && m->parent_ref[i] < 0)) return 0;
  return 1;
}
// ... and their program for this mechanism (rbd_jit.hip spec_bank_source): empty when the banked mapping does not apply
static std::string bank_program_source(const rbd_model* m, int dtype, int simple) {
  if (m->bank_lps <= 0) return std::string();
  return spec_bank_source(m->nlevels, m->bank_L0, m->nslots.data(), (unsigned long long)m->bank_perm_down, simple, dtype);
}
// the loop tables of a small loop mechanism as rbd_jit.hip's generator takes them (xi as rbd_workspace_create uploads it for loop_fused_small_kernel)
static std::string loop_program_source(const rbd_model* m, int dtype, std::vector<int32_t>* xi_store) {
  if (!m->loop_fused_ok) return std::string();
  xi_store->assign(3 * (size_t)m->nb, 0);
  for (int i = 0; i < m->nb; ++i) { (*xi_store)[3 * i] = m->parent_ref[i]; (*xi_store)[3 * i + 1] = m->qoff_ref[i]; (*xi_store)[3 * i + 2] = m->slot_of[i]; }
  LoopTables L{m->nb, m->nq, m->nv, m->nc, m->nloops, &m->loop_i, &m->loop_path, &m->jt_ref, &m->voff_ref, xi_store, &m->loop_r, &m->axis_ref, &m->axis2_ref, &m->rb, m->gravity};
  return spec_loop_source(L, dtype);
}
// the walk plan of a tree (the original one, or the one re-rooted at its centre) as rbd_jit.hip's generator takes it
static bool walk_tables(const rbd_model* m, bool rerooted, WalkTables* W) {
  const TrackPlan* P; const WalkPlan* K;
  if (rerooted) { if (!m->rrs.ok || !m->rrs.track.ok || !m->rrs.walk.ok) return false; P = &m->rrs.track; K = &m->rrs.walk; }
  else { if (!m->track.ok || !m->walk.ok) return false; P = &m->track; K = &m->walk; }
  *W = WalkTables{};
  W->ns = P->ns; W->G = P->G; W->nA = P->nA; W->nB = P->nB; W->nS = K->nS; W->nq = m->nq; W->nv = m->nv; W->flt = P->has_floating; W->gen = P->general; W->rr = rerooted;
  W->ri = &P->ri; W->wk = &K->wk; W->rrc = &P->rr;
  W->mk1 = &m->mk1; W->mkf = &m->mkf;
  for (int k = 0; k < 5; ++k) { W->sfm[k] = 0; for (int s2 = 0; s2 < P->ns; ++s2) W->sfm[k] |= (uint64_t)((P->sf[s2] >> k) & 1) << s2; }
  if (rerooted) {
    W->nchain = (int)(m->rr.chain_i.size() / RC_I_STRIDE); W->fq = m->rr.fq; W->fv = m->rr.fv; W->chain_i = &m->rr.chain_i; W->chain_r = &m->rr.chain_r; W->fXp = m->rr.fXp;
    if (W->nchain < 1) return false;
  }
  return true;
}
// (the plan rbd_dynamics picks for the walk kernel: the re-rooted tree when there is one and its rows fit the LDS)
static bool walk_program_rerooted(const rbd_model* m, int dtype, int pair) {
  WalkTables W;
  return tune("walk_no_reroot", 0) == 0 && walk_tables(m, true, &W) && walk_lds_bytes(W.ns, W.G, W.nq, W.nv, W.nA, W.nB, W.nS, (dtype == RBD_F64 || pair) ? 8 : 4, dtype == RBD_F64 ? 8 : 4) <= 160 * 1024;
}
static std::string walk_program_source(const rbd_model* m, int dtype, bool rerooted, int kind, int pair) {
  WalkTables W;
  if (!walk_tables(m, rerooted, &W)) return std::string();
  return walk_spec_source(W, dtype, kind, pair);
}
static bool capturing(rbd_ws* w);
// aba_walk_kernel compiled for the mechanism (aba_walk_spec of rbd_walk.hpp): nullptr when unavailable (or while it is being compiled)
static hipFunction_t spec_walk(rbd_ws* w, bool rerooted, int kind = 0, int pair = 0) {
  const int k = 4 * kind + (rerooted ? 2 : 0) + (pair ? 1 : 0);  // kind 0: dynamics!, 1: inverse dynamics, 2: dynamics! with the four stages of a `simulate` step in one launch
  if (w->spec_walk_tried[k]) return w->spec_walk[k];
  if (!jit_available()) { w->spec_walk_tried[k] = true; return nullptr; }
  if (capturing(w)) return nullptr;
  std::string& src = w->spec_walk_src[k];
  if (src.empty()) src = walk_program_source(w->model, w->dtype, rerooted, kind, pair);
  if (src.empty()) { w->spec_walk_tried[k] = true; return nullptr; }
  std::string log;
  std::vector<char> code;
  const int js = jit_walk_code_object_get(src, !jit_async(), &code, &log);
  if (js == JIT_PENDING) return nullptr;  // being compiled on a background thread: the interpreting walk kernel meanwhile
  w->spec_walk_tried[k] = true;
  if (js == JIT_FAILED || code.empty()) { g_last_hip_error = "run-time compilation failed (the interpreting kernel is used): " + log; src.clear(); src.shrink_to_fit(); return nullptr; }
  if (hipModuleLoadData(&w->spec_walk_mod[k], code.data()) != hipSuccess) { (void)hipGetLastError(); w->spec_walk_mod[k] = nullptr; jit_cache_discard(src); return nullptr; }
  const std::string fname = std::string(kind == 1 ? "rnea_walk_spec_" : kind == 2 ? "aba_walk_sim_spec_" : "aba_walk_spec_") + walk_spec_suffix(w->dtype, pair);
  if (hipModuleGetFunction(&w->spec_walk[k], w->spec_walk_mod[k], fname.c_str()) != hipSuccess) { (void)hipGetLastError(); w->spec_walk[k] = nullptr; }
  int scratch = 0;
  const int max_scratch = w->spec_max_scratch;  // bytes per lane
  if (w->spec_walk[k] && (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, w->spec_walk[k]) != hipSuccess || scratch > max_scratch)) { (void)hipGetLastError(); w->spec_walk[k] = nullptr; }
  src.clear(); src.shrink_to_fit();
  return w->spec_walk[k];
}
// scratch of the any-size kernels (rbd_big_kernels.hip)
static int big_scratch(rbd_ws* w, int32_t B) { return ensure(&w->d_big_scratch, &w->d_big_scratch_bytes, esize(w) * big_scratch_elems(w->big, B)); }
// small loop mechanisms compiled for the mechanism (rbd_loop_small.hpp against constant tables): nullptr when unavailable
static hipFunction_t spec_loop(rbd_ws* w) {
  if (w->spec_loop_tried) return w->spec_loop;
  if (!jit_available()) { w->spec_loop_tried = true; return nullptr; }
  if (capturing(w)) return nullptr;
  std::vector<int32_t> xi;
  std::string& src = w->spec_loop_src;
  if (src.empty()) src = loop_program_source(w->model, w->dtype, &xi);
  if (src.empty()) { w->spec_loop_tried = true; return nullptr; }
  std::string log;
  std::vector<char> code;
  const int js = jit_code_object_get(src, !jit_async(), &code, &log);
  if (js == JIT_PENDING) return nullptr;  // the generic loop kernels meanwhile
  w->spec_loop_tried = true;
  if (js == JIT_FAILED || code.empty()) { g_last_hip_error = "run-time compilation failed (the generic loop kernels are used): " + log; return nullptr; }
  if (hipModuleLoadData(&w->spec_loop_mod, code.data()) != hipSuccess) { (void)hipGetLastError(); w->spec_loop_mod = nullptr; jit_cache_discard(src); return nullptr; }
  if (hipModuleGetFunction(&w->spec_loop, w->spec_loop_mod, w->dtype == RBD_F64 ? "loop_spec_f64" : "loop_spec_f32") != hipSuccess) { (void)hipGetLastError(); w->spec_loop = nullptr; }
  int scratch = 0;
  if (w->spec_loop && (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, w->spec_loop) != hipSuccess || scratch > 0)) { (void)hipGetLastError(); w->spec_loop = nullptr; }
  src.clear(); src.shrink_to_fit();
  return w->spec_loop;
}

// the two-bodies-per-lane kernels compiled for the mechanism (rbd_bank.hpp with the level structure as constants); false while unavailable
static bool spec_bank(rbd_ws* w) {
  if (w->spec_bank_tried) return w->spec_bank_aba != nullptr;
  if (!jit_available() || w->model->bank_lps <= 0) { w->spec_bank_tried = true; return false; }
  if (capturing(w)) return false;
  std::string& src = w->spec_bank_src;
  if (src.empty()) src = bank_program_source(w->model, w->dtype, w->bm.simple);
  if (src.empty()) { w->spec_bank_tried = true; return false; }
  std::string log;
  std::vector<char> code;
  const int js = jit_code_object_get(src, !jit_async(), &code, &log);
  if (js == JIT_PENDING) return false;  // the kernels built with the library meanwhile
  w->spec_bank_tried = true;
  if (js == JIT_FAILED || code.empty()) { g_last_hip_error = "run-time compilation failed (the banked kernels built with the library are used): " + log; return false; }
  if (hipModuleLoadData(&w->spec_bank_mod, code.data()) != hipSuccess) { (void)hipGetLastError(); w->spec_bank_mod = nullptr; jit_cache_discard(src); return false; }
  const char* sfx = w->dtype == RBD_F64 ? "f64" : "f32";
  auto get = [&](hipFunction_t* f, const std::string& name) {
    int scratch = 0;
    if (hipModuleGetFunction(f, w->spec_bank_mod, name.c_str()) != hipSuccess) { (void)hipGetLastError(); *f = nullptr; }
    else if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, *f) != hipSuccess || scratch > w->spec_max_scratch) { (void)hipGetLastError(); *f = nullptr; }
  };
  get(&w->spec_bank_aba, std::string("aba_bank_spec_") + sfx);
  get(&w->spec_bank_fused, std::string("aba_bank_fused_spec_") + sfx);
  get(&w->spec_bank_rnea, std::string("rnea_bank_spec_") + sfx);
  src.clear(); src.shrink_to_fit();
  return w->spec_bank_aba != nullptr;
}

namespace {
template <typename T>
int dynamics_loops_t(rbd_ws* w, int32_t B, const Opts& o, const void* dq, const void* dv, const void* dtau, const void* df, void* dvd, void* dqd,
                     void* dlam) {
  const rbd_model* m = w->model;
  const size_t es = sizeof(T);
  const int nv = m->nv, nc = m->nc;
  const long stride = (long)nv * nv + 2L * nc * nv + 2L * nc * nc + 2L * nv + 2L * nc;
  int st;
  if ((st = ensure(&w->d_M, &w->d_M_bytes, es * (size_t)nv * nv * B)) || (st = ensure(&w->d_c, &w->d_c_bytes, es * (size_t)nv * B)) ||
      (st = ensure(&w->d_K, &w->d_K_bytes, es * (size_t)(nc * nv > 0 ? nc * nv : 1) * B)) || (st = ensure(&w->d_k, &w->d_k_bytes, es * (size_t)(nc > 0 ? nc : 1) * B)) ||
      (st = ensure(&w->d_body, &w->d_body_bytes, es * (size_t)m->nb * 24 * B)) || (st = ensure(&w->d_scratch, &w->d_scratch_bytes, es * (size_t)stride * B)))
    return st;
  w->result_layout = o.layout; w->result_B = B;
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, nv, B), Lf = layout_of(o.layout, 6L * m->nb, B);
  const Layout Lm = layout_of(o.layout, (long)nv * nv, B), Lc = layout_of(o.layout, nc, B), Lk = layout_of(o.layout, (long)nc * nv, B);
  LoopView<T> V;
  V.nloops = m->nloops; V.nc = nc; V.nv = nv; V.nb = m->nb;
  V.li = (const int32_t*)w->d_loop_i; V.lr = (const T*)w->d_loop_r; V.path = (const int32_t*)w->d_loop_path;
  V.jt = (const int32_t*)w->d_jt_ref; V.voff = (const int32_t*)w->d_voff_ref; V.axis = (const T*)w->d_axis_ref; V.axis2 = (const T*)w->d_axis2_ref;
  V.xi = (const int32_t*)w->d_fused_i; V.rb = (const T*)w->d_rb;
  Timed t(w);
  const bool no_fused = w->loop_no_fused;  // tests: the three-launch route on a mechanism the fused kernel would take
  if (hipFunction_t f = (m->loop_fused_ok && !no_fused) ? spec_loop(w) : nullptr) {  // the whole evaluation as straight-line code for this mechanism
    long Bl = B;
    int stab = o.stabilization;
    void* dM = w->d_M; void* dc = w->d_c; void* dK = w->d_K; void* dk = w->d_k; int* notpd = w->d_notpd;
    Layout a_Lq = Lq, a_Lm = Lm, a_Lv = Lv, a_Lf = Lf, a_Lc = Lc, a_Lk = Lk;
    const T* gains = w->custom_gains ? (const T*)w->d_loop_r + 24 : nullptr;  // (nullptr: the model's gains, constants of the compiled code)
    void* args[] = {&Bl, &stab, &dq, &dv, &dtau, &df, &dM, &dc, &dvd, &dqd, &dlam, &dK, &dk, &a_Lq, &a_Lm, &a_Lv, &a_Lf, &a_Lc, &a_Lk, &notpd, &gains};
    HIP_TRY(hipModuleLaunchKernel(f, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, w->stream, args, nullptr));
    w->last_kernel = "loop_spec (compiled for the mechanism at run time)";
    return RBD_OK;
  }
  if (m->loop_fused_ok && !no_fused &&
      launch_loop_fused<T>(V, B, o.stabilization, dq, dv, dtau, df, w->d_body, w->d_M, w->d_c, dvd, dqd, dlam, w->d_K, w->d_k, Lq, Lm, Lv, Lf, Lc, Lk, m->gravity,
                           w->d_notpd, w->stream)) {
    HIP_TRY(hipGetLastError());
    w->last_kernel = "loop_fused_small_kernel";
    return RBD_OK;
  }
  if (m->big) {  // more than 64 bodies: bias forces + per-body kinematics and the mass matrix from the any-size kernels, then the same constrained solve
    if ((st = big_scratch(w, B))) return st;
    HIP_TRY(launch_big_rnea<T>(w->big, B, dq, dv, nullptr, df, w->d_c, dqd, w->d_big_scratch, nullptr, nullptr, Lq, Lv, Lf, w->stream));
    HIP_TRY(launch_big_export_body<T>(w->big, B, w->d_big_scratch, w->d_body, w->stream));
    HIP_TRY(launch_big_crba<T>(w->big, B, dq, w->d_M, w->d_big_scratch, Lq, Lm, w->stream));
    HIP_TRY(launch_loop_solve<T>(V, B, o.stabilization, w->d_body, w->d_M, w->d_c, dtau, dvd, dlam, w->d_K, w->d_k, w->d_scratch, stride, Lm, Lv, Lc, Lk,
                                 m->gravity, w->d_notpd, w->stream));
    w->last_kernel = "big_rnea + big_crba + loop_solve kernel";
    return RBD_OK;
  }
  HIP_TRY(launch_rnea<T>(w->dm, B, dq, dv, nullptr, df, w->d_c, dqd, w->d_body, Lq, Lv, Lf, w->stream));
  HIP_TRY(launch_crba<T>(w->dm, B, dq, w->d_M, Lq, Lm, 1, w->stream));
  HIP_TRY(launch_loop_solve<T>(V, B, o.stabilization, w->d_body, w->d_M, w->d_c, dtau, dvd, dlam, w->d_K, w->d_k, w->d_scratch, stride, Lm, Lv, Lc, Lk,
                               m->gravity, w->d_notpd, w->stream));
  w->last_kernel = "rnea_kernel + crba_kernel + loop_solve kernel";
  return RBD_OK;
}
int dynamics_loops(rbd_ws* w, int32_t B, const Opts& o, const void* dq, const void* dv, const void* dtau, const void* df, void* dvd, void* dqd, void* dlam) {
  return w->dtype == RBD_F64 ? dynamics_loops_t<double>(w, B, o, dq, dv, dtau, df, dvd, dqd, dlam)
                             : dynamics_loops_t<float>(w, B, o, dq, dv, dtau, df, dvd, dqd, dlam);
}
}  // namespace


// inverse_dynamics! / dynamics_bias! (vdot == nullptr) through the lane mapping that fits the batch: same rule as run_aba
static void spec_load(rbd_ws* w, int family, bool force = false);  // the kernels compiled for the mechanism (below)

// The first result of a run-time compiled dynamics! program against the interpreting one-body-per-lane kernel (aba_kernel) on the first states of the same call
// (up to 256): max |difference| <= tol max(1, max |reference|), tol 1e-7 in fp64, 5e-3 in fp32 (two fp32 evaluations in different operation orders).  One
// allocation, two small launches and a synchronisation — once per program and workspace, on a call that has just waited for the module to load.
static bool capturing(rbd_ws* w);
static int first_use_check(rbd_ws* w, long B, const void* dq, const void* dv, const void* dtau, const void* df, const void* dvd, Layout Lq, Layout Lv, Layout Lf,
                           const double* gravity, bool* same, bool inverse = false) {  // inverse: inverse_dynamics! — `dtau` is v̇ (input), `dvd` the torques (output), against rnea_kernel
  const rbd_model* m = w->model;
  const long n = std::min<long>(B, 256);
  const size_t es = w->dtype == RBD_F64 ? 8 : 4;
  const size_t elems = (size_t)(layout_base(Lv, n - 1) + (long)(m->nv - 1) * Lv.sk + 1);  // v̇'s layout, its first n states
  void* ref = nullptr;
  double* out = nullptr;
  HIP_TRY(hipMalloc(&ref, elems * es));
  if (hipMalloc((void**)&out, 2 * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ref); return RBD_ERR_OUT_OF_MEMORY; }
  DevModel dm = w->dm;
  if (gravity) memcpy(dm.gravity, gravity, sizeof dm.gravity);
  double h[2] = {0, 0};
  hipError_t e = inverse ? (w->dtype == RBD_F64 ? launch_rnea<double>(dm, n, dq, dv, dtau, df, ref, nullptr, nullptr, Lq, Lv, Lf, w->stream)
                                                : launch_rnea<float>(dm, n, dq, dv, dtau, df, ref, nullptr, nullptr, Lq, Lv, Lf, w->stream))
                 : w->dtype == RBD_F64 ? launch_aba<double>(dm, n, dq, dv, dtau, df, ref, nullptr, Lq, Lv, Lf, w->stream)
                                       : launch_aba<float>(dm, n, dq, dv, dtau, df, ref, nullptr, Lq, Lv, Lf, w->stream);
  if (e == hipSuccess) e = w->dtype == RBD_F64 ? launch_max_diff<double>(n, m->nv, dvd, ref, Lv, out, w->stream) : launch_max_diff<float>(n, m->nv, dvd, ref, Lv, out, w->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, w->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
  (void)hipFree(ref);
  (void)hipFree(out);
  if (e != hipSuccess) { g_last_hip_error = std::string("first_use_check: ") + hipGetErrorString(e); (void)hipGetLastError(); return RBD_ERR_HIP; }
  w->spec_check_err = h[0] / std::max(1.0, h[1]);
  *same = w->spec_check_err <= (w->dtype == RBD_F64 ? 1e-7 : 5e-3) && !w->spec_first_use_inject;
  if (!*same) {
    char msg[320];
    snprintf(msg, sizeof msg, "%s differs from the interpreting kernel on this call's first states by %.3g of the largest %s%s: the program is dropped, the kernels built with the library serve",
             w->last_kernel, w->spec_check_err, inverse ? "torque" : "acceleration", w->spec_first_use_inject ? " (RBD_TUNE first_use_inject)" : "");
    g_last_hip_error = msg;
    fprintf(stderr, "[rbd] %s\n", msg);
  }
  return RBD_OK;
}

static int run_rnea(rbd_ws* w, int32_t B, int mapping, const void* dq, const void* dv, const void* dvd, const void* df, void* dtau, void* dqd,
                    Layout Lq, Layout Lv, Layout Lf, void* dacc = nullptr, void* djw = nullptr) {
  const rbd_model* m = w->model;
  if (m->big) {
    int st = big_scratch(w, B);
    if (st) return st;
    if (w->dtype == RBD_F64) HIP_TRY(launch_big_rnea<double>(w->big, B, dq, dv, dvd, df, dtau, dqd, w->d_big_scratch, dacc, djw, Lq, Lv, Lf, w->stream));
    else HIP_TRY(launch_big_rnea<float>(w->big, B, dq, dv, dvd, df, dtau, dqd, w->d_big_scratch, dacc, djw, Lq, Lv, Lf, w->stream));
    return RBD_OK;
  }
  if (mapping == RBD_ALGO_ABA_BANKS && m->bank_lps == 0) return RBD_ERR_UNSUPPORTED;
  // the per-body outputs (accelerations, joint wrenches) are written by the lane-per-body kernels (one or two bodies per lane)
  const bool banks = m->bank_lps > 0 && (mapping == RBD_ALGO_ABA_BANKS || (mapping != RBD_ALGO_ABA_LANES && B >= w->rnea_bank_min_batch));
  const bool can_walk = m->track.ok && m->walk.ok && (w->walk_lds_bytes > 0 || (w->walk_lds_bytes_pair > 0 && B >= w->walk_pair_min_batch));
  if (mapping == RBD_ALGO_ABA_WALK && !can_walk) return RBD_ERR_UNSUPPORTED;
  // the kernel compiled for the mechanism: large batches (q̇ is not one of its outputs).  Per-body outputs: a lane storing its own state-major row writes
  // 24-byte pieces (65 536 fp32 states: 120 us against 42 without them, 51 into batch-innermost buffers), so for a state-major caller the fp32 kernel
  // stores into batch-innermost scratch and rows_to_state_major_kernel moves it; fp64 state-major stays with the walk kernel under RBD_ALGO_ABA
  // (118 us against 146), fp64 batch-innermost comes here.
  const bool bodies = dacc || djw, rows_out = Lf.sb == 1;
  if (!dqd && (mapping == RBD_ALGO_ABA_COMPILED || (mapping == RBD_ALGO_ABA && !(bodies && w->dtype == RBD_F64 && !rows_out)))) {
    if (mapping == RBD_ALGO_ABA_COMPILED || B >= w->spec_rnea_min_batch) spec_load(w, SPEC_RNEA, mapping == RBD_ALGO_ABA_COMPILED);
    if (w->spec_rnea && (mapping == RBD_ALGO_ABA_COMPILED || (B >= w->spec_rnea_min_batch && w->spec_rnea_scratch == 0))) {
      long Bl = B;
      const long ld = (long)B + 64;  // scratch rows 256 bytes past a power of two apart
      const size_t es = w->dtype == RBD_F64 ? 8 : 4, each = es * 6 * (size_t)m->nb * (size_t)ld;
      const bool staged = bodies && !rows_out && w->dtype == RBD_F32;
      Layout Lo = Lf;
      void *oacc = dacc, *ojw = djw;
      if (staged) {
        if (int st = ensure(&w->d_rows, &w->d_rows_bytes, 2 * each)) return st;
        Lo = Layout{ld, 1};
        if (dacc) oacc = w->d_rows;
        if (djw) ojw = (char*)w->d_rows + each;
      }
      void* args[] = {&Bl, &dq, &dv, &dvd, &df, &dtau, &Lq, &Lv, &Lf, &oacc, &ojw, &Lo};
      HIP_TRY(hipModuleLaunchKernel(w->spec_rnea, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, w->stream, args, nullptr));
      if (staged) {
        if (dacc && djw) HIP_TRY(launch_rows_to_state_major<float>(6 * m->nb, B, ld, oacc, dacc, ojw, djw, w->stream));
        else HIP_TRY(launch_rows_to_state_major<float>(6 * m->nb, B, ld, dacc ? oacc : ojw, dacc ? dacc : djw, nullptr, nullptr, w->stream));
      }
      w->last_kernel = w->dtype == RBD_F64 ? "rnea_spec_f64 (compiled for the mechanism at run time)"
                       : staged            ? "rnea_spec_f32 (compiled for the mechanism at run time) + rows_to_state_major_kernel"
                                           : "rnea_spec_f32 (compiled for the mechanism at run time)";
      if (dvd && dtau && w->spec_first_use_check && !w->spec_rnea_checked && !capturing(w)) {  // (first use of this program by this workspace: first_use_check)
        w->spec_rnea_checked = true;
        bool same = true;
        if (int st = first_use_check(w, B, dq, dv, dvd, df, dtau, Lq, Lv, Lf, nullptr, &same, true)) return st;
        if (!same) {
          w->spec_rnea = nullptr;
          return mapping == RBD_ALGO_ABA_COMPILED ? RBD_ERR_UNSUPPORTED : run_rnea(w, B, mapping, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, dacc, djw);
        }
      }
      return RBD_OK;
    }
  }
  if (mapping == RBD_ALGO_ABA_COMPILED) return RBD_ERR_UNSUPPORTED;
  if (can_walk && (mapping == RBD_ALGO_ABA_WALK || (mapping != RBD_ALGO_ABA_BANKS && mapping != RBD_ALGO_ABA_LANES && B >= w->rnea_walk_min_batch))) {
    // one wavefront per track, one lane per state (rnea_walk_kernel): large batches
    const int pair = w->dtype == RBD_F32 && w->walk_lds_bytes_pair > 0 && B >= w->walk_pair_min_batch;
    if (hipFunction_t f = ((w->dtype == RBD_F64 || w->spec_walk_f32) && B >= w->spec_walk_min_batch) ? spec_walk(w, false, 1, pair) : nullptr) {  // the same kernel compiled for this mechanism
      long Bl = B;
      Layout lq = Lq, lv = Lv, lf = Lf;
      double gx = w->wm.gravity[0], gy = w->wm.gravity[1], gz = w->wm.gravity[2];
      void* args[] = {&Bl, (void*)&dq, (void*)&dv, (void*)&dvd, (void*)&df, (void*)&dtau, (void*)&dqd, &lq, &lv, &lf, (void*)&dacc, (void*)&djw, &gx, &gy, &gz};
      const long per = pair ? 128 : 64;
      HIP_TRY(hipModuleLaunchKernel(f, (unsigned)((B + per - 1) / per), 1, 1, 64u * (unsigned)w->wm.G, 1, 1, 0, w->stream, args, nullptr));
      w->last_kernel = pair ? "rnea_walk_spec (compiled for the mechanism, two fp32 states per lane)" : "rnea_walk_spec (compiled for the mechanism)";
      const int k = 4 + (pair ? 1 : 0);
      if (dvd && dtau && w->spec_first_use_check && !w->spec_walk_checked[k] && !capturing(w)) {
        w->spec_walk_checked[k] = true;
        bool same = true;
        if (int st = first_use_check(w, B, dq, dv, dvd, df, dtau, Lq, Lv, Lf, nullptr, &same, true)) return st;
        if (!same) { w->spec_walk[k] = nullptr; return run_rnea(w, B, mapping, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, dacc, djw); }
      }
      return RBD_OK;
    }
    w->last_kernel = pair ? "rnea_walk_kernel (two fp32 states per lane)" : "rnea_walk_kernel";
    if (w->dtype == RBD_F64) HIP_TRY(launch_rnea_walk<double>(w->wm, m->track.has_floating, m->track.general, 0, B, w->walk_lds_bytes, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, w->stream, dacc, djw));
    else HIP_TRY(launch_rnea_walk<float>(w->wm, m->track.has_floating, m->track.general, pair, B, pair ? w->walk_lds_bytes_pair : w->walk_lds_bytes, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, w->stream, dacc, djw));
  } else if (w->state_aot && !dacc && !djw && mapping != RBD_ALGO_ABA_BANKS && mapping != RBD_ALGO_ABA_LANES && B >= w->state_min_batch) {  // one lane per state
    w->last_kernel = "rnea_state_kernel";
    if (w->dtype == RBD_F64) HIP_TRY(launch_rnea_state<double>(w->sm, B, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, w->stream));
    else HIP_TRY(launch_rnea_state<float>(w->sm, B, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, w->stream));
  } else if (banks) {
    const int ncol = m->has3dof ? 3 : 1;
    w->last_kernel = "rnea_bank_kernel";
    if (spec_bank(w) && w->spec_bank_rnea) {  // the same kernel compiled against this mechanism's level structure (rbd_jit.hip spec_bank_source)
      BankModel bm = w->bm;
      long Bl = B;
      int nc = ncol;
      const long spw = 64 / bm.lps, waves = (B + spw - 1) / spw;
      void* args[] = {&bm, &Bl, &nc, (void*)&dq, (void*)&dv, (void*)&dvd, (void*)&df, (void*)&dtau, (void*)&dqd, &Lq, &Lv, &Lf, (void*)&dacc, (void*)&djw};
      HIP_TRY(hipModuleLaunchKernel(w->spec_bank_rnea, (unsigned)((waves + 3) / 4), 1, 1, 256, 1, 1, 0, w->stream, args, nullptr));
      w->last_kernel = "rnea_bank_kernel (compiled for the mechanism at run time)";
      if (dvd && dtau && w->spec_first_use_check && !w->spec_bank_rnea_checked && !capturing(w)) {
        w->spec_bank_rnea_checked = true;
        bool same = true;
        if (int st = first_use_check(w, B, dq, dv, dvd, df, dtau, Lq, Lv, Lf, nullptr, &same, true)) return st;
        if (!same) { w->spec_bank_rnea = nullptr; return run_rnea(w, B, mapping, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, dacc, djw); }
      }
    } else
    if (w->dtype == RBD_F64) HIP_TRY(launch_rnea_bank<double>(w->bm, B, ncol, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, w->stream, dacc, djw));
    else HIP_TRY(launch_rnea_bank<float>(w->bm, B, ncol, dq, dv, dvd, df, dtau, dqd, Lq, Lv, Lf, w->stream, dacc, djw));
  } else {
    w->last_kernel = "rnea_kernel";
    if (w->dtype == RBD_F64) HIP_TRY(launch_rnea<double>(w->dm, B, dq, dv, dvd, df, dtau, dqd, nullptr, Lq, Lv, Lf, w->stream, dacc, djw));
    else HIP_TRY(launch_rnea<float>(w->dm, B, dq, dv, dvd, df, dtau, dqd, nullptr, Lq, Lv, Lf, w->stream, dacc, djw));
  }
  return RBD_OK;
}

// The fused articulated-body pass through whichever lane mapping fits: `algorithm` RBD_ALGO_ABA chooses by batch size
// (measured crossovers, profiles/r01_mapping_sweep.txt), the RBD_ALGO_ABA_* values force one.  `gravity` overrides the
// model's (the M^-1 solve runs the pass with g = 0); `fuse` folds a Munthe-Kaas stage into the launch (lanes / banks only).
static const MkStage kNoStage{-1, 0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
// `mk` (simulate_core): the launch is a stage of a Munthe-Kaas step folded into a kernel compiled for the mechanism (rbd_mk_fuse.hpp) — only those kernels take it:
// RBD_ERR_UNSUPPORTED when the batch would go to another kernel (the caller then keeps the stage in its own launches)
static int run_aba(rbd_ws* w, int32_t B, int algorithm, const void* dq, const void* dv, const void* dtau, const void* df, void* dvd, void* dqd,
                   Layout Lq, Layout Lv, Layout Lf, const double* gravity, const MkFuse* fuse, const MkStage* mk = nullptr) {
  const rbd_model* m = w->model;
  const bool can_bank = m->bank_lps > 0 && m->bank_aba_ok;
  // the track kernel addresses its batch buffers with 32-bit byte offsets
  const bool can_walk = m->track.ok && m->walk.ok && (w->walk_lds_bytes > 0 || (w->walk_lds_bytes_pair > 0 && B >= w->walk_pair_min_batch)) && !fuse;
  // (RBD_ALGO_ABA_CHAINS: removed in round 3; RBD_ALGO_ABA_TRACKS, RBD_ALGO_ABA_PIPE: the two round-2 experiments, removed in round 4 — all three lost at every
  // batch size; the values stay reserved)
  if (algorithm == RBD_ALGO_ABA_PIPE || algorithm == RBD_ALGO_ABA_TRACKS || algorithm == RBD_ALGO_ABA_CHAINS) return RBD_ERR_UNSUPPORTED;
  // the kernels that take the integrator's stage write the next stage state over their own q / v inputs (rbd_walk.hpp, rbd_spec.hpp: the stage state is the launch's
  // own block of rows) and do not look at MkStage::q_state / v_state: a caller with stage buffers of its own would have its inputs overwritten (round-5 advice)
  if (mk && mk->stage >= 0 && (mk->q_state != dq || mk->v_state != dv)) return RBD_ERR_INVALID_ARGUMENT;
  if (algorithm == RBD_ALGO_ABA_WALK && !can_walk) return RBD_ERR_UNSUPPORTED;
  if (algorithm == RBD_ALGO_ABA_BANKS && !can_bank) return RBD_ERR_UNSUPPORTED;
  // fp64 (round 6): mechanisms with 3-dof joints / 6-dof joints below the world — what no walk or banked kernel takes — have the lane-per-state program in doubles
  // too (rbd_jit.hip spec_has), the integrator's stage of `simulate` included (the same template: rbd_spec.hpp aba_spec)
  const bool spec_f64 = w->dtype == RBD_F64 && m->state_wide.ok && !m->state.ok;
  if ((algorithm == RBD_ALGO_ABA || algorithm == RBD_ALGO_ABA_COMPILED) && !fuse && (w->dtype == RBD_F32 || spec_f64) && !(mk && mk->stage == 4)) {  // (all four stages in one launch: the walk kernels only)
    // with the integrator stage folded in, the lane-per-state kernel is ahead of the walk kernel earlier than without it (Atlas fp32, RK4 step: 24 576 states
    // 209 against 226 us, 32 768: 220 against 240; 16 384: 201 against 143) — the stage costs this kernel 10 us per launch, the walk kernel 28
    const long spec_from = mk ? std::min<long>(w->spec_aba_min_batch, w->spec_aba_fused_min_batch) : w->spec_aba_min_batch;
    if (algorithm == RBD_ALGO_ABA_COMPILED || B >= spec_from) spec_load(w, SPEC_ABA, algorithm == RBD_ALGO_ABA_COMPILED);
    // without external wrenches: the instantiation that holds no registers for them.  Left to itself (RBD_ALGO_ABA) the library takes a compiled kernel only
    // when it spilled nothing
    const bool nofext = df == nullptr && w->spec_aba_nofext != nullptr;
    hipFunction_t faba = nofext ? w->spec_aba_nofext : w->spec_aba;
    const int faba_scratch = nofext ? w->spec_aba_nofext_scratch : w->spec_aba_scratch;
    // fp64: the program with the spare rows in the HBM stash runs two wavefronts per CU where the one with every row in LDS runs one, on a chain 1.7 times as
    // long (1.2 with wrenches: randmech(), 16 384 states 84 against 50 us, 65 536 states 176 against 200; with wrenches 218 against 393) — the one that needs
    // less time for this batch's rounds
    bool stash_program = false;
    if (spec_f64 && faba) {
      hipFunction_t const fg = nofext ? w->spec_aba_gst_nofext : w->spec_aba_gst;
      const long waves = (B + 63) / 64, r_lds = (waves + w->spec_ncu - 1) / w->spec_ncu, r_gst = (waves + 2 * w->spec_ncu - 1) / (2 * w->spec_ncu);
      const long ratio = nofext ? w->spec_f64_stash_ratio : w->spec_f64_stash_ratio_fext;
      if (fg && (w->spec_f64_stash > 0 || (w->spec_f64_stash < 0 && r_lds * 100 > r_gst * ratio))) { faba = fg; stash_program = true; }
    }
    // (fp64: the program spills by construction — its per-body leave-behind does not fit 512 registers in doubles; it is taken up to spec_f64_max_scratch bytes
    //  per lane because what it replaces is the one-body-per-lane kernel, not a walk kernel)
    if (faba && (algorithm == RBD_ALGO_ABA_COMPILED || (B >= spec_from && (faba_scratch == 0 || (spec_f64 && faba_scratch <= w->spec_f64_max_scratch))))) {
      Timed t(w);
      long Bl = B;
      const double* gv = gravity ? gravity : m->gravity;
      float gx = (float)gv[0], gy = (float)gv[1], gz = (float)gv[2];
      double gxd = gv[0], gyd = gv[1], gzd = gv[2];
      MkStage F = mk ? *mk : kNoStage;
      void* args[] = {&Bl, &dq, &dv, &dtau, &df, &dvd, &dqd, &Lq, &Lv, &Lf, &gx, &gy, &gz, &F};
      void* stash = nullptr;
      if (w->dtype == RBD_F64) {  // the HBM stash of aba_spec_gst_f64 (rbd_spec.hpp aba_spec GST): (nb + 10 n3) values per state, batch-innermost; grown on demand like every workspace buffer
        if (stash_program) {
          if (int st = ensure(&w->d_rows, &w->d_rows_bytes, sizeof(double) * (size_t)(m->nb + 10 * m->spec_plan().n3) * (size_t)B)) return st;
          stash = w->d_rows;
        }
      }
      void* args64[] = {&Bl, &dq, &dv, &dtau, &df, &dvd, &dqd, &Lq, &Lv, &Lf, &gxd, &gyd, &gzd, &F, &stash};
      HIP_TRY(hipModuleLaunchKernel(faba, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, w->stream, w->dtype == RBD_F64 ? args64 : args, nullptr));
      w->last_kernel = stash_program ? "aba_spec_gst_f64 (compiled for the mechanism at run time; spare rows in the HBM stash)"
                       : w->dtype == RBD_F64 ? "aba_spec_f64 (compiled for the mechanism at run time)" : "aba_spec_f32 (compiled for the mechanism at run time)";
      // The first result a workspace gets from one of these programs is held against the interpreting one-body-per-lane kernel on the call's first states: a
      // program that hiprtc miscompiled (round 6 met one in an experiment: profiles/r06_experiments.txt §8) is dropped, loudly, and the call recomputed
      bool& checked = w->spec_aba_checked[(stash_program ? 2 : 0) + (nofext ? 1 : 0)];
      if (mk || !dv || !dvd || !w->spec_first_use_check || checked || capturing(w)) return RBD_OK;
      checked = true;
      bool same = true;
      if (int st = first_use_check(w, B, dq, dv, dtau, df, dvd, Lq, Lv, Lf, gravity, &same)) return st;
      if (same) return RBD_OK;
      (stash_program ? (nofext ? w->spec_aba_gst_nofext : w->spec_aba_gst) : (nofext ? w->spec_aba_nofext : w->spec_aba)) = nullptr;
      if (algorithm == RBD_ALGO_ABA_COMPILED) return RBD_ERR_UNSUPPORTED;
      // (falls through: the call is recomputed below)
    }
  }
  if (algorithm == RBD_ALGO_ABA_COMPILED) return RBD_ERR_UNSUPPORTED;
  int pick = algorithm;
  if (algorithm == RBD_ALGO_ABA) pick = (can_walk && B >= w->walk_min_batch) ? RBD_ALGO_ABA_WALK : (can_bank && B >= w->bank_min_batch) ? RBD_ALGO_ABA_BANKS : RBD_ALGO_ABA_LANES;
  if (algorithm == RBD_ALGO_ABA && pick == RBD_ALGO_ABA_BANKS && can_walk && B >= w->walk_one_round_batch && B >= w->spec_walk_min_batch && (w->dtype == RBD_F64 || w->spec_walk_f32)) {
    // the banked kernel would need a second workgroup per CU: the walk kernel compiled for the mechanism is ahead from here (see walk_one_round_batch)
    const bool no_rr = w->no_reroot;
    if (spec_walk(w, w->walk_rr && !no_rr && w->walk_rr_lds_bytes > 0, 0, 0)) pick = RBD_ALGO_ABA_WALK;
  }
  if (mk && pick != RBD_ALGO_ABA_WALK) return RBD_ERR_UNSUPPORTED;
  Timed t(w);
  w->last_kernel = pick == RBD_ALGO_ABA_BANKS ? "aba_bank_kernel" : "aba_kernel";
  if (pick == RBD_ALGO_ABA_WALK) {
    // the tree re-rooted at its centre (rbd_reroot.hpp) when there is one: fewer steps per track, better balanced tracks (RBD_WALK_NO_REROOT=1: the original tree)
    const bool no_rr = w->no_reroot;
    const int pair = w->dtype == RBD_F32 && w->walk_lds_bytes_pair > 0 && B >= w->walk_pair_min_batch;
    const bool rr = w->walk_rr && !no_rr && (pair ? w->walk_rr_lds_bytes_pair : w->walk_rr_lds_bytes) > 0;
    WalkModel wm = rr ? w->wm_rr : w->wm;
    if (gravity) memcpy(wm.gravity, gravity, sizeof wm.gravity);
    w->last_kernel = pair ? "aba_walk_kernel (two fp32 states per lane)" : "aba_walk_kernel";
    const TrackPlan& TP = rr ? m->rrs.track : m->track;
    const size_t lds = rr ? (pair ? w->walk_rr_lds_bytes_pair : w->walk_rr_lds_bytes) : (pair ? w->walk_lds_bytes_pair : w->walk_lds_bytes);
    const int wkind = (mk && mk->stage == 4) ? 2 : 0;  // (all four stages of a `simulate` step in one launch: the instantiation with the passes inside a loop)
    if (hipFunction_t f = ((w->dtype == RBD_F64 || w->spec_walk_f32) && (B >= w->spec_walk_min_batch || wkind == 2)) ? spec_walk(w, rr, wkind, pair) : nullptr) {  // the same kernel compiled for this mechanism
      w->last_kernel = pair ? "aba_walk_spec (compiled for the mechanism, two fp32 states per lane)" : "aba_walk_spec (compiled for the mechanism)";
      long Bl = B;
      Layout lq = Lq, lv = Lv, lf = Lf;
      double gx = wm.gravity[0], gy = wm.gravity[1], gz = wm.gravity[2];
      MkStage F = mk ? *mk : kNoStage;
      void* args[] = {&Bl, (void*)&dq, (void*)&dv, (void*)&dtau, (void*)&df, (void*)&dvd, (void*)&dqd, &lq, &lv, &lf, &gx, &gy, &gz, &F};
      const long per = pair ? 128 : 64;
      HIP_TRY(hipModuleLaunchKernel(f, (unsigned)((B + per - 1) / per), 1, 1, 64u * (unsigned)wm.G, 1, 1, 0, w->stream, args, nullptr));
      // (first use of this program by this workspace: see first_use_check; a program that differs is dropped and the call recomputed on the kernel built with the library)
      const int k = 4 * wkind + (rr ? 2 : 0) + (pair ? 1 : 0);
      if (!mk && dv && dvd && w->spec_first_use_check && !w->spec_walk_checked[k] && !capturing(w)) {
        w->spec_walk_checked[k] = true;
        bool same = true;
        if (int st = first_use_check(w, B, dq, dv, dtau, df, dvd, Lq, Lv, Lf, gravity, &same)) return st;
        if (!same) { w->spec_walk[k] = nullptr; return run_aba(w, B, algorithm, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, gravity, fuse, mk); }
      }
    } else if (mk) return RBD_ERR_UNSUPPORTED;
    else
    if (w->dtype == RBD_F64) HIP_TRY(launch_aba_walk<double>(wm, TP.has_floating, TP.general, 0, B, lds, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, w->stream));
    else HIP_TRY(launch_aba_walk<float>(wm, TP.has_floating, TP.general, pair, B, lds, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, w->stream));
  } else if (pick == RBD_ALGO_ABA_BANKS) {
    BankModel bm = w->bm;
    if (gravity) memcpy(bm.gravity, gravity, sizeof bm.gravity);
    hipFunction_t f = spec_bank(w) ? (fuse ? w->spec_bank_fused : w->spec_bank_aba) : nullptr;
    if (f) {  // the same kernel compiled against this mechanism's level structure: the level loops unrolled (rbd_jit.hip spec_bank_source)
      MkFuse F{};
      F.stage = -1;
      if (fuse) F = *fuse;
      long Bl = B;
      const long spw = 64 / bm.lps, waves = (B + spw - 1) / spw;
      void* args[] = {&bm, &Bl, (void*)&dq, (void*)&dv, (void*)&dtau, (void*)&df, (void*)&dvd, (void*)&dqd, &Lq, &Lv, &Lf, &F};
      HIP_TRY(hipModuleLaunchKernel(f, (unsigned)((waves + 3) / 4), 1, 1, 256, 1, 1, 0, w->stream, args, nullptr));
      w->last_kernel = "aba_bank_kernel (compiled for the mechanism at run time)";
      if (!fuse && dv && dvd && w->spec_first_use_check && !w->spec_bank_checked && !capturing(w)) {  // (first use: see first_use_check)
        w->spec_bank_checked = true;
        bool same = true;
        if (int st = first_use_check(w, B, dq, dv, dtau, df, dvd, Lq, Lv, Lf, gravity, &same)) return st;
        if (!same) { w->spec_bank_aba = w->spec_bank_fused = nullptr; return run_aba(w, B, algorithm, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, gravity, fuse, mk); }
      }
    } else
    if (w->dtype == RBD_F64) HIP_TRY(launch_aba_bank<double>(bm, B, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, w->stream, fuse));
    else HIP_TRY(launch_aba_bank<float>(bm, B, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, w->stream, fuse));
  } else {
    DevModel dm = w->dm;
    if (gravity) memcpy(dm.gravity, gravity, sizeof dm.gravity);
    if (w->dtype == RBD_F64) HIP_TRY(launch_aba<double>(dm, B, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, w->stream, fuse));
    else HIP_TRY(launch_aba<float>(dm, B, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, w->stream, fuse));
  }
  return RBD_OK;
}

// The run-time specialised form of the one-lane-per-state kernels (rbd_spec.hpp): compiled for this mechanism on the workspace's first use of
// them (or loaded from the on-disk cache), nullptr when hiprtc is unavailable, RBD_JIT=0, or the compile failed — callers then keep the
// interpreting kernels.  Its stores address M with a 32-bit lane offset: buffers of 4 GB and more stay with the interpreting kernel.
// `force`: the caller asked for the compiled kernel by name (RBD_ALGO_ABA_COMPILED) — wait for the compiler.  Otherwise (RBD_JIT_ASYNC != 0, the default) a
// program that is not in the cache is compiled on a background thread (rbd_jit.hip) while the calls keep their interpreting kernels, and the module is
// loaded by the first call that finds it ready: the first dynamics! on a new mechanism returns in milliseconds, not after a minute of hiprtc.
static bool capturing(rbd_ws* w) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(w->stream, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cs != hipStreamCaptureStatusNone;
}
static void spec_load(rbd_ws* w, int family, bool force) {
  const int slot = spec_slot(family);
  if (w->spec_tried[slot]) return;
  const rbd_model* m = w->model;
  if (!m->spec_plan().ok || !jit_available() || !spec_has(family, w->dtype, m->nb, m->nq, m->nv, m->spec_plan().n3)) { w->spec_tried[slot] = true; return; }
  if (capturing(w)) return;  // a module cannot be loaded inside a stream capture: the interpreting kernels serve it, the next call outside tries again
  std::string log;
  std::string& src = w->spec_src[slot];
  if (src.empty()) src = spec_source(m->spec_plan(), m->nb, m->nq, m->nv, m->row_mask.data(), m->gravity, w->dtype, family);
  if (src.empty()) { w->spec_tried[slot] = true; return; }  // (no such program for this mechanism: fp64 dynamics! of a tree the walk kernels take)
  std::vector<char> code;
  const int js = jit_code_object_get(src, force || !jit_async(), &code, &log);
  if (js == JIT_PENDING) return;
  w->spec_tried[slot] = true;
  if (js == JIT_FAILED || code.empty()) { g_last_hip_error = "run-time compilation failed (the interpreting kernels are used): " + log; src.clear(); src.shrink_to_fit(); return; }
  hipModule_t& mod = w->spec_mod[slot];
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { (void)hipGetLastError(); mod = nullptr; jit_cache_discard(src); return; }
  auto get = [&](hipFunction_t* f, const char* name) { if (hipModuleGetFunction(f, mod, name) != hipSuccess) { (void)hipGetLastError(); *f = nullptr; } };
  // a kernel whose registers spilled beyond a few values is slower than the kernels that interpret the mechanism: it steps aside
  auto fits = [&](hipFunction_t* f, int* bytes = nullptr) {
    int scratch = 0;
    // bytes per lane (the fp64 dynamics! program of round 6 spills by construction and is taken with it: what it replaces is the one-body-per-lane kernel)
    const int max_scratch = (family == SPEC_ABA && w->dtype == RBD_F64) ? std::max(w->spec_max_scratch, w->spec_f64_max_scratch) : w->spec_max_scratch;
    if (*f && (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, *f) != hipSuccess || scratch > max_scratch)) { (void)hipGetLastError(); *f = nullptr; }
    if (bytes) *bytes = scratch;
  };
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, w->device);
  if (family == SPEC_MASS) {
    get(&w->spec_crba, w->dtype == RBD_F64 ? "crba_spec_f64" : "crba_spec_f32");
    if (spec_has_chol(w->dtype, m->nv)) {
      get(&w->spec_crba_perm, "crba_spec_perm_f32");
      get(&w->spec_chol, "chol_spec_f32");
      get(&w->spec_chol_packed, "chol_spec_packed_f32");  // M as the packed lower triangle (rbd_mass_matrix_solve_packed); optional
      get(&w->spec_chol_nom, "chol_spec_nom_f32");        // no M wanted: a kernel of its own (rbd_spec.hpp: EMIT)
      if (!w->spec_chol_nom) w->spec_chol = nullptr;
      get(&w->spec_emit, "emit_spec_f32");
      if (!w->spec_crba_perm || !w->spec_chol || !w->spec_emit) w->spec_crba_perm = w->spec_chol = w->spec_emit = nullptr;
    } else if (w->dtype == RBD_F64 && spec_has_chol(RBD_F32, m->nv)) {
      get(&w->spec_emit, "emit_spec_f64");  // fp64: the emitter alone (the staging buffer in the original order; the dense kernel is rbd_kernels.hip's)
    }
  } else if (family == SPEC_ABA) {
    get(&w->spec_aba, w->dtype == RBD_F64 ? "aba_spec_f64" : "aba_spec_f32");
    fits(&w->spec_aba, &w->spec_aba_scratch);
    get(&w->spec_aba_nofext, w->dtype == RBD_F64 ? "aba_spec_nofext_f64" : "aba_spec_nofext_f32");  // the instantiation for calls without external wrenches (rbd_spec.hpp: FEXT)
    fits(&w->spec_aba_nofext, &w->spec_aba_nofext_scratch);
    if (w->dtype == RBD_F64) {  // (the stash programs step in only where the LDS ones are taken: no threshold of their own)
      get(&w->spec_aba_gst, "aba_spec_gst_f64"); fits(&w->spec_aba_gst);
      get(&w->spec_aba_gst_nofext, "aba_spec_gst_nofext_f64"); fits(&w->spec_aba_gst_nofext);
    }
  } else if (family == SPEC_RNEA) {
    get(&w->spec_rnea, w->dtype == RBD_F64 ? "rnea_spec_f64" : "rnea_spec_f32");
    fits(&w->spec_rnea, &w->spec_rnea_scratch);
  } else if (family == SPEC_KIN) {  // (a kernel that spilled steps aside: the lane-per-body kin_kernel serves)
    int sc = 0;
    get(&w->spec_kin, w->dtype == RBD_F64 ? "kin_spec_f64" : "kin_spec_f32"); fits(&w->spec_kin, &sc); if (sc) w->spec_kin = nullptr;
    get(&w->spec_jac, w->dtype == RBD_F64 ? "jac_spec_f64" : "jac_spec_f32"); fits(&w->spec_jac, &sc); if (sc) w->spec_jac = nullptr;
    get(&w->spec_mom, w->dtype == RBD_F64 ? "mom_spec_f64" : "mom_spec_f32"); fits(&w->spec_mom, &sc); if (sc) w->spec_mom = nullptr;
    get(&w->spec_energy, w->dtype == RBD_F64 ? "energy_spec_f64" : "energy_spec_f32"); fits(&w->spec_energy, &sc); if (sc) w->spec_energy = nullptr;
    get(&w->spec_com, w->dtype == RBD_F64 ? "com_spec_f64" : "com_spec_f32"); fits(&w->spec_com, &sc); if (sc) w->spec_com = nullptr;
  }
}
// one launch of a by-product kernel compiled for the mechanism (rbd_spec.hpp kin_spec<T, WHAT>): a wavefront of 64 states per workgroup
static hipError_t launch_kin_spec(rbd_ws* w, hipFunction_t f, long B, const void* q, const void* v, void* A, void* com, void* energy, void* J, unsigned long long jplus,
                                  unsigned long long jminus, void* mom, Layout Lq, Layout Lv, Layout La, Layout L3, Layout L2, Layout L12) {
  void* args[] = {&B, &q, &v, &A, &com, &energy, &J, &jplus, &jminus, &mom, &Lq, &Lv, &La, &L3, &L2, &L12};
  return hipModuleLaunchKernel(f, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, w->stream, args, nullptr);
}
static bool spec_crba_fits(const rbd_ws* w) { return (size_t)w->model->nq * 4 * 65 * esize(w) <= 160u * 1024u; }  // four wavefronts' staged q (rows of 65) in one CU's LDS
static hipFunction_t spec_crba(rbd_ws* w, size_t buffer_bytes) {
  spec_load(w, SPEC_MASS);
  return buffer_bytes < ((size_t)1 << 32) && spec_crba_fits(w) ? w->spec_crba : nullptr;
}
static hipError_t launch_crba_spec(rbd_ws* w, hipFunction_t f, long B, const void* q, void* Mout, Layout Lq, Layout Lm, int zero_fill) {
  const unsigned lds = (unsigned)((size_t)w->model->nq * 4 * 65 * esize(w));  // four wavefronts' staged q, rows of 65 (rbd_spec.hpp RS)
  void* args[] = {&B, &q, &Mout, &Lq, &Lm, &zero_fill};
  return hipModuleLaunchKernel(f, (unsigned)((B + 255) / 256), 1, 1, 256, 1, 1, lds, w->stream, args, nullptr);
}

// the sparsity-specialised tile Cholesky on the permuted staging buffer (and, before it, the caller's M from the same buffer)
static hipError_t launch_chol_spec(rbd_ws* w, long B, const void* Mg, const void* tau, const void* c, void* x, Layout Lv, void* Mcopy, Layout Lc, bool packed = false) {
  int* notpd = w->d_notpd;
  void* args[] = {&B, &Mg, &tau, &c, &x, &Lv, &notpd, &Mcopy, &Lc};
  return hipModuleLaunchKernel(!Mcopy ? w->spec_chol_nom : packed ? w->spec_chol_packed : w->spec_chol, (unsigned)((B + 15) / 16), 1, 1, 64, 1, 1, 0, w->stream, args, nullptr);
}

// The staging buffer of M for the one-lane-per-state CRBA when the caller's layout is AOS: grouped by 16 states (Layout{16, -nv nv}).  Its
// structural zeros are written once per batch size and ordering (the specialised route stores M in the factorisation's order): the CRBA kernels
// only store the non-zeros.
static int stage_m(rbd_ws* w, int32_t B, bool permuted) {
  const rbd_model* m = w->model;
  void* const before = w->d_Msoa;
  const size_t bytes = esize(w) * (size_t)m->nv * m->nv * (((size_t)B + 15) & ~(size_t)15);
  int st;
  if ((st = ensure(&w->d_Msoa, &w->d_Msoa_bytes, bytes))) return st;
  if (w->Msoa_B != B || w->d_Msoa != before || w->Msoa_perm != (int)permuted) {
    HIP_TRY(hipMemsetAsync(w->d_Msoa, 0, bytes, w->stream));
    w->Msoa_B = B;
    w->Msoa_perm = (int)permuted;
  }
  return RBD_OK;
}

// mass_matrix! alone, into the caller's buffer
static int run_crba(rbd_ws* w, int32_t B, int layout, const void* dq, void* dM, Layout Lq, Layout Lm) {
  if (w->model->big) {
    int st = big_scratch(w, B);
    if (st) return st;
    if (w->dtype == RBD_F64) HIP_TRY(launch_big_crba<double>(w->big, B, dq, dM, w->d_big_scratch, Lq, Lm, w->stream));
    else HIP_TRY(launch_big_crba<float>(w->big, B, dq, dM, w->d_big_scratch, Lq, Lm, w->stream));
    return RBD_OK;
  }
  if (B >= w->mass_min_batch && layout == RBD_LAYOUT_AOS && Lm.sk == 1 && ((Lm.sb * (long)esize(w)) & 15) == 0 && (reinterpret_cast<uintptr_t>(dM) & 15) == 0 &&
      esize(w) * (size_t)w->model->nv * w->model->nv * (((size_t)B + 15) & ~(size_t)15) < ((size_t)1 << 32) &&
      (spec_load(w, SPEC_MASS), w->spec_emit != nullptr && spec_crba_fits(w) && (w->dtype == RBD_F32 ? w->spec_crba_perm : w->spec_crba) != nullptr)) {
    // one lane per state into the staging buffer, then whole cache lines of the caller's column-per-state M (the full square: emit_spec, rbd_spec.hpp);
    // fp32 stages in the factorisation's order (the emitter's gather lists follow the program's PERM), fp64 in the original one
    const bool perm = w->dtype == RBD_F32;
    int st = stage_m(w, B, perm);
    if (st) return st;
    const Layout Ls{16, -(long)w->model->nv * w->model->nv};
    HIP_TRY(launch_crba_spec(w, perm ? w->spec_crba_perm : w->spec_crba, B, dq, w->d_Msoa, Lq, Ls, 0));
    long Bl = B;
    void* args[] = {&Bl, &w->d_Msoa, &dM, &Lm};
    HIP_TRY(hipModuleLaunchKernel(w->spec_emit, (unsigned)((B + 15) / 16), 1, 1, 64, 1, 1, 0, w->stream, args, nullptr));
    w->last_kernel = perm ? "crba_spec_perm_f32 + emit_spec_f32 (compiled for the mechanism at run time)" : "crba_spec_f64 + emit_spec_f64 (compiled for the mechanism at run time)";
    return RBD_OK;
  }
  hipFunction_t const fsoa = B >= w->state_min_batch && layout == RBD_LAYOUT_SOA ? spec_crba(w, esize(w) * (size_t)w->model->nv * w->model->nv * B) : nullptr;
  if (B >= w->state_min_batch && layout == RBD_LAYOUT_SOA && (fsoa || w->state_aot)) {  // one lane per state: its stores are coalesced when the batch is innermost
    w->last_kernel = fsoa ? (w->dtype == RBD_F64 ? "crba_spec_f64 (compiled for the mechanism at run time)" : "crba_spec_f32 (compiled for the mechanism at run time)") : "crba_state_kernel";
    if (hipFunction_t f = fsoa) HIP_TRY(launch_crba_spec(w, f, B, dq, dM, Lq, Lm, 1));
    else if (w->dtype == RBD_F64) HIP_TRY(launch_crba_state<double>(w->sm, B, dq, dM, Lq, Lm, 1, w->stream));
    else HIP_TRY(launch_crba_state<float>(w->sm, B, dq, dM, Lq, Lm, 1, w->stream));
  } else {
    w->last_kernel = "crba_kernel";
    if (w->dtype == RBD_F64) HIP_TRY(launch_crba<double>(w->dm, B, dq, dM, Lq, Lm, 1, w->stream));
    else HIP_TRY(launch_crba<float>(w->dm, B, dq, dM, Lq, Lm, 1, w->stream));
  }
  return RBD_OK;
}

// mass_matrix! into dM (caller's layout) followed by the Cholesky solve of M x = tau - c (either may be null).  Large batches
// build M with one lane per state; for an AOS caller that kernel writes a batch-innermost staging copy which the tile Cholesky
// reads (coalesced) and re-emits as the caller's M.
// ... and of the fp32 mass_matrix! + Cholesky solve on the two kernels compiled for the mechanism (crba_spec_perm + chol_spec[_nom | _packed]) — taken from 256 states
// since round 6 —: x on the first states of the call against crba_kernel + the dense Cholesky kernel, 2e-2 of the largest |x| (two fp32 factorisations in different
// elimination orders of a matrix of condition 1e4).  Column-per-state callers only (the compiled pair's scope).
static int first_use_check_solve(rbd_ws* w, long B, const void* dq, const void* dtau, const void* dc, const void* dx, Layout Lq, Layout Lm, Layout Lv, bool* same) {
  const rbd_model* m = w->model;
  const long n = std::min<long>(B, 256);
  const size_t es = esize(w);
  void *Mt = nullptr, *xt = nullptr;
  double* out = nullptr;
  HIP_TRY(hipMalloc(&Mt, es * (size_t)m->nv * m->nv * n));
  if (hipMalloc(&xt, es * (size_t)m->nv * n) != hipSuccess || hipMalloc((void**)&out, 2 * sizeof(double)) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(Mt); (void)hipFree(xt); return RBD_ERR_OUT_OF_MEMORY;
  }
  double h[2] = {0, 0};
  hipError_t e = launch_crba<float>(w->dm, n, dq, Mt, Lq, Lm, 1, w->stream);
  if (e == hipSuccess) e = launch_chol_solve<float>(m->nv, n, Mt, dtau, dc, xt, nullptr, Lm, Lv, w->d_notpd, w->stream);
  if (e == hipSuccess) e = launch_max_diff<float>(n, m->nv, dx, xt, Lv, out, w->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, w->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
  (void)hipFree(Mt); (void)hipFree(xt); (void)hipFree(out);
  if (e != hipSuccess) { g_last_hip_error = std::string("first_use_check_solve: ") + hipGetErrorString(e); (void)hipGetLastError(); return RBD_ERR_HIP; }
  w->spec_check_err = h[0] / std::max(1e-30, h[1]);
  *same = w->spec_check_err <= 2e-2 && !w->spec_first_use_inject;
  if (!*same) {
    char msg[320];
    snprintf(msg, sizeof msg, "%s differs from crba_kernel + the dense Cholesky kernel on this call's first states by %.3g of the largest solution component%s: the programs are dropped, the kernels built with the library serve",
             w->last_kernel, w->spec_check_err, w->spec_first_use_inject ? " (RBD_TUNE first_use_inject)" : "");
    g_last_hip_error = msg;
    fprintf(stderr, "[rbd] %s\n", msg);
    w->spec_chol = w->spec_chol_nom = w->spec_chol_packed = nullptr;  // (the pair's second kernel in its three forms: every route that needs one falls back)
  }
  return RBD_OK;
}

static int run_crba_chol(rbd_ws* w, int32_t B, int layout, const void* dq, void* dM, const void* dtau, const void* dc, void* dx, Layout Lq,
                         Layout Lm, Layout Lv) {
  const rbd_model* m = w->model;
  const size_t es = esize(w);
  int st;
  if (m->big) {  // any-size fallback: M into dM (never null here), then one thread per state factors a COPY (the reference's result.L, :763-764) and solves
    if ((st = run_crba(w, B, layout, dq, dM, Lq, Lm))) return st;
    if ((st = ensure(&w->d_big_L, &w->d_big_L_bytes, es * (size_t)m->nv * m->nv * B))) return st;
    if (w->dtype == RBD_F64) HIP_TRY(launch_big_chol_solve<double>(m->nv, B, dM, w->d_big_L, dtau, dc, dx, Lm, Lv, w->d_notpd, w->stream));
    else HIP_TRY(launch_big_chol_solve<float>(m->nv, B, dM, w->d_big_L, dtau, dc, dx, Lm, Lv, w->d_notpd, w->stream));
    return RBD_OK;
  }
  const bool state = B >= w->mass_solve_min_batch;  // (below state_min_batch: only with both kernels compiled for the mechanism — spec_route)
  if (state && layout == RBD_LAYOUT_AOS && chol_copies_m((int)es, m->nv)) {
    spec_load(w, SPEC_MASS);
    const bool mcopy_ok = !dM || (Lm.sk == 1 && (Lm.sb & 3) == 0 && (reinterpret_cast<uintptr_t>(dM) & 15) == 0);
    const bool spec_route = w->spec_chol && spec_crba_fits(w) && es * (size_t)m->nv * m->nv * (((size_t)B + 15) & ~(size_t)15) < ((size_t)1 << 32) && mcopy_ok;
    // the compiled mass-matrix kernel for THIS call's staging size, asked for once (the buffer itself keeps the high-water mark of earlier, larger batches: asking
    // again with its size could answer differently — 4 GB and more — and leave a mechanism the interpreting kernel does not take without any kernel)
    hipFunction_t const spec = spec_route ? nullptr : spec_crba(w, es * (size_t)m->nv * m->nv * (((size_t)B + 15) & ~(size_t)15));
    if (!spec_route && !w->state_aot && !spec) goto lanes;  // (a mechanism only the compiled kernels take, and they are not there)
    if (!spec_route && B < w->state_min_batch) goto lanes;
    if ((st = stage_m(w, B, spec_route))) return st;
    const Layout Ls{16, -(long)m->nv * m->nv};  // grouped by 16 states = one wavefront of the tile Cholesky (layout_base, rbd_device.hpp)
    if (spec_route) {
      // both kernels compiled for the mechanism: M in the factorisation's own order (children before parents: no fill-in), only the tiles
      // that hold non-zeros are factored; the same launch writes the caller's M
      HIP_TRY(launch_crba_spec(w, w->spec_crba_perm, B, dq, w->d_Msoa, Lq, Ls, 0));
      // (the emission of M as a launch of its own on a second stream beside the factorisation — emit_spec beside chol_spec without M — was measured: 145 us
      //  against 120 for the pair in one launch; what pays is the staggered order inside chol_spec, rbd_spec.hpp)
      HIP_TRY(launch_chol_spec(w, B, w->d_Msoa, dtau, dc, dx, Lv, dM, Lm));
      w->last_kernel = "crba_spec_perm_f32 + chol_spec_f32 (compiled for the mechanism at run time)";
      bool& checked = w->spec_mass_checked[dM ? 0 : 1];
      if (w->spec_first_use_check && !checked && dx && !capturing(w)) {  // (first use of the pair by this workspace: first_use_check_solve)
        checked = true;
        bool same = true;
        if ((st = first_use_check_solve(w, B, dq, dtau, dc, dx, Lq, Lm, Lv, &same))) return st;
        if (!same) goto lanes;
      }
      return RBD_OK;
    }
    if (spec) HIP_TRY(launch_crba_spec(w, spec, B, dq, w->d_Msoa, Lq, Ls, 0));
    else HIP_TRY(launch_crba_state<float>(w->sm, B, dq, w->d_Msoa, Lq, Ls, 0, w->stream));
    HIP_TRY(launch_chol_solve<float>(m->nv, B, w->d_Msoa, dtau, dc, dx, nullptr, Ls, Lv, w->d_notpd, w->stream, dM, Lm));
    w->last_kernel = spec ? "crba_spec_f32 (compiled for the mechanism at run time) + chol_mfma_kernel" : "crba_state_kernel + chol_mfma_kernel";
    return RBD_OK;
  }
lanes:
  if (!dM) {  // (the caller left M out counting on the staged route)
    if ((st = ensure(&w->d_M, &w->d_M_bytes, es * (size_t)m->nv * m->nv * B))) return st;
    dM = w->d_M;
  }
  if ((st = run_crba(w, B, layout, dq, dM, Lq, Lm))) return st;
  if (w->dtype == RBD_F64) HIP_TRY(launch_chol_solve<double>(m->nv, B, dM, dtau, dc, dx, nullptr, Lm, Lv, w->d_notpd, w->stream));
  else HIP_TRY(launch_chol_solve<float>(m->nv, B, dM, dtau, dc, dx, nullptr, Lm, Lv, w->d_notpd, w->stream));
  w->last_kernel = strstr(w->last_kernel, "crba_spec") ? "crba_spec (compiled for the mechanism at run time) + chol kernel"
                   : strstr(w->last_kernel, "crba_state") ? "crba_state_kernel + chol kernel" : "crba_kernel + chol kernel";
  return RBD_OK;
}

// dynamics! on device pointers: ABA, the reference's CRBA + Cholesky route, or the loop-joint branch
static int run_dynamics(rbd_ws* w, int32_t B, const Opts& o, const void* dq, const void* dv, const void* dtau, const void* df, void* dvd,
                        void* dqd, void* dlam) {
  const rbd_model* m = w->model;
  const size_t es = esize(w);
  const bool loops = m->nloops > 0;  // has_loops(mechanism): only the CRBA route exists (src/mechanism_algorithms.jl:858-861)
  int st;
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), Lf = layout_of(o.layout, 6L * m->nb, B);
  if (loops) {
    if ((st = dynamics_loops(w, B, o, dq, dv, dtau, df, dvd, dqd, dlam))) return st;
  } else if (o.algorithm == RBD_ALGO_CRBA_CHOLESKY || m->big) {  // (more than 64 bodies: the reference's route is the only one built for any size)
    // the reference's own route (src/mechanism_algorithms.jl:856-862): c = dynamics_bias!, M = mass_matrix!, then
    // potrf!/potrs!.  M and c stay in the workspace (layout of this call) for rbd_dynamics_result.
    const Layout Lm = layout_of(o.layout, (long)m->nv * m->nv, B);
    // M and c: the caller's own buffers when bound (rbd_workspace_bind_result: no copy afterwards), else the workspace's
    // (bound buffers are laid out like the DEVICE buffers of the call: a host-memory call — its q, v are staged copies — keeps the workspace's)
    void* const bM = o.memory == RBD_MEM_HOST ? nullptr : w->bound_M;
    void* const bc = o.memory == RBD_MEM_HOST ? nullptr : w->bound_c;
    if (!bM && (st = ensure(&w->d_M, &w->d_M_bytes, es * (size_t)m->nv * m->nv * B))) return st;
    if (!bc && (st = ensure(&w->d_c, &w->d_c_bytes, es * (size_t)m->nv * B))) return st;
    void* const Md = bM ? bM : w->d_M;
    void* const cd = bc ? bc : w->d_c;
    w->result_layout = o.layout; w->result_B = B;
    Timed t(w);
    if ((st = run_rnea(w, B, RBD_ALGO_ABA, dq, dv, nullptr, df, cd, dqd, Lq, Lv, Lf))) return st;
    if ((st = run_crba_chol(w, B, o.layout, dq, Md, dtau, cd, dvd, Lq, Lm, Lv))) return st;
  } else {
    if ((st = run_aba(w, B, o.algorithm, dq, dv, dtau, df, dvd, dqd, Lq, Lv, Lf, nullptr, nullptr))) return st;
  }
  return RBD_OK;
}

extern "C" {

int rbd_dynamics(rbd_ws_t* w, int32_t B, const void* q, const void* v, const void* tau, const void* fext, void* vdot, void* qdot,
                 void* lambda, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || missing(v, m->nv) || missing(vdot, m->nv)) return RBD_ERR_INVALID_ARGUMENT;
  // the reference's dynamics! always runs contact_dynamics! (src/mechanism_algorithms.jl:849-856): a mechanism with contact points and an
  // environment needs its additional state s — rbd_dynamics_contact; this entry point must not silently leave the contact wrenches out
  if (m->ncp > 0 && m->nhs > 0) return RBD_ERR_UNSUPPORTED;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const void *dq = q, *dv = v, *dtau = tau, *df = fext;
  void *dvd = vdot, *dqd = qdot, *dlam = lambda;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_out_alloc(w, 7, lambda, es * (m->nc > 0 ? m->nc : 1) * B, &dlam))) return st;
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_in(w, 1, v, es * m->nv * B, &dv)) ||
        (st = stage_in(w, 2, tau, es * m->nv * B, &dtau)) || (st = stage_in(w, 3, fext, es * 6 * m->nb * B, &df)) ||
        (st = stage_out_alloc(w, 4, vdot, es * m->nv * B, &dvd)) || (st = stage_out_alloc(w, 5, qdot, es * m->nq * B, &dqd)))
      return st;
  }
  if ((st = run_dynamics(w, B, o, dq, dv, dtau, df, dvd, dqd, dlam))) return st;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_out_copy(w, vdot, dvd, es * m->nv * B)) || (st = stage_out_copy(w, qdot, dqd, es * m->nq * B)) ||
        (st = stage_out_copy(w, lambda, dlam, es * m->nc * B)))
      return st;
  }
  return RBD_OK;
}

static int rnea_common(rbd_ws_t* w, int32_t B, const void* q, const void* v, const void* vdot, const void* fext, void* tau_out,
                       const rbd_opts_t* opts, void* jw_out = nullptr, void* acc_out = nullptr) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || missing(v, m->nv) || missing(tau_out, m->nv)) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const void *dq = q, *dv = v, *dvd = vdot, *df = fext;
  void *dt = tau_out, *djw = jw_out, *dacc = acc_out;
  const size_t bbytes = es * 6 * (size_t)m->nb * B;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_in(w, 1, v, es * m->nv * B, &dv)) ||
        (st = stage_in(w, 2, vdot, es * m->nv * B, &dvd)) || (st = stage_in(w, 3, fext, es * 6 * m->nb * B, &df)) ||
        (st = stage_out_alloc(w, 4, tau_out, es * m->nv * B, &dt)) || (st = stage_out_alloc(w, 5, jw_out, bbytes, &djw)) ||
        (st = stage_out_alloc(w, 6, acc_out, bbytes, &dacc)))
      return st;
  }
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), Lf = layout_of(o.layout, 6L * m->nb, B);
  {
    Timed t(w);
    if ((st = run_rnea(w, B, o.algorithm, dq, dv, dvd, df, dt, nullptr, Lq, Lv, Lf, dacc, djw))) return st;
  }
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_out_copy(w, tau_out, dt, es * m->nv * B)) || (st = stage_out_copy(w, jw_out, djw, bbytes)) || (st = stage_out_copy(w, acc_out, dacc, bbytes))) return st;
  }
  return RBD_OK;
}

int rbd_inverse_dynamics(rbd_ws_t* w, int32_t B, const void* q, const void* v, const void* vdot, const void* fext, void* tau_out,
                         const rbd_opts_t* opts) {
  if (w && w->model->nloops > 0) return RBD_ERR_HAS_LOOPS;  // src/mechanism_algorithms.jl:549
  if (!vdot && w && w->model->nv > 0) return RBD_ERR_INVALID_ARGUMENT;
  return rnea_common(w, B, q, v, vdot, fext, tau_out, opts);
}

int rbd_dynamics_bias(rbd_ws_t* w, int32_t B, const void* q, const void* v, const void* fext, void* c_out, const rbd_opts_t* opts) {
  return rnea_common(w, B, q, v, nullptr, fext, c_out, opts);
}

int rbd_inverse_dynamics_bodies(rbd_ws_t* w, int32_t B, const void* q, const void* v, const void* vdot, const void* fext, void* tau_out,
                                void* jointwrenches_out, void* accelerations_out, const rbd_opts_t* opts) {
  if (w && w->model->nloops > 0) return RBD_ERR_HAS_LOOPS;  // src/mechanism_algorithms.jl:549
  if (!vdot && w && w->model->nv > 0) return RBD_ERR_INVALID_ARGUMENT;
  return rnea_common(w, B, q, v, vdot, fext, tau_out, opts, jointwrenches_out, accelerations_out);
}

int rbd_dynamics_bias_bodies(rbd_ws_t* w, int32_t B, const void* q, const void* v, const void* fext, void* c_out, void* jointwrenches_out,
                             void* accelerations_out, const rbd_opts_t* opts) {
  return rnea_common(w, B, q, v, nullptr, fext, c_out, opts, jointwrenches_out, accelerations_out);
}

int rbd_mass_matrix(rbd_ws_t* w, int32_t B, const void* q, void* M_out, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || missing(M_out, m->nv)) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const void* dq = q;
  void* dM = M_out;
  const size_t mbytes = es * (size_t)m->nv * m->nv * B;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_out_alloc(w, 6, M_out, mbytes, &dM))) return st;
  }
  const Layout Lq = layout_of(o.layout, m->nq, B), Lm = layout_of(o.layout, (long)m->nv * m->nv, B);
  {
    Timed t(w);
    if ((st = run_crba(w, B, o.layout, dq, dM, Lq, Lm))) return st;
  }
  if (o.memory == RBD_MEM_HOST) return stage_out_copy(w, M_out, dM, mbytes);
  return RBD_OK;
}

int rbd_mass_matrix_solve(rbd_ws_t* w, int32_t B, const void* q, const void* rhs, void* x, void* M_out, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || missing(rhs, m->nv) || missing(x, m->nv)) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const size_t mbytes = es * (size_t)m->nv * m->nv * B;
  const void *dq = q, *dr = rhs;
  void *dx = x, *dM = M_out;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_in(w, 2, rhs, es * m->nv * B, &dr)) ||
        (st = stage_out_alloc(w, 4, x, es * m->nv * B, &dx)) || (st = stage_out_alloc(w, 6, M_out, mbytes, &dM)))
      return st;
  }
  // M_out == NULL: the caller wants x only.  The lane-per-state route (large batches) then skips the emission of M altogether — its factorization
  // reads the staged triangle, and the whole-square store is 340 MB of the route's ~560 MB at 65 536 Atlas states; the other routes factor M in
  // place and need a buffer of their own
  const bool crba_route = o.algorithm == RBD_ALGO_CRBA_CHOLESKY || m->big;
  const bool state_route = !m->big && B >= w->mass_solve_min_batch && o.layout == RBD_LAYOUT_AOS && chol_copies_m((int)es, m->nv);
  if (!dM && crba_route && !state_route) {
    if ((st = ensure(&w->d_M, &w->d_M_bytes, mbytes))) return st;
    dM = w->d_M;
  }
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), Lm = layout_of(o.layout, (long)m->nv * m->nv, B);
  if (crba_route) {
    Timed t(w);
    if ((st = run_crba_chol(w, B, o.layout, dq, dM, dr, nullptr, dx, Lq, Lm, Lv))) return st;
  } else {
    // O(n) solve: x = M(q)^-1 rhs is forward dynamics with v = 0, no gravity and tau = rhs (then c = 0), i.e. one pass of
    // the articulated-body kernel — no matrix is formed unless the caller asked for it.
    const double g0[3] = {0.0, 0.0, 0.0};
    const Layout Lf = layout_of(o.layout, 6L * m->nb, B);
    if ((st = run_aba(w, B, RBD_ALGO_ABA, dq, nullptr, dr, nullptr, dx, nullptr, Lq, Lv, Lf, g0, nullptr))) return st;
    if (dM && (st = run_crba(w, B, o.layout, dq, dM, Lq, Lm))) return st;
  }
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_out_copy(w, x, dx, es * m->nv * B)) || (st = stage_out_copy(w, M_out, dM, mbytes))) return st;
  }
  return RBD_OK;
}

// x = M(q)^-1 rhs with M as LAPACK's packed lower triangle — the part of M the reference defines (Symmetric, uplo 'L': src/dynamics_result.jl:42), half the bytes
// of the square, and bytes are what the emission of M costs.  Large fp32 batches of a mechanism with compiled kernels: the tile Cholesky sends the triangle
// on its way from its own tiles (chol_spec<PACKED>, rbd_spec.hpp); everything else forms the square in the workspace and packs it.
int rbd_mass_matrix_solve_packed(rbd_ws_t* w, int32_t B, const void* q, const void* rhs, void* x, void* M_packed_out, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || missing(rhs, m->nv) || missing(x, m->nv) || !M_packed_out) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const long np = (long)m->nv * (m->nv + 1) / 2;
  const size_t pbytes = es * (size_t)np * B;
  const void *dq = q, *dr = rhs;
  void *dx = x, *dP = M_packed_out;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_in(w, 2, rhs, es * m->nv * B, &dr)) ||
        (st = stage_out_alloc(w, 4, x, es * m->nv * B, &dx)) || (st = stage_out_alloc(w, 6, M_packed_out, pbytes, &dP)))
      return st;
  }
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), Lm = layout_of(o.layout, (long)m->nv * m->nv, B), Lp = layout_of(o.layout, np, B);
  const bool fast = !m->big && m->nv > 0 && B >= w->mass_solve_min_batch && o.layout == RBD_LAYOUT_AOS && chol_copies_m((int)es, m->nv) &&
                    (spec_load(w, SPEC_MASS), w->spec_chol_packed != nullptr && w->spec_crba_perm != nullptr) && spec_crba_fits(w) &&
                    es * (size_t)m->nv * m->nv * (((size_t)B + 15) & ~(size_t)15) < ((size_t)1 << 32) &&
                    (reinterpret_cast<uintptr_t>(dP) & 15) == 0;  // (the triangle leaves in 16-byte pieces)
  {
    Timed t(w);
    bool fast_failed = false;
    if (fast) {
      if ((st = stage_m(w, B, true))) return st;
      const Layout Ls{16, -(long)m->nv * m->nv};
      HIP_TRY(launch_crba_spec(w, w->spec_crba_perm, B, dq, w->d_Msoa, Lq, Ls, 0));
      HIP_TRY(launch_chol_spec(w, B, w->d_Msoa, dr, nullptr, dx, Lv, dP, Lp, true));
      w->last_kernel = "crba_spec_perm_f32 + chol_spec_packed_f32 (compiled for the mechanism at run time)";
      if (w->spec_first_use_check && !w->spec_mass_checked[2] && dx && !capturing(w)) {
        w->spec_mass_checked[2] = true;
        bool same = true;
        if ((st = first_use_check_solve(w, B, dq, dr, nullptr, dx, Lq, Lm, Lv, &same))) return st;
        if (!same) fast_failed = true;
      }
    }
    if (!fast || fast_failed) {
      if ((st = ensure(&w->d_M, &w->d_M_bytes, es * (size_t)m->nv * m->nv * B))) return st;
      if ((st = run_crba_chol(w, B, o.layout, dq, w->d_M, dr, nullptr, dx, Lq, Lm, Lv))) return st;
      if (w->dtype == RBD_F64) HIP_TRY(launch_pack_lower<double>(m->nv, B, w->d_M, dP, Lm, Lp, w->stream));
      else HIP_TRY(launch_pack_lower<float>(m->nv, B, w->d_M, dP, Lm, Lp, w->stream));
    }
  }
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_out_copy(w, x, dx, es * m->nv * B)) || (st = stage_out_copy(w, M_packed_out, dP, pbytes))) return st;
  }
  return RBD_OK;
}

int rbd_dynamics_result(rbd_ws_t* w, int32_t B, void* M, void* c, void* K, void* k, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (B != w->result_B || o.layout != w->result_layout) return RBD_ERR_DIMENSION_MISMATCH;  // must match the producing rbd_dynamics call
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const hipMemcpyKind kind = (o.memory == RBD_MEM_HOST) ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  // (a buffer bound with rbd_workspace_bind_result already holds its field: the tree-mechanism route wrote it in place)
  const bool tree = m->nloops == 0;
  if (M && !(tree && M == w->bound_M)) { if (!w->d_M) return RBD_ERR_INVALID_ARGUMENT; HIP_TRY(hipMemcpyAsync(M, w->d_M, es * (size_t)m->nv * m->nv * B, kind, w->stream)); }
  if (c && !(tree && c == w->bound_c)) { if (!w->d_c) return RBD_ERR_INVALID_ARGUMENT; HIP_TRY(hipMemcpyAsync(c, w->d_c, es * (size_t)m->nv * B, kind, w->stream)); }
  if (K && m->nc > 0) { if (!w->d_K) return RBD_ERR_INVALID_ARGUMENT; HIP_TRY(hipMemcpyAsync(K, w->d_K, es * (size_t)m->nc * m->nv * B, kind, w->stream)); }
  if (k && m->nc > 0) { if (!w->d_k) return RBD_ERR_INVALID_ARGUMENT; HIP_TRY(hipMemcpyAsync(k, w->d_k, es * (size_t)m->nc * B, kind, w->stream)); }
  return RBD_OK;
}


static int mk_ensure(rbd_ws* w, int32_t B) {
  const rbd_model* m = w->model;
  const size_t es = esize(w);
  if (w->mk_elems >= (size_t)B) return RBD_OK;
  void** ptrs[] = {&w->mk.q0, &w->mk.v0, &w->mk.phid[0], &w->mk.phid[1], &w->mk.phid[2], &w->mk.phid[3], &w->mk.vd[0], &w->mk.vd[1], &w->mk.vd[2],
                   &w->mk.vd[3], &w->d_vdwork};
  for (void** p : ptrs) { if (*p) HIP_TRY(hipFree(*p)); *p = nullptr; }
  w->mk_elems = 0;
  const size_t nq = (size_t)(m->nq > 0 ? m->nq : 1), nv = (size_t)(m->nv > 0 ? m->nv : 1);
  HIP_TRY(hipMalloc(&w->mk.q0, es * nq * B));
  HIP_TRY(hipMalloc(&w->mk.v0, es * nv * B));
  for (int k = 0; k < 4; ++k) { HIP_TRY(hipMalloc(&w->mk.phid[k], es * nv * B)); HIP_TRY(hipMalloc(&w->mk.vd[k], es * nv * B)); }
  HIP_TRY(hipMalloc(&w->d_vdwork, es * nv * B));
  w->mk_elems = (size_t)B;
  return RBD_OK;
}

// one stage of the integrator in a launch of its own: lane-per-body tables, or the any-size ones (more than 64 bodies)
static hipError_t stage_launch(rbd_ws* w, long B, int stage, double dt, void* q, void* v, const void* vdot_prev, Layout Lq, Layout Lv, int close_prev = 0) {
  if (w->model->big)
    return w->dtype == RBD_F64 ? launch_big_mk_stage<double>(w->big, B, stage, dt, q, v, vdot_prev, w->mk, Lq, Lv, w->stream, close_prev)
                               : launch_big_mk_stage<float>(w->big, B, stage, dt, q, v, vdot_prev, w->mk, Lq, Lv, w->stream, close_prev);
  return w->dtype == RBD_F64 ? launch_mk_stage<double>(w->dm, B, stage, dt, q, v, vdot_prev, w->mk, Lq, Lv, w->stream, close_prev)
                             : launch_mk_stage<float>(w->dm, B, stage, dt, q, v, vdot_prev, w->mk, Lq, Lv, w->stream, close_prev);
}

int rbd_mk_stage(rbd_ws_t* w, int32_t B, int32_t stage, double dt, void* q, void* v, const void* vdot_prev, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  if (!q || !v || stage < 0 || stage > 4 || (stage > 0 && !vdot_prev) || o.memory != RBD_MEM_DEVICE) return RBD_ERR_INVALID_ARGUMENT;
  if (w->model->ncp > 0 && w->model->nhs > 0) return RBD_ERR_UNSUPPORTED;  // the additional contact state is not part of this stage form
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  if ((st = mk_ensure(w, B))) return st;
  const rbd_model* m = w->model;
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B);
  HIP_TRY(stage_launch(w, B, stage, dt, q, v, vdot_prev, Lq, Lv));
  return RBD_OK;
}

// simulate with a controller descriptor (rbd_simulate: constant τ)
static int simulate_core(rbd_ws_t* w, int32_t B, void* q, void* v, const rbd_control_t& ctl, const void* fext, double dt, int32_t nsteps, const rbd_opts_t* opts) {
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  if (!q || !v || nsteps < 0 || !(dt > 0)) return RBD_ERR_INVALID_ARGUMENT;
  if (w->model->ncp > 0 && w->model->nhs > 0) return RBD_ERR_UNSUPPORTED;  // contact points: rbd_simulate_contact (carries the additional state)
  if (ctl.kind != RBD_CONTROL_CONSTANT && (o.memory != RBD_MEM_DEVICE || w->model->nloops > 0)) return RBD_ERR_UNSUPPORTED;
  if (ctl.kind == RBD_CONTROL_TABLE && !ctl.tau) return RBD_ERR_INVALID_ARGUMENT;
  if (ctl.kind == RBD_CONTROL_PD && (!ctl.kp || !ctl.kd)) return RBD_ERR_INVALID_ARGUMENT;
  if (ctl.kind < RBD_CONTROL_CONSTANT || ctl.kind > RBD_CONTROL_PD) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0 || nsteps == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const rbd_model* m = w->model;
  const size_t es = esize(w);
  void *dq = q, *dv = v;
  const void *dtau = ctl.tau, *df = fext;
  if (o.memory == RBD_MEM_HOST) {
    const void *cq, *cv;
    if ((st = stage_in(w, 0, q, es * m->nq * B, &cq)) || (st = stage_in(w, 1, v, es * m->nv * B, &cv)) ||
        (st = stage_in(w, 2, ctl.tau, es * m->nv * B, &dtau)) || (st = stage_in(w, 3, fext, es * 6 * m->nb * B, &df)))
      return st;
    dq = const_cast<void*>(cq); dv = const_cast<void*>(cv);
  }
  if ((st = mk_ensure(w, B))) return st;
  Opts od = o; od.memory = RBD_MEM_DEVICE;
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B);
  // the torques of (step, stage): constant / PD feed-forward, or an entry of the table
  const size_t entry = es * (size_t)m->nv * B;
  auto tau_at = [&](int step, int stage) -> const void* {
    if (ctl.kind != RBD_CONTROL_TABLE) return dtau;
    return (const char*)dtau + entry * (size_t)(ctl.per_stage ? 4 * step + stage : step);
  };
  const bool pd = ctl.kind == RBD_CONTROL_PD;
  // large batches: the walk kernel (one wavefront per track, §3.4 of DESIGN.md) with the stage bookkeeping in its own launches beats the
  // lane-per-body kernels with the stage fused in (fp64 Atlas, 65 536 states: 4 x 147 us + 5 stage launches vs 4 x 290 us)
  const bool can_walk_sim = (m->nloops == 0) && (o.algorithm == RBD_ALGO_ABA) && m->track.ok && m->walk.ok && w->walk_lds_bytes > 0;
  bool walk_sim = can_walk_sim && B >= w->walk_min_batch;
  // ... and at EVERY smaller batch when the walk program compiled for the mechanism with the four stages in a loop is there: one launch per step whose time does
  // not depend on the batch up to 16 384 states — Atlas, us per step against the banked kernel with the stage fused in (four launches): fp64 256 states 102.8 / 115.3,
  // 2048: 103.6 / 107.8, 4096: 104.6 / 109.9; fp32 256: 81.1 / 99.9, 4096: 82.3 / 93.8, 8192: 84.6 / 120.7 (round 6; until then from walk_min_batch = 4097 / 8193)
  if (!walk_sim && can_walk_sim && B >= w->sim_walk_min_batch && (w->dtype == RBD_F64 || w->spec_walk_f32) &&
      tune("sim_fuse", 1) != 0 && tune("sim_one_launch", 1) != 0 && !(w->dtype == RBD_F32 && w->walk_lds_bytes_pair > 0 && B >= w->walk_pair_min_batch))
    walk_sim = spec_walk(w, w->walk_rr && !w->no_reroot && w->walk_rr_lds_bytes > 0, 2, 0) != nullptr;
  const Layout Lf = layout_of(o.layout, 6L * m->nb, B);
  // ... and when the batch goes to a kernel COMPILED for the mechanism (aba_walk_spec, aba_spec_f32), the stage is folded into that launch (rbd_mk_fuse.hpp):
  // four launches per step and nothing else.  The first launch decides: a kernel that takes the stage runs it, any other returns RBD_ERR_UNSUPPORTED untouched.
  bool spec_sim = false;
  // (mechanisms the walk kernels do not take — 3-dof joints, 6-dof joints below the world: aba_spec_f32 takes them, stage included, from its own batch threshold on)
  const bool lane_per_state_sim = (m->nloops == 0) && (o.algorithm == RBD_ALGO_ABA) && (w->dtype == RBD_F32 || (m->state_wide.ok && !m->state.ok)) && m->spec_plan().ok && B >= std::min<long>(w->spec_aba_min_batch, w->spec_aba_fused_min_batch);  // (fp64, round 6: the mechanisms whose dynamics! runs on aba_spec_f64)
  const bool try_spec_sim = (walk_sim || lane_per_state_sim) && tune("sim_fuse", 1) != 0;  // (RBD_TUNE sim_fuse=0: the stage in its own launches, for A/B measurements)
  // The kernel of the FIRST launch serves the whole call: aba_spec keeps the stage buffers in a layout of its own (rbd_spec.hpp), and a compilation that finishes
  // in the background must not move a step from one kernel to the other between two of its stages.
  int sim_algo = RBD_ALGO_ABA;
  bool sim_lane_per_state = false;
  // the walk kernel compiled for the mechanism takes ALL FOUR stages of a step in one launch (rbd_walk.hpp aba_walk_spec, MkStage::stage = 4): the stage states
  // never leave its LDS rows.  Tried first where the lane-per-state kernel is not in line for the batch; RBD_ERR_UNSUPPORTED (the program is not compiled yet,
  // or the batch goes elsewhere) leaves everything untouched for the routes below.
  const bool walk_round = w->dtype == RBD_F32 && w->walk_lds_bytes_pair > 0 && B >= w->walk_pair_min_batch && B <= w->sim_walk_max_batch;  // (see sim_walk_max_batch)
  bool one_launch = try_spec_sim && walk_sim && (!lane_per_state_sim || walk_round) && tune("sim_one_launch", 1) != 0;
  // (Round 5 ran this kernel against the single-stage one inside the first call before using it — a hipMalloc and two synchronisations in a hot-path call — because
  //  the looped program's register allocator may take accumulation registers of its own beside the stash the passes address by number.  Since round 6 the
  //  admission is static: rbd_jit.hip loads the program only when the allocator's registers a0 .. a[.agpr_count − 1] end below the stash's lowest register, so
  //  they cannot alias whatever the live ranges are; a program that does not qualify is not loaded and the four-launch form below serves.  The comparison of the
  //  two forms lives in the tests: tests/test_gpu_parity.py test_simulate_stage_folded_into_the_compiled_kernels[one_launch], scripts/stress_simulate_walk.py.)
  if (one_launch) {
    const bool per_stage = ctl.kind == RBD_CONTROL_TABLE && ctl.per_stage;
    bool ok = true;
    for (int step = 0; step < nsteps; ++step) {
      MkStage F{4, pd ? 1 : 0, dt, w->mk.q0, w->mk.v0, w->mk.phid[0], w->mk.vd[0], dq, dv, ctl.kp, ctl.kd, ctl.q_des, 0};
      F.tau_stride = per_stage ? (int64_t)m->nv * B : 0;
      st = run_aba(w, B, RBD_ALGO_ABA_WALK, dq, dv, tau_at(step, 0), df, nullptr, nullptr, Lq, Lv, Lf, nullptr, nullptr, &F);
      if (st == RBD_ERR_UNSUPPORTED && step == 0) { ok = false; break; }
      if (st) return st;
    }
    if (ok) {
      w->last_kernel = "aba_walk_spec with the Munthe-Kaas stage folded in (compiled for the mechanism; four stages per launch)";
      if (o.memory == RBD_MEM_HOST) {
        if ((st = stage_out_copy(w, q, dq, es * m->nq * B)) || (st = stage_out_copy(w, v, dv, es * m->nv * B))) return st;
      }
      return RBD_OK;
    }
  }
  for (int step = 0; try_spec_sim && step < nsteps; ++step) {
    for (int stage = 0; stage < 4; ++stage) {
      const MkStage F{stage, pd ? 1 : 0, dt, w->mk.q0, w->mk.v0, w->mk.phid[0], w->mk.vd[0], dq, dv, ctl.kp, ctl.kd, ctl.q_des};
      st = run_aba(w, B, sim_algo, dq, dv, tau_at(step, stage), df, nullptr, nullptr, Lq, Lv, Lf, nullptr, nullptr, &F);
      if (st == RBD_ERR_UNSUPPORTED && step == 0 && stage == 0) break;
      if (st) return st;
      if (!spec_sim) {
        sim_lane_per_state = strstr(w->last_kernel, "aba_spec_") != nullptr;  // (aba_spec_f32 / aba_spec_f64 / aba_spec_gst_f64)
        sim_algo = sim_lane_per_state ? RBD_ALGO_ABA_COMPILED : RBD_ALGO_ABA_WALK;
      }
      spec_sim = true;
    }
    if (!spec_sim) break;
  }
  if (spec_sim) {
    const bool sim_stash = strstr(w->last_kernel, "aba_spec_gst_") != nullptr;
    w->last_kernel = sim_lane_per_state ? (sim_stash ? "aba_spec_gst_f64 with the Munthe-Kaas stage folded in (compiled for the mechanism at run time; spare rows in the HBM stash)"
                                           : w->dtype == RBD_F64 ? "aba_spec_f64 with the Munthe-Kaas stage folded in (compiled for the mechanism at run time)"
                                                               : "aba_spec_f32 with the Munthe-Kaas stage folded in (compiled for the mechanism at run time)")
                                        : "aba_walk_spec with the Munthe-Kaas stage folded in (compiled for the mechanism)";
    if (o.memory == RBD_MEM_HOST) {
      if ((st = stage_out_copy(w, q, dq, es * m->nq * B)) || (st = stage_out_copy(w, v, dv, es * m->nv * B))) return st;
    }
    return RBD_OK;
  }
  const bool fused = (m->nloops == 0) && (o.algorithm == RBD_ALGO_ABA) && !walk_sim && !m->big;  // (more than 64 bodies: the stage in launches of its own around rbd_dynamics' any-size route)
  for (int step = 0; fused && step < nsteps; ++step) {
    // tree mechanism, articulated-body route: each of the four stages is ONE launch (stage bookkeeping — and the PD law, on the stage state —
    // fused into the ABA kernel; the closing stage of a step rides in the first launch of the next one; only the last step closes on its own)
    for (int stage = 0; stage < 4; ++stage) {
      MkFuse F{};
      F.stage = stage; F.close_prev = (stage == 0 && step > 0) ? 1 : 0; F.dt = dt; F.W = w->mk; F.q_state = dq; F.v_state = dv;
      if (pd) { F.pd_kp = ctl.kp; F.pd_kd = ctl.kd; F.pd_qdes = ctl.q_des; }
      if ((st = run_aba(w, B, RBD_ALGO_ABA, dq, dv, tau_at(step, stage), df, nullptr, nullptr, Lq, Lv, Lf, nullptr, &F))) return st;
    }
    if (step == nsteps - 1) {
      if (w->dtype == RBD_F64) HIP_TRY(launch_mk_stage<double>(w->dm, B, 4, dt, dq, dv, nullptr, w->mk, Lq, Lv, w->stream));
      else HIP_TRY(launch_mk_stage<float>(w->dm, B, 4, dt, dq, dv, nullptr, w->mk, Lq, Lv, w->stream));
    }
  }
  if (!fused && pd && (st = ensure(&w->d_tauwork, &w->d_tauwork_bytes, entry))) return st;
  for (int step = 0; !fused && step < nsteps; ++step) {
    // the closing stage of a step rides in the stage-0 launch of the next one (as in the fused kernels); only the last step closes on its own
    for (int stage = 0; stage < 4; ++stage) {
      const int close_prev = (stage == 0 && step > 0) ? 1 : 0;
      HIP_TRY(stage_launch(w, B, stage, dt, dq, dv, w->d_vdwork, Lq, Lv, close_prev));
      const void* ts = tau_at(step, stage);
      if (pd) {  // the PD law on the stage state the launch above left in (q, v): one element-wise launch, no host round trip
        if (m->big) {  // (trees of more than 64 bodies: the same law over the any-size tables, round 6)
          if (w->dtype == RBD_F64) HIP_TRY(launch_big_pd_control<double>(w->big, B, dq, dv, ts, ctl.q_des, ctl.kp, ctl.kd, w->d_tauwork, Lq, Lv, w->stream));
          else HIP_TRY(launch_big_pd_control<float>(w->big, B, dq, dv, ts, ctl.q_des, ctl.kp, ctl.kd, w->d_tauwork, Lq, Lv, w->stream));
        } else
        if (w->dtype == RBD_F64) HIP_TRY(launch_pd_control<double>(w->dm, B, dq, dv, ts, ctl.q_des, ctl.kp, ctl.kd, w->d_tauwork, Lq, Lv, w->stream));
        else HIP_TRY(launch_pd_control<float>(w->dm, B, dq, dv, ts, ctl.q_des, ctl.kp, ctl.kd, w->d_tauwork, Lq, Lv, w->stream));
        ts = w->d_tauwork;
      }
      if ((st = run_dynamics(w, B, od, dq, dv, ts, df, w->d_vdwork, nullptr, nullptr))) return st;
    }
    if (step == nsteps - 1) HIP_TRY(stage_launch(w, B, 4, dt, dq, dv, w->d_vdwork, Lq, Lv));
  }
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_out_copy(w, q, dq, es * m->nq * B)) || (st = stage_out_copy(w, v, dv, es * m->nv * B))) return st;
  }
  return RBD_OK;
}
int rbd_simulate(rbd_ws_t* w, int32_t B, void* q, void* v, const void* tau, const void* fext, double dt, int32_t nsteps, const rbd_opts_t* opts) {
  BigOk big_ok;
  rbd_control_t ctl{};
  ctl.kind = RBD_CONTROL_CONSTANT;
  ctl.tau = tau;
  return simulate_core(w, B, q, v, ctl, fext, dt, nsteps, opts);
}
int rbd_simulate_controlled(rbd_ws_t* w, int32_t B, void* q, void* v, const rbd_control_t* control, const void* fext, double dt, int32_t nsteps,
                            const rbd_opts_t* opts) {
  BigOk big_ok;
  if (!control) return RBD_ERR_INVALID_ARGUMENT;
  return simulate_core(w, B, q, v, *control, fext, dt, nsteps, opts);
}


// ---- soft contact ------------------------------------------------------------------------------------------------------------------
int rbd_model_contact_dims(const rbd_model_t* m, int32_t* n_contact_points, int32_t* n_halfspaces, int32_t* n_additional_states) {
  if (!m) return RBD_ERR_INVALID_ARGUMENT;
  if (n_contact_points) *n_contact_points = m->ncp;
  if (n_halfspaces) *n_halfspaces = m->nhs;
  if (n_additional_states) *n_additional_states = 3 * m->ncp * m->nhs;  // num_additional_states (src/mechanism.jl:143-149)
  return RBD_OK;
}

// contact_dynamics! on device pointers: per-body kinematics (the RNEA launch exports them; its bias torques go to the workspace),
// then the contact kernel.  dcw / dtw nullable.
static int run_contact(rbd_ws* w, int32_t B, const Opts& o, const void* dq, const void* dv, void* ds, void* dsd, const void* df, void* dcw, void* dtw) {
  const rbd_model* m = w->model;
  const size_t es = esize(w);
  int st;
  if ((st = ensure(&w->d_body, &w->d_body_bytes, es * (size_t)m->nb * 24 * B)) || (st = ensure(&w->d_c, &w->d_c_bytes, es * (size_t)m->nv * B))) return st;
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), Lf = layout_of(o.layout, 6L * m->nb, B);
  const Layout Ls = layout_of(o.layout, 3L * m->ncp * m->nhs, B);
  if (m->big) {  // more than 64 bodies: the per-body kinematics from the any-size kernels
    if ((st = big_scratch(w, B))) return st;
    if (w->dtype == RBD_F64) {
      HIP_TRY(launch_big_rnea<double>(w->big, B, dq, dv, nullptr, nullptr, w->d_c, nullptr, w->d_big_scratch, nullptr, nullptr, Lq, Lv, Lf, w->stream));
      HIP_TRY(launch_big_export_body<double>(w->big, B, w->d_big_scratch, w->d_body, w->stream));
      HIP_TRY(launch_contact<double>(w->ctm, B, w->d_body, ds, dsd, df, dcw, dtw, Ls, Lf, w->stream));
    } else {
      HIP_TRY(launch_big_rnea<float>(w->big, B, dq, dv, nullptr, nullptr, w->d_c, nullptr, w->d_big_scratch, nullptr, nullptr, Lq, Lv, Lf, w->stream));
      HIP_TRY(launch_big_export_body<float>(w->big, B, w->d_big_scratch, w->d_body, w->stream));
      HIP_TRY(launch_contact<float>(w->ctm, B, w->d_body, ds, dsd, df, dcw, dtw, Ls, Lf, w->stream));
    }
    return RBD_OK;
  }
  if (w->dtype == RBD_F64) {
    HIP_TRY(launch_rnea<double>(w->dm, B, dq, dv, nullptr, nullptr, w->d_c, nullptr, w->d_body, Lq, Lv, Lf, w->stream));
    HIP_TRY(launch_contact<double>(w->ctm, B, w->d_body, ds, dsd, df, dcw, dtw, Ls, Lf, w->stream));
  } else {
    HIP_TRY(launch_rnea<float>(w->dm, B, dq, dv, nullptr, nullptr, w->d_c, nullptr, w->d_body, Lq, Lv, Lf, w->stream));
    HIP_TRY(launch_contact<float>(w->ctm, B, w->d_body, ds, dsd, df, dcw, dtw, Ls, Lf, w->stream));
  }
  return RBD_OK;
}

static int contact_scope(const rbd_ws* w, const Opts& o) {
  if (w->model->ncp == 0 || w->model->nhs == 0) return RBD_ERR_INVALID_ARGUMENT;  // nothing to do: use rbd_dynamics
  if (w->model->nloops > 0) return RBD_ERR_UNSUPPORTED;
  if (o.memory != RBD_MEM_DEVICE) return RBD_ERR_UNSUPPORTED;  // device pointers only
  return RBD_OK;
}

int rbd_contact_dynamics(rbd_ws_t* w, int32_t B, const void* q, const void* v, void* s, void* contactwrenches, void* sdot, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  if ((st = contact_scope(w, o))) return st;
  if (!q || !v || !s) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  return run_contact(w, B, o, q, v, s, sdot, nullptr, contactwrenches, nullptr);
}

int rbd_dynamics_contact(rbd_ws_t* w, int32_t B, const void* q, const void* v, void* s, const void* tau, const void* fext, void* vdot, void* qdot,
                         void* sdot, void* contactwrenches, void* totalwrenches, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  if ((st = contact_scope(w, o))) return st;
  if (!q || !v || !s || !vdot) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const rbd_model* m = w->model;
  void* dtw = totalwrenches;
  if (!dtw) {
    if ((st = ensure(&w->d_tw, &w->d_tw_bytes, esize(w) * (size_t)6 * m->nb * B))) return st;
    dtw = w->d_tw;
  }
  if ((st = run_contact(w, B, o, q, v, s, sdot, fext, contactwrenches, dtw))) return st;
  return run_dynamics(w, B, o, q, v, tau, dtw, vdot, qdot, nullptr);
}

int rbd_simulate_contact(rbd_ws_t* w, int32_t B, void* q, void* v, void* s, const void* tau, const void* fext, double dt, int32_t nsteps,
                         const rbd_opts_t* opts) {
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  if ((st = contact_scope(w, o))) return st;
  if (!q || !v || !s || nsteps < 0 || !(dt > 0)) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0 || nsteps == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const rbd_model* m = w->model;
  const size_t es = esize(w);
  const long ns = 3L * m->ncp * m->nhs * B;
  if ((st = mk_ensure(w, B)) || (st = ensure(&w->d_tw, &w->d_tw_bytes, es * (size_t)6 * m->nb * B)) || (st = ensure(&w->d_s0, &w->d_s0_bytes, es * (size_t)ns)) ||
      (st = ensure(&w->d_sacc, &w->d_sacc_bytes, es * (size_t)ns)) || (st = ensure(&w->d_sdot, &w->d_sdot_bytes, es * (size_t)ns)))
    return st;
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B);
  for (int step = 0; step < nsteps; ++step) {
    // MuntheKaasIntegrator.step (src/ode_integrators.jl:233-299): (q, v) through mk_stage_kernel, s beside them with the same tableau
    for (int stage = 0; stage <= 4; ++stage) {
      if (w->dtype == RBD_F64) {
        HIP_TRY(launch_mk_stage<double>(w->dm, B, stage, dt, q, v, w->d_vdwork, w->mk, Lq, Lv, w->stream));
        HIP_TRY(launch_contact_stage<double>(ns, stage, dt, s, w->d_sdot, w->d_s0, w->d_sacc, w->stream));
      } else {
        HIP_TRY(launch_mk_stage<float>(w->dm, B, stage, dt, q, v, w->d_vdwork, w->mk, Lq, Lv, w->stream));
        HIP_TRY(launch_contact_stage<float>(ns, stage, dt, s, w->d_sdot, w->d_s0, w->d_sacc, w->stream));
      }
      if (stage < 4) {
        if ((st = run_contact(w, B, o, q, v, s, w->d_sdot, fext, nullptr, w->d_tw))) return st;
        if ((st = run_dynamics(w, B, o, q, v, tau, w->d_tw, w->d_vdwork, nullptr, nullptr))) return st;
      }
    }
  }
  return RBD_OK;
}

int rbd_cholesky_solve(rbd_ws_t* w, int32_t B, const void* M, const void* rhs, void* x, void* L_out, const rbd_opts_t* opts) {
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  if (!M || !rhs || !x || o.memory != RBD_MEM_DEVICE) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const rbd_model* m = w->model;
  const Layout Lv = layout_of(o.layout, m->nv, B), Lm = layout_of(o.layout, (long)m->nv * m->nv, B);
  Timed t(w);
  if (w->dtype == RBD_F64) HIP_TRY(launch_chol_solve<double>(m->nv, B, M, rhs, nullptr, x, L_out, Lm, Lv, w->d_notpd, w->stream));
  else HIP_TRY(launch_chol_solve<float>(m->nv, B, M, rhs, nullptr, x, L_out, Lm, Lv, w->d_notpd, w->stream));
  return RBD_OK;
}


int rbd_kinematics(rbd_ws_t* w, int32_t B, const void* q, const void* v, void* momentum_matrix, void* com, void* energy, const rbd_opts_t* opts) {
  BigOk big_ok;  // (round 6: trees of more than 64 bodies through big_kin_kernel)
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || (energy && missing(v, m->nv))) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const void *dq = q, *dv = v;
  void *dA = momentum_matrix, *dcom = com, *den = energy;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_in(w, 1, v, es * m->nv * B, &dv)) ||
        (st = stage_out_alloc(w, 4, momentum_matrix, es * 6 * m->nv * B, &dA)) || (st = stage_out_alloc(w, 5, com, es * 3 * B, &dcom)) ||
        (st = stage_out_alloc(w, 6, energy, es * 2 * B, &den)))
      return st;
  }
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), La = layout_of(o.layout, 6L * m->nv, B);
  const Layout L3 = layout_of(o.layout, 3, B), L2 = layout_of(o.layout, 2, B);
  w->last_kernel = "kin_kernel";
  if (!m->big && B >= w->spec_kin_min_batch) spec_load(w, SPEC_KIN, false);
  if (!m->big && B >= w->spec_kin_min_batch && w->spec_kin && w->spec_energy && w->spec_com) {
    // large batches: one lane per state, compiled for the mechanism — the momentum matrix (with the centre of mass) and the energies (with the centre of mass
    // when the matrix is not asked for) are a walk each, the centre of mass alone the lightest one: what a walk carries is decided at compile time (rbd_spec.hpp
    // kin_spec)
    Timed t(w);
    if (dA) HIP_TRY(launch_kin_spec(w, w->spec_kin, B, dq, nullptr, dA, dcom, nullptr, nullptr, 0, 0, nullptr, Lq, Lv, La, L3, L2, L2));
    if (den) HIP_TRY(launch_kin_spec(w, w->spec_energy, B, dq, dv, nullptr, dA ? nullptr : dcom, den, nullptr, 0, 0, nullptr, Lq, Lv, La, L3, L2, L2));
    if (!dA && !den && dcom) HIP_TRY(launch_kin_spec(w, w->spec_com, B, dq, nullptr, nullptr, dcom, nullptr, nullptr, 0, 0, nullptr, Lq, Lv, La, L3, L2, L2));
    w->last_kernel = "kin_spec (compiled for the mechanism at run time)";
  } else
  if (m->big) {
    if ((st = big_scratch(w, B))) return st;
    Timed t(w);
    w->last_kernel = "big_kin_kernel";
    if (w->dtype == RBD_F64) HIP_TRY(launch_big_kin<double>(w->big, B, dq, dv, dA, dcom, den, nullptr, -1, -1, nullptr, w->d_big_scratch, Lq, Lv, La, L3, L2, L2, w->stream));
    else HIP_TRY(launch_big_kin<float>(w->big, B, dq, dv, dA, dcom, den, nullptr, -1, -1, nullptr, w->d_big_scratch, Lq, Lv, La, L3, L2, L2, w->stream));
  } else {
    Timed t(w);
    if (w->dtype == RBD_F64) HIP_TRY(launch_kin<double>(w->dm, B, dq, dv, dA, dcom, den, nullptr, 0, 0, Lq, Lv, La, L3, L2, w->stream));
    else HIP_TRY(launch_kin<float>(w->dm, B, dq, dv, dA, dcom, den, nullptr, 0, 0, Lq, Lv, La, L3, L2, w->stream));
  }
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_out_copy(w, momentum_matrix, dA, es * 6 * m->nv * B)) || (st = stage_out_copy(w, com, dcom, es * 3 * B)) ||
        (st = stage_out_copy(w, energy, den, es * 2 * B)))
      return st;
  }
  return RBD_OK;
}


int rbd_geometric_jacobian(rbd_ws_t* w, int32_t B, const void* q, int32_t base_body, int32_t target_body, void* jac, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || missing(jac, m->nv)) return RBD_ERR_INVALID_ARGUMENT;
  if (base_body < -1 || base_body >= m->nb || target_body < -1 || target_body >= m->nb) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  // path(mechanism, base, target): walk both ends up to the lowest common ancestor (src/graphs/tree_path.jl:41-63); the joints on the
  // base side are traversed upwards (-S), those on the target side downwards (+S)
  uint64_t plus = 0, minus = 0;
  int a = base_body, b = target_body;
  while (!m->big && a != b) {  // (the any-size kernel walks the path itself)
    if (a > b) { minus |= (uint64_t)1 << m->slot_of[a]; a = m->parent_ref[a]; }
    else { plus |= (uint64_t)1 << m->slot_of[b]; b = m->parent_ref[b]; }
  }
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), La = layout_of(o.layout, 6L * m->nv, B);
  const Layout L3 = layout_of(o.layout, 3, B), L2 = layout_of(o.layout, 2, B);
  const size_t es = esize(w);
  const void* dq = q;
  void* dJ = jac;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_out_alloc(w, 4, jac, es * 6 * m->nv * B, &dJ))) return st;
  }
  w->last_kernel = "kin_kernel";
  if (!m->big && B >= w->spec_kin_min_batch) spec_load(w, SPEC_KIN, false);
  if (!m->big && B >= w->spec_kin_min_batch && w->spec_jac) {
    Timed t(w);
    HIP_TRY(launch_kin_spec(w, w->spec_jac, B, dq, nullptr, nullptr, nullptr, nullptr, dJ, plus, minus, nullptr, Lq, Lv, La, L3, L2, L2));
    w->last_kernel = "jac_spec (compiled for the mechanism at run time)";
  } else
  if (m->big) {
    if ((st = big_scratch(w, B))) return st;
    Timed t(w);
    w->last_kernel = "big_kin_kernel";
    if (w->dtype == RBD_F64) HIP_TRY(launch_big_kin<double>(w->big, B, dq, nullptr, nullptr, nullptr, nullptr, dJ, base_body, target_body, nullptr, w->d_big_scratch, Lq, Lv, La, L3, L2, L2, w->stream));
    else HIP_TRY(launch_big_kin<float>(w->big, B, dq, nullptr, nullptr, nullptr, nullptr, dJ, base_body, target_body, nullptr, w->d_big_scratch, Lq, Lv, La, L3, L2, L2, w->stream));
  } else {
    Timed t(w);
    if (w->dtype == RBD_F64) HIP_TRY(launch_kin<double>(w->dm, B, dq, nullptr, nullptr, nullptr, nullptr, dJ, plus, minus, Lq, Lv, La, L3, L2, w->stream));
    else HIP_TRY(launch_kin<float>(w->dm, B, dq, nullptr, nullptr, nullptr, nullptr, dJ, plus, minus, Lq, Lv, La, L3, L2, w->stream));
  }
  if (o.memory == RBD_MEM_HOST) return stage_out_copy(w, jac, dJ, es * 6 * m->nv * B);
  return RBD_OK;
}


int rbd_momentum(rbd_ws_t* w, int32_t B, const void* q, const void* v, void* out12, const rbd_opts_t* opts) {
  BigOk big_ok;
  const Opts o = read_opts(opts);
  int st = check_common(w, B, o);
  if (st != RBD_OK) return st;
  const rbd_model* m = w->model;
  if (missing(q, m->nq) || missing(v, m->nv) || !out12) return RBD_ERR_INVALID_ARGUMENT;
  if (B == 0) return RBD_OK;
  HIP_TRY(hipSetDevice(w->device));
  const size_t es = esize(w);
  const void *dq = q, *dv = v;
  void* dout = out12;
  if (o.memory == RBD_MEM_HOST) {
    if ((st = stage_in(w, 0, q, es * m->nq * B, &dq)) || (st = stage_in(w, 1, v, es * m->nv * B, &dv)) ||
        (st = stage_out_alloc(w, 4, out12, es * 12 * B, &dout)))
      return st;
  }
  const Layout Lq = layout_of(o.layout, m->nq, B), Lv = layout_of(o.layout, m->nv, B), L12 = layout_of(o.layout, 12, B);
  w->last_kernel = "momentum_kernel";
  if (!m->big && B >= w->spec_kin_min_batch) spec_load(w, SPEC_KIN, false);
  if (!m->big && B >= w->spec_kin_min_batch && w->spec_mom) {
    Timed t(w);
    HIP_TRY(launch_kin_spec(w, w->spec_mom, B, dq, dv, nullptr, nullptr, nullptr, nullptr, 0, 0, dout, Lq, Lv, Lq, Lq, Lq, L12));
    w->last_kernel = "mom_spec (compiled for the mechanism at run time)";
  } else
  if (m->big) {
    if ((st = big_scratch(w, B))) return st;
    Timed t(w);
    w->last_kernel = "big_kin_kernel";
    if (w->dtype == RBD_F64) HIP_TRY(launch_big_kin<double>(w->big, B, dq, dv, nullptr, nullptr, nullptr, nullptr, -1, -1, dout, w->d_big_scratch, Lq, Lv, Lq, Lq, Lq, L12, w->stream));
    else HIP_TRY(launch_big_kin<float>(w->big, B, dq, dv, nullptr, nullptr, nullptr, nullptr, -1, -1, dout, w->d_big_scratch, Lq, Lv, Lq, Lq, Lq, L12, w->stream));
  } else {
    Timed t(w);
    if (w->dtype == RBD_F64) HIP_TRY(launch_momentum<double>(w->dm, B, dq, dv, dout, Lq, Lv, L12, w->stream));
    else HIP_TRY(launch_momentum<float>(w->dm, B, dq, dv, dout, Lq, Lv, L12, w->stream));
  }
  if (o.memory == RBD_MEM_HOST) return stage_out_copy(w, out12, dout, es * 12 * B);
  return RBD_OK;
}

}  // extern "C"
