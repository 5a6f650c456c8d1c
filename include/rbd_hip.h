/*
 * rbd_hip.h — C ABI of librbd_hip.so, the MI355X (gfx950) batched rigid-body
 * dynamics engine that replaces the data-parallel hot path of
 * RigidBodyDynamics.jl (reference paths are relative to the upstream repo):
 *
 *   rbd_dynamics          <->  dynamics!(result, state, τ, wext)       src/mechanism_algorithms.jl:845-864
 *   rbd_inverse_dynamics  <->  inverse_dynamics!(τ, jw, acc, state, v̇, wext)  src/mechanism_algorithms.jl:542-553
 *   rbd_dynamics_bias     <->  dynamics_bias!(result, state)           src/mechanism_algorithms.jl:484-498
 *   rbd_mass_matrix       <->  mass_matrix!(M::Symmetric, state)       src/mechanism_algorithms.jl:248-272
 *   rbd_mass_matrix_solve <->  dynamics_solve!(result, τ) (no-loop branch) src/mechanism_algorithms.jl:747-822
 *   rbd_model_create      <->  MechanismState(mechanism) index tables  src/mechanism_state.jl:79-172
 *   rbd_simulate          <->  simulate(state0, T; Δt) / MuntheKaasIntegrator.step  src/simulate.jl:36-55, src/ode_integrators.jl:233-299
 *
 * The reference is pure Julia and has no FFI layer; this header is what a
 * Julia `ccall` shim (julia/RigidBodyDynamicsGPU.jl) binds.
 *
 * Conventions
 *  - every entry point returns an int status (RBD_OK == 0); nothing throws or
 *    longjmps across the boundary; caller buffers are never owned by the library;
 *  - a *state* is one (q, v) of the mechanism; a *batch* is B independent states
 *    of the same mechanism (one evaluation per state);
 *  - batch buffers are DEVICE pointers of the workspace's dtype unless
 *    opts->memory == RBD_MEM_HOST (then the library stages through its own
 *    device buffers; PCIe time is then inside the call);
 *  - layout RBD_LAYOUT_SOA: element k of state b lives at x[k*B + b]
 *    (coordinate-major, coalesced);  RBD_LAYOUT_AOS: x[b*n + k] — the Julia
 *    `n × B` column-major matrix with one state per column;
 *  - motion vectors are (angular; linear), force vectors (torque; force), as in
 *    src/spatial/spatialmotion.jl:122-153 and src/spatial/spatialforce.jl:73-111;
 *  - calls are asynchronous on the workspace's HIP stream; rbd_sync() waits.
 *    (Since 600 one exception per program and workspace: the FIRST result of a
 *    kernel compiled for the mechanism at run time — dynamics!, inverse_dynamics!,
 *    fp32 mass_matrix! + solve — is compared with the interpreting kernel on the
 *    call's first states: a small allocation and one synchronisation in that call,
 *    never inside a stream capture; a program that differs is dropped, the call
 *    recomputed, rbd_last_hip_error says so.)
 *    A model handle is immutable and shareable; a workspace must not be used
 *    from two host threads at once (same rule as MechanismState/DynamicsResult,
 *    which are mutable caches: src/mechanism_state.jl:35-78).
 */
#ifndef RBD_HIP_H
#define RBD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
enum {
  RBD_OK = 0,
  RBD_ERR_INVALID_ARGUMENT = 1,   /* Julia side: ArgumentError           */
  RBD_ERR_DIMENSION_MISMATCH = 2, /* Julia side: DimensionMismatch       */
  RBD_ERR_UNSUPPORTED = 3,        /* joint type / feature not built      */
  RBD_ERR_NO_DEVICE = 4,          /* no HIP device: the product has NO CPU fallback */
  RBD_ERR_HIP = 5,                /* a HIP runtime call failed (see rbd_last_hip_error) */
  RBD_ERR_OUT_OF_MEMORY = 6,
  RBD_ERR_HAS_LOOPS = 7,          /* inverse_dynamics! on a mechanism with loop joints
                                     (src/mechanism_algorithms.jl:549)   */
  RBD_ERR_NOT_POSITIVE_DEFINITE = 8
};

/* ---- joint types (src/joint_types/ *.jl) ---------------------------------- */
enum {
  RBD_JOINT_FIXED = 0,          /* fixed.jl            nq=0 nv=0 */
  RBD_JOINT_REVOLUTE = 1,       /* revolute.jl         nq=1 nv=1 */
  RBD_JOINT_PRISMATIC = 2,      /* prismatic.jl        nq=1 nv=1 */
  RBD_JOINT_QUAT_FLOATING = 3,  /* quaternion_floating.jl nq=7 (w,x,y,z,px,py,pz) nv=6 (ω; v) body frame */
  RBD_JOINT_PLANAR = 4,         /* planar.jl           nq=3 nv=3 */
  RBD_JOINT_QUAT_SPHERICAL = 5, /* quaternion_spherical.jl nq=4 nv=3 */
  RBD_JOINT_SINCOS_REVOLUTE = 6 /* sin_cos_revolute.jl nq=2 nv=1 */
};

enum { RBD_F64 = 0, RBD_F32 = 1 };
enum { RBD_LAYOUT_SOA = 0, RBD_LAYOUT_AOS = 1 };
enum { RBD_MEM_DEVICE = 0, RBD_MEM_HOST = 1 };
/* forward-dynamics algorithm behind rbd_dynamics */
enum {
  RBD_ALGO_ABA = 0,          /* fused articulated-body algorithm (default; tree mechanisms)   */
  RBD_ALGO_CRBA_CHOLESKY = 1,/* the reference's own route: bias-RNEA + CRBA + Cholesky; also
                                fills M and c in the workspace; the only route with loop joints */
  /* RBD_ALGO_ABA picks between three lane mappings of the same algorithm by how many wavefronts the batch makes of each
     (rbd_capi.hip run_aba); these force one (tests, benchmarks): */
  RBD_ALGO_ABA_LANES = 2,    /* one lane per (state, body), level-synchronous sweeps: small batches, every tree joint type          */
  RBD_ALGO_ABA_CHAINS = 3,   /* (round 1: chains of the tree on a few lanes per state.  Removed in round 3 — it lost at every batch
                                size; the value stays reserved and returns RBD_ERR_UNSUPPORTED)                                     */
  RBD_ALGO_ABA_BANKS = 4,    /* lane-per-body with two bodies per lane (levels split into two banks): twice the states per
                                wavefront; 1-dof / fixed tree joints, 6-dof joints on the world.  The bench workload (4096 Atlas states) */
  RBD_ALGO_ABA_TRACKS = 5,   /* (round 2: chains of the tree on a few lanes per state, per-body results in lane-private LDS rows.  Removed in
                                round 4 like RBD_ALGO_ABA_PIPE — both lost at every batch size; reserved, RBD_ERR_UNSUPPORTED)          */
  RBD_ALGO_ABA_WALK = 6,     /* the track schedule with one WAVEFRONT per track and one lane per state: a workgroup is up to four
                                wavefronts over the same 64 states, q / v / tau staged through LDS, per-joint results in accumulation
                                registers.  Trees of revolute / prismatic / fixed joints, 6-dof joints on the world, at most 11 steps per
                                track; RBD_ERR_UNSUPPORTED elsewhere or when the rows of 64 states do not fit one compute unit's LDS.
                                Large batches.  `simulate` on this mapping takes the four stages of a step in ONE launch (a looped instantiation of its own:
                                admitted statically — rbd_jit_check_walk_object — since 600)                                        */
  RBD_ALGO_ABA_PIPE = 7,     /* (round 2: a body-step cut into stages on the four SIMDs of a compute unit.  Removed; reserved)     */
  RBD_ALGO_ABA_COMPILED = 8  /* one lane per state, straight-line code compiled for the mechanism at run time (rbd_jit_* below): fp32,
                                trees of every joint type above (since 400: Planar, QuaternionSpherical, QuaternionFloating below the world
                                too), no loop joints, nv <= 64; RBD_ALGO_ABA picks it from half a chip-full of wavefronts up.  Since 600 also
                                fp64 for the mechanisms no walk kernel takes (3-dof joints, 6-dof joints below the world — the reference's
                                randmech(): 65 536 states 177 us against 684 one body per lane; two programs — every row in LDS, or the spare
                                rows in an HBM stash of the workspace and two wavefronts per CU — picked by batch); RBD_ALGO_ABA picks it from
                                8193 states.
                                RBD_ERR_UNSUPPORTED without hiprtc, outside that scope, or when the program's first result in this workspace
                                differs from the interpreting kernel's on the same states (first-use check, since 600: the program is
                                dropped, rbd_last_hip_error says so; RBD_ALGO_ABA recomputes the call on the interpreting kernels)    */
};

/* ---- loop (non-tree) joint: src/mechanism_modification.jl:38-43,
 *      constraint rows src/mechanism_algorithms.jl:574-673 ------------------ */
typedef struct rbd_loop_joint {
  int32_t predecessor;      /* moving-body index, -1 = world                          */
  int32_t successor;        /* moving-body index, -1 = world                          */
  int32_t joint_type;       /* RBD_JOINT_*                                            */
  int32_t _pad;
  double axis[3];           /* Revolute/Prismatic axis in frame_before                */
  double pred_rot[9];       /* joint_to_predecessor: frame_before -> predecessor body frame, row-major R */
  double pred_trans[3];
  double succ_rot[9];       /* joint_to_successor:  frame_after  -> successor body frame, row-major R   */
  double succ_trans[3];
  double rotation_from_z_aligned[9]; /* Revolute / Prismatic: rotation_from_z_aligned, src/joint_types/revolute.jl:12-17;
                                        Planar: the matrix with columns (x_axis, y_axis, x_axis × y_axis), planar.jl:22-35 */
  double gains[4];          /* Baumgarte SE3PDGains: angular k, d; linear k, d (default 100,20,100,20:
                               src/mechanism_algorithms.jl:610-612); all zero = stabilization off */
} rbd_loop_joint_t;

/* ---- soft contact (src/contact.jl): contact points on bodies against half-spaces of the environment --------------------------
 * DefaultContactPoint = ContactPoint{SoftContactModel{HuntCrossleyModel, ViscoelasticCoulombModel}} (src/contact.jl:196) and
 * HalfSpace3D (:202-222).  Every (contact point, half-space) pair owns 3 additional states (the tangential displacement of the
 * friction model; the normal model has none): s has 3 * n_contact_points * n_halfspaces entries per state of the batch, ordered
 * point-major as MechanismState lays them out (src/mechanism_state.jl:139-152: for body, for point, for half-space) — i.e. in the
 * order the points are listed here.                                                                                              */
typedef struct rbd_contact_point {
  int32_t body;            /* moving-body index                                                        */
  int32_t _pad;
  double location[3];      /* Point3D in the body's default frame (= frame_after of its joint)         */
  double hc_k, hc_lambda, hc_n;   /* HuntCrossleyModel k, λ, n   (hunt_crossley_hertz: λ = 3/2 α k, n = 3/2)   */
  double mu, k, b;         /* ViscoelasticCoulombModel μ, k, b                                         */
} rbd_contact_point_t;
typedef struct rbd_halfspace {
  double point[3];          /* root frame */
  double outward_normal[3]; /* root frame; normalized on model creation like HalfSpace3D's constructor */
} rbd_halfspace_t;

/* ---- the flattened mechanism ---------------------------------------------
 * Exactly the tables MechanismState builds once (src/mechanism_state.jl:85-118):
 * moving body i (0-based) is the successor of tree joint i; bodies are in the
 * reference's tree-joint order (topological: parents first). After
 * canonicalize_frame_definitions! (src/mechanism.jl:250-260) the body frame IS
 * frame_after(joint i) and joint_to_successor of tree joints is identity.      */
typedef struct rbd_flat_model {
  int32_t n_bodies;            /* number of moving bodies == tree joints               */
  int32_t nq, nv;              /* num_positions, num_velocities                        */
  int32_t n_loops;             /* non-tree joints                                      */
  const int32_t* parent;       /* [n_bodies] predecessor body of joint i, -1 = world   */
  const int32_t* joint_type;   /* [n_bodies] RBD_JOINT_*                               */
  const int32_t* q_offset;     /* [n_bodies] first q index (qranges)                   */
  const int32_t* v_offset;     /* [n_bodies] first v index (vranges)                   */
  const double* joint_axis;    /* [n_bodies*3] axis in frame_before (Revolute/Prismatic); Planar: x_axis */
  const double* joint_axis2;   /* [n_bodies*3] Planar y_axis (else ignored; may be NULL) */
  const double* pred_rot;      /* [n_bodies*9] joint_to_predecessor R, row-major (src/joint.jl:49) */
  const double* pred_trans;    /* [n_bodies*3]                                         */
  const double* inertia_moment;/* [n_bodies*9] J about the body-frame origin, row-major (src/spatial/motion_force_interaction.jl:28-37) */
  const double* inertia_cross; /* [n_bodies*3] cross_part = mass * com                 */
  const double* inertia_mass;  /* [n_bodies]                                           */
  double gravity[3];           /* gravitational_acceleration in the root frame (default 0,0,-9.81: src/mechanism.jl:1) */
  const rbd_loop_joint_t* loops; /* [n_loops] or NULL                                  */
  int32_t n_contact_points;    /* contact_points of all bodies (0: none, like every URDF-parsed mechanism) */
  int32_t n_halfspaces;        /* mechanism.environment.halfspaces                     */
  const rbd_contact_point_t* contact_points; /* [n_contact_points] or NULL, in the order of the additional state */
  const rbd_halfspace_t* halfspaces;         /* [n_halfspaces] or NULL                 */
} rbd_flat_model_t;

typedef struct rbd_model rbd_model_t; /* opaque, immutable after create               */
typedef struct rbd_ws rbd_ws_t;       /* opaque, one per host thread / HIP stream     */

typedef struct rbd_opts {
  int32_t layout;     /* RBD_LAYOUT_*  (default SOA)                                  */
  int32_t memory;     /* RBD_MEM_*     (default DEVICE)                               */
  int32_t algorithm;  /* RBD_ALGO_*    (rbd_dynamics only)                            */
  int32_t stabilization; /* 1 = Baumgarte stabilization with the workspace's gains (the model's — the
                            reference's default — unless rbd_workspace_set_loop_gains replaced them),
                            0 = `stabilization_gains=nothing`                         */
} rbd_opts_t;

/* ---- model / workspace lifetime ------------------------------------------ */
/* Sizes: mechanisms of up to 64 moving bodies with at most 8 children per body run on the wavefront-shaped kernels (any nv).  Mechanisms of MORE than 64 bodies, or
 * with a body of more than 8 children (since header 600; refused before), are accepted too and run on one-thread-per-state kernels with an HBM scratch (no speed
 * claim) — every entry point: rbd_dynamics with its loop branch, rbd_inverse_dynamics[_bodies], rbd_dynamics_bias[_bodies], rbd_mass_matrix[_solve], rbd_dynamics_result,
 * the contact entry points, rbd_simulate / rbd_simulate_controlled (the PD law included) / rbd_mk_stage (since 500), and since 600 the kinematics by-products
 * rbd_kinematics, rbd_geometric_jacobian, rbd_momentum.  The reference has no size limit (src/mechanism_algorithms.jl:28-50, :80-99, :313-327). */
int rbd_model_create(const rbd_flat_model_t* desc, rbd_model_t** out); /* deep-copies desc */
int rbd_model_destroy(rbd_model_t* model);
/* Introspection of the chain schedule under the track / walk plans: tracks per state, steps per pass, (lds_fields: 0, kept for ABI
 * stability), and the steps×tracks table of reference body indices (-1 = idle).  RBD_ERR_UNSUPPORTED when the mechanism is
 * outside the scope of those mappings.  Host-only, no device needed. */
/* ... and of the two-bodies-per-lane ("banked") mapping: lanes per state, the level at which bank 1 starts, bodies per bank, and whether
 * the banked ABA (not only the banked RNEA) takes the mechanism.  RBD_ERR_UNSUPPORTED when the split would not save lanes. */
int rbd_model_bank_plan(const rbd_model_t* model, int32_t* lanes, int32_t* first_level_of_bank1, int32_t* bodies_bank0, int32_t* bodies_bank1,
                        int32_t* aba_in_scope);
/* ... and of the track plan under the walk kernels (RBD_ALGO_ABA_WALK): dims[6] = tracks per state, steps, A/C mailboxes, B mailboxes, has a 6-dof root,
 * has prismatic/fixed joints; table: steps×tracks reference body indices (-1 = idle); ri / rr: the packed per-(step, track) records
 * the kernel reads (rbd_device.hpp TI_*, TR_*) — what tests/emu feeds to the CPU emulation of the kernel's step code.        */
int rbd_model_track_plan(const rbd_model_t* model, int32_t* dims, int32_t* table, int32_t table_cap, int32_t* ri, int32_t ri_cap, double* rr, int32_t rr_cap);
/* ... and of the plan the walk kernels run when the mechanism hangs on the world by one 6-dof joint and is shallower seen from another
 * body: the tree RE-ROOTED at its centre (csrc/rbd_reroot.hpp; the floating joint's coordinates stay where they are).
 * dims[12] = tracks, steps, A/C mailboxes, B mailboxes, 6-dof root, prismatic/fixed joints, parking slots, chain length, q / v offset of the
 * floating joint, new root body, old floating body;  ri: packed records followed by the per-step flags;  rr: constants;  wk: parking words
 * (BFD_* flags in bits 8..9);  chain_i[4·chain], chain_r[15·chain], fxp[12]: the joints between the old floating body and the new root.
 * RBD_ERR_UNSUPPORTED when the mechanism is not re-rooted.  Host-only. */
int rbd_model_reroot_plan(const rbd_model_t* model, int32_t* dims, int32_t* ri, int32_t ri_cap, double* rr, int32_t rr_cap, int32_t* wk, int32_t wk_cap,
                          int32_t* chain_i, double* chain_r, double* fxp);
int rbd_model_chain_plan(const rbd_model_t* model, int32_t* tracks, int32_t* steps, int32_t* lds_fields, int32_t* table, int32_t capacity);
int rbd_model_dims(const rbd_model_t* model, int32_t* n_bodies, int32_t* nq, int32_t* nv, int32_t* nc);

/* stream: a hipStream_t passed as void* (NULL = the device's default stream) */
int rbd_workspace_create(const rbd_model_t* model, int32_t max_batch, int32_t device,
                         int32_t dtype, void* stream, rbd_ws_t** out);
int rbd_workspace_destroy(rbd_ws_t* ws);
int rbd_workspace_set_stream(rbd_ws_t* ws, void* stream);
/* `stabilization_gains` as an AbstractDict{JointID, SE3PDGains} / ConstDict (src/mechanism_algorithms.jl:614-632, :848; src/simulate.jl:37): the Baumgarte
 * gains of the calls that FOLLOW on this workspace (rbd_dynamics, rbd_simulate*, rbd_mk_stage with opts->stabilization = 1).  gains: HOST array of
 * 4 × n_loops doubles — (angular k, angular d, linear k, linear d) for every loop joint in the order of rbd_flat_model_t.loops; NULL restores the
 * model's (rbd_loop_joint_t.gains).  Cheap when the gains are the ones already in force (no device access), so a binding may call it before every
 * dynamics!; otherwise it waits for the workspace's stream.  Non-finite gains: RBD_ERR_INVALID_ARGUMENT.  No-op for tree mechanisms. */
int rbd_workspace_set_loop_gains(rbd_ws_t* ws, const double* gains);
int rbd_sync(rbd_ws_t* ws);

/* ---- the hot path ----------------------------------------------------------
 * q[nq×B], v[nv×B], tau[nv×B] (NULL => zeros, like the ConstVector default),
 * fext[6*n_bodies×B] external wrench on each moving body in the ROOT frame,
 * (torque; force) (NULL => none, like NullDict); outputs vdot[nv×B],
 * qdot[nq×B] (nullable), lambda[nc×B] (nullable).                              */
/* Mechanisms WITH contact points and an environment: rbd_dynamics / rbd_simulate / rbd_mk_stage return RBD_ERR_UNSUPPORTED — the reference's
 * dynamics! always adds the contact wrenches (src/mechanism_algorithms.jl:849-856), which needs the additional state: use
 * rbd_dynamics_contact / rbd_simulate_contact. */
int rbd_dynamics(rbd_ws_t* ws, int32_t B, const void* q, const void* v, const void* tau,
                 const void* fext, void* vdot, void* qdot, void* lambda, const rbd_opts_t* opts);

/* tau_out = M(q) vdot + c(q, v, fext); tree mechanisms only */
int rbd_inverse_dynamics(rbd_ws_t* ws, int32_t B, const void* q, const void* v, const void* vdot,
                         const void* fext, void* tau_out, const rbd_opts_t* opts);

/* c_out = c(q, v, fext) = inverse dynamics with vdot = 0 */
int rbd_dynamics_bias(rbd_ws_t* ws, int32_t B, const void* q, const void* v, const void* fext,
                      void* c_out, const rbd_opts_t* opts);

/* The same two entry points with the per-body outputs the reference fills beside the torques (every output nullable):
 *   jointwrenches_out[6*n_bodies×B]  jointwrenchesout of inverse_dynamics! / result.jointwrenches of dynamics_bias! — the wrench across
 *                                    the joint above each body after joint_wrenches_and_torques! (src/mechanism_algorithms.jl:442-459);
 *   accelerations_out[6*n_bodies×B]  accelerations of inverse_dynamics! (spatial_accelerations!, :387-417: root acceleration −g included)
 *                                    / result.accelerations of dynamics_bias! (bias_accelerations!, :377-385);
 * both in the ROOT frame, (angular; linear) / (torque; force), bodies in the flat model's order, layout as fext.  With them
 * DynamicsResult's accelerations / jointwrenches / totalwrenches (src/dynamics_result.jl:26-29) are complete: totalwrenches of a
 * mechanism without contact points is the caller's own fext (mechanism_algorithms.jl:851-855).
 * Large batches: the batch-innermost layout (RBD_LAYOUT_SOA) is the cheap one for these outputs — Atlas, 65 536 states, both outputs: fp32 51 us
 * (state-major 100 us: the kernel stores batch-innermost scratch and a second kernel moves it), fp64 78 us (state-major 118 us).      */
int rbd_inverse_dynamics_bodies(rbd_ws_t* ws, int32_t B, const void* q, const void* v, const void* vdot, const void* fext, void* tau_out,
                                void* jointwrenches_out, void* accelerations_out, const rbd_opts_t* opts);
int rbd_dynamics_bias_bodies(rbd_ws_t* ws, int32_t B, const void* q, const void* v, const void* fext, void* c_out, void* jointwrenches_out,
                             void* accelerations_out, const rbd_opts_t* opts);

/* M_out: nv×nv column-major per state (element (i,j) of state b at
 * M[(j*nv+i)*B + b] for SOA, M[b*nv*nv + j*nv + i] for AOS). Like the
 * reference (Symmetric, uplo 'L') the LOWER triangle i>=j is the result; the strict upper triangle is not to
 * be read (the reference leaves it undefined; most paths do not touch it, the fp32 path of
 * rbd_mass_matrix_solve on the kernels compiled for the mechanism — column-per-state callers, from 256 states since 600 (32 768 before: it is ahead at
 * every batch) — writes the mirror image there because whole cache lines leave the chip faster). */
int rbd_mass_matrix(rbd_ws_t* ws, int32_t B, const void* q, void* M_out, const rbd_opts_t* opts);

/* x = M(q)^-1 rhs; rhs, x: nv×B.  opts->algorithm == RBD_ALGO_CRBA_CHOLESKY: CRBA + batched lower Cholesky, the potrf/potrs
 * of dynamics_solve!;  RBD_ALGO_ABA (default): the O(n) articulated-body solve (forward dynamics with v = 0, g = 0,
 * tau = rhs), which never forms M.  M_out nullable (same layout as rbd_mass_matrix); tree mechanisms.              */
int rbd_mass_matrix_solve(rbd_ws_t* ws, int32_t B, const void* q, const void* rhs, void* x,
                          void* M_out, const rbd_opts_t* opts);
/* ... (CRBA + Cholesky) with M as LAPACK's PACKED lower triangle: element (i, j), i >= j, of state b at index i + j (2 nv − j − 1) / 2 of its
 * np = nv (nv + 1) / 2 values — M_packed_out: np × B in opts->layout (AOS: the np values of a state contiguous).  This is the part of M the reference defines
 * (`Symmetric(…, :L)`: src/dynamics_result.jl:42, mass_matrix! :248-272) and half the bytes of the square, which is what the emission of M costs at large
 * batches (Atlas, 65 536 fp32 states: 175 MB instead of 357 MB).  SURVEY.md §8(b) "lower triangle valid; or packed".                                    */
int rbd_mass_matrix_solve_packed(rbd_ws_t* ws, int32_t B, const void* q, const void* rhs, void* x, void* M_packed_out, const rbd_opts_t* opts);

/* The dense step of dynamics_solve! on its own: L = potrf!('L', M) and x = potrs!(L, rhs) for B caller-provided nv×nv SPD
 * matrices (lower triangles read; device pointers).  L_out (nullable) receives the factor — DynamicsResult.L
 * (src/dynamics_result.jl:32).  fp32 with nv <= 40 runs on the matrix cores (chol_mfma_kernel).                        */
int rbd_cholesky_solve(rbd_ws_t* ws, int32_t B, const void* M, const void* rhs, void* x, void* L_out, const rbd_opts_t* opts);

/* after rbd_dynamics(..., RBD_ALGO_CRBA_CHOLESKY): copy the DynamicsResult side
 * products out of the workspace (any pointer may be NULL). M: nv×nv (lower),
 * c: nv, K: nc×nv column-major, k: nc; layout per opts.                        */
int rbd_dynamics_result(rbd_ws_t* ws, int32_t B, void* M, void* c, void* K, void* k,
                        const rbd_opts_t* opts);
/* rbd_dynamics on the CRBA route leaves M and c in the workspace for rbd_dynamics_result to copy out.  A caller that wants them in its own buffers anyway
 * (the reference's dynamics!(result, ...) fills result.massmatrix / result.dynamicsbias in place) binds those buffers first: tree mechanisms then have M and c
 * written there directly — at 65 536 fp32 Atlas states the copy of M alone was a third of the call — and rbd_dynamics_result skips the bound fields.
 * Device pointers in the layout of the rbd_dynamics calls that follow; NULL unbinds.  Mechanisms with loop joints keep the workspace copies. */
int rbd_workspace_bind_result(rbd_ws_t* ws, void* M, void* c);

/* ---- the caller of the hot path: batched `simulate` --------------------------------------------
 * simulate(state0, final_time; Δt) src/simulate.jl:36-55 == MuntheKaasIntegrator.step (src/ode_integrators.jl:233-299) with
 * the runge_kutta_4 tableau (:48-55), for every state of the batch in lockstep, entirely on the device (no host round trip
 * per stage).  rbd_simulate: nsteps steps with constant tau / fext (the reference's default control is zero_torque!); q, v
 * are advanced in place.  rbd_mk_stage: one stage at a time for callers that evaluate a controller per stage —
 * stage 0 snapshots (q, v) as the base point and leaves the stage-1 state in (q, v); stage s = 1..3 takes the v̇ of the
 * previous stage state and leaves the next stage state in (q, v); stage 4 takes the last v̇ and leaves the end-of-step
 * state.  Loop mechanisms use the loop branch with opts->stabilization.                                              */
int rbd_simulate(rbd_ws_t* ws, int32_t B, void* q, void* v, const void* tau, const void* fext, double dt, int32_t nsteps,
                 const rbd_opts_t* opts);
int rbd_mk_stage(rbd_ws_t* ws, int32_t B, int32_t stage, double dt, void* q, void* v, const void* vdot_prev, const rbd_opts_t* opts);

/* ---- `simulate(state, T, control!)` without a host round trip per stage (src/simulate.jl:36-48: the reference calls control!(τ, t, state)
 *      before every stage's dynamics!).  Two device-side controller forms cover the common cases; anything else keeps the
 *      rbd_mk_stage + rbd_dynamics loop with the controller on the host.
 *   RBD_CONTROL_CONSTANT : τ = tau (nv × B, NULL = 0) — what rbd_simulate does.
 *   RBD_CONTROL_TABLE    : an open-loop τ(t): `tau` holds entries of nv × B each (batch layout of opts), entry 4·step + stage when
 *                          per_stage = 1 — the stage times t, t + h/2, t + h/2, t + h control! is called with — or entry `step` when
 *                          per_stage = 0 (zero-order hold over the step).
 *   RBD_CONTROL_PD       : τ_i = tau_i − kp_i (q_i − q_des_i) − kd_i v_i on every Revolute / Prismatic joint, evaluated on the STAGE state inside
 *                          the dynamics launch (other joint types: τ = tau).  kp, kd: [nv] gains, q_des: nq × B (NULL = 0), all DEVICE arrays of
 *                          the workspace's scalar type.
 * Device memory only (opts->memory = RBD_MEM_DEVICE); tree mechanisms without contact points. */
enum { RBD_CONTROL_CONSTANT = 0, RBD_CONTROL_TABLE = 1, RBD_CONTROL_PD = 2 };
typedef struct rbd_control {
  int32_t kind;       /* RBD_CONTROL_*                                   */
  int32_t per_stage;  /* TABLE: 1 = four entries per step, 0 = one       */
  const void* tau;    /* see above                                       */
  const void* q_des;  /* PD                                              */
  const void* kp;     /* PD                                              */
  const void* kd;     /* PD                                              */
} rbd_control_t;
int rbd_simulate_controlled(rbd_ws_t* ws, int32_t B, void* q, void* v, const rbd_control_t* control, const void* fext, double dt, int32_t nsteps,
                            const rbd_opts_t* opts);

/* ---- soft contact: contact_dynamics! (src/mechanism_algorithms.jl:680-723) and the entry points that include it --------------------
 * rbd_contact_dynamics: for every contact point inside a half-space the force of contact_dynamics! (src/contact.jl:79-93: Hunt–Crossley
 *   normal force, viscoelastic Coulomb friction) as a wrench on its body in the ROOT frame, and the state derivative of the friction
 *   model; for a pair that is not in contact the reference resets the state and zeroes the derivative — so `s` is IN/OUT.
 *   s, sdot: ns×B with ns = 3·n_contact_points·n_halfspaces; contactwrenches[6·n_bodies×B] = result.contactwrenches (nullable).
 * rbd_dynamics_contact: dynamics!(result, state, τ, wext) of a mechanism with contact points (:845-864): contact_dynamics!, then
 *   totalwrenches = wext + contactwrenches (:851-856), then forward dynamics with those.  totalwrenches nullable.
 * rbd_simulate_contact: simulate (src/simulate.jl:36-55) with the additional state integrated beside (q, v) by the same Runge–Kutta
 *   tableau (src/ode_integrators.jl:233-299).  Tree mechanisms.                                                                */
int rbd_model_contact_dims(const rbd_model_t* model, int32_t* n_contact_points, int32_t* n_halfspaces, int32_t* n_additional_states);
int rbd_contact_dynamics(rbd_ws_t* ws, int32_t B, const void* q, const void* v, void* s, void* contactwrenches, void* sdot, const rbd_opts_t* opts);
int rbd_dynamics_contact(rbd_ws_t* ws, int32_t B, const void* q, const void* v, void* s, const void* tau, const void* fext, void* vdot, void* qdot,
                         void* sdot, void* contactwrenches, void* totalwrenches, const rbd_opts_t* opts);
int rbd_simulate_contact(rbd_ws_t* ws, int32_t B, void* q, void* v, void* s, const void* tau, const void* fext, double dt, int32_t nsteps,
                         const rbd_opts_t* opts);

/* ---- kinematics by-products of the same forward-kinematics pass (every output nullable; opts->memory as for rbd_dynamics) ---------------
 * momentum_matrix: 6×nv column-major per state, root frame, (angular; linear) — momentum_matrix!(out, state)
 *   src/mechanism_algorithms.jl:313-327;  com: 3×B — center_of_mass(state) :28-50;  energy: 2×B = (kinetic_energy,
 *   gravitational_potential_energy) src/mechanism_state.jl:886-903 (needs v).
 * Larger batches (from 8193 states on an MI355X: a wavefront of 64 states for every other CU) take kernels compiled for the mechanism with one lane per state (family 11; Atlas,
 * 65 536 fp64 states: momentum matrix 56 us, energies 34, Jacobian 46, momentum 41 against 207 / 149 / 164 on the lane-per-body kernels).  A column of the momentum
 * matrix / Jacobian leaves as the lane's own 48 bytes: callers that can take the batch-innermost layout (RBD_LAYOUT_SOA) get whole 512-byte runs instead. */
int rbd_kinematics(rbd_ws_t* ws, int32_t B, const void* q, const void* v, void* momentum_matrix, void* com, void* energy,
                   const rbd_opts_t* opts);

/* momentum(state) and momentum_rate_bias(state), root frame — src/mechanism_state.jl:878-884, :975-987.  out: 12×B per state =
 * (momentum: angular 3, linear 3; momentum_rate_bias: torque 3, force 3).  d/dt momentum = momentum_matrix·v̇ + momentum_rate_bias.
*/
int rbd_momentum(rbd_ws_t* ws, int32_t B, const void* q, const void* v, void* out, const rbd_opts_t* opts);

/* geometric_jacobian!(jac, state, path(mechanism, base, target)) in the root frame — src/mechanism_algorithms.jl:80-99, :126-131.
 * base_body / target_body: body indices of the flat model, -1 = the root body.  jac: 6×nv column-major per state, (angular; linear);
 * columns off the path are written as zeros. */
int rbd_geometric_jacobian(rbd_ws_t* ws, int32_t B, const void* q, int32_t base_body, int32_t target_body, void* jac, const rbd_opts_t* opts);

/* ---- multi-GPU: the one exchange step (SURVEY.md §8 e) -------------------------------------------------------------------------
 * The batch shards by state — rank g of G owns states [g·B/G, (g+1)·B/G), model constants replicated, NO collective inside the
 * dynamics — and the shards of DynamicsResult.v̇ are brought together by one RCCL all-gather (or gather to one rank) over xGMI.
 * One process per GPU.  rbd_comm_unique_id on one rank, the 128 bytes handed to the others by the caller (as with ncclGetUniqueId),
 * then rbd_comm_create on every rank.  rbd_gather is asynchronous on `stream`; gathered holds world × count scalars in rank order
 * (AOS buffers: the (world·B/G) × n matrix; SOA: one n × B/G block per rank).  librccl is opened on first use.                  */
typedef struct rbd_comm rbd_comm_t;
int rbd_comm_unique_id(void* id128);
int rbd_comm_create(const void* id128, int32_t world, int32_t rank, int32_t device, rbd_comm_t** out);
int rbd_comm_destroy(rbd_comm_t* comm);
int rbd_comm_info(const rbd_comm_t* comm, int32_t* world, int32_t* rank);
int rbd_gather(rbd_comm_t* comm, int32_t dtype, const void* shard, void* gathered, int64_t count, int32_t root /* < 0: every rank */, void* stream);
/* ... with shards of different sizes (a batch that does not divide by the number of ranks): counts[world] scalars per rank, the shards back to back in rank
 * order in `gathered`; an empty shard is legal.  Every rank passes the same counts: ranks that disagree wait for each other forever, like mismatched ncclSend /
 * ncclRecv sizes — unless RBD_COMM_CHECK=1 is set (on every rank), in which case the ranks compare their counts first (one small all-gather and a wait for the
 * stream) and a rank that finds a disagreement returns RBD_ERR_DIMENSION_MISMATCH. */
int rbd_gatherv(rbd_comm_t* comm, int32_t dtype, const void* shard, void* gathered, const int64_t* counts, int32_t root /* < 0: every rank */, void* stream);
const char* rbd_comm_last_error(void);

/* ---- diagnostics ------------------------------------------------------------ */
const char* rbd_status_string(int status);
const char* rbd_last_hip_error(void);   /* thread-local text of the last HIP failure     */
/* average device time (ms) of the dominant kernel of the last hot-path call on this
 * workspace, measured with hipEvents recorded around the launch on ws's stream.
 * Timing is off by default; enable=1 brackets every launch with events.            */
int rbd_workspace_enable_timing(rbd_ws_t* ws, int32_t enable);
int rbd_workspace_last_kernel_ms(rbd_ws_t* ws, float* ms);
/* name of the articulated-body kernel (lane mapping) the last rbd_dynamics / rbd_simulate / rbd_mass_matrix_solve call launched */
const char* rbd_workspace_last_kernel(const rbd_ws_t* ws);
/* 100·round + revision of this header.  rbd_flat_model_t grew its four contact fields at 200; a caller built against an older header must
 * not call a newer library (the Python and Julia loaders compare this with the value they were written for).  400: rbd_workspace_set_loop_gains.
 * 500: rbd_mass_matrix_solve_packed, rbd_gatherv.  600: rbd_jit_check_walk_object; no size limit left on any entry point; program family 11; family 1 in fp64. */
#define RBD_HIP_H_VERSION 600
int rbd_version(void);
/* Run-time specialisation.  The one-lane-per-state kernels (mass_matrix! and mass_matrix! + Cholesky at large batches) exist in a second form
 * that is compiled for the mechanism at hand with hiprtc the first time a workspace needs it (the walk of the tree, joint types, offsets and
 * body constants become compile-time constants — what Julia's JIT does for the reference's `mass_matrix!`), cached on disk beside the library
 * (jit_cache/) or in $RBD_JIT_CACHE.  Without libhiprtc, or with RBD_JIT=0, the interpreting kernels run instead; results agree to rounding.
 * One program per family of kernels and scalar type — family 0: mass_matrix! (+ the sparse tile Cholesky and the emitter of M in fp32), 1: dynamics!
 * (fp32; since 600 fp64 for mechanisms with 3-dof joints / 6-dof joints below the world), 2: inverse_dynamics! / dynamics_bias! (fp32; fp64), 3: the whole loop-joint branch of dynamics! for small loop mechanisms (<= 4 bodies, nv <= 4,
 * nc <= 6: the four-bar linkage), 4 / 5: the one-wavefront-per-track kernels of dynamics! / inverse_dynamics! (batches beyond what the two-bodies-per-lane kernels hold at once), 6 / 7: the same with two
 * fp32 states per lane, 8: the two-bodies-per-lane kernels themselves (small batches — the bench workload — with the loops over the tree's levels unrolled against the
 * mechanism's level structure), 9 / 10: 4 / 6 with the four stages of a `simulate` step in one launch, 11 (since 600): the kinematics by-products one lane per state
 * (rbd_kinematics / rbd_geometric_jacobian / rbd_momentum at large batches: kin_spec, energy_spec, com_spec, jac_spec, mom_spec) — each compiled when a workspace first takes
 * that route.  Families 0-2 and 11 take every tree joint type; 4-7, 9, 10 and the fast form of 8 trees of 1-dof / fixed joints with 6-dof joints on the world.
 * rbd_jit_precompile compiles all of a model's programs of one scalar type (RBD_F32 / RBD_F64) into the cache ahead of time (no device needed);
 * rbd_jit_source returns the generated source of one (length without the terminator; buf may be NULL; -1: no such program for this mechanism). */
int rbd_jit_precompile(const rbd_model_t* model, int32_t dtype, char* log, int64_t log_capacity);
/* Compilation never blocks a hot-path call (since 400): a call that would use a program that is not in the cache starts its compilation on a background thread,
 * runs the interpreting kernel, and the first call after the compiler has finished switches over (RBD_JIT_ASYNC=0: the first call waits instead, as does a
 * call that names the compiled kernel, RBD_ALGO_ABA_COMPILED).  rbd_jit_precompile is the blocking form (all programs of the model in parallel; the log lists
 * the seconds each took).  rbd_jit_status is the non-blocking query for ONE program (family as above): 1 = ready, 0 = being compiled (started by this
 * call if nobody had), -1 = no such program / no hiprtc / compilation failed. */
int rbd_jit_status(const rbd_model_t* model, int32_t dtype, int32_t family);
/* Waits for every compilation this process has started in the background (no-op when there is none).  A host process calls it before it exits — the
 * Python mirror and the Julia shim register it with their `atexit`: the compiler's own teardown at process exit does not wait for a thread that is still
 * inside it (a process that exited seconds after its first large-batch call on a new mechanism ended in a segmentation fault). */
void rbd_jit_wait_idle(void);
int64_t rbd_jit_source(const rbd_model_t* model, int32_t dtype, int32_t family, char* buf, int64_t capacity);
/* The admission test the library applies to the code object of a one-wavefront-per-track ("walk") program before it loads it, on an object the caller hands in
 * (diagnostics and tests; host only, nothing is loaded).  Those kernels keep per-joint results in accumulation registers addressed BY NUMBER, counted down from a255
 * (csrc/rbd_walk.hpp WalkStash), behind the compiler's back.  Admitted: no scratch; the metadata's .agpr_count == 0 and <= 248 VGPRs — or, for the `simulate` program
 * whose source carries the marker "// rbd-walk-sim-loop stash-from=a<N>" (its register allocator may take accumulation registers of its own: a0 upwards), .agpr_count
 * <= N, so that the allocator's registers and the stash cannot alias whatever is live when; and a kernel descriptor of the known kind (AMDGPU HSA code object v5 / v6,
 * gfx950) that could be rewritten to cover all 256 accumulation registers.  RBD_OK: would be loaded; RBD_ERR_UNSUPPORTED: refused (log says why) — the kernels that
 * interpret the mechanism, or the four-launch form of `simulate`, serve then. */
int rbd_jit_check_walk_object(const char* source, const void* code, int64_t size, char* log, int64_t log_capacity);

#ifdef __cplusplus
}
#endif
#endif /* RBD_HIP_H */
