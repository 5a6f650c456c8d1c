/*
 * rbd_oracle_impl.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, one state at a time) of the reference's hot path,
 * RigidBodyDynamics.jl v2.5.0.  Included twice by rbd_oracle.c with
 * REAL/SFX = double/_f64 and float/_f32.  Each function cites the reference
 * file:line it follows (paths relative to the upstream repo).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this; librbd_hip.so never links or calls it.
 *
 * Parity pin: the reference stores NO golden vectors and Julia
 * cannot run here, so this restatement is pinned by the reference's own
 * closed-form and invariant tests, re-run against it in tests/test_oracle_*.py:
 * test/test_double_pendulum.jl:54-75 (M, C, G closed form, atol 1e-12 — the known answers),
 * test/test_mechanism_algorithms.jl:310-327 (J v = relative twist), 527-545 (A v = total momentum),
 * 564-572, 600-614, 616-652 (Coriolis skew symmetry), 654-675, 707-727 (momentum rate with external
 * wrenches on random floating trees), 729-753, 773-797 (power flow) — the invariants,
 * test/test_urdf.jl:85-101 (RPY golden matrices), test/test_simulate.jl:203-222 (four-bar loop closure).
 * Not pinned: outputs of the Julia implementation itself (no Julia toolchain in the image, so no
 * oracle/_ref and no generated golden vectors).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

typedef struct { REAL R[9]; REAL p[3]; } FN(xf_t); /* x_root = R x + p, R row-major */
#define XF FN(xf_t)

/* ---- 3-vector helpers --------------------------------------------------- */
static inline void FN(cross3)(const REAL* a, const REAL* b, REAL* o) {
  REAL x = a[1] * b[2] - a[2] * b[1];
  REAL y = a[2] * b[0] - a[0] * b[2];
  REAL z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void FN(matvec3)(const REAL* R, const REAL* x, REAL* o) {
  REAL a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  REAL b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  REAL c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static inline void FN(matmul3)(const REAL* A, const REAL* B, REAL* C) {
  REAL t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
/* H = A * B for rigid transforms: src/spatial/transform3d.jl:60-64 (4x4 product) */
static inline void FN(xf_mul)(const XF* A, const XF* B, XF* H) {
  XF t;
  FN(matmul3)(A->R, B->R, t.R);
  FN(matvec3)(A->R, B->p, t.p);
  for (int k = 0; k < 3; ++k) t.p[k] += A->p[k];
  *H = t;
}
/* inv: src/spatial/transform3d.jl:66-69 */
static inline void FN(xf_inv)(const XF* A, XF* H) {
  XF t;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t.R[3 * i + j] = A->R[3 * j + i];
  FN(matvec3)(t.R, A->p, t.p);
  for (int k = 0; k < 3; ++k) t.p[k] = -t.p[k];
  *H = t;
}
/* transform_spatial_motion: src/spatial/util.jl:104-108 */
static inline void FN(xm)(const XF* H, const REAL* w, const REAL* v, REAL* ow, REAL* ov) {
  REAL a[3], l[3], c[3];
  FN(matvec3)(H->R, w, a);
  FN(matvec3)(H->R, v, l);
  FN(cross3)(H->p, a, c);
  for (int k = 0; k < 3; ++k) { ow[k] = a[k]; ov[k] = l[k] + c[k]; }
}
/* wrench transform: src/spatial/spatialforce.jl:152-158  (Rτ + p×Rf ; Rf) */
static inline void FN(xforce)(const XF* H, const REAL* t, const REAL* f, REAL* ot, REAL* of) {
  REAL a[3], l[3], c[3];
  FN(matvec3)(H->R, t, a);
  FN(matvec3)(H->R, f, l);
  FN(cross3)(H->p, l, c);
  for (int k = 0; k < 3; ++k) { ot[k] = a[k] + c[k]; of[k] = l[k]; }
}
/* se3_commutator: src/spatial/util.jl:117-121 */
static inline void FN(se3_comm)(const REAL* x, const REAL* y, REAL* o) {
  REAL a[3], b[3], c[3];
  FN(cross3)(x, y, a);
  FN(cross3)(x, y + 3, b);
  FN(cross3)(x + 3, y, c);
  for (int k = 0; k < 3; ++k) { o[k] = a[k]; o[3 + k] = b[k] + c[k]; }
}

/* spatial inertia in some frame: J (sym 3x3 row-major), c = m*com, m */
typedef struct { REAL J[9]; REAL c[3]; REAL m; } FN(inertia_t);
#define INERTIA FN(inertia_t)

/* mul_inertia: src/spatial/util.jl:110-114 */
static inline void FN(mul_inertia)(const INERTIA* I, const REAL* T, REAL* o) {
  REAL a[3], b[3], d[3];
  FN(matvec3)(I->J, T, a);
  FN(cross3)(I->c, T + 3, b);
  FN(cross3)(I->c, T, d);
  for (int k = 0; k < 3; ++k) { o[k] = a[k] + b[k]; o[3 + k] = I->m * T[3 + k] - d[k]; }
}
/* transform(inertia, t): src/spatial/motion_force_interaction.jl:160-176 */
static inline void FN(inertia_transform)(const INERTIA* I, const XF* H, INERTIA* O) {
  const REAL* R = H->R; const REAL* p = H->p;
  REAL Rmc[3], mp[3], X[9], Y[9], RJ[9], RJRt[9], Rt[9];
  FN(matvec3)(R, I->c, Rmc);
  for (int k = 0; k < 3; ++k) mp[k] = I->m * p[k];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) X[3 * i + j] = Rmc[i] * p[j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Y[3 * i + j] = X[3 * i + j] + X[3 * j + i] + mp[i] * p[j];
  REAL trY = Y[0] + Y[4] + Y[8];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rt[3 * i + j] = R[3 * j + i];
  FN(matmul3)(R, I->J, RJ);
  FN(matmul3)(RJ, Rt, RJRt);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O->J[3 * i + j] = RJRt[3 * i + j] - Y[3 * i + j] + (i == j ? trY : (REAL)0);
  for (int k = 0; k < 3; ++k) O->c[k] = Rmc[k] + mp[k];
  O->m = I->m;
}
/* newton_euler: src/spatial/motion_force_interaction.jl:244-263 */
static inline void FN(newton_euler)(const INERTIA* I, const REAL* acc, const REAL* T, REAL* w) {
  REAL Ia[6], h[6], a[3], b[3], c[3];
  FN(mul_inertia)(I, acc, Ia);
  FN(mul_inertia)(I, T, h);
  FN(cross3)(T, h, a);         /* ω × angular momentum */
  FN(cross3)(T + 3, h + 3, b); /* v × linear momentum  */
  FN(cross3)(T, h + 3, c);     /* ω × linear momentum  */
  for (int k = 0; k < 3; ++k) { w[k] = Ia[k] + a[k] + b[k]; w[3 + k] = Ia[3 + k] + c[k]; }
}

static inline int FN(joint_nq)(int t) {
  switch (t) { case RBD_JOINT_FIXED: return 0; case RBD_JOINT_REVOLUTE: case RBD_JOINT_PRISMATIC: return 1;
    case RBD_JOINT_QUAT_FLOATING: return 7; case RBD_JOINT_PLANAR: return 3; case RBD_JOINT_QUAT_SPHERICAL: return 4;
    case RBD_JOINT_SINCOS_REVOLUTE: return 2; default: return -1; }
}
static inline int FN(joint_nv)(int t) {
  switch (t) { case RBD_JOINT_FIXED: return 0; case RBD_JOINT_REVOLUTE: case RBD_JOINT_PRISMATIC: return 1;
    case RBD_JOINT_QUAT_FLOATING: return 6; case RBD_JOINT_PLANAR: return 3; case RBD_JOINT_QUAT_SPHERICAL: return 3;
    case RBD_JOINT_SINCOS_REVOLUTE: return 1; default: return -1; }
}

/* AngleAxis -> RotMatrix given sin, cos: formula quoted verbatim in
 * src/joint_types/sin_cos_revolute.jl:69-96 (column-major constructor). */
static inline void FN(rot_axis_sc)(const REAL* ax, REAL s, REAL c, REAL* R) {
  REAL c1 = (REAL)1 - c;
  REAL c1x2 = c1 * ax[0] * ax[0], c1y2 = c1 * ax[1] * ax[1], c1z2 = c1 * ax[2] * ax[2];
  REAL c1xy = c1 * ax[0] * ax[1], c1xz = c1 * ax[0] * ax[2], c1yz = c1 * ax[1] * ax[2];
  REAL sx = s * ax[0], sy = s * ax[1], sz = s * ax[2];
  R[0] = (REAL)1 - c1y2 - c1z2; R[3] = c1xy + sz;             R[6] = c1xz - sy;
  R[1] = c1xy - sz;             R[4] = (REAL)1 - c1x2 - c1z2; R[7] = c1yz + sx;
  R[2] = c1xz + sy;             R[5] = c1yz - sx;             R[8] = (REAL)1 - c1x2 - c1y2;
}
/* QuatRotation(w,x,y,z, normalize=false) -> RotMatrix (Rotations.jl 1.x; unit-quaternion formula,
 * SURVEY.md App. C — third-party, pinned only for unit quaternions) */
static inline void FN(rot_quat)(const REAL* q, REAL* R) {
  REAL w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* joint_transform (frame_after -> frame_before) per type:
 * revolute.jl:59-62, prismatic.jl:69-73, quaternion_floating.jl:81-83,
 * quaternion_spherical.jl (rotation only), sin_cos_revolute.jl:69-96, fixed.jl, planar.jl:65-70 */
static void FN(joint_transform)(const rbd_flat_model_t* m, int i, const REAL* q, XF* T) {
  const REAL I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  REAL ax[3] = {(REAL)m->joint_axis[3 * i], (REAL)m->joint_axis[3 * i + 1], (REAL)m->joint_axis[3 * i + 2]};
  const REAL* qi = q + m->q_offset[i];
  memcpy(T->R, I3, sizeof I3);
  T->p[0] = T->p[1] = T->p[2] = 0;
  switch (m->joint_type[i]) {
    case RBD_JOINT_REVOLUTE: FN(rot_axis_sc)(ax, SIN(qi[0]), COS(qi[0]), T->R); break;
    case RBD_JOINT_SINCOS_REVOLUTE: FN(rot_axis_sc)(ax, qi[0], qi[1], T->R); break;
    case RBD_JOINT_PRISMATIC: for (int k = 0; k < 3; ++k) T->p[k] = qi[0] * ax[k]; break;
    case RBD_JOINT_QUAT_FLOATING: FN(rot_quat)(qi, T->R); T->p[0] = qi[4]; T->p[1] = qi[5]; T->p[2] = qi[6]; break;
    case RBD_JOINT_QUAT_SPHERICAL: FN(rot_quat)(qi, T->R); break;
    case RBD_JOINT_PLANAR: {
      /* planar.jl:65-70: rot = AngleAxis(q[3], rot_axis), trans = x_axis*q[1] + y_axis*q[2], rot_axis = x × y */
      REAL ay[3] = {(REAL)m->joint_axis2[3 * i], (REAL)m->joint_axis2[3 * i + 1], (REAL)m->joint_axis2[3 * i + 2]};
      REAL az[3];
      FN(cross3)(ax, ay, az);
      FN(rot_axis_sc)(az, SIN(qi[2]), COS(qi[2]), T->R);
      for (int k = 0; k < 3; ++k) T->p[k] = ax[k] * qi[0] + ay[k] * qi[1];
    } break;
    default: break;
  }
}

/* local motion subspace column k of joint i in frame_after (angular; linear):
 * revolute.jl:83-89, prismatic.jl:93-99, quaternion_floating.jl:85-91,
 * quaternion_spherical.jl, planar.jl:118-126 */
static void FN(local_subspace)(const rbd_flat_model_t* m, int i, int k, const REAL* q, REAL* S) {
  REAL ax[3] = {(REAL)m->joint_axis[3 * i], (REAL)m->joint_axis[3 * i + 1], (REAL)m->joint_axis[3 * i + 2]};
  for (int j = 0; j < 6; ++j) S[j] = 0;
  switch (m->joint_type[i]) {
    case RBD_JOINT_REVOLUTE: case RBD_JOINT_SINCOS_REVOLUTE: S[0] = ax[0]; S[1] = ax[1]; S[2] = ax[2]; break;
    case RBD_JOINT_PRISMATIC: S[3] = ax[0]; S[4] = ax[1]; S[5] = ax[2]; break;
    case RBD_JOINT_QUAT_FLOATING: S[k] = 1; break;
    case RBD_JOINT_QUAT_SPHERICAL: S[k] = 1; break;
    case RBD_JOINT_PLANAR: {
      /* planar.jl motion_subspace: angular = [0 0 rot_axis], linear = [x_axis y_axis 0], constant in frame_after */
      REAL ay[3] = {(REAL)m->joint_axis2[3 * i], (REAL)m->joint_axis2[3 * i + 1], (REAL)m->joint_axis2[3 * i + 2]};
      REAL az[3];
      FN(cross3)(ax, ay, az);
      (void)q;
      if (k == 2) { S[0] = az[0]; S[1] = az[1]; S[2] = az[2]; }
      else { const REAL* a = (k == 0) ? ax : ay; S[3] = a[0]; S[4] = a[1]; S[5] = a[2]; }
    } break;
    default: break;
  }
}

typedef struct {
  int nb, nq, nv;
  XF* H;        /* [nb] transforms_to_root                          */
  REAL* S;      /* [nv*6] world-frame motion subspace columns       */
  REAL* T;      /* [nb*6] twists wrt world                           */
  REAL* A;      /* [nb*6] accelerations                              */
  REAL* W;      /* [nb*6] wrenches                                   */
  INERTIA* I;   /* [nb] world-frame inertias                         */
} FN(cache_t);
#define CACHE FN(cache_t)

/* per-thread scratch, grown on demand and reused across evaluations (keeps malloc out of the timed baseline) */
static __thread CACHE FN(tls_cache);
static __thread int FN(tls_nb) = 0, FN(tls_nv) = 0;
static int FN(cache_alloc)(const rbd_flat_model_t* m, CACHE* c) {
  CACHE* t = &FN(tls_cache);
  if (FN(tls_nb) < m->n_bodies || FN(tls_nv) < m->nv) {
    if (FN(tls_nb) > 0) { free(t->H); free(t->S); free(t->T); free(t->A); free(t->W); free(t->I); }
    int nb = m->n_bodies, nv = m->nv > 0 ? m->nv : 1;
    t->H = (XF*)malloc(sizeof(XF) * (size_t)nb);
    t->S = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nv);
    t->T = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nb);
    t->A = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nb);
    t->W = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nb);
    t->I = (INERTIA*)malloc(sizeof(INERTIA) * (size_t)nb);
    if (!(t->H && t->S && t->T && t->A && t->W && t->I)) { FN(tls_nb) = 0; return -1; }
    FN(tls_nb) = nb; FN(tls_nv) = nv;
  }
  *c = *t;
  c->nb = m->n_bodies; c->nq = m->nq; c->nv = m->nv;
  return 0;
}
static void FN(cache_free)(CACHE* c) { (void)c; }

static void FN(load_xpred)(const rbd_flat_model_t* m, int i, XF* X) {
  for (int k = 0; k < 9; ++k) X->R[k] = (REAL)m->pred_rot[9 * i + k];
  for (int k = 0; k < 3; ++k) X->p[k] = (REAL)m->pred_trans[3 * i + k];
}
static void FN(load_inertia)(const rbd_flat_model_t* m, int i, INERTIA* I) {
  for (int k = 0; k < 9; ++k) I->J[k] = (REAL)m->inertia_moment[9 * i + k];
  for (int k = 0; k < 3; ++k) I->c[k] = (REAL)m->inertia_cross[3 * i + k];
  I->m = (REAL)m->inertia_mass[i];
}

/* update_transforms!: src/mechanism_state.jl:687-700  H_b = H_p * joint_to_predecessor * joint_transform(q) */
static void FN(update_transforms)(const rbd_flat_model_t* m, const REAL* q, CACHE* c) {
  for (int i = 0; i < c->nb; ++i) {
    XF Xp, Tj, X;
    FN(load_xpred)(m, i, &Xp);
    FN(joint_transform)(m, i, q, &Tj);
    int p = m->parent[i];
    if (p < 0) {
      FN(xf_mul)(&Xp, &Tj, &c->H[i]); /* root transform is identity */
    } else {
      FN(xf_mul)(&c->H[p], &Xp, &X);
      FN(xf_mul)(&X, &Tj, &c->H[i]);
    }
  }
}
/* update_motion_subspaces!: src/mechanism_state.jl:744-763 */
static void FN(update_motion_subspaces)(const rbd_flat_model_t* m, const REAL* q, CACHE* c) {
  for (int i = 0; i < c->nb; ++i) {
    int nvi = FN(joint_nv)(m->joint_type[i]);
    for (int k = 0; k < nvi; ++k) {
      REAL Sl[6];
      FN(local_subspace)(m, i, k, q, Sl);
      REAL* S = c->S + 6 * (m->v_offset[i] + k);
      FN(xm)(&c->H[i], Sl, Sl + 3, S, S + 3);
    }
  }
}
/* update_twists_wrt_world!: src/mechanism_state.jl:769-780 (joint twist = S_local * v: revolute.jl:64-68 etc.) */
static void FN(update_twists)(const rbd_flat_model_t* m, const REAL* v, CACHE* c) {
  for (int i = 0; i < c->nb; ++i) {
    int p = m->parent[i];
    REAL* T = c->T + 6 * i;
    for (int j = 0; j < 6; ++j) T[j] = (p < 0) ? (REAL)0 : c->T[6 * p + j];
    int nvi = FN(joint_nv)(m->joint_type[i]);
    REAL tj[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < nvi; ++k) {
      const REAL* S = c->S + 6 * (m->v_offset[i] + k);
      REAL vk = v[m->v_offset[i] + k];
      for (int j = 0; j < 6; ++j) tj[j] += S[j] * vk;
    }
    for (int j = 0; j < 6; ++j) T[j] += tj[j];
  }
}
/* update_spatial_inertias!: src/mechanism_state.jl:836-846 */
static void FN(update_inertias)(const rbd_flat_model_t* m, CACHE* c) {
  for (int i = 0; i < c->nb; ++i) {
    INERTIA Ib;
    FN(load_inertia)(m, i, &Ib);
    FN(inertia_transform)(&Ib, &c->H[i], &c->I[i]);
  }
}

/* accelerations for dynamics_bias!: bias_accelerations! src/mechanism_algorithms.jl:377-385 on top of
 * update_bias_accelerations_wrt_world! src/mechanism_state.jl:814-830 (all joint biases are zero).
 * with_vd != 0: spatial_accelerations! src/mechanism_algorithms.jl:387-417 instead.                 */
static void FN(accelerations)(const rbd_flat_model_t* m, const REAL* vd, CACHE* c) {
  REAL g[6] = {0, 0, 0, -(REAL)m->gravity[0], -(REAL)m->gravity[1], -(REAL)m->gravity[2]};
  for (int i = 0; i < c->nb; ++i) {
    int p = m->parent[i];
    const REAL* Tb = c->T + 6 * i;
    REAL Tp[6], Ap[6], x[6], cr[6];
    for (int j = 0; j < 6; ++j) { Tp[j] = (p < 0) ? (REAL)0 : c->T[6 * p + j]; Ap[j] = (p < 0) ? g[j] : c->A[6 * p + j]; }
    /* (-T_b) × T_p   (== T_b × (T_b - T_p), the bias form) */
    for (int j = 0; j < 6; ++j) x[j] = -Tb[j];
    FN(se3_comm)(x, Tp, cr);
    REAL* A = c->A + 6 * i;
    for (int j = 0; j < 6; ++j) A[j] = Ap[j] + cr[j];
    if (vd) {
      int nvi = FN(joint_nv)(m->joint_type[i]);
      for (int k = 0; k < nvi; ++k) {
        const REAL* S = c->S + 6 * (m->v_offset[i] + k);
        REAL a = vd[m->v_offset[i] + k];
        for (int j = 0; j < 6; ++j) A[j] += S[j] * a;
      }
    }
  }
}

/* newton_euler! :428-439 then joint_wrenches_and_torques! :442-459 */
static void FN(wrenches_and_torques)(const rbd_flat_model_t* m, const REAL* fext, CACHE* c, REAL* tau) {
  for (int i = 0; i < c->nb; ++i) {
    REAL* w = c->W + 6 * i;
    FN(newton_euler)(&c->I[i], c->A + 6 * i, c->T + 6 * i, w);
    if (fext) for (int j = 0; j < 6; ++j) w[j] -= fext[6 * i + j];
  }
  for (int i = c->nb - 1; i >= 0; --i) {
    int p = m->parent[i];
    const REAL* w = c->W + 6 * i;
    if (p >= 0) for (int j = 0; j < 6; ++j) c->W[6 * p + j] += w[j];
    int nvi = FN(joint_nv)(m->joint_type[i]);
    for (int k = 0; k < nvi; ++k) {
      const REAL* S = c->S + 6 * (m->v_offset[i] + k);
      REAL d = 0;
      for (int j = 0; j < 6; ++j) d += S[j] * w[j];
      tau[m->v_offset[i] + k] = d;
    }
  }
}

/* inverse_dynamics!: src/mechanism_algorithms.jl:542-553; vd == NULL gives dynamics_bias! :484-494 */
int FN(rbdo_inverse_dynamics)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, const REAL* vd,
                              const REAL* fext, REAL* tau) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(update_inertias)(m, &c);
  FN(accelerations)(m, vd, &c);
  FN(wrenches_and_torques)(m, fext, &c, tau);
  FN(cache_free)(&c);
  return RBD_OK;
}
int FN(rbdo_dynamics_bias)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, const REAL* fext, REAL* cvec) {
  return FN(rbdo_inverse_dynamics)(m, q, v, NULL, fext, cvec);
}
/* inverse_dynamics!(τ, jointwrenchesout, accelerations, state, v̇, externalwrenches) with its per-body outputs (:542-553): accelerations[b]
 * as spatial_accelerations! leaves them (root acceleration −gravity included, :405-415; vd == NULL: bias_accelerations! :377-385) and
 * jointwrenchesout[b] after joint_wrenches_and_torques! (:442-459); both in the root frame, 6 per body.                               */
int FN(rbdo_inverse_dynamics_bodies)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, const REAL* vd, const REAL* fext, REAL* tau,
                                     REAL* jointwrenches, REAL* accelerations) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(update_inertias)(m, &c);
  FN(accelerations)(m, vd, &c);
  FN(wrenches_and_torques)(m, fext, &c, tau);
  if (accelerations) memcpy(accelerations, c.A, sizeof(REAL) * 6 * c.nb);
  if (jointwrenches) memcpy(jointwrenches, c.W, sizeof(REAL) * 6 * c.nb);
  FN(cache_free)(&c);
  return RBD_OK;
}

static int FN(supports)(const rbd_flat_model_t* m, int jointj, int bodyi) {
  /* support_set_masks: src/mechanism_state.jl:95-98 — joint j supports body i iff body(j) is i or an ancestor */
  for (int b = bodyi; b >= 0; b = m->parent[b]) if (b == jointj) return 1;
  return 0;
}

/* mass_matrix!: src/mechanism_algorithms.jl:248-272 with update_crb_inertias! src/mechanism_state.jl:852-868.
 * M is nv×nv column-major; lower triangle written, strict upper zero-filled. */
static void FN(mass_matrix_cached)(const rbd_flat_model_t* m, CACHE* c, REAL* M) {
  int nv = c->nv;
  INERTIA* Ic = (INERTIA*)malloc(sizeof(INERTIA) * (size_t)c->nb);
  int* body_of_v = (int*)malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1));
  for (int i = 0; i < c->nb; ++i) Ic[i] = c->I[i];
  for (int i = c->nb - 1; i >= 0; --i) {
    int p = m->parent[i];
    if (p >= 0) {
      for (int k = 0; k < 9; ++k) Ic[p].J[k] += Ic[i].J[k];
      for (int k = 0; k < 3; ++k) Ic[p].c[k] += Ic[i].c[k];
      Ic[p].m += Ic[i].m;
    }
  }
  for (int i = 0; i < c->nb; ++i) {
    int nvi = FN(joint_nv)(m->joint_type[i]);
    for (int k = 0; k < nvi; ++k) body_of_v[m->v_offset[i] + k] = i;
  }
  for (int i = 0; i < nv * nv; ++i) M[i] = 0;
  for (int i = 0; i < nv; ++i) {
    int bi = body_of_v[i];
    REAL F[6];
    FN(mul_inertia)(&Ic[bi], c->S + 6 * i, F); /* Ici * Si: motion_force_interaction.jl:223-233 */
    for (int j = 0; j <= i; ++j) {
      if (FN(supports)(m, body_of_v[j], bi)) {
        const REAL* Sj = c->S + 6 * j;
        REAL d = 0;
        for (int k = 0; k < 6; ++k) d += F[k] * Sj[k];
        M[(size_t)j * nv + i] = d;
      }
    }
  }
  free(Ic); free(body_of_v);
}
int FN(rbdo_mass_matrix)(const rbd_flat_model_t* m, const REAL* q, REAL* M) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_inertias)(m, &c);
  FN(mass_matrix_cached)(m, &c, M);
  FN(cache_free)(&c);
  return RBD_OK;
}

/* LAPACK potrf!('L') restated (unblocked, dpotf2-style): src/mechanism_algorithms.jl:764 */
static int FN(chol_lower)(REAL* L, int n) {
  for (int j = 0; j < n; ++j) {
    REAL d = L[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)k * n + j] * L[(size_t)k * n + j];
    if (!(d > 0)) return RBD_ERR_NOT_POSITIVE_DEFINITE;
    d = SQRT(d);
    L[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      REAL s = L[(size_t)j * n + i];
      for (int k = 0; k < j; ++k) s -= L[(size_t)k * n + i] * L[(size_t)k * n + j];
      L[(size_t)j * n + i] = s / d;
    }
  }
  return RBD_OK;
}
/* x <- L^-1 x (trsv 'L','N') */
static void FN(fwd_subst)(const REAL* L, int n, REAL* x) {
  for (int i = 0; i < n; ++i) {
    REAL s = x[i];
    for (int k = 0; k < i; ++k) s -= L[(size_t)k * n + i] * x[k];
    x[i] = s / L[(size_t)i * n + i];
  }
}
/* x <- L^-T x */
static void FN(bwd_subst)(const REAL* L, int n, REAL* x) {
  for (int i = n - 1; i >= 0; --i) {
    REAL s = x[i];
    for (int k = i + 1; k < n; ++k) s -= L[(size_t)i * n + k] * x[k];
    x[i] = s / L[(size_t)i * n + i];
  }
}

/* configuration_derivative!: src/mechanism_state.jl:905-910; quaternion_floating.jl:126-136,
 * spatial/util.jl:127-134; sin_cos_revolute.jl q̇ = (c v, -s v); planar.jl:128-140; default q̇ = v */
void FN(rbdo_configuration_derivative)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, REAL* qd) {
  for (int i = 0; i < m->n_bodies; ++i) {
    const REAL* qi = q + m->q_offset[i];
    const REAL* vi = v + m->v_offset[i];
    REAL* o = qd + m->q_offset[i];
    switch (m->joint_type[i]) {
      case RBD_JOINT_REVOLUTE: case RBD_JOINT_PRISMATIC: o[0] = vi[0]; break;
      case RBD_JOINT_SINCOS_REVOLUTE: o[0] = qi[1] * vi[0]; o[1] = -qi[0] * vi[0]; break;
      case RBD_JOINT_QUAT_FLOATING: case RBD_JOINT_QUAT_SPHERICAL: {
        REAL w = qi[0], x = qi[1], y = qi[2], z = qi[3];
        o[0] = (-x * vi[0] - y * vi[1] - z * vi[2]) / 2;
        o[1] = (w * vi[0] - z * vi[1] + y * vi[2]) / 2;
        o[2] = (z * vi[0] + w * vi[1] - x * vi[2]) / 2;
        o[3] = (-y * vi[0] + x * vi[1] + w * vi[2]) / 2;
        if (m->joint_type[i] == RBD_JOINT_QUAT_FLOATING) {
          REAL R[9];
          FN(rot_quat)(qi, R);
          FN(matvec3)(R, vi + 3, o + 4);
        }
      } break;
      case RBD_JOINT_PLANAR: {
        /* planar.jl: ẋ,ẏ = R2(θ) * (vx, vy); θ̇ = ω */
        REAL s = SIN(qi[2]), c = COS(qi[2]);
        o[0] = c * vi[0] - s * vi[1];
        o[1] = s * vi[0] + c * vi[1];
        o[2] = vi[2];
      } break;
      default: break;
    }
  }
}

/* dynamics! (tree mechanisms): src/mechanism_algorithms.jl:845-864 — the reference's route:
 * q̇, dynamics_bias!, mass_matrix!, dynamics_solve! (potrf/potrs branch :764,:819).
 * Mout (nv×nv), cout (nv) nullable. */
int FN(rbdo_dynamics)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, const REAL* tau, const REAL* fext,
                      REAL* vdot, REAL* qdot, REAL* Mout, REAL* cout) {
  int nv = m->nv;
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  REAL* M = (REAL*)malloc(sizeof(REAL) * (size_t)(nv * nv + 1));
  REAL* cb = (REAL*)malloc(sizeof(REAL) * (size_t)(nv + 1));
  if (qdot) FN(rbdo_configuration_derivative)(m, q, v, qdot);
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(update_inertias)(m, &c);
  FN(accelerations)(m, NULL, &c);
  FN(wrenches_and_torques)(m, fext, &c, cb);
  FN(mass_matrix_cached)(m, &c, M);
  if (Mout) memcpy(Mout, M, sizeof(REAL) * (size_t)(nv * nv));
  if (cout) memcpy(cout, cb, sizeof(REAL) * (size_t)nv);
  int st = FN(chol_lower)(M, nv);
  if (st == RBD_OK) {
    for (int i = 0; i < nv; ++i) vdot[i] = (tau ? tau[i] : (REAL)0) - cb[i];
    FN(fwd_subst)(M, nv, vdot);
    FN(bwd_subst)(M, nv, vdot);
  }
  free(M); free(cb);
  FN(cache_free)(&c);
  return st;
}

/* x = M(q)^-1 rhs */
int FN(rbdo_mass_matrix_solve)(const rbd_flat_model_t* m, const REAL* q, const REAL* rhs, REAL* x) {
  int nv = m->nv;
  REAL* M = (REAL*)malloc(sizeof(REAL) * (size_t)(nv * nv + 1));
  int st = FN(rbdo_mass_matrix)(m, q, M);
  if (st == RBD_OK) st = FN(chol_lower)(M, nv);
  if (st == RBD_OK) {
    for (int i = 0; i < nv; ++i) x[i] = rhs[i];
    FN(fwd_subst)(M, nv, x);
    FN(bwd_subst)(M, nv, x);
  }
  free(M);
  return st;
}

/* ---- loop (non-tree) joints ------------------------------------------------------------------------------
 * constraint_jacobian! src/mechanism_algorithms.jl:574-598, constraint_bias! :630-673 (Baumgarte stabilization through
 * pd(..., SE3PDMethod{:Linearized}) src/pdcontrol.jl:109-122), dynamics_solve! loop branch :768-816.            */
static int FN(loop_nc)(int t) { return 6 - FN(joint_nv)(t); }

/* constraint_wrench_subspace in frame_after(loop joint): revolute.jl:91-98, prismatic.jl:101-108, fixed.jl (identity) */
static int FN(constraint_basis)(const rbd_loop_joint_t* lj, REAL* Tl /* 6 x nc, column c at Tl + 6 c */) {
  int nc = FN(loop_nc)(lj->joint_type);
  const double* R = lj->rotation_from_z_aligned; /* row-major */
  for (int k = 0; k < 6 * nc; ++k) Tl[k] = 0;
  switch (lj->joint_type) {
    case RBD_JOINT_REVOLUTE: case RBD_JOINT_SINCOS_REVOLUTE:
      for (int r = 0; r < 3; ++r) { Tl[6 * 0 + r] = (REAL)R[3 * r + 0]; Tl[6 * 1 + r] = (REAL)R[3 * r + 1];
        Tl[6 * 2 + 3 + r] = (REAL)R[3 * r + 0]; Tl[6 * 3 + 3 + r] = (REAL)R[3 * r + 1]; Tl[6 * 4 + 3 + r] = (REAL)R[3 * r + 2]; }
      return nc;
    case RBD_JOINT_PRISMATIC:
      for (int r = 0; r < 3; ++r) { Tl[6 * 0 + r] = (REAL)R[3 * r + 0]; Tl[6 * 1 + r] = (REAL)R[3 * r + 1]; Tl[6 * 2 + r] = (REAL)R[3 * r + 2];
        Tl[6 * 3 + 3 + r] = (REAL)R[3 * r + 0]; Tl[6 * 4 + 3 + r] = (REAL)R[3 * r + 1]; }
      return nc;
    case RBD_JOINT_FIXED:
      for (int c = 0; c < 6; ++c) Tl[6 * c + c] = 1;
      return nc;
    case RBD_JOINT_QUAT_SPHERICAL: /* quaternion_spherical.jl:50-55 */
      for (int c = 0; c < 3; ++c) Tl[6 * c + 3 + c] = 1;
      return nc;
    case RBD_JOINT_PLANAR: /* planar.jl:96-101: (0; rot_axis), (x_axis; 0), (y_axis; 0); the columns of R are (x_axis, y_axis, x × y) */
      for (int r = 0; r < 3; ++r) { Tl[6 * 0 + 3 + r] = (REAL)R[3 * r + 2]; Tl[6 * 1 + r] = (REAL)R[3 * r + 0]; Tl[6 * 2 + r] = (REAL)R[3 * r + 1]; }
      return nc;
    case RBD_JOINT_QUAT_FLOATING: return 0;
    default: return -1;
  }
}

/* symmetric eigen-decomposition by cyclic Jacobi: A (n x n, row-major, destroyed: diagonal = eigenvalues), V = eigenvectors (columns) */
static void FN(jacobi_eig)(REAL* A, REAL* V, int n) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    REAL off = 0, diag = 0;
    for (int i = 0; i < n; ++i) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
    if (off <= (REAL)1e-60 + diag * (REAL)1e-34) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        REAL apq = A[p * n + q];
        if (apq == 0) continue;
        REAL theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
        REAL t = (theta >= 0 ? (REAL)1 : (REAL)-1) / ((theta >= 0 ? theta : -theta) + SQRT(theta * theta + 1));
        REAL c = 1 / SQRT(t * t + 1), sn = t * c;
        for (int k = 0; k < n; ++k) { REAL akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - sn * akq; A[k * n + q] = sn * akp + c * akq; }
        for (int k = 0; k < n; ++k) { REAL apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - sn * aqk; A[q * n + k] = sn * apk + c * aqk; }
        for (int k = 0; k < n; ++k) { REAL vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - sn * vkq; V[k * n + q] = sn * vkp + c * vkq; }
      }
  }
}

/* dynamics! for mechanisms with loop joints (the full :845-864 path).  Outputs: vdot[nv], lambda[nc] (min-norm, like
 * LAPACK gelsy! with rcond 1e-10 — third-party, restated as the truncated pseudo-inverse of the PSD Schur matrix),
 * Kout [nc x nv] column-major, kout [nc] (nullable).  stabilize != 0: Baumgarte with the loop joints' gains.          */
int FN(rbdo_dynamics_loops)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, const REAL* tau, const REAL* fext, int stabilize,
                            REAL* vdot, REAL* qdot, REAL* lambda, REAL* Kout, REAL* kout, REAL* Mout, REAL* cout) {
  int nv = m->nv, nb = m->n_bodies, nc = 0;
  for (int l = 0; l < m->n_loops; ++l) nc += FN(loop_nc)(m->loops[l].joint_type);
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  REAL* M = (REAL*)malloc(sizeof(REAL) * (size_t)(nv * nv + 1));
  REAL* cb = (REAL*)malloc(sizeof(REAL) * (size_t)(nv + 1));
  REAL* K = (REAL*)calloc((size_t)(nc * nv + 1), sizeof(REAL)); /* row-major here: K[ci*nv + vi] */
  REAL* kk = (REAL*)calloc((size_t)(nc + 1), sizeof(REAL));
  REAL* Y = (REAL*)malloc(sizeof(REAL) * (size_t)(nc * nv + 1));
  REAL* A = (REAL*)malloc(sizeof(REAL) * (size_t)(nc * nc + 1));
  REAL* V = (REAL*)malloc(sizeof(REAL) * (size_t)(nc * nc + 1));
  REAL* z = (REAL*)malloc(sizeof(REAL) * (size_t)(nv + 1));
  REAL* bvec = (REAL*)malloc(sizeof(REAL) * (size_t)(nc + 1));
  REAL* Abias = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nb);
  if (qdot) FN(rbdo_configuration_derivative)(m, q, v, qdot);
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(update_inertias)(m, &c);
  FN(accelerations)(m, NULL, &c);
  /* bias accelerations wrt world WITHOUT gravity (state.bias_accelerations_wrt_world): c.A carries -g at the root */
  for (int i = 0; i < nb; ++i) for (int j = 0; j < 6; ++j) Abias[6 * i + j] = c.A[6 * i + j] - (j >= 3 ? -(REAL)m->gravity[j - 3] : (REAL)0);
  FN(wrenches_and_torques)(m, fext, &c, cb);
  FN(mass_matrix_cached)(m, &c, M);
  if (Mout) memcpy(Mout, M, sizeof(REAL) * (size_t)(nv * nv));
  if (cout) memcpy(cout, cb, sizeof(REAL) * (size_t)nv);
  int row0 = 0, st = RBD_OK;
  const REAL I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int l = 0; l < m->n_loops && st == RBD_OK; ++l) {
    const rbd_loop_joint_t* lj = &m->loops[l];
    REAL Tl[36];
    int ncl = FN(constraint_basis)(lj, Tl);
    if (ncl < 0) { st = RBD_ERR_UNSUPPORTED; break; }
    XF Hp, Hs, Xp, Xs, Fb, Fa;
    memcpy(Hp.R, I3, sizeof I3); Hp.p[0] = Hp.p[1] = Hp.p[2] = 0; Hs = Hp;
    if (lj->predecessor >= 0) Hp = c.H[lj->predecessor];
    if (lj->successor >= 0) Hs = c.H[lj->successor];
    for (int k = 0; k < 9; ++k) { Xp.R[k] = (REAL)lj->pred_rot[k]; Xs.R[k] = (REAL)lj->succ_rot[k]; }
    for (int k = 0; k < 3; ++k) { Xp.p[k] = (REAL)lj->pred_trans[k]; Xs.p[k] = (REAL)lj->succ_trans[k]; }
    FN(xf_mul)(&Hp, &Xp, &Fb); /* before_to_root: mechanism_state.jl:707 */
    FN(xf_mul)(&Hs, &Xs, &Fa); /* after_to_root:  :708, also the frame of the constraint wrench basis :791 */
    REAL Tw[36];
    for (int ci = 0; ci < ncl; ++ci) FN(xforce)(&Fa, Tl + 6 * ci, Tl + 6 * ci + 3, Tw + 6 * ci, Tw + 6 * ci + 3);
    /* path predecessor -> successor (src/graphs/tree_path.jl:41-63): 'up' edges from the predecessor side (sign -1),
     * 'down' edges on the successor side (sign +1) */
    int a = lj->predecessor, b = lj->successor;
    while (a != b) {
      int body, sign;
      if (a > b) { body = a; sign = -1; a = m->parent[a]; } else { body = b; sign = 1; b = m->parent[b]; }
      int nvj = FN(joint_nv)(m->joint_type[body]);
      for (int col = 0; col < nvj; ++col) {
        int vi = m->v_offset[body] + col;
        const REAL* S = c.S + 6 * vi;
        for (int ci = 0; ci < ncl; ++ci) {
          REAL d = 0;
          for (int j = 0; j < 6; ++j) d += Tw[6 * ci + j] * S[j];
          K[(row0 + ci) * nv + vi] = sign < 0 ? -d : d;
        }
      }
    }
    /* constraint_bias! */
    REAL Tp[6], Ts[6], Ap[6], As[6], cr[6], ba[6];
    for (int j = 0; j < 6; ++j) {
      Tp[j] = lj->predecessor >= 0 ? c.T[6 * lj->predecessor + j] : (REAL)0; Ts[j] = lj->successor >= 0 ? c.T[6 * lj->successor + j] : (REAL)0;
      Ap[j] = lj->predecessor >= 0 ? Abias[6 * lj->predecessor + j] : (REAL)0; As[j] = lj->successor >= 0 ? Abias[6 * lj->successor + j] : (REAL)0;
    }
    FN(se3_comm)(Ts, Tp, cr);
    for (int j = 0; j < 6; ++j) ba[j] = cr[j] + (As[j] - Ap[j]);
    if (stabilize) {
      XF Fbi, Tn, Fai;
      FN(xf_inv)(&Fb, &Fbi);
      FN(xf_mul)(&Fbi, &Fa, &Tn); /* joint transform of the loop joint: frame_after -> frame_before */
      FN(xf_inv)(&Fa, &Fai);
      REAL jt[6], jl[6], stab[6], sw[6], Rtp[3];
      for (int j = 0; j < 6; ++j) jt[j] = Ts[j] - Tp[j];
      FN(xm)(&Fai, jt, jt + 3, jl, jl + 3); /* joint twist in frame_after */
      REAL psi[3] = {(Tn.R[7] - Tn.R[5]) / 2, (Tn.R[2] - Tn.R[6]) / 2, (Tn.R[3] - Tn.R[1]) / 2}; /* spatial/util.jl:178-183 */
      for (int i = 0; i < 3; ++i) Rtp[i] = Tn.R[i] * Tn.p[0] + Tn.R[3 + i] * Tn.p[1] + Tn.R[6 + i] * Tn.p[2];
      for (int i = 0; i < 3; ++i) {
        stab[i] = -(REAL)lj->gains[0] * psi[i] - (REAL)lj->gains[1] * jl[i];
        stab[3 + i] = -(REAL)lj->gains[2] * Rtp[i] - (REAL)lj->gains[3] * jl[3 + i];
      }
      FN(xm)(&Fa, stab, stab + 3, sw, sw + 3);
      for (int j = 0; j < 6; ++j) ba[j] -= sw[j];
    }
    for (int ci = 0; ci < ncl; ++ci) { REAL d = 0; for (int j = 0; j < 6; ++j) d += Tw[6 * ci + j] * ba[j]; kk[row0 + ci] = d; }
    row0 += ncl;
  }
  if (st == RBD_OK) st = FN(chol_lower)(M, nv); /* L */
  if (st == RBD_OK) {
    for (int i = 0; i < nv; ++i) z[i] = (tau ? tau[i] : (REAL)0) - cb[i];
    REAL* rhs = (REAL*)malloc(sizeof(REAL) * (size_t)(nv + 1));
    memcpy(rhs, z, sizeof(REAL) * (size_t)nv);
    if (nc > 0) {
      FN(fwd_subst)(M, nv, z);                                   /* z = L^-1 (tau - c) */
      for (int ci = 0; ci < nc; ++ci) {                          /* Y = K L^-T  <=>  L Y' = K' */
        memcpy(Y + ci * nv, K + ci * nv, sizeof(REAL) * (size_t)nv);
        FN(fwd_subst)(M, nv, Y + ci * nv);
      }
      for (int i = 0; i < nc; ++i) {
        for (int j = 0; j < nc; ++j) { REAL s2 = 0; for (int k = 0; k < nv; ++k) s2 += Y[i * nv + k] * Y[j * nv + k]; A[i * nc + j] = s2; }
        REAL s2 = kk[i]; for (int k = 0; k < nv; ++k) s2 += Y[i * nv + k] * z[k]; bvec[i] = s2;
      }
      /* lambda = min-norm least-squares solution of A lambda = b (A PSD, possibly singular): gelsy!(A, b, 1e-10) */
      FN(jacobi_eig)(A, V, nc);
      REAL emax = 0;
      for (int i = 0; i < nc; ++i) if (A[i * nc + i] > emax) emax = A[i * nc + i];
      for (int i = 0; i < nc; ++i) lambda[i] = 0;
      for (int e = 0; e < nc; ++e) {
        REAL ev = A[e * nc + e];
        if (ev > (REAL)1e-10 * emax) {
          REAL d = 0; for (int i = 0; i < nc; ++i) d += V[i * nc + e] * bvec[i];
          d /= ev;
          for (int i = 0; i < nc; ++i) lambda[i] += V[i * nc + e] * d;
        }
      }
      for (int vi = 0; vi < nv; ++vi) { REAL s2 = 0; for (int ci = 0; ci < nc; ++ci) s2 += K[ci * nv + vi] * lambda[ci]; rhs[vi] -= s2; }
    }
    FN(fwd_subst)(M, nv, rhs); FN(bwd_subst)(M, nv, rhs);
    memcpy(vdot, rhs, sizeof(REAL) * (size_t)nv);
    free(rhs);
    if (Kout) for (int ci = 0; ci < nc; ++ci) for (int vi = 0; vi < nv; ++vi) Kout[vi * nc + ci] = K[ci * nv + vi];
    if (kout) memcpy(kout, kk, sizeof(REAL) * (size_t)nc);
  }
  free(M); free(cb); free(K); free(kk); free(Y); free(A); free(V); free(z); free(bvec); free(Abias);
  FN(cache_free)(&c);
  return st;
}

/* ---- independent cross-check: world-frame articulated-body algorithm (NOT in the reference;
 * SURVEY.md App. A item 12).  Must reproduce rbdo_dynamics' v̇. ----------------------------- */
static void FN(sym6_from_inertia)(const INERTIA* I, REAL* A /*6x6 row-major*/) {
  /* [ J   ĉ ; ĉᵀ  m 1 ]  with ĉ = hat(c):  I*(ω,v) = (Jω + c×v, m v − c×ω) */
  for (int k = 0; k < 36; ++k) A[k] = 0;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[6 * i + j] = I->J[3 * i + j];
  const REAL* c = I->c;
  REAL hat[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[6 * i + 3 + j] = hat[3 * i + j]; A[6 * (3 + i) + j] = hat[3 * j + i]; }
  for (int i = 0; i < 3; ++i) A[6 * (3 + i) + 3 + i] = I->m;
}
int FN(rbdo_aba)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, const REAL* tau, const REAL* fext, REAL* vdot) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  int nb = c.nb;
  REAL* IA = (REAL*)malloc(sizeof(REAL) * 36 * (size_t)nb);
  REAL* pA = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nb);
  REAL* cb = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nb);
  REAL* U = (REAL*)malloc(sizeof(REAL) * 36 * (size_t)nb);   /* 6 x n_i, col k at U + 36 b + 6 k */
  REAL* Dl = (REAL*)malloc(sizeof(REAL) * 36 * (size_t)nb);  /* chol factor of D, n_i x n_i col-major */
  REAL* u = (REAL*)malloc(sizeof(REAL) * 6 * (size_t)nb);
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(update_inertias)(m, &c);
  int st = RBD_OK;
  for (int b = 0; b < nb; ++b) {
    int p = m->parent[b];
    const REAL* Tb = c.T + 6 * b;
    REAL vJ[6], h[6], a3[3], b3[3], c3[3];
    for (int j = 0; j < 6; ++j) vJ[j] = Tb[j] - (p < 0 ? (REAL)0 : c.T[6 * p + j]);
    FN(se3_comm)(Tb, vJ, cb + 6 * b);
    FN(sym6_from_inertia)(&c.I[b], IA + 36 * b);
    FN(mul_inertia)(&c.I[b], Tb, h);
    FN(cross3)(Tb, h, a3); FN(cross3)(Tb + 3, h + 3, b3); FN(cross3)(Tb, h + 3, c3);
    for (int k = 0; k < 3; ++k) { pA[6 * b + k] = a3[k] + b3[k]; pA[6 * b + 3 + k] = c3[k]; }
    if (fext) for (int j = 0; j < 6; ++j) pA[6 * b + j] -= fext[6 * b + j];
  }
  for (int b = nb - 1; b >= 0 && st == RBD_OK; --b) {
    int p = m->parent[b];
    int n = FN(joint_nv)(m->joint_type[b]);
    const REAL* A = IA + 36 * b;
    REAL* Ub = U + 36 * b; REAL* D = Dl + 36 * b; REAL* ub = u + 6 * b;
    REAL Ia[36], pa[6];
    for (int k = 0; k < 36; ++k) Ia[k] = A[k];
    for (int j = 0; j < 6; ++j) pa[j] = pA[6 * b + j];
    if (n > 0) {
      for (int k = 0; k < n; ++k) {
        const REAL* S = c.S + 6 * (m->v_offset[b] + k);
        for (int i = 0; i < 6; ++i) { REAL s = 0; for (int j = 0; j < 6; ++j) s += A[6 * i + j] * S[j]; Ub[6 * k + i] = s; }
      }
      for (int k = 0; k < n; ++k) {
        const REAL* Sk = c.S + 6 * (m->v_offset[b] + k);
        for (int l = 0; l < n; ++l) { REAL s = 0; for (int j = 0; j < 6; ++j) s += Sk[j] * Ub[6 * l + j]; D[l * n + k] = s; }
        REAL s = 0; for (int j = 0; j < 6; ++j) s += Sk[j] * pA[6 * b + j];
        ub[k] = (tau ? tau[m->v_offset[b] + k] : (REAL)0) - s;
      }
      st = FN(chol_lower)(D, n);
      if (st != RBD_OK) break;
      /* Ia = IA - U D^-1 Uᵀ ;  pa = pA + Ia cb + U D^-1 u */
      REAL Y[36]; /* Y = L^-1 Uᵀ  (n x 6), column j = L^-1 (row j of U) */
      for (int j = 0; j < 6; ++j) {
        REAL x[6];
        for (int k = 0; k < n; ++k) x[k] = Ub[6 * k + j];
        FN(fwd_subst)(D, n, x);
        for (int k = 0; k < n; ++k) Y[6 * k + j] = x[k];
      }
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { REAL s = 0; for (int k = 0; k < n; ++k) s += Y[6 * k + i] * Y[6 * k + j]; Ia[6 * i + j] -= s; }
      REAL x[6];
      for (int k = 0; k < n; ++k) x[k] = ub[k];
      FN(fwd_subst)(D, n, x); FN(bwd_subst)(D, n, x);
      for (int i = 0; i < 6; ++i) { REAL s = 0; for (int j = 0; j < 6; ++j) s += Ia[6 * i + j] * cb[6 * b + j]; for (int k = 0; k < n; ++k) s += Ub[6 * k + i] * x[k]; pa[i] += s; }
    } else {
      for (int i = 0; i < 6; ++i) { REAL s = 0; for (int j = 0; j < 6; ++j) s += Ia[6 * i + j] * cb[6 * b + j]; pa[i] += s; }
    }
    if (p >= 0) {
      for (int k = 0; k < 36; ++k) IA[36 * p + k] += Ia[k];
      for (int j = 0; j < 6; ++j) pA[6 * p + j] += pa[j];
    }
  }
  if (st == RBD_OK) {
    REAL g[6] = {0, 0, 0, -(REAL)m->gravity[0], -(REAL)m->gravity[1], -(REAL)m->gravity[2]};
    for (int b = 0; b < nb; ++b) {
      int p = m->parent[b];
      int n = FN(joint_nv)(m->joint_type[b]);
      REAL ap[6], x[6];
      for (int j = 0; j < 6; ++j) ap[j] = (p < 0 ? g[j] : c.A[6 * p + j]) + cb[6 * b + j];
      for (int k = 0; k < n; ++k) { REAL s = 0; for (int j = 0; j < 6; ++j) s += U[36 * b + 6 * k + j] * ap[j]; x[k] = u[6 * b + k] - s; }
      if (n > 0) { FN(fwd_subst)(Dl + 36 * b, n, x); FN(bwd_subst)(Dl + 36 * b, n, x); }
      for (int k = 0; k < n; ++k) {
        vdot[m->v_offset[b] + k] = x[k];
        const REAL* S = c.S + 6 * (m->v_offset[b] + k);
        for (int j = 0; j < 6; ++j) ap[j] += S[j] * x[k];
      }
      for (int j = 0; j < 6; ++j) c.A[6 * b + j] = ap[j];
    }
  }
  free(IA); free(pA); free(cb); free(U); free(Dl); free(u);
  FN(cache_free)(&c);
  return st;
}

/* kinetic_energy / gravitational_potential_energy: src/mechanism_state.jl:886-903 */
int FN(rbdo_energy)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, REAL* ke, REAL* pe) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(update_inertias)(m, &c);
  REAL K = 0, P = 0;
  for (int i = 0; i < c.nb; ++i) {
    REAL h[6];
    FN(mul_inertia)(&c.I[i], c.T + 6 * i, h);
    REAL d = 0;
    for (int j = 0; j < 6; ++j) d += h[j] * c.T[6 * i + j];
    K += d / 2;
    /* -m g · com_world ; m*com_world = world-frame cross_part */
    for (int k = 0; k < 3; ++k) P -= (REAL)m->gravity[k] * c.I[i].c[k];
  }
  *ke = K; *pe = P;
  FN(cache_free)(&c);
  return RBD_OK;
}

/* momentum_matrix! in the root frame: src/mechanism_algorithms.jl:313-327 — column i = crb_inertia(body(i)) * S_i.
 * A: 6 x nv column-major (angular; linear).  Also momentum = sum_b I_b T_b (src/mechanism_state.jl:878-880) for the check
 * of test/test_mechanism_algorithms.jl:527-545, and center_of_mass: src/mechanism_algorithms.jl:28-50.               */
int FN(rbdo_momentum_matrix)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, REAL* A, REAL* hsum, REAL* com) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_inertias)(m, &c);
  INERTIA* Ic = (INERTIA*)malloc(sizeof(INERTIA) * (size_t)c.nb);
  for (int i = 0; i < c.nb; ++i) Ic[i] = c.I[i];
  for (int i = c.nb - 1; i >= 0; --i) {
    int p = m->parent[i];
    if (p >= 0) {
      for (int k = 0; k < 9; ++k) Ic[p].J[k] += Ic[i].J[k];
      for (int k = 0; k < 3; ++k) Ic[p].c[k] += Ic[i].c[k];
      Ic[p].m += Ic[i].m;
    }
  }
  for (int i = 0; i < c.nb; ++i) {
    int nvi = FN(joint_nv)(m->joint_type[i]);
    for (int k = 0; k < nvi; ++k) FN(mul_inertia)(&Ic[i], c.S + 6 * (m->v_offset[i] + k), A + 6 * (m->v_offset[i] + k));
  }
  if (hsum && v) {
    FN(update_twists)(m, v, &c);
    for (int j = 0; j < 6; ++j) hsum[j] = 0;
    for (int i = 0; i < c.nb; ++i) { REAL h[6]; FN(mul_inertia)(&c.I[i], c.T + 6 * i, h); for (int j = 0; j < 6; ++j) hsum[j] += h[j]; }
  }
  if (com) {
    REAL mass = 0, s3[3] = {0, 0, 0};
    for (int i = 0; i < c.nb; ++i)
      if (c.I[i].m > 0) { mass += c.I[i].m; for (int k = 0; k < 3; ++k) s3[k] += c.I[i].c[k]; } /* m * com_world = world cross_part */
    for (int k = 0; k < 3; ++k) com[k] = s3[k] / mass;
  }
  free(Ic);
  FN(cache_free)(&c);
  return RBD_OK;
}

/* geometric_jacobian!(jac, state, path) in the root frame: src/mechanism_algorithms.jl:80-99 with path(mechanism, base, body)
 * (src/graphs/tree_path.jl:41-63): columns of the joints from `base` up to the lowest common ancestor get -S (direction up),
 * those from the ancestor down to `body` +S, every other column 0.  base / body: body indices, -1 = the root body.
 * J: 6 x nv column-major (angular; linear).  trel (optional, needs v): relative_twist(state, body, base) in the root frame,
 * for the check of test/test_mechanism_algorithms.jl:310-327 (J v == relative twist).                                      */
int FN(rbdo_geometric_jacobian)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, int base, int body, REAL* J, REAL* trel) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  for (int i = 0; i < 6 * m->nv; ++i) J[i] = 0;
  int a = base, b = body;
  while (a != b) {
    const int up = a > b; /* parents come first, so the deeper side is the larger index (tree_path.jl:47-58) */
    const int x = up ? a : b;
    const int nvi = FN(joint_nv)(m->joint_type[x]);
    for (int k = 0; k < nvi; ++k)
      for (int r = 0; r < 6; ++r) J[6 * (m->v_offset[x] + k) + r] = (up ? -1 : 1) * c.S[6 * (m->v_offset[x] + k) + r];
    if (up) a = m->parent[a]; else b = m->parent[b];
  }
  if (trel && v) {
    FN(update_twists)(m, v, &c);
    for (int r = 0; r < 6; ++r) trel[r] = (body >= 0 ? c.T[6 * body + r] : 0) - (base >= 0 ? c.T[6 * base + r] : 0);
  }
  FN(cache_free)(&c);
  return RBD_OK;
}

/* momentum(state) and momentum_rate_bias(state), root frame: src/mechanism_state.jl:878-884, :975-987 —
 * h = sum_b I_b T_b ;  hdot_bias = sum_b newton_euler(I_b, bias_acceleration_b, T_b) with the bias accelerations wrt world WITHOUT
 * gravity (update_bias_accelerations_wrt_world!).  The rate of change of momentum is then momentum_matrix * vdot + hdot_bias
 * (test/test_mechanism_algorithms.jl:719).                                                                                      */
int FN(rbdo_momentum)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, REAL* h, REAL* hdot_bias) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(update_inertias)(m, &c);
  FN(accelerations)(m, NULL, &c);
  for (int j = 0; j < 6; ++j) { h[j] = 0; hdot_bias[j] = 0; }
  for (int i = 0; i < c.nb; ++i) {
    REAL hb[6], ab[6], w[6];
    FN(mul_inertia)(&c.I[i], c.T + 6 * i, hb);
    for (int j = 0; j < 6; ++j) ab[j] = c.A[6 * i + j] - (j >= 3 ? -(REAL)m->gravity[j - 3] : (REAL)0);  /* c.A carries -g from the root */
    FN(newton_euler)(&c.I[i], ab, c.T + 6 * i, w);
    for (int j = 0; j < 6; ++j) { h[j] += hb[j]; hdot_bias[j] += w[j]; }
  }
  FN(cache_free)(&c);
  return RBD_OK;
}

/* spatial_accelerations!(result, state) with result.v̇ = vd (src/mechanism_algorithms.jl:387-417), then
 * relative_acceleration(result, body, root_body) (src/dynamics_result.jl accessor): the root's fictitious -gravity is subtracted,
 * so acc[6*i .. 6*i+5] is the true spatial acceleration of body i in the root frame.  Also returns transforms_to_root
 * (R row-major 9, p 3) and twists_wrt_world (6) per body — what the maximal-coordinates test needs to build the equivalent state
 * (test/test_mechanism_modification.jl:274-318).                                                                          */
int FN(rbdo_body_kinematics)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, const REAL* vd, REAL* H12, REAL* twist, REAL* acc) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);
  FN(accelerations)(m, vd, &c);
  for (int i = 0; i < c.nb; ++i) {
    if (H12) { memcpy(H12 + 12 * i, c.H[i].R, sizeof(REAL) * 9); memcpy(H12 + 12 * i + 9, c.H[i].p, sizeof(REAL) * 3); }
    for (int j = 0; j < 6; ++j) {
      if (twist) twist[6 * i + j] = c.T[6 * i + j];
      if (acc) acc[6 * i + j] = c.A[6 * i + j] - (j >= 3 ? -(REAL)m->gravity[j - 3] : (REAL)0);
    }
  }
  FN(cache_free)(&c);
  return RBD_OK;
}

/* transforms_to_root of every moving body, for FK checks: out[b*12 ..] = R (9, row-major), p (3) */
int FN(rbdo_transforms)(const rbd_flat_model_t* m, const REAL* q, REAL* out) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);
  for (int i = 0; i < c.nb; ++i) { memcpy(out + 12 * i, c.H[i].R, sizeof(REAL) * 9); memcpy(out + 12 * i + 9, c.H[i].p, sizeof(REAL) * 3); }
  FN(cache_free)(&c);
  return RBD_OK;
}

/* ---- soft contact --------------------------------------------------------------------------------------------------------------
 * contact_dynamics!(result, state) (src/mechanism_algorithms.jl:680-723) with the reference's default point model
 * SoftContactModel{HuntCrossleyModel, ViscoelasticCoulombModel} (src/contact.jl:79-93, :98-119, :122-178) and HalfSpace3D (:202-228).
 * s (in/out: reset! for pairs not in contact), sdot, contactwrenches [6 n_bodies] in the root frame (torque about the root origin; force). */
int FN(rbdo_contact_dynamics)(const rbd_flat_model_t* m, const REAL* q, const REAL* v, REAL* s, REAL* sdot, REAL* cw) {
  CACHE c;
  if (FN(cache_alloc)(m, &c)) return RBD_ERR_OUT_OF_MEMORY;
  FN(update_transforms)(m, q, &c);          /* :681 */
  FN(update_motion_subspaces)(m, q, &c);
  FN(update_twists)(m, v, &c);              /* :682 */
  for (int i = 0; i < 6 * c.nb; ++i) cw[i] = 0;
  const int nh = m->n_halfspaces;
  for (int ip = 0; ip < m->n_contact_points; ++ip) {
    const rbd_contact_point_t* cp = &m->contact_points[ip];
    const XF* H = &c.H[cp->body];
    const REAL* tw = &c.T[6 * cp->body];
    REAL loc[3] = {(REAL)cp->location[0], (REAL)cp->location[1], (REAL)cp->location[2]}, pt[3], vel[3], t3[3];
    FN(matvec3)(H->R, loc, pt);                                   /* point = body_to_root * location(c)   :697 */
    for (int j = 0; j < 3; ++j) pt[j] += H->p[j];
    FN(cross3)(tw, pt, t3);                                       /* point_velocity(twist, point)          :698 */
    for (int j = 0; j < 3; ++j) vel[j] = t3[j] + tw[3 + j];
    for (int h = 0; h < nh; ++h) {
      const rbd_halfspace_t* hs = &m->halfspaces[h];
      REAL n[3] = {(REAL)hs->outward_normal[0], (REAL)hs->outward_normal[1], (REAL)hs->outward_normal[2]};
      const REAL nn = SQRT(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);  /* HalfSpace3D normalizes its normal (contact.jl:208-211) */
      for (int j = 0; j < 3; ++j) n[j] /= nn;
      REAL* x = s + 3 * (ip * nh + h);
      REAL* xd = sdot + 3 * (ip * nh + h);
      REAL sep = 0;                                               /* separation (contact.jl:224) */
      for (int j = 0; j < 3; ++j) sep += (pt[j] - (REAL)hs->point[j]) * n[j];
      if (sep <= 0) {                                             /* point_inside (:225) */
        const REAL z = -sep;                                      /* penetration */
        REAL zd = 0;
        for (int j = 0; j < 3; ++j) zd -= vel[j] * n[j];          /* ż = −dot(velocity, normal)   contact.jl:84 */
        const REAL zn = (REAL)pow((double)z, cp->hc_n);
        REAL fn = (REAL)cp->hc_lambda * zn * zd + (REAL)cp->hc_k * zn;   /* normal_force :115-118 */
        if (!(fn > 0)) fn = 0;                                    /* max(., 0) :85 */
        REAL vt[3], fs[3], n2 = 0;
        for (int j = 0; j < 3; ++j) vt[j] = vel[j] + zd * n[j];   /* tangential_velocity :88 */
        for (int j = 0; j < 3; ++j) { fs[j] = -(REAL)cp->k * x[j] - (REAL)cp->b * vt[j]; n2 += fs[j] * fs[j]; }   /* fstick :159 */
        const REAL m2 = ((REAL)cp->mu * fn) * ((REAL)cp->mu * fn);
        if (n2 > m2) { const REAL sc = SQRT(m2 / n2); for (int j = 0; j < 3; ++j) fs[j] *= sc; }                    /* :163-167 */
        for (int j = 0; j < 3; ++j) xd[j] = (-(REAL)cp->k * x[j] - fs[j]) / (REAL)cp->b;                             /* :171-178 */
        REAL f[3], tq[3];
        for (int j = 0; j < 3; ++j) f[j] = fn * n[j] + fs[j];     /* :92 */
        FN(cross3)(pt, f, tq);                                    /* Wrench(point, force) */
        for (int j = 0; j < 3; ++j) { cw[6 * cp->body + j] += tq[j]; cw[6 * cp->body + 3 + j] += f[j]; }
      } else {
        for (int j = 0; j < 3; ++j) { x[j] = 0; xd[j] = 0; }      /* reset!, zero!  :714-715 */
      }
    }
  }
  FN(cache_free)(&c);
  return RBD_OK;
}

/* ---- batch drivers (AOS: one state per column, x[b*n + k]); OpenMP over states.  Used by the parity
 * tests and by bench.py's cpu_baseline leg. what: 0 dynamics (reference route), 1 inverse dynamics,
 * 2 dynamics_bias, 3 mass matrix, 4 ABA cross-check ------------------------------------------- */
int FN(rbdo_batch)(const rbd_flat_model_t* m, int what, int B, int nthreads, const REAL* q, const REAL* v,
                   const REAL* x /*tau or vdot*/, const REAL* fext, REAL* out, REAL* qdot) {
  int nq = m->nq, nv = m->nv, nb = m->n_bodies;
  int status = RBD_OK;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int b = 0; b < B; ++b) {
    const REAL* qb = q + (size_t)b * nq;
    const REAL* vb = v ? v + (size_t)b * nv : NULL;
    const REAL* xb = x ? x + (size_t)b * nv : NULL;
    const REAL* fb = fext ? fext + (size_t)b * 6 * nb : NULL;
    int st = RBD_OK;
    switch (what) {
      case 0: st = FN(rbdo_dynamics)(m, qb, vb, xb, fb, out + (size_t)b * nv, qdot ? qdot + (size_t)b * nq : NULL, NULL, NULL); break;
      case 1: st = FN(rbdo_inverse_dynamics)(m, qb, vb, xb, fb, out + (size_t)b * nv); break;
      case 2: st = FN(rbdo_dynamics_bias)(m, qb, vb, fb, out + (size_t)b * nv); break;
      case 3: st = FN(rbdo_mass_matrix)(m, qb, out + (size_t)b * nv * nv); break;
      case 4: st = FN(rbdo_aba)(m, qb, vb, xb, fb, out + (size_t)b * nv); break;
      default: st = RBD_ERR_INVALID_ARGUMENT;
    }
    if (st != RBD_OK) {
#pragma omp critical
      status = st;
    }
  }
  return status;
}

#undef XF
#undef INERTIA
#undef CACHE
#undef FN
#undef CAT
#undef CAT_
