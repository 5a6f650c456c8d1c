"""Host-side model: the part of RigidBodyDynamics.jl's `Mechanism` API that is needed to
*describe* a mechanism and flatten it once into the index tables / per-joint constants
the HIP kernels consume (`rbd_flat_model_t`, include/rbd_hip.h).

This mirrors (names, argument meaning) the reference's model layer but is NOT a
re-implementation of it — it only carries what the flattener reads:

  Mechanism, RigidBody, Joint, attach!           src/mechanism.jl:10-34, src/rigid_body.jl:12-29,
                                                 src/joint.jl:43-67, src/mechanism_modification.jl:21-46
  canonicalize_frame_definitions!                src/mechanism.jl:250-266
  remove_fixed_tree_joints!                      src/mechanism_modification.jl:260-317
  MechanismState index tables                    src/mechanism_state.jl:85-118

All host-side geometry is float64 numpy; nothing here runs per state.
"""
from __future__ import annotations

import ctypes
import itertools
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# joint type tags — keep in sync with include/rbd_hip.h
JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_QUAT_FLOATING, JOINT_PLANAR, JOINT_QUAT_SPHERICAL, JOINT_SINCOS_REVOLUTE = range(7)

DEFAULT_GRAVITATIONAL_ACCELERATION = np.array([0.0, 0.0, -9.81])  # src/mechanism.jl:1

_frame_ids = itertools.count()


class CartesianFrame3D:
    """src/spatial/frame.jl:20-40 — identity-compared frame tag."""

    __slots__ = ("name", "id")

    def __init__(self, name: str = "anonymous"):
        self.name = name
        self.id = next(_frame_ids)

    def __repr__(self):
        return f"CartesianFrame3D({self.name!r})"


class Transform3D:
    """Rigid transform `from` -> `to`: x_to = R x_from + p  (src/spatial/transform3d.jl:7-15)."""

    __slots__ = ("frm", "to", "R", "p")

    def __init__(self, frm: CartesianFrame3D, to: CartesianFrame3D, R=None, p=None):
        self.frm, self.to = frm, to
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64).reshape(3, 3)
        self.p = np.zeros(3) if p is None else np.asarray(p, dtype=np.float64).reshape(3)

    def __mul__(self, other: "Transform3D") -> "Transform3D":  # transform3d.jl:60-64
        if self.frm is not other.to:
            raise ValueError(f"frame mismatch: {self.frm} vs {other.to}")
        return Transform3D(other.frm, self.to, self.R @ other.R, self.R @ other.p + self.p)

    def inv(self) -> "Transform3D":  # transform3d.jl:66-69
        Rt = self.R.T
        return Transform3D(self.to, self.frm, Rt, -(Rt @ self.p))


def rot_z_y_x(yaw, pitch, roll):
    """RotZYX(yaw, pitch, roll) = Rz·Ry·Rx, as used for URDF rpy (src/urdf/parse.jl:46-51)."""
    cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    Ry = np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    Rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return Rz @ Ry @ Rx


def rotation_between(u, v):
    """Rotations.rotation_between(u, v): shortest rotation taking u to v (Rotations.jl 1.x, third-party;
    restated from the public formula — SURVEY.md App. C; affects only the loop-joint constraint basis)."""
    u = np.asarray(u, float)
    v = np.asarray(v, float)
    nu, nv = np.linalg.norm(u), np.linalg.norm(v)
    w = nu * nv + u @ v
    xyz = np.cross(u, v)
    if w < 1e-12 * nu * nv and np.linalg.norm(xyz) < 1e-12 * nu * nv:  # antiparallel: any perpendicular axis
        a = np.array([1.0, 0, 0]) if abs(u[0]) < 0.9 * nu else np.array([0, 1.0, 0])
        xyz = np.cross(u, a)
        w = 0.0
    q = np.array([w, *xyz])
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


class SpatialInertia:
    """src/spatial/motion_force_interaction.jl:28-37: moment about the frame origin, cross_part = m·com, mass."""

    __slots__ = ("frame", "moment", "cross_part", "mass")

    def __init__(self, frame: CartesianFrame3D, moment=None, cross_part=None, mass=0.0, com=None, moment_about_com=None):
        self.frame = frame
        self.mass = float(mass)
        if com is not None:  # keyword constructor :70-86 (moment given about `frame` origin unless moment_about_com)
            com = np.asarray(com, float)
            cross_part = self.mass * com
            if moment_about_com is not None:
                c = com
                hat2 = np.outer(c, c) - (c @ c) * np.eye(3)  # hat(c)^2
                moment = np.asarray(moment_about_com, float) - self.mass * hat2
        self.moment = np.zeros((3, 3)) if moment is None else np.asarray(moment, float).reshape(3, 3).copy()
        self.cross_part = np.zeros(3) if cross_part is None else np.asarray(cross_part, float).reshape(3).copy()

    def transform(self, t: Transform3D) -> "SpatialInertia":  # :160-176
        if t.frm is not self.frame:
            raise ValueError("frame mismatch in SpatialInertia.transform")
        R, p, J, mc, m = t.R, t.p, self.moment, self.cross_part, self.mass
        Rmc = R @ mc
        mp = m * p
        X = np.outer(Rmc, p)
        Y = X + X.T + np.outer(mp, p)
        Jnew = R @ J @ R.T - Y + np.trace(Y) * np.eye(3)
        return SpatialInertia(t.to, Jnew, Rmc + mp, m)

    def __add__(self, o: "SpatialInertia") -> "SpatialInertia":  # :143-153
        if o.frame is not self.frame:
            raise ValueError("frame mismatch in SpatialInertia +")
        return SpatialInertia(self.frame, self.moment + o.moment, self.cross_part + o.cross_part, self.mass + o.mass)


# ---- joint types (src/joint_types/*.jl) -----------------------------------------------------------
@dataclass
class JointType:
    tag: int = JOINT_FIXED
    nq: int = 0
    nv: int = 0
    axis: np.ndarray = field(default_factory=lambda: np.zeros(3))
    axis2: np.ndarray = field(default_factory=lambda: np.zeros(3))
    rotation_from_z_aligned: np.ndarray = field(default_factory=lambda: np.eye(3))


def Fixed() -> JointType:
    return JointType(JOINT_FIXED, 0, 0)


def _normalized(a):
    a = np.asarray(a, float).reshape(3)
    return a / np.linalg.norm(a)


def Revolute(axis) -> JointType:  # revolute.jl:10-18
    a = _normalized(axis)
    return JointType(JOINT_REVOLUTE, 1, 1, a, np.zeros(3), rotation_between([0.0, 0, 1], a))


def Prismatic(axis) -> JointType:  # prismatic.jl:10-18
    a = _normalized(axis)
    return JointType(JOINT_PRISMATIC, 1, 1, a, np.zeros(3), rotation_between([0.0, 0, 1], a))


def SinCosRevolute(axis) -> JointType:  # sin_cos_revolute.jl
    a = _normalized(axis)
    return JointType(JOINT_SINCOS_REVOLUTE, 2, 1, a, np.zeros(3), rotation_between([0.0, 0, 1], a))


def QuaternionFloating() -> JointType:  # quaternion_floating.jl:19-27
    return JointType(JOINT_QUAT_FLOATING, 7, 6)


def QuaternionSpherical() -> JointType:  # quaternion_spherical.jl
    return JointType(JOINT_QUAT_SPHERICAL, 4, 3)


def Planar(x_axis, y_axis) -> JointType:  # planar.jl:22-35
    x, y = _normalized(x_axis), _normalized(y_axis)
    assert abs(x @ y) < 1e-12
    # as a loop joint its constraint wrench basis needs both axes: carried in `rotation_from_z_aligned` as the columns (x, y, x × y)
    return JointType(JOINT_PLANAR, 3, 3, x, y, np.column_stack([x, y, np.cross(x, y)]))


class Joint:
    """src/joint.jl:43-67."""

    def __init__(self, name: str, joint_type: JointType):
        self.name = name
        self.joint_type = joint_type
        self.frame_before = CartesianFrame3D("before_" + name)
        self.frame_after = CartesianFrame3D("after_" + name)
        self.joint_to_predecessor: Optional[Transform3D] = None
        self.joint_to_successor: Optional[Transform3D] = None
        # Baumgarte gains when used as a loop joint (default_constraint_stabilization_gains: mechanism_algorithms.jl:610-612)
        self.stabilization_gains = (100.0, 20.0, 100.0, 20.0)

    def __repr__(self):
        return f"Joint({self.name!r})"


class RigidBody:
    """src/rigid_body.jl:12-29. `inertia is None` only for the root body."""

    def __init__(self, name_or_inertia, inertia: Optional[SpatialInertia] = None):
        if isinstance(name_or_inertia, SpatialInertia):  # RigidBody(inertia): name = frame name (rigid_body.jl:38)
            inertia = name_or_inertia
            name = inertia.frame.name
        else:
            name = name_or_inertia
        self.name = name
        self.inertia = inertia
        self.default_frame = inertia.frame if inertia is not None else CartesianFrame3D(name)
        # frame definitions: transforms from body-fixed frames to the default frame (rigid_body.jl:24)
        self.frame_definitions: Dict[int, Transform3D] = {self.default_frame.id: Transform3D(self.default_frame, self.default_frame)}
        self.contact_points: List["ContactPoint"] = []  # rigid_body.jl:25

    def __repr__(self):
        return f"RigidBody({self.name!r})"

    def frame_definition(self, frame: CartesianFrame3D) -> Transform3D:
        try:
            return self.frame_definitions[frame.id]
        except KeyError:
            raise ValueError(f"{frame} not found among body fixed frame definitions for {self}") from None

    def add_frame(self, t: Transform3D) -> Transform3D:  # add_frame!: rigid_body.jl:133-143
        if t.to is not self.default_frame:
            t = self.frame_definition(t.to) * t
        self.frame_definitions[t.frm.id] = t
        return t

    def fixed_transform(self, frm: CartesianFrame3D, to: CartesianFrame3D) -> Transform3D:
        return self.frame_definition(to).inv() * self.frame_definition(frm)

    def change_default_frame(self, new_default: CartesianFrame3D):  # rigid_body.jl:151-166
        if new_default is self.default_frame:
            return
        old_to_new = self.frame_definition(new_default).inv()
        self.frame_definitions = {k: old_to_new * tf for k, tf in self.frame_definitions.items()}
        if self.inertia is not None:
            self.inertia = self.inertia.transform(old_to_new)
        for point in self.contact_points:  # rigid_body.jl:161-163
            point.location = old_to_new.R @ point.location + old_to_new.p
            point.frame = new_default
        self.default_frame = new_default


class Mechanism:
    """src/mechanism.jl:10-34 — a rooted tree of bodies/joints plus optional non-tree (loop) joints."""

    def __init__(self, root_body: RigidBody, gravity=DEFAULT_GRAVITATIONAL_ACCELERATION):
        self.root_body = root_body
        self.gravitational_acceleration = np.asarray(gravity, float).reshape(3).copy()
        self.bodies: List[RigidBody] = [root_body]
        self.tree_joints: List[Joint] = []
        self.non_tree_joints: List[Joint] = []
        self._pred: Dict[int, RigidBody] = {}
        self._succ: Dict[int, RigidBody] = {}
        self.environment: List["HalfSpace3D"] = []  # ContactEnvironment (mechanism.jl:13, contact.jl:231-238)
        self.modcount = 0

    # -- graph accessors
    def predecessor(self, joint: Joint) -> RigidBody:
        return self._pred[id(joint)]

    def successor(self, joint: Joint) -> RigidBody:
        return self._succ[id(joint)]

    @property
    def joints(self) -> List[Joint]:
        return self.tree_joints + self.non_tree_joints

    def has_loops(self) -> bool:
        return bool(self.non_tree_joints)

    def findbody(self, name: str) -> RigidBody:
        for b in self.bodies:
            if b.name == name:
                return b
        raise KeyError(name)

    def findjoint(self, name: str) -> Joint:
        for j in self.joints:
            if j.name == name:
                return j
        raise KeyError(name)

    def num_positions(self) -> int:
        return sum(j.joint_type.nq for j in self.tree_joints)

    def num_velocities(self) -> int:
        return sum(j.joint_type.nv for j in self.tree_joints)

    def num_constraints(self) -> int:
        return sum(6 - j.joint_type.nv for j in self.non_tree_joints)

    def joint_to_parent(self, body: RigidBody) -> Joint:
        for j in self.tree_joints:
            if self._succ[id(j)] is body:
                return j
        raise KeyError(body)

    def _canonicalize_frame_definitions(self, body: RigidBody):  # mechanism.jl:250-260
        if body is not self.root_body:
            body.change_default_frame(self.joint_to_parent(body).frame_after)
        for j in self.joints:
            if id(j) not in self._succ:
                continue
            if self._succ[id(j)] is body:
                j.joint_to_successor = body.frame_definition(j.frame_after)
            if self._pred[id(j)] is body:
                j.joint_to_predecessor = body.frame_definition(j.frame_before)


def attach_(mechanism: Mechanism, predecessor: RigidBody, successor: RigidBody, joint: Joint,
            joint_pose: Optional[Transform3D] = None, successor_pose: Optional[Transform3D] = None) -> Mechanism:
    """`attach!(mechanism, predecessor, successor, joint; joint_pose, successor_pose)`
    src/mechanism_modification.jl:21-46.  If `successor` is already part of the mechanism the joint becomes a
    non-tree (loop) joint enforced with Lagrange multipliers."""
    if joint_pose is None:
        joint_pose = Transform3D(joint.frame_before, predecessor.default_frame)
    if successor_pose is None:
        successor_pose = Transform3D(successor.default_frame, joint.frame_after)
    assert joint_pose.frm is joint.frame_before
    assert successor_pose.to is joint.frame_after
    assert any(b is predecessor for b in mechanism.bodies)
    assert all(j is not joint for j in mechanism.joints)
    joint.joint_to_predecessor = predecessor.add_frame(joint_pose)
    joint.joint_to_successor = successor.add_frame(successor_pose.inv())
    mechanism._pred[id(joint)] = predecessor
    mechanism._succ[id(joint)] = successor
    if any(b is successor for b in mechanism.bodies):
        mechanism.non_tree_joints.append(joint)
    else:
        mechanism.bodies.append(successor)
        mechanism.tree_joints.append(joint)
        mechanism._canonicalize_frame_definitions(successor)
    mechanism.modcount += 1
    return mechanism


def remove_fixed_tree_joints_(mechanism: Mechanism) -> Mechanism:
    """`remove_fixed_tree_joints!` src/mechanism_modification.jl:260-317: merge the successor of every fixed tree
    joint into its predecessor (frames migrate, inertias add), preserving the order of the remaining joints."""
    fixed = [j for j in mechanism.tree_joints if j.joint_type.tag == JOINT_FIXED]
    remaining = [j for j in mechanism.tree_joints if j.joint_type.tag != JOINT_FIXED]
    for fj in fixed:
        pred, succ = mechanism._pred[id(fj)], mechanism._succ[id(fj)]
        pred.add_frame(Transform3D(fj.frame_after, fj.frame_before))  # identity joint transform :276-277
        for tf in list(succ.frame_definitions.values()):               # migrate frames :280-282
            pred.add_frame(tf)
        for point in list(succ.contact_points):                         # migrate contact points :284-287
            add_contact_point_(pred, point)
        if pred.inertia is not None:                                    # :286-291 (root has no inertia)
            inertia = succ.inertia
            toparent = pred.fixed_transform(inertia.frame, pred.inertia.frame)
            pred.inertia = pred.inertia + inertia.transform(toparent)
        for j in mechanism.joints:                                      # rewire :294-303
            if j is fj or id(j) not in mechanism._succ:
                continue
            if mechanism._succ[id(j)] is succ:
                mechanism._succ[id(j)] = pred
            if mechanism._pred[id(j)] is succ:
                mechanism._pred[id(j)] = pred
        mechanism.bodies = [b for b in mechanism.bodies if b is not succ]
        del mechanism._pred[id(fj)], mechanism._succ[id(fj)]
    mechanism.tree_joints = remaining
    for b in mechanism.bodies:
        mechanism._canonicalize_frame_definitions(b)
    mechanism.modcount += 1
    return mechanism


# ---- the flat model ---------------------------------------------------------------------------------
# ---- soft contact (src/contact.jl) -------------------------------------------------------------------------------------------
class HuntCrossleyModel:  # contact.jl:98-119
    def __init__(self, k, lam, n):
        self.k, self.lam, self.n = float(k), float(lam), float(n)


def hunt_crossley_hertz(k=50e3, alpha=0.2) -> HuntCrossleyModel:  # contact.jl:104-107: λ = 3/2 α k  ((12) in Marhefka, Orin), n = 3/2
    return HuntCrossleyModel(k, 1.5 * alpha * k, 1.5)


class ViscoelasticCoulombModel:  # contact.jl:122-126
    def __init__(self, mu, k, b):
        self.mu, self.k, self.b = float(mu), float(k), float(b)


class SoftContactModel:  # contact.jl:38-41
    def __init__(self, normal: HuntCrossleyModel, friction: ViscoelasticCoulombModel):
        self.normal, self.friction = normal, friction


class ContactPoint:
    """ContactPoint(location::Point3D, model) (contact.jl:72-77): `location` is a 3-vector in `frame` (default: the body's default frame
    at the time it is added)."""

    def __init__(self, location, model: SoftContactModel, frame: Optional[CartesianFrame3D] = None):
        self.location = np.asarray(location, float).reshape(3).copy()
        self.model = model
        self.frame = frame


class HalfSpace3D:
    """HalfSpace3D(point, outward_normal) in the root frame; the normal is normalized (contact.jl:202-222)."""

    def __init__(self, point, outward_normal):
        self.point = np.asarray(point, float).reshape(3).copy()
        n = np.asarray(outward_normal, float).reshape(3)
        self.outward_normal = n / np.linalg.norm(n)


def add_contact_point_(body: RigidBody, point: ContactPoint):  # add_contact_point!: rigid_body.jl:173-179
    if point.frame is not None and point.frame is not body.default_frame:
        tf = body.fixed_transform(point.frame, body.default_frame)
        point.location = tf.R @ point.location + tf.p
    point.frame = body.default_frame
    body.contact_points.append(point)


def add_environment_primitive_(mechanism: "Mechanism", halfspace: HalfSpace3D):  # mechanism_modification.jl:375
    mechanism.environment.append(halfspace)


class _ContactPointC(ctypes.Structure):
    _fields_ = [("body", ctypes.c_int32), ("_pad", ctypes.c_int32), ("location", ctypes.c_double * 3),
                ("hc_k", ctypes.c_double), ("hc_lambda", ctypes.c_double), ("hc_n", ctypes.c_double),
                ("mu", ctypes.c_double), ("k", ctypes.c_double), ("b", ctypes.c_double)]


class _HalfSpaceC(ctypes.Structure):
    _fields_ = [("point", ctypes.c_double * 3), ("outward_normal", ctypes.c_double * 3)]


class _LoopJointC(ctypes.Structure):
    _fields_ = [
        ("predecessor", ctypes.c_int32), ("successor", ctypes.c_int32), ("joint_type", ctypes.c_int32), ("_pad", ctypes.c_int32),
        ("axis", ctypes.c_double * 3), ("pred_rot", ctypes.c_double * 9), ("pred_trans", ctypes.c_double * 3),
        ("succ_rot", ctypes.c_double * 9), ("succ_trans", ctypes.c_double * 3),
        ("rotation_from_z_aligned", ctypes.c_double * 9), ("gains", ctypes.c_double * 4),
    ]


class _FlatModelC(ctypes.Structure):
    _fields_ = [
        ("n_bodies", ctypes.c_int32), ("nq", ctypes.c_int32), ("nv", ctypes.c_int32), ("n_loops", ctypes.c_int32),
        ("parent", ctypes.POINTER(ctypes.c_int32)), ("joint_type", ctypes.POINTER(ctypes.c_int32)),
        ("q_offset", ctypes.POINTER(ctypes.c_int32)), ("v_offset", ctypes.POINTER(ctypes.c_int32)),
        ("joint_axis", ctypes.POINTER(ctypes.c_double)), ("joint_axis2", ctypes.POINTER(ctypes.c_double)),
        ("pred_rot", ctypes.POINTER(ctypes.c_double)), ("pred_trans", ctypes.POINTER(ctypes.c_double)),
        ("inertia_moment", ctypes.POINTER(ctypes.c_double)), ("inertia_cross", ctypes.POINTER(ctypes.c_double)),
        ("inertia_mass", ctypes.POINTER(ctypes.c_double)),
        ("gravity", ctypes.c_double * 3),
        ("loops", ctypes.POINTER(_LoopJointC)),
        ("n_contact_points", ctypes.c_int32), ("n_halfspaces", ctypes.c_int32),
        ("contact_points", ctypes.POINTER(_ContactPointC)), ("halfspaces", ctypes.POINTER(_HalfSpaceC)),
    ]


class FlatModel:
    """numpy image of `rbd_flat_model_t` (include/rbd_hip.h) — what MechanismState's constructor tabulates
    (src/mechanism_state.jl:85-118): parents, joint types, q/v ranges, joint_to_predecessor, body inertias."""

    def __init__(self, mechanism: Mechanism):
        tj = mechanism.tree_joints
        n = len(tj)
        body_index = {id(mechanism.root_body): -1}
        for i, j in enumerate(tj):
            body_index[id(mechanism.successor(j))] = i
        self.n_bodies = n
        self.body_names = [mechanism.successor(j).name for j in tj]
        self.joint_names = [j.name for j in tj]
        self.parent = np.array([body_index[id(mechanism.predecessor(j))] for j in tj], dtype=np.int32).reshape(n)
        self.joint_type = np.array([j.joint_type.tag for j in tj], dtype=np.int32).reshape(n)
        nqs = [j.joint_type.nq for j in tj]
        nvs = [j.joint_type.nv for j in tj]
        self.q_offset = np.concatenate([[0], np.cumsum(nqs)[:-1]]).astype(np.int32) if n else np.zeros(0, np.int32)
        self.v_offset = np.concatenate([[0], np.cumsum(nvs)[:-1]]).astype(np.int32) if n else np.zeros(0, np.int32)
        self.nq, self.nv = int(sum(nqs)), int(sum(nvs))
        self.joint_axis = np.array([j.joint_type.axis for j in tj], dtype=np.float64).reshape(n, 3)
        self.joint_axis2 = np.array([j.joint_type.axis2 for j in tj], dtype=np.float64).reshape(n, 3)
        self.pred_rot = np.zeros((n, 3, 3))
        self.pred_trans = np.zeros((n, 3))
        self.inertia_moment = np.zeros((n, 3, 3))
        self.inertia_cross = np.zeros((n, 3))
        self.inertia_mass = np.zeros(n)
        for i, j in enumerate(tj):
            succ = mechanism.successor(j)
            assert succ.default_frame is j.frame_after, "frames not canonical"
            assert np.allclose(j.joint_to_successor.R, np.eye(3)) and np.allclose(j.joint_to_successor.p, 0)
            self.pred_rot[i] = j.joint_to_predecessor.R
            self.pred_trans[i] = j.joint_to_predecessor.p
            assert succ.inertia.frame is succ.default_frame
            self.inertia_moment[i] = succ.inertia.moment
            self.inertia_cross[i] = succ.inertia.cross_part
            self.inertia_mass[i] = succ.inertia.mass
        self.gravity = mechanism.gravitational_acceleration.copy()
        # parents-first check (topological order, src/mechanism_state.jl:44)
        assert all(self.parent[i] < i for i in range(n))
        # loop joints
        self.loops = []
        for j in mechanism.non_tree_joints:
            self.loops.append(dict(
                name=j.name, predecessor=body_index[id(mechanism.predecessor(j))], successor=body_index[id(mechanism.successor(j))],
                joint_type=j.joint_type.tag, axis=j.joint_type.axis.copy(),
                pred_rot=j.joint_to_predecessor.R.copy(), pred_trans=j.joint_to_predecessor.p.copy(),
                succ_rot=j.joint_to_successor.R.copy(), succ_trans=j.joint_to_successor.p.copy(),
                rotation_from_z_aligned=j.joint_type.rotation_from_z_aligned.copy(), gains=tuple(j.stabilization_gains)))
        self.n_loops = len(self.loops)
        self.nc = sum(6 - _NV[l["joint_type"]] for l in self.loops)
        # soft contact: points in the order MechanismState lays out their states (for body in bodies(m), for point: mechanism_state.jl:143)
        self.contact_points = []
        for body in mechanism.bodies:
            for pt in body.contact_points:
                if body is mechanism.root_body:
                    raise ValueError("contact points on the root body are not supported")
                assert pt.frame is body.default_frame
                self.contact_points.append(dict(body=body_index[id(body)], location=pt.location.copy(), hc_k=pt.model.normal.k, hc_lambda=pt.model.normal.lam,
                                                hc_n=pt.model.normal.n, mu=pt.model.friction.mu, k=pt.model.friction.k, b=pt.model.friction.b))
        self.halfspaces = [dict(point=h.point.copy(), outward_normal=h.outward_normal.copy()) for h in mechanism.environment]
        self.ns = 3 * len(self.contact_points) * len(self.halfspaces)  # num_additional_states (mechanism.jl:143-149)
        self._c = None

    # levels (BFS depth) — used by docs/tests; the kernels only need parents-first order
    def levels(self) -> np.ndarray:
        lv = np.zeros(self.n_bodies, dtype=np.int32)
        for i in range(self.n_bodies):
            lv[i] = 0 if self.parent[i] < 0 else lv[self.parent[i]] + 1
        return lv

    def total_mass(self) -> float:
        return float(self.inertia_mass.sum())

    def c_struct(self) -> _FlatModelC:
        """ctypes view (arrays are kept alive by `self`)."""
        if self._c is None:
            def p(a, t):
                return a.ctypes.data_as(ctypes.POINTER(t))
            for name in ("parent", "joint_type", "q_offset", "v_offset"):
                setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.int32))
            for name in ("joint_axis", "joint_axis2", "pred_rot", "pred_trans", "inertia_moment", "inertia_cross", "inertia_mass"):
                setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.float64))
            c = _FlatModelC()
            c.n_bodies, c.nq, c.nv, c.n_loops = self.n_bodies, self.nq, self.nv, self.n_loops
            c.parent, c.joint_type = p(self.parent, ctypes.c_int32), p(self.joint_type, ctypes.c_int32)
            c.q_offset, c.v_offset = p(self.q_offset, ctypes.c_int32), p(self.v_offset, ctypes.c_int32)
            c.joint_axis, c.joint_axis2 = p(self.joint_axis, ctypes.c_double), p(self.joint_axis2, ctypes.c_double)
            c.pred_rot, c.pred_trans = p(self.pred_rot, ctypes.c_double), p(self.pred_trans, ctypes.c_double)
            c.inertia_moment, c.inertia_cross = p(self.inertia_moment, ctypes.c_double), p(self.inertia_cross, ctypes.c_double)
            c.inertia_mass = p(self.inertia_mass, ctypes.c_double)
            c.gravity = (ctypes.c_double * 3)(*self.gravity)
            if self.n_loops:
                arr = (_LoopJointC * self.n_loops)()
                for k, l in enumerate(self.loops):
                    arr[k].predecessor, arr[k].successor, arr[k].joint_type = l["predecessor"], l["successor"], l["joint_type"]
                    arr[k].axis = (ctypes.c_double * 3)(*l["axis"])
                    arr[k].pred_rot = (ctypes.c_double * 9)(*l["pred_rot"].ravel())
                    arr[k].pred_trans = (ctypes.c_double * 3)(*l["pred_trans"])
                    arr[k].succ_rot = (ctypes.c_double * 9)(*l["succ_rot"].ravel())
                    arr[k].succ_trans = (ctypes.c_double * 3)(*l["succ_trans"])
                    arr[k].rotation_from_z_aligned = (ctypes.c_double * 9)(*l["rotation_from_z_aligned"].ravel())
                    arr[k].gains = (ctypes.c_double * 4)(*l["gains"])
                self._loops_c = arr
                c.loops = ctypes.cast(arr, ctypes.POINTER(_LoopJointC))
            c.n_contact_points, c.n_halfspaces = len(self.contact_points), len(self.halfspaces)
            if self.contact_points:
                arr = (_ContactPointC * len(self.contact_points))()
                for k, d in enumerate(self.contact_points):
                    arr[k].body = d["body"]
                    arr[k].location = (ctypes.c_double * 3)(*d["location"])
                    arr[k].hc_k, arr[k].hc_lambda, arr[k].hc_n = d["hc_k"], d["hc_lambda"], d["hc_n"]
                    arr[k].mu, arr[k].k, arr[k].b = d["mu"], d["k"], d["b"]
                self._cp_c = arr
                c.contact_points = ctypes.cast(arr, ctypes.POINTER(_ContactPointC))
            if self.halfspaces:
                arr = (_HalfSpaceC * len(self.halfspaces))()
                for k, d in enumerate(self.halfspaces):
                    arr[k].point = (ctypes.c_double * 3)(*d["point"])
                    arr[k].outward_normal = (ctypes.c_double * 3)(*d["outward_normal"])
                self._hs_c = arr
                c.halfspaces = ctypes.cast(arr, ctypes.POINTER(_HalfSpaceC))
            self._c = c
        return self._c


_NV = {JOINT_FIXED: 0, JOINT_REVOLUTE: 1, JOINT_PRISMATIC: 1, JOINT_QUAT_FLOATING: 6, JOINT_PLANAR: 3,
       JOINT_QUAT_SPHERICAL: 3, JOINT_SINCOS_REVOLUTE: 1}
_NQ = {JOINT_FIXED: 0, JOINT_REVOLUTE: 1, JOINT_PRISMATIC: 1, JOINT_QUAT_FLOATING: 7, JOINT_PLANAR: 3,
       JOINT_QUAT_SPHERICAL: 4, JOINT_SINCOS_REVOLUTE: 2}


def flatten(mechanism: Mechanism) -> FlatModel:
    return FlatModel(mechanism)


# ---- random inputs with the reference's distributions (SURVEY.md §8 d) -------------------------------
def rand_configuration(model: FlatModel, B: int, rng: np.random.Generator) -> np.ndarray:
    """`rand_configuration!` per joint type: revolute q ~ N(0,1) (revolute.jl:54-57); floating: quaternion uniform
    on S³, translation U(-½,½)³ (quaternion_floating.jl:175-180); planar (planar.jl rand_configuration!).
    Returns q as (B, nq) float64 (AOS: one state per row == Julia column)."""
    q = np.zeros((B, model.nq))
    for i in range(model.n_bodies):
        o, t = int(model.q_offset[i]), int(model.joint_type[i])
        if t in (JOINT_REVOLUTE, JOINT_PRISMATIC):
            q[:, o] = rng.standard_normal(B)
        elif t == JOINT_SINCOS_REVOLUTE:
            th = rng.standard_normal(B)
            q[:, o], q[:, o + 1] = np.sin(th), np.cos(th)
        elif t in (JOINT_QUAT_FLOATING, JOINT_QUAT_SPHERICAL):
            quat = rng.standard_normal((B, 4))
            quat /= np.linalg.norm(quat, axis=1, keepdims=True)
            q[:, o:o + 4] = quat
            if t == JOINT_QUAT_FLOATING:
                q[:, o + 4:o + 7] = rng.random((B, 3)) - 0.5
        elif t == JOINT_PLANAR:
            q[:, o:o + 2] = rng.random((B, 2)) - 0.5
            q[:, o + 2] = rng.standard_normal(B)
    return q


def rand_velocity(model: FlatModel, B: int, rng: np.random.Generator) -> np.ndarray:
    """`rand_velocity!`: v ~ U[0,1) (src/mechanism_state.jl:342-346)."""
    return rng.random((B, model.nv))
