"""Programmatic mechanisms with the literal constants of the reference's examples/tests — BASELINE.json
configs[0] (double pendulum) and configs[4] (four-bar linkage)."""
from __future__ import annotations

import numpy as np

from .mechanism import (CartesianFrame3D, Joint, Mechanism, QuaternionFloating, Revolute, RigidBody, SpatialInertia, Transform3D, attach_)


def double_pendulum(lc1=-0.5, l1=-1.0, m1=1.0, I1=0.333, lc2=-1.0, m2=1.0, I2=1.33, g=-9.81) -> Mechanism:
    """The test-suite double pendulum, test/test_double_pendulum.jl:3-30 (axis y, inertias about the joint).
    The Quickstart example ("examples/1. Quickstart - double pendulum":21-82) is the same construction with
    I1 = I2 = 0.333, lc1 = lc2 = -0.5, l1 = -1."""
    axis = np.array([0.0, 1.0, 0.0])
    world = RigidBody("world")
    mech = Mechanism(world, gravity=[0, 0, g])
    inertia1 = SpatialInertia(CartesianFrame3D("upper_link"), moment=I1 * np.outer(axis, axis), com=[0, 0, lc1], mass=m1)
    body1 = RigidBody(inertia1)
    joint1 = Joint("shoulder", Revolute(axis))
    attach_(mech, world, body1, joint1, joint_pose=Transform3D(joint1.frame_before, world.default_frame))
    inertia2 = SpatialInertia(CartesianFrame3D("lower_link"), moment=I2 * np.outer(axis, axis), com=[0, 0, lc2], mass=m2)
    body2 = RigidBody(inertia2)
    joint2 = Joint("elbow", Revolute(axis))
    attach_(mech, body1, body2, joint2, joint_pose=Transform3D(joint2.frame_before, body1.default_frame, p=[0, 0, l1]))
    return mech


def quickstart_double_pendulum() -> Mechanism:
    """examples/1. Quickstart - double pendulum/1. Quickstart - double pendulum.jl:21-82."""
    return double_pendulum(lc1=-0.5, l1=-1.0, m1=1.0, I1=0.333, lc2=-0.5, m2=1.0, I2=0.333)


def four_bar_linkage() -> Mechanism:
    """test/test_simulate.jl:127-190 (the example file differs in com_3; the test's constants are used)."""
    g = -9.81
    l_0, l_1, l_2, l_3 = 1.10, 0.5, 1.20, 0.75
    m_1, m_2, m_3 = 0.5, 1.0, 0.75
    c_1, c_2, c_3 = 0.25, 0.60, 0.375
    I_1, I_2, I_3 = 0.333, 0.537, 0.4
    axis = np.array([0.0, -1.0, 0.0])
    world = RigidBody("world")
    mech = Mechanism(world, gravity=[0.0, 0.0, g])

    def link(name, I, m):
        return RigidBody(SpatialInertia(CartesianFrame3D(name), moment=I * np.outer(axis, axis), com=np.zeros(3), mass=m))

    joint1 = Joint("joint1", Revolute(axis))
    link1 = link("inertia1_centroidal", I_1, m_1)
    attach_(mech, world, link1, joint1, joint_pose=Transform3D(joint1.frame_before, world.default_frame),
            successor_pose=Transform3D(link1.default_frame, joint1.frame_after, p=[c_1, 0, 0]))
    joint2 = Joint("joint2", Revolute(axis))
    link2 = link("inertia2_centroidal", I_2, m_2)
    attach_(mech, link1, link2, joint2, joint_pose=Transform3D(joint2.frame_before, joint1.frame_after, p=[l_1, 0, 0]),
            successor_pose=Transform3D(link2.default_frame, joint2.frame_after, p=[c_2, 0, 0]))
    joint3 = Joint("joint3", Revolute(axis))
    link3 = link("inertia3_centroidal", I_3, m_3)
    attach_(mech, world, link3, joint3, joint_pose=Transform3D(joint3.frame_before, world.default_frame, p=[l_0, 0, 0]),
            successor_pose=Transform3D(link3.default_frame, joint3.frame_after, p=[c_3, 0, 0]))
    # loop joint between link2 and link3
    joint4 = Joint("joint4", Revolute(axis))
    attach_(mech, link2, link3, joint4, joint_pose=Transform3D(joint4.frame_before, joint2.frame_after, p=[l_2, 0, 0]),
            successor_pose=Transform3D(joint3.frame_after, joint4.frame_after, p=[-l_3, 0, 0]))
    return mech


FOUR_BAR_INITIAL_Q = np.array([1.6707963267948966, -1.4591054166649482, 1.5397303602625536])  # test_simulate.jl:195-197
FOUR_BAR_INITIAL_V = np.array([0.5, -0.47295, 0.341])                                          # :198-200


# ---- random mechanisms in the style of the reference's test fixtures ------------------------------------------------
def _rand_rotation(rng) -> np.ndarray:
    q = rng.standard_normal(4)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rand_spatial_inertia(rng, frame) -> SpatialInertia:
    """`rand(SpatialInertia{T}, frame)` src/spatial/motion_force_interaction.jl:178-196."""
    ixx, iyy = rng.random() / 10.0, rng.random() / 10.0
    lb, ub = abs(ixx - iyy), ixx + iyy
    izz = rng.random() * (ub - lb) + lb
    R = _rand_rotation(rng)
    return SpatialInertia(frame, moment_about_com=R @ np.diag([ixx, iyy, izz]) @ R.T, com=rng.random(3) - 0.5, mass=rng.random())


def rand_joint_type(rng, name: str):
    """`rand(JointType)`: random unit axes (revolute.jl:30-33, prismatic.jl, planar.jl:37-42)."""
    from .mechanism import Fixed, Planar, Prismatic, QuaternionFloating, QuaternionSpherical, Revolute, SinCosRevolute
    ax = lambda: rng.standard_normal(3)
    if name == "Planar":
        x = ax(); x /= np.linalg.norm(x)
        y = ax(); y -= (x @ y) * x
        return Planar(x, y)
    return {"Revolute": lambda: Revolute(ax()), "Prismatic": lambda: Prismatic(ax()), "SinCosRevolute": lambda: SinCosRevolute(ax()),
            "Fixed": Fixed, "QuaternionFloating": QuaternionFloating, "QuaternionSpherical": QuaternionSpherical}[name]()


def rand_tree_mechanism(rng, joint_types, parentselector=None) -> Mechanism:
    """`rand_tree_mechanism(parentselector, jointtypes...)` src/mechanism_modification.jl:382-396: each new body hangs off a
    parent chosen among ALL bodies (the world included) through a random joint pose."""
    world = RigidBody("world")
    mech = Mechanism(world)
    parent = world
    for i, name in enumerate(joint_types):
        joint = Joint(f"joint{i + 1}", rand_joint_type(rng, name))
        pose = Transform3D(joint.frame_before, parent.default_frame, _rand_rotation(rng), rng.random(3))
        body = RigidBody(rand_spatial_inertia(rng, CartesianFrame3D(f"body{i + 1}")))
        attach_(mech, parent, body, joint, joint_pose=pose, successor_pose=Transform3D(body.default_frame, joint.frame_after))
        parent = parentselector(mech, rng) if parentselector else mech.bodies[rng.integers(len(mech.bodies))]
    return mech


def randmech(rng) -> Mechanism:
    """test/test_mechanism_algorithms.jl:1-11 with SPQuatFloating (out of scope) replaced by QuaternionSpherical."""
    return rand_tree_mechanism(rng, ["QuaternionFloating"] + ["Revolute"] * 5 + ["Fixed"] * 5 + ["Prismatic"] * 5 + ["Planar"] * 5 +
                               ["QuaternionSpherical"] * 2 + ["SinCosRevolute"] * 2)


def maximal_coordinates(mechanism: Mechanism):
    """`maximal_coordinates(mechanism)` src/mechanism_modification.jl:335-362: a dynamically equivalent mechanism with a flat
    tree — every body hangs off the root on its own QuaternionFloating joint (frame_before = the root's frame, frame_after = the
    body's default frame) and the joints of the input become non-tree joints enforced with Lagrange multipliers.  Bodies and
    joints keep their order, so body i / loop joint i of the result corresponds to body i / tree joint i of the input."""
    import copy
    old_joints = mechanism.tree_joints + mechanism.non_tree_joints
    bodies, joints = copy.deepcopy((mechanism.bodies, old_joints))  # one deepcopy: shared frame objects stay shared
    bodymap = {id(o): n for o, n in zip(mechanism.bodies, bodies)}
    root = bodies[0]
    ret = Mechanism(root, gravity=mechanism.gravitational_acceleration)
    for body in bodies[1:]:
        fj = Joint(body.name, QuaternionFloating())
        fj.frame_before, fj.frame_after = root.default_frame, body.default_frame
        attach_(ret, root, body, fj, joint_pose=Transform3D(root.default_frame, root.default_frame),
                successor_pose=Transform3D(body.default_frame, body.default_frame))
    for oj, nj in zip(old_joints, joints):  # _copyjoint!: mechanism_modification.jl:48-61
        sp, ss = mechanism.predecessor(oj), mechanism.successor(oj)
        j2p = sp.fixed_transform(oj.frame_before, sp.default_frame)
        s2j = ss.fixed_transform(ss.default_frame, oj.frame_after)
        dp, ds = bodymap[id(sp)], bodymap[id(ss)]
        attach_(ret, dp, ds, nj, joint_pose=Transform3D(nj.frame_before, dp.default_frame, j2p.R, j2p.p),
                successor_pose=Transform3D(ds.default_frame, nj.frame_after, s2j.R, s2j.p))
    return ret


def tree_mechanism(rng, spec, axis_aligned=False) -> Mechanism:
    """A tree of a GIVEN shape with random bodies: `spec` is a list of (joint type name, [children specs]) hanging off the world.  Limbs of the same shape
    (the same nested spec twice under one parent) get their own random poses and inertias — what the kernels that walk such limbs in lockstep must cope with.
    axis_aligned: joint axes and poses drawn from the coordinate axes / quarter turns, the way URDF robots are built (many structural zeros in the constants)."""
    from .mechanism import Fixed, Planar, Prismatic, QuaternionFloating, QuaternionSpherical, Revolute, SinCosRevolute
    world = RigidBody("world")
    mech = Mechanism(world)
    count = [0]

    def pose_rotation():
        if not axis_aligned:
            return _rand_rotation(rng)
        P = np.eye(3)[rng.permutation(3)] * rng.choice([-1.0, 1.0], 3)[:, None]
        return P if np.linalg.det(P) > 0 else -P

    def joint_type(name):
        if not axis_aligned or name in ("Fixed", "QuaternionFloating", "QuaternionSpherical", "Planar"):
            return rand_joint_type(rng, name)
        ax = np.eye(3)[rng.integers(3)] * rng.choice([-1.0, 1.0])
        return {"Revolute": Revolute, "Prismatic": Prismatic, "SinCosRevolute": SinCosRevolute}[name](ax)

    def grow(parent, node):
        name, children = node
        count[0] += 1
        i = count[0]
        joint = Joint(f"joint{i}", joint_type(name))
        p = rng.random(3) * (rng.random(3) < 0.5 if axis_aligned else 1.0)
        pose = Transform3D(joint.frame_before, parent.default_frame, pose_rotation(), p)
        body = RigidBody(rand_spatial_inertia(rng, CartesianFrame3D(f"body{i}")))
        attach_(mech, parent, body, joint, joint_pose=pose, successor_pose=Transform3D(body.default_frame, joint.frame_after))
        for c in children:
            grow(body, c)

    for node in spec:
        grow(world, node)
    return mech
