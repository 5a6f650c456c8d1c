"""Minimal URDF front-end reproducing `parse_urdf`'s *semantics* (src/urdf/parse.jl:162-221) for the
flattener — needed because the benchmark mechanism (Atlas) is a URDF fixture:

  * links/joints form a graph; the spanning tree is built breadth-first with a FIFO frontier and
    out-edges in document order (src/graphs/spanning_tree.jl:45-83) — this fixes the joint order and
    therefore the q/v index order;
  * joint <origin> = joint_to_predecessor, rpy -> RotZYX(yaw, pitch, roll) (parse.jl:46-51, 136-137);
  * <inertial>: moment about the COM in the <origin> frame, re-expressed in the link frame
    (parse.jl:104-112); missing <inertial> => zero inertia (parse.jl:116);
  * root link attached to the world with a Fixed or QuaternionFloating joint named
    "<root>_to_world" (parse.jl:121-127);
  * fixed tree joints removed afterwards by default (parse.jl:216-218).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from typing import Dict, Optional

import numpy as np

from .mechanism import (DEFAULT_GRAVITATIONAL_ACCELERATION, CartesianFrame3D, Fixed, Joint, JointType, Mechanism, Planar,
                        Prismatic, QuaternionFloating, Revolute, RigidBody, SpatialInertia, Transform3D, attach_,
                        remove_fixed_tree_joints_, rot_z_y_x, rotation_between)


def default_urdf_joint_types() -> Dict[str, object]:  # parse.jl:6-15
    return {"revolute": Revolute, "continuous": Revolute, "prismatic": Prismatic,
            "floating": QuaternionFloating, "fixed": Fixed, "planar": Planar}


def _vec(e: Optional[ET.Element], name: str, default: str):
    s = default if (e is None or e.get(name) is None) else e.get(name)
    return [float(x) for x in s.split()]


def parse_pose(e: Optional[ET.Element]):  # parse.jl:40-51
    if e is None:
        return np.eye(3), np.zeros(3)
    rpy = _vec(e, "rpy", "0 0 0")
    return rot_z_y_x(rpy[2], rpy[1], rpy[0]), np.array(_vec(e, "xyz", "0 0 0"))


def _parse_joint_type(xml_joint: ET.Element, joint_types) -> JointType:  # parse.jl:53-74
    t = xml_joint.get("type")
    ctor = joint_types[t]
    if t in ("revolute", "continuous", "prismatic"):
        return ctor(np.array(_vec(xml_joint.find("axis"), "xyz", "1 0 0")))
    if t in ("floating", "fixed"):
        return ctor()
    if t == "planar":
        axis = np.array(_vec(xml_joint.find("axis"), "xyz", "1 0 0"))
        R = rotation_between([0.0, 0.0, 1.0], axis)
        return ctor(R @ np.array([1.0, 0, 0]), R @ np.array([0, 1.0, 0]))
    raise ValueError(f"joint type {t} not recognized")


def _parse_body(xml_link: ET.Element, frame: Optional[CartesianFrame3D] = None) -> RigidBody:  # parse.jl:104-119
    name = xml_link.get("name")
    frame = frame or CartesianFrame3D(name)
    xi = xml_link.find("inertial")
    if xi is None:
        inertia = SpatialInertia(frame)
    else:
        ie = xi.find("inertia")
        g = lambda k: float(ie.get(k, "0")) if ie is not None else 0.0
        moment = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        me = xi.find("mass")
        mass = float(me.get("value", "0")) if me is not None else 0.0
        helper = CartesianFrame3D("inertia urdf helper")
        R, p = parse_pose(xi.find("origin"))
        inertia = SpatialInertia(helper, moment, np.zeros(3), mass).transform(Transform3D(helper, frame, R, p))
    return RigidBody(name, inertia)


def parse_urdf(filename: str, floating: bool = False, joint_types=None, root_joint_type: Optional[JointType] = None,
               remove_fixed_tree_joints: bool = True, gravity=DEFAULT_GRAVITATIONAL_ACCELERATION) -> Mechanism:
    """`parse_urdf(filename; floating, joint_types, root_joint_type, remove_fixed_tree_joints, gravity)`."""
    joint_types = joint_types or default_urdf_joint_types()
    if root_joint_type is None:
        root_joint_type = joint_types["floating" if floating else "fixed"]()
    if floating and root_joint_type.nv != 6:
        raise ValueError("Ambiguous input arguments: `floating` specified, but `root_joint_type` is not a floating joint type.")
    xroot = ET.parse(filename).getroot()
    assert xroot.tag == "robot"
    links = xroot.findall("link")
    joints = xroot.findall("joint")
    name_to_link = {l.get("name"): l for l in links}
    out_edges = {l.get("name"): [] for l in links}
    in_count = {l.get("name"): 0 for l in links}
    for j in joints:
        parent, child = j.find("parent").get("link"), j.find("child").get("link")
        out_edges[parent].append(j)
        in_count[child] += 1
    roots = [l for l in links if in_count[l.get("name")] == 0]
    if len(roots) != 1:
        raise ValueError("Can only handle a single root")
    # breadth-first spanning tree, FIFO frontier, out-edges in document order (spanning_tree.jl:45-83).
    # (URDFs are trees, so no edge ever needs flipping.)
    tree_edges, visited = [], {roots[0].get("name")}
    frontier = list(out_edges[roots[0].get("name")])
    while frontier:
        e = frontier.pop(0)
        child = e.find("child").get("link")
        if child in visited:
            continue
        visited.add(child)
        tree_edges.append(e)
        frontier.extend(x for x in out_edges[child] if x.find("child").get("link") not in visited)
    if len(visited) != len(links):
        raise ValueError("Graph is not connected.")

    world = RigidBody("world")
    mechanism = Mechanism(world, gravity=gravity)
    # parse_root_link: parse.jl:121-127
    rootbody = _parse_body(roots[0])
    rootjoint = Joint(f"{rootbody.name}_to_world", root_joint_type)
    attach_(mechanism, world, rootbody, rootjoint, joint_pose=Transform3D(rootjoint.frame_before, world.default_frame))
    bodies = {rootbody.name: rootbody}
    for e in tree_edges:  # parse_joint_and_link: parse.jl:129-140
        parent = bodies[e.find("parent").get("link")]
        joint = Joint(e.get("name"), _parse_joint_type(e, joint_types))
        R, p = parse_pose(e.find("origin"))
        joint_to_parent = Transform3D(joint.frame_before, parent.default_frame, R, p)
        body = _parse_body(name_to_link[e.find("child").get("link")], joint.frame_after)
        if body.name in bodies:
            raise ValueError(f"Duplicate name: {body.name}")
        bodies[body.name] = body
        attach_(mechanism, parent, body, joint, joint_pose=joint_to_parent)
    if remove_fixed_tree_joints:
        remove_fixed_tree_joints_(mechanism)
    return mechanism


# ---- write_urdf (src/urdf/write.jl) ---------------------------------------------------------------------------------------------
def _fmt(vec) -> str:
    return " ".join(repr(float(x)) for x in vec)


def _rpy_of(R: np.ndarray):
    """RotZYX(R) -> (roll, pitch, yaw) with R = Rz(yaw) Ry(pitch) Rx(roll) (write.jl:108, 119; parse.jl:47-48 is its inverse)."""
    pitch = np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0]))
    yaw = np.arctan2(R[1, 0], R[0, 0])
    roll = np.arctan2(R[2, 1], R[2, 2])
    return roll, pitch, yaw


def _link_xml(body: RigidBody, is_root: bool) -> ET.Element:  # to_urdf(body): write.jl:5-37
    x = ET.Element("link", {"name": body.name})
    if not is_root and body.inertia is not None:
        inertia = body.inertia
        xi = ET.SubElement(x, "inertial")
        if inertia.mass > 0:
            com = inertia.cross_part / inertia.mass
            centroidal = CartesianFrame3D("centroidal")
            inertia = inertia.transform(Transform3D(inertia.frame, centroidal, np.eye(3), -com))  # moment about the centre of mass
        else:
            com = np.zeros(3)
        ET.SubElement(xi, "origin", {"xyz": _fmt(com), "rpy": _fmt(np.zeros(3))})
        ET.SubElement(xi, "mass", {"value": repr(float(inertia.mass))})
        J = inertia.moment
        ET.SubElement(xi, "inertia", {"ixx": repr(float(J[0, 0])), "ixy": repr(float(J[0, 1])), "ixz": repr(float(J[0, 2])),
                                      "iyy": repr(float(J[1, 1])), "iyz": repr(float(J[1, 2])), "izz": repr(float(J[2, 2]))})
    return x


def _joint_xml(joint: Joint, mechanism: Mechanism) -> ET.Element:  # to_urdf(joint, mechanism): write.jl:102-123
    from .mechanism import (JOINT_FIXED, JOINT_PLANAR, JOINT_PRISMATIC, JOINT_QUAT_FLOATING, JOINT_REVOLUTE)
    to_parent = joint.joint_to_predecessor
    x = ET.Element("joint", {"name": joint.name})
    ET.SubElement(x, "parent", {"link": mechanism.predecessor(joint).name})
    ET.SubElement(x, "child", {"link": mechanism.successor(joint).name})
    ET.SubElement(x, "origin", {"xyz": _fmt(to_parent.p), "rpy": _fmt(_rpy_of(to_parent.R))})
    jt = joint.joint_type
    if jt.tag == JOINT_QUAT_FLOATING:
        x.set("type", "floating")
    elif jt.tag == JOINT_FIXED:
        x.set("type", "fixed")
    elif jt.tag == JOINT_PLANAR:
        x.set("type", "planar")
        ET.SubElement(x, "axis", {"xyz": _fmt(np.cross(jt.axis, jt.axis2))})
    elif jt.tag in (JOINT_REVOLUTE, JOINT_PRISMATIC):
        ET.SubElement(x, "axis", {"xyz": _fmt(jt.axis)})
        # joint bounds are not part of the hot path's model: every revolute joint is written as the reference writes an unbounded one
        x.set("type", "continuous" if jt.tag == JOINT_REVOLUTE else "prismatic")
        lim = {"effort": "Inf", "velocity": "Inf"}
        if jt.tag == JOINT_PRISMATIC:
            lim.update({"lower": "-Inf", "upper": "Inf"})
        ET.SubElement(x, "limit", lim)
    else:
        raise ValueError(f"Joint type of {joint.name} not handled.")  # write.jl:46
    return x


def write_urdf(filename: str, mechanism: Mechanism, robot_name: Optional[str] = None, include_root: bool = True) -> None:
    """`write_urdf(filename, mechanism; robot_name, include_root)` (src/urdf/write.jl:125-200): <link> with <inertial> (origin at the centre
    of mass, centroidal moment), <joint> with <origin>, <parent>, <child>, <axis>, <limit>.  Tree mechanisms only."""
    assert not mechanism.has_loops()
    root = ET.Element("robot")
    if robot_name is not None:
        root.set("name", robot_name)
    for body in mechanism.bodies:
        if body is mechanism.root_body and not include_root:
            continue
        root.append(_link_xml(body, body is mechanism.root_body))
    for joint in mechanism.tree_joints:
        if mechanism.predecessor(joint) is mechanism.root_body and not include_root:
            continue
        root.append(_joint_xml(joint, mechanism))
    ET.ElementTree(root).write(filename, encoding="utf-8", xml_declaration=True)
