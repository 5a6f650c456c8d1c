"""(De)serialise a FlatModel as JSON — lets a flattened mechanism travel without its URDF
(the GPU box has no /root/reference; tests/golden/models/*.json are produced by tests/golden/make_models.py)."""
from __future__ import annotations

import json

import numpy as np

from .mechanism import FlatModel

_ARRAYS = ("parent", "joint_type", "q_offset", "v_offset", "joint_axis", "joint_axis2", "pred_rot", "pred_trans",
           "inertia_moment", "inertia_cross", "inertia_mass", "gravity")


def save_flat_model(model: FlatModel, path: str, meta=None):
    d = {k: np.asarray(getattr(model, k)).tolist() for k in _ARRAYS}
    d.update(n_bodies=model.n_bodies, nq=model.nq, nv=model.nv, body_names=model.body_names, joint_names=model.joint_names,
             loops=[{k: (np.asarray(v).tolist() if isinstance(v, (np.ndarray, tuple)) else v) for k, v in l.items()} for l in model.loops],
             meta=meta or {})
    if getattr(model, "contact_points", None) or getattr(model, "halfspaces", None):  # only mechanisms that have them (the URDF fixtures do not)
        d["contact_points"] = [dict(c, location=np.asarray(c["location"]).tolist()) for c in model.contact_points]
        d["halfspaces"] = [dict(point=np.asarray(h["point"]).tolist(), outward_normal=np.asarray(h["outward_normal"]).tolist()) for h in model.halfspaces]
    with open(path, "w") as f:
        json.dump(d, f, indent=0)


def load_flat_model(path: str) -> FlatModel:
    with open(path) as f:
        d = json.load(f)
    m = FlatModel.__new__(FlatModel)
    m.n_bodies, m.nq, m.nv = d["n_bodies"], d["nq"], d["nv"]
    m.body_names, m.joint_names = d["body_names"], d["joint_names"]
    for k in ("parent", "joint_type", "q_offset", "v_offset"):
        setattr(m, k, np.array(d[k], dtype=np.int32).reshape(m.n_bodies))
    m.joint_axis = np.array(d["joint_axis"], dtype=np.float64).reshape(m.n_bodies, 3)
    m.joint_axis2 = np.array(d["joint_axis2"], dtype=np.float64).reshape(m.n_bodies, 3)
    m.pred_rot = np.array(d["pred_rot"], dtype=np.float64).reshape(m.n_bodies, 3, 3)
    m.pred_trans = np.array(d["pred_trans"], dtype=np.float64).reshape(m.n_bodies, 3)
    m.inertia_moment = np.array(d["inertia_moment"], dtype=np.float64).reshape(m.n_bodies, 3, 3)
    m.inertia_cross = np.array(d["inertia_cross"], dtype=np.float64).reshape(m.n_bodies, 3)
    m.inertia_mass = np.array(d["inertia_mass"], dtype=np.float64).reshape(m.n_bodies)
    m.gravity = np.array(d["gravity"], dtype=np.float64)
    m.loops = []
    for l in d["loops"]:
        l = dict(l)
        for k in ("axis", "pred_rot", "pred_trans", "succ_rot", "succ_trans", "rotation_from_z_aligned"):
            l[k] = np.array(l[k], dtype=np.float64)
        l["pred_rot"] = l["pred_rot"].reshape(3, 3)
        l["succ_rot"] = l["succ_rot"].reshape(3, 3)
        l["rotation_from_z_aligned"] = l["rotation_from_z_aligned"].reshape(3, 3)
        l["gains"] = tuple(l["gains"])
        m.loops.append(l)
    m.n_loops = len(m.loops)
    from .mechanism import _NV
    m.nc = sum(6 - _NV[l["joint_type"]] for l in m.loops)
    m.contact_points = [dict(c, location=np.array(c["location"], dtype=np.float64)) for c in d.get("contact_points", [])]
    m.halfspaces = [dict(point=np.array(h["point"], dtype=np.float64), outward_normal=np.array(h["outward_normal"], dtype=np.float64)) for h in d.get("halfspaces", [])]
    m.ns = 3 * len(m.contact_points) * len(m.halfspaces)
    m._c = None
    return m
