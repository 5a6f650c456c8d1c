"""Checks julia/RigidBodyDynamicsGPU.jl against include/rbd_hip.h without Julia (the build image has none):
  * every `ccall`: the symbol exists, the argument count matches the prototype and every argument is of the same kind (pointer / 32-bit
    integer / 64-bit integer / double);
  * every C struct the shim mirrors (RbdLoopJoint, RbdContactPoint, RbdHalfSpace, RbdFlatModel, RbdOpts): field by field, the offset and
    size Julia's isbits layout rules give (= the platform C rules) against `offsetof` / `sizeof` printed by a g++-compiled dump of the
    header, and the field names.
Exit status 0 = consistent; run by tests/test_capi_symbols.py."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_kind(param):
    p = re.sub(r"/\*.*?\*/", "", param).strip()
    if p in ("void", ""):
        return None
    if "*" in p:
        return "ptr"
    t = p.rsplit(" ", 1)[0] if " " in p else p
    if "int64_t" in t:
        return "i64"
    if "double" in t:
        return "f64"
    if "int32_t" in t or t.strip() in ("int", "const int"):
        return "i32"
    raise ValueError(f"unclassified C parameter: {param!r}")


def jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
        return "ptr"
    if t in ("Int32", "Cint", "UInt32"):
        return "i32"
    if t in ("Int64", "Clong", "Csize_t"):
        return "i64"
    if t in ("Cdouble", "Float64"):
        return "f64"
    raise ValueError(f"unclassified Julia ccall type: {t!r}")


def header_prototypes():
    h = open(os.path.join(ROOT, "include", "rbd_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", " ", h, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|void|const char\*)\s+(rbd_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        params = [c_kind(p) for p in split_top(" ".join(m.group(2).split()))]
        protos[m.group(1)] = [k for k in params if k is not None]
    return protos


def julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "RigidBodyDynamicsGPU.jl")).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*(\w+)\[\]\),\s*(\w+),\s*\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = src[m.end():i - 1]
        calls.append((m.group(1), m.group(2), [jl_kind(t) for t in split_top(args) if t.strip()], src.count("\n", 0, m.start()) + 1))
    return calls


STRUCTS = {"RbdLoopJoint": "rbd_loop_joint_t", "RbdContactPoint": "rbd_contact_point_t", "RbdHalfSpace": "rbd_halfspace_t",
           "RbdFlatModel": "rbd_flat_model_t", "RbdOpts": "rbd_opts_t", "RbdControl": "rbd_control_t"}


def header_struct_fields():
    """{c struct: [field names in declaration order]} from the header text."""
    h = open(os.path.join(ROOT, "include", "rbd_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", " ", h, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct \w+\s*\{(.*?)\}\s*(\w+)\s*;", h, flags=re.S):
        names = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            first, *rest = split_top(decl)
            names.append(re.sub(r"\[.*", "", first.split()[-1].lstrip("*")))
            names += [re.sub(r"\[.*", "", r.strip().lstrip("*")) for r in rest]
        out[m.group(2)] = names
    return out


def c_layout(fields):
    """{c struct: (sizeof, [(name, offset, size)])} by compiling a dump of the header with g++."""
    import subprocess, tempfile
    lines = ['#include <cstdio>', '#include <cstddef>', '#include "rbd_hip.h"', "int main() {"]
    for cs, names in fields.items():
        lines.append(f'  std::printf("S {cs} %zu\\n", sizeof({cs}));')
        for n in names:
            lines.append(f'  std::printf("F {cs} {n} %zu %zu\\n", offsetof({cs}, {n}), sizeof((({cs}*)0)->{n}));')
    lines += ["  return 0;", "}"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "dump.cpp"), os.path.join(d, "dump")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        txt = subprocess.check_output([exe], text=True)
    out = {}
    for ln in txt.splitlines():
        p = ln.split()
        if p[0] == "S":
            out[p[1]] = (int(p[2]), [])
        else:
            out[p[1]][1].append((p[2], int(p[3]), int(p[4])))
    return out


def jl_layout(name):
    """(sizeof, [(field, offset, size)]) of an isbits Julia struct by the C layout rules Julia follows."""
    src = open(os.path.join(ROOT, "julia", "RigidBodyDynamicsGPU.jl")).read()
    m = re.search(r"^struct %s\b[^\n]*\n(.*?)^end" % name, src, flags=re.S | re.M)
    if not m:
        raise ValueError(f"struct {name} not found in the Julia shim")
    body = re.sub(r"#[^\n]*", "", m.group(1))
    off, fields, maxal = 0, [], 1
    for item in re.split(r"[;\n]", body):
        item = item.strip()
        if not item:
            continue
        fname, ftype = [x.strip() for x in item.split("::")]
        if ftype in ("Int32", "Cint", "UInt32"):
            size, al = 4, 4
        elif ftype in ("Float64", "Cdouble", "Int64") or ftype.startswith("Ptr{"):
            size, al = 8, 8
        else:
            t = re.match(r"NTuple\{\s*(\d+)\s*,\s*(Float64|Int32)\s*\}", ftype)
            if not t:
                raise ValueError(f"{name}.{fname}: unclassified Julia field type {ftype!r}")
            al = 8 if t.group(2) == "Float64" else 4
            size = int(t.group(1)) * al
        off = (off + al - 1) // al * al
        fields.append((fname, off, size))
        off += size
        maxal = max(maxal, al)
    return (off + maxal - 1) // maxal * maxal, fields


def check_structs():
    hf = header_struct_fields()
    cl = c_layout({cs: hf[cs] for cs in STRUCTS.values()})
    bad = []
    for js, cs in STRUCTS.items():
        jsize, jf = jl_layout(js)
        csize, cf = cl[cs]
        if jsize != csize:
            bad.append(f"{js}: sizeof {jsize} in Julia, {csize} in C ({cs})")
        if len(jf) != len(cf):
            bad.append(f"{js}: {len(jf)} fields in Julia, {len(cf)} in C ({cs})")
        for (jn, jo, jz), (cn, co, cz) in zip(jf, cf):
            if (jo, jz) != (co, cz) or jn != cn:
                bad.append(f"{js}.{jn}: offset {jo} size {jz} in Julia; {cs}.{cn}: offset {co} size {cz} in C")
    return bad, sum(len(v[1]) for v in cl.values())


def main():
    protos = header_prototypes()
    bad = []
    n = 0
    for name, lib, kinds, line in julia_ccalls():
        if lib != "librbd_hip":
            continue  # hipMalloc & co. from libamdhip64
        n += 1
        if name not in protos:
            bad.append(f"line {line}: {name} is not declared in include/rbd_hip.h")
        elif kinds != protos[name]:
            bad.append(f"line {line}: {name} is bound as {kinds}, the header declares {protos[name]}")
    print(f"{n} ccalls into librbd_hip checked against {len(protos)} prototypes; {len(bad)} mismatches")
    sbad, nf = check_structs()
    print(f"{len(STRUCTS)} structs ({nf} fields) checked against offsetof / sizeof of the header; {len(sbad)} mismatches")
    for b in bad + sbad:
        print("  " + b)
    return 1 if bad or sbad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
