"""Checks every `ccall` of julia/RigidBodyDynamicsGPU.jl against include/rbd_hip.h: the symbol exists, the argument count matches the
prototype and every argument is of the same kind (pointer / 32-bit integer / 64-bit integer / double).  Julia is not available in the
build image, so this is what keeps the shim's bindings from drifting off the header.  Exit status 0 = consistent; run by
tests/test_capi_symbols.py."""
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
    for m in re.finditer(r"\b(?:int|const char\*)\s+(rbd_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
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
    for b in bad:
        print("  " + b)
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
