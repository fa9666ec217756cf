"""Binding drift check.  The reference's seam is Julia multiple dispatch (src/multigrid.jl:7-13); the drop-in
boundary is the C ABI of include/hmg.h, bound twice: by `ccall` in julia/HomogenizationHIP.jl (never executed -- no Julia in the
build image) and by ctypes in homogenization.jl_amd/_lib.py (executed by every GPU test).  Without a Julia toolchain the only
check available is textual: every prototype of the header against the `(:hmg_x, LIB), Ret, (Args...)` tuples of the Julia file
and against `_lib.SIGNATURES` -- name, arity, and the type class of the return value and of every argument."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_prototypes():
    txt = open(os.path.join(ROOT, "include", "hmg.h")).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    txt = re.sub(r"^\s*#[^\n]*", " ", txt, flags=re.M)                                     # preprocessor lines
    txt = re.sub(r'extern\s+"C"\s*\{', " ", txt)
    txt = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", " ", txt)                            # opaque handle typedefs
    txt = re.sub(r"typedef[^;]*\(\s*\*\s*\w+\s*\)\s*\([^;]*;", " ", txt, flags=re.S)      # callback typedefs
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(hmg_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        if "typedef" in ret:
            continue
        # split arguments at top-level commas (function-pointer arguments contain commas in their own parentheses)
        parts, depth, cur = [], 0, ""
        for ch in args:
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip() and cur.strip() != "void":
            parts.append(cur)
        protos[name] = (c_class(ret, True), [c_class(a, False) for a in parts])
    return protos


def c_class(decl, is_ret):
    d = " ".join(decl.replace("\n", " ").split())
    if "(" in d:                                   # function pointer argument: int (*end)(void *user)
        return "ptr"
    if not is_ret:
        d = re.sub(r"\b\w+$", "", d).strip() if re.search(r"[\w\*]\s+\w+$|\*\s*\w+$", d) else d      # drop the argument name
    d = d.replace("const ", "").strip()
    stars = d.count("*")
    base = d.replace("*", "").strip()
    if base.endswith("_fn") and stars == 0:
        return "ptr"
    if stars == 0:
        return {"int": "int", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "void": "void"}[base]
    if stars >= 2:
        return "ptr_ptr"
    return {"char": "cstr", "double": "ptr_f64", "int64_t": "ptr_i64", "int32_t": "ptr_i32", "int": "ptr_int",
            "void": "ptr", "hmg_ctx": "ptr", "hmg_grid": "ptr", "hmg_vec": "ptr"}[base]


JL = {"Cint": "int", "Int64": "i64", "UInt64": "u64", "Float64": "f64", "Cdouble": "f64", "Cstring": "cstr", "Cvoid": "void",
      "Ptr{Cvoid}": "ptr", "Ptr{UInt8}": "ptr", "Ptr{Float64}": "ptr_f64", "Ref{Float64}": "ptr_f64", "Ptr{Int64}": "ptr_i64",
      "Ref{Int64}": "ptr_i64", "Ptr{Int32}": "ptr_i32", "Ptr{Cint}": "ptr_int", "Ref{Cint}": "ptr_int",
      "Ref{Ptr{Cvoid}}": "ptr_ptr", "Ptr{Ptr{Cvoid}}": "ptr_ptr"}


def julia_bindings():
    txt = open(os.path.join(ROOT, "julia", "HomogenizationHIP.jl")).read()
    txt = re.sub(r"#[^\n]*", " ", txt)
    out = {}
    for m in re.finditer(r"ccall\(\(:(hmg_\w+),\s*LIB\),\s*([\w{}]+),\s*\(([^()]*)\)", txt, flags=re.S):
        name, ret, args = m.group(1), m.group(2), m.group(3)
        toks = [a.strip() for a in re.split(r",(?![^{]*})", args) if a.strip()]
        sig = (JL[ret], [JL[t] for t in toks])
        out.setdefault(name, []).append(sig)
    return out


def ctypes_bindings():
    import ctypes
    from importlib import import_module
    L = import_module("homogenization_jl_amd._lib")
    cls = {L.c_int: "int", L.c_i64: "i64", ctypes.c_uint64: "u64", L.c_f64: "f64", ctypes.c_char_p: "cstr", L.vp: "ptr",
           L.p_f64: "ptr_f64", L.p_i64: "ptr_i64", L.p_i32: "ptr_i32", L.pp: "ptr_ptr", None: "void",
           L.EXCHANGE_FN: "ptr", L.EXCHANGE_END_FN: "ptr", L.P2P_FN: "ptr"}
    return {n: (cls[r], [cls[a] for a in args]) for n, (r, args) in L.SIGNATURES.items()}


def compatible(c, other):
    """Same class, or the loose pairs a byte buffer / opaque pointer allows: `void *` <-> a typed byte pointer or C string."""
    return c == other or {c, other} <= {"ptr", "cstr"} or {c, other} <= {"ptr", "ptr_int"} or {c, other} <= {"ptr", "ptr_f64"} or \
        {c, other} <= {"ptr", "ptr_i64"}


def test_header_is_parsed_completely():
    protos = header_prototypes()
    # every exported hmg_* symbol named in the header text is found as a prototype
    names = set(re.findall(r"\b(hmg_\w+)\s*\(", re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "hmg.h")).read(), flags=re.S)))
    assert names - set(protos) == set(), sorted(names - set(protos))
    assert len(protos) >= 83


def test_ctypes_signatures_match_the_header():
    protos, py = header_prototypes(), ctypes_bindings()
    assert set(py) == set(protos), (sorted(set(protos) - set(py)), sorted(set(py) - set(protos)))
    for name, (ret, args) in protos.items():
        pret, pargs = py[name]
        assert compatible(ret, pret), (name, "return", ret, pret)
        assert len(args) == len(pargs), (name, "arity", args, pargs)
        for i, (a, b) in enumerate(zip(args, pargs)):
            assert compatible(a, b), (name, i, a, b)


def test_julia_ccalls_match_the_header_and_cover_it():
    protos, jl = header_prototypes(), julia_bindings()
    assert set(jl) - set(protos) == set(), sorted(set(jl) - set(protos))
    missing = sorted(set(protos) - set(jl))
    assert missing == [], f"entry points of include/hmg.h without a ccall in julia/HomogenizationHIP.jl: {missing}"
    for name, sigs in jl.items():
        ret, args = protos[name]
        for jret, jargs in sigs:
            assert compatible(ret, jret), (name, "return", ret, jret)
            assert len(args) == len(jargs), (name, "arity", args, jargs)
            for i, (a, b) in enumerate(zip(args, jargs)):
                assert compatible(a, b), (name, i, a, b)
