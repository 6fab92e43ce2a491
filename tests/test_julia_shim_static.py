"""Static consistency of julia/SthenoMI355X.jl with include/sthenomi.h (through the ctypes mirror the symbol tests
pin to the header).  Julia is not installed in the build image, so the shim has never been executed; what CAN be
checked without it: every `ccall` names an exported entry point, passes the declared number of arguments with
compatible C types and return type, and the three boundary structs are declared field by field as in the header."""
import ctypes as C
import os
import re

import stheno_jl_amd as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "julia", "SthenoMI355X.jl"), encoding="utf-8").read()


def _matching(text, start):
    """index just past the parenthesis that closes the one opening at text[start]"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise AssertionError("unbalanced parentheses")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _ccalls():
    for m in re.finditer(r"ccall\(", SRC):
        end = _matching(SRC, m.end() - 1)
        parts = _split_top(SRC[m.end():end - 1])
        name = re.match(r"\(:(\w+),\s*LIB\)", parts[0]).group(1)
        ret = parts[1]
        argt = parts[2]
        assert argt.startswith("(") and argt.endswith(")"), (name, argt)
        types = _split_top(argt[1:-1])
        yield name, ret, types, len(parts) - 3


def _julia_kind(t):
    t = t.strip()
    if t in ("Cint", "Int32"):
        return "i32"
    if t in ("Int64", "Clonglong"):
        return "i64"
    if t in ("Cdouble", "Float64"):
        return "f64"
    if t == "Cstring":
        return "str"
    if t.startswith("Ptr{") or t.startswith("Ref{"):
        return "ptr"
    raise AssertionError(f"unknown Julia C type {t}")


def _ctypes_kind(t):
    if t is C.c_int or t is C.c_int32:
        return "i32"
    if t is C.c_int64 or t is C.c_longlong:
        return "i64"
    if t is C.c_double:
        return "f64"
    if t is C.c_char_p:
        return "str"
    if t is C.c_void_p or (isinstance(t, type) and issubclass(t, C._Pointer)):
        return "ptr"
    raise AssertionError(f"unknown ctypes type {t}")


def test_every_ccall_matches_the_declared_signature():
    sigs = P.lib._SIGS
    seen = set()
    for name, ret, types, nvalues in _ccalls():
        assert name in sigs, f"{name} is not declared in include/sthenomi.h"
        res, args = sigs[name]
        assert len(types) == len(args), (name, len(types), len(args))
        assert nvalues == len(types), (name, "values passed", nvalues, "types declared", len(types))
        assert _julia_kind(ret) == _ctypes_kind(res), (name, ret, res)
        for k, (jt, ct) in enumerate(zip(types, args)):
            assert _julia_kind(jt) == _ctypes_kind(ct), (name, k, jt, ct)
        seen.add(name)
    # the operator surface the shim promises: every one of these is bound
    for name in ("sgp_ctx_create", "sgp_ctx_create_multi", "sgp_logpdf", "sgp_logpdf_f32", "sgp_rand", "sgp_posterior_create",
                 "sgp_posterior_predict", "sgp_posterior_destroy", "sgp_elbo", "sgp_sparse_posterior_create",
                 "sgp_sparse_posterior_predict", "sgp_sparse_posterior_destroy", "sgp_kernelmatrix", "sgp_kernelmatrix_diag",
                 "sgp_logpdf_grad", "sgp_logpdf_grad_x", "sgp_logpdf_grad_xs", "sgp_elbo_grad", "sgp_kernelmatrix_diag_grad"):
        assert name in seen, f"the shim does not bind {name}"


def _julia_struct(name):
    m = re.search(r"struct " + name + r"\b(.*?)\bend", SRC, re.S)
    assert m, name
    fields = re.findall(r"(\w+)::([\w{}]+)", m.group(1))
    return fields


def test_boundary_structs_are_declared_like_the_header():
    for jname, mirror in (("CInput", P.lib.sgp_input), ("CTerm", P.lib.sgp_term), ("CSpec", P.lib.sgp_cov_spec)):
        fields = _julia_struct(jname)
        assert [f for f, _ in fields] == [f for f, _ in mirror._fields_], (jname, fields)
        for (fname, jt), (_, ct) in zip(fields, mirror._fields_):
            assert _julia_kind(jt) == _ctypes_kind(ct), (jname, fname, jt, ct)
