"""CPU suite: the Julia binding cannot run here (no Julia toolchain in the image), so every `ccall` in
julia/BlockSparseMatricesROCm.jl is checked STATICALLY against the prototype it binds in include/bsm_rocm.h: the symbol
exists, the arity matches, every argument has the same integer width / pointer-ness, and so does the return type.  One
`Int64` where the header says `int32_t` is silent stack corruption on a user's machine.

Interface this binding stands in for: LinearMaps._unsafe_mul! methods of the reference (src/abstractblockmatrix.jl:27-34,
src/vbcrs.jl:78-80 and the other constructors)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "BlockSparseMatricesROCm.jl")
HDR = os.path.join(ROOT, "include", "bsm_rocm.h")

HANDLE_TYPEDEFS = ("bsm_matrix_t", "bsm_ctx_t")  # typedef struct ... *name


def c_class(t):
    t = re.sub(r"/\*.*?\*/", "", t).strip()
    t = re.sub(r"\b(const|restrict|volatile|struct)\b", "", t).strip()
    if "*" in t or "[" in t or any(h in t.split() for h in HANDLE_TYPEDEFS):
        return "ptr"
    t = " ".join(t.split())
    return {"int": "i32", "int32_t": "i32", "unsigned": "u32", "uint32_t": "u32", "int64_t": "i64", "long long": "i64",
            "uint64_t": "u64", "size_t": "u64", "void": "void", "double": "f64", "float": "f32"}[t]


def header_prototypes(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?m)^([A-Za-z_][\w \*]*?)\b(bsm_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        if ret.startswith(("typedef", "#")):
            continue
        params = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier) unless the declarator is a bare type
                mm = re.match(r"^(.*?)(\b[A-Za-z_]\w*)\s*(\[\s*\d*\s*\])?$", a)
                typ = (mm.group(1) + (" *" if mm.group(3) else "")) if mm and mm.group(1).strip() else a
                params.append(c_class(typ))
        protos[name] = (c_class(ret), params)
    return protos


def jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr"):
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Cuint": "u32", "UInt32": "u32", "Int64": "i64", "Clonglong": "i64", "Clong": "i64",
            "UInt64": "u64", "Csize_t": "u64", "Cvoid": "void", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32"}[t]


def _balanced(text, i, open_, close):
    assert text[i] == open_
    depth = 0
    for j in range(i, len(text)):
        if text[j] == open_:
            depth += 1
        elif text[j] == close:
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [p.strip() for p in out if p.strip()]


def julia_ccalls(text):
    text = re.sub(r"(?m)#.*$", "", text)
    calls = []
    for m in re.finditer(r"ccall\(\(\s*:(bsm_\w+)\s*,\s*libbsm\s*\)\s*,", text):
        i = m.end()
        j = text.index(",", i)
        ret = text[i:j].strip()
        k = j + 1
        while text[k].isspace():
            k += 1
        assert text[k] == "(", (m.group(1), text[k:k + 40])
        e = _balanced(text, k, "(", ")")
        args = _split_top(text[k + 1:e])
        line = text.count("\n", 0, m.start()) + 1
        calls.append((m.group(1), line, jl_class(ret), [jl_class(a) for a in args]))
    return calls


def mismatches(jl_text, hdr_text):
    protos = header_prototypes(hdr_text)
    bad = []
    for name, line, ret, args in julia_ccalls(jl_text):
        if name not in protos:
            bad.append(f"{name} (line {line}): not declared in include/bsm_rocm.h")
            continue
        cret, cargs = protos[name]
        if ret != cret:
            bad.append(f"{name} (line {line}): returns {ret}, header says {cret}")
        if len(args) != len(cargs):
            bad.append(f"{name} (line {line}): {len(args)} arguments, header has {len(cargs)}")
            continue
        for n, (a, c) in enumerate(zip(args, cargs)):
            if a != c:
                bad.append(f"{name} (line {line}): argument {n + 1} is {a}, header says {c}")
    return bad


def test_every_ccall_matches_its_prototype():
    jl, hdr = open(JL).read(), open(HDR).read()
    calls = julia_ccalls(jl)
    names = {c[0] for c in calls}
    assert len(calls) >= 17 and {"bsm_mul", "bsm_mul_multi", "bsm_mul_parts", "bsm_vbcrs_create", "bsm_symmetric_create",
                                 "bsm_blocksparse_create", "bsm_ctx_create", "bsm_part_info", "bsm_get_bookkeeping"} <= names
    assert mismatches(jl, hdr) == []


def test_the_checker_sees_a_wrong_width_a_wrong_arity_and_a_missing_symbol():
    jl, hdr = open(JL).read(), open(HDR).read()
    # one integer width edited in one ccall
    broken = jl.replace("(:bsm_host_register, libbsm), Cint, (Ptr{Cvoid}, Int64)", "(:bsm_host_register, libbsm), Cint, (Ptr{Cvoid}, Cint)")
    assert broken != jl and any("bsm_host_register" in b and "argument 2" in b for b in mismatches(broken, hdr))
    # a pointer passed where the header takes an integer
    broken = jl.replace("(:bsm_part_info, libbsm), Cint, (Ptr{Cvoid}, Int32, Ref{BsmPartInfo})", "(:bsm_part_info, libbsm), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ref{BsmPartInfo})")
    assert broken != jl and any("bsm_part_info" in b for b in mismatches(broken, hdr))
    # an argument dropped
    broken = jl.replace("(:bsm_destroy, libbsm), Cint, (Ptr{Cvoid},)", "(:bsm_destroy, libbsm), Cint, ()")
    assert broken != jl and any("bsm_destroy" in b and "arguments" in b for b in mismatches(broken, hdr))
    # a symbol the header does not have
    broken = jl.replace(":bsm_options_default", ":bsm_options_defaults")
    assert any("not declared" in b for b in mismatches(broken, hdr))
    # and the header parser itself: every exported prototype is seen with its true shape
    protos = header_prototypes(hdr)
    assert protos["bsm_mul"] == ("i32", ["ptr", "i32", "ptr", "ptr", "ptr", "ptr", "i32", "i32", "ptr"])
    assert protos["bsm_host_register"] == ("i32", ["ptr", "i64"])
    assert protos["bsm_last_error"] == ("ptr", [])
    assert protos["bsm_options_default"] == ("void", ["ptr"])


def test_the_ctypes_mirror_matches_the_header_too():
    """the same check for blocksparsematrices.jl_amd/_lib.py (the binding the parity tests call through)"""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    from bsm_amd import _lib
    L = _lib.lib()
    protos = header_prototypes(open(HDR).read())

    def cls(t):
        if t is None:
            return "void"
        if t in (C.c_int, C.c_int32):
            return "i32"
        if t in (C.c_int64, C.c_longlong):
            return "i64"
        if t in (C.c_uint64, C.c_size_t):
            return "u64"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or issubclass(t, C._Pointer):
            return "ptr"
        raise AssertionError(t)
    seen = 0
    for name, (ret, args) in protos.items():
        fn = getattr(L, name)
        if fn.argtypes is None:
            continue
        seen += 1
        assert [cls(t) for t in fn.argtypes] == args, name
        assert cls(fn.restype) == ret, name
    assert seen >= 20
