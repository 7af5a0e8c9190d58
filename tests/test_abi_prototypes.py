"""The ctypes binding (capi.py) against the C header it claims to bind: every prototype of include/act_mi355x.h is parsed
and its parameter list compared, type by type, with the argtypes / restype the binding declares -- an argument added,
dropped, reordered or widened in one place and not the other fails here, without a GPU.  (tests/abi_conformance.cpp is the
complementary check: a C++ caller compiled against the header and linked to the library, run on the GPU.)"""
import ctypes as C
import os
import re

from conftest import ROOT


def parse_header():
    src = open(os.path.join(ROOT, "include", "act_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|void|size_t|const char \*|act_ctx \*)\s*(act_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        ps = [p.strip() for p in params.split(",")] if params.strip() and params.strip() != "void" else []
        protos[name] = (ret, ps)
    return protos


def kind(param: str) -> str:
    """C parameter -> the ctypes class the binding must use."""
    p = re.sub(r"\s+", " ", param)
    if "*" in p or "[" in p:
        if re.match(r"(const )?char \*", p):
            return "char_p"
        if "**" in p:
            return "ptrptr"
        if re.match(r"(const )?(double|uint64_t|size_t|int) \*", p):
            return "ptr:" + re.match(r"(const )?(\w+) \*", p).group(2)
        return "void_p"                      # uint8_t* / uint32_t* / opaque handles
    t = p.rsplit(" ", 1)[0].replace("const ", "")
    if t == "act_host_range_fn":
        return "fnptr"
    return {"int": "int", "size_t": "size_t", "uint64_t": "u64", "uint32_t": "u32", "double": "double"}[t]


CT = {"int": C.c_int, "size_t": C.c_size_t, "u64": C.c_uint64, "u32": C.c_uint32, "double": C.c_double, "void_p": C.c_void_p, "char_p": C.c_char_p}


def test_every_prototype_matches_the_binding():
    from act_amd import capi
    lib = capi.load()
    protos = parse_header()
    assert len(protos) >= 66, sorted(protos)
    assert set(protos) == set(capi.EXPORTS), set(protos) ^ set(capi.EXPORTS)
    for name, (ret, params) in protos.items():
        fn = getattr(lib, name)
        assert fn.argtypes is not None, name + ": no argtypes declared"
        assert len(fn.argtypes) == len(params), (name, params, fn.argtypes)
        for p, a in zip(params, fn.argtypes):
            k = kind(p)
            if k == "fnptr":
                assert issubclass(a, C._CFuncPtr), (name, p, a)
            elif k in CT:
                assert a is CT[k], (name, p, a)
            elif k == "ptrptr":
                assert a is C.POINTER(C.c_void_p), (name, p, a)
            else:                                               # typed out-pointer
                want = {"double": C.c_double, "uint64_t": C.c_uint64, "size_t": C.c_size_t, "int": C.c_int}[k[4:]]
                assert a in (C.POINTER(want), C.c_void_p), (name, p, a)
        want_ret = {"int": C.c_int, "void": None, "size_t": C.c_size_t, "const char *": C.c_char_p, "act_ctx *": C.c_void_p}[ret]
        assert fn.restype is want_ret, (name, ret, fn.restype)
