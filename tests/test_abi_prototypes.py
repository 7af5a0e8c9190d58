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
    return {"int": "int", "size_t": "size_t", "uint64_t": "u64", "uint32_t": "u32", "double": "double", "int64_t": "i64"}[t]


CT = {"int": C.c_int, "size_t": C.c_size_t, "u64": C.c_uint64, "u32": C.c_uint32, "double": C.c_double, "void_p": C.c_void_p, "char_p": C.c_char_p, "i64": C.c_int64}


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


# ---- the Rust binding's extern "C" block against the same header ---------------------------------------------
# Nothing compiles rust/src/mi355x.rs here, so a parameter dropped, swapped or mistyped in its declarations would first be seen by a
# maintainer's segfault.  Types AND names are compared: two neighbouring `*const u8` that changed places have the same types.
RUST_KIND = {"c_int": "int", "usize": "size_t", "u64": "u64", "u32": "u32", "f64": "double"}


def rust_kind(t: str) -> str:
    t = t.strip()
    if t.startswith("*mut *mut ") or t.startswith("*mut *const "):
        return "ptrptr"
    m = re.match(r"\*(const|mut) (\w+)$", t)
    if m:
        inner = m.group(2)
        if inner == "c_char":
            return "char_p"
        if inner in ("c_int", "u64", "f64", "usize"):
            return "ptr:" + {"c_int": "int", "u64": "uint64_t", "f64": "double", "usize": "size_t"}[inner]
        return "void_p"                   # u8 buffers, opaque handles (ActNode, ActNodeNullifierSet)
    return RUST_KIND[t]


def parse_rust_externs():
    src = open(os.path.join(ROOT, "rust", "src", "mi355x.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    block = re.sub(r"//[^\n]*", "", block)
    fns = {}
    for m in re.finditer(r"fn (act_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        params = [p.strip() for p in m.group(2).split(",") if p.strip()]
        fns[m.group(1)] = ([tuple(x.strip() for x in p.split(":", 1)) for p in params], (m.group(3) or "").strip())
    return fns


def c_param_name(p: str) -> str:
    return re.sub(r"\[[^\]]*\]", "", p).strip().split()[-1].lstrip("*")


def test_rust_extern_block_matches_the_header():
    protos = parse_header()
    fns = parse_rust_externs()
    assert len(fns) >= 22, sorted(fns)
    # everything a maintainer needs for the hot path, the wire path and the redemption step is declared
    for need in ("act_node_create", "act_node_refund_cbor_batch", "act_node_redeem_cbor_batch", "act_node_redeem_batch", "act_node_verify_spend_cbor_batch",
                 "act_node_nullifier_set_create", "act_node_verify_spend_batch", "act_node_refund_sign_batch", "act_node_prove_spend_batch"):
        assert need in fns, need
    for name, (params, ret) in fns.items():
        assert name in protos, name + " is not in include/act_mi355x.h"
        c_ret, c_params = protos[name]
        assert len(params) == len(c_params), (name, params, c_params)
        for (rname, rtype), cp in zip(params, c_params):
            assert rust_kind(rtype) == kind(cp), (name, rname, rtype, cp)
            assert rname.lower() == c_param_name(cp).lower(), (name, rname, cp)
            # const-ness of buffers: an output declared *const (or an input *mut) is a binding bug waiting for an optimiser
            if rtype.startswith("*") and "char" not in rtype and not rtype.startswith("*mut *mut"):
                c_const = cp.strip().startswith("const ")
                handle = re.match(r"\*(const|mut) Act", rtype) is not None
                if not handle:
                    assert rtype.startswith("*const") == c_const, (name, rname, rtype, cp)
        want_ret = {"int": "c_int", "void": "", "size_t": "usize", "const char *": "*const c_char"}[c_ret]
        assert ret == want_ret, (name, ret, c_ret)
    # the callback struct: two pointer-sized fields in the header's order
    rs = open(os.path.join(ROOT, "rust", "src", "mi355x.rs")).read()
    hd = open(os.path.join(ROOT, "include", "act_mi355x.h")).read()
    assert re.search(r"typedef struct act_rng_source \{ act_rng_draw_fn draw; void \*rng_ctx; \} act_rng_source;", hd)
    assert "typedef int (*act_rng_draw_fn)(void *rng_ctx, uint8_t *dst, size_t len);" in hd      # 0 = drawn; anything else fails the call (ACT_ERR_RNG)
    assert re.search(r"pub struct ActRngSource \{\s*draw: unsafe extern \"C\" fn\(rng_ctx: \*mut c_void, dst: \*mut u8, len: usize\) -> c_int,\s*rng_ctx: \*mut c_void,\s*\}", rs)
    assert "const ACT_RNG_CALLBACK: c_int = 2;" in rs and "#define ACT_RNG_CALLBACK 2" in hd
    assert "const ACT_RNG_SEQUENTIAL: c_int = 1;" in rs and "#define ACT_RNG_SEQUENTIAL 1" in hd
