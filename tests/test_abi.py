"""The C-ABI library: builds for gfx950 in this container, loads, exports every symbol include/*.h declares,
and fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from act_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "act_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(act_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from act_amd import capi
    syms = declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == syms        # the ctypes binding covers the whole header


def test_no_cpu_fallback_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from act_amd import capi
    ctx = C.c_void_p()
    h = (C.c_uint8 * 96)()
    assert lib.act_ctx_create(h, 128, 0, 0, C.byref(ctx)) == 4          # ACT_ERR_NO_DEVICE
    with pytest.raises(capi.ActError):
        capi.params_new("a", "b", "c", "d")
    with pytest.raises(capi.ActError):
        capi.Engine(bytes(96))


def test_product_does_not_link_the_oracle(lib):
    from act_amd import capi
    import subprocess
    out = subprocess.run(["nm", "-D", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle_" not in out
    for f in os.listdir(os.path.join(ROOT, "anonymous-credit-tokens_amd", "csrc")):
        if f.endswith((".hip", ".h", ".cpp")):
            assert "oracle" not in open(os.path.join(ROOT, "anonymous-credit-tokens_amd", "csrc", f)).read().lower().replace("oracle-independent", ""), f
    for f in ("capi.py", "api.py", "__init__.py"):
        assert "oracle" not in open(os.path.join(ROOT, "anonymous-credit-tokens_amd", f)).read().lower(), f


def test_simd_host_blake3_matches_upstream_vectors(lib):
    """The 16-lane host hasher of the host-transcript mode (csrc/host_hash.cpp) against the LLVM-BLAKE3 fixtures:
    sixteen copies of each message, each lane must give the upstream XOF bytes.  Hashing only; no GPU involved."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "blake3_llvm.json")))
    lib.act_host_b3_xof64_x16.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    for v in g["vectors"]:
        n = v["len"]
        if n > 200000:
            continue
        stride = (n + 64 + 15) & ~15
        buf = bytearray(stride * 16)
        for i in range(16):
            buf[i * stride:i * stride + n] = bytes((j + 0) % 251 for j in range(n))
        if n:                                  # make lane 5 differ to catch lane mix-ups
            buf[5 * stride] ^= 0xFF
        arr = (C.c_uint8 * len(buf)).from_buffer(buf)
        out = (C.c_uint32 * 256)()
        lib.act_host_b3_xof64_x16(arr, stride, n, out)
        raw = bytes(out)
        for i in range(16):
            if i == 5 and n:
                assert raw[64 * i:64 * i + 64].hex() != v["xof"][:128]
            else:
                assert raw[64 * i:64 * i + 64].hex() == v["xof"][:128], (n, i)
