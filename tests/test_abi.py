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
