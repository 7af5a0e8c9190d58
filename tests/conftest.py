import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

ELL = 2**252 + 27742317777372353535851937790883648493


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def shake(label: str, n: int) -> bytes:
    return hashlib.shake_256(label.encode()).digest(n)


def scb(v: int) -> bytes:
    return (v % ELL).to_bytes(32, "little")


def load_golden(name: str):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (checker only)."""
    from oracle_c import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def hostcheck():
    """Host build of the device arithmetic headers (test-only; see tests/hostcheck/hostcheck.cpp)."""
    import ctypes
    src = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")
    out = os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so")
    csrc = os.path.join(ROOT, "anonymous-credit-tokens_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".inc"))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", out, src], check=True)
    return ctypes.CDLL(out)


@pytest.fixture(scope="session")
def bench_params(oracle):
    return oracle.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")


_engines = {}


@pytest.fixture(scope="session")
def engine_factory():
    """HIP engines through the C ABI (GPU tests only)."""
    from act_amd import capi

    def make(h: bytes, L: int = 128, max_batch: int = 4096, transcript=None):
        # engines are cached for the session: keep their workspaces small (the library default, 65536 records per
        # launch = 29 GB at L = 128, is exercised by bench.py and test_device_memory_path_and_full_size_properties)
        key = (h, L, max_batch)
        if key not in _engines:
            _engines[key] = capi.Engine(h, L, max_batch=max_batch)
        e = _engines[key]
        capi.forward_tuning_env()          # the ACT_* measurement knobs as the test has set them (the library reads no environment itself)
        e.set_transcript_mode(capi.TRANSCRIPT_HOST if transcript is None else transcript)
        return e
    return make
