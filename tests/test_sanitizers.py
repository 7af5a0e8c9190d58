"""AddressSanitizer + UndefinedBehaviorSanitizer over everything of ours that compiles for the CPU: the C oracle and the
host build of the device arithmetic headers INCLUDING the spend-verification kernels' lane bodies (csrc/spend_lanes.h).
GPU sanitizers are not available on the pool, so this is where out-of-bounds indexing, misaligned or overlapping
accesses, shifts out of range and signed overflow in the kernel code get caught.  Never run on the GPU box's device."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def test_oracle_and_kernel_lane_bodies_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    o_so, h_so = str(tmp_path / "libact_oracle_asan.so"), str(tmp_path / "libhostcheck_asan.so")
    subprocess.run(["gcc", "-std=c11", "-fPIC", "-shared", "-pthread", "-Wall", *FLAGS, "-o", o_so, os.path.join(ROOT, "oracle", "act_oracle.c")], check=True)
    subprocess.run(["g++", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", *FLAGS, "-o", h_so, os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")], check=True)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize_driver.py"), o_so, h_so], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0 and "SANITIZERS CLEAN" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
