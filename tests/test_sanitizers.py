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


def _host_sources():
    csrc = os.path.join(ROOT, "anonymous-credit-tokens_amd", "csrc")
    return [os.path.join(csrc, "host_pool.cpp"), os.path.join(csrc, "host_hash.cpp"), os.path.join(csrc, "node.cpp"),
            os.path.join(ROOT, "tests", "node_mock", "node_mock.cpp"), os.path.join(ROOT, "tests", "tsan", "tsan_host.cpp")]


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_host_side_concurrency_under_sanitizers(tmp_path, san):
    """The library's host-side threads -- the process-wide worker pool (hashing, parallel-for), the node-level nullifier set's
    routing on it, a node handle used by several threads with small calls bypassing its lock -- under ThreadSanitizer, and again
    under AddressSanitizer + UBSan (tests/tsan/tsan_host.cpp; the single-GPU entry points are the mock's)."""
    lib = subprocess.run(["gcc", "-print-file-name=lib%s.so" % ("tsan" if san == "thread" else "asan")], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(lib) or not os.path.exists(lib):
        pytest.skip("no %s runtime in this toolchain" % san)
    exe = str(tmp_path / "tsan_host")
    # host_hash.cpp (pure computation on its arguments, no shared state) is compiled WITHOUT the sanitizer: its target_clones resolvers run
    # before the ThreadSanitizer runtime exists and an instrumented resolver segfaults at load
    hh = str(tmp_path / "host_hash.o")
    srcs = _host_sources()
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wno-psabi", "-c", "-o", hh, srcs[1]], check=True)
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-Wno-unknown-pragmas", "-Wno-psabi",
                    "-DACT_MOCK_NO_PARALLEL_FOR", "-o", exe, hh] + [x for i, x in enumerate(srcs) if i != 1], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("the ThreadSanitizer runtime cannot map its shadow memory on this kernel (address-space layout)")
    assert r.returncode == 0 and "TSAN DRIVER DONE" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-6000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
