"""The two builds of the library must produce the same bytes.  libact_mi355x.so (the default) is address-free for every secret,
the client's included (-DACT_CT_SECRET_TABLES: matrix-core table look-ups, register-only chains) -- the reference's constant-time
posture; libact_mi355x_fast.so lets the CLIENT's secrets (prover, request) address 16- / 24-bit tables and Pippenger buckets.  The
whole GPU suite runs against the default; the libsodium-made lifecycles, the oracle-checked random batches and the hygiene test are
re-run here against the fast build in a child process."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_parity_suite_under_the_fast_build():
    from act_amd import capi
    fast = os.path.join(os.path.dirname(capi.LIB_PATH), "libact_mi355x_fast.so")
    assert os.path.exists(fast), "build it: make -C anonymous-credit-tokens_amd/csrc fast"
    env = dict(os.environ, ACT_LIB_PATH=fast)
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from act_amd import capi; print(capi.load().act_build_has_ct_secret_tables())" % ROOT],
                           capture_output=True, text=True, env=env)
    assert probe.stdout.strip() == "0", probe.stderr[-500:]
    if not os.environ.get("ACT_LIB_PATH"):
        assert capi.load().act_build_has_ct_secret_tables() == 1          # the default build: no secret ever selects an address
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_sodium.py", "tests/test_gpu_hygiene.py",
                        "tests/test_gpu_parity.py::test_golden_lifecycle", "tests/test_gpu_parity.py::test_random_batches_against_oracle",
                        "tests/test_gpu_parity.py::test_seeded_prover_equals_the_prover_on_the_expanded_bytes",
                        "tests/test_gpu_node.py::test_node_equals_single_context"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
