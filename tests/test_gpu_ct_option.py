"""libact_mi355x_ct.so (-DACT_CT_SECRET_TABLES: every table entry / Pippenger bucket that a digit of a SECRET scalar selects
is read and written in full and picked with masks) must produce the same bytes as the default build: the libsodium-made
lifecycles, the oracle-checked random batches and the hygiene test are re-run against it in a child process."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_parity_suite_under_ct_secret_tables():
    from act_amd import capi
    ct = os.path.join(os.path.dirname(capi.LIB_PATH), "libact_mi355x_ct.so")
    assert os.path.exists(ct), "build it: make -C anonymous-credit-tokens_amd/csrc ct"
    env = dict(os.environ, ACT_LIB_PATH=ct)
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from act_amd import capi; print(capi.load().act_build_has_ct_secret_tables())" % ROOT],
                           capture_output=True, text=True, env=env)
    assert probe.stdout.strip() == "1", probe.stderr[-500:]
    assert capi.load().act_build_has_ct_secret_tables() == 0
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_sodium.py", "tests/test_gpu_hygiene.py",
                        "tests/test_gpu_parity.py::test_golden_lifecycle", "tests/test_gpu_parity.py::test_random_batches_against_oracle",
                        "tests/test_gpu_node.py::test_node_equals_single_context"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
