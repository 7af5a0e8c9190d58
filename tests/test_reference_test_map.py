"""tests/REFERENCE_TESTS.md says where each of the reference's own tests is restated.  Kept honest here: every test it points to exists
in the file it names, and -- where the reference is at hand (the authoring container; not the GPU box) -- every test function of
src/tests.rs appears in it with its line."""
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DOC = open(os.path.join(HERE, "REFERENCE_TESTS.md")).read()
REF = "/root/reference/src/tests.rs"


def test_every_restatement_named_in_the_map_exists():
    pointed = re.findall(r"`(test_[a-z_0-9]+\.py)(?:::(test_[a-z_0-9]+))?`", DOC)
    assert len(pointed) > 30
    for fname, func in pointed:
        path = os.path.join(HERE, fname)
        assert os.path.exists(path), fname
        if func:
            assert re.search(r"^def %s\(" % func, open(path).read(), re.M), (fname, func)


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference is not on this machine")
def test_every_reference_test_is_in_the_map():
    helpers = {"fast_config", "new", "is_spent", "record_spent", "test_params"}
    rows = [(re.findall(r"`([a-z_0-9*]+)`", a), [int(x) for x in re.findall(r"\d+", b)]) for a, b in re.findall(r"^\| ([^|]+) \| ([0-9 ,–-]+) \|", DOC, re.M)]
    for i, line in enumerate(open(REF).read().split("\n"), 1):
        m = re.match(r"\s*fn ([a-z_0-9]+)\(", line)
        if not m or m.group(1) in helpers or m.group(1).endswith("_strategy"):
            continue
        name = m.group(1)
        hit = [ls for ns, ls in rows if name in ns or (name.startswith("prop_cbor_round_trip_") and "prop_cbor_round_trip_*" in ns)]
        assert hit, name
        if not name.startswith("prop_cbor_round_trip_"):
            assert any(abs(i - l) <= 2 for l in hit[0]), (name, i, hit[0])
