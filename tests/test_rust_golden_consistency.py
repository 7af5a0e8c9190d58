"""rust/tests/golden.rs cannot be compiled here (no Rust toolchain), so what CAN be checked without one is checked: that every
fixture field, rng label and label suffix the Rust test names exists in the committed fixtures with the meaning the schema page
gives it, that its SHAKE-256 (written out in the file, Keccak round constants included) is FIPS 202's -- the constants are compared
with ones derived here from the LFSR of the standard -- and that its CBOR framing of a record is the crate's (rebuilt here from the
same head / bstr rules and compared with the Python model's to_cbor on every fixture record)."""
import hashlib
import os
import re

import pymodel as m
from conftest import ROOT, load_golden

SRC = open(os.path.join(ROOT, "rust", "tests", "golden.rs")).read()


def test_fields_and_labels_exist_in_the_fixtures():
    calls = re.findall(r'run_file\("([^"]+)", "([^"]+)", "([^"]+)", "([^"]+)"\)', SRC)
    assert len(calls) == 2
    fields = set(re.findall(r'rec\((?:case|&fx), "(\w+)"\)', SRC)) | set(re.findall(r'case\["(\w+)"\]', SRC)) | set(re.findall(r'amount\(case, "(\w+)"\)', SRC))
    assert {"pre", "request", "response", "token", "proof", "prerefund", "refund", "token2", "status", "status_other_issuer", "tamper", "c", "s", "sk", "sk_other"} <= fields
    suffixes = set(re.findall(r'rng\("(\w+)", ', SRC))
    assert suffixes == {"pre", "request", "issue", "prove", "refund"}
    for name, sk_label, sk_other_label, tag_prefix in calls:
        g = load_golden(name)
        L = g["L"]
        assert L == 128
        # the key labels: PrivateKey::random over SHAKE-256(label) gives the fixture's key (checked with the Python model)
        for label, key in ((sk_label, "sk"), (sk_other_label, "sk_other")):
            sk = m.PrivateKey.random(m.ByteRng(hashlib.shake_256(label.encode()).digest(64)))
            assert sk.record().hex() == g[key], (name, label)
        if "tag_fmt" in g:
            assert g["tag_fmt"] % 7 == tag_prefix + "7" and g["sk_label"] == sk_label
        for i, case in enumerate(g["cases"]):
            for f in fields - {"sk", "sk_other", "token2", "kprime", "challenge"}:
                assert f in case, (name, i, f)
            if case["status"] == 0:
                assert "token2" in case
            # the per-call labels reproduce the fixture: PreIssuance::random from tag + "-pre"
            pre = m.PreIssuance.random(m.ByteRng(hashlib.shake_256((tag_prefix + "%d-pre" % i).encode()).digest(128)))
            assert pre.record().hex() == case["pre"], (name, i)
            # decimal amounts below 2^128, as the Rust side parses them (u128)
            assert 0 <= int(case["c"]) < 2**128 and 0 <= int(case["s"]) < 2**128
            assert case["status"] in (0, 6, 7, 255) and case["status_other_issuer"] != 0


def _keccak_constants():
    """round constants, rotation offsets and lane order of Keccak-f[1600] from the standard's definitions (LFSR x^8 + x^6 + x^5 + x^4 + 1; (x, y) -> (y, 2x + 3y))"""
    rc, r = [], 1
    for _ in range(24):
        c = 0
        for j in range(7):
            if r & 1:
                c |= 1 << ((1 << j) - 1)
            r = ((r << 1) ^ ((r >> 7) * 0x71)) & 0xff
        rc.append(c)
    rot, pil, x, y = [], [], 1, 0
    for t in range(24):
        rot.append(((t + 1) * (t + 2) // 2) % 64)
        x, y = y, (2 * x + 3 * y) % 5
        pil.append(x + 5 * y)
    return rc, rot, pil


def test_the_keccak_tables_in_golden_rs_are_fips_202s():
    rc, rot, pil = _keccak_constants()
    got_rc = [int(v, 16) for v in re.search(r"const RC: \[u64; 24\] = \[(.*?)\];", SRC, re.S).group(1).replace("\n", " ").split(",") if v.strip()]
    got_rot = [int(v) for v in re.search(r"const ROT: \[u32; 24\] = \[(.*?)\];", SRC).group(1).split(",")]
    got_pil = [int(v) for v in re.search(r"const PIL: \[usize; 24\] = \[(.*?)\];", SRC).group(1).split(",")]
    assert got_rc == rc and got_rot == rot and got_pil == pil
    assert "0x1F" in SRC and "RATE: usize = 136" in SRC and "46b9dd2b0ba88d13233b3feb743eeb243fcd52ea62b81b82b50c27646ed5762f" in SRC
    assert hashlib.shake_256(b"").digest(32).hex() == "46b9dd2b0ba88d13233b3feb743eeb243fcd52ea62b81b82b50c27646ed5762f"


def _head(major, n):
    return bytes([major << 5 | n]) if n < 24 else bytes([major << 5 | 24, n]) if n < 256 else bytes([major << 5 | 25, n >> 8, n & 255])


def _frame_flat(rec):
    """what golden.rs frame_flat does: map {1: bstr32, ..., n: bstr32}"""
    n = len(rec) // 32
    return _head(5, n) + b"".join(_head(0, i + 1) + b"\x58\x20" + rec[32 * i:32 * i + 32] for i in range(n))


def _frame_proof(rec, L):
    """what golden.rs frame_proof does (same order of heads and byte strings)"""
    f = lambda i: b"\x58\x20" + rec[32 * i:32 * i + 32]
    out = _head(5, 17)
    for i in range(4):
        out += _head(0, i + 1) + f(i)
    out += _head(0, 5) + _head(4, L) + b"".join(f(4 + j) for j in range(L))
    for i in range(8):
        out += _head(0, 6 + i) + f(4 + L + i)
    out += _head(0, 14) + _head(4, L) + b"".join(f(12 + L + j) for j in range(L))
    out += _head(0, 15) + _head(4, L) + b"".join(_head(4, 2) + f(12 + 2 * L + 2 * j) + f(13 + 2 * L + 2 * j) for j in range(L))
    out += _head(0, 16) + f(12 + 4 * L) + _head(0, 17) + f(13 + 4 * L)
    return out


def test_the_framing_golden_rs_rebuilds_is_the_crates():
    # the index arithmetic golden.rs uses appears in its source as written here
    for frag in ("f(4 + j)", "f(4 + L + i)", "f(12 + L + j)", "f(12 + 2 * L + 2 * j)", "f(13 + 2 * L + 2 * j)", "f(12 + 4 * L)", "f(13 + 4 * L)", "head(&mut out, 5, 17)"):
        assert frag in SRC, frag
    g = load_golden("sodium_lifecycle_L128.json")
    for case in g["cases"][:6]:
        for t, key in (("IssuanceRequest", "request"), ("IssuanceResponse", "response"), ("CreditToken", "token"), ("PreRefund", "prerefund"), ("PreIssuance", "pre"), ("Refund", "refund")):
            rec = bytes.fromhex(case[key])
            if any(rec):
                assert _frame_flat(rec) == m.cbor_encode(t, rec, 128), (t,)
        rec = bytes.fromhex(case["proof"])
        assert _frame_proof(rec, 128) == m.cbor_encode("SpendProof", rec, 128)
    assert _frame_flat(bytes.fromhex(g["sk"])) == m.cbor_encode("PrivateKey", bytes.fromhex(g["sk"]), 128)
