"""The HIP engine, through the C ABI, against known answers computed by libsodium 1.0.18 + LLVM BLAKE3
(tests/golden/sodium_*.json, generated in the authoring container by tests/golden/make_sodium_golden.py): arithmetic
the engine shares nothing with.  Bit-exact on every record, status, K' and "spend" transcript."""
import hashlib

import pytest

from conftest import ELL, load_golden, shake

pytestmark = pytest.mark.gpu

hx = bytes.fromhex
MODES = [0, 1]   # ACT_TRANSCRIPT_HOST, ACT_TRANSCRIPT_DEVICE


@pytest.fixture(scope="module")
def prim():
    return load_golden("sodium_primitives.json")


@pytest.fixture(scope="module")
def eng8(engine_factory, prim):
    return engine_factory(hx(prim["params"][0]["h"]), 8, max_batch=128)


def test_scalarmult_known_answers(eng8, prim):
    """320 x `RistrettoPoint * Scalar` through the production chain / decode / encode (ragged over max_batch = 128)."""
    v = prim["scalarmult"]
    st, out = eng8.debug_scalarmult(b"".join(hx(x["point"]) for x in v), b"".join(hx(x["scalar"]) for x in v))
    assert st == bytes(len(v))
    for i, x in enumerate(v):
        assert out[32 * i:32 * i + 32].hex() == x["out"], i
    # undecodable points: status 255, zero record; valid ones decode (and 1 * P re-encodes to the same bytes)
    d = prim["decode_validity"]
    one = (1).to_bytes(32, "little")
    st, out = eng8.debug_scalarmult(b"".join(hx(x["bytes"]) for x in d), one * len(d))
    for i, x in enumerate(d):
        assert (st[i] == 0) == x["valid"], x
        assert out[32 * i:32 * i + 32] == (hx(x["bytes"]) if x["valid"] else bytes(32))


def test_base_mult_and_wide_reduction_known_answers(eng8, prim):
    """PrivateKey::random = wide reduction + generator mult; PreIssuance::random = two wide reductions."""
    wide = prim["sc_reduce_wide"]
    base = {x["scalar"]: x["out"] for x in prim["scalarmult_base"]}
    for x in wide[:60]:
        sk = eng8.private_key_random(hx(x["in"]))
        assert sk[:32].hex() == x["out"]
    # generator multiples: feed the scalar itself as the low half of the 64 rng bytes (already < l: reduction is a no-op)
    for s, out in list(base.items()):
        sk = eng8.private_key_random(hx(s) + bytes(32))
        assert sk[:32].hex() == s and sk[32:].hex() == out
    n = len(wide) // 2
    pre = eng8.pre_issuance_random(b"".join(hx(wide[2 * i]["in"]) + hx(wide[2 * i + 1]["in"]) for i in range(n)))
    for i in range(n):
        assert pre[64 * i:64 * i + 32].hex() == wide[2 * i]["out"] and pre[64 * i + 32:64 * i + 64].hex() == wide[2 * i + 1]["out"]


def test_one_way_map_and_params_known_answers(prim):
    from act_amd import capi
    fh = prim["from_uniform_bytes"]
    for i in range(0, len(fh) - 2, 3):
        got = capi.params_random(b"".join(hx(fh[i + k]["uniform"]) for k in range(3)))
        assert got.hex() == "".join(fh[i + k]["encoding"] for k in range(3)), i
    for v in prim["params"]:
        assert capi.params_new(*v["args"]).hex() == v["h"]
    for v in prim["params_random"]:
        assert capi.params_random(hx(v["rng"])).hex() == v["h"]


def test_cbor_decode_validity_and_scalar_reduction(eng8, prim):
    """decode_point / decode_scalar (src/cbor.rs:59-91) on the GPU unframe path: every rejection class, 32-byte reduction."""
    d = prim["decode_validity"]
    msgs = [b"\x58\x20" + hx(x["bytes"]) for x in d]                 # PublicKey = one bare 32-byte byte string
    st, rec = eng8.cbor_decode("PublicKey", msgs)
    for i, x in enumerate(d):
        assert st[i] == (0 if x["valid"] else 3), x
        assert rec[32 * i:32 * i + 32] == (hx(x["bytes"]) if x["valid"] else bytes(32))
    r = prim["sc_reduce32"]
    msgs = [b"\xa2\x01\x58\x20" + hx(r[i]["in"]) + b"\x02\x58\x20" + hx(r[(i + 1) % len(r)]["in"]) for i in range(len(r))]   # PreIssuance {1: r, 2: k}
    st, rec = eng8.cbor_decode("PreIssuance", msgs)
    assert st == bytes(len(r))
    for i in range(len(r)):
        assert rec[64 * i:64 * i + 32].hex() == r[i]["out"] and rec[64 * i + 32:64 * i + 64].hex() == r[(i + 1) % len(r)]["out"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["sodium_lifecycle_L128.json", "sodium_lifecycle_L64.json"])
def test_sodium_lifecycles(engine_factory, name, mode):
    g = load_golden(name)
    L = g["L"]
    eng = engine_factory(hx(g["params"]), L, max_batch=7, transcript=mode)      # ragged chunks
    sk, sk2 = hx(g["sk"]), hx(g["sk_other"])
    assert eng.private_key_random(shake(g["sk_label"], 64)) == sk
    cases = g["cases"]
    n = len(cases)
    tag = lambda i: g["tag_fmt"] % i
    cat = lambda f: b"".join(f(i) for i in range(n))
    scb = lambda v: (v % ELL).to_bytes(32, "little")
    pre = eng.pre_issuance_random(cat(lambda i: shake(tag(i) + "-pre", 128)))
    assert pre == cat(lambda i: hx(cases[i]["pre"]))
    req = eng.request(pre, cat(lambda i: shake(tag(i) + "-request", 128)))
    assert req == cat(lambda i: hx(cases[i]["request"]))
    st, resp = eng.issue(sk, req, cat(lambda i: scb(int(cases[i]["c"]))), cat(lambda i: shake(tag(i) + "-issue", 128)))
    assert st == bytes(n) and resp == cat(lambda i: hx(cases[i]["response"]))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(n) and tok == cat(lambda i: hx(cases[i]["token"]))
    st, proofs, prer = eng.prove_spend(tok, cat(lambda i: scb(int(cases[i]["s"]))), cat(lambda i: shake(tag(i) + "-prove", eng.prove_rng_bytes)))
    assert st == bytes(n) and prer == cat(lambda i: hx(cases[i]["prerefund"]))
    pb = eng.proof_bytes
    for i, c in enumerate(cases):
        if c["tamper"] is None:
            assert proofs[pb * i:pb * i + pb].hex() == c["proof"], i
    proofs = cat(lambda i: hx(cases[i]["proof"]))
    st, kp = eng.verify_spend(sk, proofs, True)
    assert list(st) == [c["status"] for c in cases]
    # the verifier's transcript pre-images, byte for byte (through their SHA-256), chunk by chunk
    for lo in range(0, n, 7):
        hi = min(n, lo + 7)
        st_c = eng.verify_spend(sk, proofs[pb * lo:pb * hi])
        trs = eng.last_spend_transcripts(hi - lo)
        for i in range(lo, hi):
            if "verifier_transcript_sha256" in cases[i]:
                assert hashlib.sha256(trs[i - lo]).hexdigest() == cases[i]["verifier_transcript_sha256"], i
    for i, c in enumerate(cases):                                   # K' for accepted lanes, a zero record for every rejected one
        assert kp[32 * i:32 * i + 32].hex() == (c["kprime"] if c["status"] == 0 else "00" * 32), i
    st, rf = eng.refund(sk, proofs, cat(lambda i: shake(tag(i) + "-refund", 128)))
    assert list(st) == [c["status"] for c in cases]
    assert rf == cat(lambda i: hx(cases[i]["refund"]))
    st2, tok2 = eng.refund_to_credit_token(prer, proofs, rf, sk[32:])
    for i, c in enumerate(cases):
        if c["status"] == 0:
            assert st2[i] == 0 and tok2[160 * i:160 * i + 160].hex() == c["token2"]
        else:
            assert st2[i] != 0 and tok2[160 * i:160 * i + 160] == bytes(160)
    assert list(eng.verify_spend(sk2, proofs)) == [c["status_other_issuer"] for c in cases]
