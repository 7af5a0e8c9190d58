"""Pins all three of our CPU-side implementations -- oracle/pymodel.py, the C oracle, and the host build of the device
arithmetic headers -- against tests/golden/sodium_*.json: known answers computed by libsodium 1.0.18 (ristretto255 group
and scalar arithmetic) and LLVM's BLAKE3 through oracle/sodium_model.py, i.e. by arithmetic none of them shares
(tests/golden/make_sodium_golden.py).  The HIP engine is held to the same files in tests/test_gpu_sodium.py."""
import ctypes as C
import hashlib

import pytest

import pymodel as m
from conftest import load_golden, shake

PRIM = load_golden("sodium_primitives.json")
hx = bytes.fromhex
i_le = lambda b: int.from_bytes(b, "little")


def test_fixture_is_large_enough():
    assert PRIM["count"] >= 1000
    assert len(PRIM["decode_validity"]) >= 200
    classes = {v["class"] for v in PRIM["decode_validity"]}
    assert classes == {"valid", "non_canonical", "negative_s", "non_square", "negative_t", "zero_y"}
    assert len(load_golden("sodium_lifecycle_L128.json")["cases"]) >= 8
    assert len(load_golden("sodium_lifecycle_L64.json")["cases"]) >= 3


# ---- primitives: Python model ----------------------------------------------------------------------------------------
def test_pymodel_primitives():
    for v in PRIM["scalarmult"]:
        p = m.ristretto_decode(hx(v["point"]))
        assert p is not None
        assert m.ristretto_encode(m.pt_mul(p, i_le(hx(v["scalar"])))).hex() == v["out"]
    for v in PRIM["scalarmult_base"][::4]:
        assert m.ristretto_encode(m.pt_mul(m.BASEPOINT, i_le(hx(v["scalar"])))).hex() == v["out"]
    for v in PRIM["from_uniform_bytes"]:
        assert m.ristretto_encode(m.ristretto_from_uniform_bytes(hx(v["uniform"]))).hex() == v["encoding"]
    for v in PRIM["sc_reduce_wide"]:
        assert m.sc_bytes(m.sc_from_wide(hx(v["in"]))).hex() == v["out"]
    for v in PRIM["sc_reduce32"]:
        assert m.sc_bytes(m.sc_from_bytes_mod_order(hx(v["in"]))).hex() == v["out"]
    for v in PRIM["sc_ring"]:
        a, b = i_le(hx(v["a"])), i_le(hx(v["b"]))
        assert m.sc_bytes(a + b).hex() == v["add"] and m.sc_bytes(a - b).hex() == v["sub"] and m.sc_bytes(a * b).hex() == v["mul"]
        assert m.sc_bytes(-a).hex() == v["neg_a"] and m.sc_bytes(m.sc_inv(a)).hex() == v["inv_a"]
    for v in PRIM["point_add_sub"]:
        p, q = m.ristretto_decode(hx(v["p"])), m.ristretto_decode(hx(v["q"]))
        assert m.ristretto_encode(m.pt_add(p, q)).hex() == v["add"] and m.ristretto_encode(m.pt_sub(p, q)).hex() == v["sub"]
    for v in PRIM["decode_validity"]:
        assert (m.ristretto_decode(hx(v["bytes"])) is not None) == v["valid"], v
    for v in PRIM["params"]:
        assert m.Params.new(*v["args"]).encoded().hex() == v["h"]
    for v in PRIM["params_random"]:
        assert m.Params.random(m.ByteRng(hx(v["rng"]))).encoded().hex() == v["h"]


# ---- primitives: C oracle --------------------------------------------------------------------------------------------
def test_c_oracle_primitives(oracle):
    o = oracle
    one = (1).to_bytes(32, "little")
    zero = bytes(32)
    for v in PRIM["scalarmult"]:
        assert o.mul(hx(v["point"]), hx(v["scalar"])).hex() == v["out"]
    for v in PRIM["scalarmult_base"]:
        assert o.mul_base(hx(v["scalar"])).hex() == v["out"]
    for v in PRIM["from_uniform_bytes"]:
        assert o.from_uniform(hx(v["uniform"])).hex() == v["encoding"]
    for v in PRIM["sc_reduce_wide"]:
        assert o.sc_reduce_wide(hx(v["in"])).hex() == v["out"]
    for v in PRIM["sc_reduce32"]:
        assert o.sc_reduce_wide(hx(v["in"]) + bytes(32)).hex() == v["out"]
        assert o.sc_muladd(hx(v["in"]), one, zero).hex() == v["out"]           # the 32-byte loader reduces too
    ell_m1 = (m.ELL - 1).to_bytes(32, "little")
    for v in PRIM["sc_ring"]:
        a, b = hx(v["a"]), hx(v["b"])
        assert o.sc_muladd(a, one, b).hex() == v["add"] and o.sc_muladd(a, b, zero).hex() == v["mul"]
        assert o.sc_muladd(b, ell_m1, a).hex() == v["sub"] and o.sc_muladd(a, ell_m1, zero).hex() == v["neg_a"]
        assert o.sc_invert(a).hex() == v["inv_a"]
    for v in PRIM["point_add_sub"]:
        assert o.add(hx(v["p"]), hx(v["q"])).hex() == v["add"]
        assert o.add(hx(v["p"]), o.mul(hx(v["q"]), ell_m1)).hex() == v["sub"]
    for v in PRIM["decode_validity"]:
        ok, enc = o.decode_encode(hx(v["bytes"]))
        assert ok == v["valid"], v
        if ok:
            assert enc == hx(v["bytes"])                                         # compress(decompress(x)) == x
    for v in PRIM["params"]:
        assert o.params_new(*v["args"]).hex() == v["h"]


# ---- primitives: the device arithmetic headers, host build -----------------------------------------------------------
def _hc(hc, fn, *ins, nout=1, outlen=32):
    outs = [C.create_string_buffer(outlen) for _ in range(nout)]
    r = getattr(hc, fn)(*ins, *outs)
    return (r, *[x.raw for x in outs])


def test_device_headers_primitives(hostcheck):
    hc = hostcheck
    sms = PRIM["scalarmult"]
    for i, v in enumerate(sms):
        w = sms[(i + 1) % len(sms)]
        # every variable-base chain shape of msm.h on the same base: s0 * P and s1 * P
        for fn in ("hc_chain_b2", "hc_chain_bu", "hc_chain2u", "hc_chain_ct2") if i % 4 == 0 else ("hc_chain_b2", "hc_chain_bu"):
            ok, o0, o1 = _hc(hc, fn, hx(v["point"]), hx(v["scalar"]), hx(w["scalar"]), nout=2)
            assert ok and o0.hex() == v["out"], (fn, i)
        if i % 16 == 0:
            ok, o0 = _hc(hc, "hc_chain1", hx(v["point"]), hx(v["scalar"]))
            assert ok and o0.hex() == v["out"]
    gen = PRIM["scalarmult_base"]
    gen_enc = next(v["out"] for v in gen if i_le(hx(v["scalar"])) == 1)
    for v in gen[:8]:                                                            # fixed-base windows (table rebuilt per call: keep it short)
        ok, o0 = _hc(hc, "hc_fixed_base", hx(gen_enc), hx(v["scalar"]))
        assert ok and o0.hex() == v["out"]
    for v in gen[:24]:                                                           # the ct build's scanned radix-16 tables
        ok, o0 = _hc(hc, "hc_fixed_base_ct", hx(gen_enc), hx(v["scalar"]))
        assert ok and o0.hex() == v["out"]
    for v in gen[8:]:
        ok, o0, o1 = _hc(hc, "hc_chain_b2", hx(gen_enc), hx(v["scalar"]), hx(v["scalar"]), nout=2)
        assert ok and o0.hex() == v["out"] and o1.hex() == v["out"]
    for v in PRIM["from_uniform_bytes"]:
        assert _hc(hc, "hc_from_uniform", hx(v["uniform"]))[1].hex() == v["encoding"]
    for v in PRIM["sc_reduce_wide"]:
        assert _hc(hc, "hc_sc_reduce_wide", hx(v["in"]))[1].hex() == v["out"]
    for v in PRIM["sc_reduce32"]:
        assert _hc(hc, "hc_sc_from_bytes", hx(v["in"]))[1].hex() == v["out"]
    one, zero = (1).to_bytes(32, "little"), bytes(32)
    for v in PRIM["sc_ring"]:
        a, b = hx(v["a"]), hx(v["b"])
        assert _hc(hc, "hc_sc_muladd", a, one, b)[1].hex() == v["add"]
        assert _hc(hc, "hc_sc_muladd", a, b, zero)[1].hex() == v["mul"]
        assert _hc(hc, "hc_sc_sub", a, b)[1].hex() == v["sub"]
        assert _hc(hc, "hc_sc_neg", a)[1].hex() == v["neg_a"]
        assert _hc(hc, "hc_sc_invert", a)[1].hex() == v["inv_a"]
    for v in PRIM["point_add_sub"]:
        ok, oa, os_, od = _hc(hc, "hc_add_sub_dbl", hx(v["p"]), hx(v["q"]), nout=3)
        assert ok and oa.hex() == v["add"] and os_.hex() == v["sub"]
    for v in PRIM["decode_validity"]:
        ok, enc = _hc(hc, "hc_decode_encode", hx(v["bytes"]))
        assert bool(ok) == v["valid"], v
        if ok:
            assert enc == hx(v["bytes"])


# ---- lifecycles --------------------------------------------------------------------------------------------------
LIFECYCLES = ["sodium_lifecycle_L128.json", "sodium_lifecycle_L64.json"]


@pytest.mark.parametrize("name", LIFECYCLES)
def test_c_oracle_lifecycles(oracle, name):
    g = load_golden(name)
    L = g["L"]
    ctx = oracle.ctx(hx(g["params"]), L)
    sk, sk2 = hx(g["sk"]), hx(g["sk_other"])
    assert ctx.private_key_random(shake(g["sk_label"], 64)) == sk
    for idx, c in enumerate(g["cases"]):
        tag = g["tag_fmt"] % idx
        pre = ctx.pre_issuance_random(shake(tag + "-pre", 128))
        assert pre.hex() == c["pre"]
        req = ctx.request(pre, shake(tag + "-request", 128))
        assert req.hex() == c["request"]
        st, resp = ctx.issue(sk, req, (int(c["c"]) % m.ELL).to_bytes(32, "little"), shake(tag + "-issue", 128))
        assert st == 0 and resp.hex() == c["response"]
        st, tok = ctx.issuance_to_credit_token(pre, sk[32:], req, resp)
        assert st == 0 and tok.hex() == c["token"]
        st, proof, prer = ctx.prove_spend(tok, (int(c["s"]) % m.ELL).to_bytes(32, "little"), shake(tag + "-prove", ctx.prove_rng_bytes))
        assert st == 0 and prer.hex() == c["prerefund"]
        if c["tamper"] is None:
            assert proof.hex() == c["proof"]
        proof = hx(c["proof"])
        st, kp, tr = ctx.verify_spend(sk, proof, True)
        assert st == c["status"], (idx, st)
        if "kprime" in c:
            assert kp.hex() == c["kprime"]
            assert hashlib.sha256(tr).hexdigest() == c["verifier_transcript_sha256"]
        st, rf = ctx.refund(sk, proof, shake(tag + "-refund", 128))
        assert st == c["status"] and rf.hex() == c["refund"]
        if st == 0:
            st2, tok2 = ctx.refund_to_credit_token(prer, proof, rf, sk[32:])
            assert st2 == 0 and tok2.hex() == c["token2"]
        assert ctx.refund(sk2, proof, shake(tag + "-refund", 128))[0] == c["status_other_issuer"]


def _pymodel_case(g, idx):
    L, c = g["L"], g["cases"][idx]
    tag = g["tag_fmt"] % idx
    params = m.Params.new(*g["params_args"])
    assert params.encoded().hex() == g["params"]
    sk = m.PrivateKey.random(m.ByteRng(shake(g["sk_label"], 64)))
    assert sk.record().hex() == g["sk"]
    pre = m.PreIssuance.random(m.ByteRng(shake(tag + "-pre", 128)))
    req = m.request(pre, params, m.ByteRng(shake(tag + "-request", 128)))
    assert req.record().hex() == c["request"]
    resp = m.issue(sk, params, req, int(c["c"]), m.ByteRng(shake(tag + "-issue", 128)))
    assert resp.record().hex() == c["response"]
    tok = m.issuance_to_credit_token(pre, params, sk.w, req, resp)
    assert tok.record().hex() == c["token"]
    proof, prer = m.prove_spend(tok, params, int(c["s"]), m.ByteRng(shake(tag + "-prove", 64 * (4 * L + 12))), L)
    assert prer.record().hex() == c["prerefund"]
    if c["tamper"] is None:
        assert proof.record().hex() == c["proof"]
    try:
        pr = m.parse_spend_proof(hx(c["proof"]), L)
    except Exception:
        assert c["status"] == 255
        return
    try:
        rf = m.refund(sk, params, pr, m.ByteRng(shake(tag + "-refund", 128)))
        assert c["status"] == 0 and rf.record().hex() == c["refund"]
        tok2 = m.refund_to_credit_token(prer, params, pr, rf, sk.w)
        assert tok2.record().hex() == c["token2"]
    except m.ActError as e:
        assert e.code == c["status"]


@pytest.mark.parametrize("name,idx", [("sodium_lifecycle_L128.json", i) for i in (0, 4, 6, 11, 14, 15)] +
                         [("sodium_lifecycle_L64.json", i) for i in (0, 1, 4)])
def test_pymodel_lifecycles(name, idx):
    """The big-integer model is slow (seconds per lifecycle): a spread of cases; the C oracle above runs all of them."""
    _pymodel_case(load_golden(name), idx)


# ---- the generator is reproducible where libsodium is present (this container); skipped on the GPU box ----------------
def test_fixtures_regenerate_from_libsodium():
    import sodium_model as sm
    if not sm.available():
        pytest.skip("libsodium / LLVM BLAKE3 not present on this box: the committed JSON is the anchor")
    assert sm.sodium_version().startswith("1.0.")
    for v in PRIM["scalarmult"][:40]:
        assert sm.pmul(hx(v["point"]), hx(v["scalar"])).hex() == v["out"]
    for v in PRIM["from_uniform_bytes"][:40]:
        assert sm.from_uniform(hx(v["uniform"])).hex() == v["encoding"]
    for v in PRIM["decode_validity"]:
        assert sm.is_valid_point(hx(v["bytes"])) == v["valid"]
    g = load_golden("sodium_lifecycle_L64.json")
    L = g["L"]
    params = sm.params_new(*g["params_args"])
    sk = sm.private_key_random(sm.ByteRng(shake(g["sk_label"], 64)))
    for idx in (0, 1):
        c = g["cases"][idx]
        tag = g["tag_fmt"] % idx
        pre = sm.pre_issuance_random(sm.ByteRng(shake(tag + "-pre", 128)))
        req = sm.request(pre, params, sm.ByteRng(shake(tag + "-request", 128)))
        resp = sm.issue(sk, params, req, int(c["c"]).to_bytes(32, "little"), sm.ByteRng(shake(tag + "-issue", 128)))
        tok = sm.issuance_to_credit_token(pre, params, sk[1], req, resp)
        proof, prer, _ = sm.prove_spend(tok, params, int(c["s"]).to_bytes(32, "little"), sm.ByteRng(shake(tag + "-prove", 64 * (4 * L + 12))), L)
        assert proof.hex() == c["proof"] and prer.hex() == c["prerefund"]
        d = sm.decode_spend_proof(proof, L)
        try:
            rf = sm.refund(sk, params, d, sm.ByteRng(shake(tag + "-refund", 128)), L)
            assert c["status"] == 0 and rf.hex() == c["refund"]
        except sm.ActError as e:
            assert e.code == c["status"]
