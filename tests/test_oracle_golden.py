"""The C oracle against the committed fixtures produced by the independent Python model (tests/golden/)."""
import hashlib

import pytest

from conftest import load_golden, shake, scb


def test_primitives(oracle):
    g = load_golden("primitives.json")
    for k, h in enumerate(g["generator_multiples"]):
        assert oracle.mul_base(scb(k)).hex() == h
    for v in g["sc_from_wide"]:
        assert oracle.sc_reduce_wide(bytes.fromhex(v["in"])).hex() == v["out"]
    for v in g["sc_muladd_invert"]:
        a, b, c = (bytes.fromhex(v[k]) for k in "abc")
        assert oracle.sc_muladd(a, b, c).hex() == v["muladd"]
        assert oracle.sc_invert(a).hex() == v["inv_a"]
    for v in g["from_uniform_bytes"]:
        enc = oracle.from_uniform(bytes.fromhex(v["uniform"]))
        assert enc.hex() == v["encoding"]
        s = bytes.fromhex(v["scalar"])
        assert oracle.mul(enc, s).hex() == v["mul"]
        assert oracle.add(enc, enc).hex() == v["double"]
        assert oracle.add(enc, oracle.mul_base(s)).hex() == v["plus_gen_mul"]
    for v in g["decode_validity"]:
        ok, _ = oracle.decode_encode(bytes.fromhex(v["bytes"]))
        assert ok == v["valid"], v["bytes"]
    for v in g["params"]:
        assert oracle.params_new(*v["args"]).hex() == v["h"]


@pytest.mark.parametrize("name", ["lifecycle_L128.json", "lifecycle_L64.json"])
def test_lifecycle(oracle, name):
    g = load_golden(name)
    L = g["L"]
    ctx = oracle.ctx(bytes.fromhex(g["params"]), L)
    assert oracle.params_new(*g["params_args"]).hex() == g["params"]
    sk, sk2 = bytes.fromhex(g["sk"]), bytes.fromhex(g["sk_other"])
    assert ctx.private_key_random(shake("golden-sk", 64)) == sk
    for idx, c in enumerate(g["cases"]):
        tag = "L%d-case%d" % (L, idx)
        pre = ctx.pre_issuance_random(shake(tag + "-pre", 128))
        assert pre.hex() == c["pre"]
        req = ctx.request(pre, shake(tag + "-request", 128))
        assert req.hex() == c["request"]
        st, resp = ctx.issue(sk, req, scb(int(c["c"])), shake(tag + "-issue", 128))
        assert st == 0 and resp.hex() == c["response"]
        st, tok = ctx.issuance_to_credit_token(pre, sk[32:], req, resp)
        assert st == 0 and tok.hex() == c["token"]
        st, proof, prer = ctx.prove_spend(tok, scb(int(c["s"])), shake(tag + "-prove", ctx.prove_rng_bytes))
        assert st == 0 and prer.hex() == c["prerefund"]
        if c["tamper"] is None:
            assert proof.hex() == c["proof"]
        proof = bytes.fromhex(c["proof"])
        st, kp = ctx.verify_spend(sk, proof)
        assert st == c["status"]
        if "kprime" in c:
            assert kp.hex() == c["kprime"]
        st, rf = ctx.refund(sk, proof, shake(tag + "-refund", 128))
        assert st == c["status"] and rf.hex() == c["refund"]
        if st == 0:
            st2, tok2 = ctx.refund_to_credit_token(prer, proof, rf, sk[32:])
            assert st2 == 0 and tok2.hex() == c["token2"]
        assert ctx.refund(sk2, proof, shake(tag + "-refund", 128))[0] == c["status_other_issuer"]
