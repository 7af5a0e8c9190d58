"""Runs inside a subprocess started by tests/test_sanitizers.py with libasan preloaded: replays fixtures through the
AddressSanitizer + UBSan builds of the C oracle and of the device arithmetic headers / spend-kernel lane bodies
(tests/hostcheck).  Any report aborts the process (-fno-sanitize-recover, ASAN abort_on_error)."""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_c import Oracle  # noqa: E402

hx = bytes.fromhex
ELL = 2**252 + 27742317777372353535851937790883648493
shake = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
load = lambda n: json.load(open(os.path.join(ROOT, "tests", "golden", n)))


def oracle_lifecycles(path):
    o = Oracle(path)
    prim = load("sodium_primitives.json")
    for v in prim["scalarmult"][:60]:
        assert o.mul(hx(v["point"]), hx(v["scalar"])).hex() == v["out"]
    for v in prim["decode_validity"]:
        assert o.decode_encode(hx(v["bytes"]))[0] == v["valid"]
    for v in prim["from_uniform_bytes"][:40]:
        assert o.from_uniform(hx(v["uniform"])).hex() == v["encoding"]
    for name, picks in (("sodium_lifecycle_L64.json", None), ("sodium_lifecycle_L128.json", (0, 6, 11, 14, 15))):
        g = load(name)
        L = g["L"]
        ctx = o.ctx(hx(g["params"]), L)
        sk = hx(g["sk"])
        for idx, c in enumerate(g["cases"]):
            if picks is not None and idx not in picks:
                continue
            tag = g["tag_fmt"] % idx
            pre = ctx.pre_issuance_random(shake(tag + "-pre", 128))
            req = ctx.request(pre, shake(tag + "-request", 128))
            assert req.hex() == c["request"]
            st, resp = ctx.issue(sk, req, (int(c["c"]) % ELL).to_bytes(32, "little"), shake(tag + "-issue", 128))
            assert resp.hex() == c["response"]
            st, tok = ctx.issuance_to_credit_token(pre, sk[32:], req, resp)
            st, proof, prer = ctx.prove_spend(tok, (int(c["s"]) % ELL).to_bytes(32, "little"), shake(tag + "-prove", ctx.prove_rng_bytes))
            assert prer.hex() == c["prerefund"]
            st, rf = ctx.refund(sk, hx(c["proof"]), shake(tag + "-refund", 128))
            assert st == c["status"] and rf.hex() == c["refund"]
            if st == 0:
                assert ctx.refund_to_credit_token(prer, hx(c["proof"]), rf, sk[32:])[1].hex() == c["token2"]
        # the threaded batch entry points (what bench.py's cpu_baseline drives)
        proofs = b"".join(hx(c["proof"]) for c in g["cases"])
        assert list(ctx.verify_spend_batch(sk, proofs, 4)) == [c["status"] for c in g["cases"]]
    print("oracle: ok")


def device_headers(path):
    hc = C.CDLL(path)
    prim = load("sodium_primitives.json")

    def call(fn, *ins, nout=1):
        outs = [C.create_string_buffer(32) for _ in range(nout)]
        r = getattr(hc, fn)(*ins, *outs)
        return (r, *[x.raw for x in outs])
    sms = prim["scalarmult"]
    for i, v in enumerate(sms[:48]):
        w = sms[(i + 1) % len(sms)]
        for fn in ("hc_chain_b2", "hc_chain_bu", "hc_chain2u"):
            ok, o0, o1 = call(fn, hx(v["point"]), hx(v["scalar"]), hx(w["scalar"]), nout=2)
            assert ok and o0.hex() == v["out"], (fn, i)
    for v in prim["decode_validity"]:
        ok, enc = call("hc_decode_encode", hx(v["bytes"]))
        assert bool(ok) == v["valid"]
    for v in prim["from_uniform_bytes"][:40]:
        assert call("hc_from_uniform", hx(v["uniform"]))[1].hex() == v["encoding"]
    for v in prim["sc_reduce_wide"]:
        assert call("hc_sc_reduce_wide", hx(v["in"]))[1].hex() == v["out"]
    for v in prim["sc_ring"][:30]:
        assert call("hc_sc_invert", hx(v["a"]))[1].hex() == v["inv_a"]
    gen = next(x["out"] for x in prim["scalarmult_base"] if int.from_bytes(hx(x["scalar"]), "little") == 1)
    for v in prim["scalarmult_base"][:3]:
        assert call("hc_fixed_base", hx(gen), hx(v["scalar"]))[1].hex() == v["out"]
    for v in load("blake3_llvm.json")["vectors"]:
        if v["len"] <= 16385:
            data = bytes(i % 251 for i in range(v["len"]))
            out = C.create_string_buffer(64)
            hc.hc_blake3_xof64(data, v["len"], out)
            assert out.raw.hex() == v["xof"][:128]
    # the spend-verification kernels' lane bodies on whole proofs (L = 64 fixture: every case; L = 3: ragged encode batches)
    g = load("sodium_lifecycle_L64.json")
    L, cases = g["L"], g["cases"]
    n = len(cases)
    proofs = b"".join(hx(c["proof"]) for c in cases)
    tb = 184 + 40 * (6 + 3 * L)
    tr = C.create_string_buffer(n * tb); st = C.create_string_buffer(n); kp = C.create_string_buffer(32 * n)
    counts = (C.c_uint64 * 25)()
    assert hc.hc_spend_verify(hx(g["params"]), L, hx(g["sk"]), n, proofs, tr, st, kp, counts) == 1
    assert list(st.raw) == [c["status"] for c in cases]
    for i, c in enumerate(cases):
        if "verifier_transcript_sha256" in c:
            assert hashlib.sha256(tr.raw[i * tb:(i + 1) * tb]).hexdigest() == c["verifier_transcript_sha256"]
    print("device headers: ok")


if __name__ == "__main__":
    oracle_lifecycles(sys.argv[1])
    device_headers(sys.argv[2])
    print("SANITIZERS CLEAN")
