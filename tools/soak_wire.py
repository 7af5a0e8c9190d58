"""Randomised soak of the wire-level path: engine-made spend proofs as CBOR messages with random damage -- bits flipped anywhere
(framing or payload), bytes replaced, messages cut short or extended, every non-canonical spelling of tests/test_cbor.py -- through
act_refund_cbor_batch (sequential rng, both transcript modes) and act_redeem_cbor_batch, against the server loop restated: the Python
model's from_cbor, the C oracle's refund, the Python model's to_cbor (tests/test_gpu_wire.py _loop).  Every status, every refund
byte and the number of rng bytes drawn must agree.  Test infrastructure (uses oracle/); run on a GPU box:
    python tools/soak_wire.py [messages] [seed]        ACT_SOAK_L=8,128"""
import hashlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from act_amd import capi
from oracle_c import Oracle
import pymodel as m
from test_cbor import _variants
from test_gpu_wire import _loop

sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
scb = lambda v: (v % m.ELL).to_bytes(32, "little")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = random.Random(seed)
total = 0
for L in [int(x) for x in os.environ.get("ACT_SOAK_L", "8,128").split(",")]:
    o = Oracle(); h = o.params_new("soak-wire", "svc", "env", "v%d" % seed); octx = o.ctx(h, L)
    eng = capi.Engine(h, L, max_batch=int(os.environ.get("ACT_SOAK_MAX_BATCH", "100")), transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(sh("wsk%d" % seed, 64))
    base = max(8, n // 4)                              # distinct proofs; messages repeat them (double spends for redeem)
    pre = eng.pre_issuance_random(sh("wpre%d" % seed, 128 * base)); req = eng.request(pre, sh("wrq%d" % seed, 128 * base))
    amounts = [r.randrange(1, 1 << min(L, 60)) for _ in range(base)]
    st, resp = eng.issue(sk, req, b"".join(scb(a) for a in amounts), sh("wir%d" % seed, 128 * base))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    spend = [r.randrange(0, a + 1) for a in amounts]
    st, proofs, prer = eng.prove_spend(tok, b"".join(scb(s) for s in spend), sh("wpr%d" % seed, eng.prove_rng_bytes * base))
    assert st == bytes(base)
    pb = eng.proof_bytes
    canon = eng.cbor_encode("SpendProof", proofs)
    msgs = []
    for i in range(n):
        b = r.randrange(base)
        msg = bytearray(canon[b])
        x = r.random()
        if x < 0.30:
            pass                                       # as sent
        elif x < 0.55:                                 # one bit anywhere
            k = r.randrange(len(msg)); msg[k] ^= 1 << r.randrange(8)
        elif x < 0.65:                                 # a bit in the framing (the first bytes, or a head byte in front of a payload)
            k = r.choice([0, 1, 2, 3, 36, 37, 38, len(msg) - 35, len(msg) - 34, len(msg) - 33]); msg[k] ^= 1 << r.randrange(8)
        elif x < 0.72:                                 # a byte replaced
            msg[r.randrange(len(msg))] = r.randrange(256)
        elif x < 0.78:                                 # cut short / extended
            msg = msg[:r.randrange(len(msg))] if r.random() < 0.7 else msg + bytes(r.randrange(256) for _ in range(r.randrange(1, 9)))
        else:                                          # another spelling of the same record (non-canonical but acceptable, or broken)
            vs = _variants("SpendProof", proofs[pb * b:pb * b + pb], L)
            msg = bytearray(vs[r.randrange(len(vs))][0])
        if len(msg) and r.random() < 0.3:              # ... and damaged a SECOND time: the status must still be from_cbor's (the first
            k = r.randrange(len(msg)); msg[k] ^= 1 << r.randrange(8)     # failure in wire order), not merely "rejected"
        msgs.append(bytes(msg))
    # the codec alone, all nine types, every spelling of tests/test_cbor.py damaged once more at random: code equality with from_cbor
    recs9 = {"IssuanceRequest": req[:128], "IssuanceResponse": resp[:160], "SpendProof": proofs[:pb], "PrivateKey": sk, "PublicKey": sk[32:],
             "PreIssuance": pre[:64], "CreditToken": tok[:160], "PreRefund": prer[:96]}
    st0, rf0 = eng.refund(sk, proofs[:pb], sh("wr0%d" % seed, 128)); recs9["Refund"] = rf0
    for t, rec in recs9.items():
        vs = [bytearray(v) for v, _ in _variants(t, rec, L)]
        more = []
        for v in vs:
            for _ in range(3):
                w = bytearray(v)
                if len(w):
                    k = r.randrange(len(w)); w[k] ^= 1 << r.randrange(8)
                more.append(w)
        batch = [bytes(v) for v in vs + more]
        st9, out9 = eng.cbor_decode(t, batch)
        rb9 = len(rec)
        for i, msg9 in enumerate(batch):
            es, er = m.cbor_decode(t, msg9, L)
            assert st9[i] == es and out9[rb9 * i:rb9 * i + rb9] == er, ("codec", t, L, i, st9[i], es, msg9[:40].hex())
        total += len(batch)
    stream = sh("wrr%d" % seed, 128 * n)
    want = _loop(octx, sk, L, msgs, stream)
    db = set()
    want_r = _loop(octx, sk, L, msgs, stream, db)
    for mode in (capi.TRANSCRIPT_DEVICE, capi.TRANSCRIPT_HOST):
        eng.set_transcript_mode(mode)
        g = capi.ReplayRng(stream)
        st, out = eng.refund_cbor(sk, msgs, g, capi.RNG_CALLBACK)
        assert st == want[0], ("status mismatch", L, mode, [(i, st[i], want[0][i]) for i in range(n) if st[i] != want[0][i]][:10])
        assert out == want[1] and g.pos == want[2], ("refund bytes / rng position", L, mode)
        ns = capi.NullifierSet(4 * n)
        g = capi.ReplayRng(stream)
        st, out = eng.redeem_cbor(ns, sk, msgs, g, capi.RNG_CALLBACK)
        assert (st, out) == want_r[:2] and g.pos == want_r[2] and len(ns) == len(db), ("redeem mismatch", L, mode)
        ns.close()
    hist = {}
    for s in want_r[0]:
        hist[s] = hist.get(s, 0) + 1
    print("L = %d: %d messages agree (statuses after redeem: %s)" % (L, n, dict(sorted(hist.items()))))
    total += n
    eng.close()
print("soak_wire ok: %d messages, seed %d" % (total, seed))
