"""GPU lifecycle (request->issue->token->prove_spend->refund->token) vs the C oracle, byte for byte."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from act_amd import capi
from oracle_c import Oracle
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
o = Oracle(); hp = o.params_new("example-corp", "payment-api", "production", "2024-01-15")
for L in (128, 64):
    octx = o.ctx(hp, L); eng = capi.Engine(hp, L, max_batch=5)
    N = 12
    sk = eng.private_key_random(sh("sk", 64)); w = sk[32:]
    for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
        eng.set_transcript_mode(mode)
        pre = eng.pre_issuance_random(sh("pre", 128 * N)); req = eng.request(pre, sh("rq", 128 * N))
        cvals = [0, 1, 2**L - 1, 1000, 5, 6, 7, 8, 9, 10, 11, 2**64 % (2**L)]
        cam = b"".join(scb(v) for v in cvals)
        st, resp = eng.issue(sk, req, cam, sh("ir", 128 * N)); assert st == bytes(N)
        resp_t = bytearray(resp); resp_t[160*2 + 32] ^= 1; resp_t[160*5 + 100] ^= 1; resp_t[160*6 + 1] ^= 0x20; resp_t = bytes(resp_t)
        st_t, tok_t = eng.issuance_to_credit_token(pre, w, req, resp_t)
        for i in range(N):
            so, to = octx.issuance_to_credit_token(pre[64*i:64*i+64], w, req[128*i:128*i+128], resp_t[160*i:160*i+160])
            assert so == st_t[i] and to == tok_t[160*i:160*i+160], ("tok", i, so, st_t[i])
        st, toks = eng.issuance_to_credit_token(pre, w, req, resp); assert st == bytes(N)
        spend = [0, 1, 2**L - 1, 1001, 0, 6, 3, 8, 1, 10, 12, 5]      # lanes 3 and 10 overspend
        s_b = b"".join(scb(v) for v in spend)
        prng = sh("pr", octx.prove_rng_bytes * N)
        t = time.time(); st, proofs, prers = eng.prove_spend(toks, s_b, prng); dt = time.time() - t
        assert st == bytes(N)
        po, pro = octx.prove_spend_batch(toks, s_b, prng, 16)
        assert proofs == po, "proofs differ"; assert prers == pro
        rrng = sh("rr", 128 * N)
        st_r, rf = eng.refund(sk, proofs, rrng)
        st_o, rf_o = octx.refund_batch(sk, proofs, rrng, 16)
        assert st_r == st_o and rf == rf_o
        exp = [0 if 0 <= c - s < 2**L else 7 for c, s in zip(cvals, spend)]
        assert list(st_r) == exp, (list(st_r), exp)
        rf_t = bytearray(rf); rf_t[128*0 + 40] ^= 1; rf_t[128*1 + 70] ^= 1; rf_t[128*4 + 2] ^= 0x80; rf_t = bytes(rf_t)
        st_c, tok2 = eng.refund_to_credit_token(prers, proofs, rf_t, w)
        for i in range(N):
            so, to = octx.refund_to_credit_token(prers[96*i:96*i+96], proofs[octx.proof_bytes*i:octx.proof_bytes*(i+1)], rf_t[128*i:128*i+128], w)
            assert so == st_c[i] and to == tok2[160*i:160*i+160], ("tok2", i, so, st_c[i])
        print("L=%d mode=%d ok (prove %.3fs) refund statuses %s client statuses %s" % (L, mode, dt, list(st_r), list(st_c)))
    # second spend from a refunded token
    st_c, tok2 = eng.refund_to_credit_token(prers, proofs, rf, w)
    good = [i for i in range(N) if st_c[i] == 0]
    i = good[3]
    st, p2, pr2 = eng.prove_spend(tok2[160*i:160*i+160], scb(1), sh("pr2", octx.prove_rng_bytes))
    assert eng.verify_spend(sk, p2) == b"\0"
print("LIFECYCLE OK")
