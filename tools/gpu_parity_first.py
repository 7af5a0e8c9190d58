"""First end-to-end GPU parity run: HIP engine (through the C ABI) vs the C oracle on seeded inputs."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import act_amd
from act_amd import capi
from oracle_c import Oracle

sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L = int(os.environ.get("ACT_L", "128"))
o = Oracle()
hp = o.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
t = time.time(); hg = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01"); print("params_new", time.time() - t)
assert hg == hp, "Params::new mismatch"
octx = o.ctx(hp, L)
t = time.time(); eng = capi.Engine(hp, L, max_batch=64); print("ctx create", time.time() - t)
sk = octx.private_key_random(sh("sk", 64)); assert eng.private_key_random(sh("sk", 64)) == sk
N = 12
pre = b"".join(octx.pre_issuance_random(sh("pre%d" % i, 128)) for i in range(N))
assert eng.pre_issuance_random(b"".join(sh("pre%d" % i, 128) for i in range(N))) == pre
rq = b"".join(sh("rq%d" % i, 128) for i in range(N))
req_o = octx.request_batch(pre, rq)
for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
    eng.set_transcript_mode(mode)
    assert eng.request(pre, rq) == req_o, "request mismatch mode %d" % mode
print("request ok")
cam = b"".join(scb(1000 + i) for i in range(N))
req_bad = bytearray(req_o); req_bad[128 * 3 + 64] ^= 1; req_bad[128 * 7 + 5] ^= 0x40   # lane 3 tampered k_bar, lane 7 broken K
req_bad = bytes(req_bad)
irng = b"".join(sh("ir%d" % i, 128) for i in range(N))
st_o, resp_o = octx.issue_batch(sk, req_bad, cam, irng)
for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
    eng.set_transcript_mode(mode)
    st_g, resp_g = eng.issue(sk, req_bad, cam, irng)
    assert st_g == st_o, (list(st_g), list(st_o)); assert resp_g == resp_o, "issue mismatch"
print("issue ok, statuses", list(st_o))
# sequential rng mode == loop sharing one stream
acc = [i for i in range(N) if st_o[i] == 0]
st_s, resp_s = eng.issue(sk, req_bad, cam, irng, capi.RNG_SEQUENTIAL)
cur = 0
for i in range(N):
    s1, r1 = octx.issue(sk, req_bad[128*i:128*i+128], cam[32*i:32*i+32], irng[128*cur:128*cur+128])
    assert s1 == st_s[i] and r1 == resp_s[160*i:160*i+160], i
    cur += (s1 == 0)
print("issue sequential-rng ok")
# tokens + proofs from the oracle
st_o, resp_o = octx.issue_batch(sk, req_o, cam, irng)
toks = b"".join(octx.issuance_to_credit_token(pre[64*i:64*i+64], sk[32:], req_o[128*i:128*i+128], resp_o[160*i:160*i+160])[1] for i in range(N))
spend = [0, 1, 77, 1003, 1004 + 1, 500, 2, 3, 4, 5, 6, 7]      # lane 3 spends everything, lane 4 overspends
s_b = b"".join(scb(v) for v in spend)
prng = b"".join(sh("pr%d" % i, octx.prove_rng_bytes) for i in range(N))
t = time.time(); proofs, prers = octx.prove_spend_batch(toks, s_b, prng, 8); print("oracle prove", time.time() - t)
pb = octx.proof_bytes
pr = bytearray(proofs)
pr[pb*5 + 32] ^= 1                      # lane 5: tampered s
pr[pb*6 + 64: pb*6 + 96] = bytes(32)    # lane 6: A' = identity
pr[pb*7 + 32*(4+9) + 3] ^= 0x10         # lane 7: Com_9 corrupted (likely undecodable)
pr[pb*8 + 32*(4+L) + 1] ^= 2            # lane 8: tampered gamma
proofs_t = bytes(pr)
t = time.time(); st_o = octx.verify_spend_batch(sk, proofs_t, 8); print("oracle verify", time.time() - t, list(st_o))
rrng = b"".join(sh("rr%d" % i, 128) for i in range(N))
st_ro, rf_o = octx.refund_batch(sk, proofs_t, rrng, 8)
for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
    eng.set_transcript_mode(mode)
    t = time.time(); st_g, kp = eng.verify_spend(sk, proofs_t, True); dt = time.time() - t
    print("gpu verify mode", mode, dt, list(st_g))
    assert st_g == st_o
    trs = eng.last_spend_transcripts(N)
    for i in range(N):
        if st_o[i] in (0, 7):
            so, kpo, tro = octx.verify_spend(sk, proofs_t[pb*i:pb*i+pb], True)
            assert trs[i] == tro, "transcript mismatch lane %d" % i
            assert kp[32*i:32*i+32] == kpo
    st_g, rf_g = eng.refund(sk, proofs_t, rrng)
    assert st_g == st_ro and rf_g == rf_o, "refund mismatch"
print("verify + refund ok")
st_s, rf_s = eng.refund(sk, proofs_t, rrng, capi.RNG_SEQUENTIAL)
cur = 0
for i in range(N):
    s1, r1 = octx.refund(sk, proofs_t[pb*i:pb*i+pb], rrng[128*cur:128*cur+128])
    assert s1 == st_s[i] and r1 == rf_s[128*i:128*i+128], i
    cur += (s1 == 0)
print("refund sequential-rng ok")
# multi-chunk (max_batch=64) with a bigger tiled batch
big = proofs_t * 12
eng.prof_enable(True)
t = time.time(); st_b = eng.verify_spend(sk, big); dt = time.time() - t
assert st_b == st_o * 12
print("tiled batch %d proofs in %.3fs" % (len(big)//pb, dt), eng.prof())
print("ALL OK")
