import sys, os, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np
import act_amd
from act_amd import capi
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L, n = 128, 1 << 16
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=1 << 14, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("d-sk", 64))
pre = eng.pre_issuance_random(sh("d-pre", 128 * n)); req = eng.request(pre, sh("d-rq", 128 * n))
cam = b"".join(scb(1000 + i) for i in range(n))
st, resp = eng.issue(sk, req, cam, sh("d-ir", 128 * n)); assert st == bytes(n)
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp); assert st == bytes(n)
s_b = b"".join(scb(i % 300) for i in range(n))
seed = sh("d-seed", 32)
want = eng.prove_spend_seeded(tok, s_b, seed, 0)
assert want[0] == bytes(n)
print("engine verify of its own seeded proofs:", sum(eng.verify_spend(sk, want[1])))
pb = eng.proof_bytes
for tail in (4, 0, 8):
    node = capi.Node(h, L, devices=(0, 0), max_batch=1 << 14, transcript=capi.TRANSCRIPT_DEVICE)
    node.set_balance(True, tail)
    got = node.prove_spend_seeded(tok, s_b, seed, 0)
    print("tail", tail, "stats", node.device_stats(), node.balance_state())
    bad = [i for i in range(n) if got[1][pb * i:pb * (i + 1)] != want[1][pb * i:pb * (i + 1)]]
    print("  seeded prover: lanes differing from one context:", len(bad), bad[:5], bad[-5:])
    got_b = node.prove_spend(tok[:160 * n], s_b, b"\x07" * 0 + sh("d-pr", 64) * (eng.prove_rng_bytes // 64) * n) if False else None
    stv = node.verify_spend(sk, want[1])
    print("  node verify statuses nonzero:", sum(1 for v in stv if v), node.device_stats())
    rrng = sh("d-rr", 128 * 64) * (n // 64)
    str_, rf = node.refund(sk, want[1], rrng, capi.RNG_PER_LANE)
    st1, rf1 = eng.refund(sk, want[1], rrng, capi.RNG_PER_LANE)
    print("  node refund: nonzero", sum(1 for v in str_ if v), "equal to one context:", (str_, rf) == (st1, rf1), node.device_stats())
    node.close()
# direct: a 4096-lane seeded call on one context, at an offset
sub = eng.prove_spend_seeded(tok[160 * 61440:], s_b[32 * 61440:], seed, 61440)
print("direct 4096-lane seeded call equals the big call's lanes:", sub[1] == want[1][pb * 61440:], sub[0] == bytes(4096))
sub = eng.prove_spend_seeded(tok[160 * 61440:160 * 61440 + 160 * 100], s_b[32 * 61440:32 * 61440 + 3200], seed, 61440)
print("direct 100-lane seeded call:", sub[1] == want[1][pb * 61440:pb * 61540])
