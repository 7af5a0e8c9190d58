#!/usr/bin/env python3
"""Latency of every entry point over ONE item from host memory (the crate's call shape, benches/benchmark.rs:166-212), L = 128,
device transcripts, median of 9 -- beside bench.py's cpu_baseline.config1 (the C port on one core) this says which single-item calls
a deployment should leave on the CPU."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from act_amd import capi

sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
mode = capi.TRANSCRIPT_HOST if (len(sys.argv) > 1 and sys.argv[1] == "host") else capi.TRANSCRIPT_DEVICE
eng = capi.Engine(h, 128, max_batch=4096, transcript=mode)
sk = eng.private_key_random(sh("sl-sk", 64))
pre = eng.pre_issuance_random(sh("sl-pre", 128)); req = eng.request(pre, sh("sl-rq", 128))
c = (777).to_bytes(32, "little")
st, resp = eng.issue(sk, req, c, sh("sl-ir", 128))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
s = (123).to_bytes(32, "little")
st, proof, prer = eng.prove_spend(tok, s, sh("sl-pr", eng.prove_rng_bytes))
st, rf = eng.refund(sk, proof, sh("sl-rr", 128))
assert st == bytes(1)
calls = [("request", lambda: eng.request(pre, sh("sl-rq", 128))),
         ("issue", lambda: eng.issue(sk, req, c, sh("sl-ir", 128))),
         ("issuance_to_credit_token", lambda: eng.issuance_to_credit_token(pre, sk[32:], req, resp)),
         ("prove_spend", lambda: eng.prove_spend(tok, s, sh("sl-pr", eng.prove_rng_bytes))),
         ("verify_spend", lambda: eng.verify_spend(sk, proof)),
         ("refund", lambda: eng.refund(sk, proof, sh("sl-rr", 128))),
         ("refund_to_credit_token", lambda: eng.refund_to_credit_token(prer, proof, rf, sk[32:])),
         # the wire-level call over ONE message (CBOR SpendProof in, CBOR Refund out): verification, then the signature behind it
         ("refund_cbor (1 message)", lambda: eng.refund_cbor(sk, [msg1], sh("sl-rr", 128))),
         # the redemption step for one item: refund in one call, then the nullifier store decides (a fresh store each time would be
         # cheating the other way: the proof is a double spend after the first call, which costs the same)
         ("redeem (1 item)", lambda: eng.redeem(ns, sk, proof, sh("sl-rr", 128), capi.RNG_SEQUENTIAL)),
         ("redeem_cbor (1 message)", lambda: eng.redeem_cbor(ns, sk, [msg1], sh("sl-rr", 128), capi.RNG_SEQUENTIAL)),
         # the two library calls the Rust binding's strict `refund` makes (nothing is drawn for a rejected proof)
         ("verify + refund_sign", lambda: two_calls())]
msg1 = eng.cbor_encode("SpendProof", proof)[0]
ns = capi.NullifierSet(1024)
import numpy as np
def two_calls():
    st, kp = eng.verify_spend(sk, proof, True)
    out = np.zeros(128, np.uint8); st2 = np.zeros(1, np.uint8)
    p_sk, k1 = capi._in(sk, 64); p_kp, k2 = capi._in(kp, 32); p_st, k3 = capi._in(st, 1); p_r, k4 = capi._in(sh("sl-rr", 128), 128)
    eng._ck(eng.lib.act_refund_sign_batch(eng.ctx, 1, capi.MEM_HOST, p_sk, p_kp, p_st, p_r, capi.RNG_SEQUENTIAL, out.ctypes.data, st2.ctypes.data))
    return st2.tobytes(), out.tobytes()
assert two_calls() == eng.refund(sk, proof, sh("sl-rr", 128)) and eng.refund_cbor(sk, [msg1], sh("sl-rr", 128))[0] == bytes(1)
print("one item per call, %s transcripts, ms (median of 9):" % ("host" if mode == capi.TRANSCRIPT_HOST else "device"))
for name, f in calls:
    f(); ts = []
    for _ in range(9):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    print("  %-26s %.2f" % (name, 1e3 * sorted(ts)[4]))
# where a single-item call's time goes: the kernels' own durations (HIP events on the engine's streams) against the wall time
if len(sys.argv) > 2 and sys.argv[2] == "kernels":
    print("per call: wall ms, then the kernels / copies the engine timed (ms, launches)")
    for name, f in calls:
        eng.prof_reset(); eng.prof_enable(True)
        t = time.perf_counter(); f(); wall = time.perf_counter() - t
        eng.prof_enable(False)
        pr = eng.prof()
        print("  %-26s %.2f   %s" % (name, 1e3 * wall, "  ".join("%s %.3f x%d" % (k, v["ms"], v["launches"]) for k, v in pr.items())))
