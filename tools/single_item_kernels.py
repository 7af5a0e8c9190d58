#!/usr/bin/env python3
"""Kernel time of the single-item calls, measured so that clock states do not decide the comparison: every call 30 times back to back,
min and median of each kernel's own duration (HIP events on the engine's streams) and of the call's wall time.  Run under different
ACT_NO_* knobs to A/B the tiny-call kernels:  ACT_NO_FUSED_TINY=1 / ACT_NO_WIDE_SIGN=1 / ACT_NO_WIDE_CLIENT=1 / ACT_NO_WIDE_PROVE=1."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from act_amd import capi

sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, 128, max_batch=4096, transcript=capi.TRANSCRIPT_DEVICE)
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
         ("refund_to_credit_token", lambda: eng.refund_to_credit_token(prer, proof, rf, sk[32:]))]
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("ACT_NO_"))
print("single-item calls, device transcripts, 30 calls each [%s]: wall ms min / median; kernels ms min / median" % (knobs or "default build"))
for name, f in calls:
    walls, kern = [], {}
    f(); f()
    for _ in range(30):
        eng.prof_reset(); eng.prof_enable(True)
        t = time.perf_counter(); f(); walls.append(time.perf_counter() - t)
        eng.prof_enable(False)
        for k, v in eng.prof().items():
            if not k.startswith("copy"):
                kern.setdefault(k, []).append(v["ms"])
    walls.sort()
    ks = "  ".join("%s %.3f/%.3f" % (k, min(v), sorted(v)[len(v) // 2]) for k, v in kern.items())
    print("  %-26s %.2f / %.2f    %s" % (name, 1e3 * walls[0], 1e3 * walls[len(walls) // 2], ks))
