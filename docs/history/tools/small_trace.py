#!/usr/bin/env python3
"""ACT_SMALL_TRACE=1 python docs/history/tools/small_trace.py: host-side stamps of one-proof verify / refund calls (small_impl.inc spend_small_locked)
beside the call's wall time: how much of a small call is enqueueing, how much GPU, how much the wipe and the return."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ACT_SMALL_TRACE", "1")
from act_amd import capi
sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, 128, max_batch=4096, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("sl-sk", 64))
pre = eng.pre_issuance_random(sh("sl-pre", 128)); req = eng.request(pre, sh("sl-rq", 128))
st, resp = eng.issue(sk, req, (777).to_bytes(32, "little"), sh("sl-ir", 128))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proof, prer = eng.prove_spend(tok, (123).to_bytes(32, "little"), sh("sl-pr", eng.prove_rng_bytes))
rr = sh("sl-rr", 128)
for name, f in (("verify", lambda: eng.verify_spend(sk, proof)), ("refund", lambda: eng.refund(sk, proof, rr))):
    f(); f()
    for i in range(8):
        t = time.perf_counter(); f(); w = time.perf_counter() - t
        sys.stderr.write("   %s wall %.0f us\n" % (name, 1e6 * w))
