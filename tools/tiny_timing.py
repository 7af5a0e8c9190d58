"""Measurement build only: libact_timing.so (csrc built with -DACT_TINY_TIMING) stamps the 100 MHz wall clock inside k_sign_fused; one
single-item issue per line group.  ACT_LIB_PATH=anonymous-credit-tokens_amd/libact_timing.so python tools/tiny_timing.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from act_amd import capi
sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, 128, max_batch=4096, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("sl-sk", 64))
pre = eng.pre_issuance_random(sh("sl-pre", 128)); req = eng.request(pre, sh("sl-rq", 128))
c = (777).to_bytes(32, "little")
for i in range(6):
    sys.stderr.write("---- issue %d (columns: start, X_A done, inverse done, role done, [finisher:] arrived, A and Y_A encoded, record written; us since the first stamp)\n" % i)
    st, resp = eng.issue(sk, req, c, sh("sl-ir", 128))
    assert st == b"\0"
