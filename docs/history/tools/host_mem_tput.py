"""PCIe-inclusive verify rate: proofs in (pageable) host memory, statuses back to host memory."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from act_amd import capi
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L, D, N = 128, 1024, 1 << 17
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=16384, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("sk", 64))
pre = eng.pre_issuance_random(sh("pre", 128 * D)); req = eng.request(pre, sh("rq", 128 * D))
st, resp = eng.issue(sk, req, scb(500) * D, sh("ir", 128 * D)); st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, scb(7) * D, sh("pr", eng.prove_rng_bytes * D))
host = np.tile(np.frombuffer(proofs, np.uint8), N // D)
for mode, name in ((capi.TRANSCRIPT_DEVICE, "device transcripts"), (capi.TRANSCRIPT_HOST, "host transcripts")):
    eng.set_transcript_mode(mode)
    eng.verify_spend(sk, host[: 16384 * eng.proof_bytes])
    t = time.perf_counter(); st = eng.verify_spend(sk, host); dt = time.perf_counter() - t
    assert st == bytes(N)
    print("host-memory inputs, %s: %d proofs in %.3f s -> %.0f verifies/s (%.2f GB/s of proofs over PCIe)" % (name, N, dt, N / dt, N * eng.proof_bytes / dt / 1e9))
