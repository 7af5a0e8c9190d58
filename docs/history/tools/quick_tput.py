"""Throughput probe: tiled valid proofs (made by the engine's own prover) resident in HBM; device- and host-transcript verify."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from act_amd import capi
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L = int(os.environ.get("ACT_L", "128")); NB = int(os.environ.get("NB", "65536")); MB = int(os.environ.get("MB", "16384")); D = 256
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=MB, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("sk", 64))
pre = eng.pre_issuance_random(sh("pre", 128 * D)); req = eng.request(pre, sh("rq", 128 * D))
st, resp = eng.issue(sk, req, b"".join(scb(1000 + 7 * i) for i in range(D)), sh("ir", 128 * D))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, b"".join(scb(3 * i) for i in range(D)), sh("pr", eng.prove_rng_bytes * D))
assert st == bytes(D)
dev = torch.from_numpy(np.frombuffer(proofs, np.uint8).copy().reshape(D, -1)).cuda().repeat(NB // D, 1).contiguous()
status = torch.zeros(NB, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
eng.prof_enable(True)
for mode, name in ((capi.TRANSCRIPT_DEVICE, "device-transcript"), (capi.TRANSCRIPT_HOST, "host-transcript")):
    eng.set_transcript_mode(mode)
    eng.verify_spend_dev(sk, min(NB, MB), dev.data_ptr(), status.data_ptr())   # warm
    eng.prof_reset()
    torch.cuda.synchronize(); t = time.time()
    eng.verify_spend_dev(sk, NB, dev.data_ptr(), status.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t
    ok = int((status == 0).sum())
    print("%s: %d proofs in %.3fs -> %.0f verifies/s (ok=%d)" % (name, NB, dt, NB / dt, ok))
    for k, v in eng.prof().items(): print("   %-20s %9.2f ms  %3d launches  %.3f us/lane" % (k, v["ms"], v["launches"], 1e3 * v["ms"] / v["lanes"]))
