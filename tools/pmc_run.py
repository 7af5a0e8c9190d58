"""Small fixed workload for rocprofv3 --pmc runs: one verify batch of NB proofs (device transcripts)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from act_amd import capi
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L = 128; NB = int(os.environ.get("NB", "65536")); D = 256
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=NB, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("sk", 64))
pre = eng.pre_issuance_random(sh("pre", 128 * D)); req = eng.request(pre, sh("rq", 128 * D))
st, resp = eng.issue(sk, req, b"".join(scb(100 + i) for i in range(D)), sh("ir", 128 * D))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, b"".join(scb(i % 90) for i in range(D)), sh("pr", eng.prove_rng_bytes * D))
dev = torch.from_numpy(np.frombuffer(proofs, np.uint8).copy().reshape(D, -1)).cuda().repeat(NB // D, 1).contiguous()
status = torch.zeros(NB, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(int(os.environ.get("REPS", "2"))):
    eng.verify_spend_dev(sk, NB, dev.data_ptr(), status.data_ptr())
torch.cuda.synchronize()
assert int((status == 0).sum()) == NB
print("ok")
