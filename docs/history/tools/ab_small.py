#!/usr/bin/env python3
"""same-process-free A/B helper: median wall time of verify calls over k proofs (HBM-resident), 21 calls each; run under different env knobs"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from act_amd import capi
sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, 128, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
N = 64
sk = eng.private_key_random(sh("ms-sk", 64))
pre = eng.pre_issuance_random(sh("ms-pre", 128 * N)); req = eng.request(pre, sh("ms-rq", 128 * N))
st, resp = eng.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(N)), sh("ms-ir", 128 * N))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, prer = eng.prove_spend(tok, b"".join((i + 1).to_bytes(32, "little") for i in range(N)), sh("ms-pr", eng.prove_rng_bytes * N))
pb = eng.proof_bytes
base = torch.from_numpy(np.frombuffer(proofs, np.uint8).copy()).reshape(N, pb)
out = []
for k in [int(x) for x in os.environ.get("SIZES", "1,64,256,384").split(",")]:
    dp = base.repeat((k + N - 1) // N, 1)[:k].contiguous().cuda(); d_st = torch.zeros(k, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    ts = []
    for i in range(25):
        t0 = time.perf_counter(); eng.verify_spend_ptr(sk, k, capi.MEM_DEVICE, dp.data_ptr(), d_st.data_ptr()); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[4:]); out.append("k=%d %.3f/%.3f" % (k, 1e3 * ts[0], 1e3 * ts[len(ts) // 2]))
print(" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("ACT_")) or "default", "| min/median ms:", "  ".join(out), flush=True)
