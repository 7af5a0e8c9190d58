#!/usr/bin/env python3
"""How long is ONE ROUND of the range kernel (131 072 lanes = two wavefronts on every SIMD) as a function of the launch size?
k_spend_bits timed with HIP events for launches of 1 024 ... 65 536 proofs (device-memory input, pipelined schedule, one chunk in
flight), repeated back to back so that the clock is where a sustained load leaves it."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from act_amd import capi

sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
L, D = 128, 256
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
eng.set_small_batch_max(0); eng.set_pipeline_depth(1)
sk = eng.private_key_random(sh("sw-sk", 64))
pre = eng.pre_issuance_random(sh("sw-pre", 128 * D)); req = eng.request(pre, sh("sw-rq", 128 * D))
st, resp = eng.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(D)), sh("sw-ir", 128 * D))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, b"".join((i % 900).to_bytes(32, "little") for i in range(D)), sh("sw-pr", eng.prove_rng_bytes * D))
PB = eng.proof_bytes
dev = torch.from_numpy(np.frombuffer(proofs, np.uint8).reshape(D, PB).copy()).cuda().repeat(65536 // D, 1).contiguous()
status = torch.zeros(65536, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for n in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    reps = max(3, 32768 // n)
    for _ in range(2):
        eng.verify_spend_dev(sk, n, dev.data_ptr(), status.data_ptr())
    eng.prof_reset(); eng.prof_enable(True)
    for _ in range(reps):
        eng.verify_spend_dev(sk, n, dev.data_ptr(), status.data_ptr())
    eng.prof_enable(False)
    p = eng.prof()
    b = p["k_spend_bits"]
    ms = b["ms"] / b["launches"]
    print("n = %6d: k_spend_bits %8.3f ms per launch = %6.3f ms per 1024 proofs (%d launches); prep %.3f tail %.3f enc %.3f hash %.3f"
          % (n, ms, ms / (n / 1024), b["launches"], p["k_spend_prep"]["ms"] / b["launches"], p["k_spend_tail"]["ms"] / b["launches"],
             p["k_spend_enc"]["ms"] / b["launches"], p["k_hash_xof(spend)"]["ms"] / b["launches"]))
