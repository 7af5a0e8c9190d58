#!/usr/bin/env python3
"""Several callers, one GPU: T threads, each with its own context, each verifying k proofs per call from pinned host memory, calls
back to back.  One caller alone leaves the chip idle between kernels of a call and its clock low (small
calls); a server has several.  Prints the whole-GPU rate for T = 1, 2, 4, 8 and k = 1, 64, 1024, 4096."""
import hashlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from act_amd import capi

sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
L, D = 128, 64
mode = capi.TRANSCRIPT_HOST if (len(sys.argv) > 1 and sys.argv[1] == "host") else capi.TRANSCRIPT_DEVICE
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng0 = capi.Engine(h, L, max_batch=8192, transcript=mode)
sk = eng0.private_key_random(sh("sw-sk", 64))
pre = eng0.pre_issuance_random(sh("sw-pre", 128 * D)); req = eng0.request(pre, sh("sw-rq", 128 * D))
st, resp = eng0.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(D)), sh("sw-ir", 128 * D))
st, tok = eng0.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng0.prove_spend(tok, b"".join((i % 900).to_bytes(32, "little") for i in range(D)), sh("sw-pr", eng0.prove_rng_bytes * D))
PB = eng0.proof_bytes
KMAX = int(os.environ.get("KMAX", "4096"))
hp = torch.empty((KMAX, PB), dtype=torch.uint8, pin_memory=True)
hp.numpy()[:] = np.tile(np.frombuffer(proofs, np.uint8).reshape(D, PB), (KMAX // D, 1))
TMAX = max(int(x) for x in os.environ.get("TS", "1,2,4,8").split(","))
engines = [eng0] + [capi.Engine(h, L, max_batch=8192, transcript=mode) for _ in range(TMAX - 1)]
if os.environ.get("SHARE_CTX"):
    engines = [eng0] * TMAX                             # one context, callers queue on its lock
stat = [torch.zeros(KMAX, dtype=torch.uint8, pin_memory=True) for _ in range(TMAX)]
print("transcripts: %s%s" % ("host" if mode == capi.TRANSCRIPT_HOST else "device", ", ONE shared context" if os.environ.get("SHARE_CTX") else ""))
KS = [int(x) for x in os.environ.get("KS", "1,64,1024,4096").split(",")]; TS = [int(x) for x in os.environ.get("TS", "1,2,4,8").split(",")]
for k in KS:
    row = []
    for T in TS:
        calls = max(8, min(200, 40000 // max(k, 32)))
        def work(t):
            for _ in range(calls):
                engines[t].verify_spend_ptr(sk, k, capi.MEM_HOST, hp.data_ptr(), stat[t].data_ptr())
        for t in range(T):
            engines[t].verify_spend_ptr(sk, k, capi.MEM_HOST, hp.data_ptr(), stat[t].data_ptr())
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        assert all(not stat[t][:k].any() for t in range(T))
        row.append("T=%d: %8.0f verifies/s (%.2f ms per call)" % (T, T * calls * k / dt, 1e3 * dt / calls))
    print("k = %4d   " % k + "   ".join(row), flush=True)
