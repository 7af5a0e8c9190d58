#!/usr/bin/env python3
"""One GPU behind a node handle that lists it once, twice, four times (act_node_create(devices = (0,), (0, 0), (0, 0, 0, 0))): every entry
is a context with two chunks in flight, so a call cut over k entries has 2k chunks in flight.  act_node_verify_spend_batch over k proofs
in pinned host memory, host transcripts, median of 5.  Does a mid-size call gain from more chunks in flight than one context gives it?"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from act_amd import capi

sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, 128, max_batch=4096, transcript=capi.TRANSCRIPT_DEVICE)
N = 64
sk = eng.private_key_random(sh("nc-sk", 64))
pre = eng.pre_issuance_random(sh("nc-pre", 128 * N)); req = eng.request(pre, sh("nc-rq", 128 * N))
st, resp = eng.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(N)), sh("nc-ir", 128 * N))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, b"".join((i + 1).to_bytes(32, "little") for i in range(N)), sh("nc-pr", eng.prove_rng_bytes * N))
pb = eng.proof_bytes
eng.close()
base = torch.from_numpy(np.frombuffer(proofs, np.uint8).copy()).reshape(N, pb)
sizes = [int(x) for x in os.environ.get("SIZES", "16384,32768,65536,131072,262144").split(",")]
hp = base.repeat((max(sizes) + N - 1) // N, 1)[:max(sizes)].contiguous().pin_memory()
hst = torch.empty(max(sizes), dtype=torch.uint8, pin_memory=True)
print("ms per act_node_verify_spend_batch call, median of 5 (k proofs/s); host transcripts, pinned host memory")
for devs in ((0,), (0, 0), (0, 0, 0, 0)):
    node = capi.Node(h, 128, devices=devs, transcript=capi.TRANSCRIPT_HOST)
    node.set_fixed_base_bits(1, 24); node.set_fixed_base_bits(3, 24)
    row = []
    for k in sizes:
        ts = []
        for it in range(7):
            t0 = time.perf_counter()
            node.verify_spend_ptr(sk, k, hp.data_ptr(), hst.data_ptr())
            ts.append(time.perf_counter() - t0)
        assert not hst[:k].any()
        ms = 1e3 * sorted(ts[2:])[2]
        row.append("k=%d %8.2f (%4.0f k/s)" % (k, ms, k / ms))
    print("%d entr%s: " % (len(devs), "y" if len(devs) == 1 else "ies") + "   ".join(row))
    node.close()
