#!/usr/bin/env python3
"""One act_verify_spend_batch call over k proofs (k = 256 ... 32 768), median of 7, three ways of handing the proofs over:
   host     ACT_MEM_HOST, pinned host memory (the engine stages it: one copy, then the kernels)
   hbm      ACT_MEM_DEVICE, proofs resident in HBM
   mapped   ACT_MEM_DEVICE with the PINNED HOST pointer: the kernels read the proofs over PCIe themselves (hipHostMalloc memory is
            mapped into the device's address space), no staging copy in front of the first kernel
Statuses go to device memory in all three.  What the staging copy of a single-chunk call costs, and whether reading in place wins."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from act_amd import capi

sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
MB = int(os.environ.get("MAX_BATCH", "65536"))
tr = capi.TRANSCRIPT_HOST if os.environ.get("HOST_TR") else capi.TRANSCRIPT_DEVICE
eng = capi.Engine(h, 128, max_batch=MB, transcript=tr)
if os.environ.get("SMALL_MAX"): eng.set_small_batch_max(int(os.environ["SMALL_MAX"]))      # where the small-batch schedule ends (library default 8 192)
eng.set_wide_range_tables(24)      # as bench.py: 24-bit windows on h1 / h3 where the device has the room (act_ctx_create itself never widens)
N = 64
sk = eng.private_key_random(sh("ms-sk", 64))
pre = eng.pre_issuance_random(sh("ms-pre", 128 * N)); req = eng.request(pre, sh("ms-rq", 128 * N))
cam = b"".join((1000 + i).to_bytes(32, "little") for i in range(N))
st, resp = eng.issue(sk, req, cam, sh("ms-ir", 128 * N))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
s_b = b"".join((i + 1).to_bytes(32, "little") for i in range(N))
st, proofs, prer = eng.prove_spend(tok, s_b, sh("ms-pr", eng.prove_rng_bytes * N))
pb = eng.proof_bytes
t = bytearray(proofs); t[pb * 5 + 33] ^= 1; proofs = bytes(t)
base = torch.from_numpy(np.frombuffer(proofs, np.uint8).copy()).reshape(N, pb)
want64 = torch.from_numpy(np.frombuffer(eng.verify_spend(sk, proofs), np.uint8).copy())
sizes = [int(x) for x in os.environ.get("SIZES", "256,1024,2048,4096,8192,16384,32768").split(",")]
print("ms per call, median of 7 (k proofs/s)   [max_batch %d, %s transcripts]" % (MB, "host" if os.environ.get("HOST_TR") else "device"))
for k in sizes:
    hp = base.repeat((k + N - 1) // N, 1)[:k].contiguous().pin_memory()
    dp = hp.cuda()
    want = want64.repeat((k + N - 1) // N)[:k].cuda()
    d_st = torch.zeros(k, dtype=torch.uint8, device="cuda")
    h_st = torch.zeros(k, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize()
    row = []
    for name, mem, ptr, stp in (("host", capi.MEM_HOST, hp.data_ptr(), h_st), ("hbm", capi.MEM_DEVICE, dp.data_ptr(), d_st), ("mapped", capi.MEM_DEVICE, hp.data_ptr(), d_st)):
        ts = []
        for i in range(9):
            stp.zero_(); torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.verify_spend_ptr(sk, k, mem, ptr, stp.data_ptr()); ts.append(time.perf_counter() - t0)
            assert torch.equal(stp.cuda(), want), (name, k)
        ts = sorted(ts[2:]); m = ts[len(ts) // 2]
        row.append("%s %7.3f (%6.0f k/s)" % (name, 1e3 * m, k / m / 1e3))
    print("  k=%-6d %s" % (k, "   ".join(row)), flush=True)
