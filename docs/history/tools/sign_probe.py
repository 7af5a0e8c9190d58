#!/usr/bin/env python3
"""Latency of the issuer's signature phase alone (act_refund_sign_batch over n accepted lanes, device memory) -- the one-lane-per-
signature kernel against the eight-lanes-per-signature one used for short launches (ACT_NO_WIDE_SIGN=1 forces the former)."""
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
eng = capi.Engine(h, 8, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("sp-sk", 64))
# any valid points do as K': the public key bytes repeated
N = 65536
kp = torch.from_numpy(np.tile(np.frombuffer(sk[32:], np.uint8), (N, 1)).copy()).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(3)
rng = torch.randint(0, 256, (N, 128), dtype=torch.uint8, device="cuda", generator=g)
sin = torch.zeros(N, dtype=torch.uint8, device="cuda"); st = torch.zeros(N, dtype=torch.uint8, device="cuda"); out = torch.zeros((N, 128), dtype=torch.uint8, device="cuda")
import ctypes as C
skb = (C.c_uint8 * 64).from_buffer_copy(sk)
torch.cuda.synchronize()
print("ACT_NO_WIDE_SIGN" in os.environ and "one lane per signature" or "eight lanes per signature up to 8 192")
row = []
digests = []
for n in (1, 64, 1024, 2048, 4096, 8192, 16384, 24576, 32768, 49152, 65536):
    best = 1e9
    for _ in range(9):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = eng.lib.act_refund_sign_batch(eng.ctx, n, capi.MEM_DEVICE, skb, kp.data_ptr(), sin.data_ptr(), rng.data_ptr(), capi.RNG_PER_LANE, out.data_ptr(), st.data_ptr())
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        assert rc == 0
    assert not st[:n].any()
    row.append("n=%d: %.2f ms" % (n, 1e3 * best)); digests.append(hashlib.sha256(out[:n].cpu().numpy().tobytes()).hexdigest()[:8])
print("   ".join(row))
print("digests of the signatures per n (equal in both runs):", " ".join(digests))
