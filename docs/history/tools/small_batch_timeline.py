#!/usr/bin/env python3
"""Per-kernel timeline (HIP events on the engine's streams, ACT_TIMELINE_FILE) of ONE small-batch call: where the latency of
act_verify_spend_batch over n proofs goes.  Usage: python docs/history/tools/small_batch_timeline.py n [host|dev] [hbm]   (hbm: proofs resident
in device memory instead of pinned host memory; long timelines print their head, tail and the gaps between range kernels)"""
import hashlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
tl = tempfile.mktemp(suffix=".csv")
os.environ["ACT_TIMELINE_FILE"] = tl
import numpy as np
import torch
from act_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = capi.TRANSCRIPT_HOST if (len(sys.argv) > 2 and sys.argv[2] == "host") else capi.TRANSCRIPT_DEVICE
sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
L, D = 128, 64
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=mode)
sk = eng.private_key_random(sh("sw-sk", 64))
pre = eng.pre_issuance_random(sh("sw-pre", 128 * D)); req = eng.request(pre, sh("sw-rq", 128 * D))
st, resp = eng.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(D)), sh("sw-ir", 128 * D))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, b"".join((i % 900).to_bytes(32, "little") for i in range(D)), sh("sw-pr", eng.prove_rng_bytes * D))
PB = eng.proof_bytes
hp = torch.empty((max(n, D), PB), dtype=torch.uint8, pin_memory=True)
hp.numpy()[:] = np.tile(np.frombuffer(proofs, np.uint8).reshape(D, PB), ((max(n, D) + D - 1) // D, 1))[:max(n, D)]
hs = torch.zeros(max(n, D), dtype=torch.uint8, pin_memory=True)
import time
hbm = len(sys.argv) > 3 and sys.argv[3] == "hbm"
refund = "refund" in sys.argv            # verify + sign (act_refund_batch), device memory
if refund:
    d_p = hp.cuda(); d_s = torch.zeros(max(n, D), dtype=torch.uint8, device="cuda"); d_r = torch.randint(0, 256, (max(n, D), 128), dtype=torch.uint8, device="cuda")
    d_o = torch.zeros((max(n, D), 128), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    call = lambda: eng.refund_dev(sk, n, d_p.data_ptr(), d_r.data_ptr(), capi.RNG_PER_LANE, d_o.data_ptr(), d_s.data_ptr())
elif hbm:
    d_p = hp.cuda(); d_s = torch.zeros(max(n, D), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    call = lambda: eng.verify_spend_dev(sk, n, d_p.data_ptr(), d_s.data_ptr())
else:
    call = lambda: eng.verify_spend_ptr(sk, n, capi.MEM_HOST, hp.data_ptr(), hs.data_ptr())
for _ in range(3):
    call()
eng.prof_reset(); eng.prof_enable(True)
open(tl, "w").close()
t0 = time.perf_counter(); call(); wall = time.perf_counter() - t0
eng.prof_enable(False)
rows = [l.strip().split(",") for l in open(tl) if l.strip()]
rows = [(r[0], float(r[2]), float(r[3]), int(r[4])) for r in rows]
base = min(r[1] for r in rows)
print("n = %d, %s transcripts: wall %.3f ms (with event overhead)" % (n, "host" if mode == capi.TRANSCRIPT_HOST else "device", 1e3 * wall))
rows = sorted(rows, key=lambda r: r[1])
show = rows if len(rows) <= 60 else rows[:24] + [None] + rows[-24:]
for r in show:
    if r is None:
        print("  ...")
        continue
    name, a, b, lanes = r
    print("  %-24s %8.3f -> %8.3f  (%7.3f ms)  lanes %d" % (name, a - base, b - base, b - a, lanes))
bits = [(a - base, b - base) for name, a, b, lanes in rows if name == "k_spend_bits"]
if len(bits) > 1:
    end = max(b for name, a, b, lanes in rows) - base
    gaps = [round(bits[i + 1][0] - bits[i][1], 3) for i in range(len(bits) - 1)]
    print("range kernels: first starts at %.3f ms, last ends at %.3f ms, call's last kernel ends at %.3f ms; gaps between consecutive range kernels (ms): %s"
          % (bits[0][0], bits[-1][1], end, gaps))
