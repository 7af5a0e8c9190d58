#!/usr/bin/env python3
"""One context, max_batch 65 536, device transcripts: wall time of ONE call over n proofs from pinned host memory for the calls that
move proofs over PCIe -- prove_spend (proofs out), refund (proofs in), refund_to_credit_token (proofs in) -- and the same calls on
device memory.  Run with ACT_NO_TAPER=1 for the one-chunk schedules (same-box A/B of the host-memory chunk schedules)."""
import ctypes as C
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
L, D = 128, 64
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("hc-sk", 64))
pre = eng.pre_issuance_random(sh("hc-pre", 128 * D)); req = eng.request(pre, sh("hc-rq", 128 * D))
st, resp = eng.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(D)), sh("hc-ir", 128 * D))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
pb = eng.proof_bytes
pin = lambda *shape: torch.empty(shape, dtype=torch.uint8, pin_memory=True)
skb = (C.c_uint8 * 64).from_buffer_copy(sk); wb = (C.c_uint8 * 32).from_buffer_copy(sk[32:]); seed = (C.c_uint8 * 32).from_buffer_copy(sh("hc-seed", 32))
lib, ctx, ck = eng.lib, eng.ctx, eng._ck
print("ACT_NO_TAPER" in os.environ and "one-chunk schedules (ACT_NO_TAPER)" or "host-memory chunk schedules")
for n in [int(x) for x in os.environ.get("NS", "16384,65536,131072").split(",")]:
    rep = (n + D - 1) // D
    h_tok = pin(n, 160); h_tok.numpy()[:] = np.tile(np.frombuffer(tok, np.uint8).reshape(D, 160), (rep, 1))[:n]
    h_s = pin(n, 32); h_s.numpy()[:] = np.tile(np.frombuffer(b"".join((i % 900).to_bytes(32, "little") for i in range(D)), np.uint8).reshape(D, 32), (rep, 1))[:n]
    h_proof, h_prer, h_st, h_rf, h_tok2 = pin(n, pb), pin(n, 96), pin(n), pin(n, 128), pin(n, 160)
    h_r = pin(n, 128); h_r.numpy()[:] = np.frombuffer(hashlib.shake_256(b"hc-r").digest(128 * 1024), np.uint8).reshape(1024, 128)[np.arange(n) % 1024]
    d = {k: v.cuda() for k, v in dict(tok=h_tok, s=h_s, proof=h_proof, prer=h_prer, st=h_st, rf=h_rf, tok2=h_tok2, r=h_r).items()}
    p = lambda t: t.data_ptr()
    calls = [
        ("prove_spend (seeded)", lambda m, b: lib.act_prove_spend_seeded_batch(ctx, n, m, p(b["tok"]), p(b["s"]), seed, C.c_uint64(0), p(b["proof"]), p(b["prer"]), p(b["st"]))),
        ("refund", lambda m, b: lib.act_refund_batch(ctx, n, m, skb, p(b["proof"]), p(b["r"]), capi.RNG_PER_LANE, p(b["rf"]), p(b["st"]))),
        ("refund_to_credit_token", lambda m, b: lib.act_refund_to_credit_token_batch(ctx, n, m, p(b["prer"]), p(b["proof"]), p(b["rf"]), wb, p(b["tok2"]), p(b["st"]))),
    ]
    hb = dict(tok=h_tok, s=h_s, proof=h_proof, prer=h_prer, st=h_st, rf=h_rf, tok2=h_tok2, r=h_r)
    row = []
    for name, f in calls:
        ts = {}
        for label, m, b in (("host", capi.MEM_HOST, hb), ("device", capi.MEM_DEVICE, d)):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t = time.perf_counter(); ck(f(m, b)); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            ts[label] = best
        assert not h_st.any() and not d["st"].any()
        row.append("%s: host %.1f ms, device %.1f ms" % (name, 1e3 * ts["host"], 1e3 * ts["device"]))
    assert torch.equal(d["tok2"].cpu(), h_tok2)
    print("n = %6d   " % n + "   ".join(row), flush=True)
