#!/usr/bin/env python3
"""The crate's API under a server's load: T threads share ONE node handle (what the Rust binding keeps inside `Params`) and each
calls the single-item `refund` in a loop -- act_node_verify_spend_batch over one proof, then act_node_refund_sign_batch with 128 rng
bytes (rust/src/mi355x.rs refund_batch with n = 1).  Without coalescing the calls queue on the handle; with
act_node_set_coalescing they merge.  Prints refunds/s over the GPU and the latency of one refund, L = 128."""
import ctypes as C
import hashlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
from act_amd import capi

sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
L, D = 128, 64
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=4096, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("si-sk", 64))
pre = eng.pre_issuance_random(sh("si-pre", 128 * D)); req = eng.request(pre, sh("si-rq", 128 * D))
st, resp = eng.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(D)), sh("si-ir", 128 * D))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, b"".join((i % 900).to_bytes(32, "little") for i in range(D)), sh("si-pr", eng.prove_rng_bytes * D))
PB = eng.proof_bytes
items = [np.frombuffer(proofs[PB * i:PB * (i + 1)], np.uint8).copy() for i in range(D)]
rngs = np.frombuffer(sh("si-r", 128 * D), np.uint8).reshape(D, 128).copy()
mode = capi.TRANSCRIPT_HOST if (len(sys.argv) > 1 and sys.argv[1] == "host") else capi.TRANSCRIPT_DEVICE
node = capi.Node(h, L, devices=tuple(int(x) for x in os.environ.get("DEVS", "0").split(",")), max_batch=8192, transcript=mode)
lib, nd = node.lib, node.nd
skb = (C.c_uint8 * 64).from_buffer_copy(sk)


def refund_one(i, st, kp, st2, rf):
    rc = lib.act_node_verify_spend_batch(nd, 1, skb, items[i].ctypes.data, st.ctypes.data, kp.ctypes.data)
    assert rc == 0 and st[0] == 0
    rc = lib.act_node_refund_sign_batch(nd, 1, skb, kp.ctypes.data, st.ctypes.data, rngs[i].ctypes.data, capi.RNG_SEQUENTIAL, rf.ctypes.data, st2.ctypes.data)
    assert rc == 0 and st2[0] == 0


print("transcripts: %s" % ("host" if mode == capi.TRANSCRIPT_HOST else "device"))
want = {}
for co in (0, 64):
    node.set_coalescing(co)
    row = []
    for T in [int(x) for x in os.environ.get("TS", "1,4,16,64").split(",")]:
        calls = max(6, 240 // T) if not co else 60
        lat = [0.0] * T
        outs = [None] * T

        def work(t):
            st, kp, st2, rf = np.zeros(1, np.uint8), np.zeros(32, np.uint8), np.zeros(1, np.uint8), np.zeros(128, np.uint8)
            t0 = time.perf_counter()
            for c in range(calls):
                refund_one((t + c) % D, st, kp, st2, rf)
            lat[t] = (time.perf_counter() - t0) / calls
            outs[t] = ((t + calls - 1) % D, rf.tobytes())
        work(0)
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        for i, rf in outs:                      # the same proof and rng bytes give the same refund, merged or not
            assert want.setdefault(i, rf) == rf
        row.append("T=%d: %6.0f refunds/s (%.2f ms each)" % (T, T * calls / dt, 1e3 * sum(lat) / T))
    print(("coalescing off" if not co else "coalescing %3d" % co) + "   " + "   ".join(row), flush=True)
