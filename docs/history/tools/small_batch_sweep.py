#!/usr/bin/env python3
"""Small-batch schedule sweep (small_impl.inc spend_small_locked): wall time of one act_verify_spend_batch call over n proofs in pinned
host memory, for sub-chunk sizes ACT_SMALL_SUB (read once per process: this script re-runs itself per setting) and with the
schedule switched off (the two-slot pipeline).  Usage: python docs/history/tools/small_batch_sweep.py [--child]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def child():
    import hashlib
    import numpy as np
    import torch
    from act_amd import capi
    sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
    ell = 2**252 + 27742317777372353535851937790883648493
    L, D, NMAX = 128, 256, 16384
    h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
    eng = capi.Engine(h, L, max_batch=int(os.environ.get("SWEEP_MAX_BATCH", "65536")), transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(sh("sw-sk", 64))
    pre = eng.pre_issuance_random(sh("sw-pre", 128 * D)); req = eng.request(pre, sh("sw-rq", 128 * D))
    st, resp = eng.issue(sk, req, b"".join((1000 + i).to_bytes(32, "little") for i in range(D)), sh("sw-ir", 128 * D))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, b"".join((i % 900).to_bytes(32, "little") for i in range(D)), sh("sw-pr", eng.prove_rng_bytes * D))
    assert st == bytes(D)
    PB = eng.proof_bytes
    hp = torch.empty((NMAX, PB), dtype=torch.uint8, pin_memory=True)
    hp.numpy()[:] = np.tile(np.frombuffer(proofs, np.uint8).reshape(D, PB), (NMAX // D, 1))
    hs = torch.zeros(NMAX, dtype=torch.uint8, pin_memory=True)
    if os.environ.get("SWEEP_OFF"):
        eng.set_small_batch_max(0)
    out = {}
    for mode, key in ((capi.TRANSCRIPT_DEVICE, "dev"), (capi.TRANSCRIPT_HOST, "host")):
        eng.set_transcript_mode(mode)
        for n in (1, 64, 256, 1024, 2048, 4096, 8192, 16384):
            ts = []
            for _ in range(6):
                t0 = time.perf_counter(); eng.verify_spend_ptr(sk, n, capi.MEM_HOST, hp.data_ptr(), hs.data_ptr()); ts.append(time.perf_counter() - t0)
            assert not hs[:n].any()
            out["%s/%d" % (key, n)] = round(1e3 * sorted(ts)[len(ts) // 2], 3)
    print(json.dumps(out))


def main():
    rows = {}
    settings = [("off (pipeline)", {"SWEEP_OFF": "1"})] + [("sub=%d" % s, {"ACT_SMALL_SUB": str(s)}) for s in (1024, 2048, 4096, 8192, 16384)]
    settings += [("sub=%d normal-prio" % s, {"ACT_SMALL_SUB": str(s), "ACT_SMALL_NORMAL_PRIO": "1"}) for s in (1024, 4096, 16384)]
    for name, env in settings:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rows[name] = json.loads(line[0]) if line else {"error": r.stderr[-400:]}
    keys = list(next(iter(rows.values())).keys())
    print("%-22s" % "ms per call" + "".join("%11s" % k for k in keys))
    for name, row in rows.items():
        print("%-22s" % name + "".join("%11s" % row.get(k, "-") for k in keys))
    print(json.dumps(rows))


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
