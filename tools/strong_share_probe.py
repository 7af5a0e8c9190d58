#!/usr/bin/env python3
"""What one GPU does with its share of ONE 2^20 batch at N = 1, 2, 4, 8 GPUs (bench.py's strong scaling: 2^20 / N proofs per rank and
step, HBM-resident, host transcripts), for host-transcript chunk sizes ACT_HOST_CHUNK (read once per process: this script re-runs
itself per setting; --one = the environment as it is, one row).  Predicts the shape of the strong-scaling curve from one GPU.
Usage: python tools/strong_share_probe.py [--one]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def child():
    import numpy as np
    import torch
    import bench
    from act_amd import capi
    L = 128
    h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
    eng = capi.Engine(h, L, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
    eng.set_wide_range_tables(24)      # as bench.py: 24-bit windows on h1 / h3 where the device has the room (act_ctx_create itself never widens)
    sk = eng.private_key_random(bench.shake("bench-sk", 64))
    n = 1 << 19
    dev, _ = bench.make_distinct_proofs_on_device(eng, capi, torch, np, sk, n, L, 0, 65536)
    status = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    out = {}
    for mode, key in ((capi.TRANSCRIPT_HOST, "host"), (capi.TRANSCRIPT_DEVICE, "dev")):
        eng.set_transcript_mode(mode)
        for share in (1 << 19, 1 << 18, 1 << 17):
            for _ in range(2):
                eng.verify_spend_dev(sk, share, dev.data_ptr(), status.data_ptr())
            steps = 6
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                eng.verify_spend_dev(sk, share, dev.data_ptr(), status.data_ptr())
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            assert int(status[:share].sum()) == 0
            out["%s/2^%d" % (key, share.bit_length() - 1)] = round(share * steps / dt)
    print(json.dumps(out))


def main():
    rows = {}
    settings = [("default", {})] + ([] if "--one" in sys.argv else [("ACT_HOST_CHUNK=%d" % c, {"ACT_HOST_CHUNK": str(c)}) for c in (16384, 32768, 65536)])
    for name, env in settings:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rows[name] = json.loads(line[0]) if line else {"error": r.stderr[-400:]}
    keys = list(next(iter(rows.values())).keys())
    print("%-24s" % "verifies/s per GPU" + "".join("%12s" % k for k in keys))
    for name, row in rows.items():
        print("%-24s" % name + "".join("%12s" % row.get(k, "-") for k in keys))
    print(json.dumps(rows))


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
