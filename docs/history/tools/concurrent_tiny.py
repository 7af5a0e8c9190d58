#!/usr/bin/env python3
"""T threads, a context each (or one shared node handle), one-proof verify calls back to back: the whole-GPU rate of tiny calls from
several callers, against the hardware-queue count HIP maps a process's streams onto (GPU_MAX_HW_QUEUES, default 4).
    python docs/history/tools/concurrent_tiny.py                  # sweeps queues x threads in child processes
    python docs/history/tools/concurrent_tiny.py child T calls    # one measurement in this process"""
import hashlib
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(T, calls, what):
    from act_amd import capi
    sh = lambda l, k: hashlib.shake_256(l.encode()).digest(k)
    h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
    engs = [capi.Engine(h, 128, max_batch=256, transcript=capi.TRANSCRIPT_DEVICE) for _ in range(T)]
    eng = engs[0]
    sk = eng.private_key_random(sh("ct-sk", 64))
    pre = eng.pre_issuance_random(sh("ct-pre", 128)); req = eng.request(pre, sh("ct-rq", 128))
    st, resp = eng.issue(sk, req, (5).to_bytes(32, "little"), sh("ct-ir", 128))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proof, prer = eng.prove_spend(tok, (3).to_bytes(32, "little"), sh("ct-pr", eng.prove_rng_bytes))
    f = {"verify": lambda e: e.verify_spend(sk, proof),
         "request": lambda e: e.request(pre, sh("ct-rq", 128)),
         "issue": lambda e: e.issue(sk, req, (5).to_bytes(32, "little"), sh("ct-ir", 128)),
         "refund": lambda e: e.refund(sk, proof, sh("ct-rr", 128))}[what]
    for e in engs:
        f(e); f(e)

    def work(e):
        for _ in range(calls):
            f(e)
    th = [threading.Thread(target=work, args=(e,)) for e in engs]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    print("  %-8s queues=%-3s threads=%d: %7.0f calls/s   %.2f ms per call" % (what, os.environ.get("GPU_MAX_HW_QUEUES", "4*"), T, T * calls / dt, 1e3 * dt / calls), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    for what in ("verify", "request", "issue"):
        for q in ("4", "8", "16", "32"):
            for T in (1, 2, 4, 8):
                env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(T), "200", what], env=env, check=False)
