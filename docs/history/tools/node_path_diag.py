"""Where does the host-memory path lose time?  2^LOG2 distinct proofs in pinned (or pageable: PAGEABLE=1) host memory through
act_verify_spend_batch(ACT_MEM_HOST) with the engine's own HIP-event timeline (ACT_TIMELINE_FILE: kernels AND the bulk copies),
TR = 0 host transcripts (default) / 1 device transcripts.  Prints the rate, per-kernel busy time, and per-chunk gaps."""
import os, sys, time, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
tl = os.environ.setdefault("ACT_TIMELINE_FILE", "/tmp/act_timeline_%d.csv" % os.getpid())
import numpy as np, torch
import bench
from act_amd import capi
L = 128; PB = bench.proof_bytes(L); n = 1 << int(os.environ.get("LOG2", "19")); tr = int(os.environ.get("TR", "0"))
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=tr)
sk = eng.private_key_random(bench.shake("bench-sk", 64))
dev, _ = bench.make_distinct_proofs_on_device(eng, capi, torch, np, sk, n, L, 0)
if os.environ.get("PAGEABLE"):
    hp = dev.cpu().numpy().copy(); hs = np.zeros(n, np.uint8); pp, ps = hp.ctypes.data, hs.ctypes.data
else:
    hp = torch.empty((n, PB), dtype=torch.uint8, pin_memory=True); hp.copy_(dev); hs = torch.zeros(n, dtype=torch.uint8, pin_memory=True); pp, ps = hp.data_ptr(), hs.data_ptr()
del dev
torch.cuda.synchronize()
if os.environ.get("NODE"):              # the same call through a node handle over [device 0] (a second context on the GPU, sharing its tables)
    import ctypes
    node = capi.Node(h, L, devices=(0,), max_batch=65536, transcript=tr)
    e2 = capi.Engine.__new__(capi.Engine); e2.lib = node.lib; e2.ctx = ctypes.c_void_p(node.lib.act_node_ctx(node.nd, 0))
    class _E:                            # verify through the node entry point, profile through its context
        prof_reset = e2.prof_reset; prof_enable = e2.prof_enable; prof = e2.prof
        def verify_spend_ptr(self, sk, n, mem, pp, ps): node.verify_spend_ptr(sk, n, pp, ps)
    e2.close = lambda: None
    eng = _E()
eng.verify_spend_ptr(sk, n, capi.MEM_HOST, pp, ps)
open(tl, "w").close()
eng.prof_reset(); eng.prof_enable(True)
t = time.perf_counter(); eng.verify_spend_ptr(sk, n, capi.MEM_HOST, pp, ps); dt = time.perf_counter() - t
eng.prof_enable(False)
print("rate %.0f verifies/s, %.1f ms, %.2f GB/s of proofs" % (n / dt, 1e3 * dt, n * PB / dt / 1e9), flush=True)
for k, v in eng.prof().items():
    print("  %-24s busy %8.1f ms  sum %8.1f ms  launches %d" % (k, v["busy_ms"], v["ms"], v["launches"]))
rows = [l.strip().split(",") for l in open(tl) if l.strip()]
ev = [(r[0], int(r[1]), float(r[2]), float(r[3])) for r in rows]
bits = sorted([e for e in ev if e[0] == "k_spend_bits"], key=lambda e: e[2])
print("k_spend_bits launches: start, end, gap to the previous one's end")
prev = None
for e in bits:
    print("   slot %d  %9.1f -> %9.1f  (%6.1f ms)  gap %6.1f" % (e[1], e[2], e[3], e[3] - e[2], (e[2] - prev) if prev is not None else 0.0)); prev = e[3]
h2d = sorted([e for e in ev if e[0].startswith("copy_h2d")], key=lambda e: e[2])
print("bulk H2D copies: start, end, ms, GB/s")
for e in h2d[:40]:
    print("   slot %d  %9.1f -> %9.1f  (%6.1f ms)" % (e[1], e[2], e[3], e[3] - e[2]))
d2h = sorted([e for e in ev if e[0].startswith("copy_d2h")], key=lambda e: e[2])
if d2h:
    print("transcript D2H pieces: %d, total %.1f ms, first %.1f last end %.1f" % (len(d2h), sum(e[3] - e[2] for e in d2h), d2h[0][2], d2h[-1][3]))
