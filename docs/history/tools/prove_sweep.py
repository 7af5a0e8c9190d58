import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from act_amd import capi
L = 128; n = 1 << 18
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(bench.shake("bench-sk", 64))
t0 = time.perf_counter()
dev, t_prove = bench.make_distinct_proofs_on_device(eng, capi, torch, np, sk, n, L, 0)
dev, t_prove = bench.make_distinct_proofs_on_device(eng, capi, torch, np, sk, n, L, 1)
print(os.environ.get("ACT_FB_WIDE_BITS"), os.environ.get("ACT_FB_ALL_WIDE"), eng.fixed_base_bits(), "prove_spend/s", round(n / t_prove))
