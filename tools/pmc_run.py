"""Small fixed workload for rocprofv3 --pmc runs: one verify batch of NB DISTINCT proofs (device transcripts), made by the
engine's own prover on the device exactly as bench.py makes its batch (no proof twice in the launch: the scalar-addressed
look-ups into the wide fixed-base tables are as cold as in the bench)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from act_amd import capi
L = 128; NB = int(os.environ.get("NB", "65536"))
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=NB, transcript=capi.TRANSCRIPT_DEVICE)
eng.set_wide_range_tables(24)      # as bench.py: 24-bit windows on h1 / h3 where the device has the room (act_ctx_create itself never widens)
eng.set_pipeline_depth(1)
sk = eng.private_key_random(bench.shake("bench-sk", 64))
dev, _ = bench.make_distinct_proofs_on_device(eng, capi, torch, np, sk, NB, L, 0, NB)
status = torch.zeros(NB, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(int(os.environ.get("REPS", "2"))):
    eng.verify_spend_dev(sk, NB, dev.data_ptr(), status.data_ptr())
torch.cuda.synchronize()
assert os.environ.get("SKIP_ASSERT") or int((status == 0).sum()) == NB
print("ok")
