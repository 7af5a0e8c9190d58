import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from act_amd import capi
L = 128; PB = bench.proof_bytes(L); n = 1 << 18
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=int(os.environ.get("TR", "1")))
sk = eng.private_key_random(bench.shake("bench-sk", 64))
proofs = bench.make_inputs(eng, sk, 4096)
hp = torch.from_numpy(np.frombuffer(proofs, np.uint8).reshape(4096, PB).copy()).repeat(n // 4096, 1).contiguous().pin_memory()
hs = torch.zeros(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter(); eng.verify_spend_ptr(sk, n, capi.MEM_HOST, hp.data_ptr(), hs.data_ptr()); dt = time.perf_counter() - t
print("hostmem rate", round(n / dt), flush=True)
