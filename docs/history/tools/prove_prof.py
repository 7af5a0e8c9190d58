"""Per-kernel time of prove_spend / issue / refund-sign on one GPU (HIP events on the engine's streams)."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from act_amd import capi
L = 128; n = 1 << 18; D = 4096
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
if os.environ.get("DEPTH"): eng.set_pipeline_depth(int(os.environ["DEPTH"]))      # DEPTH=1: one chunk in flight, the kernels' own durations
sk = eng.private_key_random(bench.shake("sk", 64))
pre = eng.pre_issuance_random(bench.shake("pre", 128 * D)); req = eng.request(pre, bench.shake("rq", 128 * D))
cs = [(i * 2654435761) % 2**64 for i in range(D)]
st, resp = eng.issue(sk, req, b"".join(bench.scb(c) for c in cs), bench.shake("ir", 128 * D))
st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
dev = lambda b, rows: torch.from_numpy(np.frombuffer(b, np.uint8).copy().reshape(rows, -1)).cuda()
d_tok = dev(tok, D).repeat(n // D, 1).contiguous(); d_s = dev(b"".join(bench.scb(c // 3) for c in cs), D).repeat(n // D, 1).contiguous()
d_rng = torch.randint(0, 256, (n, eng.prove_rng_bytes), dtype=torch.uint8, device="cuda")
d_proof = torch.empty((n, eng.proof_bytes), dtype=torch.uint8, device="cuda"); d_pre = torch.empty((n, 96), dtype=torch.uint8, device="cuda"); d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
run = lambda: eng.prove_spend_dev(n, d_tok.data_ptr(), d_s.data_ptr(), d_rng.data_ptr(), d_proof.data_ptr(), d_pre.data_ptr(), d_st.data_ptr())
run(); torch.cuda.synchronize()
eng.prof_reset(); eng.prof_enable(True)
t = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
eng.prof_enable(False)
print("prove_spend", round(n / dt), "per s;", {k: round(v["busy_ms"], 1) for k, v in eng.prof().items()}, "total ms", round(1e3 * dt, 1), eng.fixed_base_bits())
