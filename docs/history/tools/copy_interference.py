"""Does PCIe traffic slow the verify kernels down?  Device-mode verify of 2^19 HBM-resident proofs, alone and while a
second stream copies pinned host memory to / from the device continuously (torch non_blocking copies = hipMemcpyAsync)."""
import os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from act_amd import capi
L = 128; PB = bench.proof_bytes(L); n = 1 << 19
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, max_batch=65536, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(bench.shake("bench-sk", 64))
proofs = bench.make_inputs(eng, sk, 4096)
dev = torch.from_numpy(np.frombuffer(proofs, np.uint8).reshape(4096, PB).copy()).cuda().repeat(n // 4096, 1).contiguous()
st = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def run():
    eng.verify_spend_dev(sk, n, dev.data_ptr(), st.data_ptr()); torch.cuda.synchronize()
    t = time.perf_counter(); eng.verify_spend_dev(sk, n, dev.data_ptr(), st.data_ptr()); torch.cuda.synchronize(); return n / (time.perf_counter() - t)
print("alone", round(run()), flush=True)
hbuf = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True); dbuf = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for mode in ("h2d", "d2h", "both"):
    stop = False; moved = [0]
    def pump():
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        while not stop:
            if mode in ("h2d", "both"):
                with torch.cuda.stream(s1): dbuf.copy_(hbuf, non_blocking=True)
            if mode in ("d2h", "both"):
                with torch.cuda.stream(s2): hbuf.copy_(dbuf, non_blocking=True)
            s1.synchronize(); s2.synchronize(); moved[0] += 1
    th = threading.Thread(target=pump); th.start()
    t0 = time.perf_counter(); r = run(); dt = time.perf_counter() - t0
    stop = True; th.join()
    print(mode, round(r), "copies GB/s per direction ~", round(moved[0] * 1.07 / dt, 1), flush=True)
