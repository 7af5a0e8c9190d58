import hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from act_amd import capi
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L, D, NB = 128, 256, 262144
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
for mb in (16384, 65536):
    eng = capi.Engine(h, L, max_batch=mb, transcript=capi.TRANSCRIPT_HOST)
    sk = eng.private_key_random(sh("sk", 64))
    pre = eng.pre_issuance_random(sh("pre", 128 * D)); req = eng.request(pre, sh("rq", 128 * D))
    st, resp = eng.issue(sk, req, scb(500) * D, sh("ir", 128 * D)); st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, scb(7) * D, sh("pr", eng.prove_rng_bytes * D))
    dev = torch.from_numpy(np.frombuffer(proofs, np.uint8).copy().reshape(D, -1)).cuda().repeat(NB // D, 1).contiguous()
    status = torch.zeros(NB, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    for nt in (0, 16, 32, 8):
        eng._ck(eng.lib.act_ctx_set_host_threads(eng.ctx, nt))
        eng.verify_spend_dev(sk, min(NB, mb), dev.data_ptr(), status.data_ptr()); torch.cuda.synchronize()
        t = time.time(); eng.verify_spend_dev(sk, NB, dev.data_ptr(), status.data_ptr()); torch.cuda.synchronize(); dt = time.time() - t
        print("MB=%d host_threads=%d: %.0f verifies/s" % (mb, nt, NB / dt))
    eng.close()
