"""Throwaway throughput probe: tiled valid proofs resident in HBM, device-transcript and host-transcript verify."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from act_amd import capi
from oracle_c import Oracle
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L = int(os.environ.get("ACT_L", "128")); NB = int(os.environ.get("NB", "65536")); MB = int(os.environ.get("MB", "32768"))
o = Oracle(); hp = o.params_new("bench-org", "bench-service", "bench-env", "2024-01-01"); octx = o.ctx(hp, L)
sk = octx.private_key_random(sh("sk", 64))
D = 256
pre = b"".join(octx.pre_issuance_random(sh("pre%d" % i, 128)) for i in range(D))
req = octx.request_batch(pre, sh("rq", 128 * D), 64)
cam = b"".join(scb(1000 + 7 * i) for i in range(D))
st, resp = octx.issue_batch(sk, req, cam, sh("ir", 128 * D), 64)
toks = b"".join(octx.issuance_to_credit_token(pre[64*i:64*i+64], sk[32:], req[128*i:128*i+128], resp[160*i:160*i+160])[1] for i in range(D))
t = time.time(); proofs, _ = octx.prove_spend_batch(toks, b"".join(scb(3 * i) for i in range(D)), sh("pr", octx.prove_rng_bytes * D), 128); print("oracle prove %d: %.2fs" % (D, time.time() - t))
t = time.time(); st = octx.verify_spend_batch(sk, proofs, 128); dt = time.time() - t; print("oracle verify %d on 128 threads: %.3fs -> %.0f/s; ok=%d" % (D, dt, D / dt, st.count(0)))
t = time.time(); st = octx.verify_spend_batch(sk, proofs[:octx.proof_bytes * 8], 1); dt = time.time() - t; print("oracle verify 1 thread: %.1f ms/proof" % (dt / 8 * 1e3))
pb = octx.proof_bytes
host = np.frombuffer(proofs, np.uint8)
dev = torch.from_numpy(np.tile(host, NB // D)).cuda()
status = torch.zeros(NB, dtype=torch.uint8, device="cuda")
eng = capi.Engine(hp, L, max_batch=MB, transcript=capi.TRANSCRIPT_DEVICE)
torch.cuda.synchronize()
eng.prof_enable(True)
for mode, name in ((capi.TRANSCRIPT_DEVICE, "device-transcript"), (capi.TRANSCRIPT_HOST, "host-transcript")):
    eng.set_transcript_mode(mode)
    eng.verify_spend_dev(sk, min(NB, MB), dev.data_ptr(), status.data_ptr())   # warm
    eng.prof_reset()
    torch.cuda.synchronize(); t = time.time()
    eng.verify_spend_dev(sk, NB, dev.data_ptr(), status.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t
    ok = int((status == 0).sum())
    print("%s: %d proofs in %.3fs -> %.0f verifies/s (ok=%d)" % (name, NB, dt, NB / dt, ok))
    for k, v in eng.prof().items(): print("   %-20s %9.2f ms  %3d launches  %.3f us/lane" % (k, v["ms"], v["launches"], 1e3 * v["ms"] / v["lanes"]))
