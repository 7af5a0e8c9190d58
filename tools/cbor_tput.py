"""C-ABI-only timing of the batch CBOR codec (SpendProof, L = 128): host memory and device memory, no Python copies in the timed region."""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from act_amd import capi
sh = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
ELL = 2**252 + 27742317777372353535851937790883648493
scb = lambda v: (v % ELL).to_bytes(32, "little")
L, D, N = 128, 512, 1 << 16
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, transcript=capi.TRANSCRIPT_DEVICE)
sk = eng.private_key_random(sh("sk", 64))
pre = eng.pre_issuance_random(sh("pre", 128 * D)); req = eng.request(pre, sh("rq", 128 * D))
st, resp = eng.issue(sk, req, scb(500) * D, sh("ir", 128 * D)); st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
st, proofs, _ = eng.prove_spend(tok, scb(7) * D, sh("pr", eng.prove_rng_bytes * D))
T = capi.CBOR_TYPES["SpendProof"]; lib, ctx = eng.lib, eng.ctx
rb = lib.act_cbor_record_bytes(ctx, T); ml = lib.act_cbor_size(ctx, T)
recs = np.tile(np.frombuffer(proofs, np.uint8), N // D)
wire = np.zeros(ml * N, np.uint8); back = np.zeros(rb * N, np.uint8); stt = np.zeros(N, np.uint8)
offs = (np.arange(N + 1, dtype=np.uint64) * ml)
def timed(f):
    f(); torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t
dt = timed(lambda: eng._ck(lib.act_cbor_encode_batch(ctx, T, N, 0, recs.ctypes.data, wire.ctypes.data)))
print("host memory  encode: %.0f msgs/s (%.2f GB/s of wire bytes)" % (N / dt, N * ml / dt / 1e9))
dt = timed(lambda: eng._ck(lib.act_cbor_decode_batch(ctx, T, N, 0, wire.ctypes.data, offs.ctypes.data, back.ctypes.data, stt.ctypes.data)))
assert bytes(stt) == bytes(N) and np.array_equal(back, recs)
print("host memory  decode: %.0f msgs/s" % (N / dt))
d_recs = torch.from_numpy(recs).cuda(); d_wire = torch.zeros(ml * N, dtype=torch.uint8, device="cuda"); d_back = torch.zeros(rb * N, dtype=torch.uint8, device="cuda")
d_st = torch.zeros(N, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
dt = timed(lambda: eng._ck(lib.act_cbor_encode_batch(ctx, T, N, 1, d_recs.data_ptr(), d_wire.data_ptr())))
print("device memory encode: %.0f msgs/s" % (N / dt))
try:
    dt = timed(lambda: eng._ck(lib.act_cbor_decode_batch(ctx, T, N, 1, d_wire.data_ptr(), offs.ctypes.data, d_back.data_ptr(), d_st.data_ptr())))
    assert int((d_st == 0).sum()) == N and torch.equal(d_back, d_recs)
    print("device memory decode: %.0f msgs/s" % (N / dt))
except Exception as e:
    print("device memory decode:", e)
