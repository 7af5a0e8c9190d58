"""C-ABI-only timing of the batch CBOR codec and of the fused wire-to-verdict call (SpendProof, L = 128), no Python copies in
the timed regions; prints one JSON object (docs/history/profiles/r03_cbor_c_abi.json).
  encode / decode          act_cbor_encode_batch / act_cbor_decode_batch, host memory and device memory
  records_from_host        act_verify_spend_batch over raw records in pinned host memory (the reference point)
  wire_to_status           act_verify_spend_cbor_batch over the same proofs as CBOR messages in pinned host memory
  decode_then_verify       the two unfused calls one after the other (records through host memory in between)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from act_amd import capi
L = 128; N = 1 << int(os.environ.get("LOG2", "18")); PB = bench.proof_bytes(L)
h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
eng = capi.Engine(h, L, transcript=int(os.environ.get("TR", "0")))
sk = eng.private_key_random(bench.shake("bench-sk", 64))
dev, _ = bench.make_distinct_proofs_on_device(eng, capi, torch, np, sk, N, L, 0)
expect, _ = bench.tamper(torch, dev, N)
exp_host = expect.cpu().numpy()
T = capi.CBOR_TYPES["SpendProof"]; lib, ctx = eng.lib, eng.ctx
rb = lib.act_cbor_record_bytes(ctx, T); ml = lib.act_cbor_size(ctx, T)
assert rb == PB
pin = lambda *shape: torch.empty(shape, dtype=torch.uint8, pin_memory=True)
recs = pin(N, PB); recs.copy_(dev); torch.cuda.synchronize()
wire = pin(N * ml); back = pin(N * rb); stt = pin(N); st2 = pin(N)
def timed(f):
    f(); torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t
out = {"proofs": N, "message_bytes": ml, "record_bytes": rb, "transcripts": "host" if os.environ.get("TR", "0") == "0" else "device"}
ck = eng._ck
dt = timed(lambda: ck(lib.act_cbor_encode_batch(ctx, T, N, 0, recs.data_ptr(), wire.data_ptr())))
out["encode_host_memory_msgs_per_s"] = N / dt
dt = timed(lambda: ck(lib.act_cbor_decode_batch(ctx, T, N, 0, wire.data_ptr(), None, back.data_ptr(), stt.data_ptr())))
out["decode_host_memory_msgs_per_s"] = N / dt
d_wire = torch.zeros(ml * N, dtype=torch.uint8, device="cuda"); d_back = torch.zeros(rb * N, dtype=torch.uint8, device="cuda"); d_st = torch.zeros(N, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
dt = timed(lambda: ck(lib.act_cbor_encode_batch(ctx, T, N, 1, dev.data_ptr(), d_wire.data_ptr())))
out["encode_device_memory_msgs_per_s"] = N / dt
dt = timed(lambda: ck(lib.act_cbor_decode_batch(ctx, T, N, 1, d_wire.data_ptr(), None, d_back.data_ptr(), d_st.data_ptr())))
out["decode_device_memory_msgs_per_s"] = N / dt
del d_wire, d_back
# verification: records vs wire bytes, both from pinned host memory
dt = timed(lambda: eng.verify_spend_ptr(sk, N, capi.MEM_HOST, recs.data_ptr(), stt.data_ptr()))
assert np.array_equal(stt.numpy(), exp_host)
out["records_from_host_verifies_per_s"] = N / dt
dt = timed(lambda: eng.verify_spend_cbor_ptr(sk, N, capi.MEM_HOST, wire.data_ptr(), 0, st2.data_ptr()))
assert np.array_equal(st2.numpy(), exp_host), "fused statuses differ"
out["wire_to_status_verifies_per_s"] = N / dt
out["wire_to_status_over_records"] = out["wire_to_status_verifies_per_s"] / out["records_from_host_verifies_per_s"]
def unfused():
    ck(lib.act_cbor_decode_batch(ctx, T, N, 0, wire.data_ptr(), None, back.data_ptr(), stt.data_ptr()))
    eng.verify_spend_ptr(sk, N, capi.MEM_HOST, back.data_ptr(), st2.data_ptr())
dt = timed(unfused)
out["decode_then_verify_verifies_per_s"] = N / dt
# lanes whose proof carries an undecodable / identity point come back 255 / 6 from both; tampered scalars 7
out["statuses_seen"] = sorted(set(int(x) for x in st2.numpy()))
print(json.dumps(out, indent=1))
