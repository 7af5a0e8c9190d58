#!/usr/bin/env python3
"""Measures BASELINE.json's other configs on ONE MI355X (inputs resident in HBM, device transcripts unless noted):
  config 2: 2^16 spend-proof verifies at L = 64        config 3: 2^N prove_spend at L = 128
  config 4: issue + refund (verify + sign), per-GPU share of the 8-GPU batch     (+ request, client verifiers)
Prints one JSON object; the numbers quoted in DESIGN.md come from this script (profiles/)."""
import argparse, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from act_amd import capi
ELL = 2**252 + 27742317777372353535851937790883648493
shake = lambda l, n: hashlib.shake_256(l.encode()).digest(n)
scb = lambda v: (v % ELL).to_bytes(32, "little")


def dev_bytes(b, rows):
    return torch.from_numpy(np.frombuffer(b, np.uint8).copy().reshape(rows, -1)).cuda()


def timed(fn, reps=2):
    torch.cuda.synchronize()          # inputs produced on torch's stream must be complete: the engine uses its own streams
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prove-log2", type=int, default=18)
    ap.add_argument("--issue-log2", type=int, default=19)
    ap.add_argument("--refund-log2", type=int, default=18)
    ap.add_argument("--lifecycle-log2", type=int, default=18)
    ap.add_argument("--max-batch", type=int, default=65536)
    a = ap.parse_args()
    out = {}
    h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
    D = 4096
    for L in (64, 128):
        eng = capi.Engine(h, L, max_batch=a.max_batch, transcript=capi.TRANSCRIPT_DEVICE)
        eng.set_wide_range_tables(24)      # as bench.py: 24-bit windows on h1 / h3 where the device has the room (act_ctx_create itself never widens)
        sk = eng.private_key_random(shake("sk", 64))
        pre = eng.pre_issuance_random(shake("pre", 128 * D)); req = eng.request(pre, shake("rq", 128 * D))
        cs = [(i * 2654435761) % (2**min(L, 64)) for i in range(D)]
        st, resp = eng.issue(sk, req, b"".join(scb(c) for c in cs), shake("ir", 128 * D))
        st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
        ss = [c // 3 for c in cs]
        st, proofs, prer = eng.prove_spend(tok, b"".join(scb(s) for s in ss), shake("pr", eng.prove_rng_bytes * D))
        assert st == bytes(D)
        pb = eng.proof_bytes
        if L == 64:
            n = 1 << 16
            dev = dev_bytes(proofs, D).repeat(n // D, 1).contiguous(); status = torch.zeros(n, dtype=torch.uint8, device="cuda")
            dt = timed(lambda: eng.verify_spend_dev(sk, n, dev.data_ptr(), status.data_ptr()), 3)
            assert int((status == 0).sum()) == n
            out["config2_verify_L64_2^16"] = {"verifies_per_s": n / dt, "ms": 1e3 * dt}
            del dev
            eng.close(); continue
        # config 3: prove_spend, L = 128; rng and tokens resident in HBM
        n = 1 << a.prove_log2
        d_tok = dev_bytes(tok, D).repeat(n // D, 1).contiguous(); d_s = dev_bytes(b"".join(scb(s) for s in ss), D).repeat(n // D, 1).contiguous()
        d_rng = torch.randint(0, 256, (n, eng.prove_rng_bytes), dtype=torch.uint8, device="cuda")
        d_proof = torch.empty((n, pb), dtype=torch.uint8, device="cuda"); d_pre = torch.empty((n, 96), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dt = timed(lambda: eng.prove_spend_dev(n, d_tok.data_ptr(), d_s.data_ptr(), d_rng.data_ptr(), d_proof.data_ptr(), d_pre.data_ptr(), d_st.data_ptr()), 1)
        out["config3_prove_spend_L128_2^%d" % a.prove_log2] = {"proofs_per_s": n / dt, "ms": 1e3 * dt}
        # the fresh proofs verify
        nv = min(n, 1 << 16); status = torch.zeros(nv, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        eng.verify_spend_dev(sk, nv, d_proof.data_ptr(), status.data_ptr()); torch.cuda.synchronize()
        assert int((status == 0).sum()) == nv, "GPU-made proofs must verify"
        del d_rng, d_tok, d_s
        # config 4: refund (verify + sign) and issue
        n = 1 << a.refund_log2
        d_rr = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda"); d_out = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dt = timed(lambda: eng.refund_dev(sk, n, d_proof.data_ptr(), d_rr.data_ptr(), capi.RNG_PER_LANE, d_out.data_ptr(), d_st.data_ptr()), 1)
        assert int((d_st == 0).sum()) == n
        out["config4_refund_L128_2^%d" % a.refund_log2] = {"refunds_per_s": n / dt, "ms": 1e3 * dt}
        # the whole redemption step: verify -> nullifier look-up and record -> sign (act_redeem_batch).  The proofs are tiled, so
        # all but the first copy of each are double spends: the look-up and the record are exercised, 1 / (n / D) of the lanes are signed
        ns = capi.NullifierSet(2 * n)
        t0 = time.perf_counter(); eng.redeem_dev(ns, sk, n, d_proof.data_ptr(), d_rr.data_ptr(), capi.RNG_PER_LANE, d_out.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert int((d_st == 0).sum()) == D and int((d_st == 3).sum()) == n - D and len(ns) == D
        out["redeem_L128_2^%d" % a.refund_log2] = {"redemptions_per_s": n / dt, "ms": 1e3 * dt, "signed": D, "double_spends": n - D}
        ns.close()
        del d_proof, d_pre
        n = 1 << a.issue_log2
        d_req = dev_bytes(req, D).repeat(n // D, 1).contiguous(); d_c = dev_bytes(b"".join(scb(c) for c in cs), D).repeat(n // D, 1).contiguous()
        d_ir = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda"); d_resp = torch.empty((n, 160), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
        dt = timed(lambda: eng.issue_dev(sk, n, d_req.data_ptr(), d_c.data_ptr(), d_ir.data_ptr(), capi.RNG_PER_LANE, d_resp.data_ptr(), d_st.data_ptr()), 2)
        assert int((d_st == 0).sum()) == n
        out["config4_issue_2^%d" % a.issue_log2] = {"issues_per_s": n / dt, "ms": 1e3 * dt}
        d_pre = dev_bytes(pre, D).repeat(n // D, 1).contiguous(); d_out = torch.empty((n, 128), dtype=torch.uint8, device="cuda")
        dt = timed(lambda: eng.request_dev(n, d_pre.data_ptr(), d_ir.data_ptr(), d_out.data_ptr()), 2)
        out["request_2^%d" % a.issue_log2] = {"requests_per_s": n / dt, "ms": 1e3 * dt}
        del d_req, d_c, d_ir, d_resp, d_pre, d_out
        print("configs 2-4 done", out, file=sys.stderr, flush=True)
        # config 5: full lifecycles request -> issue -> token -> prove_spend -> refund -> token, everything resident in HBM,
        # streamed in chunks of 2^16 lanes (device RNG bytes stand in for the callers' generators)
        nl, chunk = 1 << a.lifecycle_log2, 1 << 16
        w = sk[32:]
        u8 = lambda *shape: torch.empty(shape, dtype=torch.uint8, device="cuda")
        rnd = lambda *shape: torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")
        b = {k: u8(chunk, v) for k, v in dict(pre=64, req=128, resp=160, tok=160, proof=pb, prer=96, rf=128, tok2=160).items()}
        stt = u8(chunk); d_c = dev_bytes(b"".join(scb(c) for c in cs), D).repeat(chunk // D, 1).contiguous()
        d_s = dev_bytes(b"".join(scb(s) for s in ss), D).repeat(chunk // D, 1).contiguous()
        r_pre, r_rq, r_ir, r_pr, r_rr = rnd(chunk, 128), rnd(chunk, 128), rnd(chunk, 128), rnd(chunk, eng.prove_rng_bytes), rnd(chunk, 128)
        lib, ctx = eng.lib, eng.ctx
        import ctypes as C
        wbuf = (C.c_uint8 * 32).from_buffer_copy(w); skbuf = (C.c_uint8 * 64).from_buffer_copy(sk)
        seedbuf = (C.c_uint8 * 32).from_buffer_copy(shake("lifecycle-seed", 32))
        seeded = [False]
        spent = {}
        def tm(key, fn):              # every entry point returns with its outputs complete: its wall time is its cost
            t = time.perf_counter(); r = fn(); spent[key] = spent.get(key, 0.0) + time.perf_counter() - t
            return r
        def lifecycle_chunk():
            ck = eng._ck
            ck(tm("pre_issuance_random", lambda: lib.act_pre_issuance_random_batch(ctx, chunk, 1, r_pre.data_ptr(), b["pre"].data_ptr())))
            ck(tm("request", lambda: lib.act_request_batch(ctx, chunk, 1, b["pre"].data_ptr(), r_rq.data_ptr(), b["req"].data_ptr())))
            ck(tm("issue", lambda: lib.act_issue_batch(ctx, chunk, 1, skbuf, b["req"].data_ptr(), d_c.data_ptr(), r_ir.data_ptr(), 0, b["resp"].data_ptr(), stt.data_ptr())))
            ck(tm("issuance_to_credit_token", lambda: lib.act_issuance_to_credit_token_batch(ctx, chunk, 1, b["pre"].data_ptr(), wbuf, b["req"].data_ptr(), b["resp"].data_ptr(), b["tok"].data_ptr(), stt.data_ptr())))
            if seeded[0]:
                ck(tm("prove_spend", lambda: lib.act_prove_spend_seeded_batch(ctx, chunk, 1, b["tok"].data_ptr(), d_s.data_ptr(), seedbuf, C.c_uint64(0), b["proof"].data_ptr(), b["prer"].data_ptr(), stt.data_ptr())))
            else:
                ck(tm("prove_spend", lambda: lib.act_prove_spend_batch(ctx, chunk, 1, b["tok"].data_ptr(), d_s.data_ptr(), r_pr.data_ptr(), b["proof"].data_ptr(), b["prer"].data_ptr(), stt.data_ptr())))
            ck(tm("refund", lambda: lib.act_refund_batch(ctx, chunk, 1, skbuf, b["proof"].data_ptr(), r_rr.data_ptr(), 0, b["rf"].data_ptr(), stt.data_ptr())))
            ck(tm("refund_to_credit_token", lambda: lib.act_refund_to_credit_token_batch(ctx, chunk, 1, b["prer"].data_ptr(), b["proof"].data_ptr(), b["rf"].data_ptr(), wbuf, b["tok2"].data_ptr(), stt.data_ptr())))
        torch.cuda.synchronize(); lifecycle_chunk(); torch.cuda.synchronize()
        assert int((stt == 0).sum()) == chunk, "every lifecycle must close"
        spent.clear()
        t = time.perf_counter()
        for _ in range(nl // chunk):
            lifecycle_chunk()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("lifecycle done", file=sys.stderr, flush=True)
        out["config5_lifecycles_L128_2^%d" % a.lifecycle_log2] = {"lifecycles_per_s": nl / dt, "ms": 1e3 * dt,
                                                                   "ms_per_2^16_lanes_by_call": {k: round(1e3 * v / (nl // chunk), 2) for k, v in spent.items()}}
        seeded[0] = True                  # the prover's generators expanded on the device from a seed (act_prove_spend_seeded_batch)
        torch.cuda.synchronize(); lifecycle_chunk(); torch.cuda.synchronize()
        assert int((stt == 0).sum()) == chunk
        t = time.perf_counter()
        for _ in range(nl // chunk):
            lifecycle_chunk()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        out["config5_lifecycles_L128_2^%d_seeded_prover" % a.lifecycle_log2] = {"lifecycles_per_s": nl / dt, "ms": 1e3 * dt}
        # wire codec and nullifier set
        # timed through the C ABI on pinned host memory and on device memory: the Python wrappers (capi.Engine.cbor_encode /
        # cbor_decode) build lists of byte strings, which costs more than the codec
        nc = 1 << 17
        T = capi.CBOR_TYPES["SpendProof"]; lib, ctx = eng.lib, eng.ctx
        ml = lib.act_cbor_size(ctx, T)
        recs = torch.from_numpy(np.frombuffer(proofs, np.uint8).reshape(D, pb).copy())
        h_recs = torch.empty((nc, pb), dtype=torch.uint8, pin_memory=True); h_recs.copy_(recs.repeat(nc // D, 1))
        h_wire = torch.empty(nc * ml, dtype=torch.uint8, pin_memory=True); h_back = torch.empty((nc, pb), dtype=torch.uint8, pin_memory=True)
        h_st = torch.empty(nc, dtype=torch.uint8, pin_memory=True)
        d_recs = h_recs.cuda(); d_wire = torch.zeros(nc * ml, dtype=torch.uint8, device="cuda"); d_back = torch.zeros((nc, pb), dtype=torch.uint8, device="cuda")
        d_st = torch.zeros(nc, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        def timed2(f):
            f(); torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t
        ck = eng._ck
        dt_e = timed2(lambda: ck(lib.act_cbor_encode_batch(ctx, T, nc, capi.MEM_HOST, h_recs.data_ptr(), h_wire.data_ptr())))
        dt_d = timed2(lambda: ck(lib.act_cbor_decode_batch(ctx, T, nc, capi.MEM_HOST, h_wire.data_ptr(), None, h_back.data_ptr(), h_st.data_ptr())))
        assert not h_st.any() and torch.equal(h_back, h_recs)
        dt_ed = timed2(lambda: ck(lib.act_cbor_encode_batch(ctx, T, nc, capi.MEM_DEVICE, d_recs.data_ptr(), d_wire.data_ptr())))
        dt_dd = timed2(lambda: ck(lib.act_cbor_decode_batch(ctx, T, nc, capi.MEM_DEVICE, d_wire.data_ptr(), None, d_back.data_ptr(), d_st.data_ptr())))
        assert not d_st.any() and torch.equal(d_back, d_recs) and torch.equal(d_wire.cpu(), h_wire)
        print("cbor done", file=sys.stderr, flush=True)
        out["cbor_spend_proof_2^17"] = {"wire_bytes": ml, "record_bytes": pb,
                                        "pinned_host_memory": {"encode_msgs_per_s": nc / dt_e, "decode_msgs_per_s": nc / dt_d,
                                                               "encode_GBps_both_ways": nc * (ml + pb) / dt_e / 1e9, "decode_GBps_both_ways": nc * (ml + pb) / dt_d / 1e9},
                                        "device_memory": {"encode_msgs_per_s": nc / dt_ed, "decode_msgs_per_s": nc / dt_dd,
                                                          "encode_GBps_read_plus_written": nc * (ml + pb) / dt_ed / 1e9, "decode_GBps_read_plus_written": nc * (ml + pb) / dt_dd / 1e9}}
        del h_recs, h_wire, h_back, d_recs, d_wire, d_back
        nn = 1 << 22
        keys = torch.randint(0, 256, (nn, 32), dtype=torch.uint8, device="cuda"); spent = torch.zeros(nn, dtype=torch.uint8, device="cuda")
        ns = capi.NullifierSet(capacity=4 * nn)
        torch.cuda.synchronize()
        ns.check_and_insert_dev(nn, keys.data_ptr(), 32, 0, spent.data_ptr()); torch.cuda.synchronize()
        keys2 = torch.randint(0, 256, (nn, 32), dtype=torch.uint8, device="cuda"); keys2[::2] = keys[::2]
        torch.cuda.synchronize()
        t = time.perf_counter(); ns.check_and_insert_dev(nn, keys2.data_ptr(), 32, 0, spent.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
        assert int(spent.sum()) == nn // 2
        out["nullifier_set_check_insert_2^22"] = {"nullifiers_per_s": nn / dt, "ms": 1e3 * dt}
        # the node-level set (act_node_nullifier_*: host memory in, host-side routing by owner, one HBM set per device) on this one GPU
        nh = 1 << 20
        hk = keys2[:nh].cpu().numpy().copy(); hk0 = keys[:nh].cpu().numpy().copy()       # numpy buffers handed over by pointer (C ABI)
        nns = capi.NodeNullifierSet(4 * nh, devices=(0,))
        sp = np.zeros(nh, np.uint8)
        assert eng.lib.act_node_nullifier_check_and_insert_batch(nns.h, nh, hk0.ctypes.data, 32, None, sp.ctypes.data) == 0
        t = time.perf_counter()
        assert eng.lib.act_node_nullifier_check_and_insert_batch(nns.h, nh, hk.ctypes.data, 32, None, sp.ctypes.data) == 0
        dt = time.perf_counter() - t
        assert int(sp.sum()) == nh // 2
        out["node_nullifier_set_1gpu_host_memory_2^20"] = {"nullifiers_per_s": nh / dt, "ms": 1e3 * dt}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
