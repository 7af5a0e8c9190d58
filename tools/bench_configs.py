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
    ap.add_argument("--max-batch", type=int, default=16384)
    a = ap.parse_args()
    out = {}
    h = capi.params_new("bench-org", "bench-service", "bench-env", "2024-01-01")
    D = 4096
    for L in (64, 128):
        eng = capi.Engine(h, L, max_batch=a.max_batch, transcript=capi.TRANSCRIPT_DEVICE)
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
    print(json.dumps(out))


if __name__ == "__main__":
    main()
