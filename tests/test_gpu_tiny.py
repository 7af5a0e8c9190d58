"""Tiny calls (at most 64 lanes: the crate's one-item call shape, src/lib.rs:463, 621, 781): one kernel per call, roles on different
wavefronts, the single-chunk transcript hashed in the kernel.  The bytes must be those of the multi-launch path and of the oracle,
in both transcript settings, at the sizes on either side of the limit, from host and device memory, with every rng convention."""
import numpy as np
import pytest

from conftest import shake, scb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("L", [128, 8])
def test_tiny_calls_equal_the_oracle_and_the_multi_launch_path(engine_factory, oracle, bench_params, L, mode):
    import torch
    from act_amd import capi
    eng = engine_factory(bench_params, L, max_batch=256, transcript=mode)
    octx = oracle.ctx(bench_params, L)
    sk = octx.private_key_random(shake("ty-sk", 64))
    N = 130                                                   # > 64: the multi-launch path; its lanes are the reference for the tiny calls
    pre = eng.pre_issuance_random(shake("ty-pre", 128 * N))
    rq = shake("ty-rq", 128 * N)
    req_big = eng.request(pre, rq)
    assert req_big == octx.request_batch(pre, rq, 4)
    for n in (1, 2, 63, 64, 65):
        assert eng.request(pre[:64 * n], rq[:128 * n]) == req_big[:128 * n], n
    eng.set_tiny_calls(False)                                  # the multi-launch path at the same sizes: same bytes
    try:
        for n in (1, 5, 64):
            assert eng.request(pre[:64 * n], rq[:128 * n]) == req_big[:128 * n], n
    finally:
        eng.set_tiny_calls(True)
    # device memory
    d = lambda b: torch.from_numpy(np.frombuffer(b, np.uint8).copy()).cuda()
    d_pre, d_rq = d(pre[:64 * 5]), d(rq[:128 * 5]); d_out = torch.zeros(128 * 5, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.request_dev(5, d_pre.data_ptr(), d_rq.data_ptr(), d_out.data_ptr())
    assert d_out.cpu().numpy().tobytes() == req_big[:128 * 5]
    # issue: rejected lanes among the first 64 (tampered gamma / undecodable K)
    bad = bytearray(req_big)
    bad[128 * 1 + 40] ^= 1; bad[128 * 7 + 3] ^= 0x08; bad[128 * 40 + 100] ^= 2; bad[128 * 100 + 70] ^= 1
    bad = bytes(bad)
    cam = b"".join(scb(10 + i) for i in range(N))
    irng = shake("ty-ir", 128 * N)
    for rng_mode in (capi.RNG_PER_LANE, capi.RNG_SEQUENTIAL):
        want = octx.issue_batch(sk, bad, cam, irng, 4) if rng_mode == capi.RNG_PER_LANE else None
        big = eng.issue(sk, bad, cam, irng, rng_mode)
        if want:
            assert big == want
        for n in (1, 2, 9, 64):
            got = eng.issue(sk, bad[:128 * n], cam[:32 * n], irng, rng_mode)
            if rng_mode == capi.RNG_PER_LANE:
                assert got == (big[0][:n], big[1][:160 * n]), (rng_mode, n)
            else:                                             # sequential: lane i draws slice (accepted lanes before it) of THIS call
                cur, exp = 0, b""
                for i in range(n):
                    s1, r1 = octx.issue(sk, bad[128 * i:128 * i + 128], cam[32 * i:32 * i + 32], irng[128 * cur:128 * cur + 128])
                    assert s1 == got[0][i]
                    exp += r1
                    cur += s1 == 0
                assert got[1] == exp, n
        assert {0, 1, 255} <= set(big[0])
    # the halves: check, then sign with exactly the accepted lanes' bytes
    for n in (1, 9, 64):
        st = capi_issue_check(eng, bad[:128 * n])                # (one kernel: k_sign_fused's check role on its own)
        assert st == big[0][:n]
        d_rq2 = d(bad[:128 * n]); d_st = torch.full((n,), 9, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        eng._ck(eng.lib.act_issue_check_batch(eng.ctx, n, capi.MEM_DEVICE, d_rq2.data_ptr(), d_st.data_ptr()))
        assert d_st.cpu().numpy().tobytes() == st and d_rq2.cpu().numpy().tobytes() == bad[:128 * n]
        acc = sum(1 for v in st if v == 0)
        got = capi_issue_sign(eng, sk, bad[:128 * n], cam[:32 * n], st, irng[:128 * acc])
        assert got == eng.issue(sk, bad[:128 * n], cam[:32 * n], irng, capi.RNG_SEQUENTIAL)
    # refund's signature half on tiny calls (verification is the spend path), tokens of the honest lanes
    st, resp = eng.issue(sk, req_big, cam, irng)
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req_big, resp)
    assert st == bytes(N)
    for n in (1, 3, 64):
        assert eng.issuance_to_credit_token(pre[:64 * n], sk[32:], req_big[:128 * n], resp[:160 * n]) == (bytes(n), tok[:160 * n])
    M = 12
    s_b = b"".join(scb(i) for i in range(M))
    prng = shake("ty-pr", eng.prove_rng_bytes * M)
    st, proofs, prer = eng.prove_spend(tok[:160 * M], s_b, prng)
    assert (proofs, prer) == octx.prove_spend_batch(tok[:160 * M], s_b, prng, 4)
    pb = eng.proof_bytes
    t = bytearray(proofs); t[pb * 2 + 33] ^= 1; t[pb * 5 + 64:pb * 5 + 96] = bytes(32); t = bytes(t)
    rrng = shake("ty-rr", 128 * M)
    for rng_mode in (capi.RNG_PER_LANE, capi.RNG_SEQUENTIAL):
        got = eng.refund(sk, t, rrng, rng_mode)
        cur = 0
        for i in range(M):
            slot = i if rng_mode == capi.RNG_PER_LANE else cur
            so, ro = octx.refund(sk, t[pb * i:pb * i + pb], rrng[128 * slot:128 * slot + 128])
            assert (so, ro) == (got[0][i], got[1][128 * i:128 * i + 128]), (rng_mode, i)
            cur += so == 0
        assert set(got[0]) == {0, 6, 7}
    # one proof per call (the crate's call shape; the signature is computed beside the verification in either rng convention), an
    # accepted, a rejected and an identity-A' proof; then the same from device memory
    for i in (0, 2, 5):
        for rng_mode in (capi.RNG_PER_LANE, capi.RNG_SEQUENTIAL):
            so, ro = octx.refund(sk, t[pb * i:pb * i + pb], rrng[:128])
            assert eng.refund(sk, t[pb * i:pb * i + pb], rrng[:128], rng_mode) == (bytes([so]), ro), (i, rng_mode)
    d_p, d_r = d(t), d(rrng); d_o = torch.full((128 * M,), 7, dtype=torch.uint8, device="cuda"); d_s = torch.full((M,), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.refund_dev(sk, M, d_p.data_ptr(), d_r.data_ptr(), capi.RNG_PER_LANE, d_o.data_ptr(), d_s.data_ptr())
    assert (d_s.cpu().numpy().tobytes(), d_o.cpu().numpy().tobytes()) == eng.refund(sk, t, rrng, capi.RNG_PER_LANE)
    eng.set_tiny_calls(False)                                  # the signature behind the verification: same bytes
    try:
        assert eng.refund(sk, t, rrng, capi.RNG_PER_LANE) == (d_s.cpu().numpy().tobytes(), d_o.cpu().numpy().tobytes())
    finally:
        eng.set_tiny_calls(True)
    st, rf = eng.refund(sk, proofs, rrng)
    got = eng.refund_to_credit_token(prer, proofs, rf, sk[32:])
    for i in range(M):
        so, to = octx.refund_to_credit_token(prer[96 * i:96 * i + 96], proofs[pb * i:pb * i + pb], rf[128 * i:128 * i + 128], sk[32:])
        assert (so, to) == (got[0][i], got[1][160 * i:160 * i + 160]), i
    assert eng.secret_residue() == 0


def capi_issue_check(eng, req):
    from act_amd import capi
    n = len(req) // 128; st = np.zeros(n, np.uint8); a = np.frombuffer(req, np.uint8)
    eng._ck(eng.lib.act_issue_check_batch(eng.ctx, n, capi.MEM_HOST, a.ctypes.data, st.ctypes.data))
    return st.tobytes()


def capi_issue_sign(eng, sk, req, cam, status_in, rng):
    from act_amd import capi
    n = len(status_in); out = np.zeros(160 * n, np.uint8); st = np.zeros(n, np.uint8)
    a = np.frombuffer(req, np.uint8); c = np.frombuffer(cam, np.uint8); si = np.frombuffer(status_in, np.uint8); r = np.frombuffer(rng + b"\0", np.uint8); k = np.frombuffer(sk, np.uint8)
    eng._ck(eng.lib.act_issue_sign_batch(eng.ctx, n, capi.MEM_HOST, k.ctypes.data, a.ctypes.data, c.ctypes.data, si.ctypes.data, r.ctypes.data, capi.RNG_SEQUENTIAL, out.ctypes.data, st.ctypes.data))
    return st.tobytes(), out.tobytes()
