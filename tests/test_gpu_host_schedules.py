"""Host-memory callers get chunk schedules of their own (engine.hip): a verify / refund call with device transcripts opens with an
eighth of the call so that the rest's proofs arrive under its kernels; prove_spend cuts its last chunk in two so that the first
part leaves under the second part's kernels; the client's refund check alternates quarters between the two slots.  Device-memory
callers take none of these.  Same call, both memory kinds, sizes at which the schedules engage: the bytes must be equal -- and the
statuses what the tampering says."""
import ctypes as C

import pytest

from conftest import shake, scb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [20100, 12000, 5000])    # 20 100: the pipelined schedule; 12 000: two halves under host transcripts, the small-batch schedule
                                                       # under device transcripts (its limit there is 16 384); 5 000: the small-batch one in both modes,
@pytest.mark.parametrize("mode", [0, 1])        # whose copy-in and range kernel go in two halves.  mode: ACT_TRANSCRIPT_HOST, _DEVICE
def test_host_and_device_memory_calls_agree_at_sizes_that_split(engine_factory, bench_params, mode, N):
    import numpy as np
    import torch
    from act_amd import capi
    L, D = 8, 64
    eng = engine_factory(bench_params, L, max_batch=32768, transcript=mode)
    sk = eng.private_key_random(shake("hs-sk", 64))
    pre = eng.pre_issuance_random(shake("hs-pre", 128 * D)); req = eng.request(pre, shake("hs-rq", 128 * D))
    st, resp = eng.issue(sk, req, b"".join(scb(100 + i) for i in range(D)), shake("hs-ir", 128 * D))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(D)
    pb, rb = eng.proof_bytes, eng.prove_rng_bytes
    pin = lambda *shape: torch.empty(shape, dtype=torch.uint8, pin_memory=True)
    # ---- prove_spend: N lanes (the D tokens repeated, every lane its own generator bytes), host memory vs device memory ----
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    d_rng = torch.randint(0, 256, (N, rb), dtype=torch.uint8, device="cuda", generator=g)
    tok_np = np.tile(np.frombuffer(tok, np.uint8).reshape(D, 160), ((N + D - 1) // D, 1))[:N].copy()
    s_np = np.tile(np.frombuffer(b"".join(scb(i % 50) for i in range(D)), np.uint8).reshape(D, 32), ((N + D - 1) // D, 1))[:N].copy()
    h_tok = pin(N, 160); h_tok.numpy()[:] = tok_np
    h_s = pin(N, 32); h_s.numpy()[:] = s_np
    h_rng = pin(N, rb); h_rng.copy_(d_rng)
    d_tok, d_s = h_tok.cuda(), h_s.cuda()
    d_proof = torch.zeros((N, pb), dtype=torch.uint8, device="cuda"); d_prer = torch.zeros((N, 96), dtype=torch.uint8, device="cuda")
    d_st = torch.full((N,), 9, dtype=torch.uint8, device="cuda")
    h_proof, h_prer, h_st = pin(N, pb), pin(N, 96), pin(N)
    torch.cuda.synchronize()
    eng.prove_spend_dev(N, d_tok.data_ptr(), d_s.data_ptr(), d_rng.data_ptr(), d_proof.data_ptr(), d_prer.data_ptr(), d_st.data_ptr())
    eng._ck(eng.lib.act_prove_spend_batch(eng.ctx, N, capi.MEM_HOST, h_tok.data_ptr(), h_s.data_ptr(), h_rng.data_ptr(), h_proof.data_ptr(), h_prer.data_ptr(), h_st.data_ptr()))
    torch.cuda.synchronize()
    assert not d_st.any() and not h_st.any()
    assert torch.equal(d_proof.cpu(), h_proof) and torch.equal(d_prer.cpu(), h_prer)
    # ---- verify / refund: some lanes tampered ----
    bad = np.arange(5, N, 97)
    h_proof.numpy()[bad, 40] ^= 1                 # a bit of the second scalar: InvalidClientSpendProof
    d_proof = h_proof.cuda()
    exp = np.zeros(N, np.uint8); exp[bad] = 7
    d_kp = torch.zeros((N, 32), dtype=torch.uint8, device="cuda"); h_kp = pin(N, 32)
    torch.cuda.synchronize()
    eng.verify_spend_dev(sk, N, d_proof.data_ptr(), d_st.data_ptr(), d_kp.data_ptr())
    eng.verify_spend_ptr(sk, N, capi.MEM_HOST, h_proof.data_ptr(), h_st.data_ptr(), h_kp.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_st.cpu().numpy(), exp) and np.array_equal(h_st.numpy(), exp)
    assert torch.equal(d_kp.cpu(), h_kp) and bool(h_kp[torch.from_numpy(exp == 0)].any(dim=1).all())
    d_r128 = torch.randint(0, 256, (N, 128), dtype=torch.uint8, device="cuda", generator=g)
    h_r128 = pin(N, 128); h_r128.copy_(d_r128)
    d_rf = torch.zeros((N, 128), dtype=torch.uint8, device="cuda"); h_rf = pin(N, 128)
    torch.cuda.synchronize()
    eng.refund_dev(sk, N, d_proof.data_ptr(), d_r128.data_ptr(), capi.RNG_PER_LANE, d_rf.data_ptr(), d_st.data_ptr())
    skb = (C.c_uint8 * 64).from_buffer_copy(sk)
    eng._ck(eng.lib.act_refund_batch(eng.ctx, N, capi.MEM_HOST, skb, h_proof.data_ptr(), h_r128.data_ptr(), capi.RNG_PER_LANE, h_rf.data_ptr(), h_st.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(d_st.cpu().numpy(), exp) and np.array_equal(h_st.numpy(), exp)
    assert torch.equal(d_rf.cpu(), h_rf) and not h_rf[torch.from_numpy(exp != 0)].any()
    # ---- the client's refund check: host memory (quarters on two slots) vs the byte-string wrapper over a few lanes' oracle ----
    wb = (C.c_uint8 * 32).from_buffer_copy(sk[32:])
    h_tok2, h_st2 = pin(N, 160), pin(N)
    d_tok2 = torch.zeros((N, 160), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng._ck(eng.lib.act_refund_to_credit_token_batch(eng.ctx, N, capi.MEM_HOST, h_prer.data_ptr(), h_proof.data_ptr(), h_rf.data_ptr(), wb, h_tok2.data_ptr(), h_st2.data_ptr()))
    eng._ck(eng.lib.act_refund_to_credit_token_batch(eng.ctx, N, capi.MEM_DEVICE, d_prer.data_ptr(), d_proof.data_ptr(), d_rf.data_ptr(), wb, d_tok2.data_ptr(), d_st.data_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(d_st.cpu(), h_st2) and torch.equal(d_tok2.cpu(), h_tok2)
    exp2 = np.where(exp == 0, 0, 4).astype(np.uint8)            # a lane without a refund: InvalidRefundProof
    assert np.array_equal(h_st2.numpy(), exp2)
    # balances of the accepted lanes: c - s
    want = (100 + np.arange(N) % D) - (np.arange(N) % D) % 50          # L = 8: amounts below 256
    got = h_tok2.numpy()[:, 128:136].copy().view("<u8").reshape(-1)
    assert np.array_equal(got[exp == 0], want[exp == 0].astype(np.uint64))
    assert eng.secret_residue() == 0


@pytest.mark.parametrize("L,N,max_batch", [(128, 6000, 2000), (8, 40000, 4096)])      # range kernels of 1 000 workgroups (more than the 512 resident) / of 128
def test_the_release_point_of_the_next_range_kernel_changes_no_byte(bench_params, monkeypatch, L, N, max_batch):
    """Host-transcript callers run their chunks' range kernels one behind the other; the later one is released by a
    hipStreamWaitValue32 on the earlier one's finished-workgroup counter (engine.hip spend_stage1), or -- knob `hard_stagger`, and on a
    device without the attribute -- by its completion event.  An ordering hint: statuses and K' must not know which."""
    import numpy as np
    import torch
    from act_amd import capi
    D = 50
    outs = []
    for hard in (0, 1):
        monkeypatch.setenv("ACT_HARD_STAGGER", str(hard))      # capi.Engine forwards the ACT_* knobs as they stand (act_tuning_set)
        try:
            eng = capi.Engine(bench_params, L, max_batch=max_batch, transcript=capi.TRANSCRIPT_HOST)
            sk = eng.private_key_random(shake("rp-sk", 64))
            pre = eng.pre_issuance_random(shake("rp-pre", 128 * D)); req = eng.request(pre, shake("rp-rq", 128 * D))
            st, resp = eng.issue(sk, req, b"".join(scb(200 + i) for i in range(D)), shake("rp-ir", 128 * D))
            st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
            st, proofs, _ = eng.prove_spend(tok, b"".join(scb(i) for i in range(D)), shake("rp-pr", eng.prove_rng_bytes * D))
            assert st == bytes(D)
            pb = eng.proof_bytes
            rec = np.tile(np.frombuffer(proofs, np.uint8).reshape(D, pb), ((N + D - 1) // D, 1))[:N].copy()
            bad = np.arange(3, N, 61); rec[bad, 40] ^= 1
            d_proof = torch.from_numpy(rec).cuda()
            d_st = torch.full((N,), 9, dtype=torch.uint8, device="cuda"); d_kp = torch.zeros((N, 32), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for _ in range(2):
                eng.verify_spend_dev(sk, N, d_proof.data_ptr(), d_st.data_ptr(), d_kp.data_ptr())
            torch.cuda.synchronize()
            exp = np.zeros(N, np.uint8); exp[bad] = 7
            assert np.array_equal(d_st.cpu().numpy(), exp)
            outs.append(d_kp.cpu())
            eng.close()
        finally:
            monkeypatch.delenv("ACT_HARD_STAGGER"); capi.forward_tuning_env()
    assert torch.equal(outs[0], outs[1]) and bool(outs[0][torch.from_numpy(exp == 0)].any(dim=1).all())
