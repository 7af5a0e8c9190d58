"""Proofs in pinned host memory are read by the kernels in place (engine.hip mapped_view: hipHostMalloc memory is mapped into the
device's address space); pageable memory is staged as before.  Both hand-overs must give the same statuses, K' and refunds as HBM
and the oracle, on every schedule (tiny, small-batch, pipelined chunks), from the middle of an allocation, and the pinned path must
really skip the staging copy.  (Wire bytes are staged from either kind of memory.)"""
import numpy as np
import pytest

from conftest import shake
from test_gpu_cbor_verify import _proofs

pytestmark = pytest.mark.gpu


def _h2d_bytes(eng):
    return eng.prof().get("copy_h2d(bulk)", {}).get("lanes", 0)


@pytest.mark.parametrize("mode", [0, 1])
def test_pinned_host_proofs_are_read_in_place(engine_factory, oracle, bench_params, mode):
    import torch
    from act_amd import capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=96, transcript=mode)
    octx = oracle.ctx(bench_params, L)
    sk = octx.private_key_random(shake("mr-sk", 64))
    N = 64
    proofs = bytearray(_proofs(eng, sk, N, "mr"))
    pb = eng.proof_bytes
    proofs[pb * 3 + 33] ^= 1; proofs[pb * 9 + 64:pb * 9 + 96] = bytes(32); proofs[pb * 20 + 32 * 5:pb * 20 + 32 * 6] = b"\xff" * 32
    proofs = bytes(proofs)
    want64 = octx.verify_spend_batch(sk, proofs, 4)
    base = np.frombuffer(proofs, np.uint8).reshape(N, pb)
    for n in (1, 40, 64, 500):                                 # tiny / small-batch schedule / (500 > 96 x ...) pipelined chunks
        eng.set_small_batch_max(64)
        tiled = np.tile(base, ((n + N - 1) // N, 1))[:n].copy()
        pad = 3 * pb + 7                                       # the range starts in the middle of the pinned allocation, unaligned
        pinned = torch.zeros(pad + n * pb + 11, dtype=torch.uint8).pin_memory()
        pinned[pad:pad + n * pb] = torch.from_numpy(tiled.reshape(-1))
        ref_st, ref_kp = eng.verify_spend(sk, tiled.tobytes(), True)           # pageable memory: staged
        assert ref_st[:min(n, N)] == want64[:min(n, N)]
        st = torch.full((n,), 9, dtype=torch.uint8).pin_memory(); kp = torch.zeros(32 * n, dtype=torch.uint8).pin_memory()
        eng.prof_reset(); eng.prof_enable(True)
        eng.verify_spend_ptr(sk, n, capi.MEM_HOST, pinned.data_ptr() + pad, st.data_ptr(), kp.data_ptr())
        eng.prof_enable(False)
        assert _h2d_bytes(eng) < n * pb, (n, _h2d_bytes(eng))   # no staging copy of the proofs
        assert st.numpy().tobytes() == ref_st and kp.numpy().tobytes() == ref_kp, n
        # refund from the same pinned range: same bytes as from pageable memory
        rng = shake("mr-rr", 128 * n)
        ref = eng.refund(sk, tiled.tobytes(), rng)
        out = torch.zeros(128 * n, dtype=torch.uint8).pin_memory(); r_t = torch.from_numpy(np.frombuffer(rng, np.uint8).copy()).pin_memory()
        ps, ks = capi._in(sk, 64)
        eng._ck(eng.lib.act_refund_batch(eng.ctx, n, capi.MEM_HOST, ps, pinned.data_ptr() + pad, r_t.data_ptr(), capi.RNG_PER_LANE, out.data_ptr(), st.data_ptr()))
        assert (st.numpy().tobytes(), out.numpy().tobytes()) == ref, n
    assert {0, 6, 7, 255} <= set(ref_st)
    # wire bytes from pinned memory: same verdicts
    msgs = eng.cbor_encode("SpendProof", proofs)
    blob = b"".join(msgs); offs = np.zeros(N + 1, np.uint64); offs[1:] = np.cumsum([len(x) for x in msgs], dtype=np.uint64)
    ref = eng.verify_spend_cbor(sk, msgs, True)
    pin = torch.from_numpy(np.frombuffer(blob, np.uint8).copy()).pin_memory()
    st = torch.full((N,), 9, dtype=torch.uint8).pin_memory(); kp = torch.zeros(32 * N, dtype=torch.uint8).pin_memory()
    eng.verify_spend_cbor_ptr(sk, N, capi.MEM_HOST, pin.data_ptr(), offs.ctypes.data, st.data_ptr(), kp.data_ptr())      # (wire bytes are staged either way)
    assert (st.numpy().tobytes(), kp.numpy().tobytes()) == ref
    eng.set_small_batch_max(8192)
    # the client's side: PreRefund::to_credit_token reads the same SpendProofs (its own, public) from pinned memory in place
    from conftest import scb
    M = 70
    pre = eng.pre_issuance_random(shake("mr2-pre", 128 * M)); req = eng.request(pre, shake("mr2-rq", 128 * M))
    s0, resp = eng.issue(sk, req, b"".join(scb(90 + i) for i in range(M)), shake("mr2-ir", 128 * M))
    s0, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    s0, pr, prer = eng.prove_spend(tok, b"".join(scb(i) for i in range(M)), shake("mr2-pr", eng.prove_rng_bytes * M))
    s0, rf = eng.refund(sk, pr, shake("mr2-rr", 128 * M))
    assert s0 == bytes(M)
    rf = bytearray(rf); rf[128 * 4 + 70] ^= 1; rf = bytes(rf)
    ref = eng.refund_to_credit_token(prer, pr, rf, sk[32:])
    assert len(set(ref[0])) == 2 and ref[0][4] != 0
    pin_of = lambda b: torch.from_numpy(np.frombuffer(b, np.uint8).copy()).pin_memory()
    t_prer, t_pr, t_rf = pin_of(prer), pin_of(pr), pin_of(rf)
    t_out = torch.zeros(160 * M, dtype=torch.uint8).pin_memory(); t_st = torch.full((M,), 9, dtype=torch.uint8).pin_memory()
    pw, kw = capi._in(sk[32:], 32)
    eng.prof_reset(); eng.prof_enable(True)
    eng._ck(eng.lib.act_refund_to_credit_token_batch(eng.ctx, M, capi.MEM_HOST, t_prer.data_ptr(), t_pr.data_ptr(), t_rf.data_ptr(), pw, t_out.data_ptr(), t_st.data_ptr()))
    eng.prof_enable(False)
    assert _h2d_bytes(eng) < M * pb
    assert (t_st.numpy().tobytes(), t_out.numpy().tobytes()) == ref
    assert eng.secret_residue() == 0


def test_registered_host_memory_is_read_in_place_too(engine_factory, oracle, bench_params):
    """hipHostRegister on an ordinary allocation (what a server would do with its receive buffers): mapped, so read in place; the same
    buffer after hipHostUnregister is pageable again and is staged.  Same verdicts either way."""
    import ctypes as C
    from act_amd import capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=96, transcript=1)
    octx = oracle.ctx(bench_params, L)
    sk = octx.private_key_random(shake("mr-sk", 64))
    N = 48
    proofs = bytearray(_proofs(eng, sk, N, "mr3"))
    pb = eng.proof_bytes
    proofs[pb * 7 + 33] ^= 1
    want = octx.verify_spend_batch(sk, bytes(proofs), 4)
    buf = np.frombuffer(bytes(proofs), np.uint8).copy()
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]; hip.hipHostUnregister.argtypes = [C.c_void_p]
    assert hip.hipHostRegister(buf.ctypes.data, buf.nbytes, 0) == 0
    try:
        st = np.full(N, 9, np.uint8)
        eng.prof_reset(); eng.prof_enable(True)
        eng.verify_spend_ptr(sk, N, capi.MEM_HOST, buf.ctypes.data, st.ctypes.data)
        eng.prof_enable(False)
        assert st.tobytes() == want and _h2d_bytes(eng) < N * pb
    finally:
        assert hip.hipHostUnregister(buf.ctypes.data) == 0
    st = np.full(N, 9, np.uint8)
    eng.prof_reset(); eng.prof_enable(True)
    eng.verify_spend_ptr(sk, N, capi.MEM_HOST, buf.ctypes.data, st.ctypes.data)
    eng.prof_enable(False)
    assert st.tobytes() == want and _h2d_bytes(eng) >= N * pb
