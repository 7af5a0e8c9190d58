"""act_verify_spend_cbor_batch (wire bytes in, verdict out: SpendProof::from_cbor, src/cbor.rs:236-408, then PrivateKey::refund
up to the challenge check, src/lib.rs:787-844, in one pass) must give exactly what the two calls it fuses give one after the
other -- act_cbor_decode_batch, then act_verify_spend_batch on the records that decoded -- on canonical messages, on every
non-canonical / broken variant of tests/test_cbor.py, on tampered and identity proofs, in both transcript modes, from host and
from device memory, over ragged chunks; and, through the Python model's from_cbor + the C oracle's verifier, what the
reference's own pair of calls would give."""
import numpy as np
import pytest

import pymodel as m
from conftest import shake, scb
from test_cbor import _variants

pytestmark = pytest.mark.gpu


def _proofs(eng, sk, n, tag):
    pre = eng.pre_issuance_random(shake(tag + "-pre", 128 * n)); req = eng.request(pre, shake(tag + "-rq", 128 * n))
    st, resp = eng.issue(sk, req, b"".join(scb(50 + i) for i in range(n)), shake(tag + "-ir", 128 * n))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(i % 40) for i in range(n)), shake(tag + "-pr", eng.prove_rng_bytes * n))
    assert st == bytes(n)
    return proofs


def _expected(eng, sk, msgs):
    """the unfused composition: decode, then verify what decoded"""
    st_d, recs = eng.cbor_decode("SpendProof", msgs)
    pb = eng.proof_bytes
    good = [i for i in range(len(msgs)) if st_d[i] == 0]
    st_v, kp_v = eng.verify_spend(sk, b"".join(recs[pb * i:pb * i + pb] for i in good), True) if good else (b"", b"")
    exp_st = [{1: 254, 2: 253, 3: 255}.get(s, 0) for s in st_d]
    exp_kp = [bytes(32)] * len(msgs)
    for k, i in enumerate(good):
        exp_st[i] = st_v[k]; exp_kp[i] = kp_v[32 * k:32 * k + 32]
    return bytes(exp_st), b"".join(exp_kp)


@pytest.mark.parametrize("L,max_batch", [(128, 5), (8, 3)])
def test_fused_wire_verify_equals_decode_then_verify(engine_factory, oracle, bench_params, L, max_batch):
    import torch
    from act_amd import capi
    eng = engine_factory(bench_params, L, max_batch=max_batch)
    sk = eng.private_key_random(shake("cv-sk", 64))
    n = 13
    proofs = _proofs(eng, sk, n, "cv%d" % L)
    pb = eng.proof_bytes
    recs = [bytearray(proofs[pb * i:pb * i + pb]) for i in range(n)]
    recs[2][32] ^= 1                                   # charge s        -> 7
    recs[3][64:96] = bytes(32)                          # A' = identity   -> 6
    recs[4][32 * (4 + 1):32 * (4 + 2)] = b"\xff" * 32   # Com_1 undecodable -> 255 (from_cbor: InvalidValue)
    recs[5][32 * (4 + L)] ^= 4                          # gamma           -> 7
    recs[6][0:32] = (m.ELL + 9).to_bytes(32, "little")  # k given as k' + l: accepted by from_cbor (reduced), then the proof fails
    msgs = eng.cbor_encode("SpendProof", b"".join(bytes(r) for r in recs))
    for r in (recs[0], recs[7]):                       # every non-canonical / broken spelling of two valid proofs
        msgs += [v for v, _ in _variants("SpendProof", bytes(r), L)]
    exp_st, exp_kp = _expected(eng, sk, msgs)
    assert {0, 6, 7, 253, 254, 255}.issubset(set(exp_st))
    octx = oracle.ctx(bench_params, L)
    for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
        eng.set_transcript_mode(mode)
        st, kp = eng.verify_spend_cbor(sk, msgs, True)
        assert st == exp_st, [(i, st[i], exp_st[i]) for i in range(len(msgs)) if st[i] != exp_st[i]]
        assert kp == exp_kp
    # the reference's pair of calls, restated: the Python model's from_cbor, then the C oracle's verifier
    for i, msg in enumerate(msgs):
        es, rec = m.cbor_decode("SpendProof", msg, L)
        want = {1: 254, 2: 253, 3: 255}[es] if es else octx.verify_spend(sk, rec)[0]
        assert exp_st[i] == want, (i, exp_st[i], want)
    # device memory, offsets on the host; and canonical-size messages without offsets
    blob = b"".join(msgs)
    offs = np.zeros(len(msgs) + 1, np.uint64); offs[1:] = np.cumsum([len(x) for x in msgs], dtype=np.uint64)
    d_blob = torch.from_numpy(np.frombuffer(blob + b"\0", np.uint8).copy()).cuda()
    d_st = torch.full((len(msgs),), 99, dtype=torch.uint8, device="cuda"); d_kp = torch.zeros(32 * len(msgs), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.verify_spend_cbor_ptr(sk, len(msgs), capi.MEM_DEVICE, d_blob.data_ptr(), offs.ctypes.data, d_st.data_ptr(), d_kp.data_ptr())
    assert d_st.cpu().numpy().tobytes() == exp_st and d_kp.cpu().numpy().tobytes() == exp_kp
    canon = msgs[:n]
    cb = np.frombuffer(b"".join(canon), np.uint8).copy(); st2 = np.full(n, 99, np.uint8)
    eng.verify_spend_cbor_ptr(sk, n, capi.MEM_HOST, cb.ctypes.data, 0, st2.ctypes.data)
    assert st2.tobytes() == exp_st[:n]
    assert eng.verify_spend_cbor(sk, []) == b""
    assert eng.secret_residue() == 0


def test_fused_wire_verify_through_the_node(engine_factory, bench_params):
    from act_amd import capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=4)
    sk = eng.private_key_random(shake("cvn-sk", 64))
    n = 11
    proofs = bytearray(_proofs(eng, sk, n, "cvn"))
    proofs[eng.proof_bytes * 9 + 33] ^= 8
    msgs = eng.cbor_encode("SpendProof", bytes(proofs))
    msgs[4] = b"\xbf" + msgs[4][1:] + b"\xff"          # indefinite-length map: same content, not canonical
    want = eng.verify_spend_cbor(sk, msgs)
    node = capi.Node(bench_params, L, devices=(0, 0, 0), max_batch=2)
    try:
        offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(x) for x in msgs], dtype=np.uint64)
        blob = np.frombuffer(b"".join(msgs) + b"\0", np.uint8).copy(); st = np.full(n, 99, np.uint8)
        import ctypes as C
        skb = (C.c_uint8 * 64).from_buffer_copy(sk)
        node._ck(node.lib.act_node_verify_spend_cbor_batch(node.nd, n, skb, blob.ctypes.data, offs.ctypes.data, st.ctypes.data, None))
        assert st.tobytes() == want and list(want).count(0) == n - 1
    finally:
        node.close()
