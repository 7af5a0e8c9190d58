"""Wire bytes in, wire bytes out (act_refund_cbor_batch, act_redeem_cbor_batch and their node forms): what a server built on the
crate does with the bytes a client sent --

    let proof  = SpendProof::from_cbor(&msg)?;                         // src/cbor.rs:276-408
    if db.is_used(&proof.nullifier()) { DoubleSpend } else { db.mark_used(..) }   // examples/act.rs:62-67
    let refund = private_key.refund(&params, &proof, &mut rng)?;       // src/lib.rs:781-869
    refund.to_cbor()                                                    // src/cbor.rs:421-433

-- as one call over a batch, compared lane by lane with that loop restated: the Python model's from_cbor, the C oracle's refund fed
the rng slice the sequential loop would have handed it, the Python model's to_cbor.  Canonical, non-canonical and broken messages,
tampered and identity proofs, every rng convention (per lane, sequential bytes, the generator itself through ACT_RNG_CALLBACK), host
and device memory, both transcript modes, the one-message call, the node handle."""
import numpy as np
import pytest

import pymodel as m
from conftest import shake, scb
from test_cbor import _variants
from test_gpu_cbor_verify import _proofs

pytestmark = pytest.mark.gpu

WIRE = {1: 254, 2: 253, 3: 255}       # CborError::{Ciborium, InvalidStructure, InvalidValue} as lane statuses


def _messages(eng, sk, L, n, tag):
    proofs = _proofs(eng, sk, n, tag)
    pb = eng.proof_bytes
    recs = [bytearray(proofs[pb * i:pb * i + pb]) for i in range(n)]
    recs[2][32] ^= 1                                   # charge s          -> 7
    recs[3][64:96] = bytes(32)                          # A' = identity     -> 6
    recs[4][32 * (4 + 1):32 * (4 + 2)] = b"\xff" * 32   # Com_1 undecodable -> 255
    recs[5][32 * (4 + L)] ^= 4                          # gamma             -> 7
    msgs = eng.cbor_encode("SpendProof", b"".join(bytes(r) for r in recs))
    for r in (recs[0], recs[7]):                       # every non-canonical / broken spelling of two valid proofs (repeats of their nullifiers)
        msgs += [v for v, _ in _variants("SpendProof", bytes(r), L)]
    return msgs


def _loop(octx, sk, L, msgs, stream, db=None, per_lane=False):
    """the server loop over the messages, one generator (or lane i's own slice): statuses, Refund messages, bytes drawn"""
    st, out, cur = [], [], 0
    for i, msg in enumerate(msgs):
        es, rec = m.cbor_decode("SpendProof", msg, L)
        if es:
            st.append(WIRE[es]); out.append(b""); continue
        v = octx.verify_spend(sk, rec)[0]
        if v:
            st.append(v); out.append(b""); continue
        if db is not None:
            k = int.from_bytes(rec[:32], "little") % m.ELL
            if k in db:
                st.append(3); out.append(b""); continue
            db.add(k)
        rng = stream[128 * i:128 * i + 128] if per_lane else stream[128 * cur:128 * cur + 128]
        s2, rf = octx.refund(sk, rec, rng)
        assert s2 == 0
        cur += 1
        st.append(0); out.append(m.cbor_encode("Refund", rf, L))
    return bytes(st), out, 128 * cur


@pytest.mark.parametrize("L,max_batch", [(128, 6), (8, 3)])
def test_refund_and_redeem_on_wire_bytes_equal_the_server_loop(engine_factory, oracle, bench_params, L, max_batch):
    import torch
    from act_amd import capi
    eng = engine_factory(bench_params, L, max_batch=max_batch)
    sk = eng.private_key_random(shake("wr-sk", 64))
    octx = oracle.ctx(bench_params, L)
    n = 13
    msgs = _messages(eng, sk, L, n, "wr%d" % L)
    N = len(msgs)
    stream = shake("wr-rng", 128 * N)
    want_st, want_out, drawn = _loop(octx, sk, L, msgs, stream)
    assert {0, 6, 7, 253, 254, 255}.issubset(set(want_st)) and want_st.count(0) > 12
    want_pl = _loop(octx, sk, L, msgs, stream, per_lane=True)
    ml = eng.cbor_size("Refund")
    assert ml == 141 and all(len(x) in (0, ml) for x in want_out)
    for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
        eng.set_transcript_mode(mode)
        # the generator itself: drawn once, after the verdicts, exactly what the loop drew
        g = capi.ReplayRng(stream)
        st, out = eng.refund_cbor(sk, msgs, g, capi.RNG_CALLBACK)
        assert st == want_st, [(i, st[i], want_st[i]) for i in range(N) if st[i] != want_st[i]]
        assert out == want_out and g.draws == [drawn] and g.pos == drawn
        # the same bytes pre-drawn, and lane i's own slice
        assert eng.refund_cbor(sk, msgs, stream, capi.RNG_SEQUENTIAL) == (want_st, want_out)
        assert eng.refund_cbor(sk, msgs, stream, capi.RNG_PER_LANE) == want_pl[:2]
        # one message per call (the small-batch schedule starts from the unframed record)
        for i in (0, 2, 4, n + 3, N - 1):
            assert eng.refund_cbor(sk, [msgs[i]], stream[:128], capi.RNG_SEQUENTIAL) == _loop(octx, sk, L, [msgs[i]], stream)[:2]
        # a few canonical messages with per-lane slices: unframing, verification and the signature beside it in one call (cbor_impl.inc
        # refund_cbor_tiny_records); with a non-canonical message among them the same call falls back to the general path
        k = min(6, max_batch)
        assert eng.refund_cbor(sk, msgs[:k], stream[:128 * k], capi.RNG_PER_LANE) == _loop(octx, sk, L, msgs[:k], stream, per_lane=True)[:2]
        mixed = [msgs[1], msgs[n + 3], msgs[0]][:max_batch]
        assert eng.refund_cbor(sk, mixed, stream[:128 * len(mixed)], capi.RNG_PER_LANE) == _loop(octx, sk, L, mixed, stream, per_lane=True)[:2]
        # the halves: verdict + K' + nullifier, then sign + frame
        stv, kp, nul = eng.verify_spend_cbor_keys(sk, msgs)
        assert stv == want_st
        for i, msg in enumerate(msgs):
            es, rec = m.cbor_decode("SpendProof", msg, L)
            # the nullifier comes back as it stood on the wire: reduced or not, it names the same scalar (zero if the message did not
            # parse; a message that parses but holds an undecodable point still has one -- its lane is rejected, nobody looks it up)
            if es != 3:
                assert int.from_bytes(nul[32 * i:32 * i + 32], "little") % m.ELL == (0 if es else int.from_bytes(rec[:32], "little") % m.ELL), i
        assert eng.refund_sign_cbor(sk, kp, stv, stream, capi.RNG_SEQUENTIAL) == (want_st, want_out)
        # redeem: the loop with the nullifier store; the repeats of proofs 0 and 7 are double spends
        db = set()
        r_st, r_out, r_drawn = _loop(octx, sk, L, msgs, stream, db)
        assert r_st.count(3) > 5
        ns = capi.NullifierSet(4 * N)
        g = capi.ReplayRng(stream)
        assert eng.redeem_cbor(ns, sk, msgs, g, capi.RNG_CALLBACK) == (r_st, r_out)
        assert g.draws == [r_drawn] and len(ns) == len(db)
        g = capi.ReplayRng(stream)                      # everything again: nothing is fresh, nothing is drawn
        st2, out2 = eng.redeem_cbor(ns, sk, msgs, g, capi.RNG_CALLBACK)
        assert st2 == bytes(3 if s == 0 else s for s in want_st) and not any(out2) and g.pos == 0
        ns.close()
        # one message per call with its 128 bytes (the refund in one call, the signature beside the verification, THEN the nullifier
        # store: nullifier_impl.inc): message by message what the loop with the store answers, the same store at the end
        ns = capi.NullifierSet(4 * N)
        cur = 0
        for i, msg in enumerate(msgs):
            st1, out1 = eng.redeem_cbor(ns, sk, [msg], stream[128 * cur:128 * cur + 128], capi.RNG_SEQUENTIAL)
            assert (st1, out1) == (r_st[i:i + 1], r_out[i:i + 1]), i
            cur += st1 == b"\0"
        assert len(ns) == len(db) and 128 * cur == r_drawn
        ns.close()
    # device memory (offsets stay on the host): refund with pre-drawn sequential bytes, redeem through the callback
    blob = b"".join(msgs)
    offs = np.zeros(N + 1, np.uint64); offs[1:] = np.cumsum([len(x) for x in msgs], dtype=np.uint64)
    d = lambda b: torch.from_numpy(np.frombuffer(b, np.uint8).copy()).cuda()
    d_blob, d_rng = d(blob + b"\0"), d(stream)
    d_st = torch.full((N,), 99, dtype=torch.uint8, device="cuda"); d_out = torch.full((ml * N,), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.wire_ptr("refund", sk, N, capi.MEM_DEVICE, d_blob.data_ptr(), offs.ctypes.data, d_rng.data_ptr(), capi.RNG_SEQUENTIAL, d_out.data_ptr(), d_st.data_ptr())
    flat = b"".join(x if x else bytes(ml) for x in want_out)
    assert d_st.cpu().numpy().tobytes() == want_st and d_out.cpu().numpy().tobytes() == flat
    ns = capi.NullifierSet(4 * N)
    g = capi.ReplayRng(stream)
    d_st.fill_(99); d_out.fill_(9); torch.cuda.synchronize()
    eng.wire_ptr("redeem", sk, N, capi.MEM_DEVICE, d_blob.data_ptr(), offs.ctypes.data, g.ptr, capi.RNG_CALLBACK, d_out.data_ptr(), d_st.data_ptr(), nullifier_set=ns)
    assert d_st.cpu().numpy().tobytes() == r_st and d_out.cpu().numpy().tobytes() == b"".join(x if x else bytes(ml) for x in r_out) and g.pos == r_drawn
    ns.close()
    # the failure contract holds on wire bytes too: the signature step fails after the nullifiers were recorded
    ns = capi.NullifierSet(4 * N)
    assert eng.lib.act_debug_fail_next_signs(eng.ctx, 1) == 0
    rc, st3, out3 = eng.redeem_cbor(ns, sk, msgs, stream, capi.RNG_SEQUENTIAL, raw=True)
    assert rc != 0 and st3 == bytes(251 if s == 0 else s for s in r_st) and not any(out3) and len(ns) == len(db)
    ns.close()
    assert eng.refund_cbor(sk, [], stream) == (b"", [])
    assert eng.secret_residue() == 0


def test_wire_calls_through_the_node(engine_factory, oracle, bench_params):
    from act_amd import capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=4)
    sk = eng.private_key_random(shake("wn-sk", 64))
    octx = oracle.ctx(bench_params, L)
    msgs = _messages(eng, sk, L, 11, "wn")
    N = len(msgs)
    stream = shake("wn-rng", 128 * N)
    want_st, want_out, drawn = _loop(octx, sk, L, msgs, stream)
    db = set()
    r_st, r_out, r_drawn = _loop(octx, sk, L, msgs, stream, db)
    node = capi.Node(bench_params, L, devices=(0, 0, 0), max_batch=3)
    ns = capi.NodeNullifierSet(4 * N, devices=(0, 0))
    try:
        for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
            node.set_transcript_mode(mode)
            g = capi.ReplayRng(stream)
            assert node.refund_cbor(sk, msgs, g, capi.RNG_CALLBACK) == (want_st, want_out) and g.draws == [drawn]
            assert node.refund_cbor(sk, msgs, stream, capi.RNG_SEQUENTIAL) == (want_st, want_out)
            stv, kp, nul = node.verify_spend_cbor_keys(sk, msgs)
            assert stv == want_st and node.refund_sign_cbor(sk, kp, stv, stream) == (want_st, want_out)
        g = capi.ReplayRng(stream)
        assert node.redeem_cbor(ns, sk, msgs, g, capi.RNG_CALLBACK) == (r_st, r_out) and g.pos == r_drawn and len(ns) == len(db)
        # small calls on a node: one context each, same bytes
        for i in (0, 1, 2, N - 1):
            assert node.refund_cbor(sk, [msgs[i]], stream[:128], capi.RNG_SEQUENTIAL) == _loop(octx, sk, L, [msgs[i]], stream)[:2]
    finally:
        ns.close(); node.close()


def test_settle_windows_and_sparse_non_canonical_messages(engine_factory, bench_params, monkeypatch):
    """untrusted clients choose the encoding.  Two non-canonical messages at the two ends of a device-memory batch must not
    make the library copy everything between them to the host: the flagged messages are read one by one when they lie far apart.
    And more flagged messages than one settle window (4 096) are settled window by window."""
    import torch
    from act_amd import capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=4096)
    sk = eng.private_key_random(shake("sw-sk", 64))
    base = _proofs(eng, sk, 8, "sw")
    pb = eng.proof_bytes
    canon = eng.cbor_encode("SpendProof", base)
    loose = [b"\xbf" + c[1:] + b"\xff" for c in canon]                 # indefinite-length map: same content, not canonical
    n = 6000
    msgs = [canon[i % 8] for i in range(n)]
    msgs[0] = loose[0]; msgs[n - 1] = loose[5]
    bad = bytearray(loose[3]); bad[40] ^= 1                             # a tampered non-canonical one (the charge): parses, fails the proof
    msgs[n - 2] = bytes(bad)
    blob = b"".join(msgs)
    offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum([len(x) for x in msgs], dtype=np.uint64)
    d_blob = torch.from_numpy(np.frombuffer(blob + b"\0", np.uint8).copy()).cuda()
    d_st = torch.full((n,), 99, dtype=torch.uint8, device="cuda"); d_kp = torch.zeros(32 * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    eng.verify_spend_cbor_ptr(sk, n, capi.MEM_DEVICE, d_blob.data_ptr(), offs.ctypes.data, d_st.data_ptr(), d_kp.data_ptr())
    st = d_st.cpu().numpy()
    assert st[n - 2] == 7 and st.sum() == 7
    kp = d_kp.cpu().numpy().tobytes()
    st_ref, kp_ref = eng.verify_spend(sk, base, True)
    which = {0: 0, 1: 1, 9: 1, n - 3: (n - 3) % 8, n - 1: 5}                    # lane -> the base proof it carries
    assert all(kp[32 * i:32 * i + 32] == kp_ref[32 * b:32 * b + 32] for i, b in which.items()) and kp[32 * (n - 2):32 * (n - 1)] == bytes(32)
    # every message non-canonical: 6 000 flagged = two settle windows
    msgs2 = [loose[i % 8] for i in range(n)]
    msgs2[4500] = bytes(bad)
    st2, kp2 = eng.verify_spend_cbor(sk, msgs2, True)
    assert st2[4500] == 7 and sum(st2) == 7 and kp2[32 * 4499:32 * 4500] == kp_ref[32 * (4499 % 8):32 * (4499 % 8) + 32]


def test_a_failing_generator_fails_the_call_and_signs_nothing(engine_factory, bench_params):
    """ACT_RNG_CALLBACK could not report a failed draw -- act_rng_draw_fn returned void, the library zero-filled its
    buffer and signed with e = alpha = 0 (z = gamma * x: the issuer's key falls out of one published refund), and ctypes swallows an
    exception raised inside a callback, so a Python generator that ran dry looked exactly like that.  Now draw() returns int: a
    generator that raises (the binding's trampoline catches it and reports 1), one that returns too few bytes, and one that is
    exhausted all fail the CALL with ACT_ERR_RNG, and not one Refund message is emitted."""
    from act_amd import api, capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=6)
    sk = eng.private_key_random(shake("rf-sk", 64))
    msgs = eng.cbor_encode("SpendProof", _proofs(eng, sk, 5, "rf"))
    lib, ml = eng.lib, eng.cbor_size("Refund")

    def call(source_ptr, keep):
        n = len(msgs); p0, k0, offs = capi._msgs(msgs); ps, ks = capi._in(sk, 64)
        st = np.zeros(n, np.uint8); out = np.full(ml * n, 0x55, np.uint8)
        rc = lib.act_refund_cbor_batch(eng.ctx, n, capi.MEM_HOST, ps, p0, offs.ctypes.data, source_ptr, capi.RNG_CALLBACK, out.ctypes.data, st.ctypes.data)
        return rc, out.tobytes()

    class Raises:
        def fill_bytes(self, n):
            raise RuntimeError("entropy source unavailable")

    class Short:
        def fill_bytes(self, n):
            return bytes(n - 1)

    for gen in (Raises(), Short()):
        cb = capi.rng_trampoline(gen.fill_bytes)
        src = capi.RngSource(cb, None)
        import ctypes as C
        rc, out = call(C.addressof(src), (cb, src))
        assert rc == 5 and len(cb.errors) == 1, (rc, cb.errors)              # ACT_ERR_RNG; the exception is kept for the caller
        assert b"\xa4" not in out[::ml], "a Refund message was framed with nonces nobody drew"
    dry = capi.ReplayRng(shake("rf-rng", 128 * 5 - 1))                         # one byte short of what five signatures need
    with pytest.raises(capi.ActError, match="ACT_ERR_RNG"):
        eng.refund_cbor(sk, msgs, dry, capi.RNG_CALLBACK)
    assert dry.pos == 0 and eng.secret_residue() == 0
    # the same generator with enough bytes: five refunds, 640 bytes drawn
    ok = capi.ReplayRng(shake("rf-rng", 128 * 5))
    st, outm = eng.refund_cbor(sk, msgs, ok, capi.RNG_CALLBACK)
    assert st == bytes(5) and all(len(x) == ml for x in outm) and ok.pos == 640
    # redeem: the nullifiers are recorded by the time the generator is asked; the lanes say that their refund is owed
    ns = capi.NullifierSet(64)
    rc, st, outb = eng.redeem_cbor(ns, sk, msgs, capi.ReplayRng(b""), capi.RNG_CALLBACK, raw=True)
    assert rc == 5 and st == bytes([251] * 5) and outb == bytes(ml * 5) and len(ns) == 5
    ns.close()
