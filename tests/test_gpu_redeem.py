"""act_redeem_batch / act_node_redeem_batch: the issuer's whole redemption step (verify -> nullifier look-up and record -> sign) must
equal the sequential loop a server built on the crate runs (examples/act.rs:62-73 with verification first): statuses incl.
DoubleSpendError, refund bytes under ACT_RNG_SEQUENTIAL (one rng stream, drawn from only by lanes that are signed), and the set's
contents afterwards -- checked against the C oracle + a Python set, from host memory, from device memory, and over a node handle."""
import numpy as np
import pytest

from conftest import ELL, shake, scb

pytestmark = pytest.mark.gpu


def _make(eng, sk, n, tag):
    pre = eng.pre_issuance_random(shake(tag + "-pre", 128 * n)); req = eng.request(pre, shake(tag + "-rq", 128 * n))
    st, resp = eng.issue(sk, req, b"".join(scb(30 + i) for i in range(n)), shake(tag + "-ir", 128 * n))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(i % 20) for i in range(n)), shake(tag + "-pr", eng.prove_rng_bytes * n))
    assert st == bytes(n)
    pb = eng.proof_bytes
    return [proofs[pb * i:pb * i + pb] for i in range(n)]


def _sequential_loop(octx, sk, proofs, rng, spent):
    """the reference loop: verification, then the nullifier database, then the signature with the next 128 rng bytes"""
    st_out, rf_out, cur = [], [], 0
    for p in proofs:
        st, _ = octx.verify_spend(sk, p)
        k = int.from_bytes(p[:32], "little") % ELL                 # HashSet<Scalar>: the reduced scalar
        if st == 0 and k in spent:
            st = 3                                                 # Error::DoubleSpendError
        if st:
            st_out.append(st); rf_out.append(bytes(128)); continue
        spent.add(k)
        s2, rf = octx.refund(sk, p, rng[128 * cur:128 * cur + 128]); cur += 1
        assert s2 == 0
        st_out.append(0); rf_out.append(rf)
    return bytes(st_out), b"".join(rf_out), cur


def _batches(eng, sk, L):
    good = _make(eng, sk, 12, "rd%d" % L)
    t = [bytearray(p) for p in good]
    t[2][32] ^= 1                                                   # charge           -> 7, nullifier NOT recorded
    t[5][64:96] = bytes(32)                                         # A' = identity     -> 6
    t[7][32 * (4 + 1):32 * (4 + 2)] = b"\xff" * 32                  # Com_1 undecodable -> 255
    first = [bytes(x) for x in t] + [good[0], good[3]]             # lanes 12, 13 repeat lanes 0, 3 inside the batch -> 3
    alias = bytearray(good[4]); alias[0:32] = (int.from_bytes(good[4][:32], "little") + ELL).to_bytes(32, "little")
    second = [good[2], bytes(alias), good[9]] + _make(eng, sk, 3, "rd%d-b" % L)    # the honest form of lane 2 is still spendable;
    return first, second                                            # k + l and lane 9 were recorded by the first batch -> 3


@pytest.mark.parametrize("L", [8, 128])
def test_redeem_equals_the_sequential_loop(engine_factory, oracle, bench_params, L):
    import torch
    from act_amd import capi
    eng = engine_factory(bench_params, L, max_batch=5, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("rd-sk", 64))
    octx = oracle.ctx(bench_params, L)
    first, second = _batches(eng, sk, L)
    rng = shake("rd-rng%d" % L, 128 * 32)
    # oracle: one database across both batches
    db = set()
    want1 = _sequential_loop(octx, sk, first, rng, db)
    want2 = _sequential_loop(octx, sk, second, rng[128 * want1[2]:], db)
    assert list(want1[0]) == [0, 0, 7, 0, 0, 6, 0, 255, 0, 0, 0, 0, 3, 3] and list(want2[0]) == [0, 3, 3, 0, 0, 0]
    # host memory
    ns = capi.NullifierSet(1000)
    st1, rf1 = eng.redeem(ns, sk, b"".join(first), rng, capi.RNG_SEQUENTIAL)
    assert (st1, rf1) == want1[:2] and len(ns) == 9
    st2, rf2 = eng.redeem(ns, sk, b"".join(second), rng[128 * want1[2]:], capi.RNG_SEQUENTIAL)
    assert (st2, rf2) == want2[:2] and len(ns) == 13 == len(db)
    ns.close()
    # device memory
    ns = capi.NullifierSet(1000)
    d = lambda b: torch.from_numpy(np.frombuffer(b, np.uint8).copy()).cuda()
    for batch, want, r in ((first, want1, rng), (second, want2, rng[128 * want1[2]:])):
        n = len(batch)
        d_p, d_r = d(b"".join(batch)), d(r)
        d_o = torch.full((n * 128,), 7, dtype=torch.uint8, device="cuda"); d_s = torch.full((n,), 99, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        eng.redeem_dev(ns, sk, n, d_p.data_ptr(), d_r.data_ptr(), capi.RNG_SEQUENTIAL, d_o.data_ptr(), d_s.data_ptr())
        assert d_s.cpu().numpy().tobytes() == want[0] and d_o.cpu().numpy().tobytes() == want[1]
    assert len(ns) == 13
    ns.close()
    # per-lane rng: lane i signs with slice i
    ns = capi.NullifierSet(1000)
    st, rf = eng.redeem(ns, sk, b"".join(first), rng, capi.RNG_PER_LANE)
    assert st == want1[0]
    for i, p in enumerate(first):
        if st[i] == 0:
            assert octx.refund(sk, p, rng[128 * i:128 * i + 128]) == (0, rf[128 * i:128 * i + 128])
    ns.close()
    # one item per call (the server's own rhythm: examples/act.rs:62-73), its 128 bytes handed over with it: the refund is computed as
    # one call with the signature beside the verification, THEN the nullifier decides (nullifier_impl.inc) -- the same answers and the
    # same database as the loop, item by item
    ns = capi.NullifierSet(1000)
    cur = 0
    for i, p in enumerate(first + second):
        want = (want1 if i < len(first) else want2)
        j = i if i < len(first) else i - len(first)
        st, rf = eng.redeem(ns, sk, p, rng[128 * cur:128 * cur + 128], capi.RNG_SEQUENTIAL)
        assert (st, rf) == (want[0][j:j + 1], want[1][128 * j:128 * j + 128]), i
        cur += st == b"\0"
    assert len(ns) == 13 and cur == want1[2] + want2[2]
    ns.close()
    # ... and a few items with per-lane slices (the same path, n <= max_batch here)
    ns = capi.NullifierSet(1000)
    st, rf = eng.redeem(ns, sk, b"".join(first[:5]), rng, capi.RNG_PER_LANE)
    assert st == want1[0][:5]
    for i, p in enumerate(first[:5]):
        assert rf[128 * i:128 * i + 128] == (octx.refund(sk, p, rng[128 * i:128 * i + 128])[1] if st[i] == 0 else bytes(128))
    assert eng.redeem(ns, sk, b"".join(first[:5]), rng, capi.RNG_PER_LANE)[0] == bytes(3 if v == 0 else v for v in want1[0][:5])      # everything again: double spends
    # empty batch; a set on a context's own device only
    assert eng.redeem(ns, sk, b"", b"") == (b"", b"")
    ns.close()
    ns = capi.NullifierSet(1000)
    eng.redeem(ns, sk, b"".join(first), rng, capi.RNG_PER_LANE)
    assert eng.redeem(ns, sk, b"", b"") == (b"", b"") and len(ns) == 9
    ns.close()
    assert eng.secret_residue() == 0


def test_redeem_over_a_node(engine_factory, oracle, bench_params):
    from act_amd import capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=5, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("rd-sk", 64))
    octx = oracle.ctx(bench_params, L)
    first, second = _batches(eng, sk, L)
    rng = shake("rdn-rng", 128 * 32)
    db = set()
    want1 = _sequential_loop(octx, sk, first, rng, db)
    want2 = _sequential_loop(octx, sk, second, rng[128 * want1[2]:], db)
    node = capi.Node(bench_params, L, devices=(0, 0, 0), max_batch=3, transcript=capi.TRANSCRIPT_DEVICE)
    ns = capi.NodeNullifierSet(1000, devices=(0, 0))
    try:
        assert node.redeem(ns, sk, b"".join(first), rng) == want1[:2]
        assert node.redeem(ns, sk, b"".join(second), rng[128 * want1[2]:]) == want2[:2]
        assert len(ns) == len(db)
    finally:
        ns.close(); node.close()


def test_redeem_failures_keep_every_decision(engine_factory, oracle, bench_params, monkeypatch):
    """a redeem call that fails after verification must still say, lane by lane, what happened.
    (a) the set has no room for the batch: refused as a whole -- verified lanes 252 (not recorded, not signed), rejected lanes keep
        their verdict, the set is unchanged, and the same batch goes through on a set that is large enough;
    (b) the signature step fails after the nullifiers were recorded: lanes that were to be signed report 251 with a zero record,
        their nullifiers ARE in the set, and verify + refund_sign on exactly those lanes yields the refunds of the loop."""
    import torch
    from act_amd import capi
    L = 8
    eng = engine_factory(bench_params, L, max_batch=5, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("rd-sk", 64))
    octx = oracle.ctx(bench_params, L)
    first, _ = _batches(eng, sk, L)
    rng = shake("rdf-rng", 128 * 32)
    want = _sequential_loop(octx, sk, first, rng, set())
    verdict = [7 if i == 2 else 6 if i == 5 else 255 if i == 7 else 0 for i in range(len(first))]
    # (a) capacity: a set made for 4 nullifiers holds 1024 slots / 2 = 512: fill it to the brim first
    small = capi.NullifierSet(4)
    filler = b"".join((10**9 + i).to_bytes(32, "little") for i in range(505))
    assert small.check_and_insert(filler) == bytes(505)
    for mem in ("host", "device"):
        if mem == "host":
            with pytest.raises(capi.ActError, match="capacity"):
                eng.redeem(small, sk, b"".join(first), rng, capi.RNG_SEQUENTIAL)
        else:
            d = lambda b: torch.from_numpy(np.frombuffer(b, np.uint8).copy()).cuda()
            n = len(first)
            d_p, d_r = d(b"".join(first)), d(rng)
            d_o = torch.full((n * 128,), 7, dtype=torch.uint8, device="cuda"); d_s = torch.full((n,), 99, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            with pytest.raises(capi.ActError, match="capacity"):
                eng.redeem_dev(small, sk, n, d_p.data_ptr(), d_r.data_ptr(), capi.RNG_SEQUENTIAL, d_o.data_ptr(), d_s.data_ptr())
            assert list(d_s.cpu().numpy()) == [v if v else 252 for v in verdict]
            assert not d_o.cpu().numpy().any()
        assert len(small) == 505
    # the host-memory form wrote its statuses too (capi raises before returning them: call the ABI directly)
    n = len(first)
    st = np.zeros(n, np.uint8); out = np.full(128 * n, 7, np.uint8)
    pr = np.frombuffer(b"".join(first), np.uint8); rg = np.frombuffer(rng, np.uint8); skb = np.frombuffer(sk, np.uint8)
    rc = eng.lib.act_redeem_batch(eng.ctx, small.h, n, capi.MEM_HOST, skb.ctypes.data, pr.ctypes.data, rg.ctypes.data, capi.RNG_SEQUENTIAL, out.ctypes.data, st.ctypes.data)
    assert rc != 0 and list(st) == [v if v else 252 for v in verdict] and not out.any()
    small.close()
    big = capi.NullifierSet(1000)
    assert eng.redeem(big, sk, b"".join(first), rng, capi.RNG_SEQUENTIAL) == want[:2]
    big.close()
    # (b) signature step fails (test hook): recorded, unsigned
    ns = capi.NullifierSet(1000)
    assert eng.lib.act_debug_fail_next_signs(eng.ctx, 1) == 0
    st = np.zeros(n, np.uint8); out = np.full(128 * n, 7, np.uint8)
    rc = eng.lib.act_redeem_batch(eng.ctx, ns.h, n, capi.MEM_HOST, skb.ctypes.data, pr.ctypes.data, rg.ctypes.data, capi.RNG_SEQUENTIAL, out.ctypes.data, st.ctypes.data)
    assert rc != 0 and b"signature step" in eng.lib.act_last_error(eng.ctx)
    assert list(st) == [251 if w == 0 else w for w in want[0]] and not out.any()
    assert len(ns) == sum(1 for w in want[0] if w == 0)                 # the nullifiers ARE recorded ...
    owed = [i for i in range(n) if st[i] == 251]
    sub = b"".join(first[i] for i in owed)
    st_v, kp = eng.verify_spend(sk, sub, want_kprime=True)              # ... so the refunds are signed, not redeemed again
    assert st_v == bytes(len(owed))
    rf = np.zeros(128 * len(owed), np.uint8); st2 = np.zeros(len(owed), np.uint8)
    kpa = np.frombuffer(kp, np.uint8); sin = np.zeros(len(owed), np.uint8)
    assert eng.lib.act_refund_sign_batch(eng.ctx, len(owed), capi.MEM_HOST, skb.ctypes.data, kpa.ctypes.data, sin.ctypes.data, rg.ctypes.data, capi.RNG_SEQUENTIAL,
                                         rf.ctypes.data, st2.ctypes.data) == 0
    assert rf.tobytes() == b"".join(want[1][128 * i:128 * i + 128] for i in owed)
    ns.close()
