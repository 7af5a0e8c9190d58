"""The reference's own behavioural tests (/root/reference/src/tests.rs), restated against the API mirror
(act_amd.api: same type and method names) running on the HIP engine."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from act_amd import api
    params = api.Params.new("test-org", "test-service", "test", "2024-01-01")
    rng = api.OsRng()
    sk = api.PrivateKey.random(rng, params)
    return api, params, rng, sk


def issue_token(api, params, rng, sk, c):
    pre = api.PreIssuance.random(rng, params)
    req = pre.request(params, rng)
    resp = sk.issue(params, req, c, rng)
    return pre.to_credit_token(params, sk.public(), req, resp)


def test_issuance(env):                                   # src/tests.rs:52-77
    api, params, rng, sk = env
    for _ in range(5):
        tok = issue_token(api, params, rng, sk, 20)
        assert api.scalar_to_u128(tok.credits()) == 20


def test_full_cycle_and_sequential_spends(env):           # :79-125, :259-337
    api, params, rng, sk = env
    tok = issue_token(api, params, rng, sk, 40)
    seen = set()
    for charge, remaining in ((20, 20), (5, 15), (15, 0)):
        proof, prerefund = tok.prove_spend(params, charge, rng)
        assert api.scalar_to_u128(proof.charge()) == charge
        assert proof.nullifier() == tok.nullifier() and proof.nullifier() not in seen
        seen.add(proof.nullifier())
        refund = sk.refund(params, proof, rng)
        tok = prerefund.to_credit_token(params, proof, refund, sk.public())
        assert api.scalar_to_u128(tok.credits()) == remaining


def test_zero_spend_and_zero_credit(env):                 # :377-426, :875-914
    api, params, rng, sk = env
    tok = issue_token(api, params, rng, sk, 0)
    proof, prerefund = tok.prove_spend(params, 0, rng)
    tok2 = prerefund.to_credit_token(params, proof, sk.refund(params, proof, rng), sk.public())
    assert api.scalar_to_u128(tok2.credits()) == 0


def test_large_amounts(env):                              # :641-689, :1007-1059
    api, params, rng, sk = env
    top = 2**128 - 1
    tok = issue_token(api, params, rng, sk, top)
    proof, prerefund = tok.prove_spend(params, 1, rng)
    tok2 = prerefund.to_credit_token(params, proof, sk.refund(params, proof, rng), sk.public())
    assert api.scalar_to_u128(tok2.credits()) == top - 1


def test_remaining_balance_bit_patterns(env):              # bits_of_ (src/tests.rs:521-567) seen through the range proof
    """The prover decomposes the remaining balance m = c - s into bits (src/lib.rs:902-915, :996); the patterns of the
    reference's own bits_of test must all give proofs the issuer accepts, and the refunded token carries exactly m."""
    api, params, rng, sk = env
    for m_ in (2**128 - 1, 0, 0b001, 0b100000000, 7, int("10" * 64, 2), int("01" * 64, 2)):
        s = 5 if m_ + 5 < 2**128 else 0
        tok = issue_token(api, params, rng, sk, m_ + s)
        proof, prerefund = tok.prove_spend(params, s, rng)
        tok2 = prerefund.to_credit_token(params, proof, sk.refund(params, proof, rng), sk.public())
        assert api.scalar_to_u128(tok2.credits()) == m_


def test_overspend_is_rejected(env):                      # :339-375
    api, params, rng, sk = env
    tok = issue_token(api, params, rng, sk, 10)
    proof, _ = tok.prove_spend(params, 11, rng)          # the prover still emits a proof
    with pytest.raises(api.Error) as e:
        sk.refund(params, proof, rng)
    assert e.value.name == "InvalidClientSpendProof"


def test_invalid_proofs_and_requests(env):                # :570-639, :850-873
    api, params, rng, sk = env
    pre = api.PreIssuance.random(rng, params)
    req = pre.request(params, rng)
    bad = bytearray(req.record); bad[64] ^= 1
    with pytest.raises(api.Error) as e:
        sk.issue(params, api.IssuanceRequest(bytes(bad)), 20, rng)
    assert e.value.name == "InvalidIssuanceRequestProof"
    tok = issue_token(api, params, rng, sk, 20)
    proof, _ = tok.prove_spend(params, 5, rng)
    bad = bytearray(proof.record); bad[32] ^= 1
    with pytest.raises(api.Error) as e:
        sk.refund(params, api.SpendProof(bytes(bad)), rng)
    assert e.value.name == "InvalidClientSpendProof"
    bad = bytearray(proof.record); bad[64:96] = bytes(32)
    with pytest.raises(api.Error) as e:
        sk.refund(params, api.SpendProof(bytes(bad)), rng)
    assert e.value.name == "IdentityPointError"


def test_client_rejects_tampered_responses(env):          # :691-720, :780-848
    api, params, rng, sk = env
    pre = api.PreIssuance.random(rng, params)
    req = pre.request(params, rng)
    resp = sk.issue(params, req, 20, rng)
    for off in (32, 64, 96):
        bad = bytearray(resp.record); bad[off] ^= 1
        with pytest.raises(api.Error) as e:
            pre.to_credit_token(params, sk.public(), req, api.IssuanceResponse(bytes(bad)))
        assert e.value.name == "InvalidIssuanceResponseProof"
    tok = pre.to_credit_token(params, sk.public(), req, resp)
    proof, prerefund = tok.prove_spend(params, 3, rng)
    refund = sk.refund(params, proof, rng)
    for off in (32, 64, 96):
        bad = bytearray(refund.record); bad[off] ^= 1
        with pytest.raises(api.Error) as e:
            prerefund.to_credit_token(params, proof, api.Refund(bytes(bad)), sk.public())
        assert e.value.name == "InvalidRefundProof"


def test_multiple_issuers_are_independent(env):           # :1997
    api, params, rng, sk = env
    sk2 = api.PrivateKey.random(rng, params)
    tok = issue_token(api, params, rng, sk, 30)
    proof, _ = tok.prove_spend(params, 4, rng)
    with pytest.raises(api.Error):
        sk2.refund(params, proof, rng)
    sk.refund(params, proof, rng)


def test_sequential_rng_is_consumed_only_by_accepted_lanes(env):   # src/lib.rs:638-643, :842-846
    api, params, _, sk = env
    stream = api.ByteStreamRng(os.urandom(128 * 4 + 64 * 600 * 3))
    pres = [api.PreIssuance.random(stream, params) for _ in range(3)]
    reqs = [p.request(params, stream) for p in pres]
    bad = bytearray(reqs[1].record); bad[70] ^= 1
    reqs[1] = api.IssuanceRequest(bytes(bad))
    pos = stream.pos
    out = sk.issue_batch(params, reqs, [5, 6, 7], stream)
    assert isinstance(out[1], api.Error) and not isinstance(out[0], api.Error)
    assert stream.pos == pos + 2 * 128                     # two accepted lanes drew 2 scalars each


def test_a_rejected_item_leaves_any_generator_untouched(env):   # VERDICT r5 #1b; src/lib.rs:638-643, :842-846
    """The crate draws e, alpha only after its checks have passed, so `issue` / `refund` of a REJECTED item never call the generator --
    whatever kind of generator it is (round 5's mirror drew 128 bytes up front unless the generator could peek, and the Rust
    binding's `*_eager` methods did the same).  A generator that only counts its calls: none for a rejected item, ONE fill_bytes(128 k)
    for the k accepted lanes of a batch, single-item calls included."""
    api, params, rng, sk = env

    class Counting:
        def __init__(self):
            self.calls = []

        def fill_bytes(self, n):
            self.calls.append(n)
            return os.urandom(n)

    pre = api.PreIssuance.random(rng, params)
    req = pre.request(params, rng)
    bad = bytearray(req.record); bad[70] ^= 1
    g = Counting()
    with pytest.raises(api.Error):
        sk.issue(params, api.IssuanceRequest(bytes(bad)), 9, g)
    assert g.calls == []
    resp = sk.issue(params, req, 9, g)
    assert g.calls == [128]
    tok = pre.to_credit_token(params, sk.public(), req, resp)
    proof, _ = tok.prove_spend(params, 4, rng)
    tampered = bytearray(proof.record); tampered[40] ^= 1
    g = Counting()
    with pytest.raises(api.Error):
        sk.refund(params, api.SpendProof(bytes(tampered), proof.nbits), g)
    assert g.calls == []
    out = sk.refund_batch(params, [api.SpendProof(bytes(tampered), proof.nbits), proof], g)
    assert isinstance(out[0], api.Error) and not isinstance(out[1], api.Error) and g.calls == [128]


def test_cbor_round_trips_of_every_type(env):             # src/tests.rs:1450-1496, 1776-1860, 2216-2233
    """prop_cbor_round_trip_* and prop_cbor_encoding_canonical, restated: to_cbor / from_cbor of the nine wire and state types
    through the API mirror; encoding twice and re-encoding the decoded value give the same bytes; CborError on broken input."""
    api, params, rng, sk = env
    pre = api.PreIssuance.random(rng, params)
    req = pre.request(params, rng)
    resp = sk.issue(params, req, 33, rng)
    tok = pre.to_credit_token(params, sk.public(), req, resp)
    proof, prerefund = tok.prove_spend(params, 11, rng)
    refund = sk.refund(params, proof, rng)
    for v in (req, resp, tok, proof, prerefund, refund, pre, sk, sk.public()):
        b1, b2 = v.to_cbor(), v.to_cbor()
        assert b1 == b2                                                  # canonical: encoding twice
        back = type(v).from_cbor(b1)
        assert back.record == v.record and back.to_cbor() == b1          # decode and re-encode
        assert type(v).from_cbor(b1 + b"trailing").record == v.record    # ciborium reads one item
    assert len(proof.to_cbor()) == 18036
    with pytest.raises(api.CborError) as e:
        api.CreditToken.from_cbor(tok.to_cbor()[:-1])
    assert e.value.name == "Ciborium"
    with pytest.raises(api.CborError) as e:
        api.CreditToken.from_cbor(b"\x80")
    assert e.value.name == "InvalidStructure"
    bad = bytearray(tok.to_cbor()); bad[4:36] = b"\x01" + bytes(31)      # field 1 (A): not a Ristretto encoding
    with pytest.raises(api.CborError) as e:
        api.CreditToken.from_cbor(bytes(bad))
    assert e.value.name == "InvalidValue"
    # a token that went through CBOR spends like the original (src/tests.rs:1470-1483 uses the round-tripped token)
    tok2 = api.CreditToken.from_cbor(tok.to_cbor())
    p2, _ = tok2.prove_spend(params, 3, rng)
    sk.refund(params, p2, rng)


def test_redeem_batch_and_nullifier_db(env):              # examples/act.rs:62-73; NullifierDb of src/tests.rs:29-50, :127-207
    """The server loop as one call: double spends come back as DoubleSpendError, fresh spends as refunds the client accepts."""
    api, params, rng, sk = env
    db = api.NullifierDb(1 << 10)
    toks = [issue_token(api, params, rng, sk, 50) for _ in range(4)]
    spends = [t.prove_spend(params, 7, rng) for t in toks]
    proofs = [p for p, _ in spends] + [toks[1].prove_spend(params, 9, rng)[0]]      # the last one re-spends token 1
    res = sk.redeem_batch(params, db, proofs, rng)
    assert [isinstance(r, api.Refund) for r in res] == [True, True, True, True, False]
    assert isinstance(res[4], api.Error) and res[4].name == "DoubleSpendError" and len(db) == 4
    for (proof, prerefund), refund in zip(spends, res):
        new = prerefund.to_credit_token(params, proof, refund, sk.public())
        assert api.scalar_to_u128(new.credits()) == 43
    # the same nullifiers again, one at a time: all spent; a fresh one is not
    assert db.spend_batch([p.nullifier() for p in proofs[:4]]) == [False] * 4
    fresh = issue_token(api, params, rng, sk, 5)
    assert db.spend(fresh.nullifier()) is True and db.spend(fresh.nullifier()) is False and len(db) == 5


def test_wire_level_api(env):                             # rust/src/mi355x.rs refund_cbor_batch / redeem_cbor_batch; examples/act.rs:62-73
    """Bytes in, bytes out: the messages a client sends (`SpendProof::to_cbor`), the messages it gets back (`Refund::to_cbor`), the
    crate's two error families per lane, and a generator that is drawn for signed lanes only."""
    api, params, rng, sk = env
    toks = [issue_token(api, params, rng, sk, 30) for _ in range(5)]
    spends = [t.prove_spend(params, 4 + i, rng) for i, t in enumerate(toks)]
    msgs = [p.to_cbor(params) for p, _ in spends]
    bad = bytearray(msgs[1]); bad[40] ^= 1                          # the charge: parses, InvalidClientSpendProof
    wire = [msgs[0], bytes(bad), msgs[2][:200], b"\x80", msgs[3], msgs[0], msgs[4]]
    stream = api.ByteStreamRng(os.urandom(128 * len(wire)))
    res = sk.refund_cbor_batch(params, wire, stream)
    kinds = [type(r).__name__ if not isinstance(r, bytes) else "bytes" for r in res]
    assert kinds == ["bytes", "Error", "CborError", "CborError", "bytes", "bytes", "bytes"], kinds
    assert res[1].name == "InvalidClientSpendProof" and res[2].name == "Ciborium" and res[3].name == "InvalidStructure"
    assert stream.pos == 128 * 4                                    # drawn for the four signed messages only
    assert len(res[0]) == 141 and res[0] != res[5]                  # the same proof signed twice: two different nonces
    # what came back is what the struct-level call gives for the same bytes of the generator
    again = api.ByteStreamRng(stream.data)
    want = sk.refund_batch(params, [spends[0][0], api.SpendProof.from_cbor(bytes(bad), params), spends[3][0], spends[0][0], spends[4][0]], again)
    assert [r.to_cbor(params) for r in want if isinstance(r, api.Refund)] == [res[0], res[4], res[5], res[6]]
    # the client accepts the refund it unpacks from the wire
    new = spends[4][1].to_credit_token(params, spends[4][0], api.Refund.from_cbor(res[6], params), sk.public())
    assert api.scalar_to_u128(new.credits()) == 30 - 8
    # redeem on wire bytes: the repeat of message 0 is now a double spend
    db = api.NullifierDb(1 << 10)
    stream = api.ByteStreamRng(os.urandom(128 * len(wire)))
    red = sk.redeem_cbor_batch(params, db, wire, stream)
    assert [isinstance(r, bytes) for r in red] == [True, False, False, False, True, False, True]
    assert red[5].name == "DoubleSpendError" and len(db) == 3 and stream.pos == 128 * 3


def test_roofline_probes_run(engine_factory, bench_params):
    """The library's three roofline probes (bench.py: the MAD issue rate; the random 128-byte read rate on a fresh allocation and on a
    context's own table) return sane numbers."""
    from act_amd import capi
    rate, ms = capi.ubench_mad(0)
    assert 1e13 < rate < 6e13 and ms > 0
    gbps, ms = capi.ubench_random_read(0, 1, 2, 2)
    assert 50 < gbps < 20000 and ms > 0
    eng = engine_factory(bench_params, 8, max_batch=4)
    gbps, ms = eng.ubench_table_read(3, 2, 2)
    assert 50 < gbps < 20000 and ms > 0
