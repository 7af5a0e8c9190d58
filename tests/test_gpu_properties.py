"""Property tests of the reference (/root/reference/src/tests.rs:1283-2234, 8 cases each there) restated against
the HIP engine with seeded randomness, batched so each property is a handful of launches."""
import random

import pytest

from conftest import shake, scb, ELL

pytestmark = pytest.mark.gpu
CASES = 8


def lifecycle(eng, sk, tag, cs):
    n = len(cs)
    pre = eng.pre_issuance_random(shake(tag + "-pre", 128 * n))
    req = eng.request(pre, shake(tag + "-rq", 128 * n))
    st, resp = eng.issue(sk, req, b"".join(scb(c) for c in cs), shake(tag + "-ir", 128 * n))
    assert st == bytes(n)
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(n)
    return tok


def spend(eng, sk, tag, tok, ss):
    n = len(ss)
    st, proofs, prer = eng.prove_spend(tok, b"".join(scb(s) for s in ss), shake(tag + "-pr", eng.prove_rng_bytes * n))
    assert st == bytes(n)
    st, rf = eng.refund(sk, proofs, shake(tag + "-rr", 128 * n))
    st2, tok2 = eng.refund_to_credit_token(prer, proofs, rf, sk[32:])
    return st, proofs, prer, rf, st2, tok2


@pytest.fixture(scope="module")
def env(engine_factory, bench_params):
    from act_amd import capi
    eng = engine_factory(bench_params, 128, max_batch=16, transcript=capi.TRANSCRIPT_DEVICE)
    return eng, eng.private_key_random(shake("prop-sk", 64))


def test_prop_balance_preservation_and_sequential_spends(env):      # :1334, :1556, :1733, :2036
    eng, sk = env
    r = random.Random(1)
    cs = [r.randrange(1, 2**64) for _ in range(CASES)] + [2**128 - 1]
    tok = lifecycle(eng, sk, "bal", cs)
    remaining = list(cs)
    for rnd in range(3):
        ss = [r.randrange(0, c + 1) for c in remaining]
        st, proofs, prer, rf, st2, tok = spend(eng, sk, "bal%d" % rnd, tok, ss)
        assert st == bytes(len(cs)) and st2 == bytes(len(cs))
        remaining = [c - s for c, s in zip(remaining, ss)]
        for i, c in enumerate(remaining):
            assert int.from_bytes(tok[160 * i + 128:160 * i + 160], "little") == c
            assert tok[160 * i + 64:160 * i + 96] == prer[96 * i + 32:96 * i + 64]      # new nullifier k* (src/lib.rs:1249)
    ss = list(remaining)                                                                   # exhaust every token
    st, proofs, prer, rf, st2, tok = spend(eng, sk, "bal-final", tok, ss)
    assert st == bytes(len(cs)) and all(int.from_bytes(tok[160 * i + 128:160 * i + 160], "little") == 0 for i in range(len(cs)))


def test_prop_overspend_always_fails(env):                            # :1523
    eng, sk = env
    r = random.Random(2)
    cs = [r.randrange(0, 2**32) for _ in range(CASES)]
    tok = lifecycle(eng, sk, "over", cs)
    ss = [c + r.randrange(1, 2**32) for c in cs]
    st, proofs, prer, rf, st2, _ = spend(eng, sk, "over", tok, ss)
    assert st == bytes([7]) * CASES and rf == bytes(128 * CASES) and all(s != 0 for s in st2)


def test_prop_nullifiers_unique_and_deterministic(env):               # :1376, :1412, :2174
    eng, sk = env
    tok = lifecycle(eng, sk, "null", [100] * CASES)
    nulls = [tok[160 * i + 64:160 * i + 96] for i in range(CASES)]
    assert len(set(nulls)) == CASES
    st, p1, _ = eng.prove_spend(tok, scb(1) * CASES, shake("null-a", eng.prove_rng_bytes * CASES))
    st, p2, _ = eng.prove_spend(tok, scb(2) * CASES, shake("null-b", eng.prove_rng_bytes * CASES))
    pb = eng.proof_bytes
    for i in range(CASES):                                             # same token -> same nullifier, whatever the spend
        assert p1[pb * i:pb * i + 32] == p2[pb * i:pb * i + 32] == nulls[i]


def test_prop_determinism_and_params_separation(engine_factory, env):  # :1602, :1662, :722-748
    from act_amd import capi
    eng, sk = env
    tok = lifecycle(eng, sk, "det", [55] * 2)
    a = eng.prove_spend(tok, scb(5) * 2, shake("det-pr", eng.prove_rng_bytes * 2))
    b = eng.prove_spend(tok, scb(5) * 2, shake("det-pr", eng.prove_rng_bytes * 2))
    assert a == b
    h2 = capi.params_new("other-org", "bench-service", "bench-env", "2024-01-01")
    assert h2 != eng.h and capi.params_new("other-org", "bench-service", "bench-env", "2024-01-01") == h2
    eng2 = engine_factory(h2, 128, max_batch=16, transcript=capi.TRANSCRIPT_DEVICE)
    assert eng2.verify_spend(sk, a[1]) == bytes([7, 7])               # a proof under other params does not verify
    assert eng.verify_spend(sk, a[1]) == bytes(2)


def test_prop_token_tampering_detected(env):                          # :1898
    eng, sk = env
    tok = bytearray(lifecycle(eng, sk, "tamper", [500] * 4))
    tok[160 * 0 + 33] ^= 1          # e
    tok[160 * 1 + 129] ^= 1         # c
    tok[160 * 2 + 97] ^= 1          # r
    good = bytes(tok[160 * 3:160 * 4])
    st, proofs, _ = eng.prove_spend(bytes(tok), scb(7) * 4, shake("tamper-pr", eng.prove_rng_bytes * 4))
    assert list(eng.verify_spend(sk, proofs)) == [7, 7, 7, 0]
    # a token whose point is replaced by another valid point
    other = lifecycle(eng, sk, "tamper2", [500])
    t2 = other[:32] + good[32:]
    st, proofs, _ = eng.prove_spend(t2, scb(7), shake("tamper-pr2", eng.prove_rng_bytes))
    assert eng.verify_spend(sk, proofs) == bytes([7])


def test_prop_spend_proof_structure(env):                             # :1862
    eng, sk = env
    tok = lifecycle(eng, sk, "struct", [321] * 3)
    st, proofs, prer = eng.prove_spend(tok, b"".join(scb(v) for v in (0, 21, 321)), shake("struct-pr", eng.prove_rng_bytes * 3))
    pb = eng.proof_bytes
    assert pb == 32 * (14 + 4 * 128)
    for i, s in enumerate((0, 21, 321)):
        rec = proofs[pb * i:pb * i + pb]
        assert rec[:32] == tok[160 * i + 64:160 * i + 96] and rec[:32] != bytes(32)          # k echoed, non-zero
        assert int.from_bytes(rec[32:64], "little") == s                                        # s echoed
        assert rec[64:96] != bytes(32)                                                          # A' is not the identity
        for f in range(0, pb, 32):                                                              # scalars canonical
            if not (64 <= f < 32 * (4 + 128)):
                assert int.from_bytes(rec[f:f + 32], "little") < ELL
        assert int.from_bytes(prer[96 * i + 64:96 * i + 96], "little") == 321 - s              # PreRefund.m = c - s
