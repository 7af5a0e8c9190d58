"""GPU nullifier set against the sequential meaning of the reference tests' NullifierDb
(/root/reference/src/tests.rs:29-50: `if is_spent(k) { reject } else { insert(k) }` per spend, in order)."""
import os
import random

import numpy as np
import pytest

from conftest import shake, scb

pytestmark = pytest.mark.gpu


def sequential(db: set, keys, mask=None):
    out = []
    for i, k in enumerate(keys):
        if mask is not None and mask[i]:
            out.append(0); continue
        if k in db:
            out.append(1)
        else:
            db.add(k); out.append(0)
    return bytes(out)


def test_matches_sequential_hashset_semantics():
    from act_amd import capi
    r = random.Random(7)
    ns = capi.NullifierSet(capacity=200_000)
    db = set()
    pool = [shake("nul%d" % i, 32) for i in range(3000)]
    for rnd in range(6):
        n = [1, 17, 1000, 4096, 5000, 2][rnd]
        keys = [r.choice(pool) if r.random() < 0.6 else os.urandom(32) for _ in range(n)]     # plenty of intra- and inter-batch repeats
        mask = bytes(1 if r.random() < 0.1 else 0 for _ in range(n)) if rnd % 2 else None
        got = ns.check_and_insert(b"".join(keys), 32, mask)
        assert got == sequential(db, keys, mask), rnd
        assert len(ns) == len(db)
    # every lane the same nullifier: only lane 0 is fresh
    k = os.urandom(32)
    assert ns.check_and_insert(k * 500) == bytes([0] + [1] * 499)
    assert ns.check_and_insert(k * 3) == bytes([1, 1, 1])
    assert ns.check_and_insert(b"") == b""


def test_reads_nullifiers_straight_from_spend_proofs(engine_factory, bench_params):
    """double_spend_prevention (src/tests.rs:127-207): the same token spent twice is caught; rejected proofs are masked."""
    from act_amd import capi
    eng = engine_factory(bench_params, 128, max_batch=16, transcript=capi.TRANSCRIPT_DEVICE)
    sk = eng.private_key_random(shake("ns-sk", 64))
    n = 6
    pre = eng.pre_issuance_random(shake("ns-pre", 128 * n)); req = eng.request(pre, shake("ns-rq", 128 * n))
    st, resp = eng.issue(sk, req, scb(50) * n, shake("ns-ir", 128 * n))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, p1, _ = eng.prove_spend(tok, scb(5) * n, shake("ns-pr1", eng.prove_rng_bytes * n))
    st, p2, _ = eng.prove_spend(tok, scb(9) * n, shake("ns-pr2", eng.prove_rng_bytes * n))    # second spend of the same tokens
    pb = eng.proof_bytes
    batch = bytearray(p1 + p2[:pb * 3])
    batch[pb * 1 + 33] ^= 1                         # lane 1: tampered -> rejected by verification, must not burn the nullifier
    batch = bytes(batch)
    status = eng.verify_spend(sk, batch)
    assert list(status) == [0, 7, 0, 0, 0, 0, 0, 0, 0]
    ns = capi.NullifierSet(capacity=1000)
    spent = ns.check_and_insert(batch, stride=pb, skip_mask=status)
    assert list(spent) == [0, 0, 0, 0, 0, 0, 1, 0, 1]          # lanes 6, 8 re-spend tokens 0, 2; lane 7 = token 1 whose first spend was rejected
    assert len(ns) == 6
    assert list(ns.check_and_insert(p2, stride=pb)) == [1] * 6          # every token's nullifier is now in the set


def test_million_keys_on_device():
    import torch
    from act_amd import capi
    n = 1 << 20
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    keys = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    keys[n // 2:] = keys[: n // 2]                   # second half repeats the first half
    spent = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    ns = capi.NullifierSet(capacity=2 * n)
    torch.cuda.synchronize()                         # inputs written on torch's stream must be complete before the call
    ns.check_and_insert_dev(n, keys.data_ptr(), 32, 0, spent.data_ptr())
    torch.cuda.synchronize()
    assert int(spent[: n // 2].sum()) == 0 and int(spent[n // 2:].sum()) == n // 2 and len(ns) == n // 2
    ns.check_and_insert_dev(n, keys.data_ptr(), 32, 0, spent.data_ptr())
    torch.cuda.synchronize()
    assert int(spent.sum()) == n and len(ns) == n // 2


def test_keys_are_scalars_not_byte_strings():
    """A raw record may carry k + l (the verifier reduces it and accepts): it must be the same key as k, as in the
    reference's HashSet<Scalar> (a Rust Scalar is always canonical, src/cbor.rs:85)."""
    from act_amd import capi
    from conftest import ELL
    ns = capi.NullifierSet(capacity=1000)
    k = int.from_bytes(shake("ns-canon", 32), "little") % ELL
    le = lambda v: v.to_bytes(32, "little")
    assert list(ns.check_and_insert(le(k) + le(k + ELL))) == [0, 1]                      # same batch
    assert list(ns.check_and_insert(le(k + 2 * ELL) + le(k + 15 * ELL) + le(k))) == [1, 1, 1]   # across batches
    k2 = (k + 12345) % ELL
    assert list(ns.check_and_insert(le(k2 + 3 * ELL))) == [0]                            # first seen in non-canonical form
    assert list(ns.check_and_insert(le(k2) + le(k2 + ELL))) == [1, 1]
    assert len(ns) == 2
    # two sets with OS-drawn salts agree on the answers (the salt only moves slots around)
    a, b = capi.NullifierSet(capacity=100), capi.NullifierSet(capacity=100)
    keys = b"".join(shake("ns-salt%d" % (i % 7), 32) for i in range(20))
    assert a.check_and_insert(keys) == b.check_and_insert(keys)


def test_node_level_set_keeps_the_sequential_meaning():
    """act_node_nullifier_*: three per-device sets behind host-side routing answer like one sequential HashSet<Scalar>."""
    from act_amd import capi
    from conftest import ELL
    r = random.Random(11)
    ns = capi.NodeNullifierSet(capacity_per_device=100_000, devices=(0, 0, 0))
    db = set()
    pool = [(int.from_bytes(shake("nnul%d" % i, 32), "little") % ELL).to_bytes(32, "little") for i in range(2000)]
    for rnd in range(5):
        n = [1, 19, 3000, 4096, 2][rnd]
        keys = [r.choice(pool) for _ in range(n)]
        mask = bytes(1 if r.random() < 0.15 else 0 for _ in range(n)) if rnd % 2 else None
        assert ns.check_and_insert(b"".join(keys), 32, mask) == sequential(db, keys, mask), rnd
        assert len(ns) == len(db)
    k = int.from_bytes(shake("nnul-canon", 32), "little") % ELL
    le = lambda v: v.to_bytes(32, "little")
    assert list(ns.check_and_insert(le(k + 3 * ELL) + le(k) + le(k + ELL))) == [0, 1, 1]      # one scalar, three byte strings, one owner
    assert ns.check_and_insert(b"") == b""
