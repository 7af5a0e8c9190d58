"""Parity tests proper: the HIP engine, called through the C ABI, against the committed golden fixtures and
against the C oracle on the same seeded inputs.  Bit-exact (integer / byte work): every record, every status."""
import pytest

from conftest import ELL, load_golden, shake, scb

pytestmark = pytest.mark.gpu

MODES = [0, 1]   # ACT_TRANSCRIPT_HOST, ACT_TRANSCRIPT_DEVICE


def test_params_new_matches_golden():
    from act_amd import capi
    for v in load_golden("primitives.json")["params"]:
        assert capi.params_new(*v["args"]).hex() == v["h"]
    u = shake("params-random", 192)
    import pymodel as m
    assert capi.params_random(u) == m.Params.random(m.ByteRng(u)).encoded()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["lifecycle_L128.json", "lifecycle_L64.json"])
def test_golden_lifecycle(engine_factory, name, mode):
    g = load_golden(name)
    L = g["L"]
    eng = engine_factory(bytes.fromhex(g["params"]), L, max_batch=4, transcript=mode)
    sk, sk2 = bytes.fromhex(g["sk"]), bytes.fromhex(g["sk_other"])
    assert eng.private_key_random(shake("golden-sk", 64)) == sk
    cases = g["cases"]
    n = len(cases)
    tag = lambda i: "L%d-case%d" % (L, i)
    cat = lambda f: b"".join(f(i) for i in range(n))
    pre = eng.pre_issuance_random(cat(lambda i: shake(tag(i) + "-pre", 128)))
    assert pre == cat(lambda i: bytes.fromhex(cases[i]["pre"]))
    req = eng.request(pre, cat(lambda i: shake(tag(i) + "-request", 128)))
    assert req == cat(lambda i: bytes.fromhex(cases[i]["request"]))
    st, resp = eng.issue(sk, req, cat(lambda i: scb(int(cases[i]["c"]))), cat(lambda i: shake(tag(i) + "-issue", 128)))
    assert st == bytes(n) and resp == cat(lambda i: bytes.fromhex(cases[i]["response"]))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(n) and tok == cat(lambda i: bytes.fromhex(cases[i]["token"]))
    st, proofs, prer = eng.prove_spend(tok, cat(lambda i: scb(int(cases[i]["s"]))), cat(lambda i: shake(tag(i) + "-prove", eng.prove_rng_bytes)))
    assert st == bytes(n) and prer == cat(lambda i: bytes.fromhex(cases[i]["prerefund"]))
    pb = eng.proof_bytes
    for i, c in enumerate(cases):
        if c["tamper"] is None:
            assert proofs[pb * i:pb * i + pb].hex() == c["proof"], i
    proofs = cat(lambda i: bytes.fromhex(cases[i]["proof"]))
    st, kp = eng.verify_spend(sk, proofs, True)
    assert list(st) == [c["status"] for c in cases]
    for i, c in enumerate(cases):
        # K' for accepted lanes; the output record of every rejected lane is zero (include/act_mi355x.h)
        assert kp[32 * i:32 * i + 32].hex() == (c["kprime"] if c["status"] == 0 else "00" * 32)
    st, rf = eng.refund(sk, proofs, cat(lambda i: shake(tag(i) + "-refund", 128)))
    assert list(st) == [c["status"] for c in cases]
    assert rf == cat(lambda i: bytes.fromhex(cases[i]["refund"]))
    st2, tok2 = eng.refund_to_credit_token(prer, proofs, rf, sk[32:])
    for i, c in enumerate(cases):
        if c["status"] == 0:
            assert st2[i] == 0 and tok2[160 * i:160 * i + 160].hex() == c["token2"]
        else:
            assert st2[i] != 0 and tok2[160 * i:160 * i + 160] == bytes(160)
    assert list(eng.verify_spend(sk2, proofs)) == [c["status_other_issuer"] for c in cases]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("L", [128, 64, 8, 100, 3])
def test_random_batches_against_oracle(engine_factory, oracle, bench_params, L, mode):
    """Seeded random batch with tampered / undecodable / identity / overspend lanes, ragged chunks (max_batch = 7)."""
    octx = oracle.ctx(bench_params, L)
    eng = engine_factory(bench_params, L, max_batch=7, transcript=mode)
    N = 23
    sk = octx.private_key_random(shake("pk-%d" % L, 64))
    pre = b"".join(octx.pre_issuance_random(shake("pre-%d-%d" % (L, i), 128)) for i in range(N))
    rq = shake("rq-%d" % L, 128 * N)
    req = eng.request(pre, rq)
    assert req == octx.request_batch(pre, rq, 8)
    bad_req = bytearray(req)
    bad_req[128 * 2 + 70] ^= 1          # k_bar
    bad_req[128 * 5 + 3] ^= 0x08        # K (almost surely undecodable)
    bad_req[128 * 9 + 40] ^= 1          # gamma
    bad_req = bytes(bad_req)
    cam = b"".join(scb((1 << min(L, 60)) // (i + 1) + i) for i in range(N))
    irng = shake("ir-%d" % L, 128 * N)
    for rng_mode in (0, 1):
        st, resp = eng.issue(sk, bad_req, cam, irng, rng_mode)
        cur = 0
        for i in range(N):
            slot = i if rng_mode == 0 else cur
            so, ro = octx.issue(sk, bad_req[128 * i:128 * i + 128], cam[32 * i:32 * i + 32], irng[128 * slot:128 * slot + 128])
            assert so == st[i] and ro == resp[160 * i:160 * i + 160], (rng_mode, i)
            cur += so == 0
    st, resp = eng.issue(sk, req, cam, irng)
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(N)
    amounts = [(1 << min(L, 60)) // (i + 1) + i for i in range(N)]
    spend = [a // 3 for a in amounts]
    spend[4] = amounts[4] + 1           # overspend
    spend[6] = amounts[6]               # spend everything
    spend[7] = 0
    s_b = b"".join(scb(v) for v in spend)
    prng = shake("pr-%d" % L, octx.prove_rng_bytes * N)
    st, proofs, prer = eng.prove_spend(tok, s_b, prng)
    po, pro = octx.prove_spend_batch(tok, s_b, prng, 8)
    assert st == bytes(N) and proofs == po and prer == pro
    pb = octx.proof_bytes
    t = bytearray(proofs)
    t[pb * 1 + 33] ^= 2                             # s
    t[pb * 3 + 64:pb * 3 + 96] = bytes(32)          # A' = identity
    t[pb * 8 + 32 * (4 + (L - 1)) + 9] ^= 0x40      # last Com
    t[pb * 10 + 32 * (12 + L) + 5] ^= 1             # gamma0[0]
    t[pb * 11 + 32 * (13 + 4 * L)] ^= 1             # s_bar
    t[pb * 12 + 96 + 1] ^= 0x20                     # B_bar
    t[pb * 13 + 32 * (4 + 2):pb * 13 + 32 * (4 + 3)] = bytes(32)      # Com_2 = identity: a valid point, the proof is just wrong
    t[pb * 14 + 96:pb * 14 + 128] = bytes(32)                         # B_bar = identity (the reference only checks A')
    t[pb * 15 + 32 * (12 + 2 * L):pb * 15 + 32 * (13 + 2 * L)] = b"\xff" * 32   # z[0][0] not canonical: reduced mod l on input
    t = bytes(t)
    rrng = shake("rr-%d" % L, 128 * N)
    st_o = octx.verify_spend_batch(sk, t, 8)
    st, kp = eng.verify_spend(sk, t, True)
    assert st == st_o
    assert {0, 6, 7, 255} <= set(st) or L <= 8
    trs = eng.last_spend_transcripts(7)            # last chunk: lanes 21, 22
    for k, i in enumerate(range(21, N)):
        so, kpo, tro = octx.verify_spend(sk, t[pb * i:pb * i + pb], True)
        assert trs[k] == tro and kp[32 * i:32 * i + 32] == (kpo if so == 0 else bytes(32))
    for rng_mode in (0, 1):
        st, rf = eng.refund(sk, t, rrng, rng_mode)
        cur = 0
        for i in range(N):
            slot = i if rng_mode == 0 else cur
            so, ro = octx.refund(sk, t[pb * i:pb * i + pb], rrng[128 * slot:128 * slot + 128])
            assert so == st[i] and ro == rf[128 * i:128 * i + 128], (rng_mode, i)
            cur += so == 0
    st, rf = eng.refund(sk, proofs, rrng)
    st_c, tok2 = eng.refund_to_credit_token(prer, proofs, rf, sk[32:])
    for i in range(N):
        so, to = octx.refund_to_credit_token(prer[96 * i:96 * i + 96], proofs[pb * i:pb * i + pb], rf[128 * i:128 * i + 128], sk[32:])
        assert so == st_c[i] and to == tok2[160 * i:160 * i + 160]


@pytest.mark.parametrize("L", [128, 64])
def test_crafted_scalars_reproduce_the_oracle_transcript(engine_factory, oracle, bench_params, L):
    """Recoding corner cases on the device: gamma (wave-uniform width-3 NAF) and gamma0_j / z_j* (per-lane radix-16 buckets,
    fixed-base windows) set to runs of ones, alternating digits, l-1, 0 and non-canonical values.  The proofs are
    invalid by construction (status 7 / 3), but every commitment the verifier recomputes (src/lib.rs:790-829) is in
    the transcript, which must equal the oracle's byte for byte."""
    octx = oracle.ctx(bench_params, L)
    pats = [0, 1, 3, 7, ELL - 1, ELL - 3, (1 << 252) - 1, 1 << 252] + [int(c * 63, 16) % ELL for c in "37bf5"]
    # the range kernel works on halved scalars: 2w mod l puts the digit string of w into its recoders.  Strings that send every
    # chain point into ONE bucket, or into one NAF accumulator at the closest allowed spacing (the d-free additions of
    # msm.h chain_bu_pre must never meet their exceptional cases)
    pats += [2 * w % ELL for w in [int(("%x" % d) * 63, 16) for d in range(1, 9)] + [int("08" * 31, 16), sum(1 << i for i in range(0, 252, 3)),
                                                                                   sum(3 << i for i in range(0, 249, 3))]]
    N = len(pats)
    eng = engine_factory(bench_params, L, max_batch=N, transcript=MODES[0])
    sk = octx.private_key_random(shake("cs-pk", 64))
    pre = octx.pre_issuance_random(shake("cs-pre", 128))
    req = eng.request(pre, shake("cs-rq", 128))
    st, resp = eng.issue(sk, req, scb(1000), shake("cs-ir", 128))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proof, _ = eng.prove_spend(tok, scb(321), shake("cs-pr", octx.prove_rng_bytes))
    assert st == b"\0"
    pb = octx.proof_bytes
    recs = []
    for n, v in enumerate(pats):
        t = bytearray(proof)
        t[32 * (4 + L):32 * (5 + L)] = scb(v)                                           # gamma
        for j in range(L):
            w = pats[(n + j) % N]
            t[32 * (12 + L + j):32 * (13 + L + j)] = scb(w)                             # gamma0_j
            t[32 * (12 + 2 * L + 2 * j):32 * (13 + 2 * L + 2 * j)] = scb(pats[(n + 2 * j + 1) % N])       # z_j0
        recs.append(bytes(t))
    nc = bytearray(recs[3]); nc[32 * (4 + L):32 * (5 + L)] = ((1 << 253) - 1).to_bytes(32, "little"); recs[3] = bytes(nc)   # gamma >= l
    batch = b"".join(recs)
    want = [octx.verify_spend(sk, r, True) for r in recs]
    try:
        for small_max in (8192, 0):          # the small-batch schedule (N <= max_batch: one chunk on four streams), then the pipelined one
            eng.set_small_batch_max(small_max)
            st, kp = eng.verify_spend(sk, batch, True)
            trs = eng.last_spend_transcripts(N)
            for i in range(N):
                so, kpo, tro = want[i]
                assert so == st[i] and so != 0, (small_max, i)
                assert trs[i] == tro, (small_max, i)
    finally:
        eng.set_small_batch_max(8192)


def test_empty_and_single_lane_batches(engine_factory, bench_params):
    eng = engine_factory(bench_params, 128, max_batch=7)
    sk = eng.private_key_random(shake("sk-empty", 64))
    assert eng.request(b"", b"") == b""
    assert eng.verify_spend(sk, b"") == b""
    assert eng.refund(sk, b"", b"") == (b"", b"")
    assert eng.issue(sk, b"", b"", b"") == (b"", b"")
    assert eng.prove_spend(b"", b"", b"") == (b"", b"", b"")


def test_workspace_that_cannot_be_allocated_fails_cleanly(bench_params):
    """A max_batch whose workspace exceeds the GPU's memory (2^22 records at L = 128 is about 0.9 TB) must come back as an
    error with a message, release what it had taken, and leave the device usable."""
    from act_amd import capi
    with pytest.raises(capi.ActError) as e:
        capi.Engine(bench_params, 128, max_batch=1 << 22)
    assert "hip" in str(e.value).lower() or "memory" in str(e.value).lower()
    with pytest.raises(capi.ActError) as e:                 # above the 32-bit lane-index limit: refused, not clamped
        capi.Engine(bench_params, 128, max_batch=(1 << 22) + 1)
    assert "ACT_ERR_ARG" in str(e.value) and "2^22" in str(e.value)
    eng = capi.Engine(bench_params, 128, max_batch=4)
    sk = eng.private_key_random(shake("oom-sk", 64))
    assert len(sk) == 64
    eng.close()


def test_device_memory_path_and_full_size_properties(engine_factory, bench_params):
    """BASELINE sizes through size-independent properties: 2^16 (config 2 count) and 2^20 (metric batch) tiled proofs
    resident in HBM; every valid lane accepted, every tampered lane (1 in 1024) rejected with the right code, and the
    result is independent of chunking."""
    import numpy as np
    import torch
    from act_amd import capi
    L, D = 128, 512
    eng = engine_factory(bench_params, L, max_batch=0, transcript=capi.TRANSCRIPT_DEVICE)      # 0 = library default (65536 per launch)
    sk = eng.private_key_random(shake("sk-big", 64))
    pre = eng.pre_issuance_random(shake("pre-big", 128 * D))
    req = eng.request(pre, shake("rq-big", 128 * D))
    cam = b"".join(scb(10 + (i * 7919) % 990) for i in range(D))
    st, resp = eng.issue(sk, req, cam, shake("ir-big", 128 * D))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    s_b = b"".join(scb(1 + (i * 31) % (10 + (i * 7919) % 990 - 1)) for i in range(D))
    st, proofs, _ = eng.prove_spend(tok, s_b, shake("pr-big", eng.prove_rng_bytes * D))
    assert st == bytes(D)
    pb = eng.proof_bytes
    host = np.frombuffer(proofs, np.uint8).reshape(D, pb)
    for n_total in (1 << 16, 1 << 20):
        reps = n_total // D
        dev = torch.from_numpy(host.copy()).cuda().repeat(reps, 1).contiguous()
        # tamper 1 lane in 1024: flip a bit of s (-> 7), or zero A' (-> 6)
        idx = torch.arange(513, n_total, 1024, device="cuda")
        dev[idx[0::2], 32] ^= 1
        dev[idx[1::2], 64:96] = 0
        status = torch.full((n_total,), 99, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()          # the engine uses its own streams: torch's writes must have landed
        eng.verify_spend_dev(sk, n_total, dev.data_ptr(), status.data_ptr())
        torch.cuda.synchronize()
        exp = torch.zeros(n_total, dtype=torch.uint8, device="cuda")
        exp[idx[0::2]] = 7
        exp[idx[1::2]] = 6
        assert torch.equal(status, exp), n_total
        del dev


def test_simd_host_hash_on_this_cpu_matches_scalar_and_device(engine_factory, bench_params):
    """Host-transcript mode on the GPU box's CPU (AVX-512 / AVX2 clone of csrc/host_hash.cpp) must give the same
    statuses and refunds as the device-transcript mode on a batch large enough to use the 16-lane groups."""
    from act_amd import capi
    eng = engine_factory(bench_params, 128, max_batch=64)
    sk = eng.private_key_random(shake("sk-simd", 64))
    n = 83                                    # 5 SIMD groups + 3 scalar leftovers, two chunks
    pre = eng.pre_issuance_random(shake("pre-simd", 128 * n))
    req = eng.request(pre, shake("rq-simd", 128 * n))
    st, resp = eng.issue(sk, req, scb(900) * n, shake("ir-simd", 128 * n))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, b"".join(scb(i) for i in range(n)), shake("pr-simd", eng.prove_rng_bytes * n))
    bad = bytearray(proofs); bad[eng.proof_bytes * 40 + 33] ^= 1
    out = {}
    for mode in (capi.TRANSCRIPT_HOST, capi.TRANSCRIPT_DEVICE):
        eng.set_transcript_mode(mode)
        out[mode] = eng.refund(sk, bytes(bad), shake("rr-simd", 128 * n))
    assert out[capi.TRANSCRIPT_HOST] == out[capi.TRANSCRIPT_DEVICE]
    assert list(out[0][0]) == [7 if i == 40 else 0 for i in range(n)]


def test_rejected_private_key_leaves_the_cached_key_untouched(engine_factory, bench_params):
    """good key -> key whose w is not a canonical encoding (ACT_ERR_PARAMS) -> good key again: the second good call must
    verify with the good x (a half-updated key cache would report status 7 for a valid proof)."""
    from act_amd import capi
    eng = engine_factory(bench_params, 8, max_batch=4)
    sk = eng.private_key_random(shake("sk-cache", 64))
    pre = eng.pre_issuance_random(shake("pre-cache", 128)); req = eng.request(pre, shake("rq-cache", 128))
    st, resp = eng.issue(sk, req, scb(9), shake("ir-cache", 128))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proof, prer = eng.prove_spend(tok, scb(4), shake("pr-cache", eng.prove_rng_bytes))
    assert eng.verify_spend(sk, proof) == b"\x00"
    bad = bytearray(eng.private_key_random(shake("sk-cache-other", 64))); bad[32] |= 1        # w: negative s -> undecodable
    with pytest.raises(capi.ActError) as e:
        eng.verify_spend(bytes(bad), proof)
    assert "ACT_ERR_PARAMS" in str(e.value)
    assert eng.verify_spend(sk, proof) == b"\x00"
    st, rf = eng.refund(sk, proof, shake("rr-cache", 128))
    assert st == b"\x00" and eng.refund_to_credit_token(prer, proof, rf, sk[32:])[0] == b"\x00"


@pytest.mark.parametrize("L", [128, 8])
def test_seeded_prover_equals_the_prover_on_the_expanded_bytes(engine_factory, oracle, bench_params, L):
    """act_prove_spend_seeded_batch: lane i draws from the BLAKE3 XOF of seed | u64_le(first_lane + i), expanded on the device.  Must equal
    the ordinary prover (and the oracle's) fed the same bytes expanded by the oracle's BLAKE3 (whose extended output is pinned by the
    upstream vectors, tests/golden/blake3_llvm.json), across chunk boundaries and over a node handle whatever the number of shards."""
    from act_amd import capi
    import pymodel as m
    octx = oracle.ctx(bench_params, L)
    eng = engine_factory(bench_params, L, max_batch=5, transcript=MODES[0])
    N = 12
    sk = octx.private_key_random(shake("sd-pk", 64))
    pre = eng.pre_issuance_random(shake("sd-pre-%d" % L, 128 * N)); req = eng.request(pre, shake("sd-rq", 128 * N))
    st, resp = eng.issue(sk, req, b"".join(scb(50 + i) for i in range(N)), shake("sd-ir", 128 * N))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    assert st == bytes(N)
    s_b = b"".join(scb(i) for i in range(N))
    seed, first = shake("sd-seed", 32), (1 << 40) + 7
    rb = eng.prove_rng_bytes
    rng = b"".join(oracle.blake3(seed + (first + i).to_bytes(8, "little"), rb) for i in range(N))
    assert rng[:200] == m.blake3(seed + first.to_bytes(8, "little"), 200)                     # the Python model's XOF agrees on the first blocks
    got = eng.prove_spend_seeded(tok, s_b, seed, first)
    assert got[0] == bytes(N)
    assert got == eng.prove_spend(tok, s_b, rng)
    assert got[1:] == octx.prove_spend_batch(tok, s_b, rng, 4)
    assert eng.prove_spend_seeded(tok, s_b, seed, first + 1)[1] != got[1]                     # another lane number, another generator
    for devices in ((0,), (0, 0, 0)):
        node = capi.Node(bench_params, L, devices=devices, max_batch=5, transcript=MODES[1])
        try:
            assert node.prove_spend_seeded(tok, s_b, seed, first) == got
        finally:
            node.close()
    assert eng.secret_residue() == 0
