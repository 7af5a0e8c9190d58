"""The prover kernels' own lane bodies (csrc/prove_lanes.h: what k_prove_head / bits / enc / tail / resp execute per lane),
compiled for the host by tests/hostcheck and run lane by lane on the CPU, against the libsodium-made fixtures and the C oracle:
SpendProof and PreRefund bytes.  A CPU unit test of kernel code (the product has no CPU path); the same bodies run on the GPU in
tests/test_gpu_*.py.  The operation counts it returns are what bench.py's `roofline_prover` is computed from."""
import ctypes as C

import pytest

from conftest import load_golden, shake, scb

hx = bytes.fromhex


def host_prove(hc, h, L, tok, s, rng):
    pb = 32 * (14 + 4 * L)
    n = len(tok) // 160
    proofs = C.create_string_buffer(pb * n); prer = C.create_string_buffer(96 * n); st = C.create_string_buffer(n)
    counts = (C.c_uint64 * 31)()
    assert hc.hc_prove_spend(h, L, n, tok, s, rng, proofs, prer, st, counts) == 1
    return st.raw, proofs.raw, prer.raw, list(counts)


@pytest.mark.parametrize("L", [128, 8, 3, 100])
def test_prover_lane_bodies_reproduce_the_oracle(hostcheck, oracle, bench_params, L):
    octx = oracle.ctx(bench_params, L)
    sk = octx.private_key_random(shake("pl-sk-%d" % L, 64))
    n = 3
    toks, ss = [], []
    for i in range(n):
        pre = octx.pre_issuance_random(shake("pl-pre-%d-%d" % (L, i), 128))
        req = octx.request(pre, shake("pl-rq-%d-%d" % (L, i), 128))
        c = (1 << (L - 1)) - 1 + i if L > 3 else 5 + i
        st, resp = octx.issue(sk, req, scb(c), shake("pl-ir-%d-%d" % (L, i), 128))
        st, tok = octx.issuance_to_credit_token(pre, sk[32:], req, resp)
        assert st == 0
        toks.append(tok); ss.append(scb((0, 1, c + 1)[i]))               # lane 2 overspends: produced all the same (Appendix E), rejected later
    rng = shake("pl-pr-%d" % L, octx.prove_rng_bytes * n)
    st, proofs, prer, counts = host_prove(hostcheck, bench_params, L, b"".join(toks), b"".join(ss), rng)
    po, pro = octx.prove_spend_batch(b"".join(toks), b"".join(ss), rng, 1)
    assert st == bytes(n) and proofs == po and prer == pro
    # counts: k_prove_bits multiplies four fixed bases per (proof, bit) lane; the encodes take 3 points per lane
    assert sum(counts[6 * 1 + 2:6 * 1 + 6]) == 4 * L * n
    assert counts[6 * 1 + 2 + 3] == 3 * L * n and counts[6 * 1 + 2 + 1] == L * n            # three on h3, one on h1


@pytest.mark.parametrize("name", ["sodium_lifecycle_L128.json", "sodium_lifecycle_L64.json"])
def test_prover_lane_bodies_reproduce_the_libsodium_fixtures(hostcheck, name):
    """Third-party arithmetic (libsodium + LLVM BLAKE3, tests/golden/make_sodium_golden.py): the untampered cases' proofs and
    PreRefunds from their tokens and the fixture's rng labels."""
    g = load_golden(name)
    L = g["L"]
    ELL = 2**252 + 27742317777372353535851937790883648493
    pb = 32 * (14 + 4 * L)
    idxs = [i for i, c in enumerate(g["cases"]) if c["tamper"] is None]
    assert idxs
    tok = b"".join(hx(g["cases"][i]["token"]) for i in idxs)
    s = b"".join((int(g["cases"][i]["s"]) % ELL).to_bytes(32, "little") for i in idxs)
    rng = b"".join(shake((g["tag_fmt"] % i) + "-prove", 64 * (4 * L + 12)) for i in idxs)
    st, proofs, prer, _ = host_prove(hostcheck, hx(g["params"]), L, tok, s, rng)
    assert st == bytes(len(idxs))
    for k, i in enumerate(idxs):
        assert proofs[pb * k:pb * k + pb].hex() == g["cases"][i]["proof"], i
        assert prer[96 * k:96 * k + 96].hex() == g["cases"][i]["prerefund"], i


def test_undecodable_token_gives_a_zero_record(hostcheck, bench_params):
    L = 8
    tok = b"\xff" * 32 + bytes(128)
    rng = shake("pl-bad", 64 * (4 * L + 12))
    st, proofs, prer, _ = host_prove(hostcheck, bench_params, L, tok, scb(0), rng)
    assert st == b"\xff" and not any(proofs) and not any(prer)
