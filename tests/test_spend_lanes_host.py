"""The spend-verification kernels' own lane bodies (csrc/spend_lanes.h: what k_spend_prep / bits / enc / tail / finish
execute per lane), compiled for the host by tests/hostcheck and run lane by lane on the CPU, against the libsodium-made
fixtures and the C oracle: statuses, enc(K') and the complete "spend" transcript pre-images.  This is a CPU unit test of
kernel code (the product has no CPU path); the same bodies run on the GPU in tests/test_gpu_*.py."""
import ctypes as C
import hashlib

import pytest

from conftest import load_golden, shake, scb

hx = bytes.fromhex


def host_verify(hc, h, L, sk, proofs):
    """Runs the lane bodies in the order of the pipelined schedule AND in the small-batch schedule's (the roles of k_spend_prep as
    kernels of their own, k_spend_tail in front of k_spend_bits, small_impl.inc spend_small_locked): the two must agree on every byte
    and on the operation counts; returns the former."""
    pb = 32 * (14 + 4 * L)
    n = len(proofs) // pb
    tb = 184 + 40 * (6 + 3 * L)
    outs = []
    for fn in (hc.hc_spend_verify, hc.hc_spend_verify_small):
        tr = C.create_string_buffer(n * tb); st = C.create_string_buffer(n); kp = C.create_string_buffer(32 * n)
        counts = (C.c_uint64 * 25)()
        assert fn(h, L, sk, n, proofs, tr, st, kp, counts) == 1
        outs.append((st.raw, kp.raw, [tr.raw[i * tb:(i + 1) * tb] for i in range(n)], op_counts(list(counts), n)))
    a, b = outs
    assert a[0] == b[0] and a[1] == b[1]
    for i in range(n):
        if a[0][i] != 6:                   # A' = identity: the pipelined prep still writes A1 / A2 of a proof that is rejected anyway
            assert a[2][i] == b[2][i], i
    small_ops = {k: dict(v) for k, v in b[3].items()}
    # the small schedule decodes every Com_j twice (once for the tail, once in the range kernel)
    assert all(small_ops[k]["fe_mul"] == a[3][k]["fe_mul"] for k in ("k_spend_prep", "k_spend_bits", "k_spend_enc"))
    return a


def op_counts(c, n, product_windows=(16, 16, 16, 16)):
    """Per-proof field operations of each kernel restated for the product's fixed-base windows per base g, h1, h2, h3
    (this build: c[24] windows of 6 bits)."""
    out = {}
    for k, name in enumerate(("k_spend_prep", "k_spend_bits", "k_spend_enc", "k_spend_tail")):
        mul, sq = c[6 * k], c[6 * k + 1]
        fb = c[6 * k + 2:6 * k + 6]
        out[name] = {"fe_mul": (mul - sum(fb[b] * (c[24] - product_windows[b]) * 7 for b in range(4))) / n, "fe_sq": sq / n,
                     "fixed_base_mults": sum(fb) / n}
    return out


@pytest.mark.parametrize("name", ["sodium_lifecycle_L128.json", "sodium_lifecycle_L64.json"])
def test_kernel_lane_bodies_reproduce_the_libsodium_fixtures(hostcheck, oracle, name):
    g = load_golden(name)
    L, cases = g["L"], g["cases"]
    sk = hx(g["sk"])
    proofs = b"".join(hx(c["proof"]) for c in cases)
    st, kp, trs, counts = host_verify(hostcheck, hx(g["params"]), L, sk, proofs)
    assert list(st) == [c["status"] for c in cases]
    octx = oracle.ctx(hx(g["params"]), L)
    pb = octx.proof_bytes
    for i, c in enumerate(cases):
        assert kp[32 * i:32 * i + 32].hex() == (c["kprime"] if c["status"] == 0 else "00" * 32)
        if "verifier_transcript_sha256" in c:
            assert hashlib.sha256(trs[i]).hexdigest() == c["verifier_transcript_sha256"], i
            so, kpo, tro = octx.verify_spend(sk, proofs[pb * i:pb * i + pb], True)
            assert trs[i] == tro
    # operation counts per proof are what bench.py's ALU roofline is computed from: sanity-bound them
    mul = sum(v["fe_mul"] for v in counts.values()); sq = sum(v["fe_sq"] for v in counts.values())
    assert 2000 * L < mul < 3000 * L and 1000 * L < sq < 1700 * L, (mul, sq)
    assert counts["k_spend_bits"]["fixed_base_mults"] == 3 * L


def test_ragged_and_small_widths(hostcheck, oracle, bench_params):
    """L = 3 and L = 100 (lanes of several proofs share a 32-point encode batch), a tampered lane, an identity A' and a proof
    with identity commitments."""
    for L in (3, 100):
        octx = oracle.ctx(bench_params, L)
        sk = octx.private_key_random(shake("hl-sk-%d" % L, 64))
        recs = []
        for i in range(4):
            pre = octx.pre_issuance_random(shake("hl-pre-%d-%d" % (L, i), 128))
            req = octx.request(pre, shake("hl-rq-%d-%d" % (L, i), 128))
            st, resp = octx.issue(sk, req, scb(5 + i), shake("hl-ir-%d-%d" % (L, i), 128))
            st, tok = octx.issuance_to_credit_token(pre, sk[32:], req, resp)
            st, proof, _ = octx.prove_spend(tok, scb(i), shake("hl-pr-%d-%d" % (L, i), octx.prove_rng_bytes))
            recs.append(bytearray(proof))
        recs[1][33] ^= 1
        recs[2][64:96] = bytes(32)
        for j in (0, L - 1):                       # Com_j = identity: the one base the d-free additions of msm.h chain_bu_pre cannot take
            recs[3][32 * (4 + j):32 * (5 + j)] = bytes(32)
        proofs = b"".join(bytes(r) for r in recs)
        st, kp, trs, _ = host_verify(hostcheck, bench_params, L, sk, proofs)
        for i in range(4):
            so, kpo, tro = octx.verify_spend(sk, bytes(recs[i]), True)
            assert st[i] == so, (L, i)
            if so in (0, 7):                       # the reference returns before building a transcript when A' is the identity
                assert trs[i] == tro, (L, i)
        assert list(st) == [0, 7, 6, 7]


def test_lane_bodies_stay_inside_the_limb_budget(hostcheck, oracle, bench_params):
    """The verifier's lane bodies, run through the instrumented host build: every product inside the column budget of
    fe25519.h (12.5), every subtrahend below its 2p / 4p offset."""
    import ctypes as C
    hostcheck.hc_bounds_reset()
    test_ragged_and_small_widths(hostcheck, oracle, bench_params)
    bd = (C.c_uint64 * 6)()
    hostcheck.hc_bounds(bd)
    for v, l in zip(bd, [12.5, 12.5, 1.0, 1.0, 7.5]):
        assert 0 < v <= l * 2**16, list(bd)
    assert bd[5] == 0, "a subtrahend limb above its offset"
