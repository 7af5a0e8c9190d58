"""The device arithmetic headers (csrc/*.h), compiled for the host by tests/hostcheck/hostcheck.cpp, against
the Python model: field / scalar / ristretto255 / shared-doubling chain / fixed-base windows / BLAKE3, plus the
limb-size discipline the lazily-reduced field code relies on.  CPU-only unit tests of kernel math; the product
has no CPU path."""
import ctypes as C

import pymodel as m
from conftest import load_golden, shake

P = m.P


def call(hc, fn, *ins, nout=1, outlen=32):
    outs = [C.create_string_buffer(outlen) for _ in range(nout)]
    r = getattr(hc, fn)(*ins, *outs)
    return (r, *[o.raw for o in outs])


def le(x):
    return x.to_bytes(32, "little")


def test_field(hostcheck):
    hc = hostcheck
    edge = [0, 1, 2, 19, P - 1, P - 2, P - 19, 2**255 - 20, 2**254, (1 << 26) - 1, (1 << 51) - 1, 2**128, 2**255 - 1]
    vals = edge + [int.from_bytes(shake("f%d" % i, 32), "little") % 2**255 for i in range(200)]
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        fi = lambda x: int.from_bytes(x, "little")
        assert fi(call(hc, "hc_fe_mul", le(a), le(b))[1]) == a * b % P
        assert fi(call(hc, "hc_fe_sq", le(a))[1]) == a * a % P
        assert fi(call(hc, "hc_fe_add", le(a), le(b))[1]) == (a + b) % P
        assert fi(call(hc, "hc_fe_sub", le(a), le(b))[1]) == (a - b) % P
        x, y = a % P, b % P
        exp = ((2 * x - y) * (x - y)) % P
        exp = ((2 * exp - (x + y)) * (x - y) ** 2) % P
        assert fi(call(hc, "hc_fe_stress", le(a), le(b))[1]) == exp
        if i < 40 and a % P:
            assert fi(call(hc, "hc_fe_invert", le(a))[1]) == pow(a, P - 2, P)
            ok, r = call(hc, "hc_fe_invsqrt", le(a))
            ok2, r2 = m.sqrt_ratio_m1(1, a)
            assert bool(ok) == ok2 and fi(r) == r2


POS = [-(-85 * i // 3) for i in range(10)]
WID = [POS[i + 1] - POS[i] for i in range(9)]


def test_field_at_the_column_budget(hostcheck):
    """fe_mul / fe_sq on operands whose EVERY limb sits at the top of its size class (phi * gamma up to the 12.5 the header allows,
    squares up to phi = 3.5): a 64-bit column that overflowed would give a wrong product.  Also: outputs are tight, and fe_carry
    brings any element below phi = 7.5 back to tight."""
    import random
    hc = hostcheck
    r = random.Random(9)
    val = lambda limbs: sum(l << POS[i] for i, l in enumerate(limbs))
    arr = lambda limbs: (C.c_uint32 * 9)(*limbs)
    fi = lambda x: int.from_bytes(x, "little")
    tight_max = [(1 << w) + ((1 << 12) if i == 1 else 0) for i, w in enumerate(WID)]

    def operand(phi, kind):
        top = [min(int(phi * (1 << w)), (1 << 32) - 1) for w in WID]
        if kind == "max":
            return top
        if kind == "alt":
            return [t if i % 2 else t // 3 for i, t in enumerate(top)]
        return [r.randrange(t // 2, t + 1) for t in top]

    pairs = [(1, 1), (2, 2), (3, 3), (4, 3), (3, 4), (6, 2), (2, 6), (7.4, 1.68), (1.68, 7.4), (5, 2.5), (3.5, 3.5)]   # every operand below phi = 7.5
    for pa, pb in pairs:
        for kind in ("max", "alt", "rnd", "rnd"):
            a, b = operand(pa, kind), operand(pb, kind)
            out = C.create_string_buffer(32)
            hc.hc_fe_mul_limbs(arr(a), arr(b), out)
            assert fi(out.raw) == val(a) * val(b) % P, (pa, pb, kind)
            lo = (C.c_uint32 * 9)()
            hc.hc_fe_mul_out_limbs(arr(a), arr(b), lo)
            assert all(l <= t for l, t in zip(lo, tight_max)), (pa, pb, kind, list(lo))
    for phi in (1, 2, 3, 3.5):
        for kind in ("max", "alt", "rnd", "rnd"):
            a = operand(phi, kind)
            out = C.create_string_buffer(32)
            hc.hc_fe_sq_limbs(arr(a), out)
            assert fi(out.raw) == val(a) ** 2 % P, (phi, kind)
    for phi in (1, 2, 5, 6, 7.4):
        for kind in ("max", "alt", "rnd"):
            a = operand(phi, kind)
            lo, out = (C.c_uint32 * 9)(), C.create_string_buffer(32)
            hc.hc_fe_carry_limbs(arr(a), lo, out)
            assert fi(out.raw) == val(a) % P and all(l <= t for l, t in zip(lo, tight_max)), (phi, kind, list(lo))


def test_scalars(hostcheck):
    hc = hostcheck
    for v in load_golden("primitives.json")["sc_from_wide"]:
        assert call(hc, "hc_sc_reduce_wide", bytes.fromhex(v["in"]))[1].hex() == v["out"]
    for i in range(100):
        a, b, c = (shake("%s%d" % (t, i), 32) for t in "abc")
        if i == 0:
            a = b"\xff" * 32
        ai, bi, ci = (int.from_bytes(x, "little") for x in (a, b, c))
        assert call(hc, "hc_sc_from_bytes", a)[1] == m.sc_bytes(ai)
        assert call(hc, "hc_sc_muladd", a, b, c)[1] == m.sc_bytes(ai * bi + ci)
        assert call(hc, "hc_sc_sub", a, b)[1] == m.sc_bytes(ai - bi)
        assert call(hc, "hc_sc_neg", a)[1] == m.sc_bytes(-ai)
        if i < 5:
            assert call(hc, "hc_sc_invert", a)[1] == m.sc_bytes(m.sc_inv(ai % m.ELL))


def test_blake3_against_upstream_vectors(hostcheck):
    for v in load_golden("blake3_llvm.json")["vectors"]:
        if v["len"] > 200000:
            continue
        data = bytes(i % 251 for i in range(v["len"]))
        out = C.create_string_buffer(64)
        hostcheck.hc_blake3_xof64(data, v["len"], out)
        assert out.raw.hex() == v["xof"][:128], v["len"]
        # the chunk-parallel form (k_hash_xof_par: chaining values of the chunks first, then the fold)
        hostcheck.hc_blake3_xof64_par(data, v["len"], out)
        assert out.raw.hex() == v["xof"][:128], ("par", v["len"])


def test_group_and_msm_shapes(hostcheck):
    hc = hostcheck
    g = load_golden("primitives.json")
    for v in g["decode_validity"]:
        ok, _ = call(hc, "hc_decode_encode", bytes.fromhex(v["bytes"]))
        assert bool(ok) == v["valid"], v["bytes"]
    for i, v in enumerate(g["from_uniform_bytes"]):
        e = bytes.fromhex(v["encoding"])
        assert call(hc, "hc_from_uniform", bytes.fromhex(v["uniform"]))[1] == e
        ok, r = call(hc, "hc_decode_encode", e)
        assert ok and r == e
        s = bytes.fromhex(v["scalar"])
        ok, o0 = call(hc, "hc_chain1", e, s)
        assert ok and o0.hex() == v["mul"]
        s1 = m.sc_from_wide(shake("hc-s1-%d" % i, 64))
        pm = m.ristretto_decode(e)
        for fn in ("hc_chain2", "hc_chain2u", "hc_chain_bu", "hc_chain_bu_pre", "hc_chain_b2"):
            ok, o0, o1 = call(hc, fn, e, s, m.sc_bytes(s1), nout=2)
            assert o0.hex() == v["mul"] and o1 == m.ristretto_encode(m.pt_mul(pm, s1)), fn
        if i < 4:
            ok, o = call(hc, "hc_fixed_base", e, s)
            assert o.hex() == v["mul"]
    # digit-recoding corner cases of the radix-4 chain
    e = bytes.fromhex(g["generator_multiples"][1])
    for s in (0, 1, 2, 3, 4, m.ELL - 1, (1 << 252) + 5, (1 << 252) - 1, int("3" * 60, 16) % m.ELL, int("2" * 63, 16) % m.ELL):
        for fn in ("hc_chain2", "hc_chain2u", "hc_chain_bu", "hc_chain_bu_pre", "hc_chain_b2", "hc_chain_ct2"):      # radix-4 / radix-4, radix-4 / NAF, buckets / NAF (twice), buckets / buckets, the ct build's address-free form
            ok, o0, o1 = call(hc, fn, e, m.sc_bytes(s), m.sc_bytes(m.ELL - 1 - s), nout=2)
            assert o0 == m.ristretto_encode(m.pt_mul(m.BASEPOINT, s)) and o1 == m.ristretto_encode(m.pt_mul(m.BASEPOINT, m.ELL - 1 - s)), (fn, s)
    # width-3 NAF recoding of the wave-uniform scalar (msm.h naf3_next): runs of 1s, alternating digits, carries
    # rippling to the top, and values in [l, 2^253) that only the raw recoder (not the reduced scalar type) can see
    pats = [int(c * 63, 16) for c in "1357bdf"] + [(1 << 253) - 1, (1 << 253) - 3, (1 << 252) + (1 << 251) + 3,
                                                   0b011, 0b101, 0b111, 0b1011, (1 << 200) - 1, ((1 << 253) - 1) // 3, ((1 << 253) - 1) // 7 * 3]
    for su in pats:
        su %= 1 << 253
        for fn in ("hc_chain_bu", "hc_chain_bu_pre"):
            ok, o0, o1 = call(hc, fn, e, m.sc_bytes(9), su.to_bytes(32, "little"), nout=2)
            assert o0 == m.ristretto_encode(m.pt_mul(m.BASEPOINT, 9)) and o1 == m.ristretto_encode(m.pt_mul(m.BASEPOINT, su)), (fn, hex(su))
    for fn in ("hc_chain2", "hc_chain2u", "hc_chain_bu", "hc_chain_bu_pre", "hc_chain_b2", "hc_chain_ct2"):   # identity base
        ok, o0, o1 = call(hc, fn, bytes(32), m.sc_bytes(5), m.sc_bytes(m.ELL - 7), nout=2)
        assert o0 == bytes(32) and o1 == bytes(32), fn


def test_batched_double_and_compress(hostcheck):
    """ge25519.h dc_* / msm.h dc_encode_batch against the model: enc(2Q) for projective Q, batches of every size up to 8,
    with identity-class members (whose zero must not poison the shared inversion) and torsion-shifted representatives."""
    hc = hostcheck
    g = load_golden("primitives.json")
    mults = [bytes.fromhex(x) for x in g["generator_multiples"]]
    pts = [m.ristretto_decode(e) for e in mults]
    ident = bytes(32)
    for count in range(1, 9):
        a = [mults[(3 * i + count) % len(mults)] for i in range(count)]
        b = [mults[(5 * i + 1) % len(mults)] for i in range(count)]
        if count >= 3:
            a[1] = ident; b[1] = ident                                   # Q = identity
        if count >= 5:
            b[3] = m.ristretto_encode(m.pt_neg(m.ristretto_decode(a[3])))   # Q = P - P
        out = (C.c_uint8 * (32 * count))()
        assert hc.hc_dc_encode_batch(b"".join(a), b"".join(b), count, out) == 1
        for i in range(count):
            q = m.pt_add(m.ristretto_decode(a[i]), m.ristretto_decode(b[i]))
            assert bytes(out[32 * i:32 * i + 32]) == m.ristretto_encode(m.pt_double(q)), (count, i)
    # halving: s/2 mod l
    for s_ in (0, 1, 2, 3, m.ELL - 1, m.ELL - 2, (1 << 252) + 1, int("5" * 62, 16) % m.ELL):
        ok, o = call(hc, "hc_sc_half", m.sc_bytes(s_))
        assert int.from_bytes(o, "little") == s_ * pow(2, m.ELL - 2, m.ELL) % m.ELL, s_


def test_dedicated_addition_never_exceptional(hostcheck):
    """msm.h chain_bu_pre adds chain points into buckets / NAF accumulators with the d-free formulas (ge25519.h ge_add_ded),
    which return (0,0,0,0) when the operands differ by an element of E[4].  (i) The formulas: sums, the identity as first
    operand, p + (-p), and the failures p + p, p + (p + order-2 point).  (ii) The arithmetic fact the proof in msm.h rests on:
    no multiple t*l, |t| < 40, is a signed radix-16 string over {-1, 0, 1} or a NAF whose non-zero digits are >= 3 apart.
    (iii) Digit strings that put many chain points into one accumulator, and every single-bucket string."""
    hc = hostcheck
    g = load_golden("primitives.json")
    enc = [e for e in (bytes.fromhex(v["encoding"]) for v in g["from_uniform_bytes"]) if e != bytes(32)][:6]
    pts = [m.ristretto_decode(e) for e in enc]
    for i, (ea, pa) in enumerate(zip(enc, pts)):
        for eb, pb in zip(enc[i + 1:], pts[i + 1:]):
            rc, o = call(hc, "hc_add_ded", ea, eb, C.c_int(0))
            assert rc == 1 and o == m.ristretto_encode(m.pt_add(pa, pb))
            rc, o = call(hc, "hc_add_ded", ea, eb, C.c_int(1))
            assert rc == 1 and o == m.ristretto_encode(m.pt_add(pa, pb))          # + (0,-1): same ristretto element
        rc, o = call(hc, "hc_add_ded", bytes(32), ea, C.c_int(0))
        assert rc == 1 and o == ea                                                 # identity + q
        rc, o = call(hc, "hc_add_ded", ea, m.ristretto_encode(m.pt_neg(pa)), C.c_int(0))
        assert rc == 1 and o == bytes(32)                                          # p + (-p) is not exceptional
        assert call(hc, "hc_add_ded", ea, ea, C.c_int(0))[0] == 2                  # p + p
        assert call(hc, "hc_add_ded", ea, ea, C.c_int(1))[0] == 2                  # p + (p + (0,-1))
    assert call(hc, "hc_add_ded", bytes(32), bytes(32), C.c_int(0))[0] == 2        # identity + identity: why n_small exists

    def signed16(n):
        d = []
        while n:
            r = n % 16
            r -= 16 if r >= 8 else 0
            d.append(r); n = (n - r) // 16
        return d

    def naf(n):
        d = []
        while n:
            r = (2 - n % 4) if n & 1 else 0
            n -= r; d.append(r); n //= 2
        return d
    for t in range(1, 40):
        assert sum(abs(x) > 1 for x in signed16(t * m.ELL)) >= 24
        pos = [i for i, x in enumerate(naf(t * m.ELL)) if x]
        assert sum(b - a == 2 for a, b in zip(pos, pos[1:])) >= 15

    e = enc[0]; pm = pts[0]
    one_bucket = [int(("%x" % v) * 64, 16) % m.ELL for v in range(1, 9)] + [int("0" + "f" * 62 + "1", 16), int("08" * 32, 16), int("80" * 31 + "08", 16) % m.ELL]
    sparse_naf = [sum(1 << i for i in range(0, 253, 3)), sum(3 << i for i in range(0, 250, 3)) % (1 << 253), sum(((-1) ** (i // 3)) * (1 << i) for i in range(0, 252, 3)) % m.ELL, 1 << 252, (1 << 253) - 1]
    for sl in one_bucket + [0, 1, m.ELL - 1]:
        for su in sparse_naf + [0, 1, 3]:
            ok, o0, o1 = call(hc, "hc_chain_bu_pre", e, m.sc_bytes(sl), su.to_bytes(32, "little"), nout=2)
            assert ok and o0 == m.ristretto_encode(m.pt_mul(pm, sl)) and o1 == m.ristretto_encode(m.pt_mul(pm, su)), (hex(sl), hex(su))


def test_dedicated_addition_chain_agrees_with_the_complete_one_on_structured_digit_strings(hostcheck):
    """3 000 digit strings built to be as structured as the recoders allow (few distinct digit values, long runs, sparse
    supports, values next to multiples of l): msm.h chain_bu_pre (d-free additions) against chain_bu (complete additions
    only), both on the host build; a sample against the Python model."""
    import random
    hc = hostcheck
    g = load_golden("primitives.json")
    encs = [e for e in (bytes.fromhex(v["encoding"]) for v in g["from_uniform_bytes"]) if e != bytes(32)][:8]
    r = random.Random(20260101)

    def radix16_string():
        kind = r.randrange(5)
        if kind == 0:                                   # two digit values only
            a, b = r.randrange(16), r.randrange(16)
            return int("".join("%x" % r.choice((a, b)) for _ in range(63)), 16)
        if kind == 1:                                   # sparse support
            return sum(r.randrange(1, 16) << (4 * i) for i in r.sample(range(63), r.randrange(1, 6)))
        if kind == 2:                                   # runs
            d = []
            while len(d) < 63:
                d += ["%x" % r.randrange(16)] * r.randrange(1, 20)
            return int("".join(d[:63]), 16)
        if kind == 3:                                   # next to a multiple of l / a power of 16
            return (r.randrange(1, 8) * m.ELL + r.randrange(-40, 40)) % m.ELL if r.random() < 0.5 else ((1 << (4 * r.randrange(1, 63))) + r.randrange(-3, 4)) % m.ELL
        return r.randrange(m.ELL)

    def naf_string():
        kind = r.randrange(4)
        if kind == 0:                                   # non-zero digits exactly three apart, random signs and magnitudes
            return sum(r.choice((1, 3, -1, -3)) << i for i in range(r.randrange(3), 250, 3)) % (1 << 253)
        if kind == 1:
            return sum(1 << i for i in r.sample(range(253), r.randrange(1, 8)))
        if kind == 2:
            return (r.randrange(1, 3) * m.ELL + r.randrange(-40, 40)) % (1 << 253)
        return r.randrange(1 << 253)

    for it in range(3000):
        e = encs[it % len(encs)]
        sl, su = radix16_string() % m.ELL, naf_string()
        ok, a0, a1 = call(hc, "hc_chain_bu_pre", e, m.sc_bytes(sl), su.to_bytes(32, "little"), nout=2)
        ok2, b0, b1 = call(hc, "hc_chain_bu", e, m.sc_bytes(sl), su.to_bytes(32, "little"), nout=2)
        assert ok and ok2 and a0 == b0 and a1 == b1, (it, hex(sl), hex(su))
        if it % 300 == 0:
            pm = m.ristretto_decode(e)
            assert a0 == m.ristretto_encode(m.pt_mul(pm, sl)) and a1 == m.ristretto_encode(m.pt_mul(pm, su))


def test_limb_bounds_hold(hostcheck):
    """Every operand recorded by the instrumented host build stays inside its class (fe25519.h header comment)."""
    hostcheck.hc_bounds_reset()
    test_field(hostcheck)
    test_group_and_msm_shapes(hostcheck)
    test_batched_double_and_compress(hostcheck)
    test_dedicated_addition_never_exceptional(hostcheck)
    bd = (C.c_uint64 * 6)()
    hostcheck.hc_bounds(bd)
    # units of 2^-16: product budget phi*gamma (mul), phi^2 (sq) <= 12.5 (a column holds 5 * 2^58 * budget < 2^64);
    # subtrahend / (2p resp. 4p) limb-wise <= 1; no element anywhere reaches phi = 7.5 (limbs stay below 2^32 through a carry)
    lim = [12.5, 12.5, 1.0, 1.0, 7.5]
    print('limb bounds (units of 2^-16):', [round(v / 65536, 4) for v in bd])
    for v, l in zip(bd, lim):
        assert 0 < v <= l * 2**16, (list(bd), lim)
    assert bd[5] == 0, "a subtraction met a subtrahend limb above its offset (fe_limb_sub: |c - g| + f would be wrong on the GPU)"


def test_generated_field_sources_are_current():
    """csrc/fe25519_gen.inc and fe25519_consts.inc are what tools/gen_fe_mul.py / gen_fe_consts.py print (no hand edits, no stale
    output), and the constants are the right numbers: each ACT_FE_CONST row, read back through the 9-limb positions, equals the
    value gen_fe_consts.py derives it from."""
    import subprocess, sys, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "anonymous-credit-tokens_amd", "csrc")
    for tool, inc in (("gen_fe_mul.py", "fe25519_gen.inc"), ("gen_fe_consts.py", "fe25519_consts.inc")):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", tool)], capture_output=True, text=True, check=True).stdout
        assert out == open(os.path.join(csrc, inc)).read(), inc
    rows = dict((m.group(1), [int(x.rstrip("u"), 16) for x in m.group(2).split(", ")])
                for m in re.finditer(r"ACT_FE_CONST\((\w+), ([^)]*)\)", open(os.path.join(csrc, "fe25519_consts.inc")).read()))
    val = lambda limbs: sum(l << POS[i] for i, l in enumerate(limbs))
    d = (-121665 * pow(121666, P - 2, P)) % P
    sqrt_m1 = pow(2, (P - 1) // 4, P)
    assert val(rows["fe_d"]) == d and val(rows["fe_d2"]) == 2 * d % P
    assert val(rows["fe_sqrt_m1"]) in (sqrt_m1, P - sqrt_m1) and val(rows["fe_sqrt_m1"]) ** 2 % P == P - 1
    assert all(len(v) == 9 and all(l < (1 << WID[i]) for i, l in enumerate(v)) for v in rows.values())
