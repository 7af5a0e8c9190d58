#!/usr/bin/env python3
"""Generates tests/golden/sodium_*.json with oracle/sodium_model.py (run in the authoring container only).

Every group / scalar result in these files was computed by libsodium 1.0.18's ristretto255 API and every hash by the
BLAKE3 C implementation bundled in LLVM -- arithmetic that neither the Python model, the C oracle nor the HIP kernels
share a line with.  The files are the parity anchor for all three (tests/test_sodium_pins.py, tests/test_gpu_sodium.py);
libsodium itself never travels to the GPU box.

  sodium_primitives.json     >= 1000 known answers: variable-base and base-point multiplications, one-way map
                             (from_uniform_bytes), wide and 32-byte scalar reduction, inversion, scalar ring ops,
                             point add/sub, and >= 200 encodings with libsodium's accept/reject verdict covering every
                             rejection class of CompressedRistretto::decompress (/root/reference/src/cbor.rs:62-77)
  sodium_lifecycle_L128.json full request -> issue -> token -> prove_spend -> refund -> token runs (s = 0, s = c,
  sodium_lifecycle_L64.json  c = 2^L - 1, overspend, tampered fields, A' = identity, non-canonical scalar fields)

rng streams are SHAKE-256(label), so the fixtures store labels only.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sodium_model as sm  # noqa: E402

P = 2**255 - 19
ELL = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def shake(label: str, n: int) -> bytes:
    return hashlib.shake_256(label.encode()).digest(n)


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", name)


def le(v: int, n: int = 32) -> bytes:
    return v.to_bytes(n, "little")


def decode_class(b: bytes) -> str:
    """Label only (plain integer arithmetic per RFC 9496 4.3.1): which check of decompress() rejects this string.
    The accept/reject verdict stored in the fixture is libsodium's; main() asserts that the two agree."""
    v = int.from_bytes(b, "little")
    if v >= P:
        return "non_canonical"
    if v & 1:
        return "negative_s"
    ss = v * v % P
    u1, u2 = (1 - ss) % P, (1 + ss) % P
    u2s = u2 * u2 % P
    vv = (-(D * u1 * u1) - u2s) % P
    w = vv * u2s % P
    # SQRT_RATIO_M1(1, w), RFC 9496 4.2
    r = pow(w, 3, P) * pow(pow(w, 7, P), (P - 5) // 8, P) % P
    check = w * r * r % P
    correct, flipped, flipped_i = check == 1, check == P - 1, check == (P - SQRT_M1) % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    was_square = correct or flipped
    if r & 1:
        r = P - r
    dx = r * u2 % P
    dy = r * dx % P * vv % P
    x = 2 * v * dx % P
    if x & 1:
        x = P - x
    y = u1 * dy % P
    t = x * y % P
    if not was_square:
        return "non_square"
    if t & 1:
        return "negative_t"
    if y == 0:
        return "zero_y"
    return "valid"


def gen_primitives():
    out = {"source": "libsodium %s crypto_core_ristretto255_* / crypto_scalarmult_ristretto255*" % sm.sodium_version()}
    special = [0, 1, 2, 3, 8, 16, ELL - 1, ELL - 2, (ELL - 1) // 2, (ELL + 1) // 2, 2**252, 2**252 - 1, 2**128, 2**128 - 1,
               int("10" * 126, 2), int("01" * 126, 2), 0x7777777777777777777777777777777777777777777777777777777777777777 % ELL,
               0x0888888888888888888888888888888888888888888888888888888888888888, 0x0fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff]
    sm_list, base_list = [], []
    for i in range(320):
        pt = sm.from_uniform(shake("sodium-smP%d" % i, 64))
        s = le(special[i] % ELL) if i < len(special) else sm.sc_reduce_wide(shake("sodium-sms%d" % i, 64))
        sm_list.append({"point": pt.hex(), "scalar": s.hex(), "out": sm.pmul(pt, s).hex()})
        base_list.append({"scalar": s.hex(), "out": sm.bmul(s).hex()})
    out["scalarmult"] = sm_list
    out["scalarmult_base"] = base_list

    fh = []
    for i in range(220):
        u = shake("sodium-fh%d" % i, 64)
        if i == 0: u = bytes(64)
        if i == 1: u = b"\xff" * 64
        if i == 2: u = le(P, 32) + le(P - 1, 32)
        if i == 3: u = le(1, 32) + le(2**255 - 1, 32)
        if i == 4: u = le(2**255, 32) + le(2**255 + 1, 32)        # bit 255 must be ignored
        if i == 5: u = shake("sodium-fh-half", 32) * 2            # both halves equal: MAP(r) + MAP(r)
        fh.append({"uniform": u.hex(), "encoding": sm.from_uniform(u).hex()})
    out["from_uniform_bytes"] = fh

    wide = [bytes(64), b"\xff" * 64, le(ELL, 64), le(ELL - 1, 64), le(ELL + 1, 64), le(2**252, 64), le(2**256 - 1, 64), le(2**256, 64),
            le(2**512 - 1 - 2**255, 64), le(ELL * ELL, 64), le(ELL * ELL - 1, 64), le((2**512 - 1) // ELL * ELL, 64)]
    wide += [shake("sodium-wide%d" % i, 64) for i in range(200 - len(wide))]
    out["sc_reduce_wide"] = [{"in": w.hex(), "out": sm.sc_reduce_wide(w).hex()} for w in wide]
    r32 = [bytes(32), b"\xff" * 32, le(ELL), le(ELL - 1), le(ELL + 1), le(2 * ELL), le(15 * ELL), le(16 * ELL - 1) if 16 * ELL - 1 < 2**256 else le(2**256 - 1), le(2**255), le(2**253)]
    r32 += [shake("sodium-r32-%d" % i, 32) for i in range(100 - len(r32))]
    out["sc_reduce32"] = [{"in": w.hex(), "out": sm.sc_reduce32(w).hex()} for w in r32]

    ring = []
    for i in range(100):
        a, b = (sm.sc_reduce_wide(shake("sodium-ring%s%d" % (t, i), 64)) for t in "ab")
        if i == 0: a = le(1)
        if i == 1: a = le(ELL - 1)
        if i == 2: a, b = le(ELL - 1), le(ELL - 1)
        if i == 3: b = bytes(32)
        ring.append({"a": a.hex(), "b": b.hex(), "add": sm.sc_add(a, b).hex(), "sub": sm.sc_sub(a, b).hex(), "mul": sm.sc_mul(a, b).hex(),
                     "neg_a": sm.sc_neg(a).hex(), "inv_a": sm.sc_invert(a).hex()})
    out["sc_ring"] = ring

    ps = []
    for i in range(60):
        p_, q_ = (sm.from_uniform(shake("sodium-ps%s%d" % (t, i), 64)) for t in "pq")
        if i == 0: q_ = p_
        if i == 1: q_ = sm.IDENTITY
        if i == 2: p_ = sm.IDENTITY
        if i == 3: q_ = sm.psub(sm.IDENTITY, p_)
        ps.append({"p": p_.hex(), "q": q_.hex(), "add": sm.padd(p_, q_).hex(), "sub": sm.psub(p_, q_).hex()})
    out["point_add_sub"] = ps

    enc = []
    cands = []
    for i in range(120):
        cands.append(shake("sodium-dec%d" % i, 32))                                     # arbitrary strings (mostly bit 255 or odd)
    for i in range(120):
        b = bytearray(shake("sodium-dec-even%d" % i, 32)); b[31] &= 0x7F; b[0] &= 0xFE
        cands.append(bytes(b))                                                          # canonical, non-negative: square / t / y classes
    for i in range(40):
        v = sm.from_uniform(shake("sodium-dec-valid%d" % i, 64))
        cands.append(v)
        if i < 10:
            b = bytearray(v); b[31] |= 0x80; cands.append(bytes(b))                    # valid encoding with bit 255 set
        elif i < 20:
            b = bytearray(v); b[0] |= 1; cands.append(bytes(b))                        # ... made negative
        elif i < 26:
            cands.append(le(int.from_bytes(v, "little") + P) if int.from_bytes(v, "little") + P < 2**256 else v)   # s + p: same field element, non-canonical
    cands += [bytes(32), le(1), le(2), le(P - 1), le(P), le(P + 1), le(P + 2), le(2**255 - 1), le(2**255), le(2**256 - 1), le(2**255 - 20),
              le(SQRT_M1), le(P - SQRT_M1), sm.generator(), le((P - 1) // 2), le((P + 1) // 2)]
    classes = {}
    for b in cands:
        cls = decode_class(b)
        ok = sm.is_valid_point(b)
        assert ok == (cls == "valid"), (b.hex(), cls, ok)
        classes[cls] = classes.get(cls, 0) + 1
        enc.append({"bytes": b.hex(), "valid": ok, "class": cls, "libsodium_1_0_18_raw": sm.is_valid_point_libsodium_raw(b)})
    print("decode classes:", classes)
    assert all(classes.get(c, 0) >= 1 for c in ("valid", "non_canonical", "negative_s", "non_square", "negative_t", "zero_y"))
    out["decode_validity"] = enc

    out["params"] = []
    for args in [("bench-org", "bench-service", "bench-env", "2024-01-01"), ("example-corp", "payment-api", "production", "2024-01-15"),
                 ("test-org", "test-service", "test", "2024-01-01"), ("", "", "", ""), ("a:b", "c", "d", "e"), ("a", "b:c", "d", "e")]:
        out["params"].append({"args": list(args), "h": b"".join(sm.params_new(*args)).hex()})
    pr = []
    for i in range(4):
        u = shake("sodium-params-random%d" % i, 192)
        pr.append({"rng": u.hex(), "h": b"".join(sm.params_random(sm.ByteRng(u))).hex()})
    out["params_random"] = pr
    n = sum(len(v) for v in out.values() if isinstance(v, list))
    out["count"] = n
    print("primitive KATs:", n)
    dump("sodium_primitives.json", out)


def field_off(L, name, j=0, b=0):
    base = {"k": 0, "s": 1, "a_prime": 2, "b_bar": 3, "com": 4 + j, "gamma": 4 + L, "e_bar": 5 + L, "r2_bar": 6 + L, "r3_bar": 7 + L,
            "c_bar": 8 + L, "r_bar": 9 + L, "w00": 10 + L, "w01": 11 + L, "gamma0": 12 + L + j, "z": 12 + 2 * L + 2 * j + b,
            "k_bar": 12 + 4 * L, "s_bar": 13 + 4 * L}[name]
    return 32 * base


def apply_tamper(rec: bytes, L: int, tamper):
    rec = bytearray(rec)
    if tamper is None:
        pass
    elif tamper == "s":
        rec[field_off(L, "s")] ^= 1
    elif tamper == "gamma":
        rec[field_off(L, "gamma") + 7] ^= 0x10
    elif tamper == "identity":
        o = field_off(L, "a_prime"); rec[o:o + 32] = bytes(32)
    elif tamper == "z":
        rec[field_off(L, "z", 5 % L, 0) + 3] ^= 4
    elif tamper == "com_swap":
        a, b = field_off(L, "com", 3 % L), field_off(L, "com", 4 % L)
        rec[a:a + 32], rec[b:b + 32] = rec[b:b + 32], rec[a:a + 32]
    elif tamper == "com_undecodable":
        o = field_off(L, "com", L - 1); rec[o] |= 1                # negative s: decompress() fails
    elif tamper == "noncanonical_scalars":                          # r_bar + l and z[1][1] + l: decode_scalar reduces, proof stays valid
        for o in (field_off(L, "r_bar"), field_off(L, "z", 1 % L, 1), field_off(L, "gamma")):
            v = int.from_bytes(rec[o:o + 32], "little") + ELL
            rec[o:o + 32] = le(v)
    elif tamper == "b_bar_identity":
        o = field_off(L, "b_bar"); rec[o:o + 32] = bytes(32)
    elif tamper == "k":
        rec[field_off(L, "k") + 1] ^= 2
    else:
        raise ValueError(tamper)
    return bytes(rec)


def gen_lifecycle(L, cases, name):
    params_args = ("bench-org", "bench-service", "bench-env", "2024-01-01")
    params = sm.params_new(*params_args)
    sk = sm.private_key_random(sm.ByteRng(shake("sodium-sk", 64)))
    sk2 = sm.private_key_random(sm.ByteRng(shake("sodium-sk-other", 64)))
    tag_fmt = "sodium-L%d-case%%d" % L
    out = {"L": L, "params_args": list(params_args), "params": b"".join(params).hex(), "sk": (sk[0] + sk[1]).hex(),
           "sk_other": (sk2[0] + sk2[1]).hex(), "sk_label": "sodium-sk", "tag_fmt": tag_fmt,
           "source": "oracle/sodium_model.py over libsodium %s + LLVM BLAKE3" % sm.sodium_version(),
           "rng": "SHAKE-256(label) truncated to the length each call consumes", "cases": []}
    for idx, (c, s, tamper) in enumerate(cases):
        tag = tag_fmt % idx
        pre = sm.pre_issuance_random(sm.ByteRng(shake(tag + "-pre", 128)))
        req = sm.request(pre, params, sm.ByteRng(shake(tag + "-request", 128)))
        resp = sm.issue(sk, params, req, le(c % ELL), sm.ByteRng(shake(tag + "-issue", 128)))
        tok = sm.issuance_to_credit_token(pre, params, sk[1], req, resp)
        proof, prer, pre_img = sm.prove_spend(tok, params, le(s % ELL), sm.ByteRng(shake(tag + "-prove", 64 * (4 * L + 12))), L)
        rec = apply_tamper(proof, L, tamper)
        case = {"c": str(c), "s": str(s), "tamper": tamper, "pre": (pre[0] + pre[1]).hex(), "request": req.hex(), "response": resp.hex(),
                "token": tok.hex(), "proof": rec.hex(), "prerefund": prer.hex(),
                "prover_transcript_sha256": hashlib.sha256(pre_img).hexdigest()}
        d = sm.decode_spend_proof(rec, L)
        if d is None:
            case.update(status=255, refund=bytes(128).hex(), status_other_issuer=255)
        else:
            try:
                if d["a_prime"] == sm.IDENTITY:
                    raise sm.ActError(sm.E_IDENTITY_POINT)
                gamma, kprime, vpre = sm.spend_challenge(sk[0], params, d, L)
                case["kprime"] = kprime.hex()
                case["challenge"] = gamma.hex()
                case["verifier_transcript_sha256"] = hashlib.sha256(vpre).hexdigest()
                rf = sm.refund(sk, params, d, sm.ByteRng(shake(tag + "-refund", 128)), L)
                tok2 = sm.refund_to_credit_token(prer, params, d, rf, sk[1], L)
                case.update(status=0, refund=rf.hex(), token2=tok2.hex())
            except sm.ActError as e:
                case.update(status=e.code, refund=bytes(128).hex())
            try:
                sm.refund(sk2, params, d, sm.ByteRng(shake(tag + "-refund", 128)), L)
                case["status_other_issuer"] = 0
            except sm.ActError as e:
                case["status_other_issuer"] = e.code
        out["cases"].append(case)
        print(name, idx, (c, s, tamper), "status", case["status"])
    dump(name, out)


if __name__ == "__main__":
    gen_primitives()
    M = 2**128 - 1
    gen_lifecycle(128, [(500, 123, None), (1000, 0, None), (1000, 1000, None), (M, 2**127, None), (M, M, None), (M, 0, None),
                        (20, 21, None), (0, 1, None), (0, 0, None), (500, 7, "s"), (500, 7, "gamma"), (500, 7, "identity"), (77, 9, "z"),
                        (12345, 12344, "com_swap"), (999, 1, "com_undecodable"), (31337, 337, "noncanonical_scalars"),
                        (64, 32, "b_bar_identity"), (64, 32, "k")], "sodium_lifecycle_L128.json")
    gen_lifecycle(64, [(2**64 - 1, 12345, None), (10, 11, None), (5, 5, None), (2**64 - 1, 0, None), (9, 3, "noncanonical_scalars")],
                  "sodium_lifecycle_L64.json")
