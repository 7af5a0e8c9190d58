#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/ (run in the authoring container only).

Sources of truth, none of which is the code under test:
  * blake3_llvm.json      BLAKE3 outputs of the upstream C implementation bundled in LLVM
                          (/opt/rocm/lib/llvm/lib/libclang-cpp.so, llvm_blake3_hasher_*), i.e. the
                          same algorithm the reference's `blake3 1.8.2` dependency implements.
  * ed25519_openssl.json  Ed25519 public keys derived by OpenSSL 3 (libcrypto) from fixed seeds:
                          pins GF(2^255-19) + the Edwards group law + the base point.
  * rfc9496.json          the RFC 9496 ristretto255 vectors recorded in SURVEY.md Appendix A.
  * primitives.json       scalar / ristretto255 known answers computed by oracle/pymodel.py
                          (Python big integers; shares no code with the C oracle or the kernels).
  * lifecycle_L*.json     full request -> issue -> token -> prove_spend -> refund -> token runs of
                          oracle/pymodel.py with every input, rng seed, output record, status and
                          the SHA-256 of every "spend" transcript pre-image.
The reference crate itself cannot be run here (Rust, un-vendored deps) and holds no golden
vectors, so these files ARE the parity anchor; see DESIGN.md.

Deterministic rng streams are SHAKE-256(label) so the fixtures only store labels.
"""
import ctypes
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pymodel as m  # noqa: E402


def shake(label: str, n: int) -> bytes:
    return hashlib.shake_256(label.encode()).digest(n)


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", name)


def gen_blake3():
    lib = ctypes.CDLL("/opt/rocm/lib/llvm/lib/libclang-cpp.so")

    def llvm_b3(data, n):
        st = ctypes.create_string_buffer(4096)
        lib.llvm_blake3_hasher_init(st)
        lib.llvm_blake3_hasher_update(st, data, ctypes.c_size_t(len(data)))
        out = ctypes.create_string_buffer(n)
        lib.llvm_blake3_hasher_finalize(st, out, ctypes.c_size_t(n))
        return out.raw

    lens = [0, 1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 184, 185, 186, 266, 425, 466, 1023, 1024, 1025, 2047, 2048,
            2049, 3072, 3073, 4096, 4097, 5120, 5121, 6144, 7168, 8192, 8193, 8104, 15784, 16384, 16385, 31744, 65536, 102400]
    dump("blake3_llvm.json", {"input": "byte i = i mod 251", "xof_len": 131,
                              "vectors": [{"len": n, "xof": llvm_b3(bytes(i % 251 for i in range(n)), 131).hex()} for n in lens]})


def gen_ed25519():
    crypto = ctypes.CDLL("libcrypto.so.3")
    crypto.EVP_PKEY_new_raw_private_key.restype = ctypes.c_void_p
    crypto.EVP_PKEY_new_raw_private_key.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    crypto.EVP_PKEY_get_raw_public_key.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
    vec = []
    for i in range(16):
        seed = hashlib.sha256(b"act-golden-ed25519-%d" % i).digest()
        pk = crypto.EVP_PKEY_new_raw_private_key(1087, None, seed, 32)
        out = ctypes.create_string_buffer(32)
        n = ctypes.c_size_t(32)
        assert crypto.EVP_PKEY_get_raw_public_key(pk, out, ctypes.byref(n)) == 1
        h = bytearray(hashlib.sha512(seed).digest()[:32])
        h[0] &= 248; h[31] &= 127; h[31] |= 64
        vec.append({"seed": seed.hex(), "clamped_scalar": bytes(h).hex(), "public_key": out.raw.hex()})
    dump("ed25519_openssl.json", {"vectors": vec})


def gen_rfc9496():
    dump("rfc9496.json", {
        "generator_multiples": ["00" * 32,
                                "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
                                "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
                                "94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259"],
        "one_way_map": [{"sha512_of": "Ristretto is traditionally a short shot of espresso coffee",
                         "encoding": "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46"}]})


def gen_primitives():
    out = {"generator_multiples": [m.ristretto_encode(m.pt_mul(m.BASEPOINT, k)).hex() for k in range(16)]}
    wide = [b"\x00" * 64, b"\xff" * 64, m.ELL.to_bytes(64, "little"), (m.ELL - 1).to_bytes(64, "little"),
            (2**252).to_bytes(64, "little"), (2**512 - 1 - 2**255).to_bytes(64, "little")] + [shake("wide%d" % i, 64) for i in range(26)]
    out["sc_from_wide"] = [{"in": w.hex(), "out": m.sc_bytes(m.sc_from_wide(w)).hex()} for w in wide]
    ma = []
    for i in range(16):
        a, b, c = (m.sc_from_wide(shake("ma%s%d" % (t, i), 64)) for t in "abc")
        ma.append({"a": m.sc_bytes(a).hex(), "b": m.sc_bytes(b).hex(), "c": m.sc_bytes(c).hex(), "muladd": m.sc_bytes(a * b + c).hex(),
                   "inv_a": m.sc_bytes(m.sc_inv(a)).hex()})
    out["sc_muladd_invert"] = ma
    fu = []
    for i in range(24):
        u = shake("uniform%d" % i, 64)
        if i == 0: u = b"\x00" * 64
        if i == 1: u = b"\xff" * 64
        p = m.ristretto_from_uniform_bytes(u)
        s = m.sc_from_wide(shake("fus%d" % i, 64))
        fu.append({"uniform": u.hex(), "encoding": m.ristretto_encode(p).hex(), "scalar": m.sc_bytes(s).hex(),
                   "mul": m.ristretto_encode(m.pt_mul(p, s)).hex(), "double": m.ristretto_encode(m.pt_double(p)).hex(),
                   "plus_gen_mul": m.ristretto_encode(m.pt_add(p, m.pt_mul(m.BASEPOINT, s))).hex()})
    out["from_uniform_bytes"] = fu
    dec = []
    for i in range(96):
        b = bytearray(shake("decode%d" % i, 32)); b[31] &= 0x7F
        dec.append({"bytes": bytes(b).hex(), "valid": m.ristretto_decode(bytes(b)) is not None})
    # special encodings: non-canonical field element, negative s, s = p, the identity
    for b in [(m.P).to_bytes(32, "little"), (m.P + 2).to_bytes(32, "little"), (1).to_bytes(32, "little"), bytes(32), (2**255 - 1).to_bytes(32, "little"),
              bytes([0] * 31 + [0x80])]:
        dec.append({"bytes": b.hex(), "valid": m.ristretto_decode(b) is not None if int.from_bytes(b, "little") < 2**255 else False})
    out["decode_validity"] = dec
    out["params"] = []
    for args in [("bench-org", "bench-service", "bench-env", "2024-01-01"), ("example-corp", "payment-api", "production", "2024-01-15"),
                 ("test-org", "test-service", "test", "2024-01-01"), ("", "", "", "")]:
        out["params"].append({"args": list(args), "h": m.Params.new(*args).encoded().hex()})
    dump("primitives.json", out)


def gen_lifecycle(L, cases, name):
    params_args = ("bench-org", "bench-service", "bench-env", "2024-01-01")
    params = m.Params.new(*params_args)
    sk = m.PrivateKey.random(m.ByteRng(shake("golden-sk", 64)))
    sk2 = m.PrivateKey.random(m.ByteRng(shake("golden-sk-other", 64)))
    out = {"L": L, "params_args": list(params_args), "params": params.encoded().hex(), "sk": sk.record().hex(), "sk_other": sk2.record().hex(),
           "rng": "SHAKE-256(label) truncated to the length each call consumes", "cases": []}
    for idx, (c, s, tamper) in enumerate(cases):
        tag = "L%d-case%d" % (L, idx)
        pre = m.PreIssuance.random(m.ByteRng(shake(tag + "-pre", 128)))
        req = m.request(pre, params, m.ByteRng(shake(tag + "-request", 128)))
        resp = m.issue(sk, params, req, c, m.ByteRng(shake(tag + "-issue", 128)))
        tok = m.issuance_to_credit_token(pre, params, sk.w, req, resp)
        proof, prer = m.prove_spend(tok, params, s, m.ByteRng(shake(tag + "-prove", 64 * (4 * L + 12))), L)
        rec = bytearray(proof.record())
        if tamper == "s":
            rec[32] ^= 1
        elif tamper == "gamma":
            rec[32 * (4 + L) + 7] ^= 0x10
        elif tamper == "identity":
            rec[64:96] = bytes(32)
        elif tamper == "z":
            rec[32 * (12 + 2 * L + 2 * 5) + 3] ^= 4
        rec = bytes(rec)
        case = {"c": str(c), "s": str(s), "tamper": tamper, "pre": pre.record().hex(), "request": req.record().hex(),
                "response": resp.record().hex(), "token": tok.record().hex(), "proof": rec.hex(), "prerefund": prer.record().hex()}
        pr = m.parse_spend_proof(rec, L)
        try:
            if m.pt_eq(pr.a_prime, m.IDENTITY):
                raise m.ActError(m.ERR_IDENTITY_POINT)
            gamma, kprime = m.spend_verify_challenge(sk.x, params, pr)
            case["kprime"] = m.ristretto_encode(kprime).hex()
            case["challenge"] = m.sc_bytes(gamma).hex()
            rf = m.refund(sk, params, pr, m.ByteRng(shake(tag + "-refund", 128)))
            tok2 = m.refund_to_credit_token(prer, params, pr, rf, sk.w)
            case.update(status=0, refund=rf.record().hex(), token2=tok2.record().hex())
        except m.ActError as e:
            case.update(status=e.code, refund=(b"\0" * 128).hex())
        # the same proof checked by a different issuer (prop_multiple_issuers_independence, src/tests.rs:1997)
        try:
            m.refund(sk2, params, pr, m.ByteRng(shake(tag + "-refund", 128)))
            case["status_other_issuer"] = 0
        except m.ActError as e:
            case["status_other_issuer"] = e.code
        out["cases"].append(case)
        print(name, idx, "status", case["status"])
    dump(name, out)


if __name__ == "__main__":
    gen_blake3()
    gen_ed25519()
    gen_rfc9496()
    gen_primitives()
    # (c, s, tamper): honest spends incl. s = 0, s = c, c = 2^L - 1 (src/tests.rs:209-257, 377-426, 1007-1059), overspend (:339-375),
    # tampered s / gamma / z (:603-639, :1681), A' = identity (:850-873)
    gen_lifecycle(128, [(500, 123, None), (1000, 0, None), (1000, 1000, None), (2**128 - 1, 2**127, None), (20, 21, None),
                        (500, 7, "s"), (500, 7, "gamma"), (500, 7, "identity"), (0, 0, None), (77, 9, "z")], "lifecycle_L128.json")
    gen_lifecycle(64, [(2**64 - 1, 12345, None), (10, 11, None), (5, 5, None)], "lifecycle_L64.json")
