"""Batch CBOR codec (/root/reference/src/cbor.rs) against the Python model: deterministic encodings
byte for byte, and the decoder on canonical and deliberately non-canonical / broken messages."""
import pytest

import pymodel as m
from conftest import load_golden, shake, scb


def test_model_encoding_sizes_and_roundtrip():
    g = load_golden("lifecycle_L128.json")
    c = g["cases"][0]
    recs = {"IssuanceRequest": c["request"], "IssuanceResponse": c["response"], "SpendProof": c["proof"], "Refund": c["refund"],
            "PrivateKey": g["sk"], "PreIssuance": c["pre"], "CreditToken": c["token"], "PreRefund": c["prerefund"]}
    for t, hx in recs.items():
        rec = bytes.fromhex(hx)
        enc = m.cbor_encode(t, rec)
        assert m.cbor_decode(t, enc) == (0, rec)
        assert m.cbor_decode(t, enc + b"trailing") == (0, rec)            # ciborium reads one item
    assert len(m.cbor_encode("SpendProof", bytes.fromhex(c["proof"]))) == 18036
    pk = bytes.fromhex(g["sk"])[32:]
    assert m.cbor_encode("PublicKey", pk) == b"\x58\x20" + pk and m.cbor_decode("PublicKey", b"\x58\x20" + pk) == (0, pk)


def _variants(t, rec, L):
    """(message, expected) pairs built from a valid record: non-canonical but acceptable, and broken ones."""
    enc = m.cbor_encode(t, rec, L)
    spec = m.CBOR_TYPES[t]
    out = [(enc, None), (enc + b"\x00\x01", None), (enc[:-1], None), (enc[:len(enc) // 2], None), (b"", None), (b"\xff", None)]
    bstr = lambda b: b"\x58\x20" + b
    f = [rec[i:i + 32] for i in range(0, len(rec), 32)]
    if spec is None:
        out += [(b"\x5f\x50" + rec[:16] + b"\x50" + rec[16:] + b"\xff", None),          # chunked byte string
                (b"\x58\x1f" + rec[:31], None), (b"\x81" + bstr(rec), None), (b"\xc1" + bstr(rec), None),
                (bstr(b"\x01" + bytes(31)), None)]                                        # not a Ristretto encoding
        return out
    # rebuild entries
    ents, i = [], 0
    for key, kind, shape in spec:
        if shape == 0:
            ents.append((key, bstr(f[i]))); i += 1
        elif shape == 1:
            ents.append((key, m._cbor_head(4, L) + b"".join(bstr(x) for x in f[i:i + L]))); i += L
        else:
            ents.append((key, m._cbor_head(4, L) + b"".join(b"\x82" + bstr(f[i + 2 * j]) + bstr(f[i + 2 * j + 1]) for j in range(L)))); i += 2 * L
    hd = lambda n: m._cbor_head(5, n)
    kb = lambda k: m._cbor_head(0, k)
    body = lambda es: b"".join(kb(k) + v for k, v in es)
    n = len(ents)
    out.append((hd(n) + body(ents[::-1]), None))                                          # reversed key order
    out.append((hd(n + 2) + kb(99) + b"\x63abc" + body(ents) + b"\x20" + b"\x82\x01\xf6", None))   # unknown int key, negative-int key
    out.append((hd(n + 1) + b"\x61k" + b"\xa1\x01\x02" + body(ents), None))               # text key with a nested map value
    out.append((b"\xbf" + body(ents) + b"\xff", None))                                    # indefinite-length map
    out.append((hd(n + 1) + b"\x62\xc3\xa9" + b"\x01" + body(ents), None))                 # text key "e-acute": valid UTF-8, ignored
    out.append((hd(n + 1) + b"\x61\xff" + b"\x01" + body(ents), None))                     # text key that is not UTF-8: ciborium refuses to parse
    out.append((hd(n + 1) + kb(98) + b"\x63\xed\xa0\x80" + body(ents), None))              # text VALUE holding a UTF-16 surrogate: parse error
    out.append((hd(n + 1) + kb(98) + b"\x7f\x62\xc3\xa9\x61\xc3\xff" + body(ents), None))  # chunked text whose second chunk is a truncated sequence
    out.append((hd(n) + b"\x18\x01" + ents[0][1] + body(ents[1:]), None))                 # non-minimal key encoding
    out.append((hd(n + 1) + kb(ents[0][0]) + bstr(bytes(32)) + body(ents), None))         # duplicate key: the last one wins
    out.append((hd(n - 1) + body(ents[1:]), None))                                        # missing field 1
    out.append((hd(n) + kb(ents[0][0]) + b"\x58\x1f" + bytes(31) + body(ents[1:]), None))   # 31-byte string
    out.append((hd(n) + kb(ents[0][0]) + b"\x01" + body(ents[1:]), None))                 # integer instead of bytes
    out.append((b"\x80", None)); out.append((b"\xc0" + enc, None))                        # not a map / tagged map
    out.append((hd(n) + kb(ents[0][0]) + b"\x5f\x50" + f[0][:16] + b"\x50" + f[0][16:] + b"\xff" + body(ents[1:]), None))   # chunked bstr
    # scalar >= l must come out reduced; an invalid point must be rejected
    for idx, (key, kind, shape) in enumerate(spec):
        if kind == "S" and shape == 0:
            big = (m.ELL + 5).to_bytes(32, "little")
            out.append((hd(n) + body(ents[:idx] + [(key, bstr(big))] + ents[idx + 1:]), None)); break
    for idx, (key, kind, shape) in enumerate(spec):
        if kind == "P" and shape == 0:
            out.append((hd(n) + body(ents[:idx] + [(key, bstr(b"\x01" + bytes(31)))] + ents[idx + 1:]), None)); break
    # broken in TWO ways: from_cbor reports the first failure in wire order (src/cbor.rs:276-388), and so must the codec -- an invalid
    # point (InvalidValue) against a mis-shaped or missing field (InvalidStructure), in both orders, and through a duplicate key
    bad_pt, short = bstr(b"\x01" + bytes(31)), b"\x58\x1f" + bytes(31)
    p_idx = next((i for i, (_, kind, shape) in enumerate(spec) if kind == "P" and shape == 0), None)
    s_idx = next((i for i, (_, kind, shape) in enumerate(spec) if kind == "S" and shape == 0), None)
    if p_idx is not None and s_idx is not None:
        both = list(ents); both[p_idx] = (ents[p_idx][0], bad_pt); both[s_idx] = (ents[s_idx][0], short)
        out.append((hd(n) + body(both), None)); out.append((hd(n) + body(both[::-1]), None))          # whichever comes first on the wire
        only_pt = list(ents); only_pt[p_idx] = (ents[p_idx][0], bad_pt)
        rest = [e for i, e in enumerate(only_pt) if i != s_idx]
        out.append((hd(n - 1) + body(rest), None))                                                    # invalid point + a missing field: the point is met first
        out.append((hd(n + 1) + kb(ents[p_idx][0]) + bad_pt + body(ents), None))                      # invalid point under a key that a later duplicate overwrites
        out.append((hd(n + 1) + body(ents) + kb(ents[p_idx][0]) + bad_pt, None))                      # ... and the other way round
    if t == "SpendProof":
        com = dict(ents)[5]
        ch = len(m._cbor_head(4, L))
        bad_com = com[:ch] + com[ch:ch + 34 * (L - 1)] + bad_pt                                       # last element invalid
        sub = lambda k2, v2, es=ents: [(k, v) if k != k2 else (k2, v2) for k, v in es]
        out.append((hd(n) + body(sub(5, m._cbor_head(4, L - 1) + bad_com[ch + 34:])), None))          # Com too short AND its last element invalid: the element is met first
        out.append((hd(n) + body(sub(5, m._cbor_head(4, L + 1) + com[ch:] + bad_pt)), None))          # over-long Com whose extra element is invalid: InvalidValue, not "wrong size"
        out.append((hd(n) + body(sub(5, m._cbor_head(4, L + 1) + com[ch:] + com[ch:ch + 34])), None)) # over-long Com, all valid: wrong size
        out.append((hd(n) + body(sub(5, bstr(bytes(32)), sub(4, bad_pt))), None))                      # B_bar invalid, Com not an array (= missing, reported last)
        z3 = dict(ents)[15]; z3 = z3[:ch] + b"\x83" + z3[ch + 1:ch + 69] + bstr(bytes(32)) + z3[ch + 69:]
        out.append((hd(n) + body(sub(15, z3, sub(5, bad_com))), None))                                 # invalid Com element in front of a 3-element z pair
        out.append((hd(n) + body(sub(15, z3, sub(5, bad_com))[::-1]), None))                           # ... and behind it
        out.append((hd(n) + body([(k, v) if k != 5 else (5, m._cbor_head(4, L - 1) + com[len(m._cbor_head(4, L)):-34]) for k, v in ents]), None))   # Com too short
        out.append((hd(n) + body([(k, v) if k != 5 else (5, b"\x9f" + com[len(m._cbor_head(4, L)):] + b"\xff") for k, v in ents]), None))          # indefinite array
        out.append((hd(n) + body([(k, v) if k != 5 else (5, bstr(bytes(32))) for k, v in ents]), None))                                                # not an array -> missing
        z = dict(ents)[15]
        zh = len(m._cbor_head(4, L))
        out.append((hd(n) + body([(k, v) if k != 15 else (15, z[:zh] + b"\x83" + z[zh + 1:zh + 69] + bstr(bytes(32)) + z[zh + 69:]) for k, v in ents]), None))   # a 3-element pair
        out.append((hd(n) + body([(k, v) if k != 15 else (15, z[:zh] + bstr(bytes(32)) + z[zh + 69:]) for k, v in ents]), None))                                    # pair is not an array
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [None, "3"])
@pytest.mark.parametrize("L", [128, 8])
def test_codec_against_model(engine_factory, bench_params, L, chunk, monkeypatch):
    """chunk = "3": the codec's chunk pipeline (alternating slots, copies out beside copies in) over batches this small"""
    if chunk:
        monkeypatch.setenv("ACT_CBOR_CHUNK_MSGS", chunk)
    eng = engine_factory(bench_params, L, max_batch=8)
    sk = eng.private_key_random(shake("cbor-sk", 64))
    pre = eng.pre_issuance_random(shake("cbor-pre", 128 * 3)); req = eng.request(pre, shake("cbor-rq", 128 * 3))
    st, resp = eng.issue(sk, req, scb(77) * 3, shake("cbor-ir", 128 * 3))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, prer = eng.prove_spend(tok, scb(7) * 3, shake("cbor-pr", eng.prove_rng_bytes * 3))
    st, rf = eng.refund(sk, proofs, shake("cbor-rr", 128 * 3))
    recs = {"IssuanceRequest": req, "IssuanceResponse": resp, "SpendProof": proofs, "Refund": rf, "PrivateKey": sk, "PublicKey": sk[32:],
            "PreIssuance": pre, "CreditToken": tok, "PreRefund": prer}
    for t, blob in recs.items():
        rb = len(blob) // (1 if t in ("PrivateKey", "PublicKey") else 3)
        records = [blob[i:i + rb] for i in range(0, len(blob), rb)]
        enc = eng.cbor_encode(t, blob)
        assert enc == [m.cbor_encode(t, r, L) for r in records], t
        assert len(enc[0]) == eng.cbor_size(t)
        msgs, exp = [], []
        for r in records[:2]:
            for msg, _ in _variants(t, r, L):
                msgs.append(msg); exp.append(m.cbor_decode(t, msg, L))
        st, out = eng.cbor_decode(t, msgs)
        for i, (es, er) in enumerate(exp):
            assert st[i] == es, (t, i, st[i], es, msgs[i][:24].hex())
            assert out[rb * i:rb * i + rb] == er, (t, i)
        assert {0, 1, 2}.issubset(set(st)) and (3 in st or t in ("PreIssuance", "PreRefund"))      # every error class exercised


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [None, "5"])
def test_codec_device_memory_round_trip(engine_factory, bench_params, chunk, monkeypatch):
    """Device-memory callers (one stream, no staging), whole and cut into chunks: encode == the model's bytes, decode gives the
    records back, and messages broken in the middle of a chunk are the only ones the host reader is asked about."""
    import numpy as np
    import torch
    from act_amd import capi
    if chunk:
        monkeypatch.setenv("ACT_CBOR_CHUNK_MSGS", chunk)
    L, D = 8, 23
    eng = engine_factory(bench_params, L, max_batch=32)
    sk = eng.private_key_random(shake("cbd-sk", 64))
    pre = eng.pre_issuance_random(shake("cbd-pre", 128 * D)); req = eng.request(pre, shake("cbd-rq", 128 * D))
    st, resp = eng.issue(sk, req, scb(77) * D, shake("cbd-ir", 128 * D))
    st, tok = eng.issuance_to_credit_token(pre, sk[32:], req, resp)
    st, proofs, _ = eng.prove_spend(tok, scb(7) * D, shake("cbd-pr", eng.prove_rng_bytes * D))
    T = capi.CBOR_TYPES["SpendProof"]; pb = eng.proof_bytes; ml = eng.cbor_size("SpendProof")
    d_recs = torch.from_numpy(np.frombuffer(proofs, np.uint8).copy()).cuda()
    d_wire = torch.zeros(D * ml, dtype=torch.uint8, device="cuda"); d_back = torch.zeros(D * pb, dtype=torch.uint8, device="cuda")
    d_st = torch.full((D,), 9, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    eng._ck(eng.lib.act_cbor_encode_batch(eng.ctx, T, D, capi.MEM_DEVICE, d_recs.data_ptr(), d_wire.data_ptr()))
    wire = d_wire.cpu().numpy().tobytes()
    assert [wire[i * ml:(i + 1) * ml] for i in range(D)] == [m.cbor_encode("SpendProof", proofs[i * pb:(i + 1) * pb], L) for i in range(D)]
    d_wire[7 * ml] = 0x40                               # message 7: a byte string, not a map
    d_wire[12 * ml + 1] = 0x18                          # message 12: its first key now reads as a two-byte head: no longer canonical
    torch.cuda.synchronize()
    eng._ck(eng.lib.act_cbor_decode_batch(eng.ctx, T, D, capi.MEM_DEVICE, d_wire.data_ptr(), None, d_back.data_ptr(), d_st.data_ptr()))
    st_h = d_st.cpu().numpy(); back = d_back.cpu().numpy().tobytes()
    broken = wire[:7 * ml] + b"\x40" + wire[7 * ml + 1:12 * ml + 1] + b"\x18" + wire[12 * ml + 2:]
    for i in range(D):
        es, er = m.cbor_decode("SpendProof", broken[i * ml:(i + 1) * ml], L)
        assert st_h[i] == es, (i, st_h[i], es)
        assert back[i * pb:(i + 1) * pb] == er, i
    assert st_h[7] != 0 and st_h[12] != 0 and int((st_h == 0).sum()) == D - 2
