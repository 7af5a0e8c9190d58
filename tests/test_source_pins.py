"""Orders and polarities of the model against the reference's SOURCE TEXT.

The oracle's arithmetic is pinned by third-party code (libsodium + LLVM BLAKE3, tests/golden/sodium_*.json), but the crate holds
no test vectors and cannot be built here, so one failure mode stays open: a misreading of /root/reference/src/lib.rs that every
restatement shares -- the order of the `Scalar::random` draws, the order of the elements a `Transcript::with` closure adds, which
argument of a `conditional_select` is taken, the key order of the CBOR maps the raw records follow.  oracle/pymodel.py keeps exactly
those as tables (DRAW_ORDER, TRANSCRIPT_ORDER, SELECTS, RECORD_ORDER) and is DRIVEN by them; this test extracts the same tables
from the Rust source with regular expressions and compares.  Swap one draw, one transcript element or one select argument in the
model and it fails here; swap it in the C oracle, the libsodium model or the kernels and the byte-for-byte tests against pymodel fail.

Build container only: nothing of the reference travels to the GPU box (skipped when /root/reference is absent).  This does not make
parity "green" -- only running the crate can -- it removes the one failure mode the libsodium anchor cannot see."""
import os
import re

import pytest

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "lib.rs")), reason="the reference source is not on this machine")


def _src(name):
    text = open(os.path.join(REF, name)).read()
    text = re.sub(r"//[^\n]*", "", text)                 # comments (doc examples among them) are not code
    return text


def _functions(text):
    """[(name, body)] for every `pub fn` / `fn`, body = text up to the next fn at the same or lower indentation (good enough here:
    the closures we look for live inside their function's braces)."""
    heads = [(m.start(), m.group(1)) for m in re.finditer(r"\bfn\s+([a-z_0-9]+)\s*[(<]", text)]
    out = []
    for k, (pos, name) in enumerate(heads):
        end = heads[k + 1][0] if k + 1 < len(heads) else len(text)
        out.append((name, text[pos:end]))
    return out


def _norm(name):
    name = name.strip().lstrip("&").strip()
    for pre in ("self.public.", "self.", "spend_proof.", "request.", "response.", "refund.", "public_key.", "public."):
        if name.startswith(pre):
            name = name[len(pre):]
    return name


# ---- rng draws ---------------------------------------------------------------------------------------------------------------
def _draws(body):
    """Bindings of Scalar::random(&mut rng) in textual (= execution) order.  Three shapes occur in src/lib.rs: `let x = Scalar::random`,
    `let x: Vec<Scalar> = (0..L).map(|_| Scalar::random(&mut rng)).collect()`, and `for v in x.iter_mut() { *v = Scalar::random }`."""
    out = []
    pat = re.compile(r"let\s+(?:mut\s+)?([a-z_0-9]+)\s*(?::\s*[^=]+)?=\s*(\(0\.\.L\)\s*\.map\(\|_\|\s*)?Scalar::random\(&mut rng\)"
                     r"|for\s+[a-z_]+\s+in\s+([a-z_0-9]+)\.iter_mut\(\)\s*\{\s*\*[a-z_]+\s*=\s*Scalar::random\(&mut rng\);")
    for m in pat.finditer(body):
        if m.group(3):
            out.append(m.group(3) + "[]")
        else:
            out.append(m.group(1) + ("[]" if m.group(2) else ""))
    return tuple(out)


def test_rng_draw_order():
    import pymodel as m
    fns = dict(_functions(_src("lib.rs")))
    for fn, want in m.DRAW_ORDER.items():
        got = _draws(fns[fn])
        assert got == want, (fn, got, want)
        assert fns[fn].count("Scalar::random(&mut rng)") == len(want), fn      # no draw of another shape slipped past the pattern
    # and the issuer draws only after its check has passed (src/lib.rs:638-643, 842-846)
    for fn, err in (("issue", "InvalidIssuanceRequestProof"), ("refund", "InvalidClientSpendProof")):
        assert fns[fn].index(err) < fns[fn].index("Scalar::random(&mut rng)"), fn


# ---- transcript elements ------------------------------------------------------------------------------------------------------
def _closure_items(closure):
    items = []
    pos = 0
    pat = re.compile(r"for\s+[a-z_]+\s+in\s+([a-z_0-9.]+)\.iter\(\)\s*\{\s*transcript\.add_elements\([a-z_]+\.iter\(\)\);\s*\}"
                     r"|transcript\.add_(scalar|element)\(([^;]*?)\);"
                     r"|transcript\.add_(scalars|elements)\(([^;]*?)\);")
    for m in pat.finditer(closure):
        assert closure[pos:m.start()].strip() == "", closure[pos:m.start()]       # nothing between two adds that we do not understand
        pos = m.end()
        if m.group(1):
            items.append(_norm(m.group(1)) + "[][]")
        elif m.group(2):
            items.append(_norm(m.group(3)))
        else:
            arg = m.group(5).strip()
            lst = re.fullmatch(r"\[([^\]]*)\]\.into_iter\(\)", arg)
            it = re.fullmatch(r"([a-z_0-9.]+)\.iter\(\)", arg)
            assert lst or it, arg
            if lst:
                items += [_norm(x) for x in lst.group(1).split(",") if x.strip()]
            else:
                items.append(_norm(it.group(1)) + "[]")
    assert closure[pos:].strip() == "", closure[pos:]
    return tuple(items)


def test_transcript_element_order():
    import pymodel as m
    seen = {}
    for fn, body in _functions(_src("lib.rs")):
        for mm in re.finditer(r'Transcript::with\(params,\s*b"([a-z]+)",\s*\|transcript\|\s*\{(.*?)\}\);', body, re.S):
            seen[(fn, mm.group(1))] = _closure_items(mm.group(2))
    assert seen == dict(m.TRANSCRIPT_ORDER), {k: (seen.get(k), m.TRANSCRIPT_ORDER.get(k)) for k in set(seen) | set(m.TRANSCRIPT_ORDER)
                                                 if seen.get(k) != m.TRANSCRIPT_ORDER.get(k)}
    # scalars go in as `as_bytes`, points compressed, each behind a u64_be length (src/transcript.rs): add_scalar(s) / add_element(e)
    tr = _src("transcript.rs")
    assert re.search(r"fn add_element.*?compress\(\)", tr, re.S) and re.search(r"fn add_scalar.*?as_bytes\(\)", tr, re.S)
    assert "to_be_bytes()" in tr


# ---- conditional_select ---------------------------------------------------------------------------------------------------------
def _split_args(text):
    """top-level comma split of a call's argument text"""
    args, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur)
    return [re.sub(r"\s+", "", a) for a in args]


def _classify(expr):
    e = re.sub(r"big_c\[(?:0|j)\]\[([01])\]", r"big_c[.]<\1>", expr)      # the second index is the OR branch, not the bit
    e = re.sub(r"\[(0|j)\]", "[.]", e).replace("<0>", "[0]").replace("<1>", "[1]")      # bit 0 and bit j share one table entry
    rules = [
        (r"^&\((&params\.h2\*&w0\+)?&params\.h3\*&z\[\.\]-big_c\[\.\]\[([01])\]\*gamma_i\[\.\]\)$", lambda mm: "sim[%s]" % mm.group(2)),
        (r"^&\((&params\.h2\*&k0_prime\+)?&params\.h3\*&s_i_prime\[\.\]\)$", lambda mm: "real"),
        (r"^&gamma_i\[\.\]$", lambda mm: "gamma_i"),
        (r"^&\(gamma-gamma_i\[\.\]\)$", lambda mm: "gamma-gamma_i"),
        (r"^&w0$", lambda mm: "w0"),
        (r"^&z\[\.\]$", lambda mm: "z"),
        (r"^&\(gamma00\[\.\]\*k_star\+k0_prime\)$", lambda mm: "resp_k0"),
        (r"^&\(\(gamma-gamma00\[\.\]\)\*k_star\+k0_prime\)$", lambda mm: "resp_k1"),
        (r"^&\(gamma00\[\.\]\*s_i\[\.\]\+s_i_prime\[\.\]\)$", lambda mm: "resp0"),
        (r"^&\(\(gamma-gamma00\[\.\]\)\*s_i\[\.\]\+s_i_prime\[\.\]\)$", lambda mm: "resp1"),
    ]
    for pat, tag in rules:
        mm = re.match(pat, e)
        if mm:
            return tag(mm)
    raise AssertionError("conditional_select argument not understood: " + expr)


def test_conditional_select_polarity():
    import pymodel as m
    body = dict(_functions(_src("lib.rs")))["prove_spend"]
    found, calls = {}, 0
    for mm in re.finditer(r"(?:let\s+)?([a-z_0-9\[\]]+)\s*=\s*(?:RistrettoPoint|Scalar)::conditional_select\(", body):
        calls += 1
        # the call's argument text: up to the matching parenthesis
        i, depth = mm.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(body[i], 0)
            i += 1
        a, b, choice = _split_args(body[mm.end():i - 1])
        assert re.fullmatch(r"i\[(0|j)\]\.ct_eq\(&Scalar::ZERO\)", choice), choice      # choice true <=> the bit is 0
        target = re.sub(r"\[(0|j)\]", "[.]", mm.group(1), count=1)
        pair = (_classify(a), _classify(b))
        # the bit-0 and bit-j statements of one target must agree with each other ...
        assert found.setdefault(target, pair) == pair, (target, found[target], pair)
        # ... and the branch index of a simulated commitment must be the target's
        for tag in pair:
            if tag.startswith("sim["):
                assert tag == "sim[%s]" % target[-2], (target, pair)
    assert found == dict(m.SELECTS), {k: (found.get(k), m.SELECTS.get(k)) for k in set(found) | set(m.SELECTS) if found.get(k) != m.SELECTS.get(k)}
    assert body.count("conditional_select(") == calls == 12        # C'[.][0], C'[.][1], gamma00, z00[.][0], z00[.][1] at bit 0 and in the loop; w00, w01
    # subtle: conditional_select(a, b, choice) yields b when choice is true -- pymodel._select takes b when the bit is 0
    proof0, _ = _prove(m, 0)
    proof1, _ = _prove(m, 1)
    assert proof0.record() != proof1.record()


def _prove(m, low_bit):
    """prove_spend on a toy token whose remaining balance has the given low bit: both branches of every select run."""
    import hashlib
    rng = m.ByteRng(hashlib.shake_256(b"select-polarity").digest(64 * 40))
    params = m.Params.new("a", "b", "c", "d")
    tok = m.CreditToken(m.BASEPOINT, 5, 7, 9, 10 + low_bit)
    return m.prove_spend(tok, params, 10, rng, nbits=2)


# ---- record / CBOR key order --------------------------------------------------------------------------------------------------
def test_record_field_order():
    import dataclasses
    import pymodel as m
    text = _src("cbor.rs")
    found = {}
    for mm in re.finditer(r"impl\s+([A-Za-z]+)\s*\{(.*?)\n\}", text, re.S):
        tc = re.search(r"fn to_cbor.*?let map = vec!\[(.*?)\];", mm.group(2), re.S)
        if not tc:
            continue
        keys, names = [], []
        for e in re.finditer(r"\(Value::Integer\((\d+)\.into\(\)\),\s*(.*?)\),\s*(?=\(Value::Integer|$)", tc.group(1).strip() + "\n", re.S):
            keys.append(int(e.group(1)))
            arg = re.sub(r"\s+", "", e.group(2))
            f = re.fullmatch(r"encode_(?:point|scalar)\((.*)\)", arg) or re.fullmatch(r"Value::Array\(([a-z0-9_]+)_array\)", arg)
            assert f, arg
            names.append(_norm(f.group(1)))
        assert keys == list(range(1, len(keys) + 1)), (mm.group(1), keys)               # deterministic encoding: ascending integer keys
        found[mm.group(1)] = tuple(names)
    assert found == dict(m.RECORD_ORDER), {k: (found.get(k), m.RECORD_ORDER.get(k)) for k in set(found) | set(m.RECORD_ORDER) if found.get(k) != m.RECORD_ORDER.get(k)}
    # the model's dataclasses are built positionally by its parsers: their field order must be the record order too
    for name, order in m.RECORD_ORDER.items():
        assert tuple(f.name for f in dataclasses.fields(getattr(m, name))) == order, name


def test_the_tables_drive_the_model():
    """Swapping an entry of a table changes the model's output: the tables are what the functions execute, not documentation."""
    import hashlib
    import pymodel as m
    params = m.Params.new("a", "b", "c", "d")
    pre = m.PreIssuance(3, 4)
    base = m.request(pre, params, m.ByteRng(hashlib.shake_256(b"drive").digest(128))).record()
    saved = dict(m.DRAW_ORDER), dict(m.TRANSCRIPT_ORDER), dict(m.RECORD_ORDER), dict(m.SELECTS)
    try:
        m.DRAW_ORDER["request"] = ("r_prime", "k_prime")
        assert m.request(pre, params, m.ByteRng(hashlib.shake_256(b"drive").digest(128))).record() != base
        m.DRAW_ORDER.update(saved[0])
        m.TRANSCRIPT_ORDER[("request", "request")] = ("k1", "big_k")
        assert m.request(pre, params, m.ByteRng(hashlib.shake_256(b"drive").digest(128))).record() != base
        m.TRANSCRIPT_ORDER.update(saved[1])
        m.RECORD_ORDER["IssuanceRequest"] = ("big_k", "k_bar", "gamma", "r_bar")
        assert m.request(pre, params, m.ByteRng(hashlib.shake_256(b"drive").digest(128))).record() != base
        m.RECORD_ORDER.update(saved[2])
        p0 = _prove(m, 0)[0].record()
        m.SELECTS["w00"] = ("resp_k0", "w0")
        assert _prove(m, 0)[0].record() != p0
    finally:
        m.DRAW_ORDER.update(saved[0]); m.TRANSCRIPT_ORDER.update(saved[1]); m.RECORD_ORDER.update(saved[2]); m.SELECTS.update(saved[3])
    assert m.request(pre, params, m.ByteRng(hashlib.shake_256(b"drive").digest(128))).record() == base
