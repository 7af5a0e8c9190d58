"""rust/ cannot be compiled here (no toolchain), so every .rs file is at least READ the way a compiler's lexer reads it (VERDICT r5
#3, weak #14): comments (nested block comments included), string / raw-string / byte-string and character literals are recognised
and removed, and what is left must have balanced (), [] and {} -- a missing brace, an unterminated string or a stray quote in 1 400
uncompiled lines fails here instead of at a maintainer's first `cargo build`.  On top of that:
  * every kept single-call signature in rust/src/mi355x.rs is compared, type by type and name by name, with the reference's own
    declaration in /root/reference/src/lib.rs (463, 528-534, 621-627, 781-786, 972-977, 1217-1223) -- skipped where the reference is
    absent;
  * every `extern "C"` declaration is a well-formed `fn name(args) -> ret;` (its types against the header: tests/test_abi_prototypes.py);
  * rust/pin is the self-contained package its README says it is (a manifest that names the crate at =0.2.1, a test that is the
    same file as rust/tests/golden.rs, fixtures reachable from it)."""
import os
import re

import pytest

from conftest import ROOT

RUST = os.path.join(ROOT, "rust")
FILES = sorted(os.path.join(d, f) for d, _, fs in os.walk(RUST) for f in fs if f.endswith(".rs") and not os.path.islink(os.path.join(d, f)))


def strip_rust(src: str, path: str = "") -> str:
    """Source text with comments and literals replaced by spaces (newlines kept); raises on anything unterminated."""
    out, i, n = [], 0, len(src)
    line = lambda k: src.count("\n", 0, k) + 1
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            depth, j = 1, i + 2
            while depth and j < n:
                if src.startswith("/*", j):
                    depth += 1; j += 2
                elif src.startswith("*/", j):
                    depth -= 1; j += 2
                else:
                    j += 1
            assert depth == 0, "%s:%d: unterminated block comment" % (path, line(i))
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == '"' or (c in "br" and re.match(r'b?r#*"|b"', src[i:i + 8]) and (i == 0 or not (src[i - 1].isalnum() or src[i - 1] == "_"))):
            m = re.match(r'(b?)(r(#*))?"', src[i:i + 40])
            raw, hashes = m.group(2) is not None, m.group(3) or ""
            j = i + m.end()
            if raw:
                end = src.find('"' + hashes, j)
                assert end >= 0, "%s:%d: unterminated raw string" % (path, line(i))
                j = end + 1 + len(hashes)
            else:
                while j < n and src[j] != '"':
                    j += 2 if src[j] == "\\" else 1
                assert j < n, "%s:%d: unterminated string literal" % (path, line(i))
                j += 1
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == "'" or (c == "b" and src.startswith("b'", i) and (i == 0 or not (src[i - 1].isalnum() or src[i - 1] == "_"))):
            k = i + (2 if c == "b" else 1)
            m = re.match(r"(\\(x[0-9a-fA-F]{2}|u\{[0-9a-fA-F_]{1,8}\}|.)|[^\\'\n])'", src[k:k + 14])
            if m:                                   # a character literal
                out.append(" " * (k - i + m.end())); i = k + m.end()
            else:                                   # a lifetime / loop label: 'a, 'static, '_
                assert c == "'" and re.match(r"[A-Za-z_]", src[k:k + 1] or " "), "%s:%d: stray quote" % (path, line(i))
                out.append(c); i += 1
        else:
            out.append(c); i += 1
    return "".join(out)


def check_balanced(code: str, path: str):
    pairs = {")": "(", "]": "[", "}": "{"}
    stack = []
    ln = 1
    for ch in code:
        if ch == "\n":
            ln += 1
        elif ch in "([{":
            stack.append((ch, ln))
        elif ch in ")]}":
            assert stack, "%s:%d: unmatched %r" % (path, ln, ch)
            op, at = stack.pop()
            assert op == pairs[ch], "%s:%d: %r closes %r opened at line %d" % (path, ln, ch, op, at)
    assert not stack, "%s: %r opened at line %d is never closed" % (path, stack[-1][0], stack[-1][1])


def test_the_lexer_itself():
    ok = 'fn f<\'a>(x: &\'a str) -> char { let _s = "a}\\"{"; let _r = r#"}"{"#; let _b = b\'{\'; /* { /* nested } */ } */ \'}\' } // }'
    check_balanced(strip_rust(ok), "ok")
    for bad in ('fn f() { let s = "abc; }', "fn f() { (1, 2] }", "fn f() { /* never closed", "fn f() {{ }"):
        with pytest.raises(AssertionError):
            check_balanced(strip_rust(bad, "bad"), "bad")


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, RUST) for f in FILES])
def test_every_rust_file_lexes_and_balances(path):
    src = open(path).read()
    code = strip_rust(src, path)
    check_balanced(code, path)
    assert "\t" not in src
    # statement-level sanity a lexer can offer: no two items glued together by a lost brace or semicolon
    assert not re.search(r"\bfn\s+\w+\s*\([^)]*\)\s*(->[^{;]+)?\s*\bfn\b", code), path


def _sig(text: str, name: str, start: int = 0):
    """(params as [(name, type)], return type) of the first `pub fn name(` at or after `start`, whitespace-normalised; `mut` bindings and
    a trailing comma do not count"""
    m = re.compile(r"pub fn %s\s*\(" % name).search(text, start)
    assert m, name
    depth, j = 1, m.end()
    while depth:
        depth += {"(": 1, ")": -1}.get(text[j], 0); j += 1
    params = text[m.end():j - 1]
    rest = text[j:]
    ret = re.match(r"\s*->\s*([^{]+)\{", rest)
    norm = lambda t: re.sub(r"\s+", " ", t).strip().rstrip(",").strip()
    ps = []
    for p in re.split(r",(?![^()<>]*[)>])", norm(params)):
        p = norm(p)
        if not p:
            continue
        if p in ("&self", "self", "&mut self"):
            ps.append((p, ""))
        else:
            nm, ty = p.split(":", 1)
            ps.append((norm(nm).replace("mut ", ""), norm(ty)))
    return ps, norm(ret.group(1)) if ret else "", m.start()


REF = "/root/reference/src/lib.rs"


@pytest.mark.skipif(not os.path.exists(REF), reason="the reference source is not on this machine")
def test_kept_signatures_are_the_crates():
    ref = strip_rust(open(REF).read(), REF)
    mine = strip_rust(open(os.path.join(RUST, "src", "mi355x.rs")).read())
    # the kept signatures are the last impl blocks of the file (the lexer has blanked the marker comment): from the `request` that
    # takes &self (the batch sibling above it takes a slice)
    kept = mine[mine.index("impl PreIssuance {\n    pub fn request(&self"):]

    def in_impl(text, impl, name):
        """signature of `name` inside the first `impl <impl> {` block that has it"""
        for m in re.finditer(r"impl %s \{" % impl, text):
            depth, j = 1, m.end()
            while depth:
                depth += {"{": 1, "}": -1}.get(text[j], 0); j += 1
            block = text[m.end():j]
            if re.search(r"pub fn %s\s*\(" % name, block):
                return _sig(block, name)[:2]
        raise AssertionError("%s::%s not found" % (impl, name))

    for impl, name, line in (("PreIssuance", "request", 463), ("PreIssuance", "to_credit_token", 528), ("PrivateKey", "issue", 621),
                             ("PrivateKey", "refund", 781), ("CreditToken", "prove_spend", 972), ("PreRefund", "to_credit_token", 1217)):
        want = in_impl(ref, impl, name)
        got = in_impl(kept, impl, name)
        assert got == want, "%s::%s (src/lib.rs:%d): binding %r, crate %r" % (impl, name, line, got, want)
        # and the declaration really sits where the docs cite it
        assert re.search(r"pub fn %s\s*\(" % name, "\n".join(open(REF).read().split("\n")[line - 1:line + 1])), (name, line)


def test_extern_block_is_well_formed():
    src = strip_rust(open(os.path.join(RUST, "src", "mi355x.rs")).read())
    blocks = list(re.finditer(r"extern\s+\{", src))            # `extern "C" {`: the lexer has blanked the string literal
    assert blocks, 'no extern block found (string literal "C" is blanked by the lexer)'
    depth, j = 1, blocks[0].end()
    while depth:
        depth += {"{": 1, "}": -1}.get(src[j], 0); j += 1
    body = src[blocks[0].end():j - 1]
    decls = [d.strip() for d in body.split(";") if d.strip()]
    assert len(decls) >= 20
    for d in decls:
        assert re.fullmatch(r"fn act_\w+\s*\((?:\s*\w+\s*:\s*[^,()]+,?)*\s*\)(\s*->\s*[\w* ]+)?", re.sub(r"\s+", " ", d)), d[:160]


def test_pin_package_is_self_contained():
    pin = os.path.join(RUST, "pin")
    man = open(os.path.join(pin, "Cargo.toml")).read()
    assert re.search(r'^anonymous-credit-tokens = "=0\.2\.1"$', man, re.M) and "serde_json" in man and "curve25519-dalek" in man and "rand_core" in man
    assert re.search(r'^version = "0\.2\.1"$', open("/root/reference/Cargo.toml").read(), re.M) if os.path.exists("/root/reference/Cargo.toml") else True
    t = os.path.join(pin, "tests", "golden.rs")
    assert os.path.islink(t) and os.path.samefile(t, os.path.join(RUST, "tests", "golden.rs"))
    src = open(t).read()
    assert 'env!("CARGO_MANIFEST_DIR")' in src and "../../tests/golden" in src
    for f in re.findall(r'run_file\("([^"]+)"', src):
        assert os.path.exists(os.path.join(pin, "..", "..", "tests", "golden", f)), f
    # every external crate the test names is a dependency of the package
    for crate in set(re.findall(r"^use (\w+)::", src, re.M)) | set(re.findall(r"\b(serde_json)::", src)):
        assert crate in man or crate.replace("_", "-") in man, crate
    assert os.path.exists(os.path.join(pin, "src", "lib.rs")) and "cargo test" in open(os.path.join(pin, "README.md")).read()
