"""Builds tests/abi_conformance.cpp against include/act_mi355x.h, links it to libact_mi355x.so and runs it on the GPU: a
caller with no Python in the process replays the libsodium-made L = 128 lifecycle fixture through every hot-path entry
point (single-context, split check/sign, node handle) and must reproduce every byte."""
import os
import struct
import subprocess

import pytest

from conftest import ELL, ROOT, load_golden, shake

pytestmark = pytest.mark.gpu
hx = bytes.fromhex


def write_fixture(path, oracle):
    g = load_golden("sodium_lifecycle_L128.json")
    L, cases = g["L"], g["cases"]
    n = len(cases)
    tag = lambda i: g["tag_fmt"] % i
    cat = lambda f: b"".join(f(i) for i in range(n))
    scb = lambda v: (v % ELL).to_bytes(32, "little")
    rb = 64 * (4 * L + 12)
    sk = hx(g["sk"])
    proof_in = cat(lambda i: hx(cases[i]["proof"]))
    refund_rng = cat(lambda i: shake(tag(i) + "-refund", 128))
    # proofs as the prover makes them (before tampering): recompute with the oracle for tampered cases
    octx = oracle.ctx(hx(g["params"]), L)
    made, seq, cur, tok2, st_tok2 = [], [], 0, [], []
    pb = octx.proof_bytes
    for i, c in enumerate(cases):
        st, proof, prer = octx.prove_spend(hx(c["token"]), scb(int(c["s"])), shake(tag(i) + "-prove", rb))
        assert prer.hex() == c["prerefund"]
        if c["tamper"] is None:
            assert proof.hex() == c["proof"]
        made.append(proof)
        st, rf = octx.refund(sk, hx(c["proof"]), refund_rng[128 * cur:128 * cur + 128])     # ACT_RNG_SEQUENTIAL: next slice of the one stream
        assert st == c["status"]
        seq.append(rf); cur += st == 0
        if c["status"] == 0:
            tok2.append(hx(c["token2"])); st_tok2.append(0)
        else:
            s2, t2 = octx.refund_to_credit_token(hx(c["prerefund"]), hx(c["proof"]), bytes(128), sk[32:])
            assert s2 != 0
            tok2.append(bytes(160)); st_tok2.append(s2)
    blobs = [a.encode() for a in g["params_args"]] + [
        hx(g["params"]), shake(g["sk_label"], 64), sk, hx(g["sk_other"]),
        cat(lambda i: shake(tag(i) + "-pre", 128)), cat(lambda i: hx(cases[i]["pre"])),
        cat(lambda i: shake(tag(i) + "-request", 128)), cat(lambda i: hx(cases[i]["request"])),
        cat(lambda i: scb(int(cases[i]["c"]))), cat(lambda i: shake(tag(i) + "-issue", 128)), cat(lambda i: hx(cases[i]["response"])),
        cat(lambda i: hx(cases[i]["token"])), cat(lambda i: scb(int(cases[i]["s"]))), cat(lambda i: shake(tag(i) + "-prove", rb)),
        b"".join(made), cat(lambda i: hx(cases[i]["prerefund"])), proof_in,
        bytes(c["status"] for c in cases), cat(lambda i: hx(cases[i]["kprime"]) if cases[i]["status"] == 0 else bytes(32)),
        refund_rng, cat(lambda i: hx(cases[i]["refund"])), b"".join(tok2), bytes(st_tok2), bytes(c["status_other_issuer"] for c in cases),
        b"".join(seq)]
    with open(path, "wb") as f:
        f.write(struct.pack("<QQ", L, n))
        for b in blobs:
            f.write(struct.pack("<Q", len(b))); f.write(b)


def test_cpp_caller_reproduces_the_fixture(tmp_path, oracle):
    from act_amd import capi
    capi.load()
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = str(tmp_path / "abi_conformance")
    rocm = "/opt/rocm/lib"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_conformance.cpp"),
                    "-o", exe, "-L" + libdir, "-lact_mi355x", "-L" + rocm, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + rocm], check=True)
    fx = str(tmp_path / "fixture.bin")
    write_fixture(fx, oracle)
    r = subprocess.run([exe, fx], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "conformance: all entry points reproduce the fixture" in r.stdout
