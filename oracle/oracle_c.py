"""ctypes binding of oracle/libact_oracle.so (the C oracle).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)


def build(native_out: str | None = None) -> str:
    """Compile the oracle with gcc.  native_out: also build a -march=native copy at that path."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    if native_out:
        subprocess.run(["make", "-C", _HERE, "-s", "native", f"OUT={native_out}"], check=True)
        return native_out
    return os.path.join(_HERE, "libact_oracle.so")


def _buf(b: bytes):
    return (C.c_uint8 * len(b)).from_buffer_copy(b)


def _out(n: int):
    return (C.c_uint8 * n)()


class Oracle:
    """Thin wrapper; every method takes and returns `bytes` in the raw record layouts."""

    def __init__(self, path: str | None = None):
        path = path or os.path.join(_HERE, "libact_oracle.so")
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        lib.oracle_ctx_new.restype = C.c_void_p
        lib.oracle_ctx_new.argtypes = [C.c_void_p, C.c_int]
        lib.oracle_ctx_free.argtypes = [C.c_void_p]
        for name in ("oracle_spend_proof_bytes", "oracle_prove_rng_bytes", "oracle_spend_transcript_bytes"):
            getattr(lib, name).restype = C.c_size_t
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.oracle_blake3.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        self._ctxs = []

    # primitives -----------------------------------------------------------------------------
    def blake3(self, data: bytes, n: int = 32) -> bytes:
        o = _out(n)
        self.lib.oracle_blake3(_buf(data) if data else None, len(data), o, n)
        return bytes(o)

    def sc_reduce_wide(self, b: bytes) -> bytes:
        o = _out(32); self.lib.oracle_sc_reduce_wide(_buf(b), o); return bytes(o)

    def sc_muladd(self, a: bytes, b: bytes, c: bytes) -> bytes:
        o = _out(32); self.lib.oracle_sc_muladd(_buf(a), _buf(b), _buf(c), o); return bytes(o)

    def sc_invert(self, a: bytes) -> bytes:
        o = _out(32); self.lib.oracle_sc_invert(_buf(a), o); return bytes(o)

    def decode_encode(self, b: bytes):
        o = _out(32); ok = self.lib.oracle_ristretto_decode_encode(_buf(b), o); return bool(ok), bytes(o)

    def from_uniform(self, b: bytes) -> bytes:
        o = _out(32); self.lib.oracle_ristretto_from_uniform(_buf(b), o); return bytes(o)

    def mul(self, pt: bytes, sc: bytes) -> bytes:
        o = _out(32); assert self.lib.oracle_ristretto_mul(_buf(pt), _buf(sc), o); return bytes(o)

    def mul_base(self, sc: bytes) -> bytes:
        o = _out(32); self.lib.oracle_ristretto_mul_base(_buf(sc), o); return bytes(o)

    def add(self, a: bytes, b: bytes) -> bytes:
        o = _out(32); assert self.lib.oracle_ristretto_add(_buf(a), _buf(b), o); return bytes(o)

    def params_new(self, org: str, svc: str, dep: str, ver: str) -> bytes:
        o = _out(96)
        self.lib.oracle_params_new(org.encode(), svc.encode(), dep.encode(), ver.encode(), o)
        return bytes(o)

    # context --------------------------------------------------------------------------------
    def ctx(self, h: bytes, L: int = 128) -> "OracleCtx":
        p = self.lib.oracle_ctx_new(_buf(h), L)
        if not p:
            raise ValueError("oracle_ctx_new failed (bad params encoding or L)")
        return OracleCtx(self, p, L)


class OracleCtx:
    def __init__(self, o: Oracle, p: int, L: int):
        self.o, self.lib, self.p, self.L = o, o.lib, C.c_void_p(p), L
        self.proof_bytes = self.lib.oracle_spend_proof_bytes(self.p)
        self.prove_rng_bytes = self.lib.oracle_prove_rng_bytes(self.p)
        self.transcript_bytes = self.lib.oracle_spend_transcript_bytes(self.p)

    def __del__(self):
        try:
            self.lib.oracle_ctx_free(self.p)
        except Exception:
            pass

    def private_key_random(self, rng: bytes) -> bytes:
        o = _out(64); self.lib.oracle_private_key_random(_buf(rng), o); return bytes(o)

    def pre_issuance_random(self, rng: bytes) -> bytes:
        o = _out(64); self.lib.oracle_pre_issuance_random(_buf(rng), o); return bytes(o)

    def request(self, pre: bytes, rng: bytes) -> bytes:
        o = _out(128); self.lib.oracle_request(self.p, _buf(pre), _buf(rng), o); return bytes(o)

    def issue(self, sk: bytes, req: bytes, c: bytes, rng: bytes):
        o = _out(160); st = self.lib.oracle_issue(self.p, _buf(sk), _buf(req), _buf(c), _buf(rng), o); return st, bytes(o)

    def issuance_to_credit_token(self, pre: bytes, w: bytes, req: bytes, resp: bytes):
        o = _out(160)
        st = self.lib.oracle_issuance_to_credit_token(self.p, _buf(pre), _buf(w), _buf(req), _buf(resp), o)
        return st, bytes(o)

    def prove_spend(self, tok: bytes, s: bytes, rng: bytes):
        assert len(rng) >= self.prove_rng_bytes
        o = _out(self.proof_bytes); pr = _out(96)
        st = self.lib.oracle_prove_spend(self.p, _buf(tok), _buf(s), _buf(rng), o, pr)
        return st, bytes(o), bytes(pr)

    def verify_spend(self, sk: bytes, proof: bytes, want_transcript: bool = False):
        tr = _out(self.transcript_bytes) if want_transcript else None
        kp = _out(32)
        st = self.lib.oracle_verify_spend(self.p, _buf(sk), _buf(proof), tr, kp)
        return (st, bytes(kp), bytes(tr)) if want_transcript else (st, bytes(kp))

    def refund(self, sk: bytes, proof: bytes, rng: bytes):
        o = _out(128); st = self.lib.oracle_refund(self.p, _buf(sk), _buf(proof), _buf(rng), o); return st, bytes(o)

    def refund_to_credit_token(self, prerefund: bytes, proof: bytes, refund: bytes, w: bytes):
        o = _out(160)
        st = self.lib.oracle_refund_to_credit_token(self.p, _buf(prerefund), _buf(proof), _buf(refund), _buf(w), o)
        return st, bytes(o)

    # batches (threaded) ---------------------------------------------------------------------
    def verify_spend_batch(self, sk: bytes, proofs: bytes, nthreads: int = 1) -> bytes:
        n = len(proofs) // self.proof_bytes
        st = _out(n)
        self.lib.oracle_verify_spend_batch(self.p, _buf(sk), C.c_size_t(n), _buf(proofs), st, nthreads)
        return bytes(st)

    def refund_batch(self, sk: bytes, proofs: bytes, rng: bytes, nthreads: int = 1):
        n = len(proofs) // self.proof_bytes
        st = _out(n); o = _out(128 * n)
        self.lib.oracle_refund_batch(self.p, _buf(sk), C.c_size_t(n), _buf(proofs), _buf(rng), o, st, nthreads)
        return bytes(st), bytes(o)

    def prove_spend_batch(self, toks: bytes, s: bytes, rng: bytes, nthreads: int = 1):
        n = len(toks) // 160
        o = _out(self.proof_bytes * n); pr = _out(96 * n)
        self.lib.oracle_prove_spend_batch(self.p, C.c_size_t(n), _buf(toks), _buf(s), _buf(rng), o, pr, nthreads)
        return bytes(o), bytes(pr)

    def issue_batch(self, sk: bytes, reqs: bytes, c: bytes, rng: bytes, nthreads: int = 1):
        n = len(reqs) // 128
        st = _out(n); o = _out(160 * n)
        self.lib.oracle_issue_batch(self.p, _buf(sk), C.c_size_t(n), _buf(reqs), _buf(c), _buf(rng), o, st, nthreads)
        return bytes(st), bytes(o)

    def request_batch(self, pres: bytes, rng: bytes, nthreads: int = 1) -> bytes:
        n = len(pres) // 64
        o = _out(128 * n)
        self.lib.oracle_request_batch(self.p, C.c_size_t(n), _buf(pres), _buf(rng), o, nthreads)
        return bytes(o)
