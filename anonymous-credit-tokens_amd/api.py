"""Host-side mirror of the reference crate's public interface (/root/reference/src/lib.rs) over the
C ABI: same type names, method names, argument meaning and error behaviour, with `*_batch`
siblings that take lists.  Every method is one (batched) call into the HIP engine; nothing here
computes curve or scalar arithmetic.  Values are held as the raw records of include/act_mi355x.h.

    params = Params.new("org", "svc", "prod", "2024-01-15")                 # src/lib.rs:291
    sk = PrivateKey.random(OsRng())                                            # :188
    pre = PreIssuance.random(rng); req = pre.request(params, rng)              # :432, :463
    resp = sk.issue(params, req, 20, rng)                                      # :621  (raises Error)
    tok = pre.to_credit_token(params, sk.public(), req, resp)                  # :528
    proof, prerefund = tok.prove_spend(params, 5, rng)                         # :972
    refund = sk.refund(params, proof, rng)                                     # :781  (raises Error)
    tok2 = prerefund.to_credit_token(params, proof, refund, sk.public())       # :1217
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

from . import capi

L = 128  # src/lib.rs:116

_ELL = 2**252 + 27742317777372353535851937790883648493


class Error(Exception):
    """Mirror of `enum Error` (src/lib.rs:102-112); `code` = 1 + discriminant, 255 = undecodable point."""
    NAMES = {1: "InvalidIssuanceRequestProof", 2: "InvalidIssuanceResponseProof", 3: "DoubleSpendError",
             4: "InvalidRefundProof", 5: "InvalidRefundResponseProof", 6: "IdentityPointError",
             7: "InvalidClientSpendProof", 8: "AmountTooBigError", 9: "ScalarOutOfRangeError", 255: "UndecodablePoint"}

    def __init__(self, code: int):
        super().__init__(self.NAMES.get(code, f"status {code}"))
        self.code = code
        self.name = self.NAMES.get(code, str(code))


class CborError(Exception):
    """Mirror of `enum CborError` (src/cbor.rs:30-37): Ciborium (malformed CBOR), InvalidStructure, InvalidValue."""
    NAMES = {1: "Ciborium", 2: "InvalidStructure", 3: "InvalidValue"}

    def __init__(self, code: int):
        super().__init__(self.NAMES.get(code, f"status {code}"))
        self.code = code
        self.name = self.NAMES.get(code, str(code))


class _Cbor:
    """to_cbor / from_cbor of the crate's wire and state types (src/cbor.rs:94-695) through the batch codec of the engine:
    deterministic RFC 8949 encoding with integer keys; from_cbor accepts what ciborium accepts, reduces scalars mod l and rejects
    points that are not canonical Ristretto encodings."""
    CBOR_TYPE = None

    def to_cbor(self, params: "Params" = None) -> bytes:
        nbits = getattr(self, "nbits", L)
        return _codec_engine(params, nbits).cbor_encode(self.CBOR_TYPE, self.record)[0]

    @classmethod
    def from_cbor(cls, data: bytes, params: "Params" = None, nbits: int = L):
        st, rec = _codec_engine(params, nbits).cbor_decode(cls.CBOR_TYPE, [bytes(data)])
        if st[0]:
            raise CborError(st[0])
        return cls(rec, nbits) if cls.CBOR_TYPE == "SpendProof" else cls(rec)


def _codec_engine(params, nbits):
    # the codec does not depend on the Params; any context of the right range width will do
    return (params or Params.new("act", "keygen", "default", "1970-01-01")).engine(nbits)


class OsRng:
    """CryptoRngCore stand-in backed by os.urandom."""

    def fill_bytes(self, n: int) -> bytes:
        return os.urandom(n)


class ByteStreamRng:
    """Deterministic CryptoRngCore stand-in replaying a byte string (tests; parity runs)."""

    def __init__(self, data: bytes):
        self.data, self.pos = bytes(data), 0

    def fill_bytes(self, n: int) -> bytes:
        if self.pos + n > len(self.data):
            raise ValueError("rng stream exhausted")
        out = self.data[self.pos:self.pos + n]
        self.pos += n
        return out

    def peek(self, n: int) -> bytes:
        return self.data[self.pos:self.pos + n].ljust(n, b"\0")

    def advance(self, n: int):
        self.pos += n


def _wire_error(status: int):
    """lane status of a wire-level call -> the error the crate's two calls would have produced"""
    return CborError({254: 1, 253: 2, 255: 3}[status]) if status in (253, 254, 255) else Error(status)


def _draw_signed(rng, n_lanes: int, run):
    """The wire-level calls take the generator itself (ACT_RNG_CALLBACK): the library draws 128 bytes per lane it SIGNS, once, after
    every verdict is known.  `rng.fill_bytes(k)` is called with exactly that many bytes."""
    import ctypes as C

    cb = capi.rng_trampoline(rng.fill_bytes)          # an exception in fill_bytes, or a short read, fails the call: nothing is signed
    src = capi.RngSource(cb, None)
    holder = type("Src", (), {"ptr": C.addressof(src), "keep": (cb, src)})()
    return run(_CallbackRng(holder), capi.RNG_CALLBACK)


class _CallbackRng(capi.ReplayRng):
    """adapter: a capi.ReplayRng-shaped object (what capi._rng_arg accepts) around an arbitrary act_rng_source"""

    def __init__(self, holder):
        self._holder = holder

    @property
    def ptr(self):
        return self._holder.ptr


def scalar(v) -> bytes:
    """Scalar::from(u128) / canonical 32-byte little-endian scalar."""
    if isinstance(v, (bytes, bytearray)):
        assert len(v) == 32
        return bytes(v)
    return (int(v) % _ELL).to_bytes(32, "little")


def scalar_to_u128(s: bytes):
    """src/lib.rs:146-153: Some(value) iff the high 16 bytes are zero."""
    return int.from_bytes(s[:16], "little") if not any(s[16:]) else None


def _check_then_sign(rng, check, sign):
    """issue / refund draw e, alpha only AFTER the checks have passed (src/lib.rs:638-643, 842-846): a rejected item leaves the caller's
    generator untouched.  So: `check()` -> (statuses, whatever the signature needs); exactly 128 bytes per ACCEPTED lane are drawn, in
    lane order, with ONE fill_bytes call; `sign(state, statuses, bytes)` -> (statuses, records).  Two library calls, the crate's contract
    for ANY generator (round 5 drew 128 bytes per lane up front unless the generator could peek)."""
    st, state = check()
    accepted = sum(1 for s in st if s == 0)
    drawn = rng.fill_bytes(128 * accepted) if accepted else b""
    if len(drawn) != 128 * accepted:
        raise ValueError("generator returned %d bytes, %d asked" % (len(drawn), 128 * accepted))
    return sign(state, st, drawn + b"\0")          # (+ one byte: never an empty buffer at the C boundary)


class Params:
    _engines: dict = {}

    def __init__(self, h: bytes, device: int = 0):
        self.h, self.device = bytes(h), device

    @staticmethod
    def new(organization: str, service: str, deployment_id: str, version: str, device: int = 0) -> "Params":
        return Params(capi.params_new(organization, service, deployment_id, version, device), device)

    @staticmethod
    def random(rng, device: int = 0) -> "Params":
        return Params(capi.params_random(rng.fill_bytes(192), device), device)

    def engine(self, nbits: int = L) -> capi.Engine:
        key = (self.h, nbits, self.device)
        if key not in Params._engines:
            mode = capi.TRANSCRIPT_DEVICE if os.environ.get("ACT_TRANSCRIPT", "host") == "device" else capi.TRANSCRIPT_HOST
            Params._engines[key] = capi.Engine(self.h, nbits, self.device, transcript=mode)
        return Params._engines[key]

    def __eq__(self, other):
        # the reference compares self.h3 with other.h2 (src/lib.rs:233); mirrored, quirk included
        return self.h[0:32] == other.h[0:32] and self.h[32:64] == other.h[32:64] and self.h[64:96] == other.h[32:64]

    def __ne__(self, other):
        return not self.__eq__(other)


class PublicKey(_Cbor):
    CBOR_TYPE = "PublicKey"

    def __init__(self, w: bytes):
        self.w = bytes(w)
        self.record = self.w


class PrivateKey(_Cbor):
    CBOR_TYPE = "PrivateKey"

    def __init__(self, record: bytes):
        assert len(record) == 64
        self.record = bytes(record)

    @staticmethod
    def random(rng, params: Params = None) -> "PrivateKey":
        params = params or Params.new("act", "keygen", "default", "1970-01-01")
        return PrivateKey(params.engine().private_key_random(rng.fill_bytes(64)))

    def public(self) -> PublicKey:
        return PublicKey(self.record[32:])

    def issue(self, params: Params, request: "IssuanceRequest", c, rng) -> "IssuanceResponse":
        return self.issue_batch(params, [request], [c], rng)[0]

    def issue_batch(self, params: Params, requests: Sequence["IssuanceRequest"], cs: Sequence, rng) -> List["IssuanceResponse"]:
        e = params.engine()
        req = b"".join(r.record for r in requests); cc = b"".join(scalar(c) for c in cs)
        st, out = _check_then_sign(rng, lambda: (e.issue_check(req), None),
                                   lambda _s, st, rb: e.issue_sign(self.record, req, cc, st, rb, capi.RNG_SEQUENTIAL))
        res = [IssuanceResponse(out[160 * i:160 * i + 160]) if st[i] == 0 else Error(st[i]) for i in range(len(requests))]
        if len(res) == 1 and isinstance(res[0], Error):
            raise res[0]
        return res

    def refund(self, params: Params, spend_proof: "SpendProof", rng) -> "Refund":
        return self.refund_batch(params, [spend_proof], rng)[0]

    def refund_batch(self, params: Params, proofs: Sequence["SpendProof"], rng) -> List["Refund"]:
        nbits = proofs[0].nbits if proofs else L
        e = params.engine(nbits)
        pb = b"".join(p.record for p in proofs)
        st, out = _check_then_sign(rng, lambda: e.verify_spend(self.record, pb, True),
                                   lambda kp, st, rb: e.refund_sign(self.record, kp, st, rb, capi.RNG_SEQUENTIAL))
        res = [Refund(out[128 * i:128 * i + 128]) if st[i] == 0 else Error(st[i]) for i in range(len(proofs))]
        if len(res) == 1 and isinstance(res[0], Error):
            raise res[0]
        return res

    def redeem_batch(self, params: Params, db: "NullifierDb", proofs: Sequence["SpendProof"], rng) -> List["Refund"]:
        """The server loop of examples/act.rs:62-73 as one call: verify, look the nullifier up, record it, sign.  Entry i is a
        Refund, or Error (DoubleSpendError for a nullifier already recorded / spent by an earlier accepted proof of the batch)."""
        nbits = proofs[0].nbits if proofs else L
        e = params.engine(nbits)
        pb = b"".join(p.record for p in proofs)
        st, out = _draw_signed(rng, len(proofs), lambda src, mode: e.redeem(db.set, self.record, pb, src, mode))      # the generator itself: drawn once, after the nullifier step, for the lanes that are signed
        return [Refund(out[128 * i:128 * i + 128]) if st[i] == 0 else Error(st[i]) for i in range(len(proofs))]

    # ---- wire bytes in, wire bytes out (rust/src/mi355x.rs refund_cbor_batch / redeem_cbor_batch) ----------
    def refund_cbor_batch(self, params: Params, msgs: Sequence[bytes], rng, nbits: int = L) -> list:
        """`[SpendProof::from_cbor(m).and_then(|p| self.refund(params, &p, rng)).map(to_cbor) for m in msgs]` as one call: entry i is
        the CBOR Refund message, a CborError (from_cbor's) or an Error (refund's); `rng` is drawn for the accepted messages only, in
        message order, after all verdicts (src/lib.rs:842-852)."""
        e = params.engine(nbits)
        st, out = _draw_signed(rng, len(msgs), lambda src, mode: e.refund_cbor(self.record, list(msgs), src, mode))
        return [out[i] if st[i] == 0 else _wire_error(st[i]) for i in range(len(msgs))]

    def redeem_cbor_batch(self, params: Params, db: "NullifierDb", msgs: Sequence[bytes], rng, nbits: int = L) -> list:
        """The server loop of examples/act.rs:62-73 on wire bytes: from_cbor, refund's checks, the nullifier store (DoubleSpendError),
        the signature, to_cbor."""
        e = params.engine(nbits)
        st, out = _draw_signed(rng, len(msgs), lambda src, mode: e.redeem_cbor(db.set, self.record, list(msgs), src, mode))
        return [out[i] if st[i] == 0 else _wire_error(st[i]) for i in range(len(msgs))]

    def verify_spend_batch(self, params: Params, proofs: Sequence["SpendProof"]) -> bytes:
        nbits = proofs[0].nbits if proofs else L
        return params.engine(nbits).verify_spend(self.record, b"".join(p.record for p in proofs))


class PreIssuance(_Cbor):
    CBOR_TYPE = "PreIssuance"

    def __init__(self, record: bytes):
        assert len(record) == 64
        self.record = bytes(record)   # r | k

    @staticmethod
    def random(rng, params: Params = None) -> "PreIssuance":
        params = params or Params.new("act", "keygen", "default", "1970-01-01")
        return PreIssuance(params.engine().pre_issuance_random(rng.fill_bytes(128)))

    def request(self, params: Params, rng) -> "IssuanceRequest":
        return IssuanceRequest(params.engine().request(self.record, rng.fill_bytes(128)))

    def to_credit_token(self, params: Params, public: PublicKey, request: "IssuanceRequest", response: "IssuanceResponse") -> "CreditToken":
        st, out = params.engine().issuance_to_credit_token(self.record, public.w, request.record, response.record)
        if st[0]:
            raise Error(st[0])
        return CreditToken(out)


class IssuanceRequest(_Cbor):
    CBOR_TYPE = "IssuanceRequest"

    def __init__(self, record: bytes):
        assert len(record) == 128
        self.record = bytes(record)   # K | gamma | k_bar | r_bar


class IssuanceResponse(_Cbor):
    CBOR_TYPE = "IssuanceResponse"

    def __init__(self, record: bytes):
        assert len(record) == 160
        self.record = bytes(record)   # A | e | gamma | z | c


class CreditToken(_Cbor):
    CBOR_TYPE = "CreditToken"

    def __init__(self, record: bytes):
        assert len(record) == 160
        self.record = bytes(record)   # a | e | k | r | c

    def nullifier(self) -> bytes:
        return self.record[64:96]

    def credits(self) -> bytes:
        return self.record[128:160]

    def prove_spend(self, params: Params, s, rng, nbits: int = L) -> Tuple["SpendProof", "PreRefund"]:
        e = params.engine(nbits)
        st, proof, pre = e.prove_spend(self.record, scalar(s), rng.fill_bytes(e.prove_rng_bytes))
        if st[0]:
            raise Error(st[0])
        return SpendProof(proof, nbits), PreRefund(pre)

    def __eq__(self, other):
        return self.record == other.record


class SpendProof(_Cbor):
    CBOR_TYPE = "SpendProof"

    def __init__(self, record: bytes, nbits: int = L):
        assert len(record) == 32 * (14 + 4 * nbits)
        self.record, self.nbits = bytes(record), nbits

    def nullifier(self) -> bytes:
        return self.record[0:32]

    def charge(self) -> bytes:
        return self.record[32:64]


class PreRefund(_Cbor):
    CBOR_TYPE = "PreRefund"

    def __init__(self, record: bytes):
        assert len(record) == 96
        self.record = bytes(record)   # r | k | m

    def to_credit_token(self, params: Params, spend_proof: SpendProof, refund: "Refund", public_key: PublicKey) -> CreditToken:
        st, out = params.engine(spend_proof.nbits).refund_to_credit_token(self.record, spend_proof.record, refund.record, public_key.w)
        if st[0]:
            raise Error(st[0])
        return CreditToken(out)


class Refund(_Cbor):
    CBOR_TYPE = "Refund"

    def __init__(self, record: bytes):
        assert len(record) == 128
        self.record = bytes(record)   # A* | e | gamma | z

    def __eq__(self, other):
        return self.record == other.record


class NullifierDb:
    """The double-spend database the crate leaves to the caller (src/lib.rs:741-745; `NullifierDb` of src/tests.rs:29-50), as a
    hash set in the GPU's memory.  The tests' `if is_spent(k) { reject } else { record_spent(k) }` pair is one atomic call here,
    `spend(k)` (there is no read-only query: a look-up that does not record is how double spends slip through concurrent
    callers); whole batches go through PrivateKey.redeem_batch.  Keys are compared as scalars (k and k + l are one nullifier)."""

    def __init__(self, capacity: int = 1 << 20, device: int = 0):
        self.set = capi.NullifierSet(capacity, device)

    def spend(self, nullifier: bytes) -> bool:
        """record `nullifier`; True if it was fresh, False if it had been spent before (DoubleSpendError)"""
        return self.set.check_and_insert(bytes(nullifier))[0] == 0

    def spend_batch(self, nullifiers: Sequence[bytes]) -> List[bool]:
        """the same for a list, with the meaning of the loop in list order (a repeat inside the list is a double spend)"""
        return [b == 0 for b in self.set.check_and_insert(b"".join(bytes(k) for k in nullifiers))] if nullifiers else []

    def __len__(self):
        return len(self.set)
