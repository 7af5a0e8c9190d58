"""Independent big-integer model of the anonymous-credit-tokens sigma-protocol hot path.

TEST INFRASTRUCTURE ONLY (oracle).  Nothing in the product path imports this file; it exists so
that the C oracle (oracle/act_oracle.c) and the HIP engine can be pinned against a second,
independently written implementation, and to generate the committed fixtures in tests/golden/.

Parity status: the reference crate (Rust, curve25519-dalek 4.1.3 / blake3 1.8.2, neither vendored
under /root/reference, no Rust toolchain in the image) cannot be run here and ships no golden
vectors, so byte-parity with the crate itself is "unpinned".  What IS pinned (tests/test_oracle_*.py):
  * BLAKE3 against the upstream C implementation bundled in LLVM (libclang-cpp's llvm_blake3_*),
  * Edwards25519 arithmetic against OpenSSL's Ed25519 public-key derivation,
  * ristretto255 encode/decode/one-way-map against the RFC 9496 vectors recorded in SURVEY.md App. A,
  * the protocol algebra against the reference's accept/reject behaviour.

Every protocol function below cites the reference lines it restates (paths relative to
/root/reference).  All arithmetic is Python ints: slow, obviously-correct, no shared code with
the C oracle or the HIP kernels.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------------------------
# Field GF(2^255-19), scalar field Z_l, curve constants (RFC 9496 section 4.1 / SURVEY.md App. A)
# --------------------------------------------------------------------------------------------
P = 2**255 - 19
ELL = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = 19681161376707505956807079304988542015446066515923890162744021073123829784752
SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
INVSQRT_A_MINUS_D = 54469307008909316920995813868745141605393597292927456921205312896311721017578
ONE_MINUS_D_SQ = 1159843021668779879193775521855586647937357759715417654439879720876111806838
D_MINUS_ONE_SQ = 40440834346308536858101042469323190826248399146238708352240133220865137265952

assert D == 37095705934669439343138083508754565189542113879843219016388785533085940283555
assert SQRT_M1 == pow(2, (P - 1) // 4, P) and SQRT_M1 * SQRT_M1 % P == P - 1
assert SQRT_AD_MINUS_ONE * SQRT_AD_MINUS_ONE % P == (-D - 1) % P
assert INVSQRT_A_MINUS_D * INVSQRT_A_MINUS_D * (-1 - D) % P == 1
assert ONE_MINUS_D_SQ == (1 - D * D) % P and D_MINUS_ONE_SQ == (D - 1) * (D - 1) % P

L_DEFAULT = 128  # src/lib.rs:116


def fe_inv(x: int) -> int:
    return pow(x, P - 2, P)


def fe_is_neg(x: int) -> bool:
    return (x % P) & 1 == 1


def fe_abs(x: int) -> int:
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u: int, v: int) -> Tuple[bool, int]:
    """RFC 9496 section 4.2 SQRT_RATIO_M1."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u) * SQRT_M1 % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    r = fe_abs(r)
    return (correct or flipped), r


# Edwards points in extended coordinates (X, Y, Z, T), a = -1.
Point = Tuple[int, int, int, int]
IDENTITY: Point = (0, 1, 1, 0)
_BY = 4 * pow(5, P - 2, P) % P
_ok, _bx = sqrt_ratio_m1((_BY * _BY - 1) % P, (D * _BY * _BY + 1) % P)
assert _ok
BASEPOINT: Point = (_bx, _BY, 1, _bx * _BY % P)  # x even ("positive"), RFC 8032 section 5.1
assert _BY == 46316835694926478169428394003475163141307993866256225615783033603165251855960


def pt_add(p: Point, q: Point) -> Point:
    x1, y1, z1, t1 = p
    x2, y2, z2, t2 = q
    a = (y1 - x1) * (y2 - x2) % P
    b = (y1 + x1) * (y2 + x2) % P
    c = 2 * D * t1 % P * t2 % P
    d = 2 * z1 * z2 % P
    e, f, g, h = b - a, d - c, d + c, b + a
    return (e * f % P, g * h % P, f * g % P, e * h % P)


def pt_neg(p: Point) -> Point:
    x, y, z, t = p
    return ((-x) % P, y, z, (-t) % P)


def pt_sub(p: Point, q: Point) -> Point:
    return pt_add(p, pt_neg(q))


def pt_double(p: Point) -> Point:
    return pt_add(p, p)


def pt_mul(p: Point, s: int) -> Point:
    """Plain double-and-add; s is reduced mod l first (group order of the ristretto group)."""
    s %= ELL
    acc = IDENTITY
    for bit in bin(s)[2:] if s else "":
        acc = pt_add(acc, acc)
        if bit == "1":
            acc = pt_add(acc, p)
    return acc


def pt_eq(p: Point, q: Point) -> bool:
    """ristretto255 equality, RFC 9496 section 4.3.3."""
    x1, y1, _, _ = p
    x2, y2, _, _ = q
    return (x1 * y2 - y1 * x2) % P == 0 or (y1 * y2 - x1 * x2) % P == 0


def pt_on_curve(p: Point) -> bool:
    x, y, z, t = p
    return (-x * x + y * y - z * z - D * t * t) % P == 0 and (x * y - z * t) % P == 0


def ristretto_encode(p: Point) -> bytes:
    """RFC 9496 section 4.3.2; used at src/transcript.rs:106 and src/cbor.rs:53."""
    x0, y0, z0, t0 = p
    u1 = (z0 + y0) * (z0 - y0) % P
    u2 = x0 * y0 % P
    _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
    den1 = invsqrt * u1 % P
    den2 = invsqrt * u2 % P
    z_inv = den1 * den2 % P * t0 % P
    ix0 = x0 * SQRT_M1 % P
    iy0 = y0 * SQRT_M1 % P
    enchanted = den1 * INVSQRT_A_MINUS_D % P
    if fe_is_neg(t0 * z_inv):
        x, y, den_inv = iy0, ix0, enchanted
    else:
        x, y, den_inv = x0, y0, den2
    if fe_is_neg(x * z_inv):
        y = (-y) % P
    s = fe_abs(den_inv * (z0 - y) % P)
    return s.to_bytes(32, "little")


def ristretto_decode(b: bytes) -> Optional[Point]:
    """RFC 9496 section 4.3.1 (the check done by src/cbor.rs:62-77 on every wire point)."""
    assert len(b) == 32
    s = int.from_bytes(b, "little")
    if s >= P or s & 1:
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2s = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2s) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2s % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = fe_abs(2 * s * den_x % P)
    y = u1 * den_y % P
    t = x * y % P
    if (not was_square) or fe_is_neg(t) or y == 0:
        return None
    return (x, y, 1, t)


def ristretto_map(t: int) -> Point:
    """RFC 9496 section 4.3.4 MAP (Elligator)."""
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = (-fe_abs(s * t % P)) % P
    if not was_square:
        s = s_prime
        c = r
    else:
        c = P - 1
    n = (c * (r - 1) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = n * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return (w0 * w3 % P, w2 * w1 % P, w1 * w3 % P, w0 * w2 % P)


def ristretto_from_uniform_bytes(b: bytes) -> Point:
    """RistrettoPoint::from_uniform_bytes (src/lib.rs:353, :261-263): two MAPs added.
    Each 32-byte half is read little-endian with bit 255 masked, then reduced mod p."""
    assert len(b) == 64
    r0 = (int.from_bytes(b[:32], "little") & ((1 << 255) - 1)) % P
    r1 = (int.from_bytes(b[32:], "little") & ((1 << 255) - 1)) % P
    return pt_add(ristretto_map(r0), ristretto_map(r1))


# Scalars ---------------------------------------------------------------------------------------
def sc_from_wide(b: bytes) -> int:
    """Scalar::from_bytes_mod_order_wide (src/transcript.rs:153; inside every Scalar::random)."""
    assert len(b) == 64
    return int.from_bytes(b, "little") % ELL


def sc_bytes(s: int) -> bytes:
    return (s % ELL).to_bytes(32, "little")


def sc_from_bytes_mod_order(b: bytes) -> int:
    return int.from_bytes(b, "little") % ELL


def sc_inv(s: int) -> int:
    return pow(s, ELL - 2, ELL)


# --------------------------------------------------------------------------------------------
# BLAKE3 (hash mode only; finalize / finalize_xof), written from the BLAKE3 specification.
# --------------------------------------------------------------------------------------------
_IV = (0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19)
_PERM = (2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
_CHUNK_START, _CHUNK_END, _PARENT, _ROOT = 1, 2, 4, 8
_M32 = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & _M32


def _g(v, a, b, c, d, mx, my):
    v[a] = (v[a] + v[b] + mx) & _M32
    v[d] = _rotr(v[d] ^ v[a], 16)
    v[c] = (v[c] + v[d]) & _M32
    v[b] = _rotr(v[b] ^ v[c], 12)
    v[a] = (v[a] + v[b] + my) & _M32
    v[d] = _rotr(v[d] ^ v[a], 8)
    v[c] = (v[c] + v[d]) & _M32
    v[b] = _rotr(v[b] ^ v[c], 7)


def _compress(cv, block_words, counter, block_len, flags):
    v = list(cv) + list(_IV[:4]) + [counter & _M32, (counter >> 32) & _M32, block_len, flags]
    m = list(block_words)
    for r in range(7):
        _g(v, 0, 4, 8, 12, m[0], m[1])
        _g(v, 1, 5, 9, 13, m[2], m[3])
        _g(v, 2, 6, 10, 14, m[4], m[5])
        _g(v, 3, 7, 11, 15, m[6], m[7])
        _g(v, 0, 5, 10, 15, m[8], m[9])
        _g(v, 1, 6, 11, 12, m[10], m[11])
        _g(v, 2, 7, 8, 13, m[12], m[13])
        _g(v, 3, 4, 9, 14, m[14], m[15])
        if r < 6:
            m = [m[i] for i in _PERM]
    for i in range(8):
        v[i] ^= v[i + 8]
        v[i + 8] ^= cv[i]
    return v


def _words(block: bytes):
    block = block + b"\x00" * (64 - len(block))
    return struct.unpack("<16I", block)


def _chunk_output(chunk: bytes, counter: int):
    """Returns (cv, words, counter, block_len, flags) of the chunk's LAST block, un-finalised."""
    cv = list(_IV)
    blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
    for i, blk in enumerate(blocks):
        flags = (_CHUNK_START if i == 0 else 0)
        if i == len(blocks) - 1:
            return (cv, _words(blk), counter, len(blk), flags | _CHUNK_END)
        cv = _compress(cv, _words(blk), counter, 64, flags)[:8]


def _output_cv(out):
    cv, w, ctr, bl, fl = out
    return _compress(cv, w, ctr, bl, fl)[:8]


def _parent_output(left_cv, right_cv):
    return (list(_IV), list(left_cv) + list(right_cv), 0, 64, _PARENT)


def _subtree(data: bytes, first_chunk: int):
    """Un-finalised output node for `data` laid out as a left-heavy BLAKE3 tree."""
    if len(data) <= 1024:
        return _chunk_output(data, first_chunk)
    nchunks = (len(data) + 1023) // 1024
    left_chunks = 1 << ((nchunks - 1).bit_length() - 1)  # largest power of two < nchunks
    left = _subtree(data[: left_chunks * 1024], first_chunk)
    right = _subtree(data[left_chunks * 1024:], first_chunk + left_chunks)
    return _parent_output(_output_cv(left), _output_cv(right))


def blake3(data: bytes, out_len: int = 32) -> bytes:
    cv, w, _ctr, bl, fl = _subtree(bytes(data), 0)
    out = b""
    block = 0
    while len(out) < out_len:
        v = _compress(cv, w, block, bl, fl | _ROOT)
        out += struct.pack("<16I", *v)
        block += 1
    return out[:out_len]


# --------------------------------------------------------------------------------------------
# Deterministic byte-stream RNG standing in for `impl CryptoRngCore`.
# --------------------------------------------------------------------------------------------
class ByteRng:
    def __init__(self, data: bytes):
        self.data = bytes(data)
        self.pos = 0

    def take(self, n: int) -> bytes:
        assert self.pos + n <= len(self.data), "rng stream exhausted"
        out = self.data[self.pos:self.pos + n]
        self.pos += n
        return out

    def scalar(self) -> int:
        """Scalar::random: one 64-byte fill_bytes then wide reduction."""
        return sc_from_wide(self.take(64))


# --------------------------------------------------------------------------------------------
# Protocol: Params, Transcript, request / issue / prove_spend / refund and the client verifiers
# --------------------------------------------------------------------------------------------
PROTOCOL_VERSION = b"curve25519-ristretto anonymous-credits v1.0"  # src/transcript.rs:29

ERR_OK = 0
ERR_INVALID_ISSUANCE_REQUEST_PROOF = 1   # src/lib.rs:103
ERR_INVALID_ISSUANCE_RESPONSE_PROOF = 2  # :104
ERR_DOUBLE_SPEND = 3
ERR_INVALID_REFUND_PROOF = 4             # :106
ERR_INVALID_REFUND_RESPONSE_PROOF = 5
ERR_IDENTITY_POINT = 6                   # :108
ERR_INVALID_CLIENT_SPEND_PROOF = 7       # :109
ERR_AMOUNT_TOO_BIG = 8
ERR_SCALAR_OUT_OF_RANGE = 9
ERR_UNDECODABLE = 255


class ActError(Exception):
    def __init__(self, code: int):
        super().__init__(f"act error {code}")
        self.code = code


@dataclass
class Params:
    h1: Point
    h2: Point
    h3: Point

    @staticmethod
    def new(organization: str, service: str, deployment_id: str, version: str) -> "Params":
        """src/lib.rs:291-315."""
        ds = f"ACT-v1:{organization}:{service}:{deployment_id}:{version}".encode()
        seed = blake3(struct.pack(">Q", len(ds)) + ds, 32)
        return Params(*(Params._hash_to_ristretto(ds, seed, i) for i in range(3)))

    @staticmethod
    def _hash_to_ristretto(ds: bytes, seed: bytes, counter: int) -> Point:
        """src/lib.rs:332-354."""
        msg = struct.pack(">Q", len(ds)) + ds + struct.pack(">Q", len(seed)) + seed
        msg += struct.pack(">Q", 4) + struct.pack("<I", counter)
        return ristretto_from_uniform_bytes(blake3(msg, 64))

    @staticmethod
    def random(rng: ByteRng) -> "Params":
        """src/lib.rs:259-265 (RistrettoPoint::random = 64-byte fill + from_uniform_bytes)."""
        return Params(*(ristretto_from_uniform_bytes(rng.take(64)) for _ in range(3)))

    def encoded(self) -> bytes:
        return ristretto_encode(self.h1) + ristretto_encode(self.h2) + ristretto_encode(self.h3)


def _lp(b: bytes) -> bytes:
    return struct.pack(">Q", len(b)) + b


def transcript_preimage(params: Params, label: bytes, items: Sequence[bytes]) -> bytes:
    """Byte string hashed by Transcript (src/transcript.rs:54-74, :95-98); items are the 32-byte
    encodings of scalars (as_bytes) or points (compress) in the order they are added."""
    out = _lp(PROTOCOL_VERSION)
    for h in (params.h1, params.h2, params.h3):
        out += _lp(ristretto_encode(h))
    out += _lp(label)
    for it in items:
        out += _lp(it)
    return out


def transcript_challenge(params: Params, label: bytes, items: Sequence[bytes]) -> int:
    """Transcript::with(...).challenge(): 64 XOF bytes -> scalar mod l (src/transcript.rs:149-154)."""
    return sc_from_wide(blake3(transcript_preimage(params, label, items), 64))


E = ristretto_encode
G = BASEPOINT


# --------------------------------------------------------------------------------------------
# Orders and polarities.  The arithmetic of this model is pinned by third-party code (libsodium, LLVM BLAKE3:
# tests/golden/sodium_*.json); what that cannot see is a misreading of /root/reference/src/lib.rs that every restatement
# shares -- the order of the rng draws, the order of the transcript elements, which argument of a conditional_select is taken,
# the key order of the records.  So those are TABLES here, the functions below are driven by them (they draw, hash, select and
# serialise through the helpers that read the tables), and tests/test_source_pins.py extracts the same tables from the reference's
# source text and compares: a swap in either place fails.
# --------------------------------------------------------------------------------------------
# function -> names of its `Scalar::random(&mut rng)` bindings in source order; "x[]" = L draws in a loop / collect
DRAW_ORDER = {
    "request": ("k_prime", "r_prime"),                                                    # src/lib.rs:468-469
    "issue": ("e", "alpha"),                                                              # :643, :649
    "refund": ("e", "alpha"),                                                             # :846, :852
    "prove_spend": ("r1", "r2", "c_prime", "r_prime", "e_prime", "r2_prime", "r3_prime",  # :978-984
                    "k_star", "s_i[]", "k0_prime", "s_i_prime[]", "gamma_i[]", "w0", "z[]",   # :998-1023
                    "k_prime", "s_prime"),                                                # :1057-1058
}
# (function, label) -> what its Transcript::with closure adds, in order; "x[]" = every element of an array, "x[][]" = of an
# array of pairs.  Names are the source's with `self.` / `request.` / `response.` / `spend_proof.` / `refund.` taken off.
TRANSCRIPT_ORDER = {
    ("request", "request"): ("big_k", "k1"),                                              # :473-475
    ("issue", "request"): ("big_k", "k1"),                                                # :633-635
    ("issue", "respond"): ("c", "e", "a", "x_a", "x_g", "y_a", "y_g"),                    # :654-657
    ("to_credit_token", "respond"): ("c", "e", "a", "x_a", "x_g", "y_a", "y_g"),          # :544-547
    ("prove_spend", "spend"): ("k", "a_prime", "b_bar", "a1", "a2", "com[]", "big_c_prime[][]", "c_"),        # :1061-1070
    ("refund", "spend"): ("k", "a_prime", "b_bar", "a1", "a2", "com[]", "big_c_prime[][]", "big_c"),          # :831-840
    ("refund", "refund"): ("e", "a", "x_a", "x_g", "y_a", "y_g"),                         # :856-859
    ("to_credit_token", "refund"): ("e", "a", "x_a", "x_g", "y_a", "y_g"),                # :1236-1239
}
# prove_spend's conditional_select(a, b, i[j].ct_eq(&Scalar::ZERO)) calls (:1025-1118): target -> (a, b).  subtle's
# conditional_select returns b when the choice is true, i.e. when the bit is 0, and a when the bit is 1.
#   sim[b]  = the simulated commitment of branch b: (h2 w0 +) h3 z[j] - big_c[j][b] gamma_i[j]
#   real    = the real commitment: (h2 k0' +) h3 s_i'[j]
#   resp0/1 = gamma00[j] s_i[j] + s_i'[j]  /  (gamma - gamma00[j]) s_i[j] + s_i'[j];  resp_k0/1 = the same with k* and k0'
SELECTS = {
    "big_c_prime[.][0]": ("sim[0]", "real"),                                              # :1025-1029, :1041-1045
    "big_c_prime[.][1]": ("real", "sim[1]"),                                              # :1031-1035, :1046-1050
    "gamma00[.]": ("gamma_i", "gamma-gamma_i"),                                           # :1078-1082, :1105-1109
    "w00": ("w0", "resp_k0"),                                                             # :1083-1087
    "w01": ("resp_k1", "w0"),                                                             # :1088-1092
    "z00[.][0]": ("z", "resp0"),                                                          # :1094-1098, :1110-1114
    "z00[.][1]": ("resp1", "z"),                                                          # :1099-1103, :1115-1119
}
# record = the struct's fields as consecutive 32-byte strings in the key order of its CBOR map (src/cbor.rs to_cbor)
RECORD_ORDER = {
    "IssuanceRequest": ("big_k", "gamma", "k_bar", "r_bar"),                              # src/cbor.rs:105-110
    "IssuanceResponse": ("a", "e", "gamma", "z", "c"),                                    # :163-169
    "SpendProof": ("k", "s", "a_prime", "b_bar", "com", "gamma", "e_bar", "r2_bar", "r3_bar", "c_bar", "r_bar", "w00", "w01",
                   "gamma0", "z", "k_bar", "s_bar"),                                      # :250-268
    "Refund": ("a", "e", "gamma", "z"),                                                   # :422-427
    "PrivateKey": ("x", "w"),                                                             # :477-480
    "PreIssuance": ("r", "k"),                                                            # :546-549
    "CreditToken": ("a", "e", "k", "r", "c"),                                             # :596-602
    "PreRefund": ("r", "k", "m"),                                                         # :656-660
}


def _draw(rng: "ByteRng", fn: str, nbits: int = 0) -> dict:
    """The function's Scalar::random draws, in DRAW_ORDER, as name -> scalar (or list of L scalars)."""
    out = {}
    for name in DRAW_ORDER[fn]:
        if name.endswith("[]"):
            out[name[:-2]] = [rng.scalar() for _ in range(nbits)]
        else:
            out[name] = rng.scalar()
    return out


def _enc32(v) -> bytes:
    """Scalar -> as_bytes, point -> compress, array (of arrays) -> its elements in order."""
    if isinstance(v, int):
        return sc_bytes(v)
    if isinstance(v, tuple) and len(v) == 4 and all(isinstance(c, int) for c in v):
        return E(v)
    return b"".join(_enc32(x) for x in v)


def _challenge(params: "Params", fn: str, label: str, env: dict) -> int:
    """Transcript::with(params, label, |t| ...) of function `fn`: the elements TRANSCRIPT_ORDER names, taken from `env`."""
    items = []
    for name in TRANSCRIPT_ORDER[(fn, label)]:
        blob = _enc32(env[name.rstrip("[]")])
        items += [blob[i:i + 32] for i in range(0, len(blob), 32)]
    return transcript_challenge(params, label.encode(), items)


def _select(target: str, bit: int, env: dict):
    """conditional_select(a, b, bit.ct_eq(0)) for the (a, b) SELECTS records: b when the bit is 0, a when it is 1.
    env values may be thunks (only the chosen one is evaluated: this is a model, not constant-time code)."""
    a, b = SELECTS[target]
    v = env[b] if bit == 0 else env[a]
    return v() if callable(v) else v


def _record(obj, type_name: str) -> bytes:
    return b"".join(_enc32(getattr(obj, name)) for name in RECORD_ORDER[type_name])


@dataclass
class PrivateKey:
    x: int
    w: Point

    @staticmethod
    def random(rng: ByteRng) -> "PrivateKey":
        x = rng.scalar()                      # src/lib.rs:189
        return PrivateKey(x, pt_mul(G, x))    # :191

    def record(self) -> bytes:                # CBOR order x, w (src/cbor.rs:477-480)
        return _record(self, "PrivateKey")


@dataclass
class PreIssuance:
    r: int
    k: int

    @staticmethod
    def random(rng: ByteRng) -> "PreIssuance":
        r = rng.scalar()                      # src/lib.rs:434
        k = rng.scalar()                      # :435
        return PreIssuance(r, k)

    def record(self) -> bytes:                # r, k (src/cbor.rs:546-549)
        return _record(self, "PreIssuance")


@dataclass
class IssuanceRequest:
    big_k: Point
    gamma: int
    k_bar: int
    r_bar: int

    def record(self) -> bytes:                # K, gamma, k_bar, r_bar (src/cbor.rs:105-110)
        return _record(self, "IssuanceRequest")


@dataclass
class IssuanceResponse:
    a: Point
    e: int
    gamma: int
    z: int
    c: int

    def record(self) -> bytes:                # A, e, gamma, z, c (src/cbor.rs:163-169)
        return _record(self, "IssuanceResponse")


@dataclass
class CreditToken:
    a: Point
    e: int
    k: int
    r: int
    c: int

    def record(self) -> bytes:                # a, e, k, r, c (src/cbor.rs:596-602)
        return _record(self, "CreditToken")


@dataclass
class SpendProof:
    k: int
    s: int
    a_prime: Point
    b_bar: Point
    com: List[Point]
    gamma: int
    e_bar: int
    r2_bar: int
    r3_bar: int
    c_bar: int
    r_bar: int
    w00: int
    w01: int
    gamma0: List[int]
    z: List[Tuple[int, int]]
    k_bar: int
    s_bar: int

    def record(self) -> bytes:
        """k,s,A',B_bar,Com[L],gamma,e_bar,r2_bar,r3_bar,c_bar,r_bar,w00,w01,gamma0[L],z[L][2],
        k_bar,s_bar (src/cbor.rs:250-268): 32*(14+4L) bytes."""
        return _record(self, "SpendProof")


@dataclass
class PreRefund:
    r: int
    k: int
    m: int

    def record(self) -> bytes:                # r, k, m (src/cbor.rs:656-660)
        return _record(self, "PreRefund")


@dataclass
class Refund:
    a: Point
    e: int
    gamma: int
    z: int

    def record(self) -> bytes:                # A*, e, gamma, z (src/cbor.rs:422-427)
        return _record(self, "Refund")


def request(pre: PreIssuance, params: Params, rng: ByteRng) -> IssuanceRequest:
    """PreIssuance::request, src/lib.rs:463-487."""
    big_k = pt_add(pt_mul(params.h2, pre.k), pt_mul(params.h3, pre.r))           # :465
    d = _draw(rng, "request")                                                     # :468-469
    k_prime, r_prime = d["k_prime"], d["r_prime"]
    k1 = pt_add(pt_mul(params.h2, k_prime), pt_mul(params.h3, r_prime))           # :470
    gamma = _challenge(params, "request", "request", {"big_k": big_k, "k1": k1})  # :473-475
    k_bar = (k_prime + pre.k * gamma) % ELL                                       # :478
    r_bar = (r_prime + pre.r * gamma) % ELL                                       # :479
    return IssuanceRequest(big_k, gamma, k_bar, r_bar)


def issue(sk: PrivateKey, params: Params, req: IssuanceRequest, c: int, rng: ByteRng) -> IssuanceResponse:
    """PrivateKey::issue, src/lib.rs:621-663.  RNG is drawn only after the PoK verifies."""
    k1 = pt_sub(pt_add(pt_mul(params.h2, req.k_bar), pt_mul(params.h3, req.r_bar)),
                pt_mul(req.big_k, req.gamma))                                     # :629-630
    gamma = _challenge(params, "issue", "request", {"big_k": req.big_k, "k1": k1})   # :633-635
    if gamma != req.gamma:                                                        # :638
        raise ActError(ERR_INVALID_ISSUANCE_REQUEST_PROOF)
    # e before alpha (:643, :649): both are drawn before either is used, so one _draw is the source's order
    d = _draw(rng, "issue")
    e, alpha = d["e"], d["alpha"]
    x_a = pt_add(pt_add(G, pt_mul(params.h1, c)), req.big_k)                      # :644
    a = pt_mul(x_a, sc_inv((e + sk.x) % ELL))                                     # :645
    x_g = pt_add(pt_mul(G, e), sk.w)                                              # :646
    y_a = pt_mul(a, alpha)                                                        # :650
    y_g = pt_mul(G, alpha)                                                        # :651
    gamma = _challenge(params, "issue", "respond", {"c": c % ELL, "e": e, "a": a, "x_a": x_a, "x_g": x_g, "y_a": y_a, "y_g": y_g})  # :654-657
    z = (gamma * (sk.x + e) + alpha) % ELL                                        # :660
    return IssuanceResponse(a, e, gamma, z, c % ELL)


def issuance_to_credit_token(pre: PreIssuance, params: Params, w: Point, req: IssuanceRequest,
                             resp: IssuanceResponse) -> CreditToken:
    """PreIssuance::to_credit_token, src/lib.rs:528-562."""
    x_a = pt_add(pt_add(G, pt_mul(params.h1, resp.c)), req.big_k)                 # :536
    x_g = pt_add(pt_mul(G, resp.e), w)                                            # :537
    ng = (-resp.gamma) % ELL
    y_a = pt_add(pt_mul(resp.a, resp.z), pt_mul(x_a, ng))                         # :540
    y_g = pt_add(pt_mul(G, resp.z), pt_mul(x_g, ng))                              # :541
    gamma = _challenge(params, "to_credit_token", "respond",
                       {"c": resp.c, "e": resp.e, "a": resp.a, "x_a": x_a, "x_g": x_g, "y_a": y_a, "y_g": y_g})   # :544-547
    if gamma != resp.gamma:                                                       # :550
        raise ActError(ERR_INVALID_ISSUANCE_RESPONSE_PROOF)
    return CreditToken(resp.a, resp.e, pre.k, pre.r, resp.c)


def bits_of(s: int, nbits: int) -> List[int]:
    """src/lib.rs:902-915: low `L` bits of the canonical little-endian encoding."""
    b = sc_bytes(s)
    return [(b[i // 8] >> (i % 8)) & 1 for i in range(nbits)]


def prove_spend(tok: CreditToken, params: Params, s: int, rng: ByteRng, nbits: int = L_DEFAULT
                ) -> Tuple[SpendProof, PreRefund]:
    """CreditToken::prove_spend, src/lib.rs:972-1152.  Every draw happens before any later draw's value is needed and nothing
    else consumes the generator, so drawing all of DRAW_ORDER["prove_spend"] up front is the source's order."""
    h1, h2, h3 = params.h1, params.h2, params.h3
    d = _draw(rng, "prove_spend", nbits)
    r1, r2, c_prime, r_prime = d["r1"], d["r2"], d["c_prime"], d["r_prime"]                 # :978-981
    e_prime, r2_prime, r3_prime = d["e_prime"], d["r2_prime"], d["r3_prime"]                # :982-984
    k_star, s_i, k0_prime, s_i_prime = d["k_star"], d["s_i"], d["k0_prime"], d["s_i_prime"]  # :998-1014
    gamma_i, w0, z, k_prime, s_prime = d["gamma_i"], d["w0"], d["z"], d["k_prime"], d["s_prime"]   # :1016-1023, :1057-1058
    b = pt_add(pt_add(pt_add(G, pt_mul(h1, tok.c)), pt_mul(h2, tok.k)), pt_mul(h3, tok.r))  # :986-989
    a_prime = pt_mul(tok.a, r1 * r2 % ELL)                                                  # :990
    b_bar = pt_mul(b, r1)                                                                   # :991
    r3 = sc_inv(r1)                                                                         # :992
    a1 = pt_add(pt_mul(a_prime, e_prime), pt_mul(b_bar, r2_prime))                          # :993
    a2 = pt_add(pt_add(pt_mul(b_bar, r3_prime), pt_mul(h1, c_prime)), pt_mul(h3, r_prime))  # :994
    i = bits_of((tok.c - s) % ELL, nbits)                                                   # :996
    com = [None] * nbits
    com[0] = pt_add(pt_add(pt_mul(h1, i[0]), pt_mul(h2, k_star)), pt_mul(h3, s_i[0]))       # :1001
    for j in range(1, nbits):
        com[j] = pt_add(pt_mul(h1, i[j]), pt_mul(h3, s_i[j]))                               # :1003
    big_c_prime = [[None, None] for _ in range(nbits)]
    for j in range(nbits):
        big_c = (com[j], pt_sub(com[j], h1))                                                # :1007-1008, :1038-1039
        if j == 0:                                                                          # the h2 terms of bit 0 (:1025-1035)
            real = lambda: pt_add(pt_mul(h2, k0_prime), pt_mul(h3, s_i_prime[0]))
            sim = lambda base: pt_sub(pt_add(pt_mul(h2, w0), pt_mul(h3, z[0])), pt_mul(base, gamma_i[0]))
        else:                                                                               # :1041-1050
            real = lambda j=j: pt_mul(h3, s_i_prime[j])
            sim = lambda base, j=j: pt_sub(pt_mul(h3, z[j]), pt_mul(base, gamma_i[j]))
        env = {"real": real, "sim[0]": lambda: sim(big_c[0]), "sim[1]": lambda: sim(big_c[1])}
        big_c_prime[j][0] = _select("big_c_prime[.][0]", i[j], env)
        big_c_prime[j][1] = _select("big_c_prime[.][1]", i[j], env)
    r_star = sum(si << idx for idx, si in enumerate(s_i)) % ELL                             # :1052-1056
    c_ = pt_add(pt_add(pt_mul(h1, (-c_prime) % ELL), pt_mul(h2, k_prime)), pt_mul(h3, s_prime))  # :1059
    gamma = _challenge(params, "prove_spend", "spend", {"k": tok.k, "a_prime": a_prime, "b_bar": b_bar, "a1": a1, "a2": a2, "com": com,
                                                        "big_c_prime": big_c_prime, "c_": c_})   # :1061-1070
    ng = (-gamma) % ELL
    e_bar = (ng * tok.e + e_prime) % ELL                                                    # :1072
    r2_bar = (gamma * r2 + r2_prime) % ELL                                                  # :1073
    r3_bar = (gamma * r3 + r3_prime) % ELL                                                  # :1074
    c_bar = (ng * tok.c + c_prime) % ELL                                                    # :1075
    r_bar = (ng * tok.r + r_prime) % ELL                                                    # :1076
    gamma00 = [0] * nbits
    zz = [(0, 0)] * nbits
    for j in range(nbits):
        gamma00[j] = _select("gamma00[.]", i[j], {"gamma_i": gamma_i[j], "gamma-gamma_i": (gamma - gamma_i[j]) % ELL})   # :1078-1082, :1105-1109
        env = {"z": z[j], "resp0": (gamma00[j] * s_i[j] + s_i_prime[j]) % ELL, "resp1": ((gamma - gamma00[j]) * s_i[j] + s_i_prime[j]) % ELL}
        zz[j] = (_select("z00[.][0]", i[j], env), _select("z00[.][1]", i[j], env))          # :1094-1103, :1110-1119
    env = {"w0": w0, "resp_k0": (gamma00[0] * k_star + k0_prime) % ELL, "resp_k1": ((gamma - gamma00[0]) * k_star + k0_prime) % ELL}
    w00 = _select("w00", i[0], env)                                                         # :1083-1087
    w01 = _select("w01", i[0], env)                                                         # :1088-1092
    k_bar = (gamma * k_star + k_prime) % ELL                                                # :1121
    s_bar = (gamma * r_star + s_prime) % ELL                                                # :1122
    proof = SpendProof(tok.k, s % ELL, a_prime, b_bar, com, gamma, e_bar, r2_bar, r3_bar, c_bar, r_bar,
                       w00, w01, gamma00, zz, k_bar, s_bar)
    return proof, PreRefund(r_star, k_star, (tok.c - s) % ELL)                              # :1124-1128


def spend_verify_challenge(sk_x: int, params: Params, pr: SpendProof) -> Tuple[int, Point]:
    """src/lib.rs:791-840: recompute every commitment and the challenge; returns (gamma', K')."""
    h1, h2, h3 = params.h1, params.h2, params.h3
    nbits = len(pr.com)
    ng = (-pr.gamma) % ELL
    a_bar = pt_mul(pr.a_prime, sk_x)                                                        # :791
    big_h1 = pt_add(G, pt_mul(h2, pr.k))                                                    # :792
    a1 = pt_add(pt_add(pt_mul(pr.a_prime, pr.e_bar), pt_mul(pr.b_bar, pr.r2_bar)), pt_mul(a_bar, ng))   # :793-795
    a2 = pt_add(pt_add(pt_add(pt_mul(pr.b_bar, pr.r3_bar), pt_mul(h1, pr.c_bar)), pt_mul(h3, pr.r_bar)),
                pt_mul(big_h1, ng))                                                         # :796-799
    cps = []
    for j in range(nbits):
        g0 = pr.gamma0[j]
        g1 = (pr.gamma - g0) % ELL                                                          # :801, :811
        cj0 = pr.com[j]
        cj1 = pt_sub(pr.com[j], h1)                                                         # :804, :813
        p0 = pt_sub(pt_mul(h3, pr.z[j][0]), pt_mul(cj0, g0))
        p1 = pt_sub(pt_mul(h3, pr.z[j][1]), pt_mul(cj1, g1))
        if j == 0:
            p0 = pt_add(pt_mul(h2, pr.w00), p0)                                             # :806-807
            p1 = pt_add(pt_mul(h2, pr.w01), p1)                                             # :808-809
        cps += [p0, p1]
    k_prime = IDENTITY
    for idx, c in enumerate(pr.com):                                                        # :819-824
        k_prime = pt_add(k_prime, pt_mul(c, 1 << idx))
    com_ = pt_add(pt_mul(h1, pr.s), k_prime)                                                # :825
    big_c = pt_sub(pt_add(pt_add(pt_mul(h1, (-pr.c_bar) % ELL), pt_mul(h2, pr.k_bar)), pt_mul(h3, pr.s_bar)),
                   pt_mul(com_, pr.gamma))                                                  # :826-829
    big_c_prime = [[cps[2 * j], cps[2 * j + 1]] for j in range(nbits)]
    gamma = _challenge(params, "refund", "spend", {"k": pr.k, "a_prime": pr.a_prime, "b_bar": pr.b_bar, "a1": a1, "a2": a2, "com": pr.com,
                                                   "big_c_prime": big_c_prime, "big_c": big_c})   # :831-840
    return gamma, k_prime


def refund(sk: PrivateKey, params: Params, pr: SpendProof, rng: ByteRng) -> Refund:
    """PrivateKey::refund, src/lib.rs:781-869.  RNG is drawn only after the proof verifies."""
    if pt_eq(pr.a_prime, IDENTITY):                                                         # :787
        raise ActError(ERR_IDENTITY_POINT)
    gamma, k_prime = spend_verify_challenge(sk.x, params, pr)
    if gamma != pr.gamma:                                                                   # :842
        raise ActError(ERR_INVALID_CLIENT_SPEND_PROOF)
    d = _draw(rng, "refund")                                                                # e (:846), alpha (:852)
    e, alpha = d["e"], d["alpha"]
    x_a = pt_add(G, k_prime)                                                                # :848
    a = pt_mul(x_a, sc_inv((e + sk.x) % ELL))                                               # :849
    x_g = pt_add(pt_mul(G, e), sk.w)                                                        # :851
    y_a = pt_mul(a, alpha)                                                                  # :853
    y_g = pt_mul(G, alpha)                                                                  # :854
    rg = _challenge(params, "refund", "refund", {"e": e, "a": a, "x_a": x_a, "x_g": x_g, "y_a": y_a, "y_g": y_g})   # :856-859
    z = (rg * (sk.x + e) + alpha) % ELL                                                     # :861
    return Refund(a, e, rg, z)


def refund_to_credit_token(pre: PreRefund, params: Params, pr: SpendProof, rf: Refund, w: Point) -> CreditToken:
    """PreRefund::to_credit_token, src/lib.rs:1217-1253."""
    k_prime = IDENTITY
    for idx, c in enumerate(pr.com):
        k_prime = pt_add(k_prime, pt_mul(c, 1 << idx))
    x_a = pt_add(G, k_prime)                                                                # :1224-1230
    x_g = pt_add(pt_mul(G, rf.e), w)                                                        # :1232
    ng = (-rf.gamma) % ELL
    y_a = pt_add(pt_mul(rf.a, rf.z), pt_mul(x_a, ng))                                       # :1233
    y_g = pt_add(pt_mul(G, rf.z), pt_mul(x_g, ng))                                          # :1234
    gamma = _challenge(params, "to_credit_token", "refund", {"e": rf.e, "a": rf.a, "x_a": x_a, "x_g": x_g, "y_a": y_a, "y_g": y_g})   # :1236-1239
    if gamma != rf.gamma:                                                                   # :1241
        raise ActError(ERR_INVALID_REFUND_PROOF)
    return CreditToken(rf.a, rf.e, pre.k, pre.r, pre.m)


# Record parsers (raw 32-byte-field layouts) -------------------------------------
def _pt(b: bytes) -> Point:
    p = ristretto_decode(b)
    if p is None:
        raise ActError(ERR_UNDECODABLE)
    return p


def _sc(b: bytes) -> int:
    # a Rust `Scalar` is always canonical: decode_scalar reduces with from_bytes_mod_order (src/cbor.rs:85)
    return int.from_bytes(b, "little") % ELL


def parse_spend_proof(rec: bytes, nbits: int = L_DEFAULT) -> SpendProof:
    assert len(rec) == 32 * (14 + 4 * nbits)
    f = [rec[i:i + 32] for i in range(0, len(rec), 32)]
    o = 4
    com = [_pt(f[o + j]) for j in range(nbits)]
    o += nbits
    sc8 = [_sc(f[o + i]) for i in range(8)]
    o += 8
    gamma0 = [_sc(f[o + j]) for j in range(nbits)]
    o += nbits
    z = [(_sc(f[o + 2 * j]), _sc(f[o + 2 * j + 1])) for j in range(nbits)]
    o += 2 * nbits
    return SpendProof(_sc(f[0]), _sc(f[1]), _pt(f[2]), _pt(f[3]), com, *sc8[:6], sc8[6], sc8[7], gamma0, z,
                      _sc(f[o]), _sc(f[o + 1]))


# --------------------------------------------------------------------------------------------
# CBOR wire codec (restates /root/reference/src/cbor.rs; ciborium 0.2.2 does the RFC 8949 parsing there)
# --------------------------------------------------------------------------------------------
# type -> list of (key, kind 'S'/'P', shape 0 single / 1 array[L] / 2 array[L] of pairs); PublicKey is a bare bstr
CBOR_TYPES = {
    "IssuanceRequest": [(1, "P", 0), (2, "S", 0), (3, "S", 0), (4, "S", 0)],                       # src/cbor.rs:105-110
    "IssuanceResponse": [(1, "P", 0), (2, "S", 0), (3, "S", 0), (4, "S", 0), (5, "S", 0)],         # :163-169
    "SpendProof": [(1, "S", 0), (2, "S", 0), (3, "P", 0), (4, "P", 0), (5, "P", 1), (6, "S", 0), (7, "S", 0), (8, "S", 0), (9, "S", 0),
                   (10, "S", 0), (11, "S", 0), (12, "S", 0), (13, "S", 0), (14, "S", 1), (15, "S", 2), (16, "S", 0), (17, "S", 0)],  # :250-268
    "Refund": [(1, "P", 0), (2, "S", 0), (3, "S", 0), (4, "S", 0)],                                 # :422-427
    "PrivateKey": [(1, "S", 0), (2, "P", 0)],                                                       # :477-480
    "PublicKey": None,                                                                              # :522-526
    "PreIssuance": [(1, "S", 0), (2, "S", 0)],                                                      # :546-549
    "CreditToken": [(1, "P", 0), (2, "S", 0), (3, "S", 0), (4, "S", 0), (5, "S", 0)],               # :596-602
    "PreRefund": [(1, "S", 0), (2, "S", 0), (3, "S", 0)],                                           # :656-660
}
CBOR_OK, CBOR_ERR_PARSE, CBOR_ERR_STRUCTURE, CBOR_ERR_VALUE = 0, 1, 2, 3


def _cbor_head(major: int, n: int) -> bytes:
    if n < 24:
        return bytes([major << 5 | n])
    if n < 256:
        return bytes([major << 5 | 24, n])
    return bytes([major << 5 | 25, n >> 8, n & 255])


def cbor_encode(type_name: str, record: bytes, nbits: int = L_DEFAULT) -> bytes:
    """to_cbor: deterministic encoding of a raw record."""
    f = [record[i:i + 32] for i in range(0, len(record), 32)]
    bstr = lambda b: _cbor_head(2, 32) + b
    spec = CBOR_TYPES[type_name]
    if spec is None:
        return bstr(f[0])
    out = _cbor_head(5, len(spec))
    i = 0
    for key, _kind, shape in spec:
        out += _cbor_head(0, key)
        if shape == 0:
            out += bstr(f[i]); i += 1
        elif shape == 1:
            out += _cbor_head(4, nbits) + b"".join(bstr(f[i + j]) for j in range(nbits)); i += nbits
        else:
            out += _cbor_head(4, nbits) + b"".join(_cbor_head(4, 2) + bstr(f[i + 2 * j]) + bstr(f[i + 2 * j + 1]) for j in range(nbits)); i += 2 * nbits
    assert i == len(f)
    return out


class _CborParseError(Exception):
    pass


def _cbor_parse(b: bytes, pos: int, depth: int = 0):
    """One RFC 8949 data item -> (value, new_pos).  Values: ('int', n) | ('bytes', b) | ('text', b) | ('array', [..]) |
    ('map', [(k, v)..]) | ('tag', n, v) | ('other',)."""
    if depth > 256 or pos >= len(b):
        raise _CborParseError
    ib = b[pos]; pos += 1
    major, ai = ib >> 5, ib & 31
    ind = False
    if ai < 24:
        val = ai
    elif ai == 31:
        if major in (0, 1, 6):
            raise _CborParseError
        ind, val = True, 0
    elif ai <= 27:
        ln = 1 << (ai - 24)
        if pos + ln > len(b):
            raise _CborParseError
        val = int.from_bytes(b[pos:pos + ln], "big"); pos += ln
    else:
        raise _CborParseError
    if major == 0:
        return ("int", val), pos
    if major == 1:
        return ("int", -1 - val), pos
    if major in (2, 3):
        if not ind:
            if pos + val > len(b):
                raise _CborParseError
            if major == 3:
                try:                                    # ciborium hands text strings to str::from_utf8: not UTF-8 = parse error
                    b[pos:pos + val].decode("utf-8")
                except UnicodeDecodeError:
                    raise _CborParseError
            return (("bytes" if major == 2 else "text"), b[pos:pos + val]), pos + val
        acc = b""
        while True:
            if pos >= len(b):
                raise _CborParseError
            if b[pos] == 0xFF:
                return (("bytes" if major == 2 else "text"), acc), pos + 1
            # a chunk is a DEFINITE-length string of the same major type (RFC 8949 3.2.3); anything else is not well-formed
            if b[pos] >> 5 != major or (b[pos] & 31) == 31:
                raise _CborParseError
            v, pos = _cbor_parse(b, pos, depth + 1)
            acc += v[1]
    if major in (4, 5):
        items = []
        per = 2 if major == 5 else 1
        if ind:
            while True:
                if pos >= len(b):
                    raise _CborParseError
                if b[pos] == 0xFF:
                    pos += 1
                    break
                group = []
                for _ in range(per):
                    v, pos = _cbor_parse(b, pos, depth + 1); group.append(v)
                items.append(tuple(group) if per == 2 else group[0])
        else:
            if val > len(b) - pos:
                raise _CborParseError
            for _ in range(val):
                group = []
                for _ in range(per):
                    v, pos = _cbor_parse(b, pos, depth + 1); group.append(v)
                items.append(tuple(group) if per == 2 else group[0])
        return (("map" if major == 5 else "array"), items), pos
    if major == 6:
        v, pos = _cbor_parse(b, pos, depth + 1)
        return ("tag", val, v), pos
    if ind:
        raise _CborParseError     # stray break
    return ("other",), pos


def cbor_decode(type_name: str, msg: bytes, nbits: int = L_DEFAULT):
    """from_cbor -> (status, record).  Status 1 malformed CBOR, 2 InvalidStructure, 3 InvalidValue; record zero on failure.
    A message that is broken in more than one way reports what the reference reports: the whole item is parsed first
    (ciborium::from_reader), then the map's entries are visited in WIRE order and the first failing decode_scalar /
    decode_point returns (src/cbor.rs:276-388: `?` inside the `for (key, val) in map` loop; array elements are decoded in
    order BEFORE the length check, :307-319), and missing fields are reported last (:390-407)."""
    spec = CBOR_TYPES[type_name]
    nf = 1 if spec is None else sum(1 if s == 0 else nbits if s == 1 else 2 * nbits for _, _, s in spec)
    fail = lambda code: (code, bytes(32 * nf))
    try:
        value, _ = _cbor_parse(msg, 0)            # trailing bytes are not read
    except _CborParseError:
        return fail(CBOR_ERR_PARSE)

    def b32(v):
        return v[1] if v[0] == "bytes" and len(v[1]) == 32 else None

    fields = [None] * nf
    kinds = []
    if spec is None:
        kinds = ["P"]
        x = b32(value)
        if x is None:
            return fail(CBOR_ERR_STRUCTURE)
        if ristretto_decode(x) is None:
            return fail(CBOR_ERR_VALUE)
        fields[0] = x
    else:
        if value[0] != "map":
            return fail(CBOR_ERR_STRUCTURE)
        first, i = {}, 0
        for key, kind, shape in spec:
            first[key] = (i, kind, shape)
            cnt = 1 if shape == 0 else nbits if shape == 1 else 2 * nbits
            kinds += [kind] * cnt; i += cnt
        present = set()
        for k, v in value[1]:
            if k[0] != "int" or k[1] not in first:
                continue
            i, kind, shape = first[k[1]]
            if shape == 0:
                x = b32(v)
                if x is None:
                    return fail(CBOR_ERR_STRUCTURE)
                if kind == "P" and ristretto_decode(x) is None:      # decode_point, src/cbor.rs:62-77: returns at once
                    return fail(CBOR_ERR_VALUE)
                fields[i] = x; present.add(k[1])
            else:
                if v[0] != "array":
                    continue                       # silently not set (src/cbor.rs:306, :331, :347)
                elems = []
                for el in v[1]:
                    if shape == 1:
                        x = b32(el)
                        if x is None:
                            return fail(CBOR_ERR_STRUCTURE)
                        if kind == "P" and ristretto_decode(x) is None:      # every element, also those beyond L, before the length check
                            return fail(CBOR_ERR_VALUE)
                        elems.append(x)
                    else:
                        if el[0] != "array" or len(el[1]) != 2:
                            return fail(CBOR_ERR_STRUCTURE)
                        x0, x1 = b32(el[1][0]), b32(el[1][1])
                        if x0 is None or x1 is None:
                            return fail(CBOR_ERR_STRUCTURE)
                        elems += [x0, x1]
                if len(v[1]) != nbits:
                    return fail(CBOR_ERR_STRUCTURE)
                fields[i:i + len(elems)] = elems; present.add(k[1])
        if len(present) != len(spec):
            return fail(CBOR_ERR_STRUCTURE)
    out = b""
    for kind, x in zip(kinds, fields):
        if kind == "S":
            out += sc_bytes(sc_from_bytes_mod_order(x))      # decode_scalar, src/cbor.rs:80-91
        else:
            out += x                                         # validated where it was read
    return CBOR_OK, out
