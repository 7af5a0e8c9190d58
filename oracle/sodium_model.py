"""Third implementation of the anonymous-credit-tokens hot path whose ARITHMETIC IS NOT OURS.

TEST INFRASTRUCTURE ONLY (oracle; authoring container only).  Every group / scalar operation below is a call
into libsodium 1.0.18 (`/opt/conda/lib/libsodium.so.23`: crypto_core_ristretto255_*, crypto_scalarmult_ristretto255*)
and every hash is the upstream BLAKE3 C implementation bundled in LLVM (`libclang-cpp.so: llvm_blake3_hasher_*`).
The only code of this repository's authors in here is the protocol glue, written to follow the reference line by
line (paths relative to /root/reference):

    Params::new / hash_to_ristretto     src/lib.rs:291-354
    PrivateKey::random                  src/lib.rs:188-194
    PreIssuance::random / request       src/lib.rs:432-437, 463-487
    PreIssuance::to_credit_token        src/lib.rs:528-562
    PrivateKey::issue                   src/lib.rs:621-663
    PrivateKey::refund                  src/lib.rs:781-869
    bits_of                             src/lib.rs:902-915
    CreditToken::prove_spend            src/lib.rs:972-1152
    PreRefund::to_credit_token          src/lib.rs:1217-1253
    Transcript                          src/transcript.rs:54-154
    decode_point / decode_scalar        src/cbor.rs:59-91

It exists to pin `oracle/pymodel.py`, `oracle/act_oracle.c` and the HIP engine against arithmetic none of them
shares: `tests/golden/make_sodium_golden.py` runs it and commits `tests/golden/sodium_*.json`; neither libsodium nor
this module travels to the GPU box (only the JSON does).  What even this cannot give is a diff against the Rust
crate itself (no Rust toolchain in the image): DESIGN.md section 5.

Points are 32-byte Ristretto encodings, scalars 32-byte little-endian canonical strings, exactly what libsodium's API
takes; nothing here touches a coordinate or a limb.
"""
import ctypes
import os

_SODIUM_PATHS = ["/opt/conda/lib/libsodium.so.23", "/opt/conda/lib/libsodium.so"]
_LLVM_PATH = "/opt/rocm/lib/llvm/lib/libclang-cpp.so"

_sodium = None
_llvm = None


def available() -> bool:
    return any(os.path.exists(p) for p in _SODIUM_PATHS) and os.path.exists(_LLVM_PATH)


def _libs():
    global _sodium, _llvm
    if _sodium is None:
        path = next(p for p in _SODIUM_PATHS if os.path.exists(p))
        _sodium = ctypes.CDLL(path)
        if _sodium.sodium_init() < 0:
            raise RuntimeError("sodium_init failed")
        _llvm = ctypes.CDLL(_LLVM_PATH)
    return _sodium, _llvm


def sodium_version() -> str:
    s, _ = _libs()
    s.sodium_version_string.restype = ctypes.c_char_p
    return s.sodium_version_string().decode()


# ---- BLAKE3 (LLVM's copy of the upstream C implementation) ------------------------------------------------------
class Blake3:
    def __init__(self):
        _, self._l = _libs()
        self._st = ctypes.create_string_buffer(4096)
        self._l.llvm_blake3_hasher_init(self._st)

    def update(self, data: bytes):
        self._l.llvm_blake3_hasher_update(self._st, data, ctypes.c_size_t(len(data)))
        return self

    def finalize(self, n: int = 32) -> bytes:
        out = ctypes.create_string_buffer(n)
        self._l.llvm_blake3_hasher_finalize(self._st, out, ctypes.c_size_t(n))
        return out.raw


# ---- libsodium wrappers --------------------------------------------------------------------------------------------
IDENTITY = bytes(32)
ZERO = bytes(32)


def _buf():
    return ctypes.create_string_buffer(32)


def is_valid_point(p: bytes) -> bool:
    """CompressedRistretto::decompress().is_some() (src/cbor.rs:68-71)."""
    s, _ = _libs()
    if len(p) != 32 or p[31] & 0x80:
        # libsodium 1.0.18's ristretto255_is_canonical masks bit 255 away before comparing with p (changed upstream in
        # 1.0.19); RFC 9496 4.3.1 step 1 and dalek's decompress() (encoding must round-trip byte for byte) reject it.
        # This one bit test is the only validity logic in this file that is not libsodium's.
        return False
    return s.crypto_core_ristretto255_is_valid_point(p) == 1


def is_valid_point_libsodium_raw(p: bytes) -> bool:
    s, _ = _libs()
    return s.crypto_core_ristretto255_is_valid_point(p) == 1


def from_uniform(b64: bytes) -> bytes:
    """RistrettoPoint::from_uniform_bytes (src/lib.rs:353; RistrettoPoint::random :261-263)."""
    s, _ = _libs()
    assert len(b64) == 64
    out = _buf()
    assert s.crypto_core_ristretto255_from_hash(out, b64) == 0
    return out.raw


def padd(p: bytes, q: bytes) -> bytes:
    s, _ = _libs()
    out = _buf()
    if s.crypto_core_ristretto255_add(out, p, q) != 0:
        raise ValueError("invalid point")
    return out.raw


def psub(p: bytes, q: bytes) -> bytes:
    s, _ = _libs()
    out = _buf()
    if s.crypto_core_ristretto255_sub(out, p, q) != 0:
        raise ValueError("invalid point")
    return out.raw


def pmul(p: bytes, n: bytes) -> bytes:
    """point * scalar.  crypto_scalarmult_ristretto255 returns -1 both for an invalid point and for an identity
    result (it still writes the 32 zero bytes the identity encodes to), so validity is checked first."""
    s, _ = _libs()
    if not is_valid_point(p):
        raise ValueError("invalid point")
    assert len(n) == 32 and n[31] < 0x80          # the call clears bit 255 of the scalar; canonical scalars never set it
    out = _buf()
    rc = s.crypto_scalarmult_ristretto255(out, n, p)
    if rc != 0:
        assert out.raw == IDENTITY
    return out.raw


def bmul(n: bytes) -> bytes:
    """RistrettoPoint::generator() * scalar."""
    s, _ = _libs()
    assert len(n) == 32 and n[31] < 0x80
    out = _buf()
    rc = s.crypto_scalarmult_ristretto255_base(out, n)
    if rc != 0:
        assert out.raw == IDENTITY
    return out.raw


def sc_reduce_wide(b64: bytes) -> bytes:
    """Scalar::from_bytes_mod_order_wide (src/transcript.rs:153; inside every Scalar::random)."""
    s, _ = _libs()
    assert len(b64) == 64
    out = _buf()
    s.crypto_core_ristretto255_scalar_reduce(out, b64)
    return out.raw


def sc_reduce32(b32: bytes) -> bytes:
    """Scalar::from_bytes_mod_order (src/cbor.rs:85)."""
    return sc_reduce_wide(b32 + bytes(32))


def _sc2(name):
    def f(a: bytes, b: bytes) -> bytes:
        s, _ = _libs()
        out = _buf()
        getattr(s, "crypto_core_ristretto255_scalar_" + name)(out, a, b)
        return out.raw
    return f


sc_add = _sc2("add")
sc_sub = _sc2("sub")
sc_mul = _sc2("mul")


def sc_neg(a: bytes) -> bytes:
    s, _ = _libs()
    out = _buf()
    s.crypto_core_ristretto255_scalar_negate(out, a)
    return out.raw


def sc_invert(a: bytes) -> bytes:
    """Scalar::invert; dalek maps 0 to 0, libsodium reports -1 and writes zeros."""
    s, _ = _libs()
    out = _buf()
    s.crypto_core_ristretto255_scalar_invert(out, a)
    return out.raw


def sc_from_int(v: int) -> bytes:
    """Scalar::from(u128) (src/lib.rs:823, 911, 1055): little-endian embedding of a value < 2^128."""
    assert 0 <= v < 2**128
    return v.to_bytes(32, "little")


GENERATOR = None


def generator() -> bytes:
    global GENERATOR
    if GENERATOR is None:
        GENERATOR = bmul((1).to_bytes(32, "little"))
    return GENERATOR


class ByteRng:
    """The bytes a CryptoRngCore would have produced; Scalar::random = one 64-byte fill + wide reduction."""

    def __init__(self, data: bytes):
        self.data, self.pos = data, 0

    def fill(self, n: int) -> bytes:
        if self.pos + n > len(self.data):
            raise ValueError("rng exhausted")
        out = self.data[self.pos:self.pos + n]
        self.pos += n
        return out

    def scalar(self) -> bytes:
        return sc_reduce_wide(self.fill(64))


class ActError(Exception):
    def __init__(self, code):
        super().__init__(code)
        self.code = code


# 1 + discriminant of `Error`, src/lib.rs:102-112
E_INVALID_ISSUANCE_REQUEST_PROOF, E_INVALID_ISSUANCE_RESPONSE_PROOF, E_INVALID_REFUND_PROOF = 1, 2, 4
E_IDENTITY_POINT, E_INVALID_CLIENT_SPEND_PROOF = 6, 7


# ---- Params (src/lib.rs:291-354) --------------------------------------------------------------------------------
def _be64(n: int) -> bytes:
    return n.to_bytes(8, "big")


def hash_to_ristretto(ds: bytes, seed: bytes, counter: int) -> bytes:        # :332-354
    h = Blake3()
    h.update(_be64(len(ds))).update(ds)
    h.update(_be64(len(seed))).update(seed)
    h.update(_be64(4)).update(counter.to_bytes(4, "little"))
    return from_uniform(h.finalize(64))


def params_new(organization: str, service: str, deployment_id: str, version: str):       # :291-315
    ds = ("ACT-v1:%s:%s:%s:%s" % (organization, service, deployment_id, version)).encode()
    seed = Blake3().update(_be64(len(ds))).update(ds).finalize(32)
    return tuple(hash_to_ristretto(ds, seed, i) for i in range(3))


def params_random(rng: ByteRng):                                              # :259-265
    return tuple(from_uniform(rng.fill(64)) for _ in range(3))


# ---- Transcript (src/transcript.rs) ---------------------------------------------------------------------------------
PROTOCOL_VERSION = b"curve25519-ristretto anonymous-credits v1.0"             # :29


class Transcript:
    def __init__(self, params, label: bytes):                                 # :54-74
        self.h = Blake3()
        self.pre = bytearray()
        self._raw(_be64(len(PROTOCOL_VERSION)))
        self._raw(PROTOCOL_VERSION)
        for hp in params:
            self.add_element(hp)
        self._raw(_be64(len(label)))
        self._raw(label)

    def _raw(self, b: bytes):
        self.h.update(bytes(b))
        self.pre += b

    def update(self, b: bytes):                                               # :95-98
        self._raw(_be64(len(b)))
        self._raw(b)

    def add_element(self, p: bytes):                                          # :105-107 (compress() of a point == its encoding)
        assert len(p) == 32
        self.update(p)

    def add_scalar(self, s: bytes):                                           # :125-128
        assert len(s) == 32
        self.update(s)

    def challenge(self) -> bytes:                                             # :149-154
        return sc_reduce_wide(self.h.finalize(64))


# ---- keys / issuance --------------------------------------------------------------------------------------------
def private_key_random(rng: ByteRng):                                         # src/lib.rs:188-194
    x = rng.scalar()
    return x, bmul(x)


def pre_issuance_random(rng: ByteRng):                                        # :432-437 -> (r, k)
    r = rng.scalar()
    k = rng.scalar()
    return r, k


def request(pre, params, rng: ByteRng):                                       # :463-487
    r, k = pre
    h1, h2, h3 = params
    big_k = padd(pmul(h2, k), pmul(h3, r))
    k_prime = rng.scalar()
    r_prime = rng.scalar()
    k1 = padd(pmul(h2, k_prime), pmul(h3, r_prime))
    t = Transcript(params, b"request")
    t.add_element(big_k)
    t.add_element(k1)
    gamma = t.challenge()
    k_bar = sc_add(k_prime, sc_mul(k, gamma))
    r_bar = sc_add(r_prime, sc_mul(r, gamma))
    return big_k + gamma + k_bar + r_bar                                      # record K | gamma | k_bar | r_bar (src/cbor.rs:105-110)


def _split(rec: bytes, n: int):
    assert len(rec) == 32 * n
    return [rec[32 * i:32 * i + 32] for i in range(n)]


def issue(sk, params, req: bytes, c: bytes, rng: ByteRng) -> bytes:           # :621-663
    x, w = sk
    h1, h2, h3 = params
    big_k, rq_gamma, k_bar, r_bar = _split(req, 4)
    k1 = psub(padd(pmul(h2, k_bar), pmul(h3, r_bar)), pmul(big_k, rq_gamma))
    t = Transcript(params, b"request")
    t.add_element(big_k)
    t.add_element(k1)
    if t.challenge() != rq_gamma:
        raise ActError(E_INVALID_ISSUANCE_REQUEST_PROOF)
    e = rng.scalar()
    x_a = padd(padd(generator(), pmul(h1, c)), big_k)
    a = pmul(x_a, sc_invert(sc_add(e, x)))
    x_g = padd(bmul(e), w)
    alpha = rng.scalar()
    y_a = pmul(a, alpha)
    y_g = bmul(alpha)
    t = Transcript(params, b"respond")
    t.add_scalar(c)
    t.add_scalar(e)
    for p in (a, x_a, x_g, y_a, y_g):
        t.add_element(p)
    gamma = t.challenge()
    z = sc_add(sc_mul(gamma, sc_add(x, e)), alpha)
    return a + e + gamma + z + c                                              # A | e | gamma | z | c (src/cbor.rs:163-169)


def issuance_to_credit_token(pre, params, w: bytes, req: bytes, resp: bytes) -> bytes:    # :528-562
    r, k = pre
    h1, h2, h3 = params
    big_k = req[:32]
    a, e, gamma_r, z, c = _split(resp, 5)
    x_a = padd(padd(generator(), pmul(h1, c)), big_k)
    x_g = padd(bmul(e), w)
    ng = sc_neg(gamma_r)
    y_a = padd(pmul(a, z), pmul(x_a, ng))
    y_g = padd(bmul(z), pmul(x_g, ng))
    t = Transcript(params, b"respond")
    t.add_scalar(c)
    t.add_scalar(e)
    for p in (a, x_a, x_g, y_a, y_g):
        t.add_element(p)
    if t.challenge() != gamma_r:
        raise ActError(E_INVALID_ISSUANCE_RESPONSE_PROOF)
    return a + e + k + r + c                                                  # a | e | k | r | c (src/cbor.rs:596-602)


# ---- spend --------------------------------------------------------------------------------------------------------
def bits_of(s: bytes, L: int):                                                # :902-915
    return [(s[i // 8] >> (i % 8)) & 1 for i in range(L)]


class ProofLayout:
    """SpendProof record: k | s | A' | B_bar | Com[L] | gamma | e_bar | r2_bar | r3_bar | c_bar | r_bar | w00 | w01 |
    gamma0[L] | z[L][2] | k_bar | s_bar (src/cbor.rs:250-268)."""

    def __init__(self, L):
        self.L = L
        self.fields = 14 + 4 * L

    def parse(self, rec: bytes):
        L = self.L
        f = _split(rec, self.fields)
        d = {"k": f[0], "s": f[1], "a_prime": f[2], "b_bar": f[3], "com": f[4:4 + L]}
        o = 4 + L
        for i, name in enumerate(["gamma", "e_bar", "r2_bar", "r3_bar", "c_bar", "r_bar", "w00", "w01"]):
            d[name] = f[o + i]
        o += 8
        d["gamma0"] = f[o:o + L]
        o += L
        d["z"] = [(f[o + 2 * j], f[o + 2 * j + 1]) for j in range(L)]
        o += 2 * L
        d["k_bar"], d["s_bar"] = f[o], f[o + 1]
        return d


def decode_spend_proof(rec: bytes, L: int):
    """What from_cbor does to the fields (src/cbor.rs:59-91): scalars reduced mod l, points must decompress."""
    d = ProofLayout(L).parse(rec)
    for name in ("a_prime", "b_bar"):
        if not is_valid_point(d[name]):
            return None
    if not all(is_valid_point(c) for c in d["com"]):
        return None
    for name in ("k", "s", "gamma", "e_bar", "r2_bar", "r3_bar", "c_bar", "r_bar", "w00", "w01", "k_bar", "s_bar"):
        d[name] = sc_reduce32(d[name])
    d["gamma0"] = [sc_reduce32(v) for v in d["gamma0"]]
    d["z"] = [(sc_reduce32(a), sc_reduce32(b)) for a, b in d["z"]]
    return d


def prove_spend(tok: bytes, params, s: bytes, rng: ByteRng, L: int):          # :972-1152
    a, e, k, r, c = _split(tok, 5)
    h1, h2, h3 = params
    r1, r2, c_prime, r_prime, e_prime, r2_prime, r3_prime = (rng.scalar() for _ in range(7))
    b = padd(padd(padd(generator(), pmul(h1, c)), pmul(h2, k)), pmul(h3, r))
    a_prime = pmul(a, sc_mul(r1, r2))
    b_bar = pmul(b, r1)
    r3 = sc_invert(r1)
    a1 = padd(pmul(a_prime, e_prime), pmul(b_bar, r2_prime))
    a2 = padd(padd(pmul(b_bar, r3_prime), pmul(h1, c_prime)), pmul(h3, r_prime))
    i = bits_of(sc_sub(c, s), L)
    isc = [sc_from_int(v) for v in i]
    k_star = rng.scalar()
    s_i = [rng.scalar() for _ in range(L)]
    com = [None] * L
    com[0] = padd(padd(pmul(h1, isc[0]), pmul(h2, k_star)), pmul(h3, s_i[0]))
    for j in range(1, L):
        com[j] = padd(pmul(h1, isc[j]), pmul(h3, s_i[j]))
    big_c = [(com[j], psub(com[j], h1)) for j in range(L)]
    k0_prime = rng.scalar()
    s_i_prime = [rng.scalar() for _ in range(L)]
    gamma_i = [rng.scalar() for _ in range(L)]
    w0 = rng.scalar()
    z = [rng.scalar() for _ in range(L)]
    sel = lambda if_false, if_true, cond: if_true if cond else if_false     # conditional_select(a, b, choice) = choice ? b : a
    cp = [None] * L
    zero0 = i[0] == 0
    real0 = padd(pmul(h2, k0_prime), pmul(h3, s_i_prime[0]))
    cp[0] = (sel(psub(padd(pmul(h2, w0), pmul(h3, z[0])), pmul(big_c[0][0], gamma_i[0])), real0, zero0),
             sel(real0, psub(padd(pmul(h2, w0), pmul(h3, z[0])), pmul(big_c[0][1], gamma_i[0])), zero0))
    for j in range(1, L):
        zj = i[j] == 0
        real = pmul(h3, s_i_prime[j])
        cp[j] = (sel(psub(pmul(h3, z[j]), pmul(big_c[j][0], gamma_i[j])), real, zj),
                 sel(real, psub(pmul(h3, z[j]), pmul(big_c[j][1], gamma_i[j])), zj))
    r_star = ZERO
    for j in range(L):
        r_star = sc_add(r_star, sc_mul(s_i[j], sc_from_int(1 << j)))
    k_prime = rng.scalar()
    s_prime = rng.scalar()
    c_ = padd(padd(pmul(h1, sc_neg(c_prime)), pmul(h2, k_prime)), pmul(h3, s_prime))
    t = Transcript(params, b"spend")
    t.add_scalar(k)
    for p in (a_prime, b_bar, a1, a2):
        t.add_element(p)
    for p in com:
        t.add_element(p)
    for pair in cp:
        t.add_element(pair[0])
        t.add_element(pair[1])
    t.add_element(c_)
    gamma = t.challenge()
    ng = sc_neg(gamma)
    e_bar = sc_add(sc_mul(ng, e), e_prime)
    r2_bar = sc_add(sc_mul(gamma, r2), r2_prime)
    r3_bar = sc_add(sc_mul(gamma, r3), r3_prime)
    c_bar = sc_add(sc_mul(ng, c), c_prime)
    r_bar = sc_add(sc_mul(ng, r), r_prime)
    gamma00 = [sel(gamma_i[j], sc_sub(gamma, gamma_i[j]), i[j] == 0) for j in range(L)]
    w00 = sel(w0, sc_add(sc_mul(gamma00[0], k_star), k0_prime), zero0)
    w01 = sel(sc_add(sc_mul(sc_sub(gamma, gamma00[0]), k_star), k0_prime), w0, zero0)
    z00 = []
    for j in range(L):
        zj = i[j] == 0
        z00.append((sel(z[j], sc_add(sc_mul(gamma00[j], s_i[j]), s_i_prime[j]), zj),
                    sel(sc_add(sc_mul(sc_sub(gamma, gamma00[j]), s_i[j]), s_i_prime[j]), z[j], zj)))
    k_bar = sc_add(sc_mul(gamma, k_star), k_prime)
    s_bar = sc_add(sc_mul(gamma, r_star), s_prime)
    proof = (k + s + a_prime + b_bar + b"".join(com) + gamma + e_bar + r2_bar + r3_bar + c_bar + r_bar + w00 + w01 +
             b"".join(gamma00) + b"".join(p[0] + p[1] for p in z00) + k_bar + s_bar)
    prerefund = r_star + k_star + sc_sub(c, s)                                 # r | k | m (src/cbor.rs:656-660)
    return proof, prerefund, bytes(t.pre)


def _k_prime(com, L):                                                         # :819-824
    acc = IDENTITY
    for j in range(L):
        acc = padd(acc, pmul(com[j], sc_from_int(1 << j)))
    return acc


def spend_challenge(x: bytes, params, d, L: int):
    """src/lib.rs:791-840 -> (gamma', K', transcript pre-image)."""
    h1, h2, h3 = params
    a_bar = pmul(d["a_prime"], x)
    big_h1 = padd(generator(), pmul(h2, d["k"]))
    ng = sc_neg(d["gamma"])
    a1 = padd(padd(pmul(d["a_prime"], d["e_bar"]), pmul(d["b_bar"], d["r2_bar"])), pmul(a_bar, ng))
    a2 = padd(padd(padd(pmul(d["b_bar"], d["r3_bar"]), pmul(h1, d["c_bar"])), pmul(h3, d["r_bar"])), pmul(big_h1, ng))
    cp = []
    for j in range(L):
        gamma01 = sc_sub(d["gamma"], d["gamma0"][j])
        c0 = d["com"][j]
        c1 = psub(d["com"][j], h1)
        if j == 0:
            p0 = psub(padd(pmul(h2, d["w00"]), pmul(h3, d["z"][0][0])), pmul(c0, d["gamma0"][0]))
            p1 = psub(padd(pmul(h2, d["w01"]), pmul(h3, d["z"][0][1])), pmul(c1, gamma01))
        else:
            p0 = psub(pmul(h3, d["z"][j][0]), pmul(c0, d["gamma0"][j]))
            p1 = psub(pmul(h3, d["z"][j][1]), pmul(c1, gamma01))
        cp.append((p0, p1))
    k_prime = _k_prime(d["com"], L)
    com_ = padd(pmul(h1, d["s"]), k_prime)
    big_c = psub(padd(padd(pmul(h1, sc_neg(d["c_bar"])), pmul(h2, d["k_bar"])), pmul(h3, d["s_bar"])), pmul(com_, d["gamma"]))
    t = Transcript(params, b"spend")
    t.add_scalar(d["k"])
    for p in (d["a_prime"], d["b_bar"], a1, a2):
        t.add_element(p)
    for p in d["com"]:
        t.add_element(p)
    for pair in cp:
        t.add_element(pair[0])
        t.add_element(pair[1])
    t.add_element(big_c)
    return t.challenge(), k_prime, bytes(t.pre)


def refund(sk, params, d, rng: ByteRng, L: int) -> bytes:                     # :781-869
    x, w = sk
    if d["a_prime"] == IDENTITY:                                              # :787-789 (the identity has one encoding)
        raise ActError(E_IDENTITY_POINT)
    gamma, k_prime, _ = spend_challenge(x, params, d, L)
    if gamma != d["gamma"]:
        raise ActError(E_INVALID_CLIENT_SPEND_PROOF)
    e = rng.scalar()
    x_a = padd(generator(), k_prime)
    a = pmul(x_a, sc_invert(sc_add(e, x)))
    x_g = padd(bmul(e), w)
    alpha = rng.scalar()
    y_a = pmul(a, alpha)
    y_g = bmul(alpha)
    t = Transcript(params, b"refund")
    t.add_scalar(e)
    for p in (a, x_a, x_g, y_a, y_g):
        t.add_element(p)
    rg = t.challenge()
    z = sc_add(sc_mul(rg, sc_add(x, e)), alpha)
    return a + e + rg + z                                                     # A* | e | gamma | z (src/cbor.rs:422-427)


def refund_to_credit_token(prerefund: bytes, params, d, rf: bytes, w: bytes, L: int) -> bytes:    # :1217-1253
    r, k, m = _split(prerefund, 3)
    a, e, rg, z = _split(rf, 4)
    x_a = padd(generator(), _k_prime(d["com"], L))
    x_g = padd(bmul(e), w)
    ng = sc_neg(rg)
    y_a = padd(pmul(a, z), pmul(x_a, ng))
    y_g = padd(bmul(z), pmul(x_g, ng))
    t = Transcript(params, b"refund")
    t.add_scalar(e)
    for p in (a, x_a, x_g, y_a, y_g):
        t.add_element(p)
    if t.challenge() != rg:
        raise ActError(E_INVALID_REFUND_PROOF)
    return a + e + k + r + m                                                  # a | e | k | r | c := m
