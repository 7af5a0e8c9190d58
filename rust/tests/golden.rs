//! tests/golden.rs — the UNMODIFIED crate against the committed lifecycle fixtures.  No GPU, no `mi355x` feature, no Python:
//!
//!     cd <this repo>/rust/pin && cargo test                        (a self-contained package: rust/pin/README.md)
//! or, inside a checkout of the crate:
//!     cp <this repo>/rust/tests/golden.rs tests/ && cargo add --dev serde_json
//!     ACT_GOLDEN_DIR=<this repo>/tests/golden cargo test --test golden
//!
//! UNCOMPILED / UNRUN in the authoring environment (no Rust toolchain there; rust/README.md).  What it pins: the fixtures
//! `sodium_lifecycle_L128.json` (18 lifecycles computed with libsodium 1.0.18 + LLVM's BLAKE3) and `lifecycle_L128.json` (10, Python
//! big integers) are what every implementation in the MI355X repository -- the C oracle, the Python model, the HIP kernels through the
//! C ABI -- reproduces byte for byte.  If the crate reproduces them too, "the engine equals the oracle" becomes "the engine equals the
//! crate" (parity "partial" for want of exactly this run).  Schema: tests/golden/README.md of that repository.
//!
//! Every rng stream of a fixture is SHAKE-256(label) (FIPS 202), read front to back by the one call it belongs to; a `Scalar::random`
//! is one `fill_bytes(&mut [u8; 64])`.  Records are the structs' fields as consecutive 32-byte strings in CBOR key order
//! (src/cbor.rs:105-110, 163-169, 250-268, 422-427, 477-480, 546-549, 596-602, 656-660); the structs' fields are private, so records
//! are compared through `to_cbor()` / `from_cbor()` with the deterministic framing rebuilt below.
use anonymous_credit_tokens::*;
use curve25519_dalek::Scalar;
use rand_core::{CryptoRng, RngCore};

// ---- SHAKE-256 (FIPS 202): Keccak-f[1600], rate 136, domain 0x1F -------------------------------------------------------------
fn keccak_f(st: &mut [u64; 25]) {
    const RC: [u64; 24] = [
        0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
        0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
        0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
        0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
    ];
    const ROT: [u32; 24] = [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44];
    const PIL: [usize; 24] = [10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1];
    for rc in RC.iter() {
        let mut bc = [0u64; 5];
        for i in 0..5 {
            bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        }
        for i in 0..5 {
            let t = bc[(i + 4) % 5] ^ bc[(i + 1) % 5].rotate_left(1);
            for j in (0..25).step_by(5) {
                st[j + i] ^= t;
            }
        }
        let mut t = st[1];
        for i in 0..24 {
            let j = PIL[i];
            let b = st[j];
            st[j] = t.rotate_left(ROT[i]);
            t = b;
        }
        for j in (0..25).step_by(5) {
            let row = [st[j], st[j + 1], st[j + 2], st[j + 3], st[j + 4]];
            for i in 0..5 {
                st[j + i] = row[i] ^ (!row[(i + 1) % 5] & row[(i + 2) % 5]);
            }
        }
        st[0] ^= rc;
    }
}
fn shake256(label: &str, out_len: usize) -> Vec<u8> {
    const RATE: usize = 136;
    let mut st = [0u64; 25];
    let mut msg = label.as_bytes().to_vec();
    msg.push(0x1F);
    while msg.len() % RATE != 0 {
        msg.push(0);
    }
    let last = msg.len() - 1;
    msg[last] |= 0x80;
    for block in msg.chunks(RATE) {
        for (i, lane) in block.chunks(8).enumerate() {
            st[i] ^= u64::from_le_bytes(lane.try_into().unwrap());
        }
        keccak_f(&mut st);
    }
    let mut out = Vec::with_capacity(out_len + RATE);
    loop {
        for lane in st.iter().take(RATE / 8) {
            out.extend_from_slice(&lane.to_le_bytes());
        }
        if out.len() >= out_len {
            break;
        }
        keccak_f(&mut st);
    }
    out.truncate(out_len);
    out
}
#[test]
fn shake256_known_answer() {
    // FIPS 202 / NIST CAVP: SHAKE256(""), first 32 bytes
    assert_eq!(hex::encode(shake256("", 32)), "46b9dd2b0ba88d13233b3feb743eeb243fcd52ea62b81b82b50c27646ed5762f");
}

/// Hands out a fixed byte string; panics when asked for more (a call that draws more than the fixture says is a finding).
struct ReplayRng {
    bytes: Vec<u8>,
    pos: usize,
}
impl ReplayRng {
    fn shake(label: &str, len: usize) -> Self {
        ReplayRng { bytes: shake256(label, len), pos: 0 }
    }
}
impl RngCore for ReplayRng {
    fn next_u32(&mut self) -> u32 { let mut b = [0u8; 4]; self.fill_bytes(&mut b); u32::from_le_bytes(b) }
    fn next_u64(&mut self) -> u64 { let mut b = [0u8; 8]; self.fill_bytes(&mut b); u64::from_le_bytes(b) }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        dest.copy_from_slice(&self.bytes[self.pos..self.pos + dest.len()]);
        self.pos += dest.len();
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand_core::Error> { self.fill_bytes(dest); Ok(()) }
}
impl CryptoRng for ReplayRng {}

// ---- records <-> the crate's deterministic CBOR (src/cbor.rs) -----------------------------------------------------------------
fn bstr(out: &mut Vec<u8>, field: &[u8]) {
    out.extend_from_slice(&[0x58, 0x20]);
    out.extend_from_slice(field);
}
fn head(out: &mut Vec<u8>, major: u8, n: usize) {
    if n < 24 { out.push(major << 5 | n as u8) } else if n < 256 { out.extend_from_slice(&[major << 5 | 24, n as u8]) } else { out.extend_from_slice(&[major << 5 | 25, (n >> 8) as u8, n as u8]) }
}
/// a record of `n` single 32-byte fields -> map {1: bstr, ..., n: bstr}
fn frame_flat(rec: &[u8]) -> Vec<u8> {
    let n = rec.len() / 32;
    let mut out = Vec::new();
    head(&mut out, 5, n);
    for i in 0..n {
        head(&mut out, 0, i + 1);
        bstr(&mut out, &rec[32 * i..32 * i + 32]);
    }
    out
}
/// SpendProof record (32 * (14 + 4L) bytes) -> the 17-entry map of src/cbor.rs:250-268
fn frame_proof(rec: &[u8]) -> Vec<u8> {
    assert_eq!(rec.len(), 32 * (14 + 4 * L));
    let f = |i: usize| &rec[32 * i..32 * i + 32];
    let mut out = Vec::new();
    head(&mut out, 5, 17);
    let mut key = 0usize;
    let mut single = |out: &mut Vec<u8>, i: usize| { key += 1; head(out, 0, key); bstr(out, f(i)); };
    for i in 0..4 { single(&mut out, i); }                       // k, s, A', B_bar
    head(&mut out, 0, 5); head(&mut out, 4, L);                   // Com[L]
    for j in 0..L { bstr(&mut out, f(4 + j)); }
    let mut key = 5usize;
    for i in 0..8 { key += 1; head(&mut out, 0, key); bstr(&mut out, f(4 + L + i)); }      // gamma, e_bar, r2_bar, r3_bar, c_bar, r_bar, w00, w01
    head(&mut out, 0, 14); head(&mut out, 4, L);                  // gamma0[L]
    for j in 0..L { bstr(&mut out, f(12 + L + j)); }
    head(&mut out, 0, 15); head(&mut out, 4, L);                  // z[L][2]
    for j in 0..L { head(&mut out, 4, 2); bstr(&mut out, f(12 + 2 * L + 2 * j)); bstr(&mut out, f(13 + 2 * L + 2 * j)); }
    head(&mut out, 0, 16); bstr(&mut out, f(12 + 4 * L));         // k_bar
    head(&mut out, 0, 17); bstr(&mut out, f(13 + 4 * L));         // s_bar
    out
}

// ---- fixtures ------------------------------------------------------------------------------------------------------------------
fn load(name: &str) -> serde_json::Value {
    // ACT_GOLDEN_DIR, or -- run from rust/pin of the MI355X repository -- the fixtures where they lie: <repo>/tests/golden
    let dir = std::env::var("ACT_GOLDEN_DIR").unwrap_or_else(|_| format!("{}/../../tests/golden", env!("CARGO_MANIFEST_DIR")));
    let text = std::fs::read_to_string(std::path::Path::new(&dir).join(name)).expect("fixture file");
    serde_json::from_str(&text).expect("fixture JSON")
}
fn rec(case: &serde_json::Value, field: &str) -> Vec<u8> {
    hex::decode(case[field].as_str().unwrap_or_else(|| panic!("field {field}"))).unwrap()
}
fn amount(case: &serde_json::Value, field: &str) -> Scalar {
    Scalar::from(case[field].as_str().unwrap().parse::<u128>().unwrap()) // decimal strings: c, s < 2^128
}
fn status_of(e: &Error) -> u64 {
    // 1 + discriminant, src/lib.rs:102-112
    match e {
        Error::InvalidIssuanceRequestProof => 1,
        Error::InvalidIssuanceResponseProof => 2,
        Error::DoubleSpendError => 3,
        Error::InvalidRefundProof => 4,
        Error::InvalidRefundResponseProof => 5,
        Error::IdentityPointError => 6,
        Error::InvalidClientSpendProof => 7,
        Error::AmountTooBigError => 8,
        Error::ScalarOutOfRangeError => 9,
    }
}

fn run_file(name: &str, sk_label: &str, sk_other_label: &str, tag_prefix: &str) {
    let fx = load(name);
    assert_eq!(fx["L"].as_u64().unwrap() as usize, L);
    let a: Vec<&str> = fx["params_args"].as_array().unwrap().iter().map(|v| v.as_str().unwrap()).collect();
    let params = Params::new(a[0], a[1], a[2], a[3]);                                  // src/lib.rs:291-315
    let sk = PrivateKey::random(ReplayRng::shake(sk_label, 64));                        // :188-194
    let sk_other = PrivateKey::random(ReplayRng::shake(sk_other_label, 64));
    assert_eq!(sk.to_cbor().unwrap(), frame_flat(&rec(&fx, "sk")), "{name}: private key");
    assert_eq!(sk_other.to_cbor().unwrap(), frame_flat(&rec(&fx, "sk_other")));
    for (idx, case) in fx["cases"].as_array().unwrap().iter().enumerate() {
        let tag = format!("{tag_prefix}{idx}");
        let at = |what: &str| format!("{name} case {idx} ({:?}): {what}", case["tamper"]);
        let rng = |suffix: &str, len: usize| ReplayRng::shake(&format!("{tag}-{suffix}"), len);
        let pre = PreIssuance::random(rng("pre", 128));                                 // :432-437
        assert_eq!(pre.to_cbor().unwrap(), frame_flat(&rec(case, "pre")), "{}", at("pre"));
        let req = pre.request(&params, rng("request", 128));                            // :463-487
        assert_eq!(req.to_cbor().unwrap(), frame_flat(&rec(case, "request")), "{}", at("request"));
        let resp = sk.issue(&params, &req, amount(case, "c"), rng("issue", 128)).expect("issue");   // :621-663
        assert_eq!(resp.to_cbor().unwrap(), frame_flat(&rec(case, "response")), "{}", at("response"));
        let tok = pre.to_credit_token(&params, sk.public(), &req, &resp).expect("to_credit_token");  // :528-562
        assert_eq!(tok.to_cbor().unwrap(), frame_flat(&rec(case, "token")), "{}", at("token"));
        let (proof, prer) = tok.prove_spend(&params, amount(case, "s"), rng("prove", 64 * (4 * L + 12)));   // :972-1152
        assert_eq!(prer.to_cbor().unwrap(), frame_flat(&rec(case, "prerefund")), "{}", at("prerefund"));
        let wire = frame_proof(&rec(case, "proof"));
        if case["tamper"].is_null() {
            assert_eq!(proof.to_cbor().unwrap(), wire, "{}", at("proof"));
        }
        // the fixture's proof (tampered or not) as the issuer receives it
        let want = case["status"].as_u64().unwrap();
        let parsed = match SpendProof::from_cbor(&wire) {                               // src/cbor.rs:276-408
            Ok(p) => p,
            Err(_) => {
                assert_eq!(want, 255, "{}", at("from_cbor rejected a proof the fixture accepts"));
                continue;
            }
        };
        assert_ne!(want, 255, "{}", at("from_cbor accepted an undecodable point"));
        let mut refund_rng = rng("refund", 128);
        match sk.refund(&params, &parsed, &mut refund_rng) {                            // :781-869
            Ok(rf) => {
                assert_eq!(want, 0, "{}", at("refund accepted"));
                assert_eq!(refund_rng.pos, 128);
                assert_eq!(rf.to_cbor().unwrap(), frame_flat(&rec(case, "refund")), "{}", at("refund"));
                let tok2 = prer.to_credit_token(&params, &parsed, &rf, sk.public()).expect("refund to_credit_token");   // :1217-1253
                assert_eq!(tok2.to_cbor().unwrap(), frame_flat(&rec(case, "token2")), "{}", at("token2"));
            }
            Err(e) => {
                assert_eq!(status_of(&e), want, "{}", at("refund error"));
                assert_eq!(refund_rng.pos, 0, "{}", at("a rejected proof must not draw (src/lib.rs:842-846)"));
            }
        }
        let other = match sk_other.refund(&params, &parsed, rng("refund", 128)) { Ok(_) => 0, Err(e) => status_of(&e) };
        assert_eq!(other, case["status_other_issuer"].as_u64().unwrap(), "{}", at("another issuer's verdict"));
    }
}

#[test]
fn libsodium_lifecycles() {
    run_file("sodium_lifecycle_L128.json", "sodium-sk", "sodium-sk-other", "sodium-L128-case");
}
#[test]
fn python_model_lifecycles() {
    run_file("lifecycle_L128.json", "golden-sk", "golden-sk-other", "L128-case");
}
