//! tests/parity.rs — the unmodified crate against its MI355X batch siblings, byte for byte (needs a GPU; the GPU-free half of the
//! pin -- the crate against the committed fixtures -- is tests/golden.rs).
//!
//! UNCOMPILED / UNRUN in the authoring environment (no Rust toolchain there; rust/README.md).  Run on a box with a GPU:
//!     ACT_MI355X_LIB_DIR=.../anonymous-credit-tokens_amd cargo test --features mi355x --test parity
//! Both sides get the same bytes from a byte-replay generator; outputs are compared through `to_cbor()` (the structs'
//! fields are private) and the two generators must have consumed exactly the same number of bytes afterwards — i.e. the
//! batch drew 128 bytes per ACCEPTED lane of issue / refund, like the sequential loop (src/lib.rs:638-643, 842-846).
//! The sequential side must be built from the crate's original method bodies: compile them under
//! `#[cfg(any(test, not(feature = "mi355x")))]` as `*_reference` (in-crate), or run this file against two builds.
use anonymous_credit_tokens::*;
use curve25519_dalek::Scalar;
use rand_core::{CryptoRng, RngCore};

/// Hands out a fixed byte string; panics when exhausted.  `Scalar::random` = one `fill_bytes(&mut [u8; 64])`.
#[derive(Clone)]
struct ReplayRng {
    bytes: std::sync::Arc<Vec<u8>>,
    pos: usize,
}
impl ReplayRng {
    fn new(seed: u64, len: usize) -> Self {
        // any deterministic filler will do: both sides see the same bytes
        let mut x = seed.wrapping_mul(0x9E37_79B9_7F4A_7C15) | 1;
        let bytes = (0..len).map(|_| { x ^= x << 13; x ^= x >> 7; x ^= x << 17; (x >> 24) as u8 }).collect();
        ReplayRng { bytes: std::sync::Arc::new(bytes), pos: 0 }
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

#[test]
fn lifecycle_batch_equals_sequential_loop() {
    const N: usize = 19;
    let params = Params::new("bench-org", "bench-service", "bench-env", "2024-01-01"); // benches/benchmark.rs:9-16
    let sk = PrivateKey::random(ReplayRng::new(1, 64));
    let pres: Vec<PreIssuance> = (0..N).map(|i| PreIssuance::random(ReplayRng::new(100 + i as u64, 128))).collect();

    // request
    let (mut a, mut b) = (ReplayRng::new(2, 128 * N), ReplayRng::new(2, 128 * N));
    let seq: Vec<IssuanceRequest> = pres.iter().map(|p| p.request_reference(&params, &mut a)).collect();
    let bat = PreIssuance::request_batch(&pres, &params, &mut b);
    assert_eq!(seq.iter().map(|r| r.to_cbor().unwrap()).collect::<Vec<_>>(), bat.iter().map(|r| r.to_cbor().unwrap()).collect::<Vec<_>>());
    assert_eq!(a.pos, b.pos);

    // issue, with lanes 3 and 11 carrying a tampered request (random K and gamma: src/tests.rs:570-601)
    let mut reqs = seq;
    for &i in &[3usize, 11] {
        reqs[i] = PreIssuance::random(ReplayRng::new(900 + i as u64, 128)).request_reference(&params, ReplayRng::new(901 + i as u64, 128));
        // ... then overwrite gamma through CBOR so that the PoK fails
        let mut cb = reqs[i].to_cbor().unwrap();
        let at = cb.len() - 3 * 35 + 2; // key 2 (gamma) payload; IssuanceRequest = map(4) of 1 + 34-byte entries
        cb[at] ^= 1;
        reqs[i] = IssuanceRequest::from_cbor(&cb).unwrap();
    }
    let amounts: Vec<Scalar> = (0..N).map(|i| Scalar::from(20u128 + i as u128)).collect();
    let (mut a, mut b) = (ReplayRng::new(3, 128 * N), ReplayRng::new(3, 128 * N));
    let seq: Vec<_> = reqs.iter().zip(&amounts).map(|(r, c)| sk.issue_reference(&params, r, *c, &mut a)).collect();
    let bat = sk.issue_batch(&params, &reqs, &amounts, &mut b);
    assert_eq!(a.pos, b.pos, "the batch must draw 128 bytes per ACCEPTED lane only");
    assert_eq!(a.pos, 128 * (N - 2));
    for i in 0..N {
        match (&seq[i], &bat[i]) {
            (Ok(x), Ok(y)) => assert_eq!(x.to_cbor().unwrap(), y.to_cbor().unwrap(), "lane {i}"),
            (Err(x), Err(y)) => assert_eq!(x, y, "lane {i}"),
            _ => panic!("lane {i}: accept/reject differs"),
        }
    }

    // tokens of the honest lanes
    let ok: Vec<usize> = (0..N).filter(|i| seq[*i].is_ok()).collect();
    let toks: Vec<CreditToken> = ok.iter().map(|&i| pres[i].to_credit_token_reference(&params, sk.public(), &reqs[i], seq[i].as_ref().unwrap()).unwrap()).collect();
    let toks_b = PreIssuance::to_credit_token_batch(&ok.iter().map(|&i| pres[i].clone()).collect::<Vec<_>>(), &params, sk.public(),
                                                    &ok.iter().map(|&i| reqs[i].clone()).collect::<Vec<_>>(),
                                                    &ok.iter().map(|&i| seq[i].clone().unwrap()).collect::<Vec<_>>());
    for (x, y) in toks.iter().zip(&toks_b) {
        assert_eq!(x.to_cbor().unwrap(), y.as_ref().unwrap().to_cbor().unwrap());
    }

    // prove_spend: s = 0, s = c, overspend among the lanes (src/tests.rs:209-257, 339-426)
    let m = toks.len();
    let charges: Vec<Scalar> = (0..m).map(|j| match j { 0 => Scalar::ZERO, 1 => amounts[ok[1]], 2 => amounts[ok[2]] + Scalar::ONE, _ => Scalar::from(5u128) }).collect();
    let per = 64 * (4 * L + 12);
    let (mut a, mut b) = (ReplayRng::new(4, per * m), ReplayRng::new(4, per * m));
    let seq_p: Vec<_> = toks.iter().zip(&charges).map(|(t, s)| t.prove_spend_reference(&params, *s, &mut a)).collect();
    let bat_p = CreditToken::prove_spend_batch(&toks, &params, &charges, &mut b);
    assert_eq!(a.pos, b.pos);
    for (x, y) in seq_p.iter().zip(&bat_p) {
        assert_eq!(x.0.to_cbor().unwrap(), y.0.to_cbor().unwrap());
        assert_eq!(x.1.to_cbor().unwrap(), y.1.to_cbor().unwrap());
    }

    // refund: lane 2 is the overspend (InvalidClientSpendProof), everything else signs
    let proofs: Vec<SpendProof> = seq_p.iter().map(|p| p.0.clone()).collect();
    let (mut a, mut b) = (ReplayRng::new(5, 128 * m), ReplayRng::new(5, 128 * m));
    let seq_r: Vec<_> = proofs.iter().map(|p| sk.refund_reference(&params, p, &mut a)).collect();
    let bat_r = sk.refund_batch(&params, &proofs, &mut b);
    assert_eq!(a.pos, b.pos);
    assert_eq!(a.pos, 128 * (m - 1));
    for i in 0..m {
        match (&seq_r[i], &bat_r[i]) {
            (Ok(x), Ok(y)) => assert_eq!(x.to_cbor().unwrap(), y.to_cbor().unwrap(), "lane {i}"),
            (Err(x), Err(y)) => assert_eq!(x, y, "lane {i}"),
            _ => panic!("lane {i}: accept/reject differs"),
        }
    }
    assert_eq!(bat_r[2], Err(Error::InvalidClientSpendProof));

    // new tokens
    let okr: Vec<usize> = (0..m).filter(|i| seq_r[*i].is_ok()).collect();
    let t2 = PreRefund::to_credit_token_batch(&okr.iter().map(|&i| seq_p[i].1.clone()).collect::<Vec<_>>(), &params,
                                              &okr.iter().map(|&i| proofs[i].clone()).collect::<Vec<_>>(),
                                              &okr.iter().map(|&i| seq_r[i].clone().unwrap()).collect::<Vec<_>>(), sk.public());
    for (j, &i) in okr.iter().enumerate() {
        let want = seq_p[i].1.to_credit_token_reference(&params, &proofs[i], seq_r[i].as_ref().unwrap(), sk.public()).unwrap();
        assert_eq!(want.to_cbor().unwrap(), t2[j].as_ref().unwrap().to_cbor().unwrap());
    }
}

/// Wire bytes in, wire bytes out: `refund_cbor_batch` / `redeem_cbor_batch` against the server loop they replace
/// (`SpendProof::from_cbor` -> nullifier store -> `refund` -> `Refund::to_cbor`; examples/act.rs:62-73), including the state the
/// caller's generator is left in and the messages `from_cbor` rejects.
#[test]
fn wire_level_calls_equal_the_server_loop() {
    use std::collections::HashSet;
    const N: usize = 12;
    let params = Params::new("bench-org", "bench-service", "bench-env", "2024-01-01");
    let sk = PrivateKey::random(ReplayRng::new(1, 64));
    // N tokens of 50 credits, spends of 0..N credits; everything on this side is the crate's own code path
    let mut msgs: Vec<Vec<u8>> = Vec::new();
    for i in 0..N {
        let pre = PreIssuance::random(ReplayRng::new(200 + i as u64, 128));
        let req = pre.request_reference(&params, ReplayRng::new(300 + i as u64, 128));
        let resp = sk.issue_reference(&params, &req, Scalar::from(50u128), ReplayRng::new(400 + i as u64, 128)).unwrap();
        let tok = pre.to_credit_token_reference(&params, sk.public(), &req, &resp).unwrap();
        let (proof, _) = tok.prove_spend_reference(&params, Scalar::from(i as u128), ReplayRng::new(500 + i as u64, 64 * (4 * L + 12)));
        msgs.push(proof.to_cbor().unwrap());
    }
    let flip = |m: &mut Vec<u8>, at: usize| m[at] ^= 1;
    flip(&mut msgs[2], 4 + 34 + 5);             // a bit of the charge `s` (key 2): parses, InvalidClientSpendProof
    msgs[3].truncate(100);                      // cut short: CborError::Ciborium
    msgs[4][0] = 0x80;                          // an array where the map should be: CborError::InvalidStructure
    msgs[5] = msgs[1].clone();                  // a second submission of message 1: a double spend for redeem
    let mut loose = vec![0xbfu8];               // message 6 again as an indefinite-length map: same content, not canonical
    loose.extend_from_slice(&msgs[6][1..]);
    loose.push(0xff);
    msgs[6] = loose;
    let refs: Vec<&[u8]> = msgs.iter().map(|m| m.as_slice()).collect();

    // refund: the loop, then the one call; same bytes, same generator position
    let (mut a, mut b) = (ReplayRng::new(7, 128 * N), ReplayRng::new(7, 128 * N));
    let seq: Vec<Result<Vec<u8>, ()>> = refs.iter().map(|m| match SpendProof::from_cbor(m) {
        Err(_) => Err(()),
        Ok(p) => sk.refund_reference(&params, &p, &mut a).map(|r| r.to_cbor().unwrap()).map_err(|_| ()),
    }).collect();
    let bat = sk.refund_cbor_batch(&params, &refs, &mut b);
    assert_eq!(a.pos, b.pos, "128 bytes per ACCEPTED message, drawn after the verdicts");
    for i in 0..N {
        match (&seq[i], &bat[i]) {
            (Ok(x), Ok(y)) => assert_eq!(x, y, "message {i}"),
            (Err(()), Err(_)) => {}
            _ => panic!("message {i}: accept/reject differs"),
        }
    }
    assert_eq!(bat[2], Err(WireError::Protocol(Error::InvalidClientSpendProof)));
    assert_eq!(bat[3], Err(WireError::Malformed));
    assert_eq!(bat[4], Err(WireError::InvalidStructure));
    assert!(bat[5].is_ok() && bat[6].is_ok());

    // redeem: the same loop with the example's nullifier store in front of refund
    let (mut a, mut b) = (ReplayRng::new(8, 128 * N), ReplayRng::new(8, 128 * N));
    let mut used: HashSet<[u8; 32]> = HashSet::new();
    let seq: Vec<Result<Vec<u8>, Option<Error>>> = refs.iter().map(|m| {
        let p = SpendProof::from_cbor(m).map_err(|_| None::<Error>)?;
        // the engine verifies BEFORE it records (a proof that does not verify must not burn a nullifier); for valid proofs the
        // example's order (mark, then refund) gives the same result
        let mut probe = a.clone();
        if sk.refund_reference(&params, &p, &mut probe).is_err() {
            return sk.refund_reference(&params, &p, &mut a).map(|r| r.to_cbor().unwrap()).map_err(Some);
        }
        if !used.insert(p.nullifier().to_bytes()) {
            return Err(Some(Error::DoubleSpendError));
        }
        sk.refund_reference(&params, &p, &mut a).map(|r| r.to_cbor().unwrap()).map_err(Some)
    }).collect();
    let store = GpuNullifierStore::new(1 << 16);
    let red = sk.redeem_cbor_batch(&params, &store, &refs, &mut b);
    assert!(red.engine_failure.is_none());
    assert_eq!(a.pos, b.pos, "128 bytes per SIGNED message");
    assert_eq!(store.len(), used.len());
    for i in 0..N {
        match (&seq[i], &red.lanes[i]) {
            (Ok(x), Ok(y)) => assert_eq!(x, y, "message {i}"),
            (Err(Some(e)), Err(WireError::Protocol(f))) => assert_eq!(e, f, "message {i}"),
            (Err(None), Err(WireError::Malformed | WireError::InvalidStructure | WireError::InvalidValue)) => {}
            _ => panic!("message {i}: outcome differs"),
        }
    }
    assert_eq!(red.lanes[5], Err(WireError::Protocol(Error::DoubleSpendError)));
}

/// `predrawn::*` (one library call per item, the signature beside the check, the caller hands over the 128 nonce bytes -- NOT
/// methods that take a generator, so the crate's "nothing is drawn for a rejected item" contract of `issue` / `refund` is not
/// touched): for an accepted item the bytes the crate's method produces from a generator whose next 128 bytes are those; for a
/// rejected one the same error.
#[test]
fn predrawn_one_item_calls_equal_the_methods() {
    use anonymous_credit_tokens::mi355x::predrawn;
    let params = Params::new("bench-org", "bench-service", "bench-env", "2024-01-01");
    let sk = PrivateKey::random(ReplayRng::new(1, 64));
    let pre = PreIssuance::random(ReplayRng::new(21, 128));
    let req = pre.request_reference(&params, ReplayRng::new(22, 128));
    // issue
    let mut a = ReplayRng::new(23, 256);
    let nonces = predrawn::nonce_bytes(ReplayRng::new(23, 256));
    let want = sk.issue_reference(&params, &req, Scalar::from(50u128), &mut a).unwrap();
    let got = predrawn::issue(&sk, &params, &req, Scalar::from(50u128), &nonces).unwrap();
    assert_eq!(want.to_cbor().unwrap(), got.to_cbor().unwrap());
    assert_eq!(a.pos, 128);
    let tok = pre.to_credit_token_reference(&params, sk.public(), &req, &want).unwrap();
    let (proof, _) = tok.prove_spend_reference(&params, Scalar::from(7u128), ReplayRng::new(24, 64 * (4 * L + 12)));
    // refund, records and wire bytes
    let mut a = ReplayRng::new(25, 256);
    let nonces = predrawn::nonce_bytes(ReplayRng::new(25, 256));
    let want = sk.refund_reference(&params, &proof, &mut a).unwrap().to_cbor().unwrap();
    assert_eq!(predrawn::refund(&sk, &params, &proof, &nonces).unwrap().to_cbor().unwrap(), want);
    let msg = proof.to_cbor().unwrap();
    assert_eq!(predrawn::refund_cbor(&sk, &params, &msg, &nonces).unwrap(), want);
    // redeem of one message: fresh the first time, a double spend the second; the store holds one nullifier
    let store = GpuNullifierStore::new(1 << 10);
    let first = predrawn::redeem_cbor(&sk, &params, &store, &msg, &nonces);
    assert!(first.engine_failure.is_none());
    assert_eq!(first.lanes[0].as_ref().unwrap(), &want);
    let second = predrawn::redeem_cbor(&sk, &params, &store, &msg, &predrawn::nonce_bytes(ReplayRng::new(27, 128)));
    assert_eq!(second.lanes[0], Err(WireError::Protocol(Error::DoubleSpendError)));
    assert_eq!(store.len(), 1);
    // a rejected item: the same error as the method; and the METHOD (the drop-in surface) leaves its generator untouched
    let mut bad = msg.clone();
    bad[4 + 34 + 5] ^= 1;
    let mut a = ReplayRng::new(26, 128);
    let p_bad = SpendProof::from_cbor(&bad).unwrap();
    assert!(sk.refund_reference(&params, &p_bad, &mut a).is_err() && a.pos == 0);
    let mut b = ReplayRng::new(26, 128);
    assert!(sk.refund(&params, &p_bad, &mut b).is_err() && b.pos == 0);
    assert_eq!(predrawn::refund_cbor(&sk, &params, &bad, &nonces), Err(WireError::Protocol(Error::InvalidClientSpendProof)));
}
