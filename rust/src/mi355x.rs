//! src/mi355x.rs — MI355X batch engine behind the crate's own API (feature `mi355x`).
//!
//! UNCOMPILED in the authoring environment (no Rust toolchain there; rust/README.md).  Binds `libact_mi355x.so`
//! (C ABI: include/act_mi355x.h).  Every `*_batch` method equals the loop
//! `items.iter().map(|x| x.method(params, .., &mut rng))` byte for byte, including the state `rng` is left in:
//! lane i of a batch is one call of the method, raw records are the structs' fields in CBOR key order
//! (src/cbor.rs:105-110, 163-169, 250-268, 422-427, 546-549, 596-602, 656-660), every `Scalar::random` is one 64-byte
//! `fill_bytes`, and issue / refund draw only for accepted lanes (src/lib.rs:638-643, 842-846).
//!
//! The existing single-call signatures are kept (bottom of this file): with the feature on they are batches of one.
use crate::{
    CreditToken, Error, IssuanceRequest, IssuanceResponse, Params, PreIssuance, PreRefund, PrivateKey, PublicKey, Refund,
    SpendProof, L,
};
use curve25519_dalek::{ristretto::CompressedRistretto, RistrettoPoint, Scalar};
use rand_core::CryptoRngCore;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::sync::OnceLock;

#[repr(C)]
pub struct ActNode {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ActNodeNullifierSet {
    _private: [u8; 0],
}
/// `act_rng_source` (include/act_mi355x.h): the caller's generator handed to the library, which draws 128 bytes per lane it signs --
/// once, after every verdict is known -- so the generator ends where a sequential loop over `refund` would have left it.
#[repr(C)]
pub struct ActRngSource {
    draw: unsafe extern "C" fn(rng_ctx: *mut c_void, dst: *mut u8, len: usize) -> c_int,
    rng_ctx: *mut c_void,
}
const ACT_RNG_SEQUENTIAL: c_int = 1;
const ACT_RNG_CALLBACK: c_int = 2;
const REFUND_CBOR_BYTES: usize = 141; // act_cbor_size(ctx, ACT_CBOR_REFUND): A4, then four times (key, 58 20, 32 bytes)
const PROOF_FIELDS: usize = 14 + 4 * L;
const PROOF_BYTES: usize = 32 * PROOF_FIELDS; // 16 832
const PROVE_RNG_BYTES: usize = 64 * (4 * L + 12); // 33 536: draw order of src/lib.rs:978-1058

// one declaration per entry point used, prototypes as in include/act_mi355x.h
extern "C" {
    fn act_node_create(h: *const u8, l: c_int, devices: *const c_int, n_devices: c_int, max_batch: usize, out: *mut *mut ActNode) -> c_int;
    fn act_node_destroy(node: *mut ActNode);
    fn act_node_last_error(node: *const ActNode) -> *const c_char;
    fn act_node_request_batch(node: *mut ActNode, n: usize, pre: *const u8, rng: *const u8, out_req: *mut u8) -> c_int;
    fn act_node_issue_check_batch(node: *mut ActNode, n: usize, req: *const u8, status: *mut u8) -> c_int;
    fn act_node_issue_sign_batch(node: *mut ActNode, n: usize, sk: *const u8, req: *const u8, c: *const u8, status_in: *const u8,
                                 rng: *const u8, rng_mode: c_int, out_resp: *mut u8, status: *mut u8) -> c_int;
    fn act_node_issuance_to_credit_token_batch(node: *mut ActNode, n: usize, pre: *const u8, w: *const u8, req: *const u8,
                                               resp: *const u8, out_token: *mut u8, status: *mut u8) -> c_int;
    fn act_node_prove_spend_batch(node: *mut ActNode, n: usize, token: *const u8, s: *const u8, rng: *const u8,
                                  out_proof: *mut u8, out_prerefund: *mut u8, status: *mut u8) -> c_int;
    fn act_node_prove_spend_seeded_batch(node: *mut ActNode, n: usize, token: *const u8, s: *const u8, seed: *const u8, first_lane: u64,
                                         out_proof: *mut u8, out_prerefund: *mut u8, status: *mut u8) -> c_int;
    fn act_node_verify_spend_batch(node: *mut ActNode, n: usize, sk: *const u8, proof: *const u8, status: *mut u8, out_kprime: *mut u8) -> c_int;
    fn act_node_issue_batch(node: *mut ActNode, n: usize, sk: *const u8, req: *const u8, c: *const u8, rng: *const u8, rng_mode: c_int,
                            out_resp: *mut u8, status: *mut u8) -> c_int;
    fn act_node_refund_batch(node: *mut ActNode, n: usize, sk: *const u8, proof: *const u8, rng: *const u8, rng_mode: c_int,
                             out_refund: *mut u8, status: *mut u8) -> c_int;
    fn act_node_refund_sign_batch(node: *mut ActNode, n: usize, sk: *const u8, kprime: *const u8, status_in: *const u8,
                                  rng: *const u8, rng_mode: c_int, out_refund: *mut u8, status: *mut u8) -> c_int;
    fn act_node_refund_to_credit_token_batch(node: *mut ActNode, n: usize, prerefund: *const u8, proof: *const u8, refund: *const u8,
                                             w: *const u8, out_token: *mut u8, status: *mut u8) -> c_int;
    // wire bytes in, wire bytes out (src/cbor.rs:276-408, src/lib.rs:781-869, src/cbor.rs:421-433) and the redemption step
    // (examples/act.rs:62-73); `rng` with ACT_RNG_CALLBACK points to an ActRngSource
    fn act_node_verify_spend_cbor_batch(node: *mut ActNode, n: usize, sk: *const u8, cbor: *const u8, offsets: *const u64,
                                        status: *mut u8, out_kprime: *mut u8) -> c_int;
    fn act_node_refund_cbor_batch(node: *mut ActNode, n: usize, sk: *const u8, cbor: *const u8, offsets: *const u64,
                                  rng: *const u8, rng_mode: c_int, out_refund_cbor: *mut u8, status: *mut u8) -> c_int;
    fn act_node_redeem_batch(node: *mut ActNode, set: *mut ActNodeNullifierSet, n: usize, sk: *const u8, proof: *const u8,
                             rng: *const u8, rng_mode: c_int, out_refund: *mut u8, status: *mut u8) -> c_int;
    fn act_node_redeem_cbor_batch(node: *mut ActNode, set: *mut ActNodeNullifierSet, n: usize, sk: *const u8, cbor: *const u8,
                                  offsets: *const u64, rng: *const u8, rng_mode: c_int, out_refund_cbor: *mut u8, status: *mut u8) -> c_int;
    fn act_node_nullifier_set_create(devices: *const c_int, n_devices: c_int, capacity_per_device: usize, salt: *const u8,
                                     out: *mut *mut ActNodeNullifierSet) -> c_int;
    fn act_node_nullifier_set_destroy(set: *mut ActNodeNullifierSet);
    fn act_node_nullifier_set_len(set: *const ActNodeNullifierSet) -> usize;
    fn act_node_nullifier_set_last_error(set: *const ActNodeNullifierSet) -> *const c_char;
    fn act_node_device_stats(node: *mut ActNode, k: c_int, weight: *mut f64, last_lanes: *mut u64, last_seconds: *mut f64, last_calls: *mut u64) -> c_int;
}

/// ACT_MI355X_DEVICES = "0,1,2,3,4,5,6,7" (default: device 0)
fn devices_from_env() -> Vec<c_int> {
    std::env::var("ACT_MI355X_DEVICES").map(|s| s.split(',').filter_map(|d| d.trim().parse().ok()).collect()).unwrap_or_else(|_| vec![0])
}

/// All GPUs of the node behind one handle (contiguous pieces, load-balanced, no collective).
///
/// How `Params` (src/lib.rs:222-229, `#[derive(Clone)]`) carries it: add the field
/// `#[cfg(feature = "mi355x")] gpu: GpuSlot` and nothing else -- `GpuSlot` below is `Clone` (a clone starts EMPTY and builds its own
/// handle on first use, so `#[derive(Clone)]` on `Params` keeps compiling and clones never share engine state), `Default`, `Send`
/// and `Sync`.
///
/// Threads: safe Rust may call `request` / `issue` / `refund` on one `&Params` from many threads at once.  The library serialises
/// them itself -- every `act_node_*_batch` call takes the node handle's lock and every context entry point the context's
/// (include/act_mi355x.h, "A node handle ... may be shared between host threads") -- so concurrent callers are served one after
/// the other and never observe each other's staging buffers, cached key or error string.  That is why `Gpu` may be `Sync`; the
/// guarantee lives in the library, not in a promise by the caller.
pub struct Gpu(*mut ActNode);
// SAFETY: the handle is only ever passed to act_node_* entry points.  Calls that are cut over the GPUs take the node's lock
// (csrc/node.cpp `node_lock`); the few-item calls that go to one context are serialised by that context's lock -- either way no two
// threads are ever inside the same context's buffers.  act_node_destroy runs from Drop, i.e. with exclusive access.  The error text is safe too: act_node_last_error copies the
// CALLING thread's own last failure on the handle (kept per thread by the library) into a buffer of that thread, valid until it asks
// again, so `check` below neither reads a string another thread's failing call is rewriting nor reports another thread's error.
unsafe impl Send for Gpu {}
unsafe impl Sync for Gpu {}
impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { act_node_destroy(self.0) }
    }
}

/// The `Params` field: a lazily built `Gpu` that does not get in the way of `#[derive(Clone)]`.
#[derive(Default)]
pub struct GpuSlot(OnceLock<Gpu>);
impl Clone for GpuSlot {
    fn clone(&self) -> Self {
        GpuSlot(OnceLock::new()) // a cloned Params builds its own node handle on first use
    }
}
impl std::fmt::Debug for GpuSlot {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str(if self.0.get().is_some() { "GpuSlot(ready)" } else { "GpuSlot(empty)" })
    }
}

impl Gpu {
    fn new(params: &Params) -> Gpu {
        let mut h = [0u8; 96]; // enc(h1) | enc(h2) | enc(h3), as Transcript::new hashes them (src/transcript.rs:64-66)
        h[..32].copy_from_slice(params.h1.basepoint().compress().as_bytes());
        h[32..64].copy_from_slice(params.h2.basepoint().compress().as_bytes());
        h[64..].copy_from_slice(params.h3.basepoint().compress().as_bytes());
        let devices = devices_from_env();
        // The library never touches the process environment.  A service with many HIP streams exports GPU_MAX_HW_QUEUES=8 before its
        // first HIP call; the engine measures whether its two pipeline streams overlap either way.
        // ACT_MI355X_MAX_BATCH: records per internal launch.  Unset = the library default, 65 536: 27 GB of workspace per GPU
        // + 0.5 GB of tables, and nothing else: wider fixed-base windows (+47 GB, +3 % verifies/s) are asked for explicitly with
        // act_node_set_fixed_base_bits (include/act_mi355x.h).  A service that only ever sees small batches sets e.g. 4096 (1.7 GB).
        let max_batch: usize = std::env::var("ACT_MI355X_MAX_BATCH").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut node = std::ptr::null_mut();
        let rc = unsafe { act_node_create(h.as_ptr(), L as c_int, devices.as_ptr(), devices.len() as c_int, max_batch, &mut node) };
        if rc != 0 {
            let msg = if node.is_null() { String::new() } else { unsafe { CStr::from_ptr(act_node_last_error(node)) }.to_string_lossy().into_owned() };
            panic!("act_node_create failed ({rc}): {msg}"); // infrastructure failure, not a protocol error
        }
        Gpu(node)
    }
    fn check(&self, rc: c_int) {
        if rc != 0 {
            let msg = unsafe { CStr::from_ptr(act_node_last_error(self.0)) }.to_string_lossy().into_owned();
            panic!("MI355X engine failure ({rc}): {msg}");
        }
    }
}

impl Params {
    pub(crate) fn gpu(&self) -> &Gpu {
        self.gpu.0.get_or_init(|| Gpu::new(self))
    }
}

fn status_to_error(s: u8) -> Error {
    match s {
        // 1 + discriminant, src/lib.rs:102-112
        1 => Error::InvalidIssuanceRequestProof,
        2 => Error::InvalidIssuanceResponseProof,
        3 => Error::DoubleSpendError,
        4 => Error::InvalidRefundProof,
        5 => Error::InvalidRefundResponseProof,
        6 => Error::IdentityPointError,
        7 => Error::InvalidClientSpendProof,
        8 => Error::AmountTooBigError,
        _ => Error::ScalarOutOfRangeError, // 255 cannot occur: every point of a Rust struct is a valid RistrettoPoint
    }
}

// ---- record marshalling ------------------------------------------------------------------------------------------
fn put_s(out: &mut Vec<u8>, s: &Scalar) {
    out.extend_from_slice(s.as_bytes());
}
fn put_p(out: &mut Vec<u8>, p: &RistrettoPoint) {
    out.extend_from_slice(p.compress().as_bytes());
}
fn get_s(rec: &[u8], field: usize) -> Scalar {
    // the engine writes canonical scalars
    Scalar::from_canonical_bytes(rec[32 * field..32 * field + 32].try_into().unwrap()).unwrap()
}
fn get_p(rec: &[u8], field: usize) -> RistrettoPoint {
    CompressedRistretto::from_slice(&rec[32 * field..32 * field + 32]).unwrap().decompress().expect("engine wrote a valid point")
}
/// `count` x `Scalar::random` worth of bytes, drawn exactly as dalek draws them: one 64-byte `fill_bytes` each.
fn draw(rng: &mut impl CryptoRngCore, count: usize) -> Vec<u8> {
    let mut buf = vec![0u8; 64 * count];
    for chunk in buf.chunks_exact_mut(64) {
        rng.fill_bytes(chunk);
    }
    buf
}

// Marshalling is the expensive part of the struct-level batch API: a `RistrettoPoint` becomes 32 bytes by `compress()` (one inverse
// square root, ~4 us on one core) and comes back by `decompress()` (another).  A SpendProof holds 130 points: ~0.5 ms of compress per
// proof = ~2 000 proofs/s per core in front of an engine that verifies ~514 000/s per GPU (INTEGRATION.md section 6 has the measured
// figures).  So records are written and read on `std::thread::scope` workers, one slice of the batch each; a server that can should
// use the wire-level calls further down (`refund_cbor_batch`), where no point is ever built on the host.
fn workers_for(items: usize) -> usize {
    let cpus = std::thread::available_parallelism().map(|n| n.get()).unwrap_or(1);
    cpus.min(items / 8).max(1) // fewer than 8 items per worker is not worth a thread
}
/// `rec_bytes` per item, written by `write(item, out)` (which appends exactly `rec_bytes`), on scoped worker threads.
fn marshal<T: Sync>(items: &[T], rec_bytes: usize, write: impl Fn(&T, &mut Vec<u8>) + Sync) -> Vec<u8> {
    let n = items.len();
    let workers = workers_for(n);
    if workers == 1 {
        let mut out = Vec::with_capacity(n * rec_bytes);
        items.iter().for_each(|it| write(it, &mut out));
        return out;
    }
    let per = (n + workers - 1) / workers;
    let parts: Vec<Vec<u8>> = std::thread::scope(|sc| {
        let handles: Vec<_> = items
            .chunks(per)
            .map(|chunk| {
                let write = &write;
                sc.spawn(move || {
                    let mut out = Vec::with_capacity(chunk.len() * rec_bytes);
                    chunk.iter().for_each(|it| write(it, &mut out));
                    out
                })
            })
            .collect();
        handles.into_iter().map(|h| h.join().expect("marshalling worker")).collect()
    });
    let mut out = Vec::with_capacity(n * rec_bytes);
    parts.iter().for_each(|p| out.extend_from_slice(p));
    out
}
/// The reverse: `n` results from `rec_bytes`-sized records + statuses, `read(record)` on scoped worker threads.
fn unmarshal<T: Send>(records: &[u8], rec_bytes: usize, status: &[u8], read: impl Fn(&[u8]) -> T + Sync) -> Vec<Result<T, Error>> {
    let n = status.len();
    let one = |i: usize| if status[i] == 0 { Ok(read(&records[rec_bytes * i..rec_bytes * (i + 1)])) } else { Err(status_to_error(status[i])) };
    let workers = workers_for(n);
    if workers == 1 {
        return (0..n).map(one).collect();
    }
    let per = (n + workers - 1) / workers;
    let parts: Vec<Vec<Result<T, Error>>> = std::thread::scope(|sc| {
        let handles: Vec<_> = (0..n)
            .step_by(per)
            .map(|lo| {
                let one = &one;
                sc.spawn(move || (lo..(lo + per).min(n)).map(one).collect::<Vec<_>>())
            })
            .collect();
        handles.into_iter().map(|h| h.join().expect("unmarshalling worker")).collect()
    });
    parts.into_iter().flatten().collect()
}

impl PrivateKey {
    fn record(&self) -> [u8; 64] {
        // x | w (src/cbor.rs:477-480)
        let mut r = [0u8; 64];
        r[..32].copy_from_slice(self.x.as_bytes());
        r[32..].copy_from_slice(self.public.w.compress().as_bytes());
        r
    }
}
impl IssuanceRequest {
    fn write_record(&self, out: &mut Vec<u8>) {
        // K | gamma | k_bar | r_bar (src/cbor.rs:105-110)
        put_p(out, &self.big_k);
        put_s(out, &self.gamma);
        put_s(out, &self.k_bar);
        put_s(out, &self.r_bar);
    }
    fn from_record(r: &[u8]) -> Self {
        IssuanceRequest { big_k: get_p(r, 0), gamma: get_s(r, 1), k_bar: get_s(r, 2), r_bar: get_s(r, 3) }
    }
}
impl IssuanceResponse {
    fn write_record(&self, out: &mut Vec<u8>) {
        // A | e | gamma | z | c (src/cbor.rs:163-169)
        put_p(out, &self.a);
        for s in [&self.e, &self.gamma, &self.z, &self.c] {
            put_s(out, s);
        }
    }
    fn from_record(r: &[u8]) -> Self {
        IssuanceResponse { a: get_p(r, 0), e: get_s(r, 1), gamma: get_s(r, 2), z: get_s(r, 3), c: get_s(r, 4) }
    }
}
impl CreditToken {
    fn write_record(&self, out: &mut Vec<u8>) {
        // a | e | k | r | c (src/cbor.rs:596-602)
        put_p(out, &self.a);
        for s in [&self.e, &self.k, &self.r, &self.c] {
            put_s(out, s);
        }
    }
    fn from_record(r: &[u8]) -> Self {
        CreditToken { a: get_p(r, 0), e: get_s(r, 1), k: get_s(r, 2), r: get_s(r, 3), c: get_s(r, 4) }
    }
}
impl SpendProof {
    /// k | s | A' | B_bar | Com[L] | gamma | e_bar | r2_bar | r3_bar | c_bar | r_bar | w00 | w01 | gamma0[L] | z[L][2] | k_bar | s_bar
    /// (src/cbor.rs:250-268)
    fn write_record(&self, out: &mut Vec<u8>) {
        put_s(out, &self.k);
        put_s(out, &self.s);
        put_p(out, &self.a_prime);
        put_p(out, &self.b_bar);
        self.com.iter().for_each(|c| put_p(out, c));
        for s in [&self.gamma, &self.e_bar, &self.r2_bar, &self.r3_bar, &self.c_bar, &self.r_bar, &self.w00, &self.w01] {
            put_s(out, s);
        }
        self.gamma0.iter().for_each(|g| put_s(out, g));
        self.z.iter().for_each(|z| {
            put_s(out, &z[0]);
            put_s(out, &z[1]);
        });
        put_s(out, &self.k_bar);
        put_s(out, &self.s_bar);
    }
    fn from_record(r: &[u8]) -> Self {
        let mut com = [RistrettoPoint::default(); L];
        let mut gamma0 = [Scalar::ZERO; L];
        let mut z = [[Scalar::ZERO; 2]; L];
        for j in 0..L {
            com[j] = get_p(r, 4 + j);
            gamma0[j] = get_s(r, 12 + L + j);
            z[j] = [get_s(r, 12 + 2 * L + 2 * j), get_s(r, 13 + 2 * L + 2 * j)];
        }
        SpendProof {
            k: get_s(r, 0), s: get_s(r, 1), a_prime: get_p(r, 2), b_bar: get_p(r, 3), com,
            gamma: get_s(r, 4 + L), e_bar: get_s(r, 5 + L), r2_bar: get_s(r, 6 + L), r3_bar: get_s(r, 7 + L),
            c_bar: get_s(r, 8 + L), r_bar: get_s(r, 9 + L), w00: get_s(r, 10 + L), w01: get_s(r, 11 + L),
            gamma0, z, k_bar: get_s(r, 12 + 4 * L), s_bar: get_s(r, 13 + 4 * L),
        }
    }
}
impl Refund {
    fn write_record(&self, out: &mut Vec<u8>) {
        // A* | e | gamma | z (src/cbor.rs:422-427)
        put_p(out, &self.a);
        for s in [&self.e, &self.gamma, &self.z] {
            put_s(out, s);
        }
    }
    fn from_record(r: &[u8]) -> Self {
        Refund { a: get_p(r, 0), e: get_s(r, 1), gamma: get_s(r, 2), z: get_s(r, 3) }
    }
}

// ---- the batch siblings ----------------------------------------------------------------------------------------------
impl PreIssuance {
    /// Batch sibling of `request` (src/lib.rs:463-487): lane i draws k', r' (2 x 64 bytes), in lane order.
    pub fn request_batch(pres: &[PreIssuance], params: &Params, mut rng: impl CryptoRngCore) -> Vec<IssuanceRequest> {
        let n = pres.len();
        let mut rec = Vec::with_capacity(64 * n);
        for p in pres {
            put_s(&mut rec, &p.r); // r | k (src/cbor.rs:546-549)
            put_s(&mut rec, &p.k);
        }
        let rng_bytes = draw(&mut rng, 2 * n);
        let mut out = vec![0u8; 128 * n];
        let gpu = params.gpu();
        gpu.check(unsafe { act_node_request_batch(gpu.0, n, rec.as_ptr(), rng_bytes.as_ptr(), out.as_mut_ptr()) });
        out.chunks_exact(128).map(IssuanceRequest::from_record).collect()
    }

    /// Batch sibling of `to_credit_token` (src/lib.rs:528-562).
    pub fn to_credit_token_batch(pres: &[PreIssuance], params: &Params, public: &PublicKey, requests: &[IssuanceRequest],
                                 responses: &[IssuanceResponse]) -> Vec<Result<CreditToken, Error>> {
        let n = pres.len();
        assert!(requests.len() == n && responses.len() == n);
        let (mut pre, mut req, mut resp) = (Vec::with_capacity(64 * n), Vec::with_capacity(128 * n), Vec::with_capacity(160 * n));
        for i in 0..n {
            put_s(&mut pre, &pres[i].r);
            put_s(&mut pre, &pres[i].k);
            requests[i].write_record(&mut req);
            responses[i].write_record(&mut resp);
        }
        let w = public.w.compress();
        let (mut out, mut status) = (vec![0u8; 160 * n], vec![0u8; n]);
        let gpu = params.gpu();
        gpu.check(unsafe {
            act_node_issuance_to_credit_token_batch(gpu.0, n, pre.as_ptr(), w.as_bytes().as_ptr(), req.as_ptr(), resp.as_ptr(), out.as_mut_ptr(), status.as_mut_ptr())
        });
        (0..n).map(|i| if status[i] == 0 { Ok(CreditToken::from_record(&out[160 * i..160 * i + 160])) } else { Err(status_to_error(status[i])) }).collect()
    }
}

impl PrivateKey {
    /// Batch sibling of `issue` (src/lib.rs:621-663).  e, alpha are drawn only for lanes whose PoK verified, in lane order.
    pub fn issue_batch(&self, params: &Params, requests: &[IssuanceRequest], amounts: &[Scalar], mut rng: impl CryptoRngCore)
        -> Vec<Result<IssuanceResponse, Error>> {
        let n = requests.len();
        assert_eq!(amounts.len(), n);
        let (mut req, mut c) = (Vec::with_capacity(128 * n), Vec::with_capacity(32 * n));
        requests.iter().for_each(|r| r.write_record(&mut req));
        amounts.iter().for_each(|a| put_s(&mut c, a));
        let sk = self.record();
        let gpu = params.gpu();
        let mut checked = vec![0u8; n];
        gpu.check(unsafe { act_node_issue_check_batch(gpu.0, n, req.as_ptr(), checked.as_mut_ptr()) }); // :629-640
        let accepted = checked.iter().filter(|&&s| s == 0).count();
        let rng_bytes = draw(&mut rng, 2 * accepted); // :643, :649 -- exactly what the sequential loop would have drawn
        let (mut out, mut status) = (vec![0u8; 160 * n], vec![0u8; n]);
        gpu.check(unsafe {
            act_node_issue_sign_batch(gpu.0, n, sk.as_ptr(), req.as_ptr(), c.as_ptr(), checked.as_ptr(), rng_bytes.as_ptr(), ACT_RNG_SEQUENTIAL,
                                      out.as_mut_ptr(), status.as_mut_ptr())
        });
        (0..n).map(|i| if status[i] == 0 { Ok(IssuanceResponse::from_record(&out[160 * i..160 * i + 160])) } else { Err(status_to_error(status[i])) }).collect()
    }

    /// Batch sibling of `refund` (src/lib.rs:781-869): verification (:787-844) on the GPUs, then e, alpha for the accepted
    /// lanes only (:846, :852), then the BBS signature on K' (:848-868).
    pub fn refund_batch(&self, params: &Params, proofs: &[SpendProof], mut rng: impl CryptoRngCore) -> Vec<Result<Refund, Error>> {
        let n = proofs.len();
        let rec = marshal(proofs, PROOF_BYTES, |p, out| p.write_record(out)); // 130 compress() per proof: on all cores
        let sk = self.record();
        let gpu = params.gpu();
        let (mut checked, mut kprime) = (vec![0u8; n], vec![0u8; 32 * n]);
        gpu.check(unsafe { act_node_verify_spend_batch(gpu.0, n, sk.as_ptr(), rec.as_ptr(), checked.as_mut_ptr(), kprime.as_mut_ptr()) });
        let accepted = checked.iter().filter(|&&s| s == 0).count();
        let rng_bytes = draw(&mut rng, 2 * accepted);
        let (mut out, mut status) = (vec![0u8; 128 * n], vec![0u8; n]);
        gpu.check(unsafe {
            act_node_refund_sign_batch(gpu.0, n, sk.as_ptr(), kprime.as_ptr(), checked.as_ptr(), rng_bytes.as_ptr(), ACT_RNG_SEQUENTIAL,
                                       out.as_mut_ptr(), status.as_mut_ptr())
        });
        unmarshal(&out, 128, &status, Refund::from_record)
    }
}

impl CreditToken {
    /// Batch sibling of `prove_spend` (src/lib.rs:972-1152): lane i draws its 4L + 12 scalars in the order of :978-1058.
    pub fn prove_spend_batch(tokens: &[CreditToken], params: &Params, charges: &[Scalar], mut rng: impl CryptoRngCore)
        -> Vec<(SpendProof, PreRefund)> {
        let n = tokens.len();
        assert_eq!(charges.len(), n);
        let tok = marshal(tokens, 160, |t, out| t.write_record(out));
        let mut s = Vec::with_capacity(32 * n);
        charges.iter().for_each(|c| put_s(&mut s, c));
        let rng_bytes = draw(&mut rng, (4 * L + 12) * n);
        debug_assert_eq!(rng_bytes.len(), PROVE_RNG_BYTES * n);
        let (mut proofs, mut prer, mut status) = (vec![0u8; PROOF_BYTES * n], vec![0u8; 96 * n], vec![0u8; n]);
        let gpu = params.gpu();
        gpu.check(unsafe {
            act_node_prove_spend_batch(gpu.0, n, tok.as_ptr(), s.as_ptr(), rng_bytes.as_ptr(), proofs.as_mut_ptr(), prer.as_mut_ptr(), status.as_mut_ptr())
        });
        proofs_out(&proofs, &prer)
    }
}

/// (proof records, PreRefund records) -> structs; 130 `decompress()` per proof, on scoped worker threads
fn proofs_out(proofs: &[u8], prer: &[u8]) -> Vec<(SpendProof, PreRefund)> {
    let n = prer.len() / 96;
    let zeros = vec![0u8; n]; // prove_spend has no failing lanes: every status is Ok
    let ps = unmarshal(proofs, PROOF_BYTES, &zeros, SpendProof::from_record);
    ps.into_iter()
        .enumerate()
        .map(|(i, p)| {
            let r = &prer[96 * i..96 * i + 96]; // r | k | m (src/cbor.rs:656-660)
            (p.unwrap_or_else(|_| unreachable!()), PreRefund { r: get_s(r, 0), k: get_s(r, 1), m: get_s(r, 2) })
        })
        .collect()
}

/// The generator `prove_spend_seeded_batch` gives lane `lane`: the BLAKE3 XOF of `seed | lane.to_le_bytes()`, read sequentially.
/// `tokens[i].prove_spend(params, charges[i], XofRng::new(&seed, first_lane + i as u64))` is the CPU twin of lane i.
pub struct XofRng(blake3::OutputReader);
impl XofRng {
    pub fn new(seed: &[u8; 32], lane: u64) -> Self {
        let mut h = blake3::Hasher::new();
        h.update(seed);
        h.update(&lane.to_le_bytes());
        XofRng(h.finalize_xof())
    }
}
impl rand_core::RngCore for XofRng {
    fn next_u32(&mut self) -> u32 { let mut b = [0u8; 4]; self.0.fill(&mut b); u32::from_le_bytes(b) }
    fn next_u64(&mut self) -> u64 { let mut b = [0u8; 8]; self.0.fill(&mut b); u64::from_le_bytes(b) }
    fn fill_bytes(&mut self, dest: &mut [u8]) { self.0.fill(dest) }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand_core::Error> { self.0.fill(dest); Ok(()) }
}
impl rand_core::CryptoRng for XofRng {}

impl CreditToken {
    /// `prove_spend_batch` with seeded generators: the 33 536 rng bytes per proof are expanded in HBM (k_xof_expand) instead of being
    /// drawn on the host and copied over PCIe.  `seed` is a secret of the prover; a (seed, lane) pair must never be used twice.
    pub fn prove_spend_seeded_batch(tokens: &[CreditToken], params: &Params, charges: &[Scalar], seed: &[u8; 32], first_lane: u64)
        -> Vec<(SpendProof, PreRefund)> {
        let n = tokens.len();
        assert_eq!(charges.len(), n);
        let tok = marshal(tokens, 160, |t, out| t.write_record(out));
        let mut s = Vec::with_capacity(32 * n);
        charges.iter().for_each(|c| put_s(&mut s, c));
        let (mut proofs, mut prer, mut status) = (vec![0u8; PROOF_BYTES * n], vec![0u8; 96 * n], vec![0u8; n]);
        let gpu = params.gpu();
        gpu.check(unsafe {
            act_node_prove_spend_seeded_batch(gpu.0, n, tok.as_ptr(), s.as_ptr(), seed.as_ptr(), first_lane, proofs.as_mut_ptr(), prer.as_mut_ptr(), status.as_mut_ptr())
        });
        proofs_out(&proofs, &prer)
    }
}

impl PreRefund {
    /// Batch sibling of `to_credit_token` (src/lib.rs:1217-1253).
    pub fn to_credit_token_batch(pres: &[PreRefund], params: &Params, proofs: &[SpendProof], refunds: &[Refund], public_key: &PublicKey)
        -> Vec<Result<CreditToken, Error>> {
        let n = pres.len();
        assert!(proofs.len() == n && refunds.len() == n);
        let (mut pre, mut rf) = (Vec::with_capacity(96 * n), Vec::with_capacity(128 * n));
        for i in 0..n {
            put_s(&mut pre, &pres[i].r);
            put_s(&mut pre, &pres[i].k);
            put_s(&mut pre, &pres[i].m);
            refunds[i].write_record(&mut rf);
        }
        let rec = marshal(proofs, PROOF_BYTES, |p, out| p.write_record(out));
        let w = public_key.w.compress();
        let (mut out, mut status) = (vec![0u8; 160 * n], vec![0u8; n]);
        let gpu = params.gpu();
        gpu.check(unsafe {
            act_node_refund_to_credit_token_batch(gpu.0, n, pre.as_ptr(), rec.as_ptr(), rf.as_ptr(), w.as_bytes().as_ptr(), out.as_mut_ptr(), status.as_mut_ptr())
        });
        (0..n).map(|i| if status[i] == 0 { Ok(CreditToken::from_record(&out[160 * i..160 * i + 160])) } else { Err(status_to_error(status[i])) }).collect()
    }
}

// ---- wire bytes in, wire bytes out ---------------------------------------------------------------------------------------
// A server does not receive `SpendProof`s, it receives bytes.  The loop it runs today --
//     let proof = SpendProof::from_cbor(&msg)?;  let refund = key.refund(&params, &proof, &mut rng)?;  refund.to_cbor()
// (src/cbor.rs:276-408, src/lib.rs:781-869, src/cbor.rs:421-433) -- builds 130 RistrettoPoints per message on the host only to
// have them compressed again for the GPU.  These calls hand the bytes over as they came: no point, no scalar, no `SpendProof` and
// no `Refund` ever exists on the host; message i in, message i out.

/// What a lane of a wire-level call can come back with instead of bytes.
#[derive(Debug, PartialEq)]
pub enum WireError {
    /// `CborError::Ciborium`: not well-formed CBOR
    Malformed,
    /// `CborError::InvalidStructure`: not a map, a field missing, wrong shape or length
    InvalidStructure,
    /// `CborError::InvalidValue`: a point that is not a canonical Ristretto encoding
    InvalidValue,
    /// the message decoded; `refund` (or the nullifier store: `DoubleSpendError`) rejected it
    Protocol(Error),
    /// redeem only, and only together with an engine failure: verified, but the nullifier store could not answer -- NOT recorded,
    /// NOT signed, safe to resubmit (include/act_mi355x.h ACT_STATUS_NULLIFIER_UNDETERMINED)
    NullifierUndetermined,
    /// redeem only, and only together with an engine failure: the nullifier IS recorded and the signature step failed -- the refund
    /// is owed: sign it (`refund`), never redeem it again (ACT_STATUS_RECORDED_UNSIGNED)
    RecordedUnsigned,
}
fn status_to_wire_error(s: u8) -> WireError {
    match s {
        254 => WireError::Malformed,
        253 => WireError::InvalidStructure,
        255 => WireError::InvalidValue,
        252 => WireError::NullifierUndetermined,
        251 => WireError::RecordedUnsigned,
        other => WireError::Protocol(status_to_error(other)),
    }
}

/// The caller's generator as the library's draw callback.  dalek draws a `Scalar::random` with one `fill_bytes(&mut [u8; 64])`;
/// so does this, 64 bytes at a time, so that generators that are not plain byte streams still see the calls they would have seen.
/// Returns 0 = all `len` bytes written; a generator that can fail reports it through `try_fill_bytes` and the library then signs
/// nothing (ACT_ERR_RNG).  (A panic inside the generator cannot unwind through the C frames: the process aborts.)
unsafe extern "C" fn draw_trampoline<R: CryptoRngCore>(rng_ctx: *mut c_void, dst: *mut u8, len: usize) -> c_int {
    let rng = &mut *(rng_ctx as *mut R);
    let buf = std::slice::from_raw_parts_mut(dst, len);
    for chunk in buf.chunks_mut(64) {
        if rng.try_fill_bytes(chunk).is_err() {
            return 1;
        }
    }
    0
}
fn rng_source<R: CryptoRngCore>(rng: &mut R) -> ActRngSource {
    ActRngSource { draw: draw_trampoline::<R>, rng_ctx: rng as *mut R as *mut c_void }
}

/// `msgs` gathered into one buffer + offsets (what the C ABI takes).  The copy is 18 KB per message at memcpy speed; a server that
/// already holds its messages in one buffer calls the `*_blob` forms and skips it.
fn gather(msgs: &[&[u8]]) -> (Vec<u8>, Vec<u64>) {
    let total: usize = msgs.iter().map(|m| m.len()).sum();
    let mut offsets = Vec::with_capacity(msgs.len() + 1);
    let mut blob = Vec::with_capacity(total + 1);
    offsets.push(0u64);
    for m in msgs {
        blob.extend_from_slice(m);
        offsets.push(blob.len() as u64);
    }
    blob.push(0); // never an empty allocation: the library wants a non-null pointer
    (blob, offsets)
}
fn refund_messages(out: &[u8], status: &[u8]) -> Vec<Result<Vec<u8>, WireError>> {
    status
        .iter()
        .enumerate()
        .map(|(i, &s)| if s == 0 { Ok(out[REFUND_CBOR_BYTES * i..REFUND_CBOR_BYTES * (i + 1)].to_vec()) } else { Err(status_to_wire_error(s)) })
        .collect()
}

impl PrivateKey {
    /// `msgs.iter().map(|m| SpendProof::from_cbor(m).map(|p| self.refund(params, &p, &mut rng)).map(|r| r.to_cbor()))` as ONE call:
    /// CBOR `SpendProof` messages in, CBOR `Refund` messages out, byte for byte what that loop returns, `rng` left where it leaves
    /// it (e, alpha are drawn for accepted lanes only, in lane order, after all verdicts: src/lib.rs:842-852).
    pub fn refund_cbor_batch(&self, params: &Params, msgs: &[&[u8]], rng: impl CryptoRngCore) -> Vec<Result<Vec<u8>, WireError>> {
        let (blob, offsets) = gather(msgs);
        self.refund_cbor_blob(params, &blob, &offsets, rng)
    }
    /// The same over messages that already lie in one buffer: message i = `blob[offsets[i]..offsets[i + 1]]`.
    pub fn refund_cbor_blob(&self, params: &Params, blob: &[u8], offsets: &[u64], mut rng: impl CryptoRngCore) -> Vec<Result<Vec<u8>, WireError>> {
        assert!(!offsets.is_empty() && *offsets.last().unwrap() as usize <= blob.len());
        let n = offsets.len() - 1;
        let sk = self.record();
        let gpu = params.gpu();
        let src = rng_source(&mut rng);
        let (mut out, mut status) = (vec![0u8; REFUND_CBOR_BYTES * n + 1], vec![0u8; n + 1]);
        gpu.check(unsafe {
            act_node_refund_cbor_batch(gpu.0, n, sk.as_ptr(), blob.as_ptr(), offsets.as_ptr(), &src as *const ActRngSource as *const u8,
                                       ACT_RNG_CALLBACK, out.as_mut_ptr(), status.as_mut_ptr())
        });
        refund_messages(&out, &status[..n])
    }
    /// Verdicts only (`refund` up to the challenge check, src/lib.rs:787-844) for CBOR `SpendProof` messages.
    pub fn verify_spend_cbor_batch(&self, params: &Params, msgs: &[&[u8]]) -> Vec<Result<(), WireError>> {
        let (blob, offsets) = gather(msgs);
        let n = msgs.len();
        let sk = self.record();
        let gpu = params.gpu();
        let mut status = vec![0u8; n + 1];
        gpu.check(unsafe { act_node_verify_spend_cbor_batch(gpu.0, n, sk.as_ptr(), blob.as_ptr(), offsets.as_ptr(), status.as_mut_ptr(), std::ptr::null_mut()) });
        status[..n].iter().map(|&s| if s == 0 { Ok(()) } else { Err(status_to_wire_error(s)) }).collect()
    }
}

// ---- the redemption step (examples/act.rs:62-73) ------------------------------------------------------------------------------
/// The double-spend database the crate leaves to the caller (src/lib.rs:741-745; `HashSet<Scalar>` in src/tests.rs:29-50 and
/// examples/act.rs:10-30) as a hash set in the HBM of the node's GPUs: one set per GPU, a nullifier owned by exactly one of them.
pub struct GpuNullifierStore(*mut ActNodeNullifierSet);
// SAFETY: every entry point that takes the set locks it (csrc/node.cpp act_node_nullifier_check_and_insert_batch); destroy runs from Drop.
unsafe impl Send for GpuNullifierStore {}
unsafe impl Sync for GpuNullifierStore {}
impl Drop for GpuNullifierStore {
    fn drop(&mut self) {
        unsafe { act_node_nullifier_set_destroy(self.0) }
    }
}
impl GpuNullifierStore {
    /// `capacity_per_device`: nullifiers each GPU's share must be able to hold (32 bytes + 4 per slot, load factor 1/2).
    pub fn new(capacity_per_device: usize) -> Self {
        let devices = devices_from_env();
        let mut set = std::ptr::null_mut();
        // salt = null: the routing and slot hashes are keyed with 16 bytes from the OS -- clients choose their nullifiers
        let rc = unsafe { act_node_nullifier_set_create(devices.as_ptr(), devices.len() as c_int, capacity_per_device, std::ptr::null(), &mut set) };
        if rc != 0 {
            let msg = if set.is_null() { String::new() } else { unsafe { CStr::from_ptr(act_node_nullifier_set_last_error(set)) }.to_string_lossy().into_owned() };
            panic!("act_node_nullifier_set_create failed ({rc}): {msg}");
        }
        GpuNullifierStore(set)
    }
    pub fn len(&self) -> usize {
        unsafe { act_node_nullifier_set_len(self.0) }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
}

/// Outcome of a redeem call.  `engine_failure` is set when a GPU or the store failed AFTER verification: every lane still has its
/// decision (`WireError::NullifierUndetermined` / `RecordedUnsigned` name the lanes that were left over) -- nothing is lost, nothing
/// may be blindly resubmitted (include/act_mi355x.h, act_redeem_batch).
pub struct Redeemed<T> {
    pub lanes: Vec<Result<T, WireError>>,
    pub engine_failure: Option<String>,
}

impl PrivateKey {
    /// The server loop of examples/act.rs:62-73 over a batch of wire messages --
    ///     from_cbor -> refund's checks -> `if store.is_used(k) { DoubleSpendError } else { store.mark_used(k) }` -> sign -> to_cbor
    /// -- with the meaning of that loop run in lane order (a nullifier repeated inside the batch is a double spend from its second
    /// occurrence on; a proof that does not verify cannot burn a nullifier) and `rng` drawn for exactly the lanes that are signed.
    pub fn redeem_cbor_batch(&self, params: &Params, store: &GpuNullifierStore, msgs: &[&[u8]], mut rng: impl CryptoRngCore) -> Redeemed<Vec<u8>> {
        let (blob, offsets) = gather(msgs);
        let n = msgs.len();
        let sk = self.record();
        let gpu = params.gpu();
        let src = rng_source(&mut rng);
        let (mut out, mut status) = (vec![0u8; REFUND_CBOR_BYTES * n + 1], vec![0u8; n + 1]);
        let rc = unsafe {
            act_node_redeem_cbor_batch(gpu.0, store.0, n, sk.as_ptr(), blob.as_ptr(), offsets.as_ptr(), &src as *const ActRngSource as *const u8,
                                       ACT_RNG_CALLBACK, out.as_mut_ptr(), status.as_mut_ptr())
        };
        let engine_failure = if rc != 0 { Some(unsafe { CStr::from_ptr(act_node_last_error(gpu.0)) }.to_string_lossy().into_owned()) } else { None };
        Redeemed { lanes: refund_messages(&out, &status[..n]), engine_failure }
    }
    /// The same over `SpendProof`s (marshalled on all cores: 130 `compress()` per proof).
    pub fn redeem_batch(&self, params: &Params, store: &GpuNullifierStore, proofs: &[SpendProof], mut rng: impl CryptoRngCore) -> Redeemed<Refund> {
        let n = proofs.len();
        let rec = marshal(proofs, PROOF_BYTES, |p, out| p.write_record(out));
        let sk = self.record();
        let gpu = params.gpu();
        let src = rng_source(&mut rng);
        let (mut out, mut status) = (vec![0u8; 128 * n + 1], vec![0u8; n + 1]);
        let rc = unsafe {
            act_node_redeem_batch(gpu.0, store.0, n, sk.as_ptr(), rec.as_ptr(), &src as *const ActRngSource as *const u8, ACT_RNG_CALLBACK,
                                  out.as_mut_ptr(), status.as_mut_ptr())
        };
        let engine_failure = if rc != 0 { Some(unsafe { CStr::from_ptr(act_node_last_error(gpu.0)) }.to_string_lossy().into_owned()) } else { None };
        let lanes = (0..n).map(|i| if status[i] == 0 { Ok(Refund::from_record(&out[128 * i..128 * i + 128])) } else { Err(status_to_wire_error(status[i])) }).collect();
        Redeemed { lanes, engine_failure }
    }
}

/// What the node dispatcher knows about its GPUs (load balance: include/act_mi355x.h "Load balance"): per device its weight
/// (relative speed, mean 1) and the lanes / seconds / calls of the most recent cut call.
pub fn gpu_device_stats(params: &Params) -> Vec<(f64, u64, f64, u64)> {
    let gpu = params.gpu();
    let mut out = Vec::new();
    let mut k = 0;
    loop {
        let (mut w, mut lanes, mut secs, mut calls) = (0f64, 0u64, 0f64, 0u64);
        if unsafe { act_node_device_stats(gpu.0, k, &mut w, &mut lanes, &mut secs, &mut calls) } != 0 {
            break;
        }
        out.push((w, lanes, secs, calls));
        k += 1;
    }
    out
}

// ---- the kept single-call signatures: batches of one ------------------------------------------------------------------
// In src/lib.rs the six existing method bodies become `#[cfg(not(feature = "mi355x"))]`; these take their place otherwise: with the
// feature on, nothing of the protocol runs on the CPU.  One item per call is latency, not throughput, and a GPU's latency is a
// dependent chain on a few lanes (docs/history/profiles/r04_single_item_latency.txt; the C port on one core of the same box in brackets):
//     request 0.59 ms (0.05)    issue 2.7 (0.25)    PreIssuance::to_credit_token 2.3 (0.21)
//     prove_spend 3.1 (15.5)    refund 3.5 (17.3)   PreRefund::to_credit_token 3.0 (5.0)
// The `*_batch` siblings are where the rates are (24 M issues/s, 112 M requests/s, 520 k refunds/s).
#[cfg(feature = "mi355x")]
impl PreIssuance {
    pub fn request(&self, params: &Params, rng: impl CryptoRngCore) -> IssuanceRequest {
        // src/lib.rs:463
        Self::request_batch(std::slice::from_ref(self), params, rng).pop().unwrap()
    }
    pub fn to_credit_token(&self, params: &Params, public: &PublicKey, request: &IssuanceRequest, response: &IssuanceResponse)
        -> Result<CreditToken, Error> {
        // src/lib.rs:528-534
        Self::to_credit_token_batch(std::slice::from_ref(self), params, public, std::slice::from_ref(request), std::slice::from_ref(response)).pop().unwrap()
    }
}
#[cfg(feature = "mi355x")]
impl PrivateKey {
    pub fn issue(&self, params: &Params, request: &IssuanceRequest, c: Scalar, rng: impl CryptoRngCore) -> Result<IssuanceResponse, Error> {
        // src/lib.rs:621-627
        self.issue_batch(params, std::slice::from_ref(request), &[c], rng).pop().unwrap()
    }
    pub fn refund(&self, params: &Params, spend_proof: &SpendProof, rng: impl CryptoRngCore) -> Result<Refund, Error> {
        // src/lib.rs:781-786
        self.refund_batch(params, std::slice::from_ref(spend_proof), rng).pop().unwrap()
    }
}
#[cfg(feature = "mi355x")]
impl CreditToken {
    pub fn prove_spend(&self, params: &Params, s: Scalar, rng: impl CryptoRngCore) -> (SpendProof, PreRefund) {
        // src/lib.rs:972-977
        Self::prove_spend_batch(std::slice::from_ref(self), params, &[s], rng).pop().unwrap()
    }
}
#[cfg(feature = "mi355x")]
impl PreRefund {
    pub fn to_credit_token(&self, params: &Params, spend_proof: &SpendProof, refund: &Refund, public_key: &PublicKey) -> Result<CreditToken, Error> {
        // src/lib.rs:1217-1223
        Self::to_credit_token_batch(std::slice::from_ref(self), params, std::slice::from_ref(spend_proof), std::slice::from_ref(refund), public_key).pop().unwrap()
    }
}

// ---- one library call per item, with nonce bytes the CALLER has already drawn ----------------------------------------------------
// NOT part of the drop-in surface, and deliberately not methods that take a generator.  `PrivateKey::issue` / `refund` above keep the
// crate's contract exactly -- e and alpha are drawn only after the checks have passed (src/lib.rs:638-643, 842-846), so a rejected
// item leaves the caller's generator untouched -- and pay for it with two library calls (check, then sign).  The functions below are
// ONE call: the signature is computed BESIDE the check (issue 1.25 ms instead of ~2.0, refund 1.95 instead of ~2.8, a wire-level
// redemption 2.1 instead of 3.2), which is only possible if the 128 bytes that seed (e, alpha) exist before the verdict does.  They
// therefore take the BYTES, not a generator: there is no generator state here that could disagree with the crate's, and what
// happens to the bytes of a rejected item is the caller's explicit decision (they were read by the GPU and wiped there; never use
// them again).  A caller whose generator is the operating system's loses nothing; a caller that replays a deterministic stream and
// must match the crate's draws byte for byte stays with the methods above.
#[cfg(feature = "mi355x")]
pub mod predrawn {
    use super::*;

    /// 128 bytes from `rng`, drawn as two `Scalar::random` would draw them (two 64-byte `fill_bytes`).
    pub fn nonce_bytes(mut rng: impl CryptoRngCore) -> [u8; 128] {
        let v = draw(&mut rng, 2);
        let mut out = [0u8; 128];
        out.copy_from_slice(&v);
        out
    }
    /// `sk.refund(params, proof, rng)` for an rng whose next 128 bytes are `nonces` -- if the proof verifies.  If it does not, the
    /// result is the same `Err`, and `nonces` have been spent on nothing.
    pub fn refund(sk: &PrivateKey, params: &Params, spend_proof: &SpendProof, nonces: &[u8; 128]) -> Result<Refund, Error> {
        let mut rec = Vec::with_capacity(PROOF_BYTES);
        spend_proof.write_record(&mut rec);
        let (skr, gpu) = (sk.record(), params.gpu());
        let (mut out, mut status) = ([0u8; 128], [0u8; 1]);
        gpu.check(unsafe { act_node_refund_batch(gpu.0, 1, skr.as_ptr(), rec.as_ptr(), nonces.as_ptr(), ACT_RNG_SEQUENTIAL, out.as_mut_ptr(), status.as_mut_ptr()) });
        if status[0] == 0 { Ok(Refund::from_record(&out)) } else { Err(status_to_error(status[0])) }
    }
    /// `sk.issue(params, request, c, rng)` for an rng whose next 128 bytes are `nonces`, the signature beside the request's proof of
    /// knowledge (src/lib.rs:629-660).
    pub fn issue(sk: &PrivateKey, params: &Params, request: &IssuanceRequest, c: Scalar, nonces: &[u8; 128]) -> Result<IssuanceResponse, Error> {
        let (mut req, mut cb) = (Vec::with_capacity(128), Vec::with_capacity(32));
        request.write_record(&mut req);
        put_s(&mut cb, &c);
        let (skr, gpu) = (sk.record(), params.gpu());
        let (mut out, mut status) = ([0u8; 160], [0u8; 1]);
        gpu.check(unsafe { act_node_issue_batch(gpu.0, 1, skr.as_ptr(), req.as_ptr(), cb.as_ptr(), nonces.as_ptr(), ACT_RNG_SEQUENTIAL, out.as_mut_ptr(), status.as_mut_ptr()) });
        if status[0] == 0 { Ok(IssuanceResponse::from_record(&out)) } else { Err(status_to_error(status[0])) }
    }
    /// One CBOR `SpendProof` message in, one CBOR `Refund` message out: unframing, verification and the signature beside it.
    pub fn refund_cbor(sk: &PrivateKey, params: &Params, msg: &[u8], nonces: &[u8; 128]) -> Result<Vec<u8>, WireError> {
        let offsets = [0u64, msg.len() as u64];
        let (skr, gpu) = (sk.record(), params.gpu());
        let (mut out, mut status) = (vec![0u8; REFUND_CBOR_BYTES + 1], vec![0u8; 2]);
        gpu.check(unsafe {
            act_node_refund_cbor_batch(gpu.0, 1, skr.as_ptr(), msg.as_ptr(), offsets.as_ptr(), nonces.as_ptr(), ACT_RNG_SEQUENTIAL, out.as_mut_ptr(), status.as_mut_ptr())
        });
        refund_messages(&out, &status[..1]).pop().unwrap()
    }
    /// The whole redemption step for one message: the refund is computed first (one call), THEN the store decides whether it is handed
    /// out; a rejected message or a double spend has spent `nonces` on nothing.
    pub fn redeem_cbor(sk: &PrivateKey, params: &Params, store: &GpuNullifierStore, msg: &[u8], nonces: &[u8; 128]) -> Redeemed<Vec<u8>> {
        let offsets = [0u64, msg.len() as u64];
        let (skr, gpu) = (sk.record(), params.gpu());
        let (mut out, mut status) = (vec![0u8; REFUND_CBOR_BYTES + 1], vec![0u8; 2]);
        let rc = unsafe {
            act_node_redeem_cbor_batch(gpu.0, store.0, 1, skr.as_ptr(), msg.as_ptr(), offsets.as_ptr(), nonces.as_ptr(), ACT_RNG_SEQUENTIAL, out.as_mut_ptr(), status.as_mut_ptr())
        };
        let engine_failure = if rc != 0 { Some(unsafe { CStr::from_ptr(act_node_last_error(gpu.0)) }.to_string_lossy().into_owned()) } else { None };
        Redeemed { lanes: refund_messages(&out, &status[..1]), engine_failure }
    }
}
