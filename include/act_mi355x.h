/* act_mi355x.h — C ABI of libact_mi355x.so: the MI355X (gfx950) batch engine for the sigma-protocol
 * hot path of anonymous-credit-tokens v0.2.1.
 *
 * The reference crate has no FFI of its own; the drop-in boundary is its public Rust API
 * (/root/reference/src/lib.rs).  Each *_batch entry point below is what a `mod mi355x` inside the
 * crate binds for the corresponding method (INTEGRATION.md shows the Rust side), lane i of a batch
 * being one call of the method:
 *
 *   act_params_new                      <-> Params::new                  src/lib.rs:291-315
 *   act_request_batch                   <-> PreIssuance::request         src/lib.rs:463-487
 *   act_issue_batch                     <-> PrivateKey::issue            src/lib.rs:621-663
 *   act_prove_spend_batch               <-> CreditToken::prove_spend     src/lib.rs:972-1152
 *   act_refund_batch                    <-> PrivateKey::refund           src/lib.rs:781-869
 *   act_verify_spend_batch              <-> PrivateKey::refund up to the challenge check (:787-844)
 *   act_issuance_to_credit_token_batch  <-> PreIssuance::to_credit_token src/lib.rs:528-562
 *   act_refund_to_credit_token_batch    <-> PreRefund::to_credit_token   src/lib.rs:1217-1253
 *
 * Records are the crate's structs as consecutive 32-byte fields in CBOR key order
 * (src/cbor.rs:105-110, 163-169, 250-268, 422-427, 477-480, 546-549, 596-602, 656-660) with the CBOR
 * framing stripped; scalars little-endian (Scalar::as_bytes), points compressed Ristretto:
 *   PrivateKey        x | w                                                              64 B
 *   PreIssuance       r | k                                                              64 B
 *   IssuanceRequest   K | gamma | k_bar | r_bar                                         128 B
 *   IssuanceResponse  A | e | gamma | z | c                                             160 B
 *   CreditToken       a | e | k | r | c                                                 160 B
 *   SpendProof        k | s | A' | B_bar | Com[L] | gamma | e_bar | r2_bar | r3_bar | c_bar | r_bar |
 *                     w00 | w01 | gamma0[L] | z[L][2] | k_bar | s_bar                 32*(14+4L) B
 *   PreRefund         r | k | m                                                          96 B
 *   Refund            A* | e | gamma | z                                               128 B
 *
 * RNG: where the crate takes `impl CryptoRngCore`, the ABI takes the bytes that generator would
 * have produced; every Scalar::random is one 64-byte draw reduced mod l (draw order per function:
 * src/lib.rs:468-469, 643/649, 846/852, 978-1058).  issue/refund draw only for accepted lanes
 * (src/lib.rs:638-643, 842-846): ACT_RNG_PER_LANE gives lane i the slice rng[128*i..], and
 * ACT_RNG_SEQUENTIAL hands consecutive 128-byte slices to the accepted lanes in lane order — the
 * byte-for-byte equivalent of a sequential loop sharing one generator.
 *
 * Errors: the function result reports infrastructure failures only.  Per-lane results are
 * status[i] = 0 for Ok, else 1 + the discriminant of the crate's `Error` (src/lib.rs:102-112);
 * 255 = a point field that is not a canonical Ristretto encoding (the crate rejects those while
 * decoding, src/cbor.rs:62-77, before any of these methods can run).  The output record of a
 * failed lane is all zero.  Scalar fields are reduced mod l on input (src/cbor.rs:85).
 *
 * Memory: every bulk pointer of one call is either host memory (ACT_MEM_HOST) or memory of the
 * context's GPU (ACT_MEM_DEVICE); `sk` is always host memory.  The engine works on its own HIP streams:
 * device-memory inputs must be complete (the producing stream synchronised) before a call, and outputs are
 * complete when the call returns.  A context is bound to one GPU and owns its streams and workspace; every batch entry
 * point takes the context's lock, so a context shared between host threads serves them one at a time; different contexts
 * (e.g. one per GPU) run concurrently (batches shard across GPUs with no collective).  Contexts of one process on the
 * same GPU with the same Params share one set of fixed-base tables.
 * Host memory may be pageable or pinned.  Spend-proof RECORDS (public inputs) that lie in pinned memory (hipHostMalloc /
 * hipHostRegister under the context's device) are read by the kernels in place over the link -- no staging copy in front of the
 * first kernel; a caller must therefore not write to them while the call runs.  Everything else (wire bytes included), and pageable
 * memory, is staged through the context's own buffers, which are wiped when the call ends.
 * There is no CPU fallback: without a HIP device every entry point fails.
 */
#ifndef ACT_MI355X_H
#define ACT_MI355X_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ======== SURVEY.md 8 rows a7 + b: result codes, memory kinds, rng conventions, per-lane statuses (records: the comment above) ==== */
#define ACT_OK 0
#define ACT_ERR_ARG 1        /* null pointer, bad L, bad mode */
#define ACT_ERR_HIP 2        /* a HIP runtime call failed; act_last_error() has the text */
#define ACT_ERR_PARAMS 3     /* h1/h2/h3 or the public key is not a canonical Ristretto encoding */
#define ACT_ERR_NO_DEVICE 4
#define ACT_ERR_RNG 5        /* ACT_RNG_CALLBACK: the caller's draw() reported a failure; nothing was signed with the bytes */

#define ACT_MEM_HOST 0
#define ACT_MEM_DEVICE 1

#define ACT_RNG_PER_LANE 0
#define ACT_RNG_SEQUENTIAL 1
/* The crate's `impl CryptoRngCore` itself, for the calls that say they take it (the wire-level and redeem entry points below): `rng`
 * points to an act_rng_source instead of bytes.  The library calls draw(rng_ctx, dst, 128 * k) ONCE per call, from the calling
 * thread, after every verdict is known, k = the number of lanes that will be signed -- the generator is advanced by exactly what a
 * sequential loop over refund() would have drawn (src/lib.rs:842-852: e, alpha only after the checks), and the slices go to the
 * signed lanes in lane order like ACT_RNG_SEQUENTIAL.  The drawn bytes are wiped before the call returns.
 * draw returns 0 when it has written all `len` bytes and non-zero when it could not (generator exhausted, an exception in a
 * language binding's trampoline ...): the call then fails with ACT_ERR_RNG and signs NOTHING -- a signature made with e = alpha = 0
 * would give the issuer's key away (z = gamma * x).  Lanes that were to be signed are reported as after a failed signature step
 * (redeem: ACT_STATUS_RECORDED_UNSIGNED, their nullifiers are recorded; the other calls: no output records). */
#define ACT_RNG_CALLBACK 2
typedef int (*act_rng_draw_fn)(void *rng_ctx, uint8_t *dst, size_t len);
typedef struct act_rng_source { act_rng_draw_fn draw; void *rng_ctx; } act_rng_source;

#define ACT_TRANSCRIPT_HOST 0    /* BLAKE3 of every transcript on host threads (src/transcript.rs stays on the host) */
#define ACT_TRANSCRIPT_DEVICE 1  /* the same bytes hashed by the device BLAKE3 kernel; no D2H/H2D inside a batch */

/* per-lane status = 1 + discriminant of reference `Error`, src/lib.rs:102-112 */
#define ACT_STATUS_OK 0
#define ACT_STATUS_INVALID_ISSUANCE_REQUEST_PROOF 1
#define ACT_STATUS_INVALID_ISSUANCE_RESPONSE_PROOF 2
#define ACT_STATUS_DOUBLE_SPEND 3
#define ACT_STATUS_INVALID_REFUND_PROOF 4
#define ACT_STATUS_INVALID_REFUND_RESPONSE_PROOF 5
#define ACT_STATUS_IDENTITY_POINT 6
#define ACT_STATUS_INVALID_CLIENT_SPEND_PROOF 7
#define ACT_STATUS_AMOUNT_TOO_BIG 8
#define ACT_STATUS_SCALAR_OUT_OF_RANGE 9
#define ACT_STATUS_UNDECODABLE 255
/* act_redeem_batch / act_node_redeem_batch only, and only together with a non-zero function result (see there) */
#define ACT_STATUS_NULLIFIER_UNDETERMINED 252   /* verified; the nullifier step could not answer: NOT recorded, NOT signed -- resubmit */
#define ACT_STATUS_RECORDED_UNSIGNED 251        /* verified, nullifier recorded, the signature step failed: the refund is owed -- sign, never redeem again */

/* ======== row a6: Params::new / Params::random ================================================================================== */
typedef struct act_ctx act_ctx;

/* Params::new (src/lib.rs:291-315): out_h = enc(h1) | enc(h2) | enc(h3).  Runs on `device`. */
int act_params_new(int device, const char *organization, const char *service, const char *deployment_id,
                   const char *version, uint8_t out_h[96]);
/* Params::random (src/lib.rs:259-265): rng = 192 bytes (three RistrettoPoint::random draws). */
int act_params_random(int device, const uint8_t rng[192], uint8_t out_h[96]);

/* ======== row b: the context (one GPU), row a5: where the transcript is hashed, and the context's knobs ========================= */
/* A context = Params (h1,h2,h3 with their device-resident fixed-base tables, cf. the three
 * RistrettoBasepointTables of src/lib.rs:222-229) + the range-proof width L (src/lib.rs:116; 128 is the
 * crate's value, 1..128 accepted) + one GPU.  max_batch bounds the records per internal launch and thereby the workspace.
 * What a context costs in HBM at L = 128:
 *     workspace      about 200 KB per record of max_batch per pipeline slot, two slots:  max_batch 65536 -> 27 GB, 16384 -> 6.8 GB,
 *                    4096 -> 1.7 GB; staging buffers of host-memory callers grow on demand on top (up to ~7 GB for a 65536-lane
 *                    prover chunk)
 *     tables         0.5 GB (16-bit windows for g, h1, h2, h3 + the matrix-core images), shared by all contexts of the process on
 *                    that GPU with the same Params
 * and nothing else, whatever the device has free: wider fixed-base windows are asked for (act_ctx_set_fixed_base_bits below).
 * 0 = default = 65536, from which size on the throughput of every entry point is flat; 16384 costs about 5 % of the verify rate and
 * two thirds of the issue/request rate.  Batches of any length are accepted and processed in such chunks.  max_batch > 2^22 is
 * refused (ACT_ERR_ARG).  On failure *out still receives a context whose only use is act_last_error() and act_ctx_destroy(). */
int act_ctx_create(const uint8_t h[96], int L, int device, size_t max_batch, act_ctx **out);
void act_ctx_destroy(act_ctx *ctx);
int act_ctx_set_transcript_mode(act_ctx *ctx, int mode);    /* default ACT_TRANSCRIPT_HOST */
/* Host BLAKE3 workers of the host-transcript mode.  All contexts of a process share ONE pool of act_host_usable_cpus() workers,
 * started on the first hashing call (never, if only device transcripts are used).  A call takes its fair share of it -- pool size /
 * (contexts hashing at that moment) -- so the 8 contexts of a node handle get an eighth each when they all hash and the whole pool
 * when they hash alone; nthreads > 0 caps the share of this context, 0 = no cap.  ACT_NUMA=1 pins worker k to the k-th CPU of the
 * process's affinity mask. */
int act_ctx_set_host_threads(act_ctx *ctx, int nthreads);
/* Telemetry of the host-transcript mode on this context since the last reset: seconds its calling thread spent waiting for
 * transcript pieces to arrive from the device, seconds it spent hashing them (with its share of the pool), bytes hashed.  hash_s over
 * the wall time of the calls = how close the host side is to being the bottleneck (bench.py reports it per rank at N > 1). */
int act_ctx_host_hash_stats(act_ctx *ctx, double *wait_s, double *hash_s, uint64_t *bytes, int reset);
int act_host_usable_cpus(void);                             /* CPUs the process may use (affinity mask, cgroup quota) = pool size */
/* The pool's hashing entry point (what the host-transcript mode calls; pure host code, usable and tested without a GPU):
 * xof[16*i ..] = first 64 XOF bytes of BLAKE3(msgs + i*stride, len), i < n; max_threads as nthreads above. */
void act_host_hash_many(const uint8_t *msgs, size_t stride, uint32_t len, size_t n, int max_threads, uint32_t *xof);
/* fn(ctx, i0, i1) over [0, n) in items of `grain` indices on the same workers, in any order, the caller among them; returns when
 * all have run.  (What act_node_nullifier_check_and_insert_batch routes keys to their owning GPU with; max_threads as above.) */
typedef void (*act_host_range_fn)(void *ctx, size_t i0, size_t i1);
void act_host_parallel_for(size_t n, size_t grain, int max_threads, act_host_range_fn fn, void *ctx);
/* diagnostics: hashing calls served, worker threads ever created by this process (constant after the first call), pool size */
void act_host_pool_stats(uint64_t *jobs, uint64_t *threads_created, int *pool_size);
/* A context pipelines two chunks on two HIP streams.  The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware
 * queues per device and priority (default 4): in a process with many streams both may land on one queue and the chunks then run
 * one after the other (measured: 415 k instead of 466 k verifies/s from host memory).  act_ctx_create measures this (two idle
 * wavefronts, 0.3 ms) and, if they share a queue, moves the second stream to another priority class, which has its own queues.
 * 1 = the streams run side by side, 0 = they still share a queue (set GPU_MAX_HW_QUEUES=8 in the embedding process before HIP
 * initialises: INTEGRATION.md), -1 = not measured.  The library never edits the process environment and, apart from the diagnostics
 * ACT_TRACE / ACT_TIMELINE_FILE and the worker pinning ACT_NUMA, never reads it. */
int act_ctx_streams_overlap(const act_ctx *ctx);
/* chunks in flight per call: 2 (default; chunk i+1's kernels overlap chunk i's low-occupancy head / tail kernels and, in
 * host-transcript mode, its host hashing) or 1 (strictly one after the other: profiling runs whose per-kernel durations
 * must not overlap) */
int act_ctx_set_pipeline_depth(act_ctx *ctx, int depth);
/* The crate's entry points take ONE proof per call (src/lib.rs:781-786).  Verify / refund calls of at most `n` proofs (default
 * 8192, and 16384 while the context hashes its transcripts on the device; 0 = never) run the small-batch schedule: one chunk, the per-proof kernels next to the range kernel on streams of their own (five in all) instead of
 * in front of and behind it -- the same lane code and bytes, 2-3 x shorter for one proof and ~1.5 x the rate at 4 096 proofs; larger
 * calls pipeline 65 536-proof chunks as before (one lane per proof is the faster form once a launch fills the chip).
 * Threads with a context each may make such calls at the same time; two of them run on a device at once, the others wait their
 * turn inside the call (more active streams than that and the driver time-slices the process's queues: INTEGRATION.md). */
int act_ctx_set_small_batch_max(act_ctx *ctx, size_t n);
/* Tiny calls -- at most 64 lanes, the crate's own one-item call shape (src/lib.rs:463, 528, 621, 781, 1217).  Their answer time is a
 * dependent chain plus fixed cost, so act_request_batch, act_issue_batch (ACT_RNG_PER_LANE, or one lane), the signing half of every
 * issue / refund call and the two to_credit_token calls run such a call as ONE kernel: the independent pieces of the method on
 * separate workgroups (a signature's two variable-base products cut into quarters), the block that arrives last assembling the
 * transcript, hashing it -- these transcripts are a single BLAKE3 chunk; the routine is the one ACT_TRANSCRIPT_DEVICE runs, so the
 * bytes are those of either transcript mode -- and finishing the record; inputs cross PCIe in one copy from a pinned buffer, which
 * the kernel zeroes itself.  One item, MI355X: request 0.27 ms (was 0.58), issue 1.15 (2.7), PreIssuance::to_credit_token 1.1 (2.3),
 * PreRefund::to_credit_token 1.5 (3.0), refund 2.0 (3.7), prove_spend 2.0 (3.2): profiles/r05_single_item_latency.txt.
 * act_ctx_set_tiny_calls(ctx, 0) keeps the multi-launch paths, whose transcripts follow
 * the context's transcript mode -- a deployment that wants every hash on the host, whatever the call size (same bytes either way;
 * tests/test_gpu_tiny.py compares). */
int act_ctx_set_tiny_calls(act_ctx *ctx, int on);           /* default 1 */
/* Secrets and memory addresses.  The reference is constant-time in its table accesses (`subtle`, src/lib.rs:98, 1025-1118;
 * dalek's table scans).  libact_mi355x.so (the default build, for which this returns 1) matches that for EVERY secret: the issuer's
 * private key and signing nonces, and the client's tokens, blinding factors and prover rng, never select a memory address --
 * variable-base products run a register-only chain that executes every digit addition, and fixed-base products pick their table
 * entries on the matrix cores: selected = Table x onehot(digits), one v_mfma_i32_32x32x32_i8 per 32 bytes x 32 lanes x 32 entries,
 * so a window holds 64 entries (37 per product) with no entry ever addressed by a digit (csrc/msm.h fixed_base_acc_mf).
 * libact_mi355x_fast.so (make fast; returns 0) differs only for the CLIENT's secrets (act_prove_spend_*, act_request_batch): they go
 * through scalar-addressed 16- / 24-bit tables and Pippenger buckets -- the instruction stream still does not depend on them, the
 * memory-access pattern does.  Same bytes either way.  Measured on one MI355X (docs/history/profiles/r04_*_other_configs_1gpu*.json): default
 * vs fast build prove_spend 0.48 x (1.16 M/s vs 2.40 M/s; round 3's masked scan: 0.26 x), request 0.55 x, a whole lifecycle
 * 0.85 x; every issuer-side call -- verify, refund, issue, redeem -- is the same code in both.  A client that owns its GPU may load
 * the fast build; an issuer gains nothing from it. */
int act_build_has_ct_secret_tables(void);
/* Window width in bits of the fixed-base table of base 0..3 = g, h1, h2, h3 in this context (a product costs ceil(253 / bits) table
 * additions).  16 after act_ctx_create (128 MiB per base), always.  act_ctx_set_fixed_base_bits(ctx, base, bits), bits in 4..24,
 * builds the table of that width now (or shares it with the process's other contexts on the device) and releases the old one; the
 * range kernel's bases are h1 (1) and h3 (3): 24 bits on both is +47 GB (23.6 GB each, built in about 2 s) for +3 % verifies/s --
 * worth it on a GPU that serves nothing else, and only the caller knows that.  Refused (ACT_ERR_HIP, width unchanged) when the
 * device has not the table's size + 16 GB free.  Not while another thread has a call on the context in flight with results that
 * matter for timing: it takes the context's lock like a batch call. */
int act_ctx_fixed_base_bits(const act_ctx *ctx, int base);
int act_ctx_set_fixed_base_bits(act_ctx *ctx, int base, int bits);
/* Text of the CALLING THREAD's last failure on this handle (of the handle's most recent failure if this thread has had none): a
 * handle is shared between threads, and a caller whose small call was merged into another thread's launch gets that launch's text.
 * Every *_last_error function copies the text into a buffer of the calling thread, valid until that thread asks again. */
const char *act_last_error(const act_ctx *ctx);
size_t act_spend_proof_bytes(const act_ctx *ctx);           /* 32*(14+4L) */
size_t act_prove_rng_bytes(const act_ctx *ctx);             /* 64*(4L+12) */
size_t act_spend_transcript_bytes(const act_ctx *ctx);      /* pre-image of the "spend" challenge */

/* ======== rows a1 - a4 and f1: request, issue, prove_spend, refund (and its verification half), the two client verifiers -- lane i = one call of the method ==== */
/* PrivateKey::random (src/lib.rs:188-194): rng 64 B -> x | w.  PreIssuance::random (:432-437): rng 128 B -> r | k. */
int act_private_key_random(act_ctx *ctx, const uint8_t rng[64], uint8_t out_sk[64]);
int act_pre_issuance_random_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *rng, uint8_t *out_pre);

/* PreIssuance::request: pre n*64, rng n*128 -> req n*128 */
int act_request_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *pre, const uint8_t *rng, uint8_t *out_req);
/* PrivateKey::issue: req n*128, c n*32, rng (n or #accepted)*128 -> resp n*160, status n.  The rng argument is BYTES: this call
 * never touches a generator, so it cannot advance one for a rejected item.  ACT_RNG_SEQUENTIAL with n == 1: the buffer must hold its
 * 128 bytes WHATEVER the verdict -- a one-lane call computes the signature beside the check and reads the slice first (the bytes of
 * a rejected lane are wiped with the call and go nowhere).  A caller that draws from a generator and must leave it where the crate
 * leaves it (src/lib.rs:638-643: e, alpha only after the check) uses the two halves below, as the Rust binding's `issue` does; the
 * binding offers the one-call form only as a free function over explicit bytes (rust/src/mi355x.rs `predrawn::issue`). */
int act_issue_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *req, const uint8_t *c,
                    const uint8_t *rng, int rng_mode, uint8_t *out_resp, uint8_t *status);
/* PreIssuance::to_credit_token: pre n*64, w 32 (host), req n*128, resp n*160 -> token n*160, status n */
int act_issuance_to_credit_token_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *pre, const uint8_t w[32],
                                       const uint8_t *req, const uint8_t *resp, uint8_t *out_token, uint8_t *status);
/* CreditToken::prove_spend: token n*160, s n*32, rng n*act_prove_rng_bytes -> proof n*act_spend_proof_bytes, prerefund n*96 */
int act_prove_spend_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *token, const uint8_t *s, const uint8_t *rng,
                          uint8_t *out_proof, uint8_t *out_prerefund, uint8_t *status);
/* The same with the generator SEEDED instead of spelled out: lane i draws from the BLAKE3 XOF of seed | u64_le(first_lane + i),
 * read sequentially by every Scalar::random -- on the crate's side `prove_spend(params, s, XofRng(seed, lane))` with an RngCore over
 * blake3::Hasher::new().update(seed).update(&lane.to_le_bytes()).finalize_xof().  The 64 (4L + 12) = 33 536 rng bytes per proof are
 * expanded in HBM and never cross PCIe (streaming 2^16 lifecycles per call from host memory: 200 k -> see DESIGN.md section 6).  The
 * seed is a secret of the prover like the rng bytes are; one seed must never serve the same lane number twice. */
int act_prove_spend_seeded_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *token, const uint8_t *s, const uint8_t seed[32],
                                 uint64_t first_lane, uint8_t *out_proof, uint8_t *out_prerefund, uint8_t *status);
/* spend-proof verification only (src/lib.rs:787-844): proof -> status n; out_kprime (nullable) n*32 = enc(K') */
int act_verify_spend_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *proof,
                           uint8_t *status, uint8_t *out_kprime);
/* PrivateKey::refund: proof, rng (n or #accepted)*128 -> refund n*128, status n.  As for act_issue_batch: a one-lane ACT_RNG_SEQUENTIAL
 * call needs its 128 bytes present whatever the verdict (the signature is computed beside the verification); generator-exact callers
 * use act_verify_spend_batch + act_refund_sign_batch (src/lib.rs:842-846), or the wire-level calls with ACT_RNG_CALLBACK. */
int act_refund_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *proof, const uint8_t *rng,
                     int rng_mode, uint8_t *out_refund, uint8_t *status);
/* PreRefund::to_credit_token: prerefund n*96, proof, refund n*128, w 32 (host) -> token n*160, status n */
int act_refund_to_credit_token_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *prerefund, const uint8_t *proof,
                                     const uint8_t *refund, const uint8_t w[32], uint8_t *out_token, uint8_t *status);

/* The two halves of issue / refund as separate calls, for callers that must see every verdict before any rng is assigned
 * (the node dispatcher below; a caller with its own admission step between check and signature):
 *   act_issue_check_batch   the PoK check alone (src/lib.rs:629-640): req n*128 -> status n
 *   act_issue_sign_batch    the BBS signature (:643-660) for lanes with status_in == 0: X_A = g + c h1 + K from the request
 *   act_refund_sign_batch   the BBS signature (:846-868) for lanes with status_in == 0: X_A = g + K', kprime n*32 as returned
 *                           by act_verify_spend_batch
 * rng / rng_mode as in act_issue_batch; check + sign with the same rng gives the bytes of the one-call form. */
int act_issue_check_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *req, uint8_t *status);
int act_issue_sign_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *req, const uint8_t *c,
                         const uint8_t *status_in, const uint8_t *rng, int rng_mode, uint8_t *out_resp, uint8_t *status);
int act_refund_sign_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *kprime, const uint8_t *status_in,
                          const uint8_t *rng, int rng_mode, uint8_t *out_refund, uint8_t *status);

/* ======== row e: the GPUs of one node behind one handle (no collective) ========================================================= */
/* Node-level dispatch: the GPUs of one node behind one handle.  act_node_create builds one context per
 * entry of devices[] (the same device may be listed more than once: each entry is its own context, stream set and
 * workspace).  Every act_node_*_batch call cuts its batch into contiguous pieces (one per context, shard k = lanes
 * [n*k/N, n*(k+1)/N), for batches below 16384 lanes per GPU; see "Load balance" below for larger ones), runs every piece on a
 * context from that context's own host thread through the single-GPU entry point of the same
 * name, and writes outputs into the matching slices of the caller's arrays: no collective, no peer traffic.  All bulk
 * pointers are host memory.  ACT_RNG_SEQUENTIAL stays exact across shards -- the bytes of one sequential loop over one
 * generator (src/lib.rs:638-643, 842-846 draw only for accepted lanes): all shards are checked first, the host counts
 * the accepted lanes in front of every shard, then all shards sign from their offsets into the stream; refund carries
 * only enc(K') between the two phases.  ACT_RNG_PER_LANE needs no such barrier and is one pass.  A node handle (like a
 * context) may be shared between host threads: every *_batch call takes the handle's lock, so concurrent callers are
 * served one after the other.  act_node_create builds the contexts concurrently (one thread per entry); entries that name
 * the same device share that device's fixed-base tables. */
typedef struct act_node act_node;
int act_node_create(const uint8_t h[96], int L, const int *devices, int n_devices, size_t max_batch, act_node **out);
void act_node_destroy(act_node *node);
int act_node_device_count(const act_node *node);
act_ctx *act_node_ctx(act_node *node, int k);                  /* context k, e.g. for act_ctx_set_* / act_prof_* */
const char *act_node_last_error(const act_node *node);
int act_node_set_transcript_mode(act_node *node, int mode);
int act_node_set_host_threads(act_node *node, int per_gpu);    /* host BLAKE3 workers of every context */
int act_node_set_fixed_base_bits(act_node *node, int base, int bits);   /* act_ctx_set_fixed_base_bits on every context */
/* Load balance.  The GPUs of a node are not equally fast (clocks differ by several percent between devices and move with
 * temperature) and a call ends when its slowest GPU does, so the cut is not n/N: (1) every throughput-sized call -- at least
 * 16384 lanes per GPU -- measures what each context did with its piece and the next call cuts in proportion (weights: relative
 * speed, mean 1); (2) the last part of such a batch is not assigned in advance but handed out in small pieces (>= 4096 lanes) to
 * whichever GPU asks next, its size following the finish-time spread the previous calls showed: a sixteenth of the batch while
 * nothing is known, nothing once the GPUs finish together (a tail costs a few small calls).  Neither changes a byte of the output:
 * a lane's result depends on its inputs and its rng slice, never on the GPU that computed it.
 *   act_node_set_balance     weighted = 0 turns (1) off (equal cut); tail_64ths = -1 adaptive (default), 0 = no tail, k = k/64 of the batch
 *   act_node_device_stats    context k: its weight, and lanes / seconds / calls it was given in the most recent cut call
 *   act_node_balance_state   finish-time spread of the heads (running average, relative to their mean; < 0 = not measured yet) and the
 *                            fraction of the last call that went through the tail */
int act_node_set_balance(act_node *node, int weighted, int tail_64ths);
int act_node_device_stats(act_node *node, int k, double *weight, uint64_t *last_lanes, double *last_seconds, uint64_t *last_calls);
int act_node_balance_state(act_node *node, double *spread, double *tail_fraction);
int act_node_request_batch(act_node *node, size_t n, const uint8_t *pre, const uint8_t *rng, uint8_t *out_req);
int act_node_issue_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *req, const uint8_t *c, const uint8_t *rng,
                         int rng_mode, uint8_t *out_resp, uint8_t *status);
int act_node_issuance_to_credit_token_batch(act_node *node, size_t n, const uint8_t *pre, const uint8_t w[32], const uint8_t *req,
                                            const uint8_t *resp, uint8_t *out_token, uint8_t *status);
/* the halves on their own (cf. act_issue_check_batch ...): a caller that draws its rng between check and signature -- the
 * Rust binding advances the caller's generator by exactly 128 bytes per accepted lane, as the sequential loop would */
int act_node_issue_check_batch(act_node *node, size_t n, const uint8_t *req, uint8_t *status);
int act_node_issue_sign_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *req, const uint8_t *c,
                              const uint8_t *status_in, const uint8_t *rng, int rng_mode, uint8_t *out_resp, uint8_t *status);
int act_node_refund_sign_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *kprime, const uint8_t *status_in,
                               const uint8_t *rng, int rng_mode, uint8_t *out_refund, uint8_t *status);
int act_node_prove_spend_batch(act_node *node, size_t n, const uint8_t *token, const uint8_t *s, const uint8_t *rng,
                               uint8_t *out_proof, uint8_t *out_prerefund, uint8_t *status);
int act_node_prove_spend_seeded_batch(act_node *node, size_t n, const uint8_t *token, const uint8_t *s, const uint8_t seed[32],
                                      uint64_t first_lane, uint8_t *out_proof, uint8_t *out_prerefund, uint8_t *status);
int act_node_verify_spend_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *proof, uint8_t *status,
                                uint8_t *out_kprime);
int act_node_refund_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *proof, const uint8_t *rng, int rng_mode,
                          uint8_t *out_refund, uint8_t *status);
int act_node_refund_to_credit_token_batch(act_node *node, size_t n, const uint8_t *prerefund, const uint8_t *proof,
                                          const uint8_t *refund, const uint8_t w[32], uint8_t *out_token, uint8_t *status);

/* ======== row f3: wire bytes -- the batch CBOR codec, wire -> verdict, wire -> wire ============================================= */
/* Batch CBOR codec (src/cbor.rs: to_cbor / from_cbor of the nine wire and state types; deterministic RFC 8949
 * encoding, int-keyed maps, 32-byte byte strings).  `type` selects the struct; records are the raw layouts above.
 * Encoding writes n canonical messages of act_cbor_size(ctx, type) bytes each, back to back.  Decoding takes n
 * messages delimited by offsets[0..n] (byte offsets into `cbor`, host memory; NULL = n canonical-size messages back to
 * back) and accepts whatever ciborium accepts (any well-formed CBOR map; unknown keys ignored; the last duplicate
 * wins; bytes after the first item ignored).  Scalars come out reduced mod l (decode_scalar, src/cbor.rs:80-91).
 * status[i]: 0 ok, 1 malformed CBOR (CborError::Ciborium), 2 CborError::InvalidStructure, 3 CborError::InvalidValue
 * (a point that is not a canonical Ristretto encoding, src/cbor.rs:59-78); the record of a failed message is zero.
 * The code is from_cbor's code, also for a message that is wrong in several ways: the whole item is parsed first (1), then the
 * first failure in WIRE order of the map's entries decides (src/cbor.rs:276-388: an invalid point in front of a mis-shaped
 * field is 3, behind it 2; array elements, those of an over-long array included, are decoded before the length is checked;
 * a value a later duplicate key overwrites still has to decode), and missing fields come last (:390-407).
 * Canonical messages are framed / unframed on the GPU (one lane per 32-byte field); others take a host reader, which hands the
 * points whose position matters to a validation kernel. */
#define ACT_CBOR_ISSUANCE_REQUEST 1
#define ACT_CBOR_ISSUANCE_RESPONSE 2
#define ACT_CBOR_SPEND_PROOF 3
#define ACT_CBOR_REFUND 4
#define ACT_CBOR_PRIVATE_KEY 5
#define ACT_CBOR_PUBLIC_KEY 6
#define ACT_CBOR_PRE_ISSUANCE 7
#define ACT_CBOR_CREDIT_TOKEN 8
#define ACT_CBOR_PRE_REFUND 9
size_t act_cbor_size(const act_ctx *ctx, int type);
size_t act_cbor_record_bytes(const act_ctx *ctx, int type);
int act_cbor_encode_batch(act_ctx *ctx, int type, size_t n, int mem, const uint8_t *records, uint8_t *out_cbor);
int act_cbor_decode_batch(act_ctx *ctx, int type, size_t n, int mem, const uint8_t *cbor, const uint64_t *offsets,
                          uint8_t *out_records, uint8_t *status);

/* Wire bytes in, verdict out: SpendProof::from_cbor (src/cbor.rs:236-408) + PrivateKey::refund up to the challenge check
 * (src/lib.rs:787-844) as ONE pass over n CBOR messages (delimited like act_cbor_decode_batch's: offsets[0..n] in host memory, or
 * NULL for canonical-size messages back to back).  Every chunk is unframed on the GPU in front of its verification kernels, and
 * each of a proof's 130 points is decoded once -- act_cbor_decode_batch followed by act_verify_spend_batch decodes them twice and
 * moves the records through the caller's memory in between.  status[i]: 0 / 6 / 7 as act_verify_spend_batch;
 * 255 = a point that is not a canonical Ristretto encoding (from_cbor's CborError::InvalidValue);
 * ACT_STATUS_CBOR_MALFORMED = not well-formed CBOR (CborError::Ciborium); ACT_STATUS_CBOR_STRUCTURE = CborError::InvalidStructure
 * (not a map, missing field, wrong shape or length).  The status is the one from_cbor followed by refund gives, also for a message
 * that is wrong in several ways (the first failure in wire order, as described at act_cbor_decode_batch).  Non-canonical but valid
 * encodings take a host reader and a second, small verification call.  out_kprime (nullable) as in act_verify_spend_batch. */
#define ACT_STATUS_CBOR_MALFORMED 254
#define ACT_STATUS_CBOR_STRUCTURE 253
int act_verify_spend_cbor_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *cbor, const uint64_t *offsets,
                                uint8_t *status, uint8_t *out_kprime);
int act_node_verify_spend_cbor_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *cbor, const uint64_t *offsets,
                                     uint8_t *status, uint8_t *out_kprime);
/* The same, also returning what the rest of a redemption needs: out_nullifier (nullable) n*32 = the `k` field of every message as it
 * stood on the wire (unreduced; the nullifier sets reduce mod l themselves), zero for a message that did not parse. */
int act_verify_spend_cbor_keys_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *cbor, const uint64_t *offsets,
                                     uint8_t *status, uint8_t *out_kprime, uint8_t *out_nullifier);
int act_node_verify_spend_cbor_keys_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *cbor, const uint64_t *offsets,
                                          uint8_t *status, uint8_t *out_kprime, uint8_t *out_nullifier);

/* Wire bytes in, wire bytes out: what a server does with the bytes a client sent (SpendProof::from_cbor, src/cbor.rs:276-408;
 * PrivateKey::refund, src/lib.rs:781-869; Refund::to_cbor, src/cbor.rs:421-433) without a SpendProof, a RistrettoPoint or a Refund
 * ever existing on the host.  Message i in, message i out: out_refund_cbor holds n slots of act_cbor_size(ctx, ACT_CBOR_REFUND)
 * bytes (141), the canonical Refund message of an accepted lane, all zero for any other lane.
 *   act_refund_sign_cbor_batch   act_refund_sign_batch + framing: the BBS signature for lanes with status_in == 0 from enc(K')
 *   act_refund_cbor_batch        act_verify_spend_cbor_batch, then the above; status as act_verify_spend_cbor_batch
 * rng / rng_mode: ACT_RNG_PER_LANE, ACT_RNG_SEQUENTIAL (both halves see every verdict before a slice is assigned, so SEQUENTIAL is the
 * sequential loop's byte stream) or ACT_RNG_CALLBACK.  The node forms cut the batch over the GPUs like every act_node_* call. */
int act_refund_sign_cbor_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *kprime, const uint8_t *status_in,
                               const uint8_t *rng, int rng_mode, uint8_t *out_refund_cbor, uint8_t *status);
int act_refund_cbor_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *cbor, const uint64_t *offsets,
                          const uint8_t *rng, int rng_mode, uint8_t *out_refund_cbor, uint8_t *status);
/* ... and the nullifier `k` of every message as it stood on the wire (n*32; zero for a message that did not parse), as
 * act_verify_spend_cbor_keys_batch returns it: for a caller that keeps the double-spend store itself and decides AFTER the refund was
 * computed whether to hand it out (what act_node_redeem_cbor_batch does with a few messages: refund in one call, then the store). */
int act_refund_cbor_keys_batch(act_ctx *ctx, size_t n, int mem, const uint8_t sk[64], const uint8_t *cbor, const uint64_t *offsets,
                               const uint8_t *rng, int rng_mode, uint8_t *out_refund_cbor, uint8_t *status, uint8_t *out_nullifier);
int act_node_refund_sign_cbor_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *kprime, const uint8_t *status_in,
                                    const uint8_t *rng, int rng_mode, uint8_t *out_refund_cbor, uint8_t *status);
int act_node_refund_cbor_batch(act_node *node, size_t n, const uint8_t sk[64], const uint8_t *cbor, const uint64_t *offsets,
                               const uint8_t *rng, int rng_mode, uint8_t *out_refund_cbor, uint8_t *status);

/* ======== row f4: the nullifier set and the issuer's whole redemption step ====================================================== */
/* Nullifier set: the double-spend database the crate leaves to the caller (src/lib.rs:741-745; `HashSet<Scalar>` with
 * "is_spent? reject : insert" per spend in src/tests.rs:29-50, examples/act.rs:10-30), as a hash set in one GPU's HBM.
 * act_nullifier_check_and_insert_batch has the meaning of that loop run over the batch in lane order: out_spent[i] = 1
 * iff nullifier i is already in the set or equals the nullifier of an earlier lane of this batch; fresh nullifiers are
 * inserted.  Nullifier i is the 32 bytes at nullifiers + i*stride (stride = act_spend_proof_bytes reads the `k` field
 * straight out of SpendProof records).  skip_mask (nullable): lanes with a non-zero byte (e.g. the status of a rejected
 * proof) are neither checked nor inserted and report 0.  Keys are compared as scalars: every nullifier is reduced mod l
 * on input (k and k + l are the same key, as in the reference's HashSet<Scalar>).  capacity = the number of nullifiers
 * the set must hold; `salt` (16 bytes) keys the SipHash-1-3 slot hash; NULL = 16 bytes from the OS (getrandom).  Multi-GPU deployments shard the key space (owner = low 64 bits mod N): one
 * set per GPU behind an all-to-all of the keys, anonymous-credit-tokens_amd/sharded_nullifier.py. */
typedef struct act_nullifier_set act_nullifier_set;
int act_nullifier_set_create(int device, size_t capacity, const uint8_t salt[16], act_nullifier_set **out);
void act_nullifier_set_destroy(act_nullifier_set *set);
size_t act_nullifier_set_len(const act_nullifier_set *set);
const char *act_nullifier_set_last_error(const act_nullifier_set *set);
int act_nullifier_check_and_insert_batch(act_nullifier_set *set, size_t n, int mem, const uint8_t *nullifiers, size_t stride,
                                         const uint8_t *skip_mask, uint8_t *out_spent);

/* The same set spread over the GPUs of a node: one set per entry of devices[], a nullifier owned by exactly one of them
 * (keyed hash of the reduced scalar), so a batch keeps the sequential meaning above in lane order.  The host buckets the
 * keys by owner (stable), every GPU checks-and-inserts its bucket from its own thread, answers are scattered back:
 * 33 bytes per spend over PCIe, no peer traffic.  Host memory only.  `salt` (16 bytes) keys the routing (SipHash-1-3 of the
 * reduced scalar, so that clients, who choose their nullifiers, cannot aim them at one GPU); NULL = 16 bytes from the OS
 * (getrandom), and creation fails if that fails.  A handle may be shared between host threads (calls are serialised).
 * Failure of one device: the other devices have checked and inserted their keys all the same, so the call fills in their
 * answers (final), marks the lanes owned by the failed device ACT_NULLIFIER_UNDETERMINED in out_spent, and returns the error;
 * only the undetermined lanes may be resubmitted -- a blind retry of the whole batch would report the honest spends that
 * were already inserted as double spends.  The single-GPU form answers the same way: a batch the set has no room for is refused as
 * a whole (nothing recorded, every unmasked lane ACT_NULLIFIER_UNDETERMINED). */
#define ACT_NULLIFIER_UNDETERMINED 2
typedef struct act_node_nullifier_set act_node_nullifier_set;
int act_node_nullifier_set_create(const int *devices, int n_devices, size_t capacity_per_device, const uint8_t salt[16],
                                  act_node_nullifier_set **out);
void act_node_nullifier_set_destroy(act_node_nullifier_set *set);
size_t act_node_nullifier_set_len(const act_node_nullifier_set *set);
const char *act_node_nullifier_set_last_error(const act_node_nullifier_set *set);
int act_node_nullifier_check_and_insert_batch(act_node_nullifier_set *set, size_t n, const uint8_t *nullifiers, size_t stride,
                                              const uint8_t *skip_mask, uint8_t *out_spent);

/* The issuer's whole redemption step -- verify, look the nullifier up, record it, sign the refund (examples/act.rs:62-73; the
 * NullifierDb loops of src/tests.rs) -- as one call with the result of the loop
 *     for i in 0..n:  refund(proof_i)'s checks fail (src/lib.rs:787-844)  -> status[i] = that error; nothing recorded, no rng drawn
 *                     nullifier_i already in the set, or spent by an earlier accepted lane of this batch
 *                                                                        -> status[i] = ACT_STATUS_DOUBLE_SPEND; no rng drawn
 *                     otherwise: recorded, signed (src/lib.rs:846-868)    -> status[i] = 0, out_refund[i]
 * Verification comes first, so a proof that does not verify cannot burn a nullifier (the crate's example marks the nullifier
 * before calling refund and unwraps: same result for valid proofs).  rng / rng_mode as in act_refund_batch: ACT_RNG_SEQUENTIAL hands
 * consecutive 128-byte slices to the lanes that are SIGNED.  The set must live on the context's device.  The node form runs the
 * verification and the signatures on all GPUs and the look-up through the node-level set.  The three steps take their handles'
 * locks one after the other: concurrent callers interleave between steps, never inside one.
 * Failures: a non-zero result never hides a decision already taken -- status[] is complete when the call returns.
 *   verification failed (a HIP error)   nothing recorded, nothing signed; status[] not written.
 *   the nullifier step failed           lanes it answered are finished as usual (0 + refund, or ACT_STATUS_DOUBLE_SPEND); lanes it could
 *                                       not answer -- the set was too small for the batch, or (node form) their owner GPU failed -- get
 *                                       ACT_STATUS_NULLIFIER_UNDETERMINED: not recorded, not signed, safe to resubmit.
 *   the signature step failed           nullifiers ARE recorded; the lanes that were to be signed (node form: those of the failing GPU's
 *                                       shard) get ACT_STATUS_RECORDED_UNSIGNED and a zero record.  Their refund is owed: call
 *                                       act_verify_spend_batch(out_kprime) and act_refund_sign_batch on exactly those lanes.  Redeeming them
 *                                       again would report DoubleSpendError and the client would lose its credits.
 * A few items at a time (host memory, at most 64, rng slices that do not depend on the verdicts: ACT_RNG_PER_LANE bytes, or ONE item
 * with its 128 bytes): the refunds are computed FIRST -- one call, the signature beside the verification -- and the store then decides
 * which of them are handed out (2.1 ms for one item instead of 3.2).  Same statuses, refunds and store; the third failure above cannot
 * occur on this road (nothing is signed behind a recorded nullifier). */
int act_redeem_batch(act_ctx *ctx, act_nullifier_set *set, size_t n, int mem, const uint8_t sk[64], const uint8_t *proof,
                     const uint8_t *rng, int rng_mode, uint8_t *out_refund, uint8_t *status);
int act_node_redeem_batch(act_node *node, act_node_nullifier_set *set, size_t n, const uint8_t sk[64], const uint8_t *proof,
                          const uint8_t *rng, int rng_mode, uint8_t *out_refund, uint8_t *status);
/* The redemption step on wire bytes: act_redeem_batch with CBOR SpendProof messages in (as act_verify_spend_cbor_batch) and CBOR Refund
 * messages out (as act_refund_cbor_batch) -- the loop of examples/act.rs:62-73 for a server that holds bytes.  status[i] additionally
 * takes the wire codes 255 / ACT_STATUS_CBOR_MALFORMED / ACT_STATUS_CBOR_STRUCTURE; failure semantics as above.
 * All four redeem entry points accept ACT_RNG_CALLBACK: the draw happens after the nullifier step, for the lanes that are signed. */
int act_redeem_cbor_batch(act_ctx *ctx, act_nullifier_set *set, size_t n, int mem, const uint8_t sk[64], const uint8_t *cbor,
                          const uint64_t *offsets, const uint8_t *rng, int rng_mode, uint8_t *out_refund_cbor, uint8_t *status);
int act_node_redeem_cbor_batch(act_node *node, act_node_nullifier_set *set, size_t n, const uint8_t sk[64], const uint8_t *cbor,
                               const uint64_t *offsets, const uint8_t *rng, int rng_mode, uint8_t *out_refund_cbor, uint8_t *status);

/* ======== row d and test infrastructure: debug hooks, measurement knobs, kernel timing, roofline probes (nothing here is on the product's path) ==== */
/* Debug / test hook: the exact "spend" transcript pre-images of the last act_verify_spend_batch /
 * act_refund_batch chunk (n_last * act_spend_transcript_bytes, copied to host memory). */
int act_debug_last_spend_transcripts(act_ctx *ctx, size_t max_lanes, uint8_t *out, size_t *n_copied);

/* Debug / test hook: number of non-zero bytes left in the context's secret-bearing device buffers -- the staging copies of
 * host inputs (tokens, PreIssuance, rng), the signer's nonces, the prover's r3 / r* / k* terms and the per-proof Pippenger
 * buckets -- all of which every entry point clears before it returns (the crate's ZeroizeOnDrop, src/lib.rs:160,362,393). */
int act_debug_secret_residue(act_ctx *ctx, size_t *nonzero_bytes);
/* Debug / test hook: every batch call on this context additionally sleeps ns_per_lane nanoseconds per lane while it holds the
 * context: a GPU that is slower than its neighbours, for the load-balance tests of the node dispatcher.  0 = off. */
int act_debug_set_slowdown(act_ctx *ctx, uint32_t ns_per_lane);
/* Debug / test hook: the signature step of the next `count` act_redeem_batch / act_redeem_cbor_batch calls on this context fails after
 * the nullifiers have been recorded (the failure ACT_STATUS_RECORDED_UNSIGNED exists for; no input can provoke it). */
int act_debug_fail_next_signs(act_ctx *ctx, int count);
/* Debug / test hook: out[i] = enc(scalars[i] * points[i]) (`RistrettoPoint * Scalar`, e.g. src/lib.rs:791) computed by the
 * engine's production variable-base chain, decode and encode; status[i] = 255 and a zero record when points[i] is not a
 * canonical encoding.  Exists so that third-party known answers can be replayed on the device one operation at a time
 * (tests/golden/sodium_primitives.json). */
int act_debug_scalarmult_batch(act_ctx *ctx, size_t n, int mem, const uint8_t *points, const uint8_t *scalars, uint8_t *out,
                               uint8_t *status);

/* Measurement hook: A/B switches and size overrides of tools/ and tests/ (csrc/kernels.h TuneKey: "no_mapped_reads", "no_fused_tiny",
 * "no_taper", "no_wide_sign", "no_wide_prove", "no_wide_client", "no_lds_isolation", "no_stream_probe", "small_sub", "small_in_flight",
 * "small_normal_prio", "small_trace", "stagger", "host_chunk", "cbor_chunk_msgs", "ubench_iters").  Process-wide; none of them changes a
 * byte of any result.  A deployment never calls this; the library reads no environment variable in its place. */
int act_tuning_set(const char *name, int64_t value);

/* Kernel timing (HIP events on the context's own stream, which torch.cuda.Event cannot see):
 * enable, run batches, then read per-kernel totals.  names: act_prof_kernel_name(i), i < act_prof_kernel_count(). */
int act_prof_enable(act_ctx *ctx, int on);
int act_prof_reset(act_ctx *ctx);
int act_prof_kernel_count(const act_ctx *ctx);
const char *act_prof_kernel_name(const act_ctx *ctx, int i);
int act_prof_get(act_ctx *ctx, int i, double *ms_total, uint64_t *launches, uint64_t *lanes);
/* ms during which at least one launch of kernel i was executing (union of the launch intervals; launches of two chunks
 * in flight overlap, so ms_total of act_prof_get can exceed the wall time) */
int act_prof_get_busy(act_ctx *ctx, int i, double *ms_busy);
/* ALU roofline probe: rate of the 64-bit multiply-accumulate (v_mad_u64_u32) the field arithmetic is made of, with every
 * SIMD of `device` saturated by register-resident dependency chains.  lane_mads_per_s is summed over lanes; ms = probe time. */
int act_ubench_mad_u64_u32(int device, double *lane_mads_per_s, double *ms);
/* Memory-side roofline probe of the scalar-addressed fixed-base tables: every lane reads pseudo-random 128-byte lines (seven 16-byte
 * loads each, as an affine-Niels table entry is read) of a `gib` GiB buffer (0 = 16) allocated for the probe, `in_flight` (1, 2 or 4;
 * 0 = 2) entries at a time, with `waves_per_simd` (0 = 2, the occupancy of the kernels that do these look-ups) wavefronts per SIMD.
 * gbytes_per_s counts 128 bytes per read. */
int act_ubench_random_read(int device, size_t gib, int waves_per_simd, int in_flight, double *gbytes_per_s, double *ms);
/* The same probe over the context's own fixed-base table of base 0..3 (g, h1, h2, h3; the largest power-of-two prefix of it): the
 * product's look-ups in the product's memory with nothing else running. */
int act_ubench_table_read(act_ctx *ctx, int base, int waves_per_simd, int in_flight, double *gbytes_per_s, double *ms);

#ifdef __cplusplus
}
#endif
#endif
