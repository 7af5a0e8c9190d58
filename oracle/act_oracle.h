/* act_oracle.h — CPU oracle for the anonymous-credit-tokens sigma-protocol hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (libact_mi355x.so) neither links nor calls it.
 *
 * Parity status: "parity unpinned" against the Rust crate itself (no Rust toolchain, the
 * arithmetic lives in un-vendored curve25519-dalek 4.1.3 / blake3 1.8.2, and the reference ships
 * no golden vectors).  Pinned instead against: upstream BLAKE3 (LLVM's bundled C implementation),
 * OpenSSL Ed25519, the RFC 9496 ristretto255 vectors, and oracle/pymodel.py (an independently
 * written big-integer model) through the committed fixtures in tests/golden/.
 *
 * Record layouts are the raw 32-byte-field layouts of SURVEY.md Appendix C (CBOR key order of
 * /root/reference/src/cbor.rs with framing stripped).  `rng` arguments are the bytes a
 * CryptoRngCore would have produced, consumed in the draw order of SURVEY.md Appendix B.
 */
#ifndef ACT_ORACLE_H
#define ACT_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_L_MAX 128

/* status codes: 0 = Ok, else 1 + discriminant of reference `Error` (src/lib.rs:102-112); 255 = undecodable input */
enum {
  ORACLE_OK = 0,
  ORACLE_ERR_INVALID_ISSUANCE_REQUEST_PROOF = 1,
  ORACLE_ERR_INVALID_ISSUANCE_RESPONSE_PROOF = 2,
  ORACLE_ERR_INVALID_REFUND_PROOF = 4,
  ORACLE_ERR_IDENTITY_POINT = 6,
  ORACLE_ERR_INVALID_CLIENT_SPEND_PROOF = 7,
  ORACLE_ERR_UNDECODABLE = 255
};

typedef struct oracle_ctx oracle_ctx;

/* primitives (exported so tests can pin them individually) */
void oracle_blake3(const uint8_t *in, size_t len, uint8_t *out, size_t outlen);
void oracle_sc_reduce_wide(const uint8_t in[64], uint8_t out[32]);
void oracle_sc_muladd(const uint8_t a[32], const uint8_t b[32], const uint8_t c[32], uint8_t out[32]); /* a*b+c */
void oracle_sc_invert(const uint8_t a[32], uint8_t out[32]);
int  oracle_ristretto_decode_encode(const uint8_t in[32], uint8_t out[32]);   /* 1 if `in` is a valid encoding */
void oracle_ristretto_from_uniform(const uint8_t in[64], uint8_t out[32]);
int  oracle_ristretto_mul(const uint8_t pt[32], const uint8_t sc[32], uint8_t out[32]);  /* variable-base */
void oracle_ristretto_mul_base(const uint8_t sc[32], uint8_t out[32]);        /* sc * generator */
int  oracle_ristretto_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);

/* Params::new (src/lib.rs:291-315): writes enc(h1)|enc(h2)|enc(h3) */
void oracle_params_new(const char *org, const char *svc, const char *dep, const char *ver, uint8_t out[96]);

/* context = Params (three generators + their fixed-base tables) + range-proof width L (src/lib.rs:116) */
oracle_ctx *oracle_ctx_new(const uint8_t h[96], int L);
void oracle_ctx_free(oracle_ctx *);
size_t oracle_spend_proof_bytes(const oracle_ctx *);   /* 32*(14+4L) */
size_t oracle_prove_rng_bytes(const oracle_ctx *);     /* 64*(4L+12) */

/* PrivateKey::random (src/lib.rs:188-194): rng 64 B -> record x|w (64 B) */
void oracle_private_key_random(const uint8_t rng[64], uint8_t sk[64]);
/* PreIssuance::random (src/lib.rs:432-437): rng 128 B -> record r|k */
void oracle_pre_issuance_random(const uint8_t rng[128], uint8_t pre[64]);

/* PreIssuance::request (src/lib.rs:463-487) */
void oracle_request(const oracle_ctx *, const uint8_t pre[64], const uint8_t rng[128], uint8_t out_req[128]);
/* PrivateKey::issue (src/lib.rs:621-663); rng consumed only on accept */
int oracle_issue(const oracle_ctx *, const uint8_t sk[64], const uint8_t req[128], const uint8_t c[32],
                 const uint8_t rng[128], uint8_t out_resp[160]);
/* PreIssuance::to_credit_token (src/lib.rs:528-562) */
int oracle_issuance_to_credit_token(const oracle_ctx *, const uint8_t pre[64], const uint8_t w[32],
                                    const uint8_t req[128], const uint8_t resp[160], uint8_t out_tok[160]);
/* CreditToken::prove_spend (src/lib.rs:972-1152) */
int oracle_prove_spend(const oracle_ctx *, const uint8_t tok[160], const uint8_t s[32], const uint8_t *rng,
                       uint8_t *out_proof, uint8_t out_prerefund[96]);
/* src/lib.rs:787-844 only: identity check, recompute commitments, challenge compare.  Optionally returns
 * the transcript pre-image (cap >= 40*(2+... ) see oracle_spend_transcript_bytes) and enc(K'). */
int oracle_verify_spend(const oracle_ctx *, const uint8_t sk[64], const uint8_t *proof,
                        uint8_t *out_transcript /* nullable */, uint8_t out_kprime[32] /* nullable */);
size_t oracle_spend_transcript_bytes(const oracle_ctx *);
/* PrivateKey::refund (src/lib.rs:781-869); rng consumed only on accept */
int oracle_refund(const oracle_ctx *, const uint8_t sk[64], const uint8_t *proof, const uint8_t rng[128],
                  uint8_t out_refund[128]);
/* PreRefund::to_credit_token (src/lib.rs:1217-1253) */
int oracle_refund_to_credit_token(const oracle_ctx *, const uint8_t prerefund[96], const uint8_t *proof,
                                  const uint8_t refund[128], const uint8_t w[32], uint8_t out_tok[160]);

/* Threaded loops over contiguous sub-batches (CPU baseline): one pthread per `nthreads`. */
void oracle_verify_spend_batch(const oracle_ctx *, const uint8_t sk[64], size_t n, const uint8_t *proofs,
                               uint8_t *status, int nthreads);
void oracle_refund_batch(const oracle_ctx *, const uint8_t sk[64], size_t n, const uint8_t *proofs,
                         const uint8_t *rng /* n*128, PER_LANE */, uint8_t *out_refunds, uint8_t *status, int nthreads);
void oracle_prove_spend_batch(const oracle_ctx *, size_t n, const uint8_t *toks, const uint8_t *s,
                              const uint8_t *rng, uint8_t *out_proofs, uint8_t *out_prerefunds, int nthreads);
void oracle_issue_batch(const oracle_ctx *, const uint8_t sk[64], size_t n, const uint8_t *reqs, const uint8_t *c,
                        const uint8_t *rng, uint8_t *out_resps, uint8_t *status, int nthreads);
void oracle_request_batch(const oracle_ctx *, size_t n, const uint8_t *pres, const uint8_t *rng,
                          uint8_t *out_reqs, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
