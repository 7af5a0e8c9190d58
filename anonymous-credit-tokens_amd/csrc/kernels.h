// kernels.h — argument blocks and launchers shared by the HIP kernels and the host engine.
// One lane = one proof (head / tail / sign kernels) or one (proof, bit) pair (range-proof kernels).
#pragma once
#include <stdlib.h>
#include <stddef.h>
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
#include <string.h>      // tests/hostcheck compiles the lane bodies (spend_lanes.h) with g++: never a product path
#endif
#include "msm.h"
#include "blake3_hd.h"
#include <atomic>

// ---- measurement knobs ----------------------------------------------------------------------------------------------------
// A/B switches and size overrides that tools/ and tests/ use (same-box comparisons; tests that want many chunks out of a small
// batch).  The library does not read the process environment for any of them: act_tuning_set(name, value) (include/act_mi355x.h,
// debug section) is the only way in, the Python binding forwards the ACT_* variables of the same names (capi.py), and a deployment
// that never calls it runs the defaults.  (engine.hip holds the table; the environment is read only for ACT_TRACE /
// ACT_TIMELINE_FILE diagnostics and ACT_NUMA.)
namespace act {
enum TuneKey {
  T_NO_MAPPED_READS, T_NO_STREAM_PROBE, T_NO_FUSED_TINY, T_NO_TAPER, T_NO_WIDE_CLIENT, T_NO_WIDE_PROVE, T_NO_WIDE_SIGN, T_NO_LDS_ISOLATION,
  T_SMALL_NORMAL_PRIO, T_SMALL_TRACE, T_SMALL_IN_FLIGHT, T_SMALL_SUB, T_STAGGER, T_HARD_STAGGER, T_HOST_CHUNK, T_CBOR_CHUNK_MSGS, T_UBENCH_ITERS, T_COUNT
};
extern std::atomic<long> g_tune[T_COUNT];
inline long tune(TuneKey k) { return g_tune[k].load(std::memory_order_relaxed); }
}  // namespace act

#if defined(__HIPCC__)
#define ACT_HDC __host__ __device__
#else
#define ACT_HDC
#endif

namespace act {

enum { LABEL_REQUEST = 0, LABEL_RESPOND = 1, LABEL_SPEND = 2, LABEL_REFUND = 3 };
enum { BASE_G = 0, BASE_H1 = 1, BASE_H2 = 2, BASE_H3 = 3 };
enum : uint32_t { FLAG_UNDECODABLE = 1u, FLAG_IDENTITY = 2u, FLAG_MISMATCH = 4u };

constexpr int PREFIX_WORDS = 48;          // transcript prefixes are 184..186 bytes (src/transcript.rs:54-74)
constexpr int SMALL_TR_STRIDE = 512;      // request 266 B, respond 466 B, refund 425 B

// Params-dependent constants, by value in every launch (~1 KiB of kernarg, read through the scalar cache)
struct DevParams {
  FbTab tab[4];                           // position-specific fixed-base tables of g, h1, h2, h3 with their window widths (msm.h)
  const uint32_t* half_h1;                // two affine-Niels entries: identity, h1 / 2 (the prover's bit term at half scale)
  const uint32_t* tab_ct[4];              // small tables the secret-scalar products scan in full (msm.h fixed_base_acc_ct)
  const uint8_t* tab_mf[4];               // matrix-core table images: 37 windows x 64 entries as MFMA A operands (msm.h fixed_base_acc_mf)
  uint32_t prefix[4][PREFIX_WORDS];       // Transcript::new(params, label) bytes, zero padded
  uint32_t prefix_len[4];
  int L;                                  // range-proof width (src/lib.rs:116)
};
struct DevKey { sc x; ge w; };            // PrivateKey (src/lib.rs:161-167), w decoded

// ---- SpendProof record field indices (32-byte fields; src/cbor.rs:250-268) ----
struct ProofLayout {
  int L;
  ACT_HDC int k() const { return 0; }
  ACT_HDC int s() const { return 1; }
  ACT_HDC int a_prime() const { return 2; }
  ACT_HDC int b_bar() const { return 3; }
  ACT_HDC int com(int j) const { return 4 + j; }
  ACT_HDC int gamma() const { return 4 + L; }
  ACT_HDC int e_bar() const { return 5 + L; }
  ACT_HDC int r2_bar() const { return 6 + L; }
  ACT_HDC int r3_bar() const { return 7 + L; }
  ACT_HDC int c_bar() const { return 8 + L; }
  ACT_HDC int r_bar() const { return 9 + L; }
  ACT_HDC int w00() const { return 10 + L; }
  ACT_HDC int w01() const { return 11 + L; }
  ACT_HDC int gamma0(int j) const { return 12 + L + j; }
  ACT_HDC int z(int j, int b) const { return 12 + 2 * L + 2 * j + b; }
  ACT_HDC int k_bar() const { return 12 + 4 * L; }
  ACT_HDC int s_bar() const { return 13 + 4 * L; }
  ACT_HDC size_t bytes() const { return 32u * (14u + 4u * (size_t)L); }
};
// "spend" transcript element slots (40 bytes each after the prefix; src/lib.rs:831-840)
struct SpendTranscript {
  int L;
  ACT_HDC int el_k() const { return 0; }
  ACT_HDC int el_a_prime() const { return 1; }
  ACT_HDC int el_b_bar() const { return 2; }
  ACT_HDC int el_a1() const { return 3; }
  ACT_HDC int el_a2() const { return 4; }
  ACT_HDC int el_com(int j) const { return 5 + j; }
  ACT_HDC int el_cprime(int j, int b) const { return 5 + L + 2 * j + b; }
  ACT_HDC int el_c() const { return 5 + 3 * L; }
  ACT_HDC size_t bytes() const { return 184u + 40u * (6u + 3u * (size_t)L); }
  ACT_HDC size_t stride() const { return (bytes() + 15u) & ~(size_t)15u; }
};

constexpr int PART_POINTS = 4;           // small-batch schedule: partial sums of A1 / A2 per proof (spend_lanes.h)
constexpr int PREP_BUCKET_SETS = 3;      // bucket sets per proof that k_spend_prep's three roles use side by side: d_buckets holds max(L, 3) per proof

struct SpendArgs {
  DevParams P;
  DevKey K;
  const uint8_t* proofs;     // n records
  uint32_t n;
  uint8_t* tr;               // n * tr_stride bytes: "spend" transcript pre-images
  uint32_t tr_stride;
  uint32_t* coords;          // n * L * NIELS_WORDS : affine Niels of every decoded Com_j
  uint32_t* d01;             // n * 2 * GE_WORDS    : w00*h2, w01*h2
  uint32_t* buckets;         // n * max(L, 3) * BUCKET_WORDS: per-lane Pippenger buckets of k_spend_bits (msm.h chain_bu); the
                             // per-proof kernels use PREP_BUCKET_SETS sets per proof (chain_b / chain_s)
  uint32_t* xa;              // n * GE_WORDS        : X_A = g + K'
  uint32_t* flags;           // n
  const uint32_t* xof;       // n * 16              : BLAKE3 XOF words of the transcript
  uint8_t* status;           // n
  uint8_t* kprime_enc;       // n * 32 or null
  uint32_t* naf;             // n * NAF_WORDS       : width-3-NAF digit string of gamma / 2 (k_spend_prep -> k_spend_bits, msm.h)
  uint32_t* dig;             // n * L * 8           : biased radix-16 digit words of gamma_j0 / 2, per (proof, bit) lane
  uint32_t* pbk;             // n * PREP_BUCKET_SETS * BUCKET_WORDS: bucket sets of the per-proof kernels (prep role B: sets 1, 2; tail: set 0).
                             // Large batches: = buckets (prep, bits, tail run one after the other).  Small-batch schedule: an area of
                             // its own, because those kernels then run NEXT TO k_spend_bits (spend_lanes.h)
  uint32_t* part;            // small-batch schedule only: n * 4 * GE_WORDS partial sums of A1 / A2 (spend_lanes.h PART_POINTS)
  uint32_t* progress;        // nullable: k_spend_bits counts its finished workgroups here (engine.hip spend_stage1: the next chunk's range kernel
                             // is released when this one's LAST ROUND of workgroups is running, not when it has drained)
};

struct SignArgs {
  DevParams P;
  DevKey K;
  uint32_t n;
  int label;                 // LABEL_RESPOND (issue) or LABEL_REFUND (refund)
  const uint32_t* xa;        // n * GE_WORDS
  const uint8_t* status;     // n: lanes with status != 0 produce an all-zero record
  const uint32_t* rng_slot;  // n: index of the lane's 128-byte rng slice
  const uint8_t* rng;
  const uint8_t* c_amount;   // n * 32 (issue) or null
  uint8_t* trs;              // n * SMALL_TR_STRIDE small transcripts
  uint32_t* state;           // n * 24 words: e | alpha | enc(A)
  const uint32_t* xof;
  uint8_t* out;              // n * 128 (refund) or n * 160 (issue)
  uint32_t* pbk;             // n * 2 * BUCKET_WORDS: per-proof Pippenger buckets (msm.h chain_b)
};

// k_sign_fused (k_sign.hip): the whole signature -- and, with `check`, the request's PoK check beside it -- in one kernel, for tiny calls
struct SignFusedArgs {
  DevParams P; DevKey K;
  uint32_t n; int label;
  const uint8_t* point; uint32_t point_stride;     // IssuanceRequest records (K first, 128 B) or enc(K') (32 B); CHECK needs the records
  const uint32_t* xa;                              // or (!CHECK, nullable) X_A itself, n * GE_WORDS: then `point` is not read
  const uint32_t* rng_slot;                        // nullable: lane p draws from slice rng_slot[p] instead of p
  const uint8_t* c_amount;                         // n * 32 (issue) or null (refund)
  const uint8_t* status_in;                        // !CHECK: verdicts of the check phase; lanes with a non-zero byte are not signed
  const uint8_t* rng;                              // lane p's 128 bytes at rng + 128 p
  uint8_t* out; uint8_t* status;
  uint32_t* pbk;                                   // n * PREP_BUCKET_SETS * BUCKET_WORDS: sets 0, 1 = the check's chain_b, set 2 = the quarters of A and Y_A
  uint8_t* trs;                                    // n * SMALL_TR_STRIDE: the "respond" / "refund" transcripts the roles assemble
  uint8_t* trs_req;                                // CHECK: n * SMALL_TR_STRIDE "request" transcripts
  uint32_t* group_counter;                         // one word per group of 64 lanes, zero between launches: which block arrives last
  int check_only;                                  // CHECK: only the check role runs (act_issue_check_batch): status out, no signature
  int before_verdict;                              // !CHECK: EVERY lane is signed and `status` is neither read nor written -- the verdicts are not known yet: a tiny
                                                   // refund signs beside its verification, into a buffer of the engine's; launch_sign_commit then hands out
                                                   // what the verdicts allow (small_impl.inc spend_small_locked)
  int wipe_rng;                                    // rng is the engine's staged copy: zero it when done
  unsigned long long* dbg;                         // -DACT_TINY_TIMING builds only: 8 time stamps per role
};

// X_A for the sign-only entry points: from K of the request (+ c h1) or from enc(K') of a verified spend proof
struct SignXaArgs {
  DevParams P;
  uint32_t n;
  const uint8_t* point;      // n encodings at point_stride: IssuanceRequest records (K first) or enc(K')
  uint32_t point_stride;
  const uint8_t* c_amount;   // n * 32 (issue) or null (refund)
  uint32_t* xa;              // n * GE_WORDS
  uint8_t* status;           // n: in = verdict of the check phase; a lane whose point does not decode becomes 255
};

struct IssueArgs {
  DevParams P;
  uint32_t n;
  const uint8_t* req;        // n * 128
  const uint8_t* c_amount;   // n * 32
  uint8_t* trs;
  uint32_t* xa;
  uint32_t* flags;
  const uint32_t* xof;
  uint8_t* status;
  uint32_t* pbk;             // n * 2 * BUCKET_WORDS (msm.h chain_b)
};

struct RequestArgs {
  DevParams P;
  uint32_t n;
  const uint8_t* pre;        // n * 64  (r | k)
  const uint8_t* rng;        // n * 128
  uint8_t* trs;
  const uint32_t* xof;
  uint8_t* out;              // n * 128
  int wipe_inputs;           // k_request_fused: pre / rng are the engine's staged copies -- zero them when done
};

// CreditToken::prove_spend (src/lib.rs:972-1152)
struct ProveArgs {
  DevParams P;
  uint32_t n;
  const uint8_t* tok;        // n * 160  (a | e | k | r | c)
  const uint8_t* s;          // n * 32
  const uint8_t* rng;        // n * 64*(4L+12), draw order of SURVEY.md Appendix B
  uint8_t* tr; uint32_t tr_stride;
  uint32_t* d3;              // n * 3 * GE_WORDS : k* h2, k0' h2, (w0 - gamma_0 k*) h2, each at half scale
  uint32_t* half;            // n * L * BUCKET_WORDS: slots 0..2 of lane (p, j) = Com_j / 2, C'_j0 / 2, C'_j1 / 2 for k_prove_enc
  uint32_t* state;           // n * 24 words: r3 | r*
  uint32_t* flags;
  const uint32_t* xof;
  uint8_t* proof;            // n * 32*(14+4L)
  uint8_t* prerefund;        // n * 96
  uint8_t* status;
  uint32_t* group_counter;   // k_prove_head_wide: one word per group of 64 proofs, zero between launches (which role block arrives last)
};
// PreIssuance::to_credit_token (src/lib.rs:528-562) / PreRefund::to_credit_token (:1217-1253)
struct ClientArgs {
  DevParams P;
  ge w;                      // issuer public key, decoded
  uint32_t n;
  int label;                 // LABEL_RESPOND or LABEL_REFUND
  const uint8_t* pre;        // n * 64 (r|k)  or  n * 96 (r|k|m)
  const uint8_t* req;        // n * 128 (issuance) or null
  const uint8_t* resp;       // n * 160 (IssuanceResponse) or n * 128 (Refund)
  const uint8_t* proofs;     // refund: n spend proofs
  uint32_t* coords;          // refund: decoded Com_j
  uint8_t* trs;
  uint32_t* flags;
  const uint32_t* xof;
  uint8_t* out_token;        // n * 160
  uint8_t* status;
  uint32_t* pbk;             // n * PREP_BUCKET_SETS * BUCKET_WORDS (msm.h chain_b); the wide kernel also parks two partial points there
  uint32_t* group_counter;   // wide kernel: one word per group of 64 items, zero between launches (which role block arrives last)
  int fused;                 // wide kernel, tiny calls: the last block also hashes the transcript and does k_client_b's part
};

struct HashArgs { const uint8_t* msg; uint32_t stride; uint32_t len; uint32_t n; uint32_t* xof; const uint32_t* len_per_lane; };

#if defined(__HIPCC__)
// launchers (defined in the .hip files)
void launch_build_table(const uint32_t* base_ext /*GE_WORDS, device*/, uint32_t* table, uint32_t wbits, hipStream_t s);
void launch_build_table_ct(const uint32_t* base_ext, uint32_t* table, hipStream_t s);
void launch_build_table_mf(const uint32_t* base_ext, uint8_t* image /* MF_TABLE_BYTES */, hipStream_t s);
constexpr size_t MF_TABLE_BYTES = (size_t)64 * 16384;       // room for any of the three window shapes (msm.h: 43 x 4, 37 x 8, 32 x 16 KiB)
void launch_half_point_table(FbTab table, uint32_t* out /*2 * NIELS_WORDS*/, hipStream_t s);
void launch_decode_points(const uint8_t* enc, uint32_t n, uint32_t* out_ext, uint32_t* ok, hipStream_t s);
void launch_from_uniform(const uint8_t* in64, uint32_t n, uint8_t* out_enc, hipStream_t s);
void launch_keygen(const DevParams& P, const uint8_t* rng64, uint32_t n, uint8_t* out_sk, hipStream_t s);
void launch_pre_issuance_random(const uint8_t* rng, uint32_t n, uint8_t* out, hipStream_t s);
void launch_hash(const HashArgs& a, hipStream_t s);
void launch_hash_par(const HashArgs& a, hipStream_t s);      // sixteen lanes per message: small calls
void launch_iota(uint32_t* out, uint32_t n, uint32_t base, hipStream_t s);
void launch_spin(uint32_t ticks_100mhz, hipStream_t s);
void launch_xof_expand(const uint32_t* d_seed /* 8 words, device */, uint64_t first_lane, uint32_t n, uint32_t blocks_per_lane, uint8_t* out, hipStream_t s);
void launch_ubench_random_read(const uint32_t* buf, uint64_t lines, uint32_t blocks, uint32_t iters, int in_flight, uint32_t* out, hipStream_t s);
constexpr int UBENCH_MADS_PER_ITER = 80;    // 8 chains x 10 dependent multiply-accumulates (k_misc.hip k_ubench_mad)
void launch_ubench_mad(uint32_t* out, uint32_t blocks, uint32_t iters, hipStream_t s);
void launch_debug_scalarmult(const uint8_t* pts, const uint8_t* scs, uint32_t n, uint32_t* pbk, uint8_t* out, uint8_t* status, hipStream_t s);
void launch_spend_prep(const SpendArgs& a, hipStream_t s);
void launch_spend_prep_role(const SpendArgs& a, int role /* 0 A, 1 B, 2 C, 3 join, 4 C1 (what k_spend_bits waits for), 5 C2 */, hipStream_t s);
void launch_spend_enc_small(const SpendArgs& a, hipStream_t s);
void launch_spend_coords(const SpendArgs& a, hipStream_t s);
void launch_spend_bits(const SpendArgs& a, hipStream_t s);
unsigned spend_bits_workgroups(const SpendArgs& a);      // the grid launch_spend_bits uses = what a.progress reaches
void launch_spend_enc(const SpendArgs& a, hipStream_t s);
// Launches that leave most of the chip empty (a call of a few hundred proofs, a one-item call's role blocks beside other kernels)
// want every wavefront on a SIMD of its own: their lanes are dependent chains, and two workgroups that the dispatcher happens to put
// on one CU run at half speed each while other CUs idle (measured: the range kernel of a 192-proof call 1.65 ms, of a 224-proof call
// 1.09 ms -- placement, not size; with it gone a 256-proof call takes 1.87 ms instead of 2.46).  LDS is the one resource a launch can
// claim without using it: with 40 KB of dynamic LDS beside its 72 KB a range-kernel workgroup owns its CU (two no longer fit the
// 160 KB), and with 60 KB a one-wavefront workgroup of the per-proof kernels cannot join it there either (and at most two of its own
// kind share a CU).  Only for grids that fit the chip that way; ACT_NO_LDS_ISOLATION=1 turns it off (A/B).
inline bool lds_isolation() { return !tune(T_NO_LDS_ISOLATION); }
// The sizes at which a launch "fits the chip that way" follow the CU count of the device the calling thread has current (256 on an
// unpartitioned MI355X; a CPX partition has 32): 7/8 of the CUs for the range kernel's workgroups, the rest for the per-proof kernels.
#if defined(__HIPCC__)
inline unsigned device_cus() {
  static unsigned cus[16] = {0};      // per device ordinal; a benign race writes the same value twice
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256u;
  if (!cus[dev]) { int n = 0; cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? (unsigned)n : 256u; }
  return cus[dev];
}
#else
inline unsigned device_cus() { return 256u; }
#endif
inline unsigned isolate_bits(unsigned blocks) { return (lds_isolation() && blocks <= device_cus() * 7u / 8u) ? 40u * 1024u : 0u; }      // 256-thread workgroups of k_spend_bits
inline unsigned isolate_role(unsigned blocks) { return (lds_isolation() && blocks <= device_cus() / 36u) ? 60u * 1024u : 0u; }        // 64-thread workgroups, several such kernels at once
inline unsigned isolate_roles(unsigned blocks) { return (lds_isolation() && blocks <= device_cus() / 8u) ? 60u * 1024u : 0u; }        // a role-block kernel (blockIdx.y = role) of a tiny call

void launch_spend_tail(const SpendArgs& a, hipStream_t s);
void launch_sign_commit(const uint8_t* status, uint8_t* held, uint8_t* out, uint32_t n, uint32_t rec_bytes, hipStream_t s);      // out[p] = status[p] == 0 ? held[p] : 0; held wiped
void launch_spend_tail_k(const SpendArgs& a, hipStream_t s);      // the tail in two launches: K', X_A ...
void launch_spend_tail_c(const SpendArgs& a, hipStream_t s);      // ... then C (spend_lanes.h spend_tail_k_lane / spend_tail_c_lane)
void launch_spend_finish(const SpendArgs& a, hipStream_t s);
void launch_sign_a(const SignArgs& a, hipStream_t s);
void launch_sign_b(const SignArgs& a, hipStream_t s);
void launch_sign_xa(const SignXaArgs& a, hipStream_t s);
void launch_sign_fused(const SignFusedArgs& a, bool check, hipStream_t s);
void launch_issue_a(const IssueArgs& a, hipStream_t s);
void launch_issue_check(const IssueArgs& a, hipStream_t s);
void launch_request_a(const RequestArgs& a, hipStream_t s);
void launch_request_b(const RequestArgs& a, hipStream_t s);
void launch_request_fused(const RequestArgs& a, hipStream_t s);      // a.n <= any: 64 requests per block of four wavefronts, hash in-kernel (tiny calls)
void launch_prove_head(const ProveArgs& a, hipStream_t s);
void launch_prove_bits(const ProveArgs& a, hipStream_t s);
void launch_prove_enc(const ProveArgs& a, hipStream_t s);
void launch_prove_tail(const ProveArgs& a, hipStream_t s);
void launch_prove_resp(const ProveArgs& a, hipStream_t s);
void launch_client_decode_com(const ClientArgs& a, hipStream_t s);
void launch_client_a(const ClientArgs& a, hipStream_t s);
void launch_client_b(const ClientArgs& a, hipStream_t s);
#endif

// ---- fixed-base products of SECRET scalars (msm.h "secret scalars") ---------------------------------------------------------
// Kernel code writes   ACT_SECRET_FB(fb, a.P);  fb.stage(BASE_G);  x = fb.mul(acc, BASE_G, e);
//   IssuerFb           the issuer's secrets (signing nonces e, alpha; key generation), in EVERY build;
//   ACT_SECRET_FB      the client's secrets in the per-proof kernels (prover head / tail, request);
//   ACT_SECRET_FB_LDS  the same for k_prove_bits (256-thread blocks).
// In the default build all three are the matrix-core look-up (msm.h fixed_base_acc_mf: 64-entry windows, 37 additions per product;
// stage() is nothing) and EVERY lane of a wavefront must call mul() together, live or not -- the table operand's rows come from all
// 64 lanes -- which is why the lane bodies multiply in every lane and drop the results of the ones past the batch.  In `make fast`
// the two client forms are fixed_base_acc on the context's wide tables (addressed look-ups).  Rounds 2-3: masked scans of 8-entry
// windows from global memory / staged in LDS (-DACT_CT_GLOBAL_SCAN / -DACT_CT_LDS_SCAN keep them for A/B: 64 additions per product).
#if defined(__HIPCC__)
#if defined(ACT_CT_GLOBAL_SCAN)
struct IssuerFb {          // round 3's form, kept for same-box A/B: 8-entry windows scanned where the table lies in global memory
  const DevParams& P;
  __device__ __forceinline__ ge mul(const ge& acc, int base, const sc& s) const { return fixed_base_acc_ct(acc, P.tab_ct[base], s); }
};
#else
struct IssuerFb {          // 64-entry windows picked on the matrix cores (msm.h fixed_base_acc_mf): EVERY lane of a wavefront calls mul()
  const DevParams& P;
  __device__ __forceinline__ ge mul(const ge& acc, int base, const sc& s) const { return fixed_base_acc_mf(acc, P.tab_mf[base], s); }
};
#endif
#if defined(ACT_CT_SECRET_TABLES)
struct SecretFb {
  const DevParams& P;
  __device__ __forceinline__ void stage(int) {}
#if defined(ACT_CT_GLOBAL_SCAN)
  __device__ __forceinline__ ge mul(const ge& acc, int base, const sc& s) const { return fixed_base_acc_ct(acc, P.tab_ct[base], s); }
#else
  __device__ __forceinline__ ge mul(const ge& acc, int base, const sc& s) const { return fixed_base_acc_mf(acc, P.tab_mf[base], s); }
#endif
};
#if defined(ACT_CT_LDS_SCAN)
// round 3's form, kept for same-box A/B (make ct CTFLAGS=-DACT_CT_LDS_SCAN): the block stages the 64 KiB scanned table in LDS
struct SecretFbLds {
  uint32_t* lds; const DevParams& P;
  __device__ __forceinline__ void stage(int base) {
    __syncthreads();                                                           // the previous base's readers are done
    const uint4* src = reinterpret_cast<const uint4*>(P.tab_ct[base]);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(CT_TABLE_WORDS / 4); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  __device__ __forceinline__ ge mul(const ge& acc, int, const sc& s) const { return fixed_base_acc_ct(acc, lds, s); }
};
#define ACT_SECRET_FB_LDS(name, P) __shared__ uint32_t name##_lds_[CT_TABLE_WORDS]; SecretFbLds name{name##_lds_, P}
#else
// the range kernel of the ct build: 64-entry windows picked on the matrix cores (msm.h fixed_base_acc_mf).  Every lane of a
// wavefront must call mul() together (non-live lanes with zero scalars): the table operand's rows come from all 64 lanes.
struct SecretFbMf {
  const DevParams& P;
  __device__ __forceinline__ void stage(int) {}
  __device__ __forceinline__ ge mul(const ge& acc, int base, const sc& s) const { return fixed_base_acc_mf(acc, P.tab_mf[base], s); }
};
#define ACT_SECRET_FB_LDS(name, P) SecretFbMf name{P}
#endif
#define ACT_SECRET_FB(name, P) SecretFb name{P}
#else
struct SecretFb {
  const DevParams& P;
  __device__ __forceinline__ void stage(int) {}
  __device__ __forceinline__ ge mul(const ge& acc, int base, const sc& s) const { return fixed_base_acc(acc, P.tab[base], s); }
};
#define ACT_SECRET_FB(name, P) SecretFb name{P}
#define ACT_SECRET_FB_LDS(name, P) SecretFb name{P}
#endif
#endif

// ---- record / transcript access helpers (uint4 / uint2 accesses on the device; memcpy in the g++ test build) ----
ACT_HD void load8(uint32_t w[8], const uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
#else
  memcpy(w, p, 32);
#endif
}
ACT_HD void store8(uint8_t* p, const uint32_t w[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]);
#else
  memcpy(p, w, 32);
#endif
}
ACT_HD void zero8(uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint4* q = reinterpret_cast<uint4*>(p); q[0] = make_uint4(0, 0, 0, 0); q[1] = make_uint4(0, 0, 0, 0);
#else
  memset(p, 0, 32);
#endif
}
ACT_HD sc load_sc(const uint8_t* p) { uint32_t w[8]; load8(w, p); return sc_from_words(w); }
ACT_HD sc load_wide(const uint8_t* p) {   // Scalar::random: 64 rng bytes -> mod l
  uint32_t w[16]; load8(w, p); load8(w + 8, p + 32); return sc_from_wide_words(w);
}
ACT_HD void store_sc(uint8_t* p, const sc& s) { store8(p, s.v); }
// one transcript element = u64_be(32) | 32 payload bytes (src/transcript.rs:95-98); `slot` 8-byte aligned
ACT_HD void tr_put_aligned(uint8_t* slot, const uint32_t w[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint2* q = reinterpret_cast<uint2*>(slot);
  q[0] = make_uint2(0u, 0x20000000u);
  q[1] = make_uint2(w[0], w[1]); q[2] = make_uint2(w[2], w[3]); q[3] = make_uint2(w[4], w[5]); q[4] = make_uint2(w[6], w[7]);
#else
  const uint32_t head[2] = {0u, 0x20000000u};
  memcpy(slot, head, 8); memcpy(slot + 8, w, 32);
#endif
}
// same at an arbitrary byte offset (the small transcripts have 185/186-byte prefixes)
ACT_HD void tr_put_bytes(uint8_t* slot, const uint32_t w[8]) {
  for (int i = 0; i < 7; i++) slot[i] = 0;
  slot[7] = 0x20;
  for (int i = 0; i < 8; i++) { uint32_t v = w[i]; slot[8 + 4 * i] = (uint8_t)v; slot[9 + 4 * i] = (uint8_t)(v >> 8); slot[10 + 4 * i] = (uint8_t)(v >> 16); slot[11 + 4 * i] = (uint8_t)(v >> 24); }
}
ACT_HD void tr_put_prefix(uint8_t* tr, const DevParams& P, int label) {
  uint32_t* q = reinterpret_cast<uint32_t*>(tr);
  for (int i = 0; i < PREFIX_WORDS; i++) if (4u * i < P.prefix_len[label]) q[i] = P.prefix[label][i];
}
// per-proof flag words are OR-ed by several lanes of one launch
#if defined(__HIP_DEVICE_COMPILE__)
#define ACT_FLAG_OR(ptr, v) atomicOr((ptr), (v))
#else
#define ACT_FLAG_OR(ptr, v) ((void)(*(ptr) |= (v)))
#endif

}  // namespace act
