// k_sign.hip — the issuer's BBS signing tail shared by PrivateKey::issue and PrivateKey::refund
// (/root/reference/src/lib.rs:643-660 and :846-861), the issuance-request PoK check (:629-640) and the
// client's PreIssuance::request (:463-487).  Each function is "phase A -> transcript hash -> phase B"
// (SURVEY.md fact 0.9); the hash runs on the host or in k_hash_xof depending on the context's mode.
#include "kernels.h"

namespace act {

// ---- sign, phase A: e, alpha <- rng;  A = (e+x)^-1 X_A;  X_g = e g + w;  Y_A = alpha A;  Y_g = alpha g --------
// A and Y_A share X_A's doubling chain: Y_A = (alpha (e+x)^-1) X_A.
__global__ void __launch_bounds__(64, 2) k_sign_a(SignArgs a) {
  IssuerFb fb{a.P};                                               // nonces and key: no digit of them ever selects an address, in either build
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  const bool live = p < a.n && a.status[p < a.n ? p : 0] == 0;    // rng is drawn only after verification (:638-643, :842-846)
  sc e = sc_zero(), alpha = sc_zero();
  ge xa = ge_identity();
  ge acc[2] = {ge_identity(), ge_identity()};
  if (live) {
    const uint8_t* rng = a.rng + (size_t)a.rng_slot[p] * 128;
    e = load_wide(rng); alpha = load_wide(rng + 64);              // :643/:649, :846/:852
    sc inv = sc_invert(sc_add(e, a.K.x));                         // :645 / :849
    xa = ge_load(a.xa + (size_t)p * GE_WORDS);
    sc s[2] = {inv, sc_mul(alpha, inv)};
    chain_ct<2>(acc, xa, s);                                                                              // acc[0] = A, acc[1] = Y_A (:650 / :853)
  }
  // the two products on g run in every lane of the wavefront, signing or not (IssuerFb: the matrix-core look-up takes its table
  // operand from all 64 lanes; a lane that does not sign multiplies by zero)
  ge xg = ge_add(fb.mul(ge_identity(), BASE_G, e), a.K.w);        // :646 / :851
  ge yg = fb.mul(ge_identity(), BASE_G, alpha);                   // :651 / :854
  if (!live) return;

  // transcript: prefix | [c] | e | A | X_A | X_g | Y_A | Y_g   (:654-657 / :856-859)
  uint8_t* tr = a.trs + (size_t)p * SMALL_TR_STRIDE;
  tr_put_prefix(tr, a.P, a.label);
  uint8_t* el = tr + a.P.prefix_len[a.label];
  if (a.label == LABEL_RESPOND) { sc c = load_sc(a.c_amount + (size_t)p * 32); tr_put_bytes(el, c.v); el += 40; }
  tr_put_bytes(el, e.v); el += 40;
  uint32_t enc[8], enc_a[8];
  ristretto_encode(enc_a, acc[0]); tr_put_bytes(el, enc_a); el += 40;
  ristretto_encode(enc, xa); tr_put_bytes(el, enc); el += 40;
  ristretto_encode(enc, xg); tr_put_bytes(el, enc); el += 40;
  ristretto_encode(enc, acc[1]); tr_put_bytes(el, enc); el += 40;
  ristretto_encode(enc, yg); tr_put_bytes(el, enc);
  uint32_t* stt = a.state + (size_t)p * 24;
  for (int i = 0; i < 8; i++) { stt[i] = e.v[i]; stt[8 + i] = alpha.v[i]; stt[16 + i] = enc_a[i]; }
}
// ---- sign, phase B: z = gamma (x + e) + alpha; write Refund {A,e,gamma,z} or IssuanceResponse {A,e,gamma,z,c} ----
__global__ void __launch_bounds__(256) k_sign_b(SignArgs a) {
  uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  const int rec = a.label == LABEL_RESPOND ? 160 : 128;
  uint8_t* out = a.out + (size_t)p * rec;
  if (a.status[p] != 0) { for (int i = 0; i < rec; i += 32) zero8(out + i); return; }
  const uint32_t* stt = a.state + (size_t)p * 24;
  sc e, alpha; uint32_t enc_a[8], w[16];
  for (int i = 0; i < 8; i++) { e.v[i] = stt[i]; alpha.v[i] = stt[8 + i]; enc_a[i] = stt[16 + i]; }
  for (int i = 0; i < 16; i++) w[i] = a.xof[(size_t)p * 16 + i];
  sc gamma = sc_from_wide_words(w);
  sc z = sc_muladd(gamma, sc_add(a.K.x, e), alpha);               // :660 / :861
  store8(out, enc_a); store_sc(out + 32, e); store_sc(out + 64, gamma); store_sc(out + 96, z);
  if (a.label == LABEL_RESPOND) store_sc(out + 128, load_sc(a.c_amount + (size_t)p * 32));
}
// The same phase for SHORT launches, four WAVEFRONTS per 64 signatures.  One lane per signature is ~6 200 dependent field operations
// (2.2 ms) however few signatures there are -- the issuer's single-item calls (one `issue`, one `refund`) are exactly that.  The
// five points of the transcript do not depend on each other, so they are spread over the wavefronts of a block (wavefront = role,
// lane = signature):
//     wave 0   A   = (e+x)^-1 X_A          wave 1   Y_A = (alpha (e+x)^-1) X_A        (a doubling chain each instead of a shared one)
//     wave 2   X_g = e g + w, and X_A (encode only)                                    wave 3   Y_g = alpha g
// Same values, same bytes; the longest wavefront is one chain + one encode (~3 500 operations).  Round 4 gave the roles to LANES of
// one wavefront (eight lanes per signature): divergent lanes of a wavefront run one after the other, so the two chains took 2 x 0.7 ms
// (profiles/r05_single_item_kernels.txt: k_sign_a 1.43 - 1.79 ms for one signature) -- the roles have to sit on different SIMDs to
// overlap.  Four wavefronts, not five: a 256-thread block leaves every wavefront a SIMD's whole register file, which these chains
// need (five wavefronts = two on one SIMD = half the registers each: the kernel spilled and took 2.9 ms).
__global__ void __launch_bounds__(64) k_sign_a_wide(SignArgs a) {
  const uint32_t role = blockIdx.y, p = blockIdx.x * 64 + threadIdx.x;      // role is uniform over the block (one wavefront): no branch below diverges
  const bool live = p < a.n && a.status[p < a.n ? p : 0] == 0;    // rng is drawn only after verification (:638-643, :842-846)
  sc e = sc_zero(), alpha = sc_zero();
  if (live) {
    const uint8_t* rng = a.rng + (size_t)a.rng_slot[p] * 128;
    e = load_wide(rng); alpha = load_wide(rng + 64);              // :643/:649, :846/:852
  }
  ge pt = ge_identity();
  if (role >= 2) {
    // the product on g runs in every lane of these two wavefronts (IssuerFb: the matrix-core look-up takes its table operand from all
    // 64 lanes); lanes that do not sign multiply by zero
    IssuerFb fb{a.P};
    pt = fb.mul(ge_identity(), BASE_G, role == 2 ? e : alpha);    // role 3: Y_g (:651 / :854)
    if (role == 2) pt = ge_add(pt, a.K.w);                        // X_g = e g + w (:646 / :851)
  }
  if (!live) return;
  // transcript: prefix | [c] | e | A | X_A | X_g | Y_A | Y_g   (:654-657 / :856-859)
  uint8_t* tr = a.trs + (size_t)p * SMALL_TR_STRIDE;
  uint8_t* el = tr + a.P.prefix_len[a.label] + (a.label == LABEL_RESPOND ? 40 : 0);
  uint32_t enc[8];
  if (role < 2) {
    const ge xa = ge_load(a.xa + (size_t)p * GE_WORDS);
    const sc inv = sc_invert(sc_add(e, a.K.x));                   // :645 / :849
    sc s1[1] = {role == 0 ? inv : sc_mul(alpha, inv)};
    ge acc[1] = {ge_identity()};
    chain_ct<1>(acc, xa, s1);                                     // A, or Y_A = (alpha (e+x)^-1) X_A (:650 / :853)
    pt = acc[0];
  } else if (role == 2) {
    ristretto_encode(enc, ge_load(a.xa + (size_t)p * GE_WORDS));  // X_A
    tr_put_bytes(el + 40 * 2, enc);
  }
  ristretto_encode(enc, pt);
  const int slot = role == 0 ? 1 : role == 2 ? 3 : role == 1 ? 4 : 5;      // after e (slot 0); X_A is slot 2
  tr_put_bytes(el + 40 * slot, enc);
  if (role == 0) {
    tr_put_prefix(tr, a.P, a.label);
    if (a.label == LABEL_RESPOND) { sc c = load_sc(a.c_amount + (size_t)p * 32); tr_put_bytes(tr + a.P.prefix_len[a.label], c.v); }
    tr_put_bytes(el, e.v);
    uint32_t* stt = a.state + (size_t)p * 24;
    for (int i = 0; i < 8; i++) { stt[i] = e.v[i]; stt[8 + i] = alpha.v[i]; stt[16 + i] = enc[i]; }
  }
}
constexpr uint32_t SIGN_WIDE_MAX = 8192;
void launch_sign_a(const SignArgs& a, hipStream_t s) {
  if (!a.n) return;
  const bool no_wide = tune(T_NO_WIDE_SIGN) != 0;     // A/B knob (act_tuning_set)
  if (a.n <= SIGN_WIDE_MAX && !no_wide) hipLaunchKernelGGL(k_sign_a_wide, dim3((a.n + 63) / 64, 4), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(k_sign_a, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
}
void launch_sign_b(const SignArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_sign_b, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }

// ---- issue, phase A: K1 = k_bar h2 + r_bar h3 - gamma K (:629-630); X_A = g + c h1 + K (:644) --------------
__global__ void __launch_bounds__(64, 2) k_issue_a(IssueArgs a) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= a.n) return;
  const uint8_t* rec = a.req + (size_t)p * 128;
  uint32_t wk[8]; load8(wk, rec);
  ge K; uint32_t flags = ristretto_decode(K, wk) ? 0u : FLAG_UNDECODABLE;
  sc gamma = load_sc(rec + 32), k_bar = load_sc(rec + 64), r_bar = load_sc(rec + 96);
  ge acc[1];
  acc[0] = fixed_base_acc(ge_identity(), a.P.tab[BASE_H2], k_bar);
  acc[0] = fixed_base_acc(acc[0], a.P.tab[BASE_H3], r_bar);
  sc s[1] = {sc_neg(gamma)};
  chain_b<1>(acc, K, s, a.pbk + (size_t)p * 2 * BUCKET_WORDS);
  uint8_t* tr = a.trs + (size_t)p * SMALL_TR_STRIDE;
  tr_put_prefix(tr, a.P, LABEL_REQUEST);
  uint8_t* el = tr + a.P.prefix_len[LABEL_REQUEST];
  uint32_t enc[8];
  tr_put_bytes(el, wk);                                           // :634 big_k (canonical bytes)
  ristretto_encode(enc, acc[0]); tr_put_bytes(el + 40, enc);      // k1
  if (a.c_amount) {                                               // check-only callers (act_issue_check_batch) pass no amounts
    sc c = load_sc(a.c_amount + (size_t)p * 32);
    ge xa = ge_add(fixed_base_acc(ge_basepoint(), a.P.tab[BASE_H1], c), K);
    ge_store(a.xa + (size_t)p * GE_WORDS, xa);
  }
  a.flags[p] = flags;
}
// X_A alone, for the sign-only entry points (act_issue_sign_batch / act_refund_sign_batch: the node dispatcher's second
// phase): issue X_A = g + c h1 + K (:644) from the request, refund X_A = g + K' (:848) from the enc(K') that
// act_verify_spend_batch returned.  Lanes whose status is already non-zero are left alone.
__global__ void __launch_bounds__(64, 2) k_sign_xa(SignXaArgs a) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= a.n || a.status[p] != 0) return;
  uint32_t w[8]; load8(w, a.point + (size_t)p * a.point_stride);
  ge K;
  if (!ristretto_decode(K, w)) { a.status[p] = 255; return; }
  ge xa = a.c_amount ? ge_add(fixed_base_acc(ge_basepoint(), a.P.tab[BASE_H1], load_sc(a.c_amount + (size_t)p * 32)), K) : ge_add(K, ge_basepoint());
  ge_store(a.xa + (size_t)p * GE_WORDS, xa);
}
void launch_sign_xa(const SignXaArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_sign_xa, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }
__global__ void __launch_bounds__(256) k_issue_check(IssueArgs a) {
  uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  sc gamma = load_sc(a.req + (size_t)p * 128 + 32);
  uint32_t w[16];
  for (int i = 0; i < 16; i++) w[i] = a.xof[(size_t)p * 16 + i];
  uint8_t stt = 0;
  if (a.flags[p] & FLAG_UNDECODABLE) stt = 255;
  else if (!sc_equal(sc_from_wide_words(w), gamma)) stt = 1;      // Error::InvalidIssuanceRequestProof (:638-640)
  a.status[p] = stt;
}
void launch_issue_a(const IssueArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_issue_a, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }
void launch_issue_check(const IssueArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_issue_check, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }

// ---- PreIssuance::request (:463-487) ---------------------------------------------------------------------
__global__ void __launch_bounds__(64, 2) k_request_a(RequestArgs a) {
  ACT_SECRET_FB(fb, a.P);
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  const bool live = p < a.n;
  sc r = sc_zero(), k = sc_zero(), kp = sc_zero(), rp = sc_zero();
  if (live) {
    r = load_sc(a.pre + (size_t)p * 64); k = load_sc(a.pre + (size_t)p * 64 + 32);
    kp = load_wide(a.rng + (size_t)p * 128); rp = load_wide(a.rng + (size_t)p * 128 + 64);        // :468-469
  }
  // K = k h2 + r h3 (:465), K1 = k' h2 + r' h3 (:470): the products grouped by base (one staged table at a time in the ct build)
  ge big_k = ge_identity(), k1 = ge_identity();
  fb.stage(BASE_H2);                                              // (every lane multiplies, live or not: prove_lanes.h prove_head_lane)
  big_k = fb.mul(big_k, BASE_H2, k); k1 = fb.mul(k1, BASE_H2, kp);
  fb.stage(BASE_H3);
  big_k = fb.mul(big_k, BASE_H3, r); k1 = fb.mul(k1, BASE_H3, rp);
  if (!live) return;
  uint8_t* tr = a.trs + (size_t)p * SMALL_TR_STRIDE;
  tr_put_prefix(tr, a.P, LABEL_REQUEST);
  uint8_t* el = tr + a.P.prefix_len[LABEL_REQUEST];
  uint32_t enc[8];
  ristretto_encode(enc, big_k); tr_put_bytes(el, enc); store8(a.out + (size_t)p * 128, enc);
  ristretto_encode(enc, k1); tr_put_bytes(el + 40, enc);
}
__global__ void __launch_bounds__(256) k_request_b(RequestArgs a) {
  uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  sc r = load_sc(a.pre + (size_t)p * 64), k = load_sc(a.pre + (size_t)p * 64 + 32);
  sc kp = load_wide(a.rng + (size_t)p * 128), rp = load_wide(a.rng + (size_t)p * 128 + 64);
  uint32_t w[16];
  for (int i = 0; i < 16; i++) w[i] = a.xof[(size_t)p * 16 + i];
  sc gamma = sc_from_wide_words(w);                               // :473-475
  uint8_t* out = a.out + (size_t)p * 128;
  store_sc(out + 32, gamma);
  store_sc(out + 64, sc_muladd(k, gamma, kp));                    // k_bar = k' + k gamma (:478)
  store_sc(out + 96, sc_muladd(r, gamma, rp));                    // r_bar = r' + r gamma (:479)
}
// ---- PrivateKey::issue / the signing half of issue and refund for TINY calls: ONE kernel -------------------------------------------
// Roles are BLOCKS of one wavefront (blockIdx.y = role, lane = item; blockIdx.x = group of 64 items):
//     role C   (CHECK only) the request's PoK: K1 = k_bar h2 + r_bar h3 - gamma K (:629-630), the 266-byte "request" transcript, its
//              BLAKE3, gamma ?= challenge (:638-640) -> status
//     roles A0-A3   the four quarters of A = (e+x)^-1 X_A        roles Y0-Y3   the four quarters of Y_A = (alpha (e+x)^-1) X_A
//                   (msm.h chain_ct_quarter: 64 i doublings + 33 chain steps each instead of 127 steps in one wavefront)
//     role M        X_g = e g + w,  Y_g = alpha g,  X_A (encode only), and the head of the transcript
// the partial points parked in the lane's bucket area, the encodings written into its "respond" / "refund" transcript (global
// memory, as k_sign_a_wide does).  The A quarters meet in whichever of their four blocks arrives last (a counter of their own), which
// adds them and encodes A while the Y_A blocks do the same for Y_A; the block of the group that ARRIVES LAST of all (a counter per
// group) hashes the transcript -- one BLAKE3 chunk, the routine of k_hash_xof -- and finishes z = gamma (x + e) + alpha (:660 / :861).  X_A = g + c h1 + K (:644) or g + K' (:848) is cheap (one table product)
// and every role that needs it makes its own.
// Why blocks and not the wavefronts of one block: where the wavefronts of a workgroup land is the dispatcher's business, and roles
// that land on one SIMD run at half speed each -- the same kernel took 0.9 or 2.0 ms from call to call (profiles/r05_tiny_ab.txt:
// min / median of 30 calls); separate workgroups go to separate CUs.
// With CHECK the signature is computed NEXT TO the check instead of behind it -- a lane's e, alpha are then read from its rng slice
// before its verdict is known, which changes nothing observable: a rejected lane's record is zero, its nonces never leave the
// registers, and the transcript the roles assemble for it in `trs` -- which does hold e and enc(A), a signature over a K nobody has
// verified -- is wiped with the call (engine.hip finish_call, Slot::d_trs_dirty; act_debug_secret_residue reads it back).  The engine
// takes this kernel only when the slice a lane would draw does not depend on other lanes' verdicts
// (ACT_RNG_PER_LANE, or one lane).  One launch instead of six; the dependent chain is max(check, signature) instead of their sum.
// Without CHECK and with `before_verdict` the same holds for a tiny refund, whose check is the whole spend-proof verification on other
// streams: every lane is signed into a buffer of the engine's and k_sign_commit hands out what the verdicts allow (small_impl.inc).
constexpr uint32_t FUSED_CTR_A = 128, FUSED_CTR_Y = 256;      // k_sign_fused: the A quarters' and the Y_A quarters' own counters, beside the group's (engine.hip: 512 words per slot)
__device__ __forceinline__ bool group_last_arrival(uint32_t* counter, uint32_t roles) {
  __shared__ uint32_t ticket;
  __threadfence();                                             // this block's global writes are visible before its arrival is
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicAdd(counter, 1u);
  __syncthreads();
  const bool last = ticket == roles - 1u;
  if (last) { __threadfence(); if (threadIdx.x == 0) *counter = 0u; }      // the counter is ready for the next launch
  return last;
}
// X_A of lane p: handed over in extended coordinates (the spend path's k_spend_tail made it), or g + c h1 + K (:644) / g + K' (:848)
// from the encoded point of the lane's record
__device__ __forceinline__ ge sign_fused_xa(const SignFusedArgs& a, uint32_t p, const uint8_t* rec, bool& dec_ok) {
  dec_ok = true;
  if (a.xa) return ge_load(a.xa + (size_t)p * GE_WORDS);
  uint32_t wk[8]; load8(wk, rec);
  ge K; dec_ok = ristretto_decode(K, wk);
  return a.c_amount ? ge_add(fixed_base_acc(ge_basepoint(), a.P.tab[BASE_H1], load_sc(a.c_amount + (size_t)p * 32)), K) : ge_add(K, ge_basepoint());
}
#if defined(ACT_TINY_TIMING)      // measurement build only (tools/tiny_timing.sh): where a role's time goes, in ticks of the 100 MHz wall clock
#define ACT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && a.dbg) a.dbg[(blockIdx.y + (CHECK ? 0 : 1)) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define ACT_STAMP(k) do { } while (0)
#endif
template <bool CHECK>
__global__ void __launch_bounds__(64) k_sign_fused(SignFusedArgs a) {
  const uint32_t lane = threadIdx.x, p = blockIdx.x * 64 + lane;
  ACT_STAMP(0);
  // roles: 0 = the check; 1..4 = the quarters of A; 5..8 = the quarters of Y_A (msm.h chain_ct_quarter); 9 = X_g, Y_g, X_A and the
  // transcript's head
  enum { R_CHECK = 0, R_A0 = 1, R_Y0 = 5, R_M = 9, ROLES = 10 };
  const uint32_t role = blockIdx.y + (CHECK ? 0u : 1u);        // uniform over the block
  const bool in = p < a.n;
  const bool spec = !CHECK && a.before_verdict;                // every lane is signed, no verdict is looked at (see kernels.h)
  const bool sign = in && (CHECK || spec || a.status_in[in ? p : 0] == 0);
  const uint8_t* rec = a.point ? a.point + (size_t)(in ? p : 0) * a.point_stride : nullptr;
  const uint8_t* rng = a.rng + (size_t)((in && a.rng_slot) ? a.rng_slot[p] : (in ? p : 0)) * 128;
  sc e = sc_zero(), alpha = sc_zero();
  if (sign && role != R_CHECK) { e = load_wide(rng); alpha = load_wide(rng + 64); }      // :643/:649, :846/:852
  uint8_t* t2 = a.trs + (size_t)(in ? p : 0) * SMALL_TR_STRIDE;
  const uint32_t plen2 = a.P.prefix_len[a.label];
  uint8_t* el2 = t2 + plen2 + (a.label == LABEL_RESPOND ? 40 : 0);
  // per lane: bucket sets 0, 1 = the check's chain (msm.h chain_b), set 2 = the eight partial points of A and Y_A
  uint32_t* park = a.pbk + ((size_t)(in ? p : 0) * PREP_BUCKET_SETS + 2) * BUCKET_WORDS;
  // transcript: prefix | [c] | e | A | X_A | X_g | Y_A | Y_g   (:654-657 / :856-859)
  uint32_t enc[8];
  if (role == R_CHECK) {
    if (CHECK && in) {
      uint32_t wk[8]; load8(wk, rec);
      ge K; const bool ok = ristretto_decode(K, wk);
      const sc gamma = load_sc(rec + 32), k_bar = load_sc(rec + 64), r_bar = load_sc(rec + 96);
      ge acc[1];
      acc[0] = fixed_base_acc(ge_identity(), a.P.tab[BASE_H2], k_bar);
      acc[0] = fixed_base_acc(acc[0], a.P.tab[BASE_H3], r_bar);
      sc s1[1] = {sc_neg(gamma)};
      chain_b<1>(acc, K, s1, a.pbk + (size_t)p * PREP_BUCKET_SETS * BUCKET_WORDS);
      uint8_t* t1 = a.trs_req + (size_t)p * SMALL_TR_STRIDE;
      tr_put_prefix(t1, a.P, LABEL_REQUEST);
      uint8_t* el = t1 + a.P.prefix_len[LABEL_REQUEST];
      tr_put_bytes(el, wk);                                      // :634 big_k (canonical bytes)
      ristretto_encode(enc, acc[0]); tr_put_bytes(el + 40, enc); // k1
      uint32_t w[16];
      b3_hash_xof64(w, reinterpret_cast<const uint32_t*>(t1), a.P.prefix_len[LABEL_REQUEST] + 80u);
      a.status[p] = !ok ? (uint8_t)255 : sc_equal(sc_from_wide_words(w), gamma) ? (uint8_t)0 : (uint8_t)1;      // Error::InvalidIssuanceRequestProof (:638-640)
      if (a.check_only && a.wipe_rng) { uint8_t* q = const_cast<uint8_t*>(rec); for (uint32_t i = 0; i < a.point_stride; i += 32) zero8(q + i); }
    }
    if (a.check_only) return;                                  // act_issue_check_batch on a tiny call: this role is the whole launch
  } else if (role == R_M) {
    IssuerFb fb{a.P};                                          // the whole wavefront multiplies: lanes that do not sign by zero
    const ge xg = ge_add(fb.mul(ge_identity(), BASE_G, e), a.K.w);   // X_g = e g + w (:646 / :851)
    const ge yg = fb.mul(ge_identity(), BASE_G, alpha);              // Y_g (:651 / :854)
    if (sign) {
      // the head of the transcript (bytes, not tr_put_prefix's word stores, whose last word reaches into the first element)
      const uint8_t* pre = reinterpret_cast<const uint8_t*>(a.P.prefix[a.label]);
      for (uint32_t i = 0; i < plen2; i++) t2[i] = pre[i];
      if (a.label == LABEL_RESPOND) { const sc c = load_sc(a.c_amount + (size_t)p * 32); tr_put_bytes(t2 + plen2, c.v); }
      tr_put_bytes(el2, e.v);
      ristretto_encode(enc, xg); tr_put_bytes(el2 + 40 * 3, enc);
      ristretto_encode(enc, yg); tr_put_bytes(el2 + 40 * 5, enc);
      bool dec_ok;
      const ge xa = sign_fused_xa(a, p, rec, dec_ok);
      if (!dec_ok && !CHECK) a.status[p] = 255;                  // (with CHECK the check role says so)
      ristretto_encode(enc, xa); tr_put_bytes(el2 + 40 * 2, enc);
    }
  } else if (sign) {
    bool dec_ok;
    const ge xa = sign_fused_xa(a, p, rec, dec_ok);
    ACT_STAMP(1);
    const sc inv = sc_invert(sc_add(e, a.K.x));                 // :645 / :849
    ACT_STAMP(2);
    const bool is_a = role < R_Y0;
    const int q = (int)(role - (is_a ? R_A0 : R_Y0));
    // a quarter of A = inv X_A, or of Y_A = (alpha inv) X_A (:650 / :853)
    ge_store(park + (size_t)(role - R_A0) * GE_WORDS, chain_ct_quarter(xa, is_a ? inv : sc_mul(alpha, inv), q));
  }
  ACT_STAMP(3);
  // The four quarters of A meet in whichever of their blocks arrives last (a counter of their own), which adds them and encodes A
  // while the Y_A blocks do the same for Y_A: two 0.1 ms encodings side by side instead of one after the other in the final block.
  uint32_t* const enc_park = park + 8 * GE_WORDS;              // enc(A) for the record, words 288..295 of the lane's set (BUCKET_WORDS = 324)
  if (role >= R_A0 && role < R_M) {
    const bool is_a = role < R_Y0;
    if (group_last_arrival(a.group_counter + (is_a ? FUSED_CTR_A : FUSED_CTR_Y) + blockIdx.x, 4u) && sign) {
      const uint32_t* q0 = park + (size_t)(is_a ? 0 : 4) * GE_WORDS;
      ge sum = ge_load(q0);
      for (int q = 1; q < 4; q++) sum = ge_add(sum, ge_load(q0 + (size_t)q * GE_WORDS));
      ristretto_encode(enc, sum); tr_put_bytes(el2 + 40 * (is_a ? 1 : 4), enc);
      if (is_a) store8(reinterpret_cast<uint8_t*>(enc_park), enc);
    }
  }
  if (!group_last_arrival(a.group_counter + blockIdx.x, CHECK ? (uint32_t)ROLES : (uint32_t)ROLES - 1u)) return;
  ACT_STAMP(4);
  // ---- the last block of the group to arrive: the hash, z, the record -----------------------------------------------------------------
  if (!in) return;
  const int rec_out = a.label == LABEL_RESPOND ? 160 : 128;
  uint8_t* out = a.out + (size_t)p * rec_out;
  const uint8_t v = spec ? (uint8_t)0 : __atomic_load_n(a.status + p, __ATOMIC_RELAXED);      // (written by another block of this launch)
  if (v == 0) {
    uint32_t enc_a[8]; load8(enc_a, reinterpret_cast<const uint8_t*>(enc_park));      // (A and Y_A are in the transcript already)
    ACT_STAMP(5);
    e = load_wide(rng); alpha = load_wide(rng + 64);
    uint32_t w[16];
    b3_hash_xof64(w, reinterpret_cast<const uint32_t*>(t2), plen2 + 40u * (a.label == LABEL_RESPOND ? 7u : 6u));
    const sc gamma = sc_from_wide_words(w);
    const sc z = sc_muladd(gamma, sc_add(a.K.x, e), alpha);     // :660 / :861
    store8(out, enc_a); store_sc(out + 32, e); store_sc(out + 64, gamma); store_sc(out + 96, z);
    if (a.label == LABEL_RESPOND) store_sc(out + 128, load_sc(a.c_amount + (size_t)p * 32));
  } else {
    for (int i = 0; i < rec_out; i += 32) zero8(out + i);
  }
  ACT_STAMP(6);
  if (sign) for (int i = 0; i < 8 * GE_WORDS + 8; i += 4) *reinterpret_cast<uint4*>(park + i) = make_uint4(0, 0, 0, 0);      // the quarters are functions of the nonces
  if (a.wipe_rng) {                                            // the engine's staged copies (issue_tiny): nothing of the call stays behind
    uint8_t* q = const_cast<uint8_t*>(rng); for (int i = 0; i < 128; i += 32) zero8(q + i);
    if (rec) { q = const_cast<uint8_t*>(rec); for (uint32_t i = 0; i < a.point_stride; i += 32) zero8(q + i); }
    if (a.c_amount) zero8(const_cast<uint8_t*>(a.c_amount) + (size_t)p * 32);
  }
}
// a signature computed before its verdict (k_sign_fused with before_verdict) is handed out, or not
__global__ void __launch_bounds__(64) k_sign_commit(const uint8_t* status, uint8_t* held, uint8_t* out, uint32_t n, uint32_t rec_bytes) {
  const uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n) return;
  const bool ok = status[p] == 0;
  for (uint32_t i = 0; i < rec_bytes; i += 16) {
    uint4* h = reinterpret_cast<uint4*>(held + (size_t)p * rec_bytes + i);
    const uint4 v = *h;
    *reinterpret_cast<uint4*>(out + (size_t)p * rec_bytes + i) = ok ? v : make_uint4(0, 0, 0, 0);
    *h = make_uint4(0, 0, 0, 0);
  }
}
void launch_sign_commit(const uint8_t* status, uint8_t* held, uint8_t* out, uint32_t n, uint32_t rec_bytes, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_sign_commit, dim3((n + 63) / 64), dim3(64), 0, s, status, held, out, n, rec_bytes);
}
void launch_sign_fused(const SignFusedArgs& a, bool check, hipStream_t s) {
  if (!a.n) return;
  const unsigned g = (a.n + 63) / 64;
  if (check && a.check_only) { hipLaunchKernelGGL(k_sign_fused<true>, dim3(g, 1), dim3(64), isolate_roles(g), s, a); return; }
  if (check) hipLaunchKernelGGL(k_sign_fused<true>, dim3(g, 10), dim3(64), isolate_roles(g * 10), s, a);
  else hipLaunchKernelGGL(k_sign_fused<false>, dim3(g, 9), dim3(64), isolate_roles(g * 9), s, a);      // (a tiny refund: beside the verification's kernels)
}

// ---- PreIssuance::request for TINY calls: the whole method in ONE kernel -------------------------------------------------------
// One request per call is the crate's call shape (src/lib.rs:463).  As k_request_a -> hash -> k_request_b it was one lane's chain of
// four fixed-base products and two encodings (0.53 ms of a 0.58 ms call, profiles/r05_single_item_kernels.txt) plus two more
// launches.  Here a block of four wavefronts serves 64 requests, wavefront = one of the four products, lane = request:
//     wave 0   k h2        wave 1   r h3        wave 2   k' h2        wave 3   r' h3            (:465, :470)
// then K = wave 0 + wave 1 and K1 = wave 2 + wave 3 through LDS, encoded by waves 0 and 2; wave 0 hashes the 266-byte transcript
// (one BLAKE3 chunk, blake3_hd.h: the same routine k_hash_xof runs, so the same bytes as either transcript mode) and finishes
// gamma, k_bar, r_bar (:473-479).  Nothing secret reaches global memory; the staged inputs are zeroed by the kernel itself.
constexpr int TINY_TR_WORDS = 68;        // a 266-byte "request" transcript rounded up to whole words
__global__ void __launch_bounds__(256) k_request_fused(RequestArgs a) {
  __shared__ uint32_t part[4][64 * GE_WORDS];
  __shared__ uint32_t trs[64][TINY_TR_WORDS];
  ACT_SECRET_FB(fb, a.P);
  const uint32_t role = threadIdx.x >> 6, lane = threadIdx.x & 63u, p = blockIdx.x * 64 + lane;      // role is wave-uniform
  const bool live = p < a.n;
  sc s = sc_zero();
  if (live) {
    const uint8_t* pre = a.pre + (size_t)p * 64; const uint8_t* rng = a.rng + (size_t)p * 128;
    s = role == 0 ? load_sc(pre + 32) : role == 1 ? load_sc(pre) : role == 2 ? load_wide(rng) : load_wide(rng + 64);      // k, r, k', r' (:468-469)
  }
  const int base = (role & 1u) ? BASE_H3 : BASE_H2;
  fb.stage(base);
  ge f = fb.mul(ge_identity(), base, s);                       // every lane multiplies, live or not (the matrix-core look-up is wave-wide)
  ge_store(part[role] + lane * GE_WORDS, f);
  for (int i = (int)threadIdx.x; i < 64 * TINY_TR_WORDS; i += 256) (&trs[0][0])[i] = 0u;
  __syncthreads();
  uint8_t* tr = reinterpret_cast<uint8_t*>(trs[lane]);
  const uint32_t plen = a.P.prefix_len[LABEL_REQUEST];
  if (role == 0) tr_put_prefix(tr, a.P, LABEL_REQUEST);        // (word stores; this lane's own K element, written next, overlaps the last of them)
  if ((role == 0 || role == 2) && live) {
    const ge pt = ge_add(f, ge_load(part[role + 1] + lane * GE_WORDS));
    uint32_t enc[8];
    ristretto_encode(enc, pt);
    tr_put_bytes(tr + plen + (role == 0 ? 0 : 40), enc);       // K, then K1 (:473-474)
    if (role == 0) store8(a.out + (size_t)p * 128, enc);
  }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < 4 * 64 * GE_WORDS; i += 256) (&part[0][0])[i] = 0u;      // the partial sums are functions of the secrets
  if (role != 0 || !live) return;
  uint32_t w[16];
  b3_hash_xof64(w, trs[lane], plen + 80u);
  const sc gamma = sc_from_wide_words(w);                      // :473-475
  const uint8_t* pre = a.pre + (size_t)p * 64; const uint8_t* rng = a.rng + (size_t)p * 128;
  const sc r = load_sc(pre), kp = load_wide(rng), rp = load_wide(rng + 64);
  uint8_t* out = a.out + (size_t)p * 128;
  store_sc(out + 32, gamma);
  store_sc(out + 64, sc_muladd(s, gamma, kp));                 // k_bar = k' + k gamma (:478)   (s = k in wave 0)
  store_sc(out + 96, sc_muladd(r, gamma, rp));                 // r_bar = r' + r gamma (:479)
  if (a.wipe_inputs) {                                         // staged copies of the caller's PreIssuance and rng bytes (engine.hip request_tiny)
    uint8_t* q = const_cast<uint8_t*>(pre); zero8(q); zero8(q + 32);
    q = const_cast<uint8_t*>(rng); for (int i = 0; i < 128; i += 32) zero8(q + i);
  }
}
void launch_request_fused(const RequestArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_request_fused, dim3((a.n + 63) / 64), dim3(256), 0, s, a); }
void launch_request_a(const RequestArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_request_a, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }
void launch_request_b(const RequestArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_request_b, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }

}  // namespace act
