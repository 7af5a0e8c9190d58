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
// The same phase for SHORT launches, eight lanes per signature.  One lane per signature is ~6 200 dependent field operations
// (2.2 ms) however few signatures there are -- the issuer's single-item calls (one `issue`, one `refund`) are exactly that.  The
// five points of the transcript do not depend on each other, so five lanes compute one each and encode it:
//     lane 0   A   = (e+x)^-1 X_A          lane 1   Y_A = (alpha (e+x)^-1) X_A        (a doubling chain each instead of a shared one)
//     lane 2   X_g = e g + w               lane 3   Y_g = alpha g                      lane 4   X_A (encode only)       lanes 5-7 idle
// Same values, same bytes; the longest lane is one chain + one encode (~3 500 operations).  Used below SIGN_WIDE_MAX signatures,
// where even eight lanes per signature leave the chip under-filled.
// (256-lane blocks: a block's four wavefronts land on the four SIMDs of a CU; with 64-lane blocks 512 of them took twice as long as
// 1 024 -- two blocks on one SIMD, other SIMDs idle: profiles/r04_sign_probe.txt)
__global__ void __launch_bounds__(256) k_sign_a_wide(SignArgs a) {
  IssuerFb fb{a.P};
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, p = gid >> 3, role = gid & 7;
  const bool live = p < a.n && a.status[p < a.n ? p : 0] == 0;    // rng is drawn only after verification (:638-643, :842-846)
  sc e = sc_zero(), alpha = sc_zero();
  if (live) {
    const uint8_t* rng = a.rng + (size_t)a.rng_slot[p] * 128;
    e = load_wide(rng); alpha = load_wide(rng + 64);              // :643/:649, :846/:852
  }
  // the product on g runs in every lane of the wavefront (IssuerFb: the matrix-core look-up takes its table operand from all 64
  // lanes); lanes that are not role 2 / 3 of a signing lane multiply by zero
  const sc fs = (live && role == 2) ? e : (live && role == 3) ? alpha : sc_zero();
  ge pt = fb.mul(ge_identity(), BASE_G, fs);                      // role 3: Y_g (:651 / :854)
  if (!live || role > 4) return;
  if (role == 2) pt = ge_add(pt, a.K.w);                          // X_g = e g + w (:646 / :851)
  if (role == 0 || role == 1 || role == 4) {
    const ge xa = ge_load(a.xa + (size_t)p * GE_WORDS);
    if (role == 4) pt = xa;
    else {
      const sc inv = sc_invert(sc_add(e, a.K.x));                 // :645 / :849
      sc s1[1] = {role == 0 ? inv : sc_mul(alpha, inv)};
      ge acc[1] = {ge_identity()};
      chain_ct<1>(acc, xa, s1);                                   // A, or Y_A = (alpha (e+x)^-1) X_A (:650 / :853)
      pt = acc[0];
    }
  }
  // transcript: prefix | [c] | e | A | X_A | X_g | Y_A | Y_g   (:654-657 / :856-859)
  uint8_t* tr = a.trs + (size_t)p * SMALL_TR_STRIDE;
  uint8_t* el = tr + a.P.prefix_len[a.label] + (a.label == LABEL_RESPOND ? 40 : 0);
  uint32_t enc[8];
  ristretto_encode(enc, pt);
  const int slot = role == 0 ? 1 : role == 4 ? 2 : role == 2 ? 3 : role == 1 ? 4 : 5;      // after e (slot 0)
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
  static const bool no_wide = getenv("ACT_NO_WIDE_SIGN") != nullptr;     // A/B knob
  if (a.n <= SIGN_WIDE_MAX && !no_wide) hipLaunchKernelGGL(k_sign_a_wide, dim3((a.n * 8 + 255) / 256), dim3(256), 0, s, a);
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
void launch_request_a(const RequestArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_request_a, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }
void launch_request_b(const RequestArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_request_b, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }

}  // namespace act
