// k_client.hip — the client-side verifiers that close the lifecycle ("next" #1):
// PreIssuance::to_credit_token (/root/reference/src/lib.rs:528-562) and PreRefund::to_credit_token
// (:1217-1253).  Both check the issuer's DLEQ proof:
//   X_g = e g + w;  Y_A = z A - gamma X_A;  Y_g = z g - gamma X_g = (z - gamma e) g - gamma w
// with X_A = g + c h1 + K (issuance) or g + K', K' = sum 2^j Com_j (refund).
#include "kernels.h"

namespace act {

// refund only: decode every Com_j of the spend proof (lane = (proof, bit))
__global__ void __launch_bounds__(256) k_client_decode_com(ClientArgs a) {
  const int L = a.P.L;
  uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  uint32_t p = gid / (uint32_t)L, j = gid % (uint32_t)L;
  if (p >= a.n) return;
  const ProofLayout pl{L};
  uint32_t wc[8]; load8(wc, a.proofs + (size_t)p * pl.bytes() + 32 * pl.com(j));
  ge C;
  if (!ristretto_decode(C, wc)) atomicOr(a.flags + p, FLAG_UNDECODABLE);
  niels_store(a.coords + ((size_t)p * L + j) * NIELS_WORDS, niels_from_affine(C));
}

__global__ void __launch_bounds__(64, 2) k_client_a(ClientArgs a) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p >= a.n) return;
  const int L = a.P.L;
  const bool issuance = a.label == LABEL_RESPOND;
  const uint8_t* resp = a.resp + (size_t)p * (issuance ? 160 : 128);
  uint32_t flags = issuance ? 0u : a.flags[p];
  uint32_t wa[8]; load8(wa, resp);
  ge A; if (!ristretto_decode(A, wa)) flags |= FLAG_UNDECODABLE;
  sc e = load_sc(resp + 32), gamma = load_sc(resp + 64), z = load_sc(resp + 96);
  ge xa; sc c = sc_zero();
  if (issuance) {
    uint32_t wk[8]; load8(wk, a.req + (size_t)p * 128);
    ge K; if (!ristretto_decode(K, wk)) flags |= FLAG_UNDECODABLE;
    c = load_sc(resp + 128);
    xa = ge_add(fixed_base_acc(ge_basepoint(), a.P.tab[BASE_H1], c), K);                   // :536
  } else {
    ge kp = ge_identity();                                                                 // :1224-1230 by Horner
    for (int j = L - 1; j >= 0; j--) { kp = ge_double(kp); kp = ge_madd(kp, niels_load(a.coords + ((size_t)p * L + j) * NIELS_WORDS)); }
    xa = ge_add(kp, ge_basepoint());
  }
  sc ng = sc_neg(gamma);
  ge xg = ge_add(fixed_base_acc(ge_identity(), a.P.tab[BASE_G], e), a.w);                  // :537 / :1232
  ge acc[2];
  acc[0] = ge_identity();                                                                  // Y_A
  acc[1] = fixed_base_acc(ge_identity(), a.P.tab[BASE_G], sc_sub(z, sc_mul(gamma, e)));   // Y_g
  uint32_t* bk = a.pbk + (size_t)p * 2 * BUCKET_WORDS;
  sc s1[1] = {z}; chain_b<1>(acc, A, s1, bk);                                                    // z A
  sc s2[1] = {ng}; chain_b<1>(acc, xa, s2, bk);                                                  // - gamma X_A   (:540 / :1233)
  ge yg[1] = {acc[1]}; chain_b<1>(yg, a.w, s2, bk);                                              // - gamma w     (:541 / :1234)
  uint8_t* tr = a.trs + (size_t)p * SMALL_TR_STRIDE;
  tr_put_prefix(tr, a.P, a.label);
  uint8_t* el = tr + a.P.prefix_len[a.label];
  if (issuance) { tr_put_bytes(el, c.v); el += 40; }
  tr_put_bytes(el, e.v); el += 40;
  uint32_t enc[8];
  tr_put_bytes(el, wa); el += 40;
  ristretto_encode(enc, xa); tr_put_bytes(el, enc); el += 40;
  ristretto_encode(enc, xg); tr_put_bytes(el, enc); el += 40;
  ristretto_encode(enc, acc[0]); tr_put_bytes(el, enc); el += 40;
  ristretto_encode(enc, yg[0]); tr_put_bytes(el, enc);
  a.flags[p] = flags;
}

// The same for SHORT launches, eight lanes per item: k_client_a is three variable-base chains, a Horner sum (refund) and four
// encodings in one lane (2.6 - 3.0 ms for one item).  Here
//     lane 0   z A, then Y_A = z A - gamma X_A (the second term from lane 1)      lane 1   X_A, then -gamma X_A
//     lane 2   Y_g = (z - gamma e) g - gamma w                                     lane 3   X_g = e g + w
//     lane 4   X_A once more, to encode it (lane 1 is busy with its chain)         lanes 5-7 idle
// Same group elements, same bytes.  The flags word is OR-ed (lanes 0 and 1 decode), so the caller clears it first.
__global__ void __launch_bounds__(256) k_client_a_wide(ClientArgs a) {
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x, p = gid >> 3, role = gid & 7;
  const bool live = p < a.n;
  const int L = a.P.L;
  const bool issuance = a.label == LABEL_RESPOND;
  ge pt = ge_identity();
  sc e = sc_zero(), c = sc_zero();
  uint32_t wa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (live && role <= 4) {
    const uint8_t* resp = a.resp + (size_t)p * (issuance ? 160 : 128);
    e = load_sc(resp + 32); const sc gamma = load_sc(resp + 64), z = load_sc(resp + 96);
    const sc ng = sc_neg(gamma);
    uint32_t* bk = a.pbk + ((size_t)p * PREP_BUCKET_SETS + (role < 3 ? role : 0)) * BUCKET_WORDS;
    if (issuance) c = load_sc(resp + 128);
    load8(wa, resp);
    if (role == 0) {
      ge A; if (!ristretto_decode(A, wa)) atomicOr(a.flags + p, FLAG_UNDECODABLE);
      ge acc[1] = {ge_identity()}; sc s1[1] = {z};
      chain_b<1>(acc, A, s1, bk);                                                              // z A
      pt = acc[0];
    } else if (role == 1 || role == 4) {
      ge xa;
      if (issuance) {
        uint32_t wk[8]; load8(wk, a.req + (size_t)p * 128);
        ge K; if (!ristretto_decode(K, wk) && role == 1) atomicOr(a.flags + p, FLAG_UNDECODABLE);
        xa = ge_add(fixed_base_acc(ge_basepoint(), a.P.tab[BASE_H1], c), K);                   // :536
      } else {
        ge kp = ge_identity();                                                                 // :1224-1230 by Horner
        for (int j = L - 1; j >= 0; j--) { kp = ge_double(kp); kp = ge_madd(kp, niels_load(a.coords + ((size_t)p * L + j) * NIELS_WORDS)); }
        xa = ge_add(kp, ge_basepoint());
      }
      if (role == 4) pt = xa;
      else { ge acc[1] = {ge_identity()}; sc s2[1] = {ng}; chain_b<1>(acc, xa, s2, bk); pt = acc[0]; }   // - gamma X_A   (:540 / :1233)
    } else if (role == 2) {
      ge yg[1] = {fixed_base_acc(ge_identity(), a.P.tab[BASE_G], sc_sub(z, sc_mul(gamma, e)))};
      sc s2[1] = {ng}; chain_b<1>(yg, a.w, s2, bk);                                            // - gamma w     (:541 / :1234)
      pt = yg[0];
    } else {
      pt = ge_add(fixed_base_acc(ge_identity(), a.P.tab[BASE_G], e), a.w);                     // X_g (:537 / :1232)
    }
  }
  // - gamma X_A travels from lane 1 of the group to lane 0 (every lane of the wavefront takes part in the shuffles)
  ge p1;
  const int src = (int)((threadIdx.x & 63u & ~7u) + 1u);
#pragma unroll
  for (int i = 0; i < FE_LIMBS; i++) {
    p1.X.v[i] = (uint32_t)__shfl((int)pt.X.v[i], src); p1.Y.v[i] = (uint32_t)__shfl((int)pt.Y.v[i], src);
    p1.Z.v[i] = (uint32_t)__shfl((int)pt.Z.v[i], src); p1.T.v[i] = (uint32_t)__shfl((int)pt.T.v[i], src);
  }
  if (!live || role > 4 || role == 1) return;
  if (role == 0) pt = ge_add(pt, p1);                                                          // Y_A
  uint8_t* tr = a.trs + (size_t)p * SMALL_TR_STRIDE;
  uint8_t* el = tr + a.P.prefix_len[a.label] + (issuance ? 40 : 0);
  // after [c] | e: A (slot 1) | X_A (2) | X_g (3) | Y_A (4) | Y_g (5)
  uint32_t enc[8];
  ristretto_encode(enc, pt);
  const int slot = role == 4 ? 2 : role == 3 ? 3 : role == 0 ? 4 : 5;
  tr_put_bytes(el + 40 * slot, enc);
  if (role == 0) {
    tr_put_prefix(tr, a.P, a.label);
    if (issuance) tr_put_bytes(tr + a.P.prefix_len[a.label], c.v);
    tr_put_bytes(el, e.v);
    tr_put_bytes(el + 40, wa);
  }
}

__global__ void __launch_bounds__(256) k_client_b(ClientArgs a) {
  uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  const bool issuance = a.label == LABEL_RESPOND;
  const uint8_t* resp = a.resp + (size_t)p * (issuance ? 160 : 128);
  uint32_t w[16];
  for (int i = 0; i < 16; i++) w[i] = a.xof[(size_t)p * 16 + i];
  sc gamma = load_sc(resp + 64);
  uint8_t stt = 0;
  if (a.flags[p] & FLAG_UNDECODABLE) stt = 255;
  else if (!sc_equal(sc_from_wide_words(w), gamma)) stt = issuance ? 2 : 4;   // InvalidIssuanceResponseProof / InvalidRefundProof
  a.status[p] = stt;
  uint8_t* out = a.out_token + (size_t)p * 160;
  if (stt) { for (int i = 0; i < 160; i += 32) zero8(out + i); return; }
  // CreditToken { a, e, k, r, c } (:554-560 / :1246-1252); pre record is r | k (| m)
  const uint8_t* pre = a.pre + (size_t)p * (issuance ? 64 : 96);
  uint32_t t[8];
  load8(t, resp); store8(out, t);
  store_sc(out + 32, load_sc(resp + 32));
  store_sc(out + 64, load_sc(pre + 32));
  store_sc(out + 96, load_sc(pre));
  store_sc(out + 128, issuance ? load_sc(resp + 128) : load_sc(pre + 64));
}

void launch_client_decode_com(const ClientArgs& a, hipStream_t s) {
  if (!a.n) return;
  size_t lanes = (size_t)a.n * a.P.L;
  hipLaunchKernelGGL(k_client_decode_com, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a);
}
constexpr uint32_t CLIENT_WIDE_MAX = 8192;
// (the wide form ORs into the flags word: client_batch clears it for both labels)
void launch_client_a(const ClientArgs& a, hipStream_t s) {
  if (!a.n) return;
  static const bool no_wide = getenv("ACT_NO_WIDE_CLIENT") != nullptr;     // A/B knob
  if (a.n <= CLIENT_WIDE_MAX && !no_wide) hipLaunchKernelGGL(k_client_a_wide, dim3((a.n * 8 + 255) / 256), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_client_a, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
}
void launch_client_b(const ClientArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_client_b, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }

}  // namespace act
