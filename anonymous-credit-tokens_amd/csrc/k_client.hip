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

// The same for SHORT launches, four BLOCKS of one wavefront per 64 items (blockIdx.y = role, lane = item): k_client_a is three
// variable-base chains, a Horner sum (refund) and four encodings in one lane (2.6 - 3.0 ms for one item).  Here
//     role 0   z A                          role 1   X_A, then -gamma X_A
//     role 2   Y_g = (z - gamma e) g - gamma w                                        role 3   X_g = e g + w, and X_A once more, to encode it
// and the block of a group that ARRIVES LAST (a counter per group) sums Y_A = z A - gamma X_A from the two partial points the
// roles left in their bucket areas and encodes it; for tiny calls (`fused`) it goes on to hash the transcript (one BLAKE3 chunk,
// the routine of k_hash_xof) and to do what k_client_b does -- the whole method in one launch.  Same group elements, same bytes.
// (Round 4 had the roles on LANES of one wavefront, which executes divergent lanes one after the other: the three chains took
// 3 x 0.7 ms.  Wavefronts of ONE block can land on one SIMD and then run at a fraction of their speed each -- 0.9 or 2.0 ms from
// call to call, profiles/r05_tiny_ab.txt; separate workgroups go to separate CUs.)  The flags word is OR-ed, so the caller clears it.
__device__ __forceinline__ bool client_group_last_arrival(uint32_t* counter, uint32_t roles) {
  __shared__ uint32_t ticket;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicAdd(counter, 1u);
  __syncthreads();
  const bool last = ticket == roles - 1u;
  if (last) { __threadfence(); if (threadIdx.x == 0) *counter = 0u; }
  return last;
}
__global__ void __launch_bounds__(64) k_client_a_wide(ClientArgs a) {
  const uint32_t role = blockIdx.y, lane = threadIdx.x, p = blockIdx.x * 64 + lane;
  const bool live = p < a.n;
  const int L = a.P.L;
  const bool issuance = a.label == LABEL_RESPOND;
  const uint8_t* resp = a.resp + (size_t)(live ? p : 0) * (issuance ? 160 : 128);
  uint8_t* tr = a.trs + (size_t)(live ? p : 0) * SMALL_TR_STRIDE;
  uint8_t* el = tr + a.P.prefix_len[a.label] + (issuance ? 40 : 0);
  // after [c] | e: A (slot 1) | X_A (2) | X_g (3) | Y_A (4) | Y_g (5)
  uint32_t enc[8];
  uint32_t* part = a.pbk + (size_t)(live ? p : 0) * PREP_BUCKET_SETS * BUCKET_WORDS;      // sets 0 and 1: the roles' bucket areas, then their results
  if (live) {
    const sc e = load_sc(resp + 32), gamma = load_sc(resp + 64), z = load_sc(resp + 96);
    const sc ng = sc_neg(gamma);
    uint32_t* bk = part + (size_t)(role < 3 ? role : 0) * BUCKET_WORDS;
    sc c = sc_zero();
    if (issuance) c = load_sc(resp + 128);
    uint32_t wa[8]; load8(wa, resp);
    if (role == 0) {
      ge A; if (!ristretto_decode(A, wa)) atomicOr(a.flags + p, FLAG_UNDECODABLE);
      ge acc[1] = {ge_identity()}; sc s1[1] = {z};
      chain_b<1>(acc, A, s1, bk);                                                              // z A
      ge_store(bk, acc[0]);
      // (bytes, not tr_put_prefix's word stores: the word behind the prefix belongs to an element another block may be writing)
      const uint8_t* pre = reinterpret_cast<const uint8_t*>(a.P.prefix[a.label]);
      for (uint32_t i = 0; i < a.P.prefix_len[a.label]; i++) tr[i] = pre[i];
      if (issuance) tr_put_bytes(tr + a.P.prefix_len[a.label], c.v);
      tr_put_bytes(el, e.v);
      tr_put_bytes(el + 40, wa);
    } else if (role == 1 || role == 3) {
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
      if (role == 3) {
        ristretto_encode(enc, xa); tr_put_bytes(el + 40 * 2, enc);
        ristretto_encode(enc, ge_add(fixed_base_acc(ge_identity(), a.P.tab[BASE_G], e), a.w)); tr_put_bytes(el + 40 * 3, enc);   // X_g (:537 / :1232)
      } else { ge acc[1] = {ge_identity()}; sc s2[1] = {ng}; chain_b<1>(acc, xa, s2, bk); ge_store(bk, acc[0]); }   // - gamma X_A   (:540 / :1233)
    } else {
      ge yg[1] = {fixed_base_acc(ge_identity(), a.P.tab[BASE_G], sc_sub(z, sc_mul(gamma, e)))};
      sc s2[1] = {ng}; chain_b<1>(yg, a.w, s2, bk);                                            // - gamma w     (:541 / :1234)
      ristretto_encode(enc, yg[0]); tr_put_bytes(el + 40 * 5, enc);
    }
  }
  if (!client_group_last_arrival(a.group_counter + blockIdx.x, 4u)) return;
  if (!live) return;
  ristretto_encode(enc, ge_add(ge_load(part), ge_load(part + BUCKET_WORDS)));                  // Y_A = z A - gamma X_A
  tr_put_bytes(el + 40 * 4, enc);
  if (!a.fused) return;
  // ---- tiny calls: the transcript's hash and k_client_b's part, here --------------------------------------------------------------
  uint32_t w[16];
  b3_hash_xof64(w, reinterpret_cast<const uint32_t*>(tr), a.P.prefix_len[a.label] + 40u * (issuance ? 7u : 6u));
  const sc gamma = load_sc(resp + 64);
  const uint32_t flags = __atomic_load_n(a.flags + p, __ATOMIC_RELAXED);
  uint8_t stt = 0;
  if (flags & FLAG_UNDECODABLE) stt = 255;
  else if (!sc_equal(sc_from_wide_words(w), gamma)) stt = issuance ? 2 : 4;   // InvalidIssuanceResponseProof / InvalidRefundProof
  a.status[p] = stt;
  uint8_t* out = a.out_token + (size_t)p * 160;
  if (stt) { for (int i = 0; i < 160; i += 32) zero8(out + i); return; }
  const uint8_t* pre = a.pre + (size_t)p * (issuance ? 64 : 96);              // CreditToken { a, e, k, r, c } (:554-560 / :1246-1252); pre = r | k (| m)
  uint32_t t[8];
  load8(t, resp); store8(out, t);
  store_sc(out + 32, load_sc(resp + 32));
  store_sc(out + 64, load_sc(pre + 32));
  store_sc(out + 96, load_sc(pre));
  store_sc(out + 128, issuance ? load_sc(resp + 128) : load_sc(pre + 64));
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
  const bool no_wide = tune(T_NO_WIDE_CLIENT) != 0;     // A/B knob (act_tuning_set)
  if (a.n <= CLIENT_WIDE_MAX && !no_wide) hipLaunchKernelGGL(k_client_a_wide, dim3((a.n + 63) / 64, 4), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(k_client_a, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
}
void launch_client_b(const ClientArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_client_b, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }

}  // namespace act
