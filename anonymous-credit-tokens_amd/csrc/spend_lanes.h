// spend_lanes.h — the per-lane bodies of the spend-verification kernels (k_spend_verify.hip), as functions that also
// compile under g++: tests/hostcheck runs them lane by lane on the CPU to (a) compare the transcript bytes this very code
// writes with the test suite's CPU checker, (b) count its field multiplications / squarings exactly (the ALU roofline of bench.py), and
// (c) put it under AddressSanitizer / UBSan, none of which can be done on the GPU pool.  The product only ever runs them
// inside the __global__ wrappers of k_spend_verify.hip; there is no CPU compute path in libact_mi355x.so.
//
// PrivateKey::refund up to the challenge check, /root/reference/src/lib.rs:787-844:
//   spend_prep_*      3 lanes per proof     A', B_bar checks (:787), A1, A2 (:791-799), w00*h2 / w01*h2 (:806,808)
//   spend_bits_lane   lane = (proof, bit)   C'_j0 / 2, C'_j1 / 2 (:800-817): decode Com_j, two fixed-base sums, one shared
//                                           doubling chain over -Com_j (msm.h chain_bu); every scalar halved mod l
//   spend_enc_lane    lane = 32 half-points encodings of C'_j0, C'_j1 = 2 * (half-point) by batched double-and-compress:
//                                           one field inversion per 32 encodings instead of one inverse square root each
//   spend_tail_*      3 lanes per proof     K' by Horner over the decoded Com_j (:819-824), Com, C (:825-829), X_A (:848)
//   spend_finish_lane lane = proof          challenge = XOF mod l ?= gamma (:842-844) -> status
//
// Algebraic regrouping that leaves every encoded point (hence every transcript byte) unchanged:
//   A1 = (e_bar - x*gamma) A' + r2_bar B_bar                       (A_bar = x A' folded in)
//   A2 = r3_bar B_bar + c_bar h1 + r_bar h3 - gamma g - (gamma k) h2
//   C'_j1 = z_j1 h3 + gamma_j1 h1 - gamma_j1 Com_j                   (C_j1 = Com_j - h1 expanded)
//   C  = (-c_bar - gamma s) h1 + k_bar h2 + s_bar h3 - gamma K'
#pragma once
#include "kernels.h"

namespace act {

// the independent pieces of a per-proof kernel are separate FUNCTIONS on the device (own register allocation each), not one inlined live range
#if defined(__HIP_DEVICE_COMPILE__) && defined(ACT_PIECES_NOINLINE)
#define ACT_PIECE __device__ __attribute__((noinline))
#else
#define ACT_PIECE ACT_HD
#endif

constexpr int ENC_BATCH = 32;   // half-points per lane of k_spend_enc

// ---- k_spend_prep: three independent pieces per proof ---------------------------------------------------------------
//   piece A  decode A' (identity check :787), transcript k | A',  A1a = (e_bar - x gamma) A'          register-only chain (msm.h chain_ct)
//   piece B  decode B_bar, transcript B_bar,  A1b = r2_bar B_bar,  A2b = r3_bar B_bar  (one chain)    buckets sets 1, 2
//   piece C  transcript prefix, A2f = c_bar h1 + r_bar h3 - gamma g - (gamma k) h2, w00/2 h2, w01/2 h2, gamma's NAF digits
//   then     A1 = A1a + A1b, A2 = A2f + A2b, encoded.
// One lane runs the three pieces in sequence (spend_prep_lane).  Measured alternatives (same-box A/B): the
// pieces on three wavefronts of one block meeting in LDS (105 instead of 303 spilled VGPRs, 3 waves per SIMD-slot instead of 1)
// was 30 % SLOWER -- 65 536 proofs already put one wavefront on every SIMD, the kernel is bound by issue slots, and three roles
// of unequal length need two rounds of blocks; the pieces as non-inlined functions (-DACT_PIECES_NOINLINE: 2 spilled VGPRs in the
// kernel, each piece with its own register allocation) ran at the same speed as the inlined form: the spills are not the cost.
ACT_PIECE ge spend_prep_role_a(const SpendArgs& a, uint32_t p, uint32_t& flags) {
  const ProofLayout pl{a.P.L};
  const SpendTranscript st{a.P.L};
  const uint8_t* rec = a.proofs + (size_t)p * pl.bytes();
  uint8_t* el = a.tr + (size_t)p * a.tr_stride + 184;
  uint32_t wk[8], wa[8];
  load8(wk, rec + 32 * pl.k()); load8(wa, rec + 32 * pl.a_prime());
  ge A;
  if (!ristretto_decode(A, wa)) flags |= FLAG_UNDECODABLE;
  if (ristretto_is_identity(A)) flags |= FLAG_IDENTITY;                       // src/lib.rs:787-789
  sc k = sc_from_words(wk);
  tr_put_aligned(el + 40 * st.el_k(), k.v);                                   // Scalar::as_bytes of the reduced k
  tr_put_aligned(el + 40 * st.el_a_prime(), wa);                              // compress(decompress(x)) == x for canonical x
  sc gamma = load_sc(rec + 32 * pl.gamma()), e_bar = load_sc(rec + 32 * pl.e_bar());
  ge acc[1] = {ge_identity()};
  sc sa[1] = {sc_sub(e_bar, sc_mul(a.K.x, gamma))};
  chain_ct<1>(acc, A, sa);                                 // the scalar depends on the issuer's x: no digit of it ever selects an address (msm.h)
  return acc[0];
}
ACT_PIECE void spend_prep_role_b(const SpendArgs& a, uint32_t p, uint32_t& flags, ge& a1b, ge& a2b) {
  const ProofLayout pl{a.P.L};
  const SpendTranscript st{a.P.L};
  const uint8_t* rec = a.proofs + (size_t)p * pl.bytes();
  uint8_t* el = a.tr + (size_t)p * a.tr_stride + 184;
  uint32_t wb[8];
  load8(wb, rec + 32 * pl.b_bar());
  ge B;
  if (!ristretto_decode(B, wb)) flags |= FLAG_UNDECODABLE;
  tr_put_aligned(el + 40 * st.el_b_bar(), wb);
  ge acc[2] = {ge_identity(), ge_identity()};
  sc sb[2] = {load_sc(rec + 32 * pl.r2_bar()), load_sc(rec + 32 * pl.r3_bar())};
  chain_b<2>(acc, B, sb, a.pbk + ((size_t)p * PREP_BUCKET_SETS + 1) * BUCKET_WORDS);
  a1b = acc[0]; a2b = acc[1];
}
ACT_PIECE ge spend_prep_role_c_a2(const SpendArgs& a, uint32_t p) {
  const ProofLayout pl{a.P.L};
  const uint8_t* rec = a.proofs + (size_t)p * pl.bytes();
  tr_put_prefix(a.tr + (size_t)p * a.tr_stride, a.P, LABEL_SPEND);
  sc k = load_sc(rec + 32 * pl.k()), gamma = load_sc(rec + 32 * pl.gamma());
  sc c_bar = load_sc(rec + 32 * pl.c_bar()), r_bar = load_sc(rec + 32 * pl.r_bar());
  sc ngamma = sc_neg(gamma);
  ge f = fixed_base_acc(ge_identity(), a.P.tab[BASE_H1], c_bar);
  f = fixed_base_acc(f, a.P.tab[BASE_H3], r_bar);
  f = fixed_base_acc(f, a.P.tab[BASE_G], ngamma);
  return fixed_base_acc(f, a.P.tab[BASE_H2], sc_mul(ngamma, k));
}
ACT_PIECE void spend_prep_role_c_bits(const SpendArgs& a, uint32_t p) {
  const ProofLayout pl{a.P.L};
  const uint8_t* rec = a.proofs + (size_t)p * pl.bytes();
  sc gamma = load_sc(rec + 32 * pl.gamma());
  sc w00 = load_sc(rec + 32 * pl.w00()), w01 = load_sc(rec + 32 * pl.w01());
  ge d0 = fixed_base_acc(ge_identity(), a.P.tab[BASE_H2], sc_half(w00));     // the bits kernel works on C'/2 (k_spend_enc)
  ge d1 = fixed_base_acc(ge_identity(), a.P.tab[BASE_H2], sc_half(w01));
  ge_store(a.d01 + (size_t)p * 2 * GE_WORDS, d0);
  ge_store(a.d01 + (size_t)p * 2 * GE_WORDS + GE_WORDS, d1);
  naf3_recode(a.naf + (size_t)p * NAF_WORDS, sc_half(gamma));                 // the proof-wide challenge's digit string for k_spend_bits (msm.h)
}
ACT_PIECE ge spend_prep_role_c(const SpendArgs& a, uint32_t p) {
  ge f = spend_prep_role_c_a2(a, p);
  spend_prep_role_c_bits(a, p);
  return f;
}
ACT_PIECE void spend_prep_put(const SpendArgs& a, uint32_t p, int element, const ge& pt) {
  uint32_t enc[8];
  ristretto_encode(enc, pt);
  tr_put_aligned(a.tr + (size_t)p * a.tr_stride + 184 + 40 * element, enc);
}
ACT_HD void spend_prep_lane(const SpendArgs& a, uint32_t p) {
  const SpendTranscript st{a.P.L};
  uint32_t flags = 0;
  ge a1a = spend_prep_role_a(a, p, flags);
  ge a1b, a2b;
  spend_prep_role_b(a, p, flags, a1b, a2b);
  ge a2f = spend_prep_role_c(a, p);
  spend_prep_put(a, p, st.el_a1(), ge_add(a1a, a1b));
  spend_prep_put(a, p, st.el_a2(), ge_add(a2f, a2b));
  a.flags[p] = flags;     // bits kernel ORs its decode failures in afterwards (same stream)
}

// ---- the small-batch schedule (small_impl.inc): the same pieces as kernels of their own ---------------------------------
// The crate's call shape is ONE proof per call (/root/reference/src/lib.rs:781-786, benches/benchmark.rs:166-212).  A lane of
// k_spend_prep is ~8 200 dependent field operations, one of k_spend_tail ~5 300, and below ~2^15 proofs a launch of either leaves
// most of the chip idle: the serial depth of prep -> bits -> enc -> tail IS the call's latency (5.7 ms for one proof in round 3).
// Nothing but the transcript hash needs all of them, so small calls run them side by side on four streams:
//     prep role C -> k_spend_bits -> k_spend_enc   |   prep role A   |   prep role B, then A1 / A2   |   Com_j decode -> k_spend_tail
// Same lane bodies, same bytes; the roles leave their partial sums in `part` (four points per proof) and every kernel ORs its
// flags in (the flag words are cleared up front).  Measured: 1 proof 5.7 -> ~2 ms, 4 096 proofs 281 k -> >400 k/s.
// For throughput at 2^16 proofs per launch the one-lane-per-proof kernels stay as they are (section 8: the three-role form was slower).
// `part` (kernels.h PART_POINTS = 4 points per proof): a1a = (e_bar - x gamma) A', a1b = r2_bar B_bar, a2b = r3_bar B_bar, a2f (fixed-base part of A2)
ACT_HD void spend_prep_a_lane(const SpendArgs& a, uint32_t p) {
  uint32_t flags = 0;
  ge_store(a.part + ((size_t)p * PART_POINTS + 0) * GE_WORDS, spend_prep_role_a(a, p, flags));
  if (flags) ACT_FLAG_OR(a.flags + p, flags);
}
ACT_HD void spend_prep_b_lane(const SpendArgs& a, uint32_t p) {
  uint32_t flags = 0;
  ge a1b, a2b;
  spend_prep_role_b(a, p, flags, a1b, a2b);
  ge_store(a.part + ((size_t)p * PART_POINTS + 1) * GE_WORDS, a1b);
  ge_store(a.part + ((size_t)p * PART_POINTS + 2) * GE_WORDS, a2b);
  if (flags) ACT_FLAG_OR(a.flags + p, flags);
}
ACT_HD void spend_prep_c_lane(const SpendArgs& a, uint32_t p) {
  ge_store(a.part + ((size_t)p * PART_POINTS + 3) * GE_WORDS, spend_prep_role_c(a, p));
}
// role C in two kernels: c1 = the part k_spend_bits waits for (gamma's digit string, w00 h2, w01 h2: two table products), c2 = the
// rest (the transcript prefix and the four table products of A2)
ACT_HD void spend_prep_c1_lane(const SpendArgs& a, uint32_t p) { spend_prep_role_c_bits(a, p); }
ACT_HD void spend_prep_c2_lane(const SpendArgs& a, uint32_t p) {
  ge_store(a.part + ((size_t)p * PART_POINTS + 3) * GE_WORDS, spend_prep_role_c_a2(a, p));
}
ACT_HD void spend_prep_join_lane(const SpendArgs& a, uint32_t p) {
  const SpendTranscript st{a.P.L};
  const uint32_t* q = a.part + (size_t)p * PART_POINTS * GE_WORDS;
  spend_prep_put(a, p, st.el_a1(), ge_add(ge_load(q), ge_load(q + GE_WORDS)));
  spend_prep_put(a, p, st.el_a2(), ge_add(ge_load(q + 3 * GE_WORDS), ge_load(q + 2 * GE_WORDS)));
}
// the decoded Com_j for k_spend_tail, which then does not have to wait for k_spend_bits (which decodes them again for itself and
// stores the same bytes)
ACT_HD void spend_coords_lane(const SpendArgs& a, uint32_t gid) {
  const int L = a.P.L;
  uint32_t p = gid / (uint32_t)L, j = gid % (uint32_t)L;
  if (p >= a.n) return;
  const ProofLayout pl{L};
  uint32_t wc[8];
  load8(wc, a.proofs + (size_t)p * pl.bytes() + 32 * pl.com(j));
  ge C;
  if (!ristretto_decode(C, wc)) ACT_FLAG_OR(a.flags + p, FLAG_UNDECODABLE);
  niels_store(a.coords + ((size_t)p * L + j) * NIELS_WORDS, niels_from_affine(C));
}

// `lds_wave` (device): 2 * GE_LDS_WORDS_PER_WAVE words of LDS owned by the calling wavefront (msm.h chain_bu); unused on the host
template <bool UNIFORM>
ACT_HD void spend_bits_lane(const SpendArgs& a, uint32_t gid, uint32_t* lds_wave) {
  const int L = a.P.L;
  uint32_t p = gid / (uint32_t)L, j = gid % (uint32_t)L;
  if (p >= a.n) return;
  const ProofLayout pl{L};
  const SpendTranscript st{L};
  const uint8_t* rec = a.proofs + (size_t)p * pl.bytes();
  uint8_t* el = a.tr + (size_t)p * a.tr_stride + 184;

  uint32_t wc[8];
  load8(wc, rec + 32 * pl.com(j));
  ge C;
  bool ok = ristretto_decode(C, wc);
  if (!ok) ACT_FLAG_OR(a.flags + p, FLAG_UNDECODABLE);
  tr_put_aligned(el + 40 * st.el_com(j), wc);
  niels_store(a.coords + ((size_t)p * L + j) * NIELS_WORDS, niels_from_affine(C));

  // Everything below is computed at half scale: Q_j0 = C'_j0 / 2, Q_j1 = C'_j1 / 2 (all scalars halved mod l), because
  // the encoding of 2Q needs only an inversion, which k_spend_enc batches (ge25519.h dc_*).
  sc gamma = sc_half(load_sc(rec + 32 * pl.gamma()));
  sc g0 = sc_half(load_sc(rec + 32 * pl.gamma0(j)));
  sc g1 = sc_sub(gamma, g0);                                                  // src/lib.rs:801, 811
  sc z0 = sc_half(load_sc(rec + 32 * pl.z(j, 0))), z1 = sc_half(load_sc(rec + 32 * pl.z(j, 1)));

  // C'_j0 = z_j0 h3 + D,  C'_j1 = z_j1 h3 + gamma_j1 h1 + G - D  with D = gamma_j0 N, G = gamma N, N = -Com_j
  // (gamma_j1 = gamma - gamma_j0).  G's scalar is the proof-wide gamma: uniform width-3 NAF digits per wavefront; D's digits go through per-lane buckets (msm.h chain_bu).
  ge acc_u = fixed_base_acc(ge_identity(), a.P.tab[BASE_H3], z1);
  acc_u = fixed_base_acc(acc_u, a.P.tab[BASE_H1], g1);
  if (j == 0) acc_u = ge_add(acc_u, ge_load(a.d01 + (size_t)p * 2 * GE_WORDS + GE_WORDS));      // + w01 h2 (:808)
  ge acc_l = ge_identity();
#if defined(ACT_ABLATE_BUCKET_FOOTPRINT)      // measurement build only: all lanes' buckets folded onto 8 MiB so that the bucket
  uint32_t* bk = a.buckets + (size_t)(gid % ACT_ABLATE_BUCKET_FOOTPRINT) * BUCKET_WORDS;      // traffic hits in L2; results are garbage
#else
  uint32_t* bk = a.buckets + (size_t)gid * BUCKET_WORDS;
#endif
  uint32_t* dg = a.dig + (size_t)gid * 8;
  { uint32_t t[8]; radix16_bias(t, g0); for (int i = 0; i < 8; i++) dg[i] = t[i]; }
  // Com_j in E[4] <=> its encoding is all zero (or it did not decode: the proof is rejected anyway): D = G = identity
  const bool n_small = !ok || (wc[0] | wc[1] | wc[2] | wc[3] | wc[4] | wc[5] | wc[6] | wc[7]) == 0u;
  chain_bu_pre<UNIFORM>(acc_l, acc_u, ge_neg(C), dg, a.naf + (size_t)p * NAF_WORDS, bk, lds_wave, n_small);
  ge f0 = fixed_base_acc(ge_identity(), a.P.tab[BASE_H3], z0);
  if (j == 0) f0 = ge_add(f0, ge_load(a.d01 + (size_t)p * 2 * GE_WORDS));                        // + w00 h2 (:806)
  ge_cached dl = ge_to_cached(acc_l);
  // the lane's bucket area is free again: its first two slots carry the half-points to k_spend_enc
  bucket_store(bk, ge_add_cached(f0, dl));
  bucket_store(bk + GE_WORDS, ge_add_cached(acc_u, ge_cached_cneg(dl, true)));
}

// point q = 2 * (p * L + j) + b lives in slot b of lane (p, j)'s bucket area; this lane encodes q0 .. q0 + ENC_BATCH - 1
// (E points per lane: ENC_BATCH in the pipelined schedule; the small-batch schedule takes ENC_BATCH_SMALL -- four times the lanes, each
// a quarter as deep, on a chip it does not fill anyway)
constexpr int ENC_BATCH_SMALL = 8;
template <int E>
ACT_HD void spend_enc_lane_e(const SpendArgs& a, uint64_t q0) {
  const uint32_t L = (uint32_t)a.P.L;
  const uint64_t total = (uint64_t)a.n * L * 2u;
  if (q0 >= total) return;
  const int count = (int)(total - q0 < (uint64_t)E ? total - q0 : (uint64_t)E);
  const SpendTranscript st{a.P.L};
  dc_encode_batch<E>(
      count,
      [&](int i) { uint64_t q = q0 + (uint64_t)i; return a.buckets + (size_t)(q >> 1) * BUCKET_WORDS + (q & 1u) * GE_WORDS; },
      [&](int i, const uint32_t* enc) {
        uint64_t q = q0 + (uint64_t)i, lane = q >> 1;
        uint32_t p = (uint32_t)(lane / L), j = (uint32_t)(lane % L);
        tr_put_aligned(a.tr + (size_t)p * a.tr_stride + 184 + 40 * st.el_cprime(j, (int)(q & 1u)), enc);
      });
}
ACT_HD void spend_enc_lane(const SpendArgs& a, uint64_t q0) { spend_enc_lane_e<ENC_BATCH>(a, q0); }

// ---- k_spend_tail ---------------------------------------------------------------------------------------------------------
//   K' = sum_j 2^j Com_j by Horner over the decoded points (src/lib.rs:819-824: the reference does 128 separate mults)
//   F  = (-c_bar - gamma s) h1 + k_bar h2 + s_bar h3;  C = F - gamma K' (:825-829), encoded;  X_A = g + K' (:848), enc(K') if asked for.
// (spend_tail_horner takes a sub-range and a shift so that K' can be cut into independent runs: the three-wavefront form of this
// kernel -- lo / hi halves of the Horner run and F on three wavefronts of a block, DESIGN.md section 8 -- measured slower.)
ACT_HD ge spend_tail_horner(const SpendArgs& a, uint32_t p, int j_lo, int j_hi, int shift) {
  const int L = a.P.L;
  ge kp = ge_identity();
  for (int j = j_hi - 1; j >= j_lo; j--) {
    kp = ge_double(kp);
    kp = ge_madd(kp, niels_load(a.coords + ((size_t)p * L + j) * NIELS_WORDS));
  }
  for (int i = 0; i < shift; i++) kp = ge_double(kp);
  return kp;
}
ACT_HD ge spend_tail_fixed(const SpendArgs& a, uint32_t p) {
  const ProofLayout pl{a.P.L};
  const uint8_t* rec = a.proofs + (size_t)p * pl.bytes();
  sc gamma = load_sc(rec + 32 * pl.gamma());
  sc s = load_sc(rec + 32 * pl.s()), c_bar = load_sc(rec + 32 * pl.c_bar());
  sc k_bar = load_sc(rec + 32 * pl.k_bar()), s_bar = load_sc(rec + 32 * pl.s_bar());
  ge f = fixed_base_acc(ge_identity(), a.P.tab[BASE_H1], sc_sub(sc_mul(sc_neg(gamma), s), c_bar));   // -c_bar - gamma s
  f = fixed_base_acc(f, a.P.tab[BASE_H2], k_bar);
  return fixed_base_acc(f, a.P.tab[BASE_H3], s_bar);
}
ACT_HD void spend_tail_c(const SpendArgs& a, uint32_t p, const ge& kp, const ge& f) {
  const ProofLayout pl{a.P.L};
  const SpendTranscript st{a.P.L};
  ge acc[1] = {f};
  sc sk_[1] = {sc_neg(load_sc(a.proofs + (size_t)p * pl.bytes() + 32 * pl.gamma()))};
  chain_b<1>(acc, kp, sk_, a.pbk + (size_t)p * PREP_BUCKET_SETS * BUCKET_WORDS);      // large batches: pbk = buckets, whose half-points k_spend_enc has consumed
  uint32_t enc[8];
  ristretto_encode(enc, acc[0]); tr_put_aligned(a.tr + (size_t)p * a.tr_stride + 184 + 40 * st.el_c(), enc);
}
ACT_HD void spend_tail_xa(const SpendArgs& a, uint32_t p, const ge& kp) {
  ge_store(a.xa + (size_t)p * GE_WORDS, ge_add(kp, ge_basepoint()));                 // X_A = g + K' (src/lib.rs:848)
  if (a.kprime_enc) { uint32_t enc[8]; ristretto_encode(enc, kp); store8(a.kprime_enc + (size_t)p * 32, enc); }
}
ACT_HD void spend_tail_lane(const SpendArgs& a, uint32_t p) {
  ge kp = spend_tail_horner(a, p, 0, a.P.L, 0);
  spend_tail_c(a, p, kp, spend_tail_fixed(a, p));
  spend_tail_xa(a, p, kp);
}
// The same in two kernels, for the call that signs beside its verification (small_impl.inc spend_small_locked): X_A leaves after the
// Horner run -- 0.5 ms into a 1.25 ms lane -- and the signature's chain starts there; C follows from K' = X_A - g.
ACT_HD void spend_tail_k_lane(const SpendArgs& a, uint32_t p) { spend_tail_xa(a, p, spend_tail_horner(a, p, 0, a.P.L, 0)); }
ACT_HD void spend_tail_c_lane(const SpendArgs& a, uint32_t p) {
  const ge kp = ge_add(ge_load(a.xa + (size_t)p * GE_WORDS), ge_neg(ge_basepoint()));
  spend_tail_c(a, p, kp, spend_tail_fixed(a, p));
}

ACT_HD void spend_finish_lane(const SpendArgs& a, uint32_t p) {
  const ProofLayout pl{a.P.L};
  sc gamma = load_sc(a.proofs + (size_t)p * pl.bytes() + 32 * pl.gamma());
  uint32_t w[16];
  for (int i = 0; i < 16; i++) w[i] = a.xof[(size_t)p * 16 + i];
  sc chal = sc_from_wide_words(w);                                            // src/transcript.rs:149-154
  uint32_t f = a.flags[p];
  uint8_t stt = 0;
  if (f & FLAG_UNDECODABLE) stt = 255;
  else if (f & FLAG_IDENTITY) stt = 6;                                        // Error::IdentityPointError
  else if (!sc_equal(chal, gamma)) stt = 7;                                   // Error::InvalidClientSpendProof
  a.status[p] = stt;
  if (a.kprime_enc && stt != 0) zero8(a.kprime_enc + (size_t)p * 32);          // the output record of a failed lane is all zero
}

}  // namespace act
