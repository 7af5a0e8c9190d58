// prove_lanes.h — the per-lane bodies of the prover kernels (k_prove.hip), as functions that also compile under g++:
// tests/hostcheck runs them lane by lane on the CPU to compare the proof bytes this very code writes with the test suite's CPU
// checker, to count its field operations and table reads exactly (bench.py roofline_prover), and to put it under ASan / UBSan.
// The product only ever runs them inside the __global__ wrappers of k_prove.hip; there is no CPU compute path in libact_mi355x.so.
//
// CreditToken::prove_spend (/root/reference/src/lib.rs:972-1152) on gfx950.
//
// The prover knows an opening of every point it commits to, so — unlike the reference, which calls
// `point * scalar` 261 times — only the two mults on the token's signature point A are
// variable-base here; everything else is regrouped over the fixed bases g, h1, h2, h3 (same group
// elements, hence the same encodings, transcript bytes and challenge):
//   B_bar = r1 B = r1 g + (r1 c) h1 + (r1 k) h2 + (r1 r) h3                              (:986-991)
//   A1 = e' A' + r2' B_bar,  A2 = r3' B_bar + c' h1 + r' h3                              (:993-994)
//   simulated branch of bit j:  z_j h3 - gamma_j C_jb = (z_j - gamma_j s_j) h3 -+ gamma_j h1
//                               (+ (w0 - gamma_0 k*) h2 for j = 0)                        (:1025-1050)
// Both OR branches are always computed and chosen with selects (no branch on a secret bit), as the
// reference does with subtle::conditional_select.
//
//   k_prove_head  lane = proof         A', B_bar, A1, A2, r3 = 1/r1, the three h2 terms of bit 0
//   k_prove_bits  lane = (proof, bit)  Com_j / 2, C'_j0 / 2, C'_j1 / 2 (every scalar halved mod l)
//   k_prove_enc   lane = 32 half-points their encodings by batched double-and-compress (ge25519.h dc_*, msm.h): one field
//                                      inversion per 32 encodings instead of an inverse square root each
//   k_prove_tail  lane = proof         r*, C
//   (transcript hash)
//   k_prove_resp  lane = (proof, bit)  gamma_j0, z_j0, z_j1; lane bit 0 also writes every proof-level response
#pragma once
#include "kernels.h"

namespace act {

// `Fb`: how a lane multiplies a fixed base by a SECRET scalar (kernels.h SecretFb / SecretFbLds; the host build passes a plain
// fixed_base_acc wrapper).  stage(base) may hold block barriers (ct build): every thread of a block calls it, live lane or not.

struct RngView {   // draw index -> 64-byte slot
  const uint8_t* base; int L;
  ACT_HD sc draw(int i) const { return load_wide(base + (size_t)i * 64); }
  ACT_HD sc r1() const { return draw(0); }
  ACT_HD sc r2() const { return draw(1); }
  ACT_HD sc c_prime() const { return draw(2); }
  ACT_HD sc r_prime() const { return draw(3); }
  ACT_HD sc e_prime() const { return draw(4); }
  ACT_HD sc r2_prime() const { return draw(5); }
  ACT_HD sc r3_prime() const { return draw(6); }
  ACT_HD sc k_star() const { return draw(7); }
  ACT_HD sc s_i(int j) const { return draw(8 + j); }
  ACT_HD sc k0_prime() const { return draw(8 + L); }
  ACT_HD sc s_i_prime(int j) const { return draw(9 + L + j); }
  ACT_HD sc gamma_i(int j) const { return draw(9 + 2 * L + j); }
  ACT_HD sc w0() const { return draw(9 + 3 * L); }
  ACT_HD sc z(int j) const { return draw(10 + 3 * L + j); }
  ACT_HD sc k_prime() const { return draw(10 + 4 * L); }
  ACT_HD sc s_prime() const { return draw(11 + 4 * L); }
};
ACT_HD size_t rng_bytes(int L) { return 64u * (4u * (size_t)L + 12u); }

template <class Fb>
ACT_HD void prove_head_lane(const ProveArgs& a, uint32_t p, Fb& fb) {
  const bool live = p < a.n;
  const int L = a.P.L;
  const ProofLayout pl{L}; const SpendTranscript st{L};
  RngView rv{a.rng + (size_t)(live ? p : 0) * rng_bytes(L), L};
  ge acc[2] = {ge_identity(), ge_identity()};
  sc k = sc_zero(), s = sc_zero(), r1 = sc_zero(), r1c = sc_zero(), r1k = sc_zero(), r1r = sc_zero();
  sc c_prime = sc_zero(), r_prime = sc_zero(), r2_prime = sc_zero(), r3_prime = sc_zero(), k_star = sc_zero();
  if (live) {
    const uint8_t* tok = a.tok + (size_t)p * 160;
    uint32_t wa[8]; load8(wa, tok);
    ge A; a.flags[p] = ristretto_decode(A, wa) ? 0u : FLAG_UNDECODABLE;
    k = load_sc(tok + 64); sc r = load_sc(tok + 96), c = load_sc(tok + 128); s = load_sc(a.s + (size_t)p * 32);
    r1 = rv.r1(); sc r2 = rv.r2(); c_prime = rv.c_prime(); r_prime = rv.r_prime();
    sc e_prime = rv.e_prime(); r2_prime = rv.r2_prime(); r3_prime = rv.r3_prime(); k_star = rv.k_star();
    sc r1r2 = sc_mul(r1, r2);
    // A' = (r1 r2) A and the A-part of A1 = e' A' share A's doubling chain
    sc sa[2] = {r1r2, sc_mul(e_prime, r1r2)};
    chain_s<2>(acc, A, sa, a.half + (size_t)p * 2 * BUCKET_WORDS);       // the half-point area is not in use yet
    r1c = sc_mul(r1, c); r1k = sc_mul(r1, k); r1r = sc_mul(r1, r);
  }
  // B_bar = r1 g + (r1 c) h1 + (r1 k) h2 + (r1 r) h3;  A1 = e' A' + r2' B_bar;  A2 = r3' B_bar + c' h1 + r' h3 -- the twelve products
  // (and the three h2 terms of bit 0, src/lib.rs:1001, 1025-1035, at half scale like everything k_prove_bits computes) grouped
  // by base: the ct build stages one base's table in LDS at a time
  // Every fb.mul below runs in EVERY lane, live or not (a lane past the batch multiplies zeros / lane 0's rng: its results are
  // dropped): the ct build's matrix-core look-up (msm.h fixed_base_acc_mf) takes its table operand from all 64 lanes of a wavefront.
  ge bbar = ge_identity(), a1 = acc[1], a2 = ge_identity();
  fb.stage(BASE_G);
  bbar = fb.mul(bbar, BASE_G, r1); a1 = fb.mul(a1, BASE_G, sc_mul(r2_prime, r1)); a2 = fb.mul(a2, BASE_G, sc_mul(r3_prime, r1));
  fb.stage(BASE_H1);
  bbar = fb.mul(bbar, BASE_H1, r1c); a1 = fb.mul(a1, BASE_H1, sc_mul(r2_prime, r1c)); a2 = fb.mul(a2, BASE_H1, sc_muladd(r3_prime, r1c, c_prime));
  fb.stage(BASE_H2);
  bbar = fb.mul(bbar, BASE_H2, r1k); a1 = fb.mul(a1, BASE_H2, sc_mul(r2_prime, r1k)); a2 = fb.mul(a2, BASE_H2, sc_mul(r3_prime, r1k));
  {
    const ge d0 = fb.mul(ge_identity(), BASE_H2, sc_half(k_star));
    const ge d1 = fb.mul(ge_identity(), BASE_H2, sc_half(rv.k0_prime()));
    const ge d2 = fb.mul(ge_identity(), BASE_H2, sc_half(sc_sub(rv.w0(), sc_mul(rv.gamma_i(0), k_star))));
    if (live) {
      uint32_t* d3 = a.d3 + (size_t)p * 3 * GE_WORDS;
      ge_store(d3, d0); ge_store(d3 + GE_WORDS, d1); ge_store(d3 + 2 * GE_WORDS, d2);
    }
  }
  fb.stage(BASE_H3);
  bbar = fb.mul(bbar, BASE_H3, r1r); a1 = fb.mul(a1, BASE_H3, sc_mul(r2_prime, r1r)); a2 = fb.mul(a2, BASE_H3, sc_muladd(r3_prime, r1r, r_prime));
  if (!live) return;

  uint8_t* rec = a.proof + (size_t)p * pl.bytes();
  uint8_t* el = a.tr + (size_t)p * a.tr_stride + 184;
  tr_put_prefix(a.tr + (size_t)p * a.tr_stride, a.P, LABEL_SPEND);
  uint32_t enc[8];
  tr_put_aligned(el + 40 * st.el_k(), k.v);
  store_sc(rec + 32 * pl.k(), k); store_sc(rec + 32 * pl.s(), s);
  ristretto_encode(enc, acc[0]); tr_put_aligned(el + 40 * st.el_a_prime(), enc); store8(rec + 32 * pl.a_prime(), enc);
  ristretto_encode(enc, bbar); tr_put_aligned(el + 40 * st.el_b_bar(), enc); store8(rec + 32 * pl.b_bar(), enc);
  ristretto_encode(enc, a1); tr_put_aligned(el + 40 * st.el_a1(), enc);
  ristretto_encode(enc, a2); tr_put_aligned(el + 40 * st.el_a2(), enc);
  sc r3 = sc_invert(r1);                                                      // :992
  uint32_t* stt = a.state + (size_t)p * 24;
  for (int i = 0; i < 8; i++) stt[i] = r3.v[i];
}

template <class Fb>
ACT_HD void prove_bits_lane(const ProveArgs& a, uint32_t gid, Fb& fb) {
  const int L = a.P.L;
  uint32_t p = gid / (uint32_t)L, j = gid % (uint32_t)L;
  const bool live = p < a.n;
  RngView rv{a.rng + (size_t)(live ? p : 0) * rng_bytes(L), L};
  uint32_t bit = 0;
  sc s_j = sc_zero(), s_jp = sc_zero(), g_j = sc_zero(), z_j = sc_zero();
  if (live) {
    const uint8_t* tok = a.tok + (size_t)p * 160;
    sc m = sc_sub(load_sc(tok + 128), load_sc(a.s + (size_t)p * 32));         // c - s (:996)
    bit = (m.v[j >> 5] >> (j & 31)) & 1u;                                     // bits_of (:902-915)
    s_j = rv.s_i(j); s_jp = rv.s_i_prime(j); g_j = rv.gamma_i(j); z_j = rv.z(j);
  }
  // Half scale throughout (k_prove_enc encodes the doubles): Com_j / 2 = i_j (h1 / 2) + (s_j / 2) h3 (+ (k* / 2) h2)
  // real branch: s'_j h3 (+ k0' h2);  simulated: (z_j - gamma_j s_j) h3 -/+ gamma_j h1 (+ (w0 - gamma_0 k*) h2)
  // (the four products run in EVERY lane, live or not -- a lane past the batch multiplies by zero: the ct build's matrix-core
  // look-up needs all 64 lanes of a wavefront in it, and a lane that sits out costs the same as one that computes)
  ge com = ge_identity(), real = ge_identity(), sim = ge_identity();
  fb.stage(BASE_H3);
  com = fb.mul(com, BASE_H3, sc_half(s_j));
  real = fb.mul(real, BASE_H3, sc_half(s_jp));
  sim = fb.mul(sim, BASE_H3, sc_half(sc_sub(z_j, sc_mul(g_j, s_j))));
  fb.stage(BASE_H1);
  sim = fb.mul(sim, BASE_H1, sc_half(bit ? sc_neg(g_j) : g_j));
  if (!live) return;
#if defined(ACT_CT_SECRET_TABLES)
  {                                                                                // both entries read, the bit picks with masks
    const ge_niels e0 = niels_load(a.P.half_h1), e1 = niels_load(a.P.half_h1 + NIELS_WORDS);
    const uint32_t m = fe_mask(bit != 0);
    ge_niels q; q.ypx = fe_select_m(e0.ypx, e1.ypx, m); q.ymx = fe_select_m(e0.ymx, e1.ymx, m); q.xy2d = fe_select_m(e0.xy2d, e1.xy2d, m);
    com = ge_madd(com, q);
  }
#else
  com = ge_madd(com, niels_load(a.P.half_h1 + (size_t)bit * NIELS_WORDS));        // entry 0 = identity, entry 1 = h1 / 2
#endif
  if (j == 0) {
    const uint32_t* d3 = a.d3 + (size_t)p * 3 * GE_WORDS;
    com = ge_add(com, ge_load(d3)); real = ge_add(real, ge_load(d3 + GE_WORDS)); sim = ge_add(sim, ge_load(d3 + 2 * GE_WORDS));
  }
  // C'_j0 = bit ? sim : real, C'_j1 = bit ? real : sim (:1025-1050), chosen with masks
  const uint32_t mb = fe_mask(bit != 0);
  ge c0, c1;
  c0.X = fe_select_m(real.X, sim.X, mb); c0.Y = fe_select_m(real.Y, sim.Y, mb); c0.Z = fe_select_m(real.Z, sim.Z, mb); c0.T = fe_select_m(real.T, sim.T, mb);
  c1.X = fe_select_m(sim.X, real.X, mb); c1.Y = fe_select_m(sim.Y, real.Y, mb); c1.Z = fe_select_m(sim.Z, real.Z, mb); c1.T = fe_select_m(sim.T, real.T, mb);
  uint32_t* hp = a.half + (size_t)gid * BUCKET_WORDS;
  bucket_store(hp, com); bucket_store(hp + GE_WORDS, c0); bucket_store(hp + 2 * GE_WORDS, c1);
}

// lane = PROVE_ENC_BATCH consecutive half-points; point q = 3 * (p * L + j) + c is slot c of lane (p, j)
constexpr int PROVE_ENC_BATCH = 32;
ACT_HD void prove_enc_lane(const ProveArgs& a, uint64_t q0) {
  const uint32_t L = (uint32_t)a.P.L;
  const uint64_t total = (uint64_t)a.n * L * 3u;
  if (q0 >= total) return;
  const int count = (int)(total - q0 < (uint64_t)PROVE_ENC_BATCH ? total - q0 : (uint64_t)PROVE_ENC_BATCH);
  const ProofLayout pl{a.P.L}; const SpendTranscript st{a.P.L};
  dc_encode_batch<PROVE_ENC_BATCH>(
      count,
      [&](int i) { uint64_t q = q0 + (uint64_t)i; return a.half + (size_t)(q / 3u) * BUCKET_WORDS + (q % 3u) * GE_WORDS; },
      [&](int i, const uint32_t* enc) {
        uint64_t q = q0 + (uint64_t)i, lane = q / 3u; uint32_t c = (uint32_t)(q % 3u);
        uint32_t p = (uint32_t)(lane / L), j = (uint32_t)(lane % L);
        uint8_t* el = a.tr + (size_t)p * a.tr_stride + 184;
        if (c == 0) { tr_put_aligned(el + 40 * st.el_com(j), enc); store8(a.proof + (size_t)p * pl.bytes() + 32 * pl.com(j), enc); }
        else tr_put_aligned(el + 40 * st.el_cprime(j, (int)c - 1), enc);
      });
}

template <class Fb>
ACT_HD void prove_tail_lane(const ProveArgs& a, uint32_t p, Fb& fb) {
  const bool live = p < a.n;
  const int L = a.P.L;
  const SpendTranscript st{L};
  RngView rv{a.rng + (size_t)(live ? p : 0) * rng_bytes(L), L};
  ge cc = ge_identity();                                                      // C = -c' h1 + k' h2 + s' h3 (:1059)
  fb.stage(BASE_H1);                                                          // (every lane multiplies: see prove_head_lane)
  cc = fb.mul(cc, BASE_H1, sc_neg(rv.c_prime()));
  fb.stage(BASE_H2);
  cc = fb.mul(cc, BASE_H2, rv.k_prime());
  fb.stage(BASE_H3);
  cc = fb.mul(cc, BASE_H3, rv.s_prime());
  if (!live) return;
  sc rstar = sc_zero();                                                       // r* = sum s_j 2^j (:1052-1056), Horner
  for (int j = L - 1; j >= 0; j--) rstar = sc_add(sc_add(rstar, rstar), rv.s_i(j));
  uint8_t* el = a.tr + (size_t)p * a.tr_stride + 184;
  uint32_t enc[8]; ristretto_encode(enc, cc); tr_put_aligned(el + 40 * st.el_c(), enc);
  uint32_t* stt = a.state + (size_t)p * 24;
  for (int i = 0; i < 8; i++) stt[8 + i] = rstar.v[i];
}

ACT_HD void prove_resp_lane(const ProveArgs& a, uint32_t gid) {
  const int L = a.P.L;
  uint32_t p = gid / (uint32_t)L, j = gid % (uint32_t)L;
  if (p >= a.n) return;
  const ProofLayout pl{L};
  const uint8_t* tok = a.tok + (size_t)p * 160;
  uint8_t* rec = a.proof + (size_t)p * pl.bytes();
  RngView rv{a.rng + (size_t)p * rng_bytes(L), L};
  const bool bad = (a.flags[p] & FLAG_UNDECODABLE) != 0;
  uint32_t w[16];
  for (int i = 0; i < 16; i++) w[i] = a.xof[(size_t)p * 16 + i];
  sc gamma = sc_from_wide_words(w);                                           // :1061-1070
  sc cs = load_sc(tok + 128), s = load_sc(a.s + (size_t)p * 32);
  sc m = sc_sub(cs, s);
  uint32_t bit = (m.v[j >> 5] >> (j & 31)) & 1u;
  sc s_j = rv.s_i(j), s_jp = rv.s_i_prime(j), g_j = rv.gamma_i(j), z_j = rv.z(j);
  sc g00 = bit ? g_j : sc_sub(gamma, g_j);                                    // :1078-1082, :1105-1109
  sc g01 = sc_sub(gamma, g00);
  sc resp0 = sc_muladd(g00, s_j, s_jp), resp1 = sc_muladd(g01, s_j, s_jp);
  sc z0 = bit ? z_j : resp0, z1 = bit ? resp1 : z_j;                          // :1094-1103, :1110-1119
  if (bad) { g00 = sc_zero(); z0 = sc_zero(); z1 = sc_zero(); }
  store_sc(rec + 32 * pl.gamma0(j), g00);
  store_sc(rec + 32 * pl.z(j, 0), z0); store_sc(rec + 32 * pl.z(j, 1), z1);
  if (bad) { zero8(rec + 32 * pl.com(j)); }
  if (j != 0) return;
  // proof-level responses (:1072-1076, :1083-1092, :1121-1122) and PreRefund (:1124-1128)
  const uint32_t* stt = a.state + (size_t)p * 24;
  sc r3, rstar;
  for (int i = 0; i < 8; i++) { r3.v[i] = stt[i]; rstar.v[i] = stt[8 + i]; }
  sc e = load_sc(tok + 32), r = load_sc(tok + 96);
  sc ng = sc_neg(gamma), k_star = rv.k_star(), k0p = rv.k0_prime(), w0 = rv.w0();
  sc w00 = bit ? w0 : sc_muladd(g00, k_star, k0p);
  sc w01 = bit ? sc_muladd(g01, k_star, k0p) : w0;
  uint8_t* pre = a.prerefund + (size_t)p * 96;
  if (bad) {
    for (int f = 0; f < 4; f++) zero8(rec + 32 * f);
    for (int f = pl.gamma(); f <= pl.w01(); f++) zero8(rec + 32 * f);
    zero8(rec + 32 * pl.k_bar()); zero8(rec + 32 * pl.s_bar());
    zero8(pre); zero8(pre + 32); zero8(pre + 64);
    a.status[p] = 255;
    return;
  }
  store_sc(rec + 32 * pl.gamma(), gamma);
  store_sc(rec + 32 * pl.e_bar(), sc_muladd(ng, e, rv.e_prime()));
  store_sc(rec + 32 * pl.r2_bar(), sc_muladd(gamma, rv.r2(), rv.r2_prime()));
  store_sc(rec + 32 * pl.r3_bar(), sc_muladd(gamma, r3, rv.r3_prime()));
  store_sc(rec + 32 * pl.c_bar(), sc_muladd(ng, cs, rv.c_prime()));
  store_sc(rec + 32 * pl.r_bar(), sc_muladd(ng, r, rv.r_prime()));
  store_sc(rec + 32 * pl.w00(), w00); store_sc(rec + 32 * pl.w01(), w01);
  store_sc(rec + 32 * pl.k_bar(), sc_muladd(gamma, k_star, rv.k_prime()));
  store_sc(rec + 32 * pl.s_bar(), sc_muladd(gamma, rstar, rv.s_prime()));
  store_sc(pre, rstar); store_sc(pre + 32, k_star); store_sc(pre + 64, m);   // PreRefund r | k | m
  a.status[p] = 0;
}

}  // namespace act
