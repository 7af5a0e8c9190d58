// sc25519.h — arithmetic mod l = 2^252 + 27742317777372353535851937790883648493 for gfx950 lanes:
// eight u32 limbs per scalar, one scalar per lane.  Restates what the reference gets from
// curve25519-dalek's Scalar (from_bytes_mod_order_wide at /root/reference/src/transcript.rs:153
// and inside every Scalar::random; + - * neg invert at src/lib.rs:478-479, 645, 660, 801, 992,
// 1052-1122).  Reduction folds 2^252 = -c (mod l), c = l - 2^252 (125 bits), three times.
// Compiles under hipcc (device) and g++ (tests/hostcheck only).
#pragma once
#include "fe25519.h"

namespace act {

struct sc { uint32_t v[8]; };

ACT_HD uint32_t sc_l_word(int i) {
  constexpr uint32_t L[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};
  return L[i];
}

template <int NA, int NB>
ACT_HD void bn_mul(uint32_t* out, const uint32_t* a, const uint32_t* b) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) out[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t; carry = (uint32_t)(t >> 32);
    }
    out[i + NB] = carry;
  }
}
template <int N>
ACT_HD uint32_t bn_add(uint32_t* out, const uint32_t* a, const uint32_t* b) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) { c += (uint64_t)a[i] + b[i]; out[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)c;
}
template <int N>
ACT_HD uint32_t bn_sub(uint32_t* out, const uint32_t* a, const uint32_t* b) {   // returns borrow
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) { c += (int64_t)a[i] - (int64_t)b[i]; out[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)(c & 1);
}
// x (N limbs) -> lo = x mod 2^252 (8 limbs), hi = x >> 252 (N-7 limbs)
template <int N>
ACT_HD void bn_split252(uint32_t* lo, uint32_t* hi, const uint32_t* x) {
#pragma unroll
  for (int i = 0; i < 8; i++) lo[i] = i < N ? x[i] : 0u;
  lo[7] &= 0x0fffffffu;
#pragma unroll
  for (int i = 0; i < N - 7; i++) {
    uint32_t a = x[i + 7] >> 28, b = (i + 8 < N) ? (x[i + 8] << 4) : 0u;
    hi[i] = a | b;
  }
}
// r (8 limbs, < 2^256): subtract l while r >= l, at most `times` times
ACT_HD void sc_cond_sub_l(uint32_t* r, int times) {
  uint32_t l[8];
#pragma unroll
  for (int i = 0; i < 8; i++) l[i] = sc_l_word(i);
  for (int t = 0; t < times; t++) {
    uint32_t d[8];
    uint32_t keep = fe_mask(bn_sub<8>(d, r, l) != 0u);   // borrow: r < l, keep r
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = sel32(d[i], r[i], keep);
  }
}
// 512-bit little-endian words -> scalar mod l
ACT_HD sc sc_reduce512(const uint32_t x[16]) {
  uint32_t c[4];
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = sc_l_word(i);
  uint32_t lo0[8], h0[9], y1[13], lo1[8], h1[6], y2[10], lo2[8], h2[3], y3[8];
  bn_split252<16>(lo0, h0, x);            // h0 < 2^260
  bn_mul<9, 4>(y1, h0, c);                // < 2^385
  bn_split252<13>(lo1, h1, y1);           // h1 < 2^133 (5 limbs used of 6)
  bn_mul<6, 4>(y2, h1, c);                // < 2^258
  bn_split252<10>(lo2, h2, y2);           // h2 < 2^6
  bn_mul<3, 4>(y3, h2, c);                // < 2^131 (7 limbs written)
  y3[7] = 0u;
  // x = lo0 - lo1 + lo2 - y3 (mod l); add 2l to stay positive
  uint32_t A[8], B[8], l2[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; i++) l[i] = sc_l_word(i);
  bn_add<8>(l2, l, l);
  bn_add<8>(A, lo0, lo2);
  bn_add<8>(A, A, l2);
  bn_add<8>(B, lo1, y3);
  bn_sub<8>(A, A, B);
  sc_cond_sub_l(A, 4);
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = A[i];
  return r;
}
ACT_HD sc sc_zero() { sc r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
ACT_HD sc sc_one() { sc r = sc_zero(); r.v[0] = 1; return r; }
// from_bytes_mod_order on 8 little-endian words (value < 2^256)
ACT_HD sc sc_from_words(const uint32_t w[8]) {
  uint32_t q = w[7] >> 28;                 // floor(x / 2^252) in 0..15
  uint32_t c[4], qc[5], lo[8], r[8], l[8], qcw[8];
#pragma unroll
  for (int i = 0; i < 4; i++) c[i] = sc_l_word(i);
#pragma unroll
  for (int i = 0; i < 8; i++) { lo[i] = w[i]; l[i] = sc_l_word(i); }
  lo[7] &= 0x0fffffffu;
  bn_mul<1, 4>(qc, &q, c);                 // q*c < 2^129
#pragma unroll
  for (int i = 0; i < 8; i++) qcw[i] = i < 5 ? qc[i] : 0u;
  bn_add<8>(r, lo, l);                     // lo + l - q*c in (0, 2l)
  bn_sub<8>(r, r, qcw);
  sc_cond_sub_l(r, 1);
  sc s;
#pragma unroll
  for (int i = 0; i < 8; i++) s.v[i] = r[i];
  return s;
}
ACT_HD sc sc_from_wide_words(const uint32_t w[16]) { return sc_reduce512(w); }
ACT_HD sc sc_add(const sc& a, const sc& b) { sc r; bn_add<8>(r.v, a.v, b.v); sc_cond_sub_l(r.v, 1); return r; }
ACT_HD sc sc_sub(const sc& a, const sc& b) {
  uint32_t l[8]; sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) l[i] = sc_l_word(i);
  bn_add<8>(r.v, a.v, l); bn_sub<8>(r.v, r.v, b.v); sc_cond_sub_l(r.v, 1); return r;
}
ACT_HD sc sc_neg(const sc& a) { return sc_sub(sc_zero(), a); }
// a / 2 mod l for canonical a: a >> 1 if even, (a + l) >> 1 otherwise (l is odd, a + l < 2^254)
ACT_HD sc sc_half(const sc& a) {
  const uint32_t odd = 0u - (a.v[0] & 1u);
  uint32_t t[8]; uint64_t c = 0;
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + (sc_l_word(i) & odd); t[i] = (uint32_t)c; c >>= 32; }
  sc r;
  for (int i = 0; i < 7; i++) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
  r.v[7] = t[7] >> 1;
  return r;
}
ACT_HD sc sc_mul(const sc& a, const sc& b) { uint32_t t[16]; bn_mul<8, 8>(t, a.v, b.v); return sc_reduce512(t); }
ACT_HD sc sc_muladd(const sc& a, const sc& b, const sc& c) { return sc_add(sc_mul(a, b), c); }
ACT_HD bool sc_equal(const sc& a, const sc& b) { uint32_t d = 0; for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i]; return d == 0; }
ACT_HD bool sc_is_zero(const sc& a) { uint32_t d = 0; for (int i = 0; i < 8; i++) d |= a.v[i]; return d == 0; }
// ---- Scalar::invert (src/lib.rs:645, 849, 992): a^(l-2), 0 -> 0 ----------------------------------------------------------------
// 253 squarings + 73 multiplications.  Through sc_mul (schoolbook product on 32-bit words with a carry per step, then the three-stage
// fold of sc_reduce512 with its conditional subtractions) that was 0.5 - 0.6 ms of one wavefront -- a third of a single-item
// `issue` (profiles/r05_tiny_timing.txt) and a fifth of k_sign_a.  Inside the exponentiation the operands are kept the way the
// field elements are (fe25519.h): ten limbs of 26 bits, so that a column of the product -- ten 52-bit terms, plus ten more from
// the reduction -- is a plain 64-bit sum of v_mad_u64_u32 results with no carry anywhere, and in Montgomery form (R = 2^260): the
// reduction is ten more passes of six multiply-accumulates (l = 2^252 + c has four zero limbs).  R > 2^7 l, so operands below
// 2 l give results below 2 l and nothing is subtracted until the end.
constexpr int SCM_LIMBS = 10;
constexpr uint32_t SCM_MASK = (1u << 26) - 1u;
ACT_HD uint32_t scm_l(int i) {
  constexpr uint32_t Lm[SCM_LIMBS] = {0x0f5d3edu, 0x098c697u, 0x1cd6581u, 0x37a8bdeu, 0x014def9u, 0u, 0u, 0u, 0u, 0x0040000u};
  return Lm[i];
}
// out = a b R^-1 mod l (lazily: < 2 l), limbs < 2^26
ACT_HD void scm_mul(uint32_t out[SCM_LIMBS], const uint32_t a[SCM_LIMBS], const uint32_t b[SCM_LIMBS]) {
  constexpr uint32_t NPRIME = 0x2547e1bu;        // -l^-1 mod 2^26
  uint64_t col[2 * SCM_LIMBS];
#pragma unroll
  for (int k = 0; k < 2 * SCM_LIMBS; k++) col[k] = 0;
#pragma unroll
  for (int i = 0; i < SCM_LIMBS; i++)
#pragma unroll
    for (int j = 0; j < SCM_LIMBS; j++) col[i + j] += (uint64_t)a[i] * b[j];
#pragma unroll
  for (int i = 0; i < SCM_LIMBS; i++) {
    const uint32_t m = ((uint32_t)col[i] * NPRIME) & SCM_MASK;
#pragma unroll
    for (int j = 0; j < SCM_LIMBS; j++) if (j < 5 || j == 9) col[i + j] += (uint64_t)m * scm_l(j);      // limbs 5..8 of l are zero
    col[i + 1] += col[i] >> 26;                  // the low 26 bits of col[i] are zero now
  }
#pragma unroll
  for (int k = SCM_LIMBS; k < 2 * SCM_LIMBS - 1; k++) { col[k + 1] += col[k] >> 26; out[k - SCM_LIMBS] = (uint32_t)col[k] & SCM_MASK; }
  out[SCM_LIMBS - 1] = (uint32_t)col[2 * SCM_LIMBS - 1];       // value < 2 l < 2^254: fits its 26 bits
}
ACT_HD sc sc_invert(const sc& a) {
  constexpr uint32_t R2[SCM_LIMBS] = {0x152d13bu, 0x274997au, 0x1bea69fu, 0x358f1c5u, 0x3687604u, 0x16f9972u, 0x33d217fu, 0x0f73bb1u, 0x37c309au, 0x0025046u};      // R^2 mod l
  constexpr uint32_t R1[SCM_LIMBS] = {0x321e6edu, 0x3d22f59u, 0x067e45au, 0x0eead6bu, 0x335e51bu, 0x3fffffau, 0x3ffffffu, 0x3ffffffu, 0x3ffffffu, 0x003ffffu};      // R mod l
  uint32_t e[8];                                 // l - 2
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = sc_l_word(i);
  e[0] -= 2u;
  uint32_t al[SCM_LIMBS], am[SCM_LIMBS], acc[SCM_LIMBS], r2[SCM_LIMBS];
#pragma unroll
  for (int i = 0; i < SCM_LIMBS; i++) {          // 8 x 32 bits -> 10 x 26 bits
    const int bit = 26 * i, w = bit >> 5, sh = bit & 31;
    uint32_t v = a.v[w] >> sh;
    if (sh > 6 && w + 1 < 8) v |= a.v[w + 1] << (32 - sh);
    al[i] = v & SCM_MASK;
    r2[i] = R2[i]; acc[i] = R1[i];
  }
  scm_mul(am, al, r2);                           // a R
  for (int i = 252; i >= 0; i--) {
    scm_mul(acc, acc, acc);
    if ((e[i >> 5] >> (i & 31)) & 1u) scm_mul(acc, acc, am);
  }
  uint32_t one[SCM_LIMBS] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  uint32_t o[SCM_LIMBS];
  scm_mul(o, acc, one);                          // out of Montgomery form: < 2 l
  sc r;
#pragma unroll
  for (int w = 0; w < 8; w++) {                  // 10 x 26 bits -> 8 x 32 bits
    const int bit = 32 * w, i = bit / 26, sh = bit % 26;
    uint32_t v = o[i] >> sh;
    if (i + 1 < SCM_LIMBS) v |= o[i + 1] << (26 - sh);
    if (sh > 20 && i + 2 < SCM_LIMBS) v |= o[i + 2] << (52 - sh);
    r.v[w] = v;
  }
  sc_cond_sub_l(r.v, 1);
  return r;
}

}  // namespace act
