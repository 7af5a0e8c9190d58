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
// 253 squarings + 73 multiplications.  Through sc_mul (schoolbook product, then the three-stage fold of sc_reduce512 with its
// conditional subtractions) that was 0.5 - 0.6 ms of one wavefront -- a third of a single-item `issue` (profiles/r05_tiny_timing.txt)
// and a fifth of k_sign_a.  Inside the exponentiation the operands stay in Montgomery form (R = 2^256): one CIOS pass per product,
// 8 x (8 + 1 + 5) multiply-accumulates (l = 2^252 + c has three zero words), no subtraction until the end -- R = 16 l, so operands
// below 2 l give results below 2 l: (a b + m l) / R < (4 l^2 + R l) / R < 2 l.
ACT_HD void sc_mont_mul(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]) {
  constexpr uint32_t NPRIME = 0x12547e1bu;      // -l^-1 mod 2^32
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; i++) t[i] = 0u;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (uint64_t)a[j] * b[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
    t[8] += (uint32_t)c;                         // (t < 2^287 throughout: nine words)
    const uint32_t m = t[0] * NPRIME;
    c = ((uint64_t)m * sc_l_word(0) + t[0]) >> 32;                    // the low word becomes zero
#pragma unroll
    for (int j = 1; j < 8; j++) {
      c += t[j];
      if (j < 4 || j == 7) c += (uint64_t)m * sc_l_word(j);           // words 4, 5, 6 of l are zero
      t[j - 1] = (uint32_t)c; c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c; t[8] = (uint32_t)(c >> 32);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = t[i];     // < 2 l < 2^254: t[8] is zero
}
ACT_HD sc sc_invert(const sc& a) {
  constexpr uint32_t R2[8] = {0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu};      // R^2 mod l
  constexpr uint32_t R1[8] = {0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu};      // R mod l
  uint32_t e[8];                                 // l - 2
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = sc_l_word(i);
  e[0] -= 2u;
  uint32_t am[8], acc[8], r2[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { r2[i] = R2[i]; acc[i] = R1[i]; }
  sc_mont_mul(am, a.v, r2);                      // a R
  for (int i = 252; i >= 0; i--) {
    sc_mont_mul(acc, acc, acc);
    if ((e[i >> 5] >> (i & 31)) & 1u) sc_mont_mul(acc, acc, am);
  }
  uint32_t one[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  sc r;
  sc_mont_mul(r.v, acc, one);                    // out of Montgomery form: < 2 l
  sc_cond_sub_l(r.v, 1);
  return r;
}

}  // namespace act
