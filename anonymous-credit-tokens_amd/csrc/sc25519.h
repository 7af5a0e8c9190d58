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
// a^(l-2); 0 -> 0 (Scalar::invert, src/lib.rs:645, 849, 992)
ACT_HD sc sc_invert(const sc& a) {
  // l - 2 as words
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = sc_l_word(i);
  e[0] -= 2u;
  sc acc = sc_one();
  for (int i = 252; i >= 0; i--) {
    acc = sc_mul(acc, acc);
    if ((e[i >> 5] >> (i & 31)) & 1u) acc = sc_mul(acc, a);
  }
  return acc;
}

}  // namespace act
