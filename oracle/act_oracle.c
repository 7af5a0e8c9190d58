/* act_oracle.c — CPU restatement (plain C11, no dependencies) of the sigma-protocol hot path of
 * anonymous-credit-tokens v0.2.1.  TEST INFRASTRUCTURE ONLY — see act_oracle.h for who may use it.
 *
 * The arithmetic the reference delegates to curve25519-dalek 4.1.3 and blake3 1.8.2 (neither is
 * under /root/reference; Cargo.lock:267-268, :83-84) is restated here from the published
 * algorithms: GF(2^255-19) in radix 2^51, Z_l with 64-bit limbs, twisted-Edwards extended
 * coordinates, ristretto255 per RFC 9496, BLAKE3 per its specification.  The PROTOCOL functions
 * follow /root/reference/src/lib.rs and src/transcript.rs line by line and keep the reference's
 * operation structure (constant-time radix-16 variable-base mult for every `point * scalar`,
 * 32x8 affine-Niels tables for every `&table * &scalar`, 128 separate mults for K', both OR-proof
 * branches in the prover) so that timing it is an honest "CPU port" baseline.
 *
 * Parity: unpinned against the crate (see header); pinned against upstream BLAKE3, OpenSSL
 * Ed25519, RFC 9496 vectors and oracle/pymodel.py via tests/golden/.
 */
#include "act_oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t fe[5];
#define M51 0x7ffffffffffffULL

/* ============================ GF(2^255-19), radix 2^51 ============================ */
static const fe FE_D = {0x34dca135978a3ULL, 0x1a8283b156ebdULL, 0x5e7a26001c029ULL, 0x739c663a03cbbULL, 0x52036cee2b6ffULL};
static const fe FE_D2 = {0x69b9426b2f159ULL, 0x35050762add7aULL, 0x3cf44c0038052ULL, 0x6738cc7407977ULL, 0x2406d9dc56dffULL};
static const fe FE_SQRT_M1 = {0x61b274a0ea0b0ULL, 0x0d5a5fc8f189dULL, 0x7ef5e9cbd0c60ULL, 0x78595a6804c9eULL, 0x2b8324804fc1dULL};
static const fe FE_SQRT_AD_MINUS_ONE = {0x7f6a0497b2e1bULL, 0x1836f0a97afd2ULL, 0x7d747f6be7638ULL, 0x456079e7e6498ULL, 0x376931bf2b834ULL};
static const fe FE_INVSQRT_A_MINUS_D = {0x0fdaa805d40eaULL, 0x2eb482e57d339ULL, 0x007610274bc58ULL, 0x6510b613dc8ffULL, 0x786c8905cfaffULL};
static const fe FE_ONE_MINUS_D_SQ = {0x409c1945fc176ULL, 0x719abc6a1fc4fULL, 0x1c37f90b20684ULL, 0x06bccca55eedfULL, 0x029072a8b2b3eULL};
static const fe FE_D_MINUS_ONE_SQ = {0x55aaa44ed4d20ULL, 0x59603c3332635ULL, 0x26d3baf4a7928ULL, 0x120a66e6997a9ULL, 0x5968b37af66c2ULL};
static const fe FE_BX = {0x62d608f25d51aULL, 0x412a4b4f6592aULL, 0x75b7171a4b31dULL, 0x1ff60527118feULL, 0x216936d3cd6e5ULL};
static const fe FE_BY = {0x6666666666658ULL, 0x4ccccccccccccULL, 0x1999999999999ULL, 0x3333333333333ULL, 0x6666666666666ULL};
static const fe FE_BT = {0x68ab3a5b7dda3ULL, 0x00eea2a5eadbbULL, 0x2af8df483c27eULL, 0x332b375274732ULL, 0x67875f0fd78b7ULL};

static void fe_copy(fe h, const fe f) { memcpy(h, f, sizeof(fe)); }
static void fe_0(fe h) { memset(h, 0, sizeof(fe)); }
static void fe_1(fe h) { fe_0(h); h[0] = 1; }

static void fe_carry(fe h) {
  uint64_t c;
  c = h[0] >> 51; h[0] &= M51; h[1] += c;
  c = h[1] >> 51; h[1] &= M51; h[2] += c;
  c = h[2] >> 51; h[2] &= M51; h[3] += c;
  c = h[3] >> 51; h[3] &= M51; h[4] += c;
  c = h[4] >> 51; h[4] &= M51; h[0] += 19 * c;
  c = h[0] >> 51; h[0] &= M51; h[1] += c;
}
static void fe_add(fe h, const fe f, const fe g) {
  for (int i = 0; i < 5; i++) h[i] = f[i] + g[i];
  fe_carry(h);
}
static void fe_sub(fe h, const fe f, const fe g) { /* f - g + 4p, limbs of g must be < 2^53 */
  h[0] = f[0] + 0x1fffffffffffb4ULL - g[0];
  for (int i = 1; i < 5; i++) h[i] = f[i] + 0x1ffffffffffffcULL - g[i];
  fe_carry(h);
}
static void fe_neg(fe h, const fe f) { fe z; fe_0(z); fe_sub(h, z, f); }

static void fe_mul(fe h, const fe f, const fe g) {
  uint64_t f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], f4 = f[4];
  uint64_t g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
  uint64_t g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4;
  u128 r0 = (u128)f0 * g0 + (u128)f1 * g4_19 + (u128)f2 * g3_19 + (u128)f3 * g2_19 + (u128)f4 * g1_19;
  u128 r1 = (u128)f0 * g1 + (u128)f1 * g0 + (u128)f2 * g4_19 + (u128)f3 * g3_19 + (u128)f4 * g2_19;
  u128 r2 = (u128)f0 * g2 + (u128)f1 * g1 + (u128)f2 * g0 + (u128)f3 * g4_19 + (u128)f4 * g3_19;
  u128 r3 = (u128)f0 * g3 + (u128)f1 * g2 + (u128)f2 * g1 + (u128)f3 * g0 + (u128)f4 * g4_19;
  u128 r4 = (u128)f0 * g4 + (u128)f1 * g3 + (u128)f2 * g2 + (u128)f3 * g1 + (u128)f4 * g0;
  uint64_t c;
  r1 += (uint64_t)(r0 >> 51); h[0] = (uint64_t)r0 & M51;
  r2 += (uint64_t)(r1 >> 51); h[1] = (uint64_t)r1 & M51;
  r3 += (uint64_t)(r2 >> 51); h[2] = (uint64_t)r2 & M51;
  r4 += (uint64_t)(r3 >> 51); h[3] = (uint64_t)r3 & M51;
  c = (uint64_t)(r4 >> 51); h[4] = (uint64_t)r4 & M51;
  h[0] += 19 * c;
  c = h[0] >> 51; h[0] &= M51; h[1] += c;
}
static void fe_sq(fe h, const fe f) {
  uint64_t f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], f4 = f[4];
  uint64_t f0_2 = 2 * f0, f1_2 = 2 * f1, f3_19 = 19 * f3, f4_19 = 19 * f4;
  u128 r0 = (u128)f0 * f0 + (u128)f1_2 * f4_19 + (u128)(2 * f2) * f3_19;
  u128 r1 = (u128)f0_2 * f1 + (u128)(2 * f2) * f4_19 + (u128)f3 * f3_19;
  u128 r2 = (u128)f0_2 * f2 + (u128)f1 * f1 + (u128)(2 * f3) * f4_19;
  u128 r3 = (u128)f0_2 * f3 + (u128)f1_2 * f2 + (u128)f4 * f4_19;
  u128 r4 = (u128)f0_2 * f4 + (u128)f1_2 * f3 + (u128)f2 * f2;
  uint64_t c;
  r1 += (uint64_t)(r0 >> 51); h[0] = (uint64_t)r0 & M51;
  r2 += (uint64_t)(r1 >> 51); h[1] = (uint64_t)r1 & M51;
  r3 += (uint64_t)(r2 >> 51); h[2] = (uint64_t)r2 & M51;
  r4 += (uint64_t)(r3 >> 51); h[3] = (uint64_t)r3 & M51;
  c = (uint64_t)(r4 >> 51); h[4] = (uint64_t)r4 & M51;
  h[0] += 19 * c;
  c = h[0] >> 51; h[0] &= M51; h[1] += c;
}
static void fe_sqn(fe h, const fe f, int n) { fe_sq(h, f); for (int i = 1; i < n; i++) fe_sq(h, h); }

static void fe_frombytes(fe h, const uint8_t s[32]) { /* ignores bit 255 */
  uint64_t w[4];
  for (int i = 0; i < 4; i++) { w[i] = 0; for (int j = 7; j >= 0; j--) w[i] = (w[i] << 8) | s[8 * i + j]; }
  h[0] = w[0] & M51;
  h[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
  h[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
  h[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
  h[4] = (w[3] >> 12) & M51;
}
static void fe_tobytes(uint8_t s[32], const fe f) { /* canonical encoding */
  fe t; fe_copy(t, f); fe_carry(t); fe_carry(t);
  uint64_t q = (t[0] + 19) >> 51;
  q = (t[1] + q) >> 51; q = (t[2] + q) >> 51; q = (t[3] + q) >> 51; q = (t[4] + q) >> 51;
  t[0] += 19 * q;
  uint64_t c;
  c = t[0] >> 51; t[0] &= M51; t[1] += c;
  c = t[1] >> 51; t[1] &= M51; t[2] += c;
  c = t[2] >> 51; t[2] &= M51; t[3] += c;
  c = t[3] >> 51; t[3] &= M51; t[4] += c;
  t[4] &= M51;
  uint64_t w[4];
  w[0] = t[0] | (t[1] << 51);
  w[1] = (t[1] >> 13) | (t[2] << 38);
  w[2] = (t[2] >> 26) | (t[3] << 25);
  w[3] = (t[3] >> 39) | (t[4] << 12);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) s[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}
static int fe_isneg(const fe f) { uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1; }
static int fe_iszero(const fe f) { uint8_t s[32]; fe_tobytes(s, f); uint8_t r = 0; for (int i = 0; i < 32; i++) r |= s[i]; return r == 0; }
static int fe_eq(const fe f, const fe g) { uint8_t a[32], b[32]; fe_tobytes(a, f); fe_tobytes(b, g); return memcmp(a, b, 32) == 0; }
static void fe_cmov(fe f, const fe g, int b) { uint64_t m = (uint64_t)0 - (uint64_t)(b != 0); for (int i = 0; i < 5; i++) f[i] ^= m & (f[i] ^ g[i]); }
static void fe_cneg(fe f, int b) { fe n; fe_neg(n, f); fe_cmov(f, n, b); }
static void fe_abs(fe f) { fe_cneg(f, fe_isneg(f)); }

/* z^(2^252-3) and the shared prefix z^(2^250-1), z^11 */
static void fe_pow_core(fe t250, fe z11, const fe z) {
  fe t0, t1, t2;
  fe_sq(t0, z);                /* 2 */
  fe_sqn(t1, t0, 2);           /* 8 */
  fe_mul(t1, z, t1);           /* 9 */
  fe_mul(z11, t0, t1);         /* 11 */
  fe_sq(t0, z11);              /* 22 */
  fe_mul(t0, t1, t0);          /* 31 = 2^5-1 */
  fe_sqn(t1, t0, 5); fe_mul(t0, t1, t0);     /* 2^10-1 */
  fe_sqn(t1, t0, 10); fe_mul(t1, t1, t0);    /* 2^20-1 */
  fe_sqn(t2, t1, 20); fe_mul(t1, t2, t1);    /* 2^40-1 */
  fe_sqn(t1, t1, 10); fe_mul(t0, t1, t0);    /* 2^50-1 */
  fe_sqn(t1, t0, 50); fe_mul(t1, t1, t0);    /* 2^100-1 */
  fe_sqn(t2, t1, 100); fe_mul(t1, t2, t1);   /* 2^200-1 */
  fe_sqn(t1, t1, 50); fe_mul(t250, t1, t0);  /* 2^250-1 */
}
static void fe_invert(fe out, const fe z) { /* z^(p-2) = z^(2^255-21) */
  fe t, z11; fe_pow_core(t, z11, z); fe_sqn(t, t, 5); fe_mul(out, t, z11);
}
static void fe_pow22523(fe out, const fe z) { /* z^((p-5)/8) = z^(2^252-3) */
  fe t, z11; fe_pow_core(t, z11, z); fe_sqn(t, t, 2); fe_mul(out, t, z);
}
/* RFC 9496 4.2 SQRT_RATIO_M1 */
static int fe_sqrt_ratio_m1(fe r, const fe u, const fe v) {
  fe v3, v7, t, check, neg_u, neg_u_i;
  fe_sq(v3, v); fe_mul(v3, v3, v);
  fe_sq(v7, v3); fe_mul(v7, v7, v);
  fe_mul(t, u, v7); fe_pow22523(t, t);
  fe_mul(r, u, v3); fe_mul(r, r, t);
  fe_sq(check, r); fe_mul(check, check, v);
  fe_neg(neg_u, u); fe_mul(neg_u_i, neg_u, FE_SQRT_M1);
  int correct = fe_eq(check, u), flipped = fe_eq(check, neg_u), flipped_i = fe_eq(check, neg_u_i);
  fe ri; fe_mul(ri, r, FE_SQRT_M1);
  fe_cmov(r, ri, flipped | flipped_i);
  fe_abs(r);
  return correct | flipped;
}

/* ============================ scalars mod l (4 x u64, little-endian) ============================ */
typedef uint64_t sc[4];
static const uint64_t SC_L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
static const uint64_t SC_C[2] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL}; /* l - 2^252 */

static void bn_mul(uint64_t *out, const uint64_t *a, int na, const uint64_t *b, int nb) {
  memset(out, 0, sizeof(uint64_t) * (size_t)(na + nb));
  for (int i = 0; i < na; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < nb; j++) {
      u128 t = (u128)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint64_t)t; carry = (uint64_t)(t >> 64);
    }
    out[i + nb] = carry;
  }
}
static uint64_t bn_add(uint64_t *out, const uint64_t *a, const uint64_t *b, int n) {
  uint64_t c = 0;
  for (int i = 0; i < n; i++) { u128 t = (u128)a[i] + b[i] + c; out[i] = (uint64_t)t; c = (uint64_t)(t >> 64); }
  return c;
}
static uint64_t bn_sub(uint64_t *out, const uint64_t *a, const uint64_t *b, int n) {
  uint64_t br = 0;
  for (int i = 0; i < n; i++) { u128 t = (u128)a[i] - b[i] - br; out[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
  return br;
}
static int bn_geq(const uint64_t *a, const uint64_t *b, int n) {
  for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
  return 1;
}
/* split x (n limbs) at bit 252: lo[4] = x mod 2^252, hi[n-3] = x >> 252 */
static void bn_split252(uint64_t lo[4], uint64_t *hi, const uint64_t *x, int n) {
  for (int i = 0; i < 4; i++) lo[i] = i < n ? x[i] : 0;
  lo[3] &= 0x0fffffffffffffffULL;
  for (int i = 0; i < n - 3; i++) {
    uint64_t a = x[i + 3] >> 60, b = (i + 4 < n) ? x[i + 4] << 4 : 0;
    hi[i] = a | b;
  }
}
/* x (8 limbs, < 2^512) mod l, by folding 2^252 = -c (mod l) three times */
static void sc_reduce512(sc out, const uint64_t x[8]) {
  uint64_t lo0[4], h0[5], y1[7], lo1[4], h1[4], y2[6], lo2[4], h2[3], y3[5];
  bn_split252(lo0, h0, x, 8);          /* h0 < 2^260 */
  bn_mul(y1, h0, 5, SC_C, 2);          /* < 2^385 (7 limbs) */
  bn_split252(lo1, h1, y1, 7);         /* h1 < 2^133 (3 limbs used, 4 written) */
  bn_mul(y2, h1, 4, SC_C, 2);          /* < 2^258 (6 limbs, top zero) */
  bn_split252(lo2, h2, y2, 6);         /* h2 < 2^6 */
  bn_mul(y3, h2, 3, SC_C, 2);          /* < 2^131 */
  /* x = lo0 - y1 = lo0 - lo1 + y2' = lo0 - lo1 + lo2 - y3  (mod l) */
  uint64_t A[4], B[4], two_l[4], r[4];
  bn_add(A, lo0, lo2, 4);              /* < 2^253 */
  bn_add(B, lo1, y3, 4);               /* < 2^252 + 2^131 */
  bn_add(two_l, SC_L, SC_L, 4);
  bn_add(r, A, two_l, 4);              /* < 2^255 */
  bn_sub(r, r, B, 4);                  /* > 0 */
  while (bn_geq(r, SC_L, 4)) bn_sub(r, r, SC_L, 4);
  memcpy(out, r, sizeof(sc));
}
static void sc_frombytes_raw(sc s, const uint8_t b[32]) {
  for (int i = 0; i < 4; i++) { s[i] = 0; for (int j = 7; j >= 0; j--) s[i] = (s[i] << 8) | b[8 * i + j]; }
}
static void sc_frombytes(sc s, const uint8_t b[32]) { /* from_bytes_mod_order */
  uint64_t x[8] = {0}; sc_frombytes_raw(x, b); sc_reduce512(s, x);
}
static void sc_from_wide(sc s, const uint8_t b[64]) { /* Scalar::from_bytes_mod_order_wide */
  uint64_t x[8];
  for (int i = 0; i < 8; i++) { x[i] = 0; for (int j = 7; j >= 0; j--) x[i] = (x[i] << 8) | b[8 * i + j]; }
  sc_reduce512(s, x);
}
static void sc_tobytes(uint8_t b[32], const sc s) { for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(s[i] >> (8 * j)); }
static void sc_copy(sc r, const sc a) { memcpy(r, a, sizeof(sc)); }
static void sc_add(sc r, const sc a, const sc b) { uint64_t t[4]; bn_add(t, a, b, 4); if (bn_geq(t, SC_L, 4)) bn_sub(t, t, SC_L, 4); memcpy(r, t, sizeof(sc)); }
static void sc_sub(sc r, const sc a, const sc b) { uint64_t t[4]; bn_add(t, a, SC_L, 4); bn_sub(t, t, b, 4); if (bn_geq(t, SC_L, 4)) bn_sub(t, t, SC_L, 4); memcpy(r, t, sizeof(sc)); }
static void sc_neg(sc r, const sc a) { sc z = {0, 0, 0, 0}; sc_sub(r, z, a); }
static void sc_mul(sc r, const sc a, const sc b) { uint64_t t[8]; bn_mul(t, a, 4, b, 4); sc_reduce512(r, t); }
static void sc_muladd(sc r, const sc a, const sc b, const sc c) { sc t; sc_mul(t, a, b); sc_add(r, t, c); }
static int sc_eq(const sc a, const sc b) { return memcmp(a, b, sizeof(sc)) == 0; }
static int sc_iszero(const sc a) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static void sc_invert(sc r, const sc a) { /* a^(l-2); 0 -> 0 */
  uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0}; bn_sub(e, SC_L, two, 4);
  sc acc = {1, 0, 0, 0};
  for (int i = 252; i >= 0; i--) {
    sc_mul(acc, acc, acc);
    if ((e[i / 64] >> (i % 64)) & 1) sc_mul(acc, acc, a);
  }
  sc_copy(r, acc);
}
static void sc_from_u64(sc r, uint64_t v) { r[0] = v; r[1] = r[2] = r[3] = 0; }
/* Scalar::from(2u128.pow(i)) for i < 128 */
static void sc_pow2(sc r, int i) { r[0] = r[1] = r[2] = r[3] = 0; r[i / 64] = 1ULL << (i % 64); }

/* ============================ Edwards points (extended, a = -1) ============================ */
typedef struct { fe X, Y, Z, T; } ge;
typedef struct { fe YpX, YmX, Z, T2d; } ge_cached;   /* "ProjectiveNiels" */
typedef struct { fe ypx, ymx, xy2d; } ge_niels;      /* "AffineNiels" */

static void ge_identity(ge *p) { fe_0(p->X); fe_1(p->Y); fe_1(p->Z); fe_0(p->T); }
static void ge_basepoint(ge *p) { fe_copy(p->X, FE_BX); fe_copy(p->Y, FE_BY); fe_1(p->Z); fe_copy(p->T, FE_BT); }
static void ge_to_cached(ge_cached *c, const ge *p) {
  fe_add(c->YpX, p->Y, p->X); fe_sub(c->YmX, p->Y, p->X); fe_copy(c->Z, p->Z); fe_mul(c->T2d, p->T, FE_D2);
}
static void ge_cached_identity(ge_cached *c) { fe_1(c->YpX); fe_1(c->YmX); fe_1(c->Z); fe_0(c->T2d); }
static void ge_niels_identity(ge_niels *c) { fe_1(c->ypx); fe_1(c->ymx); fe_0(c->xy2d); }
static void ge_cached_cmov(ge_cached *r, const ge_cached *a, int b) { fe_cmov(r->YpX, a->YpX, b); fe_cmov(r->YmX, a->YmX, b); fe_cmov(r->Z, a->Z, b); fe_cmov(r->T2d, a->T2d, b); }
static void ge_niels_cmov(ge_niels *r, const ge_niels *a, int b) { fe_cmov(r->ypx, a->ypx, b); fe_cmov(r->ymx, a->ymx, b); fe_cmov(r->xy2d, a->xy2d, b); }
static void ge_cached_cneg(ge_cached *r, int b) { fe t; fe_copy(t, r->YpX); fe_cmov(r->YpX, r->YmX, b); fe_cmov(r->YmX, t, b); fe_cneg(r->T2d, b); }
static void ge_niels_cneg(ge_niels *r, int b) { fe t; fe_copy(t, r->ypx); fe_cmov(r->ypx, r->ymx, b); fe_cmov(r->ymx, t, b); fe_cneg(r->xy2d, b); }

static void ge_from_completed(ge *r, const fe X, const fe Y, const fe Z, const fe T) {
  fe x, y, z, t; fe_copy(x, X); fe_copy(y, Y); fe_copy(z, Z); fe_copy(t, T);
  fe_mul(r->X, x, t); fe_mul(r->Y, y, z); fe_mul(r->Z, z, t); fe_mul(r->T, x, y);
}
static void ge_add_cached(ge *r, const ge *p, const ge_cached *q) {
  fe ypx, ymx, pp, mm, tt2d, zz, zz2, cx, cy, cz, ct;
  fe_add(ypx, p->Y, p->X); fe_sub(ymx, p->Y, p->X);
  fe_mul(pp, ypx, q->YpX); fe_mul(mm, ymx, q->YmX); fe_mul(tt2d, p->T, q->T2d); fe_mul(zz, p->Z, q->Z);
  fe_add(zz2, zz, zz);
  fe_sub(cx, pp, mm); fe_add(cy, pp, mm); fe_add(cz, zz2, tt2d); fe_sub(ct, zz2, tt2d);
  ge_from_completed(r, cx, cy, cz, ct);
}
static void ge_madd(ge *r, const ge *p, const ge_niels *q) {
  fe ypx, ymx, pp, mm, tt2d, zz2, cx, cy, cz, ct;
  fe_add(ypx, p->Y, p->X); fe_sub(ymx, p->Y, p->X);
  fe_mul(pp, ypx, q->ypx); fe_mul(mm, ymx, q->ymx); fe_mul(tt2d, p->T, q->xy2d);
  fe_add(zz2, p->Z, p->Z);
  fe_sub(cx, pp, mm); fe_add(cy, pp, mm); fe_add(cz, zz2, tt2d); fe_sub(ct, zz2, tt2d);
  ge_from_completed(r, cx, cy, cz, ct);
}
static void ge_add(ge *r, const ge *p, const ge *q) { ge_cached c; ge_to_cached(&c, q); ge_add_cached(r, p, &c); }
static void ge_neg(ge *r, const ge *p) { fe_neg(r->X, p->X); fe_copy(r->Y, p->Y); fe_copy(r->Z, p->Z); fe_neg(r->T, p->T); }
static void ge_sub(ge *r, const ge *p, const ge *q) { ge n; ge_neg(&n, q); ge_add(r, p, &n); }
static void ge_double(ge *r, const ge *p) {
  fe xx, yy, zz2, xpy, xpy2, yypxx, yymxx, cx, ct;
  fe_sq(xx, p->X); fe_sq(yy, p->Y); fe_sq(zz2, p->Z); fe_add(zz2, zz2, zz2);
  fe_add(xpy, p->X, p->Y); fe_sq(xpy2, xpy);
  fe_add(yypxx, yy, xx); fe_sub(yymxx, yy, xx);
  fe_sub(cx, xpy2, yypxx); fe_sub(ct, zz2, yymxx);
  ge_from_completed(r, cx, yypxx, yymxx, ct);
}

/* Scalar::as_radix_16: 64 signed digits in [-8, 8) (top digit <= 8) */
static void sc_radix16(int8_t e[64], const sc s) {
  uint8_t b[32]; sc_tobytes(b, s);
  for (int i = 0; i < 32; i++) { e[2 * i] = b[i] & 15; e[2 * i + 1] = (b[i] >> 4) & 15; }
  for (int i = 0; i < 63; i++) { int8_t c = (int8_t)((e[i] + 8) >> 4); e[i] -= (int8_t)(c << 4); e[i + 1] += c; }
}

/* `RistrettoPoint * Scalar`: constant-time radix-16 with an 8-entry table and a full-scan lookup */
static void ge_scalarmult(ge *r, const ge *p, const sc s) {
  ge_cached tab[8]; ge t; int8_t e[64];
  ge_to_cached(&tab[0], p);
  for (int i = 1; i < 8; i++) { ge_add_cached(&t, p, &tab[i - 1]); ge_to_cached(&tab[i], &t); }
  sc_radix16(e, s);
  ge q; ge_identity(&q);
  for (int i = 63; i >= 0; i--) {
    if (i != 63) { ge_double(&q, &q); ge_double(&q, &q); ge_double(&q, &q); ge_double(&q, &q); }
    int neg = e[i] < 0, a = neg ? -e[i] : e[i];
    ge_cached c; ge_cached_identity(&c);
    for (int j = 1; j <= 8; j++) ge_cached_cmov(&c, &tab[j - 1], a == j);
    ge_cached_cneg(&c, neg);
    ge_add_cached(&q, &q, &c);
  }
  *r = q;
}

/* RistrettoBasepointTable: 32 x 8 affine-Niels multiples j * 16^(2i) * B */
typedef struct { ge_niels t[32][8]; ge base; } ge_table;
static void ge_to_niels(ge_niels *n, const ge *p) {
  fe zi, x, y; fe_invert(zi, p->Z); fe_mul(x, p->X, zi); fe_mul(y, p->Y, zi);
  fe_add(n->ypx, y, x); fe_sub(n->ymx, y, x); fe_mul(n->xy2d, x, y); fe_mul(n->xy2d, n->xy2d, FE_D2);
}
static void ge_table_create(ge_table *tb, const ge *b) {
  ge p = *b; tb->base = *b;
  for (int i = 0; i < 32; i++) {
    ge q = p;
    for (int j = 0; j < 8; j++) { ge_to_niels(&tb->t[i][j], &q); ge_add(&q, &q, &p); }
    for (int k = 0; k < 8; k++) ge_double(&p, &p);   /* p *= 16^2 */
  }
}
static void ge_table_select(ge_niels *c, const ge_table *tb, int i, int8_t d) {
  int neg = d < 0, a = neg ? -d : d;
  ge_niels_identity(c);
  for (int j = 1; j <= 8; j++) ge_niels_cmov(c, &tb->t[i][j - 1], a == j);
  ge_niels_cneg(c, neg);
}
/* `&table * &scalar` */
static void ge_table_mul(ge *r, const ge_table *tb, const sc s) {
  int8_t e[64]; sc_radix16(e, s);
  ge p; ge_identity(&p); ge_niels c;
  for (int i = 1; i < 64; i += 2) { ge_table_select(&c, tb, i / 2, e[i]); ge_madd(&p, &p, &c); }
  ge_double(&p, &p); ge_double(&p, &p); ge_double(&p, &p); ge_double(&p, &p);
  for (int i = 0; i < 64; i += 2) { ge_table_select(&c, tb, i / 2, e[i]); ge_madd(&p, &p, &c); }
  *r = p;
}

/* ============================ ristretto255 (RFC 9496) ============================ */
static void ristretto_encode(uint8_t out[32], const ge *p) {
  fe u1, u2, t, inv, d1, d2, zinv, ix, iy, ench, x, y, den, zmy;
  fe_add(u1, p->Z, p->Y); fe_sub(t, p->Z, p->Y); fe_mul(u1, u1, t);
  fe_mul(u2, p->X, p->Y);
  fe_sq(t, u2); fe_mul(t, t, u1);
  fe one; fe_1(one);
  fe_sqrt_ratio_m1(inv, one, t);
  fe_mul(d1, inv, u1); fe_mul(d2, inv, u2);
  fe_mul(zinv, d1, d2); fe_mul(zinv, zinv, p->T);
  fe_mul(ix, p->X, FE_SQRT_M1); fe_mul(iy, p->Y, FE_SQRT_M1);
  fe_mul(ench, d1, FE_INVSQRT_A_MINUS_D);
  fe_mul(t, p->T, zinv);
  int rotate = fe_isneg(t);
  fe_copy(x, p->X); fe_copy(y, p->Y); fe_copy(den, d2);
  fe_cmov(x, iy, rotate); fe_cmov(y, ix, rotate); fe_cmov(den, ench, rotate);
  fe_mul(t, x, zinv);
  fe_cneg(y, fe_isneg(t));
  fe_sub(zmy, p->Z, y); fe_mul(t, den, zmy); fe_abs(t);
  fe_tobytes(out, t);
}
static int ristretto_decode(ge *p, const uint8_t in[32]) {
  fe s, ss, u1, u2, u2s, v, t, inv, dx, dy, one;
  uint8_t chk[32];
  fe_frombytes(s, in); fe_tobytes(chk, s);
  if (memcmp(chk, in, 32) != 0) return 0;        /* non-canonical (incl. bit 255 set) */
  if (in[0] & 1) return 0;                       /* negative */
  fe_1(one);
  fe_sq(ss, s); fe_sub(u1, one, ss); fe_add(u2, one, ss); fe_sq(u2s, u2);
  fe_sq(t, u1); fe_mul(t, t, FE_D); fe_neg(t, t); fe_sub(v, t, u2s);
  fe_mul(t, v, u2s);
  int was_square = fe_sqrt_ratio_m1(inv, one, t);
  fe_mul(dx, inv, u2); fe_mul(dy, inv, dx); fe_mul(dy, dy, v);
  fe_add(t, s, s); fe_mul(p->X, t, dx); fe_abs(p->X);
  fe_mul(p->Y, u1, dy); fe_1(p->Z); fe_mul(p->T, p->X, p->Y);
  if (!was_square || fe_isneg(p->T) || fe_iszero(p->Y)) return 0;
  return 1;
}
static void ristretto_map(ge *p, const fe t0) {
  fe r, u, v, c, rpd, s, sp, n, w0, w1, w2, w3, one, t;
  fe_1(one);
  fe_sq(r, t0); fe_mul(r, r, FE_SQRT_M1);
  fe_add(u, r, one); fe_mul(u, u, FE_ONE_MINUS_D_SQ);
  fe_mul(t, r, FE_D); fe_neg(c, one); fe_sub(t, c, t);        /* -1 - r*d */
  fe_add(rpd, r, FE_D); fe_mul(v, t, rpd);
  int was_square = fe_sqrt_ratio_m1(s, u, v);
  fe_mul(sp, s, t0); fe_abs(sp); fe_neg(sp, sp);
  fe_cmov(s, sp, !was_square);
  fe_cmov(c, r, !was_square);                                  /* c = -1 or r */
  fe_sub(t, r, one); fe_mul(n, c, t); fe_mul(n, n, FE_D_MINUS_ONE_SQ); fe_sub(n, n, v);
  fe_mul(w0, s, v); fe_add(w0, w0, w0);
  fe_mul(w1, n, FE_SQRT_AD_MINUS_ONE);
  fe_sq(t, s); fe_sub(w2, one, t); fe_add(w3, one, t);
  fe_mul(p->X, w0, w3); fe_mul(p->Y, w2, w1); fe_mul(p->Z, w1, w3); fe_mul(p->T, w0, w2);
}
static void ristretto_from_uniform(ge *p, const uint8_t b[64]) {
  fe r0, r1; ge p0, p1;
  fe_frombytes(r0, b); fe_frombytes(r1, b + 32);
  ristretto_map(&p0, r0); ristretto_map(&p1, r1); ge_add(p, &p0, &p1);
}
static int ristretto_eq(const ge *a, const ge *b) {
  fe l, r; int e1, e2;
  fe_mul(l, a->X, b->Y); fe_mul(r, a->Y, b->X); e1 = fe_eq(l, r);
  fe_mul(l, a->Y, b->Y); fe_mul(r, a->X, b->X); e2 = fe_eq(l, r);
  return e1 | e2;
}

/* ============================ BLAKE3 (hash mode, XOF output) ============================ */
static const uint32_t B3_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };
static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define B3_G(a, b, c, d, mx, my) do { \
  v[a] = v[a] + v[b] + (mx); v[d] = rotr32(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr32(v[b] ^ v[c], 12); \
  v[a] = v[a] + v[b] + (my); v[d] = rotr32(v[d] ^ v[a], 8);  v[c] = v[c] + v[d]; v[b] = rotr32(v[b] ^ v[c], 7); } while (0)
static void b3_compress(uint32_t out[16], const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t blen, uint32_t flags) {
  uint32_t v[16], m[16], t[16];
  memcpy(v, cv, 32); memcpy(v + 8, B3_IV, 16);
  v[12] = (uint32_t)counter; v[13] = (uint32_t)(counter >> 32); v[14] = blen; v[15] = flags;
  memcpy(m, block, 64);
  for (int r = 0; r < 7; r++) {
    B3_G(0, 4, 8, 12, m[0], m[1]); B3_G(1, 5, 9, 13, m[2], m[3]); B3_G(2, 6, 10, 14, m[4], m[5]); B3_G(3, 7, 11, 15, m[6], m[7]);
    B3_G(0, 5, 10, 15, m[8], m[9]); B3_G(1, 6, 11, 12, m[10], m[11]); B3_G(2, 7, 8, 13, m[12], m[13]); B3_G(3, 4, 9, 14, m[14], m[15]);
    for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
    memcpy(m, t, 64);
  }
  for (int i = 0; i < 8; i++) { out[i] = v[i] ^ v[i + 8]; out[i + 8] = v[i + 8] ^ cv[i]; }
}
typedef struct { uint32_t cv[8]; uint32_t block[16]; uint64_t counter; uint32_t blen; uint32_t flags; } b3_output;
static void b3_words(uint32_t w[16], const uint8_t *p, size_t n) {
  uint8_t buf[64]; memset(buf, 0, 64); memcpy(buf, p, n);
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)buf[4 * i] | (uint32_t)buf[4 * i + 1] << 8 | (uint32_t)buf[4 * i + 2] << 16 | (uint32_t)buf[4 * i + 3] << 24;
}
static void b3_chunk(b3_output *o, const uint8_t *p, size_t n, uint64_t counter) {
  uint32_t cv[8], out[16]; memcpy(cv, B3_IV, 32);
  size_t nblocks = n ? (n + 63) / 64 : 1;
  for (size_t i = 0; i < nblocks; i++) {
    size_t bl = (i == nblocks - 1) ? n - 64 * i : 64;
    uint32_t fl = (i == 0 ? B3_CHUNK_START : 0);
    uint32_t w[16]; b3_words(w, p + 64 * i, bl);
    if (i == nblocks - 1) { memcpy(o->cv, cv, 32); memcpy(o->block, w, 64); o->counter = counter; o->blen = (uint32_t)bl; o->flags = fl | B3_CHUNK_END; return; }
    b3_compress(out, cv, w, counter, 64, fl); memcpy(cv, out, 32);
  }
}
static void b3_subtree(b3_output *o, const uint8_t *p, size_t n, uint64_t first_chunk) {
  if (n <= 1024) { b3_chunk(o, p, n, first_chunk); return; }
  size_t nchunks = (n + 1023) / 1024, left = 1;
  while (left * 2 < nchunks) left *= 2;
  b3_output l, r; uint32_t out[16];
  b3_subtree(&l, p, left * 1024, first_chunk);
  b3_subtree(&r, p + left * 1024, n - left * 1024, first_chunk + left);
  memcpy(o->cv, B3_IV, 32);
  b3_compress(out, l.cv, l.block, l.counter, l.blen, l.flags); memcpy(o->block, out, 32);
  b3_compress(out, r.cv, r.block, r.counter, r.blen, r.flags); memcpy(o->block + 8, out, 32);
  o->counter = 0; o->blen = 64; o->flags = B3_PARENT;
}
void oracle_blake3(const uint8_t *in, size_t len, uint8_t *out, size_t outlen) {
  b3_output o; b3_subtree(&o, in, len, 0);
  uint64_t blk = 0; size_t done = 0;
  while (done < outlen) {
    uint32_t v[16]; b3_compress(v, o.cv, o.block, blk++, o.blen, o.flags | B3_ROOT);
    for (int i = 0; i < 64 && done < outlen; i++, done++) out[done] = (uint8_t)(v[i / 4] >> (8 * (i % 4)));
  }
}

/* ============================ context: Params + tables ============================ */
struct oracle_ctx {
  int L;
  ge h1, h2, h3;
  ge_table t1, t2, t3;
  uint8_t henc[96];
};

static void put_be64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i)); }

/* Params::new + hash_to_ristretto, src/lib.rs:291-354 */
void oracle_params_new(const char *org, const char *svc, const char *dep, const char *ver, uint8_t out[96]) {
  size_t cap = 16 + strlen(org) + strlen(svc) + strlen(dep) + strlen(ver);
  char *ds = (char *)malloc(cap);
  strcpy(ds, "ACT-v1:"); strcat(ds, org); strcat(ds, ":"); strcat(ds, svc); strcat(ds, ":"); strcat(ds, dep); strcat(ds, ":"); strcat(ds, ver);  /* :293-296 */
  size_t dl = strlen(ds);
  uint8_t *msg = (uint8_t *)malloc(dl + 8 + 8 + 32 + 8 + 4);
  uint8_t seed[32];
  put_be64(msg, dl); memcpy(msg + 8, ds, dl);
  oracle_blake3(msg, 8 + dl, seed, 32);                                     /* :299-303 */
  for (uint32_t ctr = 0; ctr < 3; ctr++) {                                  /* :306-308 */
    size_t o = 8 + dl;
    put_be64(msg + o, 32); memcpy(msg + o + 8, seed, 32); o += 40;         /* :341-342 */
    put_be64(msg + o, 4); o += 8;                                           /* :345 */
    msg[o] = (uint8_t)ctr; msg[o + 1] = msg[o + 2] = msg[o + 3] = 0; o += 4; /* :346 counter.to_le_bytes() */
    uint8_t u[64]; oracle_blake3(msg, o, u, 64);                            /* :349-351 */
    ge p; ristretto_from_uniform(&p, u);                                    /* :353 */
    ristretto_encode(out + 32 * ctr, &p);
  }
  free(msg); free(ds);
}

oracle_ctx *oracle_ctx_new(const uint8_t h[96], int L) {
  if (L < 1 || L > ORACLE_L_MAX) return NULL;
  oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof(*c));
  c->L = L; memcpy(c->henc, h, 96);
  if (!ristretto_decode(&c->h1, h) || !ristretto_decode(&c->h2, h + 32) || !ristretto_decode(&c->h3, h + 64)) { free(c); return NULL; }
  ge_table_create(&c->t1, &c->h1); ge_table_create(&c->t2, &c->h2); ge_table_create(&c->t3, &c->h3);  /* :311-313 */
  return c;
}
void oracle_ctx_free(oracle_ctx *c) { free(c); }
size_t oracle_spend_proof_bytes(const oracle_ctx *c) { return 32u * (14u + 4u * (size_t)c->L); }
size_t oracle_prove_rng_bytes(const oracle_ctx *c) { return 64u * (4u * (size_t)c->L + 12u); }
size_t oracle_spend_transcript_bytes(const oracle_ctx *c) { return 8 + 43 + 3 * 40 + 8 + 5 + 40 * (6 + 3 * (size_t)c->L); }

/* ============================ Transcript (src/transcript.rs) ============================ */
typedef struct { uint8_t *buf; size_t len; } transcript;
static const char PROTOCOL_VERSION[] = "curve25519-ristretto anonymous-credits v1.0";  /* src/transcript.rs:29 */
static void tr_update(transcript *t, const uint8_t *b, size_t n) { put_be64(t->buf + t->len, n); memcpy(t->buf + t->len + 8, b, n); t->len += 8 + n; }  /* :95-98 */
static void tr_new(transcript *t, uint8_t *buf, const oracle_ctx *c, const char *label) {   /* :54-74 */
  t->buf = buf; t->len = 0;
  tr_update(t, (const uint8_t *)PROTOCOL_VERSION, sizeof(PROTOCOL_VERSION) - 1);
  tr_update(t, c->henc, 32); tr_update(t, c->henc + 32, 32); tr_update(t, c->henc + 64, 32);   /* compress(basepoint) == canonical input bytes */
  tr_update(t, (const uint8_t *)label, strlen(label));
}
static void tr_add_element(transcript *t, const ge *p) { uint8_t e[32]; ristretto_encode(e, p); tr_update(t, e, 32); }   /* :105-107 */
static void tr_add_scalar(transcript *t, const sc s) { uint8_t e[32]; sc_tobytes(e, s); tr_update(t, e, 32); }          /* :125-128 */
static void tr_challenge(sc out, const transcript *t) { uint8_t x[64]; oracle_blake3(t->buf, t->len, x, 64); sc_from_wide(out, x); }  /* :149-154 */

/* ============================ exported primitives ============================ */
void oracle_sc_reduce_wide(const uint8_t in[64], uint8_t out[32]) { sc s; sc_from_wide(s, in); sc_tobytes(out, s); }
void oracle_sc_muladd(const uint8_t a[32], const uint8_t b[32], const uint8_t c[32], uint8_t out[32]) {
  sc x, y, z, r; sc_frombytes(x, a); sc_frombytes(y, b); sc_frombytes(z, c); sc_muladd(r, x, y, z); sc_tobytes(out, r);
}
void oracle_sc_invert(const uint8_t a[32], uint8_t out[32]) { sc x, r; sc_frombytes(x, a); sc_invert(r, x); sc_tobytes(out, r); }
int oracle_ristretto_decode_encode(const uint8_t in[32], uint8_t out[32]) { ge p; if (!ristretto_decode(&p, in)) { memset(out, 0, 32); return 0; } ristretto_encode(out, &p); return 1; }
void oracle_ristretto_from_uniform(const uint8_t in[64], uint8_t out[32]) { ge p; ristretto_from_uniform(&p, in); ristretto_encode(out, &p); }
int oracle_ristretto_mul(const uint8_t pt[32], const uint8_t s[32], uint8_t out[32]) {
  ge p, r; sc k; if (!ristretto_decode(&p, pt)) return 0; sc_frombytes(k, s); ge_scalarmult(&r, &p, k); ristretto_encode(out, &r); return 1;
}
void oracle_ristretto_mul_base(const uint8_t s[32], uint8_t out[32]) { ge b, r; sc k; ge_basepoint(&b); sc_frombytes(k, s); ge_scalarmult(&r, &b, k); ristretto_encode(out, &r); }
int oracle_ristretto_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
  ge p, q, r; if (!ristretto_decode(&p, a) || !ristretto_decode(&q, b)) return 0; ge_add(&r, &p, &q); ristretto_encode(out, &r); return 1;
}

/* ============================ protocol ============================ */
static void rng_scalar(sc s, const uint8_t **rng) { sc_from_wide(s, *rng); *rng += 64; }   /* Scalar::random = 64-byte fill + wide reduce */

void oracle_private_key_random(const uint8_t rng[64], uint8_t sk[64]) {   /* src/lib.rs:188-194 */
  sc x; ge g, w; sc_from_wide(x, rng); ge_basepoint(&g); ge_scalarmult(&w, &g, x);
  sc_tobytes(sk, x); ristretto_encode(sk + 32, &w);
}
void oracle_pre_issuance_random(const uint8_t rng[128], uint8_t pre[64]) {   /* src/lib.rs:432-437: r then k */
  sc r, k; sc_from_wide(r, rng); sc_from_wide(k, rng + 64); sc_tobytes(pre, r); sc_tobytes(pre + 32, k);
}

/* PreIssuance::request, src/lib.rs:463-487 */
void oracle_request(const oracle_ctx *c, const uint8_t pre[64], const uint8_t rng[128], uint8_t out[128]) {
  sc r, k, kp, rp, gamma, kbar, rbar; ge a, b, big_k, k1; uint8_t buf[512]; transcript t;
  sc_frombytes(r, pre); sc_frombytes(k, pre + 32);
  ge_table_mul(&a, &c->t2, k); ge_table_mul(&b, &c->t3, r); ge_add(&big_k, &a, &b);            /* :465 */
  rng_scalar(kp, &rng); rng_scalar(rp, &rng);                                                    /* :468-469 */
  ge_table_mul(&a, &c->t2, kp); ge_table_mul(&b, &c->t3, rp); ge_add(&k1, &a, &b);              /* :470 */
  tr_new(&t, buf, c, "request"); tr_add_element(&t, &big_k); tr_add_element(&t, &k1); tr_challenge(gamma, &t);  /* :473-475 */
  sc_muladd(kbar, k, gamma, kp); sc_muladd(rbar, r, gamma, rp);                                  /* :478-479 */
  ristretto_encode(out, &big_k); sc_tobytes(out + 32, gamma); sc_tobytes(out + 64, kbar); sc_tobytes(out + 96, rbar);
}

/* shared tail of issue / refund: BBS sign X_A with a DLEQ proof (src/lib.rs:643-660, :846-861) */
static void sign_tail(const oracle_ctx *c, const sc x, const ge *w, const ge *x_a, const uint8_t rng[128], const char *label,
                      const sc *c_amount /* nullable */, ge *a_out, sc e_out, sc gamma_out, sc z_out) {
  sc e, ex, exinv, alpha, gamma, xe; ge g, a, ge_e, x_g, y_a, y_g; uint8_t buf[640]; transcript t;
  ge_basepoint(&g);
  rng_scalar(e, &rng);                                                   /* :643 / :846 */
  sc_add(ex, e, x); sc_invert(exinv, ex); ge_scalarmult(&a, x_a, exinv); /* :645 / :849 */
  ge_scalarmult(&ge_e, &g, e); ge_add(&x_g, &ge_e, w);                   /* :646 / :851 */
  rng_scalar(alpha, &rng);                                               /* :649 / :852 */
  ge_scalarmult(&y_a, &a, alpha); ge_scalarmult(&y_g, &g, alpha);        /* :650-651 / :853-854 */
  tr_new(&t, buf, c, label);
  if (c_amount) tr_add_scalar(&t, *c_amount);                            /* :655 */
  tr_add_scalar(&t, e);
  tr_add_element(&t, &a); tr_add_element(&t, x_a); tr_add_element(&t, &x_g); tr_add_element(&t, &y_a); tr_add_element(&t, &y_g);
  tr_challenge(gamma, &t);                                               /* :654-657 / :856-859 */
  sc_add(xe, x, e); sc_muladd(z_out, gamma, xe, alpha);                  /* :660 / :861 */
  *a_out = a; sc_copy(e_out, e); sc_copy(gamma_out, gamma);
}

/* PrivateKey::issue, src/lib.rs:621-663 */
int oracle_issue(const oracle_ctx *c, const uint8_t sk[64], const uint8_t req[128], const uint8_t camt[32],
                 const uint8_t rng[128], uint8_t out[160]) {
  sc x, gamma_in, kbar, rbar, gamma, cs, e, g2, z; ge w, big_k, a, b, t0, k1, g, x_a, asig; uint8_t buf[512]; transcript t;
  memset(out, 0, 160);
  sc_frombytes(x, sk);
  if (!ristretto_decode(&w, sk + 32) || !ristretto_decode(&big_k, req)) return ORACLE_ERR_UNDECODABLE;
  sc_frombytes(gamma_in, req + 32); sc_frombytes(kbar, req + 64); sc_frombytes(rbar, req + 96); sc_frombytes(cs, camt);
  ge_table_mul(&a, &c->t2, kbar); ge_table_mul(&b, &c->t3, rbar); ge_add(&t0, &a, &b);
  ge_scalarmult(&a, &big_k, gamma_in); ge_sub(&k1, &t0, &a);                                     /* :629-630 */
  tr_new(&t, buf, c, "request"); tr_add_element(&t, &big_k); tr_add_element(&t, &k1); tr_challenge(gamma, &t);   /* :633-635 */
  if (!sc_eq(gamma, gamma_in)) return ORACLE_ERR_INVALID_ISSUANCE_REQUEST_PROOF;                 /* :638-640 */
  ge_basepoint(&g); ge_table_mul(&a, &c->t1, cs); ge_add(&x_a, &g, &a); ge_add(&x_a, &x_a, &big_k);   /* :644 */
  sign_tail(c, x, &w, &x_a, rng, "respond", &cs, &asig, e, g2, z);
  ristretto_encode(out, &asig); sc_tobytes(out + 32, e); sc_tobytes(out + 64, g2); sc_tobytes(out + 96, z); sc_tobytes(out + 128, cs);
  return ORACLE_OK;
}

/* shared DLEQ check of the client verifiers (src/lib.rs:536-552, :1232-1243) */
static int dleq_check(const oracle_ctx *c, const ge *w, const ge *x_a, const ge *a, const sc e, const sc gamma_in, const sc z,
                      const char *label, const sc *c_amount) {
  sc ng, gamma; ge g, t0, x_g, y_a, y_g, t1; uint8_t buf[640]; transcript t;
  ge_basepoint(&g);
  ge_scalarmult(&t0, &g, e); ge_add(&x_g, &t0, w);                                   /* :537 / :1232 */
  sc_neg(ng, gamma_in);
  ge_scalarmult(&t0, a, z); ge_scalarmult(&t1, x_a, ng); ge_add(&y_a, &t0, &t1);     /* :540 / :1233 */
  ge_scalarmult(&t0, &g, z); ge_scalarmult(&t1, &x_g, ng); ge_add(&y_g, &t0, &t1);   /* :541 / :1234 */
  tr_new(&t, buf, c, label);
  if (c_amount) tr_add_scalar(&t, *c_amount);
  tr_add_scalar(&t, e);
  tr_add_element(&t, a); tr_add_element(&t, x_a); tr_add_element(&t, &x_g); tr_add_element(&t, &y_a); tr_add_element(&t, &y_g);
  tr_challenge(gamma, &t);
  return sc_eq(gamma, gamma_in);
}

/* PreIssuance::to_credit_token, src/lib.rs:528-562 */
int oracle_issuance_to_credit_token(const oracle_ctx *c, const uint8_t pre[64], const uint8_t wenc[32], const uint8_t req[128],
                                    const uint8_t resp[160], uint8_t out[160]) {
  ge w, big_k, a, g, t0, x_a; sc e, gamma, z, cs;
  memset(out, 0, 160);
  if (!ristretto_decode(&w, wenc) || !ristretto_decode(&big_k, req) || !ristretto_decode(&a, resp)) return ORACLE_ERR_UNDECODABLE;
  sc_frombytes(e, resp + 32); sc_frombytes(gamma, resp + 64); sc_frombytes(z, resp + 96); sc_frombytes(cs, resp + 128);
  ge_basepoint(&g); ge_table_mul(&t0, &c->t1, cs); ge_add(&x_a, &g, &t0); ge_add(&x_a, &x_a, &big_k);   /* :536 */
  if (!dleq_check(c, &w, &x_a, &a, e, gamma, z, "respond", &cs)) return ORACLE_ERR_INVALID_ISSUANCE_RESPONSE_PROOF;   /* :550-552 */
  /* CreditToken{a,e,k,r,c}: record order a,e,k,r,c; PreIssuance record is r|k */
  memcpy(out, resp, 32); sc_tobytes(out + 32, e); memcpy(out + 64, pre + 32, 32); memcpy(out + 96, pre, 32); sc_tobytes(out + 128, cs);
  return ORACLE_OK;
}

/* SpendProof record field offsets (units of 32 bytes), SURVEY.md Appendix C */
#define PF_K 0
#define PF_S 1
#define PF_APRIME 2
#define PF_BBAR 3
#define PF_COM 4
#define PF_GAMMA(L) (4 + (L))
#define PF_EBAR(L) (5 + (L))
#define PF_R2BAR(L) (6 + (L))
#define PF_R3BAR(L) (7 + (L))
#define PF_CBAR(L) (8 + (L))
#define PF_RBAR(L) (9 + (L))
#define PF_W00(L) (10 + (L))
#define PF_W01(L) (11 + (L))
#define PF_GAMMA0(L) (12 + (L))
#define PF_Z(L) (12 + 2 * (L))
#define PF_KBAR(L) (12 + 4 * (L))
#define PF_SBAR(L) (13 + 4 * (L))

/* CreditToken::prove_spend, src/lib.rs:972-1152 */
int oracle_prove_spend(const oracle_ctx *c, const uint8_t tok[160], const uint8_t senc[32], const uint8_t *rng,
                       uint8_t *out, uint8_t out_pre[96]) {
  const int L = c->L;
  sc e, k, r, cc, s, r1, r2, cprime, rprime, eprime, r2prime, r3prime, r1r2, r3, m, kstar, k0prime, w0, rstar, kprime, sprime, gamma, ng, t1;
  static _Thread_local sc s_i[ORACLE_L_MAX], s_ip[ORACLE_L_MAX], gamma_i[ORACLE_L_MAX], zr[ORACLE_L_MAX];
  static _Thread_local ge com[ORACLE_L_MAX], cp[ORACLE_L_MAX][2];
  static _Thread_local uint8_t buf[8 + 43 + 3 * 40 + 8 + 5 + 40 * (6 + 3 * ORACLE_L_MAX)];
  uint8_t mb[32]; int bit[ORACLE_L_MAX];
  ge a, g, b, t0, t2, aprime, bbar, a1, a2, cfinal; transcript t;
  memset(out, 0, oracle_spend_proof_bytes(c)); memset(out_pre, 0, 96);
  if (!ristretto_decode(&a, tok)) return ORACLE_ERR_UNDECODABLE;
  sc_frombytes(e, tok + 32); sc_frombytes(k, tok + 64); sc_frombytes(r, tok + 96); sc_frombytes(cc, tok + 128); sc_frombytes(s, senc);
  rng_scalar(r1, &rng); rng_scalar(r2, &rng); rng_scalar(cprime, &rng); rng_scalar(rprime, &rng);        /* :978-981 */
  rng_scalar(eprime, &rng); rng_scalar(r2prime, &rng); rng_scalar(r3prime, &rng);                          /* :982-984 */
  ge_basepoint(&g);
  ge_table_mul(&t0, &c->t1, cc); ge_add(&b, &g, &t0); ge_table_mul(&t0, &c->t2, k); ge_add(&b, &b, &t0);
  ge_table_mul(&t0, &c->t3, r); ge_add(&b, &b, &t0);                                                       /* :986-989 */
  sc_mul(r1r2, r1, r2); ge_scalarmult(&aprime, &a, r1r2);                                                  /* :990 */
  ge_scalarmult(&bbar, &b, r1);                                                                            /* :991 */
  sc_invert(r3, r1);                                                                                       /* :992 */
  ge_scalarmult(&t0, &aprime, eprime); ge_scalarmult(&t2, &bbar, r2prime); ge_add(&a1, &t0, &t2);          /* :993 */
  ge_scalarmult(&t0, &bbar, r3prime); ge_table_mul(&t2, &c->t1, cprime); ge_add(&a2, &t0, &t2);
  ge_table_mul(&t2, &c->t3, rprime); ge_add(&a2, &a2, &t2);                                                /* :994 */
  sc_sub(m, cc, s); sc_tobytes(mb, m);
  for (int j = 0; j < L; j++) bit[j] = (mb[j / 8] >> (j % 8)) & 1;                                         /* :996, bits_of :902-915 */
  rng_scalar(kstar, &rng);                                                                                 /* :998 */
  for (int j = 0; j < L; j++) rng_scalar(s_i[j], &rng);                                                    /* :999 */
  for (int j = 0; j < L; j++) {                                                                            /* :1001-1004 */
    sc ib; sc_from_u64(ib, (uint64_t)bit[j]);
    ge_table_mul(&com[j], &c->t1, ib);
    if (j == 0) { ge_table_mul(&t0, &c->t2, kstar); ge_add(&com[j], &com[j], &t0); }
    ge_table_mul(&t0, &c->t3, s_i[j]); ge_add(&com[j], &com[j], &t0);
  }
  rng_scalar(k0prime, &rng);                                                                               /* :1010 */
  for (int j = 0; j < L; j++) rng_scalar(s_ip[j], &rng);                                                   /* :1012-1014 */
  for (int j = 0; j < L; j++) rng_scalar(gamma_i[j], &rng);                                                /* :1016-1018 */
  rng_scalar(w0, &rng);                                                                                    /* :1019 */
  for (int j = 0; j < L; j++) rng_scalar(zr[j], &rng);                                                     /* :1021-1023 */
  for (int j = 0; j < L; j++) {                                                                            /* :1025-1051: both branches always computed */
    ge cj1, real, sim0, sim1, hz;
    ge_sub(&cj1, &com[j], &c->h1);                                                                         /* :1008, :1039 */
    ge_table_mul(&hz, &c->t3, zr[j]);
    if (j == 0) { ge_table_mul(&t0, &c->t2, w0); ge_add(&hz, &t0, &hz); }
    ge_scalarmult(&t0, &com[j], gamma_i[j]); ge_sub(&sim0, &hz, &t0);
    ge_scalarmult(&t0, &cj1, gamma_i[j]); ge_sub(&sim1, &hz, &t0);
    ge_table_mul(&real, &c->t3, s_ip[j]);
    if (j == 0) { ge_table_mul(&t0, &c->t2, k0prime); ge_add(&real, &t0, &real); }
    /* conditional_select(a, b, choice) yields b when choice (= bit is zero) is set */
    cp[j][0] = bit[j] == 0 ? real : sim0;                                                                  /* :1025-1029, :1041-1045 */
    cp[j][1] = bit[j] == 0 ? sim1 : real;                                                                  /* :1031-1035, :1046-1050 */
  }
  rstar[0] = rstar[1] = rstar[2] = rstar[3] = 0;
  for (int j = 0; j < L; j++) { sc p2; sc_pow2(p2, j); sc_muladd(rstar, s_i[j], p2, rstar); }              /* :1052-1056 */
  rng_scalar(kprime, &rng); rng_scalar(sprime, &rng);                                                      /* :1057-1058 */
  sc_neg(t1, cprime);
  ge_table_mul(&cfinal, &c->t1, t1); ge_table_mul(&t0, &c->t2, kprime); ge_add(&cfinal, &cfinal, &t0);
  ge_table_mul(&t0, &c->t3, sprime); ge_add(&cfinal, &cfinal, &t0);                                        /* :1059 */
  tr_new(&t, buf, c, "spend");                                                                             /* :1061-1070 */
  tr_add_scalar(&t, k); tr_add_element(&t, &aprime); tr_add_element(&t, &bbar); tr_add_element(&t, &a1); tr_add_element(&t, &a2);
  for (int j = 0; j < L; j++) tr_add_element(&t, &com[j]);
  for (int j = 0; j < L; j++) { tr_add_element(&t, &cp[j][0]); tr_add_element(&t, &cp[j][1]); }
  tr_add_element(&t, &cfinal);
  tr_challenge(gamma, &t);
  sc_neg(ng, gamma);
  uint8_t *f = out;
  sc_tobytes(f + 32 * PF_K, k); sc_tobytes(f + 32 * PF_S, s);
  ristretto_encode(f + 32 * PF_APRIME, &aprime); ristretto_encode(f + 32 * PF_BBAR, &bbar);
  for (int j = 0; j < L; j++) ristretto_encode(f + 32 * (PF_COM + j), &com[j]);
  sc_tobytes(f + 32 * PF_GAMMA(L), gamma);
  sc_muladd(t1, ng, e, eprime); sc_tobytes(f + 32 * PF_EBAR(L), t1);                                       /* :1072 */
  sc_muladd(t1, gamma, r2, r2prime); sc_tobytes(f + 32 * PF_R2BAR(L), t1);                                 /* :1073 */
  sc_muladd(t1, gamma, r3, r3prime); sc_tobytes(f + 32 * PF_R3BAR(L), t1);                                 /* :1074 */
  sc_muladd(t1, ng, cc, cprime); sc_tobytes(f + 32 * PF_CBAR(L), t1);                                      /* :1075 */
  sc_muladd(t1, ng, r, rprime); sc_tobytes(f + 32 * PF_RBAR(L), t1);                                       /* :1076 */
  for (int j = 0; j < L; j++) {
    sc g00, g01, z0, z1;
    if (bit[j] == 0) sc_sub(g00, gamma, gamma_i[j]); else sc_copy(g00, gamma_i[j]);                        /* :1078-1082, :1105-1109 */
    sc_sub(g01, gamma, g00);
    if (bit[j] == 0) { sc_muladd(z0, g00, s_i[j], s_ip[j]); sc_copy(z1, zr[j]); }                          /* :1094-1103, :1110-1119 */
    else { sc_copy(z0, zr[j]); sc_muladd(z1, g01, s_i[j], s_ip[j]); }
    sc_tobytes(f + 32 * (PF_GAMMA0(L) + j), g00);
    sc_tobytes(f + 32 * (PF_Z(L) + 2 * j), z0); sc_tobytes(f + 32 * (PF_Z(L) + 2 * j + 1), z1);
    if (j == 0) {
      sc w00, w01;
      if (bit[0] == 0) { sc_muladd(w00, g00, kstar, k0prime); sc_copy(w01, w0); }                          /* :1083-1092 */
      else { sc_copy(w00, w0); sc_muladd(w01, g01, kstar, k0prime); }
      sc_tobytes(f + 32 * PF_W00(L), w00); sc_tobytes(f + 32 * PF_W01(L), w01);
    }
  }
  sc_muladd(t1, gamma, kstar, kprime); sc_tobytes(f + 32 * PF_KBAR(L), t1);                                /* :1121 */
  sc_muladd(t1, gamma, rstar, sprime); sc_tobytes(f + 32 * PF_SBAR(L), t1);                                /* :1122 */
  sc_tobytes(out_pre, rstar); sc_tobytes(out_pre + 32, kstar); sc_tobytes(out_pre + 64, m);                /* :1124-1128, record r|k|m */
  return ORACLE_OK;
}

/* src/lib.rs:787-844 */
static int verify_spend_core(const oracle_ctx *c, const sc x, const uint8_t *f, ge *kprime_out, uint8_t *tr_out) {
  const int L = c->L;
  static _Thread_local ge com[ORACLE_L_MAX];
  static _Thread_local uint8_t buf[8 + 43 + 3 * 40 + 8 + 5 + 40 * (6 + 3 * ORACLE_L_MAX)];
  sc k, s, gamma, ebar, r2bar, r3bar, cbar, rbar, w00, w01, kbar, sbar, ng, gchk, t1;
  ge aprime, bbar, idp, abar, g, bigh1, a1, a2, t0, kprime, com_, bigc; transcript t;
  if (!ristretto_decode(&aprime, f + 32 * PF_APRIME) || !ristretto_decode(&bbar, f + 32 * PF_BBAR)) return ORACLE_ERR_UNDECODABLE;
  for (int j = 0; j < L; j++) if (!ristretto_decode(&com[j], f + 32 * (PF_COM + j))) return ORACLE_ERR_UNDECODABLE;
  sc_frombytes(k, f + 32 * PF_K); sc_frombytes(s, f + 32 * PF_S); sc_frombytes(gamma, f + 32 * PF_GAMMA(L));
  sc_frombytes(ebar, f + 32 * PF_EBAR(L)); sc_frombytes(r2bar, f + 32 * PF_R2BAR(L)); sc_frombytes(r3bar, f + 32 * PF_R3BAR(L));
  sc_frombytes(cbar, f + 32 * PF_CBAR(L)); sc_frombytes(rbar, f + 32 * PF_RBAR(L)); sc_frombytes(w00, f + 32 * PF_W00(L));
  sc_frombytes(w01, f + 32 * PF_W01(L)); sc_frombytes(kbar, f + 32 * PF_KBAR(L)); sc_frombytes(sbar, f + 32 * PF_SBAR(L));
  ge_identity(&idp);
  if (ristretto_eq(&aprime, &idp)) return ORACLE_ERR_IDENTITY_POINT;                                        /* :787-789 */
  sc_neg(ng, gamma); ge_basepoint(&g);
  ge_scalarmult(&abar, &aprime, x);                                                                         /* :791 */
  ge_table_mul(&t0, &c->t2, k); ge_add(&bigh1, &g, &t0);                                                    /* :792 */
  ge_scalarmult(&a1, &aprime, ebar); ge_scalarmult(&t0, &bbar, r2bar); ge_add(&a1, &a1, &t0);
  ge_scalarmult(&t0, &abar, ng); ge_add(&a1, &a1, &t0);                                                     /* :793-795 */
  ge_scalarmult(&a2, &bbar, r3bar); ge_table_mul(&t0, &c->t1, cbar); ge_add(&a2, &a2, &t0);
  ge_table_mul(&t0, &c->t3, rbar); ge_add(&a2, &a2, &t0); ge_scalarmult(&t0, &bigh1, ng); ge_add(&a2, &a2, &t0);   /* :796-799 */
  tr_new(&t, buf, c, "spend");                                                                              /* :831 */
  tr_add_scalar(&t, k); tr_update(&t, f + 32 * PF_APRIME, 32); tr_update(&t, f + 32 * PF_BBAR, 32);         /* compress(decoded) == canonical wire bytes */
  tr_add_element(&t, &a1); tr_add_element(&t, &a2);
  for (int j = 0; j < L; j++) tr_update(&t, f + 32 * (PF_COM + j), 32);
  for (int j = 0; j < L; j++) {                                                                             /* :800-817 */
    sc g0, g1, z0, z1; ge cj1, p0, p1;
    sc_frombytes(g0, f + 32 * (PF_GAMMA0(L) + j)); sc_sub(g1, gamma, g0);
    sc_frombytes(z0, f + 32 * (PF_Z(L) + 2 * j)); sc_frombytes(z1, f + 32 * (PF_Z(L) + 2 * j + 1));
    ge_sub(&cj1, &com[j], &c->h1);
    ge_table_mul(&p0, &c->t3, z0); ge_table_mul(&p1, &c->t3, z1);
    if (j == 0) { ge_table_mul(&t0, &c->t2, w00); ge_add(&p0, &t0, &p0); ge_table_mul(&t0, &c->t2, w01); ge_add(&p1, &t0, &p1); }
    ge_scalarmult(&t0, &com[j], g0); ge_sub(&p0, &p0, &t0);
    ge_scalarmult(&t0, &cj1, g1); ge_sub(&p1, &p1, &t0);
    tr_add_element(&t, &p0); tr_add_element(&t, &p1);                                                       /* :836-838 */
  }
  ge_identity(&kprime);
  for (int j = 0; j < L; j++) { sc p2; sc_pow2(p2, j); ge_scalarmult(&t0, &com[j], p2); ge_add(&kprime, &kprime, &t0); }   /* :819-824 */
  ge_table_mul(&t0, &c->t1, s); ge_add(&com_, &t0, &kprime);                                                /* :825 */
  sc_neg(t1, cbar);
  ge_table_mul(&bigc, &c->t1, t1); ge_table_mul(&t0, &c->t2, kbar); ge_add(&bigc, &bigc, &t0);
  ge_table_mul(&t0, &c->t3, sbar); ge_add(&bigc, &bigc, &t0); ge_scalarmult(&t0, &com_, gamma); ge_sub(&bigc, &bigc, &t0);   /* :826-829 */
  tr_add_element(&t, &bigc);                                                                                /* :839 */
  tr_challenge(gchk, &t);
  if (tr_out) memcpy(tr_out, t.buf, t.len);
  if (kprime_out) *kprime_out = kprime;
  if (!sc_eq(gchk, gamma)) return ORACLE_ERR_INVALID_CLIENT_SPEND_PROOF;                                    /* :842-844 */
  return ORACLE_OK;
}
int oracle_verify_spend(const oracle_ctx *c, const uint8_t sk[64], const uint8_t *proof, uint8_t *tr_out, uint8_t out_kprime[32]) {
  sc x; ge kp; sc_frombytes(x, sk);
  if (out_kprime) memset(out_kprime, 0, 32);
  int st = verify_spend_core(c, x, proof, &kp, tr_out);
  if (out_kprime && (st == ORACLE_OK || st == ORACLE_ERR_INVALID_CLIENT_SPEND_PROOF)) ristretto_encode(out_kprime, &kp);
  return st;
}
/* PrivateKey::refund, src/lib.rs:781-869 */
int oracle_refund(const oracle_ctx *c, const uint8_t sk[64], const uint8_t *proof, const uint8_t rng[128], uint8_t out[128]) {
  sc x, e, gamma, z; ge w, kp, g, x_a, a;
  memset(out, 0, 128);
  sc_frombytes(x, sk);
  if (!ristretto_decode(&w, sk + 32)) return ORACLE_ERR_UNDECODABLE;
  int st = verify_spend_core(c, x, proof, &kp, NULL);
  if (st != ORACLE_OK) return st;
  ge_basepoint(&g); ge_add(&x_a, &g, &kp);                                                                  /* :848 */
  sign_tail(c, x, &w, &x_a, rng, "refund", NULL, &a, e, gamma, z);                                          /* :846-861 */
  ristretto_encode(out, &a); sc_tobytes(out + 32, e); sc_tobytes(out + 64, gamma); sc_tobytes(out + 96, z);
  return ORACLE_OK;
}
/* PreRefund::to_credit_token, src/lib.rs:1217-1253 */
int oracle_refund_to_credit_token(const oracle_ctx *c, const uint8_t pre[96], const uint8_t *proof, const uint8_t refund[128],
                                  const uint8_t wenc[32], uint8_t out[160]) {
  const int L = c->L; ge w, a, kp, t0, cj, g, x_a; sc e, gamma, z;
  memset(out, 0, 160);
  if (!ristretto_decode(&w, wenc) || !ristretto_decode(&a, refund)) return ORACLE_ERR_UNDECODABLE;
  sc_frombytes(e, refund + 32); sc_frombytes(gamma, refund + 64); sc_frombytes(z, refund + 96);
  ge_identity(&kp);
  for (int j = 0; j < L; j++) {                                                                             /* :1224-1230 */
    sc p2; if (!ristretto_decode(&cj, proof + 32 * (PF_COM + j))) return ORACLE_ERR_UNDECODABLE;
    sc_pow2(p2, j); ge_scalarmult(&t0, &cj, p2); ge_add(&kp, &kp, &t0);
  }
  ge_basepoint(&g); ge_add(&x_a, &g, &kp);
  if (!dleq_check(c, &w, &x_a, &a, e, gamma, z, "refund", NULL)) return ORACLE_ERR_INVALID_REFUND_PROOF;    /* :1241-1243 */
  /* CreditToken{a,e,k,r,c=m}; PreRefund record r|k|m */
  memcpy(out, refund, 32); sc_tobytes(out + 32, e); memcpy(out + 64, pre + 32, 32); memcpy(out + 96, pre, 32); memcpy(out + 128, pre + 64, 32);
  return ORACLE_OK;
}

/* ============================ threaded batch loops (CPU baseline) ============================ */
typedef struct { const oracle_ctx *c; int kind; size_t lo, hi; const uint8_t *sk, *in0, *in1, *rng; uint8_t *out0, *out1, *status; } job;
static void *job_run(void *arg) {
  job *j = (job *)arg; const oracle_ctx *c = j->c;
  size_t pb = oracle_spend_proof_bytes(c), rb = oracle_prove_rng_bytes(c);
  for (size_t i = j->lo; i < j->hi; i++) {
    switch (j->kind) {
      case 0: j->status[i] = (uint8_t)oracle_verify_spend(c, j->sk, j->in0 + i * pb, NULL, NULL); break;
      case 1: j->status[i] = (uint8_t)oracle_refund(c, j->sk, j->in0 + i * pb, j->rng + i * 128, j->out0 + i * 128); break;
      case 2: oracle_prove_spend(c, j->in0 + i * 160, j->in1 + i * 32, j->rng + i * rb, j->out0 + i * pb, j->out1 + i * 96); break;
      case 3: j->status[i] = (uint8_t)oracle_issue(c, j->sk, j->in0 + i * 128, j->in1 + i * 32, j->rng + i * 128, j->out0 + i * 160); break;
      case 4: oracle_request(c, j->in0 + i * 64, j->rng + i * 128, j->out0 + i * 128); break;
    }
  }
  return NULL;
}
static void run_jobs(job proto, size_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  job *jobs = (job *)malloc(sizeof(job) * (size_t)nthreads);
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = proto; jobs[t].lo = n * (size_t)t / (size_t)nthreads; jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
    if (nthreads == 1) job_run(&jobs[t]); else pthread_create(&th[t], NULL, job_run, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
}
void oracle_verify_spend_batch(const oracle_ctx *c, const uint8_t sk[64], size_t n, const uint8_t *proofs, uint8_t *status, int nthreads) {
  job p = {c, 0, 0, 0, sk, proofs, NULL, NULL, NULL, NULL, status}; run_jobs(p, n, nthreads);
}
void oracle_refund_batch(const oracle_ctx *c, const uint8_t sk[64], size_t n, const uint8_t *proofs, const uint8_t *rng,
                         uint8_t *out, uint8_t *status, int nthreads) {
  job p = {c, 1, 0, 0, sk, proofs, NULL, rng, out, NULL, status}; run_jobs(p, n, nthreads);
}
void oracle_prove_spend_batch(const oracle_ctx *c, size_t n, const uint8_t *toks, const uint8_t *s, const uint8_t *rng,
                              uint8_t *out_proofs, uint8_t *out_pre, int nthreads) {
  job p = {c, 2, 0, 0, NULL, toks, s, rng, out_proofs, out_pre, NULL}; run_jobs(p, n, nthreads);
}
void oracle_issue_batch(const oracle_ctx *c, const uint8_t sk[64], size_t n, const uint8_t *reqs, const uint8_t *camt,
                        const uint8_t *rng, uint8_t *out, uint8_t *status, int nthreads) {
  job p = {c, 3, 0, 0, sk, reqs, camt, rng, out, NULL, status}; run_jobs(p, n, nthreads);
}
void oracle_request_batch(const oracle_ctx *c, size_t n, const uint8_t *pres, const uint8_t *rng, uint8_t *out, int nthreads) {
  job p = {c, 4, 0, 0, NULL, pres, NULL, rng, out, NULL, NULL}; run_jobs(p, n, nthreads);
}
