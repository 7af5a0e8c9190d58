// ge25519.h — twisted-Edwards (a = -1) extended-coordinate group law and ristretto255
// encode / decode / one-way map (RFC 9496) for gfx950 lanes, one point per lane.
// Restates what the reference takes from curve25519-dalek's RistrettoPoint (call sites
// /root/reference/src/lib.rs:465-1239; compress at src/transcript.rs:106).
// How a group element is computed does not change its canonical 32-byte encoding, so the
// kernels are free to use different addition chains than the reference.
//
// Operand-size discipline of fe25519.h is annotated per formula: {k} = limbs at most k times a tight element's (phi <= k);
// a product needs phi * gamma <= 12.
#pragma once
#include "fe25519.h"
#include "sc25519.h"

namespace act {

struct ge { fe X, Y, Z, T; };                 // extended; all coordinates tight
struct ge_cached { fe YpX, YmX, Z, T2d; };    // "projective Niels": YpX {2}, YmX {3}; Z, T2d tight
struct ge_niels { fe ypx, ymx, xy2d; };       // affine Niels (Z = 1); tight

ACT_HD ge ge_identity() { ge p; p.X = fe_zero(); p.Y = fe_one(); p.Z = fe_one(); p.T = fe_zero(); return p; }
ACT_HD ge ge_basepoint() { ge p; p.X = fe_base_x(); p.Y = fe_base_y(); p.Z = fe_one(); p.T = fe_base_t(); return p; }
ACT_HD ge_niels ge_niels_identity() { ge_niels n; n.ypx = fe_one(); n.ymx = fe_one(); n.xy2d = fe_zero(); return n; }
ACT_HD ge ge_neg(const ge& p) { ge r; r.X = fe_carry(fe_neg(p.X)); r.Y = p.Y; r.Z = p.Z; r.T = fe_carry(fe_neg(p.T)); return r; }

ACT_HD ge_cached ge_to_cached(const ge& p) {
  ge_cached c;
  c.YpX = fe_add(p.Y, p.X);          // {2}
  c.YmX = fe_sub(p.Y, p.X);          // {3}
  c.Z = p.Z;
  c.T2d = fe_mul(p.T, fe_d2());
  return c;
}
// per-lane conditional negation of a cached point: swap Y+X / Y-X, negate 2dT
ACT_HD ge_cached ge_cached_cneg(const ge_cached& c, bool neg) {
  ge_cached r = c;
  uint32_t m = fe_mask(neg);
  fe_cswap_m(r.YpX, r.YmX, m);
  r.T2d = fe_select_m(c.T2d, fe_neg(c.T2d), m);   // {2}
  return r;
}
ACT_HD ge ge_select_m(const ge& a, const ge& b, uint32_t m) {      // m all-ones: b, all-zeros: a
  ge r; r.X = fe_select_m(a.X, b.X, m); r.Y = fe_select_m(a.Y, b.Y, m); r.Z = fe_select_m(a.Z, b.Z, m); r.T = fe_select_m(a.T, b.T, m); return r;
}
ACT_HD ge_niels ge_niels_cneg(const ge_niels& c, bool neg) {
  ge_niels r = c;
  uint32_t m = fe_mask(neg);
  fe_cswap_m(r.ypx, r.ymx, m);
  r.xy2d = fe_select_m(c.xy2d, fe_neg(c.xy2d), m);
  return r;
}

// completed point (cx : cz) x (cy : ct) -> extended.  ct * cx, cy * cz, ct * cz, cy * cx must each fit the product budget
template <bool WITH_T = true>
ACT_HD ge ge_from_completed(const fe& cx, const fe& cy, const fe& cz, const fe& ct) {
  ge r;
  // operand order: fe_mul doubles limbs 1, 4, 7 of both operands; each operand appears twice on the same side, so the
  // compiler shares those preparations.  (fe_mul2: two independent products -- one interleaved statement in the range kernel's build)
  fe_mul2(r.X, r.Y, ct, cx, cy, cz);
  if (WITH_T) fe_mul2(r.Z, r.T, ct, cz, cy, cx); else { r.Z = fe_mul(ct, cz); r.T = fe_zero(); }
  return r;
}

// p + q, q cached.  8M.
template <bool WITH_T = true>
ACT_HD ge ge_add_cached(const ge& p, const ge_cached& q) {
  fe ypx = fe_add(p.Y, p.X);                 // {2}
  fe ymx = fe_sub(p.Y, p.X);                 // {3}
  fe pp, mm, tt2d, zz;
  fe_mul2(pp, mm, ypx, q.YpX, ymx, q.YmX);   // 2 * 2, 3 * 3
  fe_mul2(tt2d, zz, p.T, q.T2d, p.Z, q.Z);   // 1 * 2
  fe zz2 = fe_dbl(zz);                       // {2}
  fe cx = fe_sub(pp, mm);                    // {3}
  fe cy = fe_add(pp, mm);                    // {2}
  fe cz = fe_add(zz2, tt2d);                 // {3}
  fe ct = fe_sub(zz2, tt2d);                 // {4}: ct * cx = ct * cz = 12
  return ge_from_completed<WITH_T>(cx, cy, cz, ct);
}
// p + q, q affine Niels.  7M.
template <bool WITH_T = true>
ACT_HD ge ge_madd(const ge& p, const ge_niels& q) {
  fe ypx = fe_add(p.Y, p.X);
  fe ymx = fe_sub(p.Y, p.X);
  fe pp, mm;
  fe_mul2(pp, mm, ypx, q.ypx, ymx, q.ymx);
  fe tt2d = fe_mul(p.T, q.xy2d);
  fe zz2 = fe_dbl(p.Z);
  fe cx = fe_sub(pp, mm);
  fe cy = fe_add(pp, mm);
  fe cz = fe_add(zz2, tt2d);
  fe ct = fe_sub(zz2, tt2d);
  return ge_from_completed<WITH_T>(cx, cy, cz, ct);
}
// ---- d-free ("dedicated") addition, Hisil-Wong-Carter-Dawson 2008 section 3.2 for a = -1 -----------------------------
// A = (Y1-X1)(Y2+X2), B = (Y1+X1)(Y2-X2), C = 2 Z1 T2, D = 2 T1 Z2; (X3 : Y3 : Z3 : T3) = ((D+C)(B-A) : (B+A)(D-C) :
// (B-A)(B+A) : (D+C)(D-C)).  8M with NO multiplication by 2d, so the second operand is the point itself plus one lazy
// sum and difference (ge_ded) instead of a ge_cached that costs a multiplication to make.  The price: it is not complete.
// With q - p in E[4] (q = p, or q = p + a point of order 2 or 4) both B-A and D-C (or B+A) vanish and the result is
// (0,0,0,0); everywhere else it is the sum (p = identity included when q is not in E[4]).  msm.h chain_bu_pre uses it only
// where q - p in E[4] is impossible for ANY digit string (proof there) and keeps the complete formulas everywhere else.
struct ge_ded { fe YpX, YmX, Z, T; };          // YpX {2}, YmX {3}; Z, T tight
ACT_HD ge_ded ge_to_ded(const ge& p) {
  ge_ded c;
  c.YpX = fe_add(p.Y, p.X);          // {2}
  c.YmX = fe_sub(p.Y, p.X);          // {3}
  c.Z = p.Z; c.T = p.T;
  return c;
}
ACT_HD ge_ded ge_ded_cneg(const ge_ded& c, bool neg) {
  ge_ded r = c;
  uint32_t m = fe_mask(neg);
  fe_cswap_m(r.YpX, r.YmX, m);
  r.T = fe_select_m(c.T, fe_neg(c.T), m);         // {2}
  return r;
}
ACT_HD ge ge_add_ded(const ge& p, const ge_ded& q) {
  fe A, B, C, D;                                 // (fe_mul2: two independent products; one interleaved statement under -DACT_FE_PAIR_ASM)
  fe_mul2(A, B, fe_sub(p.Y, p.X), q.YpX, fe_add(p.Y, p.X), q.YmX);      // 3 * 2, 2 * 3
  fe_mul2(C, D, fe_dbl(p.Z), q.T, fe_dbl(p.T), q.Z);                    // 2 * 2
  fe E = fe_add(D, C);                           // {2}
  fe F = fe_sub(B, A);                           // {3}
  fe G = fe_add(B, A);                           // {2}
  fe H = fe_sub(D, C);                           // {3}
  ge r;                                          // every product 2 * 3
  fe_mul2(r.X, r.Y, E, F, G, H);
  fe_mul2(r.Z, r.T, G, F, E, H);
  return r;
}
ACT_HD ge ge_add(const ge& p, const ge& q) { return ge_add_cached(p, ge_to_cached(q)); }
ACT_HD ge ge_sub(const ge& p, const ge& q) { return ge_add_cached(p, ge_cached_cneg(ge_to_cached(q), true)); }

// 2p with the T coordinate computed only when `with_t` (a wave-uniform flag): 4S + 3M or 4S + 4M
ACT_HD ge ge_double_opt(const ge& p, bool with_t) {
#if !defined(ACT_DBL_CARRY)
  // ct = 2 Z^2 - (Y^2 - X^2) as (2 Z^2 + X^2) - Y^2 with the sum carried inside the squaring (fe_sqda_sq): {3} instead of {6}, so the
  // doubling needs no carry pass; cx with a 3p offset (Y^2 + X^2 <= 3p limb-wise): {4}; products ct cx = 12, cy cz = 6, ct cz = 9, cy cx = 8
  // (24 instructions fewer per doubling in the range kernel's build, 9 multiply-accumulates more; same-box A/B +0.25 % verifies/s:
  // profiles/r06_ab_fused_doubling.txt.  -DACT_DBL_CARRY keeps rounds 3-5's form -- {6} carried to {1} -- for A/B)
  fe xx, yy, w, xpy2;
  fe_sq2(xx, yy, p.X, p.Y);
  fe_sqda_sq(w, xpy2, p.Z, xx, fe_add(p.X, p.Y));
  fe yypxx = fe_add(yy, xx);                  // cy {2}
  fe yymxx = fe_sub(yy, xx);                  // cz {3}
  fe cx = fe_sub3(xpy2, yypxx);               // {4}
  fe ct = fe_sub(w, yy);                      // {3}
#else
  fe xx, yy, zz, xpy2;
  fe_sq2(xx, yy, p.X, p.Y);
  fe_sq2(zz, xpy2, p.Z, fe_add(p.X, p.Y));
  fe zz2 = fe_dbl(zz);
  fe yypxx = fe_add(yy, xx);
  fe yymxx = fe_sub(yy, xx);
  fe cx = fe_sub4(xpy2, yypxx);
  fe ct = fe_carry(fe_sub4(zz2, yymxx));      // operand sizes: ge_double below
#endif
  ge r;
  fe_mul2(r.X, r.Y, ct, cx, yypxx, yymxx);
  r.T = fe_zero();
  if (with_t) fe_mul2(r.Z, r.T, ct, yymxx, yypxx, cx); else r.Z = fe_mul(ct, yymxx);
  return r;
}
// 2p.  4S + 4M (3M without T).  Input T unused.
template <bool WITH_T = true>
ACT_HD ge ge_double(const ge& p) {
  fe xx = fe_sq(p.X), yy = fe_sq(p.Y);
  fe zz2 = fe_dbl(fe_sq(p.Z));               // {2}
  fe xpy2 = fe_sq(fe_add(p.X, p.Y));         // sq operand {2}
  fe yypxx = fe_add(yy, xx);                 // cy {2}
  fe yymxx = fe_sub(yy, xx);                 // cz {3}
  fe cx = fe_sub4(xpy2, yypxx);              // {5}
  fe ct = fe_carry(fe_sub4(zz2, yymxx));     // {6} -> tight: the one carry of a doubling (ct * cz would be 18; with ct tight
                                             // the largest product is cy * cx = 10)
  return ge_from_completed<WITH_T>(cx, yypxx, yymxx, ct);
}

// ---- ristretto255 ----------------------------------------------------------------------------
// RFC 9496 4.3.2; output = canonical little-endian words of s
ACT_HD void ristretto_encode(uint32_t out[8], const ge& p) {
  fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
  fe u2 = fe_mul(p.X, p.Y);
  fe inv;
  fe_invsqrt(inv, fe_mul(u1, fe_sq(u2)));
  fe d1 = fe_mul(inv, u1), d2 = fe_mul(inv, u2);
  fe zinv = fe_mul(fe_mul(d1, d2), p.T);
  fe ix = fe_mul(p.X, fe_sqrt_m1()), iy = fe_mul(p.Y, fe_sqrt_m1());
  fe ench = fe_mul(d1, fe_invsqrt_a_minus_d());
  bool rotate = fe_is_negative(fe_mul(p.T, zinv));
  fe x = fe_select(p.X, iy, rotate);
  fe y = fe_select(p.Y, ix, rotate);
  fe den = fe_select(d2, ench, rotate);
  y = fe_cneg(y, fe_is_negative(fe_mul(x, zinv)));          // {2}
  fe s = fe_mul(fe_sub4(p.Z, y), den);                       // Z - y {5}
  fe_to_words(out, fe_abs(s));
}
// ---- double-and-compress: the encoding of 2Q from Q with a field INVERSION instead of an inverse square root, so that a
// batch shares one exponentiation through Montgomery's trick (the construction of curve25519-dalek's
// RistrettoPoint::double_and_compress_batch).  The verifier computes the half-points Q = C'/2 (every scalar halved mod
// l; the representatives may then differ from C' by 4-torsion, which Ristretto encodings do not see) and encodes 2Q.
//   e = 2XY, f = Z^2 + dT^2, g = Y^2 + X^2, h = Z^2 - dT^2;  inv = 1 / ((e g)(f h)), or 0 when that product is 0
//   (2Q in the identity class: the encoding is then 32 zero bytes, which is what the formulas give with inv = 0).
struct dc_efgh { fe e, f, g, h; };       // e tight; f, g {2}; h {3}
ACT_HD dc_efgh dc_prepare(const ge& q) {
  dc_efgh s;
  fe xx = fe_sq(q.X), yy = fe_sq(q.Y), zz = fe_sq(q.Z);
  fe dtt = fe_mul(fe_sq(q.T), fe_d());
  s.e = fe_mul(q.X, fe_dbl(q.Y));
  s.f = fe_add(zz, dtt);
  s.g = fe_add(yy, xx);
  s.h = fe_sub(zz, dtt);
  return s;
}
// eg, fh and t = eg * fh with zero replaced by one (is_zero reports it)
ACT_HD fe dc_product(fe& eg, fe& fh, bool& is_zero, const dc_efgh& s) {
  eg = fe_mul(s.e, s.g); fh = fe_mul(s.f, s.h);
  fe t = fe_mul(eg, fh);
  is_zero = fe_is_zero(t);
  return fe_select(t, fe_one(), is_zero);
}
ACT_HD void dc_finish(uint32_t out[8], const dc_efgh& s, const fe& eg, const fe& fh, const fe& inv) {
  fe zinv = fe_mul(eg, inv), tinv = fe_mul(fh, inv);
  bool neg1 = fe_is_negative(fe_mul(eg, zinv));
  fe e = fe_select(s.e, s.g, neg1);                                  // {2}
  fe g = fe_carry(fe_select(s.g, fe_neg(s.e), neg1));                // tight
  fe h = fe_carry(fe_select(s.h, fe_mul(s.f, fe_sqrt_m1()), neg1));  // tight
  fe magic = fe_select(fe_invsqrt_a_minus_d(), fe_sqrt_m1(), neg1);
  bool neg2 = fe_is_negative(fe_mul(fe_mul(h, e), zinv));
  g = fe_cneg(g, neg2);                                              // {2}
  fe r = fe_mul(fe_sub4(h, g), fe_mul(magic, fe_mul(tinv, g)));      // h - g {5}
  fe_to_words(out, fe_abs(r));
}
// RFC 9496 4.3.1; returns false for non-canonical / negative / non-square / t negative / y == 0
ACT_HD bool ristretto_decode(ge& p, const uint32_t in[8]) {
  fe s = fe_from_words(in);
  uint32_t chk[8]; fe_to_words(chk, s);
  bool canonical = true;
  for (int i = 0; i < 8; i++) canonical = canonical && (chk[i] == in[i]);
  bool ok = canonical && !(in[0] & 1u);
  fe ss = fe_sq(s);
  fe u1 = fe_sub(fe_one(), ss);                              // {3}
  fe u2 = fe_add(fe_one(), ss);                              // {2}
  fe u2s = fe_sq(u2);
  fe du1s = fe_mul(fe_sq(u1), fe_d());
  fe v = fe_sub(fe_neg(du1s), u2s);                          // {4}: -(d*u1^2) - u2^2
  fe inv;
  bool was_square = fe_invsqrt(inv, fe_mul(v, u2s));
  fe dx = fe_mul(inv, u2);
  fe dy = fe_mul(v, fe_mul(inv, dx));
  fe x = fe_carry(fe_abs(fe_mul(fe_add(s, s), dx)));
  fe y = fe_mul(u1, dy);
  fe t = fe_mul(x, y);
  ok = ok && was_square && !fe_is_negative(t) && !fe_is_zero(y);
  p.X = x; p.Y = y; p.Z = fe_one(); p.T = t;
  return ok;
}
// RFC 9496 4.3.3
ACT_HD bool ristretto_equal(const ge& a, const ge& b) {
  bool e1 = fe_equal(fe_mul(a.X, b.Y), fe_mul(a.Y, b.X));
  bool e2 = fe_equal(fe_mul(a.Y, b.Y), fe_mul(a.X, b.X));
  return e1 || e2;
}
ACT_HD bool ristretto_is_identity(const ge& a) {   // == identity (0,1,1,0): X*1 == Y*0  or  Y*1 == X*0
  return fe_is_zero(a.X) || fe_is_zero(a.Y);
}
// RFC 9496 4.3.4 MAP; t tight
ACT_HD ge ristretto_map(const fe& t0) {
  fe one = fe_one();
  fe r = fe_mul(fe_sq(t0), fe_sqrt_m1());
  fe u = fe_mul(fe_add(r, one), fe_one_minus_d_sq());
  fe m1 = fe_carry(fe_neg(one));                                   // -1
  fe v = fe_mul(fe_sub(m1, fe_mul(r, fe_d())), fe_add(r, fe_d()));  // (-1 - r d)(r + d)
  fe s;
  bool was_square = fe_sqrt_ratio_m1(s, u, v);
  fe sp = fe_carry(fe_neg(fe_carry(fe_abs(fe_mul(s, t0)))));
  s = fe_select(s, sp, !was_square);
  fe c = fe_select(m1, r, !was_square);
  fe n = fe_carry(fe_sub(fe_mul(fe_mul(c, fe_carry(fe_sub(r, one))), fe_d_minus_one_sq()), v));
  fe w0 = fe_carry(fe_dbl(fe_mul(s, v)));
  fe w1 = fe_mul(n, fe_sqrt_ad_minus_one());
  fe s2 = fe_sq(s);
  fe w2 = fe_carry(fe_sub(one, s2));
  fe w3 = fe_carry(fe_add(one, s2));
  ge p;
  p.X = fe_mul(w0, w3); p.Y = fe_mul(w2, w1); p.Z = fe_mul(w1, w3); p.T = fe_mul(w0, w2);
  return p;
}
// RistrettoPoint::from_uniform_bytes: 16 little-endian words
ACT_HD ge ristretto_from_uniform(const uint32_t w[16]) {
  ge p0 = ristretto_map(fe_carry(fe_from_words(w)));
  ge p1 = ristretto_map(fe_carry(fe_from_words(w + 8)));
  return ge_add(p0, p1);
}

}  // namespace act
