// msm.h — the two scalar-multiplication shapes every kernel of the engine is built from.
//
// 1. Variable-base, right-to-left, every scalar on a base sharing ONE doubling chain P_i = 2^i N, with no per-lane
//    table of multiples (the reference's 8-entry table per `point * scalar` -- dalek variable_base -- would be
//    1 KiB per lane, which neither LDS at 2 waves/SIMD nor VGPRs can hold) and the same instruction schedule in all
//    64 lanes (no vartime-wNAF divergence to lose on a 64-wide SIMD).  The doublings are paid once for all scalars on
//    the same base: the verifier's C'_j0 / C'_j1 share Com_j
//    (/root/reference/src/lib.rs:814-816), A1/A2 share B_bar (:793-797), A*/Y_A share X_A (:849-853).
//      chain_bu      (k_spend_bits)      per-lane scalar through signed radix-16 Pippenger buckets in global memory,
//                                        wave-uniform scalar through two width-3-NAF accumulators in LDS
//      chain_b<NACC> (per-proof kernels) every scalar through its own bucket set
//      chain<NACC>, chain2u              the earlier radix-4 / plain-NAF forms, kept as cross-checks for the host build
// 2. fixed_base_acc: `&table * &scalar` (RistrettoBasepointTable, src/lib.rs:224-228) as one mixed addition per scalar
//    window from position-specific affine-Niels tables T[pos][d] = d * 2^(w*pos) * B.  The window width w is a property of the
//    table (FbTab): 16 bits by default (16 windows x 65 536 entries x 128 B = 128 MiB per base), 24 bits for h1 and h3 -- the
//    range kernels' bases -- in contexts sized for throughput (11 windows, 23.6 GB per base; engine.hip act_ctx_create).  Tables
//    are built on the GPU (k_build_table) once per (device, base, width) and shared by every context of the process on that GPU.
//    Measured on one MI355X: 16 against 12 bits (22 additions, 11 MiB per base) verify +2 %, prove_spend +12 %; 24 against 16
//    bits on distinct proofs +3.2 % verifies/s (profiles/r03_window_ab_distinct.json).
#pragma once
#include "ge25519.h"

namespace act {

// ---- fixed-base tables -------------------------------------------------------------------------
#ifndef ACT_FB_WBITS
#define ACT_FB_WBITS 16
#endif
constexpr int FB_WBITS = ACT_FB_WBITS;                      // window width in bits
constexpr int FB_WINDOWS = (253 + FB_WBITS - 1) / FB_WBITS;  // scalars are canonical: < l < 2^253
constexpr int FB_ENTRIES = 1 << FB_WBITS;
constexpr int NIELS_WORDS = 32;         // 27 used (ypx, ymx, xy2d), padded to 128 B = one L2 line
constexpr size_t FB_TABLE_WORDS = (size_t)FB_WINDOWS * FB_ENTRIES * NIELS_WORDS;   // 128 MiB per base at 16 bits

ACT_HD ge_niels niels_load(const uint32_t* p) {
  ge_niels n;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a[7];
  for (int i = 0; i < 7; i++) a[i] = q[i];
  uint32_t w[28];
  for (int i = 0; i < 7; i++) { w[4 * i] = a[i].x; w[4 * i + 1] = a[i].y; w[4 * i + 2] = a[i].z; w[4 * i + 3] = a[i].w; }
  for (int i = 0; i < FE_LIMBS; i++) { n.ypx.v[i] = w[i]; n.ymx.v[i] = w[FE_LIMBS + i]; n.xy2d.v[i] = w[2 * FE_LIMBS + i]; }
#else
  for (int i = 0; i < FE_LIMBS; i++) { n.ypx.v[i] = p[i]; n.ymx.v[i] = p[FE_LIMBS + i]; n.xy2d.v[i] = p[2 * FE_LIMBS + i]; }
#endif
  return n;
}
ACT_HD void niels_store(uint32_t* p, const ge_niels& n) {
  for (int i = 0; i < FE_LIMBS; i++) { p[i] = n.ypx.v[i]; p[FE_LIMBS + i] = n.ymx.v[i]; p[2 * FE_LIMBS + i] = n.xy2d.v[i]; }
  for (int i = 3 * FE_LIMBS; i < NIELS_WORDS; i++) p[i] = 0;
}
// affine Niels form of an extended point with Z == 1 (x, y, t = xy all tight)
ACT_HD ge_niels niels_from_affine(const ge& p) {
  ge_niels n;
  n.ypx = fe_carry(fe_add(p.Y, p.X));
  n.ymx = fe_carry(fe_sub(p.Y, p.X));
  n.xy2d = fe_mul(p.T, fe_d2());
  return n;
}
constexpr int GE_WORDS = 4 * FE_LIMBS;    // 36 words = 144 B = nine 16-byte pieces
ACT_HD void ge_words(uint32_t w[GE_WORDS], const ge& g) {
  for (int i = 0; i < FE_LIMBS; i++) { w[i] = g.X.v[i]; w[FE_LIMBS + i] = g.Y.v[i]; w[2 * FE_LIMBS + i] = g.Z.v[i]; w[3 * FE_LIMBS + i] = g.T.v[i]; }
}
ACT_HD ge ge_of_words(const uint32_t w[GE_WORDS]) {
  ge g;
  for (int i = 0; i < FE_LIMBS; i++) { g.X.v[i] = w[i]; g.Y.v[i] = w[FE_LIMBS + i]; g.Z.v[i] = w[2 * FE_LIMBS + i]; g.T.v[i] = w[3 * FE_LIMBS + i]; }
  return g;
}
ACT_HD void ge_store(uint32_t* p, const ge& g) { ge_words(p, g); }
ACT_HD ge ge_load(const uint32_t* p) { return ge_of_words(p); }

// A fixed-base table and its window width.  The width is a property of the table, chosen per base when a context is created
// (engine.hip): 16 bits by default; the two bases of the range kernel (h1, h3) get 24-bit windows -- 11 instead of 16 table
// additions per product, 23.6 GB per base -- in contexts sized for throughput (measured on one MI355X against 16 bits:
// 20 bits +1.6 %, 22 bits +2.0 %, 24 bits +2.7 % verifies/s).  `id` = which base (host-side operation counting only).
struct FbTab { const uint32_t* p; uint32_t wbits; uint32_t id; };
ACT_HD uint32_t fb_windows(uint32_t wbits) { return (253u + wbits - 1u) / wbits; }      // scalars are canonical: < l < 2^253
ACT_HD size_t fb_table_words(uint32_t wbits) { return (size_t)fb_windows(wbits) * ((size_t)1 << wbits) * NIELS_WORDS; }

// acc += s * B using B's table; s canonical (< l).  fb_windows(wbits) mixed additions, no doublings.  The entry of window
// pos+1 is loaded before the addition of window pos (the tables live in L2 / Infinity Cache / HBM: ~550-cycle loads).
ACT_HD uint32_t fb_next_digit(uint32_t w[8], uint32_t wbits) {
  uint32_t digit = w[0] & ((1u << wbits) - 1u);
  // shift the scalar down one window (static indices only: a runtime-indexed limb array would live in scratch)
  for (int i = 0; i < 7; i++) w[i] = (w[i] >> wbits) | (w[i + 1] << (32u - wbits));
  w[7] >>= wbits;
  return digit;
}
ACT_HD ge fixed_base_acc(ge acc, const FbTab& t, const sc& s) {
  fe_count_fixed_base(t.id);
  uint32_t w[8];
  for (int i = 0; i < 8; i++) w[i] = s.v[i];
  const uint32_t nw = fb_windows(t.wbits);
  const size_t entries = (size_t)1 << t.wbits;
  ge_niels cur = niels_load(t.p + (size_t)fb_next_digit(w, t.wbits) * NIELS_WORDS);
  for (uint32_t pos = 1; pos < nw; pos++) {
    ge_niels nxt = niels_load(t.p + ((size_t)pos * entries + fb_next_digit(w, t.wbits)) * NIELS_WORDS);
    acc = ge_madd(acc, cur);
    cur = nxt;
  }
  return ge_madd(acc, cur);
}

// ---- shared-doubling variable-base chain ---------------------------------------------------------
// radix-4 digit extraction with carry: v = (low 2 bits) + carry in 0..4 -> digit in {0,1,2,-1,0}
struct digit4 { bool nonzero, two, neg; };
ACT_HD digit4 next_digit4(uint32_t w[8], uint32_t& carry) {
  uint32_t v = (w[0] & 3u) + carry;
  for (int i = 0; i < 7; i++) w[i] = (w[i] >> 2) | (w[i + 1] << 30);
  w[7] >>= 2;
  digit4 d;
  carry = v >= 3u ? 1u : 0u;
  d.nonzero = (v != 0u) && (v != 4u);
  d.two = (v == 2u);
  d.neg = (v == 3u);
  return d;
}
ACT_HD ge chain_step_add(const ge& acc, const ge_cached& c1, const ge_cached& c2, const digit4& d) {
  ge_cached q;
  uint32_t m2 = fe_mask(d.two);
  q.YpX = fe_select_m(c1.YpX, c2.YpX, m2);
  q.YmX = fe_select_m(c1.YmX, c2.YmX, m2);
  q.Z = fe_select_m(c1.Z, c2.Z, m2);
  q.T2d = fe_select_m(c1.T2d, c2.T2d, m2);
  q = ge_cached_cneg(q, d.neg);
  return ge_add_cached(acc, q);
}
// acc[a] += s[a] * N for a < NACC; scalars canonical (< l < 2^253): 127 radix-4 digits
template <int NACC>
ACT_HD void chain(ge* acc, const ge& N, const sc* s) {
  uint32_t w[NACC][8], carry[NACC];
  for (int a = 0; a < NACC; a++) { carry[a] = 0; for (int i = 0; i < 8; i++) w[a][i] = s[a].v[i]; }
  ge P = N;
  for (int step = 0; step < 127; step++) {
    ge_cached c1 = ge_to_cached(P);
    ge Q = ge_double(P);
    ge_cached c2 = ge_to_cached(Q);
    if (step < 126) P = ge_double(Q);
#pragma unroll
    for (int a = 0; a < NACC; a++) {
      digit4 d = next_digit4(w[a], carry[a]);
      if (d.nonzero) acc[a] = chain_step_add(acc[a], c1, c2, d);
    }
  }
}

// ---- chain2u: acc_l += s_l * N (per-lane scalar, radix-4 digits) and acc_u += s_u * N where s_u is the SAME
// scalar in every lane of the wavefront (the proof-wide challenge gamma: a wavefront holds 64 bits of one proof
// when L >= 64).  s_u is recoded in non-adjacent form over the single-doubling positions of the same chain;
// a zero digit is zero in every lane, so the wavefront skips that addition entirely (no exec-mask loss):
// ~84 additions instead of 127.  With mixed proofs per wavefront (L < 64) the digits differ per lane and the
// branch merely diverges; the result is the same.
struct naf_state { uint32_t w[9]; };     // scalar + carry headroom
ACT_HD naf_state naf_init(const sc& s) { naf_state n; for (int i = 0; i < 8; i++) n.w[i] = s.v[i]; n.w[8] = 0; return n; }
// returns digit in {-1, 0, +1} for the current bit and advances one bit: standard NAF (k odd -> d = 2 - (k mod 4))
ACT_HD int naf_next(naf_state& n) {
  int d = 0;
  if (n.w[0] & 1u) {
    d = 2 - (int)(n.w[0] & 3u);          // +1 or -1
    if (d < 0) {                         // k += 1
      uint64_t c = 1;
      for (int i = 0; i < 9; i++) { c += n.w[i]; n.w[i] = (uint32_t)c; c >>= 32; }
    } else {
      n.w[0] &= ~1u;                     // k -= 1
    }
  }
  for (int i = 0; i < 8; i++) n.w[i] = (n.w[i] >> 1) | (n.w[i + 1] << 31);
  n.w[8] >>= 1;
  return d;
}
// width-3 NAF (digits 0, +-1, +-3; a nonzero digit is followed by at least two zeros, density 1/4): k odd -> d = k mods 8
ACT_HD int naf3_next(naf_state& n) {
  int d = 0;
  if (n.w[0] & 1u) {
    const uint32_t m = n.w[0] & 7u;
    d = m < 4u ? (int)m : (int)m - 8;
    n.w[0] &= ~7u;                       // k -= m
    if (d < 0) {                         // ... += 8
      uint64_t c = 8;
      for (int i = 0; i < 9; i++) { c += n.w[i]; n.w[i] = (uint32_t)c; c >>= 32; }
    }
  }
  for (int i = 0; i < 8; i++) n.w[i] = (n.w[i] >> 1) | (n.w[i + 1] << 31);
  n.w[8] >>= 1;
  return d;
}
ACT_HD void chain2u(ge& acc_l, ge& acc_u, const ge& N, const sc& s_l, const sc& s_u) {
  uint32_t w[8], carry = 0;
  for (int i = 0; i < 8; i++) w[i] = s_l.v[i];
  naf_state nu = naf_init(s_u);
  ge P = N;
  for (int step = 0; step < 127; step++) {
    ge_cached c1 = ge_to_cached(P);
    ge Q = ge_double(P);
    ge_cached c2 = ge_to_cached(Q);
    if (step < 126) P = ge_double(Q);
    digit4 d = next_digit4(w, carry);
    if (d.nonzero) acc_l = chain_step_add(acc_l, c1, c2, d);
    int u0 = naf_next(nu);
    if (u0 != 0) acc_u = ge_add_cached(acc_u, ge_cached_cneg(c1, u0 < 0));
    int u1 = naf_next(nu);
    if (u1 != 0) acc_u = ge_add_cached(acc_u, ge_cached_cneg(c2, u1 < 0));
  }
  // s_u < l < 2^253 has at most 254 NAF digits (positions 0..253): all consumed by the 127 steps above
}

// ---- chain_bu: bucketed per-lane accumulator + wave-uniform NAF accumulator on one doubling chain ----------
// acc_u += s_u * N as in chain2u.  The per-lane scalar s_l is recoded in signed radix 16 (digits in [-8, 8]) and
// handled Pippenger-style: every fourth chain point P_i = 16^i N is added, with the digit's sign, into bucket
// |digit| of the lane (9 extended points per lane in global memory, AoS so a lane's bucket is 144 contiguous
// bytes; bucket 0 absorbs the zero digits so no lane ever sits out or selects).  Afterwards
// s_l * N = sum_v v * B_v by running sums: 64 + 14 additions instead of 127, one cached conversion per four
// doublings instead of four, and T is computed only for chain points that are actually added (the NAF
// positions are wave-uniform, so that is a uniform branch).  Working set: 1296 B per lane, re-touched every
// step -> served by L2 / Infinity Cache.
constexpr int BUCKETS = 9;
constexpr int BUCKET_WORDS = BUCKETS * GE_WORDS;     // 324 words = 1296 B per lane

constexpr int GE_PIECES = GE_WORDS / 4;     // 16-byte pieces of an extended point
ACT_HD ge bucket_load(const uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a[GE_PIECES];
  for (int i = 0; i < GE_PIECES; i++) a[i] = q[i];
  uint32_t w[GE_WORDS];
  for (int i = 0; i < GE_PIECES; i++) { w[4 * i] = a[i].x; w[4 * i + 1] = a[i].y; w[4 * i + 2] = a[i].z; w[4 * i + 3] = a[i].w; }
  return ge_of_words(w);
#else
  return ge_load(p);
#endif
}
ACT_HD void bucket_store(uint32_t* p, const ge& g) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t w[GE_WORDS];
  ge_words(w, g);
  uint4* q = reinterpret_cast<uint4*>(p);
  for (int i = 0; i < GE_PIECES; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
#else
  ge_store(p, g);
#endif
}
constexpr int GE_LDS_WORDS_PER_WAVE = GE_PIECES * 64 * 4;   // one extended point per lane of a wavefront

#if defined(__HIP_DEVICE_COMPILE__)
// home of an extended point in LDS: [piece][lane][4 words], so the ds_read/write_b128 are conflict-free
__device__ __forceinline__ void ge_to_lds(uint32_t* lds_wave, const ge& g) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t w[GE_WORDS];
  ge_words(w, g);
  for (int k = 0; k < GE_PIECES; k++)
    *reinterpret_cast<uint4*>(lds_wave + (k * 64 + lane) * 4) = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__device__ __forceinline__ ge ge_from_lds(const uint32_t* lds_wave) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t w[GE_WORDS];
  for (int k = 0; k < GE_PIECES; k++) {
    const uint4 q = *reinterpret_cast<const uint4*>(lds_wave + (k * 64 + lane) * 4);
    w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
  }
  return ge_of_words(w);
}
#endif

// `bk`: this lane's BUCKET_WORDS words.  Returns s_l * N in acc_l (overwritten), adds s_u * N to acc_u.
// `lds_wave` (device): 2 * GE_LDS_WORDS_PER_WAVE words of LDS owned by the calling wavefront; unused on the host.
// The wave-uniform scalar is recoded in width-3 NAF with two accumulators, U1 for digits +-1 and U3 for digits +-3
// (acc_u += U1 + U3 + 2*U3 at the end: ~63 + 3 additions instead of ~84 with plain NAF).  On the device both live in
// LDS between their additions -- one addition site per position, its operand picked by address -- which also takes
// the 36 accumulator registers out of the loop's live set.
// Digit strings of chain_bu, recoded once instead of inside the chain loop (where the 9-word NAF state with its 64-bit
// carries and the 8-word radix-16 state would sit in the loop's VGPR set):
//   naf3_recode   the wave-uniform scalar -> 64 words, word `step` = the width-3-NAF digits of chain positions 4*step .. 4*step+3
//                 as four signed bytes (0, +-1, +-3); written once per proof by k_spend_prep, read with scalar loads
//   radix16_bias  the per-lane scalar s -> t = s + 0x88...8 (8 words): nibble i of t, minus 8, is signed radix-16 digit i in
//                 [-8, 7] (sum (nib_i - 8) 16^i = t - 0x88...8 = s; s < 2^253 so t < 2^256), no carry to propagate
constexpr int NAF_WORDS = 64;
ACT_HD void naf3_recode(uint32_t out[NAF_WORDS], const sc& s) {
  naf_state n = naf_init(s);
  for (int step = 0; step < NAF_WORDS; step++) {
    uint32_t wd = 0;
    for (int k = 0; k < 4; k++) wd |= ((uint32_t)naf3_next(n) & 0xffu) << (8 * k);
    out[step] = wd;
  }
}
ACT_HD void radix16_bias(uint32_t t[8], const sc& s) {
  uint64_t c = 0;
  for (int i = 0; i < 8; i++) { c += (uint64_t)s.v[i] + 0x88888888u; t[i] = (uint32_t)c; c >>= 32; }
}
ACT_HD int naf_byte(uint32_t wd, int k) { return (int)(int8_t)(wd >> (8 * k)); }

// chain_bu with both digit strings taken from memory: `nafw` = NAF_WORDS words of the uniform scalar (naf3_recode), `dg` = the
// lane's 8 radix16_bias words (in global memory: one load per 8 steps instead of 8 live registers).  UNIFORM: every lane of
// the wavefront reads the same nafw (L a multiple of 64), so the word is moved to an SGPR and its branches are scalar.
//
// The additions of a chain point into a bucket or into U1 / U3 use the d-free formulas (ge25519.h ge_add_ded: no cached form,
// one multiplication less per addition, ~127 per lane).  They fail only when (chain point) - (accumulator) lies in E[4].
// Accumulator and chain point are multiples of N: accumulator = a N with a = sum of +-B^i over earlier positions i < k
// that carried this bucket's (this accumulator's) digit magnitude, chain point = +-B^k N (B = 16 resp. 2), so the
// difference is m N with m = a -+ B^k, a nonzero integer (|a| < B^k) below 2^257.  N is a decoded ristretto point: a point of
// order l plus an element of E[4], hence m N in E[4] iff l | m, unless N itself is in E[4], which only the all-zero encoding
// decodes to (`n_small`: the caller says so and gets the identity for both results, which is what 0 * scalar is).
// m = t l is impossible: the digits of m are in {-1, 0, 1} at distinct positions, and such a string IS the canonical signed
// radix-16 expansion (resp. the non-adjacent form: width-3 NAF positions are >= 3 apart) of its value, which is unique --
// but the expansion of t l has at least 24 digits outside {-1, 0, 1} and its NAF at least 15 adjacent-but-one pairs for
// every 1 <= |t| < 40 (tests/test_hostcheck.py::test_dedicated_addition_never_exceptional recomputes this).  An empty
// accumulator (identity) is fine: identity + q is exact for q outside E[4].  The bucket combine, U1 + 3 U3 and everything
// outside this function can meet equal operands (two empty buckets) and keep the complete formulas.
template <bool UNIFORM>
ACT_HD void chain_bu_pre(ge& acc_l, ge& acc_u, const ge& N, const uint32_t* dg, const uint32_t* nafw, uint32_t* bk, uint32_t* lds_wave = nullptr, bool n_small = false) {
  const ge id = ge_identity();
  // Bucket traffic is what this kernel pays for at the fabric (~20 KB per lane through L2 to the Infinity
  // Cache, 8 % of the kernel's time, mostly as clock).  Two cuts that cost no arithmetic: a zero digit (1 in 16) touches no
  // memory -- bucket 0 is never read again, the wavefront runs the addition anyway --, and a bucket is not initialised: `touched`
  // has a bit per bucket, a first visit starts from the identity in registers (exact for the d-free addition, see below) and
  // the combine takes the identity for buckets that were never visited.
  uint32_t touched = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  ge_to_lds(lds_wave, id);
  ge_to_lds(lds_wave + GE_LDS_WORDS_PER_WAVE, id);
  auto add_u = [&](const ge_ded& q, int d) {
    uint32_t* home = lds_wave + ((d == 3 || d == -3) ? GE_LDS_WORDS_PER_WAVE : 0);
    ge_to_lds(home, ge_add_ded(ge_from_lds(home), ge_ded_cneg(q, d < 0)));
  };
#else
  ge U[2] = {id, id};
  auto add_u = [&](const ge_ded& q, int d) {
    ge& t = U[(d == 3 || d == -3) ? 1 : 0];
    t = ge_add_ded(t, ge_ded_cneg(q, d < 0));
  };
#endif
  ge P = N;                                    // position 0, T valid
  uint32_t dw = 0;
#if defined(ACT_BUCKET_PREFETCH)
  // The bucket of step s + 1 is requested before the LAST doubling of step s (its digit is known; the doubling's live set leaves
  // room for the 36 words), so that the ~2 us the load spends between L2, Infinity Cache and HBM pass under ~630 instructions of
  // arithmetic instead of in front of the addition that needs it (SQ_WAIT_ANY was 13.9 % of the wave's cycles).
  ge Bpre = id;
  uint32_t nib_next = dg[0] & 15u;
#endif
  for (int step = 0; step < 64; step++) {
#if defined(ACT_BUCKET_PREFETCH)
    if ((step & 7) == 0) dw = dg[step >> 3] >> 4;
    const uint32_t nib = nib_next;
#else
    if ((step & 7) == 0) dw = dg[step >> 3];
    const uint32_t nib = dw & 15u; dw >>= 4;
#endif
    const bool neg = nib < 8u;
    const uint32_t mag = neg ? 8u - nib : nib - 8u;     // 0..8
    uint32_t nw = nafw[step];
#if defined(__HIP_DEVICE_COMPILE__)
    if (UNIFORM) nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)nw);
#endif
    const int u = naf_byte(nw, 0), u1 = naf_byte(nw, 1), u2 = naf_byte(nw, 2), u3 = naf_byte(nw, 3);
    ge_ded c = ge_to_ded(P);
    uint32_t* slot = bk + mag * GE_WORDS;
    if (mag != 0u) {
#if defined(ACT_BUCKET_PREFETCH)
      ge B = Bpre;                               // the identity on a first visit
#else
      ge B = id;
      if ((touched >> mag) & 1u) B = bucket_load(slot);
#endif
      touched |= 1u << mag;
      B = ge_add_ded(B, ge_ded_cneg(c, neg));
      bucket_store(slot, B);
    }
    if (u != 0) add_u(c, u);
    if (step == 63) {                          // position 253 (u1) is the last possible digit of a scalar < 2^253; 254, 255 are zero
      if (u1 != 0) { P = ge_double_opt(P, true); add_u(ge_to_ded(P), u1); }
      break;
    }
    P = ge_double_opt(P, u1 != 0);
    if (u1 != 0) add_u(ge_to_ded(P), u1);
    P = ge_double_opt(P, u2 != 0);
    if (u2 != 0) add_u(ge_to_ded(P), u2);
    P = ge_double_opt(P, u3 != 0);
    if (u3 != 0) add_u(ge_to_ded(P), u3);
#if defined(ACT_BUCKET_PREFETCH)
    {
      if (((step + 1) & 7) == 0) nib_next = dg[(step + 1) >> 3] & 15u; else { nib_next = dw & 15u; dw >>= 4; }
      const uint32_t mag_next = nib_next < 8u ? 8u - nib_next : nib_next - 8u;
      Bpre = id;
      if (mag_next != 0u && ((touched >> mag_next) & 1u)) Bpre = bucket_load(bk + mag_next * GE_WORDS);     // after this step's store: same lane, program order
    }
#endif
    P = ge_double_opt(P, true);                // next step's bucket point needs T
  }
  {                                            // acc_u += U1 + 3 * U3
#if defined(__HIP_DEVICE_COMPILE__)
    ge U1 = ge_from_lds(lds_wave), U3 = ge_from_lds(lds_wave + GE_LDS_WORDS_PER_WAVE);
#else
    ge U1 = U[0], U3 = U[1];
#endif
    const uint32_t ms = fe_mask(n_small);
    U1 = ge_select_m(U1, id, ms); U3 = ge_select_m(U3, id, ms);
    acc_u = ge_add_cached(acc_u, ge_to_cached(U1));
    acc_u = ge_add_cached(acc_u, ge_to_cached(U3));
    acc_u = ge_add_cached(acc_u, ge_to_cached(ge_double_opt(U3, true)));
  }
  auto bucket = [&](int vv) { ge B = id; if ((touched >> vv) & 1u) B = bucket_load(bk + vv * GE_WORDS); return B; };
  ge S = bucket(8);
  ge R = S;
  for (int vv = 7; vv >= 1; vv--) {
    S = ge_add_cached(S, ge_to_cached(bucket(vv)));
    R = ge_add_cached(R, ge_to_cached(S));
  }
  acc_l = ge_select_m(R, id, fe_mask(n_small));
}

ACT_HD void chain_bu(ge& acc_l, ge& acc_u, const ge& N, const sc& s_l, const sc& s_u, uint32_t* bk, uint32_t* lds_wave = nullptr) {
  const ge id = ge_identity();
  for (int b = 0; b < BUCKETS; b++) bucket_store(bk + b * GE_WORDS, id);
  uint32_t w[8], carry = 0;
  for (int i = 0; i < 8; i++) w[i] = s_l.v[i];
  naf_state nu = naf_init(s_u);
#if defined(__HIP_DEVICE_COMPILE__)
  ge_to_lds(lds_wave, id);
  ge_to_lds(lds_wave + GE_LDS_WORDS_PER_WAVE, id);
  auto add_u = [&](const ge_cached& q, int d) {
    uint32_t* home = lds_wave + ((d == 3 || d == -3) ? GE_LDS_WORDS_PER_WAVE : 0);
    ge_to_lds(home, ge_add_cached(ge_from_lds(home), ge_cached_cneg(q, d < 0)));
  };
#else
  ge U[2] = {id, id};
  auto add_u = [&](const ge_cached& q, int d) {
    ge& t = U[(d == 3 || d == -3) ? 1 : 0];
    t = ge_add_cached(t, ge_cached_cneg(q, d < 0));
  };
#endif
  ge P = N;                                    // position 0, T valid
  for (int step = 0; step < 64; step++) {
    // per-lane signed radix-16 digit
    uint32_t v = (w[0] & 15u) + carry;
    for (int i = 0; i < 7; i++) w[i] = (w[i] >> 4) | (w[i + 1] << 28);
    w[7] >>= 4;
    carry = v > 8u ? 1u : 0u;
    bool neg = v > 8u;
    uint32_t mag = neg ? 16u - v : v;          // 0..8
    ge_cached c = ge_to_cached(P);
    uint32_t* slot = bk + mag * GE_WORDS;
    ge B = bucket_load(slot);
    B = ge_add_cached(B, ge_cached_cneg(c, neg));
    bucket_store(slot, B);
    int u = naf3_next(nu);
    if (u != 0) add_u(c, u);
    // three intermediate positions: only the uniform accumulators may use them
    int u1 = naf3_next(nu), u2 = naf3_next(nu), u3 = naf3_next(nu);
    if (step == 63) {                          // position 253 (u1) is the last possible digit of a scalar < 2^253; 254, 255 are zero
      if (u1 != 0) { P = ge_double_opt(P, true); add_u(ge_to_cached(P), u1); }
      break;
    }
    P = ge_double_opt(P, u1 != 0);
    if (u1 != 0) add_u(ge_to_cached(P), u1);
    P = ge_double_opt(P, u2 != 0);
    if (u2 != 0) add_u(ge_to_cached(P), u2);
    P = ge_double_opt(P, u3 != 0);
    if (u3 != 0) add_u(ge_to_cached(P), u3);
    P = ge_double_opt(P, true);                // next step's bucket point needs T
  }
  {                                            // acc_u += U1 + 3 * U3
#if defined(__HIP_DEVICE_COMPILE__)
    const ge U1 = ge_from_lds(lds_wave), U3 = ge_from_lds(lds_wave + GE_LDS_WORDS_PER_WAVE);
#else
    const ge U1 = U[0], U3 = U[1];
#endif
    acc_u = ge_add_cached(acc_u, ge_to_cached(U1));
    acc_u = ge_add_cached(acc_u, ge_to_cached(U3));
    acc_u = ge_add_cached(acc_u, ge_to_cached(ge_double_opt(U3, true)));
  }
  // sum_v v * B_v = running sums: S = B8 + ... + Bv, R accumulates S
  ge S = bucket_load(bk + 8 * GE_WORDS);
  ge R = S;
  for (int vv = 7; vv >= 1; vv--) {
    S = ge_add_cached(S, ge_to_cached(bucket_load(bk + vv * GE_WORDS)));
    R = ge_add_cached(R, ge_to_cached(S));
  }
  acc_l = R;
}

// ---- chain_b<NACC>: acc[a] += s[a] * N for per-lane scalars, all through signed radix-16 Pippenger buckets on ONE
// doubling chain (the bucket half of chain_bu, for the per-proof kernels): 64 bucket additions + a 14-addition
// combine per scalar and one cached conversion per four doublings, instead of chain<NACC>'s 127 digit additions
// and two cached conversions per two doublings.  `bk`: NACC * BUCKET_WORDS words owned by this lane.
template <int NACC>
ACT_HD void chain_b(ge* acc, const ge& N, const sc* s, uint32_t* bk) {
  const ge id = ge_identity();
  for (int b = 0; b < NACC * BUCKETS; b++) bucket_store(bk + b * GE_WORDS, id);
  uint32_t w[NACC][8], carry[NACC];
  for (int a = 0; a < NACC; a++) { carry[a] = 0; for (int i = 0; i < 8; i++) w[a][i] = s[a].v[i]; }
  ge P = N;                                    // position 0, T valid
  for (int step = 0; step < 64; step++) {
    ge_cached c = ge_to_cached(P);
#pragma unroll
    for (int a = 0; a < NACC; a++) {
      uint32_t v = (w[a][0] & 15u) + carry[a];
      for (int i = 0; i < 7; i++) w[a][i] = (w[a][i] >> 4) | (w[a][i + 1] << 28);
      w[a][7] >>= 4;
      carry[a] = v > 8u ? 1u : 0u;
      bool neg = v > 8u;
      uint32_t mag = neg ? 16u - v : v;        // 0..8; bucket 0 absorbs the zero digits
      uint32_t* slot = bk + ((size_t)a * BUCKETS + mag) * GE_WORDS;
      bucket_store(slot, ge_add_cached(bucket_load(slot), ge_cached_cneg(c, neg)));
    }
    if (step == 63) break;
    P = ge_double_opt(P, false); P = ge_double_opt(P, false); P = ge_double_opt(P, false);
    P = ge_double_opt(P, true);                // the next bucket point needs T
  }
  for (int a = 0; a < NACC; a++) {             // sum_v v * B_v by running sums
    const uint32_t* ba = bk + (size_t)a * BUCKET_WORDS;
    ge S = bucket_load(ba + 8 * GE_WORDS);
    ge R = S;
    for (int vv = 7; vv >= 1; vv--) {
      S = ge_add_cached(S, ge_to_cached(bucket_load(ba + vv * GE_WORDS)));
      R = ge_add_cached(R, ge_to_cached(S));
    }
    acc[a] = ge_add_cached(acc[a], ge_to_cached(R));
  }
}

// ---- secret scalars --------------------------------------------------------------------------------------------
// The reference is constant-time in its table accesses (`subtle`, /root/reference/src/lib.rs:98, 1025-1118; dalek's table scans).
// Here no secret scalar ever selects a memory address in the default build (-DACT_CT_SECRET_TABLES, which the Makefile's `all` sets):
//   the ISSUER's secrets (the key x in the verifier's (e_bar - x gamma) A' and the signer's (e + x)^-1, the signing nonces e, alpha)
//                         in EVERY build: variable-base chains are chain_ct, fixed-base products go through IssuerFb (kernels.h);
//   the CLIENT's secrets  (prover: SecretFb / ACT_SECRET_FB_LDS in kernels.h and chain_s here; request) in the default build the
//                         same; in `make fast` (libact_mi355x_fast.so) fixed_base_acc / chain_b -- table entries and buckets
//                         addressed by scalar digits: the memory-access pattern depends on them, the instruction stream does not.
//   fixed-base products   fixed_base_acc_mf below: the entry of a 64-entry window picked on the MATRIX CORES (37 additions per
//                         product; measured prove_spend 0.48 x, request 0.55 x of the fast build).  Rounds 2-3 scanned 8-entry
//                         windows with masks (fixed_base_acc_ct: 64 additions per product, 0.26 x); that form is kept for the host
//                         test build and for same-box A/B (-DACT_CT_GLOBAL_SCAN / -DACT_CT_LDS_SCAN).
//   variable-base chains  chain_ct: radix-4 digits on the shared doubling chain, EVERY digit addition executed (a zero digit adds
//                         the identity, picked with masks), everything in registers: no buckets, so nothing to address; 1.36 x the
//                         bucket chain's multiplications and no memory traffic.
constexpr int CT_WINDOWS = 64, CT_ENTRIES = 8;
constexpr size_t CT_TABLE_WORDS = (size_t)CT_WINDOWS * CT_ENTRIES * NIELS_WORDS;       // T[pos][e-1] = e * 16^pos * B, e = 1..8: 64 KiB
// acc += s * B; `table` = B's CT table (in LDS on the device: SecretFb::stage)
ACT_HD ge fixed_base_acc_ct(ge acc, const uint32_t* table, const sc& s) {
  uint32_t t[8];
  radix16_bias(t, s);                                  // nibble - 8 = signed digit in [-8, 7]
#pragma unroll 1
  for (int wd = 0; wd < 8; wd++) {
    uint32_t word = t[0];
    for (int i = 0; i < 7; i++) t[i] = t[i + 1];       // static indices only: the digit words never live in scratch
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
      const uint32_t nib = word & 15u; word >>= 4;
      const bool neg = nib < 8u;
      const uint32_t mag = neg ? 8u - nib : nib - 8u;  // 0..8
      const uint32_t* row = table + (size_t)(wd * 8 + k) * CT_ENTRIES * NIELS_WORDS;
      ge_niels q = ge_niels_identity();
      for (uint32_t e = 1; e <= (uint32_t)CT_ENTRIES; e++) {
        const ge_niels c = niels_load(row + (size_t)(e - 1) * NIELS_WORDS);
        const uint32_t m = fe_mask(e == mag);
        q.ypx = fe_select_m(q.ypx, c.ypx, m); q.ymx = fe_select_m(q.ymx, c.ymx, m); q.xy2d = fe_select_m(q.xy2d, c.xy2d, m);
      }
      acc = ge_madd(acc, ge_niels_cneg(q, neg));
    }
  }
  return acc;
}
// ---- the same product with the table look-up on the MATRIX CORES (device only) ---------------------------------------------------
// "Every lane picks entry |digit_l| of a window" is a matrix product with a shared operand, selected = Table^T x onehot(digits), and
// the digit never becomes an address: D[byte][lane] = sum_e T[e][byte] * (e == |digit_lane|), one v_mfma_i32_32x32x32_i8 per 32 bytes
// x 32 lanes x 32 entries (more entries: chain MFMAs through the accumulator).  Byte values arrive exactly (one nonzero term per
// sum; a byte >= 128 comes out as value - 256, whose low byte is the value).  So a window can hold 64 entries (signed radix 128: 37
// windows instead of the scan's 64) for LESS than the scan pays for 8: measured on
// MI355X (tools/ubench_mfma_select.hip, profiles/r04_ubench_mfma_select.txt) 0.56 / 0.66 / 0.94 us of SIMD time per look-up among
// 32 / 64 / 128 entries against 0.70 / 1.27 / 2.37 us for a masked scan of 8 / 16 / 32 entries from LDS; a mixed addition is ~2 us.
// Any consistent numbering of k serves (A and B use the same one); the C/D map is col = lane & 31, row = (reg & 3) + 8 (reg >> 2) +
// 4 (lane >> 5), so after a half exchange (v_permlane32_swap) every lane holds its own entry.  Table image (k_build_table_mf): per
// window [K-step] 4 tiles of 32 bytes x 2 lane halves x 32 rows x 16 entries = 4 KiB per K-step, the 108 bytes of entry e >= 1 at
// K-slot e - 1; 296 KiB per base at 64 entries, read with wave-uniform coalesced 16-byte loads (L2 resident).
// EVERY lane of the wavefront must call this together (the table operand's rows come from all 64 lanes): non-live lanes pass s = 0.
// MF_KSTEPS chained MFMAs per tile = 32 * MF_KSTEPS entries per window = windows of 5 + log2(MF_KSTEPS) + 1 bits.  Measured in the
// kernels (two wavefronts per SIMD, where a chained accumulate's latency is NOT hidden as it is in the micro-benchmark's eight), ct
// prove_spend on one MI355X: 32 entries (43 windows) 1.118 M/s, 64 entries (37 windows) 1.162 M/s on the same box
// (profiles/r04_j_ct_mf_entries_ab.txt); 128 entries (32 windows) 1.02 M/s against 1.13 M/s on another (r04_i_ / r04_h_other_configs_1gpu_ct.json).
#ifndef ACT_MF_KSTEPS
#define ACT_MF_KSTEPS 2
#endif
constexpr int MF_KSTEPS = ACT_MF_KSTEPS, MF_ENTRIES = 32 * MF_KSTEPS;
constexpr int MF_WBITS = MF_KSTEPS == 1 ? 6 : MF_KSTEPS == 2 ? 7 : 8, MF_WINDOWS = (253 + MF_WBITS - 1) / MF_WBITS;
constexpr int MF_WINDOW_BYTES = MF_KSTEPS * 4 * 2 * 32 * 16;
static_assert(MF_KSTEPS == 1 || MF_KSTEPS == 2 || MF_KSTEPS == 4, "32, 64 or 128 entries per window");
#if defined(__HIP_DEVICE_COMPILE__)
typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ mf_v4i mf_onehot16(int idx, int h) {      // byte j of the 16 is 1 iff idx == 16 h + j (none outside [0, 32))
  const int rel = idx - 16 * h;
  const int val = (int)(1u << (8 * (rel & 3))), w = rel >> 2;
  mf_v4i b;
  b[0] = (w == 0) ? val : 0; b[1] = (w == 1) ? val : 0; b[2] = (w == 2) ? val : 0; b[3] = (w == 3) ? val : 0;
  return b;
}
__device__ __forceinline__ uint32_t mf_pack4(const mf_v16i& v, int q) {   // the low bytes of v[4q .. 4q+3] as one little-endian word
  return __builtin_amdgcn_perm((uint32_t)v[4 * q + 1], (uint32_t)v[4 * q], 0x0c0c0400u) |
         (__builtin_amdgcn_perm((uint32_t)v[4 * q + 3], (uint32_t)v[4 * q + 2], 0x0c0c0400u) << 16);
}
// w[0 .. 26] = the Niels words of entry idx + 1 of the window whose image starts at tabA (all zero for idx < 0)
__device__ __forceinline__ void mf_select(uint32_t w[27], const uint8_t* tabA, int idx) {
  const int lane = (int)(threadIdx.x & 63u), h = lane >> 5, r = lane & 31;
  const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);      // digits of lanes 0..31 resp. 32..63
  mf_v4i b_lo[MF_KSTEPS], b_hi[MF_KSTEPS];
#pragma unroll
  for (int ks = 0; ks < MF_KSTEPS; ks++) { b_lo[ks] = mf_onehot16((int)sw[0] - 32 * ks, h); b_hi[ks] = mf_onehot16((int)sw[1] - 32 * ks, h); }
  const mf_v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; t++) {
    mf_v16i x = zero, y = zero;
#pragma unroll
    for (int ks = 0; ks < MF_KSTEPS; ks++) {
      const mf_v4i a = *reinterpret_cast<const mf_v4i*>(tabA + (((ks * 4 + t) * 2 + h) * 32 + r) * 16);
      x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_lo[ks], x, 0, 0, 0);      // columns = lanes 0..31's entries
      y = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_hi[ks], y, 0, 0, 0);      // columns = lanes 32..63's entries
    }
    // lane l < 32 owns column l of x, lane l >= 32 column l - 32 of y, and each holds half the rows of both: pack the bytes of four
    // rows into a word first, then trade words -- bytes 8q .. 8q+3 of the tile come from lane half 0, 8q+4 .. 8q+7 from half 1
    const int nq = t < 3 ? 4 : 2;                                   // the last tile holds bytes 96 .. 107 only
#pragma unroll
    for (int q = 0; q < nq; q++) {
      const bool need_hi = t < 3 || q == 0;                         // word 8t + 2q + 1 < 27
      uint32_t xw = mf_pack4(x, q), yw = mf_pack4(y, q);
      const auto s = __builtin_amdgcn_permlane32_swap(xw, yw, false, false);
      w[8 * t + 2 * q] = s[0];
      if (need_hi) w[8 * t + 2 * q + 1] = s[1];
    }
  }
}
// word i of sum_w 2^(MF_WBITS - 1) * 2^(MF_WBITS * w): added to a scalar, MF_WBITS-bit group w of the sum, minus 2^(MF_WBITS - 1), is
// signed digit w in [-2^(MF_WBITS-1), 2^(MF_WBITS-1) - 1] (the sum stays below 2^(MF_WBITS * MF_WINDOWS) <= 2^259: nine words)
__device__ __forceinline__ constexpr uint32_t mf_bias_word(int i) {
  uint32_t v = 0;
  for (int w = 0; w < MF_WINDOWS; w++) { const int bit = MF_WBITS * w + MF_WBITS - 1; if (bit / 32 == i) v |= 1u << (bit % 32); }
  return v;
}
// acc += s * B through B's matrix-core table image; s canonical (< l)
__device__ __forceinline__ ge fixed_base_acc_mf(ge acc, const uint8_t* tab_mf, const sc& s) {
  uint32_t t[9];
  uint64_t c = 0;
  constexpr uint32_t bias[9] = {mf_bias_word(0), mf_bias_word(1), mf_bias_word(2), mf_bias_word(3), mf_bias_word(4), mf_bias_word(5), mf_bias_word(6), mf_bias_word(7), mf_bias_word(8)};
  for (int i = 0; i < 8; i++) { c += (uint64_t)s.v[i] + bias[i]; t[i] = (uint32_t)c; c >>= 32; }
  t[8] = (uint32_t)c + bias[8];
#pragma unroll 1
  for (int wd = 0; wd < MF_WINDOWS; wd++) {
    const int d = (int)(t[0] & ((1u << MF_WBITS) - 1u)) - (1 << (MF_WBITS - 1));
    for (int i = 0; i < 8; i++) t[i] = (t[i] >> MF_WBITS) | (t[i + 1] << (32 - MF_WBITS));      // static indices only: the digit words never live in scratch
    t[8] >>= MF_WBITS;
    const bool neg = d < 0;
    const int mag = neg ? -d : d;                                  // 0 .. MF_ENTRIES
    uint32_t w[27];
    mf_select(w, tab_mf + (size_t)wd * MF_WINDOW_BYTES, mag - 1);
    const uint32_t one = mag == 0 ? 1u : 0u;                       // digit 0: the identity (ypx = ymx = 1, xy2d = 0)
    ge_niels q;
    for (int i = 0; i < FE_LIMBS; i++) { q.ypx.v[i] = w[i]; q.ymx.v[i] = w[FE_LIMBS + i]; q.xy2d.v[i] = w[2 * FE_LIMBS + i]; }
    q.ypx.v[0] |= one; q.ymx.v[0] |= one;
    acc = ge_madd(acc, ge_niels_cneg(q, neg));
  }
  return acc;
}
#elif defined(__HIPCC__)
__device__ ge fixed_base_acc_mf(ge acc, const uint8_t* tab_mf, const sc& s);      // hipcc's host pass only parses the callers
#endif

// acc[a] += s[a] * N with no secret-dependent address and no secret-dependent branch: chain<NACC> with every addition executed
// `steps` < 127: the scalars are known to fit 2 (steps - 1) bits (the last step takes the recoding's carry): chain_ct_piece below
template <int NACC>
ACT_HD void chain_ct(ge* acc, const ge& N, const sc* s, int steps = 127) {
  uint32_t w[NACC][8], carry[NACC];
  for (int a = 0; a < NACC; a++) { carry[a] = 0; for (int i = 0; i < 8; i++) w[a][i] = s[a].v[i]; }
  ge_cached idc; idc.YpX = fe_one(); idc.YmX = fe_one(); idc.Z = fe_one(); idc.T2d = fe_zero();
  ge P = N;
#pragma unroll 1
  for (int step = 0; step < steps; step++) {
    ge_cached c1 = ge_to_cached(P);
    ge Q = ge_double(P);
    ge_cached c2 = ge_to_cached(Q);
    if (step < steps - 1) P = ge_double(Q);
#pragma unroll
    for (int a = 0; a < NACC; a++) {
      digit4 d = next_digit4(w[a], carry[a]);
      const uint32_t m2 = fe_mask(d.two), mz = fe_mask(!d.nonzero);
      ge_cached q;
      q.YpX = fe_select_m(c1.YpX, c2.YpX, m2); q.YmX = fe_select_m(c1.YmX, c2.YmX, m2);
      q.Z = fe_select_m(c1.Z, c2.Z, m2); q.T2d = fe_select_m(c1.T2d, c2.T2d, m2);
      q = ge_cached_cneg(q, d.neg);
      q.YpX = fe_select_m(q.YpX, idc.YpX, mz); q.YmX = fe_select_m(q.YmX, idc.YmX, mz);
      q.Z = fe_select_m(q.Z, idc.Z, mz); q.T2d = fe_select_m(q.T2d, idc.T2d, mz);
      acc[a] = ge_add_cached(acc[a], q);
    }
  }
}
// One QUARTER of s * N, for latency: s = sum_i 2^(64 i) s_i with 64-bit s_i, so s N = sum_i s_i (2^(64 i) N) and the four terms are
// independent -- four wavefronts each double N 64 i times and then run 33 chain steps instead of one running 127 (the longest does
// 192 doublings + 33 steps: ~0.55 of the whole chain's multiplications).  Same group element once the four are added; still no
// secret-dependent address or branch.  For single-item calls, where the chip is empty and the chain IS the answer time.
ACT_HD ge chain_ct_quarter(const ge& N, const sc& s, int i) {
  ge P = N;
#pragma unroll 1
  for (int d = 0; d < 64 * i; d++) P = ge_double(P);
  sc piece = sc_zero();
  piece.v[0] = s.v[2 * i]; piece.v[1] = s.v[2 * i + 1];
  ge acc[1] = {ge_identity()};
  sc sp[1] = {piece};
  chain_ct<1>(acc, P, sp, 33);
  return acc[0];
}
#if defined(ACT_CT_SECRET_TABLES)
template <int NACC> ACT_HD void chain_s(ge* acc, const ge& N, const sc* s, uint32_t*) { chain_ct<NACC>(acc, N, s); }
#else
template <int NACC> ACT_HD void chain_s(ge* acc, const ge& N, const sc* s, uint32_t* bk) { chain_b<NACC>(acc, N, s, bk); }
#endif

// ---- batched double-and-compress (ge25519.h dc_*): one field inversion per E encodings ----------------------
// Encodes 2*Q_i for `count` <= E points.  `slot(i)` -> the GE_WORDS words of point i (X|Y|Z|T); they are overwritten with
// e|f|g|h between the two passes.  `emit(i, words)` receives the encodings, last point first.
template <int E, typename Slot, typename Emit>
ACT_HD void dc_encode_batch(int count, Slot slot, Emit emit) {
  fe prefix[E];
  fe acc = fe_one();
#pragma unroll
  for (int i = 0; i < E; i++) {
    if (i < count) {
      uint32_t* w = slot(i);
      dc_efgh s = dc_prepare(bucket_load(w));
      ge t; t.X = s.e; t.Y = s.f; t.Z = s.g; t.T = s.h; bucket_store(w, t);
      fe eg, fh; bool z;
      fe prod = dc_product(eg, fh, z, s);
      prefix[i] = acc;
      acc = fe_mul(acc, prod);
    }
  }
  fe inv = fe_invert(acc);
#pragma unroll
  for (int i = E - 1; i >= 0; i--) {
    if (i < count) {
      ge t = bucket_load(slot(i));
      dc_efgh s; s.e = t.X; s.f = t.Y; s.g = t.Z; s.h = t.T;
      fe eg, fh; bool z;
      fe prod = dc_product(eg, fh, z, s);
      fe mine = fe_select(fe_mul(inv, prefix[i]), fe_zero(), z);
      inv = fe_mul(inv, prod);
      uint32_t enc[8];
      dc_finish(enc, s, eg, fh, mine);
      emit(i, enc);
    }
  }
}

}  // namespace act
