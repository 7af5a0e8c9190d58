// hostcheck.cpp — TEST-ONLY host build of the device arithmetic headers (csrc/fe25519.h,
// sc25519.h, ge25519.h, msm.h, blake3_hd.h) so `pytest -m "not gpu"` can check the formulas,
// the limb-size discipline (ACT_FE_BOUNDS) and the BLAKE3 code against the oracle without a GPU.
// Never linked into libact_mi355x.so; the product has no CPU compute path.
#include <cstring>
#include <ctime>
#include <vector>
#define ACT_FE_BOUNDS 1
#define ACT_FB_WBITS 6   /* host test of the window logic: small tables (43 windows x 64 entries) */
#include "../../anonymous-credit-tokens_amd/csrc/spend_lanes.h"

namespace act { fe_bounds_t fe_bounds = {0, 0, 0, 0, 0, 0}; fe_counts_t fe_counts = {0, 0, {0, 0, 0, 0}}; }
using namespace act;

static void ld(uint32_t w[8], const uint8_t* b) { memcpy(w, b, 32); }
static void st(uint8_t* b, const uint32_t w[8]) { memcpy(b, w, 32); }
static fe fe_in(const uint8_t* b) { uint32_t w[8]; ld(w, b); return fe_from_words(w); }
static void fe_out(uint8_t* b, const fe& f) { uint32_t w[8]; fe_to_words(w, f); st(b, w); }
static sc sc_in(const uint8_t* b) { uint32_t w[8]; ld(w, b); return sc_from_words(w); }
static void sc_out(uint8_t* b, const sc& s) { memcpy(b, s.v, 32); }

extern "C" {
void hc_bounds(uint64_t out[6]) { memcpy(out, &fe_bounds, sizeof(fe_bounds)); }
void hc_bounds_reset() { memset(&fe_bounds, 0, sizeof(fe_bounds)); }

void hc_fe_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) { fe_out(o, fe_mul(fe_in(a), fe_in(b))); }
void hc_fe_sq(const uint8_t* a, uint8_t* o) { fe_out(o, fe_sq(fe_in(a))); }
void hc_fe_add(const uint8_t* a, const uint8_t* b, uint8_t* o) { fe_out(o, fe_add(fe_in(a), fe_in(b))); }
void hc_fe_sub(const uint8_t* a, const uint8_t* b, uint8_t* o) { fe_out(o, fe_sub(fe_in(a), fe_in(b))); }
void hc_fe_invert(const uint8_t* a, uint8_t* o) { fe_out(o, fe_invert(fe_in(a))); }
int hc_fe_invsqrt(const uint8_t* a, uint8_t* o) { fe r; bool ok = fe_invsqrt(r, fe_in(a)); fe_out(o, r); return ok; }
int hc_fe_sqrt_ratio(const uint8_t* u, const uint8_t* v, uint8_t* o) { fe r; bool ok = fe_sqrt_ratio_m1(r, fe_in(u), fe_in(v)); fe_out(o, r); return ok; }
// stress: a chain of lazily-reduced operations at the extreme of the allowed operand classes
void hc_fe_stress(const uint8_t* a, const uint8_t* b, uint8_t* o) {
  fe x = fe_in(a), y = fe_in(b);
  fe s = fe_sub(x, y), t = fe_add(x, y);                // {3}, {2}
  fe f = fe_sub(fe_dbl(x), y);                          // {4}: 2x - y
  fe r = fe_mul(f, s);                                  // (2x - y)(x - y): the whole product budget, 4 * 3
  r = fe_mul(fe_sub4(fe_dbl(r), t), fe_sq(s));          // (2r - (x+y)) (x-y)^2: {6} * tight, square of a {3}
  fe_out(o, r);
}

// products / squares / carries of operands given limb by limb (9 words each): the extremes of the column budget, which no
// 32-byte input reaches.  out = canonical bytes; the operands' limb sizes go through the same ACT_FE_BOUNDS bookkeeping.
void hc_fe_mul_limbs(const uint32_t* a, const uint32_t* b, uint8_t* o) { fe x, y; for (int i = 0; i < FE_LIMBS; i++) { x.v[i] = a[i]; y.v[i] = b[i]; } fe_out(o, fe_mul(x, y)); }
void hc_fe_sq_limbs(const uint32_t* a, uint8_t* o) { fe x; for (int i = 0; i < FE_LIMBS; i++) x.v[i] = a[i]; fe_out(o, fe_sq(x)); }
void hc_fe_carry_limbs(const uint32_t* a, uint32_t* limbs_out, uint8_t* o) { fe x; for (int i = 0; i < FE_LIMBS; i++) x.v[i] = a[i]; fe c = fe_carry(x); for (int i = 0; i < FE_LIMBS; i++) limbs_out[i] = c.v[i]; fe_out(o, c); }
void hc_fe_mul_out_limbs(const uint32_t* a, const uint32_t* b, uint32_t* limbs_out) { fe x, y; for (int i = 0; i < FE_LIMBS; i++) { x.v[i] = a[i]; y.v[i] = b[i]; } fe c = fe_mul(x, y); for (int i = 0; i < FE_LIMBS; i++) limbs_out[i] = c.v[i]; }

void hc_sc_reduce_wide(const uint8_t* a, uint8_t* o) { uint32_t w[16]; memcpy(w, a, 64); sc_out(o, sc_from_wide_words(w)); }
void hc_sc_from_bytes(const uint8_t* a, uint8_t* o) { sc_out(o, sc_in(a)); }
void hc_sc_muladd(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* o) { sc_out(o, sc_muladd(sc_in(a), sc_in(b), sc_in(c))); }
void hc_sc_sub(const uint8_t* a, const uint8_t* b, uint8_t* o) { sc_out(o, sc_sub(sc_in(a), sc_in(b))); }
void hc_sc_neg(const uint8_t* a, uint8_t* o) { sc_out(o, sc_neg(sc_in(a))); }
void hc_sc_half(const uint8_t* a, uint8_t* o) { sc_out(o, sc_half(sc_in(a))); }
void hc_sc_invert(const uint8_t* a, uint8_t* o) { sc_out(o, sc_invert(sc_in(a))); }

int hc_decode_encode(const uint8_t* a, uint8_t* o) {
  uint32_t w[8], r[8]; ld(w, a); ge p; bool ok = ristretto_decode(p, w);
  if (ok) { ristretto_encode(r, p); st(o, r); } else memset(o, 0, 32);
  return ok;
}
// Marshalling bound of a struct-level binding: seconds for `reps` passes of decoding / encoding `count`
// points on this core -- what RistrettoPoint::decompress / compress cost a host that builds 130 of them per SpendProof.
int hc_time_point_codec(const uint8_t* enc, int count, int reps, double* s_decode, double* s_encode, uint8_t* out_check) {
  std::vector<ge> pts(count);
  uint32_t w[8], r[8], acc[8] = {0};
  auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; };
  double t0 = now();
  int ok = 1;
  for (int k = 0; k < reps; k++) for (int i = 0; i < count; i++) { ld(w, enc + 32 * i); ok &= (int)ristretto_decode(pts[i], w); }
  double t1 = now();
  for (int k = 0; k < reps; k++) for (int i = 0; i < count; i++) { ristretto_encode(r, pts[i]); for (int j = 0; j < 8; j++) acc[j] ^= r[j] + (uint32_t)k; }
  double t2 = now();
  *s_decode = t1 - t0; *s_encode = t2 - t1;
  st(out_check, acc);       // keeps the encodings alive
  return ok;
}
void hc_from_uniform(const uint8_t* a, uint8_t* o) { uint32_t w[16], r[8]; memcpy(w, a, 64); ge p = ristretto_from_uniform(w); ristretto_encode(r, p); st(o, r); }
// out = s0*P (chain<1>), and (s0*P, s1*P) via chain<2>; also s0*P + s1*Q through add/sub/double mixes
int hc_chain2(const uint8_t* pt, const uint8_t* s0, const uint8_t* s1, uint8_t* o0, uint8_t* o1) {
  uint32_t w[8], r[8]; ld(w, pt); ge p; if (!ristretto_decode(p, w)) return 0;
  ge acc[2] = {ge_identity(), ge_identity()}; sc s[2] = {sc_in(s0), sc_in(s1)};
  chain<2>(acc, p, s);
  ristretto_encode(r, acc[0]); st(o0, r); ristretto_encode(r, acc[1]); st(o1, r);
  return 1;
}
int hc_chain1(const uint8_t* pt, const uint8_t* s0, uint8_t* o0) {
  uint32_t w[8], r[8]; ld(w, pt); ge p; if (!ristretto_decode(p, w)) return 0;
  ge acc[1] = {ge_identity()}; sc s[1] = {sc_in(s0)};
  chain<1>(acc, p, s);
  ristretto_encode(r, acc[0]); st(o0, r);
  return 1;
}
// builds the position-specific table of `pt` on the host (same recurrence as k_build_table) and evaluates s*pt
int hc_fixed_base(const uint8_t* pt, const uint8_t* s, uint8_t* o) {
  uint32_t w[8], r[8]; ld(w, pt); ge b; if (!ristretto_decode(b, w)) return 0;
  std::vector<uint32_t> tab((size_t)FB_TABLE_WORDS);
  ge base = b;
  for (int pos = 0; pos < FB_WINDOWS; pos++) {
    ge acc = ge_identity();
    for (int e = 0; e < FB_ENTRIES; e++) {
      // affine-normalise acc
      fe zi = fe_invert(acc.Z); ge a; a.X = fe_mul(acc.X, zi); a.Y = fe_mul(acc.Y, zi); a.Z = fe_one(); a.T = fe_mul(a.X, a.Y);
      niels_store(&tab[((size_t)pos * FB_ENTRIES + e) * NIELS_WORDS], niels_from_affine(a));
      acc = ge_add(acc, base);
    }
    base = acc;   // 2^w * base
  }
  ge acc = fixed_base_acc(ge_identity(), FbTab{tab.data(), (uint32_t)FB_WBITS, 0u}, sc_in(s));
  ristretto_encode(r, acc); st(o, r);
  return 1;
}
int hc_add_sub_dbl(const uint8_t* a, const uint8_t* b, uint8_t* o_add, uint8_t* o_sub, uint8_t* o_dbl) {
  uint32_t w[8], r[8]; ge p, q; ld(w, a); if (!ristretto_decode(p, w)) return 0; ld(w, b); if (!ristretto_decode(q, w)) return 0;
  ristretto_encode(r, ge_add(p, q)); st(o_add, r);
  ristretto_encode(r, ge_sub(p, q)); st(o_sub, r);
  ristretto_encode(r, ge_double(p)); st(o_dbl, r);
  return 1;
}
void hc_blake3_xof64(const uint8_t* msg, uint32_t len, uint8_t* out) {
  std::vector<uint32_t> w((len + 3) / 4 + 1, 0u); if (len) memcpy(w.data(), msg, len);
  uint32_t o[16]; b3_hash_xof64(o, w.data(), len); memcpy(out, o, 64);
}
}
// the chunk-parallel form of k_hash_xof_par: every chunk's chaining value first (any order), then the one-lane fold reading them
extern "C" void hc_blake3_xof64_par(const uint8_t* msg, uint32_t len, uint8_t* out) {
  std::vector<uint32_t> w((len + 3) / 4 + 1, 0u); if (len) memcpy(w.data(), msg, len);
  const uint32_t nchunks = len ? (len + 1023u) >> 10 : 1u;
  std::vector<uint32_t> cvs((size_t)nchunks * 8, 0u);
  for (uint32_t c = nchunks; c-- > 0;) if (c + 1 < nchunks) b3_chunk_cv(&cvs[(size_t)c * 8], w.data(), len, c);
  uint32_t o[16];
  b3_hash_xof64_with(o, w.data(), len, [&](uint32_t c, uint32_t* cv) { memcpy(cv, &cvs[(size_t)c * 8], 32); });
  memcpy(out, o, 64);
}
extern "C" int hc_chain2u(const uint8_t* pt, const uint8_t* s0, const uint8_t* s1, uint8_t* o0, uint8_t* o1) {
  uint32_t w[8], r[8]; ld(w, pt); ge p; if (!ristretto_decode(p, w)) return 0;
  ge al = ge_identity(), au = ge_identity();
  chain2u(al, au, p, sc_in(s0), sc_in(s1));
  ristretto_encode(r, al); st(o0, r); ristretto_encode(r, au); st(o1, r);
  return 1;
}
// batched double-and-compress: Q_i = a_i + b_i (projective, Z != 1) for i < count <= 8; out_i must be enc(2 Q_i)
extern "C" int hc_dc_encode_batch(const uint8_t* a_enc, const uint8_t* b_enc, int count, uint8_t* out) {
  std::vector<uint32_t> slots((size_t)8 * GE_WORDS);
  for (int i = 0; i < count; i++) {
    uint32_t w[8]; ge pa, pb;
    ld(w, a_enc + 32 * i); if (!ristretto_decode(pa, w)) return 0;
    ld(w, b_enc + 32 * i); if (!ristretto_decode(pb, w)) return 0;
    ge_store(slots.data() + (size_t)i * GE_WORDS, ge_add(pa, pb));
  }
  dc_encode_batch<8>(count, [&](int i) { return slots.data() + (size_t)i * GE_WORDS; },
                     [&](int i, const uint32_t* enc) { st(out + 32 * i, enc); });
  return 1;
}
extern "C" int hc_chain_b2(const uint8_t* pt, const uint8_t* s0, const uint8_t* s1, uint8_t* o0, uint8_t* o1) {
  uint32_t w[8], r[8]; ld(w, pt); ge p; if (!ristretto_decode(p, w)) return 0;
  ge acc[2] = {ge_identity(), ge_identity()};
  sc s[2] = {sc_in(s0), sc_in(s1)};
  std::vector<uint32_t> bk((size_t)2 * BUCKET_WORDS);
  chain_b<2>(acc, p, s, bk.data());
  ristretto_encode(r, acc[0]); st(o0, r); ristretto_encode(r, acc[1]); st(o1, r);
  return 1;
}
// the address-free form the ct build uses for secret scalars (msm.h chain_ct: every digit addition executed, no buckets)
extern "C" int hc_chain_ct2(const uint8_t* pt, const uint8_t* s0, const uint8_t* s1, uint8_t* o0, uint8_t* o1) {
  uint32_t w[8], r[8]; ld(w, pt); ge p; if (!ristretto_decode(p, w)) return 0;
  ge acc[2] = {ge_identity(), ge_identity()};
  sc s[2] = {sc_in(s0), sc_in(s1)};
  chain_ct<2>(acc, p, s);
  ristretto_encode(r, acc[0]); st(o0, r); ristretto_encode(r, acc[1]); st(o1, r);
  return 1;
}
// fixed_base_acc_ct over a CT table of `pt` built here (T[pos][e-1] = e * 16^pos * pt, the recurrence of k_build_table_ct)
extern "C" int hc_fixed_base_ct(const uint8_t* pt, const uint8_t* s0, uint8_t* o0) {
  uint32_t w[8], r[8]; ld(w, pt); ge b; if (!ristretto_decode(b, w)) return 0;
  std::vector<uint32_t> tab(CT_TABLE_WORDS);
  for (int pos = 0; pos < CT_WINDOWS; pos++) {
    ge acc = ge_identity();
    const ge_cached bc = ge_to_cached(b);
    for (int e = 1; e <= CT_ENTRIES; e++) {
      acc = ge_add_cached(acc, bc);
      fe zi = fe_invert(acc.Z);
      ge af; af.X = fe_mul(acc.X, zi); af.Y = fe_mul(acc.Y, zi); af.Z = fe_one(); af.T = fe_mul(af.X, af.Y);
      niels_store(tab.data() + ((size_t)pos * CT_ENTRIES + (e - 1)) * NIELS_WORDS, niels_from_affine(af));
    }
    for (int i = 0; i < 4; i++) b = ge_double(b);
  }
  ge out = fixed_base_acc_ct(ge_identity(), tab.data(), sc_in(s0));
  ristretto_encode(r, out); st(o0, r);
  return 1;
}
extern "C" int hc_chain_bu(const uint8_t* pt, const uint8_t* s0, const uint8_t* s1, uint8_t* o0, uint8_t* o1) {
  uint32_t w[8], r[8]; ld(w, pt); ge p; if (!ristretto_decode(p, w)) return 0;
  ge al = ge_identity(), au = ge_identity();
  std::vector<uint32_t> bk(BUCKET_WORDS);
  chain_bu(al, au, p, sc_in(s0), sc_in(s1), bk.data());
  ristretto_encode(r, al); st(o0, r); ristretto_encode(r, au); st(o1, r);
  return 1;
}

// the digit-strings-from-memory form the range kernel runs (msm.h chain_bu_pre: d-free additions inside), called the way
// spend_lanes.h spend_bits_lane calls it
extern "C" int hc_chain_bu_pre(const uint8_t* pt, const uint8_t* s0, const uint8_t* s1, uint8_t* o0, uint8_t* o1) {
  uint32_t w[8], r[8]; ld(w, pt); ge p; if (!ristretto_decode(p, w)) return 0;
  ge al = ge_identity(), au = ge_identity();
  std::vector<uint32_t> bk(BUCKET_WORDS), naf(NAF_WORDS);
  uint32_t dg[8]; radix16_bias(dg, sc_in(s0));
  sc su; memcpy(su.v, s1, 32);                                   // raw: values in [l, 2^253) reach the recoder unreduced
  naf3_recode(naf.data(), su);
  const bool n_small = (w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0u;
  chain_bu_pre<false>(al, au, p, dg, naf.data(), bk.data(), nullptr, n_small);
  ristretto_encode(r, al); st(o0, r); ristretto_encode(r, au); st(o1, r);
  return 1;
}
// p + q by the d-free formulas; returns 2 when the result is (0,0,0,0) (q - p in E[4]), 1 otherwise with the encoding
extern "C" int hc_add_ded(const uint8_t* a, const uint8_t* b, int b_plus_order2, uint8_t* o) {
  uint32_t w[8], r[8]; ge p, q; ld(w, a); if (!ristretto_decode(p, w)) return 0; ld(w, b); if (!ristretto_decode(q, w)) return 0;
  if (b_plus_order2) { q.X = fe_carry(fe_neg(q.X)); q.Y = fe_carry(fe_neg(q.Y)); }       // q + (0, -1) = (-x, -y)
  ge s = ge_add_ded(p, ge_to_ded(q));
  if (fe_is_zero(s.X) && fe_is_zero(s.Y) && fe_is_zero(s.Z) && fe_is_zero(s.T)) return 2;
  ristretto_encode(r, s); st(o, r);
  return 1;
}

// ---- the spend-verification kernels' own lane bodies (csrc/spend_lanes.h), run lane by lane on the host ---------------
// Builds Params tables the way k_build_table does, then executes exactly what the five kernels execute for `n` proofs:
// prep (lane = proof), bits (lane = (proof, bit)), enc (lane = 32 half-points), tail, BLAKE3, finish.  Outputs: the
// "spend" transcript pre-images, statuses, enc(K'), and the number of field multiplications / squarings each kernel's
// lanes executed: counts[6*k .. 6*k+5] = fe_mul, fe_sq, fixed_base_acc calls on g, h1, h2, h3 for k = prep, bits, enc, tail.
// This build's fixed-base windows are ACT_FB_WBITS = 6 bits wide (43 mixed additions of 7 multiplications per call, tables
// small enough to build per test); counts[24] = that window count, so the caller can restate the multiplications for the
// product's windows exactly: mul - sum_b calls_b * (counts[24] - product_windows_b) * 7.
namespace {
struct HostTables { uint8_t h[96]; std::vector<uint32_t> tab[4]; bool valid = false; };
HostTables g_tabs;
const uint8_t kGen[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                          0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
bool build_tables(const uint8_t h[96]) {
  if (g_tabs.valid && memcmp(g_tabs.h, h, 96) == 0) return true;
  for (int b = 0; b < 4; b++) {
    uint32_t w[8]; ld(w, b == 0 ? kGen : h + 32 * (b - 1));
    ge base; if (!ristretto_decode(base, w)) return false;
    g_tabs.tab[b].assign((size_t)FB_TABLE_WORDS, 0u);
    for (int pos = 0; pos < FB_WINDOWS; pos++) {
      ge acc = ge_identity();
      for (int e = 0; e < FB_ENTRIES; e++) {
        fe zi = fe_invert(acc.Z); ge a; a.X = fe_mul(acc.X, zi); a.Y = fe_mul(acc.Y, zi); a.Z = fe_one(); a.T = fe_mul(a.X, a.Y);
        niels_store(&g_tabs.tab[b][((size_t)pos * FB_ENTRIES + e) * NIELS_WORDS], niels_from_affine(a));
        acc = ge_add(acc, base);
      }
      base = acc;
    }
  }
  memcpy(g_tabs.h, h, 96); g_tabs.valid = true;
  return true;
}
void put_lp(std::vector<uint8_t>& v, const uint8_t* b, size_t n) { for (int i = 7; i >= 0; i--) v.push_back((uint8_t)((uint64_t)n >> (8 * i))); v.insert(v.end(), b, b + n); }
}  // namespace

static int spend_verify_impl(const uint8_t* h, int L, const uint8_t* sk, uint32_t n, const uint8_t* proofs, uint8_t* out_transcripts,
                             uint8_t* out_status, uint8_t* out_kprime, uint64_t* counts, bool small_schedule) {
  if (L < 1 || L > 128 || !build_tables(h)) return 0;
  SpendArgs a{};
  for (int b = 0; b < 4; b++) a.P.tab[b] = FbTab{g_tabs.tab[b].data(), (uint32_t)FB_WBITS, (uint32_t)b};
  a.P.half_h1 = nullptr; a.P.L = L;
  static const char* const labels[4] = {"request", "respond", "spend", "refund"};
  static const char version[] = "curve25519-ristretto anonymous-credits v1.0";
  for (int l = 0; l < 4; l++) {
    std::vector<uint8_t> p; put_lp(p, (const uint8_t*)version, sizeof(version) - 1);
    put_lp(p, h, 32); put_lp(p, h + 32, 32); put_lp(p, h + 64, 32); put_lp(p, (const uint8_t*)labels[l], strlen(labels[l]));
    a.P.prefix_len[l] = (uint32_t)p.size(); p.resize(PREFIX_WORDS * 4, 0); memcpy(a.P.prefix[l], p.data(), PREFIX_WORDS * 4);
  }
  { uint32_t w[8]; ld(w, sk); a.K.x = sc_from_words(w); ld(w, sk + 32); if (!ristretto_decode(a.K.w, w)) return 0; }
  const SpendTranscript st{L}; const ProofLayout pl{L};
  std::vector<uint8_t> tr((size_t)n * st.stride(), 0), status(n, 0), kp((size_t)n * 32, 0);
  std::vector<uint32_t> coords((size_t)n * L * NIELS_WORDS), d01((size_t)n * 2 * GE_WORDS), buckets((size_t)n * (L < PREP_BUCKET_SETS ? PREP_BUCKET_SETS : L) * BUCKET_WORDS),
      xa((size_t)n * GE_WORDS), flags(n, 0), xof((size_t)n * 16), naf((size_t)n * NAF_WORDS), dig((size_t)n * L * 8);
  a.proofs = proofs; a.n = n; a.tr = tr.data(); a.tr_stride = (uint32_t)st.stride(); a.coords = coords.data(); a.d01 = d01.data();
  a.buckets = buckets.data(); a.xa = xa.data(); a.flags = flags.data(); a.xof = xof.data(); a.status = status.data(); a.kprime_enc = kp.data(); a.naf = naf.data(); a.dig = dig.data(); a.pbk = buckets.data();
  uint64_t c[25] = {0};
  c[24] = FB_WINDOWS;
  auto snap = [&](int k) {
    c[6 * k] += fe_counts.mul; c[6 * k + 1] += fe_counts.sq;
    for (int b = 0; b < 4; b++) c[6 * k + 2 + b] += fe_counts.fixed_base[b];
    fe_counts = fe_counts_t{0, 0, {0, 0, 0, 0}};
  };
  fe_counts = fe_counts_t{0, 0, {0, 0, 0, 0}};
  std::vector<uint32_t> small_scratch;
  if (!small_schedule) {
    for (uint32_t p = 0; p < n; p++) spend_prep_lane(a, p);
    snap(0);
    for (uint32_t g = 0; g < n * (uint32_t)L; g++) { if (L % 64 == 0) spend_bits_lane<true>(a, g, nullptr); else spend_bits_lane<false>(a, g, nullptr); }
    snap(1);
    for (uint64_t q0 = 0; q0 < (uint64_t)n * L * 2; q0 += ENC_BATCH) spend_enc_lane(a, q0);
    snap(2);
    for (uint32_t p = 0; p < n; p++) spend_tail_lane(a, p);
    snap(3);
  } else {
    // the small-batch schedule's kernels (engine.hip spend_small_locked) in an order the four streams allow that is as far from
    // the pipelined one as it gets: tail before bits, the roles of prep in reverse, their scratch in an area of its own
    small_scratch.assign((size_t)n * (PREP_BUCKET_SETS * BUCKET_WORDS + PART_POINTS * GE_WORDS), 0u);
    a.pbk = small_scratch.data(); a.part = small_scratch.data() + (size_t)n * PREP_BUCKET_SETS * BUCKET_WORDS;
    for (uint32_t g = 0; g < n * (uint32_t)L; g++) spend_coords_lane(a, g);
    for (uint32_t p = 0; p < n; p++) spend_tail_lane(a, p);
    snap(3);
    for (uint32_t p = 0; p < n; p++) spend_prep_c_lane(a, p);
    for (uint32_t p = 0; p < n; p++) spend_prep_b_lane(a, p);
    for (uint32_t p = 0; p < n; p++) spend_prep_a_lane(a, p);
    for (uint32_t p = 0; p < n; p++) spend_prep_join_lane(a, p);
    snap(0);
    for (uint32_t g = 0; g < n * (uint32_t)L; g++) { if (L % 64 == 0) spend_bits_lane<true>(a, g, nullptr); else spend_bits_lane<false>(a, g, nullptr); }
    snap(1);
    for (uint64_t q0 = 0; q0 < (uint64_t)n * L * 2; q0 += ENC_BATCH) spend_enc_lane(a, q0);
    snap(2);
  }
  for (uint32_t p = 0; p < n; p++) b3_hash_xof64(&xof[(size_t)p * 16], reinterpret_cast<const uint32_t*>(tr.data() + (size_t)p * st.stride()), (uint32_t)st.bytes());
  for (uint32_t p = 0; p < n; p++) spend_finish_lane(a, p);
  for (uint32_t p = 0; p < n; p++) memcpy(out_transcripts + (size_t)p * st.bytes(), tr.data() + (size_t)p * st.stride(), st.bytes());
  memcpy(out_status, status.data(), n); memcpy(out_kprime, kp.data(), (size_t)n * 32);
  if (counts) memcpy(counts, c, sizeof(c));
  return 1;
}
extern "C" int hc_spend_verify(const uint8_t* h, int L, const uint8_t* sk, uint32_t n, const uint8_t* proofs, uint8_t* out_transcripts,
                               uint8_t* out_status, uint8_t* out_kprime, uint64_t* counts) {
  return spend_verify_impl(h, L, sk, n, proofs, out_transcripts, out_status, out_kprime, counts, false);
}
extern "C" int hc_spend_verify_small(const uint8_t* h, int L, const uint8_t* sk, uint32_t n, const uint8_t* proofs, uint8_t* out_transcripts,
                                     uint8_t* out_status, uint8_t* out_kprime, uint64_t* counts) {
  return spend_verify_impl(h, L, sk, n, proofs, out_transcripts, out_status, out_kprime, counts, true);
}

// ---- the prover kernels' own lane bodies (csrc/prove_lanes.h), run lane by lane on the host -----------------------------
// CreditToken::prove_spend for `n` tokens exactly as k_prove_head / bits / enc / tail / (BLAKE3) / resp execute it: proofs,
// PreRefunds and statuses out, plus per-kernel operation counts in the layout of hc_spend_verify (counts[6*k ..] for k = head, bits,
// enc, tail; counts[24] = this build's window count).  tok == nullptr: synthetic tokens (a = the generator's encoding, e | k | r | c
// and the charges taken from the rng bytes) -- enough for counting, which does not depend on the values.
#include "../../anonymous-credit-tokens_amd/csrc/prove_lanes.h"
namespace {
struct HostFb {
  const DevParams& P;
  void stage(int) {}
  ge mul(const ge& acc, int base, const sc& s) const { return fixed_base_acc(acc, P.tab[base], s); }
};
}  // namespace
extern "C" int hc_prove_spend(const uint8_t* h, int L, uint32_t n, const uint8_t* tok, const uint8_t* s_amount, const uint8_t* rng, uint8_t* out_proofs,
                              uint8_t* out_prerefund, uint8_t* out_status, uint64_t* counts) {
  if (L < 1 || L > 128 || !build_tables(h)) return 0;
  ProveArgs a{};
  for (int b = 0; b < 4; b++) a.P.tab[b] = FbTab{g_tabs.tab[b].data(), (uint32_t)FB_WBITS, (uint32_t)b};
  a.P.L = L;
  static const char* const labels[4] = {"request", "respond", "spend", "refund"};
  static const char version[] = "curve25519-ristretto anonymous-credits v1.0";
  for (int l = 0; l < 4; l++) {
    std::vector<uint8_t> p; put_lp(p, (const uint8_t*)version, sizeof(version) - 1);
    put_lp(p, h, 32); put_lp(p, h + 32, 32); put_lp(p, h + 64, 32); put_lp(p, (const uint8_t*)labels[l], strlen(labels[l]));
    a.P.prefix_len[l] = (uint32_t)p.size(); p.resize(PREFIX_WORDS * 4, 0); memcpy(a.P.prefix[l], p.data(), PREFIX_WORDS * 4);
  }
  // the two-entry table of the bit term: identity, h1 / 2 (engine.hip launch_half_point_table)
  std::vector<uint32_t> half_h1((size_t)2 * NIELS_WORDS, 0u);
  {
    niels_store(half_h1.data(), ge_niels_identity());
    sc one = sc_zero(); one.v[0] = 1;
    ge hp = fixed_base_acc(ge_identity(), a.P.tab[BASE_H1], sc_half(one));
    fe zi = fe_invert(hp.Z); ge af; af.X = fe_mul(hp.X, zi); af.Y = fe_mul(hp.Y, zi); af.Z = fe_one(); af.T = fe_mul(af.X, af.Y);
    niels_store(half_h1.data() + NIELS_WORDS, niels_from_affine(af));
  }
  a.P.half_h1 = half_h1.data();
  std::vector<uint8_t> syn_tok, syn_s;
  const size_t rb = 64u * (4u * (size_t)L + 12u);
  if (!tok) {
    syn_tok.assign((size_t)n * 160, 0); syn_s.assign((size_t)n * 32, 0);
    for (uint32_t p = 0; p < n; p++) {
      memcpy(&syn_tok[(size_t)p * 160], kGen, 32);
      for (int f = 1; f < 5; f++) { memcpy(&syn_tok[(size_t)p * 160 + 32 * f], rng + p * rb + 64 * f, 31); }
      memcpy(&syn_s[(size_t)p * 32], rng + p * rb + 640, 15);
    }
    tok = syn_tok.data(); s_amount = syn_s.data();
  }
  const SpendTranscript st{L}; const ProofLayout pl{L};
  std::vector<uint8_t> tr((size_t)n * st.stride(), 0), status(n, 0);
  std::vector<uint32_t> d3((size_t)n * 3 * GE_WORDS), half((size_t)n * (L < 2 ? 2 : L) * BUCKET_WORDS), state((size_t)n * 24, 0), flags(n, 0), xof((size_t)n * 16);
  a.n = n; a.tok = tok; a.s = s_amount; a.rng = rng; a.tr = tr.data(); a.tr_stride = (uint32_t)st.stride(); a.d3 = d3.data(); a.half = half.data();
  a.state = state.data(); a.flags = flags.data(); a.xof = xof.data(); a.proof = out_proofs; a.prerefund = out_prerefund; a.status = status.data();
  uint64_t c[31] = {0};
  c[24] = FB_WINDOWS;
  auto snap = [&](int k) {
    c[6 * k] += fe_counts.mul; c[6 * k + 1] += fe_counts.sq;
    for (int b = 0; b < 4; b++) c[6 * k + 2 + b] += fe_counts.fixed_base[b];
    fe_counts = fe_counts_t{0, 0, {0, 0, 0, 0}};
  };
  fe_counts = fe_counts_t{0, 0, {0, 0, 0, 0}};
  HostFb fb{a.P};
  for (uint32_t p = 0; p < n; p++) prove_head_lane(a, p, fb);
  snap(0);
  for (uint32_t g = 0; g < n * (uint32_t)L; g++) prove_bits_lane(a, g, fb);
  snap(1);
  for (uint64_t q0 = 0; q0 < (uint64_t)n * L * 3; q0 += PROVE_ENC_BATCH) prove_enc_lane(a, q0);
  snap(2);
  for (uint32_t p = 0; p < n; p++) prove_tail_lane(a, p, fb);
  snap(3);
  for (uint32_t p = 0; p < n; p++) b3_hash_xof64(&xof[(size_t)p * 16], reinterpret_cast<const uint32_t*>(tr.data() + (size_t)p * st.stride()), (uint32_t)st.bytes());
  for (uint32_t g = 0; g < n * (uint32_t)L; g++) prove_resp_lane(a, g);
  memcpy(out_status, status.data(), n);
  if (counts) memcpy(counts, c, sizeof(c));
  return 1;
}
