// fe25519.h — GF(2^255-19) for gfx950 lanes: 10 unsigned limbs in radix 2^25.5 (26/25 bits
// alternating), one field element per lane, every column of a product one chain of
// v_mad_u64_u32 (measured on MI355X: one per 4.06 cycles per SIMD in blocks of dependent ones, the cost of any other
// VALU instruction in this mix, so the representation minimises instruction count, not multiplies).
//
// Replaces, for the device path, the field arithmetic the reference takes from curve25519-dalek
// 4.1.3 (not in /root/reference; call sites /root/reference/src/lib.rs:465-1239 via
// RistrettoPoint/Scalar ops).  Compiles under hipcc (device) and under g++ (tests/hostcheck only:
// formula and limb-bound unit tests — never a product code path).
//
// Limb-size discipline (checked by the ACT_FE_BOUNDS host build):
//   tight : even limbs <= 2^26 + 2^20, odd limbs <= 2^25 + 2^19   (output of mul / sq / carry)
//   mul(f, g) / sq(f): g (and f of sq) <= 1.68 * 2^27 even / 1.68 * 2^26 odd  (19*g must fit u32)
//                      f of mul       <= 1.5  * 2^28 even / 1.5  * 2^27 odd  (column sums < 2^64)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ACT_HD __host__ __device__ __forceinline__
#define ACT_D __device__ __forceinline__
#else
#define ACT_HD inline
#define ACT_D inline
#endif

#define ACT_MUL64(a, b) ((uint64_t)(a) * (uint64_t)(b))
// One column of a product: carry-in + n terms as n v_mad_u64_u32 in ONE asm statement, the carry being the first
// addend.  Written as asm because LLVM's reassociation otherwise sums the products from zero and adds the carry with a
// separate 64-bit addition (9 of the 152 / 112 instructions of a multiplication / squaring), and as one statement per
// column because the compiler puts a wait state after every asm statement whose result the next instruction reads
// (the hardware needs none between dependent v_mad_u64_u32).  "+v"(c): in the n-term blocks every input is live until its
// own instruction, so c must not share a register with an input -- it cannot: it is also an input.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ uint64_t act_col5(uint64_t c, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %0, vcc, %7, %8, %0\n\tv_mad_u64_u32 %0, vcc, %9, %10, %0" : "+v"(c) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4) : "vcc");
  return c;
}
static __device__ __forceinline__ uint64_t act_col6(uint64_t c, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %0, vcc, %7, %8, %0\n\tv_mad_u64_u32 %0, vcc, %9, %10, %0\n\tv_mad_u64_u32 %0, vcc, %11, %12, %0" : "+v"(c) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5) : "vcc");
  return c;
}
static __device__ __forceinline__ uint64_t act_col10(uint64_t c, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5, uint32_t a6, uint32_t b6, uint32_t a7, uint32_t b7, uint32_t a8, uint32_t b8, uint32_t a9, uint32_t b9) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %0, vcc, %7, %8, %0\n\tv_mad_u64_u32 %0, vcc, %9, %10, %0\n\tv_mad_u64_u32 %0, vcc, %11, %12, %0\n\tv_mad_u64_u32 %0, vcc, %13, %14, %0\n\tv_mad_u64_u32 %0, vcc, %15, %16, %0\n\tv_mad_u64_u32 %0, vcc, %17, %18, %0\n\tv_mad_u64_u32 %0, vcc, %19, %20, %0" : "+v"(c) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7), "v"(a8), "v"(b8), "v"(a9), "v"(b9) : "vcc");
  return c;
}
static __device__ __forceinline__ uint64_t act_colz6(uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5) {
  uint64_t c;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %0, vcc, %7, %8, %0\n\tv_mad_u64_u32 %0, vcc, %9, %10, %0\n\tv_mad_u64_u32 %0, vcc, %11, %12, %0" : "=&v"(c) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5) : "vcc");
  return c;
}
static __device__ __forceinline__ uint64_t act_colz10(uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5, uint32_t a6, uint32_t b6, uint32_t a7, uint32_t b7, uint32_t a8, uint32_t b8, uint32_t a9, uint32_t b9) {
  uint64_t c;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %0, vcc, %5, %6, %0\n\tv_mad_u64_u32 %0, vcc, %7, %8, %0\n\tv_mad_u64_u32 %0, vcc, %9, %10, %0\n\tv_mad_u64_u32 %0, vcc, %11, %12, %0\n\tv_mad_u64_u32 %0, vcc, %13, %14, %0\n\tv_mad_u64_u32 %0, vcc, %15, %16, %0\n\tv_mad_u64_u32 %0, vcc, %17, %18, %0\n\tv_mad_u64_u32 %0, vcc, %19, %20, %0" : "=&v"(c) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7), "v"(a8), "v"(b8), "v"(a9), "v"(b9) : "vcc");
  return c;
}
#define ACT_COL5(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4) act_col5(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4)
#define ACT_COL6(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5) act_col6(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5)
#define ACT_COL10(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7, a8, b8, a9, b9) act_col10(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7, a8, b8, a9, b9)
#define ACT_COLZ6(a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5) act_colz6(a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5)
#define ACT_COLZ10(a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7, a8, b8, a9, b9) act_colz10(a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7, a8, b8, a9, b9)
#else
#define ACT_COL5(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4) ((uint64_t)(c) + (uint64_t)(a0) * (uint64_t)(b0) + (uint64_t)(a1) * (uint64_t)(b1) + (uint64_t)(a2) * (uint64_t)(b2) + (uint64_t)(a3) * (uint64_t)(b3) + (uint64_t)(a4) * (uint64_t)(b4))
#define ACT_COL6(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5) ((uint64_t)(c) + (uint64_t)(a0) * (uint64_t)(b0) + (uint64_t)(a1) * (uint64_t)(b1) + (uint64_t)(a2) * (uint64_t)(b2) + (uint64_t)(a3) * (uint64_t)(b3) + (uint64_t)(a4) * (uint64_t)(b4) + (uint64_t)(a5) * (uint64_t)(b5))
#define ACT_COL10(c, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7, a8, b8, a9, b9) ((uint64_t)(c) + (uint64_t)(a0) * (uint64_t)(b0) + (uint64_t)(a1) * (uint64_t)(b1) + (uint64_t)(a2) * (uint64_t)(b2) + (uint64_t)(a3) * (uint64_t)(b3) + (uint64_t)(a4) * (uint64_t)(b4) + (uint64_t)(a5) * (uint64_t)(b5) + (uint64_t)(a6) * (uint64_t)(b6) + (uint64_t)(a7) * (uint64_t)(b7) + (uint64_t)(a8) * (uint64_t)(b8) + (uint64_t)(a9) * (uint64_t)(b9))
#define ACT_COLZ6(a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5) ((uint64_t)(a0) * (uint64_t)(b0) + (uint64_t)(a1) * (uint64_t)(b1) + (uint64_t)(a2) * (uint64_t)(b2) + (uint64_t)(a3) * (uint64_t)(b3) + (uint64_t)(a4) * (uint64_t)(b4) + (uint64_t)(a5) * (uint64_t)(b5))
#define ACT_COLZ10(a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7, a8, b8, a9, b9) ((uint64_t)(a0) * (uint64_t)(b0) + (uint64_t)(a1) * (uint64_t)(b1) + (uint64_t)(a2) * (uint64_t)(b2) + (uint64_t)(a3) * (uint64_t)(b3) + (uint64_t)(a4) * (uint64_t)(b4) + (uint64_t)(a5) * (uint64_t)(b5) + (uint64_t)(a6) * (uint64_t)(b6) + (uint64_t)(a7) * (uint64_t)(b7) + (uint64_t)(a8) * (uint64_t)(b8) + (uint64_t)(a9) * (uint64_t)(b9))
#endif
#include "fe25519_gen.inc"

namespace act {

struct fe { uint32_t v[10]; };

constexpr uint32_t FE_M26 = 0x3ffffffu, FE_M25 = 0x1ffffffu;

#ifdef ACT_FE_BOUNDS
// host-only instrumentation: record the largest limb ever fed to each operand class
struct fe_bounds_t { uint64_t max_g_even, max_g_odd, max_f_even, max_f_odd, max_sub_even, max_sub_odd; };
extern fe_bounds_t fe_bounds;
inline void fe_note_g(const fe& g) { for (int i = 0; i < 10; i++) { uint64_t& m = (i & 1) ? fe_bounds.max_g_odd : fe_bounds.max_g_even; if (g.v[i] > m) m = g.v[i]; } }
inline void fe_note_f(const fe& f) { for (int i = 0; i < 10; i++) { uint64_t& m = (i & 1) ? fe_bounds.max_f_odd : fe_bounds.max_f_even; if (f.v[i] > m) m = f.v[i]; } }
inline void fe_note_sub(const fe& g) { for (int i = 0; i < 10; i++) { uint64_t& m = (i & 1) ? fe_bounds.max_sub_odd : fe_bounds.max_sub_even; if (g.v[i] > m) m = g.v[i]; } }
// ... and count the multiplications / squarings executed (the exact operation counts behind bench.py's ALU roofline)
struct fe_counts_t { uint64_t mul, sq, fixed_base[4]; };
extern fe_counts_t fe_counts;
#define fe_count_mul() ((void)++fe_counts.mul)
#define fe_count_sq() ((void)++fe_counts.sq)
#define fe_count_fixed_base(id) ((void)++fe_counts.fixed_base[(id) & 3u])   // calls of msm.h fixed_base_acc per base: the host test build uses narrower windows
#else
#define fe_note_g(x) ((void)0)
#define fe_note_f(x) ((void)0)
#define fe_note_sub(x) ((void)0)
#define fe_count_mul() ((void)0)
#define fe_count_sq() ((void)0)
#define fe_count_fixed_base(id) ((void)0)
#endif

ACT_HD fe fe_zero() { fe r; for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }
ACT_HD fe fe_one() { fe r = fe_zero(); r.v[0] = 1; return r; }

// the columns arrive with the carries already propagated upwards (fe25519_gen.inc): mask, fold the top carry -> tight limbs
#define ACT_FE_CARRY_COLUMNS(h)                                                    \
  do {                                                                             \
    h0 &= FE_M26; h1 &= FE_M25; h2 &= FE_M26; h3 &= FE_M25; h4 &= FE_M26;          \
    h5 &= FE_M25; h6 &= FE_M26; h7 &= FE_M25; h8 &= FE_M26;                        \
    h0 += 19u * (h9 >> 25); h9 &= FE_M25;                                          \
    h1 += h0 >> 26; h0 &= FE_M26;                                                  \
    h.v[0] = (uint32_t)h0; h.v[1] = (uint32_t)h1; h.v[2] = (uint32_t)h2; h.v[3] = (uint32_t)h3; h.v[4] = (uint32_t)h4; \
    h.v[5] = (uint32_t)h5; h.v[6] = (uint32_t)h6; h.v[7] = (uint32_t)h7; h.v[8] = (uint32_t)h8; h.v[9] = (uint32_t)h9; \
  } while (0)

ACT_HD fe fe_mul(const fe& f, const fe& g) {
  fe_note_f(f); fe_note_g(g); fe_count_mul();
  ACT_FE_MUL_BODY
  fe h; ACT_FE_CARRY_COLUMNS(h); return h;
}
ACT_HD fe fe_sq(const fe& f) {
  fe_note_g(f); fe_count_sq();
  ACT_FE_SQ_BODY
  fe h; ACT_FE_CARRY_COLUMNS(h); return h;
}
ACT_HD fe fe_sqn(fe f, int n) { for (int i = 0; i < n; i++) f = fe_sq(f); return f; }

// 32-bit carry pass for a loose element (limbs < 2^31): result tight
ACT_HD fe fe_carry(const fe& f) {
  uint32_t h0 = f.v[0], h1 = f.v[1], h2 = f.v[2], h3 = f.v[3], h4 = f.v[4], h5 = f.v[5], h6 = f.v[6], h7 = f.v[7], h8 = f.v[8], h9 = f.v[9];
  uint32_t c9 = h9 >> 25; h9 &= FE_M25;
  h0 += 19u * c9;
  h1 += h0 >> 26; h0 &= FE_M26;
  h2 += h1 >> 25; h1 &= FE_M25;
  h3 += h2 >> 26; h2 &= FE_M26;
  h4 += h3 >> 25; h3 &= FE_M25;
  h5 += h4 >> 26; h4 &= FE_M26;
  h6 += h5 >> 25; h5 &= FE_M25;
  h7 += h6 >> 26; h6 &= FE_M26;
  h8 += h7 >> 25; h7 &= FE_M25;
  h9 += h8 >> 26; h8 &= FE_M26;
  fe r; r.v[0] = h0; r.v[1] = h1; r.v[2] = h2; r.v[3] = h3; r.v[4] = h4; r.v[5] = h5; r.v[6] = h6; r.v[7] = h7; r.v[8] = h8; r.v[9] = h9;
  return r;   // h9 <= 2^25 + 2^6: still tight
}

ACT_HD fe fe_add(const fe& f, const fe& g) { fe r; for (int i = 0; i < 10; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
// f - g + 2p  (g tight)
ACT_HD fe fe_sub(const fe& f, const fe& g) {
  fe_note_sub(g);
  fe r;
  r.v[0] = f.v[0] + 0x7ffffdau - g.v[0];
  for (int i = 1; i < 10; i++) r.v[i] = f.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - g.v[i];
  return r;
}
// f - g + 4p  (g limbs <= 2^28 - 76 even / 2^27 - 4 odd)
ACT_HD fe fe_sub4(const fe& f, const fe& g) {
  fe r;
  r.v[0] = f.v[0] + 0xfffffb4u - g.v[0];
  for (int i = 1; i < 10; i++) r.v[i] = f.v[i] + ((i & 1) ? 0x7fffffcu : 0xffffffcu) - g.v[i];
  return r;
}
ACT_HD fe fe_neg(const fe& f) { return fe_sub(fe_zero(), f); }   // loose (<= 2p limbs); f tight
ACT_HD fe fe_dbl(const fe& f) { fe r; for (int i = 0; i < 10; i++) r.v[i] = 2u * f.v[i]; return r; }

// Per-lane selects go through an all-ones/all-zeros mask and v_bfi_b32.  (A micro-benchmark loop of VOP2
// v_cndmask_b32 with the mask in VCC measured 23 cycles per instruction, but replacing the compiler's selects by
// this form left k_spend_bits unchanged within noise: the form is kept for its predictable code, not for speed.)
// The empty asm keeps LLVM from folding the mask arithmetic back into selects.
ACT_HD uint32_t fe_mask(bool b) {
  uint32_t m = 0u - (uint32_t)b;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(m));
#endif
  return m;
}
ACT_HD uint32_t sel32(uint32_t a, uint32_t b, uint32_t m) { return (b & m) | (a & ~m); }
ACT_HD fe fe_select_m(const fe& a, const fe& b, uint32_t m) { fe r; for (int i = 0; i < 10; i++) r.v[i] = sel32(a.v[i], b.v[i], m); return r; }
ACT_HD fe fe_select(const fe& a, const fe& b, bool take_b) { return fe_select_m(a, b, fe_mask(take_b)); }
ACT_HD void fe_cswap_m(fe& a, fe& b, uint32_t m) { for (int i = 0; i < 10; i++) { uint32_t x = a.v[i], y = b.v[i]; a.v[i] = sel32(x, y, m); b.v[i] = sel32(y, x, m); } }
ACT_HD void fe_cswap(fe& a, fe& b, bool sw) { fe_cswap_m(a, b, fe_mask(sw)); }

// little-endian 32 bytes given as eight u32 words; bit 255 ignored
ACT_HD fe fe_from_words(const uint32_t w[8]) {
  fe r;
  r.v[0] = w[0] & FE_M26;
  r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & FE_M25;
  r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & FE_M26;
  r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & FE_M25;
  r.v[4] = (w[3] >> 6) & FE_M26;
  r.v[5] = w[4] & FE_M25;
  r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & FE_M26;
  r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & FE_M25;
  r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & FE_M26;
  r.v[9] = (w[7] >> 6) & FE_M25;
  return r;
}
// canonical (fully reduced) little-endian words; input loose (limbs < 2^31)
ACT_HD void fe_to_words(uint32_t w[8], const fe& f) {
  fe t = fe_carry(fe_carry(f));
  // q = 1 iff t >= p  (t < 2^255 + small after two carries)
  uint32_t q = (t.v[0] + 19u) >> 26;
  q = (t.v[1] + q) >> 25; q = (t.v[2] + q) >> 26; q = (t.v[3] + q) >> 25; q = (t.v[4] + q) >> 26;
  q = (t.v[5] + q) >> 25; q = (t.v[6] + q) >> 26; q = (t.v[7] + q) >> 25; q = (t.v[8] + q) >> 26; q = (t.v[9] + q) >> 25;
  uint32_t h0 = t.v[0] + 19u * q, h1 = t.v[1], h2 = t.v[2], h3 = t.v[3], h4 = t.v[4], h5 = t.v[5], h6 = t.v[6], h7 = t.v[7], h8 = t.v[8], h9 = t.v[9];
  h1 += h0 >> 26; h0 &= FE_M26;
  h2 += h1 >> 25; h1 &= FE_M25;
  h3 += h2 >> 26; h2 &= FE_M26;
  h4 += h3 >> 25; h3 &= FE_M25;
  h5 += h4 >> 26; h4 &= FE_M26;
  h6 += h5 >> 25; h5 &= FE_M25;
  h7 += h6 >> 26; h6 &= FE_M26;
  h8 += h7 >> 25; h7 &= FE_M25;
  h9 += h8 >> 26; h8 &= FE_M26;
  h9 &= FE_M25;
  w[0] = h0 | (h1 << 26);
  w[1] = (h1 >> 6) | (h2 << 19);
  w[2] = (h2 >> 13) | (h3 << 13);
  w[3] = (h3 >> 19) | (h4 << 6);
  w[4] = h5 | (h6 << 25);
  w[5] = (h6 >> 7) | (h7 << 19);
  w[6] = (h7 >> 13) | (h8 << 12);
  w[7] = (h8 >> 20) | (h9 << 6);
}
ACT_HD bool fe_is_negative(const fe& f) { uint32_t w[8]; fe_to_words(w, f); return w[0] & 1u; }
ACT_HD bool fe_is_zero(const fe& f) { uint32_t w[8]; fe_to_words(w, f); uint32_t o = 0; for (int i = 0; i < 8; i++) o |= w[i]; return o == 0; }
ACT_HD bool fe_equal(const fe& f, const fe& g) { return fe_is_zero(fe_sub(f, fe_carry(g))); }
ACT_HD fe fe_cneg(const fe& f, bool neg) { return fe_select(f, fe_neg(f), neg); }      // f tight
ACT_HD fe fe_abs(const fe& f) { return fe_cneg(f, fe_is_negative(f)); }                   // f tight; result loose (<= 2p)

// constants (tight limbs), from oracle-independent derivation in tools/gen_fe_consts.py
#define ACT_FE_CONST(name, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9) \
  ACT_HD fe name() { fe r; r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7; r.v[8] = a8; r.v[9] = a9; return r; }
#include "fe25519_consts.inc"

// z^(2^250-1) and z^11 (shared prefix of inversion and of the (p-5)/8 power)
ACT_HD void fe_pow_core(fe& t250, fe& z11, const fe& z) {
  fe t0 = fe_sq(z);
  fe t1 = fe_sqn(t0, 2);
  t1 = fe_mul(z, t1);            // 9
  z11 = fe_mul(t0, t1);          // 11
  t0 = fe_sq(z11);               // 22
  t0 = fe_mul(t1, t0);           // 2^5 - 1
  t1 = fe_sqn(t0, 5); t0 = fe_mul(t1, t0);        // 2^10 - 1
  t1 = fe_sqn(t0, 10); t1 = fe_mul(t1, t0);       // 2^20 - 1
  fe t2 = fe_sqn(t1, 20); t1 = fe_mul(t2, t1);    // 2^40 - 1
  t1 = fe_sqn(t1, 10); t0 = fe_mul(t1, t0);       // 2^50 - 1
  t1 = fe_sqn(t0, 50); t1 = fe_mul(t1, t0);       // 2^100 - 1
  t2 = fe_sqn(t1, 100); t1 = fe_mul(t2, t1);      // 2^200 - 1
  t1 = fe_sqn(t1, 50); t250 = fe_mul(t1, t0);     // 2^250 - 1
}
ACT_HD fe fe_invert(const fe& z) { fe t, z11; fe_pow_core(t, z11, z); return fe_mul(fe_sqn(t, 5), z11); }
ACT_HD fe fe_pow22523(const fe& z) { fe t, z11; fe_pow_core(t, z11, z); return fe_mul(fe_sqn(t, 2), z); }

// RFC 9496 section 4.2 SQRT_RATIO_M1(u, v): r = |sqrt(u/v)| or |sqrt(i*u/v)|; u, v tight
ACT_HD bool fe_sqrt_ratio_m1(fe& r, const fe& u, const fe& v) {
  fe v3 = fe_mul(fe_sq(v), v);
  fe v7 = fe_mul(fe_sq(v3), v);
  fe rr = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
  fe check = fe_mul(v, fe_sq(rr));
  fe neg_u = fe_carry(fe_neg(u));
  fe neg_u_i = fe_mul(neg_u, fe_sqrt_m1());
  bool correct = fe_equal(check, u), flipped = fe_equal(check, neg_u), flipped_i = fe_equal(check, neg_u_i);
  fe ri = fe_mul(rr, fe_sqrt_m1());
  rr = fe_select(rr, ri, flipped || flipped_i);
  r = fe_carry(fe_abs(rr));
  return correct || flipped;
}
// invsqrt specialisation for u = 1 (the only form compress/decompress need): saves two muls
ACT_HD bool fe_invsqrt(fe& r, const fe& v) {
  fe v3 = fe_mul(fe_sq(v), v);
  fe v7 = fe_mul(fe_sq(v3), v);
  fe rr = fe_mul(v3, fe_pow22523(v7));
  fe check = fe_mul(v, fe_sq(rr));
  uint32_t w[8]; fe_to_words(w, check);
  // check in {1, -1, -sqrt(-1)}  <=>  correct / flipped / flipped_i
  uint32_t m1[8], mi[8];
  fe_to_words(m1, fe_neg(fe_one())); fe_to_words(mi, fe_neg(fe_sqrt_m1()));
  bool correct = (w[0] == 1u), flipped = true, flipped_i = true;
  for (int i = 1; i < 8; i++) correct = correct && (w[i] == 0u);
  for (int i = 0; i < 8; i++) { flipped = flipped && (w[i] == m1[i]); flipped_i = flipped_i && (w[i] == mi[i]); }
  fe ri = fe_mul(rr, fe_sqrt_m1());
  rr = fe_select(rr, ri, flipped || flipped_i);
  r = fe_carry(fe_abs(rr));
  return correct || flipped;
}

}  // namespace act
