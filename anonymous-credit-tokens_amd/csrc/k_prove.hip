// k_prove.hip — CreditToken::prove_spend (/root/reference/src/lib.rs:972-1152) as gfx950 kernels.  The per-lane bodies live in
// prove_lanes.h (so that the CPU test build can run, count and sanitize the same code); this file maps lanes to threads, owns the
// LDS of the ct build's staged tables and launches.
#include "prove_lanes.h"

namespace act {

__global__ void __launch_bounds__(64, 2) k_prove_head(ProveArgs a) {
  ACT_SECRET_FB(fb, a.P);
  prove_head_lane(a, blockIdx.x * 64 + threadIdx.x, fb);
}
// The same head for SHORT launches, fourteen BLOCKS of one wavefront per 64 proofs (blockIdx.y = role, lane = proof).  prove_head_lane
// is ~15 000 dependent field operations in one lane (a two-scalar chain on A, fifteen fixed-base products, four encodings): 4.2 ms
// however few proofs there are -- and the crate's prove_spend takes ONE token (src/lib.rs:972-977).  Its pieces are independent until
// A1 is summed:
//     roles 0-3    the four quarters of A' = (r1 r2) A          roles 4-7   the four quarters of P1 = (e' r1 r2) A
//                  (msm.h chain_ct_quarter: 64 i doublings + 33 chain steps each, instead of 127 steps in one wavefront)
//     role 8       B_bar                                         role 9      the fixed-base part of A1 = e' A' + r2' B_bar
//     role 10      A2                                            roles 11-13 the three h2 terms of bit 0 (d0, d1, d2)
// and the block of a group that ARRIVES LAST (a counter per group) adds the quarters, A1 = (role 9's part) + P1, and encodes A' and
// A1.  Roles 8-13 make one product per base (the matrix-core look-up wants all 64 lanes of a wavefront: lanes past the batch multiply
// by zero), roles 11-13 on h2 only.  Same group elements, hence the same bytes (tests: every small prove_spend call goes through
// here).  Round 4 had the roles on LANES of one wavefront -- the kernel was a chain plus four products in sequence, 1.95 ms for one
// proof; the wavefronts of ONE block may land on one SIMD and then share it (0.9 or 2.0 ms from call to call:
// profiles/r05_tiny_ab.txt); separate workgroups go to separate CUs.
__global__ void __launch_bounds__(64) k_prove_head_wide(ProveArgs a) {
  __shared__ uint32_t ticket;
  ACT_SECRET_FB(fb, a.P);
  enum { R_AP0 = 0, R_P10 = 4, R_BBAR = 8, R_A1F = 9, R_A2 = 10, R_D0 = 11, ROLES = 14 };
  const uint32_t role = blockIdx.y, lane = threadIdx.x, p = blockIdx.x * 64 + lane;      // role is uniform over the block
  const bool live = p < a.n;
  const int L = a.P.L;
  const ProofLayout pl{L}; const SpendTranscript st{L};
  RngView rv{a.rng + (size_t)(live ? p : 0) * rng_bytes(L), L};
  uint8_t* rec = a.proof + (size_t)(live ? p : 0) * pl.bytes();
  uint8_t* el = a.tr + (size_t)(live ? p : 0) * a.tr_stride + 184;
  uint32_t* park_q = a.half + (size_t)(live ? p : 0) * 2 * BUCKET_WORDS;                 // two areas per proof: the quarters of A' (4 points), of P1 (4 points)
  uint32_t* park_f3 = a.half + ((size_t)a.n * 2 + (live ? p : 0)) * BUCKET_WORDS;        // (the half-point area holds max(L, 3) such areas per proof)
  sc k = sc_zero(), s = sc_zero(), r1 = sc_zero();
  sc sg = sc_zero(), sh1 = sc_zero(), sh2 = sc_zero(), sh3 = sc_zero();      // this lane's scalar on g, h1, h2, h3
  if (live) {
    const uint8_t* tok = a.tok + (size_t)p * 160;
    k = load_sc(tok + 64); const sc r = load_sc(tok + 96), c = load_sc(tok + 128); s = load_sc(a.s + (size_t)p * 32);
    r1 = rv.r1();
    if (role < R_BBAR) {
      uint32_t wa[8]; load8(wa, tok);
      ge A; const bool ok = ristretto_decode(A, wa);
      if (role == R_AP0) a.flags[p] = ok ? 0u : FLAG_UNDECODABLE;
      const sc r1r2 = sc_mul(r1, rv.r2());
      const bool first = role < R_P10;
      const int q = (int)(role - (first ? R_AP0 : R_P10));
      ge_store(park_q + (size_t)(first ? 0 : BUCKET_WORDS) + (size_t)q * GE_WORDS, chain_ct_quarter(A, first ? r1r2 : sc_mul(rv.e_prime(), r1r2), q));
      if (role == R_AP0) {
        tr_put_prefix(a.tr + (size_t)p * a.tr_stride, a.P, LABEL_SPEND);
        tr_put_aligned(el + 40 * st.el_k(), k.v);
        store_sc(rec + 32 * pl.k(), k); store_sc(rec + 32 * pl.s(), s);
        const sc r3 = sc_invert(r1);                                                // :992
        uint32_t* stt = a.state + (size_t)p * 24;
        for (int i = 0; i < 8; i++) stt[i] = r3.v[i];
      }
    } else if (role <= R_A2) {
      // B_bar = r1 g + (r1 c) h1 + (r1 k) h2 + (r1 r) h3;  A1 - P1 = r2' B_bar;  A2 = r3' B_bar + c' h1 + r' h3
      const sc m = role == R_BBAR ? sc_one() : role == R_A1F ? rv.r2_prime() : rv.r3_prime();
      const sc t = sc_mul(m, r1);
      sg = t; sh1 = sc_mul(t, c); sh2 = sc_mul(t, k); sh3 = sc_mul(t, r);
      if (role == R_A2) { sh1 = sc_add(sh1, rv.c_prime()); sh3 = sc_add(sh3, rv.r_prime()); }
    } else {
      // the three h2 terms of bit 0 (src/lib.rs:1001, 1025-1035), at half scale like everything k_prove_bits computes
      const sc k_star = rv.k_star();
      sh2 = sc_half(role == R_D0 ? k_star : role == R_D0 + 1 ? rv.k0_prime() : sc_sub(rv.w0(), sc_mul(rv.gamma_i(0), k_star)));
    }
  }
  uint32_t enc[8];
  if (role >= R_BBAR) {                                        // uniform over the block: whole wavefronts multiply, or none of their lanes
    ge f = ge_identity();
    if (role <= R_A2) { fb.stage(BASE_G);  f = fb.mul(f, BASE_G, sg); fb.stage(BASE_H1); f = fb.mul(f, BASE_H1, sh1); }
    fb.stage(BASE_H2); f = fb.mul(f, BASE_H2, sh2);
    if (role <= R_A2) { fb.stage(BASE_H3); f = fb.mul(f, BASE_H3, sh3); }
    if (live) {
      if (role >= R_D0) ge_store(a.d3 + ((size_t)p * 3 + (role - R_D0)) * GE_WORDS, f);
      else if (role == R_A1F) ge_store(park_f3, f);
      else {
        ristretto_encode(enc, f);
        if (role == R_BBAR) { tr_put_aligned(el + 40 * st.el_b_bar(), enc); store8(rec + 32 * pl.b_bar(), enc); }
        else tr_put_aligned(el + 40 * st.el_a2(), enc);
      }
    }
  }
  // the last block of the group to arrive sums A' and A1
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicAdd(a.group_counter + blockIdx.x, 1u);
  __syncthreads();
  if (ticket != (uint32_t)ROLES - 1u) return;
  __threadfence();
  if (threadIdx.x == 0) a.group_counter[blockIdx.x] = 0u;
  if (!live) return;
  ge ap = ge_load(park_q), p1 = ge_load(park_q + BUCKET_WORDS);
  for (int q = 1; q < 4; q++) { ap = ge_add(ap, ge_load(park_q + (size_t)q * GE_WORDS)); p1 = ge_add(p1, ge_load(park_q + BUCKET_WORDS + (size_t)q * GE_WORDS)); }
  ristretto_encode(enc, ap);
  tr_put_aligned(el + 40 * st.el_a_prime(), enc); store8(rec + 32 * pl.a_prime(), enc);
  ristretto_encode(enc, ge_add(ge_load(park_f3), p1));
  tr_put_aligned(el + 40 * st.el_a1(), enc);
}
__global__ void __launch_bounds__(256, 2) k_prove_bits(ProveArgs a) {
  ACT_SECRET_FB_LDS(fb, a.P);
  prove_bits_lane(a, blockIdx.x * 256 + threadIdx.x, fb);
}
__global__ void __launch_bounds__(256, 2) k_prove_enc(ProveArgs a) {
  prove_enc_lane(a, ((uint64_t)blockIdx.x * 256 + threadIdx.x) * PROVE_ENC_BATCH);
}
__global__ void __launch_bounds__(64, 2) k_prove_tail(ProveArgs a) {
  ACT_SECRET_FB(fb, a.P);
  prove_tail_lane(a, blockIdx.x * 64 + threadIdx.x, fb);
}
__global__ void __launch_bounds__(256) k_prove_resp(ProveArgs a) {
  prove_resp_lane(a, blockIdx.x * 256 + threadIdx.x);
}

constexpr uint32_t PROVE_WIDE_MAX = 8192;
void launch_prove_head(const ProveArgs& a, hipStream_t s) {
  if (!a.n) return;
  const bool no_wide = tune(T_NO_WIDE_PROVE) != 0;     // A/B knob (act_tuning_set)
  if (a.n <= PROVE_WIDE_MAX && !no_wide) hipLaunchKernelGGL(k_prove_head_wide, dim3((a.n + 63) / 64, 14), dim3(64), isolate_roles((a.n + 63) / 64 * 14), s, a);
  else hipLaunchKernelGGL(k_prove_head, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
}
void launch_prove_bits(const ProveArgs& a, hipStream_t s) {
  if (!a.n) return;
  size_t lanes = (size_t)a.n * a.P.L;
  hipLaunchKernelGGL(k_prove_bits, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a);
}
void launch_prove_enc(const ProveArgs& a, hipStream_t s) {
  if (!a.n) return;
  size_t threads = ((size_t)a.n * a.P.L * 3 + PROVE_ENC_BATCH - 1) / PROVE_ENC_BATCH;
  hipLaunchKernelGGL(k_prove_enc, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
}
void launch_prove_tail(const ProveArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_prove_tail, dim3((a.n + 63) / 64), dim3(64), isolate_role((a.n + 63) / 64), s, a); }
void launch_prove_resp(const ProveArgs& a, hipStream_t s) {
  if (!a.n) return;
  size_t lanes = (size_t)a.n * a.P.L;
  hipLaunchKernelGGL(k_prove_resp, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace act
