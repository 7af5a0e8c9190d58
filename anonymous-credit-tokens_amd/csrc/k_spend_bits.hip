// k_spend_bits.hip — the range kernel of spend-proof verification (95 % of the metric's GPU time): lane = (proof, bit j), C'_j0 / 2 and
// C'_j1 / 2 of /root/reference/src/lib.rs:800-817 (lane body: spend_lanes.h spend_bits_lane; the chain: msm.h chain_bu_pre).
//
// A translation unit of its own because it is built with its own form of the field products (Makefile: -DACT_FE_PAIR_ASM).  The
// group law's independent products (ge25519.h: fe_mul2 / fe_sq2 in the doubling, the d-free addition, the completed-point
// conversion) run here as ONE asm statement per PAIR, the two instruction streams interleaved (tools/gen_fe_mul.py), and the
// remaining single products as one statement each.  Measured against the per-column form on the bench workload, same box, two
// interleaved rounds (profiles/r06_ab_pair.txt): +0.8 % verifies/s, of which the single statements alone give +0.3 %
// (profiles/r06_ab_oneasm_prefetch.txt) -- the kernel issues VALU instructions ~98 % of the time and 65 % of them are the
// multiply-accumulates no formulation avoids, so removing the compiler's 1 950 wait states between statements and giving every
// multiply-accumulate -> shift -> multiply-accumulate step an independent instruction to issue behind buys that and no more.
// The statements' temporaries are fixed registers v232 .. v254 (the sub-registers of a 64-bit operand cannot be named in a
// template); the per-proof kernels of k_spend_verify.hip (one wavefront per SIMD, long live ranges: k_spend_enc measured 2.7 x
// slower with those registers taken away) keep the per-column form.  Same instructions, same bytes: tests/test_gpu_parity.py.
#include "spend_lanes.h"

namespace act {

#ifndef ACT_BITS_BLOCK
#define ACT_BITS_BLOCK 256
#endif
// UNIFORM: L is a multiple of 64, so a wavefront holds bits of ONE proof and reads that proof's challenge digits into SGPRs
template <bool UNIFORM>
__global__ void __launch_bounds__(ACT_BITS_BLOCK, 2) k_spend_bits(SpendArgs a) {
  __shared__ uint32_t u_lds[(ACT_BITS_BLOCK / 64) * 2 * GE_LDS_WORDS_PER_WAVE];            // 18 KiB per wavefront
  spend_bits_lane<UNIFORM>(a, blockIdx.x * ACT_BITS_BLOCK + threadIdx.x, u_lds + (threadIdx.x >> 6) * 2 * GE_LDS_WORDS_PER_WAVE);
  // one count per workgroup (its first wavefront's end stands for the workgroup's): what engine.hip's hipStreamWaitValue32 gate reads
  if (threadIdx.x == 0 && a.progress) __hip_atomic_fetch_add(a.progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

unsigned spend_bits_workgroups(const SpendArgs& a) { return (unsigned)(((size_t)a.n * a.P.L + ACT_BITS_BLOCK - 1) / ACT_BITS_BLOCK); }
void launch_spend_bits(const SpendArgs& a, hipStream_t s) {
  if (!a.n) return;
  const dim3 grid(spend_bits_workgroups(a));
  if (a.P.L % 64 == 0) hipLaunchKernelGGL(k_spend_bits<true>, grid, dim3(ACT_BITS_BLOCK), isolate_bits(grid.x), s, a);
  else hipLaunchKernelGGL(k_spend_bits<false>, grid, dim3(ACT_BITS_BLOCK), isolate_bits(grid.x), s, a);
}

}  // namespace act
