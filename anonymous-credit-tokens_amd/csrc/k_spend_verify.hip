// k_spend_verify.hip — spend-proof verification (PrivateKey::refund up to the challenge check,
// /root/reference/src/lib.rs:787-844) as gfx950 kernels that between them write the exact "spend" transcript
// pre-image (src/lib.rs:831-840) of every proof into HBM, plus the status kernel.  The per-lane bodies live in
// spend_lanes.h (so that the CPU test build can run, count and sanitize the same code); this file maps lanes to
// threads, owns the LDS of the range kernel and launches.
#include "spend_lanes.h"

namespace act {

__global__ void __launch_bounds__(64, 2) k_spend_prep(SpendArgs a) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p < a.n) spend_prep_lane(a, p);
}

// the small-batch schedule's kernels (spend_lanes.h): one role of k_spend_prep each, and the Com_j decode in front of k_spend_tail
__global__ void __launch_bounds__(64, 2) k_spend_prep_a(SpendArgs a) { uint32_t p = blockIdx.x * 64 + threadIdx.x; if (p < a.n) spend_prep_a_lane(a, p); }
__global__ void __launch_bounds__(64, 2) k_spend_prep_b(SpendArgs a) { uint32_t p = blockIdx.x * 64 + threadIdx.x; if (p < a.n) spend_prep_b_lane(a, p); }
__global__ void __launch_bounds__(64, 2) k_spend_prep_c(SpendArgs a) { uint32_t p = blockIdx.x * 64 + threadIdx.x; if (p < a.n) spend_prep_c_lane(a, p); }
__global__ void __launch_bounds__(64, 2) k_spend_prep_c1(SpendArgs a) { uint32_t p = blockIdx.x * 64 + threadIdx.x; if (p < a.n) spend_prep_c1_lane(a, p); }
__global__ void __launch_bounds__(64, 2) k_spend_prep_c2(SpendArgs a) { uint32_t p = blockIdx.x * 64 + threadIdx.x; if (p < a.n) spend_prep_c2_lane(a, p); }
__global__ void __launch_bounds__(256, 2) k_spend_enc_small(SpendArgs a) {
  spend_enc_lane_e<ENC_BATCH_SMALL>(a, ((uint64_t)blockIdx.x * 256 + threadIdx.x) * ENC_BATCH_SMALL);
}
__global__ void __launch_bounds__(64, 2) k_spend_prep_join(SpendArgs a) { uint32_t p = blockIdx.x * 64 + threadIdx.x; if (p < a.n) spend_prep_join_lane(a, p); }
__global__ void __launch_bounds__(64, 2) k_spend_coords(SpendArgs a) { spend_coords_lane(a, blockIdx.x * 64 + threadIdx.x); }

// (the range kernel k_spend_bits lives in k_spend_bits.hip: a translation unit of its own, built with its own field-product form)

__global__ void __launch_bounds__(256, 2) k_spend_enc(SpendArgs a) {
  spend_enc_lane(a, ((uint64_t)blockIdx.x * 256 + threadIdx.x) * ENC_BATCH);
}

__global__ void __launch_bounds__(64, 2) k_spend_tail(SpendArgs a) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p < a.n) spend_tail_lane(a, p);
}
__global__ void __launch_bounds__(64, 2) k_spend_tail_k(SpendArgs a) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p < a.n) spend_tail_k_lane(a, p);
}
__global__ void __launch_bounds__(64, 2) k_spend_tail_c(SpendArgs a) {
  uint32_t p = blockIdx.x * 64 + threadIdx.x;
  if (p < a.n) spend_tail_c_lane(a, p);
}

__global__ void __launch_bounds__(256) k_spend_finish(SpendArgs a) {
  uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p < a.n) spend_finish_lane(a, p);
}

void launch_spend_prep(const SpendArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_spend_prep, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }
void launch_spend_prep_role(const SpendArgs& a, int role, hipStream_t s) {
  if (!a.n) return;
  const dim3 grid((a.n + 63) / 64), block(64);
  const unsigned own = isolate_role(grid.x);
  switch (role) {
    case 0: hipLaunchKernelGGL(k_spend_prep_a, grid, block, own, s, a); break;
    case 1: hipLaunchKernelGGL(k_spend_prep_b, grid, block, own, s, a); break;
    case 2: hipLaunchKernelGGL(k_spend_prep_c, grid, block, own, s, a); break;
    case 4: hipLaunchKernelGGL(k_spend_prep_c1, grid, block, own, s, a); break;
    case 5: hipLaunchKernelGGL(k_spend_prep_c2, grid, block, own, s, a); break;
    default: hipLaunchKernelGGL(k_spend_prep_join, grid, block, own, s, a); break;
  }
}
void launch_spend_coords(const SpendArgs& a, hipStream_t s) {
  if (!a.n) return;
  const size_t lanes = (size_t)a.n * a.P.L;
  hipLaunchKernelGGL(k_spend_coords, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, a);
}
void launch_spend_enc(const SpendArgs& a, hipStream_t s) {
  if (!a.n) return;
  size_t threads = ((size_t)a.n * a.P.L * 2 + ENC_BATCH - 1) / ENC_BATCH;
  hipLaunchKernelGGL(k_spend_enc, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
}
void launch_spend_enc_small(const SpendArgs& a, hipStream_t s) {
  if (!a.n) return;
  size_t threads = ((size_t)a.n * a.P.L * 2 + ENC_BATCH_SMALL - 1) / ENC_BATCH_SMALL;
  hipLaunchKernelGGL(k_spend_enc_small, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
}
void launch_spend_tail_k(const SpendArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_spend_tail_k, dim3((a.n + 63) / 64), dim3(64), isolate_role((a.n + 63) / 64), s, a); }
void launch_spend_tail_c(const SpendArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_spend_tail_c, dim3((a.n + 63) / 64), dim3(64), isolate_role((a.n + 63) / 64), s, a); }
void launch_spend_tail(const SpendArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_spend_tail, dim3((a.n + 63) / 64), dim3(64), isolate_role((a.n + 63) / 64), s, a); }
void launch_spend_finish(const SpendArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_spend_finish, dim3((a.n + 255) / 256), dim3(256), 0, s, a); }

}  // namespace act
