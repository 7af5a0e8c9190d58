// k_prove.hip — CreditToken::prove_spend (/root/reference/src/lib.rs:972-1152) as gfx950 kernels.  The per-lane bodies live in
// prove_lanes.h (so that the CPU test build can run, count and sanitize the same code); this file maps lanes to threads, owns the
// LDS of the ct build's staged tables and launches.
#include "prove_lanes.h"

namespace act {

__global__ void __launch_bounds__(64, 2) k_prove_head(ProveArgs a) {
  ACT_SECRET_FB(fb, a.P);
  prove_head_lane(a, blockIdx.x * 64 + threadIdx.x, fb);
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

void launch_prove_head(const ProveArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_prove_head, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }
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
void launch_prove_tail(const ProveArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_prove_tail, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }
void launch_prove_resp(const ProveArgs& a, hipStream_t s) {
  if (!a.n) return;
  size_t lanes = (size_t)a.n * a.P.L;
  hipLaunchKernelGGL(k_prove_resp, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace act
