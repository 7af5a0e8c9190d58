// ubench_limb_lanes.hip — the layout question BASELINE.json's north_star raises, measured: GF(2^255-19) multiplication with
//   (A) one element per lane   : csrc/fe25519.h fe_mul as the engine uses it (10 x 25.5-bit limbs in registers, carries inside the lane)
//   (B) one LIMB per lane      : 8 saturated 32-bit limbs of an element spread over 8 consecutive lanes (8 elements per wavefront),
//                                operands fetched across lanes with wavefront shuffles, the 256-bit carry propagated lane to lane with
//                                shuffles as well, 2^256 = 38 folded back (the layout a register-starved CUDA-style kernel would pick)
// Both run dependent chains a <- a * b on every lane of every SIMD (8 waves per SIMD); prints field multiplications per second.
// (B) is checked against (A) on the device for every element before it is timed.
// build: hipcc -O3 --offload-arch=gfx950 -I anonymous-credit-tokens_amd/csrc -o /tmp/ubench_limb_lanes tools/ubench_limb_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "fe25519.h"
using namespace act;

__global__ void __launch_bounds__(256) k_elem(uint32_t* io, int iters) {        // (A): io[gid][8] packed words in, out
  uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = io[gid * 16 + i];
  fe a = fe_from_words(w); for (int i = 0; i < 8; i++) w[i] = io[gid * 16 + 8 + i]; fe b = fe_from_words(w);
  for (int it = 0; it < iters; it++) a = fe_mul(a, b);
  fe_to_words(w, a); for (int i = 0; i < 8; i++) io[gid * 16 + i] = w[i];
}

// (B) lane l of an 8-lane group holds limb l.  Returns limb l of a*b mod 2^255-19 (value < 2^256, not canonical).
__device__ __forceinline__ uint32_t mul_limb_lanes(uint32_t a, uint32_t b) {
  const int lane = threadIdx.x & 63, l = lane & 7, base = lane & ~7;
  uint32_t lo0 = 0, lo1 = 0, lo2 = 0, hi0 = 0, hi1 = 0, hi2 = 0;               // 96-bit column sums: column l and column l + 8
  for (int i = 0; i < 8; i++) {
    const uint32_t ai = __shfl(a, base + i, 64);                                 // a_i to every lane of the group
    const uint32_t bj = __shfl(b, base + ((l - i) & 7), 64);                     // b_(l - i mod 8)
    const uint64_t p = (uint64_t)ai * bj;
    const bool low = i <= l;                                                     // i + j = l  or  i + j = l + 8
    uint64_t s = (uint64_t)(low ? lo0 : hi0) + (uint32_t)p; uint32_t w0 = (uint32_t)s;
    s = (s >> 32) + (low ? lo1 : hi1) + (uint32_t)(p >> 32); uint32_t w1 = (uint32_t)s; uint32_t w2 = (low ? lo2 : hi2) + (uint32_t)(s >> 32);
    if (low) { lo0 = w0; lo1 = w1; lo2 = w2; } else { hi0 = w0; hi1 = w1; hi2 = w2; }
  }
  // fold the high half: column l + 8 weighs 2^256 * 2^(32 l) = 38 * 2^(32 l).  S_l = LO_l + 38 HI_l as words s0 s1 s2 s3
  uint64_t t = (uint64_t)hi0 * 38u + lo0; const uint32_t s0 = (uint32_t)t;
  t = (t >> 32) + (uint64_t)hi1 * 38u + lo1; const uint32_t s1 = (uint32_t)t;
  t = (t >> 32) + (uint64_t)hi2 * 38u + lo2; const uint32_t s2 = (uint32_t)t, s3 = (uint32_t)(t >> 32);
  // limb k of the result collects s0_k + s1_(k-1) + s2_(k-2) + s3_(k-3): words handed UP the lanes by shuffles; what would land above
  // limb 7 wraps to the bottom with weight 38
  uint32_t in1 = __shfl(s1, base + ((l - 1) & 7), 64), in2 = __shfl(s2, base + ((l - 2) & 7), 64), in3 = __shfl(s3, base + ((l - 3) & 7), 64);
  uint64_t acc = (uint64_t)s0 + (uint64_t)in1 * (l < 1 ? 38u : 1u) + (uint64_t)in2 * (l < 2 ? 38u : 1u) + (uint64_t)in3 * (l < 3 ? 38u : 1u);
  uint32_t r = (uint32_t)acc, carry = (uint32_t)(acc >> 32);
  for (int round = 0; round < 10; round++) {                                     // 256-bit carry propagation: one lane per round
    uint32_t cin = __shfl(carry, base + ((l - 1) & 7), 64);
    acc = (uint64_t)r + (uint64_t)cin * (l == 0 ? 38u : 1u); r = (uint32_t)acc; carry = (uint32_t)(acc >> 32);
  }
  return r;
}
__global__ void __launch_bounds__(256) k_limb(uint32_t* io, int iters) {        // io[group][16]: lanes 0..7 of a group = a limbs, then b limbs
  uint32_t gid = blockIdx.x * 256 + threadIdx.x, grp = gid >> 3, l = gid & 7;
  uint32_t a = io[grp * 16 + l], b = io[grp * 16 + 8 + l];
  for (int it = 0; it < iters; it++) a = mul_limb_lanes(a, b);
  io[grp * 16 + l] = a;
}

static double run(void (*k)(uint32_t*, int), uint32_t* d, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 16);
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 8, lanes = blocks * 256;
  // correctness of (B) against (A): one multiplication of the same operands
  const int n = 4096;
  std::vector<uint32_t> h(n * 16), ha, hb;
  uint64_t s = 88172645463325252ull; auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
  for (auto& x : h) x = rnd();
  for (int i = 0; i < n; i++) { h[i * 16 + 7] &= 0x7fffffffu; h[i * 16 + 15] &= 0x7fffffffu; }
  uint32_t *da, *db; hipMalloc(&da, n * 64); hipMalloc(&db, n * 64);
  hipMemcpy(da, h.data(), n * 64, hipMemcpyHostToDevice); hipMemcpy(db, h.data(), n * 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_elem, dim3(n / 256), dim3(256), 0, 0, da, 1);
  hipLaunchKernelGGL(k_limb, dim3(n * 8 / 256), dim3(256), 0, 0, db, 1);
  ha.resize(n * 16); hb.resize(n * 16);
  hipMemcpy(ha.data(), da, n * 64, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), db, n * 64, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) {                                                  // compare mod p: reduce (B)'s < 2^256 value on the host
    unsigned __int128 c = 0; uint32_t v[8];
    for (int k = 0; k < 8; k++) v[k] = hb[i * 16 + k];
    for (int rep = 0; rep < 3; rep++) {                                          // v mod 2^255 - 19 (v < 2^256): fold bit 255, then conditional subtract
      uint32_t topbit = v[7] >> 31; v[7] &= 0x7fffffffu; c = (unsigned __int128)topbit * 19u;
      for (int k = 0; k < 8; k++) { c += v[k]; v[k] = (uint32_t)c; c >>= 32; }
    }
    uint32_t t[8]; c = 19; for (int k = 0; k < 8; k++) { c += v[k]; t[k] = (uint32_t)c; c >>= 32; }
    if (t[7] >> 31) { t[7] &= 0x7fffffffu; for (int k = 0; k < 8; k++) v[k] = t[k]; }
    for (int k = 0; k < 8; k++) if (v[k] != ha[i * 16 + k]) { bad++; break; }
  }
  printf("limb-per-lane product equals element-per-lane product on %d / %d random operand pairs\n", n - bad, n);
  uint32_t* d; hipMalloc(&d, (size_t)lanes * 64); hipMemset(d, 0x11, (size_t)lanes * 64);
  const int iters = 4096;
  double ms_a = run(k_elem, d, blocks, iters), ms_b = run(k_limb, d, blocks, iters);
  double ra = (double)lanes * iters / (ms_a * 1e-3), rb = (double)lanes / 8 * iters / (ms_b * 1e-3);
  printf("element per lane : %.3e field multiplications/s (%.2f ms)\n", ra, ms_a);
  printf("limb per lane    : %.3e field multiplications/s (%.2f ms)   ratio %.2f\n", rb, ms_b, ra / rb);
  return bad != 0;
}
