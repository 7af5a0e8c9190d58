// ubench_mix.hip — marginal issue cost on gfx950 of ten simple VALU instructions added to a block of twenty v_mad_u64_u32
// (two dependent chains), as ONE RUN after the multiply-accumulates or INTERLEAVED with them.  Everything of a block is one asm
// statement, so the compiler adds nothing.  Question behind it: k_spend_bits issues one VALU instruction per 4.06 cycles per SIMD
// (PMC), the multiply-accumulate's own rate, although a quarter of its instructions are v_and / v_add / shifts that run at 2.3
// cycles in isolation -- does grouping them make them cheaper?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mix tools/ubench_mix.hip && ./tools/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 8192;
#define M2 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %5, %4, %1\n\t"
#define M20 M2 M2 M2 M2 M2 M2 M2 M2 M2 M2
enum { BASE, AND_RUN, AND_MIX, ADD_RUN, ADD_MIX, SHL_RUN, SHL_MIX, SHR64_RUN, SHR64_MIX, SAD_RUN, SAD_MIX, BFI_RUN, BFI_MIX, AND_DEP_RUN, AND_DEP_MIX, NCASE };
static const char* names[] = {"20 mad", "20 mad, then 10 v_and", "20 mad / 10 v_and interleaved", "20 mad, then 10 v_add_u32", "20 mad / 10 v_add_u32 interleaved",
  "20 mad, then 10 v_lshlrev_b32", "20 mad / 10 v_lshlrev_b32 interleaved", "20 mad, then 10 v_lshrrev_b64", "20 mad / 10 v_lshrrev_b64 interleaved",
  "20 mad, then 10 v_sad_u32", "20 mad / 10 v_sad_u32 interleaved", "20 mad, then 10 v_bfi_b32", "20 mad / 10 v_bfi_b32 interleaved",
  "20 mad, then 10 v_lshrrev_b64 of a mad result", "20 mad / 10 v_lshrrev_b64 of the preceding mad result"};
#define RUN(op) M20 op op op op op op op op op op
#define MIX(op) M2 op M2 op M2 op M2 op M2 op M2 op M2 op M2 op M2 op M2 op
#define OPS : "+v"(r0), "+v"(r1), "+v"(x), "+v"(y) : "v"(a), "v"(b) : "vcc"
template <int C>
__global__ void __launch_bounds__(256) k_mix(uint32_t* out, uint32_t seed) {
  uint32_t a = seed * 2654435761u + threadIdx.x, b = a ^ 0x9e3779b9u;
  uint64_t r0 = a, r1 = b, x64 = ((uint64_t)a << 32) | b, y64 = ~x64;
  uint32_t x = a + 1, y = b + 1;
  for (int it = 0; it < ITER; it++) {
    if constexpr (C == BASE) asm volatile(M20 OPS);
    else if constexpr (C == AND_RUN) asm volatile(RUN("v_and_b32 %2, 0x3ffffff, %2\n\t") OPS);
    else if constexpr (C == AND_MIX) asm volatile(MIX("v_and_b32 %2, 0x3ffffff, %2\n\t") OPS);
    else if constexpr (C == ADD_RUN) asm volatile(RUN("v_add_u32 %2, %3, %2\n\t") OPS);
    else if constexpr (C == ADD_MIX) asm volatile(MIX("v_add_u32 %2, %3, %2\n\t") OPS);
    else if constexpr (C == SHL_RUN) asm volatile(RUN("v_lshlrev_b32 %2, 1, %2\n\t") OPS);
    else if constexpr (C == SHL_MIX) asm volatile(MIX("v_lshlrev_b32 %2, 1, %2\n\t") OPS);
    else if constexpr (C == SHR64_RUN) asm volatile(RUN("v_lshrrev_b64 %2, 3, %2\n\t") : "+v"(r0), "+v"(r1), "+v"(x64), "+v"(y64) : "v"(a), "v"(b) : "vcc");
    else if constexpr (C == SHR64_MIX) asm volatile(MIX("v_lshrrev_b64 %2, 3, %2\n\t") : "+v"(r0), "+v"(r1), "+v"(x64), "+v"(y64) : "v"(a), "v"(b) : "vcc");
    else if constexpr (C == SAD_RUN) asm volatile(RUN("v_sad_u32 %2, %3, %4, %2\n\t") OPS);
    else if constexpr (C == SAD_MIX) asm volatile(MIX("v_sad_u32 %2, %3, %4, %2\n\t") OPS);
    else if constexpr (C == BFI_RUN) asm volatile(RUN("v_bfi_b32 %2, %3, %4, %2\n\t") OPS);
    else if constexpr (C == BFI_MIX) asm volatile(MIX("v_bfi_b32 %2, %3, %4, %2\n\t") OPS);
    else if constexpr (C == AND_DEP_RUN) asm volatile(RUN("v_lshrrev_b64 %2, 28, %0\n\t") : "+v"(r0), "+v"(r1), "+v"(x64), "+v"(y64) : "v"(a), "v"(b) : "vcc");
    else if constexpr (C == AND_DEP_MIX) asm volatile(MIX("v_lshrrev_b64 %2, 28, %0\n\t") : "+v"(r0), "+v"(r1), "+v"(x64), "+v"(y64) : "v"(a), "v"(b) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r0 ^ (uint32_t)(r1 >> 32) ^ x ^ y ^ (uint32_t)x64 ^ (uint32_t)(y64 >> 7);
}
static double base_ns[2];
template <int C>
static void run(uint32_t* out, int ncu) {
  printf("%-52s", names[C]);
  int k = 0;
  for (int wps : {2, 8}) {
    int blocks = ncu * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_mix<C><<<blocks, 256>>>(out, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 4; rep++) k_mix<C><<<blocks, 256>>>(out, 2 + rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
    double ns_per_block = ms * 1e6 / ((double)ITER * wps);       // per asm block per SIMD
    if (C == BASE) { base_ns[k] = ns_per_block; printf("  wps=%d: %7.2f ns per block (%5.3f ns per mad)              ", wps, ns_per_block, ns_per_block / 20); }
    else printf("  wps=%d: %7.2f ns per block, extra op = %5.2f x a mad", wps, ns_per_block, (ns_per_block - base_ns[k]) / 10 / (base_ns[k] / 20));
    k++;
  }
  printf("\n");
}
template <int C> static void run_all(uint32_t* out, int ncu) { run<C>(out, ncu); if constexpr (C + 1 < NCASE) run_all<C + 1>(out, ncu); }
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  arch=%s\n", p.name, p.multiProcessorCount, p.gcnArchName);
  uint32_t* out; CK(hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4));
  run_all<0>(out, p.multiProcessorCount);
  return 0;
}
