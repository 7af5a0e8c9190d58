// ubench_dep.hip — does a simple VALU instruction ON THE DEPENDENCY PATH between two multiply-accumulate chains cost more than the same
// instruction off it?  Blocks of 4 columns x (5 dependent v_mad_u64_u32 + one carry step), as fe_mul's low half is built:
//   IND    carry step = v_lshrrev_b64 of an unrelated register (off the path)
//   SHR    carry step = v_lshrrev_b64 acc, 28, acc   (mad -> shift -> mad: what the low half does)
//   SHRAND SHR plus a v_and_b32 of the previous column's low word (the limb mask)
//   MADC   carry step = v_mad_u64_u32 acc2, hi(acc), 16, 0 feeding the next column (mad -> mad: what the high half does)
//   TWO    two independent SHR streams interleaved column by column
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dep tools/ubench_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITER = 8192;
#define M5(acc) "v_mad_u64_u32 " acc ", vcc, %4, %5, " acc "\n\tv_mad_u64_u32 " acc ", vcc, %5, %4, " acc "\n\tv_mad_u64_u32 " acc ", vcc, %4, %4, " acc "\n\tv_mad_u64_u32 " acc ", vcc, %5, %5, " acc "\n\tv_mad_u64_u32 " acc ", vcc, %4, %5, " acc "\n\t"
enum { IND, SHR, SHRAND, MADC, TWO, TWOIND, NCASE };
static const char* names[] = {"4 x (5 mad, shift64 of another register)", "4 x (5 mad, shift64 of the sum)  [mad->shift->mad]", "4 x (5 mad, v_and + v_xor of the low word, shift64 of the sum)",
  "4 x (mad(hi, 16) of the previous sum, 5 mad) [mad->mad]", "two [mad->shift->mad] streams interleaved", "two streams, shifts of another register"};
static __device__ __forceinline__ void mad5(uint64_t& acc, uint32_t a, uint32_t b) {
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
static __device__ __forceinline__ void mad5c(uint64_t& acc, uint32_t hi, uint32_t a, uint32_t b) {     // first instruction: the carry as (upper register) x 16
  asm volatile("v_mad_u64_u32 %0, vcc, %3, 16, 0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "=&v"(acc) : "v"(a), "v"(b), "v"(hi) : "vcc");
}
static __device__ __forceinline__ void shr28(uint64_t& d, uint64_t s) { asm volatile("v_lshrrev_b64 %0, 28, %1" : "=v"(d) : "v"(s)); }
static __device__ __forceinline__ void and28(uint32_t& d, uint32_t s) { asm volatile("v_and_b32 %0, 0xfffffff, %1" : "=v"(d) : "v"(s)); }
template <int C>
__global__ void __launch_bounds__(256) k_dep(uint32_t* out, uint32_t seed) {
  uint32_t a = seed * 2654435761u + threadIdx.x, b = a ^ 0x9e3779b9u;
  uint64_t r0 = a, r1 = b, r2 = a ^ b, r3 = ~(uint64_t)a;
  uint32_t m0 = a, m1 = b;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int col = 0; col < 4; col++) {
      if constexpr (C == IND) { mad5(r0, a, b); shr28(r2, r2); }
      else if constexpr (C == SHR) { mad5(r0, a, b); shr28(r0, r0); }
      else if constexpr (C == SHRAND) { mad5(r0, a, b); and28(m0, (uint32_t)r0); m1 ^= m0; shr28(r0, r0); }
      else if constexpr (C == MADC) { uint64_t n; mad5c(n, (uint32_t)(r0 >> 32), a, b); r0 = n; }
      else if constexpr (C == TWO) { if (col & 1) { mad5(r1, a, b); shr28(r1, r1); } else { mad5(r0, a, b); shr28(r0, r0); } }
      else if constexpr (C == TWOIND) { if (col & 1) { mad5(r1, a, b); shr28(r3, r3); } else { mad5(r0, a, b); shr28(r2, r2); } }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r0 ^ (uint32_t)(r1 >> 32) ^ (uint32_t)r2 ^ (uint32_t)(r3 >> 7) ^ m1;
}
template <int C>
static void run(uint32_t* out, int ncu) {
  printf("%-56s", names[C]);
  const int ninstr = C == SHRAND ? 32 : 24;   // SHRAND: + v_and + v_xor per column
  for (int wps : {1, 2, 4, 8}) {
    int blocks = ncu * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_dep<C><<<blocks, 256>>>(out, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 4; rep++) k_dep<C><<<blocks, 256>>>(out, 2 + rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
    printf("  wps=%d: %6.2f ns/block (%4.2f/instr)", wps, ms * 1e6 / ((double)ITER * wps), ms * 1e6 / ((double)ITER * wps) / ninstr);
  }
  printf("\n");
}
template <int C> static void run_all(uint32_t* out, int ncu) { run<C>(out, ncu); if constexpr (C + 1 < NCASE) run_all<C + 1>(out, ncu); }
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  arch=%s   (ns per block PER SIMD: wall time x waves-per-SIMD / blocks issued)\n", p.name, p.multiProcessorCount, p.gcnArchName);
  uint32_t* out; CK(hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4));
  run_all<0>(out, p.multiProcessorCount);
  return 0;
}
