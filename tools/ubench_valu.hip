// Instruction-rate microbenchmark for gfx950 (MI355X): which VALU forms can
// carry 255-bit modular arithmetic fastest?  Not part of the product path.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip
//   ./tools/ubench_valu
//
// Each test runs ITER iterations of 8 independent dependency chains of one
// instruction (inline asm so hipcc cannot rewrite it) and reports shader
// cycles per wave-instruction on one SIMD, at 1/2/4 waves per SIMD
// (s_memtime deltas of wave 0 of block 0, every CU busy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITER = 16384;

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, ADD_U32, ADDC_CHAIN,
          FMA_F64, MUL_F64, ADD_F64, FMA_F32, PK_FMA_F32, LSHL_ADD, ALIGNBIT, CNDMASK, MAD_I64_I32,
          MAD_MIX_ADDC, DOT4_U8, DOT2_U16, ADD3_U32, LSHLREV_B64, CNDMASK_SGPR, BFI, AND_B32, SUB_U32, XOR_B32, LSHL_ADD_U64, LSHRREV_B64, LSHRREV_B32, CNDMASK_VCC_SET, MAD_U64_SGPR, MAD_U64_ALT, MAD_AND_MIX, MAD_SHIFT_MIX, MAD_DEP10, MAD_DEP10_SHIFT_AND, MAD_2X10, MAD_DEP40, MAD10_AND10, AND20, MAD10_SHL10, NOPS };

static const char* op_name[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
  "v_add_u32", "v_add_co+v_addc_co (pair)", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_pk_fma_f32",
  "v_lshl_add_u32", "v_alignbit_b32", "v_cndmask_b32", "v_mad_i64_i32", "v_mad_u64_u32+v_addc (pair)",
  "v_dot4_u32_u8", "v_dot2_u32_u16", "v_add3_u32", "v_lshlrev_b64", "v_cndmask_b32 (sgpr mask)", "v_bfi_b32", "v_and_b32", "v_sub_u32", "v_xor_b32",
  "v_lshl_add_u64", "v_lshrrev_b64", "v_lshrrev_b32", "v_cndmask_b32 (vcc set by v_cmp)",
  "v_mad_u64_u32 sdst=s[10:11]", "v_mad_u64_u32 sdst alternating", "5 mad + 1 v_and (per instr)", "5 mad + 1 v_lshrrev_b64 (per instr)", "10 dependent mad per asm block", "10 dep mad + shift64 + and (per instr)", "2 chains x 10 mad interleaved per block", "40 dependent mad per asm block", "10 mad + 10 v_and interleaved (per instr)", "20 v_and on 2 regs per block", "10 mad + 10 v_lshlrev_b32 (per instr)"};

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
  uint32_t a = seed * 2654435761u + threadIdx.x, b = a ^ 0x9e3779b9u;
  uint64_t r[8];
  double d[8];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { r[i] = ((uint64_t)a << 32 | b) + i; d[i] = 1.0 + i * 1e-9 + a * 1e-12; f[i] = 1.0f + i; }
  double da = 1.0000001, db = 1e-9;
  float fa = 1.0001f, fb = 1e-3f;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f pf[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { pf[i].x = f[i]; pf[i].y = f[i] + 0.5f; }
  v2f pfa = {fa, fa}, pfb = {fb, fb};
  unsigned long long smask = __ballot((threadIdx.x & 3) != 0);
  uint64_t ab64 = ((uint64_t)a << 32) | b;
  if constexpr (OP == CNDMASK_VCC_SET) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t lo = (uint32_t)r[i], hi = (uint32_t)(r[i] >> 32);
      if constexpr (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      else if constexpr (OP == MAD_U64_SGPR) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "s10", "s11");
      else if constexpr (OP == MAD_U64_ALT) {
        if (i & 1) asm volatile("v_mad_u64_u32 %0, s[12:13], %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "s12", "s13");
        else if (i & 2) asm volatile("v_mad_u64_u32 %0, s[14:15], %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "s14", "s15");
        else asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "s10", "s11");
      }
      else if constexpr (OP == MAD_AND_MIX) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1" : "+v"(r[i]), "+v"(hi) : "v"(a), "v"(b) : "vcc"); r[i] += hi; }
      else if constexpr (OP == MAD_SHIFT_MIX) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_lshrrev_b64 %0, 3, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"); }
      else if constexpr (OP == MAD_DEP10) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"); }
      else if constexpr (OP == MAD_DEP10_SHIFT_AND) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %0, vcc, %3, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %3, %0\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshrrev_b64 %0, 26, %0\n\tv_and_b32 %1, 0x3ffffff, %1" : "+v"(r[i]), "+v"(lo) : "v"(a), "v"(b) : "vcc"); f[i] += (float)lo; }
      else if constexpr (OP == MAD_DEP40) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"); }
      else if constexpr (OP == MAD_2X10) { if (i < 4) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %3, %2, %1" : "+v"(r[i]), "+v"(r[i + 4]) : "v"(a), "v"(b) : "vcc"); }
      else if constexpr (OP == MAD10_AND10) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_and_b32 %1, 0x3ffffff, %1" : "+v"(r[i]), "+v"(lo) : "v"(a), "v"(b) : "vcc"); f[i] += (float)lo; }
      else if constexpr (OP == AND20) { asm volatile("v_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1\n\tv_and_b32 %0, 0x3ffffff, %0\n\tv_and_b32 %1, 0x3ffffff, %1" : "+v"(hi), "+v"(lo) : : ); r[i] = (uint64_t)hi << 32 | lo; }
      else if constexpr (OP == MAD10_SHL10) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshlrev_b32 %1, 1, %1" : "+v"(r[i]), "+v"(lo) : "v"(a), "v"(b) : "vcc"); f[i] += (float)lo; }
      else if constexpr (OP == MAD_I64_I32) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      else if constexpr (OP == MUL_LO_U32) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == MUL_HI_U32) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == MAD_U32_U24) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); r[i] = lo; }
      else if constexpr (OP == MUL_HI_U32_U24) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == ADD_U32) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == ADD3_U32) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); r[i] = lo; }
      else if constexpr (OP == ADDC_CHAIN) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc"); r[i] = (uint64_t)hi << 32 | lo; }
      else if constexpr (OP == MAD_MIX_ADDC) { asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(r[i]), "+v"(hi) : "v"(a), "v"(b) : "vcc"); }
      else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
      else if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
      else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
      else if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pf[i]) : "v"(pfa), "v"(pfb));
      else if constexpr (OP == LSHL_ADD) { asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == ALIGNBIT) { asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == CNDMASK) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == DOT4_U8) { asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); r[i] = lo; }
      else if constexpr (OP == DOT2_U16) { asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); r[i] = lo; }
      else if constexpr (OP == LSHLREV_B64) { asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(r[i])); }
      else if constexpr (OP == CNDMASK_SGPR) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "s"(smask)); r[i] = lo; }
      else if constexpr (OP == CNDMASK_VCC_SET) { asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(lo) : "v"(a) : ); r[i] = lo; }
      else if constexpr (OP == BFI) { asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); r[i] = lo; }
      else if constexpr (OP == AND_B32) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == SUB_U32) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == XOR_B32) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo) : "v"(a)); r[i] = lo; }
      else if constexpr (OP == LSHL_ADD_U64) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[i]) : "v"(ab64)); }
      else if constexpr (OP == LSHRREV_B64) { asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(r[i])); }
      else if constexpr (OP == LSHRREV_B32) { asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(lo)); r[i] = lo + a; }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0; double dacc = 0; float facc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { acc ^= r[i]; dacc += d[i]; facc += f[i] + pf[i].x + pf[i].y; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ (uint32_t)dacc ^ (uint32_t)facc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(uint32_t* out, unsigned long long* cyc, int ncu) {
  printf("%-30s", op_name[OP]);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD = blocks of 256 threads per CU
    int blocks = ncu * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<OP><<<blocks, 256>>>(out, cyc, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_rate<OP><<<blocks, 256>>>(out, cyc, 2);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
    double ninstr = (double)ITER * 8 * ((OP == ADDC_CHAIN || OP == MAD_MIX_ADDC) ? 2 : (OP == MAD_AND_MIX || OP == MAD_SHIFT_MIX) ? 6 : OP == MAD_DEP10 ? 10 : OP == MAD_DEP10_SHIFT_AND ? 12 : OP == MAD_DEP40 ? 40 : OP == MAD_2X10 ? 10 : (OP == MAD10_AND10 || OP == AND20 || OP == MAD10_SHL10) ? 20 : 1);
    // s_memtime ticks at a constant 100 MHz on gfx9 — so also derive cycles from wall time at the reported clock.
    double wave_instr_per_simd = ninstr * wps;   // instructions issued on one SIMD
    printf("  wps=%d: %7.3f ms %5.2f ns/instr/SIMD %5.2f cyc/instr/SIMD", wps, ms, ms * 1e6 / wave_instr_per_simd, (double)mx / wave_instr_per_simd);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  arch=%s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);
  int ncu = p.multiProcessorCount;
  uint32_t* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)ncu * 16 * 256 * 4)); CK(hipMalloc(&cyc, (size_t)ncu * 16 * 8));
  printf("ns/instr/SIMD x clock(GHz) = cycles per wave64 instruction on one SIMD\n");
  run<FMA_F32>(out, cyc, ncu);
  run<PK_FMA_F32>(out, cyc, ncu);
  run<ADD_U32>(out, cyc, ncu);
  run<ADD3_U32>(out, cyc, ncu);
  run<ADDC_CHAIN>(out, cyc, ncu);
  run<LSHL_ADD>(out, cyc, ncu);
  run<ALIGNBIT>(out, cyc, ncu);
  run<CNDMASK>(out, cyc, ncu);
  run<LSHLREV_B64>(out, cyc, ncu);
  run<MAD_U32_U24>(out, cyc, ncu);
  run<MUL_HI_U32_U24>(out, cyc, ncu);
  run<MUL_LO_U32>(out, cyc, ncu);
  run<MUL_HI_U32>(out, cyc, ncu);
  run<MAD_U64_U32>(out, cyc, ncu);
  run<MAD_I64_I32>(out, cyc, ncu);
  run<MAD_U64_SGPR>(out, cyc, ncu);
  run<MAD_U64_ALT>(out, cyc, ncu);
  run<MAD_AND_MIX>(out, cyc, ncu);
  run<MAD_SHIFT_MIX>(out, cyc, ncu);
  run<MAD_DEP10>(out, cyc, ncu);
  run<MAD_DEP10_SHIFT_AND>(out, cyc, ncu);
  run<MAD_2X10>(out, cyc, ncu);
  run<MAD_DEP40>(out, cyc, ncu);
  run<MAD10_AND10>(out, cyc, ncu);
  run<AND20>(out, cyc, ncu);
  run<MAD10_SHL10>(out, cyc, ncu);
  run<MAD_MIX_ADDC>(out, cyc, ncu);
  run<DOT4_U8>(out, cyc, ncu);
  run<DOT2_U16>(out, cyc, ncu);
  run<CNDMASK_SGPR>(out, cyc, ncu);
  run<CNDMASK_VCC_SET>(out, cyc, ncu);
  run<BFI>(out, cyc, ncu);
  run<AND_B32>(out, cyc, ncu);
  run<SUB_U32>(out, cyc, ncu);
  run<XOR_B32>(out, cyc, ncu);
  run<LSHL_ADD_U64>(out, cyc, ncu);
  run<LSHRREV_B64>(out, cyc, ncu);
  run<LSHRREV_B32>(out, cyc, ncu);
  run<FMA_F64>(out, cyc, ncu);
  run<MUL_F64>(out, cyc, ncu);
  run<ADD_F64>(out, cyc, ncu);
  return 0;
}
