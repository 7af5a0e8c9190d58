// ubench_mfma_select.hip — an oblivious table look-up on the matrix cores (gfx950), checked against a plain gather and timed.
//
// The ct build's fixed-base products (msm.h fixed_base_acc_ct) must pick a table entry by a SECRET digit without the digit ever
// becoming an address: today every lane reads all 8 entries of a window and keeps one with masks (8 x (7 ds_read_b128 + 27 v_bfi)
// per window), and a window cannot be wider than that without the scan eating what the saved additions give.  But "pick row
// digit_l of a table for every lane l" is a matrix product with a SHARED operand -- selected = onehot(digits) x Table -- which is
// exactly what MFMA wants: D[byte][lane] = sum_e T[e][byte] * onehot[e][lane], one v_mfma_i32_32x32x32_i8 per 32 bytes x 32 lanes
// x 32 entries.  Byte values come through exactly (one nonzero term per sum; a byte >= 128 arrives as value - 256, whose low byte is
// the value).  The result tile has the lane on the column, so after one v_permlane32_swap per packed word every lane holds its own
// entry: 8 MFMAs + 96 byte packs + 32 swaps for a 128-byte entry out of 32, against 32 x (7 + 27) instructions for a
// scan of the same width.  Any consistent numbering of k works (A and B use the same one), so only the C/D map has to be right:
// col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_select.hip -o tools/ubench_mfma_select && tools/ubench_mfma_select
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#ifndef KSTEPS
#define KSTEPS 1
#endif
constexpr int ENTRIES = 32 * KSTEPS, TILES = 4, WORDS = 32;      // 4 tiles of 32 bytes = 128 bytes = 32 words per entry (27 used by a Niels entry)

// one-hot column of this lane's MFMA-B fragment: byte j of the 16 is 1 iff idx == 16 h + j
__device__ __forceinline__ v4i onehot16(int idx, int h) {
  const int rel = idx - 16 * h;                         // outside [0, 16): no bit
  const unsigned val = 1u << (8 * (rel & 3));
  const int w = rel >> 2;
  v4i b;
  b[0] = (w == 0) ? (int)val : 0; b[1] = (w == 1) ? (int)val : 0; b[2] = (w == 2) ? (int)val : 0; b[3] = (w == 3) ? (int)val : 0;
  return b;
}

// out[32 words] = table entry idx (all zero for idx < 0) of one window; tabA = that window's A-operand image: [tile][half][row][16]
__device__ __forceinline__ void mfma_select(uint32_t out[WORDS], const uint8_t* tabA, int idx) {
  const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
  const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)idx, (unsigned)idx, false, false);
  const int idx_lo = (int)sw[0], idx_hi = (int)sw[1];   // the digit of lane (l & 31) resp. (l & 31) + 32, in every lane
  v4i b_lo[KSTEPS], b_hi[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ks++) { b_lo[ks] = onehot16(idx_lo - 32 * ks, h); b_hi[ks] = onehot16(idx_hi - 32 * ks, h); }
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < TILES; t++) {
    v16i x = zero, y = zero;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
      const v4i a = *reinterpret_cast<const v4i*>(tabA + (((ks * TILES + t) * 2 + h) * 32 + r) * 16);
      x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_lo[ks], x, 0, 0, 0);      // columns = lanes 0..31's entries
      y = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_hi[ks], y, 0, 0, 0);      // columns = lanes 32..63's entries
    }
    // lane l < 32 owns column l of x, lane l >= 32 column l - 32 of y, and each holds half the rows of both.  row = (i & 3) +
    // 8 (i >> 2) + 4 half: pack the low bytes of four rows into a word first, then trade words (as msm.h mf_select does)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t xw = __builtin_amdgcn_perm((uint32_t)x[4 * q + 1], (uint32_t)x[4 * q], 0x0c0c0400u) |
                          (__builtin_amdgcn_perm((uint32_t)x[4 * q + 3], (uint32_t)x[4 * q + 2], 0x0c0c0400u) << 16);
      const uint32_t yw = __builtin_amdgcn_perm((uint32_t)y[4 * q + 1], (uint32_t)y[4 * q], 0x0c0c0400u) |
                          (__builtin_amdgcn_perm((uint32_t)y[4 * q + 3], (uint32_t)y[4 * q + 2], 0x0c0c0400u) << 16);
      const auto s = __builtin_amdgcn_permlane32_swap(xw, yw, false, false);
      out[8 * t + 2 * q] = s[0]; out[8 * t + 2 * q + 1] = s[1];      // bytes 8q .. 8q+3 from lane half 0, 8q+4 .. 8q+7 from half 1
    }
  }
}

__global__ void __launch_bounds__(256) k_select(const uint8_t* tabA, const int* idx, uint32_t* out, int reps) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  uint32_t w[WORDS], acc[WORDS];
  for (int i = 0; i < WORDS; i++) acc[i] = 0;
  int id = idx[gid];
  for (int rep = 0; rep < reps; rep++) {
    mfma_select(w, tabA, id);
    for (int i = 0; i < WORDS; i++) acc[i] ^= w[i] + (uint32_t)rep;
    id = (id + 7 + (int)(acc[0] & 1u) * 0) % ENTRIES;   // a new digit per repetition (timing loop only)
  }
  if (reps == 1) for (int i = 0; i < WORDS; i++) out[(size_t)gid * WORDS + i] = w[i];
  else out[(size_t)gid * WORDS] = acc[0] ^ acc[5] ^ acc[31];
}

// the scan the ct build uses today, at the same width, for the timing comparison: every lane reads all entries, keeps one with masks
__global__ void __launch_bounds__(256) k_scan(const uint32_t* tab /* [entry][32 words] */, const int* idx, uint32_t* out, int reps, int entries) {
  __shared__ uint32_t lds[32 * WORDS];
  for (int i = threadIdx.x; i < entries * WORDS; i += 256) lds[i] = tab[i];
  __syncthreads();
  const int gid = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc[WORDS];
  for (int i = 0; i < WORDS; i++) acc[i] = 0;
  int id = idx[gid];
  for (int rep = 0; rep < reps; rep++) {
    uint32_t w[WORDS];
    for (int i = 0; i < WORDS; i++) w[i] = 0;
    for (int e = 0; e < entries; e++) {
      uint32_t m = 0u - (uint32_t)(e == id);
      asm("" : "+v"(m));
      for (int i = 0; i < 28; i += 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(lds + e * WORDS + i);
        w[i] |= v.x & m; w[i + 1] |= v.y & m; w[i + 2] |= v.z & m; w[i + 3] |= v.w & m;
      }
    }
    for (int i = 0; i < WORDS; i++) acc[i] ^= w[i] + (uint32_t)rep;
    id = (id + 7) % entries;
  }
  out[(size_t)gid * WORDS] = acc[0] ^ acc[5] ^ acc[27];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int blocks = 2048, lanes = blocks * 256;
  std::vector<uint8_t> T(ENTRIES * 128), tabA(KSTEPS * TILES * 2 * 32 * 16);
  srand(7);
  for (auto& b : T) b = (uint8_t)rand();
  for (int ks = 0; ks < KSTEPS; ks++) for (int t = 0; t < TILES; t++) for (int h = 0; h < 2; h++) for (int r = 0; r < 32; r++) for (int j = 0; j < 16; j++)
    tabA[(((ks * TILES + t) * 2 + h) * 32 + r) * 16 + j] = T[(32 * ks + 16 * h + j) * 128 + 32 * t + r];
  std::vector<int> idx(lanes);
  for (int i = 0; i < lanes; i++) idx[i] = (rand() % (ENTRIES + 1)) - 1;          // -1 = no entry (digit 0)
  uint8_t *d_tabA, *d_T; int* d_idx; uint32_t* d_out;
  CK(hipMalloc(&d_tabA, tabA.size())); CK(hipMalloc(&d_T, T.size())); CK(hipMalloc(&d_idx, lanes * 4)); CK(hipMalloc(&d_out, (size_t)lanes * WORDS * 4));
  CK(hipMemcpy(d_tabA, tabA.data(), tabA.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_T, T.data(), T.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, idx.data(), lanes * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_select, dim3(blocks), dim3(256), 0, 0, d_tabA, d_idx, d_out, 1);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> out((size_t)lanes * WORDS);
  CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int l = 0; l < lanes; l++)
    for (int b = 0; b < 128; b++) {
      const uint8_t want = idx[l] < 0 ? 0 : T[idx[l] * 128 + b];
      const uint8_t got = (uint8_t)(out[(size_t)l * WORDS + b / 4] >> (8 * (b % 4)));
      if (want != got && bad++ < 5) printf("lane %d byte %d: got %02x want %02x (idx %d)\n", l, b, got, want, idx[l]);
    }
  printf("mfma select of %d lanes x 128 bytes out of %d entries: %s (%zu wrong bytes)\n", lanes, ENTRIES, bad ? "WRONG" : "equal to the gather", bad);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 256;
  for (int pass = 0; pass < 2; pass++) {
    float ms[4];
    for (int which = 0; which < 4; which++) {
      CK(hipEventRecord(e0, 0));
      if (which == 0) hipLaunchKernelGGL(k_select, dim3(blocks), dim3(256), 0, 0, d_tabA, d_idx, d_out, reps);
      else hipLaunchKernelGGL(k_scan, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint32_t*>(d_T), d_idx, d_out, reps, which == 1 ? 8 : which == 2 ? 16 : 32);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[which], e0, e1));
    }
    if (pass) printf("per look-up per wavefront: mfma select (%d entries) %.1f ns; masked scan of 8 / 16 / 32 entries from LDS %.1f / %.1f / %.1f ns\n",
                     ENTRIES, 1e6 * ms[0] / reps / (lanes / 64) * 1024, 1e6 * ms[1] / reps / (lanes / 64) * 1024, 1e6 * ms[2] / reps / (lanes / 64) * 1024, 1e6 * ms[3] / reps / (lanes / 64) * 1024);
  }
  return bad ? 1 : 0;
}
