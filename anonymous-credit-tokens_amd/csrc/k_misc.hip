// k_misc.hip — set-up kernels (fixed-base tables, point decoding, Params one-way map, key
// generation) and the device BLAKE3 transcript kernel.
#include "kernels.h"

namespace act {

// T[pos][e] = e * 2^(w*pos) * B as affine Niels (msm.h).  lane = (pos, e): w*pos doublings of B, a
// w-bit double-and-add, one inversion.  Runs once per context (cf. RistrettoBasepointTable::create,
// /root/reference/src/lib.rs:311-313).
__global__ void __launch_bounds__(256) k_build_table(const uint32_t* base_ext, uint32_t* table, uint32_t wbits) {
  uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= fb_windows(wbits) << wbits) return;
  uint32_t pos = gid >> wbits, e = gid & ((1u << wbits) - 1u);
  ge b = ge_load(base_ext);
  for (uint32_t i = 0; i < wbits * pos; i++) b = ge_double(b);
  ge acc = ge_identity();
  ge_cached bc = ge_to_cached(b);
  for (int bit = (int)wbits - 1; bit >= 0; bit--) {
    acc = ge_double(acc);
    if ((e >> bit) & 1u) acc = ge_add_cached(acc, bc);
  }
  fe zi = fe_invert(acc.Z);
  ge af; af.X = fe_mul(acc.X, zi); af.Y = fe_mul(acc.Y, zi); af.Z = fe_one(); af.T = fe_mul(af.X, af.Y);
  niels_store(table + (size_t)gid * NIELS_WORDS, niels_from_affine(af));
}
void launch_build_table(const uint32_t* base_ext, uint32_t* table, uint32_t wbits, hipStream_t s) {
  const uint32_t lanes = fb_windows(wbits) << wbits;
  hipLaunchKernelGGL(k_build_table, dim3((lanes + 255) / 256), dim3(256), 0, s, base_ext, table, wbits);
}

// T[pos][e-1] = e * 16^pos * B, e = 1..8, as affine Niels: the small tables the secret-scalar products scan in full (msm.h)
__global__ void __launch_bounds__(64) k_build_table_ct(const uint32_t* base_ext, uint32_t* table) {
  uint32_t gid = blockIdx.x * 64 + threadIdx.x;
  if (gid >= (uint32_t)(CT_WINDOWS * CT_ENTRIES)) return;
  uint32_t pos = gid / CT_ENTRIES, e = gid % CT_ENTRIES + 1;
  ge b = ge_load(base_ext);
  for (uint32_t i = 0; i < 4u * pos; i++) b = ge_double(b);
  ge acc = ge_identity();
  ge_cached bc = ge_to_cached(b);
  for (int bit = 3; bit >= 0; bit--) {
    acc = ge_double(acc);
    if ((e >> bit) & 1u) acc = ge_add_cached(acc, bc);
  }
  fe zi = fe_invert(acc.Z);
  ge af; af.X = fe_mul(acc.X, zi); af.Y = fe_mul(acc.Y, zi); af.Z = fe_one(); af.T = fe_mul(af.X, af.Y);
  niels_store(table + (size_t)gid * NIELS_WORDS, niels_from_affine(af));
}
void launch_build_table_ct(const uint32_t* base_ext, uint32_t* table, hipStream_t s) {
  hipLaunchKernelGGL(k_build_table_ct, dim3((CT_WINDOWS * CT_ENTRIES + 63) / 64), dim3(64), 0, s, base_ext, table);
}

// The matrix-core table image of a base (msm.h fixed_base_acc_mf): entry e = 1..MF_ENTRIES of window w is e * 2^(MF_WBITS w) * B as
// affine Niels, its 108 bytes scattered into the window's MFMA A-operand image: byte b of the entry at [K-step (e-1) / 32][tile b / 32]
// [lane half ((e-1) % 32) / 16][row b % 32][K-slot (e-1) % 16].  lane = (w, e).  Runs once per context.
__global__ void __launch_bounds__(64) k_build_table_mf(const uint32_t* base_ext, uint8_t* image) {
  const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
  if (gid >= (uint32_t)(MF_WINDOWS * MF_ENTRIES)) return;
  const uint32_t wd = gid / (uint32_t)MF_ENTRIES, e = gid % (uint32_t)MF_ENTRIES + 1u;
  ge b = ge_load(base_ext);
  for (uint32_t i = 0; i < (uint32_t)MF_WBITS * wd; i++) b = ge_double(b);
  ge acc = ge_identity();
  const ge_cached bc = ge_to_cached(b);
  for (int bit = MF_WBITS - 1; bit >= 0; bit--) {
    acc = ge_double(acc);
    if ((e >> bit) & 1u) acc = ge_add_cached(acc, bc);
  }
  fe zi = fe_invert(acc.Z);
  ge af; af.X = fe_mul(acc.X, zi); af.Y = fe_mul(acc.Y, zi); af.Z = fe_one(); af.T = fe_mul(af.X, af.Y);
  uint32_t words[NIELS_WORDS];
  niels_store(words, niels_from_affine(af));
  const uint32_t k = e - 1u, ks = k / 32u, h = (k % 32u) / 16u, j = k % 16u;
  uint8_t* win = image + (size_t)wd * (uint32_t)MF_WINDOW_BYTES;
  for (uint32_t byte = 0; byte < 128u; byte++) {
    const uint32_t t = byte / 32u, r = byte % 32u;
    win[(((ks * 4u + t) * 2u + h) * 32u + r) * 16u + j] = byte < 108u ? (uint8_t)(words[byte / 4u] >> (8u * (byte % 4u))) : (uint8_t)0;
  }
}
void launch_build_table_mf(const uint32_t* base_ext, uint8_t* image, hipStream_t s) {
  hipLaunchKernelGGL(k_build_table_mf, dim3((MF_WINDOWS * MF_ENTRIES + 63) / 64), dim3(64), 0, s, base_ext, image);
}

// out[0] = identity, out[1] = B / 2 as affine Niels, B the base of `table`: the prover works at half scale (k_prove.hip)
__global__ void k_half_point_table(FbTab table, uint32_t* out) {
  if (blockIdx.x || threadIdx.x) return;
  ge h = fixed_base_acc(ge_identity(), table, sc_half(sc_one()));
  fe zi = fe_invert(h.Z);
  ge af; af.X = fe_mul(h.X, zi); af.Y = fe_mul(h.Y, zi); af.Z = fe_one(); af.T = fe_mul(af.X, af.Y);
  niels_store(out, niels_from_affine(ge_identity()));
  niels_store(out + NIELS_WORDS, niels_from_affine(af));
}
void launch_half_point_table(FbTab table, uint32_t* out, hipStream_t s) { hipLaunchKernelGGL(k_half_point_table, dim3(1), dim3(64), 0, s, table, out); }

__global__ void __launch_bounds__(64) k_decode_points(const uint8_t* enc, uint32_t n, uint32_t* out_ext, uint32_t* ok) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8]; load8(w, enc + (size_t)i * 32);
  ge p; bool good = ristretto_decode(p, w);
  ge_store(out_ext + (size_t)i * GE_WORDS, p);
  ok[i] = good ? 1u : 0u;
}
void launch_decode_points(const uint8_t* enc, uint32_t n, uint32_t* out_ext, uint32_t* ok, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_decode_points, dim3((n + 63) / 64), dim3(64), 0, s, enc, n, out_ext, ok);
}

// RistrettoPoint::from_uniform_bytes (src/lib.rs:353, :261-263)
__global__ void __launch_bounds__(64) k_from_uniform(const uint8_t* in64, uint32_t n, uint8_t* out_enc) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16]; load8(w, in64 + (size_t)i * 64); load8(w + 8, in64 + (size_t)i * 64 + 32);
  ge p = ristretto_from_uniform(w);
  uint32_t e[8]; ristretto_encode(e, p); store8(out_enc + (size_t)i * 32, e);
}
void launch_from_uniform(const uint8_t* in64, uint32_t n, uint8_t* out_enc, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_from_uniform, dim3((n + 63) / 64), dim3(64), 0, s, in64, n, out_enc);
}

// PrivateKey::random (src/lib.rs:188-194): x <- 64 rng bytes; w = x * g
__global__ void __launch_bounds__(64) k_keygen(DevParams P, const uint8_t* rng64, uint32_t n, uint8_t* out_sk) {
  IssuerFb fb{P};                                                 // the issuer's key never addresses memory, in either build
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  sc x = i < n ? load_wide(rng64 + (size_t)i * 64) : sc_zero();
  ge w = fb.mul(ge_identity(), BASE_G, x);                        // every lane of the wavefront (IssuerFb: matrix-core look-up)
  if (i >= n) return;
  uint32_t e[8]; ristretto_encode(e, w);
  store_sc(out_sk + (size_t)i * 64, x); store8(out_sk + (size_t)i * 64 + 32, e);
}
void launch_keygen(const DevParams& P, const uint8_t* rng64, uint32_t n, uint8_t* out_sk, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_keygen, dim3((n + 63) / 64), dim3(64), 0, s, P, rng64, n, out_sk);
}
// PreIssuance::random (src/lib.rs:432-437): r then k, record r | k
__global__ void __launch_bounds__(256) k_pre_issuance_random(const uint8_t* rng, uint32_t n, uint8_t* out) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  store_sc(out + (size_t)i * 64, load_wide(rng + (size_t)i * 128));
  store_sc(out + (size_t)i * 64 + 32, load_wide(rng + (size_t)i * 128 + 64));
}
void launch_pre_issuance_random(const uint8_t* rng, uint32_t n, uint8_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_pre_issuance_random, dim3((n + 255) / 256), dim3(256), 0, s, rng, n, out);
}

// Test hook (act_debug_scalarmult_batch): out[i] = enc(s_i * P_i) through the production variable-base chain of the
// per-proof kernels (msm.h chain_b<1>) and the production decode / encode, so that third-party known answers for
// `RistrettoPoint * Scalar` (tests/golden/sodium_primitives.json) can be replayed on the device one operation at a time.
__global__ void __launch_bounds__(64, 2) k_debug_scalarmult(const uint8_t* pts, const uint8_t* scs, uint32_t n, uint32_t* pbk, uint8_t* out, uint8_t* status) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8]; load8(w, pts + (size_t)i * 32);
  ge p; bool ok = ristretto_decode(p, w);
  ge acc[1] = {ge_identity()};
  sc s[1] = {load_sc(scs + (size_t)i * 32)};
  chain_b<1>(acc, p, s, pbk + (size_t)i * BUCKET_WORDS);
  uint32_t e[8]; ristretto_encode(e, acc[0]);
  if (ok) store8(out + (size_t)i * 32, e); else zero8(out + (size_t)i * 32);
  status[i] = ok ? 0 : 255;
}
void launch_debug_scalarmult(const uint8_t* pts, const uint8_t* scs, uint32_t n, uint32_t* pbk, uint8_t* out, uint8_t* status, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_debug_scalarmult, dim3((n + 63) / 64), dim3(64), 0, s, pts, scs, n, pbk, out, status);
}

// ALU roofline probe: 8 register-resident accumulators per lane, each advanced by one block of 10 dependent v_mad_u64_u32
// per iteration -- the form a column of fe25519.h's multiplication has (one asm statement per column; single-instruction asm
// statements would measure the wait states the compiler puts between them, 4.7 instead of 4.06 cycles) -- no memory traffic
// inside the loop, every SIMD of the chip holding 8 wavefronts.  The multiply-accumulate the whole field arithmetic is built
// from cannot issue faster than this.
__global__ void __launch_bounds__(256) k_ubench_mad(uint32_t* out, uint32_t iters) {
  uint32_t a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x9e3779b9u;
  constexpr int CHAINS = UBENCH_MADS_PER_ITER / 10;
  uint64_t r[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) r[i] = ((uint64_t)a << 32 | b) + i;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\t"
                   "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\tv_mad_u64_u32 %0, vcc, %1, %1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %2, %0\n\t"
                   "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
  }
  uint64_t x = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x ^= r[i];
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}
void launch_ubench_mad(uint32_t* out, uint32_t blocks, uint32_t iters, hipStream_t s) { hipLaunchKernelGGL(k_ubench_mad, dim3(blocks), dim3(256), 0, s, out, iters); }

// Random-read roofline probe: what the scalar-addressed look-ups into the wide fixed-base tables (msm.h fixed_base_acc: one 128-byte
// affine-Niels entry per window, 112 bytes of it read as seven 16-byte loads, every entry on a different line of a 23.6 GB table)
// can get from HBM at best.  Every lane reads `iters` pseudo-random lines of `lines` (a power of two; xorshift per lane, so no two
// lanes share a line more than by chance), IN_FLIGHT independent entries at a time (fixed_base_acc's software pipeline has two).
// The grid is `waves_per_simd` blocks of 256 threads per CU, so the lines in flight per CU can be held where the kernels hold
// theirs (k_prove_bits / k_spend_bits run two wavefronts per SIMD): with many more in flight the seven loads of an entry miss in
// the vector cache separately and every line is fetched several times.
template <int IN_FLIGHT>
__global__ void __launch_bounds__(256) k_ubench_random_read(const uint4* buf, uint64_t lines, uint32_t iters, uint32_t* out) {
  uint64_t x = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t it = 0; it < iters; it += IN_FLIGHT) {
    const uint4* q[IN_FLIGHT];
#pragma unroll
    for (int k = 0; k < IN_FLIGHT; k++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; q[k] = buf + (x & (lines - 1)) * 8; }
#pragma unroll
    for (int k = 0; k < IN_FLIGHT; k++)
#pragma unroll
      for (int i = 0; i < 7; i++) { const uint4 v = q[k][i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
void launch_ubench_random_read(const uint32_t* buf, uint64_t lines, uint32_t blocks, uint32_t iters, int in_flight, uint32_t* out, hipStream_t s) {
  const uint4* b = reinterpret_cast<const uint4*>(buf);
  if (in_flight <= 1) hipLaunchKernelGGL(k_ubench_random_read<1>, dim3(blocks), dim3(256), 0, s, b, lines, iters, out);
  else if (in_flight == 2) hipLaunchKernelGGL(k_ubench_random_read<2>, dim3(blocks), dim3(256), 0, s, b, lines, iters, out);
  else hipLaunchKernelGGL(k_ubench_random_read<4>, dim3(blocks), dim3(256), 0, s, b, lines, iters, out);
}

// Seeded prover rng (act_prove_spend_seeded_batch): lane i's generator is the BLAKE3 XOF of seed | u64_le(first_lane + i) -- what
// `blake3::Hasher::new().update(seed).update(&lane.to_le_bytes()).finalize_xof()` yields, read sequentially by every
// Scalar::random (64 bytes each, /root/reference/src/lib.rs:978-1058).  An XOF block is one compression of the same 40-byte root
// block with the output-block counter t, so the 64 (4L + 12) bytes of a lane are 4L + 12 independent compressions: thread = (lane, t).
__global__ void __launch_bounds__(256) k_xof_expand(const uint32_t* seed, uint64_t first_lane, uint32_t n, uint32_t blocks_per_lane, uint8_t* out) {
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = (uint32_t)(gid / blocks_per_lane), t = (uint32_t)(gid % blocks_per_lane);
  if (lane >= n) return;
  uint32_t m[16], cv[8], o[16];
  for (int i = 0; i < 8; i++) { m[i] = seed[i]; cv[i] = b3_iv(i); }
  const uint64_t id = first_lane + lane;
  m[8] = (uint32_t)id; m[9] = (uint32_t)(id >> 32);
  for (int i = 10; i < 16; i++) m[i] = 0;
  b3_compress(o, cv, m, t, 0u, 40u, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
  uint4* q = reinterpret_cast<uint4*>(out + ((size_t)lane * blocks_per_lane + t) * 64);
  q[0] = make_uint4(o[0], o[1], o[2], o[3]); q[1] = make_uint4(o[4], o[5], o[6], o[7]);
  q[2] = make_uint4(o[8], o[9], o[10], o[11]); q[3] = make_uint4(o[12], o[13], o[14], o[15]);
}
void launch_xof_expand(const uint32_t* d_seed, uint64_t first_lane, uint32_t n, uint32_t blocks_per_lane, uint8_t* out, hipStream_t s) {
  if (!n) return;
  const uint64_t threads = (uint64_t)n * blocks_per_lane;
  hipLaunchKernelGGL(k_xof_expand, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_seed, first_lane, n, blocks_per_lane, out);
}

// One wavefront that does nothing for `ticks` of the 100 MHz constant-rate counter (s_memrealtime).  act_ctx_create runs one on each
// of the context's two streams at the same time to learn whether the HIP runtime gave them different hardware queues (engine.hip
// streams_overlap): two of these take one `ticks` when the streams run side by side and two when they share a queue.
__global__ void __launch_bounds__(64) k_spin(uint32_t ticks) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void launch_spin(uint32_t ticks, hipStream_t s) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks); }

// out[i] = base + i: the rng slice index of lane i in ACT_RNG_PER_LANE mode (no host round trip, so a chunk's launches stay asynchronous)
__global__ void __launch_bounds__(256) k_iota(uint32_t* out, uint32_t n, uint32_t base) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = base + i;
}
void launch_iota(uint32_t* out, uint32_t n, uint32_t base, hipStream_t s) { if (n) hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, s, out, n, base); }

// Transcript::challenge's hash (src/transcript.rs:149-152): lane = message, 64 XOF bytes out.
__global__ void __launch_bounds__(64) k_hash_xof(HashArgs a) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  uint32_t o[16];
  uint32_t len = a.len_per_lane ? a.len_per_lane[i] : a.len;
  b3_hash_xof64(o, reinterpret_cast<const uint32_t*>(a.msg + (size_t)i * a.stride), len);
  uint4* q = reinterpret_cast<uint4*>(a.xof + (size_t)i * 16);
  q[0] = make_uint4(o[0], o[1], o[2], o[3]); q[1] = make_uint4(o[4], o[5], o[6], o[7]);
  q[2] = make_uint4(o[8], o[9], o[10], o[11]); q[3] = make_uint4(o[12], o[13], o[14], o[15]);
}
void launch_hash(const HashArgs& a, hipStream_t s) { if (a.n) hipLaunchKernelGGL(k_hash_xof, dim3((a.n + 63) / 64), dim3(64), 0, s, a); }

// The same hash with SIXTEEN lanes per message, for calls too small to fill the chip with one lane each (the small-batch schedule):
// the 1 KiB chunks of a BLAKE3 input are independent until the tree is folded, so lane c of a message's group computes the chaining
// values of chunks c, c + 16, ... (every chunk but the last), the group meets in LDS, and its first lane folds the tree exactly as the
// one-lane routine does (same code: b3_hash_xof64_with).  A 15 784-byte "spend" transcript: 16 + ~22 compressions deep instead of 262.
constexpr int HASH_PAR_LANES = 16, HASH_PAR_MAX_CHUNKS = 64;      // messages up to 64 KiB
__global__ void __launch_bounds__(64) k_hash_xof_par(HashArgs a) {
  __shared__ uint32_t cvs[64 / HASH_PAR_LANES][HASH_PAR_MAX_CHUNKS][8];
  const uint32_t g = threadIdx.x / HASH_PAR_LANES, lane = threadIdx.x % HASH_PAR_LANES;
  const uint32_t i = blockIdx.x * (64 / HASH_PAR_LANES) + g;
  const bool live = i < a.n;
  const uint32_t len = live ? (a.len_per_lane ? a.len_per_lane[i] : a.len) : 0u;
  const uint32_t* msg = reinterpret_cast<const uint32_t*>(a.msg + (size_t)(live ? i : 0) * a.stride);
  const uint32_t nchunks = len ? (len + 1023u) >> 10 : 1u;
  for (uint32_t c = lane; c + 1 < nchunks; c += HASH_PAR_LANES) b3_chunk_cv(cvs[g][c], msg, len, c);
  __syncthreads();
  if (!live || lane != 0) return;
  uint32_t o[16];
  b3_hash_xof64_with(o, msg, len, [&](uint32_t c, uint32_t* cv) { for (int k = 0; k < 8; k++) cv[k] = cvs[g][c][k]; });
  uint4* q = reinterpret_cast<uint4*>(a.xof + (size_t)i * 16);
  q[0] = make_uint4(o[0], o[1], o[2], o[3]); q[1] = make_uint4(o[4], o[5], o[6], o[7]);
  q[2] = make_uint4(o[8], o[9], o[10], o[11]); q[3] = make_uint4(o[12], o[13], o[14], o[15]);
}
void launch_hash_par(const HashArgs& a, hipStream_t s) {
  if (!a.n) return;
  if (a.len > (uint32_t)HASH_PAR_MAX_CHUNKS * 1024u || a.len_per_lane) { launch_hash(a, s); return; }
  hipLaunchKernelGGL(k_hash_xof_par, dim3((a.n + 3) / 4), dim3(64), 0, s, a);
}

}  // namespace act
