// host_hash.cpp — host-side BLAKE3 for the host-transcript mode (the reference hashes every transcript on the
// CPU, /root/reference/src/transcript.rs:149-152): sixteen equal-length messages hashed in lockstep, one per
// 32-bit SIMD lane (GCC vector extensions; target_clones picks AVX-512, AVX2 or baseline SSE2 at load time).
// All sixteen messages share one control flow because they have the same length — every "spend" transcript of
// a context is 184 + 40*(6+3L) bytes — so the tree/finalisation logic of blake3_hd.h carries over unchanged with
// the word type widened.  Compiled with g++ (not hipcc); results are checked against the scalar routine and
// against upstream BLAKE3 vectors in tests/.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef uint32_t v16 __attribute__((vector_size(64)));

namespace {

const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

inline __attribute__((always_inline)) v16 splat(uint32_t x) { return v16{x, x, x, x, x, x, x, x, x, x, x, x, x, x, x, x}; }
inline __attribute__((always_inline)) v16 rotr(v16 x, int n) { return (x >> n) | (x << (32 - n)); }

#define G16(a, b, c, d, mx, my)                                            \
  do {                                                                     \
    a = a + b + (mx); d = rotr(d ^ a, 16); c = c + d; b = rotr(b ^ c, 12); \
    a = a + b + (my); d = rotr(d ^ a, 8);  c = c + d; b = rotr(b ^ c, 7);  \
  } while (0)
#define ROUND16(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15)      \
  do {                                                                                     \
    G16(v0, v4, v8, v12, m0, m1); G16(v1, v5, v9, v13, m2, m3);                            \
    G16(v2, v6, v10, v14, m4, m5); G16(v3, v7, v11, v15, m6, m7);                          \
    G16(v0, v5, v10, v15, m8, m9); G16(v1, v6, v11, v12, m10, m11);                        \
    G16(v2, v7, v8, v13, m12, m13); G16(v3, v4, v9, v14, m14, m15);                        \
  } while (0)

inline __attribute__((always_inline)) void compress16(v16 out[16], const v16 cv[8], const v16 m[16], uint32_t counter, uint32_t blen, uint32_t flags) {
  v16 v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
  v16 v8 = splat(IV[0]), v9 = splat(IV[1]), v10 = splat(IV[2]), v11 = splat(IV[3]);
  v16 v12 = splat(counter), v13 = splat(0), v14 = splat(blen), v15 = splat(flags);
  ROUND16(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
  ROUND16(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]);
  ROUND16(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1]);
  ROUND16(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6]);
  ROUND16(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4]);
  ROUND16(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7]);
  ROUND16(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13]);
  out[0] = v0 ^ v8; out[1] = v1 ^ v9; out[2] = v2 ^ v10; out[3] = v3 ^ v11;
  out[4] = v4 ^ v12; out[5] = v5 ^ v13; out[6] = v6 ^ v14; out[7] = v7 ^ v15;
  out[8] = v8 ^ cv[0]; out[9] = v9 ^ cv[1]; out[10] = v10 ^ cv[2]; out[11] = v11 ^ cv[3];
  out[12] = v12 ^ cv[4]; out[13] = v13 ^ cv[5]; out[14] = v14 ^ cv[6]; out[15] = v15 ^ cv[7];
}

// block at byte offset `off` of each of the 16 messages (message i at msgs + i*stride), transposed into 16 word-vectors
inline __attribute__((always_inline)) uint32_t load_block16(v16 m[16], const uint8_t* msgs, size_t stride, uint32_t len, uint32_t off) {
  uint32_t remain = len > off ? len - off : 0u, blen = remain < 64u ? remain : 64u;
  uint32_t tmp[16][16];     // [message][word]
  for (int i = 0; i < 16; i++) {
    if (blen == 64u) memcpy(tmp[i], msgs + (size_t)i * stride + off, 64);
    else { memset(tmp[i], 0, 64); if (blen) memcpy(tmp[i], msgs + (size_t)i * stride + off, blen); }
  }
  for (int w = 0; w < 16; w++)
    m[w] = v16{tmp[0][w], tmp[1][w], tmp[2][w], tmp[3][w], tmp[4][w], tmp[5][w], tmp[6][w], tmp[7][w],
               tmp[8][w], tmp[9][w], tmp[10][w], tmp[11][w], tmp[12][w], tmp[13][w], tmp[14][w], tmp[15][w]};
  return blen;
}

}  // namespace

// xof[i*16 .. i*16+16) = first 64 XOF bytes of BLAKE3(msgs + i*stride, len), i < 16.  len < 256 KiB.
extern "C" __attribute__((target_clones("avx512f", "avx2", "default")))
void act_host_b3_xof64_x16(const uint8_t* msgs, size_t stride, uint32_t len, uint32_t* xof) {
  uint32_t nchunks = len ? (len + 1023u) >> 10 : 1u;
  v16 stack[8][8];
  int sp = 0;
  v16 cv[8], m[16], o[16], iv[8];
  for (int i = 0; i < 8; i++) iv[i] = splat(IV[i]);
  for (uint32_t c = 0; c + 1 < nchunks; c++) {
    for (int i = 0; i < 8; i++) cv[i] = iv[i];
    for (uint32_t b = 0; b < 16; b++) {
      load_block16(m, msgs, stride, len, c * 1024u + b * 64u);
      compress16(o, cv, m, c, 64u, (b == 0 ? CHUNK_START : 0u) | (b == 15 ? CHUNK_END : 0u));
      for (int i = 0; i < 8; i++) cv[i] = o[i];
    }
    uint32_t t = c + 1;
    while ((t & 1u) == 0u) {
      sp--;
      for (int i = 0; i < 8; i++) { m[i] = stack[sp][i]; m[8 + i] = cv[i]; }
      compress16(o, iv, m, 0u, 64u, PARENT);
      for (int i = 0; i < 8; i++) cv[i] = o[i];
      t >>= 1;
    }
    for (int i = 0; i < 8; i++) stack[sp][i] = cv[i];
    sp++;
  }
  uint32_t c = nchunks - 1, base = c * 1024u, clen = len - base;
  uint32_t nblocks = clen ? (clen + 63u) >> 6 : 1u;
  for (int i = 0; i < 8; i++) cv[i] = iv[i];
  uint32_t blen = 0, fl = 0;
  for (uint32_t b = 0; b < nblocks; b++) {
    blen = load_block16(m, msgs, stride, len, base + b * 64u);
    fl = (b == 0 ? CHUNK_START : 0u) | (b == nblocks - 1 ? CHUNK_END : 0u);
    if (b + 1 < nblocks) { compress16(o, cv, m, c, 64u, fl); for (int i = 0; i < 8; i++) cv[i] = o[i]; }
  }
  uint32_t ctr = c;
  while (sp > 0) {
    compress16(o, cv, m, ctr, blen, fl);
    sp--;
    for (int i = 0; i < 8; i++) { m[i] = stack[sp][i]; m[8 + i] = o[i]; }
    for (int i = 0; i < 8; i++) cv[i] = iv[i];
    ctr = 0; blen = 64u; fl = PARENT;
  }
  compress16(o, cv, m, 0u, blen, fl | ROOT);
  for (int w = 0; w < 16; w++) for (int i = 0; i < 16; i++) xof[i * 16 + w] = o[w][i];
}
