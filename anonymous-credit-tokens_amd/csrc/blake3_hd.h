// blake3_hd.h — BLAKE3 (default hash mode, 64-byte XOF output) for the Fiat–Shamir transcripts
// of /root/reference/src/transcript.rs:54-154 (blake3 1.8.2 there; not vendored).  Written from
// the BLAKE3 specification; one message per lane on the device, or per host thread when the
// context runs in host-transcript mode (the same source compiles for both, so the two modes
// cannot drift apart).  Messages are little-endian u32 words in memory (4-byte aligned; the
// buffer must be readable up to the next word boundary).
#pragma once
#include "fe25519.h"   // ACT_HD

namespace act {

enum : uint32_t { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };

ACT_HD uint32_t b3_iv(int i) {
  constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  return IV[i];
}
ACT_HD uint32_t b3_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

#define ACT_B3_G(a, b, c, d, mx, my)                                             \
  do {                                                                           \
    a = a + b + (mx); d = b3_rotr(d ^ a, 16); c = c + d; b = b3_rotr(b ^ c, 12); \
    a = a + b + (my); d = b3_rotr(d ^ a, 8);  c = c + d; b = b3_rotr(b ^ c, 7);  \
  } while (0)
#define ACT_B3_ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
  do {                                                                                     \
    ACT_B3_G(v0, v4, v8, v12, m0, m1); ACT_B3_G(v1, v5, v9, v13, m2, m3);                  \
    ACT_B3_G(v2, v6, v10, v14, m4, m5); ACT_B3_G(v3, v7, v11, v15, m6, m7);                \
    ACT_B3_G(v0, v5, v10, v15, m8, m9); ACT_B3_G(v1, v6, v11, v12, m10, m11);              \
    ACT_B3_G(v2, v7, v8, v13, m12, m13); ACT_B3_G(v3, v4, v9, v14, m14, m15);              \
  } while (0)

// out[0..8) = new chaining value, out[8..16) = upper half (XOF / root output only)
ACT_HD void b3_compress(uint32_t out[16], const uint32_t cv[8], const uint32_t m[16], uint32_t counter_lo, uint32_t counter_hi,
                        uint32_t blen, uint32_t flags) {
  uint32_t v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
  uint32_t v8 = b3_iv(0), v9 = b3_iv(1), v10 = b3_iv(2), v11 = b3_iv(3), v12 = counter_lo, v13 = counter_hi, v14 = blen, v15 = flags;
  // the message permutation {2,6,3,10,7,0,4,13,1,11,12,5,9,14,15,8} applied r times, unrolled
  ACT_B3_ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]);
  ACT_B3_ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]);
  ACT_B3_ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1]);
  ACT_B3_ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6]);
  ACT_B3_ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4]);
  ACT_B3_ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7]);
  ACT_B3_ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13]);
  out[0] = v0 ^ v8; out[1] = v1 ^ v9; out[2] = v2 ^ v10; out[3] = v3 ^ v11;
  out[4] = v4 ^ v12; out[5] = v5 ^ v13; out[6] = v6 ^ v14; out[7] = v7 ^ v15;
  out[8] = v8 ^ cv[0]; out[9] = v9 ^ cv[1]; out[10] = v10 ^ cv[2]; out[11] = v11 ^ cv[3];
  out[12] = v12 ^ cv[4]; out[13] = v13 ^ cv[5]; out[14] = v14 ^ cv[6]; out[15] = v15 ^ cv[7];
}

// loads block `b` (64 bytes) of a message of `len` bytes; bytes past the end read as zero
ACT_HD uint32_t b3_load_block(uint32_t m[16], const uint32_t* msg, uint32_t len, uint32_t byte_off) {
  uint32_t remain = len > byte_off ? len - byte_off : 0u;
  uint32_t blen = remain < 64u ? remain : 64u;
  const uint32_t* p = msg + (byte_off >> 2);
  for (int i = 0; i < 16; i++) {
    uint32_t lo = 4u * (uint32_t)i;
    uint32_t w = (lo < blen) ? p[i] : 0u;
    uint32_t valid = blen - lo;                       // only meaningful when lo < blen
    if (lo < blen && valid < 4u) w &= (1u << (8u * valid)) - 1u;
    m[i] = w;
  }
  return blen;
}

// chaining value of chunk c of a message that has more chunks after it (a full 1 KiB chunk that cannot be the root)
ACT_HD void b3_chunk_cv(uint32_t cv[8], const uint32_t* msg, uint32_t len, uint32_t c) {
  uint32_t m[16], o[16];
  for (int i = 0; i < 8; i++) cv[i] = b3_iv(i);
  for (uint32_t b = 0; b < 16; b++) {
    b3_load_block(m, msg, len, c * 1024u + b * 64u);
    uint32_t fl = (b == 0 ? B3_CHUNK_START : 0u) | (b == 15 ? B3_CHUNK_END : 0u);
    b3_compress(o, cv, m, c, 0u, 64u, fl);
    for (int i = 0; i < 8; i++) cv[i] = o[i];
  }
}

// 64 bytes of root output (finalize_xof().fill(&mut [0u8; 64]), src/transcript.rs:150-152).  `chunk_cv(c, cv)` supplies the chaining
// value of every chunk but the last: computed on the spot (b3_hash_xof64) or by other lanes beforehand (k_hash_xof_par, k_misc.hip:
// the chunks of a BLAKE3 input are independent until the tree is folded).
template <class ChunkCv>
ACT_HD void b3_hash_xof64_with(uint32_t out[16], const uint32_t* msg, uint32_t len, ChunkCv chunk_cv) {
  uint32_t nchunks = len ? (len + 1023u) >> 10 : 1u;
  uint32_t stack[8][8];      // one entry per level: enough for 2^8 chunks = messages up to 256 KiB (ours are <= 16 KiB)
  int sp = 0;
  uint32_t cv[8], m[16], o[16];
  // every chunk but the last is finished into a chaining value and merged into the stack
  for (uint32_t c = 0; c + 1 < nchunks; c++) {
    chunk_cv(c, cv);
    uint32_t t = c + 1;
    while ((t & 1u) == 0u) {       // completed subtree pairs up with the one on the stack
      sp--;
      for (int i = 0; i < 8; i++) { m[i] = stack[sp][i]; m[8 + i] = cv[i]; }
      uint32_t iv[8]; for (int i = 0; i < 8; i++) iv[i] = b3_iv(i);
      b3_compress(o, iv, m, 0u, 0u, 64u, B3_PARENT);
      for (int i = 0; i < 8; i++) cv[i] = o[i];
      t >>= 1;
    }
    for (int i = 0; i < 8; i++) stack[sp][i] = cv[i];
    sp++;
  }
  // last chunk: keep its final block un-finalised (it may be the root)
  uint32_t c = nchunks - 1, base = c * 1024u;
  uint32_t clen = len - base;
  uint32_t nblocks = clen ? (clen + 63u) >> 6 : 1u;
  for (int i = 0; i < 8; i++) cv[i] = b3_iv(i);
  uint32_t blen = 0, fl = 0;
  for (uint32_t b = 0; b < nblocks; b++) {
    blen = b3_load_block(m, msg, len, base + b * 64u);
    fl = (b == 0 ? B3_CHUNK_START : 0u) | (b == nblocks - 1 ? B3_CHUNK_END : 0u);
    if (b + 1 < nblocks) { b3_compress(o, cv, m, c, 0u, 64u, fl); for (int i = 0; i < 8; i++) cv[i] = o[i]; }
  }
  uint32_t ctr = c;
  // fold the stack: right child = current node's chaining value
  while (sp > 0) {
    b3_compress(o, cv, m, ctr, 0u, blen, fl);
    sp--;
    for (int i = 0; i < 8; i++) { m[i] = stack[sp][i]; m[8 + i] = o[i]; }
    for (int i = 0; i < 8; i++) cv[i] = b3_iv(i);
    ctr = 0; blen = 64u; fl = B3_PARENT;
  }
  b3_compress(out, cv, m, 0u, 0u, blen, fl | B3_ROOT);   // root: counter = output block index 0
}
ACT_HD void b3_hash_xof64(uint32_t out[16], const uint32_t* msg, uint32_t len) {
  b3_hash_xof64_with(out, msg, len, [&](uint32_t c, uint32_t* cv) { b3_chunk_cv(cv, msg, len, c); });
}

}  // namespace act
