// node_mock.cpp — TEST-ONLY stand-ins for the single-GPU entry points that csrc/node.cpp calls, so that the node dispatcher
// (shard cutting, per-shard pointer arithmetic, the check -> count -> sign protocol that keeps ACT_RNG_SEQUENTIAL exact
// across shards, host-side routing of the node-level nullifier set) can be exercised by `pytest -m "not gpu"` without a
// device.  Linked with the real node.cpp into tests/node_mock/libnode_mock.so; never part of the product.
//
// Mock semantics, chosen so that every slicing mistake is visible in the output:
//   a lane is ACCEPTED iff the first byte of its input record is even; status = 7 (1 for issue) otherwise
//   outputs of an accepted lane = [8-byte global tag of the input record] | the first bytes of the rng slice it was given
//   ACT_RNG_PER_LANE: lane i of a call uses rng + 128 i;  ACT_RNG_SEQUENTIAL: accepted lanes use consecutive slices
//   every context counts its calls and lanes (act_mock_lanes) so that the test can see all of them were used
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include "../../include/act_mi355x.h"

struct act_ctx { int device; int L; std::atomic<size_t> lanes{0}; std::string err; std::atomic<unsigned> ns_per_lane{0}; };     // (lanes: calls that bypass the node's lock run on several threads)
struct act_nullifier_set { std::set<std::vector<uint8_t>> keys; std::string err; int device = 0; };
// failure injection (error-path tests): the device whose nullifier set / whose signature step fails, -1 = none
static int g_fail_null_device = -1, g_fail_sign_device = -1;
static const size_t kPB = 64;        // mock "spend proof" record: 64 bytes
static std::mutex g_mu;

extern "C" {
int act_ctx_create(const uint8_t*, int L, int device, size_t, act_ctx** out) { *out = new act_ctx{device, L}; return device < 0 ? ACT_ERR_ARG : ACT_OK; }
void act_ctx_destroy(act_ctx* c) { delete c; }
const char* act_last_error(const act_ctx* c) { return c ? c->err.c_str() : ""; }
int act_ctx_set_transcript_mode(act_ctx*, int) { return ACT_OK; }
int act_ctx_set_host_threads(act_ctx*, int) { return ACT_OK; }
size_t act_spend_proof_bytes(const act_ctx*) { return kPB; }
size_t act_prove_rng_bytes(const act_ctx*) { return 256; }
size_t act_mock_lanes(const act_ctx* c) { return c->lanes; }
void act_mock_fail(int null_device, int sign_device) { g_fail_null_device = null_device; g_fail_sign_device = sign_device; }
// a context as a slower GPU: every verification call sleeps ns_per_lane per lane (load-balance tests)
void act_mock_slow(act_ctx* c, unsigned ns_per_lane) { c->ns_per_lane = ns_per_lane; }
static void mock_work(act_ctx* c, size_t n) { c->lanes += n; const unsigned ns = c->ns_per_lane; if (ns) std::this_thread::sleep_for(std::chrono::nanoseconds((uint64_t)ns * n)); }

static void emit(uint8_t* out, size_t rec, const uint8_t* in, const uint8_t* rng) { memset(out, 0, rec); memcpy(out, in, 8); if (rng) memcpy(out + 8, rng, rec - 8 < 120 ? rec - 8 : 120); }

int act_request_batch(act_ctx* c, size_t n, int, const uint8_t* pre, const uint8_t* rng, uint8_t* out) {
  c->lanes += n; for (size_t i = 0; i < n; i++) emit(out + 128 * i, 128, pre + 64 * i, rng + 128 * i); return ACT_OK;
}
static int sign_like(act_ctx* c, size_t n, const uint8_t* in, size_t in_rec, const uint8_t* status_in, uint8_t bad, const uint8_t* rng, int mode,
                     uint8_t* out, size_t out_rec, uint8_t* status) {
  c->lanes += n; size_t cur = 0;
  for (size_t i = 0; i < n; i++) {
    uint8_t st = status_in ? status_in[i] : ((in[in_rec * i] & 1) ? bad : 0);
    if (status) status[i] = st;
    if (st) { memset(out + out_rec * i, 0, out_rec); continue; }
    const uint8_t* slice = rng + 128 * (mode == ACT_RNG_PER_LANE ? i : cur++);
    emit(out + out_rec * i, out_rec, in + in_rec * i, slice);
  }
  return ACT_OK;
}
int act_issue_batch(act_ctx* c, size_t n, int, const uint8_t*, const uint8_t* req, const uint8_t*, const uint8_t* rng, int mode, uint8_t* out, uint8_t* status) {
  return sign_like(c, n, req, 128, nullptr, 1, rng, mode, out, 160, status);
}
int act_issue_check_batch(act_ctx* c, size_t n, int, const uint8_t* req, uint8_t* status) { c->lanes += n; for (size_t i = 0; i < n; i++) status[i] = (req[128 * i] & 1) ? 1 : 0; return ACT_OK; }
int act_issue_sign_batch(act_ctx* c, size_t n, int, const uint8_t*, const uint8_t* req, const uint8_t*, const uint8_t* status_in, const uint8_t* rng, int mode,
                         uint8_t* out, uint8_t* status) { return sign_like(c, n, req, 128, status_in, 1, rng, mode, out, 160, status); }
int act_issuance_to_credit_token_batch(act_ctx* c, size_t n, int, const uint8_t* pre, const uint8_t*, const uint8_t*, const uint8_t* resp, uint8_t* out, uint8_t* status) {
  c->lanes += n; for (size_t i = 0; i < n; i++) { emit(out + 160 * i, 160, pre + 64 * i, resp + 160 * i); status[i] = 0; } return ACT_OK;
}
int act_prove_spend_batch(act_ctx* c, size_t n, int, const uint8_t* tok, const uint8_t* s, const uint8_t* rng, uint8_t* proof, uint8_t* prer, uint8_t* status) {
  c->lanes += n;
  for (size_t i = 0; i < n; i++) { emit(proof + kPB * i, kPB, tok + 160 * i, rng + 256 * i); emit(prer + 96 * i, 96, s + 32 * i, rng + 256 * i + 128); status[i] = 0; }
  return ACT_OK;
}
// seeded prover: the record's tag, then seed[0..8) and the GLOBAL lane number the shard was handed
int act_prove_spend_seeded_batch(act_ctx* c, size_t n, int, const uint8_t* tok, const uint8_t* s, const uint8_t* seed, uint64_t first_lane, uint8_t* proof, uint8_t* prer, uint8_t* status) {
  c->lanes += n;
  for (size_t i = 0; i < n; i++) {
    uint8_t r[128] = {0}; memcpy(r, seed, 8); const uint64_t lane = first_lane + i; memcpy(r + 8, &lane, 8);
    emit(proof + kPB * i, kPB, tok + 160 * i, r); emit(prer + 96 * i, 96, s + 32 * i, r); status[i] = 0;
  }
  return ACT_OK;
}
int act_verify_spend_batch(act_ctx* c, size_t n, int, const uint8_t*, const uint8_t* proof, uint8_t* status, uint8_t* kp) {
  mock_work(c, n);
  for (size_t i = 0; i < n; i++) { status[i] = (proof[kPB * i] & 1) ? 7 : 0; if (kp) { memset(kp + 32 * i, 0, 32); if (!status[i]) memcpy(kp + 32 * i, proof + kPB * i, 8); } }
  return ACT_OK;
}
// mock wire messages: a SpendProof is 3 framing bytes + the record; a Refund is 0xa4 + the 128-byte mock record
size_t act_cbor_size(const act_ctx*, int type) { return type == ACT_CBOR_REFUND ? 129 : kPB + 3; }
int act_verify_spend_cbor_keys_batch(act_ctx* c, size_t n, int, const uint8_t*, const uint8_t* cbor, const uint64_t* offsets, uint8_t* status, uint8_t* kp, uint8_t* nul) {
  mock_work(c, n);
  for (size_t i = 0; i < n; i++) {
    const uint8_t* m = cbor + (offsets ? offsets[i] : i * (kPB + 3)) + 3;
    status[i] = (m[0] & 1) ? 7 : 0;
    if (kp) { memset(kp + 32 * i, 0, 32); if (!status[i]) memcpy(kp + 32 * i, m, 8); }
    if (nul) memcpy(nul + 32 * i, m, 32);                 // the record's first 32 bytes are its nullifier
  }
  return ACT_OK;
}
int act_verify_spend_cbor_batch(act_ctx* c, size_t n, int mem, const uint8_t* sk, const uint8_t* cbor, const uint64_t* offsets, uint8_t* status, uint8_t* kp) {
  return act_verify_spend_cbor_keys_batch(c, n, mem, sk, cbor, offsets, status, kp, nullptr);
}
int act_refund_batch(act_ctx* c, size_t n, int, const uint8_t*, const uint8_t* proof, const uint8_t* rng, int mode, uint8_t* out, uint8_t* status) {
  return sign_like(c, n, proof, kPB, nullptr, 7, rng, mode, out, 128, status);
}
int act_refund_sign_batch(act_ctx* c, size_t n, int, const uint8_t*, const uint8_t* kprime, const uint8_t* status_in, const uint8_t* rng, int mode, uint8_t* out, uint8_t* status) {
  if (c->device == g_fail_sign_device) { c->err = "mock: signature step failed"; return ACT_ERR_HIP; }
  return sign_like(c, n, kprime, 32, status_in, 7, rng, mode, out, 128, status);      // K' carries the record's 8-byte tag
}
int act_refund_sign_cbor_batch(act_ctx* c, size_t n, int mem, const uint8_t* sk, const uint8_t* kprime, const uint8_t* status_in, const uint8_t* rng, int mode, uint8_t* out, uint8_t* status) {
  std::vector<uint8_t> rec(128 * n + 1);
  const int rc = act_refund_sign_batch(c, n, mem, sk, kprime, status_in, rng, mode, rec.data(), status);
  if (rc) return rc;
  for (size_t i = 0; i < n; i++) { memset(out + 129 * i, 0, 129); if (!status[i]) { out[129 * i] = 0xa4; memcpy(out + 129 * i + 1, rec.data() + 128 * i, 128); } }
  return ACT_OK;
}
int act_refund_cbor_keys_batch(act_ctx* c, size_t n, int mem, const uint8_t* sk, const uint8_t* cbor, const uint64_t* offsets, const uint8_t* rng, int mode, uint8_t* out, uint8_t* status, uint8_t* nul) {
  std::vector<uint8_t> kp(32 * n + 1), st(n + 1);
  int rc = act_verify_spend_cbor_keys_batch(c, n, mem, sk, cbor, offsets, st.data(), kp.data(), nul);
  return rc ? rc : act_refund_sign_cbor_batch(c, n, mem, sk, kp.data(), st.data(), rng, mode, out, status);
}
int act_refund_cbor_batch(act_ctx* c, size_t n, int mem, const uint8_t* sk, const uint8_t* cbor, const uint64_t* offsets, const uint8_t* rng, int mode, uint8_t* out, uint8_t* status) {
  return act_refund_cbor_keys_batch(c, n, mem, sk, cbor, offsets, rng, mode, out, status, nullptr);
}
int act_refund_to_credit_token_batch(act_ctx* c, size_t n, int, const uint8_t* prer, const uint8_t* proof, const uint8_t* refund, const uint8_t*, uint8_t* out, uint8_t* status) {
  c->lanes += n; for (size_t i = 0; i < n; i++) { emit(out + 160 * i, 160, prer + 96 * i, refund + 128 * i); out[159 + 160 * i] = proof[kPB * i]; status[i] = 0; } return ACT_OK;
}

#ifndef ACT_MOCK_NO_PARALLEL_FOR      // (the ThreadSanitizer driver links the real csrc/host_pool.cpp instead)
// the host pool's parallel-for (csrc/host_pool.cpp), here on two threads that take the items from the far end: the dispatcher must
// not depend on item order
void act_host_parallel_for(size_t n, size_t grain, int, void (*fn)(void*, size_t, size_t), void* ctx) {
  if (!n) return;
  if (!grain) grain = 1;
  const size_t items = (n + grain - 1) / grain;
  auto run = [&](size_t parity) { for (size_t k = items; k-- > 0;) if ((k & 1) == parity) fn(ctx, k * grain, std::min(n, (k + 1) * grain)); };
  std::thread t(run, (size_t)1);
  run(0);
  t.join();
}
#endif

int act_ctx_set_fixed_base_bits(act_ctx*, int, int) { return ACT_OK; }
int act_nullifier_set_create(int device, size_t, const uint8_t*, act_nullifier_set** out) { *out = new act_nullifier_set(); (*out)->device = device; return ACT_OK; }
void act_nullifier_set_destroy(act_nullifier_set* s) { delete s; }
size_t act_nullifier_set_len(const act_nullifier_set* s) { return s->keys.size(); }
const char* act_nullifier_set_last_error(const act_nullifier_set* s) { return s->err.c_str(); }
int act_nullifier_check_and_insert_batch(act_nullifier_set* s, size_t n, int, const uint8_t* k, size_t stride, const uint8_t* mask, uint8_t* spent) {
  if (s->device == g_fail_null_device) { s->err = "mock: device lost"; return ACT_ERR_HIP; }
  for (size_t i = 0; i < n; i++) {                // like the real set: keys are scalars (reduced mod l) -- so an alias k + l is
    if (mask && mask[i]) { spent[i] = 0; continue; }   // reported spent only if the ROUTING sent it to the set that holds k
    static const uint64_t Lw[4] = {0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0, 0x1000000000000000ull};
    uint64_t v[4]; memcpy(v, k + i * stride, 32);
    for (;;) {
      uint64_t t[4]; unsigned __int128 b = 0;
      for (int j = 0; j < 4; j++) { unsigned __int128 d = (unsigned __int128)v[j] - Lw[j] - (uint64_t)b; t[j] = (uint64_t)d; b = (d >> 64) & 1; }
      if (b) break;
      memcpy(v, t, 32);
    }
    std::vector<uint8_t> key(32); memcpy(key.data(), v, 32);
    spent[i] = !s->keys.insert(key).second;
  }
  return ACT_OK;
}
}  // extern "C"
