// engine.hip — host side of libact_mi355x.so: contexts, workspace, chunked kernel orchestration,
// the host-transcript mode (BLAKE3 of every Fiat–Shamir transcript on host threads, as the
// reference's src/transcript.rs does) and the C ABI of include/act_mi355x.h.
// There is no CPU compute path here: the host only moves bytes and (optionally) hashes them.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "kernels.h"
#include "../../include/act_mi355x.h"

using namespace act;

namespace {

const char* const kLabels[4] = {"request", "respond", "spend", "refund"};
const char kProtocolVersion[] = "curve25519-ristretto anonymous-credits v1.0";   // src/transcript.rs:29
// RFC 9496 appendix A.1: encoding of the ristretto255 generator
const uint8_t kGeneratorEnc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                   0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};

enum ProfId { PK_SPEND_PREP, PK_SPEND_BITS, PK_SPEND_TAIL, PK_HASH_SPEND, PK_SPEND_FINISH, PK_SIGN_A, PK_HASH_SMALL, PK_SIGN_B,
              PK_ISSUE_A, PK_ISSUE_CHECK, PK_REQUEST_A, PK_REQUEST_B, PK_PROVE_HEAD, PK_PROVE_BITS, PK_PROVE_TAIL, PK_PROVE_RESP,
              PK_CLIENT, PK_COUNT };
const char* const kProfNames[PK_COUNT] = {"k_spend_prep", "k_spend_bits", "k_spend_tail", "k_hash_xof(spend)", "k_spend_finish",
                                          "k_sign_a", "k_hash_xof(small)", "k_sign_b", "k_issue_a", "k_issue_check", "k_request_a",
                                          "k_request_b", "k_prove_head", "k_prove_bits", "k_prove_tail", "k_prove_resp", "k_client_verify"};

struct PendingProf { int id; hipEvent_t e0, e1; uint64_t lanes; };

}  // namespace

struct act_ctx {
  int device = 0, L = 128;
  size_t max_batch = 0;
  hipStream_t stream = nullptr;
  DevParams P{};
  uint8_t henc[96]{};
  int tr_mode = ACT_TRANSCRIPT_HOST;
  int host_threads = 0;
  std::string err;
  // device workspace (sized for max_batch lanes)
  uint8_t *d_tr = nullptr, *d_trs = nullptr, *d_status = nullptr;
  uint32_t *d_buckets = nullptr, *d_coords = nullptr, *d_d01 = nullptr, *d_xa = nullptr, *d_flags = nullptr, *d_xof = nullptr, *d_state = nullptr, *d_slot = nullptr;
  uint32_t* d_tables = nullptr;
  // staging for host-memory callers: grow-only device buffers
  uint8_t* d_stage[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t d_stage_cap[6] = {0, 0, 0, 0, 0, 0};
  // pinned host buffers for the host-transcript mode
  uint8_t* h_tr = nullptr; size_t h_tr_cap = 0;
  uint32_t* h_xof = nullptr; size_t h_xof_cap = 0;
  // key cache
  uint8_t sk_cached[64]{}; bool sk_valid = false; DevKey key{};
  uint8_t w_cached[32]{}; bool w_valid = false; ge w_pub{};
  // profiling
  bool prof_on = false;
  double prof_ms[PK_COUNT]{}; uint64_t prof_launches[PK_COUNT]{}; uint64_t prof_lanes[PK_COUNT]{};
  std::vector<PendingProf> pending;
  size_t last_spend_lanes = 0;
};

namespace {

#define HIPCK(ctx, expr)                                                                            \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                               \
      return ACT_ERR_HIP;                                                                           \
    }                                                                                               \
  } while (0)

template <class F>
int prof_launch(act_ctx* c, int id, uint64_t lanes, F&& f) {
  if (!c->prof_on) { f(); return ACT_OK; }
  PendingProf p{id, nullptr, nullptr, lanes};
  HIPCK(c, hipEventCreate(&p.e0)); HIPCK(c, hipEventCreate(&p.e1));
  HIPCK(c, hipEventRecord(p.e0, c->stream));
  f();
  HIPCK(c, hipEventRecord(p.e1, c->stream));
  c->pending.push_back(p);
  return ACT_OK;
}
int prof_collect(act_ctx* c) {
  for (auto& p : c->pending) {
    HIPCK(c, hipEventSynchronize(p.e1));
    float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, p.e0, p.e1));
    c->prof_ms[p.id] += ms; c->prof_launches[p.id]++; c->prof_lanes[p.id] += p.lanes;
    hipEventDestroy(p.e0); hipEventDestroy(p.e1);
  }
  c->pending.clear();
  return ACT_OK;
}

int stage_reserve(act_ctx* c, int slot, size_t bytes) {
  if (bytes <= c->d_stage_cap[slot]) return ACT_OK;
  if (c->d_stage[slot]) HIPCK(c, hipFree(c->d_stage[slot]));
  c->d_stage[slot] = nullptr; c->d_stage_cap[slot] = 0;
  HIPCK(c, hipMalloc(&c->d_stage[slot], bytes));
  c->d_stage_cap[slot] = bytes;
  return ACT_OK;
}
// device view of `bytes` of caller memory: the pointer itself (device memory) or a staged H2D copy
int dev_in(act_ctx* c, int slot, int mem, const uint8_t* p, size_t bytes, const uint8_t** out) {
  if (mem == ACT_MEM_DEVICE || bytes == 0) { *out = p; return ACT_OK; }
  int rc = stage_reserve(c, slot, bytes); if (rc) return rc;
  HIPCK(c, hipMemcpyAsync(c->d_stage[slot], p, bytes, hipMemcpyHostToDevice, c->stream));
  *out = c->d_stage[slot];
  return ACT_OK;
}
int dev_out_begin(act_ctx* c, int slot, int mem, uint8_t* p, size_t bytes, uint8_t** out) {
  if (mem == ACT_MEM_DEVICE || bytes == 0) { *out = p; return ACT_OK; }
  int rc = stage_reserve(c, slot, bytes); if (rc) return rc;
  *out = c->d_stage[slot];
  return ACT_OK;
}
int dev_out_end(act_ctx* c, int mem, uint8_t* host_p, const uint8_t* dev_p, size_t bytes) {
  if (mem == ACT_MEM_DEVICE || bytes == 0) return ACT_OK;
  HIPCK(c, hipMemcpyAsync(host_p, dev_p, bytes, hipMemcpyDeviceToHost, c->stream));
  return ACT_OK;
}

void put_be64(std::vector<uint8_t>& v, uint64_t x) { for (int i = 7; i >= 0; i--) v.push_back((uint8_t)(x >> (8 * i))); }
void put_lp(std::vector<uint8_t>& v, const uint8_t* b, size_t n) { put_be64(v, n); v.insert(v.end(), b, b + n); }

// host BLAKE3 (blake3_hd.h compiled for the host) of an arbitrary byte string -> 64 XOF bytes
void host_xof64(const std::vector<uint8_t>& msg, uint8_t out[64]) {
  std::vector<uint32_t> w((msg.size() + 3) / 4 + 1, 0u);
  if (!msg.empty()) memcpy(w.data(), msg.data(), msg.size());
  uint32_t o[16]; b3_hash_xof64(o, w.data(), (uint32_t)msg.size());
  memcpy(out, o, 64);
}

int n_host_threads(const act_ctx* c) {
  int t = c->host_threads > 0 ? c->host_threads : (int)std::thread::hardware_concurrency();
  return t < 1 ? 1 : t;
}
// hash n messages of `len` bytes at `stride` (host memory) into xof[n][16] on host threads
void host_hash_many(const act_ctx* c, const uint8_t* msgs, size_t stride, uint32_t len, size_t n, uint32_t* xof) {
  int nt = (int)std::min<size_t>((size_t)n_host_threads(c), n ? n : 1);
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      size_t i0 = next.fetch_add(64);
      if (i0 >= n) break;
      size_t i1 = std::min(n, i0 + 64);
      for (size_t i = i0; i < i1; i++) b3_hash_xof64(xof + i * 16, reinterpret_cast<const uint32_t*>(msgs + i * stride), len);
    }
  };
  if (nt <= 1) { work(); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++) th.emplace_back(work);
  for (auto& t : th) t.join();
}

// transcript hashing step: device kernel, or D2H -> host threads -> H2D
int hash_step(act_ctx* c, int prof_id, const uint8_t* d_msgs, uint32_t stride, uint32_t len, uint32_t n) {
  if (c->tr_mode == ACT_TRANSCRIPT_DEVICE) {
    HashArgs h{d_msgs, stride, len, n, c->d_xof, nullptr};
    return prof_launch(c, prof_id, n, [&] { launch_hash(h, c->stream); });
  }
  size_t bytes = (size_t)n * stride;
  if (bytes > c->h_tr_cap) {
    if (c->h_tr) HIPCK(c, hipHostFree(c->h_tr));
    c->h_tr = nullptr; c->h_tr_cap = 0;
    HIPCK(c, hipHostMalloc(&c->h_tr, bytes, hipHostMallocDefault)); c->h_tr_cap = bytes;
  }
  if ((size_t)n * 64 > c->h_xof_cap) {
    if (c->h_xof) HIPCK(c, hipHostFree(c->h_xof));
    c->h_xof = nullptr; c->h_xof_cap = 0;
    HIPCK(c, hipHostMalloc(&c->h_xof, (size_t)n * 64, hipHostMallocDefault)); c->h_xof_cap = (size_t)n * 64;
  }
  HIPCK(c, hipMemcpyAsync(c->h_tr, d_msgs, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  host_hash_many(c, c->h_tr, stride, len, n, c->h_xof);
  HIPCK(c, hipMemcpyAsync(c->d_xof, c->h_xof, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
  return ACT_OK;
}

int set_key(act_ctx* c, const uint8_t sk[64]) {
  if (c->sk_valid && memcmp(c->sk_cached, sk, 64) == 0) return ACT_OK;
  uint32_t w[8]; memcpy(w, sk, 32);
  c->key.x = sc_from_words(w);
  int rc = stage_reserve(c, 5, 32 + GE_WORDS * 4 + 16); if (rc) return rc;
  uint8_t* d = c->d_stage[5];
  HIPCK(c, hipMemcpyAsync(d, sk + 32, 32, hipMemcpyHostToDevice, c->stream));
  launch_decode_points(d, 1, reinterpret_cast<uint32_t*>(d + 32), reinterpret_cast<uint32_t*>(d + 32 + GE_WORDS * 4), c->stream);
  uint32_t host[GE_WORDS + 1];
  HIPCK(c, hipMemcpyAsync(host, d + 32, sizeof(host), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (!host[GE_WORDS]) { c->err = "public key w is not a canonical Ristretto encoding"; return ACT_ERR_PARAMS; }
  c->key.w = ge_load(host);
  memcpy(c->sk_cached, sk, 64); c->sk_valid = true;
  return ACT_OK;
}

int workspace_alloc(act_ctx* c) {
  const SpendTranscript st{c->L};
  size_t B = c->max_batch;
  HIPCK(c, hipMalloc(&c->d_tr, B * st.stride()));
  HIPCK(c, hipMalloc(&c->d_coords, B * (size_t)c->L * NIELS_WORDS * 4));
  HIPCK(c, hipMalloc(&c->d_d01, B * 3 * GE_WORDS * 4));
  HIPCK(c, hipMalloc(&c->d_buckets, B * (size_t)c->L * BUCKET_WORDS * 4));
  HIPCK(c, hipMalloc(&c->d_xa, B * GE_WORDS * 4));
  HIPCK(c, hipMalloc(&c->d_flags, B * 4));
  HIPCK(c, hipMalloc(&c->d_xof, B * 64));
  HIPCK(c, hipMalloc(&c->d_status, B));
  HIPCK(c, hipMalloc(&c->d_trs, B * SMALL_TR_STRIDE));
  HIPCK(c, hipMalloc(&c->d_state, B * 24 * 4));
  HIPCK(c, hipMalloc(&c->d_slot, B * 4));
  HIPCK(c, hipMemsetAsync(c->d_trs, 0, B * SMALL_TR_STRIDE, c->stream));
  return ACT_OK;
}

int from_uniform_on_device(int device, const uint8_t* in64, int n, uint8_t* out_enc, std::string* err) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { if (err) *err = "no HIP device"; return ACT_ERR_NO_DEVICE; }
  if (hipSetDevice(device) != hipSuccess) return ACT_ERR_HIP;
  uint8_t* d = nullptr;
  if (hipMalloc(&d, (size_t)n * 96) != hipSuccess) return ACT_ERR_HIP;
  int rc = ACT_OK;
  if (hipMemcpy(d, in64, (size_t)n * 64, hipMemcpyHostToDevice) != hipSuccess) rc = ACT_ERR_HIP;
  if (!rc) { launch_from_uniform(d, (uint32_t)n, d + (size_t)n * 64, nullptr); if (hipDeviceSynchronize() != hipSuccess) rc = ACT_ERR_HIP; }
  if (!rc && hipMemcpy(out_enc, d + (size_t)n * 64, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess) rc = ACT_ERR_HIP;
  hipFree(d);
  return rc;
}

}  // namespace

extern "C" {

int act_params_new(int device, const char* org, const char* svc, const char* dep, const char* ver, uint8_t out_h[96]) {
  if (!org || !svc || !dep || !ver || !out_h) return ACT_ERR_ARG;
  // src/lib.rs:293-303: seed = BLAKE3(u64_be(len) | "ACT-v1:org:svc:dep:ver")
  std::string ds = std::string("ACT-v1:") + org + ":" + svc + ":" + dep + ":" + ver;
  std::vector<uint8_t> m; put_lp(m, (const uint8_t*)ds.data(), ds.size());
  uint8_t seed[64]; host_xof64(m, seed);
  uint8_t uni[192];
  for (uint32_t ctr = 0; ctr < 3; ctr++) {                      // src/lib.rs:332-351
    std::vector<uint8_t> msg; put_lp(msg, (const uint8_t*)ds.data(), ds.size()); put_lp(msg, seed, 32);
    uint8_t cb[4] = {(uint8_t)ctr, 0, 0, 0}; put_lp(msg, cb, 4);
    host_xof64(msg, uni + 64 * ctr);
  }
  return from_uniform_on_device(device, uni, 3, out_h, nullptr);  // src/lib.rs:353
}
int act_params_random(int device, const uint8_t rng[192], uint8_t out_h[96]) {
  if (!rng || !out_h) return ACT_ERR_ARG;
  return from_uniform_on_device(device, rng, 3, out_h, nullptr);
}

int act_ctx_create(const uint8_t h[96], int L, int device, size_t max_batch, act_ctx** out) {
  if (!h || !out || L < 1 || L > 128) return ACT_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ACT_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return ACT_ERR_ARG;
  act_ctx* c = new act_ctx();
  *out = c;
  c->device = device; c->L = L; c->max_batch = max_batch ? max_batch : 16384;
  memcpy(c->henc, h, 96);
  HIPCK(c, hipSetDevice(device));
  HIPCK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  // decode g, h1, h2, h3 and build their fixed-base tables
  uint8_t enc[128]; memcpy(enc, kGeneratorEnc, 32); memcpy(enc + 32, h, 96);
  uint8_t* d_enc = nullptr; uint32_t *d_ext = nullptr, *d_ok = nullptr;
  HIPCK(c, hipMalloc(&d_enc, 128)); HIPCK(c, hipMalloc(&d_ext, 4 * GE_WORDS * 4)); HIPCK(c, hipMalloc(&d_ok, 16));
  HIPCK(c, hipMemcpyAsync(d_enc, enc, 128, hipMemcpyHostToDevice, c->stream));
  launch_decode_points(d_enc, 4, d_ext, d_ok, c->stream);
  uint32_t ok[4];
  HIPCK(c, hipMemcpyAsync(ok, d_ok, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (!(ok[0] && ok[1] && ok[2] && ok[3])) { c->err = "h1/h2/h3 is not a canonical Ristretto encoding"; return ACT_ERR_PARAMS; }
  HIPCK(c, hipMalloc(&c->d_tables, (size_t)4 * FB_TABLE_WORDS * 4));
  for (int b = 0; b < 4; b++) {
    launch_build_table(d_ext + b * GE_WORDS, c->d_tables + (size_t)b * FB_TABLE_WORDS, c->stream);
    c->P.tab[b] = c->d_tables + (size_t)b * FB_TABLE_WORDS;
  }
  HIPCK(c, hipStreamSynchronize(c->stream));
  HIPCK(c, hipFree(d_enc)); HIPCK(c, hipFree(d_ext)); HIPCK(c, hipFree(d_ok));
  // Transcript::new(params, label) prefixes, src/transcript.rs:54-74
  for (int l = 0; l < 4; l++) {
    std::vector<uint8_t> p;
    put_lp(p, (const uint8_t*)kProtocolVersion, sizeof(kProtocolVersion) - 1);
    put_lp(p, h, 32); put_lp(p, h + 32, 32); put_lp(p, h + 64, 32);
    put_lp(p, (const uint8_t*)kLabels[l], strlen(kLabels[l]));
    c->P.prefix_len[l] = (uint32_t)p.size();
    p.resize(PREFIX_WORDS * 4, 0);
    memcpy(c->P.prefix[l], p.data(), PREFIX_WORDS * 4);
  }
  c->P.L = L;
  return workspace_alloc(c);
}

void act_ctx_destroy(act_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->d_buckets, c->d_tr, c->d_trs, c->d_status, c->d_coords, c->d_d01, c->d_xa, c->d_flags, c->d_xof, c->d_state, c->d_slot, c->d_tables};
  for (void* p : ptrs) if (p) hipFree(p);
  for (int i = 0; i < 6; i++) if (c->d_stage[i]) { hipMemset(c->d_stage[i], 0, c->d_stage_cap[i]); hipFree(c->d_stage[i]); }   // staging may hold secrets
  if (c->h_tr) hipHostFree(c->h_tr);
  if (c->h_xof) hipHostFree(c->h_xof);
  if (c->stream) hipStreamDestroy(c->stream);
  memset(&c->key, 0, sizeof(c->key)); memset(c->sk_cached, 0, 64);
  delete c;
}
int act_ctx_set_transcript_mode(act_ctx* c, int mode) {
  if (!c || (mode != ACT_TRANSCRIPT_HOST && mode != ACT_TRANSCRIPT_DEVICE)) return ACT_ERR_ARG;
  c->tr_mode = mode; return ACT_OK;
}
int act_ctx_set_host_threads(act_ctx* c, int n) { if (!c || n < 0) return ACT_ERR_ARG; c->host_threads = n; return ACT_OK; }
const char* act_last_error(const act_ctx* c) { return c ? c->err.c_str() : "null context"; }
size_t act_spend_proof_bytes(const act_ctx* c) { return ProofLayout{c->L}.bytes(); }
size_t act_prove_rng_bytes(const act_ctx* c) { return 64u * (4u * (size_t)c->L + 12u); }
size_t act_spend_transcript_bytes(const act_ctx* c) { return SpendTranscript{c->L}.bytes(); }

int act_private_key_random(act_ctx* c, const uint8_t rng[64], uint8_t out_sk[64]) {
  if (!c || !rng || !out_sk) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  int rc = stage_reserve(c, 0, 128); if (rc) return rc;
  HIPCK(c, hipMemcpyAsync(c->d_stage[0], rng, 64, hipMemcpyHostToDevice, c->stream));
  launch_keygen(c->P, c->d_stage[0], 1, c->d_stage[0] + 64, c->stream);
  HIPCK(c, hipMemcpyAsync(out_sk, c->d_stage[0] + 64, 64, hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  return ACT_OK;
}
int act_pre_issuance_random_batch(act_ctx* c, size_t n, int mem, const uint8_t* rng, uint8_t* out_pre) {
  if (!c || (n && (!rng || !out_pre))) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    const uint8_t* d_rng; uint8_t* d_out; int rc;
    if ((rc = dev_in(c, 0, mem, rng + off * 128, (size_t)m * 128, &d_rng))) return rc;
    if ((rc = dev_out_begin(c, 1, mem, out_pre + off * 64, (size_t)m * 64, &d_out))) return rc;
    launch_pre_issuance_random(d_rng, m, d_out, c->stream);
    if ((rc = dev_out_end(c, mem, out_pre + off * 64, d_out, (size_t)m * 64))) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
  }
  return ACT_OK;
}

int act_request_batch(act_ctx* c, size_t n, int mem, const uint8_t* pre, const uint8_t* rng, uint8_t* out_req) {
  if (!c || (n && (!pre || !rng || !out_req))) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    RequestArgs a{}; a.P = c->P; a.n = m; a.trs = c->d_trs; a.xof = c->d_xof; int rc;
    if ((rc = dev_in(c, 0, mem, pre + off * 64, (size_t)m * 64, &a.pre))) return rc;
    if ((rc = dev_in(c, 1, mem, rng + off * 128, (size_t)m * 128, &a.rng))) return rc;
    if ((rc = dev_out_begin(c, 2, mem, out_req + off * 128, (size_t)m * 128, &a.out))) return rc;
    if ((rc = prof_launch(c, PK_REQUEST_A, m, [&] { launch_request_a(a, c->stream); }))) return rc;
    if ((rc = hash_step(c, PK_HASH_SMALL, c->d_trs, SMALL_TR_STRIDE, c->P.prefix_len[LABEL_REQUEST] + 80, m))) return rc;
    if ((rc = prof_launch(c, PK_REQUEST_B, m, [&] { launch_request_b(a, c->stream); }))) return rc;
    if ((rc = dev_out_end(c, mem, out_req + off * 128, a.out, (size_t)m * 128))) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    if ((rc = prof_collect(c))) return rc;
  }
  return ACT_OK;
}

// rng slots for the signing phase of issue / refund.  Returns the device rng base through *d_rng.
static int prepare_rng_slots(act_ctx* c, uint32_t m, size_t off, int mem, const uint8_t* rng, int rng_mode, size_t* seq_cursor,
                             const uint8_t** d_rng) {
  std::vector<uint32_t> slot(m);
  if (rng_mode == ACT_RNG_PER_LANE) {
    if (mem == ACT_MEM_DEVICE) { for (uint32_t i = 0; i < m; i++) slot[i] = (uint32_t)(off + i); *d_rng = rng; }
    else { for (uint32_t i = 0; i < m; i++) slot[i] = i; int rc = dev_in(c, 3, mem, rng + off * 128, (size_t)m * 128, d_rng); if (rc) return rc; }
  } else {
    std::vector<uint8_t> st(m);
    HIPCK(c, hipMemcpyAsync(st.data(), c->d_status, m, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    size_t cur = *seq_cursor, base = cur;
    for (uint32_t i = 0; i < m; i++) { slot[i] = (uint32_t)(mem == ACT_MEM_DEVICE ? cur : cur - base); if (st[i] == 0) cur++; }
    *seq_cursor = cur;
    if (mem == ACT_MEM_DEVICE) *d_rng = rng;
    else { int rc = dev_in(c, 3, mem, rng + base * 128, (cur - base) * 128, d_rng); if (rc) return rc; if (cur == base) *d_rng = c->d_stage[3]; }
  }
  HIPCK(c, hipMemcpyAsync(c->d_slot, slot.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));   // `slot` is a stack-lifetime host buffer
  return ACT_OK;
}

static int sign_phase(act_ctx* c, uint32_t m, int label, const uint8_t* d_rng, const uint8_t* d_camount, uint8_t* d_out) {
  SignArgs s{}; s.P = c->P; s.K = c->key; s.n = m; s.label = label; s.xa = c->d_xa; s.status = c->d_status; s.rng_slot = c->d_slot;
  s.rng = d_rng; s.c_amount = d_camount; s.trs = c->d_trs; s.state = c->d_state; s.xof = c->d_xof; s.out = d_out;
  int rc;
  if ((rc = prof_launch(c, PK_SIGN_A, m, [&] { launch_sign_a(s, c->stream); }))) return rc;
  uint32_t len = c->P.prefix_len[label] + 40u * (label == LABEL_RESPOND ? 7u : 6u);
  if ((rc = hash_step(c, PK_HASH_SMALL, c->d_trs, SMALL_TR_STRIDE, len, m))) return rc;
  return prof_launch(c, PK_SIGN_B, m, [&] { launch_sign_b(s, c->stream); });
}

int act_issue_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* req, const uint8_t* camt, const uint8_t* rng,
                    int rng_mode, uint8_t* out_resp, uint8_t* status) {
  if (!c || !sk || (n && (!req || !camt || !rng || !out_resp || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  int rc = set_key(c, sk); if (rc) return rc;
  size_t cursor = 0;
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    IssueArgs a{}; a.P = c->P; a.n = m; a.trs = c->d_trs; a.xa = c->d_xa; a.flags = c->d_flags; a.xof = c->d_xof; a.status = c->d_status;
    if ((rc = dev_in(c, 0, mem, req + off * 128, (size_t)m * 128, &a.req))) return rc;
    if ((rc = dev_in(c, 1, mem, camt + off * 32, (size_t)m * 32, &a.c_amount))) return rc;
    uint8_t* d_out;
    if ((rc = dev_out_begin(c, 2, mem, out_resp + off * 160, (size_t)m * 160, &d_out))) return rc;
    if ((rc = prof_launch(c, PK_ISSUE_A, m, [&] { launch_issue_a(a, c->stream); }))) return rc;
    if ((rc = hash_step(c, PK_HASH_SMALL, c->d_trs, SMALL_TR_STRIDE, c->P.prefix_len[LABEL_REQUEST] + 80, m))) return rc;
    if ((rc = prof_launch(c, PK_ISSUE_CHECK, m, [&] { launch_issue_check(a, c->stream); }))) return rc;
    const uint8_t* d_rng;
    if ((rc = prepare_rng_slots(c, m, off, mem, rng, rng_mode, &cursor, &d_rng))) return rc;
    if ((rc = sign_phase(c, m, LABEL_RESPOND, d_rng, a.c_amount, d_out))) return rc;
    if ((rc = dev_out_end(c, mem, out_resp + off * 160, d_out, (size_t)m * 160))) return rc;
    HIPCK(c, hipMemcpyAsync(status + off, c->d_status, m, mem == ACT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if ((rc = prof_collect(c))) return rc;
  }
  return ACT_OK;
}

// spend verification of one chunk; leaves status in d_status and X_A in d_xa
static int verify_chunk(act_ctx* c, uint32_t m, const uint8_t* d_proofs, uint8_t* d_kprime) {
  const SpendTranscript st{c->L};
  SpendArgs a{}; a.P = c->P; a.K = c->key; a.proofs = d_proofs; a.n = m; a.tr = c->d_tr; a.tr_stride = (uint32_t)st.stride();
  a.coords = c->d_coords; a.d01 = c->d_d01; a.buckets = c->d_buckets; a.xa = c->d_xa; a.flags = c->d_flags; a.xof = c->d_xof; a.status = c->d_status; a.kprime_enc = d_kprime;
  int rc;
  if ((rc = prof_launch(c, PK_SPEND_PREP, m, [&] { launch_spend_prep(a, c->stream); }))) return rc;
  if ((rc = prof_launch(c, PK_SPEND_BITS, (uint64_t)m * c->L, [&] { launch_spend_bits(a, c->stream); }))) return rc;
  if ((rc = prof_launch(c, PK_SPEND_TAIL, m, [&] { launch_spend_tail(a, c->stream); }))) return rc;
  if ((rc = hash_step(c, PK_HASH_SPEND, c->d_tr, (uint32_t)st.stride(), (uint32_t)st.bytes(), m))) return rc;
  if ((rc = prof_launch(c, PK_SPEND_FINISH, m, [&] { launch_spend_finish(a, c->stream); }))) return rc;
  c->last_spend_lanes = m;
  return ACT_OK;
}

int act_verify_spend_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* proof, uint8_t* status, uint8_t* out_kprime) {
  if (!c || !sk || (n && (!proof || !status))) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  int rc = set_key(c, sk); if (rc) return rc;
  const size_t pb = ProofLayout{c->L}.bytes();
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    const uint8_t* d_proofs; uint8_t* d_kp = nullptr;
    if ((rc = dev_in(c, 0, mem, proof + off * pb, (size_t)m * pb, &d_proofs))) return rc;
    if (out_kprime && (rc = dev_out_begin(c, 2, mem, out_kprime + off * 32, (size_t)m * 32, &d_kp))) return rc;
    if ((rc = verify_chunk(c, m, d_proofs, d_kp))) return rc;
    if (out_kprime && (rc = dev_out_end(c, mem, out_kprime + off * 32, d_kp, (size_t)m * 32))) return rc;
    HIPCK(c, hipMemcpyAsync(status + off, c->d_status, m, mem == ACT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if ((rc = prof_collect(c))) return rc;
  }
  return ACT_OK;
}

int act_refund_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* proof, const uint8_t* rng, int rng_mode,
                     uint8_t* out_refund, uint8_t* status) {
  if (!c || !sk || (n && (!proof || !rng || !out_refund || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  int rc = set_key(c, sk); if (rc) return rc;
  const size_t pb = ProofLayout{c->L}.bytes();
  size_t cursor = 0;
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    const uint8_t *d_proofs, *d_rng; uint8_t* d_out;
    if ((rc = dev_in(c, 0, mem, proof + off * pb, (size_t)m * pb, &d_proofs))) return rc;
    if ((rc = dev_out_begin(c, 2, mem, out_refund + off * 128, (size_t)m * 128, &d_out))) return rc;
    if ((rc = verify_chunk(c, m, d_proofs, nullptr))) return rc;
    if ((rc = prepare_rng_slots(c, m, off, mem, rng, rng_mode, &cursor, &d_rng))) return rc;
    if ((rc = sign_phase(c, m, LABEL_REFUND, d_rng, nullptr, d_out))) return rc;
    if ((rc = dev_out_end(c, mem, out_refund + off * 128, d_out, (size_t)m * 128))) return rc;
    HIPCK(c, hipMemcpyAsync(status + off, c->d_status, m, mem == ACT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if ((rc = prof_collect(c))) return rc;
  }
  return ACT_OK;
}

int act_prove_spend_batch(act_ctx* c, size_t n, int mem, const uint8_t* token, const uint8_t* s, const uint8_t* rng,
                          uint8_t* out_proof, uint8_t* out_prerefund, uint8_t* status) {
  if (!c || (n && (!token || !s || !rng || !out_proof || !out_prerefund || !status))) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  const size_t pb = ProofLayout{c->L}.bytes(), rb = act_prove_rng_bytes(c);
  const SpendTranscript st{c->L};
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    ProveArgs a{}; a.P = c->P; a.n = m; a.tr = c->d_tr; a.tr_stride = (uint32_t)st.stride(); a.d3 = c->d_d01; a.state = c->d_state;
    a.flags = c->d_flags; a.xof = c->d_xof; a.status = c->d_status;
    int rc;
    if ((rc = dev_in(c, 0, mem, token + off * 160, (size_t)m * 160, &a.tok))) return rc;
    if ((rc = dev_in(c, 1, mem, s + off * 32, (size_t)m * 32, &a.s))) return rc;
    if ((rc = dev_in(c, 3, mem, rng + off * rb, (size_t)m * rb, &a.rng))) return rc;
    if ((rc = dev_out_begin(c, 2, mem, out_proof + off * pb, (size_t)m * pb, &a.proof))) return rc;
    if ((rc = dev_out_begin(c, 4, mem, out_prerefund + off * 96, (size_t)m * 96, &a.prerefund))) return rc;
    if ((rc = prof_launch(c, PK_PROVE_HEAD, m, [&] { launch_prove_head(a, c->stream); }))) return rc;
    if ((rc = prof_launch(c, PK_PROVE_BITS, (uint64_t)m * c->L, [&] { launch_prove_bits(a, c->stream); }))) return rc;
    if ((rc = prof_launch(c, PK_PROVE_TAIL, m, [&] { launch_prove_tail(a, c->stream); }))) return rc;
    if ((rc = hash_step(c, PK_HASH_SPEND, c->d_tr, (uint32_t)st.stride(), (uint32_t)st.bytes(), m))) return rc;
    if ((rc = prof_launch(c, PK_PROVE_RESP, (uint64_t)m * c->L, [&] { launch_prove_resp(a, c->stream); }))) return rc;
    if ((rc = dev_out_end(c, mem, out_proof + off * pb, a.proof, (size_t)m * pb))) return rc;
    if ((rc = dev_out_end(c, mem, out_prerefund + off * 96, a.prerefund, (size_t)m * 96))) return rc;
    HIPCK(c, hipMemcpyAsync(status + off, c->d_status, m, mem == ACT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if ((rc = prof_collect(c))) return rc;
  }
  return ACT_OK;
}

static int set_pubkey(act_ctx* c, const uint8_t w[32]) {
  if (c->w_valid && memcmp(c->w_cached, w, 32) == 0) return ACT_OK;
  int rc = stage_reserve(c, 5, 32 + GE_WORDS * 4 + 16); if (rc) return rc;
  uint8_t* d = c->d_stage[5];
  HIPCK(c, hipMemcpyAsync(d, w, 32, hipMemcpyHostToDevice, c->stream));
  launch_decode_points(d, 1, reinterpret_cast<uint32_t*>(d + 32), reinterpret_cast<uint32_t*>(d + 32 + GE_WORDS * 4), c->stream);
  uint32_t host[GE_WORDS + 1];
  HIPCK(c, hipMemcpyAsync(host, d + 32, sizeof(host), hipMemcpyDeviceToHost, c->stream));
  HIPCK(c, hipStreamSynchronize(c->stream));
  if (!host[GE_WORDS]) { c->err = "public key w is not a canonical Ristretto encoding"; return ACT_ERR_PARAMS; }
  c->w_pub = ge_load(host); memcpy(c->w_cached, w, 32); c->w_valid = true;
  return ACT_OK;
}

static int client_batch(act_ctx* c, size_t n, int mem, int label, const uint8_t* pre, const uint8_t w[32], const uint8_t* req,
                        const uint8_t* resp, const uint8_t* proofs, uint8_t* out_token, uint8_t* status) {
  HIPCK(c, hipSetDevice(c->device));
  int rc = set_pubkey(c, w); if (rc) return rc;
  const bool issuance = label == LABEL_RESPOND;
  const size_t pre_b = issuance ? 64 : 96, resp_b = issuance ? 160 : 128, pb = ProofLayout{c->L}.bytes();
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    ClientArgs a{}; a.P = c->P; a.w = c->w_pub; a.n = m; a.label = label; a.coords = c->d_coords; a.trs = c->d_trs; a.flags = c->d_flags;
    a.xof = c->d_xof; a.status = c->d_status;
    if ((rc = dev_in(c, 0, mem, pre + off * pre_b, (size_t)m * pre_b, &a.pre))) return rc;
    if ((rc = dev_in(c, 1, mem, resp + off * resp_b, (size_t)m * resp_b, &a.resp))) return rc;
    if (issuance) { if ((rc = dev_in(c, 3, mem, req + off * 128, (size_t)m * 128, &a.req))) return rc; }
    else { if ((rc = dev_in(c, 3, mem, proofs + off * pb, (size_t)m * pb, &a.proofs))) return rc; }
    if ((rc = dev_out_begin(c, 2, mem, out_token + off * 160, (size_t)m * 160, &a.out_token))) return rc;
    if (!issuance) {
      HIPCK(c, hipMemsetAsync(c->d_flags, 0, (size_t)m * 4, c->stream));
      if ((rc = prof_launch(c, PK_CLIENT, (uint64_t)m * c->L, [&] { launch_client_decode_com(a, c->stream); }))) return rc;
    }
    if ((rc = prof_launch(c, PK_CLIENT, m, [&] { launch_client_a(a, c->stream); }))) return rc;
    uint32_t len = c->P.prefix_len[label] + 40u * (issuance ? 7u : 6u);
    if ((rc = hash_step(c, PK_HASH_SMALL, c->d_trs, SMALL_TR_STRIDE, len, m))) return rc;
    if ((rc = prof_launch(c, PK_CLIENT, m, [&] { launch_client_b(a, c->stream); }))) return rc;
    if ((rc = dev_out_end(c, mem, out_token + off * 160, a.out_token, (size_t)m * 160))) return rc;
    HIPCK(c, hipMemcpyAsync(status + off, c->d_status, m, mem == ACT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if ((rc = prof_collect(c))) return rc;
  }
  return ACT_OK;
}
int act_issuance_to_credit_token_batch(act_ctx* c, size_t n, int mem, const uint8_t* pre, const uint8_t w[32], const uint8_t* req,
                                       const uint8_t* resp, uint8_t* out_token, uint8_t* status) {
  if (!c || !w || (n && (!pre || !req || !resp || !out_token || !status))) return ACT_ERR_ARG;
  return client_batch(c, n, mem, LABEL_RESPOND, pre, w, req, resp, nullptr, out_token, status);
}
int act_refund_to_credit_token_batch(act_ctx* c, size_t n, int mem, const uint8_t* prerefund, const uint8_t* proof, const uint8_t* refund,
                                     const uint8_t w[32], uint8_t* out_token, uint8_t* status) {
  if (!c || !w || (n && (!prerefund || !proof || !refund || !out_token || !status))) return ACT_ERR_ARG;
  return client_batch(c, n, mem, LABEL_REFUND, prerefund, w, nullptr, refund, proof, out_token, status);
}

int act_debug_last_spend_transcripts(act_ctx* c, size_t max_lanes, uint8_t* out, size_t* n_copied) {
  if (!c || !out || !n_copied) return ACT_ERR_ARG;
  HIPCK(c, hipSetDevice(c->device));
  const SpendTranscript st{c->L};
  size_t m = std::min(max_lanes, c->last_spend_lanes);
  std::vector<uint8_t> tmp(m * st.stride());
  HIPCK(c, hipMemcpy(tmp.data(), c->d_tr, tmp.size(), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < m; i++) memcpy(out + i * st.bytes(), tmp.data() + i * st.stride(), st.bytes());
  *n_copied = m;
  return ACT_OK;
}

int act_prof_enable(act_ctx* c, int on) { if (!c) return ACT_ERR_ARG; c->prof_on = on != 0; return ACT_OK; }
int act_prof_reset(act_ctx* c) {
  if (!c) return ACT_ERR_ARG;
  for (int i = 0; i < PK_COUNT; i++) { c->prof_ms[i] = 0; c->prof_launches[i] = 0; c->prof_lanes[i] = 0; }
  return ACT_OK;
}
int act_prof_kernel_count(const act_ctx*) { return PK_COUNT; }
const char* act_prof_kernel_name(const act_ctx*, int i) { return (i >= 0 && i < PK_COUNT) ? kProfNames[i] : ""; }
int act_prof_get(act_ctx* c, int i, double* ms_total, uint64_t* launches, uint64_t* lanes) {
  if (!c || i < 0 || i >= PK_COUNT) return ACT_ERR_ARG;
  if (ms_total) *ms_total = c->prof_ms[i];
  if (launches) *launches = c->prof_launches[i];
  if (lanes) *lanes = c->prof_lanes[i];
  return ACT_OK;
}

}  // extern "C"
