// engine.hip — host side of libact_mi355x.so: contexts, workspace, chunked kernel orchestration,
// the host-transcript mode (BLAKE3 of every Fiat–Shamir transcript on host threads, as the
// reference's src/transcript.rs does) and the C ABI of include/act_mi355x.h.
// There is no CPU compute path here: the host only moves bytes and (optionally) hashes them.
#include <algorithm>
#include <sys/random.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "kernels.h"
#include "rng_source.h"
#include "../../include/act_mi355x.h"

// ROCTX ranges (the tracing subsystem), compiled in with -DACT_ROCTX (tools/profile_round.sh builds that variant
// and runs it under `rocprofv3 --kernel-trace --marker-trace`): the pipeline the design describes -- phase A of chunk i+1 enqueued
// while chunk i's transcripts cross PCIe and are hashed on the host, then phase B -- as named host-side intervals on the same
// timeline as the kernels.  The product build carries none of it (no dependency on the profiler's library).
#if defined(ACT_ROCTX)
#include <rocprofiler-sdk-roctx/roctx.h>
namespace {
struct RoctxRange {
  explicit RoctxRange(const char* fmt, size_t a = 0, size_t b = 0) { char t[96]; snprintf(t, sizeof t, fmt, a, b); roctxRangePushA(t); }
  ~RoctxRange() { roctxRangePop(); }
};
}
#define ACT_RANGE_CAT2(a, b) a##b
#define ACT_RANGE_CAT(a, b) ACT_RANGE_CAT2(a, b)
#define ACT_RANGE(...) RoctxRange ACT_RANGE_CAT(act_roctx_range_, __LINE__)(__VA_ARGS__)
#else
#define ACT_RANGE(...) do { } while (0)
#endif

using namespace act;

// the measurement knobs (kernels.h TuneKey): defaults, and the one way to change them (act_tuning_set)
namespace act {
std::atomic<long> g_tune[T_COUNT] = {};
namespace {
struct TuneInit { TuneInit() { g_tune[T_STAGGER].store(-1); g_tune[T_SMALL_IN_FLIGHT].store(2); } } g_tune_init;
}
}  // namespace act


extern "C" void act_host_hash_many(const uint8_t* msgs, size_t stride, uint32_t len, size_t n, int max_threads, uint32_t* xof);   // host_pool.cpp

namespace {

const char* const kLabels[4] = {"request", "respond", "spend", "refund"};
const char kProtocolVersion[] = "curve25519-ristretto anonymous-credits v1.0";   // src/transcript.rs:29
// RFC 9496 appendix A.1: encoding of the ristretto255 generator
const uint8_t kGeneratorEnc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                   0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};

enum ProfId { PK_SPEND_PREP, PK_SPEND_PREP_A, PK_SPEND_PREP_B, PK_SPEND_PREP_C, PK_SPEND_PREP_JOIN, PK_SPEND_COORDS, PK_SPEND_BITS, PK_SPEND_ENC, PK_SPEND_TAIL, PK_HASH_SPEND, PK_SPEND_FINISH, PK_SIGN_A, PK_HASH_SMALL, PK_SIGN_B,
              PK_ISSUE_A, PK_ISSUE_CHECK, PK_REQUEST_A, PK_REQUEST_B, PK_PROVE_HEAD, PK_PROVE_BITS, PK_PROVE_ENC, PK_PROVE_TAIL, PK_PROVE_RESP,
              PK_CLIENT, PK_COPY_H2D, PK_COPY_D2H, PK_COUNT };
const char* const kProfNames[PK_COUNT] = {"k_spend_prep", "k_spend_prep_a", "k_spend_prep_b", "k_spend_prep_c", "k_spend_prep_join", "k_spend_coords", "k_spend_bits", "k_spend_enc", "k_spend_tail", "k_hash_xof(spend)", "k_spend_finish",
                                          "k_sign_a", "k_hash_xof(small)", "k_sign_b", "k_issue_a", "k_issue_check", "k_request_a",
                                          "k_request_b", "k_prove_head", "k_prove_bits", "k_prove_enc", "k_prove_tail", "k_prove_resp", "k_client_verify",
                                          "copy_h2d(bulk)", "copy_d2h(transcripts)"};

struct PendingProf { int id; hipEvent_t e0, e1; uint64_t lanes; };

}  // namespace

// One workspace slot: a stream plus every per-chunk device buffer.  Two slots let chunk i+1's kernels (and the
// host-side hashing of chunk i in host-transcript mode) overlap chunk i's low-occupancy head/tail kernels.
constexpr int HASH_PIECES = 8;
constexpr size_t TINY_MAX = 64;                 // lanes of a "tiny" call (one kernel, one copy each way: request_tiny ...)
constexpr size_t TINY_BYTES = (size_t)64 << 10; // its pinned + device buffers: inputs (secrets among them) in the first half, outputs in the second
constexpr size_t TINY_OUT = TINY_BYTES / 2;
constexpr size_t GROUP_CTR_WORDS = 1024;        // groups of 64 lanes per launch of a role-block kernel (launches of at most 8 192 lanes use 128)
struct Slot {
  hipStream_t stream = nullptr;
  uint8_t *d_tr = nullptr, *d_trs = nullptr, *d_status = nullptr;
  uint32_t *d_buckets = nullptr, *d_coords = nullptr, *d_d01 = nullptr, *d_xa = nullptr, *d_flags = nullptr, *d_xof = nullptr, *d_state = nullptr, *d_slot = nullptr, *d_naf = nullptr, *d_dig = nullptr;
  static constexpr int N_STAGE = 9;
  uint8_t* d_stage[N_STAGE] = {};   // staging for host-memory callers (grow-only); 5 = key decoding scratch, 6 = records unframed from wire bytes, 7 = their offsets, 8 = CBOR layout tables
  size_t d_stage_cap[N_STAGE] = {};
  size_t d_stage_dirty[N_STAGE] = {};     // bytes written since the last wipe (finish_call)
  size_t d_trs_dirty = 0;                 // lanes of d_trs that a sign-beside-the-check kernel wrote (k_sign_fused<CHECK> / before_verdict): wiped by finish_call
  std::vector<uint64_t> h_rel;            // message offsets of the chunk in flight, relative to its first byte (act_verify_spend_cbor_batch)
  uint8_t* h_tr = nullptr; size_t h_tr_cap = 0;        // pinned host buffers of the host-transcript mode
  uint32_t* h_xof = nullptr; size_t h_xof_cap = 0;
  uint8_t* h_cst = nullptr; size_t h_cst_cap = 0;      // pinned: a codec chunk's statuses on their way to the host reader (cbor_impl.inc)
  hipEvent_t h_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // one per returned piece (HASH_PIECES)
  hipEvent_t bits_ev = nullptr;        // recorded after this slot's k_spend_bits (staggering, spend_stage1)
  uint32_t* bits_sig = nullptr;        // signal memory: workgroups this slot's k_spend_bits launches have finished in the running call
  uint32_t bits_wgs = 0;               // ... and have launched
  hipEvent_t cp_in_ev = nullptr, cp_out_ev = nullptr;   // after this slot's copy in / copy out of a codec chunk (cbor_impl.inc: the two slots take turns on each link direction)
  std::vector<PendingProf> pending;
  size_t last_spend_lanes = 0;
};

// The context's last error text, under a lock of ITS OWN: act_last_error() must not wait behind a batch call that holds the context
// for seconds, and a failing call on one thread must not tear the string another thread is copying.
// The text of a handle's last failure.  A handle is shared between threads, and "last" must mean the calling thread's own last
// failing call, not whichever thread failed most recently: every assignment is also kept in a slot of the assigning thread, which
// act_last_error prefers.
struct ErrText {
  mutable std::mutex m; std::string s;
  struct Mine { const ErrText* of = nullptr; std::string text; };
  static Mine& mine() { thread_local Mine t; return t; }
  void keep(const std::string& v) const { Mine& t = mine(); t.of = this; t.text = v; }
  ErrText& operator=(const std::string& v) { { std::lock_guard<std::mutex> lk(m); s = v; } keep(v); return *this; }
  ErrText& operator=(const char* v) { return *this = std::string(v); }
  std::string get() const {
    const Mine& t = mine();
    if (t.of == this && !t.text.empty()) return t.text;
    std::lock_guard<std::mutex> lk(m); return s;
  }
  void forget_mine() const { Mine& t = mine(); if (t.of == this) t.text.clear(); }      // a new call of this thread on this handle begins
  operator std::string() const { return get(); }
};
struct act_ctx {
  int device = 0, L = 128;
  size_t max_batch = 0;
  DevParams P{};
  uint8_t henc[96]{};
  int tr_mode = ACT_TRANSCRIPT_HOST;
  int depth = 2;                       // chunks in flight (act_ctx_set_pipeline_depth): 2 = both slots, 1 = strictly one after the other
  int host_threads = 0;
  int streams_overlap = -1;            // 1 = the two slots' streams run side by side (measured at creation), 0 = they share a hardware queue
  ErrText err;
  Slot slots[2];
  std::mutex mu;                       // every batch entry point holds it: calls on one context are serialised, whatever thread they come from
  uint32_t* d_tables[4] = {nullptr, nullptr, nullptr, nullptr};     // shared with the other contexts of this device (table cache below)
  int fb_bits[4] = {0, 0, 0, 0};        // window width of each base's table
  uint32_t* d_half_h1 = nullptr;
  uint32_t* d_tables_ct = nullptr;     // the four scanned tables (msm.h fixed_base_acc_ct)
  uint8_t* d_tables_mf = nullptr;      // the four matrix-core table images (msm.h fixed_base_acc_mf)
  void* wire_layout = nullptr;         // the running wire-bytes call's CborDev (cbor_impl.inc)
  uint8_t* d_wire_flags = nullptr; size_t d_wire_flags_cap = 0;     // per message of a wire-bytes call: 0x80 = not the canonical encoding (cbor_impl.inc)
  // key cache
  uint8_t sk_cached[64]{}; bool sk_valid = false; DevKey key{};
  uint8_t w_cached[32]{}; bool w_valid = false; ge w_pub{};
  // profiling
  bool prof_on = false;
  double prof_ms[PK_COUNT]{}; uint64_t prof_launches[PK_COUNT]{}; uint64_t prof_lanes[PK_COUNT]{};
  hipEvent_t prof_base = nullptr;                                  // time origin of the launch intervals below
  std::vector<std::pair<float, float>> prof_iv[PK_COUNT];          // [start, end) of every launch, ms since prof_base
  // small-batch schedule (spend_small_locked): four more streams, the events that tie its streams together, per-proof scratch
  hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
  bool aux_used = false;               // the running call has put work on the aux streams: finish_call waits for them before it wipes
  std::vector<hipEvent_t> sm_ev;       // SM_EVENTS per sub-chunk, created on first use, kept
  uint32_t* d_small = nullptr; size_t d_small_cap = 0, d_small_dirty = 0;      // bytes
  std::atomic<size_t> small_max{8192}; // calls of at most this many proofs take the small-batch schedule (act_ctx_set_small_batch_max; 0 = never)
  // ... the caller's figure; otherwise device-transcript calls take that schedule up to twice the default (9 216 proofs 442 k/s against
  // 407 k, 16 384: 469 k against 464 k, beyond that the pipelined chunks win; host transcripts: 8 192. profiles/r06_midsize_small_limit.txt)
  std::atomic<bool> small_max_set{false};
  int last_spend_slot = 0;
  hipEvent_t last_bits_ev = nullptr;   // the most recently launched k_spend_bits of the running call
  uint32_t* last_bits_sig = nullptr; uint32_t last_bits_release = 0;      // ... its slot's counter and the count at which its last round is running
  bool wait_value = false;             // the device supports hipStreamWaitValue32 (act_ctx_create)
  double trace_wait_s = 0, trace_hash_s = 0; size_t trace_msgs = 0;      // ACT_TRACE accumulators
  double host_wait_s = 0, host_hash_s = 0; uint64_t host_hash_bytes = 0; // act_ctx_host_hash_stats: the calling thread's time in hash_end since the last reset
  // tiny calls (at most TINY_MAX lanes: the crate's one-item call shape): one pinned + one device buffer, one copy each way, one kernel
  uint8_t *d_tiny = nullptr, *h_tiny = nullptr;
  bool tiny_dirty = false;             // a tiny call staged inputs (secrets among them) there: finish_call wipes both on EVERY exit
  std::atomic<bool> tiny_on{true};     // act_ctx_set_tiny_calls
  uint32_t* d_group_ctr = nullptr;     // one word per group of 64 lanes (k_sign_fused ...: which role block arrives last), zero between launches
  uint8_t* d_tiny_tr = nullptr;        // TINY_MAX "request" transcripts of the fused issue kernel
  std::atomic<uint32_t> debug_ns_per_lane{0};    // act_debug_set_slowdown (test hook of the node dispatcher's load balance)
  std::atomic<int> debug_fail_signs{0};          // act_debug_fail_next_signs (test hook of the redeem failure contract)
};

namespace {

#define HIPCK(ctx, expr)                                                                            \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                               \
      (void)hipGetLastError(); /* reported through the C ABI: do not leave it sticky for other HIP users (torch) */ \
      return ACT_ERR_HIP;                                                                           \
    }                                                                                               \
  } while (0)

// timing events go on the stream the launch goes on (`stream`; the slot keeps the pending pair)
template <class F>
int prof_launch_on(act_ctx* c, Slot& sl, hipStream_t stream, int id, uint64_t lanes, F&& f) {
  if (!c->prof_on) { f(); return ACT_OK; }
  PendingProf p{id, nullptr, nullptr, lanes};
  HIPCK(c, hipEventCreate(&p.e0)); HIPCK(c, hipEventCreate(&p.e1));
  HIPCK(c, hipEventRecord(p.e0, stream));
  f();
  HIPCK(c, hipEventRecord(p.e1, stream));
  sl.pending.push_back(p);
  return ACT_OK;
}
template <class F>
int prof_launch(act_ctx* c, Slot& sl, int id, uint64_t lanes, F&& f) { return prof_launch_on(c, sl, sl.stream, id, lanes, static_cast<F&&>(f)); }
int prof_collect(act_ctx* c, Slot& sl) {
  for (auto& p : sl.pending) {
    HIPCK(c, hipEventSynchronize(p.e1));
    float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, p.e0, p.e1));
    c->prof_ms[p.id] += ms; c->prof_launches[p.id]++; c->prof_lanes[p.id] += p.lanes;
    if (c->prof_base) {
      float t0 = 0; HIPCK(c, hipEventElapsedTime(&t0, c->prof_base, p.e0));
      c->prof_iv[p.id].emplace_back(t0, t0 + ms);
      static FILE* const tl = [] { const char* f = getenv("ACT_TIMELINE_FILE"); return f ? fopen(f, "a") : (FILE*)nullptr; }();     // diagnostics: one line per launch / copy
      if (tl) { fprintf(tl, "%s,%d,%.3f,%.3f,%llu\n", kProfNames[p.id], (int)(&sl - c->slots), t0, t0 + ms, (unsigned long long)p.lanes); fflush(tl); }
    }
    hipEventDestroy(p.e0); hipEventDestroy(p.e1);
  }
  sl.pending.clear();
  return ACT_OK;
}

int stage_reserve(act_ctx* c, Slot& sl, int slot, size_t bytes) {
  // whatever a call reserves it is about to write: the extent is wiped when the call ends (finish_call), secret or not
  if (bytes <= sl.d_stage_cap[slot]) { sl.d_stage_dirty[slot] = std::max(sl.d_stage_dirty[slot], bytes); return ACT_OK; }
  HIPCK(c, hipStreamSynchronize(sl.stream));
  if (sl.d_stage[slot]) {
    if (sl.d_stage_dirty[slot]) HIPCK(c, hipMemset(sl.d_stage[slot], 0, sl.d_stage_dirty[slot]));     // never hand secrets back to the allocator
    HIPCK(c, hipFree(sl.d_stage[slot]));
  }
  sl.d_stage[slot] = nullptr; sl.d_stage_cap[slot] = 0; sl.d_stage_dirty[slot] = 0;
  HIPCK(c, hipMalloc(&sl.d_stage[slot], bytes));
  sl.d_stage_cap[slot] = bytes; sl.d_stage_dirty[slot] = bytes;
  return ACT_OK;
}
// Host memory the device can read in place: hipHostMalloc / hipHostRegister memory is mapped into the device's address space, and the
// spend-proof kernels read every proof byte once or twice, spread over the kernels' whole run time -- 9 GB/s at full rate, a fraction
// of the link.  Reading in place takes the staging copy out from in front of a single-chunk call's first kernel (tools/
// midsize_probe.py: 4 096 proofs 11.9 -> 10.5 ms, 16 384: 39.6 -> 35.9; what the same call takes from HBM: 10.1 / 35.3).  Only for proof
// RECORDS (public, read in 2 KiB-contiguous wavefront loads): wire bytes and secrets keep their staged (secrets: wiped) copies.  Returns the device address of p, or null (pageable memory,
// a range that leaves its allocation, memory pinned under another device, or ACT_NO_MAPPED_READS set).
const uint8_t* mapped_view(const act_ctx* c, const uint8_t* p, size_t bytes) {
  if (tune(T_NO_MAPPED_READS) || !p || !bytes) return nullptr;
  hipPointerAttribute_t a0{}, a1{};
  if (hipPointerGetAttributes(&a0, p) != hipSuccess || hipPointerGetAttributes(&a1, p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost || !a0.devicePointer || !a1.devicePointer) return nullptr;
  if (a0.device != c->device || a1.device != c->device) return nullptr;      // pinned while another device was current: staged as before
  if ((const uint8_t*)a1.devicePointer - (const uint8_t*)a0.devicePointer != (ptrdiff_t)(bytes - 1)) return nullptr;
  return (const uint8_t*)a0.devicePointer;
}
// device view of `bytes` of caller memory: the pointer itself (device memory) or a staged H2D copy
int dev_in(act_ctx* c, Slot& sl, int slot, int mem, const uint8_t* p, size_t bytes, const uint8_t** out) {
  if (mem == ACT_MEM_DEVICE || bytes == 0) { *out = p; return ACT_OK; }
  int rc = stage_reserve(c, sl, slot, bytes); if (rc) return rc;
  hipError_t ce = hipSuccess;
  rc = prof_launch(c, sl, PK_COPY_H2D, bytes, [&] { ce = hipMemcpyAsync(sl.d_stage[slot], p, bytes, hipMemcpyHostToDevice, sl.stream); }); if (rc) return rc;
  HIPCK(c, ce);
  sl.d_stage_dirty[slot] = std::max(sl.d_stage_dirty[slot], bytes);
  *out = sl.d_stage[slot];
  return ACT_OK;
}
int dev_out_begin(act_ctx* c, Slot& sl, int slot, int mem, uint8_t* p, size_t bytes, uint8_t** out) {
  if (mem == ACT_MEM_DEVICE || bytes == 0) { *out = p; return ACT_OK; }
  int rc = stage_reserve(c, sl, slot, bytes); if (rc) return rc;
  sl.d_stage_dirty[slot] = std::max(sl.d_stage_dirty[slot], bytes);
  *out = sl.d_stage[slot];
  return ACT_OK;
}
int dev_out_end(act_ctx* c, Slot& sl, int mem, uint8_t* host_p, const uint8_t* dev_p, size_t bytes) {
  if (mem == ACT_MEM_DEVICE || bytes == 0) return ACT_OK;
  HIPCK(c, hipMemcpyAsync(host_p, dev_p, bytes, hipMemcpyDeviceToHost, sl.stream));
  return ACT_OK;
}
int copy_status_out(act_ctx* c, Slot& sl, int mem, uint8_t* dst, uint32_t m) {
  HIPCK(c, hipMemcpyAsync(dst, sl.d_status, m, mem == ACT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, sl.stream));
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

// hash n messages of `len` bytes at `stride` (host memory) into xof[n][16] on this call's share of the process-wide worker pool
// (host_pool.cpp): groups of sixteen through the SIMD routine of host_hash.cpp, the remainder through the scalar routine
void host_hash_many(const act_ctx* c, const uint8_t* msgs, size_t stride, uint32_t len, size_t n, uint32_t* xof) {
  act_host_hash_many(msgs, stride, len, n, c->host_threads, xof);
}

// transcript hashing step, split so that callers can overlap the host part with other slots' GPU work:
//   hash_begin : device mode -> launch k_hash_xof;  host mode -> enqueue the D2H copy of the pre-images
//   hash_end   : device mode -> nothing;            host mode -> wait for the copy, hash on host threads, enqueue H2D of the XOF words
// pieces a batch of n pre-images comes back in: a piece must be worth waking the worker pool for (~60 us a time), so short batches
// come back whole
inline int hash_pieces(uint32_t n) { const uint32_t k = n / 512u; return k < 1u ? 1 : (k > (uint32_t)HASH_PIECES ? HASH_PIECES : (int)k); }
int hash_begin(act_ctx* c, Slot& sl, int prof_id, const uint8_t* d_msgs, uint32_t stride, uint32_t len, uint32_t n) {
  ACT_RANGE("transcript.hash_begin n=%zu (device: k_hash_xof; host: D2H of the pre-images queued)", (size_t)n);
  if (c->tr_mode == ACT_TRANSCRIPT_DEVICE) {
    HashArgs h{d_msgs, stride, len, n, sl.d_xof, nullptr};
    // few long messages (a small prove_spend / verify call's "spend" transcripts): sixteen lanes per message (k_hash_xof_par)
    if (len >= 2048 && n <= 4096) return prof_launch(c, sl, prof_id, n, [&] { launch_hash_par(h, sl.stream); });
    return prof_launch(c, sl, prof_id, n, [&] { launch_hash(h, sl.stream); });
  }
  size_t bytes = (size_t)n * stride;
  if (bytes > sl.h_tr_cap) {
    HIPCK(c, hipStreamSynchronize(sl.stream));
    if (sl.h_tr) HIPCK(c, hipHostFree(sl.h_tr));
    sl.h_tr = nullptr; sl.h_tr_cap = 0;
    HIPCK(c, hipHostMalloc(&sl.h_tr, bytes, hipHostMallocDefault)); sl.h_tr_cap = bytes;
  }
  if ((size_t)n * 64 > sl.h_xof_cap) {
    HIPCK(c, hipStreamSynchronize(sl.stream));
    if (sl.h_xof) HIPCK(c, hipHostFree(sl.h_xof));
    sl.h_xof = nullptr; sl.h_xof_cap = 0;
    HIPCK(c, hipHostMalloc(&sl.h_xof, (size_t)n * 64, hipHostMallocDefault)); sl.h_xof_cap = (size_t)n * 64;
  }
  // the pre-images come back in HASH_PIECES pieces, each followed by an event, so that hash_end can hash piece k while
  // piece k+1 is still crossing PCIe
  const int pieces = hash_pieces(n);
  for (int k = 0; k < pieces; k++) {
    size_t i0 = (size_t)n * k / pieces, i1 = (size_t)n * (k + 1) / pieces;
    if (!sl.h_ev[k]) HIPCK(c, hipEventCreateWithFlags(&sl.h_ev[k], hipEventDisableTiming));
    if (i1 > i0) {
      hipError_t ce = hipSuccess;
      int rc = prof_launch(c, sl, PK_COPY_D2H, (i1 - i0) * stride, [&] { ce = hipMemcpyAsync(sl.h_tr + i0 * stride, d_msgs + i0 * stride, (i1 - i0) * stride, hipMemcpyDeviceToHost, sl.stream); });
      if (rc) return rc;
      HIPCK(c, ce);
    }
    HIPCK(c, hipEventRecord(sl.h_ev[k], sl.stream));
  }
  return ACT_OK;
}
int hash_end(act_ctx* c, Slot& sl, uint32_t stride, uint32_t len, uint32_t n) {
  if (c->tr_mode == ACT_TRANSCRIPT_DEVICE) return ACT_OK;
  ACT_RANGE("transcript.hash_end n=%zu (wait for D2H pieces, BLAKE3 on the host pool, H2D of the XOF words)", (size_t)n);
  static const bool trace = getenv("ACT_TRACE") != nullptr;      // where the host side of the host-transcript mode spends its time
  double& t_wait = c->trace_wait_s; double& t_hash = c->trace_hash_s; size_t& n_msgs = c->trace_msgs;      // per context: contexts run on their own threads
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const int pieces = hash_pieces(n);
  for (int k = 0; k < pieces; k++) {
    size_t i0 = (size_t)n * k / pieces, i1 = (size_t)n * (k + 1) / pieces;
    const double t0 = now();
    HIPCK(c, hipEventSynchronize(sl.h_ev[k]));
    const double t1 = now();
    if (i1 > i0) host_hash_many(c, sl.h_tr + i0 * stride, stride, len, i1 - i0, sl.h_xof + i0 * 16);
    const double t2 = now();
    t_wait += t1 - t0; t_hash += t2 - t1;
    c->host_wait_s += t1 - t0; c->host_hash_s += t2 - t1; c->host_hash_bytes += (uint64_t)(i1 - i0) * len;      // act_ctx_host_hash_stats
  }
  if (trace && len > 1024) {
    n_msgs += n;
    if (n_msgs >= ((size_t)1 << 18)) { fprintf(stderr, "[act trace] %zu transcripts: waited for the device %.1f ms, hashed %.1f ms (%.2f GB/s)\n", n_msgs, 1e3 * t_wait, 1e3 * t_hash, n_msgs * (double)len / t_hash / 1e9); n_msgs = 0; t_wait = t_hash = 0; }
  }
  HIPCK(c, hipMemcpyAsync(sl.d_xof, sl.h_xof, (size_t)n * 64, hipMemcpyHostToDevice, sl.stream));
  return ACT_OK;
}
int hash_step(act_ctx* c, Slot& sl, int prof_id, const uint8_t* d_msgs, uint32_t stride, uint32_t len, uint32_t n) {
  int rc = hash_begin(c, sl, prof_id, d_msgs, stride, len, n); if (rc) return rc;
  return hash_end(c, sl, stride, len, n);
}

// decode one point on the device (public keys): returns ACT_ERR_PARAMS if it is not a canonical encoding
int decode_one(act_ctx* c, const uint8_t enc[32], ge* out) {
  Slot& sl = c->slots[0];
  int rc = stage_reserve(c, sl, 5, 32 + GE_WORDS * 4 + 16); if (rc) return rc;
  sl.d_stage_dirty[5] = std::max<size_t>(sl.d_stage_dirty[5], 32 + GE_WORDS * 4 + 16);
  uint8_t* d = sl.d_stage[5];
  HIPCK(c, hipMemcpyAsync(d, enc, 32, hipMemcpyHostToDevice, sl.stream));
  launch_decode_points(d, 1, reinterpret_cast<uint32_t*>(d + 32), reinterpret_cast<uint32_t*>(d + 32 + GE_WORDS * 4), sl.stream);
  uint32_t host[GE_WORDS + 1];
  HIPCK(c, hipMemcpyAsync(host, d + 32, sizeof(host), hipMemcpyDeviceToHost, sl.stream));
  HIPCK(c, hipStreamSynchronize(sl.stream));
  if (!host[GE_WORDS]) { c->err = "public key w is not a canonical Ristretto encoding"; return ACT_ERR_PARAMS; }
  *out = ge_load(host);
  return ACT_OK;
}
// equality of two byte strings in time that does not depend on where they differ (private keys are compared: cache hit, merging)
inline bool ct_equal(const uint8_t* a, const uint8_t* b, size_t n) {
  uint32_t d = 0;
  for (size_t i = 0; i < n; i++) d |= (uint32_t)(a[i] ^ b[i]);
  return d == 0;
}
int set_key(act_ctx* c, const uint8_t sk[64]) {
  if (c->sk_valid && ct_equal(c->sk_cached, sk, 64)) return ACT_OK;
  // nothing of the cached key changes unless the whole new key is good (a rejected w must not leave its x behind)
  uint32_t w[8]; memcpy(w, sk, 32);
  ge wpt;
  int rc = decode_one(c, sk + 32, &wpt); if (rc) return rc;
  c->sk_valid = false;
  c->key.x = sc_from_words(w); c->key.w = wpt;
  memcpy(c->sk_cached, sk, 64); c->sk_valid = true;
  return ACT_OK;
}
int set_pubkey(act_ctx* c, const uint8_t w[32]) {
  if (c->w_valid && memcmp(c->w_cached, w, 32) == 0) return ACT_OK;
  int rc = decode_one(c, w, &c->w_pub); if (rc) return rc;
  memcpy(c->w_cached, w, 32); c->w_valid = true;
  return ACT_OK;
}


#include "workspace_impl.inc"     // fixed-base tables shared between the contexts of one device (reference counted); workspace_alloc

// Do the two slots' streams run side by side?  HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues per
// device and priority (default 4); a process that holds several contexts next to the streams of its framework (measured: bench.py's
// engine + a node handle + torch) runs out, both streams of a context land on one queue, and its two-chunk pipeline silently
// runs its chunks one after the other (415 k instead of 466 k verifies/s from host memory).  Round 3 set GPU_MAX_HW_QUEUES from a
// library constructor -- a setenv in somebody else's process, racy against getenv in a multi-threaded host and without effect
// once HIP is initialised.  Now the context MEASURES it: one idle wavefront of `ticks` on each stream, started together, takes
// one `ticks` side by side and two on a shared queue.
int streams_overlap_probe(act_ctx* c, int* overlap) {
  hipStream_t s0 = c->slots[0].stream, s1 = c->slots[1].stream;
  hipEvent_t e0, e1, e2;
  HIPCK(c, hipEventCreate(&e0)); HIPCK(c, hipEventCreate(&e1)); HIPCK(c, hipEventCreate(&e2));
  const uint32_t ticks = 30000;                   // 0.3 ms of the 100 MHz counter
  launch_spin(1, s0); launch_spin(1, s1);         // code load
  HIPCK(c, hipStreamSynchronize(s0)); HIPCK(c, hipStreamSynchronize(s1));
  HIPCK(c, hipEventRecord(e0, s0));
  HIPCK(c, hipStreamWaitEvent(s1, e0, 0));
  launch_spin(ticks, s0); launch_spin(ticks, s1);
  HIPCK(c, hipEventRecord(e1, s0)); HIPCK(c, hipEventRecord(e2, s1));
  HIPCK(c, hipEventSynchronize(e1)); HIPCK(c, hipEventSynchronize(e2));
  float t1 = 0, t2 = 0;
  HIPCK(c, hipEventElapsedTime(&t1, e0, e1)); HIPCK(c, hipEventElapsedTime(&t2, e0, e2));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  *overlap = std::max(t1, t2) < 1.6f * (ticks / 1e5f) ? 1 : 0;
  return ACT_OK;
}
// On a GPU that other work is using (other contexts being created, eight ranks on one device) a single probe can read as "no
// overlap" although the streams are fine: the verdict is the best of three.
int streams_overlap_probe3(act_ctx* c, int* overlap) {
  *overlap = 0;
  for (int t = 0; t < 3 && !*overlap; t++) { int rc = streams_overlap_probe(c, overlap); if (rc) return rc; }
  return ACT_OK;
}
// ... and if they share a queue, slot 1 moves to a stream of another priority: the runtime keeps a separate set of hardware queues
// per priority, so the two cannot alias whatever else the process has created.  What is left is reported by
// act_ctx_streams_overlap() and documented for the embedding process (GPU_MAX_HW_QUEUES).
int streams_settle(act_ctx* c) {
  if (tune(T_NO_STREAM_PROBE)) { c->streams_overlap = -1; return ACT_OK; }
  int ov = 0, rc = streams_overlap_probe3(c, &ov); if (rc) return rc;
  if (!ov) {
    int least = 0, greatest = 0;
    HIPCK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int prio : {greatest, least}) {
      if (ov || least == greatest) break;
      hipStream_t s = nullptr;
      if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); continue; }
      HIPCK(c, hipStreamSynchronize(c->slots[1].stream));
      HIPCK(c, hipStreamDestroy(c->slots[1].stream));
      c->slots[1].stream = s;
      if ((rc = streams_overlap_probe3(c, &ov))) return rc;
    }
  }
  c->streams_overlap = ov;
  return ACT_OK;
}

int sync_all(act_ctx* c) {
  for (Slot& sl : c->slots) { HIPCK(c, hipStreamSynchronize(sl.stream)); int rc = prof_collect(c, sl); if (rc) return rc; }
  return ACT_OK;
}

// End of every entry point that ran kernels: wipe what the call left behind in the context's own memory, then wait.
// The reference zeroizes its secret-bearing values when they go out of scope (#[derive(ZeroizeOnDrop)],
// /root/reference/src/lib.rs:160, 362, 393, 878); here that means the staging copies of host buffers (tokens,
// PreIssuance, rng, keys), the signer's nonces / the prover's r3, r* (d_state), the prover's k* h2 terms (d_d01) and the
// per-proof Pippenger buckets, whose contents depend on secret scalar digits (x in A1, (e+x)^-1, r1 r2 ...).
// Only the extent this call touched is cleared: n lanes (at most max_batch) of the per-proof buffers.
int finish_call(act_ctx* c, size_t n) {
  const size_t lanes = std::min(n, c->max_batch);
  // a small-batch call that returned early (a HIP error half way through its schedule) may still have kernels on the aux streams
  // writing d_small and the slot buffers: nothing is wiped under them
  if (c->aux_used) { c->aux_used = false; for (hipStream_t a : c->aux) if (a) HIPCK(c, hipStreamSynchronize(a)); }
  for (Slot& sl : c->slots) {
    for (int i = 0; i < Slot::N_STAGE; i++)
      if (sl.d_stage_dirty[i]) { HIPCK(c, hipMemsetAsync(sl.d_stage[i], 0, sl.d_stage_dirty[i], sl.stream)); sl.d_stage_dirty[i] = 0; }
    if (&sl == &c->slots[0] && c->d_small_dirty) {          // the small-batch schedule's partial sums ((e_bar - x gamma) A' among them) and buckets
      HIPCK(c, hipMemsetAsync(c->d_small, 0, c->d_small_dirty, sl.stream)); c->d_small_dirty = 0;
    }
    // A kernel that signs BESIDE the check (k_sign.hip k_sign_fused<CHECK>, or before_verdict) has written e, enc(A), X_A, X_g, Y_A,
    // Y_g of EVERY lane into its small transcript, the rejected ones included: a complete signature over a K (or K') that nothing has
    // verified.  The reference draws e only after its checks (src/lib.rs:638-643, 842-846); here such a lane's transcript dies with
    // the call.  (Accepted lanes' transcripts hold only what their response publishes; they go too.)
    if (sl.d_trs_dirty) { HIPCK(c, hipMemsetAsync(sl.d_trs, 0, std::min(sl.d_trs_dirty, c->max_batch) * SMALL_TR_STRIDE, sl.stream)); sl.d_trs_dirty = 0; }
    // the tiny calls' staging pair: the kernels zero their inputs and the paths wipe the pinned side when they succeed, but an early
    // return between the copy in and that wipe skipped both
    if (&sl == &c->slots[0] && c->tiny_dirty && c->d_tiny) {
      HIPCK(c, hipMemsetAsync(c->d_tiny, 0, TINY_BYTES, sl.stream));
      volatile uint8_t* q = c->h_tiny; for (size_t i = 0; i < TINY_BYTES; i++) q[i] = 0;
      c->tiny_dirty = false;
    }
    if (lanes) {
      HIPCK(c, hipMemsetAsync(sl.d_state, 0, lanes * 24 * 4, sl.stream));
      HIPCK(c, hipMemsetAsync(sl.d_d01, 0, lanes * 3 * GE_WORDS * 4, sl.stream));
      // the per-proof kernels' bucket sets -- and, where the range kernel's per-lane buckets (n * L lane areas, nothing of the issuer's
      // in them) reach into the region those sets live in (calls shorter than max_batch), that part too
      const size_t sets = std::min(lanes * (size_t)std::max(c->L, PREP_BUCKET_SETS), c->max_batch * (size_t)PREP_BUCKET_SETS);
      HIPCK(c, hipMemsetAsync(sl.d_buckets, 0, sets * BUCKET_WORDS * 4, sl.stream));
    }
  }
  return sync_all(c);
}

// Every entry point that runs kernels owns one of these from its first line: it serialises the callers of a context (the
// Rust binding keeps the handle inside `Params`, which safe code may share between threads) and runs finish_call on EVERY
// exit -- a failed call must not leave tokens, rng bytes, nonces or key-dependent buckets behind any more than a
// successful one does.  On the failure path the wipe is best effort and the first error is the one reported.
struct Call {
  act_ctx* c; size_t n; std::unique_lock<std::mutex> lk; bool finished = false;
  Call(act_ctx* c_, size_t n_) : c(c_), n(n_), lk(c_->mu) { c->err.forget_mine(); }
  void slow() const {           // act_debug_set_slowdown: this context as a slower GPU
    const uint32_t ns = c->debug_ns_per_lane.load();
    if (ns && n) std::this_thread::sleep_for(std::chrono::nanoseconds((uint64_t)ns * n));
  }
  int finish() { finished = true; slow(); return finish_call(c, n); }
  ~Call() {
    if (finished) return;
    const std::string first = c->err;
    (void)hipSetDevice(c->device);
    // a call that failed half way may have left a role-block kernel's arrival counters mid-count: the next launch must find zeros
    if (c->d_group_ctr) { for (Slot& sl : c->slots) if (sl.stream) (void)hipStreamSynchronize(sl.stream); (void)hipMemset(c->d_group_ctr, 0, GROUP_CTR_WORDS * 4); (void)hipDeviceSynchronize(); }
    (void)finish_call(c, n);
    if (!first.empty()) c->err = first;
  }
};

int from_uniform_on_device(int device, const uint8_t* in64, int n, uint8_t* out_enc) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ACT_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return ACT_ERR_HIP;
  uint8_t* d = nullptr;
  if (hipMalloc(&d, (size_t)n * 96) != hipSuccess) return ACT_ERR_HIP;
  int rc = ACT_OK;
  if (hipMemcpy(d, in64, (size_t)n * 64, hipMemcpyHostToDevice) != hipSuccess) rc = ACT_ERR_HIP;
  if (!rc) { launch_from_uniform(d, (uint32_t)n, d + (size_t)n * 64, nullptr); if (hipDeviceSynchronize() != hipSuccess) rc = ACT_ERR_HIP; }
  if (!rc && hipMemcpy(out_enc, d + (size_t)n * 64, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess) rc = ACT_ERR_HIP;
  (void)hipFree(d);
  if (rc) (void)hipGetLastError();
  return rc;
}

// rng slices for the signing phase of issue / refund.  PER_LANE needs no knowledge of the statuses; SEQUENTIAL hands
// consecutive 128-byte slices to the accepted lanes in lane order, so it must read the chunk's statuses back first.
int prepare_rng_slots(act_ctx* c, Slot& sl, uint32_t m, size_t off, int mem, const uint8_t* rng, int rng_mode, size_t* seq_cursor,
                      const uint8_t** d_rng) {
  if (rng_mode == ACT_RNG_PER_LANE) {          // lane i owns slice off + i: filled on the device, nothing for the host to wait for
    if (mem == ACT_MEM_DEVICE) { *d_rng = rng; launch_iota(sl.d_slot, m, (uint32_t)off, sl.stream); }
    else { int rc = dev_in(c, sl, 3, mem, rng + off * 128, (size_t)m * 128, d_rng); if (rc) return rc; launch_iota(sl.d_slot, m, 0u, sl.stream); }
    return ACT_OK;
  }
  std::vector<uint32_t> slot(m);
  {
    std::vector<uint8_t> st(m);
    HIPCK(c, hipMemcpyAsync(st.data(), sl.d_status, m, hipMemcpyDeviceToHost, sl.stream));
    HIPCK(c, hipStreamSynchronize(sl.stream));
    size_t cur = *seq_cursor, base = cur;
    for (uint32_t i = 0; i < m; i++) { slot[i] = (uint32_t)(mem == ACT_MEM_DEVICE ? cur : cur - base); if (st[i] == 0) cur++; }
    *seq_cursor = cur;
    if (mem == ACT_MEM_DEVICE) *d_rng = rng;
    else if (cur > base) { int rc = dev_in(c, sl, 3, mem, rng + base * 128, (cur - base) * 128, d_rng); if (rc) return rc; }
    else *d_rng = reinterpret_cast<const uint8_t*>(sl.d_slot);   // no accepted lane reads it
  }
  HIPCK(c, hipMemcpyAsync(sl.d_slot, slot.data(), (size_t)m * 4, hipMemcpyHostToDevice, sl.stream));
  HIPCK(c, hipStreamSynchronize(sl.stream));   // `slot` is a stack-lifetime host buffer
  return ACT_OK;
}

// counters of the role-block kernels (which block of a group arrives last): a set per slot, because the two slots' launches overlap
uint32_t* group_counters(act_ctx* c, const Slot& sl) { return c->d_group_ctr + (size_t)(&sl - c->slots) * (GROUP_CTR_WORDS / 2); }
// (tiny calls: see request_tiny further down)  ACT_NO_FUSED_TINY=1 keeps the multi-launch paths (A/B; the tests compare the two)
bool tiny_enabled(const act_ctx* c) { return !tune(T_NO_FUSED_TINY) && c->tiny_on.load(); }
int sign_phase(act_ctx* c, Slot& sl, uint32_t m, int label, const uint8_t* d_rng, const uint8_t* d_camount, uint8_t* d_out) {
  ACT_RANGE("sign.phaseA+hash+phaseB m=%zu label=%zu", (size_t)m, (size_t)label);
  sl.d_trs_dirty = std::max(sl.d_trs_dirty, (size_t)m);
  if (m <= TINY_MAX && tiny_enabled(c)) {
    // tiny calls: phase A, the transcript's BLAKE3 (one chunk, in the kernel: same bytes as either transcript mode) and phase B in
    // ONE launch, the five transcript points on three wavefronts (k_sign.hip k_sign_fused); nothing secret reaches global memory
    SignFusedArgs f{}; f.P = c->P; f.K = c->key; f.n = m; f.label = label; f.xa = sl.d_xa; f.rng_slot = sl.d_slot; f.c_amount = d_camount;
    f.status_in = sl.d_status; f.rng = d_rng; f.out = d_out; f.status = sl.d_status; f.trs = sl.d_trs; f.group_counter = group_counters(c, sl); f.pbk = sl.d_buckets;
    return prof_launch(c, sl, PK_SIGN_A, m, [&] { launch_sign_fused(f, false, sl.stream); });
  }
  SignArgs s{}; s.P = c->P; s.K = c->key; s.n = m; s.label = label; s.xa = sl.d_xa; s.status = sl.d_status; s.rng_slot = sl.d_slot;
  s.rng = d_rng; s.c_amount = d_camount; s.trs = sl.d_trs; s.state = sl.d_state; s.xof = sl.d_xof; s.out = d_out; s.pbk = sl.d_buckets;
  int rc;
  if ((rc = prof_launch(c, sl, PK_SIGN_A, m, [&] { launch_sign_a(s, sl.stream); }))) return rc;
  uint32_t len = c->P.prefix_len[label] + 40u * (label == LABEL_RESPOND ? 7u : 6u);
  if ((rc = hash_step(c, sl, PK_HASH_SMALL, sl.d_trs, SMALL_TR_STRIDE, len, m))) return rc;
  return prof_launch(c, sl, PK_SIGN_B, m, [&] { launch_sign_b(s, sl.stream); });
}

// ---- spend verification, pipelined over two slots ----------------------------------------------------------
struct SpendChunk { uint32_t m = 0; size_t off = 0; const uint8_t* d_proofs = nullptr; uint8_t* d_kprime = nullptr; uint8_t* d_out = nullptr; SpendArgs a{}; bool stagger = false; };

// stage 1: everything up to (and including the start of) the transcript hash
int spend_stage1(act_ctx* c, Slot& sl, SpendChunk& ch) {
  ACT_RANGE("spend.phaseA off=%zu m=%zu (prep, bits, enc, tail enqueued; transcripts on their way)", ch.off, (size_t)ch.m);
  const SpendTranscript st{c->L};
  SpendArgs& a = ch.a;
  a = SpendArgs{}; a.P = c->P; a.K = c->key; a.proofs = ch.d_proofs; a.n = ch.m; a.tr = sl.d_tr; a.tr_stride = (uint32_t)st.stride();
  a.coords = sl.d_coords; a.d01 = sl.d_d01; a.buckets = sl.d_buckets; a.xa = sl.d_xa; a.flags = sl.d_flags; a.xof = sl.d_xof; a.status = sl.d_status;
  a.kprime_enc = ch.d_kprime; a.naf = sl.d_naf; a.dig = sl.d_dig; a.pbk = sl.d_buckets;
  int rc;
  if ((rc = prof_launch(c, sl, PK_SPEND_PREP, ch.m, [&] { launch_spend_prep(a, sl.stream); }))) return rc;
  // Staggering.  Left alone, the range kernels of the two chunks in flight run side by side, finish together, and then
  // both chunks do their copies (H2D of the next proofs, D2H of the transcripts) at the same moment with no kernel
  // running: rocprofv3 --memory-copy-trace showed 38 ms of PCIe per 290 ms of compute fully exposed for host-memory
  // callers.  When a call moves data over PCIe, every range kernel therefore waits for the previous chunk's (one of them
  // fills the GPU anyway): chunk i's encodes, tail, copies and host hashing then run under chunk i+1's range kernel.
  // "Waits for the previous chunk's" = for the START OF ITS LAST ROUND of workgroups, not for its end: a range kernel drains for half a
  // round on average (waves of one round end up to ~1 ms apart; a round is ~2 ms, a 16 384-proof chunk 16 of them), and the other
  // chunk's kernel is what can fill those CUs.  k_spend_bits counts finished workgroups in the slot's signal word and the next
  // launch sits behind a hipStreamWaitValue32 for "all but the resident ones have finished" (every workgroup of the earlier kernel
  // is then placed; the later one gets what frees up).  tools/soft_gate_probe.hip is the mechanism alone; measured on the bench
  // workload in profiles/r06_ab_soft_stagger.txt.  Only an ordering hint: no result depends on it.  Without the device attribute
  // (or with the hard_stagger knob) the wait is the previous kernel's completion event.
  const bool soft = ch.stagger && c->wait_value && sl.bits_sig && !tune(T_HARD_STAGGER);
  bool gated = false;
  if (soft && c->last_bits_sig) {
    if (hipStreamWaitValue32(sl.stream, c->last_bits_sig, c->last_bits_release, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess) gated = true;
    else { (void)hipGetLastError(); c->wait_value = false; }      // refused after all (never seen): this context orders by events from here on
  }
  if (ch.stagger && !gated && c->last_bits_ev) HIPCK(c, hipStreamWaitEvent(sl.stream, c->last_bits_ev, 0));
  a.progress = soft ? sl.bits_sig : nullptr;
  if ((rc = prof_launch(c, sl, PK_SPEND_BITS, (uint64_t)ch.m * c->L, [&] { launch_spend_bits(a, sl.stream); }))) return rc;
  if (ch.stagger) {                            // the completion event is recorded in either case: it is what a refused wait falls back to
    if (!sl.bits_ev) HIPCK(c, hipEventCreateWithFlags(&sl.bits_ev, hipEventDisableTiming));
    HIPCK(c, hipEventRecord(sl.bits_ev, sl.stream));
    c->last_bits_ev = sl.bits_ev;
  }
  c->last_bits_sig = nullptr;
  if (soft) {
    const uint32_t wgs = spend_bits_workgroups(a), resident = 2u * device_cus(), before = sl.bits_wgs;
    sl.bits_wgs += wgs;
    c->last_bits_sig = sl.bits_sig; c->last_bits_release = wgs > resident ? sl.bits_wgs - resident : before + 1;
  }
  if ((rc = prof_launch(c, sl, PK_SPEND_ENC, (uint64_t)ch.m * c->L * 2, [&] { launch_spend_enc(a, sl.stream); }))) return rc;
  if ((rc = prof_launch(c, sl, PK_SPEND_TAIL, ch.m, [&] { launch_spend_tail(a, sl.stream); }))) return rc;
  if ((rc = hash_begin(c, sl, PK_HASH_SPEND, sl.d_tr, (uint32_t)st.stride(), (uint32_t)st.bytes(), ch.m))) return rc;
  sl.last_spend_lanes = ch.m;
  return ACT_OK;
}
// stage 2: finish the hash (host mode blocks on this slot only) and set the statuses
int spend_stage2(act_ctx* c, Slot& sl, SpendChunk& ch) {
  ACT_RANGE("spend.phaseB off=%zu m=%zu (hash_end, then k_spend_finish)", ch.off, (size_t)ch.m);
  const SpendTranscript st{c->L};
  int rc;
  if ((rc = hash_end(c, sl, (uint32_t)st.stride(), (uint32_t)st.bytes(), ch.m))) return rc;
  return prof_launch(c, sl, PK_SPEND_FINISH, ch.m, [&] { launch_spend_finish(ch.a, sl.stream); });
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
  return from_uniform_on_device(device, uni, 3, out_h);         // src/lib.rs:353
}
int act_params_random(int device, const uint8_t rng[192], uint8_t out_h[96]) {
  if (!rng || !out_h) return ACT_ERR_ARG;
  return from_uniform_on_device(device, rng, 3, out_h);
}

int act_ctx_create(const uint8_t h[96], int L, int device, size_t max_batch, act_ctx** out) {
  if (!h || !out || L < 1 || L > 128) return ACT_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ACT_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return ACT_ERR_ARG;
  act_ctx* c = new act_ctx();
  *out = c;   // returned even on failure so that act_last_error() can be read; the caller destroys it
  // lanes per launch = max_batch * L must stay below 2^31 (kernels index lanes with 32-bit integers)
  if (max_batch > ((size_t)1 << 22)) { c->err = "max_batch above 2^22: lanes per launch (max_batch * L) must stay below 2^31"; return ACT_ERR_ARG; }
  c->device = device; c->L = L; c->max_batch = max_batch ? max_batch : 65536;     // throughput saturates from here on
  memcpy(c->henc, h, 96);
  HIPCK(c, hipSetDevice(device));
  int rc = workspace_alloc(c); if (rc) return rc;
  if ((rc = streams_settle(c))) return rc;
  hipStream_t s0 = c->slots[0].stream;
  // decode g, h1, h2, h3 and build their fixed-base tables
  uint8_t enc[128]; memcpy(enc, kGeneratorEnc, 32); memcpy(enc + 32, h, 96);
  uint8_t* d_enc = nullptr; uint32_t *d_ext = nullptr, *d_ok = nullptr;
  HIPCK(c, hipMalloc(&d_enc, 128)); HIPCK(c, hipMalloc(&d_ext, 4 * GE_WORDS * 4)); HIPCK(c, hipMalloc(&d_ok, 16));
  HIPCK(c, hipMemcpyAsync(d_enc, enc, 128, hipMemcpyHostToDevice, s0));
  launch_decode_points(d_enc, 4, d_ext, d_ok, s0);
  uint32_t ok[4];
  HIPCK(c, hipMemcpyAsync(ok, d_ok, 16, hipMemcpyDeviceToHost, s0));
  HIPCK(c, hipStreamSynchronize(s0));
  if (!(ok[0] && ok[1] && ok[2] && ok[3])) { c->err = "h1/h2/h3 is not a canonical Ristretto encoding"; return ACT_ERR_PARAMS; }
  // Window widths: 16 bits for every base (128 MiB each) -- ONE fixed footprint, whatever the device has free.  Wider windows are the
  // caller's decision: act_ctx_set_fixed_base_bits (below; bench.py asks for 24 bits on h1 and h3, the range kernel's bases: +47 GB,
  // +3 % verifies/s).  Tables of one width on one device are shared by all contexts of the process.
  for (int b = 0; b < 4; b++) c->fb_bits[b] = FB_WBITS;
  for (int b = 0; b < 4; b++) {
    c->d_tables[b] = table_acquire(device, enc + 32 * b, c->fb_bits[b], d_ext + b * GE_WORDS, s0, nullptr);
    if (!c->d_tables[b]) { c->err = "fixed-base table allocation failed"; return ACT_ERR_HIP; }
    c->P.tab[b] = FbTab{c->d_tables[b], (uint32_t)c->fb_bits[b], (uint32_t)b};
  }
  // the small scanned tables (64 KiB per base): the ISSUER's secrets (signing nonces, x) use them in every build, the client's
  // (prover, request) in the ct build (msm.h "secret scalars")
  HIPCK(c, hipMalloc(&c->d_tables_ct, (size_t)4 * CT_TABLE_WORDS * 4));
  for (int b = 0; b < 4; b++) {
    launch_build_table_ct(d_ext + b * GE_WORDS, c->d_tables_ct + (size_t)b * CT_TABLE_WORDS, s0);
    c->P.tab_ct[b] = c->d_tables_ct + (size_t)b * CT_TABLE_WORDS;
  }
  HIPCK(c, hipMalloc(&c->d_tables_mf, 4 * MF_TABLE_BYTES));
  for (int b = 0; b < 4; b++) {
    launch_build_table_mf(d_ext + b * GE_WORDS, c->d_tables_mf + (size_t)b * MF_TABLE_BYTES, s0);
    c->P.tab_mf[b] = c->d_tables_mf + (size_t)b * MF_TABLE_BYTES;
  }
  HIPCK(c, hipMalloc(&c->d_half_h1, (size_t)2 * NIELS_WORDS * 4));
  launch_half_point_table(c->P.tab[BASE_H1], c->d_half_h1, s0);
  c->P.half_h1 = c->d_half_h1;
  HIPCK(c, hipStreamSynchronize(s0));
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
  return ACT_OK;
}

void act_ctx_destroy(act_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (Slot& sl : c->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    // d_state held the signing nonces (e, alpha) / the prover's r3, r*; d_d01 the prover's k* h2 terms: wipe before freeing
    if (sl.d_state) (void)hipMemset(sl.d_state, 0, c->max_batch * 24 * 4);
    if (sl.d_d01) (void)hipMemset(sl.d_d01, 0, c->max_batch * 3 * GE_WORDS * 4);
    void* ptrs[] = {sl.d_buckets, sl.d_tr, sl.d_trs, sl.d_status, sl.d_coords, sl.d_d01, sl.d_xa, sl.d_flags, sl.d_xof, sl.d_state, sl.d_slot, sl.d_naf, sl.d_dig};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < Slot::N_STAGE; i++) if (sl.d_stage[i]) { (void)hipMemset(sl.d_stage[i], 0, sl.d_stage_cap[i]); (void)hipFree(sl.d_stage[i]); }   // staging may hold secrets
    if (sl.h_tr) (void)hipHostFree(sl.h_tr);
    if (sl.h_xof) (void)hipHostFree(sl.h_xof);
    if (sl.h_cst) (void)hipHostFree(sl.h_cst);
    for (hipEvent_t& e : sl.h_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (sl.bits_ev) (void)hipEventDestroy(sl.bits_ev);
    if (sl.bits_sig) (void)hipFree(sl.bits_sig);
    if (sl.cp_in_ev) (void)hipEventDestroy(sl.cp_in_ev);
    if (sl.cp_out_ev) (void)hipEventDestroy(sl.cp_out_ev);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  if (c->prof_base) (void)hipEventDestroy(c->prof_base);
  for (hipStream_t& a : c->aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); a = nullptr; }
  for (hipEvent_t e : c->sm_ev) (void)hipEventDestroy(e);
  if (c->d_small) { (void)hipMemset(c->d_small, 0, c->d_small_cap); (void)hipFree(c->d_small); }
  if (c->d_tiny) { (void)hipMemset(c->d_tiny, 0, TINY_BYTES); (void)hipFree(c->d_tiny); }
  if (c->d_group_ctr) (void)hipFree(c->d_group_ctr);
  if (c->d_tiny_tr) (void)hipFree(c->d_tiny_tr);
  if (c->h_tiny) { memset(c->h_tiny, 0, TINY_BYTES); (void)hipHostFree(c->h_tiny); }
  for (uint32_t* t : c->d_tables) table_release(c->device, t);
  if (c->d_half_h1) (void)hipFree(c->d_half_h1);
  if (c->d_tables_ct) (void)hipFree(c->d_tables_ct);
  if (c->d_tables_mf) (void)hipFree(c->d_tables_mf);
  if (c->d_wire_flags) (void)hipFree(c->d_wire_flags);
  memset(&c->key, 0, sizeof(c->key)); memset(c->sk_cached, 0, 64);
  delete c;
}
int act_ctx_set_transcript_mode(act_ctx* c, int mode) {
  if (!c || (mode != ACT_TRANSCRIPT_HOST && mode != ACT_TRANSCRIPT_DEVICE)) return ACT_ERR_ARG;
  c->tr_mode = mode; return ACT_OK;
}
int act_ctx_streams_overlap(const act_ctx* c) { return c ? c->streams_overlap : -1; }
int act_ctx_fixed_base_bits(const act_ctx* c, int base) { return (c && base >= 0 && base < 4) ? c->fb_bits[base] : 0; }
// The table of base `base` at another window width, built (or shared with the process's other contexts on the device) now; the
// old one is released.  Refused -- width unchanged -- when the table is not there yet and the device has not its size + 16 GB free.
int act_ctx_set_fixed_base_bits(act_ctx* c, int base, int bits) {
  if (!c || base < 0 || base > 3 || bits < 4 || bits > 24) return ACT_ERR_ARG;
  Call call(c, 0);
  HIPCK(c, hipSetDevice(c->device));
  if (bits == c->fb_bits[base]) return call.finish();
  uint8_t enc[32];
  memcpy(enc, base == 0 ? kGeneratorEnc : c->henc + 32 * (base - 1), 32);
  if (!table_cached(c->device, enc, bits)) {
    size_t free_b = 0, total_b = 0;
    const size_t need = fb_table_words((uint32_t)bits) * 4 + ((size_t)16 << 30);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need) {
      (void)hipGetLastError();
      c->err = "act_ctx_set_fixed_base_bits: not enough free device memory for a table of that width (its size + 16 GB)";
      return ACT_ERR_HIP;
    }
  }
  hipStream_t s0 = c->slots[0].stream;
  uint8_t* d_enc = nullptr; uint32_t *d_ext = nullptr, *d_ok = nullptr;
  HIPCK(c, hipMalloc(&d_enc, 32)); HIPCK(c, hipMalloc(&d_ext, GE_WORDS * 4)); HIPCK(c, hipMalloc(&d_ok, 4));
  HIPCK(c, hipMemcpyAsync(d_enc, enc, 32, hipMemcpyHostToDevice, s0));
  launch_decode_points(d_enc, 1, d_ext, d_ok, s0);
  uint32_t* t = table_acquire(c->device, enc, bits, d_ext, s0, nullptr);
  const hipError_t se = hipStreamSynchronize(s0);
  (void)hipFree(d_enc); (void)hipFree(d_ext); (void)hipFree(d_ok);
  if (!t || se != hipSuccess) {
    (void)hipGetLastError();
    if (t) table_release(c->device, t);
    c->err = "act_ctx_set_fixed_base_bits: table allocation failed";
    return ACT_ERR_HIP;
  }
  table_release(c->device, c->d_tables[base]);
  c->d_tables[base] = t; c->fb_bits[base] = bits;
  c->P.tab[base] = FbTab{t, (uint32_t)bits, (uint32_t)base};
  return call.finish();
}
int act_build_has_ct_secret_tables(void) {
#if defined(ACT_CT_SECRET_TABLES)
  return 1;
#else
  return 0;
#endif
}
int act_ctx_set_small_batch_max(act_ctx* c, size_t n) { if (!c) return ACT_ERR_ARG; c->small_max.store(n); c->small_max_set.store(true); return ACT_OK; }
int act_ctx_set_tiny_calls(act_ctx* c, int on) { if (!c) return ACT_ERR_ARG; c->tiny_on.store(on != 0); return ACT_OK; }
int act_debug_set_slowdown(act_ctx* c, uint32_t ns_per_lane) { if (!c) return ACT_ERR_ARG; c->debug_ns_per_lane.store(ns_per_lane); return ACT_OK; }
int act_ctx_set_pipeline_depth(act_ctx* c, int depth) { if (!c || depth < 1 || depth > 2) return ACT_ERR_ARG; c->depth = depth; return ACT_OK; }
int act_ctx_set_host_threads(act_ctx* c, int n) { if (!c || n < 0) return ACT_ERR_ARG; c->host_threads = n; return ACT_OK; }
int act_ctx_host_hash_stats(act_ctx* c, double* wait_s, double* hash_s, uint64_t* bytes, int reset) {
  if (!c) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> lk(c->mu);
  if (wait_s) *wait_s = c->host_wait_s;
  if (hash_s) *hash_s = c->host_hash_s;
  if (bytes) *bytes = c->host_hash_bytes;
  if (reset) { c->host_wait_s = c->host_hash_s = 0; c->host_hash_bytes = 0; }
  return ACT_OK;
}
// copied (under the text's own lock, not the context's: a running batch call does not delay it) into a buffer of the calling thread --
// another thread's failing call may rewrite c->err at any moment; valid until this thread's next act_last_error call
const char* act_last_error(const act_ctx* c) {
  if (!c) return "null context";
  thread_local std::string mine;
  mine = c->err.get();
  return mine.c_str();
}
size_t act_spend_proof_bytes(const act_ctx* c) { return ProofLayout{c->L}.bytes(); }
size_t act_prove_rng_bytes(const act_ctx* c) { return 64u * (4u * (size_t)c->L + 12u); }
size_t act_spend_transcript_bytes(const act_ctx* c) { return SpendTranscript{c->L}.bytes(); }

int act_private_key_random(act_ctx* c, const uint8_t rng[64], uint8_t out_sk[64]) {
  if (!c || !rng || !out_sk) return ACT_ERR_ARG;
  Call call(c, 0);
  HIPCK(c, hipSetDevice(c->device));
  Slot& sl = c->slots[0];
  int rc = stage_reserve(c, sl, 0, 128); if (rc) return rc;
  HIPCK(c, hipMemcpyAsync(sl.d_stage[0], rng, 64, hipMemcpyHostToDevice, sl.stream));
  launch_keygen(c->P, sl.d_stage[0], 1, sl.d_stage[0] + 64, sl.stream);
  HIPCK(c, hipMemcpyAsync(out_sk, sl.d_stage[0] + 64, 64, hipMemcpyDeviceToHost, sl.stream));
  sl.d_stage_dirty[0] = std::max<size_t>(sl.d_stage_dirty[0], 128);
  return call.finish();
}
int act_pre_issuance_random_batch(act_ctx* c, size_t n, int mem, const uint8_t* rng, uint8_t* out_pre) {
  if (!c || (n && (!rng || !out_pre))) return ACT_ERR_ARG;
  Call call(c, 0);
  HIPCK(c, hipSetDevice(c->device));
  Slot& sl = c->slots[0];
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    const uint8_t* d_rng; uint8_t* d_out; int rc;
    if ((rc = dev_in(c, sl, 0, mem, rng + off * 128, (size_t)m * 128, &d_rng))) return rc;
    if ((rc = dev_out_begin(c, sl, 1, mem, out_pre + off * 64, (size_t)m * 64, &d_out))) return rc;
    launch_pre_issuance_random(d_rng, m, d_out, sl.stream);
    if ((rc = dev_out_end(c, sl, mem, out_pre + off * 64, d_out, (size_t)m * 64))) return rc;
    HIPCK(c, hipStreamSynchronize(sl.stream));
  }
  return call.finish();
}

// Tiny calls.  The crate's entry points take ONE item (src/lib.rs:463, 528, 621, 781, 972, 1217); for the cheap ones the GPU's answer
// time is fixed cost -- three launches, copies from pageable memory, the wipe -- as much as arithmetic.  Calls of at most TINY_MAX
// lanes gather their inputs in a pinned buffer, cross PCIe once each way, and run ONE kernel that hashes its (single-chunk) transcript
// itself -- the device BLAKE3 of blake3_hd.h whatever the context's transcript mode: same bytes -- and zeroes its staged inputs when
// done.  ACT_NO_FUSED_TINY=1 keeps the two-phase path (A/B, and the tests compare the two).
static int tiny_buffers(act_ctx* c) {
  c->tiny_dirty = true;                 // whoever asks for the buffers is about to stage inputs in them
  if (c->d_tiny) return ACT_OK;
  HIPCK(c, hipMalloc(&c->d_tiny, TINY_BYTES));
  HIPCK(c, hipHostMalloc(&c->h_tiny, TINY_BYTES, hipHostMallocDefault));
  HIPCK(c, hipMemsetAsync(c->d_tiny, 0, TINY_BYTES, c->slots[0].stream));
  HIPCK(c, hipStreamSynchronize(c->slots[0].stream));
  memset(c->h_tiny, 0, TINY_BYTES);
  return ACT_OK;
}
static void wipe_host(uint8_t* p, size_t n) { volatile uint8_t* q = p; for (size_t i = 0; i < n; i++) q[i] = 0; }

static int request_tiny(act_ctx* c, size_t n, int mem, const uint8_t* pre, const uint8_t* rng, uint8_t* out_req) {
  Slot& sl = c->slots[0];
  RequestArgs a{}; a.P = c->P; a.n = (uint32_t)n;
  int rc;
  if (mem == ACT_MEM_DEVICE) {
    a.pre = pre; a.rng = rng; a.out = out_req;
    if ((rc = prof_launch(c, sl, PK_REQUEST_A, n, [&] { launch_request_fused(a, sl.stream); }))) return rc;
    HIPCK(c, hipStreamSynchronize(sl.stream));
    return prof_collect(c, sl);
  }
  if ((rc = tiny_buffers(c))) return rc;
  memcpy(c->h_tiny, pre, 64 * n); memcpy(c->h_tiny + 64 * n, rng, 128 * n);
  HIPCK(c, hipMemcpyAsync(c->d_tiny, c->h_tiny, 192 * n, hipMemcpyHostToDevice, sl.stream));
  a.pre = c->d_tiny; a.rng = c->d_tiny + 64 * n; a.out = c->d_tiny + TINY_OUT; a.wipe_inputs = 1;
  if ((rc = prof_launch(c, sl, PK_REQUEST_A, n, [&] { launch_request_fused(a, sl.stream); }))) return rc;
  HIPCK(c, hipMemcpyAsync(c->h_tiny + TINY_OUT, c->d_tiny + TINY_OUT, 128 * n, hipMemcpyDeviceToHost, sl.stream));
  HIPCK(c, hipStreamSynchronize(sl.stream));
  memcpy(out_req, c->h_tiny + TINY_OUT, 128 * n);
  wipe_host(c->h_tiny, 192 * n);
  return prof_collect(c, sl);
}

int act_request_batch(act_ctx* c, size_t n, int mem, const uint8_t* pre, const uint8_t* rng, uint8_t* out_req) {
  if (!c || (n && (!pre || !rng || !out_req))) return ACT_ERR_ARG;
  Call call(c, 0);
  HIPCK(c, hipSetDevice(c->device));
  if (n && n <= TINY_MAX && tiny_enabled(c)) { int rc = request_tiny(c, n, mem, pre, rng, out_req); return rc ? rc : call.finish(); }
  Slot& sl = c->slots[0];
  for (size_t off = 0; off < n; off += c->max_batch) {
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    sl.d_trs_dirty = std::max(sl.d_trs_dirty, (size_t)m);
    RequestArgs a{}; a.P = c->P; a.n = m; a.trs = sl.d_trs; a.xof = sl.d_xof; int rc;
    if ((rc = dev_in(c, sl, 0, mem, pre + off * 64, (size_t)m * 64, &a.pre))) return rc;
    if ((rc = dev_in(c, sl, 1, mem, rng + off * 128, (size_t)m * 128, &a.rng))) return rc;
    if ((rc = dev_out_begin(c, sl, 2, mem, out_req + off * 128, (size_t)m * 128, &a.out))) return rc;
    if ((rc = prof_launch(c, sl, PK_REQUEST_A, m, [&] { launch_request_a(a, sl.stream); }))) return rc;
    if ((rc = hash_step(c, sl, PK_HASH_SMALL, sl.d_trs, SMALL_TR_STRIDE, c->P.prefix_len[LABEL_REQUEST] + 80, m))) return rc;
    if ((rc = prof_launch(c, sl, PK_REQUEST_B, m, [&] { launch_request_b(a, sl.stream); }))) return rc;
    if ((rc = dev_out_end(c, sl, mem, out_req + off * 128, a.out, (size_t)m * 128))) return rc;
    if ((rc = sync_all(c))) return rc;
  }
  return call.finish();
}

// PrivateKey::issue over at most TINY_MAX lanes whose rng slices do not depend on each other's verdicts: the PoK check and the signature
// side by side in ONE kernel (k_sign.hip k_sign_fused<true>), one copy each way
static int issue_tiny(act_ctx* c, size_t n, int mem, const uint8_t* req, const uint8_t* camt, const uint8_t* rng, uint8_t* out_resp, uint8_t* status) {
  Slot& sl = c->slots[0];
  SignFusedArgs f{}; f.P = c->P; f.K = c->key; f.n = (uint32_t)n; f.label = LABEL_RESPOND; f.point_stride = 128;
  f.pbk = sl.d_buckets; f.trs = sl.d_trs; f.trs_req = c->d_tiny_tr; f.group_counter = group_counters(c, sl);
  sl.d_trs_dirty = std::max(sl.d_trs_dirty, n);       // signatures of rejected lanes land in their transcripts: finish_call wipes them
#if defined(ACT_TINY_TIMING)
  static unsigned long long* d_dbg = nullptr;
  if (!d_dbg) { HIPCK(c, hipMalloc(&d_dbg, 16 * 8 * 8)); }
  HIPCK(c, hipMemsetAsync(d_dbg, 0, 16 * 8 * 8, sl.stream));
  f.dbg = d_dbg;
  struct Dump {      // per role: its time stamps in microseconds from the earliest one
    act_ctx* c; Slot& sl;
    ~Dump() {
      unsigned long long h[128];
      if (hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
      unsigned long long t0 = ~0ull;
      for (int i = 0; i < 128; i++) if (h[i] && h[i] < t0) t0 = h[i];
      for (int r = 0; r < 10; r++) {
        fprintf(stderr, "[tiny timing] role %d:", r);
        for (int k = 0; k < 8; k++) fprintf(stderr, " %8.1f", h[r * 8 + k] ? (double)(h[r * 8 + k] - t0) / 100.0 : -1.0);
        fprintf(stderr, "  us\n");
      }
    }
  } dump{c, sl};
#endif
  int rc;
  if (mem == ACT_MEM_DEVICE) {
    f.point = req; f.c_amount = camt; f.rng = rng; f.out = out_resp; f.status = status;
    if ((rc = prof_launch(c, sl, PK_SIGN_A, n, [&] { launch_sign_fused(f, true, sl.stream); }))) return rc;
    HIPCK(c, hipStreamSynchronize(sl.stream));
    return prof_collect(c, sl);
  }
  if ((rc = tiny_buffers(c))) return rc;
  uint8_t* h = c->h_tiny;
  memcpy(h, req, 128 * n); memcpy(h + 128 * n, camt, 32 * n); memcpy(h + 160 * n, rng, 128 * n);
  HIPCK(c, hipMemcpyAsync(c->d_tiny, h, 288 * n, hipMemcpyHostToDevice, sl.stream));
  f.point = c->d_tiny; f.c_amount = c->d_tiny + 128 * n; f.rng = c->d_tiny + 160 * n; f.wipe_rng = 1;
  f.out = c->d_tiny + TINY_OUT; f.status = c->d_tiny + TINY_OUT + 160 * n;
  if ((rc = prof_launch(c, sl, PK_SIGN_A, n, [&] { launch_sign_fused(f, true, sl.stream); }))) return rc;
  HIPCK(c, hipMemcpyAsync(h + TINY_OUT, c->d_tiny + TINY_OUT, 161 * n, hipMemcpyDeviceToHost, sl.stream));
  HIPCK(c, hipStreamSynchronize(sl.stream));
  memcpy(out_resp, h + TINY_OUT, 160 * n); memcpy(status, h + TINY_OUT + 160 * n, n);
  wipe_host(h, 288 * n);                                        // (the device copies were zeroed by the kernel)
  return prof_collect(c, sl);
}

static int issue_batch_impl(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* req, const uint8_t* camt, const uint8_t* rng,
                            int rng_mode, uint8_t* out_resp, uint8_t* status);
int act_issue_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* req, const uint8_t* camt, const uint8_t* rng,
                    int rng_mode, uint8_t* out_resp, uint8_t* status) {
  if (!c || !sk || (n && (!req || !camt || !rng || !out_resp || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  return issue_batch_impl(c, n, mem, sk, req, camt, rng, rng_mode, out_resp, status);
}
static int issue_batch_impl(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* req, const uint8_t* camt, const uint8_t* rng,
                            int rng_mode, uint8_t* out_resp, uint8_t* status) {
  Call call(c, n);
  HIPCK(c, hipSetDevice(c->device));
  int rc = set_key(c, sk); if (rc) return rc;
  if (n && n <= TINY_MAX && n <= c->max_batch && (rng_mode == ACT_RNG_PER_LANE || n == 1) && tiny_enabled(c)) {
    rc = issue_tiny(c, n, mem, req, camt, rng, out_resp, status);
    return rc ? rc : call.finish();
  }
  size_t cursor = 0, chunk = 0;
  // chunks alternate between the two slots; with device transcripts and per-lane rng nothing in a chunk waits for the
  // host, so two chunks are in flight (the per-proof kernels put only one wavefront per SIMD on the GPU per chunk)
  for (size_t off = 0; off < n; off += c->max_batch, chunk++) {
    Slot& sl = c->slots[chunk % c->depth];
    if (chunk >= (size_t)c->depth) { HIPCK(c, hipStreamSynchronize(sl.stream)); if ((rc = prof_collect(c, sl))) return rc; }
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    sl.d_trs_dirty = std::max(sl.d_trs_dirty, (size_t)m);
    IssueArgs a{}; a.P = c->P; a.n = m; a.trs = sl.d_trs; a.xa = sl.d_xa; a.flags = sl.d_flags; a.xof = sl.d_xof; a.status = sl.d_status; a.pbk = sl.d_buckets;
    if ((rc = dev_in(c, sl, 0, mem, req + off * 128, (size_t)m * 128, &a.req))) return rc;
    if ((rc = dev_in(c, sl, 1, mem, camt + off * 32, (size_t)m * 32, &a.c_amount))) return rc;
    uint8_t* d_out;
    if ((rc = dev_out_begin(c, sl, 2, mem, out_resp + off * 160, (size_t)m * 160, &d_out))) return rc;
    if ((rc = prof_launch(c, sl, PK_ISSUE_A, m, [&] { launch_issue_a(a, sl.stream); }))) return rc;
    if ((rc = hash_step(c, sl, PK_HASH_SMALL, sl.d_trs, SMALL_TR_STRIDE, c->P.prefix_len[LABEL_REQUEST] + 80, m))) return rc;
    if ((rc = prof_launch(c, sl, PK_ISSUE_CHECK, m, [&] { launch_issue_check(a, sl.stream); }))) return rc;
    const uint8_t* d_rng;
    if ((rc = prepare_rng_slots(c, sl, m, off, mem, rng, rng_mode, &cursor, &d_rng))) return rc;
    if ((rc = sign_phase(c, sl, m, LABEL_RESPOND, d_rng, a.c_amount, d_out))) return rc;
    if ((rc = dev_out_end(c, sl, mem, out_resp + off * 160, d_out, (size_t)m * 160))) return rc;
    if ((rc = copy_status_out(c, sl, mem, status + off, m))) return rc;
  }
  return call.finish();
}

// The two halves of issue / refund as separate calls, for callers that must see every verdict before any rng is assigned:
// the node dispatcher (node.cpp) makes ACT_RNG_SEQUENTIAL exact across the GPUs of a node by checking on all shards,
// counting the accepted lanes of the shards in front, and only then signing.

static int tiny_buffers(act_ctx* c);
static void wipe_host(uint8_t* p, size_t n);
static int issue_check_impl(act_ctx* c, size_t n, int mem, const uint8_t* req, uint8_t* status) {
  Call call(c, n);
  HIPCK(c, hipSetDevice(c->device));
  int rc; size_t chunk = 0;
  if (n && n <= TINY_MAX && n <= c->max_batch && tiny_enabled(c)) {
    // the PoK check of a tiny call as ONE kernel (k_sign_fused's check role on its own): K1, the 266-byte transcript, its BLAKE3, the verdict
    Slot& sl = c->slots[0];
    sl.d_trs_dirty = std::max(sl.d_trs_dirty, n);
    SignFusedArgs f{}; f.P = c->P; f.n = (uint32_t)n; f.label = LABEL_RESPOND; f.point_stride = 128; f.check_only = 1;
    f.pbk = sl.d_buckets; f.trs = sl.d_trs; f.trs_req = c->d_tiny_tr; f.group_counter = group_counters(c, sl);
    if (mem == ACT_MEM_DEVICE) { f.point = req; f.status = status; }
    else {
      if ((rc = tiny_buffers(c))) return rc;
      memcpy(c->h_tiny, req, 128 * n);
      HIPCK(c, hipMemcpyAsync(c->d_tiny, c->h_tiny, 128 * n, hipMemcpyHostToDevice, sl.stream));
      f.point = c->d_tiny; f.status = c->d_tiny + TINY_OUT; f.wipe_rng = 1;
    }
    if ((rc = prof_launch(c, sl, PK_ISSUE_A, n, [&] { launch_sign_fused(f, true, sl.stream); }))) return rc;
    if (mem == ACT_MEM_HOST) {
      HIPCK(c, hipMemcpyAsync(c->h_tiny + TINY_OUT, c->d_tiny + TINY_OUT, n, hipMemcpyDeviceToHost, sl.stream));
      HIPCK(c, hipStreamSynchronize(sl.stream));
      memcpy(status, c->h_tiny + TINY_OUT, n);
      wipe_host(c->h_tiny, 128 * n);
    }
    return call.finish();
  }
  for (size_t off = 0; off < n; off += c->max_batch, chunk++) {
    Slot& sl = c->slots[chunk % c->depth];
    if (chunk >= (size_t)c->depth) { HIPCK(c, hipStreamSynchronize(sl.stream)); if ((rc = prof_collect(c, sl))) return rc; }
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    sl.d_trs_dirty = std::max(sl.d_trs_dirty, (size_t)m);
    IssueArgs a{}; a.P = c->P; a.n = m; a.trs = sl.d_trs; a.xa = sl.d_xa; a.flags = sl.d_flags; a.xof = sl.d_xof; a.status = sl.d_status; a.pbk = sl.d_buckets;
    if ((rc = dev_in(c, sl, 0, mem, req + off * 128, (size_t)m * 128, &a.req))) return rc;
    if ((rc = prof_launch(c, sl, PK_ISSUE_A, m, [&] { launch_issue_a(a, sl.stream); }))) return rc;
    if ((rc = hash_step(c, sl, PK_HASH_SMALL, sl.d_trs, SMALL_TR_STRIDE, c->P.prefix_len[LABEL_REQUEST] + 80, m))) return rc;
    if ((rc = prof_launch(c, sl, PK_ISSUE_CHECK, m, [&] { launch_issue_check(a, sl.stream); }))) return rc;
    if ((rc = copy_status_out(c, sl, mem, status + off, m))) return rc;
  }
  return call.finish();
}
int act_issue_check_batch(act_ctx* c, size_t n, int mem, const uint8_t* req, uint8_t* status) {
  if (!c || (n && (!req || !status))) return ACT_ERR_ARG;
  return issue_check_impl(c, n, mem, req, status);
}
// signs the lanes whose status_in is 0; `point` = IssuanceRequest records (label RESPOND, with amounts) or enc(K') (label REFUND)
static int sign_only_batch(act_ctx* c, size_t n, int mem, int label, const uint8_t sk[64], const uint8_t* point, size_t point_stride,
                           const uint8_t* camt, const uint8_t* status_in, const uint8_t* rng, int rng_mode, uint8_t* out, uint8_t* status) {
  Call call(c, n);
  HIPCK(c, hipSetDevice(c->device));
  int rc = set_key(c, sk); if (rc) return rc;
  const size_t rec = label == LABEL_RESPOND ? 160 : 128;
  size_t cursor = 0, chunk = 0;
  for (size_t off = 0; off < n; off += c->max_batch, chunk++) {
    Slot& sl = c->slots[chunk % c->depth];
    if (chunk >= (size_t)c->depth) { HIPCK(c, hipStreamSynchronize(sl.stream)); if ((rc = prof_collect(c, sl))) return rc; }
    uint32_t m = (uint32_t)std::min(c->max_batch, n - off);
    SignXaArgs x{}; x.P = c->P; x.n = m; x.point_stride = (uint32_t)point_stride; x.xa = sl.d_xa; x.status = sl.d_status;
    if ((rc = dev_in(c, sl, 0, mem, point + off * point_stride, (size_t)m * point_stride, &x.point))) return rc;
    if (camt && (rc = dev_in(c, sl, 1, mem, camt + off * 32, (size_t)m * 32, &x.c_amount))) return rc;
    HIPCK(c, hipMemcpyAsync(sl.d_status, status_in + off, m, mem == ACT_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, sl.stream));
    uint8_t* d_out;
    if ((rc = dev_out_begin(c, sl, 2, mem, out + off * rec, (size_t)m * rec, &d_out))) return rc;
    launch_sign_xa(x, sl.stream);
    const uint8_t* d_rng;
    if ((rc = prepare_rng_slots(c, sl, m, off, mem, rng, rng_mode, &cursor, &d_rng))) return rc;
    if ((rc = sign_phase(c, sl, m, label, d_rng, x.c_amount, d_out))) return rc;
    if ((rc = dev_out_end(c, sl, mem, out + off * rec, d_out, (size_t)m * rec))) return rc;
    if ((rc = copy_status_out(c, sl, mem, status + off, m))) return rc;
  }
  return call.finish();
}
int act_issue_sign_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* req, const uint8_t* camt, const uint8_t* status_in,
                         const uint8_t* rng, int rng_mode, uint8_t* out_resp, uint8_t* status) {
  if (!c || !sk || (n && (!req || !camt || !status_in || !rng || !out_resp || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  return sign_only_batch(c, n, mem, LABEL_RESPOND, sk, req, 128, camt, status_in, rng, rng_mode, out_resp, status);
}
int act_refund_sign_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* kprime, const uint8_t* status_in,
                          const uint8_t* rng, int rng_mode, uint8_t* out_refund, uint8_t* status) {
  if (!c || !sk || (n && (!kprime || !status_in || !rng || !out_refund || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  return sign_only_batch(c, n, mem, LABEL_REFUND, sk, kprime, 32, nullptr, status_in, rng, rng_mode, out_refund, status);
}

// Proofs that arrive as CBOR wire bytes (act_verify_spend_cbor_batch, cbor_impl.inc): every chunk's messages are copied /
// read where they are and unframed into raw records by a kernel on the chunk's own stream, in front of k_spend_prep.
struct WireSrc { const uint8_t* cbor; const uint64_t* offsets; size_t msg_len; uint8_t* out_nullifier; };      // out_nullifier (nullable): the caller's n*32 array for the `k` fields
static int wire_unframe_chunk(act_ctx* c, Slot& sl, const WireSrc& w, int mem, size_t off, uint32_t m, const uint8_t** d_records);     // cbor_impl.inc
static int copy_chain_wait(act_ctx* c, Slot& sl, bool out);        // (defined with client_batch below)
static int copy_chain_record(act_ctx* c, Slot& sl, bool out);

#include "small_impl.inc"      // the small-batch schedule: small_prepare, SmallGate, spend_small_locked

// verify (sign == false) or refund (sign == true), two-slot software pipeline: stage 1 of chunk i+1 is enqueued before
// the host touches chunk i again.  The caller holds the context (Call).
static int spend_batch_locked(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* proof, bool sign, const uint8_t* rng,
                              int rng_mode, uint8_t* out_refund, uint8_t* status, uint8_t* out_kprime, const WireSrc* wire = nullptr) {
  int rc = set_key(c, sk); if (rc) return rc;
  if (n && n <= c->small_max.load() * ((c->tr_mode == ACT_TRANSCRIPT_DEVICE && !c->small_max_set.load()) ? 2u : 1u) && n <= c->max_batch) {
    if (!wire) return spend_small_locked(c, n, mem, proof, sign, rng, rng_mode, out_refund, status, out_kprime);
    // wire bytes: all n messages are unframed on slot 0's stream, the small-batch schedule starts from those records
    const uint8_t* d_records = nullptr;
    if ((rc = wire_unframe_chunk(c, c->slots[0], *wire, mem, 0, (uint32_t)n, &d_records))) return rc;
    return spend_small_locked(c, n, mem, nullptr, sign, rng, rng_mode, out_refund, status, out_kprime, d_records);
  }
  const size_t pb = ProofLayout{c->L}.bytes();
  const uint8_t* proof_view = (mem == ACT_MEM_HOST && !wire) ? mapped_view(c, proof, n * pb) : nullptr;      // pinned: the kernels read the proofs in place
  const bool in_host = mem == ACT_MEM_HOST && !proof_view;       // the proofs travel through the staging buffers
  const size_t host_chunk_env = (size_t)tune(T_HOST_CHUNK);   // measurement knob (act_tuning_set)
  // host-transcript mode: a chunk's transcripts go to the host, are hashed there and come back before its status kernel, all
  // of which only overlaps with compute if there are other chunks to compute: long batches use full-size chunks (the
  // kernels' best size), short ones are cut finer so that there is something to pipeline
  // (measured per share of a 2^20 batch, profiles/r06_strong_share.txt: 2^19 and 2^18 proofs are fastest in full-size chunks -- 525 k
  // and 517 k/s against 501 k in 16 384s at 2^18 --, 2^17 in half-size ones -- 501 k against 493 k --, below that 16 384; a call
  // of at most 20 480 proofs goes in two halves of at least 8 192: 16 384 proofs 38.9 ms against 41.1 as one chunk, 12 288 30.4 against
  // 32.5; finer is worse -- two chunks in flight cannot cover a chunk's ~10 ms of per-proof kernels, copies and hashing with range
  // kernels of 4 ms each: profiles/r06_midsize_host_chunks.txt)
  const size_t halves = n <= 20480 ? std::max<size_t>(8192, (n / 2 + 1023) / 1024 * 1024) : (size_t)16384;
  const size_t host_chunk = host_chunk_env ? host_chunk_env
                                           : (n >= 4 * c->max_batch ? c->max_batch : n >= 2 * c->max_batch ? std::max<size_t>(c->max_batch / 2, 16384) : halves);
  const size_t chunk_len = c->tr_mode == ACT_TRANSCRIPT_HOST ? std::min<size_t>(c->max_batch, host_chunk) : c->max_batch;
  const int stagger_env = (int)tune(T_STAGGER);      // measurement knob: force on / off (-1 = decide)
  const bool stagger = stagger_env >= 0 ? stagger_env != 0 : (in_host || c->tr_mode == ACT_TRANSCRIPT_HOST);
  // Chunk schedule.  Full-size chunks are the kernels' best size, but whenever a call moves data over PCIe the pipeline has a
  // head (nothing computes until the first chunk's proofs have arrived and its per-proof kernel has run) and a tail (the
  // last chunk's transcripts travel to the host, are hashed and come back with nothing left to overlap): measured 44 ms and
  // ~60 ms of a 2.2 s call with 65 536-proof chunks.  Such calls therefore open and close with quarter- and half-size chunks.
  std::vector<std::pair<size_t, size_t>> sched;          // (offset, lanes)
  {
    const bool taper_off = tune(T_NO_TAPER) != 0;
    std::vector<size_t> head, tail;
    size_t left = n;
    if (stagger && !taper_off && chunk_len >= 4096 && n >= 4 * chunk_len) {      // (a tail halved further, down to a sixteenth: 2^17 proofs 486 k/s against 500 k, profiles/r06_ab_tail_taper.txt)
      head = {chunk_len / 4, chunk_len / 2}; tail = {chunk_len / 2, chunk_len / 4};
      left -= chunk_len / 4 * 2 + chunk_len / 2 * 2;
    }
    // A call that fits one or a few chunks and reads its proofs from host memory (device transcripts: nothing else crosses PCIe):
    // as one chunk its 16.8 KB per proof arrive before anything computes (19 ms of a 148 ms call over 65 536 proofs).  A proof
    // takes ~0.3 us to arrive and ~1.95 us to verify, so a first chunk of an eighth of the call hides the arrival of the rest.
    else if (c->tr_mode == ACT_TRANSCRIPT_DEVICE && in_host && !taper_off && n >= 4096 && n < 4 * chunk_len) {
      const size_t h = std::min(n, std::max<size_t>(2048, (n / 8 + 1023) / 1024 * 1024));
      if (h < n) { head = {h}; left -= h; }
    }
    size_t off = 0;
    for (size_t l : head) { sched.emplace_back(off, l); off += l; }
    while (left) { size_t l = std::min(chunk_len, left); sched.emplace_back(off, l); off += l; left -= l; }
    for (size_t l : tail) { sched.emplace_back(off, l); off += l; }
  }
  const size_t nchunks = sched.size();
  SpendChunk chunks[2];
  size_t cursor = 0;
  const size_t depth = (size_t)c->depth;
  c->last_bits_ev = nullptr; c->last_bits_sig = nullptr;
  for (Slot& sl : c->slots) if (sl.bits_sig) { *(volatile uint32_t*)sl.bits_sig = 0; sl.bits_wgs = 0; }      // nothing of an earlier call is in flight (sync_all)
  auto stage1 = [&](size_t i) -> int {
    Slot& sl = c->slots[i % depth]; SpendChunk& ch = chunks[i % depth];
    ch = SpendChunk{}; ch.off = sched[i].first; ch.m = (uint32_t)sched[i].second;
    ch.stagger = stagger;
    int r;
    if (wire) { if ((r = wire_unframe_chunk(c, sl, *wire, mem, ch.off, ch.m, &ch.d_proofs))) return r; }
    else if (proof_view) ch.d_proofs = proof_view + ch.off * pb;
    else if ((r = dev_in(c, sl, 0, mem, proof + ch.off * pb, (size_t)ch.m * pb, &ch.d_proofs))) return r;
    if (out_kprime && (r = dev_out_begin(c, sl, 2, mem, out_kprime + ch.off * 32, (size_t)ch.m * 32, &ch.d_kprime))) return r;
    if (sign && (r = dev_out_begin(c, sl, 4, mem, out_refund + ch.off * 128, (size_t)ch.m * 128, &ch.d_out))) return r;
    return spend_stage1(c, sl, ch);
  };
  auto stage2 = [&](size_t i) -> int {
    Slot& sl = c->slots[i % depth]; SpendChunk& ch = chunks[i % depth];
    int r;
    if ((r = spend_stage2(c, sl, ch))) return r;
    if (out_kprime && (r = dev_out_end(c, sl, mem, out_kprime + ch.off * 32, ch.d_kprime, (size_t)ch.m * 32))) return r;
    if (sign) {
      const uint8_t* d_rng;
      if ((r = prepare_rng_slots(c, sl, ch.m, ch.off, mem, rng, rng_mode, &cursor, &d_rng))) return r;
      if ((r = sign_phase(c, sl, ch.m, LABEL_REFUND, d_rng, nullptr, ch.d_out))) return r;
      if ((r = dev_out_end(c, sl, mem, out_refund + ch.off * 128, ch.d_out, (size_t)ch.m * 128))) return r;
    }
    c->last_spend_slot = (int)(i % depth);
    return copy_status_out(c, sl, mem, status + ch.off, ch.m);
  };
  for (size_t i = 0; i < nchunks; i++) {
    // slot reuse: the copies of the chunk that last used this slot must have completed before its staging buffers are overwritten
    if (i >= depth) { HIPCK(c, hipStreamSynchronize(c->slots[i % depth].stream)); }
    if ((rc = stage1(i))) return rc;
    if (i + 1 >= depth && (rc = stage2(i + 1 - depth))) return rc;
  }
  for (size_t i = nchunks >= depth ? nchunks - depth + 1 : 0; i < nchunks; i++) if ((rc = stage2(i))) return rc;
  return sync_all(c);
}

static int spend_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* proof, bool sign, const uint8_t* rng,
                       int rng_mode, uint8_t* out_refund, uint8_t* status, uint8_t* out_kprime) {
  Call call(c, n);
  HIPCK(c, hipSetDevice(c->device));
  int rc = spend_batch_locked(c, n, mem, sk, proof, sign, rng, rng_mode, out_refund, status, out_kprime);
  return rc ? rc : call.finish();
}

int act_verify_spend_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* proof, uint8_t* status, uint8_t* out_kprime) {
  if (!c || !sk || (n && (!proof || !status))) return ACT_ERR_ARG;
  return spend_batch(c, n, mem, sk, proof, false, nullptr, ACT_RNG_PER_LANE, nullptr, status, out_kprime);
}
int act_refund_batch(act_ctx* c, size_t n, int mem, const uint8_t sk[64], const uint8_t* proof, const uint8_t* rng, int rng_mode,
                     uint8_t* out_refund, uint8_t* status) {
  if (!c || !sk || (n && (!proof || !rng || !out_refund || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  return spend_batch(c, n, mem, sk, proof, true, rng, rng_mode, out_refund, status, nullptr);
}

#include "client_impl.inc"      // prove_spend and the two to_credit_token calls

#include "debug_impl.inc"       // act_debug_*, act_ubench_*, act_prof_*
}  // extern "C"

#include "cbor_impl.inc"
#include "nullifier_impl.inc"
