// node.cpp — the GPUs of one node behind one handle (include/act_mi355x.h act_node_*): a batch is cut into contiguous
// pieces, one host thread per GPU drives that GPU's context through the single-GPU entry points; outputs land in disjoint
// slices of the caller's arrays.  No collective, no peer traffic: lanes are independent.  The one
// cross-piece dependency is ACT_RNG_SEQUENTIAL for issue / refund: a lane's rng slice is the number of ACCEPTED lanes in front
// of it (/root/reference/src/lib.rs:638-643, 842-846 draw only after the proof verifies), so every lane is checked first, the
// host counts the accepted lanes in front of every piece, and only then every piece is signed from its own offset into the
// stream -- byte for byte what one loop over one generator produces, whichever GPU a piece went to.
// Plain host C++ over the C ABI: nothing here touches a device.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <sys/random.h>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/act_mi355x.h"
#include "rng_source.h"

struct act_node {
  std::vector<act_ctx*> ctx;
  std::vector<int> devices;
  int L = 0;
  std::string err;
  std::mutex mu;      // every act_node_*_batch call holds it: a handle shared between host threads is used by one at a time
  // Load balance (all under mu).  The GPUs of a node are not equally fast -- clocks differ by several percent between devices and
  // move with temperature -- and a call ends when its slowest GPU does.  Two mechanisms, both invisible in the output bytes:
  //   weights   every throughput-sized call measures what each context did with its head piece; the next call cuts the heads in
  //             proportion (weight = relative speed, mean 1, averaged over calls).  Costs nothing.
  //   tail      the last part of the batch is not assigned in advance: it is handed out in small pieces from a cursor to whichever
  //             context asks next, so a GPU that falls behind DURING a call is covered by the others.  Costs the per-call overhead
  //             of a few small calls, so its size follows the finish-time spread the heads actually show:
  //             a sixteenth of the batch while nothing is known, nothing once the heads finish together.
  std::vector<double> weight;
  double spread = -1;            // running average of (last - first head to finish) / mean head time; < 0 = nothing measured yet
  bool weighted = true;
  int tail_64ths = -1;           // act_node_set_balance: -1 = follow `spread`, 0 = no tail, k = k/64 of the batch
  double last_tail_fraction = 0;
  struct DevStats { uint64_t lanes = 0, calls = 0; double seconds = 0; };
  std::vector<DevStats> last;    // what each context did in the most recent cut call
};

// A node handle's last error is the CALLING THREAD's last failure on it where there is one (threads share the handle): every write
// also lands in a slot of the writing thread, which act_node_last_error prefers.
namespace {
struct NodeErrMine { const void* of = nullptr; std::string text; };
NodeErrMine& node_err_mine() { thread_local NodeErrMine t; return t; }
void set_node_err(act_node* nd, const std::string& text) { nd->err = text; NodeErrMine& t = node_err_mine(); t.of = nd; t.text = text; }      // caller holds nd->mu
void small_call_err(act_node* nd, act_ctx* c) { const std::string text = act_last_error(c); std::lock_guard<std::mutex> node_lock(nd->mu); set_node_err(nd, text); }
}  // namespace

namespace {

struct Shard { size_t off, m; };
struct Piece { size_t k, off, m; int rc; };       // lanes [off, off + m) ran (or were to run: k == SIZE_MAX) on context k with result rc
constexpr size_t kMinBalanceLanes = 4096;         // per device and piece: below this a call is latency, not throughput
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Plan { std::vector<Shard> head; size_t tail_off = 0, piece = 0; double tail_fraction = 0; };
Plan make_plan(const act_node* nd, size_t n) {
  const size_t N = nd->ctx.size();
  Plan p; p.head.resize(N); p.tail_off = n;
  const bool big = N > 1 && n >= N * 4 * kMinBalanceLanes;
  size_t tail = 0;
  if (big) {
    double f = nd->tail_64ths >= 0 ? nd->tail_64ths / 64.0 : (nd->spread < 0 ? 1.0 / 16 : std::min(0.125, 2 * nd->spread));
    if (f < 1.0 / 64) f = 0;
    tail = (size_t)((double)n * f);
    if (tail) {
      p.piece = std::max(kMinBalanceLanes, tail / (2 * N) / 1024 * 1024);
      tail = tail / p.piece * p.piece;
    }
    if (!tail) p.piece = 0;
  }
  const size_t body = n - tail;
  p.tail_off = body; p.tail_fraction = n ? (double)tail / (double)n : 0;
  bool equal = !big || !nd->weighted || nd->weight.size() != N;
  if (!equal) { equal = true; for (double w : nd->weight) if (w != nd->weight[0]) equal = false; }
  if (equal) { for (size_t k = 0; k < N; k++) { const size_t a = body * k / N, b = body * (k + 1) / N; p.head[k] = {a, b - a}; } return p; }
  double total = 0, cum = 0; for (double w : nd->weight) total += w;
  size_t a = 0;
  for (size_t k = 0; k < N; k++) {
    cum += nd->weight[k];
    const size_t b = k + 1 == N ? body : std::min(body, (size_t)((double)body * (cum / total)));
    p.head[k] = {a, b > a ? b - a : 0}; a = std::max(a, b);
  }
  return p;
}

// fn(k, off, m): lanes [off, off + m) on context k.  One thread per GPU: its head piece, then tail pieces while there are any.
// First failure wins and is reported with its device (the caller holds nd->mu: nd->err is only ever written under it).  A
// context that fails stops taking pieces; the others finish the tail.  `pieces` (optional) receives what ran where, and -- with
// k == SIZE_MAX -- whatever nobody ran because every context had failed.
template <class F>
int run(act_node* nd, size_t n, F fn, std::vector<Piece>* pieces = nullptr) {
  const size_t N = nd->ctx.size();
  const Plan p = make_plan(nd, n);
  std::atomic<size_t> cursor{p.tail_off};
  std::vector<std::vector<Piece>> done(N);
  std::vector<double> head_s(N, 0), all_s(N, 0);
  auto worker = [&](size_t k) {
    const double t0 = now_s();
    int rc = ACT_OK;
    if (p.head[k].m || n == 0) { rc = fn(k, p.head[k].off, p.head[k].m); done[k].push_back({k, p.head[k].off, p.head[k].m, rc}); }
    head_s[k] = now_s() - t0;
    while (!rc && p.piece) {
      const size_t off = cursor.fetch_add(p.piece);
      if (off >= n) break;
      const size_t m = std::min(p.piece, n - off);
      rc = fn(k, off, m);
      done[k].push_back({k, off, m, rc});
    }
    all_s[k] = now_s() - t0;
  };
  std::vector<std::thread> th;
  for (size_t k = 1; k < N; k++) th.emplace_back(worker, k);
  worker(0);
  for (auto& t : th) t.join();
  // telemetry + what the next call's cut learns from this one
  nd->last.assign(N, act_node::DevStats{});
  nd->last_tail_fraction = p.tail_fraction;
  bool measurable = N > 1 && n >= N * 4 * kMinBalanceLanes;
  for (size_t k = 0; k < N; k++) {
    for (const Piece& q : done[k]) { nd->last[k].lanes += q.m; nd->last[k].calls++; if (q.rc) measurable = false; }
    nd->last[k].seconds = all_s[k];
    if (p.head[k].m < kMinBalanceLanes || head_s[k] <= 0) measurable = false;
  }
  if (measurable) {
    if (nd->weight.size() != N) nd->weight.assign(N, 1.0);
    double mean_rate = 0, mean_t = 0, lo = head_s[0], hi = head_s[0];
    for (size_t k = 0; k < N; k++) { mean_rate += (double)p.head[k].m / head_s[k] / N; mean_t += head_s[k] / N; lo = std::min(lo, head_s[k]); hi = std::max(hi, head_s[k]); }
    double total = 0;
    for (size_t k = 0; k < N; k++) { nd->weight[k] = 0.5 * nd->weight[k] + 0.5 * ((double)p.head[k].m / head_s[k] / mean_rate); total += nd->weight[k]; }
    for (double& w : nd->weight) w *= (double)N / total;
    const double s = (hi - lo) / mean_t;
    nd->spread = nd->spread < 0 ? s : 0.5 * nd->spread + 0.5 * s;
  }
  int first_rc = ACT_OK;
  for (size_t k = 0; k < N && !first_rc; k++)
    for (const Piece& q : done[k])
      if (q.rc) { set_node_err(nd, "device " + std::to_string(nd->devices[k]) + ": " + act_last_error(nd->ctx[k])); first_rc = q.rc; break; }
  if (pieces) {
    pieces->clear();
    for (size_t k = 0; k < N; k++) pieces->insert(pieces->end(), done[k].begin(), done[k].end());
    if (p.piece) for (size_t off = cursor.load(); off < n; off += p.piece) pieces->push_back({(size_t)-1, off, std::min(p.piece, n - off), first_rc ? first_rc : ACT_ERR_HIP});
  }
  return first_rc;
}
inline const uint8_t* at(const uint8_t* p, size_t off, size_t rec) { return p ? p + off * rec : nullptr; }
inline uint8_t* at(uint8_t* p, size_t off, size_t rec) { return p ? p + off * rec : nullptr; }

// accepted lanes in front of any lane: counts per block of 4096 lanes, the rest counted on demand
struct AcceptedBefore {
  const uint8_t* status; std::vector<size_t> block;
  AcceptedBefore(const uint8_t* st, size_t n) : status(st), block(n / 4096 + 2, 0) {
    size_t acc = 0;
    for (size_t i = 0; i < n; i++) { if (i % 4096 == 0) block[i / 4096] = acc; acc += st[i] == 0; }
    for (size_t b = (n + 4095) / 4096; b < block.size(); b++) block[b] = acc;
  }
  size_t total() const { return block.back(); }
  size_t operator()(size_t off) const { size_t a = block[off / 4096]; for (size_t i = off / 4096 * 4096; i < off; i++) a += status[i] == 0; return a; }
};

}  // namespace

extern "C" {

int act_node_create(const uint8_t h[96], int L, const int* devices, int n_devices, size_t max_batch, act_node** out) {
  if (!h || !devices || n_devices < 1 || !out) return ACT_ERR_ARG;
  act_node* nd = new act_node();
  *out = nd;      // returned even on failure so that act_node_last_error() can be read; the caller destroys it
  nd->L = L;
  // One thread per entry: contexts on different GPUs are built at the same time (a throughput-sized context with wide tables spends
  // ~2 s constructing them); entries that name the same GPU share that GPU's tables (engine.hip table cache), so the second one
  // waits for the first one's tables instead of building its own.
  std::vector<act_ctx*> made(n_devices, nullptr);
  std::vector<int> rcs(n_devices, ACT_OK);
  std::vector<std::thread> th;
  for (int k = 1; k < n_devices; k++) th.emplace_back([&, k] { rcs[k] = act_ctx_create(h, L, devices[k], max_batch, &made[k]); });
  rcs[0] = act_ctx_create(h, L, devices[0], max_batch, &made[0]);
  for (auto& t : th) t.join();
  int bad = -1;
  for (int k = 0; k < n_devices; k++) if (rcs[k] && bad < 0) bad = k;
  if (bad >= 0) {
    nd->err = "device " + std::to_string(devices[bad]) + ": " + (made[bad] ? act_last_error(made[bad]) : "context creation failed");
    for (act_ctx* c : made) if (c) act_ctx_destroy(c);
    return rcs[bad];
  }
  for (int k = 0; k < n_devices; k++) { nd->ctx.push_back(made[k]); nd->devices.push_back(devices[k]); }
  nd->weight.assign(n_devices, 1.0); nd->last.assign(n_devices, act_node::DevStats{});
  return ACT_OK;
}
void act_node_destroy(act_node* nd) {
  if (!nd) return;
  for (act_ctx* c : nd->ctx) act_ctx_destroy(c);
  delete nd;
}
int act_node_device_count(const act_node* nd) { return nd ? (int)nd->ctx.size() : 0; }
act_ctx* act_node_ctx(act_node* nd, int k) { return (nd && k >= 0 && k < (int)nd->ctx.size()) ? nd->ctx[k] : nullptr; }
// the text is copied under the handle's lock into a buffer of the calling thread: another thread's failing call can rewrite
// nd->err at any moment, a pointer into it could dangle.  Valid until this thread's next act_node_last_error call.
const char* act_node_last_error(const act_node* nd) {
  if (!nd) return "null node";
  thread_local std::string mine;
  const NodeErrMine& t = node_err_mine();
  if (t.of == nd && !t.text.empty()) { mine = t.text; return mine.c_str(); }      // this thread's own last failure on this handle
  { std::lock_guard<std::mutex> lk(const_cast<act_node*>(nd)->mu); mine = nd->err; }
  return mine.c_str();
}
int act_node_set_transcript_mode(act_node* nd, int mode) {
  if (!nd) return ACT_ERR_ARG;
  for (act_ctx* c : nd->ctx) { int rc = act_ctx_set_transcript_mode(c, mode); if (rc) return rc; }
  return ACT_OK;
}
int act_node_set_host_threads(act_node* nd, int per_gpu) {
  if (!nd) return ACT_ERR_ARG;
  for (act_ctx* c : nd->ctx) { int rc = act_ctx_set_host_threads(c, per_gpu); if (rc) return rc; }
  return ACT_OK;
}
int act_node_set_fixed_base_bits(act_node* nd, int base, int bits) {
  if (!nd) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  // one thread per context, like act_node_create: a 24-bit table takes ~1 s to build and each GPU builds its own
  std::vector<int> rcs(nd->ctx.size(), ACT_OK);
  std::vector<std::thread> th;
  for (size_t k = 0; k < nd->ctx.size(); k++) th.emplace_back([&, k] { rcs[k] = act_ctx_set_fixed_base_bits(nd->ctx[k], base, bits); });
  for (std::thread& t : th) t.join();
  for (size_t k = 0; k < nd->ctx.size(); k++) if (rcs[k]) { set_node_err(nd, act_last_error(nd->ctx[k])); return rcs[k]; }
  return ACT_OK;
}
int act_node_set_balance(act_node* nd, int weighted, int tail_64ths) {
  if (!nd || tail_64ths < -1 || tail_64ths > 32) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  nd->weighted = weighted != 0; nd->tail_64ths = tail_64ths;
  if (!nd->weighted) nd->weight.assign(nd->ctx.size(), 1.0);
  return ACT_OK;
}
int act_node_balance_state(act_node* nd, double* spread, double* tail_fraction) {
  if (!nd) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  if (spread) *spread = nd->spread;
  if (tail_fraction) *tail_fraction = nd->last_tail_fraction;
  return ACT_OK;
}
int act_node_device_stats(act_node* nd, int k, double* weight, uint64_t* last_lanes, double* last_seconds, uint64_t* last_calls) {
  if (!nd || k < 0 || k >= (int)nd->ctx.size()) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  if (weight) *weight = nd->weight[k];
  if (last_lanes) *last_lanes = nd->last[k].lanes;
  if (last_seconds) *last_seconds = nd->last[k].seconds;
  if (last_calls) *last_calls = nd->last[k].calls;
  return ACT_OK;
}
int act_node_request_batch(act_node* nd, size_t n, const uint8_t* pre, const uint8_t* rng, uint8_t* out_req) {
  if (!nd || (n && (!pre || !rng || !out_req))) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_request_batch(nd->ctx[k], m, ACT_MEM_HOST, at(pre, off, 64), at(rng, off, 128), at(out_req, off, 128));
  });
}

int act_node_issue_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* req, const uint8_t* c, const uint8_t* rng,
                         int rng_mode, uint8_t* out_resp, uint8_t* status) {
  if (!nd || !sk || (n && (!req || !c || !rng || !out_resp || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  if (rng_mode == ACT_RNG_PER_LANE || nd->ctx.size() == 1)
    return run(nd, n, [&](size_t k, size_t off, size_t m) {
      return act_issue_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(req, off, 128), at(c, off, 32),
                             rng_mode == ACT_RNG_PER_LANE ? at(rng, off, 128) : rng, rng_mode, at(out_resp, off, 160), status + off);
    });
  int rc = run(nd, n, [&](size_t k, size_t off, size_t m) { return act_issue_check_batch(nd->ctx[k], m, ACT_MEM_HOST, at(req, off, 128), status + off); });
  if (rc) return rc;
  const std::vector<uint8_t> checked(status, status + n);
  const AcceptedBefore before(checked.data(), n);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_issue_sign_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(req, off, 128), at(c, off, 32), checked.data() + off, rng + before(off) * 128,
                                ACT_RNG_SEQUENTIAL, at(out_resp, off, 160), status + off);
  });
}

// The halves on their own, for a caller that draws its rng between them (the Rust binding advances the caller's generator
// by exactly 128 bytes per accepted lane, as the sequential loop would: INTEGRATION.md).
int act_node_issue_check_batch(act_node* nd, size_t n, const uint8_t* req, uint8_t* status) {
  if (!nd || (n && (!req || !status))) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  return run(nd, n, [&](size_t k, size_t off, size_t m) { return act_issue_check_batch(nd->ctx[k], m, ACT_MEM_HOST, at(req, off, 128), status + off); });
}
int act_node_issue_sign_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* req, const uint8_t* c, const uint8_t* status_in,
                              const uint8_t* rng, int rng_mode, uint8_t* out_resp, uint8_t* status) {
  if (!nd || !sk || (n && (!req || !c || !status_in || !rng || !out_resp || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  const std::vector<uint8_t> checked(status_in, status_in + n);
  const AcceptedBefore before(checked.data(), n);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_issue_sign_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(req, off, 128), at(c, off, 32), checked.data() + off,
                                rng + (rng_mode == ACT_RNG_PER_LANE ? off : before(off)) * 128, rng_mode, at(out_resp, off, 160), status + off);
  });
}
// refund signatures as records (out_rec = 128) or as CBOR Refund messages (cbor: out_rec = act_cbor_size(REFUND)); rng already resolved
static int refund_sign_locked(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* kprime, const uint8_t* status_in, const uint8_t* rng,
                              int rng_mode, bool cbor, uint8_t* out, uint8_t* status, std::vector<Piece>* pieces) {
  const std::vector<uint8_t> checked(status_in, status_in + n);
  const AcceptedBefore before(checked.data(), n);
  const size_t out_rec = cbor ? act_cbor_size(nd->ctx[0], ACT_CBOR_REFUND) : 128;
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    const uint8_t* r = rng + (rng_mode == ACT_RNG_PER_LANE ? off : before(off)) * 128;
    return cbor ? act_refund_sign_cbor_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(kprime, off, 32), checked.data() + off, r, rng_mode, at(out, off, out_rec), status + off)
                : act_refund_sign_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(kprime, off, 32), checked.data() + off, r, rng_mode, at(out, off, out_rec), status + off);
  }, pieces);
}
static size_t count_zero(const uint8_t* st, size_t n) { size_t a = 0; for (size_t i = 0; i < n; i++) a += st[i] == 0; return a; }
static int refund_sign_any(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* kprime, const uint8_t* status_in, const uint8_t* rng,
                           int rng_mode, bool cbor, uint8_t* out, uint8_t* status) {
  if (!nd || !sk || !rng || (n && (!kprime || !status_in || !out || !status))) return ACT_ERR_ARG;
  act::DrawnRng drawn;
  if (!cbor && rng_mode == ACT_RNG_CALLBACK) return ACT_ERR_ARG;      // (the record-level halves take bytes: include/act_mi355x.h)
  int rc = drawn.resolve(rng, rng_mode, rng_mode == ACT_RNG_CALLBACK ? count_zero(status_in, n) : 0); if (rc) return rc;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  return refund_sign_locked(nd, n, sk, kprime, status_in, rng, rng_mode, cbor, out, status, nullptr);
}
int act_node_refund_sign_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* kprime, const uint8_t* status_in, const uint8_t* rng,
                               int rng_mode, uint8_t* out_refund, uint8_t* status) {
  return refund_sign_any(nd, n, sk, kprime, status_in, rng, rng_mode, false, out_refund, status);
}
int act_node_refund_sign_cbor_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* kprime, const uint8_t* status_in, const uint8_t* rng,
                                    int rng_mode, uint8_t* out_refund_cbor, uint8_t* status) {
  return refund_sign_any(nd, n, sk, kprime, status_in, rng, rng_mode, true, out_refund_cbor, status);
}

int act_node_issuance_to_credit_token_batch(act_node* nd, size_t n, const uint8_t* pre, const uint8_t w[32], const uint8_t* req,
                                            const uint8_t* resp, uint8_t* out_token, uint8_t* status) {
  if (!nd || !w || (n && (!pre || !req || !resp || !out_token || !status))) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_issuance_to_credit_token_batch(nd->ctx[k], m, ACT_MEM_HOST, at(pre, off, 64), w, at(req, off, 128), at(resp, off, 160),
                                              at(out_token, off, 160), status + off);
  });
}

int act_node_prove_spend_batch(act_node* nd, size_t n, const uint8_t* token, const uint8_t* s_, const uint8_t* rng, uint8_t* out_proof,
                               uint8_t* out_prerefund, uint8_t* status) {
  if (!nd || (n && (!token || !s_ || !rng || !out_proof || !out_prerefund || !status))) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  const size_t pb = act_spend_proof_bytes(nd->ctx[0]), rb = act_prove_rng_bytes(nd->ctx[0]);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_prove_spend_batch(nd->ctx[k], m, ACT_MEM_HOST, at(token, off, 160), at(s_, off, 32), at(rng, off, rb), at(out_proof, off, pb),
                                 at(out_prerefund, off, 96), status + off);
  });
}

int act_node_prove_spend_seeded_batch(act_node* nd, size_t n, const uint8_t* token, const uint8_t* s_, const uint8_t seed[32], uint64_t first_lane,
                                      uint8_t* out_proof, uint8_t* out_prerefund, uint8_t* status) {
  if (!nd || !seed || (n && (!token || !s_ || !out_proof || !out_prerefund || !status))) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  const size_t pb = act_spend_proof_bytes(nd->ctx[0]);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {      // lane numbers are global: a piece starts at first_lane + its offset, whichever GPU takes it
    return act_prove_spend_seeded_batch(nd->ctx[k], m, ACT_MEM_HOST, at(token, off, 160), at(s_, off, 32), seed, first_lane + off, at(out_proof, off, pb),
                                        at(out_prerefund, off, 96), status + off);
  });
}

int act_node_verify_spend_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* proof, uint8_t* status, uint8_t* out_kprime) {
  if (!nd || !sk || (n && (!proof || !status))) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  const size_t pb = act_spend_proof_bytes(nd->ctx[0]);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_verify_spend_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(proof, off, pb), status + off, at(out_kprime, off, 32));
  });
}

// wire bytes: a piece takes messages [off, off + m); offsets are absolute into `cbor`, so every piece gets the same base
int act_node_verify_spend_cbor_keys_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* cbor, const uint64_t* offsets, uint8_t* status,
                                          uint8_t* out_kprime, uint8_t* out_nullifier) {
  if (!nd || !sk || (n && (!cbor || !status))) return ACT_ERR_ARG;
  const size_t ml = act_cbor_size(nd->ctx[0], ACT_CBOR_SPEND_PROOF);
  std::lock_guard<std::mutex> node_lock(nd->mu);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_verify_spend_cbor_keys_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, offsets ? cbor : cbor + off * ml, offsets ? offsets + off : nullptr,
                                            status + off, at(out_kprime, off, 32), at(out_nullifier, off, 32));
  });
}
int act_node_verify_spend_cbor_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* cbor, const uint64_t* offsets, uint8_t* status,
                                     uint8_t* out_kprime) {
  return act_node_verify_spend_cbor_keys_batch(nd, n, sk, cbor, offsets, status, out_kprime, nullptr);
}

int act_node_refund_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* proof, const uint8_t* rng, int rng_mode,
                          uint8_t* out_refund, uint8_t* status) {
  if (!nd || !sk || (n && (!proof || !rng || !out_refund || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  const size_t pb = act_spend_proof_bytes(nd->ctx[0]);
  if (rng_mode == ACT_RNG_PER_LANE || nd->ctx.size() == 1)
    return run(nd, n, [&](size_t k, size_t off, size_t m) {
      return act_refund_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(proof, off, pb), rng_mode == ACT_RNG_PER_LANE ? at(rng, off, 128) : rng,
                              rng_mode, at(out_refund, off, 128), status + off);
    });
  // phase 1: verification of every lane, keeping enc(K') (32 bytes per lane) -- the only thing the signature needs
  std::vector<uint8_t> kprime(n * 32);
  int rc = run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_verify_spend_batch(nd->ctx[k], m, ACT_MEM_HOST, sk, at(proof, off, pb), status + off, kprime.data() + off * 32);
  });
  if (rc) return rc;
  // phase 2: X_A = g + K', then the BBS signature (src/lib.rs:846-868), each piece from its own offset into the stream
  return refund_sign_locked(nd, n, sk, kprime.data(), status, rng, ACT_RNG_SEQUENTIAL, false, out_refund, status, nullptr);
}

// wire bytes in, wire bytes out: verification of every message first (all verdicts are then known), then the signatures framed as
// CBOR Refund messages -- ACT_RNG_SEQUENTIAL / ACT_RNG_CALLBACK hand their bytes to the signed lanes in lane order
int act_node_refund_cbor_batch(act_node* nd, size_t n, const uint8_t sk[64], const uint8_t* cbor, const uint64_t* offsets, const uint8_t* rng,
                               int rng_mode, uint8_t* out_refund_cbor, uint8_t* status) {
  if (!nd || !sk || !rng || (n && (!cbor || !out_refund_cbor || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL && rng_mode != ACT_RNG_CALLBACK) return ACT_ERR_ARG;
  if (n == 0) return ACT_OK;
  // A few messages whose rng slices do not depend on the verdicts (per-lane bytes, or one message with its 128 bytes): one context does
  // all of it in ONE call -- unframing, verification, the signature beside it (cbor_impl.inc refund_cbor_tiny_records: 2.1 ms for one message
  // instead of 3.1).  The context's own lock serialises it with whatever else that context is doing.
  if (n <= 64 && (rng_mode == ACT_RNG_PER_LANE || (rng_mode == ACT_RNG_SEQUENTIAL && n == 1))) {
    act_ctx* c = nd->ctx[0];          // nothing to cut: the node's first context
    const int rc = act_refund_cbor_batch(c, n, ACT_MEM_HOST, sk, cbor, offsets, rng, rng_mode, out_refund_cbor, status);
    if (rc) small_call_err(nd, c);
    return rc;
  }
  std::vector<uint8_t> kprime(n * 32), verdict(n);
  int rc = act_node_verify_spend_cbor_keys_batch(nd, n, sk, cbor, offsets, verdict.data(), kprime.data(), nullptr);
  if (rc) return rc;
  return refund_sign_any(nd, n, sk, kprime.data(), verdict.data(), rng, rng_mode, true, out_refund_cbor, status);
}

int act_node_refund_to_credit_token_batch(act_node* nd, size_t n, const uint8_t* prerefund, const uint8_t* proof, const uint8_t* refund,
                                          const uint8_t w[32], uint8_t* out_token, uint8_t* status) {
  if (!nd || !w || (n && (!prerefund || !proof || !refund || !out_token || !status))) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> node_lock(nd->mu);
  const size_t pb = act_spend_proof_bytes(nd->ctx[0]);
  return run(nd, n, [&](size_t k, size_t off, size_t m) {
    return act_refund_to_credit_token_batch(nd->ctx[k], m, ACT_MEM_HOST, at(prerefund, off, 96), at(proof, off, pb), at(refund, off, 128), w,
                                            at(out_token, off, 160), status + off);
  });
}

}  // extern "C"

// ---- the double-spend set over the GPUs of a node ----------------------------------------------------------------------
// One act_nullifier_set per device; a nullifier lives on exactly one of them (owner = a keyed hash of the REDUCED scalar, so
// k and k + l go to the same owner), so the answer for a key only ever depends on one set and the batch keeps the meaning
// of the reference's sequential loop (`if is_spent(k) reject else insert(k)`, /root/reference/src/tests.rs:29-50) in lane
// order: the host buckets the keys by owner preserving lane order, every GPU checks-and-inserts its bucket on its own
// thread, the answers are scattered back.  33 bytes per spend cross PCIe; no peer traffic, no collective.
struct act_node_nullifier_set {
  std::vector<act_nullifier_set*> sets;
  std::vector<int> devices;
  uint64_t route_key[2] = {0, 0};
  std::string err;
  std::mutex mu;
  // routing scratch, kept between calls (under mu) and only ever grown: a fresh 32 MB of key buckets per million-key call would
  // be page-faulted in by one thread every time
  struct Bucket { std::vector<uint32_t> lanes; std::vector<uint8_t> keys, spent; size_t count = 0; };
  std::vector<Bucket> buckets;
  std::vector<uint16_t> owner;
  std::vector<size_t> place;
};

namespace {
// 256-bit little-endian value mod l, l = 2^252 + 27742317777372353535851937790883648493 (the set itself reduces again on
// the device; here only the owner must not depend on the representative)
void reduce_mod_l(const uint8_t in[32], uint64_t out[4]) {
  static const uint64_t Lw[4] = {0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0, 0x1000000000000000ull};
  uint64_t v[4]; memcpy(v, in, 32);
  for (int rep = 0; rep < 16; rep++) {                     // v < 2^256 < 16 l: at most 15 subtractions
    uint64_t t[4]; unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)v[i] - Lw[i] - (uint64_t)borrow; t[i] = (uint64_t)d; borrow = (d >> 64) & 1; }
    if (borrow) break;
    memcpy(v, t, 32);
  }
  memcpy(out, v, 32);
}
// Owner of a key = SipHash-1-3 of the reduced scalar under the set's 128-bit routing key (the same construction the per-GPU
// tables use for their slots, nullifier_impl.inc null_hash, each under its own key).  Clients choose their nullifiers: with
// an unkeyed or weakly mixed owner function they could aim every spend at one GPU and exhaust that GPU's capacity.
uint64_t route_hash(const uint64_t k[4], const uint64_t key[2]) {
  uint64_t v0 = key[0] ^ 0x736f6d6570736575ull, v1 = key[1] ^ 0x646f72616e646f6dull, v2 = key[0] ^ 0x6c7967656e657261ull, v3 = key[1] ^ 0x7465646279746573ull;
  auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
  auto round = [&]() {
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
  };
  for (int i = 0; i < 4; i++) { v3 ^= k[i]; round(); v0 ^= k[i]; }
  const uint64_t last = (uint64_t)32 << 56;
  v3 ^= last; round(); v0 ^= last;
  v2 ^= 0xff; round(); round(); round();
  return v0 ^ v1 ^ v2 ^ v3;
}
}  // namespace

extern "C" {
int act_node_nullifier_set_create(const int* devices, int n_devices, size_t capacity_per_device, const uint8_t salt[16], act_node_nullifier_set** out) {
  if (!devices || n_devices < 1 || n_devices > 4096 || !out || !capacity_per_device) return ACT_ERR_ARG;
  act_node_nullifier_set* ns = new act_node_nullifier_set();
  *out = ns;
  if (salt) memcpy(ns->route_key, salt, 16);
  else {                                                       // no caller-supplied key: 16 bytes from the OS, or no set at all
    size_t got = 0;
    while (got < 16) {
      ssize_t r = getrandom(reinterpret_cast<uint8_t*>(ns->route_key) + got, 16 - got, 0);
      if (r <= 0) { ns->err = "getrandom failed: no routing key"; return ACT_ERR_ARG; }
      got += (size_t)r;
    }
  }
  for (int k = 0; k < n_devices; k++) {
    act_nullifier_set* s = nullptr;
    int rc = act_nullifier_set_create(devices[k], capacity_per_device, nullptr, &s);     // every table draws its own slot-hash key
    if (rc) { ns->err = "device " + std::to_string(devices[k]) + ": " + (s ? act_nullifier_set_last_error(s) : "create failed"); if (s) act_nullifier_set_destroy(s); return rc; }
    ns->sets.push_back(s); ns->devices.push_back(devices[k]);
  }
  return ACT_OK;
}
void act_node_nullifier_set_destroy(act_node_nullifier_set* ns) {
  if (!ns) return;
  for (act_nullifier_set* s : ns->sets) act_nullifier_set_destroy(s);
  delete ns;
}
size_t act_node_nullifier_set_len(const act_node_nullifier_set* ns) {
  size_t n = 0;
  if (ns) for (act_nullifier_set* s : ns->sets) n += act_nullifier_set_len(s);
  return n;
}
const char* act_node_nullifier_set_last_error(const act_node_nullifier_set* ns) {
  if (!ns) return "null set";
  thread_local std::string mine;
  { std::lock_guard<std::mutex> lk(const_cast<act_node_nullifier_set*>(ns)->mu); mine = ns->err; }
  return mine.c_str();
}

int act_node_nullifier_check_and_insert_batch(act_node_nullifier_set* ns, size_t n, const uint8_t* nullifiers, size_t stride, const uint8_t* skip_mask,
                                              uint8_t* out_spent) {
  if (!ns || (n && (!nullifiers || !out_spent)) || stride < 32) return ACT_ERR_ARG;
  std::lock_guard<std::mutex> lock(ns->mu);
  const size_t parts = ns->sets.size();
  // Bucket the keys by owner, lane order kept inside every bucket, on the host workers (host_pool.cpp): the batch is cut into
  // segments; pass 1 computes every lane's owner and the segment's count per owner, a prefix sum over (segment, owner) gives each
  // segment its place in each bucket, pass 2 copies lanes and keys there.  (A million keys at the proof stride: one thread spends
  // longer here than eight GPUs spend on their look-ups.)
  struct Route { act_node_nullifier_set* ns; const uint8_t* nullifiers; size_t stride, n, seg, parts; const uint8_t* skip; uint8_t* out_spent; }
      r{ns, nullifiers, stride, n, 0, parts, skip_mask, out_spent};
  const size_t segs = std::max<size_t>(1, std::min<size_t>(256, n / 4096));
  r.seg = (n + segs - 1) / segs;
  if (ns->owner.size() < n) ns->owner.resize(n);
  ns->place.assign(segs * parts, 0);                 // place[s * parts + p]: count, then first index, of segment s in bucket p
  ns->buckets.resize(parts);
  act_host_parallel_for(segs, 1, 0, [](void* p, size_t s0, size_t s1) {
    Route& r = *static_cast<Route*>(p);
    for (size_t s = s0; s < s1; s++)
      for (size_t i = s * r.seg, e = std::min(r.n, i + r.seg); i < e; i++) {
        r.out_spent[i] = 0;
        if (r.skip && r.skip[i]) { r.ns->owner[i] = 0xFFFF; continue; }      // e.g. the status of a rejected proof: neither checked nor inserted
        uint64_t k[4]; reduce_mod_l(r.nullifiers + i * r.stride, k);
        const size_t o = (size_t)(route_hash(k, r.ns->route_key) % r.parts);
        r.ns->owner[i] = (uint16_t)o; r.ns->place[s * r.parts + o]++;
      }
  }, &r);
  for (size_t p = 0; p < parts; p++) {
    size_t at = 0;
    for (size_t s = 0; s < segs; s++) { const size_t c = ns->place[s * parts + p]; ns->place[s * parts + p] = at; at += c; }
    auto& b = ns->buckets[p];
    b.count = at;
    if (b.lanes.size() < at) { b.lanes.resize(at); b.keys.resize(32 * at); b.spent.resize(at); }
  }
  act_host_parallel_for(segs, 1, 0, [](void* p, size_t s0, size_t s1) {
    Route& r = *static_cast<Route*>(p);
    for (size_t s = s0; s < s1; s++)
      for (size_t i = s * r.seg, e = std::min(r.n, i + r.seg); i < e; i++) {
        const size_t o = r.ns->owner[i];
        if (o == 0xFFFF) continue;
        auto& b = r.ns->buckets[o];
        const size_t at = r.ns->place[s * r.parts + o]++;
        b.lanes[at] = (uint32_t)i;
        memcpy(b.keys.data() + 32 * at, r.nullifiers + i * r.stride, 32);
      }
  }, &r);
  std::vector<int> rc(parts, ACT_OK);
  std::vector<std::thread> th;
  auto work = [&](size_t p) {
    auto& b = ns->buckets[p];
    if (b.count) rc[p] = act_nullifier_check_and_insert_batch(ns->sets[p], b.count, ACT_MEM_HOST, b.keys.data(), 32, nullptr, b.spent.data());
  };
  for (size_t p = 1; p < parts; p++) th.emplace_back(work, p);
  work(0);
  for (auto& t : th) t.join();
  // Every device answers for its own keys.  If one of them failed, the others have still inserted theirs: their lanes get
  // their (final) answers, the failed device's lanes get ACT_NULLIFIER_UNDETERMINED, and the call reports the error --
  // a caller that retries must resubmit only the undetermined lanes, or the lanes already inserted would come back "spent".
  int first_rc = ACT_OK;
  ns->err.clear();
  for (size_t p = 0; p < parts; p++) {
    if (rc[p]) {
      if (!first_rc) first_rc = rc[p];
      ns->err += (ns->err.empty() ? "device " : "; device ") + std::to_string(ns->devices[p]) + ": " + act_nullifier_set_last_error(ns->sets[p]);
      for (size_t j = 0; j < ns->buckets[p].count; j++) out_spent[ns->buckets[p].lanes[j]] = ACT_NULLIFIER_UNDETERMINED;
    } else {
      for (size_t j = 0; j < ns->buckets[p].count; j++) out_spent[ns->buckets[p].lanes[j]] = ns->buckets[p].spent[j];
    }
  }
  return first_rc;
}
}  // extern "C"

// The issuer's whole redemption step over the GPUs of a node (act_redeem_batch's meaning, include/act_mi355x.h): verification of
// every lane, the node-level nullifier set over the whole batch in lane order (verdicts as skip mask), then the signatures --
// ACT_RNG_SEQUENTIAL / ACT_RNG_CALLBACK draw only for lanes that are signed, from one stream, exactly as the sequential loop would.
// Records in / records out (act_node_redeem_batch) and wire bytes in / wire bytes out (act_node_redeem_cbor_batch) share this body.
// Failures after verification never lose a decision (same contract as act_redeem_batch): a device of the nullifier set that
// fails leaves ITS lanes ACT_STATUS_NULLIFIER_UNDETERMINED (not recorded, not signed) while every other lane is finished; a GPU
// that fails while signing leaves the lanes of ITS pieces that were to be signed ACT_STATUS_RECORDED_UNSIGNED (nullifier recorded,
// refund owed); status[] and the output are complete for all other lanes and the error code says that something was left over.
static int node_redeem(act_node* nd, act_node_nullifier_set* set, size_t n, const uint8_t sk[64], const uint8_t* proof, const uint8_t* cbor,
                       const uint64_t* offsets, const uint8_t* rng, int rng_mode, uint8_t* out, uint8_t* status) {
  const bool wire = cbor != nullptr;
  if (!nd || !set || !sk || !rng || (n && ((!proof && !cbor) || !out || !status))) return ACT_ERR_ARG;
  if (rng_mode != ACT_RNG_PER_LANE && rng_mode != ACT_RNG_SEQUENTIAL && rng_mode != ACT_RNG_CALLBACK) return ACT_ERR_ARG;
  if (n == 0) return ACT_OK;
  const size_t pb = act_spend_proof_bytes(nd->ctx[0]), out_rec = wire ? act_cbor_size(nd->ctx[0], ACT_CBOR_REFUND) : 128;
  // A few items whose rng slices do not depend on the verdicts (per-lane bytes, or one item with its 128 bytes): ONE context computes
  // the refunds in one call, the signature beside the verification (2.1 ms for one item instead of 3.1 through verify -> store ->
  // sign); then the store decides, and a lane whose nullifier was not fresh gets its status and loses its refund.  The same answers
  // and the same store as the general path below; nothing is signed behind a recorded nullifier here, so nothing can fail there.
  if (n <= 64 && (rng_mode == ACT_RNG_PER_LANE || (rng_mode == ACT_RNG_SEQUENTIAL && n == 1))) {
    act_ctx* c = nd->ctx[0];          // nothing to cut: the node's first context
    // the refunds are computed into a buffer of this function's: a refund for a token that turns out to be spent must never be visible
    // in the caller's memory, not even until the store has answered (the engine-level twin, nullifier_impl.inc redeem_impl, does the same)
    std::vector<uint8_t> st(n), sp(n), nl(wire ? n * 32 : 0), rec(n * out_rec);
    struct Wipe { std::vector<uint8_t>& v; ~Wipe() { volatile uint8_t* q = v.data(); for (size_t i = 0; i < v.size(); i++) q[i] = 0; } } wipe_rec{rec};
    int rc = wire ? act_refund_cbor_keys_batch(c, n, ACT_MEM_HOST, sk, cbor, offsets, rng, rng_mode, rec.data(), st.data(), nl.data())
                  : act_refund_batch(c, n, ACT_MEM_HOST, sk, proof, rng, rng_mode, rec.data(), st.data());
    if (rc) { small_call_err(nd, c); return rc; }
    const int rc_null = act_node_nullifier_check_and_insert_batch(set, n, wire ? nl.data() : proof, wire ? 32 : pb, st.data(), sp.data());
    for (size_t i = 0; i < n; i++) {
      if (st[i] == 0 && sp[i]) st[i] = sp[i] == 1 ? ACT_STATUS_DOUBLE_SPEND : ACT_STATUS_NULLIFIER_UNDETERMINED;
      if (st[i] == 0) memcpy(out + i * out_rec, rec.data() + i * out_rec, out_rec); else memset(out + i * out_rec, 0, out_rec);
      status[i] = st[i];
    }
    if (rc_null) { std::lock_guard<std::mutex> node_lock(nd->mu); set_node_err(nd, std::string("nullifier set: ") + act_node_nullifier_set_last_error(set)); }
    return rc_null;
  }
  std::vector<uint8_t> kprime(n * 32), verdict(n), spent(n), nul(wire ? n * 32 : 0);
  int rc = wire ? act_node_verify_spend_cbor_keys_batch(nd, n, sk, cbor, offsets, verdict.data(), kprime.data(), nul.data())
                : act_node_verify_spend_batch(nd, n, sk, proof, verdict.data(), kprime.data());
  if (rc) return rc;
  const int rc_null = act_node_nullifier_check_and_insert_batch(set, n, wire ? nul.data() : proof, wire ? 32 : pb, verdict.data(), spent.data());
  for (size_t i = 0; i < n; i++)
    if (verdict[i] == 0 && spent[i]) verdict[i] = spent[i] == 1 ? ACT_STATUS_DOUBLE_SPEND : ACT_STATUS_NULLIFIER_UNDETERMINED;
  // the caller's generator is touched only now, and only for the lanes that are signed
  act::DrawnRng drawn;
  int rc_sign = drawn.resolve(rng, rng_mode, rng_mode == ACT_RNG_CALLBACK ? count_zero(verdict.data(), n) : 0);
  std::lock_guard<std::mutex> node_lock(nd->mu);
  std::vector<Piece> pieces;
  if (!rc_sign) rc_sign = refund_sign_locked(nd, n, sk, kprime.data(), verdict.data(), rng, rng_mode, wire, out, status, &pieces);
  else pieces.push_back({(size_t)-1, 0, n, rc_sign});
  if (rc_sign) {
    for (const Piece& q : pieces) {
      if (!q.rc) continue;
      for (size_t i = q.off; i < q.off + q.m; i++) status[i] = verdict[i] == 0 ? ACT_STATUS_RECORDED_UNSIGNED : verdict[i];
      memset(out + q.off * out_rec, 0, q.m * out_rec);
    }
    return rc_sign;
  }
  if (rc_null) set_node_err(nd, std::string("nullifier set: ") + act_node_nullifier_set_last_error(set));
  return rc_null;
}
extern "C" int act_node_redeem_batch(act_node* nd, act_node_nullifier_set* set, size_t n, const uint8_t sk[64], const uint8_t* proof, const uint8_t* rng,
                                     int rng_mode, uint8_t* out_refund, uint8_t* status) {
  if (n && !proof) return ACT_ERR_ARG;
  return node_redeem(nd, set, n, sk, proof, nullptr, nullptr, rng, rng_mode, out_refund, status);
}
extern "C" int act_node_redeem_cbor_batch(act_node* nd, act_node_nullifier_set* set, size_t n, const uint8_t sk[64], const uint8_t* cbor, const uint64_t* offsets,
                                          const uint8_t* rng, int rng_mode, uint8_t* out_refund_cbor, uint8_t* status) {
  if (n && !cbor) return ACT_ERR_ARG;
  static const uint8_t none = 0;
  return node_redeem(nd, set, n, sk, nullptr, cbor ? cbor : &none, offsets, rng, rng_mode, out_refund_cbor, status);
}
