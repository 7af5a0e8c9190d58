// ThreadSanitizer driver for the library's host-side concurrency (tests/test_sanitizers.py builds it with -fsanitize=thread):
//   * csrc/host_pool.cpp   the process-wide worker pool: act_host_hash_many and act_host_parallel_for from several threads at once
//   * csrc/coalesce.h      the queueing protocol of merged small calls (engine.hip runs GPU calls through it; here the merged
//                          "call" is a stand-in that sleeps and answers every request from its own input)
//   * csrc/node.cpp        the node-level nullifier set's routing on those workers (real host_pool.cpp here, not the mock's
//                          two-thread stand-in) and a node handle used from several threads
// linked against tests/node_mock/node_mock.cpp for the single-GPU entry points.  Any report makes the process exit non-zero.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <thread>
#include <vector>
#include "../../include/act_mi355x.h"
#include "../../anonymous-credit-tokens_amd/csrc/coalesce.h"
#include <chrono>

static std::atomic<uint64_t> g_sum{0};
static void add_range(void*, size_t i0, size_t i1) { uint64_t s = 0; for (size_t i = i0; i < i1; i++) s += i; g_sum.fetch_add(s); }

int main() {
  // ---- the pool: hashing and parallel-for from six threads at once ----
  std::vector<std::thread> th;
  std::vector<std::vector<uint32_t>> xofs(6);
  for (int t = 0; t < 6; t++) th.emplace_back([t, &xofs] {
    const size_t n = 300 + 17 * t, stride = 320; const uint32_t len = 257;
    std::vector<uint8_t> msgs(n * stride);
    for (size_t i = 0; i < msgs.size(); i++) msgs[i] = (uint8_t)(i * 31 + t);
    xofs[t].assign(n * 16, 0);
    for (int rep = 0; rep < 4; rep++) {
      act_host_hash_many(msgs.data(), stride, len, n, rep & 1 ? 3 : 0, xofs[t].data());
      act_host_parallel_for(10000 + t, 64, 0, add_range, nullptr);
    }
  });
  for (auto& x : th) x.join();
  th.clear();
  uint64_t want = 0;
  for (int t = 0; t < 6; t++) want += 4 * (uint64_t)(10000 + t) * (10000 + t - 1) / 2;
  if (g_sum.load() != want) { printf("parallel_for sum wrong\n"); return 2; }
  // one-thread hashing of the same messages gives the same words
  {
    const size_t n = 300, stride = 320; std::vector<uint8_t> msgs(n * stride);
    for (size_t i = 0; i < msgs.size(); i++) msgs[i] = (uint8_t)(i * 31);
    std::vector<uint32_t> one(n * 16);
    act_host_hash_many(msgs.data(), stride, 257, n, 1, one.data());
    if (memcmp(one.data(), xofs[0].data(), n * 64) != 0) { printf("hash differs between thread counts\n"); return 3; }
  }
  // ---- node-level nullifier sets: two sets, two threads, 30 000 keys each (many routing segments) ----
  int devs[3] = {0, 1, 2};
  for (int t = 0; t < 2; t++) th.emplace_back([t, &devs] {
    act_node_nullifier_set* ns = nullptr;
    if (act_node_nullifier_set_create(devs, 3, 200000, nullptr, &ns)) { printf("set create failed\n"); _exit(4); }
    const size_t n = 30000;
    std::vector<uint8_t> keys(n * 32), spent(n);
    for (size_t i = 0; i < n; i++) { memset(&keys[i * 32], 0, 32); uint64_t v = (i % 20000) * 2654435761u + t; memcpy(&keys[i * 32], &v, 8); }
    if (act_node_nullifier_check_and_insert_batch(ns, n, keys.data(), 32, nullptr, spent.data())) { printf("check failed\n"); _exit(5); }
    size_t dup = 0; for (size_t i = 0; i < n; i++) dup += spent[i];
    if (dup != 10000 || act_node_nullifier_set_len(ns) != 20000) { printf("nullifier answers wrong: %zu\n", dup); _exit(6); }
    act_node_nullifier_set_destroy(ns);
  });
  for (auto& x : th) x.join();
  th.clear();
  // ---- one node handle, four threads, small calls that bypass the handle's lock (act_node_set_coalescing) beside large ones ----
  act_node* nd = nullptr; uint8_t h[96] = {0};
  int two[2] = {0, 1};
  if (act_node_create(h, 128, two, 2, 0, &nd)) { printf("node create failed\n"); return 7; }
  act_node_set_coalescing(nd, 4);
  for (int t = 0; t < 4; t++) th.emplace_back([t, nd] {
    uint8_t sk[64] = {0};
    for (int rep = 0; rep < 50; rep++) {
      const size_t n = (rep % 5 == 0) ? 100 : 1;                 // 1: one context, no node lock; 100: sharded under the lock
      std::vector<uint8_t> proofs(n * 64, (uint8_t)(2 * t)), st(n), kp(n * 32), st2(n), rf(n * 128), rng(n * 128, 1);
      if (act_node_verify_spend_batch(nd, n, sk, proofs.data(), st.data(), kp.data())) { printf("verify failed\n"); _exit(8); }
      if (act_node_refund_sign_batch(nd, n, sk, kp.data(), st.data(), rng.data(), ACT_RNG_PER_LANE, rf.data(), st2.data())) { printf("sign failed\n"); _exit(9); }
      (void)act_node_last_error(nd);
    }
  });
  for (auto& x : th) x.join();
  th.clear();
  act_node_destroy(nd);
  // ---- the combiner: 24 threads, requests of 1 - 3 lanes in two groups; every request must get ITS answer, merged calls must
  //      only ever hold requests of one group and stay within the cap, and nobody may be lost or served twice
  {
    struct Req { size_t n; int rc = 0; bool done = false; int group; uint64_t in[3]; uint64_t out[3] = {0, 0, 0}; int served = 0; };
    act::Combiner<Req> co;
    std::atomic<int> calls{0}, merged_lanes{0}, bad{0};
    auto run = [&](const std::vector<Req*>& batch, size_t total) {
      calls.fetch_add(1); merged_lanes.fetch_add((int)total);
      size_t sum = 0;
      for (Req* q : batch) { if (q->group != batch[0]->group) bad.fetch_add(1); sum += q->n; }
      if (sum != total || (batch.size() > 1 && total > 16)) bad.fetch_add(1);
      std::this_thread::sleep_for(std::chrono::microseconds(300));          // a call takes a while: requests pile up behind it
      for (Req* q : batch) { for (size_t i = 0; i < q->n; i++) q->out[i] = q->in[i] * 3 + 1; q->served++; }
      return 0;
    };
    for (int t = 0; t < 24; t++) th.emplace_back([t, &co, &run, &bad] {
      for (int rep = 0; rep < 60; rep++) {
        Req r; r.n = 1 + (size_t)((t + rep) % 3); r.group = (t * 7 + rep) % 2;
        for (size_t i = 0; i < r.n; i++) r.in[i] = (uint64_t)t * 1000003u + (uint64_t)rep * 17u + i;
        const int rc = co.submit(r, 16, [](const Req& a, const Req& b) { return a.group == b.group; }, run);
        if (rc || !r.done || r.served != 1) bad.fetch_add(1);
        for (size_t i = 0; i < r.n; i++) if (r.out[i] != r.in[i] * 3 + 1) bad.fetch_add(1);
      }
    });
    for (auto& x : th) x.join();
    th.clear();
    if (bad.load() || !co.q.empty() || co.leader) { printf("combiner: %d bad\n", bad.load()); return 10; }
    if (calls.load() >= 24 * 60) { printf("combiner never merged anything (%d calls)\n", calls.load()); return 11; }
    printf("combiner: %d requests in %d merged calls\n", 24 * 60, calls.load());
  }
  printf("TSAN DRIVER DONE\n");
  return 0;
}
