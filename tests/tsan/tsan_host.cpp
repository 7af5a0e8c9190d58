// ThreadSanitizer driver for the library's host-side concurrency (tests/test_sanitizers.py builds it with -fsanitize=thread):
//   * csrc/host_pool.cpp   the process-wide worker pool: act_host_hash_many and act_host_parallel_for from several threads at once
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
  // ---- one node handle, four threads, small calls beside large ones (all under the handle's lock) ----
  act_node* nd = nullptr; uint8_t h[96] = {0};
  int two[2] = {0, 1};
  if (act_node_create(h, 128, two, 2, 0, &nd)) { printf("node create failed\n"); return 7; }
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
  printf("TSAN DRIVER DONE\n");
  return 0;
}
