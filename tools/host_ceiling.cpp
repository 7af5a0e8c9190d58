// host_ceiling.cpp — the HOST side of an 8-GPU node in the contract mode (ACT_TRANSCRIPT_HOST, host-memory callers),
// run WITHOUT any GPU: how many GPUs' worth of host work do this box's cores sustain?
//
// Per proof and per GPU shard the host does exactly this (engine.hip spend_batch / hash_begin / hash_end, node.cpp run):
//   (1) staging   the 16 832-byte proof record travels from the caller's memory to the device.  For page-locked caller memory
//                 that is DMA and costs no core time; for an ordinary allocation (a Rust Vec) the HIP runtime first copies it
//                 into its own pinned staging area on the calling thread: one memcpy of 16 832 B per proof.  (--stage 0 / 1)
//   (2) hashing   the 15 784-byte "spend" transcript pre-image arrives in the shard's pinned D2H buffer and is hashed where it
//                 lies (no second copy) by act_host_b3_xof64_x16 (sixteen messages per call, host_hash.cpp), 64 XOF bytes out
//   (3) scatter   1 status byte per proof into the caller's array
// One OS thread stands for one hash worker; `shards` buffers are walked round-robin so that the working set is that of
// `shards` GPUs (2 x 65 536-proof chunks each would be 2 GB per GPU: scaled down with --lanes, the access pattern is streaming).
// Output: one JSON object per thread count: GB/s hashed, proofs/s, and "GPUs' worth" at the per-GPU verify rate given.
//
// Build: g++ -O3 -std=c++17 -pthread -o /tmp/host_ceiling tools/host_ceiling.cpp anonymous-credit-tokens_amd/csrc/host_hash.o
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" void act_host_b3_xof64_x16(const uint8_t* msgs, size_t stride, uint32_t len, uint32_t* xof);

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  int L = 128, shards = 8, stage = 1; size_t lanes = 16384; double per_gpu = 455000.0, seconds = 2.0;
  std::vector<int> tlist;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--shards")) shards = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--lanes")) lanes = (size_t)atol(argv[++i]);
    else if (!strcmp(argv[i], "--stage")) stage = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--per-gpu")) per_gpu = atof(argv[++i]);
    else if (!strcmp(argv[i], "--seconds")) seconds = atof(argv[++i]);
    else if (!strcmp(argv[i], "--threads")) { for (char* p = strtok(argv[++i], ","); p; p = strtok(nullptr, ",")) tlist.push_back(atoi(p)); }
  }
  if (tlist.empty()) tlist = {1, 2, 4, 8};
  const size_t pb = 32 * (14 + 4 * (size_t)L), tb = 184 + 40 * (6 + 3 * (size_t)L), stride = (tb + 15) & ~(size_t)15;
  // per shard: caller's proofs (pageable), the staging area, the transcript buffer (what the D2H copy delivered), xof, statuses
  struct Shard { std::vector<uint8_t> proofs, staged, tr; std::vector<uint32_t> xof; std::vector<uint8_t> status; };
  std::vector<Shard> sh(shards);
  for (auto& s : sh) {
    s.proofs.assign(lanes * pb, 0x5a); s.staged.assign(lanes * pb, 0); s.tr.assign(lanes * stride, 0xa5); s.xof.assign(lanes * 16, 0); s.status.assign(lanes, 0);
    for (size_t i = 0; i < lanes * stride; i += 4093) s.tr[i] = (uint8_t)i;
  }
  printf("[");
  bool first = true;
  for (int T : tlist) {
    std::atomic<size_t> next{0}; std::atomic<bool> stop{false}; std::atomic<uint64_t> done{0};
    const size_t groups_per_shard = lanes / 64;
    auto work = [&]() {
      uint64_t mine = 0;
      while (!stop.load(std::memory_order_relaxed)) {
        size_t g = next.fetch_add(1);
        Shard& s = sh[(g / groups_per_shard) % shards];
        size_t i0 = (g % groups_per_shard) * 64;
        if (stage) memcpy(s.staged.data() + i0 * pb, s.proofs.data() + i0 * pb, 64 * pb);
        for (size_t i = i0; i < i0 + 64; i += 16) act_host_b3_xof64_x16(s.tr.data() + i * stride, stride, (uint32_t)tb, s.xof.data() + i * 16);
        for (size_t i = i0; i < i0 + 64; i++) s.status[i] = (uint8_t)(s.xof[i * 16] & 1u);
        mine += 64;
      }
      done += mine;
    };
    std::vector<std::thread> th;
    double t0 = now();
    for (int t = 0; t < T; t++) th.emplace_back(work);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& t : th) t.join();
    double dt = now() - t0;
    double pps = (double)done.load() / dt;
    printf("%s\n {\"threads\": %d, \"shards\": %d, \"staging_memcpy\": %s, \"proofs_per_s\": %.0f, \"hash_GBps\": %.2f, \"staged_GBps\": %.2f, "
           "\"gpus_worth_at_%.0fk_per_gpu\": %.2f, \"threads_per_gpu\": %.2f}",
           first ? "" : ",", T, shards, stage ? "true" : "false", pps, pps * tb / 1e9, stage ? pps * pb / 1e9 : 0.0, per_gpu / 1e3, pps / per_gpu, T / (pps / per_gpu));
    first = false;
    fflush(stdout);
  }
  printf("\n]\n");
  return 0;
}
