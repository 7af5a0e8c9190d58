// host_pool.cpp — the host side of the host-transcript mode: ONE process-wide pool of BLAKE3 workers shared by every
// context of the process (the reference hashes each transcript on the CPU, /root/reference/src/transcript.rs:149-152).
//
// Round 3 started `usable_cpus()` fresh std::threads per hash piece per context: a node handle over 8 GPUs created and joined
// 8 x all-CPUs threads four times per chunk.  Now:
//   * the pool has usable_cpus() workers, started once, on the first hash of the process (a process that only ever uses device
//     transcripts never starts a thread);
//   * a hashing call takes a SHARE of them: pool size / (calls hashing at this moment), or the context's explicit
//     act_ctx_set_host_threads() if that is smaller -- eight contexts that hash at the same time get an eighth each, one that
//     hashes alone gets them all; the calling thread works too, so a share of 1 needs no hand-off at all;
//   * ACT_NUMA=1 pins worker k to CPU k of the process's affinity mask (workers then stay on their socket; the pinned
//     transcript buffers are first touched by the HIP runtime, which this library does not control).
// Plain host C++ (g++): nothing here touches a device.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "blake3_hd.h"

extern "C" void act_host_b3_xof64_x16(const uint8_t* msgs, size_t stride, uint32_t len, uint32_t* xof);   // host_hash.cpp

namespace {

// CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a 16-CPU container on a
// 256-thread host reports hardware_concurrency() = 256; hashing with that many threads runs at half the rate of 16)
int usable_cpus_now() {
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { int k = CPU_COUNT(&set); if (k > 0 && (n < 1 || k < n)) n = k; }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
    char q[32]; long period = 0;
    if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      long k = (atol(q) + period - 1) / period; if (k > 0 && k < n) n = (int)k;
    }
    fclose(f);
  } else {
    long quota = -1, period = 0;                                              // cgroup v1
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
    if (quota > 0 && period > 0) { long k = (quota + period - 1) / period; if (k > 0 && k < n) n = (int)k; }
  }
  return n < 1 ? 1 : n;
}

constexpr size_t GRAIN = 64;       // messages per work item: four SIMD groups of sixteen

// a job = fn(ctx, i0, i1) over [0, n) in items of `grain`; whoever holds a ticket takes items until none are left
struct Job {
  void (*fn)(void*, size_t, size_t); void* ctx; size_t n, grain;
  std::atomic<size_t> next{0}, done{0};
  void work() {
    for (;;) {
      const size_t i0 = next.fetch_add(grain, std::memory_order_relaxed);
      if (i0 >= n) return;
      const size_t i1 = std::min(n, i0 + grain);
      fn(ctx, i0, i1);
      done.fetch_add(i1 - i0, std::memory_order_release);
    }
  }
};

struct HashArgs { const uint8_t* msgs; size_t stride; uint32_t len; uint32_t* xof; };
void hash_items(void* p, size_t i0, size_t i1) {
  const HashArgs& a = *static_cast<const HashArgs*>(p);
  size_t i = i0;
  for (; i + 16 <= i1; i += 16) act_host_b3_xof64_x16(a.msgs + i * a.stride, a.stride, a.len, a.xof + i * 16);
  for (; i < i1; i++) act::b3_hash_xof64(a.xof + i * 16, reinterpret_cast<const uint32_t*>(a.msgs + i * a.stride), a.len);
}

class Pool {
 public:
  static Pool& get() { static Pool* p = new Pool(); return *p; }      // never destroyed: workers may outlive static destructors
  int size() const { return size_; }
  // hashes the job on up to `par` threads, the caller among them; returns when every message is hashed
  void run(const std::shared_ptr<Job>& job, int par) {
    const size_t items = (job->n + job->grain - 1) / job->grain;
    const int helpers = (int)std::min<size_t>((size_t)std::max(par, 1) - 1, items > 0 ? items - 1 : 0);
    if (helpers > 0) {
      start_workers();
      { std::lock_guard<std::mutex> lk(mu_); for (int k = 0; k < helpers; k++) tickets_.push_back(job); }
      if (helpers == 1) cv_.notify_one(); else cv_.notify_all();
    }
    jobs_.fetch_add(1, std::memory_order_relaxed);
    job->work();
    // the tail: helpers still inside their last work item (~100 us each); a ticket nobody has picked up yet finds the job
    // exhausted and retires at once, the job itself lives as long as any ticket holds it
    for (int spin = 0; job->done.load(std::memory_order_acquire) < job->n; spin++) { if (spin < 64) __builtin_ia32_pause(); else std::this_thread::yield(); }
  }
  uint64_t jobs() const { return jobs_.load(); }
  uint64_t threads_created() const { return created_.load(); }

 private:
  Pool() : size_(usable_cpus_now()) {}
  void start_workers() {
    if (started_.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(mu_);
    if (started_.load(std::memory_order_relaxed)) return;
    const char* e = getenv("ACT_NUMA");
    const bool pin = e && atoi(e) != 0;
    std::vector<int> cpus;
    if (pin) { cpu_set_t set; if (sched_getaffinity(0, sizeof(set), &set) == 0) for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &set)) cpus.push_back(c); }
    // size_ - 1 workers: the calling thread of every job is the size_-th
    for (int k = 0; k + 1 < size_; k++) {
      std::thread t([this] { loop(); });
      if (pin && !cpus.empty()) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[(size_t)(k + 1) % cpus.size()], &one); (void)pthread_setaffinity_np(t.native_handle(), sizeof(one), &one); }
      t.detach();
      created_.fetch_add(1);
    }
    started_.store(true, std::memory_order_release);
  }
  void loop() {
    for (;;) {
      std::shared_ptr<Job> job;
      { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [this] { return !tickets_.empty(); }); job = std::move(tickets_.front()); tickets_.pop_front(); }
      job->work();
    }
  }
  const int size_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Job>> tickets_;
  std::atomic<bool> started_{false};
  std::atomic<uint64_t> jobs_{0}, created_{0};
};

std::atomic<int> g_hashing{0};      // calls inside act_host_hash_many at this moment

}  // namespace

extern "C" {

int act_host_usable_cpus(void) { return Pool::get().size(); }

// xof[i*16 .. i*16+16) = first 64 XOF bytes of BLAKE3(msgs + i*stride, len) for i < n.  max_threads: 0 = this call's fair share
// of the pool (pool size / calls hashing right now), k > 0 = at most k (and never more than the fair share).
void act_host_hash_many(const uint8_t* msgs, size_t stride, uint32_t len, size_t n, int max_threads, uint32_t* xof) {
  if (!n) return;
  Pool& pool = Pool::get();
  const int active = g_hashing.fetch_add(1, std::memory_order_acq_rel) + 1;
  int par = std::max(1, pool.size() / active);
  if (max_threads > 0) par = std::min(par, max_threads);
  HashArgs args{msgs, stride, len, xof};
  auto job = std::make_shared<Job>();
  job->fn = hash_items; job->ctx = &args; job->n = n; job->grain = GRAIN;
  pool.run(job, par);
  g_hashing.fetch_sub(1, std::memory_order_acq_rel);
}

// fn(ctx, i0, i1) over [0, n) in items of `grain` indices on the same workers (node.cpp routes nullifiers to their owners with it).
// Items run in any order and concurrently; the call returns when all have.  max_threads as above.
void act_host_parallel_for(size_t n, size_t grain, int max_threads, void (*fn)(void*, size_t, size_t), void* ctx) {
  if (!n || !fn) return;
  Pool& pool = Pool::get();
  const int active = g_hashing.fetch_add(1, std::memory_order_acq_rel) + 1;
  int par = std::max(1, pool.size() / active);
  if (max_threads > 0) par = std::min(par, max_threads);
  auto job = std::make_shared<Job>();
  job->fn = fn; job->ctx = ctx; job->n = n; job->grain = grain ? grain : 1;
  pool.run(job, par);
  g_hashing.fetch_sub(1, std::memory_order_acq_rel);
}

// test / diagnostics hook: hashing calls served so far and worker threads ever created by this process (the second number
// stops growing after the first call: no thread is created per call)
void act_host_pool_stats(uint64_t* jobs, uint64_t* threads_created, int* pool_size) {
  Pool& pool = Pool::get();
  if (jobs) *jobs = pool.jobs();
  if (threads_created) *threads_created = pool.threads_created();
  if (pool_size) *pool_size = pool.size();
}

}  // extern "C"
