// coalesce.h — callers of one context that arrive while a call is running merge into the next call instead of queueing behind it
// (engine.hip spend_coalesced: the crate's entry points take ONE proof, /root/reference/src/lib.rs:781-786, and a server's threads
// share one context).  Plain C++ (no HIP): the queueing protocol is exercised under ThreadSanitizer in tests/tsan/tsan_host.cpp.
//
// Protocol.  A caller appends its request and then either finds a leader at work -- it sleeps until its request is done or the
// leadership is free -- or becomes the leader: it repeatedly takes the OLDEST queued request and every queued request of the same
// group (in arrival order, while the merged size stays within `cap`), runs them as one call outside the lock, marks them done and
// wakes everybody; when its own request is done it gives the leadership up, and one of the callers still waiting takes over.
// Nobody waits for company (a lone caller's request runs at once); requests pile up only while a call is running; requests are
// served oldest first, so nobody is overtaken for ever.
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
#include <vector>

namespace act {

// Req needs: size_t n (its size in lanes), int rc (result code of the call that served it), bool done.
template <class Req>
struct Combiner {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Req*> q;          // requests no merged call has taken yet, oldest first
  bool leader = false;         // some caller is running merged calls

  // same(a, b): may a and b share a call?   run(batch, total_lanes) -> rc of the merged call (stored in every request of the batch)
  template <class Same, class Run>
  int submit(Req& r, size_t cap, Same same, Run run) {
    std::unique_lock<std::mutex> lk(mu);
    q.push_back(&r);
    for (;;) {
      if (r.done) return r.rc;                              // a leader ran it
      if (!leader) break;                                   // nobody is leading: this caller does
      cv.wait(lk);
    }
    leader = true;
    while (!r.done) {
      std::vector<Req*> batch; size_t total = 0;
      Req* first = q.front();
      for (auto it = q.begin(); it != q.end();) {
        Req* x = *it;
        if (same(*first, *x) && (batch.empty() || total + x->n <= cap)) { batch.push_back(x); total += x->n; it = q.erase(it); }
        else ++it;
      }
      lk.unlock();
      const int rc = run(batch, total);
      lk.lock();
      for (Req* x : batch) { if (rc) x->rc = rc; x->done = true; }
      cv.notify_all();
    }
    leader = false;                                         // this caller's own request is done: whoever still waits takes over
    cv.notify_all();
    return r.rc;
  }
};

}  // namespace act
