// workers.h — the library's pool of persistent host worker threads.
//
// A prove needs helper threads for milliseconds at a time: the staging lanes of the witness upload, the uploader that drives
// them while the caller enqueues the head of the witness MSMs, one Horner tail per MSM, the blinding terms.  Rounds 1–4 created a
// std::thread for each of them in every prove (eight constructions, ≈ 0.2 ms of the caller's time, three of them in front of the
// first DMA) — and a std::thread constructor can throw under a container's thread limit, which must not unwind through an
// extern "C" entry point.  Here the threads are created on demand, parked on a condition variable and re-used; submit() never
// throws and never blocks: when no worker can be had (thread limit, pool at its cap) it says so and the caller runs the task inline.
//
// The pool is a leaked singleton with detached threads: idle workers sit in a futex wait and simply end with the process — no
// static destructor has to join them while the HIP runtime is shutting down.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace isnark {

struct HostTask {
  std::function<void()> fn;
  std::mutex m;
  std::condition_variable c;
  bool done = false;
  bool queued = false; // a worker owns it (else the submitter ran — or has to run — it inline)
};

class WorkerPool {
public:
  static WorkerPool& get()
  {
    static WorkerPool* p = new WorkerPool();
    return *p;
  }
  // hands the task to a parked (or a new) worker; false: no worker to be had — run it yourself (run_or_inline does)
  bool submit(HostTask* t) noexcept
  {
    try {
      std::lock_guard<std::mutex> lk(mu_);
      if (idle_ <= (int)q_.size()) {
        if (nthreads_ >= CAP) return false;
        std::thread th([this] { loop(); });
        th.detach();
        nthreads_++;
      }
      t->queued = true;
      q_.push_back(t);
    } catch (...) {
      return false;
    }
    cv_.notify_one();
    return true;
  }
  void run_or_inline(HostTask* t) noexcept
  {
    if (submit(t)) return;
    run(t);
  }
  // blocks until the task has run
  static void wait(HostTask* t)
  {
    std::unique_lock<std::mutex> lk(t->m);
    t->c.wait(lk, [t] { return t->done; });
  }

private:
  static constexpr int CAP = 96; // 8 tasks per prove × the shards of a device group, with room to spare
  static void run(HostTask* t) noexcept
  {
    try {
      t->fn();
    } catch (...) {
    }
    // notified under the lock: the waiter cannot return (and destroy the task) before this thread has let go of it
    std::lock_guard<std::mutex> lk(t->m);
    t->done = true;
    t->c.notify_all();
  }
  void loop()
  {
    for (;;) {
      HostTask* t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        idle_++;
        cv_.wait(lk, [this] { return !q_.empty(); });
        idle_--;
        t = q_.front();
        q_.pop_front();
      }
      run(t);
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<HostTask*> q_;
  int idle_ = 0, nthreads_ = 0;
};

} // namespace isnark
