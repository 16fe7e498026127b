// host-only check of csrc/workers.h (the prover's pool of persistent worker threads): every submitted task runs exactly once and
// can be waited for, nested submissions from inside a task work (the uploader submits its staging lanes), the pool re-uses parked
// workers instead of growing, and beyond its cap submit() refuses — never blocks, never throws — so that the caller runs the task inline.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <thread>
#include <vector>

#include "workers.h"

using namespace isnark;

extern "C" int workers_check()
{
  WorkerPool& P = WorkerPool::get();
  // 1. many short tasks, waited for in reverse order
  {
    std::atomic<int> ran{0};
    std::vector<std::unique_ptr<HostTask>> ts;
    for (int i = 0; i < 64; i++) {
      ts.emplace_back(new HostTask());
      ts.back()->fn = [&ran] {
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        ran.fetch_add(1);
      };
      P.run_or_inline(ts.back().get());
    }
    for (int i = 63; i >= 0; i--)
      if (ts[i]->queued) WorkerPool::wait(ts[i].get());
    if (ran.load() != 64) return 1;
  }
  // 2. nested: a task submits four more and waits for them (the uploader and its lanes)
  {
    std::atomic<int> inner{0};
    HostTask outer;
    outer.fn = [&] {
      HostTask lane[4];
      for (auto& l : lane) {
        l.fn = [&inner] { inner.fetch_add(1); };
        WorkerPool::get().run_or_inline(&l);
      }
      for (auto& l : lane)
        if (l.queued) WorkerPool::wait(&l);
    };
    P.run_or_inline(&outer);
    if (outer.queued) WorkerPool::wait(&outer);
    if (inner.load() != 4) return 2;
  }
  // 3. a task that throws is contained (nothing may unwind into the C ABI) and still counts as done
  {
    HostTask t;
    t.fn = [] { throw 1; };
    P.run_or_inline(&t);
    if (t.queued) WorkerPool::wait(&t);
    if (!t.done) return 3;
  }
  // 4. beyond the cap: 200 tasks that all block until released → the pool takes what it can (≤ its cap), refuses the rest at once
  {
    std::atomic<bool> release{false};
    std::atomic<int> ran{0};
    std::vector<std::unique_ptr<HostTask>> ts;
    int refused = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; i++) {
      ts.emplace_back(new HostTask());
      ts.back()->fn = [&] {
        while (!release.load()) std::this_thread::sleep_for(std::chrono::microseconds(100));
        ran.fetch_add(1);
      };
      if (!P.submit(ts.back().get())) refused++;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    release.store(true);
    int queued = 0;
    for (auto& t : ts)
      if (t->queued) {
        WorkerPool::wait(t.get());
        queued++;
      }
    if (queued + refused != 200 || ran.load() != queued) return 4;
    if (refused < 200 - 96 || queued < 32) return 5; // (cap 96; a container's thread limit may make the pool stop earlier)
    if (ms > 2000) return 6;                          // submit never blocked
  }
  // 5. afterwards the parked workers are re-used
  {
    std::atomic<int> ran{0};
    HostTask t[8];
    for (auto& x : t) {
      x.fn = [&ran] { ran.fetch_add(1); };
      if (!P.submit(&x)) return 7;
    }
    for (auto& x : t) WorkerPool::wait(&x);
    if (ran.load() != 8) return 8;
  }
  return 0;
}
