// team.h — a team of persistent host threads: n − 1 parked threads + the caller run one function per member (run), and
// barrier() inside that function lines the members up.  Used by the device group of the prover (one member per shard,
// prover/multi.cpp).  Waiting is blocking (condition variables): members that spin would eat the CPU quota the staging workers
// and the Horner tails of a prove need.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace isnark {

class Team {
public:
  explicit Team(int n) : n_(n)
  {
    for (int r = 1; r < n; r++) th_.emplace_back([this, r] { loop(r); });
  }
  ~Team()
  {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_job_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(const std::function<void(int)>& fn)
  {
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &fn;
      pending_ = n_ - 1;
      gen_++;
    }
    cv_job_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }
  void barrier()
  {
    std::unique_lock<std::mutex> lk(bmu_);
    const uint64_t g = bgen_;
    if (++bcount_ == n_) {
      bcount_ = 0;
      bgen_++;
      bcv_.notify_all();
    } else
      bcv_.wait(lk, [this, g] { return bgen_ != g; });
  }

private:
  void loop(int r)
  {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* fn;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_job_.wait(lk, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_;
        fn = job_;
      }
      (*fn)(r);
      {
        std::lock_guard<std::mutex> lk(mu_);
        pending_--;
      }
      cv_done_.notify_all();
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex mu_, bmu_;
  std::condition_variable cv_job_, cv_done_, bcv_;
  const std::function<void(int)>* job_ = nullptr;
  uint64_t gen_ = 0, bgen_ = 0;
  int pending_ = 0, bcount_ = 0;
  bool quit_ = false;
};

} // namespace isnark
