// host-only check of csrc/prover/cold_feed.h (the stage feed between the uploader task of a cold prove and the prove's thread):
// a waiter returns 0 exactly when ITS stage has been posted, however the posts and waits interleave; after a failure every
// waiter — also of stages never posted — returns the uploader's code and text; a feed that finishes without posting a stage
// does not leave its waiter blocked.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "prover/cold_feed.h"

using isnark::prover::ColdFeed;

extern "C" int coldfeed_check()
{
  // 1. stages posted in the uploader's order (COEF, WITNESS, B2, A, B1, C, H) with small delays; one waiter per stage started
  //    in another order: each returns 0 and never before its stage was posted
  for (int round = 0; round < 20; round++) {
    ColdFeed F;
    std::atomic<int> posted_mask{0};
    std::atomic<int> bad{0};
    std::vector<std::thread> ws;
    const int order[ColdFeed::N] = {ColdFeed::SEC_H, ColdFeed::WITNESS, ColdFeed::SEC_A, ColdFeed::COEF, ColdFeed::SEC_C, ColdFeed::SEC_B2, ColdFeed::SEC_B1};
    for (int i : order)
      ws.emplace_back([&, i] {
        const int rc = F.wait(i);
        if (rc != 0 || !(posted_mask.load() & (1 << i))) bad.fetch_add(1);
      });
    const int up[ColdFeed::N] = {ColdFeed::COEF, ColdFeed::WITNESS, ColdFeed::SEC_B2, ColdFeed::SEC_A, ColdFeed::SEC_B1, ColdFeed::SEC_C, ColdFeed::SEC_H};
    for (int i : up) {
      if (round & 1) std::this_thread::sleep_for(std::chrono::microseconds(50 * (round % 5)));
      posted_mask.fetch_or(1 << i);
      F.post(i);
    }
    F.finish();
    for (auto& t : ws) t.join();
    if (bad.load()) return 1;
    for (int i = 0; i < ColdFeed::N; i++)
      if (F.wait(i) != 0) return 2; // (a wait after the end still answers)
  }
  // 2. failure after two stages: their waiters got 0, everybody else the code; the text is the first failure's
  {
    ColdFeed F;
    int rcs[ColdFeed::N];
    std::vector<std::thread> ws;
    for (int i = 0; i < ColdFeed::N; i++) ws.emplace_back([&, i] { rcs[i] = F.wait(i); });
    F.post(ColdFeed::COEF);
    F.post(ColdFeed::WITNESS);
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
    F.fail_with(-2, "coefficient 3 out of range");
    F.fail_with(-9, "later noise");
    F.finish();
    for (auto& t : ws) t.join();
    for (int i = 0; i < ColdFeed::N; i++) {
      const bool was_posted = i == ColdFeed::COEF || i == ColdFeed::WITNESS;
      // a waiter of a posted stage may have been woken by the failure first: 0 or the code are both legal for it — but a stage that
      // never arrived must never report 0
      if (!was_posted && rcs[i] != -2) return 3;
      if (was_posted && rcs[i] != 0 && rcs[i] != -2) return 4;
    }
    if (F.rc != -2 || F.err != "coefficient 3 out of range") return 5;
    if (F.wait(ColdFeed::COEF) != 0 || F.wait(ColdFeed::SEC_H) != -2) return 6;
  }
  // 3. finished without the stage and without an error text: the waiter comes back with an error, not 0
  {
    ColdFeed F;
    int rc = 0;
    std::thread w([&] { rc = F.wait(ColdFeed::SEC_C); });
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
    F.finish();
    w.join();
    if (rc == 0) return 7;
  }
  return 0;
}
