// The stage feed of the cold pipeline (prover_internal.h: ColdUpload; prover.cpp: cold_prove) — on its own so that the CPU suite can
// drive it without a GPU (tests/test_workers.py: coldfeed_check.cc).
#pragma once
#include <condition_variable>
#include <mutex>
#include <string>

#include <hip/hip_runtime_api.h>

#include "../../../include/icicle_snark_hip.h"

namespace isnark {
namespace prover {
// Cold pipeline (round 5): inside a groth16_prove that finds no cache entry the key's sections and the witness cross PCIe WHILE the
// prove's kernels are being enqueued and run — upload (16–20 ms at 1.6 M constraints) and first proof (≈ 22 ms of GPU work) overlap
// instead of following each other.  An uploader task (a pooled worker, cache.cpp: cold_upload_task) sends, in this order, the
// coefficient records (→ CSR built on the device), the witness, and the point sections B2, A, B1, C, H (each converted to the bucket
// kernels' encoding as it lands), records an event behind each stage and posts it; the prove's thread waits (host) until a stage
// has been POSTED — an event that has not been recorded yet would not make a stream wait — and then makes the stream that needs
// the stage wait for its event.
struct ColdFeed {
  enum { COEF = 0, WITNESS = 1, SEC_A = 2, SEC_B1 = 3, SEC_B2 = 4, SEC_C = 5, SEC_H = 6, N = 7 };
  std::mutex m;
  std::condition_variable cv;
  bool posted[N] = {false, false, false, false, false, false, false};
  hipEvent_t ev[N] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int rc = 0;       // first error of the uploader (then nothing more is posted)
  std::string err;
  bool finished = false;
  void post(int i)
  {
    std::lock_guard<std::mutex> lk(m);
    posted[i] = true;
    cv.notify_all();
  }
  void fail_with(int code, const char* text)
  {
    std::lock_guard<std::mutex> lk(m);
    if (!rc) {
      rc = code;
      err = text ? text : "";
    }
    cv.notify_all();
  }
  void finish()
  {
    std::lock_guard<std::mutex> lk(m);
    finished = true;
    cv.notify_all();
  }
  // 0 once stage i has been posted; the uploader's error code when it failed first
  int wait(int i)
  {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return posted[i] || rc != 0 || finished; });
    return posted[i] ? 0 : (rc ? rc : (int)ICICLE_UNKNOWN_ERROR);
  }
};

} // namespace prover
} // namespace isnark
